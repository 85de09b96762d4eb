#!/bin/bash
# GPU box: wall time of build/testbed --save-mesh at several resolutions on a small synthetic scene
python - <<'PY'
import sys; sys.path.insert(0, ".")
from rnb_neus2_amd import synthetic
v, n, a = synthetic.make_scene(16, 256, 448.0)
synthetic.write_scene("/tmp/mesh_scene", v, n, a, scale=0.5, offset=(0.5, 0.5, 0.5))
PY
for R in 256 512 1024; do
  t0=$(date +%s%N)
  ./build/testbed --scene /tmp/mesh_scene/ --maxiter 300 --no-gui --no-albedo --save-mesh --resolution $R > /tmp/mesh_$R.log 2>/tmp/mesh_$R.err
  t1=$(date +%s%N)
  echo "res $R: $(( (t1 - t0) / 1000000 )) ms wall (training 300 steps included)"; grep "vertices" /tmp/mesh_$R.log; tail -2 /tmp/mesh_$R.err; ls -la /tmp/mesh_scene/output/*.obj | tail -1
done

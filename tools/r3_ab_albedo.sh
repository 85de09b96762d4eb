#!/bin/bash
# GPU box: bench.py --albedo under a list of environment settings -> gpurun_out/<tag>/ab_albedo.txt
tag=$1; shift
mkdir -p gpurun_out/$tag
for env in "$@"; do
  for rep in 1 2; do
    out=$(env $env timeout 120 python bench.py --albedo --steps 200 --no-cpu-baseline --profile-steps 0 --late-step 0 2>/dev/null | grep "^{")
    python - "$env" "$out" <<'PY' >> gpurun_out/$tag/ab_albedo.txt
import json, sys
d = json.loads(sys.argv[2])
print("%-50s albedo steps1000-1200 %.4f ms  window %.4f ms (p50 %.4f)" % (sys.argv[1] or "(defaults)", d["ms_per_step"], d["window_1000_2000"]["ms_per_step"], d["window_1000_2000"]["p50_ms_per_step"]))
PY
  done
done
cat gpurun_out/$tag/ab_albedo.txt

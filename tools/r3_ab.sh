#!/bin/bash
# GPU box: bench.py --steps 20 under a list of environment settings (A/B on one box) -> gpurun_out/<tag>/ab.txt
#   bash tools/r3_ab.sh <tag> "NAME=V NAME2=V" "..." ...     ("" = defaults)
tag=$1; shift
mkdir -p gpurun_out/$tag
for env in "$@"; do
  for rep in 1 2; do
    out=$(env $env python bench.py --steps 200 --no-cpu-baseline --profile-steps 0 --late-steps 100 --fixed-cost-steps 0 2>/dev/null | grep "^{")
    python - "$env" "$out" <<'PY' >> gpurun_out/$tag/ab.txt
import json, sys
d = json.loads(sys.argv[2])
print("%-50s steps1000-1200 %.4f ms  window %.4f ms (p50 %.4f)  late %.4f ms (p50 %.4f)" % (sys.argv[1] or "(defaults)", d["ms_per_step"], d["window_1000_2000"]["ms_per_step"], d["window_1000_2000"]["p50_ms_per_step"], d["late_regime"]["ms_per_step"], d["late_regime"]["p50_ms_per_step"]))
PY
  done
done
cat gpurun_out/$tag/ab.txt

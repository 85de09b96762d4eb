#!/bin/bash
# GPU box: serialised per-kernel HIP-event times (bench.py's profile pass) at one point of the run, for several environment settings.
#   BURN=980 bash tools/serial_kernels.sh "ENV" "ENV" ...
burn=${BURN:-980}
for env in "$@"; do
  out=$(env $env python bench.py --burn-in $burn --steps 20 --warmup 20 --no-cpu-baseline --profile-steps 100 --window-end 0 --late-step 0 --fixed-cost-steps 0 --parity-mode-steps 0 --no-live-pmc $BENCH_ARGS 2>/dev/null | grep "^{")
  python - "$env" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
k = d["kernels_ms_per_step"]
print("%-40s step %.4f | " % (sys.argv[1] or "(defaults)", d["ms_per_step"]) + "  ".join("%s %.1f" % (n.replace("k_grid_scatter", "sc").replace("k_", ""), 1e3 * v["ms_per_step"]) for n, v in k.items() if v["ms_per_step"] > 0.004))
PY
done

#!/bin/bash
# GPU box: A/B of environment settings on ONE slice of the run (default: 200 steps from step 1000), interleaved rounds -> gpurun_out/<tag>/ab_slice.txt
#   BURN=980 STEPS=200 bash tools/ab_slice.sh <tag> <rounds> "ENV" "ENV" ...
tag=$1; rounds=$2; shift; shift
burn=${BURN:-980}; steps=${STEPS:-200}
mkdir -p gpurun_out/$tag; rm -f gpurun_out/$tag/ab_slice_raw.txt
for r in $(seq 1 $rounds); do
  for env in "$@"; do
    out=$(env $env python bench.py --burn-in $burn --steps $steps --warmup 20 --no-cpu-baseline --profile-steps 0 --window-end 0 --late-step 0 --fixed-cost-steps 0 --parity-mode-steps 0 --no-live-pmc $BENCH_ARGS 2>/dev/null | grep "^{")
    python - "$env" "$out" <<'PY' >> gpurun_out/$tag/ab_slice_raw.txt
import json, sys
d = json.loads(sys.argv[2])
print(json.dumps({"env": sys.argv[1] or "(defaults)", "ms": d["ms_per_step"], "rays": d["config"]["rays_per_step_per_gpu"]}))
PY
  done
done
python - gpurun_out/$tag/ab_slice_raw.txt $burn $steps <<'PY' | tee gpurun_out/$tag/ab_slice.txt
import json, sys, statistics as st
rows = [json.loads(l) for l in open(sys.argv[1])]
envs = []
for r in rows:
    if r["env"] not in envs: envs.append(r["env"])
print("ms/step over %s steps from step %d, median [min .. max] of %d interleaved runs" % (sys.argv[3], int(sys.argv[2]) + 20, len(rows) // len(envs)))
for e in envs:
    g = [r["ms"] for r in rows if r["env"] == e]
    print("%-60s %.4f  [%.4f .. %.4f]" % (e, st.median(g), min(g), max(g)))
PY

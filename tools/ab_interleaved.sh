#!/bin/bash
# GPU box: A/B of environment settings, interleaved over several rounds on ONE box (box-to-box spread of the pool is larger than most effects) -> gpurun_out/<tag>/ab.txt
#   bash tools/ab_interleaved.sh <tag> <rounds> "NAME=V NAME2=V" "..." ...     ("" = defaults). BENCH_ARGS: extra bench.py arguments.
tag=$1; rounds=$2; shift; shift
mkdir -p gpurun_out/$tag
rm -f gpurun_out/$tag/ab_raw.txt
for r in $(seq 1 $rounds); do
  for env in "$@"; do
    out=$(env $env python bench.py --steps 200 --no-cpu-baseline --profile-steps 0 --late-steps 100 --fixed-cost-steps 0 --parity-mode-steps 0 --no-live-pmc $BENCH_ARGS 2>/dev/null | grep "^{")
    python - "$env" "$out" <<'PY' >> gpurun_out/$tag/ab_raw.txt
import json, sys
d = json.loads(sys.argv[2])
print(json.dumps({"env": sys.argv[1] or "(defaults)", "steps": d["ms_per_step"], "window": d["window_1000_2000"]["ms_per_step"], "p50": d["window_1000_2000"]["p50_ms_per_step"], "late": d["late_regime"]["ms_per_step"], "late_p50": d["late_regime"]["p50_ms_per_step"]}))
PY
  done
done
python - gpurun_out/$tag/ab_raw.txt <<'PY' | tee gpurun_out/$tag/ab.txt
import json, sys, statistics as st
rows = [json.loads(l) for l in open(sys.argv[1])]
envs = []
for r in rows:
    if r["env"] not in envs: envs.append(r["env"])
print("median of %d interleaved runs per setting, ms/step: steps 1000-1200 | window 1000-2000 (p50) | steps 6000-6100 (p50)   [min .. max of the window]" % (len(rows) // len(envs)))
for e in envs:
    g = [r for r in rows if r["env"] == e]
    m = lambda k: st.median(x[k] for x in g)
    print("%-52s %.4f | %.4f (%.4f) | %.4f (%.4f)   [%.4f .. %.4f]" % (e, m("steps"), m("window"), m("p50"), m("late"), m("late_p50"), min(x["window"] for x in g), max(x["window"] for x in g)))
PY

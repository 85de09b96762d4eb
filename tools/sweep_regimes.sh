#!/bin/bash
# GPU box: ms/step of 160-step slices along the training run for a list of environment settings, interleaved -> gpurun_out/<tag>/sweep.txt
#   bash tools/sweep_regimes.sh <tag> "ENV..." "ENV..." ...       slices start at steps 1000 1200 1400 1600 1800 2400 3200 4400 6000
tag=$1; shift
mkdir -p gpurun_out/$tag; rm -f gpurun_out/$tag/sweep_raw.txt
for burn in 980 1180 1380 1580 1780 2380 3180 4380 5980; do
  for rep in 1 2; do
  for env in "$@"; do
    out=$(env $env python bench.py --burn-in $burn --steps 160 --warmup 20 --no-cpu-baseline --profile-steps 0 --window-end 0 --late-step 0 --fixed-cost-steps 0 2>/dev/null | grep "^{")
    python - "$env" "$burn" "$out" <<'PY' >> gpurun_out/$tag/sweep_raw.txt
import json, sys
d = json.loads(sys.argv[3])
print(json.dumps({"env": sys.argv[1] or "(defaults)", "first_step": int(sys.argv[2]) + 20, "ms": d["ms_per_step"], "rays": d["config"]["rays_per_step_per_gpu"], "rays_per_s": d["value"]}))
PY
  done; done
done
python - gpurun_out/$tag/sweep_raw.txt <<'PY' | tee gpurun_out/$tag/sweep.txt
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1])]
envs, steps = [], []
for r in rows:
    if r["env"] not in envs: envs.append(r["env"])
    if r["first_step"] not in steps: steps.append(r["first_step"])
print("ms/step over 160 steps from the given step (mean of 2 runs); rays per step in brackets")
print("%-10s" % "step" + "".join("%-34s" % e[:32] for e in envs))
for s in steps:
    line = "%-10d" % s
    for e in envs:
        g = [r for r in rows if r["env"] == e and r["first_step"] == s]
        line += "%-34s" % ("%.4f  [%.0f]" % (sum(x["ms"] for x in g) / len(g), g[0]["rays"]))
    print(line)
PY

#!/bin/bash
# GPU box: rocprofv3 kernel trace of the steady regime, serial and overlapped schedule -> gpurun_out/<tag>/steady_{serial,overlapped}.json
#   bash tools/kt.sh <tag> [burn-in (default 980)] [steps (default 100)]
tag=${1:-kt}; burn=${2:-980}; steps=${3:-100}
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for mode in serial overlapped; do
  rm -rf /tmp/kt_$mode
  if [ $mode = serial ]; then export RNB_OVERLAP_OFF=1; else unset RNB_OVERLAP_OFF; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$mode -- python bench.py --burn-in $burn --steps $steps --warmup 20 --no-cpu-baseline --profile-steps 0 --window-end 0 --late-step 0 --fixed-cost-steps 0 --parity-mode-steps 0 --no-live-pmc > /tmp/kt_$mode.log 2>&1
  f=$(find /tmp/kt_$mode -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/$tag/kernel_stats_$mode.csv
  python tools/steady_stats.py /tmp/kt_$mode $steps > gpurun_out/$tag/steady_$mode.json
  grep "^{" /tmp/kt_$mode.log | head -c 400; echo
done
unset RNB_OVERLAP_OFF
python - <<PY
import json
for m in ('serial','overlapped'):
    d=json.load(open('gpurun_out/$tag/steady_%s.json'%m)); print(m, 'wall us/step', d['wall_us_per_step'])
    for k,v in list(d['kernels'].items())[:24]: print('   %-44s %8.2f us/step  (%.2f x %.2f)'%(k[:44], v['us_per_step'], v['calls_per_step'], v['avg_us']))
PY

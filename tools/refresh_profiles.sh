#!/bin/bash
# GPU box: everything profiles/ holds for a round (bench lines, serial / overlapped kernel statistics, PMC passes, smoke) -> gpurun_out/r01, gpurun_out/pmc
mkdir -p gpurun_out/r01 gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python bench.py 2>/dev/null | grep "^{" > gpurun_out/r01/bench_default.json
python bench.py --albedo --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/r01/bench_albedo.json
for mode in serial overlapped; do
  rm -rf /tmp/kt_$mode
  if [ $mode = serial ]; then export RNB_OVERLAP_OFF=1; else unset RNB_OVERLAP_OFF; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$mode -- python bench.py --burn-in 1980 --steps 100 --warmup 20 --no-cpu-baseline --profile-steps 0 > /tmp/kt_$mode.log 2>&1
  f=$(find /tmp/kt_$mode -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r01/kernel_stats_$mode.csv
  python tools/steady_stats.py /tmp/kt_$mode 100 > gpurun_out/r01/steady_$mode.json
done
unset RNB_OVERLAP_OFF
bash tools/collect_pmc.sh gpurun_out/pmc FETCH_SIZE WRITE_SIZE TCC_ATOMIC_sum SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE > /dev/null 2>&1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
head -c 600 gpurun_out/r01/bench_default.json; echo; head -c 300 gpurun_out/r01/bench_albedo.json; echo
python -c "
import json
for m in ('serial','overlapped'):
    d=json.load(open('gpurun_out/r01/steady_%s.json'%m)); print(m, d['wall_us_per_step'])"

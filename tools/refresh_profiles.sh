#!/bin/bash
# GPU box: everything profiles/ holds for a round -> gpurun_out/<round>p/ (copy what is to be judged into profiles/ afterwards)
#   bash tools/refresh_profiles.sh r02
R=${1:-r05}
O=gpurun_out/${R}p
mkdir -p $O $O/pmc
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python bench.py 2>$O/bench_default.err | grep "^{" > $O/bench_default.json                        # the defaults: steps 1000-2000 timed, + late regime
python bench.py --steps 20 2>/dev/null | grep "^{" > $O/bench_steps20.json                        # the driver's command line
python bench.py --albedo --no-cpu-baseline --late-step 0 --parity-mode-steps 0 2>/dev/null | grep "^{" > $O/bench_albedo.json
for regime in 980 1980; do
for mode in serial overlapped; do
  rm -rf /tmp/kt_$mode
  if [ $mode = serial ]; then export RNB_OVERLAP_OFF=1; else unset RNB_OVERLAP_OFF; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$mode -- python bench.py --burn-in $regime --steps 100 --warmup 20 --no-cpu-baseline --profile-steps 0 --window-end 0 --late-step 0 --fixed-cost-steps 0 --parity-mode-steps 0 > /tmp/kt_$mode.log 2>&1
  f=$(find /tmp/kt_$mode -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_${mode}_step$((regime+20)).csv
  python tools/steady_stats.py /tmp/kt_$mode 100 > $O/steady_${mode}_step$((regime+20)).json
done; done
unset RNB_OVERLAP_OFF
bash tools/collect_pmc.sh $O/pmc FETCH_SIZE WRITE_SIZE TCC_ATOMIC_sum SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE > /dev/null 2>&1
python tools/pmc_traffic.py $O/pmc $O/pmc_traffic.json
python tools/pmc_report.py $O/pmc $O/steady_serial_step2000.json $O/pmc_units.json > /dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python - <<PY
import json
d=json.load(open('$O/bench_default.json')); print('default:', d['value'], d['ms_per_step'], d['window_1000_2000'], d['late_regime'])
d=json.load(open('$O/bench_steps20.json')); print('steps20:', d['value'], d['ms_per_step'], d['late_regime'])
for m in ('serial_step1000','overlapped_step1000','serial_step2000','overlapped_step2000'):
    s=json.load(open('$O/steady_%s.json'%m)); print(m, s['wall_us_per_step'])
PY
# late regime (steps 6000-6100) and the SQ / TCP counter groups of the window regime
for mode in serial overlapped; do
  rm -rf /tmp/kt_$mode
  if [ $mode = serial ]; then export RNB_OVERLAP_OFF=1; else unset RNB_OVERLAP_OFF; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$mode -- python bench.py --burn-in 5980 --steps 100 --warmup 20 --no-cpu-baseline --profile-steps 0 --window-end 0 --late-step 0 --fixed-cost-steps 0 --parity-mode-steps 0 > /tmp/kt_$mode.log 2>&1
  python tools/steady_stats.py /tmp/kt_$mode 100 > $O/steady_${mode}_step6000.json
done
unset RNB_OVERLAP_OFF
python tools/timeline.py /tmp/kt_overlapped 5 > $O/timeline_overlapped_step6000.txt
rm -rf /tmp/kt_tl; rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_tl -- python bench.py --burn-in 980 --steps 60 --warmup 20 --no-cpu-baseline --profile-steps 0 --window-end 0 --late-step 0 --fixed-cost-steps 0 --parity-mode-steps 0 > /tmp/kt_tl.log 2>&1
python tools/timeline.py /tmp/kt_tl 5 > $O/timeline_overlapped_step1000.txt
tools/collect_pmc_group.sh $O/sq a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES > /dev/null 2>&1
tools/collect_pmc_group.sh $O/sq b SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM > /dev/null 2>&1
tools/collect_pmc_group.sh $O/sq c TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum > /dev/null 2>&1
python - <<PY
import json, glob
tab = {}
for f in sorted(glob.glob('$O/sq/*.json')):
    d = json.load(open(f))
    for k, v in d['kernels'].items(): tab.setdefault(k, {})[d['counter']] = round(v['avg'])
json.dump({"_source": "tools/collect_pmc_group.sh: three rocprofv3 --pmc passes (SQ x8, SQ x7, TCP x3), kernels serialised, steps 2000-2010, average per launch; SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles summed over wavefronts", "kernels": tab}, open('$O/pmc_sq.json', 'w'), indent=1)
PY

#!/bin/bash
# GPU box: per-kernel hardware counters, ONE counter per rocprofv3 --pmc pass (as the microarchitecture guide prescribes for
# the HBM counters), kernels serialised (cfg.overlap = 0) -> gpurun_out/pmc/<COUNTER>.json
#   tools/collect_pmc.sh [outdir] [COUNTER ...]        default counters: FETCH_SIZE WRITE_SIZE
export TMPDIR=/tmp
OUT=${1:-gpurun_out/pmc}
shift
COUNTERS=${@:-FETCH_SIZE WRITE_SIZE}
mkdir -p $OUT
for C in $COUNTERS; do
  rm -rf /tmp/pmc_$C
  RNB_OVERLAP_OFF=1 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -- python bench.py --steps 10 --warmup 2 --burn-in 2000 --profile-steps 0 --no-cpu-baseline --window-end 0 --late-step 0 --fixed-cost-steps 0 --parity-mode-steps 0 --no-live-pmc > /tmp/pmc_$C.log 2>&1
  python tools/pmc_summary.py /tmp/pmc_$C $C > $OUT/$C.json
  head -12 $OUT/$C.json
done

#!/bin/bash
# GPU box: HBM traffic counters per kernel (separate --pmc passes, as the microarchitecture guide prescribes) -> gpurun_out/pmc/
export TMPDIR=/tmp
OUT=${1:-gpurun_out/pmc}
mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  RNB_OVERLAP_OFF=1 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -- python bench.py --steps 10 --warmup 2 --burn-in 1000 --profile-steps 0 --no-cpu-baseline > /tmp/pmc_$C.log 2>&1
  python tools/pmc_summary.py /tmp/pmc_$C $C > $OUT/$C.json
  tail -3 $OUT/$C.json
done

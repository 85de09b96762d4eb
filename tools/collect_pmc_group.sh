#!/bin/bash
# GPU box: several counters of ONE hardware block in one rocprofv3 --pmc pass (SQ has 8 slots), kernels serialised
# (cfg.overlap = 0) -> <outdir>/<COUNTER>.json per counter.   tools/collect_pmc_group.sh <outdir> <tag> COUNTER [COUNTER ...]
export TMPDIR=/tmp
OUT=$1; TAG=$2; shift; shift
mkdir -p $OUT
rm -rf /tmp/pmcg_$TAG
RNB_OVERLAP_OFF=1 rocprofv3 --pmc $@ --kernel-trace --output-format csv -d /tmp/pmcg_$TAG -- python bench.py --steps 10 --warmup 2 --burn-in ${BURN:-2000} --profile-steps 0 --no-cpu-baseline --window-end 0 --late-step 0 --fixed-cost-steps 0 --parity-mode-steps 0 --no-live-pmc > /tmp/pmcg_$TAG.log 2>&1
tail -3 /tmp/pmcg_$TAG.log | cut -c1-300
for C in $@; do python tools/pmc_summary.py /tmp/pmcg_$TAG $C > $OUT/$C.json; done

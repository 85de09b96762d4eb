#!/bin/bash
# GPU box: kernel timelines of an ordinary overlapped step and of a step that begins with an occupancy update -> gpurun_out/<tag>/
#   bash tools/r3_tl2.sh <tag> <burn-in (multiple of 16 minus 20)> [ENV=V ...]
tag=$1; burn=$2; shift; shift
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/kt_tl
env "$@" rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_tl -- python bench.py $BENCH_ARGS --burn-in $burn --steps 60 --warmup 20 --no-cpu-baseline --profile-steps 0 --window-end 0 --late-step 0 --fixed-cost-steps 0 > /tmp/kt_tl.log 2>&1
# steps burn+20 .. burn+80; the cycle `back` from the end runs from the second loss pass of step burn+79-back to that of step burn+80-back
for back in 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20; do
  python tools/timeline.py /tmp/kt_tl $back > /tmp/tl_$back.txt
  if grep -q k_point_query /tmp/tl_$back.txt; then cp /tmp/tl_$back.txt gpurun_out/$tag/timeline_update_step.txt; else cp /tmp/tl_$back.txt gpurun_out/$tag/timeline_ordinary_step.txt; fi
done
head -60 gpurun_out/$tag/timeline_update_step.txt

#!/bin/bash
# GPU box: kernel timeline of one overlapped step at a given burn-in -> gpurun_out/<tag>/timeline_step<N>.txt
#   bash tools/r3_tl.sh <tag> <burn-in> [ENV=V ...]
tag=$1; burn=$2; shift; shift
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/kt_tl
env "$@" rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_tl -- python bench.py --burn-in $burn --steps 60 --warmup 20 --no-cpu-baseline --profile-steps 0 --window-end 0 --late-step 0 --fixed-cost-steps 0 > /tmp/kt_tl.log 2>&1
python tools/timeline.py /tmp/kt_tl 5 > gpurun_out/$tag/timeline_step$((burn+20))$(echo "$@" | tr -d ' =_' | head -c 40).txt
cat gpurun_out/$tag/timeline_step$((burn+20))*.txt | tail -40

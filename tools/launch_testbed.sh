#!/bin/bash
# One `testbed` process per GPU of this node, exchanging counters and gradients over RCCL (rnb-neus2_amd/host/testbed_main.cpp, struct Dist).
#   tools/launch_testbed.sh <n_gpus> build/testbed --scene <dir>/ --maxiter N --no-gui ...      (RNB_WEAK_SCALING=1: W x the batch instead of 1/W per rank)
# Rank 0 writes meshes / snapshots / progress lines; the exit code is the first non-zero one. A rank that fails takes the job down: the others get ten seconds to
# notice by themselves (the staged test transport tells them; RCCL ranks would wait for the dead peer forever) and are then ended by their PIDs.
# Tests: RNB_DP_TRANSPORT=staged RNB_DP_STAGE_DIR=<dir> runs the same job with the collectives staged through host files (host/dist_transport.hpp) -- two ranks
# on one GPU (`... 2 env RNB_LOCAL_RANK=0 build/testbed ...`) or the CPU-checker build.
N=$1; shift
ID=$(mktemp -u /tmp/rnb_rccl_id.XXXXXX)
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-16}
export RNB_DP_JOB_ID=${RNB_DP_JOB_ID:-$$.$(date +%s%N)}  # the staged test transport works in <RNB_DP_STAGE_DIR>/job_<id>: nothing a crashed earlier job left behind is ever read
pids=()
for ((r = 0; r < N; r++)); do
  RNB_WORLD_SIZE=$N RNB_RANK=$r RNB_LOCAL_RANK=$r RNB_RCCL_ID_FILE=$ID "$@" &
  pids+=($!)
done
rc=0
deadline=0
while :; do
  alive=0
  for i in "${!pids[@]}"; do
    p=${pids[$i]}
    [ -z "$p" ] && continue
    if kill -0 "$p" 2>/dev/null; then alive=1; continue; fi
    wait "$p"; c=$?
    pids[$i]=""
    if [ $c -ne 0 ] && [ $rc -eq 0 ]; then rc=$c; deadline=$((SECONDS + 10)); fi
  done
  [ $alive -eq 0 ] && break
  if [ $deadline -ne 0 ] && [ $SECONDS -ge $deadline ]; then
    for q in "${pids[@]}"; do [ -n "$q" ] && kill "$q" 2>/dev/null; done
    deadline=$((SECONDS + 1000000))
  fi
  sleep 0.1
done
rm -f $ID $ID.tmp
exit $rc

#!/bin/bash
# One `testbed` process per GPU of this node, exchanging counters and gradients over RCCL (rnb-neus2_amd/host/testbed_main.cpp, struct Dist).
#   tools/launch_testbed.sh <n_gpus> build/testbed --scene <dir>/ --maxiter N --no-gui ...      (RNB_WEAK_SCALING=1: W x the batch instead of 1/W per rank)
# Rank 0 writes meshes / snapshots / progress lines; the exit code is the first non-zero one.
N=$1; shift
ID=$(mktemp -u /tmp/rnb_rccl_id.XXXXXX)
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-16}
pids=()
for ((r = 0; r < N; r++)); do
  RNB_WORLD_SIZE=$N RNB_RANK=$r RNB_LOCAL_RANK=$r RNB_RCCL_ID_FILE=$ID "$@" &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || { c=$?; [ $rc -eq 0 ] && rc=$c; }; done
rm -f $ID $ID.tmp
exit $rc

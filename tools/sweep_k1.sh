#!/bin/bash
# GPU box: ms/step of 200-step slices along the run for fixed head lengths of the two-round network evaluation (RNB_FWD_K1) -> gpurun_out/sweep_k1.txt
#   bash tools/sweep_k1.sh "980 1180 1380 1680 1980 2980 5980" "16 24 32 40 48 64"
burns=${1:-"980 1380 1980 2980 5980"}; ks=${2:-"16 24 32 40 48 64"}
: > gpurun_out/sweep_k1.txt
for b in $burns; do
  envs=(); for k in $ks; do envs+=("RNB_FWD_K1=$k"); done
  BURN=$b bash tools/ab_slice.sh sweep_k1_$b 2 "${envs[@]}" > /dev/null 2>&1
  cat gpurun_out/sweep_k1_$b/ab_slice.txt >> gpurun_out/sweep_k1.txt
done
cat gpurun_out/sweep_k1.txt

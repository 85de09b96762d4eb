mkdir -p gpurun_out/loop
for i in $(seq 1 ${N:-14}); do
  timeout 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "whole_step" > gpurun_out/loop/run_$i.log 2>&1
  cp gpurun_out/r04_whole_step_vs_oracle_late.json gpurun_out/loop/late_$i.json 2>/dev/null; cp gpurun_out/r04_whole_step_vs_oracle_window.json gpurun_out/loop/window_$i.json 2>/dev/null
  tail -1 gpurun_out/loop/run_$i.log
done
grep -l "failed" gpurun_out/loop/run_*.log | head

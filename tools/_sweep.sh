export TMPDIR=/tmp
for K in "" "16,16,16,16,16,16,8,8,4,4,2,2,2,2" "16,16,16,16,16,16,16,8,8,4,4,2,2,2" "16,16,16,16,16,16,16,16,8,8,4,4,4,4" "16,16,16,16,16,8,8,8,4,4,4,2,1,1" "16,16,16,16,16,8,8,4,4,4,4,4,4,4"; do
  echo "K=$K"
  RNB_SCATTER_K=$K python bench.py --no-cpu-baseline --steps 100 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'scatter', d['kernels_ms_per_step']['k_grid_scatter']['ms_per_step'])"
done

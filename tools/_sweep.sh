export TMPDIR=/tmp
for cfg in "256 " "128 " "64 " "32 " "256 16,16,16,16,16,16,8,8,4,4,1,1,1,1" "256 16,16,16,16,16,8,8,4,4,2,1,1,1,1" "256 16,16,16,16,16,16,16,8,8,4,2,1,1,1"; do
  set -- $cfg
  echo "LDS_WG=$1 K=$2"
  RNB_SCATTER_LDS_WG=$1 RNB_SCATTER_K=$2 python bench.py --no-cpu-baseline --steps 150 --profile-steps 30 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'scatter', d['kernels_ms_per_step']['k_grid_scatter']['ms_per_step'])"
done

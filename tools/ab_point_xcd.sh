for r in 1 2; do for e in "RNB_POINT_XCD=0" "RNB_POINT_XCD=1"; do
env $e python bench.py --steps 200 --no-cpu-baseline --profile-steps 160 --late-steps 0 --fixed-cost-steps 0 --parity-mode-steps 0 --window-end 0 --no-live-pmc 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
k=d.get('kernels_ms_per_step') or {}
print('$e', 'ms/step', d['ms_per_step'], {n:v for n,v in k.items() if 'point' in n or 'density' in n or 'occup' in n or 'grid_update' in n})
"
done; done

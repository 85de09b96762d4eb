import numpy as np, sys
sys.path.insert(0, ".")
import rnb_neus2_amd as rnb
from rnb_neus2_amd import synthetic
ctx = rnb.Context(apply_no_albedo=1, mask_loss_weight=1.0, overlap=0)
ctx.init_params()
v, n, a = synthetic.make_scene(64, 800)
ctx.set_dataset(v, n, a)
for _ in range(1200):
    st = ctx.train_step()
R = ctx.rays_per_batch
print("rays", R, "measured_before", st.measured_batch_size_before_compaction, "compacted", st.measured_batch_size)
ctx.generate_training_samples(R)
ns = ctx.get("NUMSTEPS", 2 * R).reshape(R, 2).copy()
cnt = ctx.get("COUNTERS")
print("counters", cnt)
ctx.forward_infer_staged(int(cnt[0]))
ctx.compute_loss(R)
nc = ctx.get("NUMSTEPS", 2 * R).reshape(R, 2).copy()
n0, n1 = ns[:, 0].astype(int), nc[:, 0].astype(int)
kept = n0 > 0
print("rays with samples", kept.sum(), "sum n", n0.sum(), "sum compacted", n1.sum())
term = n1 < n0  # terminated early
print("terminated early: %.3f of rays with samples; mean n0 %.1f, mean used %.1f" % (term[kept].mean(), n0[kept].mean(), n1[kept].mean()))
print("percentiles of used (terminated rays):", np.percentile(n1[term], [50, 75, 90, 95, 99, 100]))
print("percentiles of n0:", np.percentile(n0[kept], [5, 25, 50, 75, 90, 99, 100]))
unt = kept & ~term
print("unterminated rays: %d, their n0 sum %d (mean %.1f)" % (unt.sum(), n0[unt].sum(), n0[unt].mean() if unt.any() else 0))
for K1 in (16, 24, 32, 40, 48):
    r1 = np.minimum(n0, K1).sum()
    need2 = kept & (n1 >= np.minimum(n0, K1)) & (n0 > K1)  # not finished within K1
    r2 = (n0[need2] - K1).sum()
    print("K1=%d: round1 %d + round2 %d = %d (%.2f of all), rays in round 2: %d" % (K1, r1, r2, r1 + r2, (r1 + r2) / n0.sum(), need2.sum()))

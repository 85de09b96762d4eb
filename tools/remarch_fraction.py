"""GPU box: which fraction of a step's rays does an occupancy update change? (Sizing of a selective re-march: the march of an update step cannot be generated
ahead because the bitfield changes -- but only rays that visit a cell whose bit flips march differently.) At steps that begin with an update: march the step's rays
with the bits before and after the update (same generator state, same ray count) and compare per ray (sample count and the t of its samples).
   python tools/remarch_fraction.py"""
import sys
sys.path.insert(0, ".")
import numpy as np
import rnb_neus2_amd as rnb
from rnb_neus2_amd import synthetic

scene = synthetic.make_scene(64, 800)
ctx = rnb.Context(apply_no_albedo=1, mask_loss_weight=1.0)
ctx.init_params(); ctx.set_dataset(*scene)


def march(R):
    ctx.generate_training_samples(R, 4096)
    cnt = ctx.get("COUNTERS")
    kept = int(cnt[2])
    idx = ctx.get("RAY_INDICES", kept)
    ns = ctx.get("NUMSTEPS", kept * 2).reshape(kept, 2)
    co = ctx.get("COORDS", int(cnt[3]) * 7).reshape(-1, 7)
    sig = np.zeros(R, dtype=np.float64)
    n = np.zeros(R, dtype=np.int64)
    n[idx] = ns[:, 0]
    cs = np.concatenate([[0.0], np.cumsum(co[:, 0].astype(np.float64) * 3.1 + co[:, 1].astype(np.float64) * 1.7 + co[:, 2].astype(np.float64))])
    sig[idx] = cs[ns[:, 1] + ns[:, 0]] - cs[ns[:, 1]]
    return n, sig, int(cnt[0])


for target in (1008, 1200, 1408, 1600, 2000, 2400, 3008, 4000, 6000):
    while ctx.training_step < target:
        st = ctx.train_step()
    assert ctx.training_step % 16 == 0
    R = ctx.rays_per_batch
    bf0 = ctx.get("DENSITY_BITFIELD", 128 ** 3 // 8).copy()
    n0, s0, m0 = march(R)
    ctx.update_density_grid()
    bf1 = ctx.get("DENSITY_BITFIELD", 128 ** 3 // 8)
    n1, s1, m1 = march(R)
    flipped = int(np.unpackbits(bf0 ^ bf1).sum())
    diff = (n0 != n1) | (np.abs(s0 - s1) > 1e-9)
    with_samples = (n0 > 0) | (n1 > 0)
    print("step %5d: %6d rays (%5d with samples), %5d cells flip, rays that march differently: %5d = %.1f %% of all, %.1f %% of those with samples; marched samples %d -> %d"
          % (target, R, int(with_samples.sum()), flipped, int(diff.sum()), 100.0 * diff.mean(), 100.0 * diff.sum() / max(1, with_samples.sum()), m0, m1), flush=True)
    # restore the controller for the run to go on (the two stage calls did not train)

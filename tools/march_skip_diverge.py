"""GPU box, diagnosis: the first training step at which k_march_count_skip (RNB_MARCH_SKIP=1) and the full march (RNB_MARCH_SKIP=0) produce different samples, and the rays that differ.
Two deterministic contexts in lockstep (identical states until the marches differ)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rnb_neus2_amd as rnb
from rnb_neus2_amd import synthetic

scene = synthetic.make_scene(64, 800)
KW = dict(apply_no_albedo=1, mask_loss_weight=1.0, overlap=0, deterministic=1)
ctx = []
for mode in ("0", "1"):
    os.environ["RNB_MARCH_SKIP"] = mode
    os.environ["RNB_MARCH_BBOX"] = mode
    c = rnb.Context(**KW)
    c.init_params()
    c.set_dataset(*scene)
    ctx.append(c)
os.environ.pop("RNB_MARCH_SKIP")
os.environ.pop("RNB_MARCH_BBOX")
max_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1100
for step in range(max_steps):
    a, b = ctx[0].train_step(), ctx[1].train_step()
    ka = (a.rays_per_batch, a.measured_batch_size_before_compaction, a.n_rays_kept, a.measured_batch_size, a.loss)
    kb = (b.rays_per_batch, b.measured_batch_size_before_compaction, b.n_rays_kept, b.measured_batch_size, b.loss)
    if ka != kb:
        print("step %d differs: full %s | skip %s" % (a.training_step, ka, kb))
        kept_a, kept_b = int(a.n_rays_kept), int(b.n_rays_kept)
        ia, ib = ctx[0].get("RAY_INDICES", kept_a), ctx[1].get("RAY_INDICES", kept_b)
        na, nb = ctx[0].get("NUMSTEPS", 2 * kept_a).reshape(-1, 2), ctx[1].get("NUMSTEPS", 2 * kept_b).reshape(-1, 2)
        da, db = dict(zip(ia.tolist(), na[:, 0].tolist())), dict(zip(ib.tolist(), nb[:, 0].tolist()))
        bad = [(r, da.get(r, 0), db.get(r, 0)) for r in sorted(set(da) | set(db)) if da.get(r, 0) != db.get(r, 0)]
        print("%d rays differ (ray index, samples full, samples skip):" % len(bad), bad[:20])
        ra, rb = ctx[0].get("RAYS", kept_a * 6).reshape(-1, 6), ctx[1].get("RAYS", kept_b * 6).reshape(-1, 6)
        for r, x, y in bad[:6]:
            if r in da:
                k = list(ia).index(r)
                print(" ray %d origin/dir (full ctx):" % r, ra[k].tolist())
                ca = ctx[0].get("COORDS", (int(na[k, 1]) + int(na[k, 0])) * 7).reshape(-1, 7)[int(na[k, 1]):]
                print("   full: first / last sample pos", ca[0, :3].tolist(), ca[-1, :3].tolist())
            if r in db:
                k = list(ib).index(r)
                cb = ctx[1].get("COORDS", (int(nb[k, 1]) + int(nb[k, 0])) * 7).reshape(-1, 7)[int(nb[k, 1]):]
                print("   skip: first / last sample pos", cb[0, :3].tolist(), cb[-1, :3].tolist())
        break
else:
    print("no difference in %d steps" % max_steps)

"""GPU box: does the overlapped schedule (cfg.overlap = 1: the next step's march runs on a side stream beside the backward pass) produce the
same ray / sample sets as the serial schedule? Clones a trained state N times, runs 6 steps each, and counts the steps whose marched-sample
counters differ from the serial reference (see DESIGN.md section 6 for what this found).   python tools/march_determinism.py [N]
Environment: GL_OVERLAP=0 (serial clones), GL_BASE_STEPS=n (trained state to clone; default 400; 3000+ = the large-batch kernels), RNB_MARCH_LATE=1, RNB_FWD_BWD_GENERIC=1, ..."""
import os, sys
import numpy as np
sys.path.insert(0, ".")
import rnb_neus2_amd as rnb
from rnb_neus2_amd import synthetic
scene = synthetic.make_scene(64, 800)
KW = dict(apply_no_albedo=1, mask_loss_weight=1.0)
base = rnb.Context(overlap=0, **KW); base.init_params(); base.set_dataset(*scene)
for _ in range(int(os.environ.get("GL_BASE_STEPS", "400"))): st = base.train_step()
state = dict(params=base.get("PARAMS_FP32").copy(), grid=base.get("DENSITY_GRID").copy(), step=base.training_step, rays=base.rays_per_batch, before=st.measured_batch_size_before_compaction)
def clone(overlap=1):
    c = rnb.Context(overlap=overlap, **KW); c.init_params(); c.set_dataset(*scene); c.set_params(state["params"]); c.put("DENSITY_GRID", state["grid"]); c.update_density_bitfield()
    c.set_controller(state["step"], state["rays"], state["before"], 0); return c
ref = clone(0)
R = [ref.train_step().as_dict() for _ in range(6)]
ref.close()
keys = ("rays_per_batch", "measured_batch_size_before_compaction", "measured_batch_size", "n_rays_kept", "loss", "ek_loss", "mask_loss")
bad = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
OVL = int(os.environ.get("GL_OVERLAP", "1"))
march = 0
for rep in range(N):
    c = clone(OVL)
    for i in range(6):
        d = c.train_step().as_dict()
        r = R[i]
        odd = [k for k in keys if (abs(d[k] - r[k]) > (3e-3 * abs(r[k]) + 1e-9) if isinstance(r[k], float) else d[k] != r[k])]
        if i == 1 and d["measured_batch_size_before_compaction"] != r["measured_batch_size_before_compaction"]:
            march += 1
            print("rep", rep, "step", i, {k: (d[k], r[k]) for k in odd}, flush=True)
    c.close()
print("march anomalies at step 1:", march, "of", N, flush=True)

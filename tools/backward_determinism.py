"""GPU box: is the backward pass of the overlapped schedule (dW GEMMs / scatter / optimizer / next march side by side) bit-identical to the serial
schedule? First step from a common trained state, N clones: MLP gradients, dL/dout and grid gradients compared bit for bit with the serial run.
   python tools/backward_determinism.py [N]      GL_ALBEDO=1: albedo mode; RNB_MARCH_LATE=1: with the march held back until k_fwd_bwd is done"""
import os, sys
import numpy as np
sys.path.insert(0, ".")
import rnb_neus2_amd as rnb
from rnb_neus2_amd import synthetic
scene = synthetic.make_scene(64, 800)
ALB = int(os.environ.get("GL_ALBEDO", "0"))
KW = dict(apply_no_albedo=0 if ALB else 1, mask_loss_weight=1.0)
base = rnb.Context(overlap=0, **KW); base.init_params(); base.set_dataset(*scene)
for _ in range(400): st = base.train_step()
state = dict(params=base.get("PARAMS_FP32").copy(), grid=base.get("DENSITY_GRID").copy(), step=base.training_step, rays=base.rays_per_batch, before=st.measured_batch_size_before_compaction)
def clone(overlap):
    c = rnb.Context(overlap=overlap, **KW); c.init_params(); c.set_dataset(*scene); c.set_params(state["params"]); c.put("DENSITY_GRID", state["grid"]); c.update_density_bitfield()
    c.set_controller(state["step"], state["rays"], state["before"], 0); return c
def grads_of(c):
    c.train_step_begin(); cnt, sums = c.train_step_local(); c.train_step_finish(cnt, sums)
    g = c.get("GRADS_FP32").copy(); d = c.get("DLOSS_DOUT").copy()
    c.train_step_apply()
    return g, d
ref = clone(0); g_ref, d_ref = grads_of(ref); lay = ref.param_layout(); ref.close()
nm = lay["grid"]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
bad_mlp = bad_dout = 0; worst_grid = 0.0
gs = np.abs(g_ref[nm:]).max()
for rep in range(N):
    c = clone(1); g, d = grads_of(c); c.close()
    if not np.array_equal(g[:nm].view(np.uint32), g_ref[:nm].view(np.uint32)):
        bad_mlp += 1
        k = np.nonzero(g[:nm] != g_ref[:nm])[0]
        print("rep", rep, "MLP gradient entries differing:", k.size, "max rel", float(np.max(np.abs(g[:nm][k] - g_ref[:nm][k])) / np.abs(g_ref[:nm]).max()), flush=True)
    if not np.array_equal(d.view(np.uint16), d_ref.view(np.uint16)): bad_dout += 1
    worst_grid = max(worst_grid, float(np.max(np.abs(g[nm:] - g_ref[nm:])) / gs))
    var_rel = abs(g[-4] - g_ref[-4]) / (abs(g_ref[-4]) + 1e-30)
    if var_rel > 1e-6: print("rep", rep, "variance gradient differs rel", var_rel, flush=True)
print("nonzero grads in ref:", int(np.count_nonzero(g_ref)), "grid max", float(gs)); print("albedo", ALB, "reps", N, "MLP-gradient mismatches", bad_mlp, "dL/dout mismatches", bad_dout, "worst grid gradient deviation / max", worst_grid)

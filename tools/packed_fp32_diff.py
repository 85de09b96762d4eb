"""GPU box, measurement aid for tools/packed_fp32_bisect.sh: WHAT differs between the serial and the overlapped schedule on a build with packed fp32 in the evaluation kernels?
One state (256 training steps), then per repetition a fresh overlapped context runs train_step_begin and its network output (RNB_BUF_MLP_OUT, 16 halfs per marched sample) is
compared with the serial context's: how many samples differ, in which lanes of their 64-sample tile, in which channels, by how much.   python tools/packed_fp32_diff.py [reps=20]"""
import sys
sys.path.insert(0, ".")
import numpy as np
import rnb_neus2_amd as rnb
from rnb_neus2_amd import synthetic

KW = dict(apply_no_albedo=1, mask_loss_weight=1.0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
scene = synthetic.make_scene(64, 800)
c = rnb.Context(overlap=0, **KW); c.init_params(); c.set_dataset(*scene)
st = None
for _ in range(1008):
    st = c.train_step()
state = dict(params=c.get("PARAMS_FP32").copy(), grid=c.get("DENSITY_GRID").copy(), step=c.training_step, rays=c.rays_per_batch, before=st.measured_batch_size_before_compaction,
             adam_m=c.get("ADAM_M").copy(), adam_v=c.get("ADAM_V").copy(), adam_steps=c.get("ADAM_STEPS").copy(), ema=c.get("PARAMS_EMA").copy())
c.close()


def fresh(overlap):
    x = rnb.Context(overlap=overlap, **KW); x.init_params(); x.set_dataset(*scene)
    x.set_params(state["params"]); x.put("ADAM_M", state["adam_m"]); x.put("ADAM_V", state["adam_v"]); x.put("ADAM_STEPS", state["adam_steps"]); x.put("PARAMS_EMA", state["ema"])
    x.set_optimizer_step(state["step"]); x.put("DENSITY_GRID", state["grid"]); x.update_density_bitfield(); x.set_controller(state["step"], state["rays"], state["before"], 0)
    return x


def run(overlap):
    x = fresh(overlap)
    try:
        x.train_step_begin()
        cnt, sums = x.train_step_local()
        n = int(cnt[0])
        return n, x.get("MLP_OUT", n * 16).view(np.uint16).reshape(n, 16).copy(), x.get("COORDS", n * 7).view(np.uint32).reshape(n, 7).copy(), x.get("DLOSS_DOUT").view(np.uint16).copy()
    finally:
        x.close()


n0, out0, co0, d0 = run(0)
print("serial: %d marched samples" % n0)
for rep in range(reps):
    n, out, co, d = run(1)
    same_coords = n == n0 and np.array_equal(co, co0)
    if not same_coords:
        print("rep %d: the march differs (n %d vs %d)" % (rep, n, n0)); continue
    ev = (out0 != 0).any(axis=1) | (out != 0).any(axis=1)  # samples the two-round evaluation touched
    diff = (out != out0)
    rows = np.flatnonzero(diff.any(axis=1))
    msg = "rep %d: %d of %d evaluated samples differ; dL/dout equal: %s" % (rep, len(rows), int(ev.sum()), np.array_equal(d, d0))
    if len(rows):
        a, b = out.view(np.float16).astype(np.float64), out0.view(np.float16).astype(np.float64)
        ch = diff[rows].sum(axis=0)
        rel = np.abs(a[rows] - b[rows]).max(axis=1) / (np.abs(b[rows]).max(axis=1) + 1e-12)
        msg += "; channels %s; max rel dev %.3g, median %.3g; nan %d; first rows %s; slot mod 64 of the first rows %s" % (ch.tolist(), rel.max(), np.median(rel), int(np.isnan(a[rows]).sum()), rows[:12].tolist(), (rows[:12] % 64).tolist())
        r = rows[0]
        msg += "\n   row %d overlapped %s\n   row %d serial     %s" % (r, a[r].round(5).tolist(), r, b[r].round(5).tolist())
    print(msg, flush=True)

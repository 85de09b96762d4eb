"""How far do deviations D1 (fp32 instead of fp16 accumulation in the MLP dot products and weight-gradient GEMMs) and D2 (fp32
instead of half2-atomic accumulation of the hash-grid gradients) move the loss and the gradients away from the reference
as coded?  TEST INFRASTRUCTURE (drives oracle/liborc.so).

The reference cannot be run here; the oracle can emulate its half accumulation (ORC_EMULATE_FP16_ACCUM, ORC_EMULATE_HALF_ATOMICS,
see oracle/rnb_oracle.cpp dot_h / emulated_dw / emulated_accumulate -- an emulation MODEL of tensor-core half accumulators, not a
bit-exact statement). From one trained state this script runs the same step in four oracle modes and prints

  * per-ray loss / loss sums of the step (forward arithmetic: D1 only),
  * gradient blocks: max and rms deviation relative to the block's scale, cosine to the default mode,
  * the parameter update after one Adam step, and the loss after `--steps` further training steps in each mode.

  python tools/oracle_deviation_report.py [--gpu-state STEPS] [--views 64 --res 800 --batch-log2 18] [--steps 30] [--out FILE]

With --gpu-state N (on the GPU box) the state is the HIP library's after N training steps of config 4 (the oracle takes about
1.3 s per full-size step on that host's 256 cores); without it the oracle itself trains a smaller scene first."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

MODES = {"default (D1+D2: fp32 accumulate)": {}, "fp16 MLP/GEMM accumulators": {"ORC_EMULATE_FP16_ACCUM": "1"},
         "half2-atomic grid gradients": {"ORC_EMULATE_HALF_ATOMICS": "1"}, "both (reference as coded, emulated)": {"ORC_EMULATE_FP16_ACCUM": "1", "ORC_EMULATE_HALF_ATOMICS": "1"}}


def make_ctx(env, **kw):
    from tests import oracle_lib
    old = {k: os.environ.get(k) for k in ("ORC_EMULATE_FP16_ACCUM", "ORC_EMULATE_HALF_ATOMICS")}
    for k in old:
        os.environ.pop(k, None)
    os.environ.update(env)
    try:
        return oracle_lib.context(**kw)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpu-state", type=int, default=0)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--res", type=int, default=160)
    ap.add_argument("--batch-log2", type=int, default=13)
    ap.add_argument("--pretrain", type=int, default=120, help="oracle training steps to reach the state when no GPU state is used")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from rnb_neus2_amd import synthetic
    scene = synthetic.make_scene(a.views, a.res)
    B = 1 << a.batch_log2
    kw = dict(apply_no_albedo=1, mask_loss_weight=1.0, target_batch_size=B, max_rays_per_batch=min(1 << 18, B), initial_rays_per_batch=min(4096, B))
    t0 = time.time()
    if a.gpu_state:
        import rnb_neus2_amd as rnb
        g = rnb.Context(**kw)
        g.init_params(); g.set_dataset(*scene)
        for _ in range(a.gpu_state):
            st = g.train_step()
        state = dict(params=g.get("PARAMS_FP32").copy(), grid=g.get("DENSITY_GRID").copy(), step=g.training_step, rays=g.rays_per_batch, before=st.measured_batch_size_before_compaction)
        g.close()
    else:
        c = make_ctx({}, **kw)
        c.init_params(); c.set_dataset(*scene)
        for _ in range(a.pretrain):
            st = c.train_step()
        state = dict(params=c.get("PARAMS_FP32").copy(), grid=c.get("DENSITY_GRID").copy(), step=c.training_step, rays=c.rays_per_batch, before=st.measured_batch_size_before_compaction)
        c.close()
    print("state: step %d, %d rays/step, %.0f s" % (state["step"], state["rays"], time.time() - t0), flush=True)
    res = {"state_step": int(state["step"]), "rays": int(state["rays"]), "batch": B, "views": a.views, "res": a.res, "modes": {}}
    ref = None
    for name, env in MODES.items():
        c = make_ctx(env, **kw)
        c.init_params(); c.set_dataset(*scene); c.set_params(state["params"]); c.put("DENSITY_GRID", state["grid"]); c.update_density_bitfield()
        step = state["step"] | 1  # not an occupancy-update step: every mode marches the same sample set
        c.set_controller(step, state["rays"], state["before"], 0)
        lay = c.param_layout()
        c.train_step_begin()
        cnt, sums = c.train_step_local()
        grads = c.get("GRADS_FP32").astype(np.float64)
        n = int(state["rays"])
        per_ray = np.stack([c.get(k, n).astype(np.float64) for k in ("LOSS", "EK_LOSS", "MASK_LOSS")])
        st = c.train_step_finish(cnt, sums)
        c.train_step_apply()
        params = c.get("PARAMS_FP32").astype(np.float64)
        losses = [float(st.loss)]
        for _ in range(a.steps):
            losses.append(float(c.train_step().loss))
        cur = dict(cnt=cnt, sums=sums, grads=grads, per_ray=per_ray, params=params, losses=losses)
        c.close()
        if ref is None:
            ref = cur
        out = {"counters": [int(x) for x in cnt], "loss_sums": [float(x) for x in sums],
               "loss_sums_rel_dev": [float(abs(x - y) / (abs(y) + 1e-30)) for x, y in zip(sums, ref["sums"])],
               "per_ray_loss_max_rel_dev": float(np.max(np.abs(per_ray[0] - ref["per_ray"][0]) / (np.abs(ref["per_ray"][0]) + 1e-6 * np.abs(ref["per_ray"][0]).max()))),
               "loss_after_%d_steps" % a.steps: losses[-1], "mean_loss_last_10": float(np.mean(losses[-10:]))}
        blocks = {"sdf_mlp": (lay["sdf"], lay["rgb"]), "hash_grid": (lay["grid"], lay["variance"])}
        for bname, (lo, hi) in blocks.items():
            x, y = grads[lo:hi], ref["grads"][lo:hi]
            sc = np.abs(y).max()
            nz = y != 0
            out[bname] = {"max_dev_over_scale": float(np.abs(x - y).max() / sc), "rms_dev_over_rms": float(np.sqrt(np.mean((x - y) ** 2)) / np.sqrt(np.mean(y ** 2))),
                          "cosine": float(x @ y / (np.linalg.norm(x) * np.linalg.norm(y))),
                          "median_rel_dev_nonzero": float(np.median(np.abs(x[nz] - y[nz]) / np.abs(y[nz]))) if nz.any() else 0.0,
                          "entries_lost_to_zero": int(np.count_nonzero((x == 0) & nz)), "nonzero_entries": int(nz.sum())}
        dp, dr = params - state["params"], ref["params"] - state["params"]
        out["adam_update_cosine"] = float(dp @ dr / (np.linalg.norm(dp) * np.linalg.norm(dr)))
        out["adam_update_entries_differing"] = float(np.mean(dp != dr))
        res["modes"][name] = out
        print(name, json.dumps(out), flush=True)
    res["seconds"] = round(time.time() - t0, 1)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()

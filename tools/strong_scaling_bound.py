"""GPU box: the compute side of strong scaling on ONE GPU. A rank of a W-GPU strong-scaling job runs the step with B/W compacted samples and R/W rays
(dp.strong_scaling_sizes); how long that step takes on one MI355X -- without any exchange -- bounds what W GPUs can gain: speed-up <= t(1) / (t(W) + exchange).
   python tools/strong_scaling_bound.py [steps]"""
import sys, time
sys.path.insert(0, ".")
import rnb_neus2_amd as rnb
from rnb_neus2_amd import synthetic, dp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
scene = synthetic.make_scene(64, 800)
base = None
for W in (1, 2, 4, 8):
    ctx = rnb.Context(apply_no_albedo=1, mask_loss_weight=1.0, **dp.strong_scaling_sizes(W))
    ctx.init_params(); ctx.set_dataset(*scene)
    for i in range(1000):
        ctx.train_step()
    t0 = time.perf_counter(); rays = 0
    for i in range(n - 1000):
        rays += ctx.train_step().rays_per_batch
    dt = (time.perf_counter() - t0) / (n - 1000)
    base = base or dt
    print("W = %d: B/W = %6d samples, %.4f ms/step over steps 1000-%d (%.1f k rays/step at the end) -> compute-side bound of the speed-up %.2fx; with a 0.12 ms exchange %.2fx"
          % (W, (1 << 18) // W, 1e3 * dt, n, ctx.rays_per_batch / 1e3, base / dt, base / (dt + 0.12e-3)), flush=True)
    ctx.close()

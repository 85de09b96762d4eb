"""GPU box: long run of the default training path (config 4) — loss, rays/step and ms/step every 1000 steps."""
import sys, time
sys.path.insert(0, ".")
import rnb_neus2_amd as rnb
from rnb_neus2_amd import synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ctx = rnb.Context(apply_no_albedo=0 if (len(sys.argv) > 2 and sys.argv[2] == "albedo") else 1, mask_loss_weight=1.0)
ctx.init_params(); ctx.set_dataset(*synthetic.make_scene(64, 800))
t0 = time.perf_counter(); rays = 0
for i in range(1, n + 1):
    st = ctx.train_step(); rays += st.rays_per_batch
    if i % 1000 == 0:
        dt = time.perf_counter() - t0
        print("step %6d loss %.6f ek %.5f mask %.5f rays/step %6d  %.3f ms/step  %.2f M rays/s" % (i, st.loss, st.ek_loss, st.mask_loss, st.rays_per_batch, 1e3 * dt / 1000, rays / dt / 1e6), flush=True)
        t0 = time.perf_counter(); rays = 0
ctx.close()

"""GPU box: long run of the default training path (config 4) -- loss, rays/step and ms/step every 1000 steps, then the surface: marching cubes at 512^3 on the EMA
weights and the distance of the mesh vertices from the synthetic scene's sphere (radius 0.25 around the cube's centre).
    python tools/soak.py [n_steps=10000] [albedo|-] [half|fp32]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import rnb_neus2_amd as rnb
from rnb_neus2_amd import synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
albedo = len(sys.argv) > 2 and sys.argv[2] == "albedo"
half = len(sys.argv) > 3 and sys.argv[3] == "half"
ctx = rnb.Context(apply_no_albedo=0 if albedo else 1, mask_loss_weight=1.0, accumulate=1 if half else 0)
ctx.init_params(); ctx.set_dataset(*synthetic.make_scene(64, 800))
print("accumulate = %s%s" % ("half" if half else "fp32", ", albedo" if albedo else ""), flush=True)
t0 = time.perf_counter(); rays = 0
for i in range(1, n + 1):
    st = ctx.train_step(); rays += st.rays_per_batch
    if i % 1000 == 0:
        dt = time.perf_counter() - t0
        print("step %6d loss %.6f ek %.5f mask %.5f rays/step %6d  %.3f ms/step  %.2f M rays/s" % (i, st.loss, st.ek_loss, st.mask_loss, st.rays_per_batch, 1e3 * dt / 1000, rays / dt / 1e6), flush=True)
        t0 = time.perf_counter(); rays = 0
res = 512
t0 = time.perf_counter()
lat = ctx.sdf_lattice(res)
verts, idx = ctx.marching_cubes(lat, res)
ctx.device_free(lat)
dt = time.perf_counter() - t0
r = np.linalg.norm(verts.astype(np.float64) - 0.5, axis=1)
print("mesh %d^3: %d vertices, %d triangles in %.1f ms; |v - centre| mean %.5f (sphere 0.25), rms deviation %.5f, max %.5f" % (res, len(verts), len(idx) // 3, 1e3 * dt, r.mean(), np.sqrt(np.mean((r - 0.25) ** 2)), np.abs(r - 0.25).max()), flush=True)
off = np.abs(r - 0.25) > 0.005
if off.any():
    d = verts[off].astype(np.float64) - 0.5
    print("   %d vertices further than 0.005 from the sphere: radius %.4f .. %.4f, mean direction %s, extent %s" % (int(off.sum()), r[off].min(), r[off].max(), np.round((d / np.linalg.norm(d, axis=1, keepdims=True)).mean(0), 3).tolist(), np.round(np.ptp(verts[off], axis=0), 4).tolist()), flush=True)
ctx.close()

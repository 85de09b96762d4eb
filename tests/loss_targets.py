"""Restates, in numpy, how the loss kernel derives a ray's light direction and shading target from the dataset
(testbed_nerf.cu:1498-1593) for the ray stored in slot ``i`` of an oracle context. Test infrastructure."""
import numpy as np

from tests.test_oracle_cpu import Pcg32


def _srgb_to_linear(s):
    return s / 12.92 if s <= 0.04045 else ((s + 0.055) / 1.055) ** 2.4


def _linear_to_srgb(v):
    return 12.92 * v if v < 0.0031308 else 1.055 * v ** 0.41666 - 0.055


def ray_light_and_target(c, views, normals, slot, n_rays, seed=1337):
    kept = int(c.get("COUNTERS")[2])
    ray_idx = int(c.get("RAY_INDICES", kept)[slot])
    n_img = len(views)
    img = ((ray_idx * n_img) // n_rays) % n_img
    rng = Pcg32(seed)
    rng.next_uint()  # density_grid_rng seed draw (testbed.cu:2236)
    rng.next_uint()  # tv_loss_rng seed draw
    state0 = (rng.state, rng.inc)

    def stream(offset):
        r = Pcg32(0)
        r.state, r.inc = state0
        for _ in range(offset):
            r.next_uint()
        return r
    r = stream(ray_idx * 8)
    w = h = views[img]["width"]
    xy = []
    for res in (w, h):
        p = float(r.next_float()) * res
        p = min(max(p, 0.0), res - 1)
        xy.append((np.float32(p) + np.float32(0.5)) / np.float32(res))
    px = min(max(int(xy[0] * w), 0), w - 1)
    py = min(max(int(xy[1] * h), 0), h - 1)
    pix = normals[img][py, px].astype(np.float64)
    alpha = pix[3] / 65535.0
    nv = np.array([_linear_to_srgb(_srgb_to_linear(pix[k] / 65535.0) * alpha) * 2 - 1 for k in range(3)])
    nv[1] *= -1
    nv[2] *= -1
    nv /= np.linalg.norm(nv)
    lr = stream(ray_idx * 8 + 7)
    k = lr.next_uint() % 3
    slant = np.radians(54.74)
    tilt = np.radians([0.0, 120.0, 240.0])[k]
    light_cam = np.array([-np.sin(slant) * np.cos(tilt), -np.sin(slant) * np.sin(tilt), -np.cos(slant)])
    R = np.asarray(views[img]["xform"], dtype=np.float64)[:, :3]
    light = R @ light_cam
    target = np.array([1.0, 1.0, 1.0, 0.0]) * float(nv @ light_cam)
    mask_gt = 1.0 if alpha > 0.99 else 0.0
    return light, target, mask_gt

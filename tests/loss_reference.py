"""Independent float64 numpy statement of the per-ray NeuS compositing loss used to check the
oracle's analytic output gradients by finite differences. Test infrastructure."""
import numpy as np

from tests import oracle_lib


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def ray_loss(out, dt, d, light, target, mask_gt, mask_w):
    """Colour (L2, RGB+ halved, no-albedo) + sigmoid-BCE mask loss of one ray; `out` = float64 [n,16] network outputs."""
    T = 1.0
    rgb = np.zeros(4)
    W = 0.0
    albedo = np.array([1.0, 1.0, 1.0, 0.0])
    for j in range(len(out)):
        if T < 1e-4:
            break
        s = np.exp(10.0 * out[j, 7])
        g = out[j, 4:7]
        cos = float(d @ g)
        ic = -max(-cos, 0.0)
        nxt = out[j, 3] + ic * dt * 0.5
        prv = out[j, 3] - ic * dt * 0.5
        p = _sigmoid(prv * s) - _sigmoid(nxt * s)
        c = _sigmoid(prv * s)
        alpha = min(max((p + 1e-5) / (c + 1e-5), 0.0), 1.0)
        w = alpha * T
        rgb += w * albedo * float(g @ light)
        W += w
        T *= 1 - alpha
    colour = 0.5 * np.sum((rgb - target) ** 2)
    Wc = min(max(W, 1e-4), 1 - 1e-4)
    mask = -(mask_gt * np.log(_sigmoid(Wc)) + (1 - mask_gt) * np.log(1 - _sigmoid(Wc)))
    return colour + mask_w * mask


def ray_loss_and_grads(out, dt, d, light, target, mask_gt, mask_w, n_rays, h=1e-5):
    out = out.astype(np.float64)
    grad = np.zeros_like(out)
    for j in range(len(out)):
        for ch in (3, 4, 5, 6):
            o1, o2 = out.copy(), out.copy()
            o1[j, ch] += h
            o2[j, ch] -= h
            grad[j, ch] = (ray_loss(o1, dt, d, light, target, mask_gt, mask_w) - ray_loss(o2, dt, d, light, target, mask_gt, mask_w)) / (2 * h)
    # the loss kernel routes the normal's part of the colour/mask gradient to channels 8..10 and keeps 4..6 for Eikonal
    full = np.zeros_like(out)
    full[:, 3] = grad[:, 3] / n_rays
    full[:, 8:11] = grad[:, 4:7] / n_rays
    return {"grad": full}


def analytic_from_oracle(seed=0):
    """Run the oracle's loss kernel on ONE synthetic ray with hand-made network outputs; return its inputs and dL/dout."""
    from rnb_neus2_amd import synthetic
    rng = np.random.default_rng(seed)
    c = oracle_lib.context(target_batch_size=1 << 12, max_rays_per_batch=1 << 12, initial_rays_per_batch=128, apply_no_albedo=1, mask_loss_weight=0.7)
    c.init_params()
    views, nm, al = synthetic.make_scene(2, 32, 56.0)
    c.set_dataset(views, nm, al)
    c.set_training_step(0)
    c.update_density_grid()
    n_rays = 64
    c.generate_training_samples(n_rays, 0)
    cnt = c.get("COUNTERS")
    kept, written = int(cnt[2]), int(cnt[3])
    numsteps = c.get("NUMSTEPS", kept * 2).reshape(kept, 2)
    coords = c.get("COORDS", written * 7).reshape(written, 7)
    # hand-made outputs: a soft surface crossing along every ray
    out = np.zeros((written, 16), np.float16)
    for n, b in numsteps:
        t = np.linspace(0.06, -0.06, n) + rng.normal(0, 0.002, n)
        out[b:b + n, 3] = t
        dirv = coords[b, 4:7] * 2 - 1
        g = -dirv / np.linalg.norm(dirv) + rng.normal(0, 0.15, (n, 3))
        out[b:b + n, 4:7] = g
        out[b:b + n, 7] = 0.3
        out[b:b + n, 8:11] = coords[b:b + n, 4:7]
    c.put("MLP_OUT", out)
    c.compute_loss(n_rays, 0)
    ns2 = c.get("NUMSTEPS", kept * 2).reshape(kept, 2)
    dl = c.get("DLOSS_DOUT").astype(np.float32).reshape(-1, 16)
    # pick the longest compacted ray
    i = int(np.argmax(ns2[:, 0]))
    ncomp, cb = int(ns2[i, 0]), int(ns2[i, 1])
    b0 = int(numsteps[i, 1])
    o = out[b0:b0 + ncomp].astype(np.float64)
    dirv = out[b0, 8:11].astype(np.float64) * 2 - 1
    dirv /= np.linalg.norm(dirv)
    # recover the ray's light / target from the oracle's own logged quantities is not possible through the ABI; restate them:
    from tests.loss_targets import ray_light_and_target
    light, target, mask_gt = ray_light_and_target(c, views, nm, i, n_rays)
    dt = (np.sqrt(3) / 1024)
    res = dict(out=o, dt=float(np.float32(1.73205080757) / np.float32(1024)), dir=dirv, light=light, target=target, mask_gt=mask_gt, mask_w=0.7,
               n_rays=n_rays, dloss=dl[cb:cb + ncomp])
    c.close()
    return res

"""GPU box: the half mode's SDF-MLP weight gradients against the oracle's model from the same loss gradients, with the reference's split-K order (default) and with
the training kernel's own tiling (RNB_DW_SLICED=0, rounds 4-5):  python tools/dw_sliced_compare.py  -> max deviation / matrix scale, cosine, share of equal halves."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_parity import _pair, _randomize, _stage_samples  # noqa: E402

for sliced, albedo in (("1", 0), ("0", 0), ("1", 1), ("0", 1)):
    os.environ["RNB_DW_SLICED"] = sliced
    gpu, cpu = _pair(apply_no_albedo=0 if albedo else 1, accumulate=1)
    _randomize(gpu, cpu, seed=1)
    for n_rays in (512, 4096):
        _stage_samples(gpu, cpu, n_rays, step=700)
        cpu.compute_loss(n_rays, 0)
        gpu.put("DLOSS_DOUT", cpu.get("DLOSS_DOUT"))
        gpu.put("COORDS_COMPACTED", cpu.get("COORDS_COMPACTED"))
        for c in (gpu, cpu):
            c.forward_backward()
        lay = cpu.param_layout()
        hi = lay["grid"] if albedo else lay["rgb"]  # with the colour MLP: both MLPs
        g = gpu.get("GRADS_FP16")[lay["sdf"]:hi].astype(np.float64)
        r = cpu.get("GRADS_FP16")[lay["sdf"]:hi].astype(np.float64)
        scale = np.abs(r).max()
        print("RNB_DW_SLICED=%s %s rays %5d: max |d| / scale %.3e  rms %.3e  cosine %.8f  equal halves %.4f  (W0 %.4f, W1 %.4f)" % (
            sliced, "albedo   " if albedo else "no albedo", n_rays, np.abs(g - r).max() / scale, np.sqrt(np.mean((g - r) ** 2)) / np.sqrt(np.mean(r ** 2)), g @ r / np.linalg.norm(g) / np.linalg.norm(r),
            np.mean(g == r), np.mean(g[:2048] == r[:2048]), np.mean(g[2048:2112] == r[2048:2112])), flush=True)
    gpu.close()
    cpu.close()

"""GPU box: how much of a backward pass of the half mode (accumulate = RNB_ACCUM_HALF) equals the oracle's model BIT FOR BIT when the order of the hash-grid atomics is taken out
(deterministic = 1: exact integer sums on both sides) -- what is left is the operands the matrix cores form in an addition order of their own.
    python tools/half_mode_equal_bits.py   -> per block: share of equal halves, largest deviation over the block's scale"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_parity import _pair, _randomize, _stage_samples  # noqa: E402

for albedo in (0, 1):
    gpu, cpu = _pair(apply_no_albedo=0 if albedo else 1, accumulate=1, deterministic=1)
    _randomize(gpu, cpu, seed=1)
    for n_rays in (512, 4096):
        _stage_samples(gpu, cpu, n_rays, step=700)
        cpu.compute_loss(n_rays, 0)
        gpu.put("DLOSS_DOUT", cpu.get("DLOSS_DOUT"))
        gpu.put("COORDS_COMPACTED", cpu.get("COORDS_COMPACTED"))
        for c in (gpu, cpu):
            c.forward_backward()
        lay = cpu.param_layout()
        g, r = gpu.get("GRADS_FP16").astype(np.float64), cpu.get("GRADS_FP16").astype(np.float64)
        parts = []
        for name, lo, hi in (("sdf mlp", lay["sdf"], lay["rgb"]), ("rgb mlp", lay["rgb"], lay["grid"]), ("hash grid", lay["grid"], lay["variance"])):
            a, b = g[lo:hi], r[lo:hi]
            if not b.any():
                continue
            nz = (a != 0) | (b != 0)
            parts.append("%s %.5f of %d touched equal, max |d| / scale %.1e" % (name, np.mean(a[nz] == b[nz]), int(nz.sum()), np.abs(a - b).max() / np.abs(b).max()))
        print("%s rays %5d: %s; variance %s" % ("albedo   " if albedo else "no albedo", n_rays, "; ".join(parts), "equal" if g[lay["variance"]] == r[lay["variance"]] else "%g vs %g" % (g[lay["variance"]], r[lay["variance"]])), flush=True)
    gpu.close()
    cpu.close()

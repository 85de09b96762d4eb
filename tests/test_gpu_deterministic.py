"""rnb_config::deterministic (include/rnb_neus2.h, ABI 5): the hash-grid gradients summed as 64-bit fixed-point integers (every addend is a half value, grid.h:415-416,
hence an exact integer at scale 2^24; integer atomics commute) and narrowed once. What that buys, asserted here bit for bit at sizes the oracle finishes in seconds:
a backward pass gives the same bits every time; the sums do not depend on which scatter kernel a level goes through (LDS-privatised / run-length / one atomic per
corner: three different groupings of the same addends); a training run is reproducible -- same state in, same bits out -- on the serial and on the overlapped
schedule alike; and the mode agrees with the oracle's statement of it (exact integer sums on the CPU) within the tolerances of the default mode's stage tests.
The full-size statements (a pinned state at step 1009, the as-coded comparison on it) are in tests/test_gpu_fullsize.py."""
import hashlib

import numpy as np
import pytest

from tests.test_gpu_parity import _env, _pair, _randomize, _stage_samples

pytestmark = pytest.mark.gpu


def _grads(c):
    return c.get("GRADS_FP16" if c.cfg.accumulate else "GRADS_FP32").copy()


def _bits(a):
    return a.view(np.uint16 if a.dtype == np.float16 else np.uint32)


def _staged(accumulate, no_albedo, env=None, seed=1):
    gpu, cpu = _pair(env=env, deterministic=1, accumulate=accumulate, apply_no_albedo=no_albedo)
    _randomize(gpu, cpu, seed=seed)
    n_rays = 512
    _stage_samples(gpu, cpu, n_rays, step=700)
    cpu.compute_loss(n_rays, 0)
    gpu.put("DLOSS_DOUT", cpu.get("DLOSS_DOUT"))
    gpu.put("COORDS_COMPACTED", cpu.get("COORDS_COMPACTED"))
    return gpu, cpu


@pytest.mark.parametrize("accumulate", [0, 1])
@pytest.mark.parametrize("no_albedo", [0, 1])
def test_backward_pass_gives_the_same_bits_every_time_and_agrees_with_the_oracle(accumulate, no_albedo):
    gpu, cpu = _staged(accumulate, no_albedo)
    try:
        lay = cpu.param_layout()
        runs = []
        for _ in range(4):
            gpu.forward_backward()
            runs.append(_grads(gpu))
        for r in runs[1:]:
            assert np.array_equal(_bits(runs[0]), _bits(r))  # MLPs (fixed-order sums), hash grid (integer sums), variance: every bit
        cpu.forward_backward()
        g, r = runs[0].astype(np.float64), _grads(cpu).astype(np.float64)
        gg, rg = g[lay["grid"]:lay["variance"]], r[lay["grid"]:lay["variance"]]
        assert rg.any()
        # the two sides sum exactly; what differs is the operands (the MLP backward's half outputs: MFMA order vs the oracle's sequential dot products)
        assert np.mean((gg != 0) != (rg != 0)) < 1e-4
        scale = np.abs(rg).max()
        assert np.abs(gg - rg).max() / scale < (4e-3 if accumulate else 2e-3)
        nz = rg != 0
        rel = np.abs(gg[nz] - rg[nz]) / (np.abs(rg[nz]) + 1e-3 * scale)
        assert np.quantile(rel, 0.999) < (4e-2 if accumulate else 2e-2)
        if accumulate:
            # the half mode with the order of the atomics taken out IS the model up to the few operands the matrix cores round the other way (round 6, with the weight gradients in the
            # reference's split-K order; tools/half_mode_equal_bits.py: 99.96 % of 745 k touched hash-grid entries, 99.9-100 % of the MLP weights, the variance gradient)
            touched = (gg != 0) | (rg != 0)
            assert np.mean(gg[touched] == rg[touched]) >= 0.999 and np.abs(gg - rg).max() / scale < 2e-4, (np.mean(gg[touched] == rg[touched]), np.abs(gg - rg).max() / scale)
            gm, rm = g[lay["sdf"]:lay["grid"]], r[lay["sdf"]:lay["grid"]]
            tm = (gm != 0) | (rm != 0)
            assert np.mean(gm[tm] == rm[tm]) >= 0.995 and np.abs(gm - rm).max() / np.abs(rm).max() < 2e-4, (np.mean(gm[tm] == rm[tm]), np.abs(gm - rm).max() / np.abs(rm).max())
            assert g[lay["variance"]] == r[lay["variance"]]
    finally:
        gpu.close()
        cpu.close()


@pytest.mark.parametrize("accumulate", [0, 1])
def test_the_sums_do_not_depend_on_the_scatter_kernel(accumulate):
    """RNB_SCATTER_PLAIN=1 sends every level through the one-atomic-per-corner kernel; the default sends levels 0-1 through private LDS tables and the middle levels
    through run-length walks. Different groupings, different atomic orders -- the same integers."""
    a, cpu_a = _staged(accumulate, 1)
    b, cpu_b = _staged(accumulate, 1, env={"RNB_SCATTER_PLAIN": "1"})
    try:
        lay = cpu_a.param_layout()
        for c in (a, b):
            c.forward_backward()
        ga, gb = _grads(a), _grads(b)
        assert ga[lay["grid"]:lay["variance"]].any()
        assert np.array_equal(_bits(ga), _bits(gb))
    finally:
        for c in (a, b, cpu_a, cpu_b):
            c.close()


def _digest(c):
    h = hashlib.sha256()
    for name in ("PARAMS_FP32", "PARAMS_FP16", "PARAMS_EMA", "ADAM_M", "ADAM_V", "ADAM_STEPS", "DENSITY_GRID", "DENSITY_BITFIELD"):
        h.update(np.ascontiguousarray(c.get(name)).tobytes())
    return h.hexdigest()


@pytest.mark.parametrize("accumulate,no_albedo", [(0, 1), (1, 1), (0, 0)])
def test_training_run_is_bit_reproducible_on_any_schedule(accumulate, no_albedo):
    """Three runs of 80 steps from the initial state (occupancy updates at steps 0..., optimizer, controller): overlapped, overlapped again, strictly serial --
    every statistic of every step and the final state (weights, Adam moments and step counts, EMA, occupancy grid) bit for bit."""
    import rnb_neus2_amd as rnb
    from rnb_neus2_amd import synthetic
    scene = synthetic.make_scene(6, 128, 224.0)
    kw = dict(target_batch_size=1 << 14, max_rays_per_batch=1 << 14, initial_rays_per_batch=1024, apply_no_albedo=no_albedo, accumulate=accumulate, deterministic=1)
    out = []
    for overlap in (1, 1, 0):
        c = rnb.Context(overlap=overlap, **kw)
        try:
            c.init_params()
            c.set_dataset(*scene)
            stats = []
            for _ in range(80):
                st = c.train_step().as_dict()
                stats.append(tuple(st[k] for k in ("training_step", "rays_per_batch", "next_rays_per_batch", "measured_batch_size", "measured_batch_size_before_compaction",
                                                   "n_rays_kept", "density_grid_updated", "loss", "ek_loss", "mask_loss")))
            out.append((stats, _digest(c)))
        finally:
            c.close()
    assert out[0][0][-1][7] < out[0][0][0][7]  # it trained
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1]
    assert out[0][0] == out[2][0] and out[0][1] == out[2][1]


def test_default_mode_is_not_reproducible_which_is_why_the_mode_exists():
    """(A measurement, not a requirement: with floating-point atomics two backward passes from one state differ in the last bits of the hash-grid sums -- as the
    reference's do, grid.h:410-430. If this ever stops being so the deterministic mode has lost its reason, not the build its correctness.)"""
    gpu, cpu = _pair(apply_no_albedo=1)
    try:
        _randomize(gpu, cpu, seed=1)
        n_rays = 512
        _stage_samples(gpu, cpu, n_rays, step=700)
        cpu.compute_loss(n_rays, 0)
        gpu.put("DLOSS_DOUT", cpu.get("DLOSS_DOUT"))
        gpu.put("COORDS_COMPACTED", cpu.get("COORDS_COMPACTED"))
        lay = cpu.param_layout()
        differs = 0
        gpu.forward_backward()
        first = gpu.get("GRADS_FP32").copy()
        for _ in range(5):
            gpu.forward_backward()
            g = gpu.get("GRADS_FP32")
            assert np.array_equal(first[:lay["grid"]].view(np.uint32), g[:lay["grid"]].view(np.uint32))  # the MLPs' fixed-order sums are reproducible in every mode
            differs += int((first.view(np.uint32) != g.view(np.uint32)).sum())
        print("default mode: %d of %d hash-grid sums differed in some bit over 5 repeats" % (differs, 5 * (lay["variance"] - lay["grid"])))
    finally:
        gpu.close()
        cpu.close()

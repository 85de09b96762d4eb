"""GPU parity tests of the product mode rnb_config::accumulate = RNB_ACCUM_HALF -- the reference's arithmetic as coded: MLP dot products whose accumulator is rounded
to half after every 16-wide k-step (WMMA half fragments, fully_fused_mlp.cu:59-68, 198), hash-grid gradients summed by atomicAdd(__half2) into a half gradient
vector (grid.h:410-430, trainer.h:78-84) -- against the oracle in the same mode (oracle/rnb_oracle.cpp: dot_h, emulated_dw, half atomics in sample order), stage by
stage at sizes the oracle finishes in seconds. The full-size step is tests/test_gpu_fullsize.py::test_hip_against_the_reference_as_coded_emulation[half].

What can and cannot be bit-exact: a k-step's 16 products are summed in fp32 by the matrix core in an order the oracle's sequential sum need not share, so a dot
product may land on the neighbouring half where the fp32 sums straddle a rounding boundary (a few elements per thousand, one ulp); the order of the half atomics
is the hardware's. Everything else (which entries are touched, the optimizer on a given gradient vector) is exact."""
import numpy as np
import pytest

from tests.test_gpu_parity import _pair, _randomize, _stage_samples

pytestmark = pytest.mark.gpu

HALF = dict(accumulate=1)


def _ulps(a, b):
    """Distance of two half arrays in units of the last place (of the larger magnitude's binade)."""
    a32, b32 = a.astype(np.float32), b.astype(np.float32)
    mag = np.maximum(np.abs(a32), np.abs(b32))
    ulp = np.exp2(np.floor(np.log2(np.maximum(mag, 6.1e-5))) - 10)
    return np.abs(a32 - b32) / ulp


@pytest.fixture(scope="module")
def hpair():
    gpu, cpu = _pair(apply_no_albedo=0, **HALF)
    _randomize(gpu, cpu)
    yield gpu, cpu
    gpu.close()
    cpu.close()


@pytest.fixture(scope="module")
def hpair_no_albedo():
    gpu, cpu = _pair(apply_no_albedo=1, **HALF)
    _randomize(gpu, cpu, seed=1)
    yield gpu, cpu
    gpu.close()
    cpu.close()


def test_buffers_of_the_two_modes(hpair):
    import rnb_neus2_amd as rnb
    gpu, cpu = hpair
    with pytest.raises(rnb.RnbError):
        gpu.get("GRADS_FP32")
    assert gpu.get("GRADS_FP16").dtype == np.float16 and gpu.get("GRADS_FP16").size == gpu.n_params
    plain = rnb.Context(target_batch_size=1 << 13, max_rays_per_batch=1 << 13)
    try:
        with pytest.raises(rnb.RnbError):
            plain.get("GRADS_FP16")
        with pytest.raises(rnb.RnbError):
            plain.update_config(accumulate=1)  # fixed at creation
    finally:
        plain.close()
    with pytest.raises(rnb.RnbError):
        rnb.Context(accumulate=2)


@pytest.mark.parametrize("step", [50, 700])
def test_half_mode_network_evaluation(hpair, step):
    """k_forward_chained_emul / k_point_query_chained_emul against the oracle's dot_h model: every channel within one half ulp of the model's, the channels the
    loss reads (sdf, grad sdf) equal on all but a few per thousand, within a few half ulps everywhere."""
    gpu, cpu = pair = hpair
    rng = np.random.default_rng(step)
    n = 4000 + step % 7
    coords = rng.random((n, 7), dtype=np.float32)
    for c in pair:
        c.set_training_step(step)
    a, b = gpu.forward_infer(coords), cpu.forward_infer(coords)
    assert np.array_equal(a[:, 7:11].view(np.uint16), b[:, 7:11].view(np.uint16))  # variance, direction
    u = _ulps(a[:, 3], b[:, 3])
    # (a hidden unit that lands on the neighbouring half moves the next layer's sums: a couple of ulps at the output, on a few samples per thousand)
    assert u.max() <= 4.0 and np.mean(u == 0) >= 0.995, (u.max(), np.mean(u == 0))
    ug = _ulps(a[:, 4:7], b[:, 4:7])
    # grad sdf = sum_k dsdf_din[k] * dy_dx[k] in fp32 over halfs that may each sit one ulp off: a few ulps of the result where the terms cancel
    assert np.mean(ug <= 1.0) >= 0.99 and np.abs(a[:, 4:7].astype(np.float32) - b[:, 4:7].astype(np.float32)).max() <= 2e-2, (np.mean(ug <= 1.0), ug.max())
    rgb = [0, 1, 2, 11, 12, 13, 14, 15]
    assert np.mean(_ulps(a[:, rgb], b[:, rgb]) <= 2.0) >= 0.98  # raw colour-MLP outputs: three more layers behind the (sdf_out, grad) inputs
    xyz = coords[:, :3]
    s_g, s_c = gpu.sdf(xyz, inference=False), cpu.sdf(xyz, inference=False)
    assert np.array_equal(s_g.view(np.uint16), a[:, 3].view(np.uint16))  # the point query is the forward pass's sdf channel, bit for bit
    assert _ulps(s_g, s_c).max() <= 4.0
    d_g, d_c = gpu.density(xyz), cpu.density(xyz)
    assert np.mean(_ulps(d_g, d_c) <= 1.0) >= 0.99


@pytest.mark.parametrize("no_albedo", [0, 1])
def test_half_mode_forward_backward_gradients(hpair, hpair_no_albedo, no_albedo):
    """k_fwd_bwd_sdf_hs (--no-albedo) and k_rgb_fwd_bwd_hs + k_fwd_bwd_sdf_full_hs, k_dw_sliced + k_dw_finish and the packed-half scatter kernels into GRADS_FP16,
    against the oracle's model from the same loss gradients."""
    gpu, cpu = pair = (hpair_no_albedo if no_albedo else hpair)
    n_rays = 512
    _stage_samples(gpu, cpu, n_rays, step=700)
    cpu.compute_loss(n_rays, 0)
    gpu.put("DLOSS_DOUT", cpu.get("DLOSS_DOUT"))
    gpu.put("COORDS_COMPACTED", cpu.get("COORDS_COMPACTED"))
    for c in pair:
        c.forward_backward()
    g16 = gpu.get("GRADS_FP16")
    g, r = g16.astype(np.float64), cpu.get("GRADS_FP16").astype(np.float64)
    lay = cpu.param_layout()
    for lo, hi, name in ((lay["sdf"], lay["rgb"], "sdf mlp"), (lay["rgb"], lay["grid"], "rgb mlp")):
        if no_albedo and name == "rgb mlp":
            assert not g[lo:hi].any() and not r[lo:hi].any()
            continue
        scale = np.abs(r[lo:hi]).max() + 1e-30
        assert scale > 1e-20, name
        # round 6: the weight gradients are summed in the reference's split-K order (k_dw_sliced, bit-identical to the model on the same operands): what is left is an
        # operand that landed on the neighbouring half in the matrix cores' k-step sums -- measured at this size: every half equal without the colour MLP, 99.95 % with it
        # (the others 7e-7 of the scale apart). Round 5 (fp32 accumulators in the kernels' own tiling, RNB_DW_SLICED=0): 44 % equal, 2.3e-3 of the scale.
        assert np.mean(g[lo:hi] == r[lo:hi]) >= 0.99 and np.abs(g[lo:hi] - r[lo:hi]).max() / scale < 1e-4, (name, np.mean(g[lo:hi] == r[lo:hi]), np.abs(g[lo:hi] - r[lo:hi]).max() / scale)
        cos = float(g[lo:hi] @ r[lo:hi] / (np.linalg.norm(g[lo:hi]) * np.linalg.norm(r[lo:hi])))
        assert cos > 0.99999, (name, cos)
    gg, rg = g[lay["grid"]:lay["variance"]], r[lay["grid"]:lay["variance"]]
    # the same entries are touched (an addend that rounds to half zero is skipped on both sides; a sum may cancel to zero on one side only)
    assert np.mean((gg != 0) != (rg != 0)) < 2e-4
    scale = np.abs(rg).max() + 1e-30
    # per entry: a handful of half roundings apart (the atomics' order and the fp32 run sums), relative to the entry plus a floor of the table's scale
    rel = np.abs(gg - rg) / (np.abs(rg) + 2e-3 * scale)
    assert np.quantile(rel, 0.999) < 1e-2 and np.abs(gg - rg).max() / scale < 4e-3, (np.quantile(rel, 0.999), np.abs(gg - rg).max() / scale)
    cos = float(gg @ rg / (np.linalg.norm(gg) * np.linalg.norm(rg)))
    assert cos > 0.999995, cos
    assert abs(g[lay["variance"]] - r[lay["variance"]]) <= 2.1e-3 * abs(r[lay["variance"]]) + 1e-7  # the fp32 sum on the device, the fp64 sum in the model, each narrowed to half once


def test_sliced_weight_gradients_equal_the_model_bit_for_bit(hpair_no_albedo):
    """k_dw_sliced + the slice sum of k_dw_finish (the half mode's weight-gradient GEMMs in the reference's split-K order: 4096-sample slices, half accumulators rounded
    after every 16-sample k-step, slices reduced in half -- tcnn cutlass_matmul.h:83, 315-322) on given operands against the oracle's emulated_dw: the same bits,
    from sums that are exact in half to an accumulator that overflows. tests/test_oracle_cpu.py holds emulated_dw to an independent numpy statement."""
    from tests import dw_sliced_cases
    gpu, cpu = hpair_no_albedo
    items = dw_sliced_cases.items()
    out, ref = gpu.eval_primitives("DW_SLICED", items), cpu.eval_primitives("DW_SLICED", items)
    assert np.array_equal(out, ref), np.argwhere(out != ref)[:8]
    items2 = dw_sliced_cases.items(seed=7)
    assert np.array_equal(gpu.eval_primitives("DW_SLICED", items2), cpu.eval_primitives("DW_SLICED", items2))


def test_half_mode_optimizer_reads_and_clears_the_half_gradient_vector(hpair):
    gpu, cpu = pair = hpair
    rng = np.random.default_rng(3)
    n = cpu.n_params
    grads = np.zeros(n, dtype=np.float16)
    idx = rng.choice(n, size=200000, replace=False)
    grads[idx] = (rng.standard_normal(idx.size) * 0.05).astype(np.float16)
    grads[:11264] = (rng.standard_normal(11264) * 0.01).astype(np.float16)
    m0 = {name: cpu.get(name).copy() for name in ("PARAMS_FP32", "ADAM_M", "ADAM_V", "ADAM_STEPS", "PARAMS_EMA")}
    for name, v in m0.items():
        gpu.put(name, v)
    for _ in range(2):
        gpu.put("GRADS_FP16", grads)
        cpu.put("GRADS_FP16", grads)
        for c in pair:
            c.optimizer_step()
    for name, tol in (("PARAMS_FP32", 2e-6), ("ADAM_M", 1e-6), ("ADAM_V", 1e-6)):
        np.testing.assert_allclose(gpu.get(name), cpu.get(name), rtol=tol, atol=1e-9, err_msg=name)
    assert np.array_equal(gpu.get("ADAM_STEPS"), cpu.get("ADAM_STEPS"))
    assert not gpu.get("GRADS_FP16").view(np.uint16).any()  # consumed: cleared to +0


def test_half_mode_train_steps_track_the_model():
    """Whole steps (overlapped schedule) in the half mode against the model: the first step's losses to 1e-4 (same weights; the march is exact), counters equal,
    and the run stays with the model's over a few optimizer steps."""
    gpu, cpu = _pair(**HALF)
    try:
        for i in range(6):
            sg, sc = gpu.train_step(), cpu.train_step()
            # (from the second optimizer step on the weights differ by the atomics' order, and with them an occupancy cell at its threshold: a sample more or less)
            slack = 0 if i == 0 else 0.01
            assert abs(int(sg.measured_batch_size_before_compaction) - int(sc.measured_batch_size_before_compaction)) <= slack * sc.measured_batch_size_before_compaction, i
            assert abs(int(sg.n_rays_kept) - int(sc.n_rays_kept)) <= slack * sc.n_rays_kept, i
            tol = 1e-4 if i == 0 else 5e-2
            for k in ("loss", "ek_loss", "mask_loss"):
                a, b = getattr(sg, k), getattr(sc, k)
                assert abs(a - b) <= tol * max(abs(b), 1e-6), (i, k, a, b)
            if i == 0:
                assert sg.measured_batch_size == sc.measured_batch_size
    finally:
        gpu.close()
        cpu.close()


def test_half_mode_is_deterministic_up_to_the_atomics(hpair_no_albedo):
    """Two backward passes from the same inputs: the MLP gradients (fixed-order sums) are bit-identical, the hash grid's differ by the atomics' order only."""
    gpu, cpu = hpair_no_albedo
    lay = cpu.param_layout()
    gpu.forward_backward()
    a = gpu.get("GRADS_FP16").copy()
    gpu.forward_backward()
    b = gpu.get("GRADS_FP16").copy()
    assert np.array_equal(a[:lay["grid"]].view(np.uint16), b[:lay["grid"]].view(np.uint16))
    ga, gb = a[lay["grid"]:lay["variance"]].astype(np.float64), b[lay["grid"]:lay["variance"]].astype(np.float64)
    assert np.array_equal(ga != 0, gb != 0) or np.mean((ga != 0) != (gb != 0)) < 1e-4
    assert np.abs(ga - gb).max() <= 4e-3 * np.abs(ga).max()

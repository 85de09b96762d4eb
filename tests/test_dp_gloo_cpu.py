"""World-size-2 data-parallel path on CPU (gloo): the sharding of the ray index space, the gradient all-reduce and the
counter exchange of rnb-neus2_amd/dp.py, exercised with the CPU checker as the compute engine (test infrastructure)."""
import os
import socket

import numpy as np
import pytest

KW = dict(target_batch_size=1 << 12, max_rays_per_batch=1 << 12, initial_rays_per_batch=256, apply_no_albedo=1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene():
    from rnb_neus2_amd import synthetic
    return synthetic.make_scene(4, 48, 84.0)


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import oracle_lib
    from rnb_neus2_amd import dp
    views, nm, al = _scene()
    c = oracle_lib.context(world_size=world, rank=rank, **KW)
    c.init_params()
    c.set_dataset(views, nm, al)

    def reduce_grads(ctx):  # host buffers: stage through a torch tensor
        t = torch.from_numpy(ctx.get("GRADS_FP32"))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        ctx.put("GRADS_FP32", t.numpy())

    tr = dp.DataParallelTrainer(c, all_reduce_grads=reduce_grads)
    out = {"rank": rank, "steps": []}
    for i in range(3):
        st = tr.step()
        if i == 0:
            kept = int(c.get("COUNTERS")[2])
            out["ray_indices"] = c.get("RAY_INDICES", kept)
            out["rays"] = c.get("RAYS", kept * 6)
        out["steps"].append(st.as_dict())
    out["params"] = c.get("PARAMS_FP32")[:20000].copy()
    out["params_tail"] = c.get("PARAMS_FP32")[-4:].copy()
    out["grid"] = c.get("DENSITY_BITFIELD").copy()
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


class _GlooShardCollectives:
    """reduce_scatter / all_gather of dp.DataParallelTrainer's sharded optimizer over gloo, in place on the checker's host
    buffers (gloo has no reduce-scatter: all-reduce a copy and keep the own chunk, which is what a reduce-scatter leaves)."""

    def __init__(self, ctx, capacity):
        self.ctx, self.capacity = ctx, capacity

    def view(self, name):
        import ctypes as C
        from rnb_neus2_amd import dp
        dt = np.dtype(dp._PARAM_BUFFERS[name])
        ptr, _ = self.ctx.buffer(name)
        return np.frombuffer((C.c_char * (self.capacity * dt.itemsize)).from_address(ptr), dtype=dt)

    def reduce_scatter(self, name, part):
        import torch
        import torch.distributed as dist
        lo, hi, own_lo, own_hi = part
        v = self.view(name)
        t = torch.from_numpy(v[lo:hi].copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        v[own_lo:own_hi] = t.numpy()[own_lo - lo:own_hi - lo]

    def all_gather(self, name, part):
        import torch
        import torch.distributed as dist
        lo, hi, own_lo, own_hi = part
        v = self.view(name)
        mine = torch.from_numpy(v[own_lo:own_hi].copy().view(np.uint8))  # bytes: gloo has no 16-bit integer type
        chunks = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(chunks, mine)
        v[lo:hi] = torch.cat(chunks).numpy().view(v.dtype)


def _worker_sharded(rank, world, port, q):
    try:
        _worker_sharded_body(rank, world, port, q)
    except Exception:  # report instead of leaving the parent waiting for its queue
        import traceback
        q.put({"rank": rank, "error": traceback.format_exc()})


def _worker_sharded_body(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import oracle_lib
    from rnb_neus2_amd import dp
    views, nm, al = _scene()
    ctxs = []
    for _ in range(2):
        c = oracle_lib.context(world_size=world, rank=rank, **KW)
        c.init_params()
        c.set_dataset(views, nm, al)
        ctxs.append(c)
    rep, sh = ctxs

    def reduce_grads(ctx):
        t = torch.from_numpy(ctx.get("GRADS_FP32"))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        ctx.put("GRADS_FP32", t.numpy())

    parts, capacity = sh.shard_layout()
    n_exchanges = [0]

    def grid_max(ptr, n, stream):  # rnb_set_grid_exchange over gloo: element-wise max of the checker's host buffer (uint32 order, as the single-rank atomicMax)
        import ctypes as C
        v = np.frombuffer((C.c_char * (n * 4)).from_address(ptr), dtype=np.int32)
        t = torch.from_numpy(v ^ np.int32(-0x80000000))  # uint32 order of the single-rank atomicMax as a signed max
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        v[:] = t.numpy() ^ np.int32(-0x80000000)
        n_exchanges[0] += 1

    tr_rep = dp.DataParallelTrainer(rep, all_reduce_grads=reduce_grads)  # replicated optimizer AND replicated occupancy updates
    tr_sh = dp.DataParallelTrainer(sh, sharded=True, shard_collectives=_GlooShardCollectives(sh, capacity), grid_exchange=grid_max)
    assert tr_sh.sharded and not tr_rep.sharded and tr_sh.grid_sharded and not tr_rep.grid_sharded
    out = {"rank": rank, "parts": parts, "capacity": capacity, "n_params": sh.n_params, "same_stats": True}
    for i in range(3):
        a, b = tr_rep.step().as_dict(), tr_sh.step().as_dict()
        for k in a:
            if k not in ("prep_ms", "step_ms") and a[k] != b[k]:
                out["same_stats"] = False
    # occupancy updates sharded over the ranks (each evaluates half of an update's samples, element-wise max of the splat targets) leave the grid of the
    # replicated update, bit for bit; the first three steps each begin with an update
    out["grid_exchanges"] = n_exchanges[0]
    out["grid_equal"] = bool(np.array_equal(rep.get("DENSITY_GRID"), sh.get("DENSITY_GRID")) and np.array_equal(rep.get("DENSITY_BITFIELD"), sh.get("DENSITY_BITFIELD")))
    # ... and so does the stage form: begin -> max -> end against one rnb_update_density_grid of the replicated context
    rep.update_density_grid()
    sh.set_grid_exchange(None)
    sh.update_density_grid_begin()
    half = sh.get("DENSITY_GRID_TMP").copy()
    ptr, nb = sh.buffer("DENSITY_GRID_TMP")
    grid_max(ptr, nb // 4, None)
    out["half_is_partial"] = bool(np.count_nonzero(half) < np.count_nonzero(sh.get("DENSITY_GRID_TMP")))
    sh.update_density_grid_end()
    out["grid_equal_stage"] = bool(np.array_equal(rep.get("DENSITY_GRID"), sh.get("DENSITY_GRID")) and np.array_equal(rep.get("DENSITY_GRID_TMP"), sh.get("DENSITY_GRID_TMP")))
    # the training weights are whole on every rank after each step; masters / EMA / Adam state only on the own chunks
    out["w16_equal"] = bool(np.array_equal(rep.get("PARAMS_FP16"), sh.get("PARAMS_FP16")))
    own = np.zeros(sh.n_params, dtype=bool)
    for lo, hi, own_lo, own_hi in parts:
        own[own_lo:min(own_hi, sh.n_params)] = True
    out["own_fraction"] = float(own.mean())
    out["own_equal"] = {n: bool(np.array_equal(rep.get(n)[own], sh.get(n)[own])) for n in ("PARAMS_FP32", "PARAMS_EMA", "ADAM_M", "ADAM_V", "ADAM_STEPS")}
    out["foreign_stale"] = bool(not np.array_equal(rep.get("ADAM_STEPS")[~own], sh.get("ADAM_STEPS")[~own]))
    tr_sh.sync_parameters()
    out["synced_equal"] = {n: bool(np.array_equal(rep.get(n), sh.get(n))) for n in ("PARAMS_FP32", "PARAMS_EMA", "ADAM_M", "ADAM_V", "ADAM_STEPS")}
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_rank_sharded_optimizer_equals_replicated():
    """Reduce-scatter -> Adam on the own chunks -> all-gather of the fp16 weights gives, on every rank, the training weights of
    the all-reduce + replicated optimizer bit for bit; after sync_parameters() also the masters, EMA weights and Adam state."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sharded, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=800) for _ in procs], key=lambda r: r["rank"])
    for r in res:
        assert "error" not in r, r["error"]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in res:
        assert r["same_stats"] and r["w16_equal"]
        assert r["grid_exchanges"] == 3 and r["grid_equal"] and r["grid_equal_stage"] and r["half_is_partial"], r
        assert all(r["own_equal"].values()), r["own_equal"]
        assert all(r["synced_equal"].values()), r["synced_equal"]
        assert r["foreign_stale"]  # the other rank's chunks really were skipped here
        assert 0.45 < r["own_fraction"] < 0.55
        assert r["capacity"] >= r["n_params"] and r["capacity"] % 8 == 0
    # the two ranks' chunks tile every block without overlap
    for pa, pb in zip(res[0]["parts"], res[1]["parts"]):
        assert pa[0] == pb[0] and pa[1] == pb[1]
        assert pa[2] == pa[0] and pa[3] == pb[2] and pb[3] == pb[1] and (pa[3] - pa[2]) == (pb[3] - pb[2]) and (pa[3] - pa[2]) % 4 == 0
    assert res[0]["parts"][0][0] == 0 and res[0]["parts"][-1][1] == res[0]["capacity"]
    p = res[0]["parts"]  # MLPs + coarse and middle levels | first half of the fine levels | second half + variance
    assert len(p) == 3 and p[0][1] == p[1][0] and p[1][1] == p[2][0]


@pytest.mark.timeout(600)
def test_two_rank_data_parallel_step():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=500) for _ in procs], key=lambda r: r["rank"])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    r0, r1 = res
    # replicas stay identical: same parameters, same occupancy grid, same controller decisions
    assert np.array_equal(r0["params"], r1["params"]) and np.array_equal(r0["params_tail"], r1["params_tail"])
    assert np.array_equal(r0["grid"], r1["grid"])
    for a, b in zip(r0["steps"], r1["steps"]):
        for k in ("training_step", "rays_per_batch", "next_rays_per_batch", "measured_batch_size", "loss", "mask_loss", "ek_loss"):
            assert a[k] == b[k], k
    # the union of the two ranks' rays at step 0 is the ray set of ONE process marching world*R rays (same RNG positions)
    from tests import oracle_lib
    views, nm, al = _scene()
    big = dict(KW)
    big["target_batch_size"] = 2 * KW["target_batch_size"]  # one process needs the sample budget of both ranks
    single = oracle_lib.context(**big)
    single.init_params()
    single.set_dataset(views, nm, al)
    single.set_training_step(0)
    single.update_density_grid()
    R = r0["steps"][0]["rays_per_batch"]
    single.generate_training_samples(2 * R, 0)
    kept = int(single.get("COUNTERS")[2])
    idx = single.get("RAY_INDICES", kept)
    rays = single.get("RAYS", kept * 6).reshape(kept, 6)
    union_idx = np.concatenate([r0["ray_indices"], r1["ray_indices"] + R])
    union_rays = np.concatenate([r0["rays"].reshape(-1, 6), r1["rays"].reshape(-1, 6)])
    assert np.array_equal(union_idx, idx)
    assert np.array_equal(union_rays.view(np.uint32), rays.view(np.uint32))
    single.close()


def test_single_rank_trainer_equals_train_step():
    from tests import oracle_lib
    from rnb_neus2_amd import dp
    views, nm, al = _scene()
    a = oracle_lib.context(**KW)
    b = oracle_lib.context(**KW)
    for c in (a, b):
        c.init_params()
        c.set_dataset(views, nm, al)
    tr = dp.DataParallelTrainer(a)
    for _ in range(2):
        sa, sb = tr.step(), b.train_step()
        da, db = sa.as_dict(), sb.as_dict()
        for k in da:
            if k not in ("prep_ms", "step_ms"):
                assert da[k] == db[k], k
    assert np.array_equal(a.get("PARAMS_FP32"), b.get("PARAMS_FP32"))
    a.close()
    b.close()


# few enough rays that nobody's compacted batch overflows at step 0: every comparison below is then unconditional
KW_STRONG = dict(KW, target_batch_size=1 << 14, initial_rays_per_batch=64)


def _worker_strong(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests import oracle_lib
        from rnb_neus2_amd import dp
        views, nm, al = _scene()
        kw = dict(KW_STRONG)
        kw.update(dp.strong_scaling_sizes(world, KW_STRONG["target_batch_size"], KW_STRONG["max_rays_per_batch"], KW_STRONG["initial_rays_per_batch"]))
        c = oracle_lib.context(world_size=world, rank=rank, **kw)
        c.init_params()
        c.set_dataset(views, nm, al)

        def reduce_grads(ctx):
            t = torch.from_numpy(ctx.get("GRADS_FP32"))
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            ctx.put("GRADS_FP32", t.numpy())

        tr = dp.DataParallelTrainer(c, all_reduce_grads=reduce_grads)
        st = tr.step()
        kept = int(c.get("COUNTERS")[2])
        n_comp = int(c.get("COUNTERS")[1])
        q.put({"rank": rank, "stats": st.as_dict(), "ray_indices": c.get("RAY_INDICES", kept), "rays": c.get("RAYS", kept * 6), "n_comp": n_comp,
               "coords": c.get("COORDS_COMPACTED", min(n_comp, kw["target_batch_size"]) * 7), "loss": c.get("LOSS", st.rays_per_batch), "sizes": kw})
    except Exception:
        import traceback
        q.put({"rank": rank, "error": traceback.format_exc()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_strong_scaling_two_ranks_run_the_single_gpu_step():
    """dp.strong_scaling_sizes: two ranks with B/2 samples and R/2 rays each run the step ONE process runs with B and R -- the same
    rays (PCG32 positions, image assignment), the same compacted samples, the same per-ray losses (scaled by the global ray
    count), the same counters after the exchange; the controller then keeps the job at B samples per step."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_strong, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=500) for _ in procs], key=lambda r: r["rank"])
    for p in procs:
        p.join(60)
    for r in res:
        assert "error" not in r, r["error"]
    r0, r1 = res
    assert r0["sizes"]["target_batch_size"] == KW_STRONG["target_batch_size"] // 2 and r0["sizes"]["initial_rays_per_batch"] == KW_STRONG["initial_rays_per_batch"] // 2
    from tests import oracle_lib
    views, nm, al = _scene()
    single = oracle_lib.context(**KW_STRONG)
    single.init_params()
    single.set_dataset(views, nm, al)
    st = single.train_step()
    R = st.rays_per_batch
    assert r0["stats"]["rays_per_batch"] == r1["stats"]["rays_per_batch"] == R // 2
    kept = int(single.get("COUNTERS")[2])
    union_idx = np.concatenate([r0["ray_indices"], r1["ray_indices"] + R // 2])
    assert np.array_equal(union_idx, single.get("RAY_INDICES", kept))
    assert np.array_equal(np.concatenate([r0["rays"], r1["rays"]]).view(np.uint32), single.get("RAYS", kept * 6).view(np.uint32))
    # compacted samples: rank 0's then rank 1's = the single process's, as long as nobody ran out of room
    n_comp = int(single.get("COUNTERS")[1])
    assert r0["n_comp"] + r1["n_comp"] == n_comp
    assert 0 < n_comp <= KW_STRONG["target_batch_size"] and max(r0["n_comp"], r1["n_comp"]) <= KW_STRONG["target_batch_size"] // 2, (n_comp, r0["n_comp"], r1["n_comp"])
    both = np.concatenate([r0["coords"], r1["coords"]])
    assert np.array_equal(both.view(np.uint32), single.get("COORDS_COMPACTED", n_comp * 7).view(np.uint32))
    # per-ray losses carry the global 1 / R scale
    # (the per-ray loss rows are indexed by kept ray)
    got, want = np.concatenate([r0["loss"][:len(r0["ray_indices"])], r1["loss"][:len(r1["ray_indices"])]]), single.get("LOSS", kept)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (bad, got[bad], want[bad])
    # exchanged statistics: the job's counters are the single process's
    for k in ("measured_batch_size_before_compaction", "measured_batch_size", "n_rays_kept"):
        assert r0["stats"][k] == r1["stats"][k]
        assert abs(2 * r0["stats"][k] - getattr(st, k)) <= 1, k  # stats report per-rank means
    for k in ("loss", "ek_loss", "mask_loss"):
        assert abs(r0["stats"][k] - getattr(st, k)) <= 2e-6 * abs(getattr(st, k)) + 1e-12, k
    assert abs(2 * r0["stats"]["next_rays_per_batch"] - st.next_rays_per_batch) <= 256  # each rank rounds its share up to a multiple of 128
    single.close()

"""Two data-parallel ranks as two PROCESSES sharing the one GPU of the test box, the HIP library as the engine and gloo as the
transport (RCCL refuses two ranks on one device): dp.DataParallelTrainer's sharded and all-reduce paths with a world of 2 —
block 0 on its side stream behind rnb_gradient_part_wait, Adam on the own chunks, the fp16 weights gathered — must leave both
ranks with the same weights, and the sharded optimizer must agree with the replicated one."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# rnb_config::deterministic: the hash-grid sums are exact integers and a two-rank sum a + b commutes, so the sharded and the replicated optimizer must agree BIT FOR BIT on every step
# (rounds 2-5 compared two chaotic trajectories with tolerances)
KW = dict(target_batch_size=1 << 14, max_rays_per_batch=1 << 14, initial_rays_per_batch=1024, apply_no_albedo=1, deterministic=1)


class _GlooDeviceShardCollectives:
    """reduce_scatter / all_gather for the trainer over gloo, in place on the library's DEVICE buffers, stream-ordered on
    torch's current stream (staged through the host: gloo reduces CUDA tensors but does not gather them)."""

    on_device = True

    def __init__(self, ctx, capacity):
        from rnb_neus2_amd import dp
        self._t = dp.TorchShardCollectives(ctx, capacity)  # only for its no-copy tensor views

    def reduce_scatter(self, name, part):
        import torch.distributed as dist
        lo, hi, own_lo, own_hi = part
        v = self._t.view(name)
        t = v[lo:hi].cpu()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        v[own_lo:own_hi].copy_(t[own_lo - lo:own_hi - lo])  # what a reduce-scatter leaves: the sum in the own chunk only

    def all_gather(self, name, part):
        import torch
        import torch.distributed as dist
        lo, hi, own_lo, own_hi = part
        v = self._t.view(name)
        mine = v[own_lo:own_hi].cpu().view(torch.uint8)
        chunks = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(chunks, mine)
        v[lo:hi].copy_(torch.cat(chunks).view(v.dtype))


def _worker(rank, world, port, q):
    try:
        import torch
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import rnb_neus2_amd as rnb
        from rnb_neus2_amd import dp, synthetic
        scene = synthetic.make_scene(8, 128, 224.0)
        ctxs = []
        for _ in range(2):
            c = rnb.Context(world_size=world, rank=rank, **KW)
            c.init_params()
            c.set_dataset(*scene)
            ctxs.append(c)
        sh, rep = ctxs

        def reduce_grads(ctx):  # all-reduce path: gloo sums the device tensor in place
            dist.all_reduce(dp.grads_tensor(ctx), op=dist.ReduceOp.SUM)

        parts, capacity = sh.shard_layout()
        tr_sh = dp.DataParallelTrainer(sh, sharded=True, shard_collectives=_GlooDeviceShardCollectives(sh, capacity))
        tr_rep = dp.DataParallelTrainer(rep, all_reduce_grads=reduce_grads)
        out = {"rank": rank, "parts": parts, "n_params": sh.n_params, "steps": []}
        for i in range(40):
            a, b = tr_sh.step(), tr_rep.step()
            out["steps"].append((a.as_dict(), b.as_dict()))
            if i == 0:  # same rays, same losses, the same update
                torch.cuda.synchronize()
                out["first_w16_absdiff"] = float(np.abs(sh.get("PARAMS_FP16").astype(np.float32) - rep.get("PARAMS_FP16").astype(np.float32)).max())
        torch.cuda.synchronize()
        out["w16"] = sh.get("PARAMS_FP16").copy()
        out["w16_rep"] = rep.get("PARAMS_FP16").copy()
        own = np.zeros(sh.n_params, dtype=bool)
        for lo, hi, own_lo, own_hi in parts:
            own[own_lo:min(own_hi, sh.n_params)] = True
        steps_before = sh.get("ADAM_STEPS").copy()
        tr_sh.sync_parameters()
        torch.cuda.synchronize()
        out["own_fraction"] = float(own.mean())
        out["foreign_steps_before_sync"] = int(steps_before[~own].max())
        out["adam_steps"] = sh.get("ADAM_STEPS").copy()
        out["w32"] = sh.get("PARAMS_FP32").copy()
        out["grads_clear"] = bool(not sh.get("GRADS_FP32").any())
        out["bitfield"] = sh.get("DENSITY_BITFIELD").copy()
        q.put(out)
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put({"rank": rank, "error": traceback.format_exc()})


@pytest.mark.timeout(900)
def test_two_processes_share_the_gpu():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=800) for _ in procs], key=lambda r: r["rank"])
    for r in res:
        assert "error" not in r, r["error"]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    r0, r1 = res
    # both ranks hold the same training weights (same gathered bytes), masters and Adam state after the sync, and occupancy grid
    assert np.array_equal(r0["w16"].view(np.uint16), r1["w16"].view(np.uint16))
    assert np.array_equal(r0["w32"].view(np.uint32), r1["w32"].view(np.uint32))
    assert np.array_equal(r0["adam_steps"], r1["adam_steps"]) and r0["adam_steps"].max() == 40
    assert np.array_equal(r0["bitfield"], r1["bitfield"])
    assert np.array_equal(r0["w16_rep"].view(np.uint16), r1["w16_rep"].view(np.uint16))
    for r in res:
        assert r["grads_clear"] and 0.45 < r["own_fraction"] < 0.55
        assert r["foreign_steps_before_sync"] == 0       # the other rank's chunks were never stepped here
        assert r["first_w16_absdiff"] == 0.0
        for a, b in r["steps"]:  # sharded == replicated on every step, every statistic
            for k in ("training_step", "rays_per_batch", "next_rays_per_batch", "measured_batch_size", "measured_batch_size_before_compaction", "n_rays_kept", "loss", "mask_loss", "ek_loss"):
                assert a[k] == b[k], k
        assert r["steps"][-1][0]["training_step"] == 40
    for (a0, _), (a1, _) in zip(r0["steps"], r1["steps"]):  # the ranks agree on every controller decision
        for k in ("training_step", "rays_per_batch", "next_rays_per_batch", "measured_batch_size", "loss"):
            assert a0[k] == a1[k], k
    # training went somewhere and the two optimizers are ONE trajectory
    assert np.array_equal(r0["w16"].view(np.uint16), r0["w16_rep"].view(np.uint16))
    assert r0["steps"][-1][0]["loss"] < r0["steps"][0][0]["loss"]

"""Worker of tests/test_gpu_rccl.py: ONE python process = ONE init_process_group("nccl") -- RCCL is initialised once per process and torn down with
it (round 5 did init -> destroy -> init inside the pytest process). `python -m tests.rccl_single_rank_worker <accumulate> <deterministic>`; exit code 0 = every assert held.
deterministic = 1 (rnb_config::deterministic): the three trainers -- plain step, sharded optimizer over reduce_scatter / all_gather, all-reduce + replicated optimizer -- must
stay BIT-IDENTICAL over every step (the sums are exact integers; a collective over one rank is the identity); deterministic = 0 (the default product mode, floating-point
atomics): the first step from a common state within the atomics' noise, then 20 steps that must run and stay finite -- three chaotic trajectories are not compared step by step
(round 5 did, with tolerances that a fresh box broke)."""
import os
import socket
import sys
import time

import numpy as np


def main(accumulate, deterministic):
    import torch
    import torch.distributed as dist
    from rnb_neus2_amd import dp, synthetic
    from tests.test_gpu_fullsize import KW, WINDOW_STEP, _clone, _state_of
    import rnb_neus2_amd as rnb

    t_phase = [time.perf_counter()]

    def phase(name):
        t_phase.append(time.perf_counter())
        print("[rccl single rank] %s: %.2f s" % (name, t_phase[-1] - t_phase[-2]), flush=True)

    scene = synthetic.make_scene(64, 800)
    ctx = rnb.Context(overlap=0, **KW)
    ctx.init_params()
    ctx.set_dataset(*scene)
    st = None
    for _ in range(WINDOW_STEP):
        st = ctx.train_step()
    state = _state_of(ctx, st)
    ctx.close()
    phase("scene + %d steps" % WINDOW_STEP)

    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    os.environ["RNB_DP_FORCE_COLLECTIVES"] = "1"
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    phase("init_process_group")
    mode = dict(accumulate=accumulate, deterministic=deterministic)
    plain = _clone(scene, state, overlap=1, **mode)
    ctxs = [_clone(scene, state, overlap=1, **mode), _clone(scene, state, overlap=1, **mode)]  # created with the variable set: data-parallel scatter order
    phase("three clones")
    try:
        trainers = [dp.DataParallelTrainer(ctxs[0], sharded=True), dp.DataParallelTrainer(ctxs[1], sharded=False)]
        assert trainers[0].sharded and not trainers[1].sharded and all(t._collectives for t in trainers)
        assert all(t.grid_sharded for t in trainers)  # the occupancy update's exchange (a max all-reduce over RCCL inside train_step_begin) is registered
        ref = plain.train_step()
        got = [t.step() for t in trainers]
        torch.cuda.synchronize()
        phase("first step x 3")
        assert ref.density_grid_updated and all(st.density_grid_updated for st in got)  # the first step began with an occupancy update: through the exchange, same grid
        for c in ctxs:
            assert np.array_equal(plain.get("DENSITY_GRID").view(np.uint32), c.get("DENSITY_GRID").view(np.uint32))
            assert np.array_equal(plain.get("DENSITY_BITFIELD"), c.get("DENSITY_BITFIELD"))
        for t in trainers:
            t.sync_parameters()
        torch.cuda.synchronize()
        phase("sync_parameters")
        pa = plain.get("PARAMS_FP32")
        gname, gview = ("GRADS_FP16", np.uint16) if accumulate else ("GRADS_FP32", np.uint32)
        for st, c in zip(got, ctxs):  # first step from a common state: identical statistics
            assert st.training_step == ref.training_step and st.loss == ref.loss and st.next_rays_per_batch == ref.next_rays_per_batch
            assert not c.get(gname).view(gview).any()
            if deterministic:
                continue
            d = np.abs(pa - c.get("PARAMS_FP32"))
            # same update up to the order of the atomics: a gradient that rounds to +-tiny is one Adam step of lr either way (half mode: the sums themselves depend on the order of the half atomics: more such entries)
            assert d.max() <= 2.5e-3 and np.mean(d > 2e-5) < (1e-3 if accumulate else 1e-5), (float(d.max()), float(np.mean(d > 2e-5)))
            if not accumulate:
                assert np.array_equal(plain.get("ADAM_STEPS"), c.get("ADAM_STEPS"))
            else:
                assert np.mean(plain.get("ADAM_STEPS") != c.get("ADAM_STEPS")) < 1e-3  # (a half sum that cancels to zero on one side only is not stepped there)
        history = []
        keys = ("training_step", "rays_per_batch", "next_rays_per_batch", "measured_batch_size", "measured_batch_size_before_compaction", "n_rays_kept", "density_grid_updated", "loss", "ek_loss", "mask_loss")
        for _ in range(20):
            ref = plain.train_step()
            got = [t.step() for t in trainers]
            history.append([tuple(getattr(st, k) for k in keys) for st in [ref] + got])
        phase("20 steps x 3")
        for t in trainers:
            t.sync_parameters()
        torch.cuda.synchronize()
        for st in got:
            assert st.training_step == ref.training_step and np.isfinite(st.loss)
        if deterministic:
            for i, (a, b, c) in enumerate(history):
                assert a == b == c, (i, a, b, c)
            for name in ("PARAMS_FP32", "PARAMS_FP16", "PARAMS_EMA", "ADAM_M", "ADAM_V", "ADAM_STEPS", "DENSITY_GRID", "DENSITY_BITFIELD"):
                x = np.ascontiguousarray(plain.get(name)).view(np.uint8)
                for c in ctxs:
                    assert np.array_equal(x, np.ascontiguousarray(c.get(name)).view(np.uint8)), name
            print("deterministic: plain == sharded == all-reduce over 21 steps, every statistic and every byte of the state", flush=True)
        else:
            h = np.array([[row[7] for row in step] for step in history])
            assert abs(h[:, 1].mean() - h[:, 0].mean()) <= 0.3 * h[:, 0].mean() and abs(h[:, 2].mean() - h[:, 0].mean()) <= 0.3 * h[:, 0].mean(), h.tolist()
    finally:
        plain.close()
        for c in ctxs:
            c.close()
        dist.destroy_process_group()
        phase("destroy_process_group")
    print("RCCL_SINGLE_RANK_OK", flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]))

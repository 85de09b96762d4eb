#!/usr/bin/env python3
"""bench.py — training rays/s of the normals-only SDF path on synthetic 64-view 800x800 normal+mask data.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

One "step" = one pass of the hot path over one batch: Testbed::train (occupancy update when due, ray generation +
march, network forward on the un-compacted samples, loss + compaction, forward/backward on 2^18 compacted samples,
Adam+EMA). Inputs are resident in HBM before the timed region. Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

# more hardware queues than HIP's default 4: the library's side streams, torch's and RCCL's must not be multiplexed onto the
# queue that carries the step's critical path (set before the HIP runtime starts)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
PMC_TRAFFIC, PMC_UNITS = "r02_pmc_traffic.json", "r02_pmc_units.json"  # summaries of the separate rocprofv3 --pmc passes (tools/collect_pmc.sh)
# Algorithmic bytes per unit (restated in DESIGN.md §measurement)
ALGO_BYTES = {
    "k_forward": 508.0,            # 448 B gathers + 28 B coords + 32 B out, per un-compacted sample
    "k_point_query": 462.0,        # 448 B gathers + 12 B position + 2 B density, per occupancy sample
    "k_fwd_bwd": 508.0,            # gathers + coords + dL/dout, per compacted sample (scratch traffic is overhead)
    "k_grid_scatter": 896.0,       # first-order + second-order scatter (RMW counted once each), per compacted sample
    "k_adam_ema": 10.0,            # >= grad 4 B + fp16 weight 2 B + EMA 2+2 B per parameter (dead entries)
    "k_march_count": 430.0,        # per RAY: view record 100 B + pixel 8 B + ~150 occupancy tests (1 B each) + setup/steps 48 B out + 4 B per marched sample (~30)
}
# what actually bounds a kernel whose HBM fraction is small by construction
LIMITER_NOTES = {
    "k_march_count": "divergent ALU work and dependent occupancy tests of a sequential per-ray walk (bit-exact replay of the reference's march); not a bandwidth kernel",
    "k_forward": "L1 lane-address rate of 112 scattered 8-byte gathers per sample",
    "k_fwd_bwd": "L1 lane-address rate of the gathers + operand stores",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000, help="timed steps; default = the window SURVEY.md section 8(d) defines: training steps 1000-2000")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--burn-in", type=int, default=980, help="untimed training steps before warmup so that the measured steps sit in the "
                    "regime the metric is quoted on (burn-in + warmup = 1000: all 14 levels live, occupancy converged; SURVEY.md §8d)")
    ap.add_argument("--views", type=int, default=64)
    ap.add_argument("--res", type=int, default=800)
    ap.add_argument("--cpu-baseline-steps", type=int, default=8, help="steps of the CPU checker timed as the baseline (≈1.4 s each on the GPU box host)")
    ap.add_argument("--profile-steps", type=int, default=100, help="serialized steps after the timed region for the per-kernel HIP-event table")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--window-end", type=int, default=2000, help="after the K timed steps the run continues (timed per step) to this training step, so that "
                    "a short --steps still reports the whole window SURVEY.md 8(d) defines (steps 1000-2000); 0 = off")
    ap.add_argument("--late-step", type=int, default=6000, help="then train on to this step and time --late-steps more: the converged regime (few samples "
                    "per ray, ~95 k rays per step) that the >= 1e8 rays/s target is about; 0 = off")
    ap.add_argument("--late-steps", type=int, default=200)
    ap.add_argument("--strong", action="store_true", help="strong scaling (SURVEY.md 8e): the job's step stays the single-GPU step (2^18 compacted samples, the "
                    "controller's ray count); every rank takes 1/N of its rays and samples (dp.strong_scaling_sizes). Default: weak scaling, 2^18 samples per rank")
    ap.add_argument("--albedo", action="store_true", help="secondary workload: stage 2 of the two-stage pipeline (colour MLP + reflectance loss live) instead of "
                    "the normals-only path the metric is quoted on")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import rnb_neus2_amd as rnb
    from rnb_neus2_amd import synthetic, dp

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or (os.environ.get("RNB_DP_FORCE_COLLECTIVES") and "MASTER_ADDR" in os.environ):  # the env var exercises the RCCL path on one rank
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    # stage 1 of run_two_stage: --mask-weight 1.0 --no-albedo (rnb_neus2/pipeline.py:63-74)
    sizes = dp.strong_scaling_sizes(world) if args.strong else {}
    ctx = rnb.Context(apply_no_albedo=0 if args.albedo else 1, mask_loss_weight=1.0, world_size=world, rank=rank, overlap=0 if os.environ.get("RNB_OVERLAP_OFF") else 1, **sizes)
    ctx.init_params()
    t0 = time.time()
    views, normals, albedos = synthetic.make_scene(args.views, args.res)
    ctx.set_dataset(views, normals, albedos)
    del normals, albedos
    setup_s = time.time() - t0
    trainer = dp.DataParallelTrainer(ctx)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.burn_in):
        trainer.step()
    for _ in range(args.warmup):
        trainer.step()
    barrier()
    t0 = time.perf_counter()
    rays = 0
    samples = 0
    samples_before = 0
    last = None
    for _ in range(args.steps):
        last = trainer.step()
        rays += last.rays_per_batch * world
        samples += last.measured_batch_size * world
        samples_before += last.measured_batch_size_before_compaction * world
    barrier()
    elapsed = time.perf_counter() - t0

    def timed_run(n_steps):
        """n_steps more training steps: (wall seconds, rays, compacted samples, per-step host milliseconds, last stats)."""
        barrier()
        tt0 = time.perf_counter()
        r = smp = 0
        per_step = []
        stl = None
        tp = tt0
        for _ in range(n_steps):
            stl = trainer.step()
            tn = time.perf_counter()
            per_step.append(1e3 * (tn - tp))
            tp = tn
            r += stl.rays_per_batch * world
            smp += stl.measured_batch_size * world
        barrier()
        return time.perf_counter() - tt0, r, smp, per_step, stl

    # The rest of the window the metric is defined on (training steps 1000-2000), so that the record carries it
    # even when --steps is small: the K steps above plus the steps up to --window-end.
    window = None
    first_timed = int(last.training_step) - args.steps
    n_more = args.window_end - int(last.training_step) if args.window_end else 0
    if n_more > 0:
        w_el, w_rays, w_smp, w_ms, last_w = timed_run(n_more)
        n_w = args.steps + n_more
        window = {"first_step": first_timed, "steps": n_w, "ms_per_step": round(1e3 * (elapsed + w_el) / n_w, 4), "rays_per_s": round((rays + w_rays) / (elapsed + w_el), 1),
                  "p50_ms_per_step": round(float(np.median(w_ms)), 4), "p90_ms_per_step": round(float(np.quantile(w_ms, 0.9)), 4),
                  "rays_per_step_first": int(last.rays_per_batch), "rays_per_step_last": int(last_w.rays_per_batch),
                  "samples_per_s_compacted": round((samples + w_smp) / (elapsed + w_el), 1)}
        last = last_w
    # Per-kernel durations: HIP events on the step's stream, taken in a second pass over the same workload right after the
    # timed region. In the timed region the next step's march and the weight-gradient GEMMs run on side streams beside the
    # backward pass (cfg.overlap), where a per-kernel duration is not well defined; with the profiler on the library runs the
    # same kernels strictly one after the other.
    ctx.profile_enable(True)
    tail = last
    for _ in range(args.profile_steps):
        tail = trainer.step()
    barrier()
    prof = ctx.profile()
    ctx.profile_enable(False)
    late = None
    if args.late_step and args.late_steps > 0:
        while ctx.training_step < args.late_step:
            tail = trainer.step()
        l_el, l_rays, l_smp, l_ms, tail = timed_run(args.late_steps)
        late = {"first_step": int(tail.training_step) - args.late_steps, "steps": args.late_steps, "ms_per_step": round(1e3 * l_el / args.late_steps, 4),
                "rays_per_s": round(l_rays / l_el, 1), "rays_per_step": round(l_rays / args.late_steps / world, 1), "p50_ms_per_step": round(float(np.median(l_ms)), 4),
                "samples_per_s_compacted": round(l_smp / l_el, 1), "loss": round(float(tail.loss), 6)}
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    result = None
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = rays / elapsed
        # roofline of the dominant kernel (by accumulated HIP-event time of the per-kernel pass), and of the two next ones
        def roofline_of(p):
            kname = p["kernel"]
            bytes_per_unit = ALGO_BYTES.get(kname)
            if not p["launches"] or bytes_per_unit is None or p["units"] <= 0:
                return None
            avg_ms = p["total_ms"] / p["launches"]
            units_per_launch = p["units"] / p["launches"]
            achieved = bytes_per_unit * units_per_launch / (avg_ms * 1e-3) / 1e9
            # HBM bytes per launch: NOT measured in this process (a PMC pass cannot run inside it) but read from the committed summary of
            # separate rocprofv3 --pmc passes over the same command (tools/collect_pmc.sh); `traffic_source` says so in the record
            traffic, traffic_source = None, None
            try:
                with open(os.path.join(ROOT, "profiles", PMC_TRAFFIC)) as f:
                    traffic = json.load(f)["per_step"][kname]["total_bytes"] / max(p["launches"] / max(args.profile_steps, 1), 1.0)
                traffic_source = "committed file profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, steps 2000-2010, builder-run), not this run" % PMC_TRAFFIC
            except Exception:
                pass
            limiter = None  # the unit that actually bounds the kernel when it is not HBM bytes (tools/pmc_report.py); also from a committed file
            try:
                with open(os.path.join(ROOT, "profiles", PMC_UNITS)) as f:
                    units = json.load(f)
                if kname == "k_grid_scatter":
                    parts = [units["kernels"][k] for k in ("k_grid_scatter_quad", "k_grid_scatter_quad_rl", "k_grid_scatter_lds")]
                    req = sum(q.get("l2_atomic_requests", 0) for q in parts)
                    limiter = {"unit": "L2 atomic lines (one 64-byte line of one instruction)", "per_launch": req, "achieved_per_s": round(req / (avg_ms * 1e-3)),
                               "probe_rate_per_s": units["_atomic_probe_requests_per_s"],
                               "frac": round(req / (avg_ms * 1e-3) / units["_atomic_probe_requests_per_s"], 3),
                               "limiter_source": "per_launch: committed file profiles/%s (TCC_ATOMIC_sum pass at steps 2000-2010, builder-run; the duration is this run's, whose "
                                                 "regime may put more lines on the path); probe rate: tools/probe_atomics4.hip" % PMC_UNITS}
            except Exception:
                pass
            if limiter is None and kname in LIMITER_NOTES:
                limiter = {"unit": LIMITER_NOTES[kname]}
            return {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source, "avg_launch_ms": round(avg_ms, 4),
                    "units_per_launch": round(units_per_launch, 1), "algorithmic_bytes_per_unit": bytes_per_unit, "limiter": limiter}

        ranked = sorted(prof, key=lambda p: -p["total_ms"])
        roofline = roofline_of(ranked[0]) if ranked else None
        rooflines_next = [r for r in (roofline_of(p) for p in ranked[1:4]) if r is not None]
        kernels = {p["kernel"]: {"ms_per_step": round(p["total_ms"] / max(args.profile_steps, 1), 4), "launches": p["launches"]} for p in prof if p["launches"]}
        result = {
            "metric": "training rays/s + ms/step, normals-only SDF 64x800^2",
            "value": round(value, 1),
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "f16 storage / f32 accumulate",
            "data": "synthetic",
            "config": {"workload": "config 4: synthetic %d-view %dx%d normals+mask sphere, %s --mask-weight 1.0, %s compacted samples/step/GPU"
                                   % (args.views, args.res, args.res, "albedo + reflectance loss (NOT the metric's workload)" if args.albedo else "--no-albedo",
                                      ("2^18 / %d" % world) if args.strong else "2^18"),
                       "burn_in_steps": args.burn_in, "first_timed_step": first_timed,
                       "rays_per_step_per_gpu": round(rays / args.steps / world, 1),
                       "samples_per_s_compacted": round(samples / elapsed, 1),
                       "samples_per_s_before_compaction": round(samples_before / elapsed, 1),
                       "loss": round(float(last.loss), 6), "parallelism": "dp%d" % world, "setup_s": round(setup_s, 1)},
            "window_1000_2000": window,
            "late_regime": late,
            "roofline": roofline,
            "rooflines_next": rooflines_next,
            "kernels_ms_per_step": kernels,
        }

    # CPU baseline: the oracle (a port, not the reference — the reference has no CPU path) continues from the GPU's
    # trained state on the same workload, all host cores, a bounded number of steps.
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.cpu_baseline_steps > 0:
        try:
            from tests import oracle_lib
            cpu = oracle_lib.context(apply_no_albedo=0 if args.albedo else 1, mask_loss_weight=1.0)
            views2, normals2, albedos2 = synthetic.make_scene(args.views, args.res)
            cpu.set_dataset(views2, normals2, albedos2)
            del normals2, albedos2
            cpu.set_params(ctx.get("PARAMS_FP32"))
            cpu.put("DENSITY_GRID", ctx.get("DENSITY_GRID"))
            cpu.update_density_bitfield()
            cpu.set_controller(ctx.training_step, ctx.rays_per_batch, tail.measured_batch_size_before_compaction, 0)
            t0 = time.perf_counter()
            crays = 0
            for _ in range(args.cpu_baseline_steps):
                st = cpu.train_step()
                crays += st.rays_per_batch
            cel = time.perf_counter() - t0
            result["cpu_baseline"] = {"value": round(crays / cel, 1), "unit": "rays/s", "cores": os.cpu_count(), "kind": "port",
                                      "sample": "%d training steps of the CPU oracle (OpenMP, all host cores) continued from the GPU's state at step %d, same dataset/flags; %.1f s"
                                                % (args.cpu_baseline_steps, ctx.training_step, cel),
                                      "ms_per_step": round(1e3 * cel / args.cpu_baseline_steps, 1)}
            cpu.close()
        except Exception as e:  # the baseline is a reported number, never a reason to lose the GPU line
            result["cpu_baseline"] = {"value": None, "unit": "rays/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}
    if rank == 0:
        print(json.dumps(result), flush=True)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

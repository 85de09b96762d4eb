#!/usr/bin/env python3
"""bench.py — training rays/s of the normals-only SDF path on synthetic 64-view 800x800 normal+mask data.

  python bench.py --gpus N --steps K --warmup W

N > 1 from a plain shell: the script starts its own N ranks (python -m torch.distributed.run, one rank per GPU, rendezvous on
127.0.0.1); launched by torch.distributed.run itself (RANK / WORLD_SIZE in the environment) it is one of those ranks.

One "step" = one pass of the hot path over one batch: Testbed::train (occupancy update when due, ray generation +
march, network forward on the un-compacted samples, loss + compaction, forward/backward on 2^18 compacted samples,
Adam+EMA). Inputs are resident in HBM before the timed region. Prints ONE JSON line (rank 0).
On several GPUs the line carries BOTH scaling modes: `value` / `scaling` for the mode asked for (weak by default: 2^18
compacted samples per rank and step; --strong: the single-GPU step divided over the ranks) and the other mode's value in
`strong_scaling` / `weak_scaling`, measured in the same job.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import signal
import socket
import subprocess
import sys
import tempfile
import time

# more hardware queues than HIP's default 4: the library's side streams, torch's and RCCL's must not be multiplexed onto the
# queue that carries the step's critical path (set before the HIP runtime starts)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
PMC_TRAFFIC, PMC_UNITS = "r06_pmc_traffic.json", "r06_pmc_units.json"  # summaries of the separate rocprofv3 --pmc passes (tools/collect_pmc.sh)
PMC_FALLBACK = {"r06_pmc_traffic.json": "r05_pmc_traffic.json", "r06_pmc_units.json": "r05_pmc_units.json"}
# committed rocprofv3 summaries of this same command from which every `frac` of the record can be recomputed (profiles/README.md)
PROFILE_FILES = {"kernel_trace_serial": "profiles/r06_steady_serial_step2000.json", "kernel_trace_overlapped": "profiles/r06_steady_overlapped_step1000.json",
                 "pmc_traffic": "profiles/" + PMC_TRAFFIC, "pmc_units": "profiles/" + PMC_UNITS, "counter_calibration": "profiles/r05_counter_calibration.json"}
# What FETCH_SIZE means on gfx950, calibrated on this path's own access patterns (tools/probe_counters.hip -> profiles/r05_counter_calibration.json): the counter is the L2's
# memory-side read REQUESTS x 64 B. A scattered 8-byte gather is one 64-byte request (p_gather8_far: 64.0 B counted per gather = the bytes moved: x1); a coalesced stream is
# 128-byte requests counted as 64 (16 B and 8 B per lane, and 64-byte records read as 4 x 16 B: 0.500 / 0.500 / 0.503 of the true bytes: x2). WRITE_SIZE is exact for
# coalesced stores (1.000 / 1.000 / 1.007); an atomic without return is ONE write request counted as 32 B and no fetch (it executes memory-side).
# Per kernel group: the factor of the pattern that makes up (nearly) all of its reads.
FETCH_CORRECTION = {"k_adam_ema": 2.0, "k_grid_scatter": 2.0, "k_grid_scatter_lds": 2.0, "k_grid_scatter_quad_rl": 2.0, "k_grid_scatter_quad": 2.0, "k_dw*7+k_dw_finish": 2.0,
                    "k_loss_pass1": 2.0, "k_loss_pass2+k_rollover": 2.0, "k_march_write": 2.0, "k_scan_rays": 2.0, "k_scan_compact": 2.0,
                    "k_forward": 1.0, "k_fwd_bwd": 1.0, "k_point_query": 1.0, "k_march_count": 1.0}
FETCH_CORRECTION_NOTE = ("FETCH_SIZE (KiB x 1024) x %s + WRITE_SIZE (KiB x 1024); calibration profiles/r05_counter_calibration.json (tools/probe_counters.hip): scattered 8-byte gathers are "
                         "64-byte requests counted at 64 B (x1: k_forward, k_fwd_bwd, k_point_query -- their coalesced share, coordinates and indices, is < 7 %% of the bytes), coalesced streams are "
                         "128-byte requests counted at 64 B (x2: k_adam_ema, the scatter's operand reads, the loss passes), stores are exact, an atomic is one write request counted as 32 B")
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


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000, help="timed steps; default = the window SURVEY.md section 8(d) defines: training steps 1000-2000")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--burn-in", type=int, default=980, help="untimed training steps before warmup so that the measured steps sit in the "
                    "regime the metric is quoted on (burn-in + warmup = 1000: all 14 levels live, occupancy converged; SURVEY.md §8d)")
    ap.add_argument("--views", type=int, default=64)
    ap.add_argument("--res", type=int, default=800)
    ap.add_argument("--focal", type=float, default=None, help="focal length in pixels (default: the generator's, 1.75 x res as in SURVEY.md 8d)")
    ap.add_argument("--batch-log2", type=int, default=18, help="log2 of the compacted samples per step; anything but 18 is NOT the metric's workload (launcher tests)")
    ap.add_argument("--cpu-baseline-steps", type=int, default=8, help="steps of the CPU checker timed as the baseline (≈1.4 s each on the GPU box host)")
    ap.add_argument("--cpu-baseline-late-steps", type=int, default=3, help="the same for the late regime's state (second baseline, next to `late_regime`)")
    ap.add_argument("--profile-steps", type=int, default=100, help="serialized steps after the timed region for the per-kernel HIP-event table")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--window-end", type=int, default=2000, help="after the K timed steps the run continues (timed per step) to this training step, so that "
                    "a short --steps still reports the whole window SURVEY.md 8(d) defines (steps 1000-2000); 0 = off")
    ap.add_argument("--late-step", type=int, default=6000, help="then train on to this step and time --late-steps more: the converged regime (few samples "
                    "per ray, ~95 k rays per step) that the >= 1e8 rays/s target is about; 0 = off")
    ap.add_argument("--late-steps", type=int, default=200)
    ap.add_argument("--strong", action="store_true", help="strong scaling (SURVEY.md 8e): the job's step stays the single-GPU step (2^18 compacted samples, the "
                    "controller's ray count); every rank takes 1/N of its rays and samples (dp.strong_scaling_sizes). Default: weak scaling, 2^18 samples per rank. "
                    "With N > 1 the other mode is measured as a second leg of the same job and reported beside `value`")
    ap.add_argument("--other-leg-steps", type=int, default=200, help="timed steps of that second leg (0 = skip it)")
    ap.add_argument("--fixed-cost-world", type=int, default=8, help="single-GPU runs: time the step ONE rank of a strong-scaling job of this many GPUs would run (B / W compacted "
                    "samples, R / W rays: dp.strong_scaling_sizes) on this GPU -- the part of the step that does not shrink with the job size and bounds strong scaling; 0 = skip")
    ap.add_argument("--fixed-cost-steps", type=int, default=400)
    ap.add_argument("--albedo", action="store_true", help="secondary workload: stage 2 of the two-stage pipeline (colour MLP + reflectance loss live) instead of "
                    "the normals-only path the metric is quoted on")
    ap.add_argument("--accumulate", choices=["fp32", "half"], default="fp32", help="rnb_config::accumulate of every leg: fp32 (the default product mode, the one `value` is quoted for) or half "
                    "(the reference's arithmetic as coded: half k-step accumulators, packed half atomics into a half gradient vector)")
    ap.add_argument("--parity-mode-steps", type=int, default=400, help="single-GPU fp32 runs: a second context in the OTHER accumulate mode (half), trained to the same step and timed over the driver's K steps "
                    "and over this many more (`parity_mode` in the record); 0 = skip")
    ap.add_argument("--burn-in-mode", choices=["deterministic", "same"], default=None, help="how the untimed burn-in steps are trained (default: deterministic on one GPU, same on several -- the multi-rank form "
                    "is covered on two CPU ranks and has never run on several GPUs, and a scaling record must not depend on it). deterministic: in a second context with "
                    "rnb_config::deterministic = 1 (hash-grid gradients summed as fixed-point integers), whose state -- the SAME bytes on every run, `burn_in.state_sha256` -- is then loaded into the context "
                    "that is timed; the timed context always runs the mode of --accumulate / --deterministic. same: in the timed context itself (rounds 1-5: the non-reproducible training reached the "
                    "timed steps in one of several states, 0.592 or 0.616 ms/step)")
    ap.add_argument("--deterministic", action="store_true", help="rnb_config::deterministic = 1 in every leg (the timed steps too)")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not collect the HBM counters of `roofline.traffic` in this run (child processes under rocprofv3 --pmc); the record "
                    "then carries the committed summary's value and says so")
    ap.add_argument("--live-pmc-steps", type=int, default=10, help="steps each of those counter passes averages over")
    return ap.parse_args(argv)


# kernels of a bench kernel group (substring of the kernel name as rocprofv3 prints it, mangled or not)
PMC_GROUPS = {"k_forward": ("k_forward_chained",), "k_fwd_bwd": ("k_fwd_bwd", "k_rgb_fwd_bwd"), "k_grid_scatter": ("k_grid_scatter",), "k_adam_ema": ("k_adam_ema",),
              "k_grid_scatter_lds": ("k_grid_scatter_lds",), "k_grid_scatter_quad_rl": ("k_grid_scatter_quad_rl",), "k_grid_scatter_quad": ("k_grid_scatter_quad_h", "k_grid_scatter_quadENS", "k_grid_scatter_quad("),
              "k_march_count": ("k_march_count",), "k_march_write": ("k_march_write",), "k_point_query": ("k_point_query",), "k_loss_pass1": ("k_loss_pass1",),
              "k_loss_pass2+k_rollover": ("k_loss_pass2",), "k_dw*7+k_dw_finish": ("k_dw_",)}


def live_pmc(args, first_step, counters=("FETCH_SIZE", "WRITE_SIZE", "TCC_ATOMIC_sum"), timeout_s=240):
    """HBM bytes per step and kernel group, measured NOW: one child process of this script per counter under `rocprofv3 --pmc <counter> --kernel-trace`
    (separate passes, as MI355X_MICROARCH.md prescribes for the HBM counters; kernels serialised with cfg.overlap = 0 so that a group is one launch per step),
    trained to `first_step` like this run and averaged over the --live-pmc-steps steps from there. Conversions as in tools/pmc_traffic.py: FETCH_SIZE / WRITE_SIZE
    count KiB; FETCH_SIZE is corrected per kernel group by the factor its access pattern calibrates to (FETCH_CORRECTION above, profiles/r05_counter_calibration.json).
    Returns (per_step dict or None, note)."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    n = max(1, args.live_pmc_steps)
    burn = max(0, first_step - 2)
    per_step, t0 = {}, time.perf_counter()
    for counter in counters:
        d = tempfile.mkdtemp(prefix="rnb_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--steps", str(n), "--warmup", "2",
               "--burn-in", str(burn), "--profile-steps", "0", "--no-cpu-baseline", "--window-end", "0", "--late-step", "0", "--fixed-cost-steps", "0", "--no-live-pmc",
               "--parity-mode-steps", "0", "--accumulate", args.accumulate, "--burn-in-mode", args.burn_in_mode,
               "--views", str(args.views), "--res", str(args.res), "--batch-log2", str(args.batch_log2)] + (["--albedo"] if args.albedo else []) + (["--focal", str(args.focal)] if args.focal else [])
        env = dict(os.environ, RNB_OVERLAP_OFF="1", TMPDIR="/tmp")
        try:
            proc = subprocess.Popen(cmd, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = proc.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)  # the exact process group started here
                proc.wait()
                shutil.rmtree(d, ignore_errors=True)
                return None, "rocprofv3 --pmc %s did not finish within %d s" % (counter, timeout_s)
            rows = []
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if r.get("Counter_Name") == counter:
                            rows.append((int(r.get("Dispatch_Id", 0)), r["Kernel_Name"], float(r["Counter_Value"])))
        finally:
            shutil.rmtree(d, ignore_errors=True)
        rows.sort()
        starts = [r[0] for r in rows if "k_march_count" in r[1]]  # serialised: every step begins with its march (an update step: a few kernels earlier)
        if rc != 0 or len(starts) < n:
            return None, "rocprofv3 --pmc %s: exit code %d, %d steps seen" % (counter, rc, len(starts))
        lo = starts[-n]
        for g, names in PMC_GROUPS.items():
            v = sum(x for i, k, x in rows if i >= lo and any(s in k for s in names)) / n
            e = per_step.setdefault(g, {"fetch_bytes": 0, "write_bytes": 0, "atomic_lines": 0})
            if counter == "FETCH_SIZE":
                e["fetch_bytes"] = round(v * 1024)
            elif counter == "WRITE_SIZE":
                e["write_bytes"] = round(v * 1024)
            else:
                e["atomic_lines"] = round(v)
    for g, e in per_step.items():
        e["fetch_correction"] = FETCH_CORRECTION.get(g, 1.0)
        e["total_bytes_as_counted"] = e["fetch_bytes"] + e["write_bytes"]
        e["total_bytes"] = round(e["fetch_bytes"] * e["fetch_correction"]) + e["write_bytes"]
    return per_step, "this run: %d child passes under rocprofv3 --pmc (%s), kernels serialised, steps %d-%d, %.0f s" % (len(counters), ", ".join(counters), burn + 2, burn + 2 + n, time.perf_counter() - t0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n, argv):
    """`python bench.py --gpus N` from a plain shell: one rank per GPU under torch.distributed.run on this node. RNB_BENCH_ENTRY names
    the script the ranks run (default: this file; the CPU test of the launcher points it at an entry that injects a gloo engine)."""
    entry = os.environ.get("RNB_BENCH_ENTRY") or os.path.abspath(__file__)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n, "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), entry] + list(argv)
    return subprocess.call(cmd, env=env)


class HipEngine:
    """The product: librnb_neus2_hip.so on the rank's GPU, collectives over RCCL (torch.distributed backend "nccl")."""
    name, backend = "hip", "nccl"

    def setup(self, local_rank):
        import torch
        self.torch = torch
        self.local_rank = local_rank
        torch.cuda.set_device(local_rank)

    def init_process_group(self):
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=self.torch.device("cuda", self.local_rank))
        return dist

    def context(self, **kw):
        import rnb_neus2_amd as rnb
        return rnb.Context(overlap=0 if os.environ.get("RNB_OVERLAP_OFF") else 1, **kw)

    def trainer(self, ctx):
        from rnb_neus2_amd import dp
        return dp.DataParallelTrainer(ctx)

    def sync(self):
        self.torch.cuda.synchronize()

    def reduce_tensor(self, values):
        return self.torch.tensor(values, dtype=self.torch.float64, device="cuda")


def main(argv=None, engine=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus, argv)
    from rnb_neus2_amd import synthetic, dp

    engine = engine or HipEngine()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    engine.setup(local_rank)
    dist = None
    if world > 1 or (os.environ.get("RNB_DP_FORCE_COLLECTIVES") and "MASTER_ADDR" in os.environ):  # the env var exercises the RCCL path on one rank
        dist = engine.init_process_group()
    comm = {"backend": dist.get_backend(), "ranks": dist.get_world_size()} if dist is not None else {"backend": None, "ranks": 1}

    def barrier():
        if dist is not None:
            dist.barrier()
        engine.sync()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = engine.reduce_tensor([x])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    B = 1 << args.batch_log2
    t0 = time.time()
    scene = synthetic.make_scene(args.views, args.res) if args.focal is None else synthetic.make_scene(args.views, args.res, args.focal)
    scene_s = time.time() - t0
    flags = dict(apply_no_albedo=0 if args.albedo else 1, mask_loss_weight=1.0)  # stage 1 of run_two_stage: --mask-weight 1.0 --no-albedo (rnb_neus2/pipeline.py:63-74)
    accumulate = 1 if args.accumulate == "half" else 0
    deterministic = 1 if args.deterministic else 0
    if args.burn_in_mode is None:
        args.burn_in_mode = "deterministic" if world == 1 else "same"

    def state_of(ctx, st):
        return dict(params=ctx.get("PARAMS_FP32").copy(), grid=ctx.get("DENSITY_GRID").copy(), step=ctx.training_step, rays=ctx.rays_per_batch,
                    before=st.measured_batch_size_before_compaction)

    def open_leg(strong, **mode):
        if strong:
            sizes = dp.strong_scaling_sizes(world, B, min(1 << 18, B), min(1 << 12, B))
        else:
            sizes = dict(target_batch_size=B, max_rays_per_batch=min(1 << 18, B), initial_rays_per_batch=min(1 << 12, B)) if B != (1 << 18) else {}
        kw = dict(accumulate=accumulate, deterministic=deterministic)
        kw.update(mode)
        ctx = engine.context(world_size=world, rank=rank, **flags, **sizes, **kw)
        ctx.init_params()
        ctx.set_dataset(*scene)
        return ctx, engine.trainer(ctx)

    def state_sha256(state):
        import hashlib
        h = hashlib.sha256()
        for k in ("params", "adam_m", "adam_v", "adam_steps", "ema", "grid"):
            h.update(np.ascontiguousarray(state[k]).tobytes())
        h.update(np.array([state["step"], state["rays"], state["before"]], dtype=np.uint64).tobytes())
        return h.hexdigest()

    burn_states = {}

    def burn_state(strong, n_steps):
        """--burn-in-mode deterministic: the state after the untimed steps, trained ONCE per leg shape in a context with rnb_config::deterministic = 1 -- created, trained and CLOSED before the
        leg's own context exists -- and kept as data (api.Context.training_state: weights, Adam state, EMA, occupancy grid, controller): every run of this script times the same steps from the same bytes.
        (Two contexts alive at once share the process's four hardware queues: a context created after another one was closed beside a live one lost its side-stream overlap, 0.27 -> 0.40 ms/step
        for the fixed-cost leg -- measured this round; hence strictly one context at a time.)"""
        if n_steps <= 0 or args.burn_in_mode != "deterministic":
            return None
        key = (bool(strong), n_steps)
        if key not in burn_states:
            dctx, dtr = open_leg(strong, deterministic=1)
            st = None
            for _ in range(n_steps):
                st = dtr.step()
            if hasattr(dtr, "sync_parameters"):
                dtr.sync_parameters()  # sharded optimizer: masters, moments and EMA live on the owning rank until asked for
            engine.sync()
            burn_states[key] = dctx.training_state(st)
            dctx.close()
            del dtr, dctx
        return burn_states[key]

    def open_leg_burnt_in(strong, n_steps, **mode):
        """A leg's context at the end of the burn-in: (context, trainer, record)."""
        state = burn_state(strong, n_steps)
        ctx, trainer = open_leg(strong, **mode)
        if state is None:
            for _ in range(max(0, n_steps)):
                trainer.step()
            return ctx, trainer, ({"steps": n_steps, "mode": "same"} if n_steps > 0 else None)
        ctx.load_training_state(state)
        return ctx, trainer, {"steps": n_steps, "mode": "deterministic", "state_step": int(state["step"]), "state_rays_per_batch": int(state["rays"]), "state_sha256": state_sha256(state)}

    def timed_run(trainer, n_steps):
        """n_steps training steps between two barriers: (wall seconds (max over ranks), rays, compacted samples, samples before compaction, per-step host ms, last stats)."""
        barrier()
        tt0 = time.perf_counter()
        r = smp = smp_before = 0
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
            smp_before += stl.measured_batch_size_before_compaction * world
        barrier()
        return max_over_ranks(time.perf_counter() - tt0), r, smp, smp_before, per_step, stl

    # ---- the leg `value` is taken from ----
    t0 = time.time()
    ctx, trainer, burn_info = open_leg_burnt_in(args.strong, args.burn_in)
    setup_s = scene_s + time.time() - t0
    for _ in range(args.warmup):
        trainer.step()
    elapsed, rays, samples, samples_before, timed_ms, last = timed_run(trainer, args.steps)
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    cpu_state = state_of(ctx, last) if want_cpu and args.cpu_baseline_steps > 0 else None  # the regime `value` was measured in

    # The rest of the window the metric is defined on (training steps 1000-2000), so that the record carries it
    # even when --steps is small: the K steps above plus the steps up to --window-end.
    window = None
    first_timed = int(last.training_step) - args.steps
    n_more = args.window_end - int(last.training_step) if args.window_end else 0
    if n_more > 0:
        w_el, w_rays, w_smp, _, w_ms, last_w = timed_run(trainer, n_more)
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
    prof = []
    tail = last
    prof_first_step = int(ctx.training_step)
    if args.profile_steps > 0:
        ctx.profile_enable(True)
        for _ in range(args.profile_steps):
            tail = trainer.step()
        barrier()
        prof = ctx.profile()
        ctx.profile_enable(False)
    late = None
    late_state = None
    if args.late_step and args.late_steps > 0:
        while ctx.training_step < args.late_step:
            tail = trainer.step()
        l_el, l_rays, l_smp, _, l_ms, tail = timed_run(trainer, args.late_steps)
        late = {"first_step": int(tail.training_step) - args.late_steps, "steps": args.late_steps, "ms_per_step": round(1e3 * l_el / args.late_steps, 4),
                "rays_per_s": round(l_rays / l_el, 1), "rays_per_step": round(l_rays / args.late_steps / world, 1), "p50_ms_per_step": round(float(np.median(l_ms)), 4),
                "samples_per_s_compacted": round(l_smp / l_el, 1), "loss": round(float(tail.loss), 6)}
        if want_cpu and args.cpu_baseline_late_steps > 0:
            late_state = state_of(ctx, tail)
    final_loss = float(last.loss)
    ctx.close()

    # ---- several ranks: the other scaling mode, same job ----
    other = None
    if world > 1 and args.other_leg_steps > 0:
        ctx2, trainer2, _ = open_leg_burnt_in(not args.strong, args.burn_in)
        for _ in range(args.warmup):
            trainer2.step()
        o_el, o_rays, o_smp, _, _, o_last = timed_run(trainer2, args.other_leg_steps)
        other = {"scaling": "weak" if args.strong else "strong", "value": round(o_rays / o_el, 1), "unit": "rays/s", "steps": args.other_leg_steps,
                 "ms_per_step": round(1e3 * o_el / args.other_leg_steps, 4), "rays_per_step_per_gpu": round(o_rays / args.other_leg_steps / world, 1),
                 "samples_per_step_per_gpu": ctx2.cfg.target_batch_size, "samples_per_s_compacted": round(o_smp / o_el, 1), "loss": round(float(o_last.loss), 6)}
        ctx2.close()

    # ---- one GPU: the fixed cost of strong scaling (tools/strong_scaling_bound.py is the stand-alone form) ----
    fixed = None
    if world == 1 and rank == 0 and args.fixed_cost_world > 1 and args.fixed_cost_steps > 0 and not args.strong and B % (128 * args.fixed_cost_world) == 0:
        Wf = args.fixed_cost_world
        fctx = engine.context(world_size=1, rank=0, accumulate=accumulate, **flags, **dp.strong_scaling_sizes(Wf, B, min(1 << 18, B), min(1 << 12, B)))
        fctx.init_params()
        fctx.set_dataset(*scene)
        ftr = engine.trainer(fctx)
        for _ in range(args.burn_in + args.warmup):
            ftr.step()
        f_el, f_rays, _, _, f_ms, f_last = timed_run(ftr, args.fixed_cost_steps)
        fctx.profile_enable(True)
        n_prof = 64
        for _ in range(n_prof):
            ftr.step()
        barrier()
        fprof = {p["kernel"]: p["total_ms"] / n_prof for p in fctx.profile() if p["launches"]}
        fctx.profile_enable(False)
        fctx.close()
        f_step = 1e3 * f_el / args.fixed_cost_steps
        f_adam, f_pq = fprof.get("k_adam_ema", 0.0), fprof.get("k_point_query", 0.0)
        f_sharded = f_step - (1.0 - 1.0 / Wf) * (f_adam + f_pq)
        fixed = {"what": "the step one rank of a %d-GPU strong-scaling job runs (2^%d / %d compacted samples, rays / %d), timed on this one GPU without any exchange: what does not "
                         "shrink when the step is divided; the single-GPU step / this = the compute-side bound of the strong-scaling speed-up" % (Wf, args.batch_log2, Wf, Wf),
                 "world": Wf, "first_step": int(f_last.training_step) - args.fixed_cost_steps, "steps": args.fixed_cost_steps, "ms_per_step": round(f_step, 4),
                 "p50_ms_per_step": round(float(np.median(f_ms)), 4), "rays_per_step": round(f_rays / args.fixed_cost_steps, 1),
                 "replicated_in_this_measurement": {"k_adam_ema_ms_per_step": round(f_adam, 4), "k_point_query_ms_per_step": round(f_pq, 4),
                                                    "note": "serialised HIP-event times of the two kernels a real job divides by W: the sharded optimizer (rnb_train_step_apply_shard) and the "
                                                            "sharded occupancy update (rnb_update_density_grid_begin / _end); this single context runs both whole"},
                 "ms_per_step_with_both_divided_estimate": round(f_step - (1.0 - 1.0 / Wf) * (f_adam + f_pq), 4),
                 "strong_scaling_bound": {"measured": round(1e3 * elapsed / args.steps / f_step, 2), "with_both_divided_estimate": round(1e3 * elapsed / args.steps / max(f_sharded, 1e-6), 2),
                                          "note": "speed-up <= single-GPU ms_per_step / this, before any exchange; weak scaling (bench.py --gpus N: N x 2^18 samples per step) is the mode the >= 6x at 8 GPUs is claimed for (DESIGN.md section 7)"}}

    # ---- one GPU: the same workload in the other accumulate mode (half: the reference's arithmetic as coded) ----
    parity = None
    if world == 1 and rank == 0 and args.parity_mode_steps > 0 and not args.strong and accumulate == 0 and engine.name == "hip":
        sizes = dict(target_batch_size=B, max_rays_per_batch=min(1 << 18, B), initial_rays_per_batch=min(1 << 12, B)) if B != (1 << 18) else {}
        pctx, ptr, _ = open_leg_burnt_in(False, args.burn_in, accumulate=1)  # the SAME pinned state (a state is mode-independent data)
        for _ in range(args.warmup):
            ptr.step()
        p_el, p_rays, _, _, _, p_last = timed_run(ptr, args.steps)          # the driver's K steps
        q_el, q_rays, _, _, q_ms, q_last = timed_run(ptr, args.parity_mode_steps)
        pctx.profile_enable(True)
        n_prof = 64
        for _ in range(n_prof):
            ptr.step()
        barrier()
        pprof = {p["kernel"]: round(p["total_ms"] / n_prof, 4) for p in pctx.profile() if p["launches"]}
        pctx.profile_enable(False)
        pctx.close()
        parity = {"what": "rnb_config::accumulate = RNB_ACCUM_HALF on the same workload, same steps from the same state: every MLP dot product rounds its accumulator to half after each 16-wide k-step "
                          "(fully_fused_mlp.cu:59-68), the hash-grid gradients go through global_atomic_pk_add_f16 into a half gradient vector (grid.h:410-430, trainer.h:78-84) that the "
                          "optimizer reads at 2 bytes per parameter, the weight-gradient GEMMs run in CUTLASS's split-K order (4096-sample slices, half accumulators per 16-sample k-step, "
                          "cutlass_matmul.h:83, 315-322: k_dw_sliced, bit-identical to the model on the same operands). One stated departure from the reference's code in this mode: the scatter "
                          "sums a cell run / a workgroup's slice in fp32 before its one packed half atomic (RNB_SCATTER_PLAIN=1: every addend its own atomic). Against the reference-as-coded model on the pinned state: "
                          "tests/test_gpu_fullsize.py::test_hip_against_the_reference_as_coded_emulation[half] (Eikonal / mask sums 4e-6 / 6e-9; colour sum 1.65e-4 -- one ray -- NOT within the 1e-4)",
                  "accumulate": "half", "first_step": int(p_last.training_step) - args.steps, "steps": args.steps, "ms_per_step": round(1e3 * p_el / args.steps, 4), "rays_per_s": round(p_rays / p_el, 1),
                  "next_steps": {"steps": args.parity_mode_steps, "ms_per_step": round(1e3 * q_el / args.parity_mode_steps, 4), "p50_ms_per_step": round(float(np.median(q_ms)), 4),
                                 "rays_per_s": round(q_rays / q_el, 1), "loss": round(float(q_last.loss), 6)},
                  "kernels_ms_per_step_serialised": pprof}

    # ---- one GPU: the same steps from the same state with rnb_config::deterministic (the mode the parity tests and the burn-in run in) ----
    det_leg = None
    if world == 1 and rank == 0 and args.parity_mode_steps > 0 and not args.strong and not deterministic and engine.name == "hip":
        dctx, dtr, _ = open_leg_burnt_in(False, args.burn_in, deterministic=1)
        for _ in range(args.warmup):
            dtr.step()
        d_el, d_rays, _, _, _, d_last = timed_run(dtr, args.steps)
        e_el, e_rays, _, _, e_ms, e_last = timed_run(dtr, args.parity_mode_steps)
        d_state = dctx.training_state(e_last)
        dctx.close()
        det_leg = {"what": "rnb_config::deterministic = 1 on the same workload, same steps from the same state: the hash-grid gradients are summed as 64-bit fixed-point integers (exact, order-independent) "
                           "by integer atomics and narrowed once (k_fixed_narrow); every run of this leg ends in the same bytes (`end_state_sha256`)",
                   "first_step": int(d_last.training_step) - args.steps, "steps": args.steps, "ms_per_step": round(1e3 * d_el / args.steps, 4), "rays_per_s": round(d_rays / d_el, 1),
                   "next_steps": {"steps": args.parity_mode_steps, "ms_per_step": round(1e3 * e_el / args.parity_mode_steps, 4), "p50_ms_per_step": round(float(np.median(e_ms)), 4),
                                  "rays_per_s": round(e_rays / e_el, 1), "loss": round(float(e_last.loss), 6)},
                   "end_state_step": int(d_state["step"]), "end_state_sha256": state_sha256(d_state)}

    result = None
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = rays / elapsed
        # the HBM counters of the record's rooflines, collected now and in the regime of the per-kernel pass above (child processes; this process's contexts are idle)
        live, live_note = (None, "--no-live-pmc") if (args.no_live_pmc or world != 1 or engine.name != "hip" or not prof) else live_pmc(args, prof_first_step)

        def pmc_file(name):
            p = os.path.join(ROOT, "profiles", name)
            return (p, name) if os.path.exists(p) else (os.path.join(ROOT, "profiles", PMC_FALLBACK[name]), PMC_FALLBACK[name])

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
            launches_per_step = max(p["launches"] / max(args.profile_steps, 1), 1.0)
            if live is not None and kname in live:
                traffic = live[kname]["total_bytes"] / launches_per_step
                traffic_source = live_note + "; " + FETCH_CORRECTION_NOTE % ("%g" % FETCH_CORRECTION.get(kname, 1.0))
            else:
                try:
                    path, fname = pmc_file(PMC_TRAFFIC)
                    with open(path) as f:
                        e = json.load(f)["per_step"][kname]
                    traffic = (e["fetch_bytes"] * FETCH_CORRECTION.get(kname, 1.0) + e["write_bytes"]) / launches_per_step
                    traffic_source = "committed file profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, steps 2000-2010, builder-run), not this run (%s); %s" % (
                        fname, live_note, FETCH_CORRECTION_NOTE % ("%g" % FETCH_CORRECTION.get(kname, 1.0)))
                except Exception:
                    pass
            limiter = None  # the unit that actually bounds the kernel when it is not HBM bytes (tools/pmc_report.py); also from a committed file
            try:
                path, fname = pmc_file(PMC_UNITS)
                with open(path) as f:
                    units = json.load(f)
                if kname == "k_grid_scatter":
                    parts = [units["kernels"][k] for k in units["kernels"] if k.startswith("k_grid_scatter")]
                    req = sum(q.get("l2_atomic_requests", 0) for q in parts)
                    src = "per_launch: committed file profiles/%s (TCC_ATOMIC_sum pass at steps 2000-2010, builder-run; the duration is this run's, whose regime may put more lines on the path)" % fname
                    if live is not None and live.get(kname, {}).get("atomic_lines"):
                        req = live[kname]["atomic_lines"]  # all launches of the group in one step = one `launch` of the group in the serialised pass
                        src = "per_launch: " + live_note
                    limiter = {"unit": "L2 atomic lines (one 64-byte line of one instruction)", "per_launch": req, "achieved_per_s": round(req / (avg_ms * 1e-3)),
                               "probe_rate_per_s": units["_atomic_probe_requests_per_s"],
                               "frac": round(req / (avg_ms * 1e-3) / units["_atomic_probe_requests_per_s"], 3),
                               "limiter_source": src + "; probe rate: tools/probe_atomics4.hip"}
                    # the same per KERNEL of the group (round 5): each kernel's own lines over its own serialised duration
                    per_kernel = {}
                    by_name = {q["kernel"]: q for q in prof}
                    for kk in ("k_grid_scatter_lds", "k_grid_scatter_quad_rl", "k_grid_scatter_quad"):
                        q = by_name.get(kk)
                        lines = live.get(kk, {}).get("atomic_lines") if live is not None else None
                        if lines is None:  # (the committed summary names the kernel as launched: k_grid_scatter_quad_rl_direct, ..._h)
                            hits = [v for k, v in units["kernels"].items() if k.startswith(kk) and (kk != "k_grid_scatter_quad" or not k.startswith("k_grid_scatter_quad_rl"))]
                            lines = sum(v.get("l2_atomic_requests", 0) for v in hits) or None
                        if q and q["launches"] and lines:
                            ms_k = q["total_ms"] / q["launches"]
                            per_kernel[kk] = {"avg_launch_ms": round(ms_k, 4), "atomic_lines_per_launch": lines, "frac": round(lines / (ms_k * 1e-3) / units["_atomic_probe_requests_per_s"], 3)}
                    limiter["per_kernel"] = per_kernel
            except Exception:
                pass
            if limiter is None and kname in LIMITER_NOTES:
                limiter = {"unit": LIMITER_NOTES[kname]}
            return {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source, "avg_launch_ms": round(avg_ms, 4),
                    "units_per_launch": round(units_per_launch, 1), "algorithmic_bytes_per_unit": bytes_per_unit, "limiter": limiter,
                    "recompute_from": "frac = algorithmic_bytes_per_unit x units_per_launch / avg_launch_ms / peak; avg_launch_ms: this run's HIP events (serialised pass), cross-check: "
                                      "the kernel's per-step duration in %s" % PROFILE_FILES["kernel_trace_serial"]}

        ranked = sorted((p for p in prof if p["kernel"] not in ("k_grid_scatter_lds", "k_grid_scatter_quad_rl", "k_grid_scatter_quad")), key=lambda p: -p["total_ms"])  # (the group's three kernels are in its entry)
        # The dominant kernel of the STEP: the largest group on the step's critical stream. The next step's ray generation + march (k_march_count, k_scan_rays, k_march_write) runs a step
        # ahead on a side stream beside this step's backward pass and optimizer (cfg.overlap), and it is not a bandwidth kernel (instruction issue of a bit-exact
        # replay: its HBM roofline is meaningless, 0.013); since round 6 its serial time at 47 k rays per step (0.199 ms) is above the scatter group's (0.193 ms), so it is named here
        # explicitly instead of silently taking the slot: `roofline` = the critical stream's dominant group, `rooflines_next` = the next three groups by time INCLUDING the march.
        side = ("k_march_count", "k_scan_rays", "k_march_write")
        critical = [p for p in ranked if p["kernel"] not in side]
        roofline = roofline_of(critical[0]) if critical else None
        if roofline is not None:
            roofline["selection"] = ("largest kernel group on the step's critical stream by serialised HIP-event time; the side-stream march chain is listed in rooflines_next "
                                     "(largest group overall: %s, %.4f ms per step)" % (ranked[0]["kernel"], ranked[0]["total_ms"] / max(args.profile_steps, 1)))
        rest = [p for p in ranked if not critical or p is not critical[0]]
        rooflines_next = [r for r in (roofline_of(p) for p in rest[:4]) if r is not None]
        kernels = {p["kernel"]: {"ms_per_step": round(p["total_ms"] / max(args.profile_steps, 1), 4), "launches": p["launches"]} for p in prof if p["launches"]}
        result = {
            "metric": "training rays/s + ms/step, normals-only SDF 64x800^2",
            "value": round(value, 1),
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            # host time of each timed step on rank 0 (K <= 64): one step in 16 begins with an occupancy update (~1.1 ms); anything else that stands out is the host
            "timed_steps_ms": [round(x, 3) for x in timed_ms] if args.steps <= 64 else None,
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "f16 storage / f16 accumulate (k-step model)" if accumulate else "f16 storage / f32 accumulate",
            "data": "synthetic",
            "config": {"workload": "config 4: synthetic %d-view %dx%d normals+mask sphere, %s --mask-weight 1.0, %s compacted samples/step/GPU"
                                   % (args.views, args.res, args.res, "albedo + reflectance loss (NOT the metric's workload)" if args.albedo else "--no-albedo",
                                      ("2^%d / %d" % (args.batch_log2, world)) if args.strong else "2^%d" % args.batch_log2),
                       "burn_in_steps": args.burn_in, "burn_in": burn_info, "first_timed_step": first_timed, "deterministic": bool(deterministic),
                       "rays_per_step_per_gpu": round(rays / args.steps / world, 1),
                       "samples_per_s_compacted": round(samples / elapsed, 1),
                       "samples_per_s_before_compaction": round(samples_before / elapsed, 1),
                       "loss": round(final_loss, 6), "parallelism": "dp%d" % world, "setup_s": round(setup_s, 1), "engine": engine.name},
            "communicator": comm,
            "window_1000_2000": window,
            "late_regime": late,
            "fixed_cost": fixed,
            "parity_mode": parity,
            "deterministic_mode": det_leg,
            "roofline": roofline,
            "pmc_live": {"note": live_note, "bytes_and_atomic_lines_per_step": live},
            "rooflines_next": rooflines_next,
            "kernels_ms_per_step": kernels,
            "profiles": PROFILE_FILES,
        }
        if comm["backend"] == "nccl":
            result["rccl_ranks"] = comm["ranks"]
        if other is not None:
            result["weak_scaling" if args.strong else "strong_scaling"] = other

    # CPU baseline: the oracle (a port, not the reference — the reference has no CPU path) continues from the GPU's state at the END OF
    # THE TIMED K STEPS (the regime `value` was measured in) on the same workload, all host cores, a bounded number of steps; a second,
    # shorter one from the late regime's state stands next to `late_regime`.
    def cpu_leg(state, n_steps):
        from tests import oracle_lib
        cpu = oracle_lib.context(accumulate=accumulate, **flags)
        try:
            cpu.set_dataset(*scene)
            cpu.set_params(state["params"])
            cpu.put("DENSITY_GRID", state["grid"])
            cpu.update_density_bitfield()
            cpu.set_controller(state["step"], state["rays"], state["before"], 0)
            t0c = time.perf_counter()
            crays = 0
            for _ in range(n_steps):
                crays += cpu.train_step().rays_per_batch
            cel = time.perf_counter() - t0c
        finally:
            cpu.close()
        return {"value": round(crays / cel, 1), "unit": "rays/s", "cores": os.cpu_count(), "kind": "port",
                "sample": "%d training steps of the CPU oracle (OpenMP, all host cores) continued from the GPU's state at step %d (%d rays per step), same dataset/flags; %.1f s"
                          % (n_steps, state["step"], state["rays"], cel),
                "ms_per_step": round(1e3 * cel / n_steps, 1), "rays_per_step": round(crays / n_steps, 1)}

    if want_cpu and cpu_state is not None:
        try:
            result["cpu_baseline"] = cpu_leg(cpu_state, args.cpu_baseline_steps)
            if late_state is not None:
                result["late_regime"]["cpu_baseline"] = cpu_leg(late_state, args.cpu_baseline_late_steps)
        except Exception as e:  # the baseline is a reported number, never a reason to lose the GPU line
            result["cpu_baseline"] = {"value": None, "unit": "rays/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())

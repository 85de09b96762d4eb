"""Data-parallel training step over torch.distributed (RCCL on MI355X, gloo in the CPU tests).

The reference is single-GPU (no collective call sites). Sharding: every rank holds the full
parameter block, occupancy grid and dataset; rank r generates the global rays [r*R, (r+1)*R) of a step of W*R rays
(same PCG32 stream positions and image assignment as a single process running W*R rays), compacts its own
``target_batch_size`` samples, and the ranks exchange
  * the fp32 gradient accumulators (10.56 M floats = 42 MB) before the optimizer — by default as a SHARDED optimizer:
    reduce-scatter of each gradient block, Adam + EMA on this rank's 1/W of the block, all-gather of the fp16 training
    weights (7/8 * (42 + 21) MB on the wire per rank at W = 8 instead of 7/8 * 84 MB for an all-reduce, and 1/W of the
    optimizer's HBM traffic); ``sharded=False`` / ``RNB_DP_SHARDED=0`` selects the plain all-reduce + replicated optimizer;
  * one 7-value all-reduce of the step counters / loss sums so all ranks draw the same rays_per_batch next step.
With the sharded optimizer a rank's fp32 masters, EMA weights and Adam state are current only on its own chunks:
``sync_parameters()`` all-gathers them (before snapshots / inference with EMA weights).
The occupancy update is sharded (round 4): rank r evaluates its 1/W of the update's 2^20 sample points and the ranks take the element-wise max of
the splat targets (8 MB every 16 steps) -- the same grid as a single rank computes, bit for bit; ``grid_exchange=None`` keeps it replicated.
"""
import os

import numpy as np


def strong_scaling_sizes(world_size, target_batch_size=1 << 18, max_rays_per_batch=1 << 18, initial_rays_per_batch=1 << 12):
    """Context sizes of ONE rank for strong scaling: the step of the whole job is the single-GPU step --
    `target_batch_size` compacted samples, `rays_per_batch` rays -- and rank r marches rays [r R/W, (r+1) R/W) of it and compacts
    B/W samples. (The library always treats a step as world_size x the per-rank sizes: ray indices, image assignment, loss scale
    128 / R_global and the Eikonal divisor B_global are those of one process running the whole step; with the default sizes on
    every rank the job is weak scaling, W x the samples per step.) What differs from the single-GPU step: each rank pads ITS
    compacted batch to B/W (fill_rollover_and_rescale, common_device.h:514-535) and the controller rounds the per-rank ray
    count, not the global one, to a multiple of 128."""
    W = int(world_size)
    if target_batch_size % (128 * W):
        raise ValueError("target_batch_size must be a multiple of 128 x world_size")
    return dict(target_batch_size=target_batch_size // W, max_rays_per_batch=max(128, max_rays_per_batch // W), initial_rays_per_batch=max(1, initial_rays_per_batch // W))


class _DeviceArray:
    """Minimal __cuda_array_interface__ view of a device buffer owned by the HIP library."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def grads_name(ctx):
    """The gradient vector the ranks exchange: the fp32 accumulators, or -- rnb_config::accumulate = RNB_ACCUM_HALF -- the half vector (summed in half over the ranks)."""
    return "GRADS_FP16" if ctx.cfg.accumulate else "GRADS_FP32"


def grads_tensor(ctx):
    """torch view (no copy) of the context's gradient vector, for dist.all_reduce."""
    import torch
    name = grads_name(ctx)
    ptr, nbytes = ctx.buffer(name)
    size = 2 if name == "GRADS_FP16" else 4
    return torch.as_tensor(_DeviceArray(ptr, nbytes // size, _PARAM_BUFFERS[name]), device="cuda")


# parameter-shaped buffers and the element type their exchange uses (uint32 counters travel as int32)
_STAGING_VIEWS = ("PARAMS_FP32", "ADAM_M", "ADAM_V", "ADAM_STEPS")  # include/rnb_neus2.h, rnb_buffer
_PARAM_BUFFERS = {"GRADS_FP32": "<f4", "GRADS_FP16": "<f2", "PARAMS_FP16": "<f2", "PARAMS_FP32": "<f4", "PARAMS_EMA": "<f2", "ADAM_M": "<f4", "ADAM_V": "<f4", "ADAM_STEPS": "<i4"}


class TorchShardCollectives:
    """reduce-scatter / all-gather of a block of a parameter-shaped device buffer, in place, over torch.distributed (RCCL)."""

    on_device = True  # stream-ordered device work: the trainer may put block 0 on a side stream behind rnb_gradient_part_wait

    def __init__(self, ctx, capacity):
        self.ctx = ctx
        self.capacity = capacity
        self._views = {}

    def view(self, name):
        import torch
        v = self._views.get(name)
        if v is None or name in _STAGING_VIEWS:
            # the optimizer-state arrays are staging views: every rnb_buffer call brings them up to date (and announces a possible write)
            ptr, _ = self.ctx.buffer(name)  # allocated up to `capacity` elements (rnb_shard_layout)
            if v is None:
                v = self._views[name] = torch.as_tensor(_DeviceArray(ptr, self.capacity, _PARAM_BUFFERS[name]), device="cuda")
        return v

    def reduce_scatter(self, name, part):
        import torch.distributed as dist
        lo, hi, own_lo, own_hi = part
        v = self.view(name)
        dist.reduce_scatter_tensor(v[own_lo:own_hi], v[lo:hi], op=dist.ReduceOp.SUM)

    def all_gather(self, name, part):
        import torch.distributed as dist
        lo, hi, own_lo, own_hi = part
        v = self.view(name)
        dist.all_gather_into_tensor(v[lo:hi], v[own_lo:own_hi])
        if name == "PARAMS_FP16":  # written through a cached pointer: the kernels' weight images are stale (rnb_params_changed)
            self.ctx.params_changed()


def _raw_stream(stream):
    """HIP stream handle of a torch stream (None: the library's default stream handling)."""
    return None if stream is None else getattr(stream, "cuda_stream", stream)


class DataParallelTrainer:
    """Drives ``ctx`` (created with world_size/rank in its config) through begin -> all-reduce -> apply -> finish.

    ``all_reduce_grads(ctx)`` sums the gradient accumulators in place over the ranks; ``all_reduce_small(vec)``
    sums a float64 numpy vector. The defaults use torch.distributed; the CPU tests inject gloo/numpy versions.
    """

    def __init__(self, ctx, all_reduce_grads=None, all_reduce_small=None, stream=None, sharded=None, shard_collectives=None, grid_exchange="default"):
        """``shard_collectives``: object with reduce_scatter(name, part) / all_gather(name, part) working in place on the
        context's buffers (default: TorchShardCollectives); the CPU tests inject a gloo version over host buffers.
        ``grid_exchange``: callable(grid_tmp_ptr, n_elements, stream_handle) taking the element-wise max of the occupancy update's splat
        target over the ranks (rnb_set_grid_exchange: the update's 2^20 network evaluations are then divided over the ranks, every 16th
        step one max all-reduce of 8 MB); "default" = torch.distributed on the device buffer, None = replicated updates
        (also RNB_DP_SHARD_GRID=0)."""
        self.ctx = ctx
        self.stream = stream
        if sharded is None:
            sharded = all_reduce_grads is None and os.environ.get("RNB_DP_SHARDED", "1") != "0"
        self.sharded = bool(sharded)
        self._shard = shard_collectives
        self._layout = None
        self._grads = None
        self._side = None
        self._vec = None
        self._vec_host = None
        self._early = None
        self._collectives = ctx.cfg.world_size > 1 or bool(os.environ.get("RNB_DP_FORCE_COLLECTIVES"))  # the env var exercises the collective path on one rank
        self._reduce_grads = all_reduce_grads or self._torch_reduce_grads
        self._reduce_small = all_reduce_small  # None: all-reduce the library's device block in place
        if grid_exchange == "default":
            grid_exchange = self._torch_grid_exchange if (all_reduce_grads is None and shard_collectives is None) else None  # injected transports bring their own
        if os.environ.get("RNB_DP_SHARD_GRID", "1") == "0" or not self._collectives:
            grid_exchange = None
        ctx.set_grid_exchange(grid_exchange)
        self.grid_sharded = grid_exchange is not None

    def _torch_grid_exchange(self, ptr, n, stream_handle):
        """Element-wise max of DENSITY_GRID_TMP over the ranks, in place on the device, in the order of the single-rank splat: atomicMax on the
        words as UINT32 (testbed_nerf.cu:634). torch has no unsigned all-reduce, so the sign bit is flipped around a signed max (uint order
        = int order of word ^ 0x80000000): a NaN density with its sign bit set wins here exactly as it does on one rank."""
        import torch
        import torch.distributed as dist
        t = torch.as_tensor(_DeviceArray(ptr, n, "<i4"), device="cuda")

        def run():
            t.bitwise_xor_(-0x80000000)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            t.bitwise_xor_(-0x80000000)
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                run()
        else:
            run()

    def _torch_reduce_grads(self, ctx):
        """Sum of the fp32 gradient accumulators over the ranks. The block of levels whose scatter finishes first is exchanged
        on a side stream while the remaining levels are still being scattered; the rest follows on the default stream."""
        import torch
        import torch.distributed as dist
        if self._grads is None:
            self._grads = grads_tensor(ctx)
        parts = ctx.gradient_parts()
        if len(parts) > 1 and dist.get_backend() == "nccl":
            if self._early is None:
                self._early = torch.cuda.Stream()
            with torch.cuda.stream(self._early):  # every block but the last as soon as its levels are final, beside the scatter of the levels behind it
                for k, (lo, hi) in enumerate(parts[:-1]):
                    ctx.gradient_part_wait(k, self._early.cuda_stream)
                    dist.all_reduce(self._grads[lo:hi], op=dist.ReduceOp.SUM)
                    if k == 0:
                        ctx.train_step_apply_early(self._early.cuda_stream)  # Adam on that block, beside the rest of the exchange
            lo, hi = parts[-1]
            dist.all_reduce(self._grads[lo:hi], op=dist.ReduceOp.SUM)
            if len(parts) > 2:
                (self.stream or torch.cuda.current_stream()).wait_stream(self._early)  # the middle block's sums: train_step_apply steps it on that stream
        else:
            dist.all_reduce(self._grads, op=dist.ReduceOp.SUM)

    def _shard_setup(self, ctx):
        parts, capacity = ctx.shard_layout()
        if self._layout is not None and self._layout != (parts, capacity):
            raise RuntimeError("the shard layout changed between steps: a parameter's optimizer state would change ranks")
        self._layout = (parts, capacity)
        if self._shard is None:
            self._shard = TorchShardCollectives(ctx, capacity)
        return parts

    def _sharded_apply(self, ctx):
        """Per gradient block, in completion order: reduce-scatter -> Adam + EMA on the own chunk -> all-gather of the fp16
        training weights. Every block but the last (everything in front of the finest levels; then the first half of the finest
        levels) goes through this on a side stream while the levels behind it are still being scattered."""
        parts = self._shard_setup(ctx)
        on_device = bool(getattr(self._shard, "on_device", False))
        early = None
        if on_device and len(parts) > 1:
            import torch
            if self._early is None:
                self._early = torch.cuda.Stream()
            early = self._early
            with torch.cuda.stream(early):
                for k in range(len(parts) - 1):
                    ctx.gradient_part_wait(k, early.cuda_stream)
                    self._shard.reduce_scatter(grads_name(ctx), parts[k])
                    ctx.train_step_apply_shard(k, early.cuda_stream)
                    self._shard.all_gather("PARAMS_FP16", parts[k])
            rest = range(len(parts) - 1, len(parts))
        else:
            rest = range(len(parts))
        handle = _raw_stream(self.stream)
        for k in rest:
            if on_device:
                ctx.gradient_part_wait(k, handle)
            self._shard.reduce_scatter(grads_name(ctx), parts[k])
            ctx.train_step_apply_shard(k, handle)
            self._shard.all_gather("PARAMS_FP16", parts[k])
        if early is not None:
            import torch
            (self.stream or torch.cuda.current_stream()).wait_stream(early)
        ctx.train_step_apply_done(handle)

    def sync_parameters(self, names=("PARAMS_FP32", "PARAMS_EMA", "ADAM_M", "ADAM_V", "ADAM_STEPS")):
        """All-gather the per-rank chunks of the fp32 masters, EMA weights and Adam state (sharded optimizer only), so that
        every rank holds them whole: before snapshots, mesh extraction / rendering with the EMA weights, or reading them."""
        if not (self.sharded and self._collectives) or self._layout is None:
            return
        parts, _ = self._layout
        for name in names:
            for part in parts:
                self._shard.all_gather(name, part)

    def _torch_reduce_step_vector(self, ctx):
        """All-reduce of the step's 7 counters / loss sums, in place on the library's device block (RNB_BUF_STEP_VECTOR): no
        host-to-device copy, one small device-to-host copy. On its own (non-blocking) stream: the default stream holds this
        step's backward pass, and waiting for it here would delay the ray controller and with it the next step's march."""
        import torch
        import torch.distributed as dist
        if dist.get_backend() != "nccl":
            counters, sums = ctx.train_step_local(self.stream)
            t = torch.from_numpy(np.concatenate([counters.astype(np.float64), sums]))
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return t.numpy()
        if self._vec is None:
            ptr, _ = ctx.buffer("STEP_VECTOR")
            self._vec = torch.as_tensor(_DeviceArray(ptr, 7, "<f8"), device="cuda")
            self._side = torch.cuda.Stream()
            self._vec_host = torch.empty(7, dtype=torch.float64, pin_memory=True)
        with torch.cuda.stream(self._side):
            dist.all_reduce(self._vec, op=dist.ReduceOp.SUM)
            self._vec_host.copy_(self._vec, non_blocking=True)
            self._side.synchronize()
        return self._vec_host.numpy().copy()

    def step(self, allow_no_samples=False):
        """begin (march .. loss .. backward queued) -> counters as soon as the loss pass is done -> 7-value all-reduce ->
        finish (ray controller; queues the NEXT step's march on the side stream) -> gradient all-reduce -> apply (Adam).
        The next step's march therefore runs beside this step's backward pass, gradient all-reduce and optimizer."""
        ctx = self.ctx
        if not self._collectives and not os.environ.get("RNB_DP_SPLIT_CALLS"):
            # one rank, nothing to exchange: the library's own step (rnb_train_step = begin, apply, local, finish in one call). The next step's march is then queued
            # by the C code the moment the loss readback arrives; through the four Python calls below it started ~25 us later (measured, late regime: its chain is what
            # the next network evaluation waits for there)
            return ctx.train_step(self.stream, allow_no_samples=allow_no_samples)
        ctx.train_step_begin(self.stream)
        counters, sums = ctx.train_step_local(self.stream)
        if self._collectives:
            if self._reduce_small is not None:
                vec = self._reduce_small(np.concatenate([counters.astype(np.float64), sums]))
            else:
                vec = self._torch_reduce_step_vector(ctx)
            counters, sums = np.rint(vec[:4]).astype(np.uint64), vec[4:]
        try:
            stats = ctx.train_step_finish(counters, sums, allow_no_samples=allow_no_samples)
        finally:  # the optimizer runs even when the step produced no samples, as in the reference
            if self._collectives and self.sharded:
                self._sharded_apply(ctx)
            else:
                if self._collectives:
                    self._reduce_grads(ctx)
                ctx.train_step_apply(self.stream)
        return stats

"""Rank entry of tests/test_bench_launcher_cpu.py: bench.main() with the CPU checker as the engine and gloo as the transport, so that
bench.py's own launcher (`python bench.py --gpus 2` -> torch.distributed.run -> ranks -> strong + weak legs -> one JSON line) runs end
to end in a container without a GPU. TEST INFRASTRUCTURE: bench.py itself never selects this engine (RNB_BENCH_ENTRY points here)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


class OracleGlooEngine:
    name, backend = "cpu-oracle (launcher test)", "gloo"

    def setup(self, local_rank):
        pass

    def init_process_group(self):
        import torch.distributed as dist
        dist.init_process_group(backend="gloo")
        return dist

    def context(self, **kw):
        from tests import oracle_lib
        return oracle_lib.context(**kw)

    def trainer(self, ctx):
        import torch
        import torch.distributed as dist
        from rnb_neus2_amd import dp
        if ctx.cfg.world_size == 1:
            return dp.DataParallelTrainer(ctx)

        def reduce_grads(c):  # host buffers: stage through a torch tensor
            t = torch.from_numpy(c.get("GRADS_FP32"))
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            c.put("GRADS_FP32", t.numpy())

        def reduce_small(vec):
            t = torch.from_numpy(np.ascontiguousarray(vec, dtype=np.float64))
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return t.numpy()

        return dp.DataParallelTrainer(ctx, all_reduce_grads=reduce_grads, all_reduce_small=reduce_small)

    def sync(self):
        pass

    def reduce_tensor(self, values):
        import torch
        return torch.tensor(values, dtype=torch.float64)


if __name__ == "__main__":
    import bench
    sys.exit(bench.main(engine=OracleGlooEngine()))

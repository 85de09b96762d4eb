"""GPU box, diagnosis aid for round 5's red driver run (pytest -m gpu: SIGABRT after 61.6 s, at the point of the suite where
test_data_parallel_trainer_over_rccl_single_rank ran its second parametrisation): does init_process_group("nccl") -> collectives -> destroy_process_group
-> init_process_group("nccl") again IN ONE PROCESS abort now and then?  N fresh python processes each do that sequence `--cycles` times (2 = what the test
did), with the library's contexts and side streams alive as in the test; the exit codes are counted.

    python tools/rccl_reinit_stress.py [--procs 30] [--cycles 2] [--with-library 1]  ->  one JSON line (and gpurun_out/r06_rccl_reinit_stress.json)"""
import argparse
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, socket, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
cycles, with_lib = %(cycles)d, %(with_lib)d
ctx = None
if with_lib:
    import rnb_neus2_amd as rnb
    from rnb_neus2_amd import dp, synthetic
    scene = synthetic.make_scene(8, 128, 224.0)
    os.environ["RNB_DP_FORCE_COLLECTIVES"] = "1"
for k in range(cycles):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%%d" %% port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    t = torch.ones(1 << 20, device="cuda")
    dist.all_reduce(t)
    if with_lib:
        ctx = rnb.Context(target_batch_size=1 << 14, max_rays_per_batch=1 << 14, initial_rays_per_batch=1024, apply_no_albedo=1, overlap=1, accumulate=k %% 2)
        ctx.init_params(); ctx.set_dataset(*scene)
        tr = dp.DataParallelTrainer(ctx, sharded=True)
        for _ in range(20):
            tr.step()
        tr.sync_parameters()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    if ctx is not None:
        ctx.close(); ctx = None
print("CYCLES_OK", flush=True)
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=30)
    ap.add_argument("--cycles", type=int, default=2)
    ap.add_argument("--with-library", type=int, default=1)
    a = ap.parse_args()
    code = CHILD % dict(root=ROOT, cycles=a.cycles, with_lib=a.with_library)
    rcs, tails = [], []
    for i in range(a.procs):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
        rcs.append(r.returncode)
        if r.returncode != 0 or "CYCLES_OK" not in r.stdout:
            tails.append({"proc": i, "rc": r.returncode, "stderr_head": r.stderr[:3000], "stderr_tail": r.stderr[-1500:]})
    out = {"what": "init_process_group(nccl) -> collectives -> destroy, %d times in one process; %d processes" % (a.cycles, a.procs), "with_library": a.with_library,
           "exit_codes": {str(k): rcs.count(k) for k in sorted(set(rcs))}, "failures": tails}
    line = json.dumps(out)
    print(line)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06_rccl_reinit_stress_c%d_l%d.json" % (a.cycles, a.with_library)), "w") as f:
        f.write(line + "\n")


if __name__ == "__main__":
    main()

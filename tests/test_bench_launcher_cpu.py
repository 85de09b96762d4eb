"""bench.py's multi-GPU entry point without a wrapper: `python bench.py --gpus 2 [--strong]` from a plain
shell starts its own two ranks under torch.distributed.run, runs both scaling modes and prints ONE JSON line. Here the ranks run
tests/bench_gloo_entry.py (gloo + the CPU checker as the engine, a 4-view scene and 2^12 samples per step); on a GPU node the same
launcher starts bench.py itself over RCCL."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--views", "4", "--res", "48", "--focal", "84", "--batch-log2", "12", "--burn-in", "2", "--warmup", "1", "--steps", "2", "--other-leg-steps", "1",
         "--window-end", "0", "--late-step", "0", "--profile-steps", "0", "--no-cpu-baseline", "--burn-in-mode", "deterministic"]


def _run(extra, tail=()):
    env = dict(os.environ)
    env["RNB_BENCH_ENTRY"] = os.path.join(ROOT, "tests", "bench_gloo_entry.py")
    env["OMP_NUM_THREADS"] = "2"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra + SMALL + list(tail), capture_output=True, text=True, timeout=850, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(900)
@pytest.mark.parametrize("strong", [True])  # one job measures both modes; --strong selects which one is `value`
def test_bench_gpus_2_spawns_its_own_ranks(strong):
    rec = _run(["--gpus", "2"] + (["--strong"] if strong else []))
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["warmup"] == 1 and rec["unit"] == "rays/s" and rec["higher_is_better"] is True
    assert rec["communicator"] == {"backend": "gloo", "ranks": 2} and "rccl_ranks" not in rec  # RCCL reports itself only when it is the transport
    assert rec["scaling"] == ("strong" if strong else "weak") and rec["value"] > 0 and rec["ms_per_step"] > 0
    other = rec["weak_scaling" if strong else "strong_scaling"]
    assert other["scaling"] == ("weak" if strong else "strong") and other["value"] > 0 and other["steps"] == 1
    # strong: the job's step is the single-GPU step (2^12 samples over both ranks); weak: 2^12 samples per rank
    assert other["samples_per_step_per_gpu"] == ((1 << 12) if strong else (1 << 11))
    assert rec["config"]["parallelism"] == "dp2" and "launcher test" in rec["config"]["engine"]
    assert rec["config"]["rays_per_step_per_gpu"] > 0 and other["rays_per_step_per_gpu"] > 0
    # the untimed burn-in ran in a deterministic context (both ranks, through the trainer's collectives) and was handed over as data; its hash is in the record
    b = rec["config"]["burn_in"]
    assert b["mode"] == "deterministic" and b["steps"] == 2 and b["state_step"] == 2 and len(b["state_sha256"]) == 64
    again = _run(["--gpus", "2"] + (["--strong"] if strong else []), tail=["--steps", "1", "--warmup", "0", "--other-leg-steps", "0"])  # (the burn-in is what is compared: the shortest timed part)
    assert again["config"]["burn_in"]["state_sha256"] == b["state_sha256"]  # the same bytes on every run


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)

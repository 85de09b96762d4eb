"""The `testbed` command line as a JOB of two processes (struct Dist, rnb-neus2_amd/host/testbed_main.cpp) on CPU: the product's source built against the CPU
checker, the collectives through the host-staged test transport (dist_transport.hpp, RNB_DP_TRANSPORT=staged) -- reduce-scatter / shard apply / all-gather over
three gradient blocks with non-zero chunk offsets, the step-vector all-reduce, the sharded occupancy update's max exchange, sync_parameters() before rank 0 writes.
Pinned three ways: sharded == all-reduce + replicated optimizer bit for bit; both == the same protocol stated in Python on two checker contexts (numpy sums);
a rank that fails takes the job down with a non-zero exit code. The GPU twin (two HIP processes sharing the one GPU) is tests/test_gpu_cli.py."""
import json
import os
import shutil
import subprocess

import msgpack
import numpy as np
import pytest

from rnb_neus2_amd import synthetic
from tests import oracle_lib
from tests.conftest import SMALL_CFG

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAUNCH = os.path.join(ROOT, "tools", "launch_testbed.sh")
N_STEPS = 10

# six levels: the checker's shard layout then has three blocks (splits in front of the four and of the two finest levels, oracle/rnb_oracle.cpp shard_layout)
CFG6 = json.loads(json.dumps(SMALL_CFG))
CFG6["encoding"].update(n_levels=6, top_resolution=128)
# (per_level_scale: the command line derives it with expf / logf, src/testbed.cu:2296-2305; numpy's float32 exp lands one ulp beside it for these sizes, which moves the
# finest level's resolution from 128 to 129 -- the Python side takes the value the job itself recorded in its snapshot)
KW6 = dict(n_levels=6, log2_hashmap_size=12, base_resolution=16, mask_loss_weight=1.0, apply_no_albedo=1)


@pytest.fixture(scope="module")
def job_install(install):
    with open(os.path.join(install, "configs", "nerf", "small6.json"), "w") as f:
        json.dump(CFG6, f)
    return install


@pytest.fixture(scope="module")
def scene_src(tmp_path_factory):
    d = tmp_path_factory.mktemp("scene2")
    data = synthetic.make_scene(4, 48, 84.0)
    synthetic.write_scene(str(d), *data)
    return d, data


def _job(install, scene_src, tmp_path, name, n_ranks, env=None, args=(), cmd=None):
    scene = tmp_path / name
    shutil.copytree(scene_src, scene)
    stage = tmp_path / (name + "_stage")
    # (the checker's fp32 sums are partitioned by OpenMP thread: the jobs run with the thread count of the in-process checker they are compared with)
    e = dict(os.environ, RNB_DP_TRANSPORT="staged", RNB_DP_STAGE_DIR=str(stage))
    e.update(env or {})
    base = [str(install / "build" / "testbed"), "--scene", str(scene) + "/", "--maxiter", str(N_STEPS), "--no-gui", "--mask-weight", "1.0", "--config", "small6.json", "--no-albedo", "--save-snapshot", *args]
    r = subprocess.run([LAUNCH, str(n_ranks)] + (cmd(base) if cmd else base), env=e, capture_output=True, text=True, timeout=900)
    return r, scene


def _snapshot(scene):
    with open(scene / "output" / ("snapshot_%d.msgpack" % N_STEPS), "rb") as f:
        return msgpack.unpackb(f.read(), raw=False)


@pytest.fixture(scope="module")
def jobs(job_install, scene_src, tmp_path_factory):
    tmp = tmp_path_factory.mktemp("jobs")
    out = {}
    for name, env in (("sharded", {}), ("allreduce", {"RNB_DP_SHARDED": "0", "RNB_DP_SHARD_GRID": "0"})):  # (the second job also keeps its occupancy updates replicated)
        r, scene = _job(job_install, scene_src[0], tmp, name, 2, env)
        assert r.returncode == 0, (name, r.stdout[-2000:], r.stderr[-2000:])
        out[name] = dict(stdout=r.stdout, snap=_snapshot(scene), scene=scene)
    return out


def test_two_rank_job_runs_the_sharded_protocol(jobs):
    s = jobs["sharded"]
    assert "staged_ranks: 2 (sharded optimizer)" in s["stdout"] and "staged_ranks: 2 (all-reduce, replicated optimizer)" in jobs["allreduce"]["stdout"]
    assert s["stdout"].count("Saving Snapshot !") == 1  # rank 0 alone writes
    snap = s["snap"]["snapshot"]
    assert snap["training_step"] == N_STEPS
    assert s["snap"]["hyperparams"]["batch_size"] == 4096  # the JOB's batch, not a rank's share


def test_sharded_equals_allreduce_equals_replicated_grid_bit_for_bit(jobs):
    """sync_parameters() has gathered the other rank's chunks of the EMA weights (non-zero offsets in all three blocks): the sharded job's snapshot is the
    all-reduce job's, and the sharded occupancy update (max exchange) leaves the replicated update's grid (the all-reduce job runs its updates replicated)."""
    ref = jobs["allreduce"]["snap"]["snapshot"]
    for name in ("sharded",):
        snap = jobs[name]["snap"]["snapshot"]
        assert snap["params_binary"] == ref["params_binary"], name
        assert snap["density_grid_binary"] == ref["density_grid_binary"], name
        assert snap["nerf"]["rgb"] == ref["nerf"]["rgb"] and snap["loss"] == ref["loss"], name
    ema = np.frombuffer(ref["params_binary"], np.float16)
    assert np.isfinite(ema.astype(np.float32)).all() and np.count_nonzero(ema) > ema.size // 2


def test_two_rank_job_equals_the_protocol_stated_in_python(jobs, scene_src):
    """Two checker contexts as ranks 0 / 1 of a strong-scaling job, driven through the stage calls with the exchanges done by numpy: the step vector summed, the
    gradient accumulators summed (a + b, the staged transport's fold), the replicated optimizer. Rank 0's EMA weights and occupancy grid are the snapshot's."""
    views, normals, albedos = scene_src[1]
    sizes = dict(target_batch_size=4096 // 2, max_rays_per_batch=max(128, (1 << 18) // 2), initial_rays_per_batch=(1 << 12) // 2)
    ranks = []
    pls = float(np.float32(jobs["allreduce"]["snap"]["encoding"]["per_level_scale"]))
    for r in range(2):
        c = oracle_lib.context(world_size=2, rank=r, per_level_scale=pls, **KW6, **sizes)
        c.init_params()
        c.set_dataset(views, normals, albedos)
        ranks.append(c)
    try:
        st = None
        for _ in range(N_STEPS):
            for c in ranks:
                c.train_step_begin()
            local = [c.train_step_local() for c in ranks]
            cnt = [int(sum(int(l[0][k]) for l in local)) for k in range(4)]
            sums = [float(np.float64(local[0][1][k]) + np.float64(local[1][1][k])) for k in range(3)]
            g = ranks[0].get("GRADS_FP32") + ranks[1].get("GRADS_FP32")
            for c in ranks:
                c.put("GRADS_FP32", g)
                st = c.train_step_finish(cnt, sums)
                c.train_step_apply()
        snap = jobs["allreduce"]["snap"]["snapshot"]
        np.testing.assert_array_equal(np.frombuffer(snap["params_binary"], np.uint16), ranks[0].get("PARAMS_EMA").view(np.uint16))
        np.testing.assert_array_equal(np.frombuffer(snap["density_grid_binary"], np.float16), ranks[0].get("DENSITY_GRID").astype(np.float16))
        assert snap["nerf"]["rgb"]["rays_per_batch"] == st.next_rays_per_batch
    finally:
        for c in ranks:
            c.close()


def test_half_gradient_vector_is_exchanged_in_half(job_install, scene_src, tmp_path):
    """--accumulate half: the ranks reduce-scatter RNB_BUF_GRADS_FP16 as halfs (the staged transport folds in half like ncclHalf sums); sharded == all-reduce."""
    snaps = []
    for name, env in (("h_sharded", {}), ("h_allreduce", {"RNB_DP_SHARDED": "0"})):
        r, scene = _job(job_install, scene_src[0], tmp_path, name, 2, env, args=("--accumulate", "half"))
        assert r.returncode == 0, (name, r.stdout[-2000:], r.stderr[-2000:])
        snaps.append(_snapshot(scene)["snapshot"])
    assert snaps[0]["params_binary"] == snaps[1]["params_binary"] and snaps[0]["density_grid_binary"] == snaps[1]["density_grid_binary"]


@pytest.mark.parametrize("failing_rank", [0, 1])
def test_a_failing_rank_takes_the_job_down(job_install, scene_src, tmp_path, failing_rank):
    """One rank exits early (a missing network config): its exit code is the job's, and the other rank does not wait for it until the transport's timeout."""
    import time

    def cmd(base):
        bad = [a if a != "small6.json" else "missing.json" for a in base]
        q = lambda xs: " ".join("'%s'" % x for x in xs)  # noqa: E731
        return ["sh", "-c", 'if [ "$RNB_RANK" = "%d" ]; then exec %s; else exec %s; fi' % (failing_rank, q(bad), q(base))]
    t0 = time.time()
    r, _ = _job(job_install, scene_src[0], tmp_path, "fail%d" % failing_rank, 2, cmd=cmd)
    assert r.returncode == 1, (r.returncode, r.stderr[-1000:])
    assert "Network config path" in r.stderr
    assert "another rank has aborted the job" in r.stderr
    assert time.time() - t0 < 60

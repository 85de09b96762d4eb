"""dp.DataParallelTrainer through real RCCL calls (a world of 1, RNB_DP_FORCE_COLLECTIVES): the sharded optimizer and the all-reduce path both reproduce the
plain training step at full size. accumulate = 1: the half mode, whose gradient vector (RNB_BUF_GRADS_FP16) travels through reduce_scatter_tensor / all_reduce
as halfs. With rnb_config::deterministic the three trainers are compared bit for bit over 21 steps; in the default mode (floating-point atomics) the first step within the
atomics' noise. Every case is its own python process (tests/rccl_single_rank_worker.py): RCCL is initialised exactly once per process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
@pytest.mark.parametrize("accumulate,deterministic", [(0, 1), (1, 1), (0, 0), (1, 0)])
def test_data_parallel_trainer_over_rccl_single_rank(accumulate, deterministic):
    r = subprocess.run([sys.executable, "-m", "tests.rccl_single_rank_worker", str(accumulate), str(deterministic)], cwd=ROOT, capture_output=True, text=True, timeout=850)
    print(r.stdout[-3000:])
    assert r.returncode == 0 and "RCCL_SINGLE_RANK_OK" in r.stdout, "exit code %s\n%s\n%s" % (r.returncode, r.stdout[-3000:], r.stderr[-6000:])

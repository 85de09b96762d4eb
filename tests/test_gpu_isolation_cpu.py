"""The GPU tier's process isolation (tests/conftest.py) exercised without a GPU: a copy of the conftest drives a made-up module whose second test
ends its python process with SIGABRT (what the HIP runtime does on a GPU memory fault). The run must name that test, fail only it, continue with
the module's remaining tests in a fresh process, keep the other module's results, and end with the ordinary summary line and exit code."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))

MOD_A = '''
import os, pytest
pytestmark = pytest.mark.gpu

@pytest.fixture(scope="module")
def state():
    return {"n": 0}

def test_one(state):
    state["n"] += 1

def test_aborts(state):
    print("about to abort", flush=True)
    os.abort()

def test_after_the_abort(state):
    assert state["n"] == 0  # a fresh process: the module fixture was rebuilt

def test_plain_failure():
    assert 1 == 2, "an ordinary failure travels with its text"

@pytest.mark.skip(reason="skipped on purpose")
def test_skipped():
    pass
'''
MOD_B = '''
import pytest
pytestmark = pytest.mark.gpu

@pytest.mark.parametrize("k", [0, 1])
def test_other_module(k):
    assert k in (0, 1)
'''


def _run(tmp_path, extra):
    t = tmp_path / "tests"
    if not t.exists():
        t.mkdir()
        shutil.copy(os.path.join(HERE, "conftest.py"), t / "conftest.py")
        (t / "__init__.py").write_text("")
        (t / "test_gpu_aaa.py").write_text(MOD_A)
        (t / "test_gpu_bbb.py").write_text(MOD_B)
    env = dict(os.environ)
    env.pop("RNB_GPU_CHILD_RESULTS", None)
    env.pop("RNB_GPU_ISOLATE", None)
    return subprocess.run([sys.executable, "-m", "pytest", "tests/", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + extra, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)


def test_a_native_abort_fails_one_named_test_and_the_run_goes_on(tmp_path):
    r = _run(tmp_path, [])
    out = r.stdout + r.stderr
    assert r.returncode == 1, out
    assert "4 passed" in out and "2 failed" in out and "1 skipped" in out, out
    assert "FAILED tests/test_gpu_aaa.py::test_aborts [call] child process ended by signal 6 (SIGABRT)" in out, out
    assert "an ordinary failure travels with its text" in out
    assert "RUN tests/test_gpu_aaa.py::test_aborts" in out and "OK tests/test_gpu_bbb.py::test_other_module[1]" in out
    # the tail names the pass count
    assert "passed=4" in out


def test_x_stops_at_the_abort_and_names_it(tmp_path):
    r = _run(tmp_path, ["-x"])
    out = r.stdout + r.stderr
    assert r.returncode == 1, out
    assert "1 passed" in out and "1 failed" in out, out
    assert "test_aborts" in out and "SIGABRT" in out
    assert "test_after_the_abort" not in out.split("child process ended by")[-1].split("FAILURES")[0]


MOD_HANG = '''
import time, pytest
pytestmark = pytest.mark.gpu

def test_passes_first():
    pass

def test_hangs():
    time.sleep(600)
'''


def test_a_hang_is_cut_by_the_time_budget_and_the_summary_still_prints(tmp_path):
    """The driver ends the GPU tier at 1200 s; the runner's own budget (RNB_GPU_SUITE_BUDGET_S, 1100 s by default; 6 s here) ends a hanging module first, reports the hanging test
    and what was left as failed / not run, and prints the summary with the tests that passed."""
    t = tmp_path / "tests"
    t.mkdir()
    shutil.copy(os.path.join(HERE, "conftest.py"), t / "conftest.py")
    (t / "__init__.py").write_text("")
    (t / "test_gpu_aaa.py").write_text(MOD_HANG)
    (t / "test_gpu_bbb.py").write_text(MOD_B)
    env = dict(os.environ, RNB_GPU_SUITE_BUDGET_S="6")
    env.pop("RNB_GPU_CHILD_RESULTS", None)
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/", "-q", "-m", "gpu", "-p", "no:cacheprovider"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=120)
    out = r.stdout + r.stderr
    assert r.returncode == 1, out
    assert "1 passed" in out and "1 failed" in out and "2 errors" in out, out  # the hanging test failed (killed), the module behind it was not run, the first test's pass stands
    assert "test_hangs" in out and "killed after" in out and "time budget" in out

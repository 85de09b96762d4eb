import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.hookimpl(trylast=True)
def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if os.environ.get("RNB_GPU_CHILD_RESULTS"):
        # a module's child process of the GPU tier (below): its reporter stays (assertion rewriting needs one) but writes nowhere -- the parent session
        # replays the reports and prints the one summary; what the child shows is the RUN / OK lines and whatever native code prints
        tr = config.pluginmanager.get_plugin("terminalreporter")
        if tr is not None:
            from _pytest._io import TerminalWriter
            tr._tw = TerminalWriter(file=open(os.devnull, "w"))


# ---------------------------------------------------------------------------------------------------------------------------------------------
# How the GPU tier runs (`pytest tests -m gpu`, the driver's command).
#
# 1. One short flushed line per test on the real stdout ("RUN <nodeid>" before, "OK|FAILED <nodeid> passed=<n>" after): the TAIL of a log always
#    names the test that was running and the number that had passed, whatever ends the process.
# 2. Every test MODULE runs in its own python process (a child `pytest` over that module's selected node ids, same conftest, results handed back
#    through a JSON-lines file and replayed into this session's reporter, so the final "N passed" line and the exit code are those of one ordinary
#    run). A native abort (SIGABRT from the HIP runtime on a GPU memory fault, std::terminate in a watchdog thread, ...) then fails ONE test -- named,
#    with the signal -- and the module's remaining tests continue in a fresh child; the other modules' results are untouched. It also bounds what one
#    process accumulates (contexts, RCCL communicators, torch's caching allocator) to one module.
# 3. Collection order: the small stage-by-stage parity files first, the full-size file and the multi-process ones last (_GPU_ORDER).
# RNB_GPU_ISOLATE=0 runs everything in the one process (debugging).
# ---------------------------------------------------------------------------------------------------------------------------------------------
_GPU_ORDER = ["test_gpu_parity.py", "test_gpu_half_mode.py", "test_gpu_mesh.py", "test_gpu_deterministic.py", "test_gpu_fullsize.py", "test_gpu_cli.py",
              "test_gpu_rccl.py", "test_gpu_dp_two_process.py"]
_progress = {"passed": 0, "failed": 0}
_CHILD_ENV = "RNB_GPU_CHILD_RESULTS"
_CHILD_WALL_LIMIT_S = 900        # one module's child process
_SUITE_BUDGET_S = float(os.environ.get("RNB_GPU_SUITE_BUDGET_S", "1100"))  # the whole GPU tier (the driver ends the run at 1200 s): a hang must not take the summary line with it
_suite_t0 = [None]


def _is_gpu_run(config):
    m = config.getoption("-m") or ""
    return "gpu" in m and "not gpu" not in m


def _isolating_parent(config):
    return _is_gpu_run(config) and not os.environ.get(_CHILD_ENV) and os.environ.get("RNB_GPU_ISOLATE", "1") != "0" and not config.getoption("collectonly", False)


def _say(line):
    out = sys.__stdout__
    try:
        out.write("\n" + line + "\n")
        out.flush()
    except Exception:
        pass


def _child_log(obj):
    path = os.environ.get(_CHILD_ENV)
    if path:
        with open(path, "a") as f:
            f.write(json.dumps(obj) + "\n")
            f.flush()
            os.fsync(f.fileno())


def pytest_runtest_logstart(nodeid, location):
    if _progress.get("replaying"):
        return
    _say("RUN %s" % nodeid)
    _child_log({"event": "start", "nodeid": nodeid})


def pytest_runtest_logreport(report):
    if _progress.get("replaying"):
        return
    if report.when == "call" and report.passed:
        _progress["passed"] += 1
        _say("OK %s passed=%d failed=%d (%.1fs)" % (report.nodeid, _progress["passed"], _progress["failed"], report.duration))
    elif report.failed:
        _progress["failed"] += 1
        _say("FAILED %s [%s] passed=%d failed=%d" % (report.nodeid, report.when, _progress["passed"], _progress["failed"]))
    if os.environ.get(_CHILD_ENV):
        lr = report.longrepr
        rec = {"event": "report", "nodeid": report.nodeid, "when": report.when, "outcome": report.outcome, "duration": report.duration,
               "longrepr": (list(lr) if isinstance(lr, tuple) else (None if lr is None else report.longreprtext)), "longrepr_is_tuple": isinstance(lr, tuple),
               "sections": [list(s) for s in report.sections]}
        if hasattr(report, "wasxfail"):
            rec["wasxfail"] = report.wasxfail
        _child_log(rec)


def pytest_sessionfinish(session, exitstatus):
    _child_log({"event": "sessionfinish", "exitstatus": int(exitstatus)})


def _replay(item, records, note=None):
    from _pytest.reports import TestReport
    item.ihook.pytest_runtest_logstart(nodeid=item.nodeid, location=item.location)
    for r in records:
        lr = r["longrepr"]
        if r.get("longrepr_is_tuple") and lr is not None:
            lr = tuple(lr)
        extra = {"wasxfail": r["wasxfail"]} if "wasxfail" in r else {}
        rep = TestReport(nodeid=item.nodeid, location=item.location, keywords={k: 1 for k in item.keywords}, outcome=r["outcome"], longrepr=lr, when=r["when"],
                         sections=[tuple(s) for s in r.get("sections", [])], duration=r.get("duration", 0.0), **extra)
        item.ihook.pytest_runtest_logreport(report=rep)
    item.ihook.pytest_runtest_logfinish(nodeid=item.nodeid, location=item.location)


def _run_module_in_children(session, items):
    """Runs `items` (one module's selected tests, in order) in child processes; replays their reports here. Returns False when the session should stop."""
    import signal
    import tempfile
    import time
    config = session.config
    by_id = {it.nodeid: it for it in items}
    pending = [it.nodeid for it in items]
    restarts = 0
    while pending:
        if _suite_t0[0] is None:
            _suite_t0[0] = time.time()
        budget_left = _SUITE_BUDGET_S - (time.time() - _suite_t0[0])
        if budget_left < 1.0:  # report what is left as not run, so that the session still ends with its summary line and the tests that passed
            for nid in pending:
                _progress["replaying"] = True
                try:
                    _replay(by_id[nid], [{"outcome": "failed", "longrepr": "not run: the GPU tier's time budget (%d s, RNB_GPU_SUITE_BUDGET_S) was used up" % _SUITE_BUDGET_S, "when": "setup", "sections": [], "duration": 0.0}])
                finally:
                    _progress["replaying"] = False
                if session.shouldfail or session.shouldstop:
                    return False
            return True
        wall_limit = min(_CHILD_WALL_LIMIT_S, budget_left)
        fd, path = tempfile.mkstemp(prefix="rnb_gpu_child_", suffix=".jsonl")
        os.close(fd)
        env = dict(os.environ)
        env[_CHILD_ENV] = path
        env["RNB_GPU_CHILD_PASSED_BEFORE"] = "%d,%d" % (_progress["passed"], _progress["failed"])
        cmd = [sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", "-m", config.getoption("-m"), "--rootdir", str(config.rootpath)]
        maxfail = config.getoption("maxfail")
        if maxfail:
            cmd += ["--maxfail", str(max(1, maxfail - session.testsfailed))]
        if config.getoption("capture") == "no":
            cmd += ["-s"]
        cmd += pending
        sys.__stdout__.flush()
        sys.__stderr__.flush()
        proc = subprocess.Popen(cmd, cwd=str(config.invocation_params.dir), env=env, start_new_session=True)
        t0 = time.time()
        timed_out = False
        while proc.poll() is None:
            time.sleep(0.2)
            if time.time() - t0 > wall_limit:
                timed_out = True
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except Exception:
                    proc.kill()
                proc.wait()
        rc = proc.returncode
        records = []
        try:
            with open(path) as f:
                for line in f:
                    try:
                        records.append(json.loads(line))
                    except ValueError:
                        pass
        finally:
            try:
                os.unlink(path)
            except OSError:
                pass
        if not records and rc is not None and rc > 0:  # the child never got as far as a test (usage / collection error): nothing to continue with
            for nid in pending:
                _progress["replaying"] = True
                try:
                    _replay(by_id[nid], [{"outcome": "failed", "longrepr": "the child pytest process could not start its tests (exit code %d)" % rc, "when": "setup", "sections": [], "duration": 0.0}])
                finally:
                    _progress["replaying"] = False
                if session.shouldfail or session.shouldstop:
                    return False
            return True
        per_test = {}
        started = []
        finished_session = False
        for r in records:
            if r["event"] == "start":
                started.append(r["nodeid"])
                per_test.setdefault(r["nodeid"], [])
            elif r["event"] == "report":
                per_test.setdefault(r["nodeid"], []).append(r)
            elif r["event"] == "sessionfinish":
                finished_session = True
        done = []
        for nid in pending:
            recs = per_test.get(nid)
            if recs is None:
                break
            whens = [r["when"] for r in recs]
            complete = "teardown" in whens
            if not complete:
                if finished_session:  # -x / maxfail ended the child between this test's reports
                    complete = bool(recs)
                if not complete:
                    break
            _progress["replaying"] = True
            try:
                _replay(by_id[nid], recs)
            finally:
                _progress["replaying"] = False
            for r in recs:
                if r["when"] == "call" and r["outcome"] == "passed":
                    _progress["passed"] += 1
                elif r["outcome"] == "failed":
                    _progress["failed"] += 1
            done.append(nid)
            if session.shouldfail or session.shouldstop:
                return False
        pending = pending[len(done):]
        if not pending:
            break
        if finished_session and rc in (0, 1):
            # the child ended in order but left tests unreported (maxfail inside the child): the session's own maxfail decides
            if session.shouldfail or session.shouldstop:
                return False
        # the child died (or was killed) while `pending[0]` was running -- or before it reported anything at all
        culprit = pending[0]
        if rc is not None and rc < 0:
            try:
                how = "signal %d (%s)" % (-rc, signal.Signals(-rc).name)
            except ValueError:
                how = "signal %d" % -rc
        else:
            how = "exit code %s" % rc
        if timed_out:
            how = "killed after %d s (the wall limit of one module's child process / what was left of the GPU tier's time budget)" % wall_limit
        partial = per_test.get(culprit, [])
        phase = "setup" if not any(r["when"] == "setup" for r in partial) else ("call" if not any(r["when"] == "call" for r in partial) else "teardown")
        text = ("the python process running this test ended by %s during its %s phase (a native abort: HIP runtime / GPU memory fault / std::terminate; "
                "the lines above this report in the log are that process's own output). The module's remaining tests continue in a fresh process." % (how, phase))
        _say("FAILED %s [%s] child process ended by %s passed=%d failed=%d" % (culprit, phase, how, _progress["passed"], _progress["failed"] + 1))
        _progress["failed"] += 1
        _progress["replaying"] = True
        try:
            _replay(by_id[culprit], [r for r in partial if r["outcome"] == "passed" and r["when"] != phase] +
                    [{"outcome": "failed", "longrepr": text, "when": phase, "sections": [], "duration": time.time() - t0}])
        finally:
            _progress["replaying"] = False
        pending = pending[1:]
        restarts += 1
        if session.shouldfail or session.shouldstop:
            return False
        if restarts > 6:
            for nid in pending:
                _progress["replaying"] = True
                try:
                    _replay(by_id[nid], [{"outcome": "failed", "longrepr": "not run: this module's child process died %d times" % restarts, "when": "setup", "sections": [], "duration": 0.0}])
                finally:
                    _progress["replaying"] = False
            return not (session.shouldfail or session.shouldstop)
    return True


@pytest.hookimpl(tryfirst=True)
def pytest_runtestloop(session):
    if not _isolating_parent(session.config):
        return None
    if session.testsfailed and not session.config.option.continue_on_collection_errors:
        raise session.Interrupted("%d error%s during collection" % (session.testsfailed, "s" if session.testsfailed != 1 else ""))
    groups = []
    for item in session.items:
        key = item.nodeid.split("::")[0]
        if not groups or groups[-1][0] != key:
            groups.append((key, []))
        groups[-1][1].append(item)
    for key, items in groups:
        _say("MODULE %s (%d tests) in its own process" % (key, len(items)))
        ok = _run_module_in_children(session, items)
        if session.shouldfail:
            raise session.Failed(session.shouldfail)
        if session.shouldstop:
            raise session.Interrupted(session.shouldstop)
        if not ok:
            break
    return True


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    """Without a GPU a plain `pytest tests` skips the gpu-marked tests instead of failing them. With `-m gpu` on a box
    that has no usable device they still run (and fail loudly): the driver's GPU tier must never pass by skipping."""
    def order(item):
        name = os.path.basename(item.nodeid.split("::")[0])
        return _GPU_ORDER.index(name) if name in _GPU_ORDER else (-1 if "gpu" not in item.keywords else len(_GPU_ORDER))
    items.sort(key=order)  # (stable: the order inside a file stays)
    if os.environ.get("RNB_GPU_CHILD_PASSED_BEFORE"):
        a, b = os.environ["RNB_GPU_CHILD_PASSED_BEFORE"].split(",")
        _progress["passed"], _progress["failed"] = int(a), int(b)
    if _is_gpu_run(config):
        return
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container (GPU tests run with -m gpu on the MI355X box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_lib
    return oracle_lib.functions()


@pytest.fixture(scope="session")
def hip():
    from rnb_neus2_amd import api
    return api.load_library()


# a network small enough for the CPU checker to train in seconds (passed to the testbed as --config small.json)
SMALL_CFG = {
    "encoding": {"n_levels": 4, "log2_hashmap_size": 12, "base_resolution": 16, "top_resolution": 64, "valid_level_scale": 0.02,
                 "base_valid_level_scale": 0.2, "base_training_step": 100},
    "network": {"sdf_bias": -0.1},
    "optimizer": {"decay": 0.95, "nested": {"decay_start": 20000, "decay_interval": 10000, "decay_base": 0.33,
                                            "nested": {"learning_rate": 0.001, "beta1": 0.9, "beta2": 0.99, "epsilon": 1e-15, "l2_reg": 1e-6}}},
    "hyperparams": {"batch_size": 4096, "mask_loss_weight": 1.0, "ek_loss_weight": 0.01},
}


def make_install(root, base_cfg=None):
    """<root>/build/testbed built from the product's testbed_main.cpp but linked against the CPU checker
    (`-include oracle/orc_prefix.h`), plus <root>/utils and <root>/configs/nerf — the layout the binary expects."""
    from tests import oracle_lib
    oracle_lib.functions()
    host = os.path.join(ROOT, "rnb-neus2_amd", "host")
    os.makedirs(os.path.join(root, "build"))
    os.makedirs(os.path.join(root, "configs", "nerf"))
    shutil.copytree(os.path.join(ROOT, "utils"), os.path.join(root, "utils"))
    if base_cfg is None:
        shutil.copy(os.path.join(ROOT, "configs", "nerf", "base.json"), os.path.join(root, "configs", "nerf", "base.json"))
    else:
        with open(os.path.join(root, "configs", "nerf", "base.json"), "w") as f:
            json.dump(base_cfg, f)
    with open(os.path.join(root, "configs", "nerf", "small.json"), "w") as f:
        json.dump(SMALL_CFG, f)
    odir = os.path.join(ROOT, "oracle")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-include", os.path.join(odir, "orc_prefix.h"),
                           os.path.join(host, "testbed_main.cpp"), "-o", os.path.join(root, "build", "testbed"), "-L" + odir, "-lorc", "-lz", "-Wl,-rpath," + odir])
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(host, "dump_dataset.cpp"), "-o", os.path.join(root, "build", "dump_dataset"), "-lz"])
    return root


@pytest.fixture(scope="session")
def install(tmp_path_factory):
    return make_install(tmp_path_factory.mktemp("install"))


@pytest.fixture(scope="session")
def small_install(tmp_path_factory):
    """Same, but the default config (configs/nerf/base.json) is the small network — for pipeline tests, which cannot pass --config."""
    return make_install(tmp_path_factory.mktemp("small_install"), base_cfg=SMALL_CFG)

import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    """Without a GPU a plain `pytest tests` skips the gpu-marked tests instead of failing them. With `-m gpu` on a box
    that has no usable device they still run (and fail loudly): the driver's GPU tier must never pass by skipping."""
    if "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or ""):
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

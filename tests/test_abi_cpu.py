"""CPU tests of the boundary: the C-ABI library loads without a GPU and exports every symbol the header declares;
the CPU checker exports the same set under orc_; the Python host side maps every prototype."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "rnb_neus2.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(rnb_[a-z_0-9]+)\s*\(", src)
    return sorted(set(n for n in names if n not in ("rnb_ctx",)))


def test_header_lists_the_expected_entry_points():
    names = header_functions()
    for must in ("rnb_create", "rnb_destroy", "rnb_init_params", "rnb_set_dataset", "rnb_update_density_grid", "rnb_forward_infer",
                 "rnb_generate_training_samples", "rnb_compute_loss", "rnb_forward_backward", "rnb_optimizer_step", "rnb_train_step"):
        assert must in names
    assert len(names) >= 35


def test_hip_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from rnb_neus2_amd import api
    lib = C.CDLL(api.library_path())
    missing = [n for n in header_functions() if not hasattr(lib, n)]
    assert not missing, missing
    # callable without a GPU: version + defaults (no compute)
    fns = api.load_library()
    assert fns.abi_version() == 5
    cfg = api.default_config()
    assert (cfg.n_levels, cfg.log2_hashmap_size, cfg.target_batch_size, cfg.seed) == (14, 19, 1 << 18, 1337)
    assert abs(cfg.per_level_scale - 1.45242) < 1e-4  # exp(ln(2048/16)/13), testbed.cu:2320-2323


def test_host_library_exports_every_declared_symbol():
    """include/rnb_host.h (PNG I/O + mesh ray casting for the preparation stages) vs librnb_host.so."""
    import __graft_entry__ as g
    g.build()
    from rnb_neus2_amd import hostlib
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rnb_host.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(rnb_[a-z_0-9]+)\s*\(", src)))
    assert len(names) == 8, names
    lib = C.CDLL(hostlib.library_path())
    assert not [n for n in names if not hasattr(lib, n)]


def test_oracle_exports_the_same_abi():
    from tests import oracle_lib
    oracle_lib.functions()
    lib = C.CDLL(os.path.join(ROOT, "oracle", "liborc.so"))
    missing = [n for n in header_functions() if not hasattr(lib, "orc_" + n[4:])]
    assert not missing, missing


def test_python_prototypes_cover_the_header():
    from rnb_neus2_amd import _abi
    declared = set(n[4:] for n in header_functions())
    assert declared == set(_abi.PROTOTYPES), declared ^ set(_abi.PROTOTYPES)


def test_config_struct_layout_matches_the_header():
    """sizeof/offsets of the ctypes mirrors against a tiny C program compiled from the header."""
    import subprocess
    import tempfile
    from rnb_neus2_amd import _abi
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "rnb_neus2.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(rnb_config), offsetof(rnb_config, target_batch_size), offsetof(rnb_config, learning_rate),
         offsetof(rnb_config, world_size), sizeof(rnb_view), sizeof(rnb_step_stats), offsetof(rnb_step_stats, loss), offsetof(rnb_config, accumulate));
  return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write(prog)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        got = [int(x) for x in subprocess.check_output([exe]).split()]
    want = [C.sizeof(_abi.Config), _abi.Config.target_batch_size.offset, _abi.Config.learning_rate.offset, _abi.Config.world_size.offset,
            C.sizeof(_abi.View), C.sizeof(_abi.StepStats), _abi.StepStats.loss.offset, _abi.Config.accumulate.offset]
    assert got == want


def test_product_refuses_to_run_without_the_hip_library(monkeypatch):
    from rnb_neus2_amd import api
    monkeypatch.setattr(api, "_FUNCS", None)
    monkeypatch.setattr(api, "_LIB_NAME", "librnb_neus2_hip_missing.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        api.load_library()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "rnb-neus2_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle_lib" not in text and "liborc" not in text and "orc_" not in text.replace("force_", ""), f

"""Loads oracle/liborc.so (the CPU checker) behind the same Python ``Context`` class the product uses.
Test infrastructure only: nothing under rnb-neus2_amd/ imports this module."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_F = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def functions():
    global _F
    if _F is None:
        so = os.path.join(ORACLE_DIR, "liborc.so")
        src = os.path.join(ORACLE_DIR, "rnb_oracle.cpp")
        if (not os.path.exists(so)) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
            build()
        from rnb_neus2_amd import _abi
        _F = _abi.declare(C.CDLL(so), "orc_")
    return _F


def context(**overrides):
    from rnb_neus2_amd import api
    f = functions()
    return api.Context(cfg=api.default_config(f, **overrides), fns=f)

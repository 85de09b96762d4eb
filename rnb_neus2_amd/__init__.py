"""Importable alias of the package directory ``rnb-neus2_amd/`` (a hyphen cannot appear in a module name).

The sources (Python host side + ``csrc/`` HIP kernels + the built ``librnb_neus2_hip.so``) live in
``rnb-neus2_amd/``; this shim only extends the package search path to that directory.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "rnb-neus2_amd")
__path__.append(_real)

from .api import *  # noqa: E402,F401,F403
from . import api as _api  # noqa: E402

__all__ = _api.__all__

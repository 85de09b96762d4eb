"""rnb-neus2_amd — MI355X-native RNb-NeuS2 training hot path (HIP kernels in csrc/, host side in api.py).

Import it as ``rnb_neus2_amd`` (the sibling shim package forwards to this directory).
"""

"""Operands and an independent statement of RNB_PRIM_DW_SLICED (include/rnb_neus2.h): the weight-gradient GEMM dW[o][i] = sum_s Y[o][s] X[i][s] as the reference's
half arithmetic forms it -- CUTLASS split-K slices of 4096 samples with half accumulators (tcnn cutlass_matmul.h:83), one rounding per 16-sample k-step, the slices'
results reduced in half (cutlass_matmul.h:315-322). `model` states that in numpy, element by element in sample order, sharing nothing with oracle/rnb_oracle.cpp's
emulated_dw or the library's k_dw_sliced; tests/test_oracle_cpu.py holds the oracle to it, tests/test_gpu_half_mode.py the library to the oracle, bit for bit."""
import numpy as np

S = 8256  # two whole slices and one of 64 samples


def items(seed=0):
    """uint32 [n, 4 + 8 * S / 2]: magnitudes from 'every sum is exact' to 'the half accumulator loses most addends' and 'it overflows'; one item with Y = 1."""
    rng = np.random.default_rng(seed)
    out = []
    for scale_y, scale_x, ones, sparse in ((1.0, 1.0, 0, 0), (0.05, 0.02, 0, 0), (8.0, 16.0, 0, 0), (1e-3, 1e-2, 0, 0), (1.0, 1.0, 1, 0), (0.3, 0.7, 0, 1), (60.0, 60.0, 0, 0)):
        y = (rng.standard_normal((4, S)) * scale_y).astype(np.float16)
        x = (rng.standard_normal((4, S)) * scale_x).astype(np.float16)
        if sparse:  # relu-masked operands: exact zeros, signed
            y[rng.random((4, S)) < 0.6] = 0
            x[rng.random((4, S)) < 0.3] = -0.0
        if scale_y == 60.0:  # all positive: the accumulator reaches 65504 and stays infinite
            y, x = np.abs(y), np.abs(x)
        head = np.array([ones, 0, 0, 0], dtype=np.uint32)
        out.append(np.concatenate([head, np.concatenate([y, x]).reshape(-1).view(np.uint32)]))
    return np.stack(out)


def model(item):
    """float32 bit patterns [16] of the half results, from the item's words."""
    rows = item[4:].view(np.float16).reshape(8, S)
    y, x = rows[:4].astype(np.float32), rows[4:].astype(np.float32)
    if item[0] & 1:
        y = np.zeros_like(y)
        y[0] = 1.0
    total = np.zeros((4, 4), np.float16)
    with np.errstate(over="ignore"):
        for s0 in range(0, S, 4096):
            acc = np.zeros((4, 4), np.float16)
            for k0 in range(s0, min(S, s0 + 4096), 16):
                part = np.zeros((4, 4), np.float32)
                for s in range(k0, k0 + 16):
                    part = part + y[:, s, None] * x[None, :, s]  # fp32; a product of two halfs is exact
                acc = (acc.astype(np.float32) + part).astype(np.float16)
            total = (total.astype(np.float32) + acc.astype(np.float32)).astype(np.float16)
    return total.astype(np.float32).reshape(-1).view(np.uint32)

"""CPU, build container only: the oracle against the REAL reference library
(oracle/_ref/libref_canon.so, compiled from the reference's own sources by
oracle/Makefile) on fresh random inputs.  Skipped where oracle/_ref is absent."""
import ctypes as C

import numpy as np
import pytest

import reflib
from oracle import orc_net as O

pytestmark = [pytest.mark.ref,
              pytest.mark.skipif(not reflib.available("canon"), reason="oracle/_ref not built (no /root/reference here)")]
FP = C.POINTER(C.c_float)
F = C.c_float


def fp(a):
    return a.ctypes.data_as(FP)


def test_conv_layer_paths_random():
    """im2col + gemm at a real yolov4-tiny layer shape, all four gemm variants."""
    R, L = reflib.lib("canon"), O.lib()
    R.gemm_cpu.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, FP, C.c_int,
                           FP, C.c_int, C.c_float, FP, C.c_int]
    rng = np.random.default_rng(11)
    c, h, w, k = 32, 13, 13, 3
    im = rng.uniform(-1, 1, (c, h, w)).astype(np.float32)
    col_r = np.zeros((c * k * k, h * w), np.float32)
    col_o = np.zeros_like(col_r)
    R.im2col_cpu_ext(fp(im), c, h, w, k, k, 1, 1, 1, 1, 1, 1, fp(col_r))
    L.orc_im2col_ext(fp(im), c, h, w, k, k, 1, 1, 1, 1, 1, 1, fp(col_o))
    assert np.array_equal(col_r, col_o)
    for ta, tb in ((0, 0), (0, 1), (1, 0), (1, 1)):
        m, n, kk = 24, 57, 96
        A = rng.uniform(-1, 1, (kk, m) if ta else (m, kk)).astype(np.float32)
        B = rng.uniform(-1, 1, (n, kk) if tb else (kk, n)).astype(np.float32)
        C1 = rng.uniform(-1, 1, (m, n)).astype(np.float32)
        C2 = C1.copy()
        R.gemm_cpu(ta, tb, m, n, kk, 1.0, fp(A), A.shape[1], fp(B), B.shape[1], 1.0, fp(C1), n)
        L.orc_gemm(ta, tb, m, n, kk, F(1.0), fp(A), A.shape[1], fp(B), B.shape[1], F(1.0), fp(C2), n)
        assert np.array_equal(C1, C2), (ta, tb)


def test_mish_and_gradients_random():
    R, L = reflib.lib("canon"), O.lib()
    rng = np.random.default_rng(12)
    x = (rng.standard_normal(200000) * 6).astype(np.float32)
    y1, y2, a1, a2 = (np.zeros_like(x) for _ in range(4))
    R.activate_array_mish(fp(x), x.size, fp(a1), fp(y1))
    L.orc_activate_array_mish(fp(x), x.size, fp(a2), fp(y2))
    assert np.array_equal(y1, y2)
    d1 = rng.uniform(-1, 1, x.size).astype(np.float32)
    d2 = d1.copy()
    R.gradient_array_mish(x.size, fp(x), fp(d1))
    L.orc_gradient_array_mish(x.size, fp(x), fp(d2))
    assert np.array_equal(d1, d2)

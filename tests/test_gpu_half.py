"""BASELINE config C5: fp16-operand / fp32-accumulate convolutions.  Oracle = the CPU
fp32 path with the eligible layers' inputs and weights pre-rounded to fp16 (the
reference has no CPU fp16 path: SURVEY.md section 8 row a17)."""
import ctypes as C
import os

import numpy as np
import pytest

import netutil
import synth
import util
from oracle import orc_net as O
from test_gpu_ops import orc_conv

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", [(2, 32, 19, 19, 64, 3, 1, 1, "LEAKY"), (2, 64, 26, 26, 128, 3, 2, 1, "MISH"),
                                  (1, 512, 19, 19, 1024, 3, 1, 1, "LEAKY"), (3, 16, 9, 7, 8, 3, 1, 1, "LINEAR")])
def test_conv_half_vs_rounded_oracle(gpu, case):
    batch, c, h, w, n, size, stride, pad, actname = case
    act = getattr(O, actname)
    rng = np.random.default_rng(util.seed_of(case))
    x = rng.uniform(-1, 1, (batch, c, h, w)).astype(np.float32)
    wt = (rng.uniform(-1, 1, (n, c, size, size)) * np.sqrt(2.0 / (size * size * c))).astype(np.float32)
    bias = rng.uniform(-.5, .5, n).astype(np.float32)
    x16 = np.ascontiguousarray(x.astype(np.float16).astype(np.float32))
    w16 = np.ascontiguousarray(wt.astype(np.float16).astype(np.float32))
    ref, _ = orc_conv(x16, w16, bias, batch, c, h, w, n, size, stride, pad, act)
    L = gpu.lib()
    L.dk_conv_forward_half.argtypes = [C.POINTER(gpu.DkConvDesc)] + [C.c_void_p] * 7
    L.dk_conv_forward_half.restype = C.c_int
    d = gpu.DkConvDesc(batch, c, h, w, n, 1, size, stride, stride, 1, pad, act)
    dx, dw, db = gpu.DeviceArray(x), gpu.DeviceArray(wt), gpu.DeviceArray(bias)
    dy = gpu.DeviceArray(n=ref.size)
    assert L.dk_conv_forward_half(C.byref(d), dx.ptr, dw.ptr, db.ptr, dy.ptr, None, None, None) == 0
    util.assert_close(dy.numpy().reshape(ref.shape), ref, "fp16-operand conv %s" % (case,))
    # and it is NOT the fp32 result (the rounding is really applied)
    full, _ = orc_conv(x, wt, bias, batch, c, h, w, n, size, stride, pad, act)
    assert np.abs(full - ref).max() > 1e-5


@pytest.mark.parametrize("case", [(2, 32, 19, 19, 64, "LEAKY"), (1, 512, 19, 19, 1024, "LEAKY"), (2, 64, 32, 32, 72, "MISH"),
                                  (1, 128, 64, 64, 128, "MISH"), (1, 16, 128, 128, 40, "LINEAR"), (3, 48, 16, 16, 136, "LEAKY")])
def test_conv_half_direct_vs_rounded_oracle(gpu, case):
    """The patch-in-LDS fp16 kernel (weights packed [n][c/16][tap][16]) against the CPU fp32 path
    on fp16-rounded inputs and weights, with and without a residual.  The products of fp16 operands are exact in
    fp32 and both sides accumulate in fp32, but in different orders (the kernel walks 16-channel stages tap-major, the
    oracle's gemm_nn is k-ascending): at K = 4608 that is 1.5e-5 x rms on single elements (measured), so the
    element-wise floor is 2e-5 x rms here instead of util's 1e-5."""
    batch, c, h, w, n, actname = case
    act = getattr(O, actname)
    rng = np.random.default_rng(util.seed_of(case))
    x = rng.uniform(-1, 1, (batch, c, h, w)).astype(np.float32)
    wt = (rng.uniform(-1, 1, (n, c, 3, 3)) * np.sqrt(2.0 / (9 * c))).astype(np.float32)
    bias = rng.uniform(-.5, .5, n).astype(np.float32)
    res = rng.uniform(-1, 1, (batch, n, h, w)).astype(np.float32)
    x16 = np.ascontiguousarray(x.astype(np.float16).astype(np.float32))
    w16 = np.ascontiguousarray(wt.astype(np.float16).astype(np.float32))
    ref, _ = orc_conv(x16, w16, bias, batch, c, h, w, n, 3, 1, 1, act)
    L = gpu.lib()
    VP = C.c_void_p
    L.dk_conv_half_direct_weights_size.argtypes = [C.POINTER(gpu.DkConvDesc)]
    L.dk_conv_half_direct_weights_size.restype = C.c_size_t
    L.dk_conv_half_pack_weights.argtypes = [C.POINTER(gpu.DkConvDesc), VP, VP, VP]
    L.dk_conv_forward_half_packed.argtypes = [C.POINTER(gpu.DkConvDesc)] + [VP] * 6
    d = gpu.DkConvDesc(batch, c, h, w, n, 1, 3, 1, 1, 1, 1, act)
    halves = L.dk_conv_half_direct_weights_size(C.byref(d))
    assert halves == n * c * 9
    dx, dw, db, dr = gpu.DeviceArray(x), gpu.DeviceArray(wt), gpu.DeviceArray(bias), gpu.DeviceArray(res)
    dp = gpu.DeviceArray(n=(halves + 1) // 2)
    dy = gpu.DeviceArray(n=ref.size)
    assert L.dk_conv_half_pack_weights(C.byref(d), dw.ptr, dp.ptr, None) == 0
    # packed layout: [n][c/16][tap][16] halves of the RNE-rounded weights
    packed = np.frombuffer(dp.numpy().tobytes(), dtype=np.float16)[:halves].reshape(n, c // 16, 9, 16)
    want = wt.astype(np.float16).reshape(n, c // 16, 16, 9).transpose(0, 1, 3, 2)
    assert np.array_equal(packed, want)
    assert L.dk_conv_forward_half_packed(C.byref(d), dx.ptr, dp.ptr, db.ptr, dy.ptr, dr.ptr, None) == 0
    util.assert_close(dy.numpy().reshape(ref.shape), ref + res, "fp16 direct conv + residual %s" % (case,), atol_rms=2e-5)
    assert L.dk_conv_forward_half_packed(C.byref(d), dx.ptr, dp.ptr, db.ptr, dy.ptr, None, None) == 0
    util.assert_close(dy.numpy().reshape(ref.shape), ref, "fp16 direct conv %s" % (case,), atol_rms=2e-5)


def test_csp_b1_half_vs_rounded_oracle(gpu, tmp_path):
    name = "yolov4-csp"
    wpath = str(tmp_path / "w.weights")
    netutil.synth_weights_for(gpu, name, wpath)
    L = gpu.lib()
    L.DkSetHalf.argtypes = [C.c_int]
    L.DkSetHalf(1)
    try:
        net = netutil.DkNet(gpu, netutil.cfg_path(name), wpath)
    finally:
        L.DkSetHalf(0)
    onet = O.load_network(netutil.cfg_path(name), wpath, batch=1)
    x = synth.make_input(1, net.c, net.h, net.w)
    net.predict(x)
    O.forward(onet, x, half=True)
    nhalf = sum(1 for l in onet.layers if O.half_eligible(l))
    assert nhalf == 43  # 43 of 115 convs are fp16-eligible
    worst = 0.0
    for i, l in enumerate(onet.layers):
        if l.type == O.YOLO:
            st = util.assert_close(net.output(i), l.output, "csp half head %d" % i, rel=2e-4, atol_rms=1e-4)
            worst = max(worst, st["max_abs_over_rms"])
    print("yolov4-csp fp16-operand heads: worst max|d|/rms %.3g" % worst)
    net.close()


@pytest.mark.parametrize("name", ["yolov4-csp", "yolov4x-mish"])
def test_c5_csp_512_b32_half_at_baseline_size(gpu, tmp_path, name):
    """BASELINE config C5 at its full size: yolov4-csp and yolov4x-mish 512x512, batch 32, fp16 operands on the
    layers the reference's rule admits (43 of csp's convs).  Item 17's decoded heads vs the fp16-pre-rounded CPU oracle (the only possible
    oracle: the reference has no CPU fp16 path -- PARITY UNPINNED by construction for this row), and
    two size-independent properties at the full batch: items with identical inputs give bitwise
    identical heads wherever they sit in the batch, and the graph replay reproduces the eager run."""
    B, K = 32, 17
    wpath = str(tmp_path / "w.weights")
    netutil.synth_weights_for(gpu, name, wpath)
    L = gpu.lib()
    L.DkSetHalf.argtypes = [C.c_int]
    L.DkSetHalf(1)
    try:
        net = netutil.DkNet(gpu, netutil.cfg_path(name), wpath, batch=B)
    finally:
        L.DkSetHalf(0)
    assert (net.w, net.h) == (512, 512)
    x = synth.make_input(B, net.c, net.h, net.w)
    x[5] = x[K]
    x[31] = x[K]
    net.predict(x)
    heads = {i: net.output(i).copy() for i in range(net.n) if net.info(i)["type"] == O.YOLO}
    for i, h in heads.items():
        assert np.array_equal(h[5], h[K]) and np.array_equal(h[31], h[K]), "head %d depends on the batch position" % i
    net.predict(x)   # graph replay
    for i, h in heads.items():
        assert np.array_equal(net.output(i), h), "head %d: graph replay differs from the eager run" % i
    onet = O.load_network(netutil.cfg_path(name), wpath, batch=1)
    O.forward(onet, x[K:K + 1], half=True)
    worst = 0.0
    for i, l in enumerate(onet.layers):
        if l.type == O.YOLO:
            st = util.assert_close(heads[i][K:K + 1], l.output, "%s b32 half head %d item %d" % (name, i, K), rel=2e-4, atol_rms=1e-4)
            worst = max(worst, st["max_abs_over_rms"])
    print("%s 512 b32 fp16 operands: item %d heads worst max|d|/rms %.3g" % (name, K, worst))
    net.close()

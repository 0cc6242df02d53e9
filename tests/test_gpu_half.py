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
    rng = np.random.default_rng(abs(hash(case)) & 0xFFFF)
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

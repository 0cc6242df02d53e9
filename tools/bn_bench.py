#!/usr/bin/env python3
"""Dev tool (GPU box): dk_bn_act_backward (the two fused batch-norm backward kernels) and dk_bn_forward_train on yolov4's
C4 shapes: time per call and GB/s on the algorithmic bytes (backward: delta + x read twice, delta written = 5 streams;
forward: x read twice, output written = 3 streams).  usage: bn_bench.py [iters]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import darknet_amd as dk  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
L = dk.lib()
L.cuda_set_device(0)
VP, i = C.c_void_p, C.c_int
L.dk_bn_act_backward.argtypes = [VP] * 10 + [i, i, i, i, VP]
L.dk_bn_forward_train.argtypes = [VP] * 11 + [i, i, i, i, i, VP]
rng = np.random.default_rng(0)
tot_b = tot_f = 0.0
for (b, c, hw, act, cnt) in ((8, 32, 608 * 608, 17, 1), (8, 64, 304 * 304, 17, 7), (8, 128, 152 * 152, 17, 4), (8, 64, 152 * 152, 17, 8),
                             (8, 128, 76 * 76, 17, 19), (8, 256, 76 * 76, 17, 4), (8, 256, 38 * 38, 17, 19), (8, 512, 38 * 38, 17, 4),
                             (8, 512, 19 * 19, 8, 14), (8, 1024, 19 * 19, 8, 8), (8, 256, 38 * 38, 8, 10), (8, 128, 76 * 76, 8, 8)):
    n = b * c * hw
    x = dk.DeviceArray(rng.normal(0, 1, n).astype(np.float32))
    d = dk.DeviceArray(rng.normal(0, 1, n).astype(np.float32))
    o = dk.DeviceArray(n=n)
    mean, var = dk.DeviceArray(np.zeros(c, np.float32)), dk.DeviceArray(np.ones(c, np.float32))
    rm, rv = dk.DeviceArray(np.zeros(c, np.float32)), dk.DeviceArray(np.ones(c, np.float32))
    sc, bi = dk.DeviceArray(np.ones(c, np.float32)), dk.DeviceArray(np.zeros(c, np.float32))
    md, vd, su, bu = (dk.DeviceArray(np.zeros(c, np.float32)) for _ in range(4))

    def bwd():
        assert L.dk_bn_act_backward(d.ptr, x.ptr, mean.ptr, var.ptr, sc.ptr, bi.ptr, md.ptr, vd.ptr, su.ptr, bu.ptr, b, c, hw, act, None) == 0

    def fwd():
        assert L.dk_bn_forward_train(x.ptr, None, None, None, o.ptr, mean.ptr, var.ptr, rm.ptr, rv.ptr, sc.ptr, bi.ptr, b, c, hw, act, 1, None) == 0
    res = []
    for fn, streams in ((bwd, 5), (fwd, 3)):
        for _ in range(3):
            fn()
        dk._sync()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        dk._sync()
        us = (time.perf_counter() - t0) / iters * 1e6
        res.append((us, streams * 4.0 * n / us / 1e3))
    tot_b += res[0][0] * cnt
    tot_f += res[1][0] * cnt
    print("b%d c%-4d hw %-6d act %2d x%-2d | backward %7.1f us %5.0f GB/s | forward %7.1f us %5.0f GB/s" % (b, c, hw, act, cnt, res[0][0], res[0][1], res[1][0], res[1][1]))
    for a in (x, d, o):
        a.free()
print("yolov4 b=8 step estimate (counts = layers of that shape): backward %.2f ms, forward %.2f ms" % (tot_b / 1e3, tot_f / 1e3))

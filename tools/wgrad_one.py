#!/usr/bin/env python3
"""Dev tool: run ONE weight-gradient shape repeatedly (for rocprofv3 --pmc).
usage: wgrad_one.py batch c h w n size stride pad iters"""
import ctypes as C
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import darknet_amd as dk

b, c, h, w, n, size, stride, pad, iters = map(int, sys.argv[1:10])
L = dk.lib()
L.cuda_set_device(0)
L.dk_conv_backward_weights.argtypes = [C.POINTER(dk.DkConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
L.dk_conv_backward_weights.restype = C.c_int
rng = np.random.default_rng(0)
d = dk.DkConvDesc(b, c, h, w, n, 1, size, stride, stride, 1, pad, 4)
oh, ow = dk.conv_out_dims(h, w, size, stride, stride, pad)
dx = dk.DeviceArray(rng.uniform(-1, 1, b * c * h * w).astype(np.float32))
dd = dk.DeviceArray(rng.uniform(-1, 1, b * n * oh * ow).astype(np.float32))
dw = dk.DeviceArray(n=n * c * size * size)
for _ in range(iters):
    assert L.dk_conv_backward_weights(C.byref(d), dx.ptr, dd.ptr, dw.ptr, None) == 0
dk._sync()
print("done")

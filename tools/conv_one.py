#!/usr/bin/env python3
"""Dev tool: run ONE conv shape with ONE tile config repeatedly (for rocprofv3 --pmc).
usage: conv_one.py batch c h w n size stride pad act cfg iters"""
import ctypes as C
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import darknet_amd as dk

b, c, h, w, n, size, stride, pad, act, cfg, iters = map(int, sys.argv[1:12])
L = dk.lib()
L.cuda_set_device(0)
rng = np.random.default_rng(0)
d = dk.DkConvDesc(b, c, h, w, n, 1, size, stride, stride, 1, pad, act)
oh, ow = dk.conv_out_dims(h, w, size, stride, stride, pad)
dx = dk.DeviceArray(rng.uniform(-1, 1, b * c * h * w).astype(np.float32))
dw = dk.DeviceArray((rng.uniform(-1, 1, n * c * size * size) * 0.05).astype(np.float32))
db = dk.DeviceArray(rng.uniform(-1, 1, n).astype(np.float32))
dy = dk.DeviceArray(n=b * n * oh * ow)
nu = L.dk_conv_wino_weights_size(C.byref(d))
if nu:   # Winograd configurations need the transformed filters
    du = dk.DeviceArray(n=nu)
    L.dk_conv_wino_transform_weights(C.byref(d), dw.ptr, du.ptr, None)
    L.dk_conv_wino_register(dw.ptr, du.ptr)
L.dk_conv_force_config(cfg)
for _ in range(iters):
    L.dk_conv_forward(C.byref(d), dx.ptr, dw.ptr, db.ptr, dy.ptr, None, None, None)
dk._sync()
print("done", L.dk_conv_config_name(cfg).decode())

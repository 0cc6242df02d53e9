#!/usr/bin/env python3
"""Dev tool: time a few conv shapes x configs x activations (GPU box)."""
import ctypes as C
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import darknet_amd as dk
L = dk.lib(); L.cuda_set_device(0)
rng = np.random.default_rng(0)
SHAPES = [  # b c h w n size stride pad
    (16, 128, 76, 76, 128, 3, 1, 1), (16, 256, 38, 38, 512, 3, 1, 1), (16, 512, 19, 19, 1024, 3, 1, 1),
    (16, 64, 304, 304, 64, 1, 1, 0), (16, 128, 76, 76, 128, 1, 1, 0), (16, 512, 19, 19, 512, 1, 1, 0),
    (16, 3, 608, 608, 32, 3, 1, 1), (16, 32, 608, 608, 64, 3, 2, 1)]
ncfg = L.dk_conv_force_config(-1)
iters = 10
for (b, c, h, w, n, size, stride, pad) in SHAPES:
    oh, ow = dk.conv_out_dims(h, w, size, stride, stride, pad)
    dx = dk.DeviceArray(rng.uniform(-1, 1, b * c * h * w).astype(np.float32))
    dw = dk.DeviceArray((rng.uniform(-1, 1, n * c * size * size) * 0.05).astype(np.float32))
    db = dk.DeviceArray(rng.uniform(-1, 1, n).astype(np.float32))
    dy = dk.DeviceArray(n=b * n * oh * ow)
    for act in (8, 8 | 0x100, 8 | 0x200, 8 | 0x300):
        d = dk.DkConvDesc(b, c, h, w, n, 1, size, stride, stride, 1, pad, act)
        res = []
        for cfg in range(ncfg):
            L.dk_conv_force_config(cfg)
            L.dk_conv_forward(C.byref(d), dx.ptr, dw.ptr, db.ptr, dy.ptr, None, None, None)
            L.dk_profile_enable(1)
            for _ in range(iters):
                L.dk_conv_forward(C.byref(d), dx.ptr, dw.ptr, db.ptr, dy.ptr, None, None, None)
            out = (C.c_double * 192)()
            L.dk_profile_read(out, 128)
            L.dk_profile_enable(0)
            res.append(sum(out[(cfg * 4 + v) * 3 + 1] for v in range(4)) / sum(out[(cfg * 4 + v) * 3 + 2] for v in range(4)))
        print("c%-4d %3dx%-3d n%-4d k%d s%d act%-4x | " % (c, h, w, n, size, stride, act) + " ".join("%6.1f" % r for r in res), flush=True)
    for a in (dx, dw, db, dy): a.free()

#!/usr/bin/env python3
"""Dev tool (GPU box): what does the activation in the conv epilogue cost?  One 1x1 shape, one gather configuration,
timed with LINEAR / LEAKY / MISH (fast) epilogues, interleaved rounds in one process.
usage: act_cost.py batch c h w n config_name [rounds]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import darknet_amd as dk  # noqa: E402

b, c, h, w, n = map(int, sys.argv[1:6])
cfgname = sys.argv[6]
rounds = int(sys.argv[7]) if len(sys.argv) > 7 else 5
L = dk.lib()
L.cuda_set_device(0)
rng = np.random.default_rng(0)
dx = dk.DeviceArray(rng.uniform(-1, 1, b * c * h * w).astype(np.float32))
dw = dk.DeviceArray((rng.uniform(-1, 1, n * c) * 0.05).astype(np.float32))
db = dk.DeviceArray(rng.uniform(-1, 1, n).astype(np.float32))
dy = dk.DeviceArray(n=b * n * h * w)
ncfg = L.dk_conv_force_config(-1)
cfg = [i for i in range(ncfg) if L.dk_conv_config_name(i).decode() == cfgname][0]
L.dk_conv_force_config(cfg)
res = {}
for r in range(rounds):
    for act, nm in ((4, "linear"), (8, "leaky"), (17, "mish")):
        d = dk.DkConvDesc(b, c, h, w, n, 1, 1, 1, 1, 1, 0, act)
        for _ in range(3):
            L.dk_conv_forward(C.byref(d), dx.ptr, dw.ptr, db.ptr, dy.ptr, None, None, None)
        dk._sync()
        t0 = time.perf_counter()
        for _ in range(50):
            L.dk_conv_forward(C.byref(d), dx.ptr, dw.ptr, db.ptr, dy.ptr, None, None, None)
        dk._sync()
        res.setdefault(nm, []).append((time.perf_counter() - t0) / 50 * 1e6)
print("%s b%d c%d %dx%d n%d:" % (cfgname, b, c, h, w, n), "  ".join("%s %.1f us (min %.1f)" % (k, np.median(v), min(v)) for k, v in res.items()))

#!/usr/bin/env python3
"""Dev tool (GPU box, DK_LIB = a -DDK_WSTAMP=1 build): phase timeline of the Winograd kernel (schedule 0) on one shape.
usage: DK_LIB=build_abl/libdk_wstamp.so wino_stamps.py batch c h w n"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import darknet_amd as dk  # noqa: E402

b, c, h, w, n = map(int, sys.argv[1:6])
cfgname = sys.argv[6] if len(sys.argv) > 6 else "wino_64x64"
act = int(sys.argv[7]) if len(sys.argv) > 7 else 17      # 17 mish, 8 leaky, 4 linear (dk_device_math.h)
with_res = len(sys.argv) > 8 and sys.argv[8] == "res"
L = dk.lib()
L.cuda_set_device(0)
rng = np.random.default_rng(0)
d = dk.DkConvDesc(b, c, h, w, n, 1, 3, 1, 1, 1, 1, act)
dx = dk.DeviceArray(rng.uniform(-1, 1, b * c * h * w).astype(np.float32))
dw = dk.DeviceArray((rng.uniform(-1, 1, n * c * 9) * 0.05).astype(np.float32))
db = dk.DeviceArray(rng.uniform(-1, 1, n).astype(np.float32))
dy = dk.DeviceArray(n=b * n * h * w)
dr = dk.DeviceArray(rng.uniform(-1, 1, b * n * h * w).astype(np.float32)) if with_res else None
du = dk.DeviceArray(n=L.dk_conv_wino_weights_size(C.byref(d)))
L.dk_conv_wino_transform_weights(C.byref(d), dw.ptr, du.ptr, None)
L.dk_conv_wino_register(dw.ptr, du.ptr)
ncfg = L.dk_conv_force_config(-1)
cfg = [i for i in range(ncfg) if L.dk_conv_config_name(i).decode() == cfgname][0]
L.dk_conv_force_config(cfg)
for _ in range(5):
    L.dk_conv_forward(C.byref(d), dx.ptr, dw.ptr, db.ptr, dy.ptr, dr.ptr if dr else None, None, None)
dk._sync()
buf = np.zeros(64 * 8 * 160, np.int64)
L.dk_wino_stamps_read.argtypes = [C.c_void_p, C.c_int]
L.dk_wino_stamps_read(buf.ctypes.data, buf.size)
s = buf.reshape(64, 8, 160)
nst = min(c // 4, 36)   # stamps of the first 36 stages fit the buffer
T = lambda i: s[:, :, i].astype(np.float64)

def show(name, a):
    x, dm = a[:, :4], a[:, 4:]
    print("%-46s xf waves med %8.0f | dma waves med %8.0f | all min %8.0f max %8.0f" % (name, np.median(x), np.median(dm), a.min(), a.max()))
print("%s shape b%d c%d %dx%d n%d act %d%s: %d stages stamped; cycles (s_memtime = shader clock)" % (cfgname, b, c, h, w, n, act, " +residual" if with_res else "", nst))
show("entry -> LDS zeroed (before barrier)", T(1) - T(0))
show("zero barrier", T(2) - T(1))
show("prologue DMA issue + wait", T(3) - T(2))
show("barrier", T(4) - T(3))
show("V(0) transform (to first loop top)", T(8) - T(4))
tops = np.stack([T(8 + 4 * t) for t in range(nst)], -1)
show("stage period (top to top), median stage", np.median(np.diff(tops, axis=-1), -1))
show("  top -> DMA landed (vmcnt wait)", np.median(np.stack([T(9 + 4 * t) - T(8 + 4 * t) for t in range(nst)], -1), -1))
show("  barrier wait", np.median(np.stack([T(10 + 4 * t) - T(9 + 4 * t) for t in range(nst)], -1), -1))
show("  released -> stamp 11 (sched 0: bundle issued; 3: MFMAs issued)", np.median(np.stack([T(11 + 4 * t) - T(10 + 4 * t) for t in range(nst)], -1), -1))
show("  stamp 11 -> next top", np.median(np.stack([T(8 + 4 * (t + 1)) - T(11 + 4 * t) for t in range(nst - 1)], -1), -1))
show("whole loop", T(5) - T(8))
show("loop end -> epilogue part 1 (row sums, writes)", T(6) - T(5))
show("exchange barrier -> every store issued", T(159) - T(6))
show("stores issued -> drained (vmcnt 0)", T(7) - T(159))
show("TOTAL", T(7) - T(0))
per = (T(7) - T(0))
print("ideal MFMA cycles per stage: 2048 (2 waves/SIMD x 16 MFMA x 64 cyc); stages in this layer: %d" % (c // 4))

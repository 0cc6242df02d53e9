#!/usr/bin/env python3
"""Dev tool (GPU box, DK_LIB = tools/build_stamp.sh conv_igemm DK_GSTAMP): phase timeline of the gather kernel on one
shape and configuration.  usage: gather_stamps.py batch c h w n size stride pad act config_name"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import darknet_amd as dk  # noqa: E402

b, c, h, w, n, size, stride, pad, act = map(int, sys.argv[1:10])
cfgname = sys.argv[10]
L = dk.lib()
L.cuda_set_device(0)
rng = np.random.default_rng(0)
d = dk.DkConvDesc(b, c, h, w, n, 1, size, stride, stride, 1, pad, act)
oh, ow = dk.conv_out_dims(h, w, size, stride, stride, pad)
dx = dk.DeviceArray(rng.uniform(-1, 1, b * c * h * w).astype(np.float32))
dw = dk.DeviceArray((rng.uniform(-1, 1, n * c * size * size) * 0.05).astype(np.float32))
db = dk.DeviceArray(rng.uniform(-1, 1, n).astype(np.float32))
dy = dk.DeviceArray(n=b * n * oh * ow)
ncfg = L.dk_conv_force_config(-1)
cfg = [i for i in range(ncfg) if L.dk_conv_config_name(i).decode() == cfgname][0]
L.dk_conv_force_config(cfg)
for _ in range(5):
    L.dk_conv_forward(C.byref(d), dx.ptr, dw.ptr, db.ptr, dy.ptr, None, None, None)
dk._sync()
buf = np.zeros(512 * 16 * 8, np.int64)
L.dk_gather_stamps_read.argtypes = [C.c_void_p, C.c_int]
L.dk_gather_stamps_read(buf.ctypes.data, buf.size)
s = buf.reshape(512, 16, 8).astype(np.float64)
nw = int((s[0, :, 0] > 0).sum())
s = s[:, :nw]
T = lambda i: s[:, :, i]
t0 = T(0).min()
print("%s on b%d c%d %dx%d n%d k%d s%d act %d: %d waves per workgroup; cycles (median | min | max over waves of the first 512 workgroups)" % (
    cfgname, b, c, h, w, n, size, stride, act, nw))
for name, a in (("launch skew: workgroup entry after the first one", T(0) - t0), ("index arithmetic", T(1) - T(0)),
                ("first tile: loads + LDS write + barrier", T(2) - T(1)), ("K loop", T(3) - T(2)), ("epilogue until every store is issued", T(4) - T(3)),
                ("stores drained", T(5) - T(4)), ("workgroup lifetime", T(5) - T(0)), ("kernel span: last exit - first entry", np.full((1, 1), T(5).max() - t0))):
    print("%-50s %9.0f | %9.0f | %9.0f" % (name, np.median(a), a.min(), a.max()))
K = c * size * size
print("MFMA cycles of a wave in the K loop if it owned its SIMD: %d per 32x32 sub-tile (K = %d)" % (K // 2 * 64, K))

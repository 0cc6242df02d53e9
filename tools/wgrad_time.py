#!/usr/bin/env python3
"""Dev tool: time ONE weight-gradient shape through a forced tile / kernel (dk_train_force knob 0) with the profile slots.
usage: wgrad_time.py batch c h w n size tile[,tile...] [det [stride]]   (tile 4 / 5 = the row-staged 3x3 kernel)"""
import ctypes as C
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import darknet_amd as dk

b, c, h, w, n, size = map(int, sys.argv[1:7])
tiles = [int(t) for t in sys.argv[7].split(",")]
det = int(sys.argv[8]) if len(sys.argv) > 8 else 0
stride = int(sys.argv[9]) if len(sys.argv) > 9 else 1
L = dk.lib()
L.cuda_set_device(0)
VP = C.c_void_p
L.dk_conv_backward_weights.argtypes = [VP, VP, VP, VP, VP]
L.dk_train_force.argtypes = [C.c_int, C.c_int]
L.dk_set_deterministic.argtypes = [C.c_int]
L.dk_conv_kernel_name.restype = C.c_char_p
rng = np.random.default_rng(0)
d = dk.DkConvDesc(b, c, h, w, n, 1, size, stride, stride, 1, size // 2, 4)
oh, ow = dk.conv_out_dims(h, w, size, stride, stride, size // 2)
dx = dk.DeviceArray(rng.uniform(-1, 1, b * c * h * w).astype(np.float32))
dd = dk.DeviceArray(rng.uniform(-1, 1, b * n * oh * ow).astype(np.float32))
dw = dk.DeviceArray(n=n * c * size * size)
L.dk_set_deterministic(det)
gf = 2.0 * n * c * size * size * b * oh * ow / 1e9
for tile in tiles:
    L.dk_train_force(0, tile)
    for _ in range(3):
        L.dk_conv_backward_weights(C.byref(d), dx.ptr, dd.ptr, dw.ptr, None)
    dk._sync()
    L.dk_profile_enable(1)
    for _ in range(10):
        L.dk_conv_backward_weights(C.byref(d), dx.ptr, dd.ptr, dw.ptr, None)
    out = (C.c_double * (3 * 512))()
    L.dk_profile_read(out, 512)
    L.dk_profile_enable(0)
    for i in range(512):
        if out[3 * i]:
            ms = out[3 * i + 2] / out[3 * i]
            print("b%d c%d %dx%d n%d k%d s%d tile %2d det %d env %s: %-36s %.3f ms  %.1f TFLOP/s" % (
                b, c, h, w, n, size, stride, tile, det, os.environ.get("DK_WGRAD3_BLOCKS", "-"), L.dk_conv_kernel_name(i).decode(), ms, gf / ms))

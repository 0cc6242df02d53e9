#!/usr/bin/env python3
"""Dev: where does conv_wgrad3_f32 differ from the gather kernel?  usage: dev_wgrad3_dbg.py b c h w n"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import darknet_amd as dk
b, c, h, w, n = map(int, sys.argv[1:6])
L = dk.lib()
L.cuda_set_device(0)
VP = C.c_void_p
L.dk_conv_backward_weights.argtypes = [VP, VP, VP, VP, VP]
L.dk_train_force.argtypes = [C.c_int, C.c_int]
rng = np.random.default_rng(0)
x = rng.uniform(-1, 1, (b, c, h, w)).astype(np.float32)
dl = rng.uniform(-1, 1, (b, n, h, w)).astype(np.float32)
d = dk.DkConvDesc(b, c, h, w, n, 1, 3, 1, 1, 1, 1, 4)
dx, dd = dk.DeviceArray(x), dk.DeviceArray(dl)
out = {}
for tile in (0, 4):
    L.dk_train_force(0, tile)
    dw = dk.DeviceArray(np.zeros(n * c * 9, np.float32))
    assert L.dk_conv_backward_weights(C.byref(d), dx.ptr, dd.ptr, dw.ptr, None) == 0
    out[tile] = dw.numpy().reshape(n, c, 9).copy()
L.dk_train_force(0, -1)
ref, got = out[0], out[4]
rms = np.sqrt((ref ** 2).mean())
bad = np.abs(ref - got) > 1e-3 * rms
print("rms %.3g, bad %d of %d" % (rms, bad.sum(), bad.size))
if bad.any():
    m, cc, t = np.nonzero(bad)
    print("bad filters:", np.unique(m)[:40], "count", len(np.unique(m)))
    print("bad channels:", np.unique(cc)[:40], "count", len(np.unique(cc)))
    print("bad taps:", np.unique(t))
    for i in range(min(8, len(m))):
        print(m[i], cc[i], t[i], ref[m[i], cc[i], t[i]], got[m[i], cc[i], t[i]])

#!/usr/bin/env python3
"""Fused stem (dk_conv_stem_forward) against the two launches it replaces, at a network's real stem shape.
usage: python tools/stem_bench.py [batch h w n1 reps]   (default 16 608 608 64 20: yolov4 C3)"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import darknet_amd as dk

MISH = 17


def main():
    a = [int(v) for v in sys.argv[1:]]
    batch, h, w, n1, reps = (a + [16, 608, 608, 64, 20][len(a):])[:5]
    L = dk.lib()
    VP = C.c_void_p
    L.dk_conv_stem_forward.argtypes = [VP, VP, VP, VP, VP, VP, VP, VP, C.c_int, VP]
    rng = np.random.default_rng(0)
    x = rng.uniform(0, 1, (batch, 3, h, w)).astype(np.float32)
    w0 = (rng.uniform(-1, 1, (32, 3, 3, 3)) * 0.27).astype(np.float32)
    w1 = (rng.uniform(-1, 1, (n1, 32, 3, 3)) * 0.083).astype(np.float32)
    b0 = rng.uniform(-.5, .5, 32).astype(np.float32)
    b1 = rng.uniform(-.5, .5, n1).astype(np.float32)
    d0 = dk.DkConvDesc(batch, 3, h, w, 32, 1, 3, 1, 1, 1, 1, MISH)
    d1 = dk.DkConvDesc(batch, 32, h, w, n1, 1, 3, 2, 2, 1, 1, MISH)
    dx, dw0, db0, dw1, db1 = (dk.DeviceArray(v) for v in (x, w0, b0, w1, b1))
    dmid = dk.DeviceArray(n=batch * 32 * h * w)
    dy2 = dk.DeviceArray(n=batch * n1 * (h // 2) * (w // 2))
    dy1 = dk.DeviceArray(n=batch * n1 * (h // 2) * (w // 2))
    hip = C.CDLL("libamdhip64.so")
    ev = [C.c_void_p() for _ in range(2)]
    for e in ev:
        hip.hipEventCreate(C.byref(e))
    hip.hipEventRecord.argtypes = [VP, VP]
    hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), VP, VP]
    hip.hipEventSynchronize.argtypes = [VP]

    L.get_cuda_stream.restype = VP
    st = L.get_cuda_stream()   # the library's stream: the launches above go there, so must the events

    def timed(fn):
        for _ in range(3):
            fn()
        hip.hipDeviceSynchronize()
        hip.hipEventRecord(ev[0], st)
        for _ in range(reps):
            fn()
        hip.hipEventRecord(ev[1], st)
        hip.hipEventSynchronize(ev[1])
        ms = C.c_float()
        hip.hipEventElapsedTime(C.byref(ms), ev[0], ev[1])
        return ms.value / reps

    def two():
        assert L.dk_conv_forward(C.byref(d0), dx.ptr, dw0.ptr, db0.ptr, dmid.ptr, None, None, None) == 0
        assert L.dk_conv_forward(C.byref(d1), dmid.ptr, dw1.ptr, db1.ptr, dy2.ptr, None, None, None) == 0

    def one():
        assert L.dk_conv_stem_forward(C.byref(d0), C.byref(d1), dx.ptr, dw0.ptr, db0.ptr, dw1.ptr, db1.ptr, dy1.ptr, 0, None) == 0

    t2, t1 = timed(two), timed(one)
    same = np.array_equal(dy1.numpy().view(np.uint32), dy2.numpy().view(np.uint32))
    fl = (2.0 * 32 * 27 * h * w + 2.0 * n1 * 288 * (h // 2) * (w // 2)) * batch
    print("stem b%d %dx%d -> %d: two launches %.3f ms, fused %.3f ms (%.1f TFLOP/s), bitwise equal: %s"
          % (batch, h, w, n1, t2, t1, fl / t1 / 1e9, same))


if __name__ == "__main__":
    main()

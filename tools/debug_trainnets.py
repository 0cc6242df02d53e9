#!/usr/bin/env python3
"""Dev tool (GPU box): step-by-step statistics of the C-level TrainNetworks path vs a single replica."""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import darknet_amd as dk  # noqa: E402
import synth  # noqa: E402
from darknet_amd import netapi  # noqa: E402

VP = C.c_void_p
L = dk.lib()
L.cuda_set_device(0)
for fn, at, rt in (("DkNetworkArrayCreate", [C.c_int], VP), ("DkNetworkArrayAt", [VP, C.c_int], VP),
                   ("LoadNetwork", [VP, C.c_char_p, C.c_char_p, C.c_bool, C.c_bool], C.c_bool),
                   ("DkTrainNetworksFlat", [VP, C.c_int, VP, C.c_int, VP, C.c_int, C.c_int, C.c_int], C.c_float),
                   ("DkSetMaxIter", [VP, C.c_int], None), ("DkLayerPull", [VP, C.c_int, C.c_int, VP, C.c_size_t], C.c_long),
                   ("TrainNetworkDatum", [VP, VP, VP], C.c_float), ("UpdateNetworkGpu", [VP], None),
                   ("DkAdvanceIteration", [VP], None), ("DkGradBucketSize", [VP], C.c_size_t)):
    getattr(L, fn).argtypes = at
    getattr(L, fn).restype = rt
d = tempfile.mkdtemp()
B = 2
txt = open(os.path.join(ROOT, "cfg", "yolov4-tiny.cfg")).read().replace("batch=64", "batch=%d" % B)
one = os.path.join(d, "one.cfg")
open(one, "w").write(txt.replace("batch=%d" % B, "batch=1"))
sub = os.path.join(d, "sub.cfg")
open(sub, "w").write(txt.replace("subdivisions=1", "subdivisions=%d" % B))
w = os.path.join(d, "w.weights")
netapi.synth_weights_for(dk, "yolov4-tiny", w)
x = synth.make_input(B, 3, 416, 416)
truth = np.zeros((B, 450), np.float32)
for b in range(B):
    for t, box in enumerate([(.3, .4, .2, .3, 1), (.6, .5, .4, .35, 17), (.8, .2, .1, .15, 60)]):
        truth[b, t * 5:(t + 1) * 5] = box


def pull(p, i, which, n):
    out = np.empty(n, np.float32)
    assert L.DkLayerPull(p, i, which, out.ctypes.data, n) == n
    return out


ref = netapi.DkNet(dk, sub, w, train=True)
L.DkSetMaxIter(ref.p, 1000)
nets = L.DkNetworkArrayCreate(B)
ps = [L.DkNetworkArrayAt(nets, i) for i in range(B)]
for p in ps:
    assert L.LoadNetwork(p, one.encode(), w.encode(), True, False)
    L.DkSetMaxIter(p, 1000)
f0 = ref.info(0)
for step in range(2):
    for i in range(B):
        L.TrainNetworkDatum(ref.p, x[i:i + 1].ctypes.data, truth[i:i + 1].ctypes.data)
    gref = pull(ref.p, 0, 7, f0["nweights"])
    L.DkAdvanceIteration(ref.p)
    L.UpdateNetworkGpu(ref.p)
    c = L.DkTrainNetworksFlat(nets, B, x.ctypes.data, x.shape[1], truth.ctypes.data, truth.shape[1], B, 4)
    wr = pull(ref.p, 0, 1, f0["nweights"])
    for k, p in enumerate(ps):
        wk = pull(p, 0, 1, f0["nweights"])
        gk = pull(p, 0, 7, f0["nweights"])
        print("step %d replica %d: cost %.4f  |w-wref|max %.3g  w rms %.3g  grad(after update) rms %.3g  ref grad(before update) rms %.3g  bucket %d"
              % (step, k, c, np.abs(wk - wr).max(), np.sqrt((wk ** 2).mean()), np.sqrt((gk.astype(np.float64) ** 2).mean()),
                 np.sqrt((gref.astype(np.float64) ** 2).mean()), L.DkGradBucketSize(p)))

#!/usr/bin/env python3
"""Dev tool (GPU box): worst per-layer parity ratio |o - ref| / (REL |ref| + ATOL_RMS rms) of a net against its golden
samples (tests/golden/net_<name>.npz), for the current kernel selection (DK_WINOGRAD / DK_WINO_MAXC ...)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import darknet_amd as dk
from darknet_amd import netapi
import synth, util

for name in sys.argv[1:] or ["yolov4"]:
    g = np.load(os.path.join(ROOT, "tests", "golden", "net_%s.npz" % name))
    w = "/tmp/_m_%s.weights" % name
    netapi.synth_weights_for(dk, name, w)
    B = int(os.environ.get("MARGIN_BATCH", "16"))   # the tuner only picks Winograd at real batch sizes
    L = dk.lib()
    L.DkSetFusion(0)
    net = netapi.DkNet(dk, netapi.cfg_path(name), w, batch=B)
    x = np.repeat(synth.make_input(1, net.c, net.h, net.w, seed=12345), B, 0)
    net.predict(x)
    worst = []
    nw = 0
    for i in range(net.n):
        o = net.output(i)[0].ravel()     # item 0 == the b = 1 golden run (batch-position invariance)
        ref = g["layer_samples"][i]
        idx = np.linspace(0, o.size - 1, 64).astype(np.int64)
        rms = np.sqrt(g["layer_sums"][i][1] / o.size)
        err = np.abs(o[idx] - ref) / (util.REL * np.abs(ref) + util.ATOL_RMS * rms)
        worst.append((float(err.max()), i, net.info(i)["type"]))
    worst.sort(reverse=True)
    print(name, "worst sample ratios (ratio, layer, type):", [(round(a, 3), b, c) for a, b, c in worst[:8]])
    net.close()

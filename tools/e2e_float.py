#!/usr/bin/env python3
"""Dev tool: the PCIe-inclusive float path (host frames in, whole yolo heads out) with the forward as a graph or as plain
launches, and the pure forward beside it.  usage: e2e_float.py [cfg batch steps]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np
import darknet_amd as dk
from darknet_amd import netapi
import synth

cfgname = sys.argv[1] if len(sys.argv) > 1 else "yolov4"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
L = dk.lib()
L.cuda_set_device(0)
tmp = tempfile.mkdtemp(prefix="dke2e_")
w = os.path.join(tmp, "w.weights")
netapi.synth_weights_for(dk, cfgname, w)
net = netapi.DkNet(dk, netapi.cfg_path(cfgname), w, batch=batch)
x = synth.make_input(batch, net.c, net.h, net.w)


def rate(fn):
    fn(); fn()
    dk._sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    dk._sync()
    return batch * steps / (time.perf_counter() - t0)


def staged():
    net.predict_staged()
    net.stage_float(x)
    net.collect()


for graph in (1, 0):
    L.DkSetGraph(graph)
    L.DkSetPullHeads(0)
    net.stage_float(x)
    fwd = rate(lambda: (net.predict_staged(), net.stage_float(x), net.collect()))
    L.DkSetPullHeads(1)
    net.stage_float(x)
    r = rate(staged)
    print("%s b%d graph %d: staged input only %.0f images/s, + heads to the host %.0f images/s" % (cfgname, batch, graph, fwd, r))

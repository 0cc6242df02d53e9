#!/usr/bin/env python3
"""Per-layer times of the three GEMMs of a training step (forward convolution, data gradient, weight gradient):
which layer runs which tile configuration and at what rate.  Diagnostics for the C4 workload.
usage: DK_TRAIN_LAYERS=1 train_layers.py [--cfg yolov4] [--batch 8]  -> table on stdout"""
import argparse, os, re, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
os.environ.setdefault("DK_TRAIN_LAYERS", "1")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="yolov4")
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    import numpy as np
    import darknet_amd as dk
    from darknet_amd import netapi
    import synth
    L, C = dk.lib(), dk.C
    L.cuda_set_device(0)
    tmp = tempfile.mkdtemp(prefix="dklayers_")
    w = os.path.join(tmp, "w.weights")
    netapi.synth_weights_for(dk, a.cfg, w)
    cfg = os.path.join(tmp, "t.cfg")
    txt = open(netapi.cfg_path(a.cfg)).read()
    txt = re.sub(r"batch=\d+", "batch=%d" % a.batch, txt, count=1)
    txt = re.sub(r"subdivisions=\d+", "subdivisions=1", txt, count=1)
    open(cfg, "w").write(txt)
    net = netapi.DkNet(dk, cfg, w, train=True)
    L.DkSetMaxIter.argtypes = [C.c_void_p, C.c_int]
    L.DkSetMaxIter(net.p, 100000)
    x = synth.make_input(a.batch, net.c, net.h, net.w)
    truth = np.zeros((a.batch, 90 * 5), np.float32)
    for b in range(a.batch):
        for t, box in enumerate([(.3, .4, .2, .3, 1), (.6, .5, .4, .35, 17), (.8, .2, .1, .15, 60)]):
            truth[b, t * 5:(t + 1) * 5] = box
    L.TrainNetworkDatum.restype = C.c_float
    L.TrainNetworkDatum.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.DkTrainLayerReport.restype = C.c_int
    L.DkTrainLayerReport.argtypes = [C.c_char_p, C.c_int]
    L.dk_conv_config_name.restype = C.c_char_p
    L.dk_conv_config_name.argtypes = [C.c_int]
    buf = C.create_string_buffer(1 << 20)
    for _ in range(3):   # first step: kernel timing; then warm
        L.TrainNetworkDatum(net.p, x.ctypes.data, truth.ctypes.data)
    L.DkTrainLayerReport(buf, len(buf))
    steps = 3
    for _ in range(steps):
        L.TrainNetworkDatum(net.p, x.ctypes.data, truth.ctypes.data)
    n = L.DkTrainLayerReport(buf, len(buf))
    rows = {}
    for line in buf.raw[:n].decode().splitlines():
        li, kind, c, gf, ms = line.split()
        k = (int(li), int(kind))
        r = rows.setdefault(k, [int(c), float(gf), 0.0])
        r[2] += float(ms) / steps
    kinds = ("fwd", "dgrad", "wgrad")
    wname = {-1: "heuristic", 0: "128x128", 1: "64x128", 2: "128x64", 3: "64x64", 4: "rows3x3", 5: "rows3x3_1wg"}
    tot = [0.0, 0.0, 0.0]
    print("layer  shape                                 kind   config                      GF      ms     TF")
    for (li, kind), (c, gf, ms) in sorted(rows.items()):
        i = net.info(li)
        shape = "c%-4d %3dx%-3d n%-4d k%d s%d" % (i["c"], i["h"], i["w"], i["n"], i["size"], i["stride"]) if "size" in i else ""
        name = wname.get(c, "?") if kind == 2 else ("heuristic" if c < 0 else L.dk_conv_config_name(c).decode())
        tot[kind] += ms
        print("L%-4d  %-36s  %-5s  %-26s %6.2f  %6.3f  %5.1f" % (li, shape, kinds[kind], name, gf, ms, gf / ms))
    print("totals per step: fwd %.2f ms, dgrad %.2f ms, wgrad %.2f ms" % tuple(tot))
    net.close()


if __name__ == "__main__":
    main()

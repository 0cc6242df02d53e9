#!/usr/bin/env python3
"""Dev tool (GPU box): time every compiled tile configuration of the implicit-GEMM
conv kernel on each distinct conv shape of a cfg and print the table.
usage: conv_sweep.py cfg/yolov4.cfg BATCH [iters]
DK_SWEEP_FILTER=k1|s2|k3s1 restricts the shapes; DK_SWEEP_TOGGLE=<exported int setter, e.g. dk_conv_set_lds_bias> measures
every configuration with the switch at 0 and at 1 in the same process (A/B on one device, guide rule 24) and prints
both best-per-layer totals."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import darknet_amd as dk  # noqa: E402
from darknet_amd import netapi  # noqa: E402


def main():
    cfg, batch = sys.argv[1], int(sys.argv[2])
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    L = dk.lib()
    L.cuda_set_device(0)
    # layer table from the product's own parser (no device work)
    p = L.DkNetworkCreate()
    L.ParseNetworkCfg.restype = C.c_bool
    L.ParseNetworkCfg.argtypes = [C.c_void_p, C.c_char_p, C.c_bool]
    assert L.ParseNetworkCfg(p, cfg.encode(), False)
    a = (C.c_int * 8)()
    L.DkNetworkInfo(p, a)
    shapes = {}
    outputs = []
    for i in range(a[0]):
        f = (C.c_int * 24)()
        L.DkLayerInfo(p, i, f)
        f = dict(zip(netapi.INFO, list(f)))
        if f["type"] != 0:   # CONVOLUTIONAL
            continue
        outputs.append(f["outputs"])
        key = (f["c"], f["h"], f["w"], f["n"], f["size"], f["stride_x"], f["pad"], f["activation"], f["groups"])
        flt = os.environ.get("DK_SWEEP_FILTER", "")   # e.g. "k1" (1x1 layers), "s2" (stride 2), "k3s1"
        tag = "k%ds%d" % (f["size"], f["stride_x"])
        if flt and flt not in tag:
            continue
        shapes.setdefault(key, []).append(i)
    L.DkNetworkDestroy(p)
    ncfg = L.dk_conv_force_config(-1)
    names = [L.dk_conv_config_name(i).decode() for i in range(ncfg)]
    rows = []
    tot_off = [0.0]
    tot_best = tot_heur = 0.0
    tot_gflop = 0.0
    maxin = max(batch * k[0] * k[1] * k[2] for k in shapes)
    maxout = max(batch * o for o in outputs)
    rng = np.random.default_rng(0)
    dx = dk.DeviceArray(rng.uniform(-1, 1, maxin).astype(np.float32))
    dy = dk.DeviceArray(n=maxout)
    for key, idxs in sorted(shapes.items(), key=lambda kv: kv[1][0]):
        c, h, w, n, size, stride, pad, act, groups = key
        d = dk.DkConvDesc(batch, c, h, w, n, groups, size, stride, stride, 1, pad, act)
        wt = dk.DeviceArray((rng.uniform(-1, 1, n * (c // groups) * size * size) * 0.05).astype(np.float32))
        bs = dk.DeviceArray(rng.uniform(-1, 1, n).astype(np.float32))
        L.dk_conv_force_config(-1)
        heur = L.dk_conv_pick_config(C.byref(d))
        du = None
        nu = L.dk_conv_wino_weights_size(C.byref(d))
        if nu:   # Winograd candidate: transformed filters registered under the weights pointer
            du = dk.DeviceArray(n=nu)
            L.dk_conv_wino_transform_weights(C.byref(d), wt.ptr, du.ptr, None)
            L.dk_conv_wino_register(wt.ptr, du.ptr)
        toggle = os.environ.get("DK_SWEEP_TOGGLE")
        times_off = []
        if toggle:
            getattr(L, toggle)(0)
            for cfgi in range(ncfg):
                L.dk_conv_force_config(cfgi)
                L.dk_conv_forward(C.byref(d), dx.ptr, wt.ptr, bs.ptr, dy.ptr, None, None, None)
                L.dk_profile_enable(1)
                for _ in range(iters):
                    L.dk_conv_forward(C.byref(d), dx.ptr, wt.ptr, bs.ptr, dy.ptr, None, None, None)
                out = (C.c_double * (3 * 256))()
                L.dk_profile_read(out, 256)
                L.dk_profile_enable(0)
                ms = sum(out[(cfgi * 4 + v) * 3 + 2] for v in range(4)) / iters
                times_off.append(ms if ms > 0 else float("inf"))
            getattr(L, toggle)(1)
            tot_off[0] += min(times_off) * len(idxs)
        times = []
        for cfgi in range(ncfg):
            L.dk_conv_force_config(cfgi)
            L.dk_conv_forward(C.byref(d), dx.ptr, wt.ptr, bs.ptr, dy.ptr, None, None, None)  # warm
            L.dk_profile_enable(1)
            for _ in range(iters):
                L.dk_conv_forward(C.byref(d), dx.ptr, wt.ptr, bs.ptr, dy.ptr, None, None, None)
            out = (C.c_double * (3 * 256))()
            L.dk_profile_read(out, 256)
            L.dk_profile_enable(0)
            ms = sum(out[(cfgi * 4 + v) * 3 + 2] for v in range(4)) / iters
            g1 = sum(out[(cfgi * 4 + v) * 3 + 1] for v in range(4)) / iters
            if g1 > 0:
                gf = g1
            # a direct-3x3 configuration that cannot take this layer falls back: not a candidate
            times.append(ms if ms > 0 else float("inf"))
        best = int(np.argmin(times))
        cnt = len(idxs)
        tot_best += times[best] * cnt
        tot_heur += times[heur] * cnt
        tot_gflop += gf * cnt
        rows.append(dict(layers=idxs, shape=key, gflop=gf, ms=times, best=best, heur=heur,
                         tf_best=gf / times[best], tf_heur=gf / times[heur]))
        print("L%-4d x%-2d c%-4d %3dx%-3d n%-4d k%d s%d  GF %7.2f | " % (idxs[0], cnt, c, h, w, n, size, stride, gf) +
              " ".join("%6.3f" % t for t in times) + " | best %d (%5.1f TF) heur %d (%5.1f TF)" %
              (best, gf / times[best], heur, gf / times[heur]), flush=True)
        if toggle:
            print("      %s: off best %.3f ms -> on best %.3f ms (%+.1f %%)" % (toggle, min(times_off), times[best], 100 * (times[best] / min(times_off) - 1)), flush=True)
        if du is not None:
            L.dk_conv_wino_register(wt.ptr, None)
            du.free()
        wt.free(); bs.free()
    L.dk_conv_force_config(-1)
    print("configs:", names)
    print("total conv GFLOP %.1f  best-per-layer %.3f ms (%.1f TF)  heuristic %.3f ms (%.1f TF)" %
          (tot_gflop, tot_best, tot_gflop / tot_best, tot_heur, tot_gflop / tot_heur))
    if os.environ.get("DK_SWEEP_TOGGLE"):
        print("%s: best-per-layer off %.3f ms -> on %.3f ms" % (os.environ["DK_SWEEP_TOGGLE"], tot_off[0], tot_best))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "conv_sweep_%s_b%d.json" % (os.path.basename(cfg), batch)), "w") as f:
        json.dump(dict(names=names, rows=rows), f)


if __name__ == "__main__":
    main()

"""CPU: the product's host-side YOLO loss (csrc/host/yolo_loss.cpp)
against golden deltas/costs dumped from the REAL reference (tests/golden/yololoss.npz).
The yolo layer's input is produced by the oracle's train-mode forward, which is
bit-identical to the reference's; the loss must then match BIT-EXACTLY."""
import ctypes as C
import os

import numpy as np
import pytest

import netutil
import synth
from oracle import orc_net as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.mark.parametrize("name", ["yolov4-tiny", "yolov4-csp", "yolov4"])
def test_yolo_loss_matches_reference(dk, name, tmp_path):
    g = np.load(os.path.join(GOLD, "yololoss.npz"))
    B = 2
    txt = open(os.path.join(ROOT, "cfg", name + ".cfg")).read()
    cfg = str(tmp_path / "l.cfg")
    open(cfg, "w").write(txt.replace("batch=64", "batch=%d" % B).replace("subdivisions=8", "subdivisions=1"))
    onet = O.parse_cfg(cfg)
    w = str(tmp_path / "w.weights")
    synth.write_weights(w, [(l.n, l.c // l.groups, l.size, l.batch_normalize) for l in onet.layers if l.type == O.CONVOLUTIONAL], seed=2024)
    x = synth.make_input(B, onet.c, onet.h, onet.w, seed=12345)
    onet = O.load_network_train(cfg, w, None)
    assert onet.batch == B
    O.forward_train(onet, x)
    L = dk.lib()
    L.ParseNetworkCfg.restype = C.c_bool
    L.ParseNetworkCfg.argtypes = [C.c_void_p, C.c_char_p, C.c_bool]
    L.DkLayerPtr.restype = C.c_void_p
    L.DkLayerPtr.argtypes = [C.c_void_p, C.c_int]
    L.DkYoloLossHost.restype = C.c_float
    L.DkYoloLossHost.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    p = L.DkNetworkCreate()
    if dk.have_gpu():
        L.cuda_set_device(0)
    assert L.ParseNetworkCfg(p, cfg.encode(), True)
    truth = np.ascontiguousarray(g[name + "_truth"])
    nyolo = 0
    for i, l in enumerate(onet.layers):
        if l.type != O.YOLO:
            continue
        nyolo += 1
        out = np.ascontiguousarray(l.output.ravel().copy())
        delta = np.full(out.size, 7.0, np.float32)
        cost = L.DkYoloLossHost(L.DkLayerPtr(p, i), onet.w, onet.h, out.ctypes.data, truth.ctypes.data, delta.ctypes.data)
        ref = np.zeros(out.size, np.float32)
        ref[g["%s_%d_idx" % (name, i)]] = g["%s_%d_val" % (name, i)]
        assert np.array_equal(delta, ref), "%s yolo %d: %d deltas differ (max %g)" % (
            name, i, np.count_nonzero(delta != ref), np.abs(delta - ref).max())
        assert np.float32(cost) == g["%s_%d_cost" % (name, i)], (cost, g["%s_%d_cost" % (name, i)])
    assert nyolo in (2, 3)
    L.DkNetworkDestroy(p)


def test_gaussian_yolo_loss_matches_reference(dk, tmp_path):
    """[Gaussian_yolo] training loss (csrc/host/yolo_loss.cpp: DkGaussianYoloLossHost) against the
    deltas / costs of the REAL reference (tests/golden/gaussianloss.npz: head 5 = mse boxes, head 12 = giou boxes
    + iou_thresh + max_delta + label smoothing + uc_normalizer).  BIT-EXACT, like the [yolo] loss."""
    g = np.load(os.path.join(GOLD, "gaussianloss.npz"))
    B = int(g["batch"])
    cfg = str(tmp_path / "g.cfg")
    open(cfg, "w").write(open(os.path.join(ROOT, "cfg", "gaussian-test.cfg")).read().replace("batch=1", "batch=%d" % B, 1))
    onet = O.parse_cfg(cfg)
    w = str(tmp_path / "w.weights")
    synth.write_weights_layers(w, synth.weight_layers_of(onet), seed=2024)
    x = synth.make_input(B, onet.c, onet.h, onet.w, seed=12345)
    onet = O.load_network_train(cfg, w, None)
    assert onet.batch == B
    O.forward_train(onet, x)
    L = dk.lib()
    L.ParseNetworkCfg.restype = C.c_bool
    L.ParseNetworkCfg.argtypes = [C.c_void_p, C.c_char_p, C.c_bool]
    L.DkLayerPtr.restype = C.c_void_p
    L.DkLayerPtr.argtypes = [C.c_void_p, C.c_int]
    L.DkGaussianYoloLossHost.restype = C.c_float
    L.DkGaussianYoloLossHost.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    p = L.DkNetworkCreate()
    if dk.have_gpu():
        L.cuda_set_device(0)
    assert L.ParseNetworkCfg(p, cfg.encode(), True)
    truth = np.ascontiguousarray(g["truth"])
    heads = 0
    for i, l in enumerate(onet.layers):
        if l.type != O.GAUSSIAN_YOLO:
            continue
        heads += 1
        out = np.ascontiguousarray(l.output.ravel().copy())
        delta = np.full(out.size, 7.0, np.float32)
        cost = L.DkGaussianYoloLossHost(L.DkLayerPtr(p, i), onet.w, onet.h, out.ctypes.data, truth.ctypes.data, delta.ctypes.data)
        ref = np.zeros(out.size, np.float32)
        ref[g["delta_%d_idx" % i]] = g["delta_%d_val" % i]
        bad = np.flatnonzero(delta != ref)
        assert bad.size == 0, "gaussian head %d: %d deltas differ (max %g), first at %d: %r vs %r" % (
            i, bad.size, np.abs(delta - ref).max(), bad[0], delta[bad[0]], ref[bad[0]])
        assert np.float32(cost) == g["cost_%d" % i], (cost, g["cost_%d" % i])
    assert heads == 2
    L.DkNetworkDestroy(p)

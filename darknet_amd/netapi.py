"""ctypes wrapper over the network-level C-ABI (include/yolo_core_hip.h) used by
tests, bench.py and __graft_entry__.smoke().  No compute happens here."""
import ctypes as C
import os

import numpy as np

import sys

ROOT_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT_, "tools"))
import synth  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INFO = ["type", "batch", "outputs", "out_c", "out_h", "out_w", "n", "size", "stride", "pad",
        "c", "h", "w", "activation", "batch_normalize", "nweights", "groups", "inputs",
        "classes", "total", "index", "dilation", "stride_x", "stride_y"]


def cfg_path(name):
    return os.path.join(ROOT, "cfg", name + ".cfg")


class DkNet:
    """Thin ctypes wrapper over the product's network API (yolo_core_hip.h)."""

    def __init__(self, dk, cfg, weights=None, batch=None, train=False):
        self.dk, self.L = dk, dk.lib()
        self.p = self.L.DkNetworkCreate()
        w = weights.encode() if weights else None
        if batch is None:
            ok = self.L.LoadNetwork(self.p, cfg.encode(), w, train, False)
        else:
            ok = self.L.LoadNetworkBatch(self.p, cfg.encode(), w, batch)
        assert ok, "LoadNetwork failed"
        a = (C.c_int * 8)()
        self.L.DkNetworkInfo(self.p, a)
        self.n, self.batch, self.w, self.h, self.c, self.inputs, self.outputs, self.gpu = list(a)

    def info(self, i):
        a = (C.c_int * 24)()
        self.L.DkLayerInfo(self.p, i, a)
        return dict(zip(INFO, list(a)))

    def convs(self):
        out = []
        for i in range(self.n):
            f = self.info(i)
            if f["type"] == 0:
                out.append((f["n"], f["c"] // f["groups"], f["size"], f["batch_normalize"]))
        return out

    def predict(self, x):
        x = np.ascontiguousarray(x, np.float32)
        assert x.size == self.batch * self.inputs
        self.L.NetworkPredict(self.p, x.ctypes.data)

    def predict_u8(self, frames, row_step=None):
        """frames: uint8 [batch, h, row_step] (interleaved HWC rows, row_step >= w*c bytes)."""
        frames = np.ascontiguousarray(frames, np.uint8)
        if row_step is None:
            row_step = self.w * self.c
        assert frames.size == self.batch * self.h * row_step
        self.L.DkNetworkPredictU8.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        self.L.DkNetworkPredictU8.restype = None
        self.L.DkNetworkPredictU8(self.p, frames.ctypes.data, row_step)

    def stage_u8(self, frames, row_step=None):
        """DkNetworkStageU8: pinned copy + H2D of the NEXT batch on the copy stream (overlaps the running forward)."""
        frames = np.ascontiguousarray(frames, np.uint8)
        if row_step is None:
            row_step = self.w * self.c
        assert frames.size == self.batch * self.h * row_step
        self.L.DkNetworkStageU8.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        self.L.DkNetworkStageU8.restype = None
        self.L.DkNetworkStageU8(self.p, frames.ctypes.data, row_step)

    def stage_float(self, x):
        """DkNetworkStageFloat: float CHW frames of the NEXT batch: pinned copy + H2D on the staging stream (overlaps the
        running forward); consumed by predict_staged()."""
        x = np.ascontiguousarray(x, np.float32)
        assert x.size == self.batch * self.inputs
        self.L.DkNetworkStageFloat.argtypes = [C.c_void_p, C.c_void_p]
        self.L.DkNetworkStageFloat.restype = None
        self.L.DkNetworkStageFloat(self.p, x.ctypes.data)

    def collect(self):
        """Wait for the forward in flight and the D2H of its heads (NetworkSync)."""
        self.L.NetworkSync.argtypes = [C.c_void_p]
        self.L.NetworkSync(self.p)

    def predict_staged(self):
        self.L.DkNetworkPredictStaged.argtypes = [C.c_void_p]
        self.L.DkNetworkPredictStaged.restype = None
        self.L.DkNetworkPredictStaged(self.p)

    def output(self, i):
        f = self.info(i)
        out = np.empty(f["batch"] * f["outputs"], np.float32)
        assert self.L.DkLayerOutput(self.p, i, out.ctypes.data, out.size) == 0
        return out.reshape(f["batch"], f["outputs"])

    def boxes(self, b, thresh, max_dets=400000):
        # the record buffers are kept between calls (a caller in a frame loop must not pay for 0.7 MB of
        # page-zeroing per image); DkGetBoxesBatch writes every field of the n records it returns
        if getattr(self, "_box_cap", 0) != max_dets:
            self._classes = self.info(self.n - 1)["classes"]
            self._box_d = np.zeros((max_dets, 5 + self._classes), np.float32)
            self._box_ids = np.zeros((max_dets, 4), np.int32)
            self._box_cap = max_dets
        d, ids = self._box_d, self._box_ids
        n = self.L.DkGetBoxesBatch(self.p, b, C.c_float(thresh), d.ctypes.data,
                                   ids.ctypes.data_as(C.POINTER(C.c_int)), max_dets)
        assert 0 <= n <= max_dets
        return d[:n].copy(), ids[:n].copy()

    def close(self):
        if self.p:
            self.L.DkNetworkDestroy(self.p)
            self.p = None


def synth_weights_for(dk, name, path, seed=2024):
    """Write the synthetic .weights for cfg `name` using the PRODUCT's own parser
    for the layer table (un-fused parse)."""
    L = dk.lib()
    p = L.DkNetworkCreate()
    L.ParseNetworkCfg.restype = C.c_bool
    L.ParseNetworkCfg.argtypes = [C.c_void_p, C.c_char_p, C.c_bool]
    assert L.ParseNetworkCfg(p, cfg_path(name).encode(), False)
    a = (C.c_int * 8)()
    L.DkNetworkInfo(p, a)
    convs = []
    for i in range(a[0]):
        f = (C.c_int * 24)()
        L.DkLayerInfo(p, i, f)
        if f[0] == 0:
            convs.append((f[6], f[10] // f[16], f[7], f[14]))
    size = L.DkWeightsFileSize(p)
    L.DkNetworkDestroy(p)
    synth.write_weights(path, convs, seed=seed)
    assert os.path.getsize(path) == size
    return convs

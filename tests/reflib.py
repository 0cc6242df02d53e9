"""ctypes access to the REAL reference CPU build (oracle/_ref/*.so), used only to
pin the oracle and to generate golden fixtures.  Absent on machines that never
had /root/reference (tests that need it skip)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")


def available(kind="canon"):
    return os.path.exists(os.path.join(REF_DIR, f"libref_{kind}.so"))


_libs = {}


def lib(kind="canon"):
    if kind not in _libs:
        L = C.CDLL(os.path.join(REF_DIR, f"libref_{kind}.so"))
        L.ref_net_load.restype = C.c_void_p
        L.ref_net_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.ref_net_predict.restype = C.POINTER(C.c_float)
        L.ref_net_predict.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.ref_net_n.argtypes = [C.c_void_p]
        L.ref_net_free.argtypes = [C.c_void_p]
        L.ref_layer_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.ref_layer_ptr.restype = C.POINTER(C.c_float)
        L.ref_layer_ptr.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ref_layer_indexes.restype = C.POINTER(C.c_int)
        L.ref_layer_indexes.argtypes = [C.c_void_p, C.c_int]
        L.ref_layer_bflops.restype = C.c_float
        L.ref_layer_bflops.argtypes = [C.c_void_p, C.c_int]
        L.ref_layer_cost.restype = C.c_float
        L.ref_layer_cost.argtypes = [C.c_void_p, C.c_int]
        L.ref_get_boxes.restype = C.c_int
        L.ref_get_boxes.argtypes = [C.c_void_p, C.c_float, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int)]
        L.ref_nms_sort.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.c_float, C.c_int, C.c_float]
        L.ref_train_datum.restype = C.c_float
        L.ref_train_datum.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.ref_forward_train.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.ref_update.argtypes = [C.c_void_p]
        L.ref_curr_lr.restype = C.c_float
        L.ref_curr_lr.argtypes = [C.c_void_p]
        L.ref_set_max_iter.argtypes = [C.c_void_p, C.c_int]
        L.ref_net_dims.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.ref_save_weights.argtypes = [C.c_void_p, C.c_char_p]
        L.init_cpu()
        _libs[kind] = L
    return _libs[kind]


INFO = ["type", "batch", "outputs", "out_c", "out_h", "out_w", "n", "size", "stride", "pad",
        "c", "h", "w", "activation", "batch_normalize", "nweights", "groups", "inputs",
        "classes", "total", "index", "dilation", "stride_x", "stride_y"]


class RefNet:
    def __init__(self, cfg, weights=None, train=False, kind="canon"):
        self.L = lib(kind)
        self.p = self.L.ref_net_load(cfg.encode(), (weights or "").encode(), int(train))
        assert self.p
        self.n = self.L.ref_net_n(self.p)
        d = (C.c_int * 8)()
        self.L.ref_net_dims(self.p, d)
        self.w, self.h, self.c, self.batch, self.subdiv = d[0], d[1], d[2], d[3], d[4]

    def info(self, i):
        a = (C.c_int * 24)()
        self.L.ref_layer_info(self.p, i, a)
        return dict(zip(INFO, list(a)))

    def predict(self, x):
        x = np.ascontiguousarray(x, np.float32)
        self.L.ref_net_predict(self.p, x.ctypes.data_as(C.POINTER(C.c_float)))

    def arr(self, i, which, n):
        p = self.L.ref_layer_ptr(self.p, i, which)
        if not p:
            return None
        return np.ctypeslib.as_array(p, shape=(n,)).copy()

    def output(self, i):
        inf = self.info(i)
        return self.arr(i, 0, inf["batch"] * inf["outputs"])

    def boxes(self, thresh, max_dets=200000):
        cls = C.c_int()
        nclass = self.info(self.n - 1)["classes"]
        buf = np.zeros((max_dets, 5 + nclass), np.float32)
        n = self.L.ref_get_boxes(self.p, thresh, buf.ctypes.data_as(C.POINTER(C.c_float)), max_dets, C.byref(cls))
        assert n <= max_dets
        return buf[:n].copy()

    def close(self):
        if self.p:
            self.L.ref_net_free(self.p)
            self.p = None

"""darknet_amd -- Python (ctypes) view of the MI355X-native Darknet conv path.

The product is the C-ABI shared library ``libdarknet_amd.so`` (hand-written
gfx950 HIP kernels + the C/C++ host side mirroring the reference's
yolo_core.h / dark_cuda.h API).  This module only binds it for tests and
bench.py; it contains no compute and no CPU fallback: if the library is
missing, or a compute entry point is called without a HIP device, it fails
loudly.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DK_LIB", os.path.join(HERE, "libdarknet_amd.so"))  # DK_LIB: dev A/B builds

# ACTIVATION ids (reference src/yolo_core.h:69-92)
LOGISTIC, RELU, LINEAR, LEAKY, MISH = 0, 1, 4, 8, 17
# LAYER_TYPE ids (reference src/yolo_core.h:112-138)
CONVOLUTIONAL, MAXPOOL, ROUTE, SHORTCUT, YOLO, UPSAMPLE = 0, 2, 7, 11, 17, 21


class DkConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("batch", "c", "h", "w", "n", "groups", "size", "stride_x",
                 "stride_y", "dilation", "pad", "activation")]


def build(verbose=False):
    """Compile every HIP source for gfx950 into libdarknet_amd.so (in-tree)."""
    cmd = ["make", "-C", os.path.join(HERE, "csrc"), "-j8"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)


_lib = None

FP = C.POINTER(C.c_float)
IP = C.POINTER(C.c_int)


def lib():
    """Load the C-ABI library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libdarknet_amd.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the HIP path)")
    L = C.CDLL(LIB_PATH)
    vp, sz, i, f = C.c_void_p, C.c_size_t, C.c_int, C.c_float
    sigs = {
        # dark_hip.h
        "cuda_set_device": (None, [i]),
        "cuda_get_device": (i, []),
        "CudaGetDeviceCount": (i, []),
        "get_gpu_compute_capability": (i, [i]),
        "show_cuda_cudnn_info": (None, []),
        "get_cuda_stream": (vp, []),
        "get_cuda_memcpy_stream": (vp, []),
        "cuda_make_array": (vp, [vp, sz]),
        "cuda_make_int_array": (vp, [sz]),
        "cuda_make_int_array_new_api": (vp, [vp, sz]),
        "cuda_make_array_pointers": (vp, [vp, sz]),
        "cuda_make_array_pinned": (vp, [vp, sz]),
        "cuda_make_array_pinned_preallocated": (vp, [vp, sz]),
        "pre_allocate_pinned_memory": (None, [sz]),
        "free_pinned_memory": (None, []),
        "cuda_free": (None, [vp]),
        "cuda_free_host": (None, [vp]),
        "cuda_push_array": (None, [vp, vp, sz]),
        "cuda_pull_array": (None, [vp, vp, sz]),
        "cuda_pull_array_async": (None, [vp, vp, sz]),
        "cuda_compare": (f, [vp, vp, sz, C.c_char_p]),
        "get_number_of_blocks": (i, [i, i]),
        "check_error": (None, [i]),
        # dk_kernels.h
        "dk_conv_forward": (i, [C.POINTER(DkConvDesc), vp, vp, vp, vp, vp, vp, vp]),
        "dk_conv_force_config": (i, [i]),
        "dk_conv_pick_config": (i, [C.POINTER(DkConvDesc)]),
        "dk_conv_config_name": (C.c_char_p, [i]),
        "dk_maxpool_forward": (i, [vp, vp, vp, i, i, i, i, i, i, i, i, vp]),
        "dk_route_copy": (i, [vp, i, i, i, i, vp, i, i, vp]),
        "dk_shortcut_forward": (i, [vp, vp, vp, sz, i, vp]),
        "dk_upsample_forward": (i, [vp, i, i, i, i, i, f, vp, vp]),
        "dk_yolo_forward": (i, [vp, vp, i, i, i, i, i, f, vp]),
        "dk_activate_array": (i, [vp, sz, i, vp]),
        "dk_activate_array_mish": (i, [vp, sz, vp, vp, vp]),
        "dk_add_bias": (i, [vp, vp, i, i, i, vp]),
        "dk_scale_bias": (i, [vp, vp, i, i, i, vp]),
        "dk_fill": (i, [sz, f, vp, vp]),
        "dk_copy": (i, [sz, vp, vp, vp]),
        "dk_axpy": (i, [sz, f, vp, vp, vp]),
        "dk_scal": (i, [sz, f, vp, vp]),
        "dk_profile_enable": (None, [i]),
        "dk_profile_read": (i, [C.POINTER(C.c_double), i]),
        "dk_conv_wino_weights_size": (sz, [C.POINTER(DkConvDesc)]),
        "dk_conv_wino_transform_weights": (i, [C.POINTER(DkConvDesc), vp, vp, vp]),
        "dk_conv_wino_register": (None, [vp, vp]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    L._sigs = sigs
    _lib = L
    _bind_net_api(L)
    return L


def _bind_net_api(L):
    """Network-level API (include/yolo_core_hip.h); bound when present."""
    vp, sz, i, f = C.c_void_p, C.c_size_t, C.c_int, C.c_float
    sigs = {
        "DkNetworkCreate": (vp, []),
        "DkNetworkDestroy": (None, [vp]),
        "LoadNetwork": (C.c_bool, [vp, C.c_char_p, C.c_char_p, C.c_bool, C.c_bool]),
        "LoadNetworkBatch": (C.c_bool, [vp, C.c_char_p, C.c_char_p, i]),
        "FreeNetwork": (None, [vp]),
        "NetworkPredict": (FP, [vp, vp]),
        "NetworkPredictDevice": (None, [vp, vp]),
        "NetworkSync": (None, [vp]),
        "DkNetworkInputGpu": (vp, [vp]),
        "DkNetworkInfo": (None, [vp, IP]),
        "DkLayerInfo": (None, [vp, i, IP]),
        "DkLayerBflops": (f, [vp, i]),
        "DkLayerOutput": (i, [vp, i, vp, sz]),
        "DkLayerOutputGpu": (vp, [vp, i]),
        "DkGetBoxesBatch": (i, [vp, i, f, vp, IP, i]),
        "DkWeightsFileSize": (sz, [vp]),
        "SaveWeights": (None, [vp, C.c_char_p]),
        "DkSetFusion": (None, [i]),
        "DkSetGraph": (None, [i]),
    }
    L._net_sigs = {}
    for name, (res, args) in sigs.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            continue
        fn.restype = res
        fn.argtypes = args
        L._net_sigs[name] = (res, args)


def have_gpu():
    return lib().CudaGetDeviceCount() > 0


class DeviceArray:
    """float32 (or int32) device buffer owned through the dark_hip C-ABI."""

    def __init__(self, host=None, n=None, dtype=np.float32):
        L = lib()
        self.dtype = np.dtype(dtype)
        if host is not None:
            host = np.ascontiguousarray(host, self.dtype)
            self.n = host.size
            if self.dtype == np.float32:
                self.ptr = L.cuda_make_array(host.ctypes.data, self.n)
            else:
                self.ptr = L.cuda_make_int_array_new_api(host.ctypes.data, self.n)
            L.cuda_pull_array  # keep symbol referenced
            _sync()
        else:
            self.n = int(n)
            if self.dtype == np.float32:
                self.ptr = L.cuda_make_array(None, self.n)
            else:
                self.ptr = L.cuda_make_int_array(self.n)

    def numpy(self):
        out = np.empty(self.n, self.dtype)
        # cuda_pull_array copies n 4-byte elements and synchronises the stream
        lib().cuda_pull_array(self.ptr, out.ctypes.data, self.n)
        return out

    def free(self):
        if self.ptr:
            lib().cuda_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _sync():
    """Synchronise the per-device compute stream (via a 1-element pull)."""
    L = lib()
    global _sync_buf
    try:
        _sync_buf
    except NameError:
        _sync_buf = L.cuda_make_array(None, 1)
    tmp = np.empty(1, np.float32)
    L.cuda_pull_array(_sync_buf, tmp.ctypes.data, 1)


def conv_out_dims(h, w, size, stride_x, stride_y, pad, dilation=1):
    keff = dilation * (size - 1) + 1
    p = pad * dilation
    return (h + 2 * p - keff) // stride_y + 1, (w + 2 * p - keff) // stride_x + 1


def conv_forward(x, weights, biases, batch, c, h, w, n, size, stride, pad,
                 activation, groups=1, dilation=1, residual=None,
                 want_act_in=False, stride_y=None, wino=False):
    """Host-array convenience wrapper over dk_conv_forward (tests).  wino=True also transforms and
    registers the Winograd copy of the filters (the caller forces the Winograd configuration)."""
    L = lib()
    sy = stride if stride_y is None else stride_y
    oh, ow = conv_out_dims(h, w, size, stride, sy, pad, dilation)
    d = DkConvDesc(batch, c, h, w, n, groups, size, stride, sy, dilation, pad, activation)
    dx, dw = DeviceArray(x), DeviceArray(weights)
    db = DeviceArray(biases) if biases is not None else None
    dr = DeviceArray(residual) if residual is not None else None
    dy = DeviceArray(n=batch * n * oh * ow)
    da = DeviceArray(n=batch * n * oh * ow) if want_act_in else None
    du = None
    if wino:
        nu = L.dk_conv_wino_weights_size(C.byref(d))
        if not nu:
            raise RuntimeError("layer does not take the Winograd kernel")
        du = DeviceArray(n=nu)
        if L.dk_conv_wino_transform_weights(C.byref(d), dw.ptr, du.ptr, None):
            raise RuntimeError("dk_conv_wino_transform_weights failed")
        L.dk_conv_wino_register(dw.ptr, du.ptr)
    rc = L.dk_conv_forward(C.byref(d), dx.ptr, dw.ptr, db.ptr if db else None,
                           dy.ptr, dr.ptr if dr else None, da.ptr if da else None, None)
    if du is not None:
        _sync()
        L.dk_conv_wino_register(dw.ptr, None)
    if rc != 0:
        raise RuntimeError("dk_conv_forward failed")
    y = dy.numpy().reshape(batch, n, oh, ow)
    if want_act_in:
        return y, da.numpy().reshape(batch, n, oh, ow)
    return y

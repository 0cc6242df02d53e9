"""Training path on the MI355X vs the CPU oracle (which is pinned bit-exact against
the real reference's train step, tests/test_oracle_golden.py::test_train_step_golden).
fp32 within util.REL / util.ATOL_RMS; maxpool indexes and SGD arithmetic bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest

import netutil
import synth
import util
from oracle import orc_net as O
from test_gpu_ops import orc_conv
from test_oracle_golden import inject_yolo_deltas, train_fixture

pytestmark = pytest.mark.gpu
F = C.c_float
VP = C.c_void_p


def bind(L):
    i, sz, f = C.c_int, C.c_size_t, C.c_float
    sig = {
        "dk_bn_forward_train": [VP] * 11 + [i, i, i, i, i, VP],
        "dk_gradient_array": [VP, VP, VP, sz, i, VP],
        "dk_backward_bias": [VP, VP, i, i, i, VP],
        "dk_bn_backward": [VP] * 10 + [i, i, i, VP],
        "dk_bn_act_backward": [VP] * 10 + [i, i, i, i, VP],
        "dk_conv_backward_weights": [VP, VP, VP, VP, VP],
        "dk_conv_backward_data": [VP, VP, VP, VP, VP],
        "dk_transpose_weights": [VP, VP, i, i, i, VP],
        "dk_transpose_weights_tapmajor": [VP, VP, i, i, i, VP],
        "dk_conv_backward_data_tapmajor": [VP, VP, VP, VP, VP],
        "dk_maxpool_backward": [VP, VP, sz, VP, VP],
        "dk_route_backward": [VP, i, i, i, i, i, i, VP, VP],
        "dk_shortcut_backward": [VP, sz, VP, VP, VP],
        "dk_upsample_backward": [VP, i, i, i, i, i, f, VP, VP],
        "dk_sgd_update": [VP, VP, sz, i, f, f, f, i, VP],
    }
    for k, a in sig.items():
        getattr(L, k).argtypes = a
        getattr(L, k).restype = i
    return L


BWD_CASES = [
    # batch, c, h, w, n, size, stride, pad, groups
    (2, 16, 13, 13, 32, 3, 1, 1, 1),
    (2, 8, 14, 14, 16, 3, 2, 1, 1),     # stride 2, even input
    (3, 8, 19, 19, 24, 3, 2, 1, 1),     # stride 2, odd input (parity classes)
    (2, 32, 13, 13, 255, 1, 1, 0, 1),   # 1x1 head, ragged M
    (1, 3, 32, 32, 32, 3, 1, 1, 1),     # first-layer shape class (K = 27)
    (2, 16, 10, 10, 16, 3, 1, 1, 2),    # groups
    (2, 64, 19, 19, 128, 3, 1, 1, 1),
    (2, 3, 416, 416, 32, 3, 2, 1, 1),   # yolov4-tiny layer 0: huge N, K = 27
    (1, 32, 208, 208, 64, 3, 2, 1, 1),  # yolov4-tiny layer 1
]


@pytest.mark.parametrize("case", BWD_CASES)
def test_conv_backward_vs_oracle(gpu, case):
    batch, c, h, w, n, size, stride, pad, groups = case
    L, G = O.lib(), bind(gpu.lib())
    rng = np.random.default_rng(util.seed_of(case))
    oh, ow = (h + 2 * pad - size) // stride + 1, (w + 2 * pad - size) // stride + 1
    x = rng.uniform(-1, 1, (batch, c, h, w)).astype(np.float32)
    wt = (rng.uniform(-1, 1, (n, c // groups, size, size)) * 0.2).astype(np.float32)
    delta = rng.uniform(-1, 1, (batch, n, oh, ow)).astype(np.float32)
    dw0 = rng.uniform(-1, 1, wt.shape).astype(np.float32)  # beta = 1: accumulates
    ref_dw = dw0.copy()
    ref_prev = np.full_like(x, 3.0)                          # dgrad overwrites
    ws = np.zeros(oh * ow * size * size * (c // groups) + 1, np.float32)
    L.orc_conv_backward(O.fptr(x), O.fptr(wt), O.fptr(delta), O.fptr(ref_dw), O.fptr(ref_prev), O.fptr(ws),
                        batch, c, h, w, n, groups, size, stride, stride, 1, pad)
    d = gpu.DkConvDesc(batch, c, h, w, n, groups, size, stride, stride, 1, pad, O.LINEAR)
    dx, dwt, dd = gpu.DeviceArray(x), gpu.DeviceArray(wt), gpu.DeviceArray(delta)
    ddw, dprev = gpu.DeviceArray(dw0), gpu.DeviceArray(np.full_like(x, 3.0))
    dt = gpu.DeviceArray(n=wt.size)
    assert G.dk_conv_backward_weights(C.byref(d), dx.ptr, dd.ptr, ddw.ptr, None) == 0
    per = wt.size // groups
    for g in range(groups):
        assert G.dk_transpose_weights(dwt.ptr + 4 * g * per, dt.ptr + 4 * g * per, n // groups, c // groups, size, None) == 0
    assert G.dk_conv_backward_data(C.byref(d), dd.ptr, dt.ptr, dprev.ptr, None) == 0
    util.assert_close(ddw.numpy().reshape(wt.shape), ref_dw, "wgrad %s" % (case,))
    util.assert_close(dprev.numpy().reshape(x.shape), ref_prev, "dgrad %s" % (case,))


PARITY_CASES = [
    # batch, c, h, w, n, size (stride 2, pad size // 2): the parity-class data gradient of the downsampling layers
    (8, 32, 608, 608, 64, 3),    # yolov4 layer 1 at the C4 per-GPU batch
    (2, 8, 14, 14, 32, 3),
    (3, 16, 38, 38, 64, 3),      # several pixel tiles per class, class tails are padding
    (1, 32, 76, 52, 96, 3),      # non-square, filters = 3 K tiles of 32
    (2, 32, 304, 304, 64, 3),    # yolov4 layer 1's shape class at half size
    (2, 24, 20, 20, 32, 2),      # 2x2 / stride 2 (one tap per class)
]


TRAIN_VARIANT_SHAPES = [
    # batch, c, h, w, n, size (stride 1, pad size // 2): yolov4's C4 layer shapes at the per-GPU batch 8, plus two small
    # shapes whose channel counts select the other tile / tap-major applicability classes
    (8, 512, 19, 19, 1024, 3),
    (8, 256, 38, 38, 512, 3),
    (8, 128, 76, 76, 128, 3),
    (8, 64, 304, 304, 64, 1),
    (2, 64, 19, 19, 128, 3),      # 64 channels per tap: only the 64-tap tiles are tap-major
    (2, 256, 38, 38, 128, 1),
    (3, 24, 13, 17, 40, 3),       # ragged everything: no tap-major tile, no 16-byte delta loads
    (2, 32, 23, 45, 128, 3),      # odd width above one segment: the row-staged 3x3 kernel's 4-byte pieces, two segments
    (1, 64, 10, 86, 256, 3),      # width = 2 mod 4, three segments (the last 6 pixels wide), 8-byte pieces
]


def _ran_kernels(L, nslots=512):
    out = (C.c_double * (3 * nslots))()
    L.dk_profile_read(out, nslots)
    L.dk_conv_kernel_name.restype = C.c_char_p
    names = {}
    for i in range(nslots):
        if out[3 * i]:
            nm = L.dk_conv_kernel_name(i)
            names[nm.decode() if nm else "slot%d" % i] = int(out[3 * i])
    return names


@pytest.mark.parametrize("case", TRAIN_VARIANT_SHAPES)
def test_every_training_kernel_variant_vs_oracle(gpu, case):
    """The training step picks its kernels per layer by first-step timing, so a variant that loses on the test box can
    win elsewhere: EVERY candidate is forced here (dk_train_force, dk_conv_force_config, dk_set_deterministic) against
    the oracle's BackwardConvolutionalLayer (src/convolutional_layer.cpp:1307-1380):
      * weight gradient: 4 tile shapes x tap-major on / off x 16-byte delta loads on / off x atomics / ordered
        (deterministic) reduction, accumulating into a non-zero dW as the reference does;
      * data gradient of the stride-1 layers as a forward convolution of delta with the transposed (1x1) or transposed
        and 180-degree-rotated (3x3) filters, through every forward configuration that can run the layer -- gather
        tiles, patch-in-LDS, LDS-DMA 1x1, Winograd schedules -- and the gather kernel in data-gradient mode;
    each launch is checked to have run the kernel it names (profile slots)."""
    batch, c, h, w, n, size = case
    pad = size // 2
    L, G = O.lib(), bind(gpu.lib())
    G.dk_train_force.argtypes = [C.c_int, C.c_int]
    G.dk_set_deterministic.argtypes = [C.c_int]
    G.dk_set_deterministic.restype = None
    G.dk_transpose_weights_flip.argtypes = [VP, VP, C.c_int, C.c_int, C.c_int, VP]
    G.dk_conv_config_can_run.argtypes = [VP, C.c_int]
    rng = np.random.default_rng(util.seed_of(case))
    x = rng.uniform(-1, 1, (batch, c, h, w)).astype(np.float32)
    wt = (rng.uniform(-1, 1, (n, c, size, size)) * 0.2).astype(np.float32)
    delta = rng.uniform(-1, 1, (batch, n, h, w)).astype(np.float32)
    dw0 = rng.uniform(-1, 1, wt.shape).astype(np.float32)
    ref_dw, ref_prev = dw0.copy(), np.full_like(x, 3.0)
    ws = np.zeros(h * w * size * size * c + 1, np.float32)
    L.orc_conv_backward(O.fptr(x), O.fptr(wt), O.fptr(delta), O.fptr(ref_dw), O.fptr(ref_prev), O.fptr(ws),
                        batch, c, h, w, n, 1, size, 1, 1, 1, pad)
    d = gpu.DkConvDesc(batch, c, h, w, n, 1, size, 1, 1, 1, pad, O.LINEAR)
    dx, dwt, dd = gpu.DeviceArray(x), gpu.DeviceArray(wt), gpu.DeviceArray(delta)
    K = c * size * size
    # a weight gradient is a sum of N = batch * oh * ow products per element; the oracle adds them sequentially in fp32
    # (gemm_nt, as the reference does), which alone carries ~sqrt(N) * 2^-24 relative noise: 5e-5 x rms at N = 739 k
    # (the 304x304 layers at b = 8; measured HIP-vs-oracle there: 3.2e-5).  Bound: util's 1e-5 x rms up to N = 10 k,
    # growing with sqrt(N) beyond.
    wg_atol = max(util.ATOL_RMS, 1e-7 * float(np.sqrt(batch * h * w)))
    # ---- weight gradient ------------------------------------------------------------------------------------------
    seen = set()
    G.dk_profile_enable(1)
    try:
        for tile in range(4):
            tk = 128 if tile in (0, 1) else 64
            for tmaj in (0, 1):
                for avec in (0, -1):
                    for det in (0, 1):
                        G.dk_train_force(0, tile)
                        G.dk_train_force(1, tmaj)
                        G.dk_train_force(2, avec)
                        G.dk_set_deterministic(det)
                        ddw = gpu.DeviceArray(dw0)
                        assert G.dk_conv_backward_weights(C.byref(d), dx.ptr, dd.ptr, ddw.ptr, None) == 0
                        ran = [k for k in _ran_kernels(gpu.lib()) if k.startswith("conv_wgrad")]
                        assert len(ran) == 1, ran
                        want_tmaj = bool(tmaj) and c % tk == 0
                        want_avec = avec != 0 and (h * w) % 4 == 0
                        want = "conv_wgrad_f32<%d, %d, %s, %s>" % (2 if tile in (0, 2) else 1, 2 if tile in (0, 1) else 1,
                                                                  "true" if want_avec else "false", "true" if want_tmaj else "false")
                        assert ran[0] == want, "forced %s, ran %s" % (want, ran[0])
                        seen.add((ran[0], det))
                        util.assert_close(ddw.numpy().reshape(wt.shape), ref_dw,
                                          "wgrad %s tile %d tap-major %d avec %d deterministic %d" % (case, tile, tmaj, avec, det),
                                          atol_rms=wg_atol)
                        ddw.free()
        # the row-staged 3x3 kernel (tile 4; the default where it applies): piece width by the row alignment, atomics / ordered
        if size == 3 and c % 32 == 0 and n % 128 == 0:
            G.dk_train_force(1, -1)
            for avec in (0, -1):
                for det in (0, 1):
                    for tile in (4, 5, -1):
                        G.dk_train_force(0, tile)
                        G.dk_train_force(2, avec)
                        G.dk_set_deterministic(det)
                        ddw = gpu.DeviceArray(dw0)
                        assert G.dk_conv_backward_weights(C.byref(d), dx.ptr, dd.ptr, ddw.ptr, None) == 0
                        ran = [k for k in _ran_kernels(gpu.lib()) if k.startswith("conv_wgrad")]
                        seg = 20 if w <= 20 else 40
                        vw = 1 if (seg == 20 or avec == 0) else 4 if w % 4 == 0 else 2 if w % 2 == 0 else 1
                        assert ran == ["conv_wgrad3_f32<%d, %d, 1>" % (seg, vw)], ran
                        seen.add((ran[0], det))
                        util.assert_close(ddw.numpy().reshape(wt.shape), ref_dw,
                                          "wgrad3 %s avec %d deterministic %d" % (case, avec, det), atol_rms=wg_atol)
                        ddw.free()
    finally:
        for k in range(3):
            G.dk_train_force(k, -1)
        G.dk_set_deterministic(-1)
        G.dk_profile_enable(0)
    assert len(seen) >= 8
    # ---- data gradient as a forward convolution on transposed / rotated filters --------------------------------
    dt = gpu.DeviceArray(n=wt.size)
    if size == 3:
        assert G.dk_transpose_weights_flip(dwt.ptr, dt.ptr, n, c, 3, None) == 0
    else:
        assert G.dk_transpose_weights(dwt.ptr, dt.ptr, n, c, 1, None) == 0
    ddesc = gpu.DkConvDesc(batch, n, h, w, c, 1, size, 1, 1, 1, pad, O.LINEAR)   # channels <-> filters
    GL = gpu.lib()
    ncfg = GL.dk_conv_force_config(-1)
    names = [GL.dk_conv_config_name(i).decode() for i in range(ncfg)]
    du = None
    nu = GL.dk_conv_wino_weights_size(C.byref(ddesc))
    if nu:
        du = gpu.DeviceArray(n=nu)
        assert GL.dk_conv_wino_transform_weights(C.byref(ddesc), dt.ptr, du.ptr, None) == 0
        GL.dk_conv_wino_register(dt.ptr, du.ptr)
    families = set()
    try:
        for cfg in range(ncfg):
            if not G.dk_conv_config_can_run(C.byref(ddesc), cfg):
                continue
            GL.dk_conv_force_config(cfg)
            G.dk_profile_enable(1)
            dprev = gpu.DeviceArray(np.full_like(x, 3.0))
            assert GL.dk_conv_forward(C.byref(ddesc), dd.ptr, dt.ptr, None, dprev.ptr, None, None, None) == 0
            ran = _ran_kernels(GL)
            G.dk_profile_enable(0)
            assert len(ran) == 1, (names[cfg], ran)
            fam = names[cfg].split("_")[0] if not names[cfg][0].isdigit() else "gather"
            kn = list(ran)[0]
            assert {"gather": "conv_igemm_f32", "direct3x3": "conv3x3_direct_f32", "dma1x1": "conv1x1_dma_f32",
                    "wino": "conv3x3_wino_f32"}[fam] in kn, "config %s ran %s" % (names[cfg], kn)
            families.add(fam)
            util.assert_close(dprev.numpy().reshape(x.shape), ref_prev, "dgrad as convolution %s through %s" % (case, names[cfg]))
            dprev.free()
        # and the gather kernel in data-gradient mode (the form the stride-2 layers use), every gather tile
        dt2 = gpu.DeviceArray(n=wt.size)
        assert G.dk_transpose_weights(dwt.ptr, dt2.ptr, n, c, size, None) == 0
        for cfg in range(ncfg):
            if not names[cfg][0].isdigit():
                break
            GL.dk_conv_force_config(cfg)
            dprev = gpu.DeviceArray(np.full_like(x, 3.0))
            assert G.dk_conv_backward_data(C.byref(d), dd.ptr, dt2.ptr, dprev.ptr, None) == 0
            util.assert_close(dprev.numpy().reshape(x.shape), ref_prev, "dgrad (gather mode 1) %s through %s" % (case, names[cfg]))
            dprev.free()
    finally:
        GL.dk_conv_force_config(-1)
        G.dk_profile_enable(0)
        if du is not None:
            GL.dk_conv_wino_register(dt.ptr, None)
    print("data gradient of %s ran through: %s" % (case, sorted(families)))
    assert "gather" in families and (size == 1 or "wino" in families or n % 4 or c % 64)


@pytest.mark.parametrize("case", PARITY_CASES)
def test_conv_backward_data_parity_classes_vs_oracle(gpu, case):
    """dk_conv_backward_data_tapmajor (stride-2 layers: pixels enumerated by parity class, tap-major contraction
    index, only the matching taps visited) against the oracle's col2im data gradient, through every gather tile
    shape, and against the masked-gather form of the same library."""
    batch, c, h, w, n, size = case
    stride, pad = 2, size // 2 if size == 3 else 0
    L, G = O.lib(), bind(gpu.lib())
    rng = np.random.default_rng(util.seed_of(case))
    oh, ow = (h + 2 * pad - size) // stride + 1, (w + 2 * pad - size) // stride + 1
    x = rng.uniform(-1, 1, (batch, c, h, w)).astype(np.float32)
    wt = (rng.uniform(-1, 1, (n, c, size, size)) * 0.2).astype(np.float32)
    delta = rng.uniform(-1, 1, (batch, n, oh, ow)).astype(np.float32)
    ref_dw, ref_prev = np.zeros_like(wt), np.full_like(x, 3.0)
    ws = np.zeros(oh * ow * size * size * c + 1, np.float32)
    L.orc_conv_backward(O.fptr(x), O.fptr(wt), O.fptr(delta), O.fptr(ref_dw), O.fptr(ref_prev), O.fptr(ws),
                        batch, c, h, w, n, 1, size, stride, stride, 1, pad)
    d = gpu.DkConvDesc(batch, c, h, w, n, 1, size, stride, stride, 1, pad, O.LINEAR)
    dwt, dd = gpu.DeviceArray(wt), gpu.DeviceArray(delta)
    dt = gpu.DeviceArray(n=wt.size)
    assert G.dk_transpose_weights_tapmajor(dwt.ptr, dt.ptr, n, c, size, None) == 0
    ncfg = gpu.lib().dk_conv_force_config(-1)
    names = []
    try:
        for cfg in range(-1, ncfg):
            if cfg >= 0 and not gpu.lib().dk_conv_config_name(cfg).decode()[0].isdigit():
                break     # the gather shapes come first; direct / DMA / Winograd shapes do not take a data gradient
            gpu.lib().dk_conv_force_config(cfg)
            dprev = gpu.DeviceArray(np.full_like(x, 3.0))
            assert G.dk_conv_backward_data_tapmajor(C.byref(d), dd.ptr, dt.ptr, dprev.ptr, None) == 0
            util.assert_close(dprev.numpy().reshape(x.shape), ref_prev, "parity dgrad %s cfg %d" % (case, cfg))
            names.append(cfg)
    finally:
        gpu.lib().dk_conv_force_config(-1)
    assert len(names) >= 8
    # and the masked form (every tap multiplied, zeros gathered) agrees to summation-order noise
    dt2, dprev2 = gpu.DeviceArray(n=wt.size), gpu.DeviceArray(np.full_like(x, 3.0))
    assert G.dk_transpose_weights(dwt.ptr, dt2.ptr, n, c, size, None) == 0
    assert G.dk_conv_backward_data(C.byref(d), dd.ptr, dt2.ptr, dprev2.ptr, None) == 0
    util.assert_close(dprev.numpy(), dprev2.numpy(), "parity vs masked dgrad %s" % (case,))
    # a layer outside the form is refused, not mis-computed
    d_odd = gpu.DkConvDesc(batch, c, h + 1, w, n, 1, size, stride, stride, 1, pad, O.LINEAR)
    assert G.dk_conv_backward_data_tapmajor(C.byref(d_odd), dd.ptr, dt.ptr, dprev.ptr, None) == 1


def test_batchnorm_forward_backward_vs_oracle(gpu):
    L, G = O.lib(), bind(gpu.lib())
    rng = np.random.default_rng(5)
    batch, c, sp = 3, 7, 11 * 13
    raw = rng.uniform(-2, 2, (batch, c, sp)).astype(np.float32)
    scales = rng.uniform(.5, 1.5, c).astype(np.float32)
    biases = rng.uniform(-.5, .5, c).astype(np.float32)
    rm0, rv0 = rng.uniform(-.1, .1, c).astype(np.float32), rng.uniform(.5, 1.5, c).astype(np.float32)
    for act in (O.LEAKY, O.MISH):
        out = raw.copy()
        rm, rv = rm0.copy(), rv0.copy()
        mean, var = np.zeros(c, np.float32), np.zeros(c, np.float32)
        xs, xn = np.zeros_like(raw), np.zeros_like(raw)
        L.orc_batchnorm_forward(O.fptr(out), batch, c, sp, O.fptr(scales), O.fptr(biases), O.fptr(rm), O.fptr(rv),
                                O.fptr(mean), O.fptr(var), O.fptr(xs), O.fptr(xn), 1)
        pre = out.copy()
        if act == O.MISH:
            L.orc_activate_array_mish(O.fptr(out), out.size, None, O.fptr(out))
        else:
            L.orc_activate_array(O.fptr(out), out.size, act)
        d = {k: gpu.DeviceArray(v) for k, v in dict(raw=raw, sc=scales, bi=biases, rm=rm0, rv=rv0).items()}
        for k in ("xn", "ain", "out"):
            d[k] = gpu.DeviceArray(n=raw.size)
        d["mean"], d["var"] = gpu.DeviceArray(n=c), gpu.DeviceArray(n=c)
        assert G.dk_bn_forward_train(d["raw"].ptr, d["raw"].ptr, d["xn"].ptr, d["ain"].ptr, d["out"].ptr,
                                     d["mean"].ptr, d["var"].ptr, d["rm"].ptr, d["rv"].ptr, d["sc"].ptr,
                                     d["bi"].ptr, batch, c, sp, act, 1, None) == 0
        util.assert_close(d["mean"].numpy(), mean, "bn mean", rel=1e-5)
        util.assert_close(d["var"].numpy(), var, "bn variance", rel=1e-5)
        util.assert_close(d["rm"].numpy(), rm, "rolling mean", rel=1e-5)
        util.assert_close(d["rv"].numpy(), rv, "rolling variance", rel=1e-5)
        util.assert_close(d["xn"].numpy().reshape(raw.shape), xn, "x_norm")
        util.assert_close(d["ain"].numpy().reshape(raw.shape), pre, "pre-activation")
        util.assert_close(d["out"].numpy().reshape(raw.shape), out, "bn+act output")
        # backward
        delta = rng.uniform(-1, 1, raw.shape).astype(np.float32)
        rdelta = delta.copy()
        su0 = rng.uniform(-1, 1, c).astype(np.float32)
        su = su0.copy()
        md, vd = np.zeros(c, np.float32), np.zeros(c, np.float32)
        L.orc_batchnorm_backward(O.fptr(rdelta), batch, c, sp, O.fptr(scales), O.fptr(xs), O.fptr(xn), O.fptr(mean),
                                 O.fptr(var), O.fptr(md), O.fptr(vd), O.fptr(su))
        dd, dsu, dbu = gpu.DeviceArray(delta), gpu.DeviceArray(su0), gpu.DeviceArray(np.zeros(c, np.float32))
        dmd, dvd = gpu.DeviceArray(n=c), gpu.DeviceArray(n=c)
        dxs, dxn, dm, dv = gpu.DeviceArray(xs), gpu.DeviceArray(xn), gpu.DeviceArray(mean), gpu.DeviceArray(var)
        assert G.dk_bn_backward(dd.ptr, dxs.ptr, dxn.ptr, dm.ptr, dv.ptr, d["sc"].ptr, dmd.ptr, dvd.ptr, dsu.ptr,
                                dbu.ptr, batch, c, sp, None) == 0
        util.assert_close(dsu.numpy(), su, "scale_updates", rel=2e-5)
        util.assert_close(dd.numpy().reshape(raw.shape), rdelta, "bn delta")
        # quirk 3: the CPU reference never fills bias_updates for BN layers; here it is the true sum(delta)
        util.assert_close(dbu.numpy(), delta.sum(axis=(0, 2), dtype=np.float64).astype(np.float32), "bias_updates", rel=2e-5)
        # fused activation-gradient + BN backward (recomputes x_norm / pre-activation from x):
        # oracle = gradient_array on the output delta, then backward_batchnorm
        gdelta = delta.copy()
        if act == O.MISH:
            L.orc_gradient_array_mish(pre.size, O.fptr(pre), O.fptr(gdelta))
        else:
            L.orc_gradient_array(O.fptr(out), out.size, act, O.fptr(gdelta))
        su2 = su0.copy()
        md2, vd2 = np.zeros(c, np.float32), np.zeros(c, np.float32)
        gsum = gdelta.sum(axis=(0, 2), dtype=np.float64).astype(np.float32)
        L.orc_batchnorm_backward(O.fptr(gdelta), batch, c, sp, O.fptr(scales), O.fptr(xs), O.fptr(xn), O.fptr(mean),
                                 O.fptr(var), O.fptr(md2), O.fptr(vd2), O.fptr(su2))
        fd, fsu, fbu = gpu.DeviceArray(delta), gpu.DeviceArray(su0), gpu.DeviceArray(np.zeros(c, np.float32))
        assert G.dk_bn_act_backward(fd.ptr, dxs.ptr, dm.ptr, dv.ptr, d["sc"].ptr, d["bi"].ptr, dmd.ptr, dvd.ptr,
                                    fsu.ptr, fbu.ptr, batch, c, sp, act, None) == 0
        util.assert_close(fsu.numpy(), su2, "fused scale_updates", rel=2e-5)
        util.assert_close(fbu.numpy(), gsum, "fused bias_updates", rel=2e-5)
        util.assert_close(fd.numpy().reshape(raw.shape), gdelta, "fused act+bn delta")


def test_glue_backward_and_sgd(gpu):
    L, G = O.lib(), bind(gpu.lib())
    rng = np.random.default_rng(9)
    # activation gradients
    y = rng.uniform(-2, 2, 5000).astype(np.float32)
    for act in (O.LEAKY, O.LOGISTIC, O.LINEAR, O.MISH):
        d0 = rng.uniform(-1, 1, y.size).astype(np.float32)
        ref = d0.copy()
        if act == O.MISH:
            L.orc_gradient_array_mish(y.size, O.fptr(y), O.fptr(ref))
        else:
            L.orc_gradient_array(O.fptr(y), y.size, act, O.fptr(ref))
        dy, dd = gpu.DeviceArray(y), gpu.DeviceArray(d0)
        assert G.dk_gradient_array(dy.ptr, dy.ptr, dd.ptr, y.size, act, None) == 0
        util.assert_close(dd.numpy(), ref, "gradient act %d" % act, rel=1e-5)
    # backward bias
    delta = rng.uniform(-1, 1, (3, 5, 77)).astype(np.float32)
    bu = rng.uniform(-1, 1, 5).astype(np.float32)
    ref = bu.copy()
    L.orc_backward_bias(O.fptr(ref), O.fptr(delta), 3, 5, 77)
    dbu, dd = gpu.DeviceArray(bu), gpu.DeviceArray(delta)
    assert G.dk_backward_bias(dbu.ptr, dd.ptr, 3, 5, 77, None) == 0
    util.assert_close(dbu.numpy(), ref, "backward_bias", rel=2e-5)
    # maxpool backward (SPP-style overlapping windows) through forward indexes
    x = rng.uniform(-1, 1, (2, 3, 19, 19)).astype(np.float32)
    yv = np.zeros((2, 3, 19, 19), np.float32)
    idx = np.zeros(yv.shape, np.int32)
    L.orc_maxpool_forward(O.fptr(x), O.fptr(yv), O.iptr(idx), 2, 3, 19, 19, 5, 1, 1, 4)
    dl = rng.uniform(-1, 1, yv.shape).astype(np.float32)
    prev0 = rng.uniform(-1, 1, x.shape).astype(np.float32)
    ref = prev0.copy()
    L.orc_maxpool_backward(O.fptr(dl), O.iptr(idx), dl.size, O.fptr(ref))
    ddl, didx, dprev = gpu.DeviceArray(dl), gpu.DeviceArray(idx, dtype=np.int32), gpu.DeviceArray(prev0)
    assert G.dk_maxpool_backward(ddl.ptr, didx.ptr, dl.size, dprev.ptr, None) == 0
    util.assert_close(dprev.numpy().reshape(x.shape), ref, "maxpool backward")
    # route backward (groups = 2, group_id = 1), shortcut backward, upsample backward
    batch = 3
    ld = rng.uniform(-1, 1, (batch, 40)).astype(np.float32)
    src0 = rng.uniform(-1, 1, (batch, 48)).astype(np.float32)
    ref = src0.copy()
    L.orc_route_backward(O.fptr(ld), 40, 16, 48, 2, 1, batch, O.fptr(ref))
    dld, dsrc = gpu.DeviceArray(ld), gpu.DeviceArray(src0)
    assert G.dk_route_backward(dld.ptr, 40, 16, 48, 2, 1, batch, dsrc.ptr, None) == 0
    assert np.array_equal(dsrc.numpy().reshape(ref.shape), ref)
    a0, b0 = rng.uniform(-1, 1, 999).astype(np.float32), rng.uniform(-1, 1, 999).astype(np.float32)
    dl = rng.uniform(-1, 1, 999).astype(np.float32)
    ra, rb = a0.copy(), b0.copy()
    L.orc_shortcut_backward(O.fptr(dl), 999, O.fptr(ra), O.fptr(rb))
    da, db, ddl = gpu.DeviceArray(a0), gpu.DeviceArray(b0), gpu.DeviceArray(dl)
    assert G.dk_shortcut_backward(ddl.ptr, 999, da.ptr, db.ptr, None) == 0
    assert np.array_equal(da.numpy(), ra) and np.array_equal(db.numpy(), rb)
    w, h, c = 7, 5, 3
    dl = rng.uniform(-1, 1, (batch, c, h * 2, w * 2)).astype(np.float32)
    p0 = rng.uniform(-1, 1, (batch, c, h, w)).astype(np.float32)
    ref = p0.copy()
    L.orc_upsample_backward(O.fptr(dl), w, h, c, batch, 2, F(1.0), O.fptr(ref))
    ddl, dp = gpu.DeviceArray(dl), gpu.DeviceArray(p0)
    assert G.dk_upsample_backward(ddl.ptr, w, h, c, batch, 2, 1.0, dp.ptr, None) == 0
    assert np.array_equal(dp.numpy().reshape(ref.shape), ref)
    # SGD: same float operations in the same order -> bit-exact
    n = 1000
    wv, wu = rng.uniform(-1, 1, n).astype(np.float32), rng.uniform(-1, 1, n).astype(np.float32)
    bv, bu = rng.uniform(-1, 1, 10).astype(np.float32), rng.uniform(-1, 1, 10).astype(np.float32)
    rw, rwu, rb2, rbu = wv.copy(), wu.copy(), bv.copy(), bu.copy()
    L.orc_conv_update(O.fptr(rw), O.fptr(rwu), n, O.fptr(rb2), O.fptr(rbu), None, None, 10, 64, F(0.00261), F(0.949), F(0.0005))
    dw, dwu, dbv, dbu = gpu.DeviceArray(wv), gpu.DeviceArray(wu), gpu.DeviceArray(bv), gpu.DeviceArray(bu)
    assert G.dk_sgd_update(dw.ptr, dwu.ptr, n, 64, 0.00261, 0.949, 0.0005, 1, None) == 0
    assert G.dk_sgd_update(dbv.ptr, dbu.ptr, 10, 64, 0.00261, 0.949, 0.0005, 0, None) == 0
    assert np.array_equal(dw.numpy(), rw) and np.array_equal(dwu.numpy(), rwu)
    assert np.array_equal(dbv.numpy(), rb2) and np.array_equal(dbu.numpy(), rbu)


def test_tiny_train_step_vs_oracle_and_reference_golden(gpu, tmp_path):
    """One yolov4-tiny train step (b=2) through the network API: forward with batch
    statistics, backward driven by the REAL reference's yolo deltas (golden fixture),
    SGD update; compared with the oracle and the reference's gradient summaries."""
    g, cfg, wpath, x = train_fixture(tmp_path)
    L = gpu.lib()
    L.DkSetYoloDelta.argtypes = [VP, C.c_int, VP]
    L.DkSetMaxIter.argtypes = [VP, C.c_int]
    L.TrainNetworkDatum.argtypes = [VP, VP, VP]
    L.TrainNetworkDatum.restype = C.c_float
    L.UpdateNetworkGpu.argtypes = [VP]
    L.DkLayerPull.argtypes = [VP, C.c_int, C.c_int, VP, C.c_size_t]
    L.DkLayerPull.restype = C.c_long
    net = netutil.DkNet(gpu, cfg, wpath, train=True)
    assert net.batch == int(g["batch"])
    onet = O.load_network_train(cfg, wpath, None)
    O.forward_train(onet, x)
    inject_yolo_deltas(onet, g)
    keep = []
    for i, l in enumerate(onet.layers):
        if l.type == O.YOLO:
            d = np.ascontiguousarray(l.delta.ravel())
            keep.append(d)
            L.DkSetYoloDelta(net.p, i, d.ctypes.data)
    truth = np.ascontiguousarray(g["truth"])
    L.TrainNetworkDatum(net.p, np.ascontiguousarray(x).ctypes.data, truth.ctypes.data)

    def pull(i, which, n):
        out = np.empty(n, np.float32)
        assert L.DkLayerPull(net.p, i, which, out.ctypes.data, n) == n
        return out

    # BN conv bias_updates: quirk 3 (oracle = CPU reference leaves them 0) -> compare with sum of
    # the oracle's delta taken BEFORE its BN backward; re-run the oracle's backward step by step
    worst = 0.0
    for i, l in enumerate(onet.layers):
        st = util.assert_close(net.output(i), l.output, "train forward layer %d" % i, atol_rms=util.TRAIN_ATOL_RMS)
        worst = max(worst, st["max_abs_over_rms"])
    print("train forward: worst max|d|/rms %.3g" % worst)
    # Where the train-mode tolerance comes from: the same forward with the oracle's batch statistics
    # accumulated in double (analysis variant orc_set_bn_stats_f64; the HIP kernels reduce in fp64 too).
    # HIP agrees with THAT oracle to 1e-4 x rms; the reference-faithful oracle (sequential fp32 sums of
    # 86 k ... 346 k terms here) sits further from it than HIP does.
    O.lib().orc_set_bn_stats_f64(1)
    try:
        o64 = O.load_network_train(cfg, wpath, None)
        O.forward_train(o64, x)
    finally:
        O.lib().orc_set_bn_stats_f64(0)
    w_hip = w_ref = 0.0
    for i, (l64, lref) in enumerate(zip(o64.layers, onet.layers)):
        st = util.assert_close(net.output(i), l64.output, "train forward layer %d vs the fp64-statistics oracle" % i, atol_rms=1e-4)
        w_hip = max(w_hip, st["max_abs_over_rms"])
        w_ref = max(w_ref, util.rel_err_stats(lref.output, l64.output)["max_abs_over_rms"])
    print("train forward vs the fp64-statistics oracle: HIP worst |d|/rms %.3g, reference-faithful oracle %.3g" % (w_hip, w_ref))
    assert w_hip < w_ref
    # The backward pass contains kinks (leaky slope, maxpool argmax): an activation that
    # sits within the forward tolerance of 0 can take the other branch and change its
    # gradient by 10x.  To check the BACKWARD arithmetic independently of that, the
    # oracle's backward is evaluated on the HIP path's own forward activations.
    for i, l in enumerate(onet.layers):
        l.output = net.output(i).reshape(l.output.shape).copy()
    for i, l in enumerate(onet.layers):
        if l.type == O.MAXPOOL:  # argmax of the same activations
            tmp = np.zeros_like(l.output)
            O.lib().orc_maxpool_forward(O.fptr(onet.layers[i - 1].output), O.fptr(tmp), O.iptr(l.indexes), l.batch,
                                        l.c, l.h, l.w, l.size, l.stride_x, l.stride_y, l.pad)
    O.backward(onet)
    which_name = {7: "weight_updates", 8: "bias_updates", 9: "scale_updates"}
    for i, l in reversed(list(enumerate(onet.layers))):
        if l.type != O.CONVOLUTIONAL:
            if i > 0 and l.type != O.YOLO:
                util.assert_close(pull(i, 6, l.batch * l.outputs), l.delta.ravel(), "delta layer %d (type %d)" % (i, l.type), rel=2e-4, atol_rms=2 * util.TRAIN_ATOL_RMS)
            continue
        if i > 0:
            util.assert_close(pull(i, 6, l.batch * l.outputs), l.delta.ravel(), "delta layer %d" % i, rel=2e-4, atol_rms=2 * util.TRAIN_ATOL_RMS)
        util.assert_close(pull(i, 7, l.nweights), l.weight_updates, "weight_updates layer %d" % i, rel=2e-4, atol_rms=2 * util.TRAIN_ATOL_RMS)
        if l.batch_normalize:
            util.assert_close(pull(i, 9, l.n), l.scale_updates, "scale_updates layer %d" % i, rel=2e-4, atol_rms=2 * util.TRAIN_ATOL_RMS)
        else:
            util.assert_close(pull(i, 8, l.n), l.bias_updates, "bias_updates layer %d" % i, rel=2e-4, atol_rms=2 * util.TRAIN_ATOL_RMS)
        if i > 0:
            util.assert_close(pull(i, 6, l.batch * l.outputs), l.delta.ravel(), "delta layer %d" % i, rel=2e-4, atol_rms=2 * util.TRAIN_ATOL_RMS)
    # the REAL reference's summaries of each gradient tensor (golden fixture).  A few
    # activations sit on a kink (see above) and take the other branch than the reference
    # did, so these whole-tensor checksums are compared loosely: L2 norm within 0.5 %
    # (the plain sums are printed, not asserted).
    worst_norm = worst_sum = 0.0
    for row in g["grad_summaries"]:
        i, which = int(row[0]), int(row[1])
        if which not in (7, 9):
            continue
        n = onet.layers[i].nweights if which == 7 else onet.layers[i].n
        a = pull(i, which, n).astype(np.float64)
        rms = np.sqrt(row[3] / n)
        dn = abs(np.sqrt((a * a).sum()) / np.sqrt(row[3]) - 1)
        worst_norm = max(worst_norm, dn)
        assert dn < 5e-3, (i, which, dn)
        worst_sum = max(worst_sum, abs(a.sum() - row[2]) / (rms * np.sqrt(n)))
    print("gradient tensors vs the reference: worst L2-norm deviation %.3g, worst |sum diff|/(rms*sqrt(n)) %.3g"
          % (worst_norm, worst_sum))
    # SGD update with the reference's learning rate schedule
    L.DkSetMaxIter(net.p, 1000)
    L.UpdateNetworkGpu(net.p)
    O.update(onet, onet.batch * onet.subdiv, float(g["lr"]), onet.momentum, onet.decay)
    for i, l in enumerate(onet.layers):
        if l.type == O.CONVOLUTIONAL:
            util.assert_close(pull(i, 1, l.nweights), l.weights, "updated weights layer %d" % i, atol_rms=util.TRAIN_ATOL_RMS)
    net.close()


def test_real_train_step_with_host_loss(gpu, tmp_path):
    """TrainNetworkDatum with truth boxes: the yolo loss runs on the host
    (csrc/host/yolo_loss.cpp, pinned bit-exactly against the reference on CPU), its
    delta is pushed and the backward sweep runs.  Cost vs the REAL reference's cost for
    the same step (golden), weight gradients vs the run driven by the reference's deltas."""
    g, cfg, wpath, x = train_fixture(tmp_path)
    L = gpu.lib()
    L.TrainNetworkDatum.argtypes = [VP, VP, VP]
    L.TrainNetworkDatum.restype = C.c_float
    L.DkLayerPull.argtypes = [VP, C.c_int, C.c_int, VP, C.c_size_t]
    L.DkLayerPull.restype = C.c_long
    net = netutil.DkNet(gpu, cfg, wpath, train=True)
    truth = np.ascontiguousarray(g["truth"])
    cost = L.TrainNetworkDatum(net.p, np.ascontiguousarray(x).ctypes.data, truth.ctypes.data)
    assert abs(cost - float(g["cost"])) <= 2e-3 * float(g["cost"]), (cost, float(g["cost"]))
    # yolo deltas produced on the host from the HIP forward vs the reference's deltas
    for i in range(net.n):
        f = net.info(i)
        if f["type"] != O.YOLO:
            continue
        n = f["batch"] * f["outputs"]
        d = np.empty(n, np.float32)
        assert L.DkLayerPull(net.p, i, 6, d.ctypes.data, n) == n
        ref = np.zeros(n, np.float32)
        ref[g["yolo_%d_delta_idx" % i]] = g["yolo_%d_delta_val" % i]
        assert np.count_nonzero(d) == np.count_nonzero(ref)
        util.assert_close(d, ref, "yolo %d delta" % i, rel=1e-3, atol_rms=1e-3)
    net.close()


def test_two_replicas_with_summed_bucket_equal_subdivisions(gpu, tmp_path):
    """Multi-GPU equivalence on one GPU: two replicas, each on its half of the batch,
    gradient buckets summed (what the RCCL all-reduce does), update with B = 2*b  ==
    one replica run with subdivisions = 2 (the reference's own accumulation)."""
    g, cfg, wpath, x = train_fixture(tmp_path)
    L = gpu.lib()
    for fn, at, rt in (("DkGradBucketSize", [VP], C.c_size_t), ("DkAttachGradBucket", [VP, VP], None),
                       ("DkSetSubdivisions", [VP, C.c_int], None), ("DkSetReplicas", [VP, C.c_int], None),
                       ("DkAdvanceIteration", [VP], None),
                       ("TrainNetworkDatum", [VP, VP, VP], C.c_float), ("UpdateNetworkGpu", [VP], None),
                       ("DkSetMaxIter", [VP, C.c_int], None), ("DkLayerPull", [VP, C.c_int, C.c_int, VP, C.c_size_t], C.c_long)):
        getattr(L, fn).argtypes = at
        getattr(L, fn).restype = rt
    B = int(g["batch"])
    one = str(tmp_path / "one.cfg")
    open(one, "w").write(open(cfg).read().replace("batch=%d" % B, "batch=1"))
    truth = np.ascontiguousarray(g["truth"])
    xs = [np.ascontiguousarray(x[i:i + 1]) for i in range(B)]
    ts = [np.ascontiguousarray(truth[i:i + 1]) for i in range(B)]
    # (a) one replica, batch=2 subdivisions=2 -> net->batch = 1, two accumulating sub-steps
    sub = str(tmp_path / "sub.cfg")
    open(sub, "w").write(open(cfg).read().replace("subdivisions=1", "subdivisions=%d" % B))
    ref = netutil.DkNet(gpu, sub, wpath, train=True)
    assert ref.batch == 1
    L.DkSetMaxIter(ref.p, 1000)
    STEPS = 2   # the second step exercises the momentum carried in the gradient buffers
    for _ in range(STEPS):
        for i in range(B):
            L.TrainNetworkDatum(ref.p, xs[i].ctypes.data, ts[i].ctypes.data)
        L.DkAdvanceIteration(ref.p)
        L.UpdateNetworkGpu(ref.p)
    # (b) two replicas with attached buckets; emulate the all-reduce with axpy
    reps, buckets = [], []
    for i in range(B):
        r = netutil.DkNet(gpu, one, wpath, train=True)
        n = L.DkGradBucketSize(r.p)
        b = gpu.DeviceArray(np.zeros(n, np.float32))
        L.DkAttachGradBucket(r.p, b.ptr)
        L.DkSetReplicas(r.p, B)
        L.DkSetMaxIter(r.p, 1000)
        reps.append(r)
        buckets.append(b)
    for _ in range(STEPS):
        for i in range(B):
            L.TrainNetworkDatum(reps[i].p, xs[i].ctypes.data, ts[i].ctypes.data)
        total = buckets[0].numpy() + buckets[1].numpy()
        for b in buckets:
            L.cuda_push_array(b.ptr, total.ctypes.data, total.size)
        for r in reps:
            L.DkAdvanceIteration(r.p)
            L.UpdateNetworkGpu(r.p)

    def weights(net, i, n):
        out = np.empty(n, np.float32)
        assert L.DkLayerPull(net.p, i, 1, out.ctypes.data, n) == n
        return out
    for i in range(ref.n):
        f = ref.info(i)
        if f["type"] == O.CONVOLUTIONAL:
            a, b0, b1 = weights(ref, i, f["nweights"]), weights(reps[0], i, f["nweights"]), weights(reps[1], i, f["nweights"])
            assert np.array_equal(b0, b1), "replicas diverged"
            util.assert_close(b0, a, "weights after %d steps, layer %d" % (STEPS, i), rel=2e-5, atol_rms=2e-6)
    for r in reps + [ref]:
        r.close()


def test_split_train_step_equals_train_network_datum(gpu, tmp_path):
    """The split step the overlapped all-reduce uses (DkTrainForward, DkBackwardRange per bucket
    segment, DkTrainFinish) produces bitwise the gradients and the cost of TrainNetworkDatum."""
    from darknet_amd.train_dist import bucket_segments
    g, cfg, wpath, x = train_fixture(tmp_path)
    L = gpu.lib()
    for fn, at, rt in (("TrainNetworkDatum", [VP, VP, VP], C.c_float), ("DkTrainForward", [VP, VP, VP], None),
                       ("DkBackwardRange", [VP, C.c_int, C.c_int], None), ("DkTrainFinish", [VP], C.c_float),
                       ("DkGradBucketOffset", [VP, C.c_int], C.c_size_t),
                       ("DkLayerPull", [VP, C.c_int, C.c_int, VP, C.c_size_t], C.c_long)):
        getattr(L, fn).argtypes = at
        getattr(L, fn).restype = rt
    truth = np.ascontiguousarray(g["truth"])
    xin = np.ascontiguousarray(x)
    a = netutil.DkNet(gpu, cfg, wpath, train=True)
    cost_a = L.TrainNetworkDatum(a.p, xin.ctypes.data, truth.ctypes.data)
    b = netutil.DkNet(gpu, cfg, wpath, train=True)
    offs = [L.DkGradBucketOffset(b.p, i) for i in range(b.n + 1)]
    convs = [i for i in range(b.n) if offs[i + 1] > offs[i]]
    segs = bucket_segments(convs, [offs[i + 1] - offs[i] for i in convs], b.n, 3)
    assert len(segs) >= 2
    L.DkTrainForward(b.p, xin.ctypes.data, truth.ctypes.data)
    for hi, lo, off, cnt in segs:
        L.DkBackwardRange(b.p, hi, lo)
    cost_b = L.DkTrainFinish(b.p)
    assert cost_a == cost_b
    for i in convs:
        f = a.info(i)
        for which, n in ((7, f["nweights"]), (8, f["n"])):
            ga, gb = np.empty(n, np.float32), np.empty(n, np.float32)
            assert L.DkLayerPull(a.p, i, which, ga.ctypes.data, n) == n
            assert L.DkLayerPull(b.p, i, which, gb.ctypes.data, n) == n
            # float atomics in the weight gradient make two runs agree only to rounding
            util.assert_close(gb, ga, "layer %d gradient %d" % (i, which), rel=1e-4, atol_rms=1e-5)
    a.close(); b.close()


def test_derived_weights_launch_equals_per_layer_form(gpu, tmp_path):
    """From the second step on, the transposed / rotated / tap-major data-gradient weights and the Winograd filters
    come from ONE launch per step (DkTrainPrepRun) into per-layer buffers, and the weight gradients run on a second
    stream.  Three steps in that form and three with the per-layer launches on one stream (DkSetTrainPrep(0),
    DkSetTrainStreams(0)) must train the same network: the tensors are copies / the same transform of the same
    weights and the streams only reorder independent kernels, so the only difference left is the float-atomic order
    of the weight gradient."""
    g, cfg, wpath, x = train_fixture(tmp_path)
    L = gpu.lib()
    for fn, at, rt in (("TrainNetworkDatum", [VP, VP, VP], C.c_float), ("UpdateNetworkGpu", [VP], None),
                       ("DkAdvanceIteration", [VP], None), ("DkSetMaxIter", [VP, C.c_int], None),
                       ("DkSetTrainPrep", [C.c_int], None), ("DkSetTrainStreams", [C.c_int], None),
                       ("DkLayerTrainCfg", [VP, C.c_int, C.c_int], C.c_int),
                       ("DkLayerPull", [VP, C.c_int, C.c_int, VP, C.c_size_t], C.c_long)):
        getattr(L, fn).argtypes = at
        getattr(L, fn).restype = rt
    truth = np.ascontiguousarray(g["truth"])
    xin = np.ascontiguousarray(x)
    res = {}
    try:
        for mode in (1, 0):
            L.DkSetTrainPrep(mode)
            L.DkSetTrainStreams(mode)
            net = netutil.DkNet(gpu, cfg, wpath, train=True)
            L.DkSetMaxIter(net.p, 1000)
            costs = []
            for _ in range(3):
                costs.append(L.TrainNetworkDatum(net.p, xin.ctypes.data, truth.ctypes.data))
                L.DkAdvanceIteration(net.p)
                L.UpdateNetworkGpu(net.p)
            convs = [i for i in range(net.n) if net.info(i)["type"] == O.CONVOLUTIONAL]
            w = {}
            for i in convs:
                n = net.info(i)["nweights"]
                out = np.empty(n, np.float32)
                assert L.DkLayerPull(net.p, i, 1, out.ctypes.data, n) == n
                w[i] = out
            kinds = {(L.DkLayerTrainCfg(net.p, i, 0), L.DkLayerTrainCfg(net.p, i, 1)) for i in convs}
            res[mode] = (costs, w, kinds)
            net.close()
    finally:
        L.DkSetTrainPrep(-1)
        L.DkSetTrainStreams(-1)
    (ca, wa, ka), (cb, wb, kb) = res[1], res[0]
    assert ka == kb and len(ka) >= 3      # same kernel choices (process-wide timing cache), several kinds of them
    assert np.allclose(ca, cb, rtol=1e-4), (ca, cb)
    for i in wa:
        util.assert_close(wa[i], wb[i], "weights of conv %d after three steps" % i, rel=1e-4, atol_rms=1e-5)


def test_maxpool_backward_gather_equals_scatter(gpu):
    """dk_maxpool_backward_gather (deterministic mode: no atomics) == dk_maxpool_backward on the SPP pools (stride 1,
    windows 5 / 9 / 13 on 19x19: overlapping windows, the case the atomics exist for) and on 2x2 / stride-2 pools,
    incl. odd maps.  Integer-valued deltas make the atomic sum order irrelevant, so the comparison is exact."""
    G = bind(gpu.lib())
    L = gpu.lib()
    i = C.c_int
    L.dk_maxpool_forward.argtypes = [VP, VP, VP, i, i, i, i, i, i, i, i, VP]
    L.dk_maxpool_forward.restype = i
    L.dk_maxpool_backward_gather.argtypes = [VP, VP, i, i, i, i, i, i, i, i, i, i, VP, VP]
    L.dk_maxpool_backward_gather.restype = i
    rng = np.random.default_rng(21)
    for batch, c, h, w, size, stride in ((2, 6, 19, 19, 5, 1), (1, 4, 19, 19, 9, 1), (2, 3, 19, 19, 13, 1),
                                         (2, 8, 26, 26, 2, 2), (1, 5, 13, 13, 2, 1), (2, 4, 15, 17, 3, 2)):
        pad = size - 1
        oh, ow = (h + pad - size) // stride + 1, (w + pad - size) // stride + 1
        x = rng.uniform(-1, 1, (batch, c, h, w)).astype(np.float32)
        delta = rng.integers(-8, 9, (batch, c, oh, ow)).astype(np.float32)
        dx, dy, didx = gpu.DeviceArray(x), gpu.DeviceArray(n=batch * c * oh * ow), gpu.DeviceArray(n=batch * c * oh * ow)
        assert L.dk_maxpool_forward(dx.ptr, dy.ptr, didx.ptr, batch, c, h, w, size, stride, stride, pad, None) == 0
        dd = gpu.DeviceArray(delta)
        base = rng.integers(-3, 4, x.shape).astype(np.float32)
        pa, pb = gpu.DeviceArray(base), gpu.DeviceArray(base)
        assert G.dk_maxpool_backward(dd.ptr, didx.ptr, delta.size, pa.ptr, None) == 0
        assert L.dk_maxpool_backward_gather(dd.ptr, didx.ptr, batch, c, h, w, oh, ow, size, stride, stride, pad, pb.ptr, None) == 0
        a, b = pa.numpy(), pb.numpy()
        assert np.array_equal(a, b), (size, stride, np.abs(a - b).max())
        assert np.abs(a - base.ravel()).sum() > 0


def test_deterministic_mode_is_bitwise_reproducible(gpu, tmp_path):
    """DkSetDeterministic(1): weight gradients, BN channel sums, bias sums and the maxpool gradient go through ordered
    workspaces instead of float / double atomics.  Two fresh networks stepped three times on the same data must then
    agree BITWISE in every weight and in the cost (same process: same kernel choices), the split step must reproduce
    the unsplit gradients bitwise, and the mode must still train the same network as the default (atomics) mode."""
    from darknet_amd.train_dist import bucket_segments
    g, cfg, wpath, x = train_fixture(tmp_path)
    L = gpu.lib()
    for fn, at, rt in (("TrainNetworkDatum", [VP, VP, VP], C.c_float), ("UpdateNetworkGpu", [VP], None),
                       ("DkAdvanceIteration", [VP], None), ("DkSetMaxIter", [VP, C.c_int], None),
                       ("DkSetDeterministic", [C.c_int], None), ("DkTrainForward", [VP, VP, VP], None),
                       ("DkBackwardRange", [VP, C.c_int, C.c_int], None), ("DkTrainFinish", [VP], C.c_float),
                       ("DkGradBucketOffset", [VP, C.c_int], C.c_size_t),
                       ("DkLayerPull", [VP, C.c_int, C.c_int, VP, C.c_size_t], C.c_long)):
        getattr(L, fn).argtypes = at
        getattr(L, fn).restype = rt
    truth = np.ascontiguousarray(g["truth"])
    xin = np.ascontiguousarray(x)

    def pull(net, i, which, n):
        out = np.empty(n, np.float32)
        assert L.DkLayerPull(net.p, i, which, out.ctypes.data, n) == n
        return out

    def run(steps=3):
        net = netutil.DkNet(gpu, cfg, wpath, train=True)
        L.DkSetMaxIter(net.p, 1000)
        costs = []
        for _ in range(steps):
            costs.append(L.TrainNetworkDatum(net.p, xin.ctypes.data, truth.ctypes.data))
            L.DkAdvanceIteration(net.p)
            L.UpdateNetworkGpu(net.p)
        convs = [i for i in range(net.n) if net.info(i)["type"] == O.CONVOLUTIONAL]
        w = {i: pull(net, i, 1, net.info(i)["nweights"]) for i in convs}
        net.close()
        return costs, w

    try:
        L.DkSetDeterministic(1)
        c1, w1 = run()
        c2, w2 = run()
        assert c1 == c2, (c1, c2)
        for i in w1:
            assert np.array_equal(w1[i].view(np.uint32), w2[i].view(np.uint32)), "conv %d weights differ between two deterministic runs" % i
        # split step == unsplit step, bitwise
        a = netutil.DkNet(gpu, cfg, wpath, train=True)
        cost_a = L.TrainNetworkDatum(a.p, xin.ctypes.data, truth.ctypes.data)
        b = netutil.DkNet(gpu, cfg, wpath, train=True)
        offs = [L.DkGradBucketOffset(b.p, i) for i in range(b.n + 1)]
        convs = [i for i in range(b.n) if offs[i + 1] > offs[i]]
        segs = bucket_segments(convs, [offs[i + 1] - offs[i] for i in convs], b.n, 3)
        L.DkTrainForward(b.p, xin.ctypes.data, truth.ctypes.data)
        for hi, lo, off, cnt in segs:
            L.DkBackwardRange(b.p, hi, lo)
        assert L.DkTrainFinish(b.p) == cost_a
        for i in convs:
            n = a.info(i)["nweights"]
            assert np.array_equal(pull(a, i, 7, n).view(np.uint32), pull(b, i, 7, n).view(np.uint32)), i
        a.close(); b.close()
        L.DkSetDeterministic(0)
        c0, w0 = run()
    finally:
        L.DkSetDeterministic(-1)
    assert np.allclose(c0, c1, rtol=1e-4), (c0, c1)
    for i in w1:
        util.assert_close(w1[i], w0[i], "deterministic vs atomics, conv %d after three steps" % i, rel=1e-4, atol_rms=1e-5)


def test_stopbackward_split_step_equals_unsplit(gpu, tmp_path):
    """`stopbackward=1` (network_kernels.cu:140-143 ends the backward sweep for good): the split step must
    not resume the sweep in the next DkBackwardRange segment -- gradients below the stop layer stay exactly what
    the unsplit step leaves there (zero), above it they agree."""
    from darknet_amd.train_dist import bucket_segments
    g, cfg, wpath, x = train_fixture(tmp_path)
    txt = open(cfg).read()
    # stop the sweep at the 8th conv section (layer 9 of yolov4-tiny's backbone)
    parts = txt.split("[convolutional]\n")
    assert len(parts) > 10
    parts[8] = "stopbackward=1\n" + parts[8]
    cfg2 = str(tmp_path / "stop.cfg")
    open(cfg2, "w").write("[convolutional]\n".join(parts))
    L = gpu.lib()
    for fn, at, rt in (("TrainNetworkDatum", [VP, VP, VP], C.c_float), ("DkTrainForward", [VP, VP, VP], None),
                       ("DkBackwardRange", [VP, C.c_int, C.c_int], None), ("DkTrainFinish", [VP], C.c_float),
                       ("DkGradBucketOffset", [VP, C.c_int], C.c_size_t),
                       ("DkLayerPull", [VP, C.c_int, C.c_int, VP, C.c_size_t], C.c_long)):
        getattr(L, fn).argtypes = at
        getattr(L, fn).restype = rt
    truth = np.ascontiguousarray(g["truth"])
    xin = np.ascontiguousarray(x)
    a = netutil.DkNet(gpu, cfg2, wpath, train=True)
    cost_a = L.TrainNetworkDatum(a.p, xin.ctypes.data, truth.ctypes.data)
    b = netutil.DkNet(gpu, cfg2, wpath, train=True)
    offs = [L.DkGradBucketOffset(b.p, i) for i in range(b.n + 1)]
    convs = [i for i in range(b.n) if offs[i + 1] > offs[i]]
    segs = bucket_segments(convs, [offs[i + 1] - offs[i] for i in convs], b.n, 4)
    L.DkTrainForward(b.p, xin.ctypes.data, truth.ctypes.data)
    for hi, lo, off, cnt in segs:
        L.DkBackwardRange(b.p, hi, lo)
    cost_b = L.DkTrainFinish(b.p)
    assert cost_a == cost_b
    zero_layers = 0
    for i in convs:
        f = a.info(i)
        ga, gb = np.empty(f["nweights"], np.float32), np.empty(f["nweights"], np.float32)
        assert L.DkLayerPull(a.p, i, 7, ga.ctypes.data, ga.size) == ga.size
        assert L.DkLayerPull(b.p, i, 7, gb.ctypes.data, gb.size) == gb.size
        if not ga.any():
            zero_layers += 1
            assert not gb.any(), "layer %d below the stopbackward layer received a gradient in the split step" % i
        else:
            util.assert_close(gb, ga, "layer %d gradient" % i, rel=1e-4, atol_rms=1e-5)
    assert zero_layers >= 3, "the stop did not cut any layer off (fixture does not exercise stopbackward)"
    a.close(); b.close()


def test_train_networks_c_entry_equals_subdivisions(gpu, tmp_path):
    """The C-level data-parallel entry (TrainNetworks, csrc/host/multigpu.cpp; reference signature
    network_kernels.cu:446-484): two replicas in one Network array, one host thread each, gradient
    buckets summed every iteration.  On a one-GPU box both replicas sit on device 0, so the collective
    is the library's local-sum path (RCCL needs distinct devices); everything else -- threads, bucket
    attachment, B = batch x subdivisions x replicas, momentum carry-over -- is the multi-GPU code.
    Must equal ONE replica run with subdivisions = 2 over two iterations."""
    g, cfg, wpath, x = train_fixture(tmp_path)
    L = gpu.lib()
    for fn, at, rt in (("DkNetworkArrayCreate", [C.c_int], VP), ("DkNetworkArrayAt", [VP, C.c_int], VP),
                       ("DkNetworkArrayDestroy", [VP, C.c_int], None),
                       ("LoadNetwork", [VP, C.c_char_p, C.c_char_p, C.c_bool, C.c_bool], C.c_bool),
                       ("DkTrainNetworksFlat", [VP, C.c_int, VP, C.c_int, VP, C.c_int, C.c_int, C.c_int], C.c_float),
                       ("DkSetMaxIter", [VP, C.c_int], None), ("DkSetSubdivisions", [VP, C.c_int], None),
                       ("TrainNetworkDatum", [VP, VP, VP], C.c_float), ("UpdateNetworkGpu", [VP], None),
                       ("DkAdvanceIteration", [VP], None), ("SyncNetworks", [VP, C.c_int], None),
                       ("DkLayerPull", [VP, C.c_int, C.c_int, VP, C.c_size_t], C.c_long)):
        getattr(L, fn).argtypes = at
        getattr(L, fn).restype = rt
    B = int(g["batch"])
    assert B == 2
    one = str(tmp_path / "one.cfg")
    open(one, "w").write(open(cfg).read().replace("batch=%d" % B, "batch=1"))
    sub = str(tmp_path / "sub.cfg")
    open(sub, "w").write(open(cfg).read().replace("subdivisions=1", "subdivisions=%d" % B))
    truth = np.ascontiguousarray(g["truth"])
    X = np.ascontiguousarray(x.reshape(B, -1))
    T = np.ascontiguousarray(truth.reshape(B, -1))
    STEPS = 2
    ref = netutil.DkNet(gpu, sub, wpath, train=True)
    L.DkSetMaxIter(ref.p, 1000)
    for _ in range(STEPS):
        for i in range(B):
            L.TrainNetworkDatum(ref.p, X[i:i + 1].ctypes.data, T[i:i + 1].ctypes.data)
        L.DkAdvanceIteration(ref.p)
        L.UpdateNetworkGpu(ref.p)
    nets = L.DkNetworkArrayCreate(B)
    for i in range(B):
        p = L.DkNetworkArrayAt(nets, i)
        assert L.LoadNetwork(p, one.encode(), wpath.encode(), True, False)
        L.DkSetMaxIter(p, 1000)
    costs = [L.DkTrainNetworksFlat(nets, B, X.ctypes.data, X.shape[1], T.ctypes.data, T.shape[1], B, 4) for _ in range(STEPS)]
    assert all(np.isfinite(c) and c > 0 for c in costs)
    L.SyncNetworks(nets, B)   # averages the (identical) weights and the per-replica rolling statistics

    def weights(p, i, n):
        out = np.empty(n, np.float32)
        assert L.DkLayerPull(p, i, 1, out.ctypes.data, n) == n
        return out
    p0, p1 = L.DkNetworkArrayAt(nets, 0), L.DkNetworkArrayAt(nets, 1)
    for i in range(ref.n):
        f = ref.info(i)
        if f["type"] == O.CONVOLUTIONAL:
            a, b0, b1 = weights(ref.p, i, f["nweights"]), weights(p0, i, f["nweights"]), weights(p1, i, f["nweights"])
            assert np.array_equal(b0, b1), "replicas diverged"
            util.assert_close(b0, a, "weights after %d TrainNetworks steps, layer %d" % (STEPS, i), rel=2e-5, atol_rms=2e-6)
    L.DkNetworkArrayDestroy(nets, B)
    ref.close()


def test_train_networks_collective_path_on_one_gpu(gpu, tmp_path):
    """The RCCL branch of TrainNetworks / SyncNetworks (csrc/host/multigpu.cpp; reference network_kernels.cu:398-484)
    EXECUTED on a one-GPU box: a child process (the environment selects the path before the library loads) runs two
    replicas of yolov4-tiny on device 0 with DK_DP_SHARED_DEVICE_COLLECTIVE=1 -- two concurrent host threads, each on
    its own compute stream with its own reduction scratch, the gradient bucket all-reduced in backward-order segments on
    the communication stream behind events, threaded update, SyncNetworks' per-tensor all-reduces -- with
    tests/libshim_rccl.so (host-staged sums; built from tests/shim_rccl.cpp) standing in for librccl through
    DK_RCCL_LIB.  In deterministic mode the replicas must stay bitwise identical and match ONE replica accumulating
    over subdivisions = 2 (same tolerance as the local-sum test: the two sums associate differently); the shim must
    have seen >= 2 segments per iteration, from both ranks at once.  What this does NOT show: anything about RCCL
    itself or xGMI -- multi-GPU runs are the driver's."""
    import subprocess
    import sys
    shim_src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shim_rccl.cpp")
    shim = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libshim_rccl.so")
    if not os.path.exists(shim) or os.path.getmtime(shim) < os.path.getmtime(shim_src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O2", shim_src, "-o", shim, "-lpthread"])
    g, cfg, wpath, x = train_fixture(tmp_path)
    np.save(str(tmp_path / "x.npy"), x)
    np.save(str(tmp_path / "truth.npy"), g["truth"])
    env = dict(os.environ, DK_RCCL_LIB=shim, DK_DP_SHARED_DEVICE_COLLECTIVE="1", DK_DETERMINISTIC="1", DK_DP_SEGMENTS="3",
               PYTHONPATH=os.pathsep.join([os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]))
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dp_child.py")
    r = subprocess.run([sys.executable, child, cfg, wpath, str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    print(out[-3000:])
    assert r.returncode == 0, out[-3000:]
    assert "DP-CHILD-OK" in out


def test_c4_yolov4_608_b8_train_step_vs_reference_golden(gpu, tmp_path):
    """BASELINE config C4 at its per-GPU size: one yolov4 608x608 train step at batch 8 (forward with
    batch statistics, host yolo loss, backward) against the REAL reference's step on the same inputs
    (tests/golden/train_yolov4_b8.npz: summaries only -- cost, per layer 64 strided samples + sum of
    squares of the train-mode output, yolo delta positions, per conv gradient norms).  Integer
    results (which predictors receive a loss gradient) must match exactly; fp32 activations within
    the train-mode tolerance of util.py; gradient L2 norms within 1 %."""
    name, B = "yolov4", 8
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_%s_b%d.npz" % (name, B)))
    cfg = str(tmp_path / "t.cfg")
    open(cfg, "w").write(open(netutil.cfg_path(name)).read().replace("batch=64", "batch=%d" % B).replace("subdivisions=8", "subdivisions=1"))
    wpath = str(tmp_path / "w.weights")
    netutil.synth_weights_for(gpu, name, wpath)
    L = gpu.lib()
    L.TrainNetworkDatum.argtypes = [VP, VP, VP]
    L.TrainNetworkDatum.restype = C.c_float
    L.DkLayerPull.argtypes = [VP, C.c_int, C.c_int, VP, C.c_size_t]
    L.DkLayerPull.restype = C.c_long
    L.DkSetMaxIter.argtypes = [VP, C.c_int]
    net = netutil.DkNet(gpu, cfg, wpath, train=True)
    assert net.batch == B and (net.w, net.h) == (608, 608)
    L.DkSetMaxIter(net.p, 1000)
    x = np.ascontiguousarray(synth.make_input(B, net.c, net.h, net.w, seed=12345))
    truth = np.ascontiguousarray(g["truth"])

    def pull(i, which, n):
        out = np.empty(n, np.float32)
        assert L.DkLayerPull(net.p, i, which, out.ctypes.data, n) == n
        return out
    # The step is run through the split API (DkTrainForward / DkBackwardRange / DkTrainFinish == TrainNetworkDatum,
    # test_split_train_step_equals_train_network_datum) so that the BACKWARD arithmetic of sampled layers can be
    # checked tightly at this size: before layer l's backward its incoming delta is pulled, after it the delta it left
    # (conv-output gradient), its weight / scale / bias gradients and the data gradient it wrote into layer l - 1 --
    # and the oracle's BackwardConvolutionalLayer (activation gradient, batchnorm backward, weight gradient, col2im
    # data gradient; oracle/orc_net.py backward) is evaluated on exactly those tensors of the HIP path (its input
    # activations, batch statistics and incoming delta), so kinks take the same branch and the reference's drifted
    # statistics play no part.  Layers: one of every kernel class the first-step timing chooses among at this size.
    for fn, at, rt in (("DkTrainForward", [VP, VP, VP], None), ("DkBackwardRange", [VP, C.c_int, C.c_int], None),
                       ("DkTrainFinish", [VP], C.c_float)):
        getattr(L, fn).argtypes = at
        getattr(L, fn).restype = rt
    SAMPLED = [1, 2, 6, 29, 59, 86, 106, 138]   # 3x3/s2 at 608, 1x1 at 304, 3x3 at 304 / 76, 1x1 at 38, 3x3/s2 -> 19, 3x3 at 19, head
    OL = O.lib()
    L.DkTrainForward(net.p, x.ctypes.data, truth.ctypes.data)
    hi = net.n
    checked = []
    for li in sorted(SAMPLED, reverse=True):
        f, fp = net.info(li), net.info(li - 1)
        assert f["type"] == O.CONVOLUTIONAL
        if hi > li + 1:
            L.DkBackwardRange(net.p, hi, li + 1)
        n_out = f["batch"] * f["outputs"]
        delta_in = pull(li, 6, n_out)
        L.DkBackwardRange(net.p, li + 1, li)
        hi = li
        delta_post, dw = pull(li, 6, n_out), pull(li, 7, f["nweights"])
        prev_delta = pull(li - 1, 6, fp["batch"] * fp["outputs"])
        x_in = np.ascontiguousarray(net.output(li - 1).reshape(B, f["c"], f["h"], f["w"]))
        w = pull(li, 1, f["nweights"])
        n, sp, act = f["n"], f["out_h"] * f["out_w"], f["activation"]
        raw, _ = orc_conv(x_in, w.reshape(n, f["c"], f["size"], f["size"]), np.zeros(n, np.float32), B, f["c"], f["h"], f["w"], n, f["size"],
                          f["stride_x"], f["pad"], O.LINEAR)
        raw = np.ascontiguousarray(raw.reshape(B, n, sp))
        d = delta_in.copy()
        out_hip = np.ascontiguousarray(net.output(li).ravel())
        if f["batch_normalize"]:
            mean, var, scales, biases = pull(li, 10, n), pull(li, 11, n), pull(li, 3, n), pull(li, 2, n)
            x_norm = ((raw - mean[None, :, None]) / np.sqrt(var[None, :, None] + np.float32(.000001))).astype(np.float32)
            pre = np.ascontiguousarray((x_norm * scales[None, :, None] + biases[None, :, None]).astype(np.float32))
            if act == O.MISH:
                OL.orc_gradient_array_mish(n_out, O.fptr(pre.ravel()), O.fptr(d))
            else:
                OL.orc_gradient_array(O.fptr(out_hip), n_out, act, O.fptr(d))
            md, vd, su = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
            xn = np.ascontiguousarray(x_norm.ravel())
            OL.orc_batchnorm_backward(O.fptr(d), B, n, sp, O.fptr(scales), O.fptr(raw.ravel()), O.fptr(xn), O.fptr(mean),
                                      O.fptr(var), O.fptr(md), O.fptr(vd), O.fptr(su))
            util.assert_close(pull(li, 9, n), su, "C4 layer %d scale_updates vs oracle on HIP tensors" % li, rel=2e-4,
                              atol_rms=2 * util.TRAIN_ATOL_RMS)
        else:
            OL.orc_gradient_array(O.fptr(out_hip), n_out, act, O.fptr(d))
            bu = np.zeros(n, np.float32)
            OL.orc_backward_bias(O.fptr(bu), O.fptr(d), B, n, sp)
            util.assert_close(pull(li, 8, n), bu, "C4 layer %d bias_updates vs oracle on HIP tensors" % li, rel=2e-4,
                              atol_rms=2 * util.TRAIN_ATOL_RMS)
        util.assert_close(delta_post, d, "C4 layer %d delta after activation / batchnorm backward" % li, rel=2e-4,
                          atol_rms=2 * util.TRAIN_ATOL_RMS)
        # the weight gradient and the data gradient from the HIP path's own post-batchnorm delta
        ref_dw, ref_prev = np.zeros(f["nweights"], np.float32), np.zeros(x_in.size, np.float32)
        ws = np.zeros(sp * f["size"] * f["size"] * f["c"] + 1, np.float32)
        OL.orc_conv_backward(O.fptr(x_in.ravel()), O.fptr(w), O.fptr(delta_post), O.fptr(ref_dw), O.fptr(ref_prev), O.fptr(ws),
                             B, f["c"], f["h"], f["w"], n, 1, f["size"], f["stride_x"], f["stride_y"], 1, f["pad"])
        st_w = util.assert_close(dw, ref_dw, "C4 layer %d weight_updates vs oracle on HIP tensors" % li, rel=2e-4,
                                 atol_rms=2 * util.TRAIN_ATOL_RMS)
        st_d = util.assert_close(prev_delta, ref_prev, "C4 layer %d data gradient vs oracle on HIP tensors" % li, rel=2e-4,
                                 atol_rms=2 * util.TRAIN_ATOL_RMS)
        checked.append((li, st_w["max_abs_over_rms"], st_d["max_abs_over_rms"]))
    if hi > 0:
        L.DkBackwardRange(net.p, hi, 0)
    cost = L.DkTrainFinish(net.p)
    print("C4 backward of sampled layers vs the oracle on the HIP path's own tensors (layer, wgrad max|d|/rms, dgrad max|d|/rms): %s"
          % [(i, float("%.3g" % a), float("%.3g" % b)) for i, a, b in checked])
    assert abs(cost - float(g["cost"])) <= 2e-3 * float(g["cost"]), (cost, float(g["cost"]))
    # Two comparators per layer (64 strided samples each):
    #  (a) the oracle with the batch statistics accumulated in DOUBLE (orc_set_bn_stats_f64; an analysis
    #      variant, the HIP kernels also reduce in fp64): train-mode tolerance of util.py;
    #  (b) the reference itself, whose mean_cpu / variance_cpu add 2.96 M fp32 terms sequentially at this
    #      size: ITS outputs sit up to ref_vs_f64stats_max_over_rms (4.9e-3 x rms at layer 0, growing to
    #      0.4 x rms near the heads as 107 batch-normalised layers amplify it) away from (a) -- the
    #      fixture stores that distance per layer, and it bounds the HIP-vs-reference distance here.
    # Tolerance vs (a): util.TRAIN_ATOL_RMS (3e-4 x rms, measured on yolov4-tiny's 21 batch-normalised layers)
    # up to layer 133.  From layer 134 on (the 19x19 neck/head: 2888 samples per channel at b=8) this random-init
    # network amplifies ANY upstream perturbation 30-100x -- the fixture shows it on the reference itself,
    # whose distance from (a) jumps from 3.4e-3 x rms (layer 130) to 0.09-0.43 x rms (layers 134-160).  The
    # HIP path's fp32 summation-order differences are amplified alike (measured on MI355X: 4.1e-4 x rms at
    # layer 134, 1.03e-3 at layer 146), so there the bound is "100x closer to (a) than the reference is".
    worst_a = worst_b = worst_ratio = 0.0
    shift = g["ref_vs_f64stats_max_over_rms"]
    bad = []
    for row, row64 in zip(g["fwd_summaries"], g["fwd_summaries_f64stats"]):
        i = int(row[0])
        a = net.output(i).ravel()
        idx = np.linspace(0, a.size - 1, 64).astype(np.int64)
        got = a[idx].astype(np.float64)
        rms = np.sqrt(row64[2] / a.size)
        atol = max(util.TRAIN_ATOL_RMS, 1e-2 * float(shift[i]))
        d64 = np.abs(got - row64[3:])
        lim64 = util.REL * np.abs(row64[3:]) + atol * rms
        dref = np.abs(got - row[3:])
        limref = util.REL * np.abs(row[3:]) + (atol + 1.5 * shift[i]) * rms
        r64 = float((d64 / np.maximum(lim64, 1e-300)).max())
        rref = float((dref / np.maximum(limref, 1e-300)).max())
        if r64 > 1 or rref > 1:
            bad.append((i, r64, rref))
        if rms > 0:
            worst_a, worst_b = max(worst_a, float(d64.max() / rms)), max(worst_b, float(dref.max() / rms))
            if shift[i] > 0:
                worst_ratio = max(worst_ratio, float(d64.max() / rms / shift[i]))
    print("C4 train forward: HIP distance / reference distance from the fp64-statistics oracle, worst layer: %.3g" % worst_ratio)
    assert not bad, "train forward layers off (layer, x tolerance vs fp64-statistics oracle, x tolerance vs reference): %s" % bad[:8]
    print("C4 train forward, 162 layers: worst |d|/rms %.3g vs the fp64-statistics oracle, %.3g vs the reference "
          "(whose own distance from that oracle reaches %.3g)" % (worst_a, worst_b, float(shift.max())))
    for i in range(net.n):
        f = net.info(i)
        if f["type"] == O.YOLO:
            d = pull(i, 6, f["batch"] * f["outputs"])
            ref_idx = g["yolo_%d_delta_idx" % i]
            mine = np.flatnonzero(d)
            # which predictors receive a loss gradient: identical but for ignore-threshold decisions that
            # the reference's drifted heads (see above) can flip
            sym = np.setxor1d(mine, ref_idx).size
            assert sym <= 2e-3 * ref_idx.size, "yolo %d: %d of %d gradient positions differ" % (i, sym, ref_idx.size)
    # Gradient L2 norms vs the reference's.  The reference back-propagates from ITS forward pass, which
    # (see above) sits up to 0.43 x rms away from the exact-statistics forward at the heads, so its yolo deltas
    # and every gradient behind them carry that drift; the norms are a loose checksum here (the exact
    # backward check is test_tiny_train_step: oracle backward on the HIP path's own activations).
    devs = []
    for row in g["grad_summaries"]:
        i, which = int(row[0]), int(row[1])
        if which not in (7, 9):
            continue
        f = net.info(i)
        n = f["nweights"] if which == 7 else f["n"]
        a = pull(i, which, n).astype(np.float64)
        devs.append((abs(np.sqrt((a * a).sum()) / np.sqrt(row[3]) - 1), i, which))
    devs.sort(reverse=True)
    worst_norm = devs[0][0]
    print("C4 gradient norms vs the reference, largest deviations (dev, layer, 7=weights/9=biases): %s; median %.3g" % (
        [(round(float(d), 4), i, w) for d, i, w in devs[:6]], float(np.median([d for d, _, _ in devs]))))
    # (the exact backward check at this size is the per-layer one above; the norms against the reference's own,
    # drifted step are a sanity bound on the whole: the median, and no gradient tensor off by a factor)
    assert float(np.median([d for d, _, _ in devs])) < 3e-2 and worst_norm < 0.5, devs[:6]
    print("C4 gradients vs the reference: worst L2-norm deviation %.3g" % worst_norm)
    net.close()


def test_adam_update_vs_oracle(gpu, tmp_path):
    """adam=1 (src/blas_kernels.cu:99-134, convolutional_kernels.cu:884-898; the reference has no CPU adam, so the
    oracle restates its GPU launch sequence: "parity unpinned", see oracle/orc_ops.c).  (1) the fused kernel on
    random tensors over three iterations vs the oracle; (2) a yolov4-tiny net with adam=1: one train step's
    weights == the oracle's update applied to the gradients the step produced."""
    L = gpu.lib()
    L.dk_adam_update.argtypes = [VP, VP, VP, VP] + [C.c_float] * 5 + [C.c_size_t, C.c_int, C.c_int, VP]
    L.dk_adam_update.restype = C.c_int
    OL = O.lib()
    OL.orc_adam_update.argtypes = [C.POINTER(C.c_float)] * 4 + [C.c_float] * 5 + [C.c_int] * 3
    OL.orc_adam_update.restype = None
    rng = np.random.default_rng(7)
    n = 100003
    w = rng.normal(0, .1, n).astype(np.float32)
    m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    dw, dm, dv = gpu.DeviceArray(w), gpu.DeviceArray(m), gpu.DeviceArray(v)
    for t in (1, 2, 3):
        d = rng.normal(0, 1, n).astype(np.float32)
        dd = gpu.DeviceArray(d)
        assert L.dk_adam_update(dw.ptr, dd.ptr, dm.ptr, dv.ptr, .9, .999, 1e-6, 5e-4, 1e-3, n, 64, t, None) == 0
        OL.orc_adam_update(O.fptr(w), O.fptr(d), O.fptr(m), O.fptr(v), .9, .999, 1e-6, 5e-4, 1e-3, n, 64, t)
        assert not dd.numpy().any(), "the gradient buffer must be zeroed by the update"
        util.assert_close(dm.numpy(), m, "adam m, t=%d" % t, rel=1e-6, atol_rms=1e-7)
        util.assert_close(dv.numpy(), v, "adam v, t=%d" % t, rel=1e-6, atol_rms=1e-7)
        util.assert_close(dw.numpy(), w, "adam w, t=%d" % t, rel=2e-6, atol_rms=1e-7)
    # ---- net level
    g, cfg, wpath, x = train_fixture(tmp_path)
    cfg2 = str(tmp_path / "adam.cfg")
    open(cfg2, "w").write(open(cfg).read().replace("momentum=0.9", "momentum=0.9\nadam=1\nB1=0.9\nB2=0.999\neps=0.000001"))
    for fn, at, rt in (("TrainNetworkDatum", [VP, VP, VP], C.c_float), ("UpdateNetworkGpu", [VP], None),
                       ("DkAdvanceIteration", [VP], None), ("DkSetMaxIter", [VP, C.c_int], None),
                       ("GetCurrLr", [VP], C.c_float),
                       ("DkLayerPull", [VP, C.c_int, C.c_int, VP, C.c_size_t], C.c_long)):
        getattr(L, fn).argtypes = at
        getattr(L, fn).restype = rt
    net = netutil.DkNet(gpu, cfg2, wpath, train=True)
    L.DkSetMaxIter(net.p, 1000)
    truth = np.ascontiguousarray(g["truth"])
    xin = np.ascontiguousarray(x)
    L.TrainNetworkDatum(net.p, xin.ctypes.data, truth.ctypes.data)

    def pull(i, which, cnt):
        out = np.empty(cnt, np.float32)
        assert L.DkLayerPull(net.p, i, which, out.ctypes.data, cnt) == cnt
        return out
    convs = [i for i in range(net.n) if net.info(i)["type"] == O.CONVOLUTIONAL]
    before = {i: (pull(i, 1, net.info(i)["nweights"]), pull(i, 7, net.info(i)["nweights"])) for i in convs[:6]}
    L.DkAdvanceIteration(net.p)       # curr_iter = 1 -> t = 1
    lr = L.GetCurrLr(net.p)
    L.UpdateNetworkGpu(net.p)
    B = net.batch                     # subdivisions = 1 in the fixture cfg
    for i, (w0, g0) in before.items():
        w1 = w0.copy(); d = g0.copy(); m = np.zeros_like(w0); v = np.zeros_like(w0)
        OL.orc_adam_update(O.fptr(w1), O.fptr(d), O.fptr(m), O.fptr(v), .9, .999, 1e-6, 5e-4, lr, w1.size, B, 1)
        util.assert_close(pull(i, 1, w0.size), w1, "adam net step, conv %d weights" % i, rel=2e-6, atol_rms=1e-7)
        assert not pull(i, 7, w0.size).any()
    net.close()


def test_clip_clamps_weights_after_the_update(gpu, tmp_path):
    """`clip=` (parser.cpp:1361; applied by UpdateConvolutionalLayerGpu, convolutional_kernels.cu:919-920, through
    constrain_ongpu blas_kernels.cu:450): dk_constrain bitwise vs the oracle's constrain_cpu restatement, and one
    deterministic yolov4-tiny train step with clip=0.02 on two layers == clamp(the same step without clip), bitwise,
    while unclipped layers are untouched.  (The reference's CPU update never clips -- this line follows its GPU path;
    the arithmetic itself is fminf / fmaxf.)"""
    L = gpu.lib()
    L.dk_constrain.argtypes = [C.c_size_t, C.c_float, VP, VP]
    rng = np.random.default_rng(0)
    v = (rng.normal(0, 0.05, 100003)).astype(np.float32)
    v[:3] = (0.02, -0.02, np.float32(0.020000001))
    dv = gpu.DeviceArray(v)
    assert L.dk_constrain(v.size, 0.02, dv.ptr, None) == 0
    want = v.copy()
    O.lib().orc_constrain(want.size, O.F(0.02), O.fptr(want))
    assert np.array_equal(dv.numpy(), want) and np.abs(want).max() == np.float32(0.02)

    g, cfg, wpath, x = train_fixture(tmp_path)
    for fn, at, rt in (("TrainNetworkDatum", [VP, VP, VP], C.c_float), ("DkSetMaxIter", [VP, C.c_int], None),
                       ("DkSetDeterministic", [C.c_int], None), ("UpdateNetworkGpu", [VP], None),
                       ("DkAdvanceIteration", [VP], None),
                       ("DkLayerPull", [VP, C.c_int, C.c_int, VP, C.c_size_t], C.c_long)):
        getattr(L, fn).argtypes = at
        getattr(L, fn).restype = rt
    txt = open(cfg).read()
    secs = txt.split("[convolutional]")
    clipped = (1, 4)   # conv sections (1-based pieces of the split): conv layers 0 and 3
    for k in clipped:
        secs[k] = "\nclip=0.02" + secs[k]
    ccfg = str(tmp_path / "clip.cfg")
    open(ccfg, "w").write("[convolutional]".join(secs))
    onet = O.parse_cfg(ccfg)
    conv_ids = [i for i, l in enumerate(onet.layers) if l.type == O.CONVOLUTIONAL]
    clip_layers = [conv_ids[k - 1] for k in clipped]
    assert [onet.layers[i].clip for i in clip_layers] == [0.02, 0.02]
    truth = np.ascontiguousarray(g["truth"])
    xin = np.ascontiguousarray(x)

    def run(c):
        net = netutil.DkNet(gpu, c, wpath, train=True)
        L.DkSetMaxIter(net.p, 1000)
        cost = L.TrainNetworkDatum(net.p, xin.ctypes.data, truth.ctypes.data)
        L.DkAdvanceIteration(net.p)     # TrainNetwork (network.cpp:210-236): curr_iter++ and then the update
        L.UpdateNetworkGpu(net.p)
        ws = {}
        for i in conv_ids[:6]:
            n = net.info(i)["nweights"]
            ws[i] = np.empty(n, np.float32)
            assert L.DkLayerPull(net.p, i, 1, ws[i].ctypes.data, n) == n
        net.close()
        return cost, ws
    L.DkSetDeterministic(1)
    try:
        c0, w0 = run(cfg)
        c1, w1 = run(ccfg)
    finally:
        L.DkSetDeterministic(0)
    assert c0 == c1
    for i in conv_ids[:6]:
        if i in clip_layers:
            assert np.abs(w0[i]).max() > 0.02, "the fixture's weights never reach the clip value"
            assert np.array_equal(w1[i], np.clip(w0[i], np.float32(-0.02), np.float32(0.02))), "layer %d" % i
        else:
            assert np.array_equal(w1[i], w0[i]), "unclipped layer %d changed" % i


WGRAD3_STRIDE2_SHAPES = [
    # batch, c, h, w, n: 3x3 / stride 2 / pad 1 on even maps (yolov4's down-sampling layers are 64->128 304^2 ... 512->1024 38^2)
    (2, 32, 16, 24, 128),      # one segment of 12 pixels, 16-byte pieces
    (1, 64, 38, 38, 128),      # 19 output columns, 8-byte pieces, two channel tiles
    (2, 32, 12, 88, 256),      # 44 output columns: three segments, the last 4 wide; two filter tiles
    (1, 32, 6, 50, 128),       # width = 2 mod 4, 25 output columns: 4-byte pieces (a delta row is odd)
    (1, 32, 8, 44, 128),       # 22 output columns: 8-byte pieces
]


@pytest.mark.parametrize("case", WGRAD3_STRIDE2_SHAPES)
def test_row_staged_weight_gradient_stride2_vs_oracle(gpu, case):
    """conv_wgrad3_f32<20, VW, 2> (conv_wgrad.hip): the weight gradient of the 3x3 / stride-2 layers, against the oracle's
    BackwardConvolutionalLayer (src/convolutional_layer.cpp:1345-1356: im2col_cpu_ext + gemm(0,1)) accumulating into a
    non-zero dW, with 16 / 8 / 4-byte pieces, atomics and the ordered reduction, both pixel splits; the gather kernel on the
    same layer as a cross-check; each launch is checked to have run the kernel it names."""
    batch, c, h, w, n = case
    L, G = O.lib(), bind(gpu.lib())
    G.dk_train_force.argtypes = [C.c_int, C.c_int]
    G.dk_set_deterministic.argtypes = [C.c_int]
    G.dk_set_deterministic.restype = None
    rng = np.random.default_rng(util.seed_of(case))
    oh, ow = h // 2, w // 2
    x = rng.uniform(-1, 1, (batch, c, h, w)).astype(np.float32)
    wt = (rng.uniform(-1, 1, (n, c, 3, 3)) * 0.2).astype(np.float32)
    delta = rng.uniform(-1, 1, (batch, n, oh, ow)).astype(np.float32)
    dw0 = rng.uniform(-1, 1, wt.shape).astype(np.float32)
    ref_dw, ref_prev = dw0.copy(), np.zeros_like(x)
    ws = np.zeros(oh * ow * 9 * c + 1, np.float32)
    L.orc_conv_backward(O.fptr(x), O.fptr(wt), O.fptr(delta), O.fptr(ref_dw), O.fptr(ref_prev), O.fptr(ws),
                        batch, c, h, w, n, 1, 3, 2, 2, 1, 1)
    d = gpu.DkConvDesc(batch, c, h, w, n, 1, 3, 2, 2, 1, 1, O.LINEAR)
    dx, dd = gpu.DeviceArray(x), gpu.DeviceArray(delta)
    G.dk_profile_enable(1)
    try:
        for tile, avec, det in [(4, -1, 0), (4, -1, 1), (5, -1, 0), (4, 0, 0), (-1, -1, 0), (0, -1, 0)]:
            G.dk_train_force(0, tile)
            G.dk_train_force(2, avec)
            G.dk_set_deterministic(det)
            ddw = gpu.DeviceArray(dw0)
            assert G.dk_conv_backward_weights(C.byref(d), dx.ptr, dd.ptr, ddw.ptr, None) == 0
            ran = [k for k in _ran_kernels(gpu.lib()) if k.startswith("conv_wgrad")]
            if tile == 0:
                assert len(ran) == 1 and ran[0].startswith("conv_wgrad_f32<"), ran
            else:
                vw = 1 if avec == 0 else 4 if (w % 4 == 0 and ow % 4 == 0) else 2 if ow % 2 == 0 else 1
                assert ran == ["conv_wgrad3_f32<20, %d, 2>" % vw], ran
            util.assert_close(ddw.numpy().reshape(wt.shape), ref_dw, "wgrad3 stride 2 %s tile %d avec %d det %d" % (case, tile, avec, det))
            ddw.free()
    finally:
        for k in range(3):
            G.dk_train_force(k, -1)
        G.dk_set_deterministic(-1)
        G.dk_profile_enable(0)

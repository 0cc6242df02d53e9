"""Per-op parity: HIP kernels (through the C-ABI) vs the CPU oracle on the same
seeded inputs.  fp32 activations within util.REL (1e-4, definition in util.py);
integer outputs (maxpool argmax indexes) bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest

import util
from oracle import orc_net as O

pytestmark = pytest.mark.gpu

F = C.c_float


def orc_conv(x, w, bias, batch, c, h, wd, n, size, stride, pad, act, groups=1, dilation=1, stride_y=None):
    L = O.lib()
    sy = stride if stride_y is None else stride_y
    keff = dilation * (size - 1) + 1
    oh = (h + 2 * pad * dilation - keff) // sy + 1
    ow = (wd + 2 * pad * dilation - keff) // stride + 1
    out = np.zeros((batch, n, oh, ow), np.float32)
    ws = np.zeros(max(1, oh * ow * size * size * (c // groups)), np.float32)
    act_in = np.zeros_like(out)
    L.orc_conv_forward_fused(O.fptr(x), O.fptr(w), O.fptr(bias), O.fptr(out), O.fptr(ws),
                             O.fptr(act_in), batch, c, h, wd, n, groups, size, stride, sy,
                             dilation, pad, act)
    return out, act_in


CONV_CASES = [
    # batch, c, h, w, n, size, stride, pad, act, groups, dilation
    (2, 3, 32, 32, 32, 3, 1, 1, "MISH", 1, 1),        # first layer shape class (K=27)
    (2, 3, 33, 29, 32, 3, 2, 1, "LEAKY", 1, 1),       # stride 2, odd, non-square
    (3, 32, 19, 19, 64, 3, 1, 1, "LEAKY", 1, 1),      # 19x19 (W not a multiple of anything)
    (2, 64, 19, 19, 255, 1, 1, 0, "LINEAR", 1, 1),    # head: M=255 (ragged M), 1x1
    (1, 128, 13, 13, 256, 1, 1, 0, "MISH", 1, 1),     # 1x1, N=169 < tile
    (2, 64, 26, 26, 128, 3, 2, 1, "MISH", 1, 1),      # downsampler
    (1, 16, 8, 8, 16, 1, 1, 0, "LOGISTIC", 1, 1),     # tiny everything (N=64, M=16)
    (2, 32, 20, 20, 64, 3, 1, 1, "LEAKY", 2, 1),      # groups=2
    (1, 8, 17, 17, 24, 3, 1, 1, "RELU", 1, 2),        # dilation 2 (cfg pad=1 -> l->pad=1, effective 2)
    (2, 16, 14, 14, 32, 5, 1, 2, "LINEAR", 1, 1),     # 5x5
    (1, 512, 19, 19, 1024, 3, 1, 1, "LEAKY", 1, 1),   # yolov4 neck shape, K=4608
    (5, 4, 9, 7, 10, 3, 1, 1, "LINEAR", 1, 1),        # N spans image boundaries inside one tile
    (2, 16, 13, 13, 24, 3, 1, 1, "SWISH", 1, 1),      # sibling-cfg activations through the fused epilogue
    (2, 8, 10, 10, 16, 1, 1, 0, "RELU6", 1, 1),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_forward_vs_oracle(gpu, case):
    batch, c, h, w, n, size, stride, pad, actname, groups, dil = case
    act = getattr(O, actname) if hasattr(O, actname) else {"RELU": 1, "RELU6": 2, "SWISH": 16}[actname]
    rng = np.random.default_rng(util.seed_of(case))
    x = rng.uniform(-1, 1, (batch, c, h, w)).astype(np.float32)
    fan = size * size * c // groups
    wt = (rng.uniform(-1, 1, (n, c // groups, size, size)) * np.sqrt(2.0 / fan)).astype(np.float32)
    bias = rng.uniform(-.5, .5, n).astype(np.float32)
    ref, _ = orc_conv(x, wt, bias, batch, c, h, w, n, size, stride, pad, act, groups, dil)
    ref_in, _ = orc_conv(x, wt, bias, batch, c, h, w, n, size, stride, pad, O.LINEAR, groups, dil)
    ncfg = gpu.lib().dk_conv_force_config(-1)
    for cfg in [-1] + list(range(ncfg)):
        gpu.lib().dk_conv_force_config(cfg)
        y, act_in = gpu.conv_forward(x, wt, bias, batch, c, h, w, n, size, stride, pad, act,
                                     groups=groups, dilation=dil, want_act_in=True)
        util.assert_close(y, ref, "conv cfg %d %s" % (cfg, case))
        util.assert_close(act_in, ref_in, "conv pre-activation cfg %d" % cfg)
    gpu.lib().dk_conv_force_config(-1)


DIRECT_CASES = [
    # batch, c, h, w, n, act : 3x3 / stride 1 / pad 1 layers for the patch-in-LDS kernel
    (3, 8, 19, 19, 64, "MISH"),       # pitch 24; 361-pixel images: tiles straddle images
    (2, 12, 30, 22, 32, "LEAKY"),     # pitch 24, non-square
    (2, 8, 38, 38, 40, "MISH"),       # pitch 40, ragged M
    (1, 8, 76, 76, 130, "LEAKY"),     # pitch 80, ragged M > one tile
    (1, 4, 152, 152, 32, "LINEAR"),   # pitch 160
    (2, 16, 64, 64, 64, "MISH"),      # pitch 80, W a power of two
    (1, 4, 26, 26, 16, "LEAKY"),      # pitch 40, N = 676 (ragged last tile)
    (3, 8, 13, 13, 32, "LEAKY"),      # 416-net sizes: pitch 24 with 169-pixel images
    (1, 8, 52, 52, 32, "MISH"),       # pitch 80, 8 patch rows
]


@pytest.mark.parametrize("case", DIRECT_CASES)
def test_conv_direct3x3_vs_oracle(gpu, case):
    """Every tile configuration (gather and direct) on 3x3/s1/p1 layers, with bias, the fused
    activation and a residual (the straight-line epilogue), against the CPU oracle."""
    batch, c, h, w, n, actname = case
    act = getattr(O, actname)
    rng = np.random.default_rng(util.seed_of(case))
    x = rng.uniform(-1, 1, (batch, c, h, w)).astype(np.float32)
    wt = (rng.uniform(-1, 1, (n, c, 3, 3)) * np.sqrt(2.0 / (9 * c))).astype(np.float32)
    bias = rng.uniform(-.5, .5, n).astype(np.float32)
    res = rng.uniform(-1, 1, (batch, n, h, w)).astype(np.float32)
    ref, _ = orc_conv(x, wt, bias, batch, c, h, w, n, 3, 1, 1, act)
    L = gpu.lib()
    ncfg = L.dk_conv_force_config(-1)
    names = [L.dk_conv_config_name(i).decode() for i in range(ncfg)]
    assert any(nm.startswith("direct3x3") for nm in names)
    for cfg in range(ncfg):
        L.dk_conv_force_config(cfg)
        y = gpu.conv_forward(x, wt, bias, batch, c, h, w, n, 3, 1, 1, act)
        util.assert_close(y, ref, "conv %s %s" % (names[cfg], case))
        y = gpu.conv_forward(x, wt, bias, batch, c, h, w, n, 3, 1, 1, act, residual=res)
        util.assert_close(y, ref + res, "conv + residual %s %s" % (names[cfg], case))
    L.dk_conv_force_config(-1)


@pytest.mark.parametrize("case", [(48, 32, 608, 608, 8, 2), (364, 64, 152, 152, 8, 1)])
def test_conv_batch_chunking_beyond_2gib(gpu, case):
    """Maximum sizes: the kernels address one launch with 32-bit byte offsets, so inputs of 2^29
    elements or more are processed in batch chunks (gather kernel: stride 2; patch-in-LDS kernel:
    stride 1).  Size-independent check: the batch repeats 3 distinct images, so every item must
    equal the item of a 3-image run through the same kernels, bit for bit."""
    B, c, h, w, n, stride = case
    assert B * c * h * w >= 2 ** 29
    rng = np.random.default_rng(5)
    x3 = rng.uniform(-1, 1, (3, c, h, w)).astype(np.float32)
    wt = (rng.uniform(-1, 1, (n, c, 3, 3)) * 0.06).astype(np.float32)
    bias = rng.uniform(-.5, .5, n).astype(np.float32)
    y3 = gpu.conv_forward(x3, wt, bias, 3, c, h, w, n, 3, stride, 1, O.LEAKY)
    ref3, _ = orc_conv(x3[:1], wt, bias, 1, c, h, w, n, 3, stride, 1, O.LEAKY)
    util.assert_close(y3[:1], ref3, "3-image run vs oracle")
    x = np.tile(x3, (B // 3 + 1, 1, 1, 1))[:B]
    y = gpu.conv_forward(x, wt, bias, B, c, h, w, n, 3, stride, 1, O.LEAKY)
    del x
    for b in range(B):
        assert np.array_equal(y[b], y3[b % 3]), "item %d of the chunked batch differs" % b


def test_conv_identity_asymmetric(gpu):
    """A = I check with an asymmetric B: catches a transposed C/D fragment map."""
    c = n = 64
    h, w = 8, 16
    x = np.arange(c * h * w, dtype=np.float32).reshape(1, c, h, w) % 251
    wt = np.eye(c, dtype=np.float32).reshape(n, c, 1, 1)
    y = gpu.conv_forward(x, wt, np.zeros(n, np.float32), 1, c, h, w, n, 1, 1, 0, O.LINEAR)
    assert np.array_equal(y, x)


def test_conv_residual_and_nobias(gpu):
    rng = np.random.default_rng(7)
    batch, c, h, w, n = 2, 32, 19, 19, 32
    x = rng.uniform(-1, 1, (batch, c, h, w)).astype(np.float32)
    wt = (rng.uniform(-1, 1, (n, c, 3, 3)) * 0.08).astype(np.float32)
    res = rng.uniform(-1, 1, (batch, n, h, w)).astype(np.float32)
    ref, _ = orc_conv(x, wt, np.zeros(n, np.float32), batch, c, h, w, n, 3, 1, 1, O.LEAKY)
    y = gpu.conv_forward(x, wt, None, batch, c, h, w, n, 3, 1, 1, O.LEAKY, residual=res)
    util.assert_close(y, ref + res, "conv + residual")


@pytest.mark.parametrize("shape", [(2, 8, 26, 26, 2, 2, 1), (2, 6, 19, 19, 5, 1, 4), (1, 4, 19, 19, 9, 1, 8),
                                   (3, 5, 19, 19, 13, 1, 12), (1, 3, 13, 11, 3, 2, 2), (1, 2, 7, 7, 2, 2, 0)])
def test_maxpool_vs_oracle(gpu, shape):
    batch, c, h, w, size, stride, pad = shape
    rng = np.random.default_rng(1)
    # quantised values -> many ties: exercises "first maximum wins"
    x = np.round(rng.uniform(-4, 4, (batch, c, h, w))).astype(np.float32)
    ow, oh = (w + pad - size) // stride + 1, (h + pad - size) // stride + 1
    ref = np.zeros((batch, c, oh, ow), np.float32)
    ridx = np.zeros((batch, c, oh, ow), np.int32)
    O.lib().orc_maxpool_forward(O.fptr(x), O.fptr(ref), O.iptr(ridx), batch, c, h, w, size, stride, stride, pad)
    dx = gpu.DeviceArray(x)
    dy = gpu.DeviceArray(n=ref.size)
    di = gpu.DeviceArray(n=ref.size, dtype=np.int32)
    assert gpu.lib().dk_maxpool_forward(dx.ptr, dy.ptr, di.ptr, batch, c, h, w, size, stride, stride, pad, None) == 0
    assert np.array_equal(dy.numpy().reshape(ref.shape), ref)
    assert np.array_equal(di.numpy().reshape(ref.shape), ridx)


def test_route_shortcut_upsample_yolo(gpu):
    L, G = O.lib(), gpu.lib()
    rng = np.random.default_rng(3)
    batch = 3
    # route: two sources, second with groups=2 group_id=1 semantics tested separately
    a = rng.uniform(-1, 1, (batch, 6 * 5 * 5)).astype(np.float32)
    b = rng.uniform(-1, 1, (batch, 4 * 5 * 5)).astype(np.float32)
    for groups, gid in ((1, 0), (2, 1)):
        outputs = (a.shape[1] + b.shape[1]) // groups
        ref = np.zeros((batch, outputs), np.float32)
        L.orc_route_copy(O.fptr(a), a.shape[1], groups, gid, batch, O.fptr(ref), outputs, 0)
        L.orc_route_copy(O.fptr(b), b.shape[1], groups, gid, batch, O.fptr(ref), outputs, a.shape[1] // groups)
        da, db, do = gpu.DeviceArray(a), gpu.DeviceArray(b), gpu.DeviceArray(n=ref.size)
        assert G.dk_route_copy(da.ptr, a.shape[1], groups, gid, batch, do.ptr, outputs, 0, None) == 0
        assert G.dk_route_copy(db.ptr, b.shape[1], groups, gid, batch, do.ptr, outputs, a.shape[1] // groups, None) == 0
        assert np.array_equal(do.numpy().reshape(ref.shape), ref)
    # shortcut (+ leaky)
    for n in (1000, 1003):
        p = rng.uniform(-1, 1, n).astype(np.float32)
        q = rng.uniform(-1, 1, n).astype(np.float32)
        for act in (O.LINEAR, O.LEAKY):
            ref = np.zeros(n, np.float32)
            L.orc_shortcut_forward(O.fptr(p), O.fptr(q), O.fptr(ref), n)
            L.orc_activate_array(O.fptr(ref), n, act)
            dp, dq, do = gpu.DeviceArray(p), gpu.DeviceArray(q), gpu.DeviceArray(n=n)
            assert G.dk_shortcut_forward(dp.ptr, dq.ptr, do.ptr, n, act, None) == 0
            assert np.array_equal(do.numpy(), ref)
    # upsample x2 with scale
    w, h, c = 7, 5, 3
    u = rng.uniform(-1, 1, (batch, c, h, w)).astype(np.float32)
    ref = np.zeros((batch, c, h * 2, w * 2), np.float32)
    L.orc_upsample_forward(O.fptr(u), w, h, c, batch, 2, F(1.0), O.fptr(ref))
    du, do = gpu.DeviceArray(u), gpu.DeviceArray(n=ref.size)
    assert G.dk_upsample_forward(du.ptr, w, h, c, batch, 2, 1.0, do.ptr, None) == 0
    assert np.array_equal(do.numpy().reshape(ref.shape), ref)
    # yolo decode
    lw, lh, na, classes = 13, 11, 3, 80
    t = rng.uniform(-6, 6, (batch, na * (5 + classes) * lw * lh)).astype(np.float32)
    for sxy in (1.0, 1.05, 2.0):
        ref = np.zeros_like(t)
        L.orc_yolo_forward(O.fptr(t), O.fptr(ref), batch, lw, lh, na, classes, F(sxy))
        dt, do = gpu.DeviceArray(t), gpu.DeviceArray(n=t.size)
        assert G.dk_yolo_forward(dt.ptr, do.ptr, batch, lw, lh, na, classes, sxy, None) == 0
        util.assert_close(do.numpy().reshape(ref.shape), ref, "yolo decode sxy=%g" % sxy, rel=2e-6, atol_rms=1e-7)


def test_add_bias_scale_bias_bitwise(gpu):
    """dk_add_bias / dk_scale_bias (the plugin slots add_bias_gpu / scale_bias_gpu) == the oracle's add_bias /
    scale_bias (convolutional_layer.cpp:916-944) bit for bit: one float add / multiply per element, incl. odd
    plane sizes and a single-channel tensor."""
    L, G = O.lib(), gpu.lib()
    VP, i = C.c_void_p, C.c_int
    for fn in (G.dk_add_bias, G.dk_scale_bias):
        fn.argtypes = [VP, VP, i, i, i, VP]
        fn.restype = i
    rng = np.random.default_rng(11)
    for batch, n, size in ((3, 7, 19 * 19), (2, 255, 13 * 13), (1, 1, 5), (4, 32, 76 * 76)):
        x = rng.uniform(-2, 2, (batch, n, size)).astype(np.float32)
        v = rng.uniform(-1.5, 1.5, n).astype(np.float32)
        for gfn, ofn in ((G.dk_add_bias, L.orc_add_bias), (G.dk_scale_bias, L.orc_scale_bias)):
            ref = x.copy()
            ofn(O.fptr(ref), O.fptr(v), batch, n, size)
            dx, dv = gpu.DeviceArray(x), gpu.DeviceArray(v)
            assert gfn(dx.ptr, dv.ptr, batch, n, size, None) == 0
            assert np.array_equal(dx.numpy().reshape(x.shape).view(np.uint32), ref.view(np.uint32)), (batch, n, size)
    assert G.dk_add_bias(None, None, 0, 0, 0, None) == 0   # empty tensors are a no-op


def test_activations_grid(gpu):
    """leaky / mish / logistic on a fixed grid incl. the +-20 softplus thresholds."""
    L, G = O.lib(), gpu.lib()
    x = np.concatenate([np.linspace(-30, 30, 24001), [20.0, -20.0, 20.000002, -20.000002, 0.0, -0.0, 1e-30, -1e-30]]).astype(np.float32)
    for act in (O.LEAKY, O.LOGISTIC):
        ref = x.copy()
        L.orc_activate_array(O.fptr(ref), ref.size, act)
        d = gpu.DeviceArray(x)
        assert G.dk_activate_array(d.ptr, x.size, act, None) == 0
        got = d.numpy()
        if act == O.LEAKY:
            assert np.array_equal(got, ref)
        else:
            util.assert_close(got, ref, "logistic", rel=1e-6, atol_rms=1e-7)
    # the rarer activation kinds: exact for the piecewise-linear ones, libm-vs-device ulps for the rest
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "ops.npz"))
    grid = np.ascontiguousarray(gold["act_grid"])
    for name, act in (("relu6", 2), ("relie", 3), ("ramp", 5), ("tanh", 6), ("plse", 7), ("elu", 9), ("loggy", 10),
                      ("hardtan", 12), ("lhtan", 13), ("selu", 14), ("gelu", 15), ("swish", 16)):
        d = gpu.DeviceArray(grid)
        assert G.dk_activate_array(d.ptr, grid.size, act, None) == 0
        got, ref = d.numpy(), gold["act_" + name]      # reference's own values
        if name in ("relu6", "relie", "ramp", "plse", "hardtan", "lhtan"):
            assert np.array_equal(got, ref), name
        else:
            assert np.all(np.abs(got - ref) <= 1e-5 * np.abs(ref) + 3e-7), (name, np.abs(got - ref).max())
    ref = np.zeros_like(x)
    ain = np.zeros_like(x)
    L.orc_activate_array_mish(O.fptr(x), x.size, O.fptr(ain), O.fptr(ref))
    d, da, do = gpu.DeviceArray(x), gpu.DeviceArray(n=x.size), gpu.DeviceArray(n=x.size)
    assert G.dk_activate_array_mish(d.ptr, x.size, da.ptr, do.ptr, None) == 0
    assert np.array_equal(da.numpy(), x)
    got = do.numpy()
    # The reference's mish formula logf(expf(x)+1) cancels catastrophically for
    # x in [-16,-4]: a 1-ulp difference between glibc's and the device's expf can
    # flip the rounding of (1+e^x), i.e. move softplus by 2^-23 and mish by up to
    # |x| * 1.2e-7 ~ 2e-6 absolute.  Bar: 1e-4 relative + 2.5e-6 absolute.
    assert np.all(np.abs(got - ref) <= 1e-4 * np.abs(ref) + 2.5e-6)


def test_rare_activation_gradients_grid(gpu):
    """dk_gradient_array for the 12 rarer activations and swish vs the reference's own gradient_array
    values on the grid (tests/golden/ops_grad.npz): exact for the piecewise ones, libm-vs-device ulps
    for the transcendental ones."""
    G = gpu.lib()
    G.dk_gradient_array.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "ops.npz"))
    gg = np.load(os.path.join(os.path.dirname(__file__), "golden", "ops_grad.npz"))
    grid = np.ascontiguousarray(gold["act_grid"])
    for name, act in (("relu6", 2), ("relie", 3), ("ramp", 5), ("tanh", 6), ("plse", 7), ("elu", 9), ("loggy", 10),
                      ("hardtan", 12), ("lhtan", 13), ("selu", 14), ("gelu", 15), ("relu", 1), ("swish", 16)):
        y = gpu.DeviceArray(np.ascontiguousarray(gold["act_" + name]))
        pre = gpu.DeviceArray(grid)
        d = gpu.DeviceArray(np.ones_like(grid))
        assert G.dk_gradient_array(y.ptr, pre.ptr, d.ptr, grid.size, act, None) == 0
        got, ref = d.numpy(), gg["grad_" + name]
        if name in ("relu6", "relie", "ramp", "plse", "hardtan", "lhtan", "relu", "tanh", "loggy", "elu", "selu"):
            assert np.array_equal(got, ref), name
        else:
            assert np.all(np.abs(got - ref) <= 2e-5 * np.abs(ref) + 1e-6), (name, np.abs(got - ref).max())


DMA1X1_CASES = [
    # batch, c, h, w, n, act : 1x1 / stride 1 layers the LDS-DMA ring kernel takes (c % 32 == 0, h*w % 4 == 0)
    (2, 64, 16, 16, 64, "MISH"),      # N = 512: full tiles
    (3, 32, 12, 12, 255, "LINEAR"),   # one K stage, ragged M (head), ragged N (432)
    (2, 128, 38, 38, 256, "LEAKY"),   # 1444-pixel images: tiles straddle images
    (1, 96, 20, 20, 40, "MISH"),      # three K stages, M < one tile
    (2, 256, 8, 8, 128, "LEAKY"),     # eight K stages: the ring wraps
    (1, 64, 152, 152, 32, "MISH"),    # M = 32
]


@pytest.mark.parametrize("case", DMA1X1_CASES)
def test_conv_dma1x1_vs_oracle(gpu, case):
    """LDS-DMA ring GEMM (conv1x1_dma.hip), every compiled shape: vs the CPU oracle, with a residual
    (straight-line epilogue), and BITWISE equal to the gather kernel (same k order, same MFMA chain)."""
    batch, c, h, w, n, actname = case
    act = getattr(O, actname)
    rng = np.random.default_rng(util.seed_of(case))
    x = rng.uniform(-1, 1, (batch, c, h, w)).astype(np.float32)
    wt = (rng.uniform(-1, 1, (n, c, 1, 1)) * np.sqrt(2.0 / c)).astype(np.float32)
    bias = rng.uniform(-.5, .5, n).astype(np.float32)
    res = rng.uniform(-1, 1, (batch, n, h, w)).astype(np.float32)
    ref, _ = orc_conv(x, wt, bias, batch, c, h, w, n, 1, 1, 0, act)
    L = gpu.lib()
    ncfg = L.dk_conv_force_config(-1)
    names = [L.dk_conv_config_name(i).decode() for i in range(ncfg)]
    dma = [i for i, nm in enumerate(names) if nm.startswith("dma1x1")]
    assert dma
    L.dk_conv_force_config(3)   # gather 64x64x16
    y_gather = gpu.conv_forward(x, wt, bias, batch, c, h, w, n, 1, 1, 0, act)
    L.dk_profile_enable(1)
    for cfg in dma:
        L.dk_conv_force_config(cfg)
        y = gpu.conv_forward(x, wt, bias, batch, c, h, w, n, 1, 1, 0, act)
        util.assert_close(y, ref, "conv %s %s" % (names[cfg], case))
        assert np.array_equal(y, y_gather), "dma1x1 %s differs bitwise from the gather kernel" % names[cfg]
        y = gpu.conv_forward(x, wt, bias, batch, c, h, w, n, 1, 1, 0, act, residual=res)
        util.assert_close(y, ref + res, "conv + residual %s %s" % (names[cfg], case))
    # the DMA kernel really ran (no silent fallback to the gather)
    out = (C.c_double * (3 * 256))()
    L.dk_profile_read(out, 256)
    L.dk_profile_enable(0)
    for cfg in dma:
        assert out[cfg * 4 * 3] == 2, "config %s did not launch" % names[cfg]
    L.dk_conv_force_config(-1)


WINO_CASES = [
    # batch, c, h, w, n, act, residual
    (2, 64, 38, 38, 64, "MISH", True),      # even map, full tiles
    (3, 32, 19, 19, 128, "LEAKY", False),   # odd map: ragged tile row / column, b32 store path
    (1, 8, 13, 13, 64, "LINEAR", True),     # one stage (C = 8), odd, residual on the b32 path
    (2, 128, 76, 76, 128, "MISH", True),    # yolov4's most frequent shape
    (1, 512, 19, 19, 512, "LEAKY", False),  # long K (64 stages), partitioned launch (U > 3 MB)
    (5, 16, 6, 10, 64, "LOGISTIC", False),  # generic activation path, tiles straddling images
    (3, 24, 4, 4, 192, "RELU", False),      # four tiles per image: 33 tile rows per strip
    (1, 8, 152, 152, 64, "LEAKY", True),    # wide rows (76 tiles > 64 per strip): windowed raw patch, 16-byte loads
    (1, 16, 304, 304, 64, "MISH", False),   # yolov4 layer 6's map: strips start mid-row at odd and even tiles
    (2, 32, 38, 38, 64, "LEAKY", True),     # 8-byte row loads (W % 4 == 2)
    (3, 16, 26, 26, 64, "MISH", False),     # yolov4-tiny map, 8-byte loads, 6 tile rows per strip
    (2, 16, 52, 52, 128, "LEAKY", True),    # 16-byte loads with bank-padding rows
    (1, 8, 130, 130, 64, "LINEAR", False),  # 65 tiles per row: wide case with 8-byte loads
    (2, 8, 17, 23, 64, "LEAKY", False),     # non-square odd map, 4-byte loads
]


@pytest.mark.parametrize("case", WINO_CASES)
def test_conv_winograd_vs_oracle(gpu, case):
    """Fused Winograd F(2x2,3x3) kernel (conv3x3_wino.hip) vs the CPU oracle of the reference's im2col+GEMM
    (convolutional_layer.cpp:1128-1305).  Not bitwise equal to the direct kernels by construction; bound =
    the fp32 tolerance of util.py, and the measured distance is printed next to the direct kernel's."""
    batch, c, h, w, n, actname, with_res = case
    act = getattr(O, actname)
    rng = np.random.default_rng(util.seed_of(case))
    x = rng.uniform(-1, 1, (batch, c, h, w)).astype(np.float32)
    wt = (rng.uniform(-1, 1, (n, c, 3, 3)) * np.sqrt(2.0 / (9 * c))).astype(np.float32)
    bias = rng.uniform(-.5, .5, n).astype(np.float32)
    res = rng.uniform(-1, 1, (batch, n, h, w)).astype(np.float32) if with_res else None
    ref, _ = orc_conv(x, wt, bias, batch, c, h, w, n, 3, 1, 1, act)
    if with_res:
        ref = ref + res
    L = gpu.lib()
    ncfg = L.dk_conv_force_config(-1)
    names = [L.dk_conv_config_name(i).decode() for i in range(ncfg)]
    wino = [i for i, nm in enumerate(names) if nm.startswith("wino")]
    assert wino
    y_direct = gpu.conv_forward(x, wt, bias, batch, c, h, w, n, 3, 1, 1, act, residual=res)
    L.dk_profile_enable(1)
    try:
        for cfg in wino:
            L.dk_conv_force_config(cfg)
            y = gpu.conv_forward(x, wt, bias, batch, c, h, w, n, 3, 1, 1, act, residual=res, wino=True)
            rms = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
            print("winograd %s %s: max|d|/rms %.3g (direct kernel: %.3g)" % (names[cfg], case,
                np.abs(y - ref).max() / rms, np.abs(y_direct - ref).max() / rms))
            util.assert_close(y, ref, "conv %s %s" % (names[cfg], case))
    finally:
        L.dk_conv_force_config(-1)
    out = (C.c_double * (3 * 256))()
    L.dk_profile_read(out, 256)
    L.dk_profile_enable(0)
    for cfg in wino:
        # variants: 16- or 4-byte row pieces x paired or single stores; a launch that fell back books none of them
        assert sum(out[(cfg * 4 + v) * 3] for v in range(4)) == 1, "config %s did not launch" % names[cfg]


def test_conv_winograd_on_caller_owned_memory_without_slack(gpu):
    """dk_conv_forward's memory contract (include/dk_kernels.h): no bytes behind a tensor are assumed readable.
    The Winograd kernel's 16-byte row pieces straddle row ends on maps whose width is not a multiple of 4 and would
    read up to 12 bytes past the input; library arrays (cuda_make_array) carry slack, a caller's hipMalloc does not.
    Input placed so that it ENDS at the end of a raw hipMalloc allocation: the launch must take the 4-byte-piece
    variant (profile slots 2 / 3) and still match the oracle; the same input in a library array takes the 16-byte
    variant (slots 0 / 1)."""
    batch, c, h, w, n = 2, 32, 38, 38, 64
    act = O.LEAKY
    rng = np.random.default_rng(11)
    x = rng.uniform(-1, 1, (batch, c, h, w)).astype(np.float32)
    wt = (rng.uniform(-1, 1, (n, c, 3, 3)) * np.sqrt(2.0 / (9 * c))).astype(np.float32)
    bias = rng.uniform(-.5, .5, n).astype(np.float32)
    ref, _ = orc_conv(x, wt, bias, batch, c, h, w, n, 3, 1, 1, act)
    L = gpu.lib()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipMemGetAddressRange.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_void_p]
    raw = C.c_void_p()
    lead = 4096                               # the tensor sits at the END of the allocation
    assert hip.hipMalloc(C.byref(raw), lead + x.nbytes) == 0
    xptr = raw.value + lead
    base, size = C.c_void_p(), C.c_size_t()
    assert hip.hipMemGetAddressRange(C.byref(base), C.byref(size), C.c_void_p(xptr)) == 0
    if base.value + size.value != xptr + x.nbytes:
        hip.hipFree(raw)
        pytest.skip("hipMemGetAddressRange reports a padded allocation (%d bytes for %d): no way to place a tensor at "
                    "the end of one" % (size.value, lead + x.nbytes))
    assert hip.hipMemcpy(C.c_void_p(xptr), x.ctypes.data, x.nbytes, 1) == 0
    d = gpu.DkConvDesc(batch, c, h, w, n, 1, 3, 1, 1, 1, 1, act)
    dw, db, dy = gpu.DeviceArray(wt), gpu.DeviceArray(bias), gpu.DeviceArray(n=batch * n * h * w)
    du = gpu.DeviceArray(n=L.dk_conv_wino_weights_size(C.byref(d)))
    assert L.dk_conv_wino_transform_weights(C.byref(d), dw.ptr, du.ptr, None) == 0
    L.dk_conv_wino_register(dw.ptr, du.ptr)
    ncfg = L.dk_conv_force_config(-1)
    names = [L.dk_conv_config_name(i).decode() for i in range(ncfg)]
    cfg = [i for i, nm in enumerate(names) if nm.startswith("wino")][0]
    dx = gpu.DeviceArray(x)
    try:
        L.dk_conv_force_config(cfg)
        for ptr, want in ((xptr, (2, 3)), (dx.ptr, (0, 1))):
            L.dk_profile_enable(1)
            assert L.dk_conv_forward(C.byref(d), C.c_void_p(ptr), dw.ptr, db.ptr, dy.ptr, None, None, None) == 0
            y = dy.numpy().reshape(ref.shape)
            out = (C.c_double * (3 * 256))()
            L.dk_profile_read(out, 256)
            L.dk_profile_enable(0)
            ran = [v for v in range(4) if out[(cfg * 4 + v) * 3] == 1]
            assert len(ran) == 1 and ran[0] in want, "variant %s ran, expected one of %s" % (ran, want)
            util.assert_close(y, ref, "winograd on %s memory" % ("caller-owned" if ptr == xptr else "library"))
    finally:
        L.dk_conv_force_config(-1)
        L.dk_conv_wino_register(dw.ptr, None)
        hip.hipFree(raw)


"""CPU ORACLE, network level.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product (darknet_amd/, include/) never does.

It restates, on top of the per-op C restatement in orc_ops.c (loaded through
ctypes), the reference's loader and layer loop for the YOLOv4 family:

  * ReadSections / option lookup ........ src/parser.cpp:59-100,
                                          src/option_list.cpp:134-243,
                                          strip(): src/utils.cpp:133-148
  * ParseNetOptions ..................... src/parser.cpp:921-1055
  * ParseConv / Yolo / Maxpool / Route /
    Shortcut / Upsample ................. src/parser.cpp:179-242, 312-415,
                                          640-659, 828-893, 720-779, 820-826
  * ParseNetworkCfg layer loop .......... src/parser.cpp:1076-1519
  * LoadWeightsUpTo / LoadConvolutionalWeights  src/parser.cpp:1778-1844, 1695-1759
  * FuseConvBatchNorm ................... src/network.cpp:647-682
  * ForwardNetwork / NetworkPredict ..... src/network.cpp:101-114, 412-430
  * GetNetworkBoxes ..................... src/network.cpp:432-516

Pinned against the real reference (oracle/_ref/libref_canon.so) by
tests/test_oracle_vs_ref.py in the build container, and against the committed
golden fixtures everywhere.
"""
import ctypes as C
import os
import struct
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# enum values, src/yolo_core.h:69-92 and :112-138
LOGISTIC, RELU, LINEAR, LEAKY, MISH = 0, 1, 4, 8, 17
ACT_NAMES = {"logistic": LOGISTIC, "relu": RELU, "linear": LINEAR,
             "leaky": LEAKY, "mish": MISH, "relu6": 2, "relie": 3, "ramp": 5, "tanh": 6, "plse": 7,
             "elu": 9, "loggy": 10, "stair": 11, "hardtan": 12, "lhtan": 13, "selu": 14, "gelu": 15,
             "swish": 16}
CONVOLUTIONAL, MAXPOOL, ROUTE, SHORTCUT, YOLO, UPSAMPLE = 0, 2, 7, 11, 17, 21
DROPOUT, AVGPOOL, SCALE_CHANNELS, BATCHNORM = 5, 9, 12, 14
GAUSSIAN_YOLO = 18   # LAYER_TYPE, yolo_core.h (YOLO = 17)

_lib = None


def build():
    """Compile orc_ops.c -> liborc.so (gcc -O2 -ffp-contract=off -fopenmp)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "liborc.so")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(
                os.path.join(HERE, "orc_ops.c")):
            build()
        _lib = C.CDLL(path)
        _lib.orc_num_threads.restype = C.c_int
        _lib.orc_yolo_num_detections.restype = C.c_int
        _lib.orc_yolo_detections.restype = C.c_int
    return _lib


def fptr(a):
    if a is None:
        return C.POINTER(C.c_float)()
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


def iptr(a):
    if a is None:
        return C.POINTER(C.c_int)()
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_int))


F = C.c_float


# ------------------------------------------------------------------ cfg parse
def read_sections(path):
    """ReadSections: every space/tab/CR/LF is stripped from each line, '[' opens
    a section, '#', ';' and empty lines are skipped, the rest is key=value split
    at the first '='."""
    sections = []
    with open(path) as f:
        for raw in f:
            line = "".join(ch for ch in raw if ch not in " \t\r\n")
            if not line or line[0] in "#;":
                continue
            if line[0] == "[":
                sections.append((line, {}))
            else:
                if "=" not in line:
                    continue
                k, v = line.split("=", 1)
                sections[-1][1][k] = v
    return sections


def _int(o, k, d):
    return int(o[k]) if k in o else d


def _float(o, k, d):
    return float(o[k]) if k in o else d


class Layer:
    pass


class Net:
    pass


def parse_cfg(path, batch=1, train=False):
    """ParseNetworkCfg restricted to the YOLOv4 family's layer kinds.
    `batch` overrides the reference's forced batch=1 for inference
    (src/parser.cpp:1114-1115); in train mode pass batch=None to use
    batch/subdivisions from the cfg (:927-929)."""
    secs = read_sections(path)
    assert secs and secs[0][0] in ("[net]", "[network]")
    o = secs[0][1]
    net = Net()
    net.cfg_batch = _int(o, "batch", 1)
    net.subdiv = _int(o, "subdivisions", 1)
    net.batch = batch if batch is not None else net.cfg_batch // net.subdiv
    net.h, net.w, net.c = _int(o, "height", 0), _int(o, "width", 0), _int(o, "channels", 0)
    net.lr = _float(o, "learning_rate", .001)
    net.momentum = _float(o, "momentum", .9)
    net.decay = _float(o, "decay", .0001)
    net.burn_in = _int(o, "burn_in", 0)
    net.power = _float(o, "power", 4)
    net.max_epoch = _int(o, "max_epoch", 0)
    net.policy = o.get("policy", "constant")
    net.steps = [float(x) for x in o["steps"].split(",")] if "steps" in o else []
    net.scales = [float(x) for x in o["scales"].split(",")] if "scales" in o else []
    net.train = train
    net.layers = []
    h, w, c = net.h, net.w, net.c
    inputs = h * w * c
    for idx, (name, o) in enumerate(secs[1:]):
        l = Layer()
        l.index = idx
        l.batch = net.batch
        l.h, l.w, l.c = h, w, c
        l.inputs = inputs
        if name == "[convolutional]":
            l.type = CONVOLUTIONAL
            l.n = _int(o, "filters", 1)
            l.groups = max(1, _int(o, "groups", 1))
            l.size = _int(o, "size", 1)
            stride = _int(o, "stride", 1)
            l.stride_x = _int(o, "stride_x", -1)
            l.stride_y = _int(o, "stride_y", -1)
            if l.stride_x < 1:
                l.stride_x = stride
            if l.stride_y < 1:
                l.stride_y = stride
            l.dilation = _int(o, "dilation", 1)
            if l.size == 1:
                l.dilation = 1
            pad = _int(o, "pad", 0)
            l.pad = _int(o, "padding", 0)
            if pad:
                l.pad = l.size // 2
            l.activation = ACT_NAMES[o.get("activation", "logistic")]
            l.batch_normalize = _int(o, "batch_normalize", 0)
            l.clip = _float(o, "clip", 0)   # src/parser.cpp:1361
            l.out_h = (h + 2 * l.pad - l.size) // l.stride_y + 1
            l.out_w = (w + 2 * l.pad - l.size) // l.stride_x + 1
            l.out_c = l.n
            l.nweights = (c // l.groups) * l.n * l.size * l.size
            # l->bflops, src/convolutional_layer.cpp:714
            l.bflops = (2.0 * l.nweights * l.out_h * l.out_w) / 1000000000.
            l.workspace = l.out_h * l.out_w * l.size * l.size * (c // l.groups)
        elif name == "[maxpool]":
            l.type = MAXPOOL
            stride = _int(o, "stride", 1)
            l.stride_x = _int(o, "stride_x", stride)
            l.stride_y = _int(o, "stride_y", stride)
            l.size = _int(o, "size", stride)
            l.pad = _int(o, "padding", l.size - 1)
            l.out_w = (w + l.pad - l.size) // l.stride_x + 1
            l.out_h = (h + l.pad - l.size) // l.stride_y + 1
            l.out_c = c
            # src/maxpool_layer.cpp:95 (float division of an int product)
            l.bflops = (l.size * l.size * c * l.out_h * l.out_w) / 1000000000.
        elif name == "[route]":
            l.type = ROUTE
            ids = [int(x) for x in o["layers"].split(",")]
            l.input_layers = [i if i >= 0 else idx + i for i in ids]
            l.input_sizes = [net.layers[i].outputs for i in l.input_layers]
            l.groups = _int(o, "groups", 1)
            l.group_id = _int(o, "group_id", 0)
            first = net.layers[l.input_layers[0]]
            l.out_w, l.out_h = first.out_w, first.out_h
            l.out_c = sum(net.layers[i].out_c for i in l.input_layers) // l.groups
            l.bflops = 0
        elif name == "[shortcut]":
            l.type = SHORTCUT
            frm = int(o["from"].split(",")[0])
            l.from_index = frm if frm >= 0 else idx + frm
            l.activation = ACT_NAMES[o.get("activation", "linear")]
            l.out_w, l.out_h, l.out_c = w, h, c
            src = net.layers[l.from_index]
            assert (src.out_w, src.out_h, src.out_c) == (w, h, c)
            # src/shortcut_layer.c: bflops = out_w*out_h*out_c*n / 1e9
            l.bflops = (l.out_w * l.out_h * l.out_c * 1) / 1000000000.
        elif name == "[upsample]":
            l.type = UPSAMPLE
            l.stride = _int(o, "stride", 2)
            l.scale = _float(o, "scale", 1)
            l.out_w, l.out_h, l.out_c = w * l.stride, h * l.stride, c
            l.bflops = 0
        elif name == "[yolo]":
            l.type = YOLO
            l.classes = _int(o, "classes", 20)
            l.total = _int(o, "num", 1)
            l.mask = [int(x) for x in o["mask"].split(",")] if "mask" in o \
                else list(range(l.total))
            l.n = len(l.mask)
            l.scale_x_y = _float(o, "scale_x_y", 1)
            l.biases = np.full(l.total * 2, .5, np.float32)
            if "anchors" in o:
                a = [float(x) for x in o["anchors"].split(",")]
                for i in range(min(len(a), l.total * 2)):
                    l.biases[i] = a[i]
            l.nms_kind = {"greedynms": 0, "diounms": 1}.get(o.get("nms_kind", "greedynms"), 0)
            l.beta_nms = _float(o, "beta_nms", 0.6)
            l.out_w, l.out_h = w, h
            l.out_c = l.n * (l.classes + 4 + 1)
            assert l.out_c == c, "filters= before [yolo] does not match classes/mask"
            l.bflops = 0
        elif name == "[Gaussian_yolo]":
            # ParseGaussianYolo / FillGaussianYoloLayer, src/parser.cpp:443-552, src/gaussian_yolo_layer.cpp:26-100
            l.type = GAUSSIAN_YOLO
            l.classes = _int(o, "classes", 20)
            l.total = _int(o, "num", 1)
            l.mask = [int(x) for x in o["mask"].split(",")] if "mask" in o else list(range(l.total))
            l.n = len(l.mask)
            l.scale_x_y = _float(o, "scale_x_y", 1)
            l.biases = np.full(l.total * 2, .5, np.float32)
            if "anchors" in o:
                a = [float(x) for x in o["anchors"].split(",")]
                for i in range(min(len(a), l.total * 2)):
                    l.biases[i] = a[i]
            l.nms_kind = {"greedynms": 0, "diounms": 1}.get(o.get("nms_kind", "greedynms"), 0)
            l.beta_nms = _float(o, "beta_nms", 0.6)
            l.out_w, l.out_h = w, h
            l.out_c = l.n * (l.classes + 8 + 1)
            assert l.out_c == c, "filters= before [Gaussian_yolo] does not match classes/mask"
            l.bflops = 0
        elif name == "[batchnorm]":
            # FillBatchnormLayer, src/batchnorm_layer.cpp:9-88
            l.type = BATCHNORM
            l.n = c
            l.out_w, l.out_h, l.out_c = w, h, c
            l.bflops = 0
        elif name in ("[avgpool]", "[avg]"):
            # FillAvgpoolLayer, src/avgpool_layer.cpp:6-40
            l.type = AVGPOOL
            l.out_w, l.out_h, l.out_c = 1, 1, c
            l.bflops = 0
        elif name == "[scale_channels]":
            # ParseScaleChannels / FillScaleChannelsLayer, src/parser.cpp:781-802, src/scale_channels_layer.c:9-48
            l.type = SCALE_CHANNELS
            frm = int(o["from"])
            l.from_index = frm if frm >= 0 else idx + frm
            l.scale_wh = _int(o, "scale_wh", 0)
            src = net.layers[l.from_index]
            l.out_w, l.out_h, l.out_c = src.out_w, src.out_h, src.out_c
            l.activation = ACT_NAMES[o.get("activation", "linear")]
            l.bflops = 0
        elif name == "[dropout]":
            # inference: identity on the previous layer's buffers (src/parser.cpp:1232-1242)
            l.type = DROPOUT
            l.probability = _float(o, "probability", .2)
            l.out_w, l.out_h, l.out_c = w, h, c
            l.bflops = 0
        else:
            raise ValueError("oracle: unsupported section %s" % name)
        l.outputs = l.out_h * l.out_w * l.out_c
        net.layers.append(l)
        h, w, c = l.out_h, l.out_w, l.out_c
        inputs = l.outputs
    net.n = len(net.layers)
    net.bflops = float(np.float32(sum(np.float32(l.bflops) for l in net.layers if l.bflops > 0)))
    net.workspace = max([getattr(l, "workspace", 0) for l in net.layers] + [1])
    return net


# ----------------------------------------------------------------- weights IO
def load_weights(net, path):
    """LoadWeightsUpTo: header int32 major, minor, revision + uint64 seen, then
    per conv: biases[n], (scales, rolling_mean, rolling_variance)[n] if BN,
    weights[nweights].  Returns the number of bytes consumed."""
    with open(path, "rb") as f:
        buf = f.read()
    major, minor, rev = struct.unpack_from("<iii", buf, 0)
    (net.seen,) = struct.unpack_from("<Q", buf, 12)
    off = 20

    def take(n):
        nonlocal off
        a = np.frombuffer(buf, np.float32, n, off).copy()
        off += 4 * n
        return a

    for l in net.layers:
        if l.type == BATCHNORM:   # LoadBatchnormWeights, src/parser.cpp:1683-1693
            l.biases, l.scales = take(l.c), take(l.c)
            l.rolling_mean, l.rolling_variance = take(l.c), take(l.c)
            continue
        if l.type != CONVOLUTIONAL:
            continue
        l.biases = take(l.n)
        if l.batch_normalize:
            l.scales = take(l.n)
            l.rolling_mean = take(l.n)
            l.rolling_variance = take(l.n)
        l.weights = take(l.nweights)
    return off


def weights_file_size(net):
    n = 20
    for l in net.layers:
        if l.type == CONVOLUTIONAL:
            n += 4 * (l.n + l.nweights + (3 * l.n if l.batch_normalize else 0))
        if l.type == BATCHNORM:
            n += 16 * l.c
    return n


def fuse_conv_batchnorm(net):
    """FuseConvBatchNorm for every BN conv; sets batch_normalize = 0."""
    L = lib()
    for l in net.layers:
        if l.type == CONVOLUTIONAL and l.batch_normalize:
            L.orc_fuse_conv_bn(fptr(l.weights), fptr(l.biases), fptr(l.scales),
                               fptr(l.rolling_mean), fptr(l.rolling_variance),
                               l.n, l.size * l.size * l.c // l.groups)
            l.batch_normalize = 0


def load_network(cfg, weights, batch=1):
    """LoadNetwork(train=false): parse, load, fuse (src/parser.cpp:1852-1876)."""
    net = parse_cfg(cfg, batch=batch, train=False)
    if weights:
        load_weights(net, weights)
        fuse_conv_batchnorm(net)
    return net


# -------------------------------------------------------------------- forward
def half_eligible(l):
    """The reference's fp16 rule, src/convolutional_kernels.cu:361-365."""
    return (l.type == CONVOLUTIONAL and l.size > 1 and l.c % 8 == 0 and l.n % 8 == 0 and
            l.groups == 1 and l.index != 0)


def forward(net, x, keep=None, upto=None, half=False):
    """ForwardNetwork in inference mode (BN folded).  x: float32 [batch, c*h*w].
    Every layer's output is kept in l.output (as the reference does).
    half=True: the oracle of BASELINE config C5 -- the reference has no CPU fp16
    path, so it is the CPU fp32 path with the eligible
    layers' inputs and weights pre-rounded to fp16 (round-to-nearest-even)."""
    L = lib()
    x = np.ascontiguousarray(x, np.float32).reshape(net.batch, -1)
    ws = np.zeros(net.workspace, np.float32)
    inp = x
    for l in net.layers:
        if upto is not None and l.index > upto:
            break
        B = l.batch
        out = np.zeros((B, l.outputs), np.float32)
        if l.type == CONVOLUTIONAL:
            assert not l.batch_normalize, "inference oracle expects fused BN"
            cin, cw = inp, l.weights
            if half and half_eligible(l):
                cin = np.ascontiguousarray(inp.astype(np.float16).astype(np.float32))
                cw = np.ascontiguousarray(l.weights.astype(np.float16).astype(np.float32))
            L.orc_conv_forward_fused(fptr(cin), fptr(cw), fptr(l.biases),
                                     fptr(out), fptr(ws), None, B, l.c, l.h, l.w,
                                     l.n, l.groups, l.size, l.stride_x, l.stride_y,
                                     l.dilation, l.pad, l.activation)
        elif l.type == MAXPOOL:
            L.orc_maxpool_forward(fptr(inp), fptr(out), None, B, l.c, l.h, l.w,
                                  l.size, l.stride_x, l.stride_y, l.pad)
        elif l.type == ROUTE:
            off = 0
            for src, size in zip(l.input_layers, l.input_sizes):
                L.orc_route_copy(fptr(net.layers[src].output), size, l.groups,
                                 l.group_id, B, fptr(out), l.outputs, off)
                off += size // l.groups
        elif l.type == SHORTCUT:
            L.orc_shortcut_forward(fptr(inp), fptr(net.layers[l.from_index].output),
                                   fptr(out), B * l.outputs)
            L.orc_activate_array(fptr(out), B * l.outputs, l.activation)
        elif l.type == UPSAMPLE:
            L.orc_upsample_forward(fptr(inp), l.w, l.h, l.c, B, l.stride,
                                   F(l.scale), fptr(out))
        elif l.type == YOLO:
            L.orc_yolo_forward(fptr(inp), fptr(out), B, l.w, l.h, l.n, l.classes,
                               F(l.scale_x_y))
        elif l.type == GAUSSIAN_YOLO:
            L.orc_gaussian_yolo_forward(fptr(inp), fptr(out), B, l.w, l.h, l.n, l.classes, F(l.scale_x_y))
        else:
            out = _forward_extra(L, net, l, inp, out, train=False)
        l.output = out
        inp = out
    return inp


def _bn_defaults(l):
    for name, v in (("biases", 0), ("scales", 1), ("rolling_mean", 0), ("rolling_variance", 0)):
        if not hasattr(l, name):
            setattr(l, name, np.full(l.c, v, np.float32))
    for name in ("mean", "variance", "mean_delta", "variance_delta"):
        if not hasattr(l, name):
            setattr(l, name, np.zeros(l.c, np.float32))


def _forward_extra(L, net, l, inp, out, train):
    """[batchnorm] / [avgpool] / [scale_channels] / [dropout] forward (src/batchnorm_layer.cpp:206-238,
    src/avgpool_layer.cpp:40-56, src/scale_channels_layer.c:70-95)."""
    B = l.batch
    if l.type == BATCHNORM:
        _bn_defaults(l)
        out[...] = inp
        l.x = np.zeros_like(out)
        l.x_norm = np.zeros_like(out)
        L.orc_batchnorm_forward(fptr(out), B, l.c, l.out_h * l.out_w, fptr(l.scales), fptr(l.biases),
                                fptr(l.rolling_mean), fptr(l.rolling_variance), fptr(l.mean),
                                fptr(l.variance), fptr(l.x), fptr(l.x_norm), 1 if train else 0)
    elif l.type == AVGPOOL:
        L.orc_avgpool_forward(fptr(inp), fptr(out), B, l.c, l.h, l.w)
    elif l.type == SCALE_CHANNELS:
        L.orc_scale_channels_forward(fptr(inp), fptr(net.layers[l.from_index].output), fptr(out), B,
                                     l.out_c, l.out_h, l.out_w, l.scale_wh)
        L.orc_activate_array(fptr(out), out.size, l.activation)
    elif l.type == DROPOUT:
        assert not train, "oracle: [dropout] in train mode needs the reference's RNG stream"
        out = inp
    else:
        raise ValueError("oracle: unsupported layer type %d" % l.type)
    return out


def get_boxes(net, thresh, b=0):
    """GetNetworkBoxes for batch item b: (dets [num, 5+classes], ids [num, 4] =
    (layer index, anchor, row, col))."""
    L = lib()
    dets, ids = [], []
    for l in net.layers:
        if l.type == GAUSSIAN_YOLO:
            # GetGaussianYoloDetections: the record also carries the four uncertainties (dropped here; see
            # get_gaussian_boxes for them)
            d, i3 = _gaussian_dets(L, net, l, thresh, b)
            dets.append(d[:, :5 + l.classes])
            ids.append(np.concatenate([np.full((len(d), 1), l.index, np.int32), i3], 1))
            continue
        if l.type != YOLO:
            continue
        num = L.orc_yolo_num_detections(fptr(l.output), b, l.w, l.h, l.n,
                                        l.classes, F(thresh))
        d = np.zeros((num, 5 + l.classes), np.float32)
        i3 = np.zeros((num, 3), np.int32)
        mask = np.array(l.mask, np.int32)
        got = L.orc_yolo_detections(fptr(l.output), b, l.w, l.h, l.n, l.classes,
                                    fptr(l.biases), iptr(mask), net.w, net.h,
                                    F(thresh), fptr(d), iptr(i3))
        assert got == num
        dets.append(d)
        ids.append(np.concatenate([np.full((num, 1), l.index, np.int32), i3], 1))
    if not dets:
        return np.zeros((0, 5), np.float32), np.zeros((0, 4), np.int32)
    return np.concatenate(dets), np.concatenate(ids)


def _gaussian_dets(L, net, l, thresh, b):
    num = L.orc_gaussian_yolo_num_detections(fptr(l.output), b, l.w, l.h, l.n, l.classes, F(thresh))
    d = np.zeros((num, 5 + l.classes + 4), np.float32)
    i3 = np.zeros((num, 3), np.int32)
    mask = np.array(l.mask, np.int32)
    got = L.orc_gaussian_yolo_detections(fptr(l.output), b, l.w, l.h, l.n, l.classes, fptr(l.biases), iptr(mask),
                                         net.w, net.h, F(thresh), fptr(d), iptr(i3))
    assert got == num
    return d, i3


def get_gaussian_boxes(net, thresh, b=0):
    """Detections of the [Gaussian_yolo] heads with their uncertainties: [num, 5 + classes + 4]."""
    L = lib()
    out = [_gaussian_dets(L, net, l, thresh, b)[0] for l in net.layers if l.type == GAUSSIAN_YOLO]
    return np.concatenate(out) if out else np.zeros((0, 9), np.float32)


# ------------------------------------------------------------------ training
def load_network_train(cfg, weights, batch):
    """LoadNetwork(train=true) semantics for the layer kinds of the path: BN is
    NOT folded; per-layer delta / update buffers are allocated and zeroed."""
    net = parse_cfg(cfg, batch=batch, train=True)
    if weights:
        load_weights(net, weights)
    for l in net.layers:
        l.delta = np.zeros((l.batch, l.outputs), np.float32)
        if l.type == CONVOLUTIONAL:
            l.weight_updates = np.zeros(l.nweights, np.float32)
            l.bias_updates = np.zeros(l.n, np.float32)
            if l.batch_normalize:
                l.scale_updates = np.zeros(l.n, np.float32)
                for name in ("mean", "variance", "mean_delta", "variance_delta"):
                    setattr(l, name, np.zeros(l.n, np.float32))
        if l.type == BATCHNORM:
            _bn_defaults(l)
            l.bias_updates = np.zeros(l.c, np.float32)
            l.scale_updates = np.zeros(l.c, np.float32)
    return net


def forward_train(net, x):
    """ForwardNetwork with state.train = 1 (src/network.cpp:101-114): every
    l->delta is zeroed, conv = GEMM -> BN with batch statistics (rolling stats
    updated) or bias -> activation (mish keeps its input).  The yolo layer only
    decodes here; its delta (the loss gradient) is injected by the caller."""
    L = lib()
    x = np.ascontiguousarray(x, np.float32).reshape(net.batch, -1)
    net.input = x
    ws = np.zeros(net.workspace, np.float32)
    inp = x
    for l in net.layers:
        B = l.batch
        l.delta[...] = 0
        out = np.zeros((B, l.outputs), np.float32)
        if l.type == CONVOLUTIONAL:
            L.orc_conv_gemm_forward(fptr(inp), fptr(l.weights), fptr(out), fptr(ws), B, l.c, l.h,
                                    l.w, l.n, l.groups, l.size, l.stride_x, l.stride_y, l.dilation, l.pad)
            sp = l.out_h * l.out_w
            if l.batch_normalize:
                l.x = np.zeros_like(out)
                l.x_norm = np.zeros_like(out)
                L.orc_batchnorm_forward(fptr(out), B, l.n, sp, fptr(l.scales), fptr(l.biases),
                                        fptr(l.rolling_mean), fptr(l.rolling_variance), fptr(l.mean),
                                        fptr(l.variance), fptr(l.x), fptr(l.x_norm), 1)
            else:
                L.orc_add_bias(fptr(out), fptr(l.biases), B, l.n, sp)
            if l.activation == MISH:
                l.activation_input = np.zeros_like(out)
                L.orc_activate_array_mish(fptr(out), out.size, fptr(l.activation_input), fptr(out))
            elif l.activation == 16:  # SWISH: activate_array_swish keeps sigmoid(x) in activation_input
                l.activation_input = out.copy()
                L.orc_activate_array(fptr(l.activation_input), out.size, LOGISTIC)
                L.orc_activate_array(fptr(out), out.size, l.activation)
            else:
                L.orc_activate_array(fptr(out), out.size, l.activation)
        elif l.type == MAXPOOL:
            l.indexes = np.zeros((B, l.outputs), np.int32)
            L.orc_maxpool_forward(fptr(inp), fptr(out), iptr(l.indexes), B, l.c, l.h, l.w, l.size,
                                  l.stride_x, l.stride_y, l.pad)
        elif l.type == ROUTE:
            off = 0
            for src, size in zip(l.input_layers, l.input_sizes):
                L.orc_route_copy(fptr(net.layers[src].output), size, l.groups, l.group_id, B,
                                 fptr(out), l.outputs, off)
                off += size // l.groups
        elif l.type == SHORTCUT:
            L.orc_shortcut_forward(fptr(inp), fptr(net.layers[l.from_index].output), fptr(out), B * l.outputs)
            L.orc_activate_array(fptr(out), B * l.outputs, l.activation)
        elif l.type == UPSAMPLE:
            L.orc_upsample_forward(fptr(inp), l.w, l.h, l.c, B, l.stride, F(l.scale), fptr(out))
        elif l.type == YOLO:
            L.orc_yolo_forward(fptr(inp), fptr(out), B, l.w, l.h, l.n, l.classes, F(l.scale_x_y))
        elif l.type == GAUSSIAN_YOLO:
            L.orc_gaussian_yolo_forward(fptr(inp), fptr(out), B, l.w, l.h, l.n, l.classes, F(l.scale_x_y))
        else:
            out = _forward_extra(L, net, l, inp, out, train=True)
        l.output = out
        inp = out
    return inp


def backward(net):
    """BackwardNetwork (src/network.cpp:160-190) for the path's layer kinds."""
    L = lib()
    ws = np.zeros(net.workspace, np.float32)
    for i in range(net.n - 1, -1, -1):
        l = net.layers[i]
        prev = net.layers[i - 1] if i > 0 else None
        p_in = prev.output if prev is not None else net.input
        p_delta = prev.delta if prev is not None else None
        B = l.batch
        tot = B * l.outputs
        if l.type == YOLO:
            p_delta += l.delta  # BackwardYoloLayer: axpy_cpu(1)
        elif l.type == CONVOLUTIONAL:
            if l.activation == MISH:
                L.orc_gradient_array_mish(tot, fptr(l.activation_input), fptr(l.delta))
            elif l.activation == 16:
                L.orc_gradient_array_swish(fptr(l.output), tot, fptr(l.activation_input), fptr(l.delta))
            else:
                L.orc_gradient_array(fptr(l.output), tot, l.activation, fptr(l.delta))
            sp = l.out_h * l.out_w
            if l.batch_normalize:
                L.orc_batchnorm_backward(fptr(l.delta), B, l.n, sp, fptr(l.scales), fptr(l.x), fptr(l.x_norm),
                                         fptr(l.mean), fptr(l.variance), fptr(l.mean_delta),
                                         fptr(l.variance_delta), fptr(l.scale_updates))
            else:
                L.orc_backward_bias(fptr(l.bias_updates), fptr(l.delta), B, l.n, sp)
            L.orc_conv_backward(fptr(p_in), fptr(l.weights), fptr(l.delta), fptr(l.weight_updates),
                                fptr(p_delta) if p_delta is not None else None, fptr(ws), B, l.c, l.h, l.w,
                                l.n, l.groups, l.size, l.stride_x, l.stride_y, l.dilation, l.pad)
        elif l.type == ROUTE:
            off = 0
            for src, size in zip(l.input_layers, l.input_sizes):
                L.orc_route_backward(fptr(l.delta), l.outputs, off, size, l.groups, l.group_id, B,
                                     fptr(net.layers[src].delta))
                off += size // l.groups
        elif l.type == SHORTCUT:
            L.orc_gradient_array(fptr(l.output), tot, l.activation, fptr(l.delta))
            L.orc_shortcut_backward(fptr(l.delta), tot, fptr(p_delta), fptr(net.layers[l.from_index].delta))
        elif l.type == MAXPOOL:
            L.orc_maxpool_backward(fptr(l.delta), iptr(l.indexes), tot, fptr(p_delta))
        elif l.type == UPSAMPLE:
            L.orc_upsample_backward(fptr(l.delta), l.w, l.h, l.c, B, l.stride, F(l.scale), fptr(p_delta))
        elif l.type == BATCHNORM:
            # BackwardBatchnormLayer, src/batchnorm_layer.cpp:240-255 (bias_updates stay untouched on the CPU)
            L.orc_batchnorm_backward(fptr(l.delta), B, l.c, l.out_h * l.out_w, fptr(l.scales), fptr(l.x), fptr(l.x_norm),
                                     fptr(l.mean), fptr(l.variance), fptr(l.mean_delta),
                                     fptr(l.variance_delta), fptr(l.scale_updates))
            if p_delta is not None:
                p_delta[...] = l.delta
        elif l.type == AVGPOOL:
            L.orc_avgpool_backward(fptr(l.delta), fptr(p_delta), B, l.c, l.h, l.w)
        elif l.type == SCALE_CHANNELS:
            L.orc_gradient_array(fptr(l.output), tot, l.activation, fptr(l.delta))
            frm = net.layers[l.from_index]
            L.orc_scale_channels_backward(fptr(l.delta), fptr(p_in), fptr(frm.output), fptr(frm.delta), fptr(p_delta),
                                          B, l.out_c, l.out_h, l.out_w, l.scale_wh)


def update(net, actual_batch, lr, momentum, decay):
    """UpdateNetwork -> UpdateConvolutionalLayer (src/convolutional_layer.cpp:1382-1399)."""
    L = lib()
    for l in net.layers:
        if l.type == CONVOLUTIONAL:
            L.orc_conv_update(fptr(l.weights), fptr(l.weight_updates), l.nweights, fptr(l.biases),
                              fptr(l.bias_updates), fptr(l.scales) if l.batch_normalize else None,
                              fptr(l.scale_updates) if l.batch_normalize else None, l.n, actual_batch,
                              F(lr), F(momentum), F(decay))
            if getattr(l, "clip", 0):   # convolutional_kernels.cu:919-920 (GPU path only in the reference)
                L.orc_constrain(l.nweights, F(l.clip), fptr(l.weights))
        elif l.type == BATCHNORM:
            L.orc_batchnorm_update(fptr(l.biases), fptr(l.bias_updates), fptr(l.scales), fptr(l.scale_updates),
                                   l.c, actual_batch, F(lr), F(momentum))

"""CPU: the oracle restatement (oracle/orc_ops.c + orc_net.py) against the golden
fixtures dumped from the REAL reference (tools/make_golden.py).  Bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest

import synth
from oracle import orc_net as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = C.c_float


@pytest.fixture(scope="module")
def ops():
    return np.load(os.path.join(GOLD, "ops.npz"))


def test_im2col_col2im(ops):
    L = O.lib()
    for i, (c, h, w, kh, kw, ph, pw, sh, sw, dh, dw) in enumerate(ops["im2col_cases"]):
        c, h, w, kh, kw, ph, pw, sh, sw, dh, dw = map(int, (c, h, w, kh, kw, ph, pw, sh, sw, dh, dw))
        im = np.ascontiguousarray(ops[f"im2col_{i}_im"])
        col = np.zeros_like(ops[f"im2col_{i}_col"])
        L.orc_im2col_ext(O.fptr(im), c, h, w, kh, kw, ph, pw, sh, sw, dh, dw, O.fptr(col))
        assert np.array_equal(col, ops[f"im2col_{i}_col"])
        cin = np.ascontiguousarray(ops[f"col2im_{i}_col"])
        back = np.full((c, h, w), 7.0, np.float32)
        L.orc_col2im_ext(O.fptr(cin), c, h, w, kh, kw, ph, pw, sh, sw, dh, dw, O.fptr(back))
        assert np.array_equal(back, ops[f"col2im_{i}_im"])


def test_gemm(ops):
    L = O.lib()
    for i, (ta, tb, m, n, k) in enumerate(ops["gemm_cases"]):
        alpha, beta = ops["gemm_alpha_beta"][i]
        A = np.ascontiguousarray(ops[f"gemm_{i}_A"])
        B = np.ascontiguousarray(ops[f"gemm_{i}_B"])
        Cm = ops[f"gemm_{i}_C0"].copy()
        L.orc_gemm(int(ta), int(tb), int(m), int(n), int(k), F(alpha), O.fptr(A), A.shape[1],
                   O.fptr(B), B.shape[1], F(beta), O.fptr(Cm), int(n))
        assert np.array_equal(Cm, ops[f"gemm_{i}_C"]), i


EXTRA_ACTIVATIONS = (('relu6', 2), ('relie', 3), ('ramp', 5), ('tanh', 6), ('plse', 7), ('elu', 9), ('loggy', 10),
                     ('hardtan', 12), ('lhtan', 13), ('selu', 14), ('gelu', 15), ('swish', 16))


def test_activations(ops):
    L = O.lib()
    g = ops["act_grid"]
    for name, a in (("leaky", O.LEAKY), ("logistic", O.LOGISTIC), ("relu", O.RELU)):
        x = g.copy()
        L.orc_activate_array(O.fptr(x), x.size, a)
        assert np.array_equal(x, ops["act_" + name]), name
    # the rarer kinds of activate() and swish (ACTIVATION enum values, src/yolo_core.h:69-92)
    for name, a in EXTRA_ACTIVATIONS:
        x = g.copy()
        L.orc_activate_array(O.fptr(x), x.size, a)
        assert np.array_equal(x, ops["act_" + name]), name
    y = np.zeros_like(g)
    ain = np.zeros_like(g)
    L.orc_activate_array_mish(O.fptr(g.copy()), g.size, O.fptr(ain), O.fptr(y))
    assert np.array_equal(y, ops["act_mish"]) and np.array_equal(ain, ops["act_mish_input"])
    d = np.ones_like(g)
    L.orc_gradient_array_mish(g.size, O.fptr(g.copy()), O.fptr(d))
    assert np.array_equal(d, ops["grad_mish"])
    for name, a in (("leaky", O.LEAKY), ("logistic", O.LOGISTIC)):
        d = np.ones_like(g)
        L.orc_gradient_array(O.fptr(np.ascontiguousarray(ops["act_" + name])), g.size, a, O.fptr(d))
        assert np.array_equal(d, ops["grad_" + name])


def test_bn_statistics(ops):
    L = O.lib()
    x = np.ascontiguousarray(ops["bn_x"])
    mean = np.zeros(5, np.float32)
    var = np.zeros(5, np.float32)
    L.orc_mean(O.fptr(x), 3, 5, 42, O.fptr(mean))
    L.orc_variance(O.fptr(x), O.fptr(mean), 3, 5, 42, O.fptr(var))
    xn = x.copy()
    L.orc_normalize(O.fptr(xn), O.fptr(mean), O.fptr(var), 3, 5, 42)
    assert np.array_equal(mean, ops["bn_mean"])
    assert np.array_equal(var, ops["bn_var"])
    assert np.array_equal(xn, ops["bn_norm"])


def test_maxpool_with_indexes():
    g = np.load(os.path.join(GOLD, "maxpool.npz"))
    L = O.lib()
    x = np.ascontiguousarray(g["x"])
    for i, (size, stride) in enumerate(g["cases"]):
        size, stride = int(size), int(stride)
        pad = size - 1
        ow = (19 + pad - size) // stride + 1
        y = np.zeros(6 * ow * ow, np.float32)
        idx = np.zeros(6 * ow * ow, np.int32)
        L.orc_maxpool_forward(O.fptr(x), O.fptr(y), O.iptr(idx), 1, 6, 19, 19, size, stride, stride, pad)
        assert np.array_equal(y, g[f"y_{i}"]) and np.array_equal(idx, g[f"idx_{i}"]), (size, stride)


@pytest.mark.parametrize("name", ["yolov4-tiny", "yolov4-csp", "yolov4"])
def test_whole_net_inference(name, tmp_path):
    g = np.load(os.path.join(GOLD, f"net_{name}.npz"))
    cfg = os.path.join(ROOT, "cfg", name + ".cfg")
    net = O.parse_cfg(cfg)
    assert net.n == int(g["n_layers"])
    convs = [(l.n, l.c // l.groups, l.size, l.batch_normalize) for l in net.layers if l.type == O.CONVOLUTIONAL]
    wpath = str(tmp_path / "w.weights")
    synth.write_weights(wpath, convs, seed=2024)
    assert os.path.getsize(wpath) == int(g["weights_bytes"]) == O.weights_file_size(net)
    net = O.load_network(cfg, wpath, batch=1)
    x = synth.make_input(1, net.c, net.h, net.w, seed=12345)
    O.forward(net, x)
    for i, l in enumerate(net.layers):
        o = l.output.ravel()
        assert l.type == g["layer_types"][i] and l.outputs == g["layer_outputs"][i]
        idx = np.linspace(0, o.size - 1, 64).astype(np.int64)
        assert np.array_equal(o[idx], g["layer_samples"][i]), f"layer {i} samples"
        assert np.sum(o, dtype=np.float64) == g["layer_sums"][i][0], f"layer {i} sum"
        assert np.sum(o.astype(np.float64) ** 2) == g["layer_sums"][i][1], f"layer {i} sum of squares"
        if l.type == O.YOLO:
            if f"head_{i}" in g:
                assert np.array_equal(o, g[f"head_{i}"])
            else:
                assert np.array_equal(o[::16], g[f"head_{i}_sub16"])
    dets, ids = O.get_boxes(net, float(g["thresh"]))
    assert len(dets) == int(g["num_dets"])
    assert np.array_equal(ids, g["det_ids"])
    assert np.array_equal(dets[:, :5], g["det_box_obj"])
    assert np.array_equal(np.argmax(dets[:, 5:], 1), g["det_best_class"])
    assert np.array_equal(np.max(dets[:, 5:], 1), g["det_best_prob"])


def train_fixture(tmp_path, name="yolov4-tiny"):
    g = np.load(os.path.join(GOLD, "train_%s.npz" % name))
    B = int(g["batch"])
    cfg = str(tmp_path / "t.cfg")
    open(cfg, "w").write(open(os.path.join(ROOT, "cfg", name + ".cfg")).read().replace("batch=64", "batch=%d" % B))
    net = O.parse_cfg(cfg)
    wpath = str(tmp_path / "w.weights")
    synth.write_weights(wpath, [(l.n, l.c // l.groups, l.size, l.batch_normalize) for l in net.layers if l.type == O.CONVOLUTIONAL], seed=2024)
    x = synth.make_input(B, net.c, net.h, net.w, seed=12345)
    return g, cfg, wpath, x


def summ(a):
    idx = np.linspace(0, a.size - 1, 16).astype(np.int64)
    return np.concatenate([[np.sum(a, dtype=np.float64), np.sum(a.astype(np.float64) ** 2)], a[idx].astype(np.float64)])


def inject_yolo_deltas(net, g):
    for i, l in enumerate(net.layers):
        if l.type == O.YOLO:
            d = np.zeros(l.batch * l.outputs, np.float32)
            d[g["yolo_%d_delta_idx" % i]] = g["yolo_%d_delta_val" % i]
            l.delta[...] = d.reshape(l.delta.shape)


def test_train_step_golden(tmp_path):
    """One train step of the real reference (BN batch statistics, backward through
    every layer kind, SGD update), driven by the reference's own yolo deltas."""
    g, cfg, wpath, x = train_fixture(tmp_path)
    net = O.load_network_train(cfg, wpath, None)
    O.forward_train(net, x)
    inject_yolo_deltas(net, g)
    O.backward(net)
    which_name = {7: "weight_updates", 8: "bias_updates", 9: "scale_updates", 6: "delta"}
    for row in g["grad_summaries"]:
        i, which = int(row[0]), int(row[1])
        a = getattr(net.layers[i], which_name[which])
        assert np.array_equal(summ(a.ravel()), row[2:]), (i, which_name[which])
    O.update(net, net.batch * net.subdiv, float(g["lr"]), net.momentum, net.decay)
    for row in g["updated_weight_summaries"]:
        assert np.array_equal(summ(net.layers[int(row[0])].weights), row[1:])


def test_extra_layer_kinds_golden(tmp_path):
    """[batchnorm] / [avgpool] / [scale_channels] / [dropout]: the oracle reproduces,
    bit for bit, the REAL reference's outputs of every layer of cfg/se-test.cfg (inference) and one
    train step (forward with batch statistics, backward, update) of the same net without [dropout]
    and [batchnorm] (tools/make_golden.py extra; see synth.se_cfgs for why those two are left out)."""
    g = np.load(os.path.join(GOLD, "extra_se-test.npz"))
    inf, tr = synth.se_cfgs(tmp_path)
    onet = O.parse_cfg(inf)
    w = str(tmp_path / "w.weights")
    synth.write_weights_layers(w, synth.weight_layers_of(onet), seed=2024)
    onet = O.load_network(inf, w, batch=1)
    O.forward(onet, synth.make_input(1, onet.c, onet.h, onet.w, seed=12345))
    kinds = set()
    for i, l in enumerate(onet.layers):
        assert np.array_equal(l.output.ravel(), g["inf_out_%d" % i]), i
        kinds.add(l.type)
    assert {O.BATCHNORM, O.AVGPOOL, O.SCALE_CHANNELS, O.DROPOUT} <= kinds
    tnet = O.parse_cfg(tr)
    tw = str(tmp_path / "tw.weights")
    synth.write_weights_layers(tw, synth.weight_layers_of(tnet), seed=2024)
    tnet = O.load_network_train(tr, tw, None)
    x = synth.make_input(tnet.batch, tnet.c, tnet.h, tnet.w, seed=int(g["train_x_seed"]))
    O.forward_train(tnet, x)
    for i, l in enumerate(tnet.layers):
        assert np.array_equal(l.output.ravel(), g["train_out_%d" % i]), i
        if l.type == O.YOLO:
            l.delta[...] = g["train_yolo_delta_%d" % i].reshape(l.delta.shape)
    O.backward(tnet)
    checked = 0
    for i, l in enumerate(tnet.layers):
        for nm in ("weight_updates", "bias_updates", "scale_updates"):
            k = "train_%s_%d" % (nm, i)
            if k in g.files:
                assert np.array_equal(getattr(l, nm), g[k]), k
                checked += 1
        k = "train_delta_%d" % i
        if k in g.files:
            assert np.array_equal(l.delta.ravel(), g[k]), k
    assert checked >= 8


RARE_GRADS = (("relu6", 2), ("relie", 3), ("ramp", 5), ("tanh", 6), ("plse", 7), ("elu", 9), ("loggy", 10),
              ("hardtan", 12), ("lhtan", 13), ("selu", 14), ("gelu", 15), ("relu", 1))


def test_rare_activation_gradients_golden(ops):
    """gradient() of the 12 rarer activations and gradient_array_swish: the oracle reproduces the
    reference's gradient_array output on the activation grid bit for bit (tests/golden/ops_grad.npz)."""
    L = O.lib()
    gg = np.load(os.path.join(GOLD, "ops_grad.npz"))
    for name, a in RARE_GRADS:
        y = np.ascontiguousarray(ops["act_" + name])
        d = np.ones_like(y)
        L.orc_gradient_array(O.fptr(y), y.size, a, O.fptr(d))
        assert np.array_equal(d, gg["grad_" + name]), name
    y = np.ascontiguousarray(ops["act_swish"])
    d = np.ones_like(y)
    L.orc_gradient_array_swish(O.fptr(y), y.size, O.fptr(np.ascontiguousarray(gg["swish_sigmoid"])), O.fptr(d))
    assert np.array_equal(d, gg["grad_swish"])


def test_gaussian_yolo_oracle_vs_reference_golden(tmp_path):
    """[Gaussian_yolo] heads: the oracle's decoded heads and detection list on
    cfg/gaussian-test.cfg are bit-identical to the REAL reference's (tests/golden/gaussian-test.npz,
    tools/make_golden.py gaussian)."""
    g = np.load(os.path.join(GOLD, "gaussian-test.npz"))
    cfg = os.path.join(ROOT, "cfg", "gaussian-test.cfg")
    net = O.parse_cfg(cfg)
    w = str(tmp_path / "w.weights")
    synth.write_weights_layers(w, synth.weight_layers_of(net), seed=2024)
    assert os.path.getsize(w) == int(g["weights_bytes"]) == O.weights_file_size(net)
    net = O.load_network(cfg, w, batch=1)
    O.forward(net, synth.make_input(1, net.c, net.h, net.w, seed=12345))
    heads = [i for i, l in enumerate(net.layers) if l.type == O.GAUSSIAN_YOLO]
    assert len(heads) == 2 and net.n == int(g["n_layers"])
    for i in heads:
        assert np.array_equal(net.layers[i].output.ravel(), g["head_%d" % i])
    d, ids = O.get_boxes(net, float(g["thresh"]))
    assert np.array_equal(d, g["dets"]) and np.array_equal(ids, g["det_ids"])
    assert np.array_equal(O.get_gaussian_boxes(net, float(g["thresh"]))[:, -4:], g["dets_uc"])

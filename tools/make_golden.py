#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the REAL reference.

Runs only in the build container (needs oracle/_ref/libref_canon.so, i.e. the
reference's own CPU sources compiled unmodified by oracle/Makefile, canonical
build: -O2 -ffp-contract=off, scalar branches).  The fixtures are data only:
inputs, seeds and the reference's outputs -- never reference source text.

  ops.npz            per-op vectors: im2col/col2im on odd shapes, gemm NN/NT/TN/TT,
                     activations on a fixed grid (incl. the +-20 softplus
                     thresholds), maxpool 2/2 and SPP 5,9,13/1 at 19x19 with
                     indexes, BN statistics, one fused-BN layer, yolo decode.
  net_<cfg>.npz      whole-net inference (b=1, synthetic weights seed 2024, input
                     seed 12345): per-layer sum / sum-of-squares / 64 strided
                     samples, the full decoded yolo heads (tiny) or a 1/16
                     subsample (yolov4, csp), and the detection list at a
                     guard-banded threshold (box fields, objectness, ids).
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import reflib  # noqa: E402
import synth  # noqa: E402
from oracle import orc_net as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
FP = C.POINTER(C.c_float)
IPT = C.POINTER(C.c_int)


def fp(a):
    return a.ctypes.data_as(FP)


def ip(a):
    return a.ctypes.data_as(IPT)


def gen_ops():
    L = reflib.lib("canon")
    rng = np.random.default_rng(20240601)
    out = {}
    # ---- im2col / col2im (Caffe-style ext, pad/stride/dilation, odd shapes)
    cases = [(3, 7, 9, 3, 3, 1, 1, 1, 1, 1, 1), (2, 8, 5, 3, 3, 1, 1, 2, 2, 1, 1),
             (4, 9, 9, 3, 3, 2, 2, 1, 1, 2, 2), (1, 6, 11, 5, 5, 2, 2, 1, 2, 1, 1),
             (5, 4, 4, 1, 1, 0, 0, 1, 1, 1, 1), (2, 10, 7, 3, 3, 0, 0, 3, 2, 1, 1)]
    out["im2col_cases"] = np.array(cases, np.int32)
    for i, (c, h, w, kh, kw, ph, pw, sh, sw, dh, dw) in enumerate(cases):
        im = rng.uniform(-1, 1, (c, h, w)).astype(np.float32)
        oh = (h + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
        ow = (w + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
        col = np.zeros((c * kh * kw, oh * ow), np.float32)
        L.im2col_cpu_ext(fp(im), c, h, w, kh, kw, ph, pw, sh, sw, dh, dw, fp(col))
        cin = rng.uniform(-1, 1, col.shape).astype(np.float32)
        back = np.full((c, h, w), 7.0, np.float32)
        L.col2im_cpu_ext(fp(cin), c, h, w, kh, kw, ph, pw, sh, sw, dh, dw, fp(back))
        out[f"im2col_{i}_im"], out[f"im2col_{i}_col"] = im, col
        out[f"col2im_{i}_col"], out[f"col2im_{i}_im"] = cin, back
    # ---- gemm (gemm_cpu, all four transposes, beta 1 and 0)
    L.gemm_cpu.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, FP, C.c_int,
                           FP, C.c_int, C.c_float, FP, C.c_int]
    gcases = [(0, 0, 7, 13, 27, 1.0, 1.0), (0, 1, 5, 9, 31, 1.0, 1.0), (1, 0, 9, 11, 6, 1.0, 0.0),
              (1, 1, 4, 6, 10, 1.0, 1.0), (0, 0, 32, 169, 288, 1.0, 1.0), (0, 0, 3, 5, 4, 0.5, 2.0)]
    out["gemm_cases"] = np.array([(a, b, m, n, k) for a, b, m, n, k, _, _ in gcases], np.int32)
    out["gemm_alpha_beta"] = np.array([(al, be) for *_, al, be in gcases], np.float32)
    for i, (ta, tb, m, n, k, alpha, beta) in enumerate(gcases):
        A = rng.uniform(-1, 1, (k, m) if ta else (m, k)).astype(np.float32)
        B = rng.uniform(-1, 1, (n, k) if tb else (k, n)).astype(np.float32)
        Cm = rng.uniform(-1, 1, (m, n)).astype(np.float32)
        C0 = Cm.copy()
        L.gemm_cpu(ta, tb, m, n, k, alpha, fp(A), A.shape[1], fp(B), B.shape[1], beta, fp(Cm), n)
        out[f"gemm_{i}_A"], out[f"gemm_{i}_B"], out[f"gemm_{i}_C0"], out[f"gemm_{i}_C"] = A, B, C0, Cm
    # ---- activations on a fixed grid
    grid = np.concatenate([np.linspace(-30, 30, 6001),
                           [20.0, -20.0, 20.000002, -20.000002, 19.999998, -19.999998, 0.0, 1e-30]]).astype(np.float32)
    out["act_grid"] = grid
    L.activate_array_cpu_custom.argtypes = [FP, C.c_int, C.c_int]
    for name, a in (("leaky", O.LEAKY), ("logistic", O.LOGISTIC), ("relu", O.RELU)):
        x = grid.copy()
        L.activate_array_cpu_custom(fp(x), x.size, a)
        out["act_" + name] = x
    # the rarer kinds of activate() and swish (sibling cfgs: efficientnet, mobilenet, ...)
    for name, a in (("relu6", 2), ("relie", 3), ("ramp", 5), ("tanh", 6), ("plse", 7), ("elu", 9), ("loggy", 10),
                    ("hardtan", 12), ("lhtan", 13), ("selu", 14), ("gelu", 15)):
        x = grid.copy()
        L.activate_array_cpu_custom(fp(x), x.size, a)
        out["act_" + name] = x
    x = grid.copy()
    sig = np.zeros_like(x)
    y = np.zeros_like(x)
    L.activate_array_swish.argtypes = [FP, C.c_int, FP, FP]
    L.activate_array_swish(fp(x), x.size, fp(sig), fp(y))
    out["act_swish"] = y
    x = grid.copy()
    ain = np.zeros_like(x)
    y = np.zeros_like(x)
    L.activate_array_mish(fp(x), x.size, fp(ain), fp(y))
    out["act_mish"], out["act_mish_input"] = y, ain
    d = np.ones_like(grid)
    L.gradient_array_mish(grid.size, fp(grid), fp(d))
    out["grad_mish"] = d
    for name, a in (("leaky", O.LEAKY), ("logistic", O.LOGISTIC)):
        d = np.ones_like(grid)
        yv = out["act_" + name]
        L.gradient_array(fp(yv), grid.size, a, fp(d))
        out["grad_" + name] = d
    # ---- maxpool via the public layer path of the reference net (cfg below)
    # ---- BN statistics
    xb = rng.uniform(-2, 2, (3, 5, 7 * 6)).astype(np.float32)
    mean = np.zeros(5, np.float32)
    var = np.zeros(5, np.float32)
    getattr(L, "_Z8mean_cpuPfiiiS_")(fp(xb), 3, 5, 42, fp(mean))
    getattr(L, "_Z12variance_cpuPfS_iiiS_")(fp(xb), fp(mean), 3, 5, 42, fp(var))
    xn = xb.copy()
    getattr(L, "_Z13normalize_cpuPfS_S_iii")(fp(xn), fp(mean), fp(var), 3, 5, 42)
    out["bn_x"], out["bn_mean"], out["bn_var"], out["bn_norm"] = xb, mean, var, xn
    np.savez_compressed(os.path.join(GOLD, "ops.npz"), **out)
    print("ops.npz:", len(out), "arrays")


MAXPOOL_CFG = """[net]
batch=1
subdivisions=1
width=19
height=19
channels=6
[maxpool]
size={size}
stride={stride}
"""


def gen_maxpool():
    """maxpool through the reference's train-mode generic loop (definition-correct,
    SURVEY quirk 5) incl. argmax indexes."""
    rng = np.random.default_rng(5)
    out = {}
    cases = [(2, 2), (5, 1), (9, 1), (13, 1), (3, 2)]
    out["cases"] = np.array(cases, np.int32)
    x = np.round(rng.uniform(-4, 4, (1, 6 * 19 * 19))).astype(np.float32)
    out["x"] = x
    for i, (size, stride) in enumerate(cases):
        path = "/tmp/_dk_maxpool.cfg"
        with open(path, "w") as f:
            f.write(MAXPOOL_CFG.format(size=size, stride=stride))
        rn = reflib.RefNet(path, None, train=True)
        rn.L.ref_forward_train(rn.p, fp(x), None)
        inf = rn.info(0)
        out[f"y_{i}"] = rn.output(0)
        idx = rn.L.ref_layer_indexes(rn.p, 0)
        out[f"idx_{i}"] = np.ctypeslib.as_array(idx, shape=(inf["outputs"],)).copy()
        rn.close()
    np.savez_compressed(os.path.join(GOLD, "maxpool.npz"), **out)
    print("maxpool.npz")


def pick_threshold(obj, lo=0.3, hi=0.7, guard=2e-3):
    """A detection threshold with no objectness/probability within `guard` of it."""
    v = np.sort(obj[(obj > lo) & (obj < hi)])
    if v.size == 0:
        return 0.5
    edges = np.concatenate([[lo], v, [hi]])
    gaps = np.diff(edges)
    i = int(np.argmax(gaps))
    assert gaps[i] > 2 * guard, "no gap wide enough for a guard-banded threshold"
    return float(np.float32((edges[i] + edges[i + 1]) / 2))


def gen_net(name, full_heads):
    cfg = os.path.join(ROOT, "cfg", name + ".cfg")
    net = O.parse_cfg(cfg)
    convs = [(l.n, l.c // l.groups, l.size, l.batch_normalize) for l in net.layers
             if l.type == O.CONVOLUTIONAL]
    wpath = f"/tmp/_dk_{name}.weights"
    synth.write_weights(wpath, convs, seed=2024)
    x = synth.make_input(1, net.c, net.h, net.w, seed=12345)
    rn = reflib.RefNet(cfg, wpath, train=False)
    rn.predict(x)
    out = {"n_layers": np.int32(rn.n), "weights_bytes": np.int64(os.path.getsize(wpath))}
    sums, samples, types, outs = [], [], [], []
    heads = {}
    for i in range(rn.n):
        o = rn.output(i)
        inf = rn.info(i)
        types.append(inf["type"])
        outs.append(inf["outputs"])
        sums.append((np.sum(o, dtype=np.float64), np.sum(o.astype(np.float64) ** 2)))
        idx = np.linspace(0, o.size - 1, 64).astype(np.int64)
        samples.append(o[idx])
        if inf["type"] == O.YOLO:
            heads[i] = o
    out["layer_types"] = np.array(types, np.int32)
    out["layer_outputs"] = np.array(outs, np.int64)
    out["layer_sums"] = np.array(sums, np.float64)
    out["layer_samples"] = np.array(samples, np.float32)
    allobj = []
    for i, o in heads.items():
        inf = rn.info(i)
        if full_heads:
            out[f"head_{i}"] = o
        else:
            out[f"head_{i}_sub16"] = o[::16].copy()
        wh = inf["out_h"] * inf["out_w"]
        v = o.reshape(3, 5 + inf["classes"], wh)
        allobj.append(v[:, 4, :].ravel())
        allobj.append((v[:, 4:5, :] * v[:, 5:, :]).ravel())
    thresh = pick_threshold(np.concatenate(allobj))
    dets = rn.boxes(thresh)
    out["thresh"] = np.float32(thresh)
    out["num_dets"] = np.int32(len(dets))
    # ids (layer, anchor, row, col) + best class, from the oracle (bit-identical
    # to the reference; asserted here)
    onet = O.load_network(cfg, wpath, batch=1)
    O.forward(onet, x)
    od, oid = O.get_boxes(onet, thresh)
    assert np.array_equal(od, dets), "oracle and reference detections differ"
    for i, l in enumerate(onet.layers):
        assert np.array_equal(l.output.ravel(), rn.output(i)), f"oracle != reference at layer {i}"
    out["det_ids"] = oid
    out["det_box_obj"] = dets[:, :5].copy()
    out["det_best_class"] = np.argmax(dets[:, 5:], 1).astype(np.int32) if len(dets) else np.zeros(0, np.int32)
    out["det_best_prob"] = np.max(dets[:, 5:], 1) if len(dets) else np.zeros(0, np.float32)
    out["det_nonzero_classes"] = np.count_nonzero(dets[:, 5:], 1).astype(np.int32)
    # NmsSort (src/box.cpp:393-419) on the reference's own detection list
    nd = dets.copy()
    classes = dets.shape[1] - 5
    last = [l for l in onet.layers if l.type == O.YOLO][-1]
    rn.L.ref_nms_sort(fp(nd), len(nd), classes, 0.45, last.nms_kind, last.beta_nms)
    out["nms_thresh"] = np.float32(0.45)
    out["nms_kind_beta"] = np.array([last.nms_kind, last.beta_nms], np.float32)
    out["nms_box_obj"] = nd[:, :5].copy()
    keep = nd[:, 5:] > 0
    out["nms_kept_per_det"] = keep.sum(1).astype(np.int32)
    out["nms_kept_prob_sum"] = np.where(keep, nd[:, 5:], 0).sum(1, dtype=np.float64)
    rn.close()
    np.savez_compressed(os.path.join(GOLD, f"net_{name}.npz"), **out)
    print(f"net_{name}.npz: layers {rn.n}, dets {len(dets)} at thresh {thresh:.6f}")


def gen_gaussian():
    """[Gaussian_yolo] heads, inference, through the REAL reference: every layer's output of
    cfg/gaussian-test.cfg and the detection list (box, objectness, probabilities weighted by 1 - mean
    uncertainty) at a guard-banded threshold; the oracle is asserted bit-identical -> gaussian-test.npz."""
    name = "gaussian-test"
    cfg = os.path.join(ROOT, "cfg", name + ".cfg")
    net = O.parse_cfg(cfg)
    wpath = "/tmp/_dk_gaussian.weights"
    synth.write_weights_layers(wpath, synth.weight_layers_of(net), seed=2024)
    x = synth.make_input(1, net.c, net.h, net.w, seed=12345)
    rn = reflib.RefNet(cfg, wpath, train=False)
    rn.predict(x)
    onet = O.load_network(cfg, wpath, batch=1)
    O.forward(onet, x)
    out = {"n_layers": np.int32(rn.n), "weights_bytes": np.int64(os.path.getsize(wpath))}
    allv = []
    for i, l in enumerate(onet.layers):
        inf = rn.info(i)
        assert (inf["type"], inf["outputs"]) == (l.type, l.outputs), (i, inf)
        r = rn.output(i)
        assert np.array_equal(r, l.output.ravel()), f"gaussian: oracle != reference at layer {i}"
        if l.type == O.GAUSSIAN_YOLO:
            out[f"head_{i}"] = r
            v = r.reshape(l.n, 9 + l.classes, l.w * l.h)
            uc = (v[:, 1] + v[:, 3] + v[:, 5] + v[:, 7]) / 4
            allv.append(v[:, 8].ravel())
            allv.append((v[:, 8:9] * v[:, 9:] * (1 - uc[:, None])).ravel())
    thresh = pick_threshold(np.concatenate(allv), lo=0.2, hi=0.6, guard=1e-4)
    dets = rn.boxes(thresh)
    od, oid = O.get_boxes(onet, thresh)
    assert len(dets) > 50 and np.array_equal(od, dets), "gaussian: oracle and reference detections differ"
    out["thresh"] = np.float32(thresh)
    out["dets"] = dets
    out["det_ids"] = oid
    out["dets_uc"] = O.get_gaussian_boxes(onet, thresh)[:, -4:]
    rn.close()
    np.savez_compressed(os.path.join(GOLD, f"{name}.npz"), **out)
    print(f"{name}.npz: layers {rn.n}, dets {len(dets)} at thresh {thresh:.6f}")


def gen_gaussianloss():
    """The reference's Gaussian-YOLO loss (ForwardGaussianYoloLayer train branch, host C++) on cfg/gaussian-test.cfg:
    sparse deltas and cost of both heads for b = 2 with 4 truths/image (first head: mse boxes, second head: giou boxes,
    iou_thresh, max_delta, label smoothing, uc_normalizer) -> gaussianloss.npz."""
    name = "gaussian-test"
    B = 2
    cfg = "/tmp/_dk_gloss.cfg"
    open(cfg, "w").write(open(os.path.join(ROOT, "cfg", name + ".cfg")).read().replace("batch=1", "batch=%d" % B, 1))
    net = O.parse_cfg(cfg)
    wpath = "/tmp/_dk_gaussian.weights"
    synth.write_weights_layers(wpath, synth.weight_layers_of(net), seed=2024)
    x = synth.make_input(B, net.c, net.h, net.w, seed=12345)
    boxes = [(.3, .4, .2, .3, 1), (.6, .5, .4, .35, 3), (.8, .2, .1, .15, 0), (.5, .5, .9, .8, 2)]
    truth = np.zeros((B, 90 * 5), np.float32)
    for b in range(B):
        for t, box in enumerate(boxes[b:] + boxes[:b]):
            truth[b, t * 5:(t + 1) * 5] = box
    rn = reflib.RefNet(cfg, wpath, train=True)
    assert rn.batch == B
    rn.L.ref_forward_train(rn.p, fp(x), fp(truth))
    onet = O.load_network_train(cfg, wpath, None)
    O.forward_train(onet, x)
    out = {"truth": truth, "batch": np.int32(B)}
    for i, l in enumerate(onet.layers):
        assert np.array_equal(rn.output(i), l.output.ravel()), f"gaussian train forward: oracle != reference at {i}"
        if l.type != O.GAUSSIAN_YOLO:
            continue
        d = rn.arr(i, 6, l.batch * l.outputs)
        nz = np.flatnonzero(d)
        out[f"delta_{i}_idx"] = nz.astype(np.int64)
        out[f"delta_{i}_val"] = d[nz]
        out[f"cost_{i}"] = np.float32(rn.L.ref_layer_cost(rn.p, i))
        print("gaussian head", i, "nonzero deltas", nz.size, "cost", rn.L.ref_layer_cost(rn.p, i))
    rn.close()
    np.savez_compressed(os.path.join(GOLD, "gaussianloss.npz"), **out)
    print("gaussianloss.npz")


def main():
    assert reflib.available("canon"), "build oracle/_ref first: make -C oracle ref"
    os.makedirs(GOLD, exist_ok=True)
    gen_ops()
    gen_maxpool()
    gen_net("yolov4-tiny", full_heads=True)
    gen_net("yolov4", full_heads=False)
    gen_net("yolov4-csp", full_heads=False)
    gen_train()
    gen_yololoss()
    gen_extra()
    gen_map()
    gen_grads()
    gen_gaussian()
    gen_gaussianloss()


def gen_train(name="yolov4-tiny", B=2):
    """One train step of the real reference (forward with batch-norm statistics,
    yolo loss, backward, SGD update) -> train_<cfg>.npz.  The yolo-layer deltas
    (the loss gradient, host C++ in the reference too) are stored sparsely so that
    the oracle's / the HIP path's backward can be driven from them."""
    cfg_txt = open(os.path.join(ROOT, "cfg", name + ".cfg")).read().replace("batch=64", "batch=%d" % B)
    cfg = f"/tmp/_dk_{name}_b{B}.cfg"
    open(cfg, "w").write(cfg_txt)
    net = O.parse_cfg(cfg)
    convs = [(l.n, l.c // l.groups, l.size, l.batch_normalize) for l in net.layers if l.type == O.CONVOLUTIONAL]
    wpath = f"/tmp/_dk_{name}.weights"
    synth.write_weights(wpath, convs, seed=2024)
    x = synth.make_input(B, net.c, net.h, net.w, seed=12345)
    truth = np.zeros((B, 90 * 5), np.float32)
    for b in range(B):
        for t, box in enumerate([(.3, .4, .2, .3, 1), (.6, .5, .4, .35, 17), (.8, .2, .1, .15, 60)]):
            truth[b, t * 5:(t + 1) * 5] = box
    rn = reflib.RefNet(cfg, wpath, train=True)
    rn.L.ref_set_max_iter(rn.p, 1000)
    cost = rn.L.ref_train_datum(rn.p, fp(x), fp(truth))
    out = {"batch": np.int32(B), "truth": truth, "cost": np.float32(cost)}
    onet = O.load_network_train(cfg, wpath, None)
    O.forward_train(onet, x)
    for i, l in enumerate(onet.layers):
        assert np.array_equal(rn.output(i), l.output.ravel()), f"train forward: oracle != reference at {i}"
        if l.type == O.YOLO:
            d = rn.arr(i, 6, l.batch * l.outputs)
            nz = np.flatnonzero(d)
            out[f"yolo_{i}_delta_idx"] = nz.astype(np.int64)
            out[f"yolo_{i}_delta_val"] = d[nz]
            l.delta[...] = d.reshape(l.delta.shape)
    O.backward(onet)

    def summ(a):
        idx = np.linspace(0, a.size - 1, 16).astype(np.int64)
        return np.concatenate([[np.sum(a, dtype=np.float64), np.sum(a.astype(np.float64) ** 2)], a[idx].astype(np.float64)])
    rows = []
    for i, l in enumerate(onet.layers):
        if l.type != O.CONVOLUTIONAL:
            continue
        for which, nm, n in ((7, "weight_updates", l.nweights), (8, "bias_updates", l.n), (9, "scale_updates", l.n)):
            r = rn.arr(i, which, n)
            o = getattr(l, nm, None)
            if r is None or o is None:
                continue
            assert np.array_equal(r, o), f"train backward: oracle != reference at {i} {nm}"
            rows.append(np.concatenate([[i, which], summ(r)]))
        rows.append(np.concatenate([[i, 6], summ(rn.arr(i, 6, l.batch * l.outputs))]))
    out["grad_summaries"] = np.array(rows)
    # SGD update (UpdateNetwork: lr from GetCurrLr at iteration 1 of burn-in)
    rn.L.ref_update(rn.p)
    lr = float(rn.L.ref_curr_lr(rn.p))
    out["lr"] = np.float32(lr)
    O.update(onet, onet.batch * onet.subdiv, lr, onet.momentum, onet.decay)
    rows = []
    for i, l in enumerate(onet.layers):
        if l.type != O.CONVOLUTIONAL:
            continue
        r = rn.arr(i, 1, l.nweights)
        assert np.array_equal(r, l.weights), f"update: oracle != reference at {i}"
        assert np.array_equal(rn.arr(i, 2, l.n), l.biases)
        rows.append(np.concatenate([[i], summ(r)]))
    out["updated_weight_summaries"] = np.array(rows)
    rn.close()
    np.savez_compressed(os.path.join(GOLD, f"train_{name}.npz"), **out)
    print(f"train_{name}.npz: cost {cost:.4f}, lr {lr:.3e}")


def gen_yololoss():
    """The reference's YOLO loss (ForwardYoloLayer train branch, host C++) on the three
    nets: sparse deltas and cost of every yolo layer for b = 2 with 5 truths/image."""
    out = {}
    boxes = [(.3, .4, .2, .3, 1), (.6, .5, .4, .35, 17), (.8, .2, .1, .15, 60), (.05, .93, .08, .1, 3),
             (.5, .5, .9, .8, 79)]
    for name in ("yolov4-tiny", "yolov4", "yolov4-csp"):
        B = 2
        cfg_txt = open(os.path.join(ROOT, "cfg", name + ".cfg")).read()
        cfg_txt = cfg_txt.replace("batch=64", "batch=%d" % B).replace("subdivisions=8", "subdivisions=1")
        cfg = f"/tmp/_dk_loss_{name}.cfg"
        open(cfg, "w").write(cfg_txt)
        net = O.parse_cfg(cfg)
        convs = [(l.n, l.c // l.groups, l.size, l.batch_normalize) for l in net.layers if l.type == O.CONVOLUTIONAL]
        wpath = f"/tmp/_dk_{name}.weights"
        synth.write_weights(wpath, convs, seed=2024)
        x = synth.make_input(B, net.c, net.h, net.w, seed=12345)
        truth = np.zeros((B, 90 * 5), np.float32)
        for b in range(B):
            for t, box in enumerate(boxes[b:] + boxes[:b]):
                truth[b, t * 5:(t + 1) * 5] = box
        rn = reflib.RefNet(cfg, wpath, train=True)
        assert rn.batch == B
        rn.L.ref_forward_train(rn.p, fp(x), fp(truth))
        out[name + "_truth"] = truth
        for i in range(rn.n):
            inf = rn.info(i)
            if inf["type"] != O.YOLO:
                continue
            d = rn.arr(i, 6, inf["batch"] * inf["outputs"])
            nz = np.flatnonzero(d)
            out[f"{name}_{i}_idx"] = nz.astype(np.int64)
            out[f"{name}_{i}_val"] = d[nz]
            out[f"{name}_{i}_cost"] = np.float32(rn.L.ref_layer_cost(rn.p, i))
            print(name, "yolo", i, "nonzero deltas", nz.size, "cost", rn.L.ref_layer_cost(rn.p, i))
        rn.close()
    np.savez_compressed(os.path.join(GOLD, "yololoss.npz"), **out)
    print("yololoss.npz")


def gen_grads():
    """gradient_array of the REAL reference for the 12 rarer activations and swish on the outputs of
    the activation grid already in ops.npz -> ops_grad.npz (delta = 1 going in)."""
    L = reflib.lib("canon")
    ops = np.load(os.path.join(GOLD, "ops.npz"))
    out = {}
    for name, a in (("relu6", 2), ("relie", 3), ("ramp", 5), ("tanh", 6), ("plse", 7), ("elu", 9), ("loggy", 10),
                    ("hardtan", 12), ("lhtan", 13), ("selu", 14), ("gelu", 15), ("relu", 1)):
        y = np.ascontiguousarray(ops["act_" + name])
        d = np.ones_like(y)
        L.gradient_array(fp(y), y.size, a, fp(d))
        out["grad_" + name] = d
    # swish: gradient_array_swish(x = swish output, sigmoid)
    grid = np.ascontiguousarray(ops["act_grid"])
    x = grid.copy()
    sig = np.zeros_like(x)
    y = np.zeros_like(x)
    L.activate_array_swish.argtypes = [FP, C.c_int, FP, FP]
    L.activate_array_swish(fp(x), x.size, fp(sig), fp(y))
    d = np.ones_like(grid)
    L.gradient_array_swish.argtypes = [FP, C.c_int, FP, FP]
    L.gradient_array_swish(fp(y), y.size, fp(sig), fp(d))
    out["grad_swish"], out["swish_sigmoid"] = d, sig
    np.savez_compressed(os.path.join(GOLD, "ops_grad.npz"), **out)
    print("ops_grad.npz:", sorted(out))


def gen_train_big(name="yolov4", B=8):
    """BASELINE config C4 per GPU: one yolov4 608x608 train step at batch 8 through the REAL reference
    -> train_<cfg>_b<B>.npz with SUMMARIES only (per layer: sum, sum of squares and 64 strided samples
    of the train-mode output; per conv: the same of weight/scale updates and delta; cost; yolo deltas
    sparse).  The reference run is finished and freed before the oracle runs (the two do not fit in
    memory together at this size); the oracle must reproduce every summary bit for bit."""
    import gc
    cfg_txt = open(os.path.join(ROOT, "cfg", name + ".cfg")).read().replace("batch=64", "batch=%d" % B).replace("subdivisions=8", "subdivisions=1")
    cfg = f"/tmp/_dk_{name}_b{B}.cfg"
    open(cfg, "w").write(cfg_txt)
    net = O.parse_cfg(cfg)
    convs = [(l.n, l.c // l.groups, l.size, l.batch_normalize) for l in net.layers if l.type == O.CONVOLUTIONAL]
    wpath = f"/tmp/_dk_{name}.weights"
    synth.write_weights(wpath, convs, seed=2024)
    x = synth.make_input(B, net.c, net.h, net.w, seed=12345)
    truth = np.zeros((B, 90 * 5), np.float32)
    boxes = [(.3, .4, .2, .3, 1), (.6, .5, .4, .35, 17), (.8, .2, .1, .15, 60), (.05, .93, .08, .1, 3), (.5, .5, .9, .8, 79)]
    for b in range(B):
        for t, box in enumerate(boxes[b % 5:] + boxes[:b % 5]):
            truth[b, t * 5:(t + 1) * 5] = box

    def summ(a, k=64):
        a = np.asarray(a).ravel()
        idx = np.linspace(0, a.size - 1, k).astype(np.int64)
        return np.concatenate([[np.sum(a, dtype=np.float64), np.sum(a.astype(np.float64) ** 2)], a[idx].astype(np.float64)])
    rn = reflib.RefNet(cfg, wpath, train=True)
    assert rn.batch == B
    rn.L.ref_set_max_iter(rn.p, 1000)
    cost = rn.L.ref_train_datum(rn.p, fp(x), fp(truth))
    out = {"batch": np.int32(B), "truth": truth, "cost": np.float32(cost)}
    fwd, grads, ydelta = [], [], {}
    for i in range(rn.n):
        inf = rn.info(i)
        fwd.append(np.concatenate([[i], summ(rn.output(i))]))
        if inf["type"] == O.YOLO:
            d = rn.arr(i, 6, inf["batch"] * inf["outputs"])
            nz = np.flatnonzero(d)
            out[f"yolo_{i}_delta_idx"], out[f"yolo_{i}_delta_val"] = nz.astype(np.int64), d[nz]
            ydelta[i] = d
        if inf["type"] == O.CONVOLUTIONAL:
            for which, n in ((7, inf["nweights"]), (8, inf["n"]), (9, inf["n"])):
                r = rn.arr(i, which, n)
                if r is not None and not (which == 9 and not inf["batch_normalize"]):
                    grads.append(np.concatenate([[i, which], summ(r, 16)]))
            grads.append(np.concatenate([[i, 6], summ(rn.arr(i, 6, inf["batch"] * inf["outputs"]), 16)]))
    out["fwd_summaries"], out["grad_summaries"] = np.array(fwd), np.array(grads)
    rn.close()
    del rn
    gc.collect()
    print("reference done, cost", cost, flush=True)
    np.savez_compressed(os.path.join(GOLD, f"train_{name}_b{B}.npz"), **out)   # kept even if the oracle check below is interrupted
    onet = O.load_network_train(cfg, wpath, None)
    O.forward_train(onet, x)
    for i, l in enumerate(onet.layers):
        assert np.array_equal(summ(l.output), out["fwd_summaries"][i][1:]), f"big train forward: oracle != reference at {i}"
        if l.type == O.YOLO:
            l.delta[...] = ydelta[i].reshape(l.delta.shape)
    print("oracle forward matches", flush=True)
    O.backward(onet)
    gi = {(int(r[0]), int(r[1])): r[2:] for r in out["grad_summaries"]}
    for i, l in enumerate(onet.layers):
        if l.type != O.CONVOLUTIONAL:
            continue
        assert np.array_equal(summ(l.weight_updates, 16), gi[(i, 7)]), f"big train backward: weight_updates at {i}"
        if l.batch_normalize:
            assert np.array_equal(summ(l.scale_updates, 16), gi[(i, 9)]), f"big train backward: scale_updates at {i}"
        assert np.array_equal(summ(l.delta, 16), gi[(i, 6)]), f"big train backward: delta at {i}"
    print(f"train_{name}_b{B}.npz: cost {cost:.4f}; oracle == reference on every summary", flush=True)
    # Second forward with the batch statistics accumulated in double (analysis variant, see
    # orc_set_bn_stats_f64): separates the reference's own fp32 summation error from the error of
    # an implementation that reduces in higher precision.
    del onet
    gc.collect()
    O.lib().orc_set_bn_stats_f64(1)
    onet = O.load_network_train(cfg, wpath, None)
    O.forward_train(onet, x)
    O.lib().orc_set_bn_stats_f64(0)
    f64 = np.array([np.concatenate([[i], summ(l.output)]) for i, l in enumerate(onet.layers)])
    out["fwd_summaries_f64stats"] = f64
    shift = []
    for a, b in zip(out["fwd_summaries"], f64):
        rms = np.sqrt(b[2] / max(1, onet.layers[int(a[0])].batch * onet.layers[int(a[0])].outputs))
        shift.append(np.abs(a[3:] - b[3:]).max() / rms if rms > 0 else 0.0)
    out["ref_vs_f64stats_max_over_rms"] = np.array(shift)
    np.savez_compressed(os.path.join(GOLD, f"train_{name}_b{B}.npz"), **out)
    print("reference vs fp64-statistics oracle: worst sample shift / rms = %.3g (layer %d)" % (max(shift), int(np.argmax(shift))))


se_cfgs = synth.se_cfgs


def gen_extra():
    """Sibling-cfg layer kinds ([batchnorm], [avgpool], [scale_channels], [dropout]) through the REAL
    reference: every layer's inference output (b=1), and one train step (forward with batch
    statistics, backward, update) of the cfg without [dropout]; the oracle is asserted bit-identical
    to the reference on all of it while the fixture is written -> extra_se-test.npz."""
    inf, tr = se_cfgs()
    out = {}
    onet = O.parse_cfg(inf)
    wpath = "/tmp/_dk_se.weights"
    synth.write_weights_layers(wpath, synth.weight_layers_of(onet), seed=2024)
    assert os.path.getsize(wpath) == O.weights_file_size(onet)
    # ---- inference, batch 1 (the reference forces it)
    x = synth.make_input(1, onet.c, onet.h, onet.w, seed=12345)
    rn = reflib.RefNet(inf, wpath)
    assert rn.n == onet.n == 10
    rn.predict(x)
    onet = O.load_network(inf, wpath, batch=1)
    O.forward(onet, x)
    for i, l in enumerate(onet.layers):
        inff = rn.info(i)
        assert (inff["type"], inff["outputs"], inff["out_c"], inff["out_h"], inff["out_w"]) == \
            (l.type, l.outputs, l.out_c, l.out_h, l.out_w), (i, inff)
        r = rn.output(i)
        assert np.array_equal(r, l.output.ravel()), f"extra inference: oracle != reference at layer {i}"
        out[f"inf_out_{i}"] = r
    rn.close()
    # ---- one train step, batch 2, no dropout
    B = 2
    tnet = O.parse_cfg(tr)
    x2 = synth.make_input(B, tnet.c, tnet.h, tnet.w, seed=777)
    truth = np.zeros((B, 90 * 5), np.float32)
    for b in range(B):
        for t, box in enumerate([(.3, .4, .2, .3, 1), (.6, .5, .4, .35, 0), (.8, .2, .1, .15, 1)]):
            truth[b, t * 5:(t + 1) * 5] = box
    twpath = "/tmp/_dk_se_train.weights"
    synth.write_weights_layers(twpath, synth.weight_layers_of(tnet), seed=2024)
    wpath = twpath
    rn = reflib.RefNet(tr, wpath, train=True)
    assert rn.batch == B
    rn.L.ref_set_max_iter(rn.p, 1000)
    cost = rn.L.ref_train_datum(rn.p, fp(x2), fp(truth))
    out["train_x_seed"], out["train_truth"], out["train_cost"] = np.int32(777), truth, np.float32(cost)
    onet = O.load_network_train(tr, wpath, None)
    O.forward_train(onet, x2)
    for i, l in enumerate(onet.layers):
        assert np.array_equal(rn.output(i), l.output.ravel()), f"extra train forward: oracle != reference at {i}"
        out[f"train_out_{i}"] = l.output.ravel().copy()
        if l.type == O.YOLO:
            d = rn.arr(i, 6, l.batch * l.outputs)
            out[f"train_yolo_delta_{i}"] = d.copy()
            l.delta[...] = d.reshape(l.delta.shape)
    O.backward(onet)
    for i, l in enumerate(onet.layers):
        for which, nm, n in ((7, "weight_updates", getattr(l, "nweights", 0)), (8, "bias_updates", getattr(l, "n", 0)),
                             (9, "scale_updates", getattr(l, "n", 0))):
            o = getattr(l, nm, None)
            r = rn.arr(i, which, n) if (o is not None and n) else None
            if r is None or o is None:
                continue
            assert np.array_equal(r, o), f"extra train backward: oracle != reference at {i} {nm}"
            out[f"train_{nm}_{i}"] = r
        if l.type != O.YOLO and l.type != O.DROPOUT:
            r = rn.arr(i, 6, l.batch * l.outputs)
            assert np.array_equal(r, l.delta.ravel()), f"extra train delta: oracle != reference at {i}"
            out[f"train_delta_{i}"] = r
    rn.L.ref_update(rn.p)
    lr = float(rn.L.ref_curr_lr(rn.p))
    out["train_lr"] = np.float32(lr)
    O.update(onet, onet.batch * onet.subdiv, lr, onet.momentum, onet.decay)
    for i, l in enumerate(onet.layers):
        if l.type == O.CONVOLUTIONAL:
            assert np.array_equal(rn.arr(i, 1, l.nweights), l.weights), f"extra update: weights at {i}"
            assert np.array_equal(rn.arr(i, 2, l.n), l.biases)
        if l.type == O.BATCHNORM:
            assert np.array_equal(rn.arr(i, 2, l.c), l.biases), f"extra update: bn biases at {i}"
            assert np.array_equal(rn.arr(i, 3, l.c), l.scales), f"extra update: bn scales at {i}"
            out[f"train_bn_scales_{i}"] = l.scales.copy()
    rn.close()
    np.savez_compressed(os.path.join(GOLD, "extra_se-test.npz"), **out)
    print("extra_se-test.npz: cost %.5f lr %.3e, %d arrays" % (cost, lr, len(out)))


def gen_map(name="yolov4-tiny", K=4):
    """Evaluator fixture: the REAL reference's post-NMS detections (NetworkPredict,
    GetNetworkBoxes, NmsSort through oracle/_ref) on K seeded u8 images, a label set derived from
    them, and the mAP the oracle's restatement of ValidateDetector's arithmetic gives on those
    detections -> map_<cfg>.npz.  (detector.cpp itself needs OpenCV and is not in the _ref build.)"""
    from oracle import orc_map
    cfg = os.path.join(ROOT, "cfg", name + ".cfg")
    net = O.parse_cfg(cfg)
    convs = [(l.n, l.c // l.groups, l.size, l.batch_normalize) for l in net.layers if l.type == O.CONVOLUTIONAL]
    wpath = f"/tmp/_dk_{name}.weights"
    synth.write_weights(wpath, convs, seed=2024)
    rn = reflib.RefNet(cfg, wpath)
    seeds = [4100 + i for i in range(K)]
    xs = [synth.u8_to_chw(synth.make_u8_image(net.w, net.h, sd)) for sd in seeds]
    # class probability (objectness x class score) of every predictor and class over the K images ->
    # a guard-banded threshold that lets ~250 (predictor, class) pairs per image through.  (With the
    # synthetic weights every score hovers around 0.5, so the reference's default .005 would pass all
    # 2535 x 80 pairs; only the sparse upper tail of the distribution has gaps wide enough for a
    # guard band.)
    probs = []
    for x in xs:
        rn.predict(x.reshape(1, -1))
        for i in range(rn.n):
            inf = rn.info(i)
            if inf["type"] == O.YOLO:
                o = rn.output(i).reshape(3, 5 + inf["classes"], -1)
                probs.append((o[:, 4:5, :] * o[:, 5:, :]).ravel())
    srt = np.sort(np.concatenate(probs))[::-1]
    lo, hi = 150 * K, 350 * K
    gaps = srt[lo:hi] - srt[lo + 1:hi + 1]
    j = lo + int(np.argmax(gaps))
    thresh = float((srt[j] + srt[j + 1]) / 2)
    print("threshold gap", gaps.max(), "rank", j)
    assert gaps.max() > 2e-5, "no guard band around the threshold"
    last = [l for l in net.layers if l.type == O.YOLO][-1]
    classes = last.classes
    nms = 0.45
    rng = np.random.default_rng(99)
    dets_all, gts_all = [], []
    for x in xs:
        rn.predict(x.reshape(1, -1))
        d = rn.boxes(thresh)
        nd = np.ascontiguousarray(d)
        rn.L.ref_nms_sort(nd.ctypes.data_as(FP), len(nd), classes, nms, last.nms_kind, last.beta_nms)
        dd = np.concatenate([nd[:, :4], nd[:, 5:]], 1)   # drop objectness: [x, y, w, h, prob...]
        dd = np.ascontiguousarray(dd[(dd[:, 4:] != 0).any(1)])   # detections without a surviving class add nothing
        dets_all.append(dd)
        best = np.argsort(-dd[:, 4:].max(1))[:6]
        g = []
        for r, k in enumerate(best):
            b = dd[k, :4].copy()
            if r % 3 == 2:
                b[2:] *= np.float32(0.9)                  # still IoU ~0.8 with the detection
            g.append([float(np.argmax(dd[k, 4:]))] + list(b))
        for _ in range(2):                                # labels nothing will match
            g.append([float(rng.integers(0, classes)), .05 + .02 * rng.uniform(), .9, .03, .04])
        gts_all.append(np.array(g, np.float32))
    rn.close()
    m, aps = orc_map.mean_average_precision(dets_all, gts_all, classes, 0.5)
    out = {"seeds": np.array(seeds, np.int32), "thresh": np.float32(thresh), "nms": np.float32(nms), "map": np.float64(m),
           "ap": np.array(aps, np.float64), "n_dets": np.array([len(d) for d in dets_all], np.int32),
           "dets": np.concatenate(dets_all), "n_gts": np.array([len(g) for g in gts_all], np.int32),
           "gts": np.concatenate(gts_all)}
    np.savez_compressed(os.path.join(GOLD, f"map_{name}.npz"), **out)
    print(f"map_{name}.npz: thresh {thresh:.6f}, dets/image {[len(d) for d in dets_all]}, mAP {m:.6f}")


LR_POLICIES = {
    # name -> (the [net] lines that replace the tiny cfg's schedule, max_iter set through ref_set_max_iter)
    "constant": "learning_rate=0.00261\nburn_in=1000\npower=4\npolicy=constant\n",
    "step": "learning_rate=0.01\nburn_in=0\npolicy=step\nstep=400\nscale=.5\n",
    "steps": "learning_rate=0.00261\nburn_in=1000\npolicy=steps\nsteps=.8,.9\nscales=.1,.1\n",
    "exp": "learning_rate=0.01\nburn_in=100\npower=2\npolicy=exp\ngamma=.999\n",
    "poly": "learning_rate=0.01\nburn_in=0\npower=4\npolicy=poly\n",
    "sig": "learning_rate=0.01\nburn_in=0\npolicy=sigmoid\ngamma=.01\nstep=2500\n",
    # (the reference dereferences steps= / scales= for sgdr too: a cfg without them crashes its parser)
    "sgdr": "learning_rate=0.01\nlearning_rate_min=.0001\nburn_in=200\npolicy=sgdr\nsgdr_cycle=700\nsgdr_mult=2\nsteps=.5\nscales=1\n",
}
LR_MAX_ITER = 5000
LR_ITERS = [0, 1, 2, 50, 99, 100, 101, 199, 200, 399, 400, 401, 699, 700, 701, 999, 1000, 1001, 1500, 2099, 2100, 2101,
            2499, 2500, 2501, 3999, 4000, 4001, 4499, 4500, 4501, 4899, 4900, 4999, 5000]


def lr_cfg_text(policy):
    """yolov4-tiny with its learning-rate schedule lines replaced (shared with the test)."""
    import re
    txt = open(os.path.join(ROOT, "cfg", "yolov4-tiny.cfg")).read()
    for key in ("learning_rate", "burn_in", "policy", "steps", "scales"):
        txt = re.sub(r"(?m)^%s=.*\n" % key, "", txt, count=1)
    txt = re.sub(r"batch=\d+", "batch=1", txt, count=1)
    return txt.replace("[net]\n", "[net]\n" + LR_POLICIES[policy], 1)


def gen_lr():
    """GetCurrLr (network.cpp:32-84) of the real reference for every deterministic policy over burn-in, the step
    boundaries and the warm restarts -> tests/golden/lr_schedule.npz."""
    import tempfile
    L = reflib.lib()
    L.ref_set_curr_iter.argtypes = [C.c_void_p, C.c_longlong]
    out = {"iters": np.array(LR_ITERS, np.int64), "max_iter": np.array(LR_MAX_ITER)}
    with tempfile.TemporaryDirectory() as d:
        for pol in LR_POLICIES:
            cfg = os.path.join(d, pol + ".cfg")
            open(cfg, "w").write(lr_cfg_text(pol))
            net = reflib.RefNet(cfg, None, train=False)
            L.ref_set_max_iter(net.p, LR_MAX_ITER)
            v = []
            for it in LR_ITERS:
                L.ref_set_curr_iter(net.p, it)
                v.append(L.ref_curr_lr(net.p))
            out["lr_" + pol] = np.array(v, np.float32)
            print(pol, v[:4], v[-3:])
    np.savez_compressed(os.path.join(GOLD, "lr_schedule.npz"), **out)


class DetStruct(C.Structure):
    """Detection (src/box.h:68-85)"""
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("w", C.c_float), ("h", C.c_float), ("classes", C.c_int),
                ("prob", FP), ("mask", FP), ("objectness", C.c_float), ("sort_class", C.c_int), ("uc", FP),
                ("points", C.c_int)]


def json_case():
    """Synthetic detections for Detection2Json: 7 boxes x 5 classes, a `dont_show` class, values around the fixed
    0.005 threshold, one box with nothing above it."""
    rng = np.random.default_rng(7)
    n, classes = 7, 5
    box = rng.uniform(0.05, 0.95, (n, 4)).astype(np.float32)
    prob = rng.uniform(0, 1, (n, classes)).astype(np.float32)
    prob[prob < 0.55] = 0
    prob[1, 2] = np.float32(0.005)      # not above the threshold
    prob[2, 0] = np.float32(0.0050001)
    prob[3, :] = 0
    prob[4, 3] = np.float32(0.75)       # class 3 is dont_show
    names = [b"person", b"traffic light", b"dog", b"dont_show_me", b"kite"]
    return box, prob, names


def call_detection2json(L, box, prob, names, frame_id, filename):
    n, classes = prob.shape
    arr = (DetStruct * n)()
    keep = []
    for i in range(n):
        pr = np.ascontiguousarray(prob[i])
        keep.append(pr)
        arr[i].x, arr[i].y, arr[i].w, arr[i].h = [float(v) for v in box[i]]
        arr[i].classes = classes
        arr[i].prob = pr.ctypes.data_as(FP)
    nm = (C.c_char_p * classes)(*names)
    L.Detection2Json.restype = C.c_void_p
    L.Detection2Json.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_char_p), C.c_longlong, C.c_char_p]
    ptr = L.Detection2Json(arr, n, classes, nm, frame_id, filename)
    txt = C.string_at(ptr)
    C.CDLL(None).free(C.c_void_p(ptr))
    return txt


def gen_json():
    """detection2json.npz: the reference's Detection2Json (src/network.cpp:518-592) on json_case(), with and
    without a file name."""
    L = reflib.lib()
    box, prob, names = json_case()
    a = call_detection2json(L, box, prob, names, 42, b"data/frame 17.jpg")
    b = call_detection2json(L, box, prob, names, 9876543210123, None)
    e = call_detection2json(L, box[:0], prob[:0], names, 0, None)
    np.savez_compressed(os.path.join(GOLD, "detection2json.npz"), box=box, prob=prob,
                        with_name=np.frombuffer(a, np.uint8), without_name=np.frombuffer(b, np.uint8),
                        empty=np.frombuffer(e, np.uint8))
    print(a.decode())


def gen_net_multi(name="yolov4", seeds=(101, 202, 303)):
    """net_<cfg>_multi.npz: the reference's decoded heads (1/16 subsample) and detection ids for several more
    inputs of the same weights (b = 1 each) -- the batched HIP run places them at different batch positions among
    other distinct images (tests/test_gpu_net.py), which is what catches a tile reading the wrong image."""
    cfg = os.path.join(ROOT, "cfg", name + ".cfg")
    net = O.parse_cfg(cfg)
    convs = [(l.n, l.c // l.groups, l.size, l.batch_normalize) for l in net.layers if l.type == O.CONVOLUTIONAL]
    wpath = f"/tmp/_dk_{name}.weights"
    synth.write_weights(wpath, convs, seed=2024)
    rn = reflib.RefNet(cfg, wpath, train=False)
    out = {"seeds": np.array(seeds, np.int32)}
    for sd in seeds:
        x = synth.make_input(1, net.c, net.h, net.w, seed=sd)
        rn.predict(x)
        allobj = []
        for i in range(rn.n):
            inf = rn.info(i)
            if inf["type"] != O.YOLO:
                continue
            o = rn.output(i)
            out[f"s{sd}_head_{i}_sub16"] = o[::16].copy()
            wh = inf["out_h"] * inf["out_w"]
            v = o.reshape(3, 5 + inf["classes"], wh)
            allobj.append(v[:, 4, :].ravel())
            allobj.append((v[:, 4:5, :] * v[:, 5:, :]).ravel())
        thresh = pick_threshold(np.concatenate(allobj))
        dets = rn.boxes(thresh)
        onet = O.load_network(cfg, wpath, batch=1)
        O.forward(onet, x)
        od, oid = O.get_boxes(onet, thresh)
        assert np.array_equal(od, dets), "oracle and reference detections differ"
        out[f"s{sd}_thresh"] = np.float32(thresh)
        out[f"s{sd}_det_ids"] = oid
        out[f"s{sd}_det_box_obj"] = dets[:, :5].copy()
        print(f"{name} seed {sd}: {len(dets)} dets at {thresh:.6f}")
    rn.close()
    np.savez_compressed(os.path.join(GOLD, f"net_{name}_multi.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "json":
        gen_json()
    elif len(sys.argv) > 1 and sys.argv[1] == "multi":
        gen_net_multi(sys.argv[2] if len(sys.argv) > 2 else "yolov4")
    elif len(sys.argv) > 1 and sys.argv[1] == "net":
        gen_net(sys.argv[2], len(sys.argv) > 3 and sys.argv[3] == "full")
    elif len(sys.argv) > 1 and sys.argv[1] == "lr":
        gen_lr()
    elif len(sys.argv) > 1 and sys.argv[1] == "map":
        gen_map()
    elif len(sys.argv) > 1 and sys.argv[1] == "gaussian":
        gen_gaussian()
        gen_gaussianloss()
    elif len(sys.argv) > 1 and sys.argv[1] == "grads":
        gen_grads()
    elif len(sys.argv) > 1 and sys.argv[1] == "train_big":
        gen_train_big(sys.argv[2] if len(sys.argv) > 2 else "yolov4", int(sys.argv[3]) if len(sys.argv) > 3 else 8)
    elif len(sys.argv) > 1 and sys.argv[1] == "extra":
        gen_extra()
    elif len(sys.argv) > 1 and sys.argv[1] == "train":
        gen_train()
    elif len(sys.argv) > 1 and sys.argv[1] == "yololoss":
        gen_yololoss()
    elif len(sys.argv) > 1 and sys.argv[1] == "ops":
        gen_ops()
    else:
        main()

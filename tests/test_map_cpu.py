"""CPU: the product's mAP routine (DkMeanAveragePrecision, csrc/host/detector.cpp -- host C++ like the
reference's ValidateDetector) against the oracle's array-for-array restatement of
src/detector.cpp:326-562, on seeded synthetic detection sets with edge cases: classes without
predictions or without ground truth, images without detections or labels, duplicate matches of one
ground-truth box (second one is a false positive), an unreachable IoU threshold."""
import ctypes as C

import numpy as np
import pytest

from oracle import orc_map


def run_product(dk, dets, gts, classes, iou):
    L = dk.lib()
    L.DkMeanAveragePrecision.restype = C.c_double
    L.DkMeanAveragePrecision.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_int),
                                         C.POINTER(C.c_float), C.c_int, C.c_float, C.POINTER(C.c_double)]
    nd = np.array([len(d) for d in dets], np.int32)
    ng = np.array([len(g) for g in gts], np.int32)
    fd = np.ascontiguousarray(np.concatenate([d.reshape(-1) for d in dets] + [np.zeros(1, np.float32)]), np.float32)
    fg = np.ascontiguousarray(np.concatenate([g.reshape(-1) for g in gts] + [np.zeros(1, np.float32)]), np.float32)
    ap = np.zeros(classes, np.float64)
    m = L.DkMeanAveragePrecision(len(dets), nd.ctypes.data_as(C.POINTER(C.c_int)), fd.ctypes.data_as(C.POINTER(C.c_float)),
                                 ng.ctypes.data_as(C.POINTER(C.c_int)), fg.ctypes.data_as(C.POINTER(C.c_float)), classes,
                                 C.c_float(iou), ap.ctypes.data_as(C.POINTER(C.c_double)))
    return m, ap


def synth_set(rng, n_images, classes, max_gt=6, max_det=25):
    dets, gts = [], []
    for i in range(n_images):
        m = int(rng.integers(0, max_gt + 1)) if i != 1 else 0          # image 1: no labels
        g = np.zeros((m, 5), np.float32)
        g[:, 0] = rng.integers(0, max(1, classes - 1), m)              # the last class never has ground truth
        g[:, 1:3] = rng.uniform(.2, .8, (m, 2))
        g[:, 3:5] = rng.uniform(.05, .3, (m, 2))
        n = int(rng.integers(0, max_det + 1)) if i != 2 else 0         # image 2: no detections
        d = np.zeros((n, 4 + classes), np.float32)
        for j in range(n):
            if m and rng.uniform() < .6:                               # near a ground-truth box (sometimes the same one twice)
                k = int(rng.integers(0, m))
                d[j, :4] = g[k, 1:5] * rng.uniform(.85, 1.15, 4).astype(np.float32)
                cls = int(g[k, 0]) if rng.uniform() < .8 else int(rng.integers(0, classes))
            else:
                d[j, :2] = rng.uniform(.1, .9, 2)
                d[j, 2:4] = rng.uniform(.05, .4, 2)
                cls = int(rng.integers(0, classes))
            d[j, 4 + cls] = rng.uniform(.01, 1)
            if rng.uniform() < .3:                                     # a second class on the same box
                d[j, 4 + int(rng.integers(0, classes))] = rng.uniform(.01, 1)
        if classes > 2:
            d[:, 4 + 1] = 0                                            # class 1 never predicted
        dets.append(d)
        gts.append(g)
    return dets, gts


@pytest.mark.parametrize("seed,n_images,classes,iou", [(1, 6, 4, .5), (2, 9, 3, .5), (3, 4, 2, .3), (4, 12, 6, .75), (5, 3, 1, .5)])
def test_map_product_equals_oracle(dk, seed, n_images, classes, iou):
    rng = np.random.default_rng(seed)
    dets, gts = synth_set(rng, n_images, classes)
    want, want_ap = orc_map.mean_average_precision(dets, gts, classes, iou)
    got, got_ap = run_product(dk, dets, gts, classes, iou)
    assert got == want, (got, want)             # same double arithmetic in the same order
    assert np.array_equal(got_ap, np.array(want_ap))
    assert 0 <= got <= 1


def test_map_edge_cases(dk):
    def both(d, g, classes, iou):
        m = run_product(dk, d, g, classes, iou)[0]
        assert m == orc_map.mean_average_precision(d, g, classes, iou)[0]
        return m
    # nothing at all
    assert both([np.zeros((0, 6), np.float32)], [np.zeros((0, 5), np.float32)], 2, .5) == 0
    g = np.array([[0, .5, .5, .2, .2]], np.float32)
    # quirk of the reference's integral (detector.cpp:525-545): the curve of the class that owns the
    # globally highest-scored prediction starts AFTER that prediction, so its first recall step is
    # never integrated -- a lone perfect detection scores AP 0
    d = np.array([[.5, .5, .2, .2, .9, 0]], np.float32)
    assert both([d], [g], 2, .5) == 0.0
    # with a higher-scored prediction of another class in front, the same detection earns AP 1
    d1 = np.array([[.1, .1, .05, .05, 0, .95], [.5, .5, .2, .2, .9, 0]], np.float32)
    assert both([d1], [g], 2, .5) == 0.5
    # two detections of the same box: the second is a false positive; the precision envelope keeps AP 1
    d2 = np.array([[.1, .1, .05, .05, 0, .95], [.5, .5, .2, .2, .9, 0], [.5, .5, .21, .2, .8, 0]], np.float32)
    assert both([d2], [g], 2, .5) == 0.5
    # an unreachable IoU threshold: nothing matches
    assert both([d1], [g], 2, 1.5) == 0.0

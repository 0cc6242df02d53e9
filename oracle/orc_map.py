"""CPU ORACLE of the evaluator's mAP arithmetic.  TEST INFRASTRUCTURE ONLY.

Restates ValidateDetector's bookkeeping (Ravicmoon/darknet src/detector.cpp:326-562) in plain
Python, array for array as the reference builds them: the ValBox list (one entry per detection and
class with a non-zero probability, matched against the image's ground truth by IoU), the global
sort by probability, the per-class running tp/fp/fn and precision/recall at EVERY position, and
the AP integral walked from the end of the curve (:525-545).  Box::Iou: src/box.cpp:36-63.

PARITY UNPINNED for the arithmetic itself: detector.cpp needs OpenCV and is not part of the
oracle/_ref build, and the reference ships no fixture of an mAP value.  What IS pinned are its
inputs: the tests feed it the REAL reference's post-NMS detections (tests/golden/map_*.npz,
produced through oracle/_ref by tools/make_golden.py map).  The reference sorts with std::sort on
the probability alone, so ties are ordered arbitrarily there; here (and in the product) ties keep
insertion order -- fixtures avoid exact ties in what they assert.
"""
import numpy as np


def _overlap(x1, w1, x2, w2):
    l1, l2 = x1 - w1 / 2, x2 - w2 / 2
    left = l1 if l1 > l2 else l2
    r1, r2 = x1 + w1 / 2, x2 + w2 / 2
    right = r1 if r1 < r2 else r2
    return right - left


def box_iou(a, b):
    """Box::Iou with the reference's float arithmetic (np.float32 throughout)."""
    f = np.float32
    a = [f(v) for v in a]
    b = [f(v) for v in b]
    w = f(_overlap(a[0], a[2], b[0], b[2]))
    h = f(_overlap(a[1], a[3], b[1], b[3]))
    inter = f(0) if (w < 0 or h < 0) else f(w * h)
    union = f(f(a[2] * a[3]) + f(b[2] * b[3]) - inter)
    return f(inter / union)


def mean_average_precision(dets_per_image, gts_per_image, classes, iou_thresh):
    """dets_per_image[i]: array [n, 4 + classes] (x, y, w, h, prob per class) AFTER NmsSort;
    gts_per_image[i]: array [m, 5] (id, x, y, w, h).  Returns (mAP, per-class AP list)."""
    val = []   # (p, cid, matched, gt_idx)
    num_gt_class = [0] * classes
    num_gt = 0
    eps = np.finfo(np.float32).eps
    for dets, gts in zip(dets_per_image, gts_per_image):
        for g in gts:
            num_gt_class[int(g[0])] += 1
        for d in dets:
            for cid in range(classes):
                p = np.float32(d[4 + cid])
                if abs(p) < eps:
                    continue
                gt_idx, max_iou = -1, np.float32(0)
                for k, g in enumerate(gts):
                    iou = box_iou(d[:4], g[1:5])
                    if iou > np.float32(iou_thresh) and iou > max_iou and cid == int(g[0]):
                        max_iou, gt_idx = iou, num_gt + k
                val.append((p, cid, gt_idx > -1, gt_idx))
        num_gt += len(gts)
    val.sort(key=lambda v: -float(v[0]))   # stable: ties keep insertion order
    nb = len(val)
    tp = [[0] * nb for _ in range(classes)]
    fp = [[0] * nb for _ in range(classes)]
    prec = [[0.0] * nb for _ in range(classes)]
    rec = [[0.0] * nb for _ in range(classes)]
    gt_flags = [False] * max(num_gt, 1)
    for i, (p, vc, matched, gt_idx) in enumerate(val):
        if i > 0:
            for cid in range(classes):
                tp[cid][i], fp[cid][i] = tp[cid][i - 1], fp[cid][i - 1]
        if matched and not gt_flags[gt_idx]:
            gt_flags[gt_idx] = True
            tp[vc][i] += 1
        else:
            fp[vc][i] += 1
        for cid in range(classes):
            t, f_ = tp[cid][i], fp[cid][i]
            fn = num_gt_class[cid] - t
            prec[cid][i] = t / (t + f_) if t + f_ > 0 else 0.0
            rec[cid][i] = t / (t + fn) if t + fn > 0 else 0.0
    aps = []
    for cid in range(classes):
        ap = 0.0
        if nb:
            last_recall, last_precision = rec[cid][-1], prec[cid][-1]
            for i in range(nb - 1, -1, -1):
                delta = last_recall - rec[cid][i]
                last_recall = rec[cid][i]
                last_precision = max(last_precision, prec[cid][i])
                ap += delta * last_precision
        aps.append(ap)
    return (sum(aps) / classes if classes else 0.0), aps

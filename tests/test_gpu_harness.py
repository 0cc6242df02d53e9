"""the headless trainer / evaluator harness on the HIP path.  The mAP must equal,
to 4 decimals, the mAP of the REAL reference's detections on the same labelled synthetic set
(tests/golden/map_yolov4-tiny.npz: reference NetworkPredict + GetNetworkBoxes + NmsSort through
oracle/_ref; ValidateDetector's arithmetic restated by oracle/orc_map.py)."""
import ctypes as C
import os

import numpy as np
import pytest

import netutil
import synth
import util
from oracle import orc_map

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAME = "yolov4-tiny"
VP = C.c_void_p


def hip_detections(gpu, g, weights):
    """Post-NMS detections of the golden's images from the HIP path: [x, y, w, h, prob...] per image."""
    L = gpu.lib()
    L.DkNmsSortFlat.argtypes = [VP, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float]
    seeds = [int(s) for s in g["seeds"]]
    net = netutil.DkNet(gpu, netutil.cfg_path(NAME), weights, batch=len(seeds))
    x = np.stack([synth.u8_to_chw(synth.make_u8_image(net.w, net.h, sd)).ravel() for sd in seeds])
    net.predict(x)
    out = []
    for b in range(len(seeds)):
        d, _ = net.boxes(b, float(g["thresh"]))
        buf = np.ascontiguousarray(d)
        L.DkNmsSortFlat(buf.ctypes.data, len(buf), buf.shape[1] - 5, float(g["nms"]), 0, 0.6)   # greedynms, beta .6 (cfg)
        dd = np.concatenate([buf[:, :4], buf[:, 5:]], 1)
        out.append(np.ascontiguousarray(dd[(dd[:, 4:] != 0).any(1)]))
    net.close()
    return out


def split(flat, counts):
    out, o = [], 0
    for n in counts:
        out.append(flat[o:o + n])
        o += n
    return out


def product_map(gpu, dets, gts, classes, iou):
    L = gpu.lib()
    L.DkMeanAveragePrecision.restype = C.c_double
    L.DkMeanAveragePrecision.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_int),
                                         C.POINTER(C.c_float), C.c_int, C.c_float, C.POINTER(C.c_double)]
    nd = np.array([len(d) for d in dets], np.int32)
    ng = np.array([len(x) for x in gts], np.int32)
    fd = np.ascontiguousarray(np.concatenate([d.reshape(-1) for d in dets] + [np.zeros(1, np.float32)]), np.float32)
    fg = np.ascontiguousarray(np.concatenate([x.reshape(-1) for x in gts] + [np.zeros(1, np.float32)]), np.float32)
    return L.DkMeanAveragePrecision(len(dets), nd.ctypes.data_as(C.POINTER(C.c_int)), fd.ctypes.data_as(C.POINTER(C.c_float)),
                                    ng.ctypes.data_as(C.POINTER(C.c_int)), fg.ctypes.data_as(C.POINTER(C.c_float)), classes,
                                    C.c_float(iou), None)


def test_map_hip_equals_reference_to_4_decimals(gpu, tmp_path):
    g = np.load(os.path.join(GOLD, "map_%s.npz" % NAME))
    w = str(tmp_path / "w.weights")
    netutil.synth_weights_for(gpu, NAME, w)
    dets = hip_detections(gpu, g, w)
    ref_dets = split(g["dets"], g["n_dets"])
    gts = split(g["gts"], g["n_gts"])
    classes = ref_dets[0].shape[1] - 4
    # the same (detection, class) pairs survive threshold + NMS; boxes and probabilities agree
    def canon(d):   # NmsSort leaves an order that depends on ties; compare as sets of boxes
        k = np.round(d[:, :4].astype(np.float64), 4)
        return d[np.lexsort((k[:, 3], k[:, 2], k[:, 1], k[:, 0]))]
    for b, (a, r) in enumerate(zip(dets, ref_dets)):
        assert a.shape == r.shape, (b, a.shape, r.shape)
        a, r = canon(a), canon(r)
        assert np.array_equal(a[:, 4:] != 0, r[:, 4:] != 0), "image %d: surviving classes differ" % b
        util.assert_close(a, r, "post-NMS detections image %d" % b)
    m_hip = product_map(gpu, dets, gts, classes, 0.5)
    m_ref = float(g["map"])
    assert m_ref > 0.01
    assert abs(m_hip - m_ref) < 5e-5, (m_hip, m_ref)
    assert "%.4f" % m_hip == "%.4f" % m_ref
    # and the oracle's arithmetic on the HIP detections gives the product's number exactly
    assert orc_map.mean_average_precision(dets, gts, classes, 0.5)[0] == m_hip
    print("mAP@0.5: HIP %.6f, reference %.6f" % (m_hip, m_ref))


def write_dataset(tmp_path, g, w, h):
    """PPM images at the network's resolution + the reference's label / list / .data files."""
    d = tmp_path / "set"
    d.mkdir()
    paths = []
    gts = split(g["gts"], g["n_gts"])
    for sd, gt in zip(g["seeds"], gts):
        img = synth.make_u8_image(w, h, int(sd))
        p = d / ("img%d.ppm" % int(sd))
        with open(p, "wb") as f:
            f.write(b"P6\n# synthetic\n%d %d\n255\n" % (w, h))
            f.write(img.tobytes())
        with open(str(p)[:-4] + ".txt", "w") as f:
            for row in gt:
                f.write("%d %.9g %.9g %.9g %.9g\n" % (int(row[0]), row[1], row[2], row[3], row[4]))
        paths.append(str(p))
    (d / "valid.txt").write_text("\n".join(paths) + "\n")
    (d / "train.txt").write_text("\n".join(paths) + "\n")
    (d / "names.txt").write_text("\n".join("class%d" % i for i in range(80)) + "\n")
    (d / "save").mkdir()
    data = d / "set.data"
    data.write_text("classes = 80\ntrain = %s\nvalid = %s\nnames = %s\nsave = %s\n" % (d / "train.txt", d / "valid.txt", d / "names.txt", d / "save"))
    return str(data), str(d / "save")


def test_validate_detector_on_files(gpu, tmp_path):
    """DkValidateDetector end to end: PPM reader, device Mat2Image, forward (batch 3: the last batch is
    short), device candidate extraction, NmsSort, label files, mAP == the reference's."""
    g = np.load(os.path.join(GOLD, "map_%s.npz" % NAME))
    L = gpu.lib()
    L.DkValidateDetectorFlat.restype = C.c_float
    L.DkValidateDetectorFlat.argtypes = [C.c_char_p, VP, C.c_float, C.c_float, C.c_float]
    L.DkSetPullHeads.argtypes = [C.c_int]
    w = str(tmp_path / "w.weights")
    netutil.synth_weights_for(gpu, NAME, w)
    net = netutil.DkNet(gpu, netutil.cfg_path(NAME), w, batch=3)
    data, _ = write_dataset(tmp_path, g, net.w, net.h)
    for pull in (1, 0):
        L.DkSetPullHeads(pull)
        m = L.DkValidateDetectorFlat(data.encode(), net.p, 0.5, float(g["thresh"]), float(g["nms"]))
        assert abs(m - float(g["map"])) < 5e-5, (pull, m, float(g["map"]))
    L.DkSetPullHeads(1)
    net.close()


def test_train_detector_checkpoints_and_resumes(gpu, tmp_path):
    """DkTrainDetector: two iterations on the synthetic set (batch 2), a checkpoint per iteration and
    the final one in the reference's .weights format; a second run resumes from the final checkpoint
    (iteration counter restored from `seen`) and continues to iteration 3."""
    g = np.load(os.path.join(GOLD, "map_%s.npz" % NAME))
    L = gpu.lib()
    L.DkTrainDetectorFlat.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]
    L.LoadNetwork.argtypes = [VP, C.c_char_p, C.c_char_p, C.c_bool, C.c_bool]
    L.LoadNetwork.restype = C.c_bool
    L.GetCurrIter.argtypes = [VP]
    cfg = str(tmp_path / "tiny_b2.cfg")
    open(cfg, "w").write(open(netutil.cfg_path(NAME)).read().replace("batch=64", "batch=2"))
    w = str(tmp_path / "w.weights")
    netutil.synth_weights_for(gpu, NAME, w)
    data, save = write_dataset(tmp_path, g, 416, 416)
    L.DkTrainDetectorFlat(data.encode(), cfg.encode(), w.encode(), 1, 1, 0, 2, 1, 0.0)
    size = os.path.getsize(w)
    for suffix in ("1", "2", "final"):
        p = os.path.join(save, "tiny_b2_%s.weights" % suffix)
        assert os.path.exists(p) and os.path.getsize(p) == size, p
    fin = os.path.join(save, "tiny_b2_final.weights")
    assert open(fin, "rb").read() != open(w, "rb").read()      # the weights moved
    p = L.DkNetworkCreate()
    assert L.LoadNetwork(p, cfg.encode(), fin.encode(), True, False)
    assert L.GetCurrIter(p) == 2
    L.DkNetworkDestroy(p)
    L.DkTrainDetectorFlat(data.encode(), cfg.encode(), fin.encode(), 1, 0, 0, 3, 1, 0.0)
    assert os.path.exists(os.path.join(save, "tiny_b2_3.weights"))

"""CPU (no GPU): the C-ABI library loads and exports every symbol include/*.h
declares; the loader (cfg parser, .weights reader/writer, BN folding) and the host
post-processing (NmsSort) match the oracle / golden fixtures; compute entry points
fail loudly without a device."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import netutil
import synth
from oracle import orc_net as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def declared_symbols():
    names = set()
    for h in ("dk_kernels.h", "dark_hip.h", "yolo_core_hip.h"):
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        for m in re.finditer(r"(?:DK_API|LIB_API)\s+[\w\s\*:<>]+?\b(\w+)\s*\(", txt):
            names.add(m.group(1))
    names.discard("Box")
    return sorted(names)


def test_every_declared_symbol_is_exported(dk):
    L = dk.lib()
    syms = declared_symbols()
    assert len(syms) > 70
    missing = []
    nm = subprocess.run(["nm", "-DC", "--defined-only", dk.LIB_PATH], stdout=subprocess.PIPE).stdout.decode()
    cxx = 0
    for s in syms:
        try:
            getattr(L, s)
        except AttributeError:
            # C++ linkage (std::string / std::vector in the signature): look for the demangled name
            if re.search(r"\b%s\(" % re.escape(s), nm):
                cxx += 1
            else:
                missing.append(s)
    assert not missing, "declared in include/*.h but not exported: %s" % missing
    assert "GetMostProbDets" in nm and cxx >= 5
    # no torch / CUDA types in the boundary
    for h in os.listdir(os.path.join(ROOT, "include")):
        incs = re.findall(r"^\s*#\s*include\s*[<\"]([^>\"]+)", open(os.path.join(ROOT, "include", h)).read(), flags=re.M)
        for inc in incs:
            assert not re.search(r"torch|cuda|cublas|cudnn|curand|ATen", inc), (h, inc)


@pytest.mark.parametrize("name,layers,bflops,wbytes", [("yolov4-tiny", 38, 6.910, 24251276),
                                                       ("yolov4", 162, 128.459, 257717640),
                                                       ("yolov4-csp", 175, 77.003, 211944840),
                                                       ("yolov4x-mish", 201, 139.974, 382983688)])
def test_parser_invariants(dk, name, layers, bflops, wbytes):
    """The format invariants of SURVEY.md section 4 + layer table == the oracle's."""
    L = dk.lib()
    L.ParseNetworkCfg.restype = C.c_bool
    L.ParseNetworkCfg.argtypes = [C.c_void_p, C.c_char_p, C.c_bool]
    p = L.DkNetworkCreate()
    assert L.ParseNetworkCfg(p, netutil.cfg_path(name).encode(), False)
    a = (C.c_int * 8)()
    L.DkNetworkInfo(p, a)
    assert a[0] == layers and a[1] == 1 and a[7] == -1  # inference forces batch 1; no device
    assert L.DkWeightsFileSize(p) == wbytes
    tot = np.float32(0)
    onet = O.parse_cfg(netutil.cfg_path(name))
    for i in range(a[0]):
        f = (C.c_int * 24)()
        L.DkLayerInfo(p, i, f)
        b = np.float32(L.DkLayerBflops(p, i))
        if b > 0:
            tot = np.float32(tot + b)
        ol = onet.layers[i]
        assert (f[0], f[2], f[3], f[4], f[5]) == (ol.type, ol.outputs, ol.out_c, ol.out_h, ol.out_w), i
    assert "%.3f" % tot == "%.3f" % bflops
    L.DkNetworkDestroy(p)


def test_weights_roundtrip_and_bn_folding(dk, tmp_path):
    name = "yolov4-tiny"
    L = dk.lib()
    w = str(tmp_path / "a.weights")
    convs = netutil.synth_weights_for(dk, name, w)
    onet = O.parse_cfg(netutil.cfg_path(name))
    assert convs == [(l.n, l.c // l.groups, l.size, l.batch_normalize) for l in onet.layers if l.type == O.CONVOLUTIONAL]
    # un-fused: LoadWeights + SaveWeights reproduces the file byte for byte
    L.ParseNetworkCfg.restype = C.c_bool
    L.ParseNetworkCfg.argtypes = [C.c_void_p, C.c_char_p, C.c_bool]
    L.LoadWeights.restype = C.c_bool
    L.LoadWeights.argtypes = [C.c_void_p, C.c_char_p]
    p = L.DkNetworkCreate()
    assert L.ParseNetworkCfg(p, netutil.cfg_path(name).encode(), False)
    assert L.LoadWeights(p, w.encode())
    w2 = str(tmp_path / "b.weights")
    L.SaveWeights(p, w2.encode())
    assert open(w, "rb").read() == open(w2, "rb").read()
    L.DkNetworkDestroy(p)
    assert not L.LoadWeights(L.DkNetworkCreate(), b"/nonexistent.weights")
    # fused: LoadNetwork(train=false) folds BN exactly like the oracle / reference
    net = netutil.DkNet(dk, netutil.cfg_path(name), w)
    onet = O.load_network(netutil.cfg_path(name), w)
    L.DkLayerHostPtr.restype = C.POINTER(C.c_float)
    L.DkLayerHostPtr.argtypes = [C.c_void_p, C.c_int, C.c_int]
    for i, l in enumerate(onet.layers):
        if l.type != O.CONVOLUTIONAL:
            continue
        assert net.info(i)["batch_normalize"] == 0
        wts = np.ctypeslib.as_array(L.DkLayerHostPtr(net.p, i, 1), shape=(l.nweights,))
        bs = np.ctypeslib.as_array(L.DkLayerHostPtr(net.p, i, 2), shape=(l.n,))
        assert np.array_equal(wts, l.weights) and np.array_equal(bs, l.biases), i
    net.close()


def test_nms_sort_vs_reference_golden(dk):
    """NmsSort on the reference's own detection list (rebuilt by the oracle, which is
    bit-identical) must reproduce the reference's post-NMS result."""
    name = "yolov4-tiny"
    g = np.load(os.path.join(GOLD, "net_%s.npz" % name))
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        w = os.path.join(d, "w.weights")
        onet = O.parse_cfg(netutil.cfg_path(name))
        synth.write_weights(w, [(l.n, l.c // l.groups, l.size, l.batch_normalize) for l in onet.layers if l.type == O.CONVOLUTIONAL])
        onet = O.load_network(netutil.cfg_path(name), w)
        O.forward(onet, synth.make_input(1, 3, 416, 416))
    dets, _ = O.get_boxes(onet, float(g["thresh"]))
    L = dk.lib()
    L.DkNmsSortFlat.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float]
    buf = np.ascontiguousarray(dets)
    kind, beta = g["nms_kind_beta"]
    L.DkNmsSortFlat(buf.ctypes.data, len(buf), buf.shape[1] - 5, float(g["nms_thresh"]), int(kind), float(beta))
    assert np.array_equal(buf[:, :5], g["nms_box_obj"])
    keep = buf[:, 5:] > 0
    assert np.array_equal(keep.sum(1), g["nms_kept_per_det"])
    assert np.array_equal(np.where(keep, buf[:, 5:], 0).sum(1, dtype=np.float64), g["nms_kept_prob_sum"])


def test_compute_fails_loudly_without_gpu(dk):
    if dk.have_gpu():
        pytest.skip("a GPU is present")
    code = ("import sys; sys.path.insert(0, %r); import numpy as np; import darknet_amd as dk; "
            "from darknet_amd import netapi; n = netapi.DkNet(dk, netapi.cfg_path('yolov4-tiny')); "
            "n.predict(np.zeros(n.inputs, np.float32))" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0
    assert b"no HIP device" in r.stderr or b"no CPU fallback" in r.stderr


def test_synth_is_deterministic_and_matches_scalar_lcg():
    g = synth.LCG(12345)
    v = g.uniform(70000)
    s = 12345
    ref = []
    for _ in range(70000):
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        ref.append((s >> 8) / 16777216.0)
    assert np.array_equal(v, np.array(ref, np.float32))
    assert np.array_equal(synth.make_input(2, 3, 8, 8), synth.make_input(2, 3, 8, 8))


def test_kernel_index_arithmetic_on_host(dk):
    """Host-callable halves of the kernels' index logic: (1) the reciprocal division is exact;
    (2) under every XCD partition the host can pick, the workgroup -> tile map covers each
    (group, M tile, N tile) exactly once (surplus workgroups are flagged invalid)."""
    import ctypes as C
    L = dk.lib()
    L.DkTestFdiv.argtypes = [C.c_int, C.c_int]
    L.DkTestFdiv.restype = C.c_int
    rng = np.random.default_rng(3)
    ds = [1, 2, 3, 7, 9, 19, 361, 1444, 5776, 23104, 92416, 369664, 1048573, (1 << 20) + 7]
    for d in ds:
        ns = np.concatenate([np.arange(0, 5 * d, max(1, d // 7)), [d - 1, d, d + 1, 2 ** 31 - 1, 2 ** 30, 2 ** 29 - 1],
                             rng.integers(0, 2 ** 31 - 1, 200), (np.arange(1, 40) * d - 1), np.arange(1, 40) * d])
        for n in ns:
            n = int(min(n, 2 ** 31 - 1))
            assert L.DkTestFdiv(n, d) == n // d, (n, d)
    L.DkTestBlockTile.argtypes = [C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_int, C.POINTER(C.c_int)]
    out = (C.c_int * 6)()
    seen_pm = set()
    for tiles_m, tiles_n, groups, wbytes in [(8, 181, 1, 4718592), (16, 46, 1, 18874368), (8, 91, 1, 9437184),
                                             (4, 181, 1, 2359296), (3, 17, 1, 40 << 20), (1, 5, 1, 64 << 20),
                                             (2, 7, 2, 64 << 20), (16, 1, 1, 50 << 20), (5, 3, 1, 7 << 20)]:
        L.DkTestBlockTile(tiles_m, tiles_n, groups, wbytes, 0, out)
        nblk, pm = out[0], out[1]
        seen_pm.add(pm)
        assert pm in (1, 2, 4, 8) and pm <= max(1, tiles_m)
        count = {}
        for bid in range(nblk):
            L.DkTestBlockTile(tiles_m, tiles_n, groups, wbytes, bid, out)
            if out[2]:
                key = (out[3], out[4], out[5])
                assert 0 <= out[3] < groups and 0 <= out[4] < tiles_m and 0 <= out[5] < tiles_n
                count[key] = count.get(key, 0) + 1
        assert len(count) == groups * tiles_m * tiles_n and set(count.values()) == {1}, (tiles_m, tiles_n, pm)
    assert {1, 2, 8} <= seen_pm


def test_adam_cfg_parses_and_oracle_update_properties(dk, tmp_path):
    """adam=1 (parser.cpp:995-1000): the cfg parses (B1 / B2 / eps consumed) -- round 1 refused it, round 2 has the
    update (dk_adam_update).  And the oracle's restatement of adam_update_gpu (blas_kernels.cu:99-134) behaves as
    Adam must: first step moves every weight by ~lr against the gradient's sign, gradients are zeroed."""
    src = open(netutil.cfg_path("yolov4-tiny")).read().replace("[net]", "[net]\nadam=1\nB1=0.9\nB2=0.999\neps=0.000001", 1)
    cfg = tmp_path / "adam.cfg"
    cfg.write_text(src)
    code = ("import sys, ctypes as C; sys.path.insert(0, %r); import darknet_amd as dk; L = dk.lib(); "
            "L.ParseNetworkCfg.restype = C.c_bool; L.ParseNetworkCfg.argtypes = [C.c_void_p, C.c_char_p, C.c_bool]; "
            "sys.exit(0 if L.ParseNetworkCfg(L.DkNetworkCreate(), %r.encode(), True) else 3)" % (ROOT, str(cfg)))
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr[-400:]
    from oracle import orc_net as O
    OL = O.lib()
    OL.orc_adam_update.argtypes = [C.POINTER(C.c_float)] * 4 + [C.c_float] * 5 + [C.c_int] * 3
    OL.orc_adam_update.restype = None
    rng = np.random.default_rng(1)
    n = 4096
    w = rng.normal(0, 1, n).astype(np.float32); w0 = w.copy()
    d = rng.normal(0, 1, n).astype(np.float32); d0 = d.copy()
    m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    OL.orc_adam_update(O.fptr(w), O.fptr(d), O.fptr(m), O.fptr(v), .9, .999, 1e-6, 0.0, 1e-3, n, 1, 1)
    assert not d.any()
    # t = 1, zero moments: mhat = g, vhat = g^2 -> step = lr * g / (|g| + eps) = lr * sign(g) (darknet ADDS: gradients
    # are stored as descent directions)
    big = np.abs(d0) > 1e-3
    assert np.allclose((w - w0)[big], 1e-3 * np.sign(d0)[big], rtol=2e-3)


def test_c_bucket_segments_equal_python_rule(dk):
    """The C trainer (TrainNetworks, csrc/host/multigpu.cpp) cuts the gradient bucket into backward-order
    slices with the same rule as darknet_amd/train_dist.py: bucket_segments."""
    from darknet_amd.train_dist import bucket_segments
    L = dk.lib()
    L.ParseNetworkCfg.restype = C.c_bool
    L.ParseNetworkCfg.argtypes = [C.c_void_p, C.c_char_p, C.c_bool]
    L.DkGradBucketOffset.restype = C.c_size_t
    L.DkGradBucketOffset.argtypes = [C.c_void_p, C.c_int]
    L.DkBucketSegments.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_longlong), C.c_int]
    for name in ("yolov4-tiny", "yolov4"):
        p = L.DkNetworkCreate()
        assert L.ParseNetworkCfg(p, netutil.cfg_path(name).encode(), True)
        a = (C.c_int * 8)()
        L.DkNetworkInfo(p, a)
        n = a[0]
        offs = [L.DkGradBucketOffset(p, i) for i in range(n + 1)]
        convs = [i for i in range(n) if offs[i + 1] > offs[i]]
        assert convs and offs[-1] > 0
        for nseg in (1, 2, 4, 7):
            want = bucket_segments(convs, [offs[i + 1] - offs[i] for i in convs], n, nseg)
            out = (C.c_longlong * (4 * 16))()
            k = L.DkBucketSegments(p, nseg, out, 16)
            got = [tuple(out[4 * i:4 * i + 4]) for i in range(k)]
            assert got == [tuple(int(v) for v in s) for s in want], (name, nseg)
        L.DkNetworkDestroy(p)


def test_learning_rate_schedules_vs_reference_golden(dk, tmp_path):
    """GetCurrLr (network.cpp:32-84) for every deterministic policy -- burn-in ramp, step / steps boundaries, exp,
    poly, sigmoid, SGDR warm restarts -- bit-exact against the real reference (tests/golden/lr_schedule.npz,
    tools/make_golden.py lr)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_golden as MG
    g = np.load(os.path.join(ROOT, "tests", "golden", "lr_schedule.npz"))
    L = dk.lib()
    L.ParseNetworkCfg.restype = C.c_bool
    L.ParseNetworkCfg.argtypes = [C.c_void_p, C.c_char_p, C.c_bool]
    L.DkNetworkCreate.restype = C.c_void_p
    L.GetCurrLr.restype = C.c_float
    L.GetCurrLr.argtypes = [C.c_void_p]
    L.DkSetCurrIter.argtypes = [C.c_void_p, C.c_longlong]
    L.DkSetMaxIter.argtypes = [C.c_void_p, C.c_int]
    L.DkNetworkDestroy.argtypes = [C.c_void_p]
    assert list(g["iters"]) == MG.LR_ITERS and int(g["max_iter"]) == MG.LR_MAX_ITER
    for pol in MG.LR_POLICIES:
        cfg = str(tmp_path / (pol + ".cfg"))
        open(cfg, "w").write(MG.lr_cfg_text(pol))
        p = L.DkNetworkCreate()
        assert L.ParseNetworkCfg(p, cfg.encode(), False)
        L.DkSetMaxIter(p, MG.LR_MAX_ITER)
        got = []
        for it in MG.LR_ITERS:
            L.DkSetCurrIter(p, it)
            got.append(L.GetCurrLr(p))
        got = np.array(got, np.float32)
        ref = g["lr_" + pol]
        assert ref.max() > ref.min() or pol == "constant"
        bad = np.nonzero(got.view(np.uint32) != ref.view(np.uint32))[0]
        assert bad.size == 0, (pol, [(MG.LR_ITERS[i], float(got[i]), float(ref[i])) for i in bad[:5]])
        L.DkNetworkDestroy(p)


def test_detection2json_vs_reference_golden(dk):
    """Detection2Json (src/network.cpp:518-592, yolo_core.h:635): byte-identical text to the reference's own
    function (fixture tests/golden/detection2json.npz, generated by tools/make_golden.py json from
    oracle/_ref): %f fields, separators, the fixed 0.005 threshold, `dont_show*` classes skipped, with / without a
    file name, empty list; and against the live reference library where it exists."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_golden as MG
    import reflib
    g = np.load(os.path.join(GOLD, "detection2json.npz"))
    L = dk.lib()
    box, prob, names = MG.json_case()
    assert np.array_equal(box, g["box"]) and np.array_equal(prob, g["prob"])
    a = MG.call_detection2json(L, box, prob, names, 42, b"data/frame 17.jpg")
    b = MG.call_detection2json(L, box, prob, names, 9876543210123, None)
    e = MG.call_detection2json(L, box[:0], prob[:0], names, 0, None)
    assert a == g["with_name"].tobytes()
    assert b == g["without_name"].tobytes()
    assert e == g["empty"].tobytes()
    assert b"dont_show" not in a and a.count(b"class_id") == 8
    if reflib.available():
        R = reflib.lib()
        rng = np.random.default_rng(3)
        box2 = rng.uniform(0, 1, (40, 4)).astype(np.float32)
        prob2 = rng.uniform(0, 0.02, (40, 5)).astype(np.float32)
        assert MG.call_detection2json(L, box2, prob2, names, 1, b"x") == MG.call_detection2json(R, box2, prob2, names, 1, b"x")

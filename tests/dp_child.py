"""Child process of tests/test_gpu_train.py::test_train_networks_collective_path_on_one_gpu: two replicas on device 0
through the collective branch of TrainNetworks (the parent sets DK_RCCL_LIB / DK_DP_SHARED_DEVICE_COLLECTIVE /
DK_DETERMINISTIC before this process loads the library).  argv: cfg weights workdir"""
import ctypes as C
import os
import sys

import numpy as np

import darknet_amd as gpu
import netutil
import util
from oracle import orc_net as O

VP = C.c_void_p
cfg, wpath, work = sys.argv[1:4]
x = np.load(os.path.join(work, "x.npy"))
truth = np.ascontiguousarray(np.load(os.path.join(work, "truth.npy")))
L = gpu.lib()
for fn, at, rt in (("DkNetworkArrayCreate", [C.c_int], VP), ("DkNetworkArrayAt", [VP, C.c_int], VP),
                   ("DkNetworkArrayDestroy", [VP, C.c_int], None),
                   ("LoadNetwork", [VP, C.c_char_p, C.c_char_p, C.c_bool, C.c_bool], C.c_bool),
                   ("DkTrainNetworksFlat", [VP, C.c_int, VP, C.c_int, VP, C.c_int, C.c_int, C.c_int], C.c_float),
                   ("DkSetMaxIter", [VP, C.c_int], None), ("TrainNetworkDatum", [VP, VP, VP], C.c_float),
                   ("UpdateNetworkGpu", [VP], None), ("DkAdvanceIteration", [VP], None), ("SyncNetworks", [VP, C.c_int], None),
                   ("DkLayerPull", [VP, C.c_int, C.c_int, VP, C.c_size_t], C.c_long)):
    getattr(L, fn).argtypes = at
    getattr(L, fn).restype = rt
B = 2
assert x.shape[0] == B
one, sub = os.path.join(work, "one.cfg"), os.path.join(work, "sub.cfg")
txt = open(cfg).read()
open(one, "w").write(txt.replace("batch=%d" % B, "batch=1"))
open(sub, "w").write(txt.replace("subdivisions=1", "subdivisions=%d" % B))
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
assert all(np.isfinite(c) and c > 0 for c in costs), costs
shim = C.CDLL(os.environ["DK_RCCL_LIB"])
st = (C.c_longlong * 3)()
shim.shim_rccl_stats(st)
calls_train = st[0]
print("shim after %d TrainNetworks steps: %d all-reduce calls, %d floats, max ranks inside a call at once %d" % (STEPS, st[0], st[1], st[2]))
assert st[0] >= 2 * STEPS, "the bucket was not all-reduced in segments"
assert st[2] == B, "the replica threads never met inside the collective"
L.SyncNetworks(nets, B)
shim.shim_rccl_stats(st)
assert st[0] > calls_train, "SyncNetworks did not go through the collective library"


def weights(p, i, n, which=1):
    out = np.empty(n, np.float32)
    assert L.DkLayerPull(p, i, which, out.ctypes.data, n) == n
    return out


p0, p1 = L.DkNetworkArrayAt(nets, 0), L.DkNetworkArrayAt(nets, 1)
nconv = 0
for i in range(ref.n):
    f = ref.info(i)
    if f["type"] == O.CONVOLUTIONAL:
        a, b0, b1 = weights(ref.p, i, f["nweights"]), weights(p0, i, f["nweights"]), weights(p1, i, f["nweights"])
        assert np.array_equal(b0, b1), "replicas diverged at layer %d" % i
        util.assert_close(b0, a, "weights after %d collective TrainNetworks steps, layer %d" % (STEPS, i), rel=2e-5, atol_rms=2e-6)
        if f["batch_normalize"]:
            assert np.array_equal(weights(p0, i, f["n"], 4), weights(p1, i, f["n"], 4)), "rolling means not synchronised"
        nconv += 1
assert nconv == 21
L.DkNetworkArrayDestroy(nets, B)
ref.close()
print("DP-CHILD-OK")

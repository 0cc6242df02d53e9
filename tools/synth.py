"""Deterministic synthetic inputs and .weights files.

There is no network access for datasets or checkpoints, so every parity test
and bench.py run uses the same generated data, produced identically for the
oracle, the real reference (build container only) and the HIP path:

  * a 32-bit LCG  s <- s*1664525 + 1013904223 (mod 2^32), value (s>>8)/2^24;
  * input: batch x 3 x H x W i.i.d. U[0,1), seed 12345;
  * weights, in .weights file order (src/parser.cpp:1695-1759), one stream:
      bias ~ 0.1(U-.5); if batch_normalize: scale ~ .75+.5U,
      rolling_mean ~ 0.1(U-.5), rolling_variance ~ .5+U;
      W ~ sqrt(2/(k*k*c/groups)) * (2U-1).
    (BN variance ~1 keeps activations O(1) after FuseConvBatchNorm.)

The file layout written here is the reference's: int32 major=0, minor=2,
revision=5, uint64 seen=0, then the per-layer tensors (src/parser.cpp:1602-1611).
"""
import struct

import numpy as np

A = np.uint32(1664525)
Cc = np.uint32(1013904223)
_BLOCK = 1 << 16


def _jump_tables():
    # a_k = A^k, c_k = C*(A^(k-1)+...+1) for k = 1.._BLOCK, mod 2^32
    a = np.empty(_BLOCK, np.uint32)
    c = np.empty(_BLOCK, np.uint32)
    ak, ck = 1, 0
    for k in range(_BLOCK):
        ak = (ak * 1664525) & 0xFFFFFFFF
        ck = (ck * 1664525 + 1013904223) & 0xFFFFFFFF
        a[k], c[k] = ak, ck
    return a, c


_TAB = None


class LCG:
    def __init__(self, seed):
        self.s = np.uint32(seed)

    def uniform(self, n):
        """Next n values of the stream as float32 in [0,1)."""
        global _TAB
        if _TAB is None:
            _TAB = _jump_tables()
        a, c = _TAB
        out = np.empty(n, np.uint32)
        pos = 0
        s = np.uint32(self.s)
        with np.errstate(over="ignore"):
            while pos < n:
                m = min(_BLOCK, n - pos)
                blk = a[:m] * s + c[:m]
                out[pos:pos + m] = blk
                s = blk[m - 1]
                pos += m
        self.s = s
        return ((out >> np.uint32(8)).astype(np.float32)) / np.float32(16777216.0)


def make_input(batch, c, h, w, seed=12345):
    return LCG(seed).uniform(batch * c * h * w).reshape(batch, c * h * w)


def write_weights(path, convs, seed=2024, head_bias=None):
    """convs: iterable of (n, c_per_group, size, batch_normalize) in layer order.
    head_bias: optional float added to the biases of non-BN (head) convs -- not
    used by default."""
    g = LCG(seed)
    with open(path, "wb") as f:
        f.write(struct.pack("<iiiQ", 0, 2, 5, 0))
        for (n, cpg, size, bn) in convs:
            bias = np.float32(0.1) * (g.uniform(n) - np.float32(.5))
            if head_bias is not None and not bn:
                bias = bias + np.float32(head_bias)
            bias.astype(np.float32).tofile(f)
            if bn:
                (np.float32(.75) + np.float32(.5) * g.uniform(n)).astype(np.float32).tofile(f)
                (np.float32(0.1) * (g.uniform(n) - np.float32(.5))).astype(np.float32).tofile(f)
                (np.float32(.5) + g.uniform(n)).astype(np.float32).tofile(f)
            fan = size * size * cpg
            sc = np.float32(np.sqrt(2.0 / fan))
            nw = n * cpg * size * size
            (sc * (np.float32(2) * g.uniform(nw) - np.float32(1))).astype(np.float32).tofile(f)


def write_weights_layers(path, layers, seed=2024):
    """Like write_weights for nets that also hold standalone [batchnorm] layers.  layers: in file
    order, ("conv", n, c_per_group, size, batch_normalize) or ("batchnorm", c): biases, scales,
    rolling_mean, rolling_variance with the conv BN distributions (src/parser.cpp:1683-1693)."""
    g = LCG(seed)
    with open(path, "wb") as f:
        f.write(struct.pack("<iiiQ", 0, 2, 5, 0))
        for ent in layers:
            if ent[0] == "batchnorm":
                c = ent[1]
                (np.float32(0.1) * (g.uniform(c) - np.float32(.5))).astype(np.float32).tofile(f)
                (np.float32(.75) + np.float32(.5) * g.uniform(c)).astype(np.float32).tofile(f)
                (np.float32(0.1) * (g.uniform(c) - np.float32(.5))).astype(np.float32).tofile(f)
                (np.float32(.5) + g.uniform(c)).astype(np.float32).tofile(f)
                continue
            _, n, cpg, size, bn = ent
            (np.float32(0.1) * (g.uniform(n) - np.float32(.5))).astype(np.float32).tofile(f)
            if bn:
                (np.float32(.75) + np.float32(.5) * g.uniform(n)).astype(np.float32).tofile(f)
                (np.float32(0.1) * (g.uniform(n) - np.float32(.5))).astype(np.float32).tofile(f)
                (np.float32(.5) + g.uniform(n)).astype(np.float32).tofile(f)
            sc = np.float32(np.sqrt(2.0 / (size * size * cpg)))
            (sc * (np.float32(2) * g.uniform(n * cpg * size * size) - np.float32(1))).astype(np.float32).tofile(f)


def weight_layers_of(onet):
    """The (oracle-parsed) network's weight-bearing layers in .weights order."""
    out = []
    for l in onet.layers:
        if l.type == 0:       # CONVOLUTIONAL
            out.append(("conv", l.n, l.c // l.groups, l.size, l.batch_normalize))
        elif l.type == 14:    # BATCHNORM
            out.append(("batchnorm", l.c))
    return out


def se_cfgs(outdir="/tmp"):
    """cfg/se-test.cfg as is (inference) and, for the train step, without the [dropout] section (the
    reference's dropout draws from rand(), which no fixture can pin) and without the standalone
    [batchnorm] (the reference's CPU path cannot train it: FillBatchnormLayer allocates l->x /
    l->x_norm only under GPU, src/batchnorm_layer.cpp:9-88, and ForwardBatchnormLayer :224-227
    writes them -> it crashes; train-mode standalone [batchnorm] is therefore pinned through the
    conv-BN path's identical kernels, not directly).  Returns (inference cfg, train cfg) paths."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "cfg", "se-test.cfg")).read()
    inf, tr = os.path.join(str(outdir), "_dk_se_inf.cfg"), os.path.join(str(outdir), "_dk_se_train.cfg")
    open(inf, "w").write(txt)
    t = txt
    for sec in ("[dropout]", "[batchnorm]"):
        a = t.index("\n" + sec + "\n") + 1
        b = t.index("[convolutional]", a)
        t = t[:a] + t[b:]
    open(tr, "w").write(t)
    return inf, tr


def make_u8_image(w, h, seed):
    """Interleaved RGB u8 test image [h, w, 3] from the LCG stream (values 0..255)."""
    v = LCG(seed).uniform(w * h * 3)
    return (v * np.float32(256.0)).astype(np.uint8).reshape(h, w, 3)


def u8_to_chw(img):
    """Mat2Image (src/visualize.cpp:26-55): chw[k][y][x] = hwc[y][x][k] / 255.0f"""
    return np.ascontiguousarray(img.transpose(2, 0, 1).astype(np.float32) / np.float32(255.0))

"""Data-parallel training step over RCCL (BASELINE config C4).

One process per GPU, identical replicas, the global batch split evenly.  Each
replica runs TrainNetworkDatum on its shard (forward with per-replica batch-norm
statistics, host yolo loss, backward); all conv gradients live in ONE contiguous
fp32 bucket allocated through torch, which is all-reduced (sum) in place with
torch.distributed (backend nccl = RCCL over xGMI); then every replica applies the
same SGD update with B = per-replica batch x world size.  This is the synchronous
form of the reference's data parallelism (one replica per GPU, host-mediated weight
averaging every 4th iteration: src/network_kernels.cu:366-484) and is arithmetically
the reference's own accumulation over `subdivisions` (one sub-batch per replica).
"""
import ctypes as C

import os

import numpy as np


def bucket_segments(conv_layers, sizes, n_layers, nseg):
    """Backward-order segments for the overlapped all-reduce.  conv_layers: indices of the layers
    that own gradients (ascending), sizes: floats each owns in the bucket (same order).  Returns
    [(hi, lo, off, cnt)]: after the backward of layers hi-1 .. lo the bucket slice [off, off+cnt)
    is final and can be reduced; segments are cut so that each carries ~1/nseg of the bucket."""
    total = float(sum(sizes))
    segs, hi, acc, cnt_hi = [], n_layers, 0.0, len(conv_layers)
    target = total / max(1, nseg)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    for k in range(len(conv_layers) - 1, -1, -1):
        acc += sizes[k]
        if acc >= target and k > 0 and len(segs) < nseg - 1:
            lo = conv_layers[k]
            segs.append((hi, lo, int(offs[k]), int(offs[cnt_hi] - offs[k])))
            hi, cnt_hi, acc = lo, k, 0.0
    segs.append((hi, 0, 0, int(offs[cnt_hi])))
    return [s for s in segs if s[0] > s[1]]


class DataParallelTrainer:
    def __init__(self, dk, net, ctx, overlap=True, segments=4):
        import torch
        self.torch, self.dk, self.net, self.ctx = torch, dk, net, ctx
        L = self.L = dk.lib()
        L.DkGradBucketSize.restype = C.c_size_t
        L.DkGradBucketSize.argtypes = [C.c_void_p]
        L.DkGradBucketOffset.restype = C.c_size_t
        L.DkGradBucketOffset.argtypes = [C.c_void_p, C.c_int]
        L.DkAttachGradBucket.argtypes = [C.c_void_p, C.c_void_p]
        L.DkSetReplicas.argtypes = [C.c_void_p, C.c_int]
        L.DkAdvanceIteration.argtypes = [C.c_void_p]
        L.TrainNetworkDatum.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.TrainNetworkDatum.restype = C.c_float
        L.DkTrainForward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.DkBackwardRange.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.DkTrainFinish.argtypes = [C.c_void_p]
        L.DkTrainFinish.restype = C.c_float
        L.UpdateNetworkGpu.argtypes = [C.c_void_p]
        L.get_cuda_stream.restype = C.c_void_p
        n = L.DkGradBucketSize(net.p)
        self.bucket = torch.zeros(n, dtype=torch.float32, device="cuda")
        L.DkAttachGradBucket(net.p, self.bucket.data_ptr())
        L.DkSetReplicas(net.p, ctx.world)
        # DK_TRAIN_OVERLAP=0: the unsplit step (TrainNetworkDatum; with DK_TRAIN_TIMING=1 it prints host issue times)
        self.overlap = overlap and os.environ.get("DK_TRAIN_OVERLAP", "1") != "0"
        # the library launches on its own HIP stream; torch collectives are ordered against it
        self.dk_stream = torch.cuda.ExternalStream(L.get_cuda_stream())
        offs = [L.DkGradBucketOffset(net.p, i) for i in range(net.n + 1)]
        convs = [i for i in range(net.n) if offs[i + 1] > offs[i]]
        self.segments = bucket_segments(convs, [offs[i + 1] - offs[i] for i in convs], net.n, segments)

    def step(self, x, truth):
        """x: [batch, c*h*w] float32 shard, truth: [batch, max_boxes*5]."""
        L, torch = self.L, self.torch
        x = np.ascontiguousarray(x, np.float32)
        truth = np.ascontiguousarray(truth, np.float32)
        if not self.overlap:
            cost = L.TrainNetworkDatum(self.net.p, x.ctypes.data, truth.ctypes.data)  # syncs the stream
            if self.ctx.world > 1:
                self.ctx.dist.all_reduce(self.bucket)   # sum over replicas, in place, over xGMI
                torch.cuda.synchronize()
        else:
            # backward in segments; each segment's slice of the bucket is all-reduced (RCCL, its own
            # stream) while the next segment's backward runs
            L.DkTrainForward(self.net.p, x.ctypes.data, truth.ctypes.data)
            works = []
            with torch.cuda.stream(self.dk_stream):
                for hi, lo, off, cnt in self.segments:
                    L.DkBackwardRange(self.net.p, hi, lo)
                    if self.ctx.world > 1 and cnt:
                        works.append(self.ctx.dist.all_reduce(self.bucket[off:off + cnt], async_op=True))
                for w in works:
                    w.wait()   # the library's stream waits for the collectives
            cost = L.DkTrainFinish(self.net.p)
        L.DkAdvanceIteration(self.net.p)
        L.UpdateNetworkGpu(self.net.p)
        return cost

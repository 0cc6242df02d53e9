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

import numpy as np


class DataParallelTrainer:
    def __init__(self, dk, net, ctx):
        import torch
        self.torch, self.dk, self.net, self.ctx = torch, dk, net, ctx
        L = self.L = dk.lib()
        L.DkGradBucketSize.restype = C.c_size_t
        L.DkGradBucketSize.argtypes = [C.c_void_p]
        L.DkAttachGradBucket.argtypes = [C.c_void_p, C.c_void_p]
        L.DkSetSubdivisions.argtypes = [C.c_void_p, C.c_int]
        L.DkAdvanceIteration.argtypes = [C.c_void_p]
        L.TrainNetworkDatum.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.TrainNetworkDatum.restype = C.c_float
        L.UpdateNetworkGpu.argtypes = [C.c_void_p]
        n = L.DkGradBucketSize(net.p)
        self.bucket = torch.zeros(n, dtype=torch.float32, device="cuda")
        L.DkAttachGradBucket(net.p, self.bucket.data_ptr())
        L.DkSetSubdivisions(net.p, ctx.world)

    def step(self, x, truth):
        """x: [batch, c*h*w] float32 shard, truth: [batch, max_boxes*5]."""
        L, torch = self.L, self.torch
        x = np.ascontiguousarray(x, np.float32)
        truth = np.ascontiguousarray(truth, np.float32)
        cost = L.TrainNetworkDatum(self.net.p, x.ctypes.data, truth.ctypes.data)  # syncs the stream
        if self.ctx.world > 1:
            self.ctx.dist.all_reduce(self.bucket)   # sum over replicas, in place, over xGMI
            torch.cuda.synchronize()
        L.DkAdvanceIteration(self.net.p)
        L.UpdateNetworkGpu(self.net.p)
        return cost

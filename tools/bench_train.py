#!/usr/bin/env python3
"""Train-step throughput (BASELINE config C4): yolov4 608x608, batch 8 per GPU,
forward (BN batch statistics) + host yolo loss + backward + [RCCL all-reduce of the
gradient bucket] + SGD update.  One process per GPU (torch.distributed.run for N>1).
Prints one JSON line.  usage: bench_train.py [--gpus N] [--steps K] [--warmup W] [--cfg yolov4] [--batch 8]"""
import argparse, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cfg", default="yolov4")
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    import numpy as np
    import torch
    import darknet_amd as dk
    from darknet_amd import dist as dkdist, netapi, train_dist
    import synth
    ctx = dkdist.DistCtx(backend="nccl")
    L = dk.lib()
    torch.cuda.set_device(ctx.local_rank)
    L.cuda_set_device(ctx.local_rank)
    tmp = tempfile.mkdtemp(prefix="dktrain%d_" % ctx.rank)
    w = os.path.join(tmp, "w.weights")
    netapi.synth_weights_for(dk, a.cfg, w)
    cfg = os.path.join(tmp, "t.cfg")
    txt = open(netapi.cfg_path(a.cfg)).read()
    import re
    txt = re.sub(r"batch=\d+", "batch=%d" % a.batch, txt, count=1)
    txt = re.sub(r"subdivisions=\d+", "subdivisions=1", txt, count=1)
    open(cfg, "w").write(txt)
    net = netapi.DkNet(dk, cfg, w, train=True)
    L.DkSetMaxIter.argtypes = [dk.C.c_void_p, dk.C.c_int]
    L.DkSetMaxIter(net.p, 100000)
    tr = train_dist.DataParallelTrainer(dk, net, ctx)
    lo, hi = dkdist.shard_range(a.batch * ctx.world, ctx.rank, ctx.world)
    x = synth.make_input(hi, net.c, net.h, net.w)[lo:hi]
    truth = np.zeros((a.batch, 90 * 5), np.float32)
    for b in range(a.batch):
        for t, box in enumerate([(.3, .4, .2, .3, 1), (.6, .5, .4, .35, 17), (.8, .2, .1, .15, 60)]):
            truth[b, t * 5:(t + 1) * 5] = box
    for _ in range(a.warmup):
        cost = tr.step(x, truth)
    torch.cuda.synchronize(); ctx.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        cost = tr.step(x, truth)
    torch.cuda.synchronize(); ctx.barrier()
    dt = time.perf_counter() - t0
    rate, tmax = dkdist.aggregate_throughput(ctx, a.batch * a.steps, dt)
    if ctx.rank == 0:
        print(json.dumps({"metric": "images/sec %s train step" % a.cfg, "value": rate, "n_gpus": ctx.world,
                          "ms_per_step": 1000 * tmax / a.steps, "batch_per_gpu": a.batch, "last_cost": cost,
                          "grad_bucket_mfloats": tr.bucket.numel() / 1e6}))
    net.close(); ctx.close()


if __name__ == "__main__":
    main()

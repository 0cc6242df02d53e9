#!/usr/bin/env python3
"""Train-step throughput (BASELINE config C4): yolov4 608x608, batch 8 per GPU,
forward (BN batch statistics) + host yolo loss + backward + [RCCL all-reduce of the
gradient bucket] + SGD update.  One process per GPU (torch.distributed.run for N>1).
Prints one JSON line.  usage: bench_train.py [--gpus N] [--steps K] [--warmup W] [--cfg yolov4] [--batch 8]"""
import argparse, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]


def train_tune_summary(dk, net):
    """What the first step's kernel timing chose: seconds spent and, per GEMM of the step (forward / data gradient /
    weight gradient), how many layers run which configuration."""
    L, C = dk.lib(), dk.C
    L.DkTrainTuneSeconds.restype = C.c_double
    L.DkLayerTrainCfg.restype = C.c_int
    L.DkLayerTrainCfg.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.dk_conv_config_name.restype = C.c_char_p
    L.dk_conv_config_name.argtypes = [C.c_int]
    out = {"seconds": L.DkTrainTuneSeconds()}
    for kind, name in enumerate(("forward", "dgrad", "wgrad")):
        hist = {}
        for i in range(net.n):
            c = L.DkLayerTrainCfg(net.p, i, kind)
            if c < -1:
                continue
            if kind == 2:
                key = {-1: "heuristic", 0: "128x128", 1: "64x128", 2: "128x64", 3: "64x64", 4: "rows3x3", 5: "rows3x3_1wg"}.get(c, str(c))
            else:
                key = "heuristic" if c < 0 else (L.dk_conv_config_name(c) or b"?").decode()
            hist[key] = hist.get(key, 0) + 1
        out[name] = hist
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cfg", default="yolov4")
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    # N > 1 without a launcher: start the N ranks as fresh child processes (one per GPU, rendezvous on 127.0.0.1) BEFORE
    # anything here touches the GPU, relay rank 0's JSON line and exit with the children's code (as bench.py does)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
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
    roofline = None
    if ctx.rank == 0:
        # dominant conv-family kernel of the step (forward convs, data gradients, weight gradients):
        # algorithmic FLOPs / HIP-event time on the kernel's own stream
        C = dk.C
        L.dk_conv_kernel_name.restype = C.c_char_p
        L.dk_conv_kernel_name.argtypes = [C.c_int]
        L.dk_profile_enable(1)
        for _ in range(2):
            tr.step(x, truth)
        out = (C.c_double * (3 * 512))()
        n = L.dk_profile_read(out, 512)
        L.dk_profile_enable(0)
        rows = sorted([(out[3 * i + 2], out[3 * i], out[3 * i + 1], i) for i in range(min(n, 512)) if out[3 * i] > 0], reverse=True)
        ms, launches, gflop, ci = rows[0]
        tot_ms, tot_gf = sum(r[0] for r in rows), sum(r[2] for r in rows)
        roofline = {"bound": "mfma", "achieved": gflop / ms, "peak": 157.3, "unit": "TFLOP/s", "frac": gflop / ms / 157.3,
                    "traffic": None, "kernel": L.dk_conv_kernel_name(ci).decode(), "launches_per_step": launches / 2,
                    "avg_launch_ms": ms / launches,
                    "all_conv_kernels": {"achieved": tot_gf / tot_ms, "ms_per_step": tot_ms / 2, "gflop_per_step": tot_gf / 2},
                    "kernels": [{"kernel": L.dk_conv_kernel_name(r[3]).decode(), "ms_per_step": r[0] / 2, "launches_per_step": r[1] / 2,
                                 "tflops": r[2] / r[0]} for r in rows[:10]]}
        gf_img = {"yolov4": 128.459, "yolov4-tiny": 6.910, "yolov4-csp": 77.003}.get(a.cfg)
        print(json.dumps({"metric": "images/sec %s train step" % a.cfg, "value": rate, "unit": "images/sec", "n_gpus": ctx.world,
                          "ms_per_step": 1000 * tmax / a.steps, "steps": a.steps, "batch_per_gpu": a.batch, "last_cost": cost,
                          "frac_of_fp32_mfma_roofline": (rate * 3 * gf_img * 1e9 / (ctx.world * 157.3e12)) if gf_img else None,
                          "config": {"workload": "%s.cfg %dx%d batch=%d/GPU train step (forward with batch statistics, host yolo loss, "
                                                 "backward, gradient all-reduce, SGD)" % (a.cfg, net.w, net.h, a.batch)},
                          "grad_bucket_mfloats": tr.bucket.numel() / 1e6, "roofline": roofline,
                          "train_tune": train_tune_summary(dk, net), "lib_sha16": __import__("bench").lib_sha16()}))
    net.close(); ctx.close()


if __name__ == "__main__":
    main()

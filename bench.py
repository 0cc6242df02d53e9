#!/usr/bin/env python3
"""bench.py -- YOLOv4 608x608 forward images/sec on N MI355X (one process per GPU).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N
          --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...)

Workload (BASELINE.json metric "images/sec YOLOv4 608x608 fwd", configs[2]):
yolov4.cfg at 608x608, batch 16 PER GPU, BN folded + mish/leaky fused into the
conv epilogue, synthetic input and random-init weights (tools/synth.py).  A step
is one forward pass of the whole hot path (110 implicit-GEMM convs, SPP maxpools,
routes, shortcuts, upsamples, 3 YOLO decodes) over one batch whose input is
already resident in HBM; the decoded heads stay in HBM (`value`); the
PCIe-inclusive rate (H2D input + D2H heads) is reported beside it as
`e2e_images_per_sec`, never as `value`.  Ranks are independent replicas on
disjoint image shards (no data-path collective): scaling = weak.

The JSON line also carries
  roofline      -- the dominant conv kernel (the instantiation with the largest share of
                   the step): ALGORITHMIC FLOPs per launch (the reference's counter
                   2*nweights*oh*ow*batch, src/convolutional_layer.cpp:714 = SURVEY 8d)
                   / its average launch duration from HIP events on the kernel's
                   own stream, against the 157.3 TFLOP/s dense fp32 MFMA peak.  A launch
                   of the Winograd kernel is booked with the DIRECT algorithm's count;
                   `mfma_executed_tflops` says what the MFMA pipe really executed (2.25x
                   fewer).  `traffic` = HBM bytes per launch of that kernel from the
                   committed rocprofv3 --pmc summary of this workload
                   (profiles/round*_c{2,3,5}/traffic_summary.json).
                   `frac_executed_mfma` = executed / peak; `algorithmic_bytes` (input + raw filters +
                   output of the layers that run the kernel, per launch) and `traffic_ratio`; a committed
                   summary is quoted only when its run.json carries the hash of the library loaded now
                   (`lib_sha16`), else `traffic` is null and `traffic_source` says why.
  cpu_baseline  -- the reference's CPU path (oracle/_ref/libref_fast.so when it
                   travelled with the repo, else this repo's oracle port) timed
                   on this box's host cores on yolov4 608x608 b=1.
  other_configs -- (N = 1) the other BASELINE GPU configs in a few seconds each, as child processes:
                   C2 yolov4-tiny 416 b=32, C5 yolov4-csp and yolov4x-mish 512 b=32 fp16 operands,
                   C4 yolov4 608 b=8 train step (tools/bench_train.py): value, ms_per_step, dominant kernel, frac.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: Peak FP32 (matrix), dense
FP16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: Peak BF16/FP16 MFMA, dense (not the 2:1-sparse figure)
CFG = "yolov4"
BATCH_PER_GPU = 16


def lib_sha16():
    """Identity of the HIP library build: first 16 hex digits of the sha256 over the SOURCES it is built from
    (darknet_amd/csrc/**, include/*.h, sorted by path).  (The .so itself is not reproducible byte for byte -- two builds
    of the same sources hash differently -- so a profile is tied to the sources; `make -q` says whether the shipped .so
    is up to date with them.)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = []
    for pat in ("darknet_amd/csrc/**/*.hip", "darknet_amd/csrc/**/*.cpp", "darknet_amd/csrc/**/*.h", "darknet_amd/csrc/Makefile", "include/*.h"):
        files += glob.glob(os.path.join(ROOT, pat), recursive=True)
    for f in sorted(set(files)):
        if os.sep + "build" + os.sep in f:
            continue
        h.update(os.path.relpath(f, ROOT).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic_for(kernel, tag):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary of THIS workload AND THIS BUILD
    (profiles/round*_<tag>/traffic_summary.json, produced by tools/make_profiles.sh with this command line:
    FETCH_SIZE / WRITE_SIZE in separate rocprofv3 --pmc passes).  A summary is used only when the `lib_sha16` its
    run.json recorded equals the hash of the library loaded now -- a profile of an older build of a kernel with the
    same name would otherwise be quoted silently.  Returns (bytes or None, why)."""
    import glob
    if not tag:
        return None, "no committed profile for this workload"
    mine = lib_sha16()
    why = "no committed profile names this kernel"
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_%s" % tag, "traffic_summary.json")), reverse=True):
        try:
            run = json.load(open(os.path.join(os.path.dirname(f), "run.json")))
            if run.get("lib_sha16") != mine:
                if why.startswith("no committed"):
                    why = "the newest committed profile (%s) measured another build (lib_sha16 %s, sources now %s)" % (
                        os.path.relpath(os.path.dirname(f), ROOT), run.get("lib_sha16"), mine)
                continue
            for r in json.load(open(f)):
                if kernel in r["kernel"]:
                    return r["hbm_bytes_per_launch"], os.path.relpath(f, ROOT)
        except Exception:
            pass
    return None, why


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--cfg", default=CFG)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="images per GPU per step")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--half", action="store_true", help="fp16 operands / fp32 accumulate on the layers the reference's rule admits (config C5)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short C2 / C4 / C5 legs of the N = 1 line")
    args = ap.parse_args()

    # N > 1 without a launcher: start the N ranks ourselves as fresh child processes (one per GPU,
    # rendezvous on 127.0.0.1) BEFORE anything here touches the GPU, relay rank 0's JSON line and
    # exit with the children's code.  Under torch.distributed.run (WORLD_SIZE set) this is skipped.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))

    import numpy as np
    import torch  # device sync + torch.distributed (RCCL) only
    import darknet_amd as dk
    from darknet_amd import dist as dkdist
    from darknet_amd import netapi
    import synth

    # DK_BENCH_REHEARSE=1: rehearsal of the N > 1 path on a box with ONE GPU (launcher, rendezvous, sharding,
    # aggregation): gloo instead of RCCL, every rank on device 0.  Never a measurement.
    rehearse = bool(os.environ.get("DK_BENCH_REHEARSE"))
    ctx = dkdist.DistCtx(backend="gloo" if rehearse else "nccl")
    if ctx.world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)"
                         % (args.gpus, ctx.world))
    L = dk.lib()
    if L.CudaGetDeviceCount() < 1:
        raise SystemExit("bench.py: no HIP device visible (there is no CPU fallback)")
    dev_index = 0 if rehearse else ctx.local_rank
    torch.cuda.set_device(dev_index)
    L.cuda_set_device(dev_index)

    if args.half:
        L.DkSetHalf.argtypes = [C.c_int]
        L.DkSetHalf(1)
    tmp = tempfile.mkdtemp(prefix="dkbench%d_" % ctx.rank)
    wpath = os.path.join(tmp, args.cfg + ".weights")
    netapi.synth_weights_for(dk, args.cfg, wpath, seed=2024)
    t_load = time.perf_counter()
    net = netapi.DkNet(dk, netapi.cfg_path(args.cfg), wpath, batch=args.batch)
    load_seconds = time.perf_counter() - t_load   # parse + weights + BN fold + plan + per-layer tile autotune
    # this rank's shard of the global synthetic batch
    lo, hi = dkdist.shard_range(args.batch * ctx.world, ctx.rank, ctx.world)
    x = synth.make_input(hi, net.c, net.h, net.w)[lo:hi]
    L.DkSetPullHeads.argtypes = [C.c_int]
    L.dk_conv_kernel_name.restype = C.c_char_p
    L.dk_conv_kernel_name.argtypes = [C.c_int]
    L.cuda_push_array(L.DkNetworkInputGpu(net.p), x.ctypes.data, x.size)
    L.NetworkSync(net.p)

    net_w, net_h = net.w, net.h

    def step():
        L.NetworkPredictDevice(net.p, None)

    # ---- device-only timed region (value) ---------------------------------
    L.DkSetPullHeads(0)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ctx.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    ctx.barrier()
    dt = time.perf_counter() - t0
    value, tmax = dkdist.aggregate_throughput(ctx, args.batch * args.steps, dt)

    # ---- PCIe-inclusive rate (reported beside, never as value) --------------
    L.DkSetPullHeads(1)
    e2e_steps = max(3, args.steps // 3)
    net.predict(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(e2e_steps):
        net.predict(x)  # H2D input, forward, D2H heads, sync
    e2e_dt = time.perf_counter() - t0
    e2e, _ = dkdist.aggregate_throughput(ctx, args.batch * e2e_steps, e2e_dt)

    # the same frames -> heads path with the input step taken off the critical path (DkNetworkStageFloat: the next
    # batch's pinned copy + H2D run on the staging stream under the current forward; heads still cross PCIe whole)
    def f32_step():
        net.predict_staged()
        net.stage_float(x)
        net.collect()
    net.stage_float(x)
    f32_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(e2e_steps):
        f32_step()
    e2e_staged_dt = time.perf_counter() - t0
    e2e_staged, _ = dkdist.aggregate_throughput(ctx, args.batch * e2e_steps, e2e_staged_dt)

    # ---- u8 frames in, detections out (Mat2Image and the candidate extraction on
    # the device: bytes cross PCIe one way, a few candidate records the other) ----------------
    # synthetic weights have no meaningful confidence scale: pick the threshold that lets
    # ~200 predictors per image through (a busy real scene), from the heads just pulled
    obj = []
    for i in range(net.n):
        f = net.info(i)
        if f["type"] == 17:  # YOLO (LAYER_TYPE, yolo_core.h)
            o = net.output(i)[0].reshape(f["n"], 5 + f["classes"], -1)
            obj.append(o[:, 4, :].ravel())
    obj = np.sort(np.concatenate(obj))
    box_thresh = float(obj[-201]) if obj.size > 201 else 0.25
    L.DkSetPullHeads(0)
    frames = (np.clip(x.reshape(args.batch, net.c, net.h, net.w), 0, 1) * 255).astype(np.uint8)
    frames = np.ascontiguousarray(frames.transpose(0, 2, 3, 1))

    # double-buffered input step: while the forward of batch k runs, batch k+1 is copied to pinned memory
    # and crosses PCIe on the copy stream (DkNetworkStageU8); the boxes of batch k are collected after that
    def u8_step():
        net.predict_staged()      # device Mat2Image of the staged frames + forward (asynchronous)
        net.stage_u8(frames)      # next batch: host copy + H2D, overlapped with the forward above
        n = 0
        for b in range(args.batch):
            n += len(net.boxes(b, box_thresh, max_dets=2048)[0])
        return n

    net.stage_u8(frames)
    u8_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(e2e_steps):
        ndet = u8_step()
    u8_dt = time.perf_counter() - t0
    e2e_u8, _ = dkdist.aggregate_throughput(ctx, args.batch * e2e_steps, u8_dt)

    # ---- roofline of the dominant kernel: HIP events per conv launch ---------
    roofline = None
    if ctx.rank == 0:
        L.DkSetPullHeads(0)
        L.dk_profile_enable(1)
        prof_steps = 3
        for _ in range(prof_steps):
            step()
        out = (C.c_double * (3 * 512))()
        ncfg = L.dk_profile_read(out, 512)
        L.dk_profile_enable(0)
        rows = [(out[3 * i + 2], out[3 * i], out[3 * i + 1], i) for i in range(min(ncfg, 512)) if out[3 * i] > 0]
        rows.sort(reverse=True)
        ms, launches, gflop, ci = rows[0]
        tot_ms = sum(r[0] for r in rows)
        tot_gf = sum(r[2] for r in rows)
        achieved = gflop / ms  # GFLOP / ms = TFLOP/s
        kname = L.dk_conv_kernel_name(ci).decode()
        # profiles/round<N>_{c3,c2,c5}: the BASELINE configs the summaries were taken on
        tag = {"yolov4": "c3", "yolov4-tiny": "c2", "yolov4-csp": "c5"}.get(args.cfg)
        traffic, traffic_src = pmc_traffic_for(kname, tag)
        # the dominant kernel's own arithmetic decides its roof: fp16-operand kernels run on the fp16 MFMA pipe
        peak = FP16_MFMA_PEAK_TFLOPS if "f16" in kname else FP32_MFMA_PEAK_TFLOPS
        # ALGORITHMIC bytes per launch of that kernel (every tensor once): input + raw filters + output
        # (+ the residual a fused shortcut reads) of the layers that run it, launch-weighted
        L.DkLayerConvCfg.argtypes = [C.c_void_p, C.c_int]
        L.DkLayerFused.argtypes = [C.c_void_p, C.c_int]
        alg, nl = 0.0, 0
        for i in range(net.n):
            f = net.info(i)
            if f["type"] != 0 or L.DkLayerConvCfg(net.p, i) != ci // 4:
                continue
            if "wino" in kname and ((f["out_w"] % 2 == 0) != ("true" in kname.split(",")[1])):
                continue   # the paired-store and the single-store instantiation are different kernels
            out_b = 4.0 * args.batch * f["outputs"]
            alg += 4.0 * args.batch * f["inputs"] + 4.0 * f["nweights"] + out_b * (2 if L.DkLayerFused(net.p, i) else 1)
            nl += 1
        alg_per_launch = alg / nl if nl else None
        executed = achieved / 2.25 if "wino" in kname else achieved
        roofline = {
            "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
            "frac": achieved / peak,
            # Winograd launches are booked with the DIRECT algorithm's FLOPs (SURVEY 8d's per-layer figure), so `frac` can
            # exceed what the pipe does: `frac_executed_mfma` is the share of the MFMA peak the pipe really executes
            "mfma_executed_tflops": executed, "frac_executed_mfma": executed / peak,
            "traffic": traffic, "traffic_source": traffic_src,
            "traffic_unit": "HBM bytes per launch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate "
                            "rocprofv3 --pmc passes of this command: tools/make_profiles.sh -> profiles/round*/)",
            "algorithmic_bytes": alg_per_launch,
            "traffic_ratio": (traffic / alg_per_launch) if (traffic and alg_per_launch) else None,
            "kernel": kname,
            "launches_per_step": launches / prof_steps,
            "gflop_per_launch": gflop / launches, "avg_launch_ms": ms / launches,
            "all_conv_kernels": {"achieved": tot_gf / tot_ms, "frac": tot_gf / tot_ms / FP32_MFMA_PEAK_TFLOPS,
                                 "ms_per_step": tot_ms / prof_steps, "gflop_per_step": tot_gf / prof_steps},
            "kernels": [{"kernel": L.dk_conv_kernel_name(r[3]).decode(), "ms_per_step": r[0] / prof_steps,
                         "launches_per_step": r[1] / prof_steps, "tflops": r[2] / r[0]} for r in rows[:8]],
        }

    # ---- CPU baseline on the host cores (rank 0, N = 1 only) -----------------
    cpu_baseline = None
    if ctx.rank == 0 and ctx.world == 1 and not args.no_cpu_baseline:
        def cpu_leg(cfg_name, w, seconds, threads=None):
            env = dict(os.environ)
            if threads:
                env["OMP_NUM_THREADS"] = str(threads)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_baseline.py"),
                                netapi.cfg_path(cfg_name), w, str(seconds)],
                               stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600, env=env)
            return json.loads(r.stdout.decode().strip().splitlines()[-1])
        try:
            cpu_baseline = cpu_leg(args.cfg, wpath, args.cpu_seconds)
            # BASELINE configs[0] (C1): yolov4-tiny 416x416 b=1 on the CPU path, all cores and one thread
            tw = os.path.join(tmp, "yolov4-tiny.weights")
            netapi.synth_weights_for(dk, "yolov4-tiny", tw, seed=2024)
            cpu_baseline["c1_yolov4_tiny_416_b1_all_cores"] = cpu_leg("yolov4-tiny", tw, 4.0)
            cpu_baseline["c1_yolov4_tiny_416_b1_one_thread"] = cpu_leg("yolov4-tiny", tw, 4.0, threads=1)
        except Exception as e:  # the baseline is reported, never required
            log("cpu_baseline failed:", e)

    # ---- the other BASELINE GPU configs, a few seconds each (N = 1 line of the headline workload only): child
    # processes of this script / tools/bench_train.py after this process has released the GPU work above
    other = None
    if (ctx.rank == 0 and ctx.world == 1 and args.cfg == CFG and args.batch == BATCH_PER_GPU and not args.half
            and not args.no_other_configs):
        def leg(cmd):
            try:
                r = subprocess.run([sys.executable] + cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=280)
                d = json.loads(r.stdout.decode().strip().splitlines()[-1])
                rf = d.get("roofline") or {}
                return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d.get("steps"),
                        "workload": d["config"]["workload"], "dominant_kernel": rf.get("kernel"), "frac": rf.get("frac"),
                        "frac_executed_mfma": rf.get("frac_executed_mfma"),
                        "frac_of_fp32_mfma_roofline": d.get("frac_of_fp32_mfma_roofline")}
            except Exception as e:
                return {"error": str(e)[:200]}
        me = os.path.abspath(__file__)
        common = ["--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-other-configs"]
        net.close()   # (12 GB of activations: the legs get the whole card)
        net = None
        other = {
            "c2_yolov4_tiny_416_b32_fwd": leg([me, "--cfg", "yolov4-tiny", "--batch", "32"] + common),
            "c5_yolov4_csp_512_b32_half_fwd": leg([me, "--cfg", "yolov4-csp", "--batch", "32", "--half"] + common),
            "c5_yolov4x_mish_512_b32_half_fwd": leg([me, "--cfg", "yolov4x-mish", "--batch", "32", "--half"] + common),
            "c4_yolov4_608_b8_train_step": leg([os.path.join(ROOT, "tools", "bench_train.py"), "--steps", "10", "--warmup", "3"]),
        }

    if ctx.rank == 0:
        res = {
            "metric": "images/sec YOLOv4 608x608 fwd" if args.cfg == "yolov4" else "images/sec %s fwd" % args.cfg,
            "value": value, "unit": "images/sec", "n_gpus": ctx.world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * tmax / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16 operands / f32 accumulate (eligible 3x3 layers), f32 elsewhere" if args.half else "f32", "data": "synthetic",
            "config": {"workload": "%s.cfg %dx%d batch=%d/GPU forward, BN folded, bias+mish/leaky fused%s"
                                   % (args.cfg, net_w, net_h, args.batch,
                                      " (BASELINE configs[2])" if (args.cfg == "yolov4" and args.batch == 16 and not args.half) else ""),
                       "global_batch": args.batch * ctx.world,
                       "parallelism": "batch-sharded replicas x%d, no data-path collective" % ctx.world},
            "frac_of_fp32_mfma_roofline": value * 128.459e9 / (ctx.world * FP32_MFMA_PEAK_TFLOPS * 1e12)
            if args.cfg == "yolov4" else None,
            "gflop_per_image": {"yolov4": 128.459, "yolov4-tiny": 6.910, "yolov4-csp": 77.003}.get(args.cfg),
            "e2e_images_per_sec": e2e,
            "e2e_staged_float_frames_images_per_sec": e2e_staged,
            "e2e_u8_frames_to_boxes_images_per_sec": e2e_u8,
            "e2e_u8_note": "u8 HWC frames -> device Mat2Image -> forward -> device candidate compaction -> "
                           "Detection arrays for every image; threshold passes %.0f predictors/image" % (ndet / args.batch),
            "load_seconds_incl_autotune": load_seconds,
            "lib_sha16": lib_sha16(),
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "other_configs": other,
        }
        print(json.dumps(res), flush=True)
    if net is not None:
        net.close()
    ctx.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Time the CPU path beside the GPU numbers (bench.py's `cpu_baseline` leg).

kind "reference": oracle/_ref/libref_fast.so, i.e. the reference's own sources
compiled the way it ships (-Ofast -mavx2 -mfma, OpenMP), driven through its public
API (LoadNetwork / NetworkPredict).  Only when that library travelled with the
repo.  kind "port": this repo's oracle (oracle/orc_ops.c, scalar, OpenMP rows).
Prints one JSON object on the last line of stdout.
usage: cpu_baseline.py CFG WEIGHTS BUDGET_SECONDS
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def host_cores():
    """CPU share actually available: affinity mask capped by the cgroup quota
    (an OpenMP team as large as os.cpu_count() oversubscribes a container)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except Exception:
        pass
    return min(n, 64)


def main():
    cfg, weights, budget = sys.argv[1], sys.argv[2], float(sys.argv[3])
    import synth
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or host_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)
    import reflib
    from oracle import orc_net as O
    net = O.parse_cfg(cfg)
    x = synth.make_input(1, net.c, net.h, net.w)
    if reflib.available("fast"):
        kind = "reference"
        rn = reflib.RefNet(cfg, weights, train=False, kind="fast")
        run = lambda: rn.predict(x)
    else:
        kind = "port"
        onet = O.load_network(cfg, weights, batch=1)
        cores = O.lib().orc_num_threads()
        run = lambda: O.forward(onet, x)
    run()  # warm-up
    t0 = time.time()
    n = 0
    while True:
        run()
        n += 1
        if time.time() - t0 >= budget or n >= 50:
            break
    dt = time.time() - t0
    print(json.dumps(dict(value=n / dt, unit="images/sec", cores=cores, kind=kind,
                          sample="%s b=1 forward x%d (%.1f s), same synthetic weights/input" % (
                              os.path.basename(cfg), n, dt))))


if __name__ == "__main__":
    main()

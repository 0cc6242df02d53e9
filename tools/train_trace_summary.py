#!/usr/bin/env python3
"""Per-kernel time of the steady-state training steps from a rocprofv3 kernel trace (csv): the steps are delimited
by the multi-tensor SGD launch that ends each of them, so the first step's kernel timing runs and the warm-up are
left out.  usage: train_trace_summary.py <kernel_trace.csv> [steps=5]  ->  markdown table on stdout"""
import csv, re, sys
from collections import defaultdict


def main():
    path, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 5
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))))
    ends = [i for i, r in enumerate(rows) if "sgd_multi_kernel" in r[2]]
    if len(ends) < steps + 1:
        sys.exit("not enough steps in the trace")
    lo, hi = ends[-steps - 1] + 1, ends[-1] + 1
    sel = rows[lo:hi]
    span = (sel[-1][1] - sel[0][0]) / 1e6 / steps
    tot, cnt = defaultdict(float), defaultdict(int)
    for s, e, n in sel:
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"^void ", "", n)
        n = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)
        tot[n] += (e - s) / 1e6
        cnt[n] += 1
    busy = sum(tot.values()) / steps
    print("steps %d: %.2f ms per step wall (kernel trace span), %.2f ms kernel time per step, %d launches per step\n" %
          (steps, span, busy, len(sel) // steps))
    print("| kernel | launches/step | avg us | ms/step | share |\n|---|---|---|---|---|")
    for n, t in sorted(tot.items(), key=lambda kv: -kv[1]):
        print("| `%s` | %.1f | %.1f | %.3f | %.1f %% |" % (n, cnt[n] / steps, 1e3 * t / cnt[n], t / steps, 100 * t / steps / busy))
    groups = [("weight gradient", "conv_wgrad|wgrad_fold"), ("forward / data-gradient convolutions", "conv_igemm|conv3x3|conv1x1"),
              ("batch norm (+ activation) kernels", "bn_|chan_"), ("copies / fills", "copyBuffer|fillBuffer"),
              ("weight transposes / Winograd filter transforms", "transpose_w|wino_weights|train_prep")]
    print("\n| group | ms/step |\n|---|---|")
    rest = busy
    for g, pat in groups:
        t = sum(v for k, v in tot.items() if re.search(pat, k)) / steps
        rest -= t
        print("| %s | %.2f |" % (g, t))
    print("| everything else | %.2f |" % rest)


if __name__ == "__main__":
    main()

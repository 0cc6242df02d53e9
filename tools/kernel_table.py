#!/usr/bin/env python3
"""Per-kernel table of the default bench command from rocprofv3 outputs (tools/make_profiles.sh):
time share, HBM GB/s (PMC FETCH_SIZE x2 + WRITE_SIZE over the kernel's own duration) and MFMA pipe
utilisation (PMC SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs)).
usage: kernel_table.py OUTDIR > table.md"""
import collections, csv, glob, os, sys
out = sys.argv[1]
CLK, SIMDS = 2.4e9, 1024


def per_kernel(counter_dir, counter):
    acc = collections.defaultdict(lambda: [0.0, 0.0, 0])   # value sum, duration ns sum, launches
    cf = glob.glob(os.path.join(out, counter_dir, "*", "*_counter_collection.csv"))
    kf = glob.glob(os.path.join(out, counter_dir, "*", "*_kernel_trace.csv"))
    if not cf or not kf:
        return acc
    dur = {}
    for r in csv.DictReader(open(kf[0])):
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for r in csv.DictReader(open(cf[0])):
        if r["Counter_Name"] != counter:
            continue
        a = acc[r["Kernel_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += dur.get(r["Dispatch_Id"], 0)
        a[2] += 1
    return acc


fetch = per_kernel("FETCH_SIZE", "FETCH_SIZE")
write = per_kernel("WRITE_SIZE", "WRITE_SIZE")
mfma = per_kernel("MFMA", "SQ_VALU_MFMA_BUSY_CYCLES")
stats = list(csv.DictReader(open(os.path.join(out, "kernel_stats.csv"))))
tot = sum(float(r["TotalDurationNs"]) for r in stats)
print("| kernel | calls | avg us | share | HBM GB/s | MFMA pipe busy |")
print("|---|---|---|---|---|---|")
for r in stats:
    k = r["Name"]
    share = 100 * float(r["TotalDurationNs"]) / tot
    if share < 0.3:
        continue
    f, w, m = fetch.get(k), write.get(k), mfma.get(k)
    gbs = "-"
    if f and w and f[1] and w[1]:
        gbs = "%.0f" % ((2.0 * f[0] * 1024 / f[1]) + (w[0] * 1024 / w[1]))   # bytes per ns = GB/s
    util = "-"
    if m and m[1]:
        util = "%.1f %%" % (100.0 * m[0] / (m[1] * 1e-9 * CLK * SIMDS))
    print("| `%s` | %s | %.1f | %.1f %% | %s | %s |" % (k.replace("void ", "").replace("(ConvArgs)", ""), r["Calls"],
                                                    float(r["AverageNs"]) / 1e3, share, gbs, util))

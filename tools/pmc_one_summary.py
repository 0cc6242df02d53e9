#!/usr/bin/env python3
"""Summarise tools/pmc_one.sh: per counter the mean over the conv launches, plus derived shares."""
import collections, csv, glob, os, sys
out = sys.argv[1]
vals = collections.defaultdict(list)
dur = []
for g in sorted(glob.glob(os.path.join(out, "g*"))):
    if not os.path.isdir(g):
        continue
    cf = glob.glob(os.path.join(g, "*", "*_counter_collection.csv"))
    kf = glob.glob(os.path.join(g, "*", "*_kernel_trace.csv"))
    if not cf:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(cf[0])):
        if "conv" not in r["Kernel_Name"]:
            continue
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    for d, cs in per.items():
        for k, v in cs.items():
            vals[k].append(v)
    if kf:
        for r in csv.DictReader(open(kf[0])):
            if "conv" in r["Kernel_Name"]:
                dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
m = {k: sum(v) / len(v) for k, v in vals.items()}
us = sum(dur) / len(dur) / 1e3 if dur else 0
lines = ["kernel avg %.1f us" % us]
for k in sorted(m):
    lines.append("%-28s %.4g" % (k, m[k]))
wc = m.get("SQ_WAVE_CYCLES", 0)
if wc:
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
        if k in m:
            lines.append("%-28s %.1f %% of wave-cycles" % (k, 100 * m[k] / wc))
if "SQ_VALU_MFMA_BUSY_CYCLES" in m and us:
    lines.append("MFMA pipe busy %.1f %% (of duration x 2.4 GHz x 1024 SIMDs)" % (100 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (us * 1e-6 * 2.4e9 * 1024)))
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))

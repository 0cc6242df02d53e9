#!/usr/bin/env python3
"""Summarise the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh per kernel."""
import collections, csv, glob, json, os, sys
out = sys.argv[1]
res = collections.defaultdict(lambda: {"launches": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(out, c, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            k = r["Kernel_Name"]
            res[k][c] += float(r["Counter_Value"])
            if c == "FETCH_SIZE":
                res[k]["launches"] += 1
rows = []
for k, v in res.items():
    if not v["launches"]:
        continue
    fetch_b = 2.0 * v["FETCH_SIZE"] * 1024 / v["launches"]   # gfx950: x2 correction, KiB -> bytes
    write_b = v["WRITE_SIZE"] * 1024 / v["launches"]
    rows.append(dict(kernel=k, launches=v["launches"], fetch_bytes_per_launch=fetch_b,
                     write_bytes_per_launch=write_b, hbm_bytes_per_launch=fetch_b + write_b))
rows.sort(key=lambda r: -r["hbm_bytes_per_launch"] * r["launches"])
json.dump(rows, open(os.path.join(out, "traffic_summary.json"), "w"), indent=1)
for r in rows[:12]:
    print("%-70s n=%5d fetch %8.1f MB write %8.1f MB per launch" % (r["kernel"][:70], r["launches"], r["fetch_bytes_per_launch"] / 1e6, r["write_bytes_per_launch"] / 1e6))

#!/bin/bash
# FETCH_SIZE vs request counters on one direct-kernel layer
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2h1
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1
grep -i -E "TCC_EA0?_RDREQ|TCC_REQ|FETCH_SIZE|TCC_EA_RD|TCC_BUBBLE|TCC_READ" $O/counters.txt | head -40
i=0
for grp in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_READ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/g$i -- python3 $R/tools/conv_one.py 16 128 76 76 128 3 1 1 17 14 6 > $O/g$i.log 2>&1 || echo "group $i failed"
done
python3 - <<PY
import csv, glob, collections
for g in sorted(glob.glob("$O/g*")):
    import os
    if not os.path.isdir(g): continue
    cf = glob.glob(os.path.join(g, "*", "*_counter_collection.csv"))
    if not cf: print(g, "no counters"); continue
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(cf[0])):
        if "conv3x3_direct" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in per.items():
        print(os.path.basename(g), k, "mean per launch %.6g over %d launches" % (sum(v) / len(v), len(v)))
PY

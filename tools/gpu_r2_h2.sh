#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2h2
mkdir -p $O; cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_net.py -q -m gpu -k "staged or extraction" > $O/test.log 2>&1; echo "pytest rc=$?"; tail -3 $O/test.log
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2h2/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['e2e_images_per_sec'], d['e2e_u8_frames_to_boxes_images_per_sec'], d['roofline']['frac'], d['roofline'].get('traffic'))
PY
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/g$i -- python3 $R/tools/conv_one.py 16 128 76 76 128 3 1 1 17 14 6 > $O/g$i.log 2>&1 || { echo "group $i failed"; tail -3 $O/g$i.log; }
done
python3 - <<PY
import csv, glob, collections, os
for g in sorted(glob.glob("$O/g*")):
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

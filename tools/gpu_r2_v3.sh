#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2v3
mkdir -p $R/$O
cd $R
DK_TRAIN_OVERLAP=0 DK_TRAIN_TIMING=1 timeout -k 10 200 python tools/bench_train.py --steps 6 --warmup 2 > $O/train_timing.json 2> $O/train_timing.err; echo "rc=$?"
grep "train timing" $O/train_timing.err | tail -4
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python3 $R/tools/bench_train.py --steps 8 --warmup 2 > $R/$O/prof.log 2>&1; echo "prof rc=$?"
cd $R
cp $O/prof/*/*kernel_stats.csv $O/kernel_stats.csv
python - <<'PY'
import csv, glob
# GPU busy time vs wall span in the kernel trace (last 60 % of the run = timed steps)
f = glob.glob('gpurun_out/r2v3/prof/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows)
n = len(ev); ev = ev[int(n*0.5):]
span = ev[-1][1]-ev[0][0]; busy = sum(e-s for s,e in ev)
print("tail half of trace: span %.1f ms, kernel busy %.1f ms (%.1f %%), kernels %d" % (span/1e6, busy/1e6, 100*busy/span, len(ev)))
PY
rm -rf $O/prof
head -45 $O/kernel_stats.csv | cut -c1-160

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2v4
mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python3 $R/tools/bench_train.py --steps 8 --warmup 2 > $R/$O/prof.log 2>&1; echo "prof rc=$?"
cd $R
python tools/train_trace_summary.py $(ls $O/prof/*/*kernel_trace.csv | head -1) 6 > $O/steady_table.md
rm -rf $O/prof
head -60 $O/steady_table.md | cut -c1-150; tail -9 $O/steady_table.md

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2o1
mkdir -p $R/$O
cd $R
S=$(date +%s); DK_BENCH_REHEARSE=1 timeout -k 10 500 python bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench2.json 2> $O/bench2.err; echo "bench --gpus 2 (rehearsal) rc=$? wall $(( $(date +%s) - S )) s"
tail -3 $O/bench2.err | cut -c1-300
cut -c1-400 $O/bench2.json

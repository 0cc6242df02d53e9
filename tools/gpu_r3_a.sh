#!/bin/bash
# round 3 call a: fp32 MFMA shape microbenchmark + baseline bench of the round-2 library
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r3a
mkdir -p $R/$O
cd $R
timeout -k 10 200 tools/ubench/mfma_shape > $O/shape.txt 2>&1; echo "shape rc=$?"
cat $O/shape.txt
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -c 1500 $O/bench.json

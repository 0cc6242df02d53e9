#!/bin/bash
# HBM traffic of the conv kernels per launch via rocprofv3 PMC, collected exactly
# as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc
# passes (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2), kernel-trace only
# (no sys/hip/hsa trace), program directly after `--`.
# gfx950 correction: FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced
# stream -> doubled; WRITE_SIZE is exact.  Units: KiB.
# usage (on the GPU box): tools/pmc_traffic.sh OUTDIR
set -e
OUT=${1:-gpurun_out/pmc_traffic}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
export DK_TUNE_FILE=$R/$OUT/tune.txt
python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$OUT/warm.json 2> $R/$OUT/warm.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$OUT/$c -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$OUT/$c.log 2>&1
done
python3 $R/tools/pmc_summarize.py $R/$OUT

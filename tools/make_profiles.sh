#!/bin/bash
# Round artifacts (GPU box): kernel-trace stats + PMC HBM traffic of the default bench
# command, with the per-layer tile choice cached so that no tuning launches are traced.
# usage: tools/make_profiles.sh OUTDIR     (then copy the summaries into profiles/)
set -e
OUT=${1:-gpurun_out/final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
export DK_TUNE_FILE=$R/$OUT/tune.txt
python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/$OUT/warm.json 2> $R/$OUT/warm.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/$OUT/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$OUT/$c -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$OUT/$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/$OUT/MFMA -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$OUT/MFMA.log 2>&1
python3 $R/tools/pmc_summarize.py $R/$OUT
cp $R/$OUT/stats/*/*kernel_stats.csv $R/$OUT/kernel_stats.csv
python3 $R/tools/kernel_table.py $R/$OUT > $R/$OUT/kernel_table.md

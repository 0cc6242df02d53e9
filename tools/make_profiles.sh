#!/bin/bash
# Round artifacts (GPU box): rocprofv3 kernel-trace stats + PMC HBM traffic + MFMA-busy of ONE
# measurement command, with the per-layer tile choice cached so that no tuning launches are traced.
# usage: tools/make_profiles.sh OUTDIR [script.py args ...]      (default: bench.py, the BASELINE configs[2] run)
#        then copy OUTDIR/{kernel_stats.csv,kernel_table.md,traffic_summary.json,run.json} into profiles/
set -e
OUT=${1:-gpurun_out/final}
shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
if [ $# -eq 0 ]; then set -- bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs; fi
SCRIPT=$R/$1
shift
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
export DK_TUNE_FILE=$R/$OUT/tune.txt
python3 $SCRIPT "$@" > $R/$OUT/run.json 2> $R/$OUT/run.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats -- python3 $SCRIPT "$@" > $R/$OUT/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$OUT/$c -- python3 $SCRIPT "$@" > $R/$OUT/$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/$OUT/MFMA -- python3 $SCRIPT "$@" > $R/$OUT/MFMA.log 2>&1
python3 $R/tools/pmc_summarize.py $R/$OUT
cp $R/$OUT/stats/*/*kernel_stats.csv $R/$OUT/kernel_stats.csv
python3 $R/tools/kernel_table.py $R/$OUT > $R/$OUT/kernel_table.md

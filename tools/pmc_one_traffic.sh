#!/bin/bash
# Dev tool (GPU box): HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of ONE conv layer/config.
# usage: tools/pmc_one_traffic.sh OUTDIR batch c h w n size stride pad act cfg
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$1; shift
mkdir -p $R/$OUT; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$OUT/$c -- python3 $R/tools/conv_one.py "$@" 6 > $R/$OUT/$c.log 2>&1 || exit 1
done
python3 $R/tools/pmc_summarize.py $R/$OUT

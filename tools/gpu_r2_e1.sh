#!/bin/bash
# round 2, GPU session E1: the whole GPU suite
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2e
mkdir -p $R/$O
cd $R
timeout -k 10 1000 python -m pytest tests -q -m gpu --durations=15 > $O/test.log 2>&1; echo "pytest rc=$?" | tee -a $O/test.log
grep -E "passed|failed|FAILED|ERROR" $O/test.log | tail -20
tail -30 $O/test.log

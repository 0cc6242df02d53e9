#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout -k 10 500 bash tools/pmc_one.sh gpurun_out/r2f3/pmc_wino_256x38 16 256 38 38 512 3 1 1 1 22 > gpurun_out/r2f3_wino.log 2>&1; echo rc=$?
tail -30 gpurun_out/r2f3_wino.log

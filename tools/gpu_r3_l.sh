#!/bin/bash
# round 3 call l: round artifacts for C3 (rocprofv3 stats + PMC traffic + MFMA busy of the bench command)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
bash tools/make_profiles.sh gpurun_out/r3l_c3 > gpurun_out/r3l_c3.log 2>&1; echo "rc=$?"
tail -15 gpurun_out/r3l_c3.log
ls gpurun_out/r3l_c3

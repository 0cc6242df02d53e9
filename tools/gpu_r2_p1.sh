#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=gpurun_out/r2p1
mkdir -p $R/$OUT; cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/$OUT/g$i -- python3 $R/tools/wgrad_one.py 8 128 76 76 128 3 1 1 6 > $R/$OUT/g$i.log 2>&1 || exit 1
done
python3 $R/tools/pmc_one_summary.py $R/$OUT

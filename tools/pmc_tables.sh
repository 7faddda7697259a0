#!/bin/bash
# PMC passes over the percentile-rank fusion with 27,943-entry tables (tables.hip).  Run on the GPU box from the repo root.
set -e
export TMPDIR=/tmp
TAG=${1:-}
OUT=$PWD/gpurun_out/pmc_tables$TAG
rm -rf $OUT; mkdir -p $OUT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM"
P3="SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_THREAD_CYCLES_VALU"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $P --kernel-trace -d $OUT/pass$i -o p --output-format csv -- python3 $OLDPWD/tools/run_tables.py > $OUT/pass$i.log 2>&1) || { tail -5 $OUT/pass$i.log; }
done
python3 tools/pmc_summary.py gpurun_out/pmc_tables$TAG.json $OUT/pass1 $OUT/pass2 $OUT/pass3 --match bigtab

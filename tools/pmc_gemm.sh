#!/bin/bash
# PMC passes over the score GEMM and the vendor fp32 GEMM on the same operands (Q = 1024, N = 276,307, d = 768).
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_gemm
rm -rf $OUT; mkdir -p $OUT
P0=""
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM"
P3="GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC"
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/pass0 -o p --output-format csv -- python3 $OLDPWD/tools/run_gemm.py > $OUT/pass0.log 2>&1) || tail -5 $OUT/pass0.log
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $P --kernel-trace -d $OUT/pass$i -o p --output-format csv -- python3 $OLDPWD/tools/run_gemm.py > $OUT/pass$i.log 2>&1) || { tail -5 $OUT/pass$i.log; }
done
python3 tools/pmc_summary.py gpurun_out/pmc_gemm.json $OUT/pass1 $OUT/pass2 $OUT/pass3
find $OUT/pass0 -name "*kernel_stats.csv" -exec cat {} \;

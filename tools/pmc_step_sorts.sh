#!/bin/bash
# SQ counters of the bench step's three row sorts (dpr_rank: fp32 keys; bm25_rank: fp64 keys; final_order: fp64 keys formed from rank planes on load)
# as they run IN the step -- separate rocprofv3 --pmc passes over `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs`.
# Run on the GPU box from the repo root; summary -> gpurun_out/pmc_step_sorts.json (copied to profiles/r05_pmc_step_sorts.json).
set -e
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/pmc_step_sorts
rm -rf $OUT; mkdir -p $OUT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM"
P3="SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_THREAD_CYCLES_VALU"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $P --kernel-trace -d $OUT/pass$i -o p --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs > $OUT/pass$i.log 2>&1) || { tail -5 $OUT/pass$i.log; }
done
python3 tools/pmc_summary.py gpurun_out/pmc_step_sorts.json $OUT/pass1 $OUT/pass2 $OUT/pass3 --match sort_rows_kernel

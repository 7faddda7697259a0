#!/bin/bash
# Everything under profiles/ that follows the default build, in one go (run on the GPU box from the repo root; ~10 minutes).
# Outputs land in gpurun_out/regen/; copy what is to be kept into profiles/ under the round's prefix.
set -o pipefail
O=gpurun_out/regen
mkdir -p $O
step() { echo "== $1" >&2; shift; timeout -k 10 "$@"; rc=$?; if [ $rc -ne 0 ]; then echo "step failed rc=$rc" >&2; exit $rc; fi; }
step "bench default"   420 python bench.py > $O/bench_default.json 2> $O/bench_default.err
step "bench Q=195"     200 python bench.py --queries 195 --no-configs > $O/bench_q195.json 2> $O/bench_q195.err
step "kernel bench"    300 python tools/bench_kernels.py > $O/kernel_bench.jsonl 2> $O/kernel_bench.err
step "sort pass cost"  120 python tools/sort_pass_cost.py > $O/sort_pass_cost.log 2>&1
step "colbert amp"     200 python tools/bench_colbert_amp.py 1024 > $O/colbert_amp.jsonl 2> $O/colbert_amp.err
step "colbert amp 195" 200 python tools/bench_colbert_amp.py 195 >> $O/colbert_amp.jsonl 2>> $O/colbert_amp.err
step "profile bench"   600 bash tools/profile_bench.sh > $O/profile_bench.log 2>&1
step "pmc sort"        300 bash tools/pmc_sort.sh > $O/pmc_sort.log 2>&1
step "pmc tables"      300 bash tools/pmc_tables.sh > $O/pmc_tables.log 2>&1
step "tables a/b"      300 python tools/run_tables_ab.py 2 > $O/tables_ab.jsonl 2>&1
step "boundary"        300 python tools/bench_boundary.py 32 > $O/boundary.json 2>&1
cp gpurun_out/hbm_traffic.json $O/ 2>/dev/null
cp gpurun_out/pmc_sort.json gpurun_out/pmc_tables.json $O/ 2>/dev/null
find gpurun_out/prof_stats -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
ls -la $O >&2

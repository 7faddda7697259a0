#!/bin/bash
# Round-6 profiles of the final build in one GPU-box call (from the repo root; ~8 minutes): outputs in gpurun_out/r06/, copied into profiles/r06_*.
set -o pipefail
O=gpurun_out/r06
mkdir -p $O
step() { echo "== $1 $(date +%T)" >&2; shift; timeout -k 10 "$@"; rc=$?; if [ $rc -ne 0 ]; then echo "step failed rc=$rc" >&2; exit $rc; fi; }
step "bench default"   420 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cp bench_detail.json $O/bench_default_detail.json
step "bench Q=195"     200 python bench.py --queries 195 --no-configs > $O/bench_q195.json 2> $O/bench_q195.err; cp bench_detail.json $O/bench_q195_detail.json
step "bench mmarco n1" 400 python bench.py --workload mmarco --steps 3 --warmup 1 > $O/bench_mmarco_n1.json 2> $O/bench_mmarco_n1.err; cp bench_detail.json $O/bench_mmarco_n1_detail.json
step "launcher 2 gloo" 400 env FUSION_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --rehearsal --mmarco-docs 2000000 --steps 2 --warmup 1 > $O/bench_launcher_2ranks_1gpu_gloo.json 2> $O/bench_launcher.err
step "kernel bench"    300 python tools/bench_kernels.py > $O/kernel_bench.jsonl 2> $O/kernel_bench.err
step "sort vs zeros"   200 python tools/bench_sort_zeros.py > $O/sort_zeros.json 2> $O/sort_zeros.err
step "bm25 module"     200 python tools/bench_bm25_tune.py > $O/bm25_module.json 2> $O/bm25_module.err
step "sparse splade"   200 python tools/bench_sparse.py > $O/sparse_splade.jsonl 2> $O/sparse_splade.err
step "bm25 expr A/B"   200 python bench.py --no-configs --no-cpu-baseline --steps 10 --bm25-per-posting-expression > $O/bench_bm25_expression.json 2> $O/bench_bm25_expression.err
step "profile bench"   600 bash tools/profile_bench.sh > $O/profile_bench.log 2>&1
cp gpurun_out/hbm_traffic.json $O/ 2>/dev/null
find gpurun_out/prof_stats -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
ls -la $O >&2

#!/bin/bash
# Round profiles of the default bench command (run on the GPU box from the repo root):
#   1. rocprofv3 --kernel-trace --stats  -> gpurun_out/prof_stats/   (per-kernel average durations)
#   2. rocprofv3 --pmc FETCH_SIZE        -> gpurun_out/prof_fetch/   (separate passes, as MI355X_MICROARCH.md prescribes)
#   3. rocprofv3 --pmc WRITE_SIZE        -> gpurun_out/prof_write/
# and folds 2 + 3 into gpurun_out/hbm_traffic.json (tools/pmc_traffic.py: FETCH_SIZE doubled, the gfx950 correction).
set -e
export TMPDIR=/tmp
ROOT=$PWD
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-configs"
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write
(cd /tmp && rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_stats -o b --output-format csv -- python3 $ROOT/bench.py $ARGS > $ROOT/gpurun_out/prof_stats.log 2>&1) || tail -5 gpurun_out/prof_stats.log
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $ROOT/gpurun_out/prof_fetch -o b --output-format csv -- python3 $ROOT/bench.py $ARGS > $ROOT/gpurun_out/prof_fetch.log 2>&1) || tail -5 gpurun_out/prof_fetch.log
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $ROOT/gpurun_out/prof_write -o b --output-format csv -- python3 $ROOT/bench.py $ARGS > $ROOT/gpurun_out/prof_write.log 2>&1) || tail -5 gpurun_out/prof_write.log
python3 tools/pmc_traffic.py gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/hbm_traffic.json "python bench.py $ARGS"
ls gpurun_out/prof_stats

for i in 1 2; do
FZ_SORT_ZERO_COMPACT=0 python tools/bench_sort_zeros.py 2>&1 | grep -E "f64 zeros=0.0|n=22400|n=11200"
FZ_SORT_ZERO_COMPACT=0 FUSION_AMD_LIB=$PWD/fusion_amd/libfusion_hip_ng.so python tools/bench_sort_zeros.py 2>&1 | grep -E "f64 zeros=0.0|n=22400|n=11200"
done

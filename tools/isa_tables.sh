#!/bin/bash
# Compile tables.hip with resource remarks and keep the ISA of the percentile kernel in /tmp/k0.s (build container diagnostics).
cd /root/repo/fusion_amd/csrc || exit 1
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -c tables.hip -o /tmp/tables.o --save-temps=obj -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A12 "error\|Function Name: _ZN2fz22fuse_nsf_bigtab" | grep -i "error\|name\|VGPRs:\|VGPRs Spill\|ScratchSize"
S=/tmp/tables-hip-amdgcn-amd-amdhsa-gfx950.s
awk '/^_ZN2fz22fuse_nsf_bigtab_kernelILb0/ {f=1} f {print NR": "$0} f && /s_endpgm/ {exit}' $S > /tmp/k0.s
wc -l /tmp/k0.s
grep -n "scratch_" /tmp/k0.s | awk -F: '{print $1}' | awk 'NR==1{s=$1;p=$1;next} {if($1-p>40){print s"-"p; s=$1} p=$1} END{print s"-"p}' | head -40

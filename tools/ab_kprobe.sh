#!/bin/bash
# A/B of the mask-path kernels on the kprobe query sets, one session: default build, variants (NXS_GPU_LIB), k_scanm
# usage: tools/ab_kprobe.sh "<sets>" <variant suffixes...>
mkdir -p gpurun_out/r5
SETS=${1:-X,K,E,M}; shift
C=$PWD/nxsearch_amd/csrc
echo "== default"; timeout -k 10 200 python tools/kprobe.py --sets $SETS --reps 5 2>&1 | grep -v amdgpu.ids
for v in "$@"; do
  echo "== variant $v"; NXS_GPU_LIB=$C/libnxsearch_gpu_$v.so timeout -k 10 200 python tools/kprobe.py --sets $SETS --reps 5 2>&1 | grep -v amdgpu.ids
done
echo "== k_scanm (NXS_GPU_NOSCANS)"; NXS_GPU_NOSCANS=1 timeout -k 10 200 python tools/kprobe.py --sets $SETS --reps 5 2>&1 | grep -v amdgpu.ids
if [ -f $C/libnxsearch_gpu_stats.so ]; then
  echo "== stats"; NXS_GPU_LIB=$C/libnxsearch_gpu_stats.so STATS_SETS=$SETS timeout -k 10 200 python tools/scans_stats.py 2>&1 | grep -v amdgpu.ids
fi

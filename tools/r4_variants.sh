#!/bin/bash
# kprobe on the default build and on library variants, one GPU session (boxes differ by ~10 %)
set -u
SETS=${SETS:-K,E,O}
echo "== default"; timeout -k 10 300 python tools/kprobe.py --sets $SETS 2>&1 | grep -v amdgpu.ids
for v in "$@"; do echo "== variant $v"; NXS_GPU_LIB=$PWD/nxsearch_amd/csrc/libnxsearch_gpu_$v.so timeout -k 10 300 python tools/kprobe.py --sets $SETS 2>&1 | grep -v amdgpu.ids; done
echo "== k_scanm"; NXS_GPU_NOSCANB=1 timeout -k 10 300 python tools/kprobe.py --sets $SETS 2>&1 | grep -v amdgpu.ids

#!/bin/bash
# quick loop: mask-path parity tests, then event stats + kprobe A/B of k_scanb
set -u
out=$PWD/gpurun_out; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "${TESTS:-mask_path or every_scan_path or sparse_terms or pipelined_string}" > $out/r4_quick_tests.log 2>&1
rc=$?; tail -3 $out/r4_quick_tests.log; [ $rc -ne 0 ] && exit $rc
NXS_GPU_LIB=$PWD/nxsearch_amd/csrc/libnxsearch_gpu_stats.so timeout -k 10 300 python tools/scanb_stats.py 2>&1 | grep -v amdgpu.ids
SETS=${SETS:-K,E,O,C}
timeout -k 10 300 python tools/kprobe.py --sets $SETS 2>&1 | grep -v amdgpu.ids
for v in ${VARIANTS:-}; do echo "== variant $v"; NXS_GPU_LIB=$PWD/nxsearch_amd/csrc/libnxsearch_gpu_$v.so timeout -k 10 300 python tools/kprobe.py --sets $SETS 2>&1 | grep -v amdgpu.ids; done

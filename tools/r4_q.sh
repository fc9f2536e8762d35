#!/bin/bash
set -u
out=$PWD/gpurun_out; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "${TESTS:-every_scan_path or sparse_terms or random_corpora or incremental_refresh_interleaved or golden or querylogic}" > $out/r4_q_tests.log 2>&1
rc=$?; tail -3 $out/r4_q_tests.log; [ $rc -ne 0 ] && exit $rc
SETS=D,G,H,I timeout -k 10 300 python tools/kprobe.py --sets D,G,H,I,J 2>&1 | grep -v amdgpu.ids
echo "== NOBLKMAP"; NXS_GPU_NOBLKMAP=1 timeout -k 10 300 python tools/kprobe.py --sets D,G,H,I,J 2>&1 | grep -v amdgpu.ids
. tools/ab2.sh
run scanq - ; run noblk - NXS_GPU_NOBLKMAP=1 ; run scanq - ; run noblk - NXS_GPU_NOBLKMAP=1

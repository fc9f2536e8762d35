#!/bin/bash
# round-4 dev loop on the GPU box: GPU tier, then kernel probes A/B (k_scanb vs k_scanm)
set -u
out=$PWD/gpurun_out
mkdir -p $out
what=${1:-all}
if [ "$what" = all ] || [ "$what" = tests ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/r4_tests.log 2>&1
  rc=$?; tail -5 $out/r4_tests.log; echo "tests rc=$rc"
  [ $rc -ne 0 ] && exit $rc
fi
if [ "$what" = all ] || [ "$what" = probe ]; then
  SETS=${SETS:-K,E,O,N,C,Q,R}
  echo "== k_scanb (default)"; timeout -k 10 600 python tools/kprobe.py --sets $SETS 2>&1 | grep -v "^$" | tee $out/r4_kprobe_b.log
  echo "== k_scanm (NXS_GPU_NOSCANB=1)"; NXS_GPU_NOSCANB=1 timeout -k 10 600 python tools/kprobe.py --sets $SETS 2>&1 | tee $out/r4_kprobe_m.log
fi
if [ "$what" = all ] || [ "$what" = bench ]; then
  . tools/ab2.sh
  run scanb - ; run scanm - NXS_GPU_NOSCANB=1 ; run scanb - ; run scanm - NXS_GPU_NOSCANB=1
fi

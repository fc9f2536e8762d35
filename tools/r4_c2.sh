#!/bin/bash
set -u
out=$PWD/gpurun_out; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "every_scan_path or random_corpora or incremental_refresh or pipelined or golden or resync" > $out/r4_c2_tests.log 2>&1
rc=$?; tail -3 $out/r4_c2_tests.log; [ $rc -ne 0 ] && exit $rc
for e in "" "NXS_PLAN_CACHE=0"; do
  echo "== C2 $e"; env $e python bench.py --workload C2 --cpu-seconds 0 --no-extras --steps 40 --warmup 8 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['host_ms_per_step'])"
done
python bench.py --cpu-seconds 0 --steps 20 --warmup 5 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['host_ms_per_step'], d['refresh'], d['latency'])"

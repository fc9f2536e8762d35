#!/bin/bash
# worker threads that poll for the next run before they sleep (NXS_POOL_SPIN_US), A/B in one session
out=gpurun_out/r4_spin; mkdir -p $out
for i in 1 2 3; do
  for sp in 0 120 500; do
    for w in C3 C2; do
      NXS_POOL_SPIN_US=$sp python3 bench.py --workload $w --cpu-seconds 0 --no-extras > $out/${w}_$sp.json 2>> $out/err.log
      echo "$w spin $sp $(python3 tools/show_bench.py $out/${w}_$sp.json | head -1) $(python3 -c "import json;d=json.loads(open('$out/${w}_$sp.json').read().strip().splitlines()[-1]);print(d['host_ms_per_step']['plan_ms'])")"
    done
  done
done

#!/bin/bash
# batches in flight 2 / 3 on the host-bound workloads (C2, C5), one session
out=gpurun_out/r4_depth2; mkdir -p $out
for w in C2 C5; do
  for d in 2 3 2 3; do
    st=""; [ $w = C5 ] && st="--steps 8 --warmup 2"
    python3 bench.py --workload $w --depth $d $st --cpu-seconds 0 --no-extras > $out/${w}_d$d.json 2>> $out/err.log
    echo "$w depth $d $(python3 tools/show_bench.py $out/${w}_d$d.json 2>/dev/null | head -1)"
  done
done

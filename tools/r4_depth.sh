#!/bin/bash
# batches in flight: default limit (1000) at depths 2 / 3 / 4 and the headline at 2 / 3, one session
out=gpurun_out/r4_depth; mkdir -p $out
for d in ${DEPTHS:-2 3 4}; do
  python3 bench.py --limit 1000 --depth $d --steps 24 --warmup 4 --cpu-seconds 0 --no-extras > $out/l1000_d$d.json 2>> $out/err.log
  python3 tools/show_bench.py $out/l1000_d$d.json 2>/dev/null | head -1
done
for d in ${C3DEPTHS:-2 3}; do
  python3 bench.py --depth $d --cpu-seconds 0 --no-extras > $out/c3_d$d.json 2>> $out/err.log
  python3 tools/show_bench.py $out/c3_d$d.json 2>/dev/null | head -1
done

#!/bin/bash
# A/B of a bench workload's host pipeline inside one session: "name:ENV=val,ENV=val" per argument
# usage: tools/ab_c4.sh [--workload C4|C5] [--rounds N] spec...
mkdir -p gpurun_out/r5/ab
W=C4; ROUNDS=2
while [ "${1#--}" != "$1" ]; do
  case "$1" in --workload) W=$2; shift 2;; --rounds) ROUNDS=$2; shift 2;; *) break;; esac
done
STEPS=$([ $W = C5 ] && echo 8 || echo 100)
for r in $(seq 1 $ROUNDS); do
for spec in "$@"; do
  name=${spec%%:*}; kv=${spec#*:}
  (
    IFS=','; for e in $kv; do case "$e" in ?*=*) export "$e";; esac; done
    timeout -k 10 400 python bench.py --workload $W --cpu-seconds 0 --no-extras --steps $STEPS > gpurun_out/r5/ab/$W.$name.$r.json 2>gpurun_out/r5/ab/$W.$name.$r.err
  )
  python - "$name" gpurun_out/r5/ab/$W.$name.$r.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print("%-12s %9.0f q/s %.4f ms/step  host %s" % (sys.argv[1], d["value"], d["ms_per_step"], d.get("host_ms_per_step")))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
done
done

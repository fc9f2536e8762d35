#!/bin/bash
# A/B of the default-limit batch (params == NULL => 1000) inside one session: "name:ENV=val,ENV=val" per argument
mkdir -p gpurun_out/r5/ab
for spec in "$@"; do
  name=${spec%%:*}; kv=${spec#*:}
  (
    IFS=','; for e in $kv; do case "$e" in ?*=*) export "$e";; esac; done
    timeout -k 10 300 python bench.py --limit 1000 --depth 4 --cpu-seconds 0 --no-extras --steps 30 > gpurun_out/r5/ab/L.$name.json 2>gpurun_out/r5/ab/L.$name.err
  )
  python - "$name" gpurun_out/r5/ab/L.$name.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print("%-12s %9.0f q/s %.4f ms/step  %s" % (sys.argv[1], d["value"], d["ms_per_step"], [(k["kernel"][:24], k["ms"]) for k in d["roofline"]["per_kernel"]][:4]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
done

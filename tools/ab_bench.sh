#!/bin/bash
# A/B of the C3 bench inside one session: "name:ENV=val,ENV=val" or "name:lib=<variant suffix>" per argument
# usage: tools/ab_bench.sh [--rounds N] spec...
mkdir -p gpurun_out/r5/ab
ROUNDS=2
if [ "$1" = "--rounds" ]; then ROUNDS=$2; shift 2; fi
C=$PWD/nxsearch_amd/csrc
for r in $(seq 1 $ROUNDS); do
for spec in "$@"; do
  name=${spec%%:*}; kv=${spec#*:}
  (
    IFS=','; for e in $kv; do
      case "$e" in
        lib=*) export NXS_GPU_LIB=$C/libnxsearch_gpu_${e#lib=}.so;;
        ?*=*) export "$e";;
      esac
    done
    NXS_BENCH_SERIAL=1 timeout -k 10 200 python bench.py --cpu-seconds 0 --no-extras --steps 100 > gpurun_out/r5/ab/$name.$r.json 2>/dev/null
  )
  python - "$name" gpurun_out/r5/ab/$name.$r.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    r = d["roofline"]
    al = {k["kernel"].split("<")[0] + ("D" if "true>" in k["kernel"] and "scanm" in k["kernel"] else ""): k["ms"] for k in r.get("per_kernel_serial", [])}
    be = {k["kernel"].split("<")[0] + ("D" if "true>" in k["kernel"] and "scanm" in k["kernel"] else ""): k["ms"] for k in r.get("per_kernel", [])}
    print("%-14s %8.0f q/s %.4f ms/step  alone %s  beside %s  requeries %s" % (sys.argv[1], d["value"], d["ms_per_step"],
          {k: round(v, 3) for k, v in al.items()}, {k: round(v, 3) for k, v in be.items()}, d.get("host_ms_per_step", {}).get("exact_requeries")))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
done
done

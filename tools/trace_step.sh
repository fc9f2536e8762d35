#!/bin/bash
# kernel timeline of one bench step under rocprofv3 (csv): usage: tools/trace_step.sh <name> [bench args...]; env passes through
name=$1; shift
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
d=gpurun_out/r5/trace_$name
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $d -o run -- python3 bench.py --steps 7 --warmup 1 --cpu-seconds 0 --no-extras --min-seconds 0 "$@" > gpurun_out/r5/trace_$name.json 2> gpurun_out/r5/trace_$name.log
f=$(find $d -name "*kernel_trace.csv" | head -1)
test -n "$f" || { echo "no trace"; tail -3 gpurun_out/r5/trace_$name.log; exit 1; }
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows]
ks.sort()
# the last full step: find the last k_cursors and print from the one before it
idx = [i for i, k in enumerate(ks) if k[2].startswith("k_cursors")]
if len(idx) < 3:
    print("few steps", len(idx)); sys.exit(0)
a, b = idx[-3], idx[-2]
t0 = ks[a][0]
for s, e, n, q in ks[a:b + 12]:
    print("%9.1f %9.1f  %7.1f us  q%s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n[:60]))
PY
rm -rf $d

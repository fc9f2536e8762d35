#!/bin/bash
# quick look at the C3 step on the GPU box: per-kernel durations + issued instructions of the scan kernels
set -u
out=$PWD/gpurun_out/probe
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o run -- python3 bench.py --steps 7 --warmup 1 --cpu-seconds 0 --no-extras "$@" > "$out/bench.json" 2> "$out/stats.log" &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d "$out/pmc" -o run -- python3 bench.py --steps 7 --warmup 1 --cpu-seconds 0 --no-extras "$@" > /dev/null 2> "$out/pmc.log" &&
python3 - "$out" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
d = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
print("bench under rocprof:", d["value"], d["ms_per_step"], d["roofline"]["step"]["span_ms"], [(k["kernel"], k["ms"]) for k in d["roofline"]["per_kernel"]])
f = glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True)[0]
for i, r in enumerate(csv.DictReader(open(f))):
    if i < 12: print("%-44s calls %4s avg %10.1f us" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e3))
f = glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_INSTS_VALU": n[k] += 1
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].values()))[:8]:
    print("%-40s x%3d per call: VALU %8.1fM SALU %8.1fM LDS %7.1fM VMEM %7.2fM" % (k[:40], n[k], *(v[c] / n[k] / 1e6 for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD"))))
PY

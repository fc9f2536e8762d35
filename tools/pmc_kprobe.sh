#!/bin/bash
# instruction / wait counters of the mask-path kernel on a kprobe query set (GPU box)
# usage: tools/pmc_kprobe.sh <set> [ENV=val ...]
set -u
SET=${1:-X}; shift
out=$PWD/gpurun_out/pmckp
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
for kv in "$@"; do export $kv; done
B="python3 tools/kprobe.py --sets $SET --reps 3"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d "$out/a" -o run -- $B > /dev/null 2> "$out/a.log" &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$out/b" -o run -- $B > /dev/null 2> "$out/b.log" &&
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d "$out/c" -o run -- $B > /dev/null 2> "$out/c.log" &&
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/s" -o run -- $B > /dev/null 2> "$out/s.log"
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in acc:
    if "k_scan" not in k and "k_cold" not in k: continue
    print(k[:44], " ".join("%s=%.1fM" % (c.replace("SQ_", ""), v / max(n[k][c], 1) / 1e6) for c, v in sorted(acc[k].items())))
for f in glob.glob(out + "/s/**/*kernel_stats.csv", recursive=True):
    for i, r in enumerate(csv.DictReader(open(f))):
        if i < 6: print("%-44s calls %4s avg %10.1f us" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e3))
PY

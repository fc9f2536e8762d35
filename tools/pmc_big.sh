#!/bin/bash
# instruction mix and wait / busy counters of the MODE_BIG kernels on the default-limit bench (GPU box)
set -u
out=$PWD/gpurun_out/pmcbig
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
for kv in "$@"; do export $kv; done
B="python3 bench.py --limit 1000 --depth 1 --steps 3 --warmup 1 --cpu-seconds 0 --no-extras"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$out/a" -o run -- $B > /dev/null 2> "$out/a.log" &&
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d "$out/b" -o run -- $B > /dev/null 2> "$out/b.log" &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d "$out/c" -o run -- $B > /dev/null 2> "$out/c.log" &&
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY --output-format csv -d "$out/d" -o run -- $B > /dev/null 2> "$out/d.log"
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in acc:
    if "k_scan" not in k and "k_replay" not in k: continue
    print(k[:44], " ".join("%s=%.1fM" % (c.replace("SQ_", ""), v / max(n[k][c], 1) / 1e6) for c, v in sorted(acc[k].items())))
PY

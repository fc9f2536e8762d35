#!/bin/bash
# instruction counters of the scan kernels for several library variants (probe builds give wrong results)
export TMPDIR=/tmp
for v in "$@"; do
  out=$PWD/gpurun_out/ic_$v; rm -rf $out; mkdir -p $out
  if [ "$v" != default ]; then export NXS_GPU_LIB=$PWD/nxsearch_amd/csrc/libnxsearch_gpu_$v.so; else unset NXS_GPU_LIB; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d "$out/pmc" -o run -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-extras > /dev/null 2> "$out/pmc.log"
  python3 - "$out" "$v" <<'PY'
import csv, glob, sys, collections
out, v = sys.argv[1:3]
f = glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_INSTS_VALU": n[k] += 1
for k, vv in acc.items():
    if "k_scanb" in k or "k_scanm<5, false, false" in k:
        print("%-10s %-34s x%3d per call: VALU %8.1fM SALU %8.1fM LDS %7.1fM VMEM %7.2fM" % (v, k[:34], n[k], *(vv[c] / n[k] / 1e6 for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD"))))
PY
done

#!/usr/bin/env python3
"""Condenses rocprofv3 output directories into the summaries kept under profiles/.

    tools/pmc_summary.py --stats DIR --pmc DIR [DIR ...] --steps N --out-prefix profiles/r1_final

* DIR of a `--kernel-trace --stats` run  -> <prefix>_kernel_stats.csv (copied as is)
* DIRs of `--pmc` passes (one counter group per pass, as MI355X_MICROARCH.md
  prescribes)                            -> <prefix>_pmc_summary.json

One bench "step" launches the scan kernel once per query class (k_scan8<0,5,1>
for pure-OR, k_scan8<0,5,0> for the rest ...), so per-step figures are the sum
over every `k_scan*` instantiation divided by the number of steps that ran
(warmup included: --steps here = timed + warmup steps of the profiled command).
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports coalesced
streaming reads by 2x (guide, HBM section), so read bytes = FETCH_SIZE*1024*2.
"""
import argparse
import collections
import csv
import glob
import json
import os
import re
import shutil
import sqlite3


def source_hash():
    """sha256 over the HIP sources + headers the profiled library is built from (bench.py recomputes it
    and refuses a summary whose kernels are not the tree's)."""
    import hashlib
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nxsearch_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(root)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]


def short(name):
    m = re.match(r"(?:void )?([A-Za-z_0-9]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats", default=None)
    ap.add_argument("--pmc", nargs="*", default=[])
    ap.add_argument("--steps", type=int, required=True, help="steps run by the profiled command (timed + warmup)")
    ap.add_argument("--out-prefix", required=True)
    ap.add_argument("--command", default="")
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--terms", type=int, default=1_000_000)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--limit", type=int, default=10)
    ap.add_argument("--name", default="C3", help="workload name (bench.py --workload)")
    ap.add_argument("--calib", default=None, help="DIR of the --pmc FETCH_SIZE pass over tools/pmc_calib.py")
    ap.add_argument("--calib-json", default=None, help="stdout of tools/pmc_calib.py (known byte counts)")
    a = ap.parse_args()

    if a.stats:
        files = glob.glob(os.path.join(a.stats, "**", "*_kernel_stats.csv"), recursive=True)
        dbs = glob.glob(os.path.join(a.stats, "**", "*.db"), recursive=True)
        if len(files) == 1:
            shutil.copy(files[0], a.out_prefix + "_kernel_stats.csv")
        elif len(dbs) == 1:          # rocpd (sqlite) output, the default of this rocprofv3
            con = sqlite3.connect(dbs[0])
            rows = con.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
            with open(a.out_prefix + "_kernel_stats.csv", "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(["Name", "Calls", "TotalDuration_us", "AverageDuration_us", "Percentage"])
                for name, calls, tot, avg, pct in rows:
                    w.writerow([short(name), calls, round(tot, 3), round(avg, 3), round(pct, 3)])
        else:
            raise SystemExit("no single kernel stats csv / rocpd db under %s" % a.stats)

    # FETCH_SIZE against known byte counts: bytes really read per counted KiB, for
    # 16 B/lane loads (the guide's case: 2.0) and for 8 B/lane global_load_dwordx2
    # (what the scan kernels issue)
    calib = None
    if a.calib and a.calib_json:
        known = None
        for line in open(a.calib_json):
            line = line.strip()
            if line.startswith("{"):
                known = json.loads(line)["bytes_per_kernel"]
        files = glob.glob(os.path.join(a.calib, "**", "*_counter_collection.csv"), recursive=True)
        dbs = glob.glob(os.path.join(a.calib, "**", "*.db"), recursive=True)
        if files:
            with open(files[0], newline="") as f:
                rows = [(r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"])) for r in csv.DictReader(f)]
        else:
            con = sqlite3.connect(dbs[0])
            rows = con.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
        per = collections.defaultdict(float)
        for kname, cname, val in rows:
            if cname == "FETCH_SIZE" and short(kname).startswith("k_hbm_read"):
                per[short(kname)] += float(val)
        if known and per:
            calib = {"bytes_per_kernel": known,
                     "FETCH_SIZE_KB": dict(per),
                     "bytes_per_counted_byte": {k: known / (v * 1024.0) for k, v in per.items() if v > 0}}

    # counter -> kernel -> [sum, dispatches]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for d in a.pmc:
        files = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
        dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
        if len(files) == 1:
            with open(files[0], newline="") as f:
                rows = [(r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"])) for r in csv.DictReader(f)]
        elif len(dbs) == 1:
            con = sqlite3.connect(dbs[0])
            rows = con.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
        else:
            raise SystemExit("no single counter_collection csv / rocpd db under %s" % d)
        for kname, cname, val in rows:
            c = acc[cname][short(kname)]
            c[0] += float(val)
            c[1] += 1
    if not acc:
        return

    counters = collections.defaultdict(dict)
    for cname, per_k in acc.items():
        for k, (s, n) in per_k.items():
            counters[k][cname] = {"sum_per_step": s / a.steps, "dispatches_per_step": n / a.steps}

    def scan_total(cname):
        return sum(v[cname]["sum_per_step"] for k, v in counters.items()
                   if (k.startswith("k_scan") or k.startswith("k_cold")) and cname in v)

    fetch_kb, write_kb = scan_total("FETCH_SIZE"), scan_total("WRITE_SIZE")
    factor, factor_src = 2.0, "MI355X_MICROARCH.md (16 B/lane streaming reads), unverified for 8 B/lane"
    if calib and "k_hbm_read_x2" in calib["bytes_per_counted_byte"]:
        factor = calib["bytes_per_counted_byte"]["k_hbm_read_x2"]
        factor_src = "measured: k_hbm_read_x2 (global_load_dwordx2, the scan kernels' width) over a known byte count"
    out = {
        "command": a.command,
        "workload": {"name": a.name, "docs": a.docs, "terms": a.terms, "batch": a.batch, "limit": a.limit},
        "source_hash": source_hash(),
        "fetch_size_calibration": calib,
        "fetch_size_factor": factor,
        "fetch_size_factor_source": factor_src,
        "kernel": "k_scan* + k_cold (all query-class launches of one step summed)",
        "steps_profiled": a.steps,
        "FETCH_SIZE_KB_per_step": fetch_kb,
        "WRITE_SIZE_KB_per_step": write_kb,
        "correction": "gfx950 FETCH_SIZE reports half the bytes of coalesced streaming reads "
                      "(MI355X_MICROARCH.md, HBM): hbm_read_bytes = FETCH_SIZE*1024*2; WRITE_SIZE*1024 as is",
        "hbm_bytes_per_launch": fetch_kb * 1024 * factor + write_kb * 1024,
        "counters": counters,
    }
    with open(a.out_prefix + "_pmc_summary.json", "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()

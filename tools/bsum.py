#!/usr/bin/env python3
"""Print the headline figures of bench.py JSON lines (argv: files)."""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable:", e); continue
    r = d.get("roofline", {})
    print("%s: %.0f q/s  %.4f ms/step  dominant %s %.3f ms frac %.3f" % (f, d["value"], d["ms_per_step"], r.get("kernel"), r.get("kernel_ms", 0), r.get("frac", 0)))
    for k in r.get("per_kernel", []):
        print("    beside: %-52s q %4d  %.4f ms" % (k["kernel"], k["queries"], k["ms"]))
    for k in r.get("per_kernel_serial", []):
        print("    alone : %-52s        %.4f ms" % (k["kernel"], k["ms"]))
    x = {k: round(d[k]) for k in d if k.endswith("_qps")}
    dl = d.get("default_limit", {})
    print("   ", x, "default_limit", dl.get("queries_per_s"), "lat p50", d.get("latency", {}).get("p50_ms"), d.get("latency_default", {}).get("p50_ms"))
    cb = d.get("cpu_baseline", {})
    print("    parity_mismatches", cb.get("parity_mismatches"), "of", cb.get("parity_checked", cb.get("queries_done")))

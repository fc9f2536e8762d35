#!/usr/bin/env python3
"""Pretty-print the roofline object and the scalar fields of a bench.py JSON line."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
print(json.dumps(d["roofline"], indent=1))
print({k: d[k] for k in d if k not in ("roofline", "config", "cpu_baseline")})

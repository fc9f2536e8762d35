#!/usr/bin/env python3
"""Timeline of the last dispatches of a rocprofv3 --kernel-trace run (rocpd sqlite):
usage: trace_tail.py <dir> [n]"""
import glob, os, sqlite3, sys
db = glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
con = sqlite3.connect(db)
names = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
kv = [x for x in names if x == "kernels"] or [x for x in names if "kernel" in x.lower()]
print("views:", [x for x in names if not x.startswith("rocpd_")][:40])
cols = [r[1] for r in con.execute("pragma table_info(%s)" % kv[0])]
print(kv[0], cols)
sel = [c for c in ("name", "start", "end", "duration", "stream_id", "queue_id") if c in cols]
rows = con.execute("select %s from %s order by start desc limit %d" % (",".join(sel), kv[0], n)).fetchall()[::-1]
t0 = rows[0][sel.index("start")]
for r in rows:
    d = dict(zip(sel, r))
    print("%-40s start %9.1f us  dur %7.1f us  %s" % (str(d["name"])[:40], (d["start"] - t0) / 1e3, (d["end"] - d["start"]) / 1e3,
          " ".join("%s=%s" % (k, d[k]) for k in sel if k in ("stream_id", "queue_id"))))

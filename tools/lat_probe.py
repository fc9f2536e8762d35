import os, sys, time, json
sys.path.insert(0, "/root/repo")
import nxsearch_amd as N
from nxsearch_amd import corpus
work = "/dev/shm/nxs_lat"
os.makedirs(work, exist_ok=True)
info = corpus.write_corpus(work, 10_000_000, 1_000_000, seed=0)
terms = corpus.term_strings(1_000_000, 0)
nxs = N.Nxs(work)
idx = nxs.open_files(info["terms"], info["dtmap"])
qs = corpus.queries_single(terms, 300, seed=3)
for q in qs[:50]:
    idx.search(q, limit=10, fuzzymatch=False)
ts = []
for q in qs:
    t0 = time.perf_counter()
    idx.search(q, limit=10, fuzzymatch=False)
    ts.append(time.perf_counter() - t0)
ts.sort()
print("python-level p50 %.1f us p95 %.1f us" % (ts[len(ts)//2]*1e6, ts[int(len(ts)*0.95)]*1e6))
if os.environ.get("NXS_GPU_LIB", "").endswith("_stats.so"):
    import ctypes as C
    L = N.lib()
    out = (C.c_ulonglong * 8)()
    L.nxsgpu_debug_rstats(out, 1)
    for q in qs:
        idx.search(q, limit=10, fuzzymatch=False)
    L.nxsgpu_debug_rstats(out, 1)
    n = max(out[0], 1)
    print("k_replay per query: to-first-counts %.1f us, candidates+heap %.1f us, sort %.1f us, output %.1f us; candidates %.1f, inserts %.1f" % (
        out[1] / n / 100.0, out[2] / n / 100.0, out[3] / n / 100.0, out[4] / n / 100.0, out[5] / n, out[6] / n))

#!/usr/bin/env python3
"""The exact two-pass path (limit > 64; the API default is 1000): C3 queries, first 64 of the
batch, nxs_index_search_batch().  Run under rocprofv3 --kernel-trace --stats for the split."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nxsearch_amd as N
from nxsearch_amd import corpus
work = "/dev/shm/nxs_lat"
os.makedirs(work, exist_ok=True)
info = corpus.write_corpus(work, 10_000_000, 1_000_000, seed=0)
terms = corpus.term_strings(1_000_000, 0)
nxs = N.Nxs(work)
idx = nxs.open_files(info["terms"], info["dtmap"])
qs = corpus.queries_bool5(terms, 1024, seed=3)[:64]
limit = int(os.environ.get("LIMIT", "1000"))
idx.search_batch(qs, limit=limit, fuzzymatch=False)
for rep in range(3):
    t0 = time.perf_counter()
    r = idx.search_batch(qs, limit=limit, fuzzymatch=False)
    dt = time.perf_counter() - t0
    print("limit %d: 64 queries in %.2f ms (%.0f q/s), %.1f results/query" % (limit, dt * 1e3, 64 / dt, sum(len(x) for x in r) / 64.0), flush=True)
if os.environ.get("NXS_GPU_LIB", "").endswith("_stats.so"):
    import ctypes as C
    L = N.lib()
    out = (C.c_ulonglong * 8)()
    L.nxsgpu_debug_rstats(out, 1)
    idx.search_batch(qs, limit=limit, fuzzymatch=False)
    L.nxsgpu_debug_rstats(out, 1)
    n = max(out[0], 1)
    print("k_replay per query (%d): to-first-counts %.1f us, candidates+heap %.1f us, sort %.1f us, output %.1f us; candidates %.1f, inserts %.1f" % (
        n, out[1] / n / 100.0, out[2] / n / 100.0, out[3] / n / 100.0, out[4] / n / 100.0, out[5] / n, out[6] / n))

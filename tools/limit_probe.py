#!/usr/bin/env python3
"""Limits above 64 (the API default is 1000): C3 queries through nxs_index_search_batch()
and the pipelined _begin/_end pair; host profile (exact re-queries!) after every run.
LIMIT=1000 NQ=1024 python tools/limit_probe.py; run under rocprofv3 --kernel-trace --stats
for the kernel split."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nxsearch_amd as N
from nxsearch_amd import corpus
work = "/dev/shm/nxs_lat"
os.makedirs(work, exist_ok=True)
ndocs = int(os.environ.get("NDOCS", "10000000"))
info = corpus.write_corpus(work, ndocs, ndocs // 10, seed=0)
terms = corpus.term_strings(ndocs // 10, 0)
nxs = N.Nxs(work)
idx = nxs.open_files(info["terms"], info["dtmap"])
limit = int(os.environ.get("LIMIT", "1000"))
allq = corpus.queries_bool5(terms, 1024, seed=3)
for nq in [int(x) for x in os.environ.get("NQ", "64,1024").split(",")]:
    qs = allq[:nq]
    idx.search_batch(qs, limit=limit, fuzzymatch=False)
    idx.host_profile()
    for rep in range(3):
        t0 = time.perf_counter()
        r = idx.search_batch(qs, limit=limit, fuzzymatch=False)
        dt = time.perf_counter() - t0
        print("limit %d blocking: %d queries in %.2f ms (%.0f q/s), %.1f results/query  %s" % (
            limit, nq, dt * 1e3, nq / dt, sum(len(x) for x in r) / float(nq), idx.host_profile()), flush=True)
    # pipelined: two batches in flight
    steps = 8
    t0 = time.perf_counter()
    idx.search_batch_begin(qs, limit=limit, fuzzymatch=False)
    for s in range(steps):
        if s + 1 < steps:
            idx.search_batch_begin(qs, limit=limit, fuzzymatch=False)
        idx.search_batch_end()
    dt = (time.perf_counter() - t0) / steps
    print("limit %d pipelined: %d queries in %.2f ms per step (%.0f q/s)  %s" % (
        limit, nq, dt * 1e3, nq / dt, idx.host_profile()), flush=True)
if os.environ.get("NXS_GPU_LIB", "").endswith("_stats.so"):
    import ctypes as C
    L = N.lib()
    out = (C.c_ulonglong * 8)()
    qs = allq[:int(os.environ.get("NQ", "64,1024").split(",")[-1])]
    L.nxsgpu_debug_rstats(out, 1)
    idx.search_batch(qs, limit=limit, fuzzymatch=False)
    L.nxsgpu_debug_rstats(out, 1)
    n = max(out[0], 1)
    if limit > 64:
        print("k_replay<LDS> over %d queries: mean to-first-counts %.1f us, candidates+heap %.1f us, sort+output %.1f us; "
              "slowest query %.1f us; mean candidates %.1f, inserts %.1f; most inserts %d (that query: ~%d candidates)" % (
                  n, out[1] / n / 100.0, out[2] / n / 100.0, out[3] / n / 100.0, out[4] / 100.0, out[5] / n, out[6] / n,
                  out[7] // 1000000, (out[7] % 1000000) * 16))
    else:
        print("k_replay per query (%d): to-first-counts %.1f us, candidates+heap %.1f us, sort %.1f us, output %.1f us; candidates %.1f, inserts %.1f" % (
            n, out[1] / n / 100.0, out[2] / n / 100.0, out[3] / n / 100.0, out[4] / n / 100.0, out[5] / n, out[6] / n))

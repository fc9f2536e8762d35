#!/usr/bin/env python3
"""C2 / C4-shaped traffic (1024 single-term queries per batch): where k_replay's time goes
(stats build: NXS_GPU_LIB=.../libnxsearch_gpu_stats.so) and the batch rate through the C loop."""
import os, sys, time, json, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nxsearch_amd as N
from nxsearch_amd import corpus
docs = int(os.environ.get("NDOCS", "1000000"))
work = "/dev/shm/nxs_c2_%d" % docs
os.makedirs(work, exist_ok=True)
info = corpus.write_corpus(work, docs, docs // 10, seed=0)
terms = corpus.term_strings(docs // 10, 0)
nxs = N.Nxs(work)
idx = nxs.open_files(info["terms"], info["dtmap"])
qs = corpus.queries_single(terms, 1024, seed=3, lo=int(os.environ.get("LO", "10")), hi=int(os.environ.get("HI", "10000")))
for _ in range(3):
    idx.search_batch(qs, limit=10, fuzzymatch=False)
idx.set_profiling(True); idx.profile(reset=True)
for _ in range(5):
    idx.search_batch(qs, limit=10, fuzzymatch=False)
p = idx.profile(reset=True); idx.set_profiling(False)
print("per batch: scan %.3f ms, after-last-scan %.3f ms (launches %d)" % (p["scan_ms"] / p["launches"], p["replay_ms"] / p["launches"], p["launches"]))
if os.environ.get("NXS_GPU_LIB", "").endswith("_stats.so"):
    L = N.lib()
    out = (C.c_ulonglong * 8)()
    L.nxsgpu_debug_rstats(out, 1)
    idx.search_batch(qs, limit=10, fuzzymatch=False)
    L.nxsgpu_debug_rstats(out, 1)
    n = max(out[0], 1)
    print("k_replay per query (%d): to-first-counts %.1f us, candidates+heap %.1f us, sort %.1f us, output %.1f us; candidates %.1f, inserts %.1f" % (
        n, out[1] / n / 100.0, out[2] / n / 100.0, out[3] / n / 100.0, out[4] / n / 100.0, out[5] / n, out[6] / n))

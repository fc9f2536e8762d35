#!/usr/bin/env python3
"""Per-step wall times of the pipelined C3 loop (DEPTH batches in flight, four rotated batches): is the mean a
steady state or a few stalls?  DEPTH=3 STEPS=300 LIMIT=10 python tools/step_hist.py"""
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
limit = int(os.environ.get("LIMIT", "10"))
depth = int(os.environ.get("DEPTH", "3"))
steps = int(os.environ.get("STEPS", "300"))
batches = [corpus.queries_bool5(terms, 1024, seed=3 + 100 * s, hi=1000) for s in range(4)]   # bench.py's four
L = N.lib()
import ctypes as C
qarrs = [(C.c_char_p * 1024)(*[q.encode() for q in b]) for b in batches]
p = N._make_params(limit, "BM25", False)
resps = (C.c_void_p * 1024)()
errs = (C.c_int * 1024)()
def begin(s):
    if L.nxs_index_search_batch_begin(idx._h, p, qarrs[s % 4], 1024) != 0:
        raise RuntimeError(nxs.error())
def end():
    if L.nxs_index_search_batch_end(idx._h, resps, errs) < 0:
        raise RuntimeError(nxs.error())
    for i in range(1024):
        if resps[i]:
            L.nxs_resp_release(C.c_void_p(resps[i]))
for rep in range(2):
    tb, te, tt = [], [], []
    for s in range(steps):
        t0 = time.perf_counter()
        begin(s)
        t1 = time.perf_counter()
        if s + 1 >= depth:
            end()
        t2 = time.perf_counter()
        tb.append(t1 - t0); te.append(t2 - t1); tt.append(t2 - t0)
    for _ in range(depth - 1):
        end()
    def q(v, f):
        v = sorted(v)
        return 1e3 * v[min(len(v) - 1, int(f * len(v)))]
    v = tt[depth + 5:]
    print("rep %d limit %d depth %d: step p50 %.3f p90 %.3f p99 %.3f max %.3f mean %.3f ms | begin p50 %.3f p99 %.3f max %.3f | end p50 %.3f p99 %.3f max %.3f" % (
        rep, limit, depth, q(v, .5), q(v, .9), q(v, .99), 1e3 * max(v), 1e3 * sum(v) / len(v),
        q(tb[8:], .5), q(tb[8:], .99), 1e3 * max(tb[8:]), q(te[8:], .5), q(te[8:], .99), 1e3 * max(te[8:])), flush=True)
    big = [(i, round(1e3 * x, 2)) for i, x in enumerate(tt) if x > 2.5 * sorted(tt)[len(tt) // 2] and i > depth + 5]
    print("  steps > 2.5 x median:", big[:20], flush=True)
    print("  mean step by (s mod 4):", [round(1e3 * sum(tt[i] for i in range(12 + r, steps, 4)) / len(range(12 + r, steps, 4)), 3) for r in range(4)], flush=True)
# each batch alone (the same batch in every step)
idx.set_profiling(True)
for b in range(4):
    idx.profile(reset=True)
    t0 = time.perf_counter()
    for s in range(40):
        if L.nxs_index_search_batch_begin(idx._h, p, qarrs[b], 1024) != 0:
            raise RuntimeError(nxs.error())
        if s + 1 >= depth:
            end()
    for _ in range(depth - 1):
        end()
    dt = (time.perf_counter() - t0) / 40
    pr = idx.profile(reset=True)
    print("batch %d alone: %.3f ms/step; classes (key hex: ms per launch, queries): %s" % (
        b, 1e3 * dt, " ".join("%x:%.3f/%d" % (c["key"], c["ms"] / max(c["launches"], 1), c["queries"] // max(c["launches"], 1)) for c in pr["classes"])), flush=True)

#!/usr/bin/env python3
"""Fuzzy-path probe: C4-style tokens over the 1M-term BK-tree (small doc set: the
tree depends on the terms only).  Run under rocprofv3 --kernel-trace --stats to see
the per-level kernels."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nxsearch_amd as N
from nxsearch_amd import corpus
work = "/dev/shm/nxs_fz_probe"
os.makedirs(work, exist_ok=True)
info = corpus.write_corpus(work, 200_000, 1_000_000, seed=0)
terms = corpus.term_strings(1_000_000, 0)
nxs = N.Nxs(work)
idx = nxs.open_files(info["terms"], info["dtmap"])
toks = corpus.queries_fuzzy(terms, 1024, seed=4)
idx.fuzzy(toks)
for rep in range(3):
    t0 = time.time()
    ids, vis = idx.fuzzy(toks, want_visited=True)
    dt = time.time() - t0
    print("fuzzy: %d tokens, %d candidates, %.2f ms wall, %.2f G cand/s" % (len(toks), sum(vis), dt * 1e3, sum(vis) / dt / 1e9), flush=True)

#!/usr/bin/env python3
"""Fuzzy-path probe: C4 tokens over the 1M-term BK-tree of the C3 corpus.  Run under
rocprofv3 --kernel-trace --stats to see the per-kernel times (match-first search:
k_fz_filter / k_fz_dist / k_fz_chain; NXS_GPU_FUZZY_BFS=1: k_bk_level per level)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nxsearch_amd as N
from nxsearch_amd import corpus
docs = int(os.environ.get("FZ_DOCS", "10000000"))
n_terms = int(os.environ.get("FZ_TERMS", "1000000"))
n_tok = int(os.environ.get("FZ_TOKENS", "1024"))
work = "/dev/shm/nxs_fz_probe_%d_%d" % (docs, n_terms)
os.makedirs(work, exist_ok=True)
info = corpus.write_corpus(work, docs, n_terms, seed=0)
terms = corpus.term_strings(n_terms, 0)
nxs = N.Nxs(work)
idx = nxs.open_files(info["terms"], info["dtmap"])
toks = corpus.queries_fuzzy(terms, n_tok, seed=4)
idx.fuzzy(toks)
idx.set_profiling(True)
for rep in range(3):
    idx.profile(reset=True)
    t0 = time.time()
    ids = idx.fuzzy(toks)
    dt = time.time() - t0
    p = idx.profile(reset=True)
    print("fuzzy: %d tokens, %.2f ms wall, device %.3f ms, evals %d, levels %s" % (
        len(toks), dt * 1e3, p["fuzzy_ms"], p["fuzzy_visits"], [x for x in p["fuzzy_level"] if x]), flush=True)
if not os.environ.get("FZ_SINGLES"):
    sys.exit(0)
t0 = time.time()
for t in toks[:64]:
    idx.fuzzy([t])
print("single-token calls: %.3f ms each" % ((time.time() - t0) / 64 * 1e3))

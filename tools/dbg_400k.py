#!/usr/bin/env python3
"""Debug: the query set of test_sparse_terms_on_a_larger_corpus (400k docs), GPU vs oracle, current env."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nxsearch_amd as N
import oracle_lib as O
from nxsearch_amd import corpus
work = "/dev/shm/nxs_dbg2"
os.makedirs(work, exist_ok=True)
c = corpus.write_corpus(work, 400_000, 20_000, seed=31)
terms = corpus.term_strings(20_000, seed=31)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
qs = []
if which in ("all", "a"): qs += corpus.queries_bool5(terms, 24, seed=7, lo=200, hi=4000)
if which in ("all", "b"): qs += corpus.queries_bool5(terms, 12, seed=8, lo=2000, hi=20_000, k=3)
if which in ("all", "c"): qs += corpus.queries_bool5(terms, 12, seed=9, lo=30, hi=3000, k=7)
if which in ("all", "d"): qs += [" OR ".join(terms[r - 1].decode() for r in (3, 300, 500, 700, 900)), " OR ".join(terms[r - 1].decode() for r in (10, 400, 600))]
print("queries", len(qs), flush=True)
oidx = O.Index(c["terms"], c["dtmap"])
with N.Nxs(work) as nxs:
    gidx = nxs.open_files(c["terms"], c["dtmap"])
    bad = 0
    got = gidx.search_batch(qs, limit=10, fuzzymatch=False)
    for q, g in zip(qs, got):
        w = oidx.search(q, limit=10, fuzzymatch=False)
        if [d for d, _ in g] != [d for d, _ in w]:
            bad += 1
            if bad <= 3:
                print("MISMATCH", q, g[:3], w[:3])
    print("env", {k: v for k, v in os.environ.items() if k.startswith("NXS_")}, which, "bad", bad, "of", len(qs), flush=True)
    gidx.close()

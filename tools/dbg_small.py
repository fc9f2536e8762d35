#!/usr/bin/env python3
"""Debug: the query set of test_every_scan_path on the 60k-doc corpus, GPU vs oracle, current env."""
import os, sys, random, struct
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nxsearch_amd as N
import oracle_lib as O
from nxsearch_amd import corpus
work = "/dev/shm/nxs_dbg"
os.makedirs(work, exist_ok=True)
c = corpus.write_corpus(work, 60_000, 3000, seed=21)
terms = corpus.term_strings(3000, seed=21)
qs = corpus.queries_bool5(terms, 24, seed=6, hi=400)
oidx = O.Index(c["terms"], c["dtmap"])
with N.Nxs(work) as nxs:
    gidx = nxs.open_files(c["terms"], c["dtmap"])
    bad = 0
    for limit in (10,):
        got = gidx.search_batch(qs, limit=limit, fuzzymatch=False)
        for q, g in zip(qs, got):
            w = oidx.search(q, limit=limit, fuzzymatch=False)
            if [d for d, _ in g] != [d for d, _ in w]:
                bad += 1
                if bad <= 3:
                    print("MISMATCH", q)
                    print("  got ", [(d, round(s, 4)) for d, s in g])
                    print("  want", [(d, round(s, 4)) for d, s in w])
    print("env", {k: v for k, v in os.environ.items() if k.startswith("NXS_")}, "bad", bad, "of", len(qs))
    gidx.close()

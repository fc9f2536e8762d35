#!/usr/bin/env python3
"""Which queries of the rotated C3 batches leave the candidate filter (exact re-queries)?
Per batch variant: re-queries at default settings and with the dense-term class / mask path off."""
import os, sys, time, json
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
tid = {t.decode(): i + 1 for i, t in enumerate(terms[:1000])}
for env in ({}, {"NXS_GPU_NODROP": "1"}, {"NXS_GPU_NOSCANM": "1"}):
    for k in ("NXS_GPU_NODROP", "NXS_GPU_NOSCANM"):
        os.environ.pop(k, None)
    os.environ.update(env)
    idx.reconfigure()
    for v in range(4):
        qs = corpus.queries_bool5(terms, 1024, seed=3 + 100 * v, hi=1000)
        idx.search_batch(qs, limit=10, fuzzymatch=False)
        idx.host_profile()
        t0 = time.perf_counter()
        for _ in range(3):
            idx.search_batch(qs, limit=10, fuzzymatch=False)
        dt = (time.perf_counter() - t0) / 3
        hp = idx.host_profile()
        print(env, "variant", v, "ms/batch %.2f" % (dt * 1e3), "exact_requeries/batch", hp["exact_requeries"] / 3.0, flush=True)
        if not env and hp["exact_requeries"]:
            # find them: one query at a time never overflows the same way; run halves
            bad = []
            for i in range(0, 1024, 64):
                idx.host_profile()
                idx.search_batch(qs[i:i + 64] * 4, limit=10, fuzzymatch=False)
                if idx.host_profile()["exact_requeries"]:
                    bad.append(i)
            print("   blocks of 64 with re-queries when run alone x4:", bad)

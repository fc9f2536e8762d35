#!/usr/bin/env python3
"""TF-IDF's dense-term caps: blocking C3 batches for a few NXS_GPU_OUTL_SHARE values (the share is
applied when the impacts are built, so the index is opened anew for each)."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nxsearch_amd as N
from nxsearch_amd import corpus
docs, nterms, batch = 10_000_000, 1_000_000, 1024
work = "/dev/shm/nxs_probe_%d_%d" % (docs, nterms)
if not os.path.exists(os.path.join(work, "done")):
    info = corpus.write_corpus(work, docs, nterms, seed=0)
    json.dump(info, open(os.path.join(work, "done"), "w"))
info = json.load(open(os.path.join(work, "done")))
terms = corpus.term_strings(nterms, 0)
qsets = [corpus.queries_bool5(terms, batch, seed=3 + v, hi=1000) for v in range(4)]
for share in sys.argv[1:] or ["0", "4", "8", "16", "32", "64"]:
    if share == "0":
        os.environ["NXS_GPU_TFIDF_NODROP"] = "1"
    else:
        os.environ.pop("NXS_GPU_TFIDF_NODROP", None)
        os.environ["NXS_GPU_OUTL_SHARE"] = share
    with N.Nxs(work) as nxs:
        idx = nxs.open_files(info["terms"], info["dtmap"], algo="TF-IDF")
        for qs in qsets:
            idx.search_batch(qs, limit=10, algo="TF-IDF", fuzzymatch=False)
        t0 = time.perf_counter()
        n = 0
        for rep in range(5):
            for qs in qsets:
                idx.search_batch(qs, limit=10, algo="TF-IDF", fuzzymatch=False)
                n += len(qs)
        dt = time.perf_counter() - t0
        print("share %s: %.0f queries/s blocking, %.3f ms per batch" % (share, n / dt, 1e3 * dt / 20), flush=True)
        idx.close()

#!/usr/bin/env python3
"""N1 in isolation: bench.py's refresh_measurements() on a fresh C3 corpus copy (both ranking
functions materialised first, as in the bench), twice in a row on the same index."""
import os, sys, json, types, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import nxsearch_amd as N
from nxsearch_amd import corpus
docs, nterms = int(os.environ.get("DOCS", 10_000_000)), int(os.environ.get("TERMS", 1_000_000))
work = "/dev/shm/nxs_refresh_probe"
shutil.rmtree(work, ignore_errors=True)
info = corpus.write_corpus(work, docs, nterms, seed=0)
terms = corpus.term_strings(nterms, 0)
args = types.SimpleNamespace(docs=docs, terms=nterms, limit=10)
with N.Nxs(work) as nxs:
    idx = nxs.open_files(info["terms"], info["dtmap"])
    idx.search(terms[5].decode(), limit=10, algo="TF-IDF", fuzzymatch=False)
    out = bench.refresh_measurements(args, idx, info, terms)
    print(json.dumps(out))
    idx.close()
shutil.rmtree(work, ignore_errors=True)

import os, sys, json, ctypes as C, time
sys.path.insert(0, "/root/repo")
import nxsearch_amd as N
from nxsearch_amd import corpus
docs, nterms = 10_000_000, 1_000_000
work = "/dev/shm/nxs_probe_%d_%d" % (docs, nterms)
if not os.path.exists(os.path.join(work, "done")):
    info = corpus.write_corpus(work, docs, nterms, seed=0)
    json.dump(info, open(os.path.join(work, "done"), "w"))
info = json.load(open(os.path.join(work, "done")))
terms = corpus.term_strings(nterms, 0)
nxs = N.Nxs(work); idx = nxs.open_files(info["terms"], info["dtmap"])
qs = corpus.queries_bool5(terms, 1024, seed=3, hi=1000)
for algo in ("BM25", "TF-IDF"):
    idx.search_batch(qs, limit=10, algo=algo, fuzzymatch=False)
    idx.host_profile()
    t0 = time.perf_counter()
    for _ in range(3):
        idx.search_batch(qs, limit=10, algo=algo, fuzzymatch=False)
    dt = (time.perf_counter() - t0) / 3
    print(algo, os.environ.get("NXS_GPU_NODROP"), "ms/batch %.2f" % (1e3 * dt), idx.host_profile(), flush=True)

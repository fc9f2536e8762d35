#!/usr/bin/env python3
"""Per-query device time of the C3 batch's OR queries that hold a dense term
(k_scanm<.., DROP> class): each query alone, so the time is its slowest
wavefront.  GPU only.  Usage: python tools/qprobe.py [n]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import nxsearch_amd as N
from nxsearch_amd import corpus
docs, nterms = 10_000_000, 1_000_000
work = "/dev/shm/nxs_probe_%d_%d" % (docs, nterms)
if not os.path.exists(os.path.join(work, "done")):
    info = corpus.write_corpus(work, docs, nterms, seed=0)
    json.dump(info, open(os.path.join(work, "done"), "w"))
info = json.load(open(os.path.join(work, "done")))
terms = corpus.term_strings(nterms, 0)
rank = {t: i + 1 for i, t in enumerate(terms[:2000])}
nxs = N.Nxs(work)
idx = nxs.open_files(info["terms"], info["dtmap"])
dev = torch.device("cuda", 0)
k = 10
qs = corpus.queries_bool5(terms, 1024, seed=3, hi=1000)
sel = []
for q in qs:
    if " OR " in q:
        rs = sorted(rank[w.encode()] for w in q.split(" OR "))
        if rs[0] <= 27:
            sel.append((q, rs))
d_ids = torch.empty((1, k), dtype=torch.int64, device=dev)
d_sc = torch.empty((1, k), dtype=torch.float32, device=dev)
d_cnt = torch.empty((1,), dtype=torch.int32, device=dev)
out = []
for q, rs in sel[:int(sys.argv[1]) if len(sys.argv) > 1 else 80]:
    plans, errs = idx.plan_batch([q], limit=k, fuzzymatch=False)
    for _ in range(2):
        idx.search_dev(plans, 1, k, N.BM25, d_ids.data_ptr(), d_sc.data_ptr(), d_cnt.data_ptr())
    idx.set_profiling(True); idx.profile(reset=True)
    for _ in range(3):
        idx.search_dev(plans, 1, k, N.BM25, d_ids.data_ptr(), d_sc.data_ptr(), d_cnt.data_ptr())
    p = idx.profile(reset=True); idx.set_profiling(False)
    out.append(((p["scan_ms"] + p["replay_ms"]) / p["launches"], rs))
out.sort(reverse=True)
for ms, rs in out:
    print("%8.3f ms  ranks %s" % (ms, rs))

#!/usr/bin/env python3
"""Diagnostic: k_scanr event counts on the C3 batch's AND half.
Needs the stats build:  make -C nxsearch_amd/csrc variant SFX=stats XFLAGS=-DNXS_STATS ; NXS_GPU_LIB=.../libnxsearch_gpu_stats.so"""
import ctypes as C, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import nxsearch_amd as N
from nxsearch_amd import corpus
docs, nterms, batch, k = 10_000_000, 1_000_000, 1024, 10
work = "/dev/shm/nxs_probe_%d_%d" % (docs, nterms)
if not os.path.exists(os.path.join(work, "done")):
    info = corpus.write_corpus(work, docs, nterms, seed=0)
    json.dump(info, open(os.path.join(work, "done"), "w"))
info = json.load(open(os.path.join(work, "done")))
terms = corpus.term_strings(nterms, 0)
nxs = N.Nxs(work)
idx = nxs.open_files(info["terms"], info["dtmap"])
dev = torch.device("cuda", 0)
d_ids = torch.empty((batch, k), dtype=torch.int64, device=dev)
d_sc = torch.empty((batch, k), dtype=torch.float32, device=dev)
d_cnt = torch.empty((batch,), dtype=torch.int32, device=dev)
qs = [q for q in corpus.queries_bool5(terms, 2 * batch, seed=3, hi=1000) if " AND " in q][:batch]
L = N.lib()
L.nxsgpu_debug_stats_req.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
names = ["waves", "rounds", "driver_docs", "slot1_visits", "later_slot_visits", "skip_rotates", "skip_jumps", "lookup_rotates", "rounds_with_survivors", "survivors"]
plans, errs = idx.plan_batch(qs, limit=k, fuzzymatch=False)
out = (C.c_ulonglong * 16)()
idx.search_dev(plans, batch, k, N.BM25, d_ids.data_ptr(), d_sc.data_ptr(), d_cnt.data_ptr())
L.nxsgpu_debug_stats_req(out, 1)
idx.search_dev(plans, batch, k, N.BM25, d_ids.data_ptr(), d_sc.data_ptr(), d_cnt.data_ptr())
L.nxsgpu_debug_stats_req(out, 1)
v = dict(zip(names, list(out)))
w = max(v["waves"], 1)
print({n: round(v[n] / w, 2) for n in names[1:]}, "waves", v["waves"], "per query:", {n: round(v[n] / batch, 1) for n in ("waves", "rounds", "driver_docs")})

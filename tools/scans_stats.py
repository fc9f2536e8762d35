#!/usr/bin/env python3
"""Diagnostic: k_scans event counts / cycle spans on the kprobe query sets.
Needs the stats build:  make -C nxsearch_amd/csrc variant SFX=stats XFLAGS=-DNXS_STATS
Run with NXS_GPU_LIB=.../libnxsearch_gpu_stats.so"""
import ctypes as C, os, sys, json, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import nxsearch_amd as N
from nxsearch_amd import corpus
docs, nterms, batch, k = 10_000_000, 1_000_000, int(os.environ.get("STATS_BATCH", "1024")), 10
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
rng = random.Random(1)
T = lambda r: terms[r - 1].decode()
sets = {
  "C": [" OR ".join(T(r) for r in rng.sample(range(1, 1001), 5)) for _ in range(batch)],
  "E": [" OR ".join(T(r) for r in rng.sample(range(500, 1001), 5)) for _ in range(batch)],
  "K": [" OR ".join(T(r) for r in rng.sample(range(100, 1001), 5)) for _ in range(batch)],
  "M": [" OR ".join(T(r) for r in rng.sample(range(30, 101), 5)) for _ in range(batch)],
  "X": [" OR ".join(T(r) for r in rng.sample(range(28, 1001), 5)) for _ in range(batch)],
  "R": [" OR ".join(T(r) for r in [rng.randint(1, 27)] + rng.sample(range(100, 1001), 4)) for _ in range(batch)],

}
if os.environ.get("STATS_SETS"):
    sets = {k: v for k, v in sets.items() if k in os.environ["STATS_SETS"].split(",")}
L = N.lib()
L.nxsgpu_debug_stats_stripe.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
names = ["waves", "stripes", "passes", "windows", "flushes", "pending", "unique", "rounds", "search_steps", "emitted",
         "flush_cyc", "wave_cyc", "postings", "overflows", "ovf_rest", "ovf_segcap"]
for name, qs in sets.items():
    plans, errs = idx.plan_batch(qs, limit=k, fuzzymatch=False)
    idx.search_dev(plans, batch, k, N.BM25, d_ids.data_ptr(), d_sc.data_ptr(), d_cnt.data_ptr())
    out = (C.c_ulonglong * 16)()
    L.nxsgpu_debug_stats_stripe(out, 1)
    idx.search_dev(plans, batch, k, N.BM25, d_ids.data_ptr(), d_sc.data_ptr(), d_cnt.data_ptr())
    L.nxsgpu_debug_stats_stripe(out, 1)
    v = dict(zip(names, list(out)))
    w = max(v["waves"], 1)
    print(name, {n: round(v[n] / w, 2) for n in names[1:]}, "waves", v["waves"], "ovf total %d (cold-pass %d) rest %d segcap %d" % (
        v["overflows"] & 0xffff, v["overflows"] >> 16, v["ovf_rest"], v["ovf_segcap"]), flush=True)

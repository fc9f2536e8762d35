#!/usr/bin/env python3
"""Kernel probe: times k_scan on query mixes of different density (GPU only).
Usage: python tools/kprobe.py [--docs N --terms T]"""
import argparse, os, sys, time, json, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import nxsearch_amd as N
from nxsearch_amd import corpus

ap = argparse.ArgumentParser()
ap.add_argument("--docs", type=int, default=10_000_000)
ap.add_argument("--terms", type=int, default=1_000_000)
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--sets", default="A,B,C,D,E,F,G,H,I,J")
a = ap.parse_args()
work = "/dev/shm/nxs_probe_%d_%d" % (a.docs, a.terms)
if not os.path.exists(os.path.join(work, "done")):
    info = corpus.write_corpus(work, a.docs, a.terms, seed=0)
    json.dump(info, open(os.path.join(work, "done"), "w"))
info = json.load(open(os.path.join(work, "done")))
terms = corpus.term_strings(a.terms, 0)
nxs = N.Nxs(work)
idx = nxs.open_files(info["terms"], info["dtmap"])
dev = torch.device("cuda", 0)
k = 10
d_ids = torch.empty((a.batch, k), dtype=torch.int64, device=dev)
d_sc = torch.empty((a.batch, k), dtype=torch.float32, device=dev)
d_cnt = torch.empty((a.batch,), dtype=torch.int32, device=dev)
rng = random.Random(1)
T = lambda r: terms[r - 1].decode()
sets = {
  "A": ("1 term, rank 1..4 (dense stream)", [T(rng.randint(1, 4)) for _ in range(a.batch)]),
  "B": ("1 term, rank 10..10000 (C2)", [T(rng.randint(10, 10000)) for _ in range(a.batch)]),
  "C": ("5-term OR rank 1..1000", [" OR ".join(T(r) for r in rng.sample(range(1, 1001), 5)) for _ in range(a.batch)]),
  "D": ("5-term AND rank 1..1000", [" AND ".join(T(r) for r in rng.sample(range(1, 1001), 5)) for _ in range(a.batch)]),
  "E": ("5-term OR rank 500..1000 (sparse)", [" OR ".join(T(r) for r in rng.sample(range(500, 1001), 5)) for _ in range(a.batch)]),
  "F": ("2-term OR rank 1..50 (dense)", [" OR ".join(T(r) for r in rng.sample(range(1, 51), 2)) for _ in range(a.batch)]),
  "G": ("2-term AND rank 1..1000", [" AND ".join(T(r) for r in rng.sample(range(1, 1001), 2)) for _ in range(a.batch)]),
  "H": ("3-term AND rank 1..1000", [" AND ".join(T(r) for r in rng.sample(range(1, 1001), 3)) for _ in range(a.batch)]),
  "I": ("a AND (b OR c) rank 1..1000", ["%s AND (%s OR %s)" % tuple(T(r) for r in rng.sample(range(1, 1001), 3)) for _ in range(a.batch)]),
  "K": ("5-term OR rank 100..1000 (no dense)", [" OR ".join(T(r) for r in rng.sample(range(100, 1001), 5)) for _ in range(a.batch)]),
  "L": ("5-term OR rank 1..30 (all dense)", [" OR ".join(T(r) for r in rng.sample(range(1, 31), 5)) for _ in range(a.batch)]),
  "M": ("5-term OR rank 30..100 (medium)", [" OR ".join(T(r) for r in rng.sample(range(30, 101), 5)) for _ in range(a.batch)]),
  "N": ("7-term OR rank 100..1000", [" OR ".join(T(r) for r in rng.sample(range(100, 1001), 7)) for _ in range(a.batch)]),
  "O": ("2-term OR rank 100..1000", [" OR ".join(T(r) for r in rng.sample(range(100, 1001), 2)) for _ in range(a.batch)]),
  "P": ("(a AND b) OR (c AND d) rank 100..1000", ["(%s AND %s) OR (%s AND %s)" % tuple(T(r) for r in rng.sample(range(100, 1001), 4)) for _ in range(a.batch)]),
  "Q": ("a OR b OR (c AND NOT d) rank 100..1000", ["%s OR %s OR (%s AND NOT %s)" % tuple(T(r) for r in rng.sample(range(100, 1001), 4)) for _ in range(a.batch)]),
  "R": ("1 dense (1..27) + 4 sparse (100..1000) OR", [" OR ".join(T(r) for r in [rng.randint(1, 27)] + rng.sample(range(100, 1001), 4)) for _ in range(a.batch)]),
  "S": ("1 very dense (1..5) + 4 sparse OR", [" OR ".join(T(r) for r in [rng.randint(1, 5)] + rng.sample(range(100, 1001), 4)) for _ in range(a.batch)]),
  "U": ("2 dense (1..27) + 3 sparse OR", [" OR ".join(T(r) for r in rng.sample(range(1, 28), 2) + rng.sample(range(100, 1001), 3)) for _ in range(a.batch)]),
  "V": ("1 dense (15..27) + 4 sparse OR", [" OR ".join(T(r) for r in [rng.randint(15, 27)] + rng.sample(range(100, 1001), 4)) for _ in range(a.batch)]),
  "W": ("1 dense (1..27) + 4 sparse (500..1000) OR", [" OR ".join(T(r) for r in [rng.randint(1, 27)] + rng.sample(range(500, 1001), 4)) for _ in range(a.batch)]),
  "X": ("5-term OR rank 28..1000 (C3 plain class)", [" OR ".join(T(r) for r in rng.sample(range(28, 1001), 5)) for _ in range(a.batch)]),
  "J": ("2-term AND rank 1..50 (dense)", [" AND ".join(T(r) for r in rng.sample(range(1, 51), 2)) for _ in range(a.batch)]),
}
for name in a.sets.split(","):
    desc, qs = sets[name]
    plans, errs = idx.plan_batch(qs, limit=k, fuzzymatch=False)
    for _ in range(2):
        idx.search_dev(plans, a.batch, k, N.BM25, d_ids.data_ptr(), d_sc.data_ptr(), d_cnt.data_ptr())
    idx.set_profiling(True); idx.profile(reset=True)
    for _ in range(a.reps):
        idx.search_dev(plans, a.batch, k, N.BM25, d_ids.data_ptr(), d_sc.data_ptr(), d_cnt.data_ptr())
    p = idx.profile(reset=True); idx.set_profiling(False)
    ms = p["scan_ms"] / p["launches"]; post = p["postings"] / p["launches"]
    print("%s %-36s scan %8.3f ms  replay %6.3f ms  postings %.3e  %7.1f GB/s  %6.1f Mpost/ms" % (
        name, desc, ms, p["replay_ms"] / p["launches"], post, post * 8 / ms / 1e6, post / ms / 1e6), flush=True)

#!/usr/bin/env python3
"""Stress of the batches-in-flight machinery against the oracle (round 4): random mixed-operator queries over a
synthetic Zipf corpus, FOUR batches in flight with the limit changing from batch to batch (10 / 1000 / 64 / 300:
top-k and MODE_BIG batches interleaved on the replay streams), plain and through a one-rank RCCL communicator
(record blocks, all-gather, copy-out kernel).  argv: docs, queries, rounds."""
import os, sys, random, struct, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nxsearch_amd as N
import oracle_lib as O
from nxsearch_amd import corpus, multi
docs, nterms = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000, 50_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 320
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
work = "/dev/shm/nxs_stress"
os.makedirs(work, exist_ok=True)
c = corpus.write_corpus(work, docs, nterms, seed=77)
terms = corpus.term_strings(nterms, seed=77)
T = lambda r: terms[r - 1].decode()
rng = random.Random(11)
qs = []
for _ in range(nq):
    n = rng.randint(1, 6)
    lo, hi = rng.choice([(1, 30), (1, 300), (20, 2000), (200, 20000), (1, 50000)])
    ts = [T(r) for r in rng.sample(range(lo, hi + 1), n)]
    q = ts[0]
    for t in ts[1:]:
        q += rng.choice([" OR ", " ", " OR ", " AND NOT ", " AND ", " AND "]) + t
    qs.append(q)
batches = [qs[i:i + 80] for i in range(0, nq, 80)]
limits = [10, 1000, 64, 300]
oidx = O.Index(c["terms"], c["dtmap"])
bits = lambda x: struct.pack("<f", x)
want, bad, checked = {}, 0, 0
t0 = time.time()
for mode in ("plain", "one-rank communicator"):
    with N.Nxs(work) as nxs:
        gidx = nxs.open_files(c["terms"], c["dtmap"])
        if mode != "plain":
            multi.attach(nxs, gidx, 0, 1)
        plan = [(b, limits[(i + r) % 4]) for r in range(rounds) for i, b in enumerate(batches)]
        inflight = []
        def collect():
            global bad, checked
            bb, kk = inflight.pop(0)
            got = gidx.search_batch_end()
            for q, g in zip(bb, got):
                key = (q, kk)
                if key not in want:
                    want[key] = oidx.search(q, limit=kk, fuzzymatch=False)
                w = want[key]
                checked += 1
                if [d for d, _ in g] != [d for d, _ in w] or [bits(s) for _, s in g] != [bits(s) for _, s in w]:
                    bad += 1
                    print("MISMATCH", mode, kk, q, flush=True)
        for b, k in plan:
            gidx.search_batch_begin(b, limit=k, fuzzymatch=False)
            inflight.append((b, k))
            if len(inflight) == 4:
                collect()
        while inflight:
            collect()
        hp = gidx.host_profile()
        print("%s: %d batches, %d searches checked so far, %d mismatches, exact re-queries %d  (%.0f s)" % (
            mode, len(plan), checked, bad, hp.get("exact_requeries", -1), time.time() - t0), flush=True)
        gidx.close()
print("stress_pipeline: %d docs, %d queries x limits %s x %d rounds x 2 modes: %d searches, %d mismatches" % (
    docs, nq, limits, rounds, checked, bad))
sys.exit(1 if bad else 0)

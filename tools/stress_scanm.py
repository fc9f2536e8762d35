#!/usr/bin/env python3
"""One-off stress of the mask path against the oracle (argv: docs, queries, [mixed]): random pure-OR
(or mixed-operator) queries of
2..8 terms over a synthetic Zipf corpus, several limits, both ranking functions,
default routing and k_scanm forced for every density."""
import os, sys, random, struct, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nxsearch_amd as N
import oracle_lib as O
from nxsearch_amd import corpus
docs, nterms = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000, 50_000
work = "/dev/shm/nxs_stress"
os.makedirs(work, exist_ok=True)
c = corpus.write_corpus(work, docs, nterms, seed=77)
terms = corpus.term_strings(nterms, seed=77)
T = lambda r: terms[r - 1].decode()
rng = random.Random(5)
qs = []
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 400):
    n = rng.randint(2, 8)
    lo, hi = rng.choice([(1, 30), (1, 300), (20, 2000), (200, 20000), (1, 50000)])
    ts = [T(r) for r in rng.sample(range(lo, hi + 1), n)]
    if len(sys.argv) > 3 and sys.argv[3] == "mixed":
        # random operators (juxtaposition = OR): pure OR, OR-like and AND-like shapes
        q = ts[0]
        for t in ts[1:]:
            q += rng.choice([" OR ", " ", " OR ", " AND NOT ", " AND "]) + t
        if rng.random() < 0.3 and n >= 4:
            q = "(%s OR %s) OR (%s AND NOT %s)" % tuple(ts[:4]) + "".join(" OR " + t for t in ts[4:])
        qs.append(q)
    else:
        qs.append(" OR ".join(ts))
oidx = O.Index(c["terms"], c["dtmap"])
bits = lambda x: struct.pack("<f", x)
bad = 0
t0 = time.time()
want = {}
for env in ({}, {"NXS_GPU_DROP_MINPOST": "1"}, {"NXS_GPU_DROP_MINPOST": "1", "NXS_GPU_WAVES": "256"},
            {"NXS_GPU_SCANM_DENS": "0.01", "NXS_GPU_DROP_MINPOST": "1"}, {"NXS_GPU_NODROP": "1"},
            {"NXS_GPU_SCANM_DENS": "1.0"}, {"NXS_GPU_SCANM_DENS": "1.0", "NXS_GPU_WAVES": "256"},
            {"NXS_GPU_NOSCANM": "1"},
            # round 4: presence bits for every eligible query, the sparse + dense class on k_scanb, conjunctions
            # through the block bitmaps always / never, no classes sent ahead
            {"NXS_GPU_SCANB_DENS": "1.0"}, {"NXS_GPU_SCANB_DENS": "1.0", "NXS_GPU_SCANM_DENS": "1.0", "NXS_GPU_WAVES": "256"},
            {"NXS_GPU_DROPB": "1", "NXS_GPU_DROP_MINPOST": "1"}, {"NXS_GPU_BM_GAIN": "0"}, {"NXS_GPU_NOBLKMAP": "1"},
            {"NXS_GPU_DROP_SPLIT": "0", "NXS_GPU_AND_NOEARLY": "1"},
            # round 5: the byte map on doc stripes (k_scans) for every density / with short ranges / off / with a rank
            # directory for every term / also as the sparse + dense class's second kernel
            {"NXS_GPU_SCANM_DENS": "1.0", "NXS_GPU_NOSCANB": "1"}, {"NXS_GPU_SCANM_DENS": "1.0", "NXS_GPU_NOSCANB": "1", "NXS_GPU_WAVES": "256"},
            {"NXS_GPU_NOSCANS": "1"}, {"NXS_GPU_BM_SHARE": "1073741824", "NXS_GPU_NOSCANB": "1"},
            {"NXS_GPU_SCANS_DROP": "1", "NXS_GPU_DROP_MINPOST": "1"}, {"NXS_GPU_SCANS_DROP": "1", "NXS_GPU_DROP_MINPOST": "1", "NXS_GPU_WAVES": "256"}):
    if os.environ.get("STRESS_ONLY") and os.environ["STRESS_ONLY"] not in env:
        continue
    for kk, v in env.items():
        os.environ[kk] = v
    with N.Nxs(work) as nxs:
        gidx = nxs.open_files(c["terms"], c["dtmap"])
        for limit in (1, 10, 64):
            for algo, name in ((1, "BM25"), (0, "TF-IDF")):
                got = gidx.search_batch(qs, limit=limit, algo=name, fuzzymatch=False)
                for q, g in zip(qs, got):
                    key = (q, limit, algo)
                    if key not in want:
                        want[key] = oidx.search(q, algo=algo, limit=limit, fuzzymatch=False)
                    w = want[key]
                    if [d for d, _ in g] != [d for d, _ in w] or [bits(s) for _, s in g] != [bits(s) for _, s in w]:
                        bad += 1
                        if bad <= 5:
                            print("MISMATCH", env, q, limit, name, g[:3], w[:3], flush=True)
                print("env %s limit %d %s done, %d mismatches so far, %.0f s" % (env, limit, name, bad, time.time() - t0), flush=True)
        gidx.close()
    for kk in env:
        del os.environ[kk]
print("TOTAL mismatches:", bad)
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X query/ranking path.

Metric (BASELINE.json): queries/sec + p50 latency, 10M-doc BM25 top-10;
fuzzy candidates/sec.  Default workload = configs[2] ("C3"): 10M docs / 1M
terms synthetic Zipf corpus, 5-term AND/OR BM25 queries, batch 1024, top-10 --
the configuration the metric is quoted on; it fits one GPU.

A "step" = one batch of 1024 resolved queries through the device path
(nxsgpu_search_dev_begin/_end: host planning, plan upload, k_cursors, the scan
kernels, k_replay, results left in HBM) plus, for N > 1, one RCCL all-gather of
the per-GPU top-k.  Steps are software-pipelined two deep: the host plans and
uploads step i+1 while the device runs step i; all K steps have completed when
the timed region ends.  The index is resident in HBM before it starts.  Launch:

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line.  Extra objects in that line:
  roofline      achieved algorithmic HBM GB/s of k_scan (HIP events on its own
                stream) against the 8 TB/s HBM3E peak
  cpu_baseline  the oracle (plain-C restatement of the reference) timed on one
                host core on a bounded sample of the same workload
  latency / fuzzy / e2e   p50 single-query latency through nxs_index_search(),
                fuzzy candidates/sec (device vs the genuine reference BK-tree
                on one core), and the rate through the full C API.
"""
import argparse
import ctypes as C
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
POSTING_BYTES = 8               # u32 doc ordinal + f32 score per posting
RESULT_BYTES = 12               # (u64 doc id, f32 score) per returned result


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--terms", type=int, default=1_000_000)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--limit", type=int, default=10)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=20.0,
                    help="budget of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the latency / fuzzy / e2e side measurements")
    ap.add_argument("--workdir", default=None)
    ap.add_argument("--keep", action="store_true")
    return ap.parse_args()


def pick_workdir(args, need_bytes):
    if args.workdir:
        return args.workdir
    for base in ("/dev/shm", tempfile.gettempdir()):
        try:
            st = os.statvfs(base)
            if st.f_bavail * st.f_frsize > need_bytes * 1.3:
                return os.path.join(base, "nxs_bench_%d_%d_%d_%d" % (
                    os.getuid(), args.docs, args.terms, args.seed))
        except OSError:
            pass
    return os.path.join(tempfile.gettempdir(), "nxs_bench_%d" % os.getuid())


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run for --gpus > 1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    import nxsearch_amd as N
    from nxsearch_amd import corpus

    if not torch.cuda.is_available() or N.lib().nxsgpu_device_count() <= 0:
        sys.exit("bench.py needs a HIP device; there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    def barrier():
        if dist is not None:
            dist.barrier()

    # ---- corpus (rank 0 writes, everyone maps the same files) -------------
    need = args.docs * (16 + 8 * 33) + args.terms * 32
    work = pick_workdir(args, need)
    t0 = time.time()
    marker = os.path.join(work, "done")
    if rank == 0 and not os.path.exists(marker):
        os.makedirs(work, exist_ok=True)
        info = corpus.write_corpus(work, args.docs, args.terms, seed=args.seed)
        with open(marker, "w") as f:
            json.dump(info, f)
    barrier()
    with open(marker) as f:
        info = json.load(f)
    t_gen = time.time() - t0
    terms = corpus.term_strings(args.terms, seed=args.seed)

    # ---- index resident in HBM --------------------------------------------
    os.environ["NXS_GPU_DEVICE"] = str(local_rank)
    t0 = time.time()
    nxs = N.Nxs(work)
    idx = nxs.open_files(info["terms"], info["dtmap"], algo="BM25")
    t_load = time.time() - t0

    # ---- the batch: C3 queries; weak scaling = 1024 queries per GPU, sharded
    #      by query (contiguous slices), index replicated on every GPU --------
    from nxsearch_amd import multi
    k = args.limit
    all_queries = corpus.queries_bool5(terms, args.batch * world, seed=3, hi=1000)
    # two batches in flight (nxsgpu_search_dev_begin/_end): the host plans and
    # uploads step i+1 while step i runs; each has its own output buffers
    sbs = [multi.ShardedBatch(len(all_queries), k, rank, world, dev) for _ in range(2)]
    sb = sbs[0]
    queries = all_queries[sb.lo:sb.hi]
    plans, errs = idx.plan_batch(queries, limit=args.limit, algo="BM25", fuzzymatch=False)
    assert not any(errs)

    def begin(i):
        o = sbs[i % 2]
        idx.search_dev_begin(plans, len(queries), k, N.BM25, o.ids.data_ptr(),
                             o.scores.data_ptr(), o.counts.data_ptr())

    def end(i):
        r = idx.search_dev_end()
        assert r == 0, "a query needed the exact two-pass path"
        # per-GPU top-k records over xGMI (RCCL all-gather); ~124 B per query
        sbs[i % 2].gather(dist)

    def run(n):
        """n steps, software-pipelined: every step is one whole batch through
        plan upload, cursors, scans, replay (+ all-gather); all have completed
        when this returns."""
        for i in range(n):
            begin(i)
            if i:
                end(i - 1)
        if n:
            end(n - 1)

    run(args.warmup)
    idx.set_profiling(True)
    idx.profile(reset=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = idx.profile(reset=True)
    idx.set_profiling(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_q = args.batch * world * args.steps
    qps = total_q / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    # ---- roofline of the dominant kernel (k_scan) on this rank --------------
    matched = int(sbs[(args.steps - 1) % 2].counts.sum().item())
    alg_bytes = prof["postings"] * POSTING_BYTES / max(prof["launches"], 1) \
        + matched * RESULT_BYTES
    scan_ms = prof["scan_ms"] / max(prof["launches"], 1)
    achieved = alg_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    # one scan launch per query class and step; on C3: k_scanm<5,false> (pure OR of
    # sparse terms), k_scan8<0,5,1> (pure OR with a dense term) and k_scanr<0,5>
    # (required terms); kernel_ms is their sum per step (HIP events on the
    # library's stream)
    roofline = {"bound": "hbm", "kernel": "k_scanm+k_scan8+k_scanr", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": pmc_traffic(args, world),
                "alg_bytes_per_launch": int(alg_bytes),
                "kernel_ms": round(scan_ms, 4),
                "replay_ms": round(prof["replay_ms"] / max(prof["launches"], 1), 4)}

    out = {
        "metric": "queries/sec (10M-doc BM25 top-10, 5-term AND/OR, batch 1024)",
        "value": round(qps, 1), "unit": "queries/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "C3: %d docs / %d terms Zipf, 5-term AND/OR BM25 top-%d, "
                               "batch %d per GPU" % (args.docs, args.terms, k, args.batch),
                   "postings": info["postings"], "parallelism": "query-sharded x%d, "
                   "index replicated" % world},
        "roofline": roofline,
        "setup_s": {"corpus": round(t_gen, 1), "index_load": round(t_load, 1)},
    }

    if rank == 0 and world == 1:
        if not args.no_extras:
            out.update(side_measurements(args, idx, terms, queries, torch))
        if args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(args, info, queries, idx)
    barrier()
    idx.close()
    nxs.close()
    if rank == 0:
        print(json.dumps(out), flush=True)
        if not args.keep and not args.workdir:
            shutil.rmtree(work, ignore_errors=True)
    if dist is not None:
        dist.destroy_process_group()


def pmc_traffic(args, world):
    """HBM bytes per step (all scan launches) from the committed rocprofv3 --pmc
    passes (profiles/r1_s2_pmc_summary.json: FETCH_SIZE x2 on gfx950 + WRITE_SIZE),
    valid only for the workload they were collected on; else None."""
    try:
        with open(os.path.join(ROOT, "profiles", "r1_s2_pmc_summary.json")) as f:
            p = json.load(f)
        w = p["workload"]
        if (w["docs"], w["terms"], w["batch"], w["limit"]) == \
                (args.docs, args.terms, args.batch, args.limit) and world == 1:
            return int(p["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        pass
    return None


def side_measurements(args, idx, terms, queries, torch):
    """p50 latency (C2-style single-term queries through nxs_index_search),
    the full C-API batch rate, and the fuzzy path (C4)."""
    import nxsearch_amd as N
    from nxsearch_amd import corpus
    res = {}
    # single-query latency, top-10, one term of rank uniform in [10, 10^4]
    singles = corpus.queries_single(terms, 200, seed=5)
    for q in singles[:20]:
        idx.search(q, limit=args.limit, fuzzymatch=False)
    lat = []
    for q in singles:
        t0 = time.perf_counter()
        idx.search(q, limit=args.limit, fuzzymatch=False)
        lat.append(time.perf_counter() - t0)
    lat.sort()
    res["latency"] = {"p50_ms": round(1e3 * lat[len(lat) // 2], 4),
                      "p95_ms": round(1e3 * lat[int(len(lat) * 0.95)], 4),
                      "what": "nxs_index_search(), single-term BM25 top-%d, n=%d" % (args.limit, len(lat))}
    # the whole C API on the batch: parse + resolve + plan + device + resp objects
    idx.search_batch(queries, limit=args.limit, fuzzymatch=False)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        idx.search_batch(queries, limit=args.limit, fuzzymatch=False)
    dt = (time.perf_counter() - t0) / reps
    res["e2e"] = {"queries_per_s": round(len(queries) / dt, 1),
                  "what": "nxs_index_search_batch(): host parse/plan + H2D + kernels + D2H + resp"}
    # fuzzy: C4 = 1024 tokens, Levenshtein d<=2 over the BK-tree of all terms
    toks = corpus.queries_fuzzy(terms, args.batch, seed=4)
    idx.fuzzy(toks[:64])
    idx.set_profiling(True)
    idx.profile(reset=True)
    t0 = time.perf_counter()
    ids = idx.fuzzy(toks)
    dt = time.perf_counter() - t0
    p = idx.profile(reset=True)
    idx.set_profiling(False)
    res["fuzzy"] = {"candidates_per_s": round(p["fuzzy_visits"] / (p["fuzzy_ms"] * 1e-3), 1)
                    if p["fuzzy_ms"] > 0 else None,
                    "tokens_per_s": round(len(toks) / dt, 1),
                    "candidates": p["fuzzy_visits"], "device_ms": round(p["fuzzy_ms"], 3),
                    "resolved": sum(1 for i in ids if i),
                    "what": "C4: %d tokens, d<=2 over a %d-term BK-tree" % (len(toks), args.terms)}
    return res


def cpu_baseline(args, info, queries, idx):
    """The oracle on ONE host core, on a bounded sample of the same batch
    (same corpus, first queries of the batch), next to the GPU number."""
    import oracle_lib as O
    t0 = time.time()
    oidx = O.Index(info["terms"], info["dtmap"])
    t_load = time.time() - t0
    budget = args.cpu_seconds
    n, pairs, t_used = 0, 0, 0.0
    mism = 0
    for q in queries:
        t0 = time.perf_counter()
        want = oidx.search(q, algo=O.BM25, limit=args.limit, fuzzymatch=False)
        t_used += time.perf_counter() - t0
        pairs += oidx.last_pairs
        got = idx.search(q, limit=args.limit, fuzzymatch=False)
        if got != want:
            mism += 1
        n += 1
        if t_used >= budget:
            break
    base = {"value": round(n / t_used, 3), "unit": "queries/s", "cores": 1, "kind": "port",
            "sample": "first %d queries of the batch, %.1f s, %d (doc,term) pairs scored; "
                      "oracle/nxs_oracle.c (reference algorithm restated in C)" % (n, t_used, pairs),
            "parity_mismatches": mism, "load_s": round(t_load, 1),
            "host_cpus": os.cpu_count()}
    # fuzzy baseline on the genuine reference bktree.c/levdist.c (oracle/_ref)
    if O.ref() is not None and not args.no_extras:
        from nxsearch_amd import corpus
        terms = corpus.term_strings(args.terms, seed=args.seed)
        t0 = time.time()
        tree = O.RefBKTree(terms)
        t_build = time.time() - t0
        toks = corpus.queries_fuzzy(terms, args.batch, seed=4)
        vis, t_used, m = 0, 0.0, 0
        for t in toks:
            t0 = time.perf_counter()
            _, nv = tree.search(t.encode(), 2)
            t_used += time.perf_counter() - t0
            vis += nv
            m += 1
            if t_used >= min(budget, 10.0):
                break
        base["fuzzy"] = {"candidates_per_s": round(vis / t_used, 1), "cores": 1,
                         "kind": "reference", "tokens": m, "build_s": round(t_build, 1),
                         "sample": "reference src/algo/bktree.c + levdist.c compiled in place (oracle/_ref)"}
        tree.close()
    oidx.close()
    return base


if __name__ == "__main__":
    main()

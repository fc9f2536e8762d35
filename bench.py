#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X query/ranking path.

Metric (BASELINE.json): queries/sec + p50 latency, 10M-doc BM25 top-10;
fuzzy candidates/sec.  Default workload = configs[2] ("C3"): 10M docs / 1M
terms synthetic Zipf corpus, 5-term AND/OR BM25 queries, batch 1024, top-10 --
the configuration the metric is quoted on; it fits one GPU.  `--workload`
selects the others (C2 single-term, C4 fuzzy, C5 mixed BM25 + fuzzy at
50M docs / batch 8192).

A "step" = one batch of query STRINGS through the public C API
(nxs_index_search_batch_begin/_end, include/nxs.h): lex + parse + resolve (fuzzy
misses on the device) + plan on the host's worker threads, upload, k_cursors,
the scan kernels, k_replay, (N > 1: ONE RCCL all-gather of the ranks' record
blocks, inside the library), copy to pinned memory, one nxs_resp_t per query,
every result read and every response released by a C consumer
(csrc/nxs_benchloop.c -- no Python inside the timed region).  Steps are
pipelined two deep, as the API allows: the host plans step i+1 while the GPU
runs step i; all K steps have completed when the timed region ends.  The index
is resident in HBM before it starts.  Launch:

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line.  Extra objects in that line:
  roofline      achieved algorithmic HBM GB/s of the scan kernels (HIP events on
                the library's own stream) against the 8 TB/s HBM3E peak and
                against the read bandwidth measured on this device
  cpu_baseline  the oracle (plain-C restatement of the reference) on ONE host
                core on a bounded sample of the same batch, and on one worker
                process per core (`np`, the reference's deployment model)
  device_resident_qps / blocking_qps / latency / tfidf / default_limit / fuzzy
                side measurements (N = 1 only)
"""
import argparse
import ctypes as C
import json
import math
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
POSTING_BYTES = 8               # u32 doc ordinal + f32 score per posting
RESULT_BYTES = 12               # (u64 doc id, f32 score) per returned result

WORKLOADS = {
    # name: (docs, terms, batch per step, what)
    "C2": (1_000_000, 100_000, 1024, "single-term BM25 top-%d"),
    "C3": (10_000_000, 1_000_000, 1024, "5-term AND/OR BM25 top-%d"),
    "C4": (10_000_000, 1_000_000, 1024, "one misspelled term per query (Levenshtein d<=2 over the BK-tree), BM25 top-%d"),
    "C5": (50_000_000, 2_000_000, 8192, "mixed: 75%% 5-term AND/OR + 25%% with one fuzzy token, BM25 top-%d"),
}


class BenchOut(C.Structure):
    _fields_ = [("seconds", C.c_double), ("results", C.c_uint64),
                ("checksum", C.c_uint64), ("failed", C.c_uint64)]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)   # (60 x 1.3 ms: a host hiccup of a few ms no longer moves the line by 10 %)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--plan-cache", action="store_true",
                    help="measure `value` with the library's plan cache on (default: off, see plan_cache_qps)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="C3")
    ap.add_argument("--docs", type=int, default=None)
    ap.add_argument("--terms", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--limit", type=int, default=10)
    ap.add_argument("--depth", type=int, default=0,
                    help="batches in flight in the timed loop (default: 3; 4 for limits above 64, whose "
                         "batches end in milliseconds of heap replay)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--min-seconds", type=float, default=1.0,
                    help="repeat the timed loop of --steps steps until this much time has been measured; the line "
                         "reports the median loop (0: one loop)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0,
                    help="budget of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--np", type=int, default=0,
                    help="worker processes of the CPU-NP baseline (0 = every core of the host share)")
    ap.add_argument("--rotate", type=int, default=4,
                    help="seed-distinct batches rotated through the timed loop")
    ap.add_argument("--docshard", type=int, default=0,
                    help="N4: cut the collection into S doc shards (all on this GPU, a stream set each) and "
                         "run the batch doc-sharded (nxs_docshard_search_batch)")
    ap.add_argument("--sparse-ids", action="store_true",
                    help="corpus with sparse random u64 doc ids (SURVEY 8d: exercises the ordinal map)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the latency / fuzzy / device-resident side measurements")
    ap.add_argument("--workdir", default=None)
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--cpu-np-worker", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()
    d, t, b, _ = WORKLOADS[a.workload]
    a.docs = a.docs or d
    a.terms = a.terms or t
    a.batch = a.batch or b
    return a


def pick_workdir(args, need_bytes):
    if args.workdir:
        return args.workdir
    for base in ("/dev/shm", tempfile.gettempdir()):
        try:
            st = os.statvfs(base)
            if st.f_bavail * st.f_frsize > need_bytes * 1.3:
                return os.path.join(base, "nxs_bench_%d_%d_%d_%d%s" % (
                    os.getuid(), args.docs, args.terms, args.seed, "_sp" if args.sparse_ids else ""))
        except OSError:
            pass
    return os.path.join(tempfile.gettempdir(), "nxs_bench_%d" % os.getuid())


def make_queries(args, terms, n, corpus, variant=0):
    """The batch of workload `args.workload`, seed-stable.
    variant 0 is THE batch (parity sample, CPU baseline); variants 1.. are the
    same distribution from other seeds, rotated through the timed loop."""
    w = args.workload
    if w == "C2":
        return corpus.queries_single(terms, n, seed=3 + 100 * variant)
    if w == "C3":
        return corpus.queries_bool5(terms, n, seed=3 + 100 * variant, hi=1000)
    if w == "C4":
        return corpus.queries_fuzzy(terms, n, seed=4 + 100 * variant)
    return corpus.queries_mixed(terms, n, seed=6 + 100 * variant, hi=1000)


def c_strings(qs):
    return (C.c_char_p * max(len(qs), 1))(*[q.encode() if isinstance(q, str) else q for q in qs])


def main():
    args = parse_args()
    if args.cpu_np_worker:
        return cpu_np_worker(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N > 1 with torch.distributed.run, "
                 "one rank per GPU)" % (args.gpus, world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    import nxsearch_amd as N
    from nxsearch_amd import corpus, multi

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
        info = corpus.write_corpus(work, args.docs, args.terms, seed=args.seed, sparse_ids=args.sparse_ids)
        with open(marker, "w") as f:
            json.dump(info, f)
    barrier()
    with open(marker) as f:
        info = json.load(f)
    t_gen = time.time() - t0
    terms = corpus.term_strings(args.terms, seed=args.seed)

    os.environ["NXS_GPU_DEVICE"] = str(local_rank)
    if args.docshard > 0:
        return docshard_bench(args, info, terms, corpus, work, t_gen)

    # ---- index resident in HBM (one replica per GPU) ------------------------
    t0 = time.time()
    nxs = N.Nxs(work)
    idx = nxs.open_files(info["terms"], info["dtmap"], algo="BM25")
    # The library keeps a plan cache (query string -> compiled plan, cleared at every refresh).  The timed
    # loop rotates a handful of batches, so with the cache on every query would be a hit and the front half
    # (lex, parse, resolve, compile) would drop out of the measured step: `value` is measured with the
    # cache OFF -- every step plans its 1024 strings afresh --, `plan_cache_qps` reports the same loop with
    # it on (a server's repeated head queries).
    idx.set_plan_cache(args.plan_cache)
    t_load = time.time() - t0
    L = N.lib()
    B = C.CDLL(os.path.join(N.CSRC, "libnxsbench.so"))
    B.nxs_bench_batches.restype = C.c_int
    B.nxs_bench_batches.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_char_p), C.c_size_t,
                                    C.c_uint, C.c_int, C.POINTER(BenchOut)]
    B.nxs_bench_batches_rot.restype = C.c_int
    B.nxs_bench_batches_rot.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_char_p), C.c_size_t, C.c_uint,
                                        C.c_uint, C.c_int, C.POINTER(BenchOut)]
    B.nxs_bench_singles.restype = C.c_int
    B.nxs_bench_singles.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_char_p), C.c_size_t,
                                    C.POINTER(C.c_double), C.POINTER(BenchOut)]

    # ---- the batch.  C2-C4: weak scaling, `batch` queries per GPU; C5: the
    #      8192-query batch is fixed and shards over the ranks (strong).  Every
    #      rank passes the SAME list; the library runs its contiguous slice and
    #      all-gathers the records (one RCCL collective per step, N > 1) --------
    strong = args.workload == "C5"
    n_total = args.batch if strong else args.batch * world
    queries = make_queries(args, terms, n_total, corpus)
    # the timed loop rotates `--rotate` seed-distinct batches (step s takes batch s mod R):
    # no step re-reads exactly what the previous one warmed up
    n_sets = max(1, args.rotate)
    batches = [queries] + [make_queries(args, terms, n_total, corpus, variant=v) for v in range(1, n_sets)]
    fuzzy_on = args.workload in ("C4", "C5")
    sharded, shard_err = False, None
    if world > 1 or os.environ.get("NXS_BENCH_FORCE_SHARD"):
        # (FORCE_SHARD: rehearse the sharded path -- RCCL communicator, all-gather of the
        # record blocks -- on a one-GPU box)
        ok = 1
        try:
            if os.environ.get("NXS_BENCH_FAIL_SHARD"):      # rehearse the safety net below
                raise RuntimeError("NXS_BENCH_FAIL_SHARD")
            multi.attach(nxs, idx, rank, world, dist, dev)
        except Exception as e:
            ok, shard_err = 0, "%s: %s" % (type(e).__name__, e)
        if dist is not None:
            t_ok = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
            ok = int(t_ok.item())
        sharded = ok == 1
        if sharded:
            # like the reference's worker processes (compose/nginx.conf:2) a rank materialises and walks the
            # responses of ITS slice only: O(n / world) host work per rank and step
            idx.shard_local(True)
        if not sharded:
            # Safety net: the communicator could not be built (on some rank).  The ranks still
            # split the batch by query -- each runs ITS slice on its replica -- but nothing is
            # gathered; `config.parallelism` says so.  The scaling curve is then the library's
            # without its one collective.
            sys.stderr.write("bench.py: rank %d: sharding unavailable (%s): every rank runs its own "
                             "slice, no collective\n" % (rank, shard_err))
            try:
                idx.shard(0, 1, None)                       # detach if this rank did attach
            except Exception:
                pass
            lo, hi = multi.shard_slice(n_total, rank, world)
            queries = queries[lo:hi]
            batches = [b[lo:hi] for b in batches]
    qarr = c_strings(queries)
    qarr_all = c_strings([q for b in batches for q in b])
    params = N._make_params(args.limit, "BM25", fuzzy_on)

    depth = args.depth if args.depth > 0 else (4 if args.limit > 64 else 3)

    def run(steps):
        out = BenchOut()
        if B.nxs_bench_batches_rot(idx._h, params, qarr_all, len(queries), n_sets, steps, depth, C.byref(out)) != 0:
            raise N.NxsError(*nxs.error())
        return out

    # measured HBM read rate of this device (roofline denominator), before the
    # timed region: it also brings the memory clocks up
    measured = L.nxsgpu_hbm_read_gbs(idx.device, 5)
    run(args.warmup)
    idx.set_profiling(True)
    idx.profile(reset=True)
    idx.host_profile()

    own_loops = []

    def timed_loop():
        """EXACTLY --steps steps between barrier + synchronize on both sides; the MAX over the ranks."""
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = run(args.steps)
        barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        own_loops.append(dt)
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return r, dt

    # A loop of K steps may last a few dozen milliseconds (the driver's `--steps 20` at C3: 20 ms) -- one host
    # hiccup moves such a figure by 10 %.  So the loop of EXACTLY K steps is REPEATED until at least
    # --min-seconds have been timed (every rank takes the same count from the all-reduced first loop), and the
    # line reports the MEDIAN loop: ms_per_step x steps is what that one loop took, `repeat_ms_per_step`
    # lists them all.
    res, first = timed_loop()
    loops = [first]
    n_loops = 1 if first <= 0 else max(1, min(400, int(math.ceil(args.min_seconds / first))))
    n_loops = max(n_loops, 1 + int(os.environ.get("NXS_BENCH_REPEATS", "0")))
    for _ in range(n_loops - 1):
        r2, dt = timed_loop()
        loops.append(dt)
        res.results += r2.results
        res.failed += r2.failed
    prof = idx.profile(reset=True)
    host_prof = idx.host_profile()
    idx.set_profiling(False)
    elapsed = sorted(loops)[(len(loops) - 1) // 2]          # the median loop (the lower one of an even count)
    # N > 1: evidence that RCCL saw N ranks -- its own count of the communicator, every rank's own loop
    # time (the line's figure is their MAX), what the all-gathers carried
    shard_ev = None
    if sharded:
        si = idx.shard_info()
        mine = 1e3 * sorted(own_loops)[(len(own_loops) - 1) // 2] / args.steps
        rank_ms = [round(mine, 4)]
        if dist is not None:
            tl = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
            dist.all_gather(tl, torch.tensor([mine], dtype=torch.float64, device=dev))
            rank_ms = [round(float(x.item()), 4) for x in tl]
        gathers = max(si["allgathers"], 1)
        hp = [host_prof.get("begin_ms", 0.0) + host_prof.get("end_ms", 0.0)]
        if dist is not None:
            tl = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
            dist.all_gather(tl, torch.tensor(hp, dtype=torch.float64, device=dev))
            hp = [float(x.item()) for x in tl]
        shard_ev = {"rccl_ranks": si["rccl_ranks"], "library_world": si["world"], "rank_ms_per_step": rank_ms,
                    "rank_host_ms_per_step": [round(x, 4) for x in hp], "responses": "own slice per rank (nxs_index_shard_local)",
                    "allgathers": si["allgathers"],
                    "allgather_bytes_per_rank": si["bytes_contributed"] // gathers,
                    "allgather_bytes_received_per_step": si["bytes_contributed"] // gathers * si["world"]}
    repeats = [round(1e3 * dt / args.steps, 4) for dt in loops]
    res_results_per_loop = res.results / len(loops)
    res_failed = res.failed

    total_q = n_total * args.steps
    qps = total_q / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    # ---- roofline of the dominant kernels (the scan launches) on this rank ----
    launches = max(prof["launches"], 1)
    my_share = 1.0      # (sharded: a rank walks the responses of its own slice only -- nxs_index_shard_local)
    matched = res_results_per_loop / max(args.steps, 1) * my_share
    alg_bytes = prof["postings"] * POSTING_BYTES / launches + matched * RESULT_BYTES
    # first scan launch -> every class scanned and replayed (classes overlap: the
    # sparse + dense OR class runs on a stream of its own beside the others)
    scan_ms = (prof["scan_ms"] + prof["replay_ms"]) / launches
    achieved = alg_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    traffic, traffic_src, traffic_per_kernel = pmc_traffic(args, world)
    # One scan launch per query class and step, each timed with HIP events on the stream it is
    # launched on (nxsgpu_profile_t::cls_*).  Per class: ms = the events' span per launch; alg_bytes =
    # 8 B x the class's postings (what the reference's loops touch) -- for classes that SKIP
    # postings (k_scanr / k_scanq: intersect first; k_cold + k_scanm<.., DROP>: dense lists not read) that
    # is work done, not bytes moved, so it is reported as a rate (work_equiv_gbs), never as a fraction;
    # moved_bytes = the kernel's FETCH_SIZE x factor + WRITE_SIZE from the committed rocprofv3 --pmc
    # summary of this command (traffic_source; not collected inside this run), frac_moved = moved / ms / peak.
    mode = 0 if args.limit <= 64 else 3
    per_kernel = []
    for c in prof.get("classes", []):
        if not c["launches"] or c["ms"] <= 0:
            continue
        names = class_kernels(c["key"], mode)
        ms = c["ms"] / c["launches"]
        algb = c["postings"] * POSTING_BYTES / c["launches"]
        moved = sum(traffic_per_kernel.get(n, 0) for n in names) if all(n in traffic_per_kernel for n in names) else None
        per_kernel.append({"kernel": "+".join(names), "queries": int(c["queries"] / c["launches"]),
                           "ms": round(ms, 4), "alg_bytes": int(algb),
                           "work_equiv_gbs": round(algb / (ms * 1e-3) / 1e9, 1),
                           "moved_bytes": moved,
                           "moved_gbs": round(moved / (ms * 1e-3) / 1e9, 1) if moved else None,
                           "frac_moved": round(moved / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if moved else None})
    per_kernel.sort(key=lambda e: -e["ms"])
    # the dominant kernel: the class that moves the most bytes (classes overlap on two streams, so a
    # latency-bound class with few wavefronts can span longer than the one that keeps the chip busy);
    # without a PMC summary: the longest one
    with_moved = [e for e in per_kernel if e["moved_bytes"]]
    # (an entry without bytes of its own -- a class's top ranges sent ahead: counted with the class -- is never the dominant one)
    with_bytes = [e for e in per_kernel if e["alg_bytes"] > 0]
    dom = max(with_moved, key=lambda e: e["moved_bytes"]) if with_moved else (with_bytes[0] if with_bytes else (per_kernel[0] if per_kernel else None))
    # the dominant kernel's own roofline point.  If it streams every posting it is charged for
    # (k_scanm, k_scan1) algorithmic == moved and the fraction is a bandwidth fraction; a kernel that
    # skips postings gets its fraction from the bytes it MOVED (never above 1).
    if dom:
        dom_alg = dom["work_equiv_gbs"]
        basis = "algorithmic"
        dom_ach = dom_alg
        if dom["moved_gbs"] is not None and (dom_alg > HBM_PEAK_GBS or dom["moved_bytes"] < 0.8 * dom["alg_bytes"]):
            dom_ach, basis = dom["moved_gbs"], "moved"
        elif dom_alg > HBM_PEAK_GBS:
            dom_ach, basis = None, "work-equivalent only (the kernel skips postings; no PMC summary for it)"
    roofline = {"bound": "hbm", "kernel": dom["kernel"] if dom else None,
                "achieved": round(dom_ach, 1) if dom and dom_ach is not None else None,
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(dom_ach / HBM_PEAK_GBS, 4) if dom and dom_ach is not None else None,
                "achieved_basis": basis if dom else None,
                "kernel_ms": dom["ms"] if dom else None,
                "peak_measured": round(measured, 1),
                "frac_measured": round(dom_ach / measured, 4) if dom and dom_ach is not None and measured > 0 else None,
                "traffic": dom["moved_bytes"] if dom else None, "traffic_source": traffic_src,
                "per_kernel": per_kernel,
                # the whole step: all classes, first scan launch -> last replay (classes overlap on two
                # streams); a rate of work done -- the AND half and the dense-OR class are charged for
                # postings they never read -- NOT a bandwidth fraction
                "step": {"alg_bytes_per_launch": int(alg_bytes), "span_ms": round(scan_ms, 4),
                         "work_equiv_gbs": round(achieved, 1),
                         "moved_bytes_per_launch": traffic,
                         "moved_gbs": round(traffic / (scan_ms * 1e-3) / 1e9, 1) if traffic and scan_ms > 0 else None,
                         "frac_moved": round(traffic / (scan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic and scan_ms > 0 else None,
                         "of_which_after_last_scan_ms": round(prof["replay_ms"] / launches, 4)}}

    what = WORKLOADS[args.workload][3] % args.limit
    out = {
        "metric": "queries/sec (10M-doc BM25 top-10, 5-term AND/OR, batch 1024)" if args.workload == "C3"
                  else "queries/sec (%s)" % what,
        "value": round(qps, 1), "unit": "queries/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic", "depth": depth,
        "config": {"workload": "%s: %d docs / %d terms Zipf, %s, batch %d %s" % (
                       args.workload, args.docs, args.terms, what, args.batch,
                       "in all (sharded by query)" if strong else "per GPU"),
                   "postings": info["postings"],
                   "boundary": "nxs_index_search_batch_begin/_end: query strings in, nxs_resp_t out, "
                               "%d batches in flight" % depth,
                   "batches_rotated": n_sets, "doc_ids": "sparse random u64" if args.sparse_ids else "1..D",
                   "parallelism": ("query-sharded x%d, index replicated, one RCCL all-gather of "
                                   "record blocks per step" % world) if (world > 1 and sharded) else
                                  ("query-sharded x%d, index replicated, NO collective (the RCCL communicator "
                                   "could not be built: %s)" % (world, shard_err)) if world > 1 else
                                  "one GPU, no collective",
                   **({"sharding": shard_ev} if shard_ev else {})},
        "roofline": roofline,
        "host_ms_per_step": host_prof,
        "timed_loops": len(loops), "repeat_ms_per_step": repeats,
        "results_per_step": int(res_results_per_loop // max(args.steps, 1)),
        "failed_queries": int(res_failed),
        "setup_s": {"corpus": round(t_gen, 1), "index_load": round(t_load, 1)},
    }

    if rank == 0 and world == 1:
        # The classes of a step overlap on three streams, so a class's event span above includes the time it
        # shares the chip with the others.  The same rotated loop once more with every class on ONE stream, one
        # after the other (nothing beside the scan stream, nothing sent ahead): each kernel's own duration, and
        # the fraction of the HBM peak its moved bytes make of THAT.
        if not args.no_extras or os.environ.get("NXS_BENCH_SERIAL"):
            serial_env = {"NXS_GPU_DROP_NOSIDE": "1", "NXS_GPU_AND_NOEARLY": "1", "NXS_GPU_DROP_NOEARLY": "1",
                          "NXS_GPU_DROP_SPLIT": "0", "NXS_GPU_REPLAY_JOIN": "1"}
            os.environ.update(serial_env)
            idx.reconfigure()
            run(2)
            idx.set_profiling(True)
            idx.profile(reset=True)
            run(8)
            sprof = idx.profile(reset=True)
            idx.set_profiling(False)
            for kk in serial_env:
                del os.environ[kk]
            idx.reconfigure()
            ser = []
            for c in sprof.get("classes", []):
                if not c["launches"] or c["ms"] <= 0:
                    continue
                names = class_kernels(c["key"], mode)
                ms = c["ms"] / c["launches"]
                moved = sum(traffic_per_kernel.get(n, 0) for n in names) if all(n in traffic_per_kernel for n in names) else None
                ser.append({"kernel": "+".join(names), "ms": round(ms, 4),
                            "moved_gbs": round(moved / (ms * 1e-3) / 1e9, 1) if moved else None,
                            "frac_moved": round(moved / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if moved else None})
            ser.sort(key=lambda e: -e["ms"])
            out["roofline"]["per_kernel_serial"] = ser
            out["roofline"]["per_kernel_serial_what"] = ("the same loop with every class on one stream, one after the other "
                                                         "(NXS_GPU_DROP_NOSIDE, NXS_GPU_AND_NOEARLY, NXS_GPU_DROP_NOEARLY, "
                                                         "NXS_GPU_DROP_SPLIT=0): each kernel alone on the chip")
        if not args.no_extras:
            out.update(side_measurements(args, nxs, idx, B, terms, queries, qarr, torch, dev))
        if not args.no_extras:
            # N1 on a private copy of the two index files, with an index of its own: the measurement
            # modifies them like an indexer would, and it runs BEFORE the CPU baseline (whose 256
            # forked workers leave the parent's first large device allocation 10x slower)
            rwork = os.path.join(work, "refresh_copy")
            try:
                os.makedirs(rwork, exist_ok=True)
                rinfo = dict(info)
                for k in ("terms", "dtmap"):
                    rinfo[k] = os.path.join(rwork, os.path.basename(info[k]))
                    shutil.copyfile(info[k], rinfo[k])
                rnxs = N.Nxs(rwork)
                ridx = rnxs.open_files(rinfo["terms"], rinfo["dtmap"], algo="BM25")
                try:
                    out["refresh"] = refresh_measurements(args, ridx, rinfo, terms)
                finally:
                    ridx.close()
                    rnxs.close()
            except Exception as e:        # a side measurement never costs the run
                out["refresh"] = {"error": "%s: %s" % (type(e).__name__, e)}
            shutil.rmtree(rwork, ignore_errors=True)
        if args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(args, info, queries, idx, fuzzy_on, work)
    barrier()
    L.nxs_params_release(params)
    idx.close()
    nxs.close()
    if rank == 0:
        print(json.dumps(out), flush=True)
        if not args.keep and not args.workdir:
            shutil.rmtree(work, ignore_errors=True)
    if dist is not None:
        dist.destroy_process_group()


def docshard_bench(args, info, terms, corpus, work, t_gen):
    """N4: the collection cut into S doc shards, every shard a device index of its own
    (here all on one GPU, each with its own streams), a batch runs on all of them at once
    and the shards' accepted candidates are merged by one more exact heap replay."""
    import nxsearch_amd as N
    S = args.docshard
    t0 = time.time()
    nxs = N.Nxs(work)
    shards = [nxs.open_shard(info["terms"], info["dtmap"], s, S) for s in range(S)]
    t_load = time.time() - t0
    B = C.CDLL(os.path.join(N.CSRC, "libnxsbench.so"))
    B.nxs_bench_docshard.restype = C.c_int
    B.nxs_bench_docshard.argtypes = [C.POINTER(C.c_void_p), C.c_uint, C.c_void_p, C.POINTER(C.c_char_p), C.c_size_t,
                                     C.c_uint, C.c_uint, C.POINTER(BenchOut)]
    n_sets = max(1, args.rotate)
    batches = [make_queries(args, terms, args.batch, corpus, variant=v) for v in range(n_sets)]
    qarr_all = c_strings([q for b in batches for q in b])
    fuzzy_on = args.workload in ("C4", "C5")
    params = N._make_params(args.limit, "BM25", fuzzy_on)
    hs = (C.c_void_p * S)(*[s._h for s in shards])
    out = BenchOut()

    def run(steps):
        if B.nxs_bench_docshard(hs, S, params, qarr_all, args.batch, n_sets, steps, C.byref(out)) != 0:
            raise N.NxsError(*nxs.error())
    run(args.warmup)
    t0 = time.perf_counter()
    run(args.steps)
    elapsed = time.perf_counter() - t0
    # parity sample against the whole-index oracle
    mism = None
    if args.cpu_seconds > 0:
        import oracle_lib as O
        oidx = O.Index(info["terms"], info["dtmap"])
        sample = batches[0][:max(4, min(24, int(args.cpu_seconds)))]
        got = nxs.docshard_search_batch(shards, sample, limit=args.limit, fuzzymatch=fuzzy_on)
        mism = sum(1 for q, g in zip(sample, got) if g != oidx.search(q, algo=O.BM25, limit=args.limit, fuzzymatch=fuzzy_on))
        oidx.close()
    what = WORKLOADS[args.workload][3] % args.limit
    res = {"metric": "queries/sec (%s), doc-sharded x%d" % (what, S),
           "value": round(args.batch * args.steps / elapsed, 1), "unit": "queries/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "%s: %d docs / %d terms Zipf, %s, batch %d, %d doc shards on one GPU" % (
                          args.workload, args.docs, args.terms, what, args.batch, S),
                      "boundary": "nxs_docshard_search_batch: query strings in, nxs_resp_t out (blocking)",
                      "parallelism": "doc-sharded x%d (N4): shards scanned concurrently, candidates merged by an "
                                     "exact heap replay; one process" % S},
           "roofline": None, "results_per_step": int(out.results // max(args.steps, 1)),
           "failed_queries": int(out.failed), "parity_mismatches_vs_whole_index_oracle": mism,
           "setup_s": {"corpus": round(t_gen, 1), "index_load": round(t_load, 1)}}
    N.lib().nxs_params_release(params)
    for s in shards:
        s.close()
    nxs.close()
    print(json.dumps(res), flush=True)
    if not args.keep and not args.workdir:
        shutil.rmtree(work, ignore_errors=True)


def source_hash():
    """sha256 over the HIP sources + headers the library is built from (tools/pmc_summary.py stamps its
    summaries with it)."""
    import hashlib
    root = os.path.join(ROOT, "nxsearch_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(root)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(args, world):
    """HBM bytes per step (all scan launches) from the newest committed rocprofv3
    --pmc summary whose workload matches this run (collected by
    tools/profile_round.sh on the same command) AND whose kernels are this tree's (the summary's
    `source_hash`), and the file it came from; (None, reason, {}) otherwise -- the counters are not
    collected inside this run, and figures of other kernels are not printed as this run's."""
    best = (None, None, {})
    head = source_hash()
    pdir = os.path.join(ROOT, "profiles")
    try:
        names = sorted(f for f in os.listdir(pdir) if f.endswith("pmc_summary.json"))
    except OSError:
        return best
    for name in names:
        try:
            with open(os.path.join(pdir, name)) as f:
                p = json.load(f)
            w = p["workload"]
            if (w["docs"], w["terms"], w["batch"], w["limit"]) == \
                    (args.docs, args.terms, args.batch, args.limit) and world == 1 and \
                    w.get("name", "C3") == args.workload:
                per_k = {}
                fac = float(p.get("fetch_size_factor", 2.0))
                for kname, ctr in p.get("counters", {}).items():
                    if "FETCH_SIZE" in ctr:
                        per_k[kname] = int(ctr["FETCH_SIZE"]["sum_per_step"] * 1024 * fac +
                                           ctr.get("WRITE_SIZE", {}).get("sum_per_step", 0.0) * 1024)
                if p.get("source_hash") != head:
                    best = (None, "profiles/%s is stale (kernel sources %s, this tree %s): traffic not reported" % (
                        name, p.get("source_hash", "unstamped"), head), {})
                    continue
                best = (int(p["hbm_bytes_per_launch"]), "profiles/" + name, per_k)
        except (OSError, KeyError, ValueError):
            pass
    return best


def class_kernels(key, mode):
    """Kernel names (as rocprofv3 prints them) of a query class; key = kind << 8 | shape << 4 | token bucket
    (nxsgpu_profile_t::cls_key), mode = 0 (limit <= 64) / 3 (MODE_BIG)."""
    ahead = (key >> 7) & 1          # a launch of the class's top doc ranges, sent ahead of the rest
    key &= ~0x80
    kind, shape, b = key >> 8, (key >> 4) & 15, key & 15
    if ahead:
        return [n + " [top ranges ahead]" for n in class_kernels(key, mode)]
    b3 = 3 if b in (2, 3) else b
    tf = lambda x: "true" if x else "false"
    if kind == 1:
        return ["k_scan1<%d>" % mode] if b == 1 else ["k_scan8<%d, %d, %d>" % (mode, b, shape)]
    if kind == 3:
        return ["k_scanr<%d, %d, %s>" % (mode, b, tf(shape == 1))]
    if kind == 4:
        return ["k_scanm<%d, %s, false>" % (b3, tf(shape != 1))] if mode == 0 else ["k_scan8<%d, %d, %d>" % (mode, b, 1 if shape == 1 else 0)]
    if kind == 5:
        return ["k_cold<%d, false>" % b3, "k_scanm<%d, false, true>" % b3] if mode == 0 else ["k_scan8<%d, %d, 1>" % (mode, b)]
    if kind == 9:
        return ["k_cold<%d, false>" % b3, "k_scans<%d, false, true>" % b3] if mode == 0 else ["k_scan8<%d, %d, 1>" % (mode, b)]
    if kind == 8:
        return ["k_scans<%d, %s, false>" % (b3, tf(shape != 1))] if mode == 0 else ["k_scan8<%d, %d, %d>" % (mode, b, 1 if shape == 1 else 0)]
    if kind == 6:
        return ["k_scanb<%d, %s, false>" % (b3 if b3 != 8 else 5, tf(shape != 1))]
    if kind == 7:
        return ["k_scanq<%d, %s>" % (b, tf(mode != 0))]
    return ["k_scan<%d, false>" % mode]


def side_measurements(args, nxs, idx, B, terms, queries, qarr, torch, dev):
    """Device-resident rate, blocking-API rate, p50 latency, TF-IDF pass, the
    default-limit path, and the fuzzy path (C4) with its own roofline object."""
    import nxsearch_amd as N
    from nxsearch_amd import corpus
    L = N.lib()
    res = {}
    k = args.limit
    fuzzy_on = args.workload in ("C4", "C5")

    # (1) pre-resolved plans, results left in HBM (round 1's headline): what the
    #     device does once the host front half and the response objects are gone
    plans, errs = idx.plan_batch(queries, limit=k, algo="BM25", fuzzymatch=fuzzy_on)
    nq = len(queries)
    bufs = [(torch.zeros((nq, k), dtype=torch.int64, device=dev),
             torch.zeros((nq, k), dtype=torch.float32, device=dev),
             torch.zeros((nq,), dtype=torch.int32, device=dev)) for _ in range(2)]

    def dev_run(n):
        for i in range(n):
            o = bufs[i % 2]
            idx.search_dev_begin(plans, nq, k, N.BM25, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
            if i:
                idx.search_dev_end()
        if n:
            idx.search_dev_end()
    dev_run(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dev_run(args.steps)
    torch.cuda.synchronize()
    res["device_resident_qps"] = round(nq * args.steps / (time.perf_counter() - t0), 1)

    # (2) the blocking call (no batch overlap)
    o = BenchOut()
    p = N._make_params(k, "BM25", fuzzy_on)
    B.nxs_bench_batches(idx._h, p, qarr, nq, 2, 1, C.byref(o))
    B.nxs_bench_batches(idx._h, p, qarr, nq, 10, 1, C.byref(o))
    res["blocking_qps"] = round(nq * 10 / o.seconds, 1)
    # (2b) the pipelined loop on ONE repeated batch (what rounds 1-2 reported as `value`,
    #      before the timed loop rotated seed-distinct batches): for comparison only
    B.nxs_bench_batches(idx._h, p, qarr, nq, 3, 2, C.byref(o))
    B.nxs_bench_batches(idx._h, p, qarr, nq, args.steps, 2, C.byref(o))
    res["repeated_batch_qps"] = round(nq * args.steps / o.seconds, 1)
    # (2c) the same loop with the plan cache on: every query string has been seen before
    if not args.plan_cache:
        idx.set_plan_cache(True)
        B.nxs_bench_batches(idx._h, p, qarr, nq, 3, 2, C.byref(o))
        B.nxs_bench_batches(idx._h, p, qarr, nq, args.steps, 2, C.byref(o))
        res["plan_cache_qps"] = round(nq * args.steps / o.seconds, 1)
        idx.set_plan_cache(False)

    # (3) single-query latency: nxs_index_search(), one term of rank uniform in
    #     [10, 10^4], top-k, n = 1000, timed inside the C consumer
    singles = corpus.queries_single(terms, 1100, seed=5)
    sarr = c_strings(singles)
    lat = (C.c_double * len(singles))()
    pl = N._make_params(k, "BM25", False)
    B.nxs_bench_singles(idx._h, pl, sarr, 100, lat, C.byref(o))
    B.nxs_bench_singles(idx._h, pl, sarr, len(singles), lat, C.byref(o))
    ls = sorted(lat[100:])
    res["latency"] = {"p50_ms": round(1e-3 * ls[len(ls) // 2], 4),
                      "p95_ms": round(1e-3 * ls[int(len(ls) * 0.95)], 4),
                      "p99_ms": round(1e-3 * ls[int(len(ls) * 0.99)], 4),
                      "what": "nxs_index_search(), single-term BM25 top-%d, n=%d" % (k, len(ls))}
    L.nxs_params_release(pl)

    # (4) TF-IDF pass over the same batch
    pt = N._make_params(k, "TF-IDF", fuzzy_on)
    B.nxs_bench_batches(idx._h, pt, qarr, nq, 2, 2, C.byref(o))
    idx.host_profile()
    B.nxs_bench_batches(idx._h, pt, qarr, nq, 10, 2, C.byref(o))
    res["tfidf"] = {"queries_per_s": round(nq * 10 / o.seconds, 1), "failed": int(o.failed),
                    "exact_requeries_per_step": idx.host_profile()["exact_requeries"] / 10.0}
    L.nxs_params_release(pt)

    # (5) default limit: params == NULL, what the reference's own callers pass
    #     (src/utils/benchmark.c:204) => limit 1000 (nxs_impl.h:39).  The WHOLE batch,
    #     pipelined like the headline (candidate filter on a histogram threshold,
    #     MODE_BIG; heap replayed in LDS by the whole wavefront), then the blocking call
    idx.host_profile()
    B.nxs_bench_batches(idx._h, None, qarr, nq, 4, 4, C.byref(o))
    B.nxs_bench_batches(idx._h, None, qarr, nq, 48, 4, C.byref(o))
    hp = idx.host_profile()
    res["default_limit"] = {"queries_per_s": round(nq * 48 / o.seconds, 1), "limit": 1000,
                            "ms_per_step": round(1e3 * o.seconds / 48, 3),
                            "results_per_query": round(o.results / (48.0 * nq), 1),
                            "exact_requeries_per_step": hp["exact_requeries"] / 52.0,
                            "what": "the whole batch (%d queries), params == NULL, "
                                    "nxs_index_search_batch_begin/_end, four batches in flight" % nq}
    B.nxs_bench_batches(idx._h, None, qarr, nq, 8, 2, C.byref(o))
    res["default_limit"]["two_in_flight_queries_per_s"] = round(nq * 8 / o.seconds, 1)
    B.nxs_bench_batches(idx._h, None, qarr, nq, 4, 1, C.byref(o))
    res["default_limit"]["blocking_queries_per_s"] = round(nq * 4 / o.seconds, 1)
    L.nxs_params_release(p)
    # ... and the latency of the reference CLI's actual call: nxs_index_search(idx, NULL, ...)
    B.nxs_bench_singles(idx._h, None, sarr, 100, lat, C.byref(o))
    B.nxs_bench_singles(idx._h, None, sarr, len(singles), lat, C.byref(o))
    ls = sorted(lat[100:])
    res["latency_default"] = {"p50_ms": round(1e-3 * ls[len(ls) // 2], 4),
                              "p95_ms": round(1e-3 * ls[int(len(ls) * 0.95)], 4),
                              "p99_ms": round(1e-3 * ls[int(len(ls) * 0.99)], 4),
                              "results_per_query": round(o.results / float(len(singles)), 1),
                              "what": "nxs_index_search(idx, NULL, ...): single-term, default limit 1000, n=%d" % len(ls)}

    # (6) fuzzy: C4 = 1024 tokens, Levenshtein d<=2 over the BK-tree of all terms
    toks = corpus.queries_fuzzy(terms, 1024, seed=4)
    idx.fuzzy(toks[:64])
    ids_ref, vis = idx.fuzzy(toks, want_visited=True)        # the reference's visit set (no pruning)
    idx.set_profiling(True)
    idx.profile(reset=True)
    best_dt = None
    for _ in range(3):
        t0 = time.perf_counter()
        ids = idx.fuzzy(toks)
        dt = time.perf_counter() - t0
        best_dt = dt if best_dt is None else min(best_dt, dt)
    pr = idx.profile(reset=True)
    idx.set_profiling(False)
    evals, pairs, ms = pr["fuzzy_visits"] / 3.0, pr["fuzzy_pairs"] / 3.0, pr["fuzzy_ms"] / 3.0
    mean_len = sum(len(t) for t in terms[:100000]) / 100000.0
    tok_len = sum(len(t) for t in toks) / float(len(toks))
    # candidates = the nodes the reference's bktree_search visits for these
    # tokens (counted by the frontier search, which reproduces them one for one);
    # algorithmic bytes = 16 B node record + len(term) B per candidate.  The match-first
    # search finds the same winners with far fewer distance evaluations (`levels`: pairs
    # screened, survivors of the screen, d = 2 matches walked to the root).
    ref_vis = float(sum(vis))
    f_ms, d_ms, c_ms = pr["fuzzy_filter_ms"] / 3.0, pr["fuzzy_dist_ms"] / 3.0, pr["fuzzy_chain_ms"] / 3.0
    checked = pr["fuzzy_checked"] / 3.0
    # Bounds in WORK DONE (round 2's object divided the reference's visits by this time: frac > 1).
    # k_fz_filter: 6 vector instructions per compared (token, term) pair, one pair per lane:
    #   wave-instructions = pairs x 6 / 64 against the chip's VALU issue rate
    #   (256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction, MI355X_MICROARCH.md);
    # k_fz_dist / k_fz_chain: dependent gathers out of L2 (node record + Peq words per
    #   evaluation; parent / slot + node per ancestor): a rate, no streaming peak applies.
    valu_peak = 256 * 4 * 2.4e9 / 2
    filt_rate = checked * 6 / 64 / (f_ms * 1e-3) if f_ms > 0 else 0.0
    res["fuzzy"] = {"tokens_per_s": round(len(toks) / best_dt, 1),
                    "reference_visits_equiv_per_s": round(ref_vis / (ms * 1e-3), 1) if ms > 0 else None,
                    "distance_evaluations_per_s": round(evals / (ms * 1e-3), 1) if ms > 0 else None,
                    "reference_visits": int(ref_vis),
                    "distance_evaluations": int(evals), "queue_items": int(pairs),
                    "device_ms": round(ms, 3), "resolved": sum(1 for i in ids if i),
                    "same_as_unpruned": ids == ids_ref,
                    "levels": [int(x / 3) for x in pr["fuzzy_level"] if x],
                    "bounds": {
                        "k_fz_filter": {"bound": "valu_issue", "ms": round(f_ms, 4), "pairs_compared": int(checked),
                                        "valu_instr_per_pair": 6, "achieved": round(filt_rate / 1e9, 2),
                                        "peak": round(valu_peak / 1e9, 1), "unit": "G wave-instr/s",
                                        "frac": round(filt_rate / valu_peak, 4)},
                        "k_fz_dist": {"bound": "l2_gather_latency", "ms": round(d_ms, 4), "gathers": int(evals),
                                      "bytes_per_gather": 104,
                                      "achieved": round(evals / (d_ms * 1e-3) / 1e9, 3) if d_ms > 0 else None,
                                      "unit": "G gathers/s",
                                      "gb_per_s_from_l2": round(evals * 104 / (d_ms * 1e-3) / 1e9, 1) if d_ms > 0 else None},
                        "k_fz_chain": {"bound": "l2_gather_latency (dependent)", "ms": round(c_ms, 4),
                                       "matches_walked": int(pr["fuzzy_level"][2] / 3)}},
                    "moved_bytes_estimate": int(len(terms) * 8 * 8 + pairs * 16 + evals * 104),
                    "note": "tokens_per_s is the headline; reference_visits_equiv_per_s restates it in the "
                            "reference's unit (nodes ITS bktree_search would visit for these tokens) -- work the "
                            "match-first search does not do, so not a rate of this device",
                    "what": "C4: %d tokens, d<=2 over a %d-term BK-tree" % (len(toks), args.terms)}
    return res


def refresh_measurements(args, idx, info, terms):
    """N1: what one appended / removed document costs the NEXT search (the
    reference re-syncs before every search, search.c:309-312).  The corpus files
    are modified in place like an indexer process would: block first, header
    counters and data_len last.  The files are not pristine after it: the caller hands in a private copy."""
    import struct
    import ctypes as C
    import nxsearch_amd as N
    L = N.lib()
    stats = (C.c_uint64 * 2)()
    L.nxs_index_refresh_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    dpath, tpath = info["dtmap"], info["terms"]

    def hdr(f):
        f.seek(0)
        h = f.read(32)
        data_len, tokens, docs = struct.unpack(">QQI", h[8:28])
        return data_len, tokens, docs

    def append_block(doc_id, pairs):
        with open(dpath, "r+b") as f:
            data_len, tokens, docs = hdr(f)
            blk = struct.pack(">QII", doc_id, sum(c for _, c in pairs), len(pairs))
            blk += b"".join(struct.pack(">II", t, c) for t, c in sorted(pairs))
            f.seek(32 + data_len)
            f.write(blk)
            f.flush()
            f.seek(8)
            f.write(struct.pack(">QQI", data_len + len(blk), tokens + sum(c for _, c in pairs), docs + 1))
        return 32 + data_len

    def remove_doc(block_off, doc_id, doc_len):
        with open(dpath, "r+b") as f:
            data_len, tokens, docs = hdr(f)
            f.seek(block_off)
            f.write(struct.pack(">Q", 0))
            f.seek(32 + data_len)
            f.write(struct.pack(">QII", doc_id, 0, 0))
            f.flush()
            f.seek(8)
            f.write(struct.pack(">QQI", data_len + 16, tokens - doc_len, docs - 1))

    def append_term(word):
        with open(tpath, "r+b") as f:
            f.seek(8)
            data_len = struct.unpack(">I", f.read(4))[0]
            blk = struct.pack(">H", len(word)) + word + b"\0"
            blk += b"\0" * (-len(blk) % 8) + struct.pack(">Q", 2)
            f.seek(16 + data_len)
            f.write(blk)
            f.flush()
            f.seek(8)
            f.write(struct.pack(">I", data_len + len(blk)))

    q = terms[99].decode()
    def timed_search(query=q, fuzzymatch=False):
        t0 = time.perf_counter()
        r = idx.search(query, limit=args.limit, fuzzymatch=fuzzymatch)
        return 1e3 * (time.perf_counter() - t0), r
    base_ms = min(timed_search()[0] for _ in range(5))
    out = {"plain_search_ms": round(base_ms, 3)}
    new_id = args.docs + 10
    pairs = [(t, 1 + (t % 3)) for t in (5, 17, 100, 101, 2000, 31337 % args.terms + 1, 7, 9, 11, 13)]
    off = append_block(new_id, pairs)
    ms, r = timed_search(terms[99].decode())
    out["append_1_doc_ms"] = round(ms, 3)
    ms2, r2 = timed_search(terms[100].decode())           # term 101 holds the new doc
    out["new_doc_found"] = any(d == new_id for d, _ in idx.search(terms[100].decode(), limit=64, fuzzymatch=False)) \
        or len(r2) > 0
    remove_doc(off, new_id, sum(c for _, c in pairs))
    ms, _ = timed_search()
    out["remove_1_doc_ms"] = round(ms, 3)
    word = b"zzqxjkvbnm"
    append_term(word)
    append_block(new_id + 5, [(args.terms + 1, 2), (5, 1)])
    ms, r = timed_search(word.decode())
    out["append_doc_with_new_term_ms"] = round(ms, 3)
    out["new_term_found"] = [d for d, _ in r] == [new_id + 5]
    ms, r = timed_search("zzqxjkvbnn", fuzzymatch=True)      # fuzzy: BK image re-flattened lazily
    out["first_fuzzy_after_new_term_ms"] = round(ms, 3)
    out["fuzzy_resolves_new_term"] = [d for d, _ in r] == [new_id + 5]
    L.nxs_index_refresh_stats(idx._h, stats)
    out["incremental_refreshes"], out["rebuilds"] = int(stats[0]), int(stats[1])
    out["what"] = ("wall time of the nxs_index_search() that picks the change up (incremental "
                   "merge into the device CSR + all impacts recomputed), %d docs" % args.docs)
    return out


def cpu_baseline(args, info, queries, idx, fuzzy_on, work):
    """The oracle on ONE host core, on a bounded sample of the same batch
    (same corpus, first queries of the batch), next to the GPU number; then one
    worker process per core (the reference's deployment model)."""
    import oracle_lib as O
    t0 = time.time()
    oidx = O.Index(info["terms"], info["dtmap"])
    t_load = time.time() - t0
    budget = args.cpu_seconds
    n, pairs, t_used = 0, 0, 0.0
    mism = 0
    for q in queries:
        t0 = time.perf_counter()
        want = oidx.search(q, algo=O.BM25, limit=args.limit, fuzzymatch=fuzzy_on)
        t_used += time.perf_counter() - t0
        pairs += oidx.last_pairs
        got = idx.search(q, limit=args.limit, fuzzymatch=fuzzy_on)
        if got != want:
            mism += 1
        n += 1
        if t_used >= budget:
            break
    base = {"value": round(n / t_used, 3), "unit": "queries/s", "cores": 1, "kind": "port",
            "sample": "first %d queries of the batch, %.1f s, %d (doc,term) pairs scored; "
                      "oracle/nxs_oracle.c (reference algorithm restated in C)" % (n, t_used, pairs),
            "parity_mismatches": mism, "load_s": round(t_load, 1),
            "host_cpus": os.cpu_count(), "cpu_model": cpu_model()}
    oidx.close()
    # fuzzy baseline on the genuine reference bktree.c/levdist.c (oracle/_ref)
    if O.ref() is not None and not args.no_extras:
        from nxsearch_amd import corpus
        terms = corpus.term_strings(args.terms, seed=args.seed)
        t0 = time.time()
        tree = O.RefBKTree(terms)
        t_build = time.time() - t0
        toks = corpus.queries_fuzzy(terms, 1024, seed=4)
        vis, t_used, m = 0, 0.0, 0
        for t in toks:
            t0 = time.perf_counter()
            _, nv = tree.search(t.encode(), 2)
            t_used += time.perf_counter() - t0
            vis += nv
            m += 1
            if t_used >= min(budget, 10.0):
                break
        base["fuzzy"] = {"candidates_per_s": round(vis / t_used, 1), "tokens_per_s": round(m / t_used, 2),
                         "cores": 1, "kind": "reference", "tokens": m, "build_s": round(t_build, 1),
                         "sample": "reference src/algo/bktree.c + levdist.c compiled in place (oracle/_ref)"}
        tree.close()
    # one worker PROCESS per core over disjoint query slices (compose/nginx.conf:2
    # `worker_processes auto`): a child program without a GPU context, so that it
    # can fork its workers after loading the index once
    if not args.no_extras:
        try:
            share = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            share = os.cpu_count() or 1
        # one worker per core of the host share (the reference's deployment model);
        # the workers fork after the index is loaded, so they share its pages
        np_ = args.np or share
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-np-worker", "--workload", args.workload,
               "--docs", str(args.docs), "--terms", str(args.terms), "--batch", str(args.batch),
               "--limit", str(args.limit), "--seed", str(args.seed), "--np", str(np_),
               "--cpu-seconds", str(min(args.cpu_seconds, 20.0)), "--workdir", work]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            base["np"] = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:            # the baseline is a report, never a reason to lose the run
            base["np"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return base


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def cpu_np_worker(args):
    """Child program (no GPU): load the oracle index once, fork `--np` workers,
    each answers a disjoint slice of the batch for the time budget."""
    import oracle_lib as O
    from nxsearch_amd import corpus
    with open(os.path.join(args.workdir, "done")) as f:
        info = json.load(f)
    terms = corpus.term_strings(args.terms, seed=args.seed)
    queries = make_queries(args, terms, args.batch, corpus)
    fuzzy_on = args.workload in ("C4", "C5")
    t0 = time.time()
    oidx = O.Index(info["terms"], info["dtmap"])
    t_load = time.time() - t0
    P = max(1, args.np)
    pipes = []
    # same mix of query shapes for every worker: shuffle before slicing; a worker
    # that runs out of queries starts over, so that all P stay busy for the budget
    import random
    random.Random(12345).shuffle(queries)
    t_start = time.time()
    for w in range(P):
        r, wfd = os.pipe()
        pid = os.fork()
        if pid == 0:
            os.close(r)
            n, t_used = 0, 0.0
            mine = queries[w::P] or queries
            t_begin = time.perf_counter()
            while t_used < args.cpu_seconds:
                for q in mine:
                    oidx.search(q, algo=O.BM25, limit=args.limit, fuzzymatch=fuzzy_on)
                    n += 1
                    t_used = time.perf_counter() - t_begin
                    if t_used >= args.cpu_seconds:
                        break
            os.write(wfd, ("%d %.6f\n" % (n, t_used)).encode())
            os._exit(0)
        os.close(wfd)
        pipes.append((pid, r))
    rate, done = 0.0, 0
    for pid, r in pipes:
        data = b""
        while True:
            chunk = os.read(r, 256)
            if not chunk:
                break
            data += chunk
        os.close(r)
        os.waitpid(pid, 0)
        n, t_used = data.split()
        done += int(n)
        if float(t_used) > 0:
            rate += int(n) / float(t_used)
    print(json.dumps({"value": round(rate, 3), "unit": "queries/s", "cores": P, "kind": "port",
                      "queries_done": done, "wall_s": round(time.time() - t_start, 1),
                      "load_s": round(t_load, 1), "host_cpus": os.cpu_count(), "cpu_model": cpu_model(),
                      "sample": "one oracle worker process per core over disjoint slices of the batch "
                                "(index loaded once, forked), %.0f s budget each" % args.cpu_seconds}),
          flush=True)


if __name__ == "__main__":
    main()

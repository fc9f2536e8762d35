import json, csv, sys
D = sys.argv[1]
def last(fn):
    return json.loads(open(D + "/" + fn).read().strip().splitlines()[-1])
def kstats(fn):
    out = {}
    for r in csv.DictReader(open(D + "/" + fn)):
        out[r["Name"]] = (int(r["Calls"]), float(r["AverageDuration_us"]))
    return out
b = last("r3_bench.json")
ks = kstats("r3_kernel_stats.csv")
pm = json.load(open(D + "/r3_pmc_summary.json"))
rows = []
hp = b["host_ms_per_step"]
rf = b["roofline"]
rows.append(("**C3 through the C API** (`value`; strings in, `nxs_resp_t` out, two batches in flight, 4 batches rotated)",
 "**%.0fk queries/s**, %.2f ms/step over the default 20 steps.  Host per step: parse/resolve/compile %.2f ms on the worker threads, queueing %.2f ms, responses %.2f ms" % (
   b["value"] / 1e3, b["ms_per_step"], hp["plan_ms"], hp["queue_ms"], hp["resps_ms"])))
rows.append(("the batch of round 2 again (seed 3 alone): the same loop on ONE repeated batch (`repeated_batch_qps`, what rounds 1-2 reported) / plans pre-resolved, results left in HBM (`device_resident_qps`) / blocking call / TF-IDF (rotated batches)",
 "%.0fk / %.0fk / %.0fk / %.0fk queries/s" % (b.get("repeated_batch_qps", 0) / 1e3, b["device_resident_qps"] / 1e3, b["blocking_qps"] / 1e3, b["tfidf"]["queries_per_s"] / 1e3)))
rows.append(("roofline (all scan launches of a step, first launch → last replay, HIP events)",
 "**%.3f ms** per step: %.0f GB/s algorithmic = **%.1f %%** of 8 TB/s, %.0f %% of the %.2f TB/s this device reads (`peak_measured`)" % (
   rf["kernel_ms"], rf["achieved"], 100 * rf["frac"], 100 * rf["frac_measured"], rf["peak_measured"] / 1e3)))
def k(name):
    return ks.get(name, (0, 0.0))
rows.append(("rocprofv3 averages (`profiles/r3_kernel_stats.csv`; kernels run one at a time under the profiler)",
 "`k_scanm<5,false,false>` %.3f ms, `k_scanr<0,5,true>` %.3f ms, on their own stream `k_cold<5>` %.3f + `k_scanm<5,false,true>` %.3f ms, retry `k_scan8<0,5,1>` %d launches × %.3f ms, `k_replay<1>` %d × %.3f ms" % (
   k("k_scanm<5, false, false>")[1] / 1e3, k("k_scanr<0, 5, true>")[1] / 1e3, k("k_cold<5, false>")[1] / 1e3, k("k_scanm<5, false, true>")[1] / 1e3,
   k("k_scan8<0, 5, 1>")[0] // 8, k("k_scan8<0, 5, 1>")[1] / 1e3, k("k_replay<1>")[0] // 8, k("k_replay<1>")[1] / 1e3)))
c = pm["counters"]
def cs(kn, cn):
    return c.get(kn, {}).get(cn, {}).get("sum_per_step", 0.0)
kn = "k_scanm<5, false, false>"
tot_hbm = pm["hbm_bytes_per_launch"]
valu, salu, lds, vm = cs(kn, "SQ_INSTS_VALU"), cs(kn, "SQ_INSTS_SALU"), cs(kn, "SQ_INSTS_LDS"), cs(kn, "SQ_INSTS_VMEM_RD")
rows.append(("HBM traffic (PMC, calibrated ×%.4f for 8 B/lane loads)" % pm["fetch_size_factor"],
 "%.2f GB per step vs %.2f GB algorithmic (%.2f×): `k_scanm<5,false,false>` %.2f GB, `k_cold` + `k_scanm<5,false,true>` %.2f GB, `k_scanr` %.2f GB" % (
   tot_hbm / 1e9, rf["alg_bytes_per_launch"] / 1e9, tot_hbm / rf["alg_bytes_per_launch"],
   (cs(kn, "FETCH_SIZE") * 1024 * pm["fetch_size_factor"] + cs(kn, "WRITE_SIZE") * 1024) / 1e9,
   sum(cs(x, "FETCH_SIZE") * 1024 * pm["fetch_size_factor"] + cs(x, "WRITE_SIZE") * 1024 for x in ("k_cold<5, false>", "k_scanm<5, false, true>")) / 1e9,
   (cs("k_scanr<0, 5, true>", "FETCH_SIZE") * 1024 * pm["fetch_size_factor"] + cs("k_scanr<0, 5, true>", "WRITE_SIZE") * 1024) / 1e9)))
win = vm if vm else 1
rows.append(("`k_scanm<5,false,false>` per step (PMC)", "%.0f M VALU + %.0f M SALU + %.0f M LDS wave-instructions for %.2f M `global_load_dwordx2` windows = **%.0f issued instructions per 64-posting window**; its moved bytes ÷ its rocprof time = %.2f TB/s" % (
   valu / 1e6, salu / 1e6, lds / 1e6, vm / 1e6, (valu + salu + lds) / win,
   (cs(kn, "FETCH_SIZE") * 1024 * pm["fetch_size_factor"] + cs(kn, "WRITE_SIZE") * 1024) / (k(kn)[1] * 1e-6) / 1e12 if k(kn)[1] else 0)))
dl = b["default_limit"]; ld = b["latency_default"]; la = b["latency"]
rows.append(("p50 / p95 / p99 latency `nxs_index_search`, single-term top-10, 10M docs, n = 1000", "**%.3f** / %.3f / %.3f ms (round 2: 0.087 / 0.126 / 0.207)" % (la["p50_ms"], la["p95_ms"], la["p99_ms"])))
rows.append(("**default limit** (`params == NULL` ⇒ 1000): the whole batch of 1024, pipelined / blocking (`default_limit`)", "**%.0fk queries/s** (%.2f ms per batch) / %.0fk queries/s; 500 results per query on average; 0 exact re-queries (round 2: 2.9k on 64 queries)" % (
   dl["queries_per_s"] / 1e3, dl["ms_per_step"], dl["blocking_queries_per_s"] / 1e3)))
rows.append(("`latency_default`: `nxs_index_search(idx, NULL, …)`, single-term, 1000 results, n = 1000", "p50 %.2f / p95 %.2f / p99 %.2f ms (≈ k (1 + ln(df / k)) heap insertions at 0.4 µs)" % (ld["p50_ms"], ld["p95_ms"], ld["p99_ms"])))
l1 = kstats("r3_l1000_kernel_stats.csv")
rows.append(("default-limit batch under rocprofv3 (`profiles/r3_l1000_kernel_stats.csv`)", "`k_scan8<3,5,1>` (MODE_BIG, OR half) %.2f ms, `k_scanr<3,5,true>` %.2f ms, `k_replay_coop` %d launches × %.2f ms" % (
   l1.get("k_scan8<3, 5, 1>", (0, 0))[1] / 1e3, l1.get("k_scanr<3, 5, true>", (0, 0))[1] / 1e3, l1.get("k_replay_coop", (0, 0))[0] // 8, l1.get("k_replay_coop", (0, 0))[1] / 1e3)))
for tag, label in (("c2", "C2 (1M docs / 100k terms, 1024 single-term queries per step)"), ("c4", "C4 as queries (1024 misspelled single terms per step: fuzzy resolve + single-term scan)")):
    x = last("r3_%s_bench.json" % tag); p = json.load(open(D + "/r3_%s_pmc_summary.json" % tag)); kk = kstats("r3_%s_kernel_stats.csv" % tag)
    rows.append((label + " (`profiles/r3_%s_*`)" % tag, "%.2f M queries/s (%.2f ms/step; host: begin %.2f ms), kernels %.3f ms per step (`k_scan1<0>` %.3f, `k_replay<1>` %.3f ms under rocprofv3); HBM traffic %.1f MB per step vs %.1f MB algorithmic; p50 %.3f ms" % (
        x["value"] / 1e6, x["ms_per_step"], x["host_ms_per_step"]["begin_ms"], x["roofline"]["kernel_ms"], kk.get("k_scan1<0>", (0, 0))[1] / 1e3, kk.get("k_replay<1>", (0, 0))[1] / 1e3,
        p["hbm_bytes_per_launch"] / 1e6, x["roofline"]["alg_bytes_per_launch"] / 1e6, x["latency"]["p50_ms"])))
fz = b["fuzzy"]; fb = fz["bounds"]
rows.append(("fuzzy device pass, 1024 tokens over the 1M-term tree (`fuzzy`)", "**%.2f ms = %.2f M tokens/s**: `k_fz_filter` %.3f ms for %.0f M compared pairs × 6 VALU = %.0f G wave-instr/s = **%.1f %% of the chip's VALU issue rate** (1228.8 G/s); `k_fz_dist` %.3f ms for %.2f M exact distances = %.1f G gathers/s (%.1f TB/s out of L2 at 104 B each); `k_fz_chain` %.3f ms for %.0f k walked matches.  In the reference's unit (its %.1f M visits for these tokens): %.0f G visits-equivalent/s -- work the search does not do" % (
   fz["device_ms"], fz["tokens_per_s"] / 1e6, fb["k_fz_filter"]["ms"], fb["k_fz_filter"]["pairs_compared"] / 1e6, fb["k_fz_filter"]["achieved"], 100 * fb["k_fz_filter"]["frac"],
   fb["k_fz_dist"]["ms"], fb["k_fz_dist"]["gathers"] / 1e6, fb["k_fz_dist"]["achieved"], fb["k_fz_dist"]["gb_per_s_from_l2"] / 1e3, fb["k_fz_chain"]["ms"], fb["k_fz_chain"]["matches_walked"] / 1e3,
   fz["reference_visits"] / 1e6, fz["reference_visits_equiv_per_s"] / 1e9)))
c5 = last("r3_c5_1gpu_bench.json"); p5 = json.load(open(D + "/r3_c5_pmc_summary.json"))
rows.append(("C5 on ONE GPU (50M docs / 2M terms, 1.6 G postings, 8192 mixed queries per step; `profiles/r3_c5_*`)", "%.0fk queries/s (%.1f ms/step, scans %.1f ms: %.2f TB/s algorithmic = %.0f %% of 8 TB/s); HBM traffic %.1f GB per step vs %.1f GB algorithmic; %s mismatches on the sampled queries" % (
   c5["value"] / 1e3, c5["ms_per_step"], c5["roofline"]["kernel_ms"], c5["roofline"]["achieved"] / 1e3, 100 * c5["roofline"]["frac"], p5["hbm_bytes_per_launch"] / 1e9, c5["roofline"]["alg_bytes_per_launch"] / 1e9,
   c5.get("cpu_baseline", {}).get("parity_mismatches", "n/a"))))
sp = last("r3_sparse_ids_bench.json")
rows.append(("sparse random u64 doc ids (SURVEY §8d; `profiles/r3_sparse_ids_bench.json`)", "%.0fk queries/s, %.2f ms/step, %s mismatches against the oracle on the sampled queries (ids only meet the device as ordinals: the same kernels)" % (
   sp["value"] / 1e3, sp["ms_per_step"], sp.get("cpu_baseline", {}).get("parity_mismatches", "n/a"))))
d3 = last("r3_docshard4_c3_bench.json"); d5 = last("r3_docshard4_c5_bench.json")
rows.append(("doc shards (N4), 4 shards on one GPU, blocking call (`profiles/r3_docshard4_*`)", "C3 %.0fk queries/s (%.1f ms per batch, %s mismatches against the whole-index oracle), C5 %.0fk queries/s (%.0f ms per batch)" % (
   d3["value"] / 1e3, d3["ms_per_step"], d3["parity_mismatches_vs_whole_index_oracle"], d5["value"] / 1e3, d5["ms_per_step"])))
rfr = b["refresh"]
rows.append(("incremental refresh at 10M docs (`refresh`)", "one appended doc %.0f ms (the first), one removed doc %.0f ms, doc with a new term %.0f ms + %.0f ms for the next fuzzy search; %d incremental, %d rebuilds" % (
   rfr["append_1_doc_ms"], rfr["remove_1_doc_ms"], rfr["append_doc_with_new_term_ms"], rfr["first_fuzzy_after_new_term_ms"], rfr["incremental_refreshes"], rfr["rebuilds"])))
cb = b["cpu_baseline"]
rows.append(("CPU oracle on the same host (%s, %d hardware threads)" % (cb["cpu_model"], cb["host_cpus"]), "1 core: %.2f queries/s (%d parity mismatches against the GPU on its sample); **%d worker processes (`np`): %.0f queries/s**; genuine reference BK-tree %.1f M candidates/s/core, %.0f tokens/s" % (
   cb["value"], cb["parity_mismatches"], cb["np"]["cores"], cb["np"]["value"], cb["fuzzy"]["candidates_per_s"] / 1e6, cb["fuzzy"]["tokens_per_s"])))
print("| | value |\n|---|---|")
for a, v in rows:
    print("| %s | %s |" % (a, v))
print("\nKEY default_limit_qps=%.0f scanm_ipw=%.0f fz_frac=%.1f ds_c3=%.0f ds_c5=%.0f value=%.0f" % (dl["queries_per_s"], (valu + salu + lds) / win, 100 * fb["k_fz_filter"]["frac"], d3["value"], d5["value"], b["value"]))

/*
 * nxs_gpu_search.hip -- work list, kernel dispatch, blocking search (exact two-pass path), pipelined batches, shard slices
 * (MI355X / gfx950 query path of nxsearch; see nxs_gpu_int.h for the map of the files)
 */
#include "nxs_gpu_int.h"

/* ---- search --------------------------------------------------------- */

/*
 * Work decomposition: every query's doc space is cut into n_groups equal
 * ranges (multiples of TILE_W), one wavefront each.  The number of ranges is
 * proportional to the query's share of the batch's postings, so a query with
 * long lists gets many wavefronts and a sparse one a single one (whose fixed
 * costs -- cursor searches, warm-up of the candidate threshold -- are then
 * paid once).  Items are grouped by kernel class (token-count bucket x
 * tile/step path) and emitted heaviest query first inside a class.
 */

void
delete_worklist(worklist_t *wl)
{
	delete wl;
}

static uint32_t
nt_bucket(uint32_t nt)
{
	return nt <= 1 ? 1 : nt <= 2 ? 2 : nt <= 3 ? 3 : nt <= 5 ? 5 : 8;
}

static void
build_worklist(const nxsgpu_index_t *ix, dev_query_t *hq, uint32_t nq, worklist_t &wl, bool solo = false,
    uint32_t big_k = 0)
{
	const uint64_t tiles = std::max<uint64_t>(1, (ix->n_docs + TILE_W - 1) / TILE_W);
	const gpu_cfg_t &cf = ix->cfg;
	/* (a batch that has the GPU to itself is latency-bound: shorter ranges, more of them) */
	const uint64_t target = cf.wave_target, min_post = solo ? std::min(cf.min_post, cf.min_post_solo) : cf.min_post;
	const bool use_scanr = cf.use_scanr && ix->n_docs < (1ull << 31);
	const bool mask_off = cf.mask_off;
	const uint32_t rmin = cf.rmin;	/* 3: "a AND b" takes k_scan8's sign-bit path */
	const bool by_level = cf.by_level;
	/* (limits > 64 -- big_k -- filter on a histogram threshold: the accumulator tiles,
	 * k_scanr and k_scan1 have that mode, the mask path and the dense-term class do not) */
	/* NXS_GPU_NOSTRAGGLER: keep tiny tile-path classes (below) as launches of their own */
	const bool merge_stragglers = !cf.no_straggler;
	const bool use_scanm = cf.use_scanm && ix->n_docs < (1ull << 31) && big_k == 0;
	const bool scanm_general = cf.scanm_general;
	const uint32_t scanm_minnt = cf.scanm_minnt, scanm_maxnt = cf.scanm_maxnt;
	/* k_scanm if the densest list holds at most this fraction of the docs */
	const double scanm_dens = cf.scanm_dens;
	std::vector<uint64_t> work(nq);
	std::vector<uint32_t> order(nq), cls(nq);
	uint64_t total = 0;

	for (uint32_t i = 0; i < nq; i++) {
		uint64_t w = 0, wmax = 0;
		for (uint32_t t = 0; t < hq[i].nt; t++) {
			const uint64_t df = hq[i].pend[t] - hq[i].pbeg[t];
			w += df;
			wmax = std::max(wmax, df);
		}
		work[i] = w;
		total += w;
		order[i] = i;
		if (hq[i].nt > 8) {
			cls[i] = 0;
		} else {
			/* pure OR: every non-empty presence mask matches => no mask array */
			bool or_only = hq[i].nt >= 2 && mask_off;
			for (uint32_t m = 1; or_only && m < (1u << hq[i].nt); m++) {
				or_only = (hq[i].truth[m >> 5] >> (m & 31)) & 1;
			}
			const uint32_t mm = or_only ? 1u : 0u;
			cls[i] = 1u * 64 + mm * 16 + nt_bucket(hq[i].nt);
			/* pure OR of 2..8 tokens whose lists are sparse: mask path (k_scanm).
			 * Dense lists stream faster through the accumulator tiles. */
			/* ... or any expression without a required token: the bound in the
			 * byte map does not depend on the operators, the truth table is
			 * applied to the few docs that get scored
			 * (only where matches are common enough for a threshold to form:
			 * at least half of the tokens satisfy the expression on their own --
			 * "(a AND b) OR (c AND d)" floods the scoring stage and stays on the
			 * accumulator tiles: 3.3 ms there, 5.4 ms here) */
			uint32_t singles = 0;
			for (uint32_t t = 0; t < hq[i].nt && t < 8; t++) {
				const uint32_t m1 = 1u << t;
				singles += (hq[i].truth[m1 >> 5] >> (m1 & 31)) & 1;
			}
			const bool no_req = hq[i].req == 0 && hq[i].nt >= 2 && hq[i].nt <= 8 && mm != 2 &&
			    2 * singles >= hq[i].nt;
			if ((or_only || (no_req && scanm_general)) && use_scanm &&
			    hq[i].nt >= scanm_minnt && hq[i].nt <= scanm_maxnt &&
			    (double)wmax <= scanm_dens * (double)ix->n_docs) {
				cls[i] = 4u * 64 + (or_only ? 16u : 0u) + nt_bucket(hq[i].nt);
				/* ... on presence bits (k_scanb) where that kernel is the faster one: its cost per
				 * posting does not fall with the lists' density as the byte map's does, so it
				 * takes the queries whose lists TOGETHER hold few docs (measured cross-over on
				 * 10M docs: 5-term ORs of rank 500-1000 -29 %, of rank 100-1000 +9 %) */
				if (cf.use_scanb && hq[i].nt <= 5 && (double)w <= cf.scanb_dens * (double)ix->n_docs) {
					cls[i] += 2u * 64;
				}
				/* ... on doc stripes (k_scans) when every term has a rank directory: the stripes'
				 * slices of the lists are table lookups, no per-term window state */
				if ((cls[i] >> 6) == 4 && cf.use_scans && ix->n_post < (1ull << 32) && ix->d_bmrank) {
					bool all = true;
					for (uint32_t t = 0; t < hq[i].nt; t++) {
						all = all && hq[i].bm_col[t] != 0xffffffffu;
					}
					if (all) {
						cls[i] += 4u * 64;
						/* (longer ranges: a stripe range's fixed costs -- set-up, the cold sub-ranges, ~1.3 flushes --
						 * are paid per wavefront) */
						if (cf.scans_workpct != 100) {
							total -= work[i];
							work[i] = std::max<uint64_t>(1, work[i] * cf.scans_workpct / 100);
							total += work[i];
						}
					}
				}
			} else if (or_only && use_scanm && cf.use_drop && hq[i].drop_mask &&
			    hq[i].nt >= scanm_minnt && hq[i].nt <= scanm_maxnt) {
				/*
				 * A pure OR of sparse terms AND dense ones: the mask path on the
				 * sparse terms, the dense lists leave the scan once the threshold
				 * exceeds their joint ceiling (k_scanm<.., DROP>).  Needs enough
				 * sparse postings for a threshold to form in every doc range; the
				 * work is what the sparse lists hold.
				 */
				uint64_t ws = 0;
				uint32_t n_sparse = 0;
				for (uint32_t t = 0; t < hq[i].nt; t++) {
					if (!((hq[i].drop_mask >> t) & 1)) {
						ws += hq[i].pend[t] - hq[i].pbeg[t];
						n_sparse++;
					} else if (hq[i].outl_tfidf) {
						const size_t c = hq[i].drop_col[t];
						ws += ix->outl_off[c + 1] - ix->outl_off[c];	/* (a dropped term's outlier list is scanned) */
					}
				}
				if (n_sparse && ws >= cf.drop_minpost) {
					/* TF-IDF: from here on the dropped tokens' lists are their outlier lists
					 * (kernels that stream the terms' own lists must not see this query again:
					 * qflags) */
					for (uint32_t t = 0; t < hq[i].nt && hq[i].outl_tfidf && !cf.drop_tiles; t++) {
						const size_t c = hq[i].drop_col[t];
						if (((hq[i].drop_mask >> t) & 1) && ix->outl_off[c + 1] > ix->outl_off[c]) {
							hq[i].pbeg[t] = ix->outl_off[c];
							hq[i].pend[t] = ix->outl_off[c + 1];
							hq[i].outl_mask |= 1u << t;
							hq[i].qflags |= 1;
						}
					}
					if (!cf.drop_tiles) {
						total -= work[i];
						work[i] = cf.drop_workmul * (ws + 16384);	/* latency-bound wavefronts: more, shorter ranges */
						total += work[i];
					}
					cls[i] = 5u * 64 + 16u + nt_bucket(hq[i].nt);
					/* ... on doc stripes (k_cold + k_scans<.., DROP>) if the sparse terms all have a rank
					 * directory and no dropped term brings an outlier list (those have none) */
					if (cf.use_scans && cf.use_scans_drop && !hq[i].outl_tfidf && ix->d_dense_q8 && ix->n_post < (1ull << 32) && ix->d_bmrank && !hq[i].outl_mask) {
						bool all = true;
						for (uint32_t t = 0; t < hq[i].nt; t++) {
							all = all && (((hq[i].drop_mask >> t) & 1) || hq[i].bm_col[t] != 0xffffffffu);
						}
						if (all) {
							cls[i] = 9u * 64 + 16u + nt_bucket(hq[i].nt);
						}
					}
				}
			}
			/* required terms: intersect first (k_scanr).  Its work is set by
			 * the shortest required list; longer lists are mostly skipped */
			if (hq[i].n_req && hq[i].nt >= rmin && use_scanr) {
				const uint64_t dfd = hq[i].pend[hq[i].slot_tok[0]] - hq[i].pbeg[hq[i].slot_tok[0]];
				uint64_t wr = 0;
				for (uint32_t t = 0; t < hq[i].nt; t++) {
					wr += std::min<uint64_t>(hq[i].pend[t] - hq[i].pbeg[t], 4 * dfd);
				}
				total -= work[i];
				work[i] = wr;
				total += wr;
				/* (four required terms and more: rounds of whole driver windows, k_scanr<.., true>) */
			cls[i] = 3u * 64 + ((SCANR_HASH && hq[i].n_req >= 4) ? 16u : 0u) + nt_bucket(hq[i].nt);
				/*
				 * Two required terms and more that all have a block-presence bitmap: AND the
				 * bitmaps and look at the postings of the surviving 64-doc blocks only
				 * (k_scanq) -- if few blocks are expected to survive (independent lists: a
				 * block holds term t with probability 1 - (1 - df_t / N)^64) against what
				 * the driver list would cost k_scanr.
				 */
				if (cf.use_blkmap && hq[i].n_req >= 2 && ix->n_post < (1ull << 32)) {
					double surv = (double)ix->n_docs / 64.0;
					double em = (double)ix->n_docs;		/* expected docs holding every required term */
					bool all = true;
					for (uint32_t t = 0; t < hq[i].nt; t++) {
						if (!((hq[i].req >> t) & 1)) {
							continue;
						}
						all = all && hq[i].bm_col[t] != 0xffffffffu;
						const double rho = (double)(hq[i].pend[t] - hq[i].pbeg[t]) / (double)std::max<uint64_t>(ix->n_docs, 1);
						double e64 = 1.0 - std::min(rho, 1.0);	/* ^64 by squaring (std::pow: 2 500 calls a batch) */
						e64 *= e64; e64 *= e64; e64 *= e64; e64 *= e64; e64 *= e64; e64 *= e64;
						surv *= 1.0 - e64;
						em *= std::min(rho, 1.0);
					}
					/* (limits > 64: k_scanq<.., BIG> emits EVERY match -- for queries that expect a
					 * handful; more than a range's candidate list holds sends the query to the exact path) */
					if (all && surv * cf.bm_gain < (double)dfd && (big_k == 0 || em < cf.bigq_em)) {
						total -= work[i];
						/* the bitmaps' words + the surviving blocks (a lane each), in posting units */
						work[i] = (uint64_t)(ix->n_docs / 256 + surv * 64.0) + 1;
						total += work[i];
						cls[i] = 7u * 64 + (cls[i] & 16u) + nt_bucket(hq[i].nt);
					}
				}
			}
		}
	}
	/*
	 * Stragglers.  A pure-OR query whose lists are too dense for the mask path and that
	 * cannot drop them either (two dense terms, a ceiling too close to the sparse ones)
	 * takes the accumulator tiles -- a class of ONE or two queries in a C3 batch: a launch
	 * of its own on the scan stream, 0.1 ms of latency for 8 MB of postings, with nothing
	 * to run beside.  The mask kernel takes any density (it is merely slower per dense
	 * posting): up to four such queries join the batch's mask-path class of their shape,
	 * where their ranges are wavefronts among tens of thousands.
	 */
	if (merge_stragglers && use_scanm) {
		uint32_t n_in[16 * 64] = { 0 };
		for (uint32_t i = 0; i < nq; i++) {
			n_in[cls[i] & 1023]++;
		}
		/* (the batch's mask-path class of a shape: on doc stripes -- k_scans -- if the query's terms all have
		 * a rank directory and that class is the populated one, else on register windows) */
		auto mask_class = [&](uint32_t i, uint32_t shape) -> uint32_t {
			bool all = cf.use_scans && ix->n_post < (1ull << 32) && ix->d_bmrank;
			for (uint32_t t = 0; all && t < hq[i].nt; t++) {
				all = hq[i].bm_col[t] != 0xffffffffu;
			}
			return (all && n_in[8u * 64 + shape] >= 32) ? 8u * 64 + shape : 4u * 64 + shape;
		};
		for (uint32_t i = 0; i < nq; i++) {
			const uint32_t c = cls[i];
			if ((c >> 6) == 1 && ((c >> 4) & 3) == 1 && (c & 15) >= 2 && n_in[c] <= 4 &&
			    hq[i].nt >= scanm_minnt && hq[i].nt <= scanm_maxnt) {
				const uint32_t to = mask_class(i, 16u + (c & 15));
				if (n_in[to] >= 32) {
					cls[i] = to;
				}
			}
			/* (the same for a handful of very sparse queries that would take k_scanb: a launch of
			 * their own only pays with enough of them) */
			if ((c >> 6) == 6 && n_in[c] < 64) {
				const uint32_t to = mask_class(i, c & 63);
				if (n_in[to] >= 32) {
					cls[i] = to;
				}
			}
		}
	}
	/* (limits > 64: a range's own threshold needs well over k matches to form, and
	 * every range that starts cold emits k candidates before it has one) */
	/* (a batch with the stripe class: somewhat longer ranges for everything -- measured on C3, 57 344 against 65 536
	 * wavefronts with the class itself at 70 %: 1.00 -> 1.05 M queries/s; single-token batches keep the finer split) */
	bool any_scans = false;
	for (uint32_t i = 0; i < nq && !any_scans; i++) {
		any_scans = (cls[i] >> 6) == 8;
	}
	/* (... and a huge batch -- C5: 29 G postings -- more wavefronts than the target: a range of more than cf.max_post
	 * postings leaves the step's tail to a few long wavefronts (C5: 358k -> 377k queries/s); at most four times the target: the staging area's bound) */
	uint64_t target_eff = any_scans ? cf.wave_target_scans : target;
	target_eff = std::min<uint64_t>(4 * target_eff, std::max<uint64_t>(target_eff, total / std::max<uint64_t>(cf.max_post, 1)));
	const uint64_t per_wave = std::max<uint64_t>(std::max<uint64_t>(min_post, (uint64_t)big_k * cf.big_minpost),
	    total / std::max<uint64_t>(target_eff, 1) + 1);
	/* launch order of the classes: the mask path first -- a class's heap replay
	 * runs beside the NEXT class's scan, and the last class (required-term
	 * queries: few candidates, short replay) is the one left exposed */
	/* (the sparse + dense class leads: it runs on a stream of its own, beside the rest) */
	auto cls_key = [&](uint32_t c) -> uint32_t { return (c >> 6) == 9 ? (c & 63) : (c >> 6) == 5 ? 32 + (c & 63) : (c >> 6) == 8 ? 64 + (c & 63) : (c >> 6) == 4 ? 128 + (c & 63) : (c >> 6) == 6 ? 192 + (c & 63) : c + 256; };
	std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
		if (cls[x] != cls[y]) return cls_key(cls[x]) < cls_key(cls[y]);
		return work[x] != work[y] ? work[x] > work[y] : x < y;
	});
	wl.qmeta.assign(nq, qmeta_t());
	wl.items.clear();
	wl.launches.clear();
	wl.n_segs = 0;
	wl.need_cursors = false;
	for (uint32_t i = 0; i < nq; i++) {
		uint64_t per_i = per_wave;
		if (solo) {
			/*
			 * Alone on the GPU every range starts cold and emits its own early
			 * maxima (~10 (1 + ln(postings / 10)) candidates each), which the replay
			 * then streams through one wavefront; a range's scan is a chain of
			 * dependent window loads.  Scan time falls with the number of ranges R,
			 * replay time grows with it: the sum is smallest near R = sqrt(n / 84),
			 * i.e. sqrt(84 n) postings per range (2M postings: 154 ranges, not 2000).
			 */
			/* (never finer than the batch-wide rule: the work list's size bound rests on it) */
			per_i = std::max<uint64_t>(per_wave, (uint64_t)std::sqrt(84.0 * (double)work[i]));
		}
		uint64_t g = std::max<uint64_t>(1, (work[i] + per_i - 1) / per_i);
		g = std::min<uint64_t>(g, tiles);
		g = std::min<uint64_t>(g, 65535);
		const uint64_t tiles_per = (tiles + g - 1) / g;
		g = (tiles + tiles_per - 1) / tiles_per;
		qmeta_t &m = wl.qmeta[i];
		m.n_groups = (uint32_t)g;
		m.group_docs = (uint32_t)std::min<uint64_t>(tiles_per * TILE_W, 0xffffffffu & ~(uint64_t)(TILE_W - 1));
		/* single-token queries on k_scan1: any split of the list into contiguous
		 * pieces, highest docs first, feeds the heap the same sequence -- split by
		 * posting index and the batch needs no k_cursors launch */
		m.pad = ((cls[i] >> 6) == 1 && (cls[i] & 15) == 1 && !cf.no_scan1 && !cf.old_scan &&
		    ix->n_docs < (1ull << 31)) ? 1u : 0u;
		wl.need_cursors = wl.need_cursors || m.pad == 0;
	}
	for (uint32_t i = 0; i < nq; i++) {
		wl.qmeta[i].seg_first = wl.n_segs;
		wl.n_segs += wl.qmeta[i].n_groups;
	}
	/* (both arrays are sized once and written through plain pointers: 130 000 push_back calls cost 0.1 ms a batch) */
	wl.bnd_q.resize((size_t)wl.n_segs + nq);
	{
		uint32_t *bp = wl.bnd_q.data();
		for (uint32_t i = 0; i < nq; i++) {
			/* query i owns boundaries seg_first + i ... + n_groups (inclusive) */
			bp = std::fill_n(bp, (size_t)wl.qmeta[i].n_groups + 1, i);
		}
	}
	wl.items.resize(wl.n_segs);
	item_t *const items = wl.items.data();
	size_t n_items = 0;
	wl.qorder = order;
	/*
	 * Inside a class, items go out level by level: level l of every query
	 * (its l-th highest doc range) before level l+1 of any.  All items carry
	 * about per_wave postings, so this costs no balance, and it spreads one
	 * query's ranges in time: when a range starts, higher ranges of its query
	 * have usually finished and published their threshold (range_hint).
	 */
	for (uint32_t o0 = 0; o0 < nq; ) {
		uint32_t o1 = o0, max_g = 0;
		while (o1 < nq && cls[order[o1]] == cls[order[o0]]) {
			max_g = std::max(max_g, wl.qmeta[order[o1]].n_groups);
			o1++;
		}
		launch_t l;
		l.postings = 0;
		l.first = (uint32_t)n_items;
		l.nt_bucket = cls[order[o0]] & 15;
		l.nomask = (cls[order[o0]] >> 4) & 3;	/* 0 mask array, 1 pure OR, 2 two-token AND */
		l.kind = cls[order[o0]] >> 6;
		if (by_level) {
			/* the class is sorted by work, so n_groups does not increase along
			 * it (checked): the queries that still have a level `lev` form a
			 * prefix, and the loop is linear in the number of items */
			bool mono = true;
			for (uint32_t oi = o0 + 1; oi < o1 && mono; oi++) {
				mono = wl.qmeta[order[oi]].n_groups <= wl.qmeta[order[oi - 1]].n_groups;
			}
			uint32_t live_end = o1;
			for (uint32_t lev = 0; lev < max_g; lev++) {
				while (mono && live_end > o0 && wl.qmeta[order[live_end - 1]].n_groups <= lev) {
					live_end--;
				}
				for (uint32_t oi = o0; oi < live_end; oi++) {
					const uint32_t i = order[oi];
					if (lev < wl.qmeta[i].n_groups) {
						item_t it;
						it.q = i;
						it.g = wl.qmeta[i].n_groups - 1 - lev;
						items[n_items++] = it;
					}
				}
				/*
				 * Single-token queries (k_scan1: a wavefront is ~20 us of streaming):
				 * a dense term is thousands of ranges that would all start at once,
				 * cold, each handing its ~10 (1 + ln(postings / 10)) early maxima to
				 * the one wavefront that replays the query -- 150 000 candidates for a
				 * term holding 90 % of 10M docs, 0.3 ms of replay behind 0.05 ms of
				 * scanning.  The TOP range of every query goes first, in a launch of
				 * its own: when the others start it has published the 10th best of its
				 * 4096 postings, and they emit a seventh of that.
				 */
				/*
				 * The sparse + dense class (k_cold + k_scanm<.., DROP>) of a mixed batch is a few
				 * thousand wavefronts: ALL of them fit the GPU at once, so no range ever finds a
				 * threshold published by a higher one -- every range walks its cold phase and
				 * pushes on a weak threshold (8 x the pending docs of the plain class).  The first
				 * level(s) go ahead in a launch of their own here too.
				 */
				if (lev + 1 == cf.drop_split && (l.kind == 5 || l.kind == 9) && max_g > cf.drop_split && !big_k && !solo) {
					launch_t l0 = l;
					l0.count = (uint32_t)n_items - l0.first;
					l0.q_first = o0;
					l0.q_count = 0;			/* (no replay behind this one) */
					l0.postings = 0;
					wl.launches.push_back(l0);
					l.first = (uint32_t)n_items;
				}
				if (lev == 0 && l.kind == 1 && l.nt_bucket == 1 && max_g >= cf.scan1_split && !big_k && !solo) {
					launch_t l0 = l;
					l0.count = (uint32_t)n_items - l0.first;
					l0.q_first = o0;
					l0.q_count = 0;			/* (no replay behind this one) */
					l0.postings = 0;		/* (the class's postings are charged to the launch its queries end in) */
					wl.launches.push_back(l0);
					l.first = (uint32_t)n_items;
				}
			}
		} else {
			for (uint32_t oi = o0; oi < o1; oi++) {
				const uint32_t i = order[oi];
				for (uint32_t g = wl.qmeta[i].n_groups; g-- > 0; ) {
					item_t it;
					it.q = i;
					it.g = g;
					items[n_items++] = it;
				}
			}
		}
		l.count = (uint32_t)n_items - l.first;
		l.q_first = o0;
		l.q_count = o1 - o0;
		l.postings = 0;
		for (uint32_t oi = o0; oi < o1; oi++) {
			const dev_query_t &dq = hq[order[oi]];
			for (uint32_t t = 0; t < dq.nt; t++) {
				/* (a dropped token whose list was replaced by its outlier list: the term's own df is not known
				 * here any more -- the class's figure then counts what is scanned) */
				l.postings += dq.pend[t] - dq.pbeg[t];
			}
		}
		wl.launches.push_back(l);
		o0 = o1;
	}
}

/* the gathered record blocks, device -> mapped pinned host memory (8-byte words; blocks are multiples of 8) */
__global__ void __launch_bounds__(256)
k_records_out(const uint64_t *__restrict__ src, uint64_t *__restrict__ dst, uint64_t n8)
{
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
		dst[i] = src[i];
	}
}

static void
launch_cursors(nxsgpu_index_t *ix, const scan_args_t &a, const uint32_t *d_bnd_q, uint32_t n_bnd,
    hipStream_t stream = NULL)
{
	nxs_launch_cursors(a, d_bnd_q, n_bnd, stream ? stream : ix->stream);
}

/*
 * One scan launch per query class.  With `ra` (top-k filter pass) the heap
 * replay of a class is queued on the second stream as soon as the class's
 * scan is: the replay is a few latency-bound wavefronts (one per query) and
 * runs beside the next class's scan instead of after all of them.
 */
static void
launch_scan(int MODE, nxsgpu_index_t *ix, const scan_args_t &a0, const worklist_t &wl,
    const replay_args_t *ra = NULL, const uint32_t *d_qorder = NULL, hipEvent_t scans_done = NULL,
    bool replays_aside = false, hipStream_t replay_stream = NULL, nxsgpu_index::dev_slot_t *psl = NULL,
    hipEvent_t ahead_done = NULL, hipStream_t early_stream = NULL, hipEvent_t early_done = NULL)
{
	const hipStream_t st_rp = replay_stream ? replay_stream : ix->stream2;
	bool forked = false, forked3 = false, early_any = false;
	/* where the replay's heap lives: across the lanes (k <= 64) or in LDS (MODE_BIG) */
	const int heap = a0.k <= WAVE ? HEAP_REG : HEAP_LDS;
	const size_t heap_lds = heap == HEAP_LDS ? (size_t)a0.k * 8 : 0;
	const launch_t *last_launch = NULL;
	size_t n_launches = 0;

	for (const launch_t &l : wl.launches) {
		n_launches += l.count != 0;
	}
	/* the sparse + dense class goes to its own stream when there is something to
	 * run it beside (top-k pass only: its replay follows it there) */
	const bool side3 = MODE == MODE_TOPK && ra && n_launches > 1 && a0.k >= 1 && a0.k <= WAVE && ix->cfg.drop_side;
	/*
	 * The conjunctive classes of a mixed batch (k_scanr, k_scanq: a few thousand latency-bound
	 * wavefronts, 0.3 ms on the scan stream with the chip nearly idle) run EARLY: on the upload
	 * stream, behind this batch's k_cursors -- i.e. beside the previous batch's big scans --, their
	 * heap replays with them; the batch's end waits for them (early_done).
	 */
	size_t n_late = 0;
	for (const launch_t &l : wl.launches) {
		n_late += l.count && !(l.kind == 3 || l.kind == 7);
	}
	/*
	 * MODE_BIG, several batches in flight: the scan stream is what a step costs (one batch's scans
	 * behind the other's; the replays -- milliseconds -- run aside, a stream per batch).  The
	 * conjunctive classes and their short replays go to the batch's REPLAY stream, in front of the
	 * other classes' replays: beside this batch's tile scans instead of behind them.
	 */
	const bool early_big = MODE == MODE_BIG && replays_aside && replay_stream && ra && n_late >= 1 && ix->cfg.and_early;
	if (early_big) {
		early_stream = st_rp;
	}
	const bool early_ok = early_big || (early_stream && early_done && MODE == MODE_TOPK && ra && n_late >= 1 &&
	    a0.k >= 1 && a0.k <= WAVE);
	for (const launch_t &l : wl.launches) {
		if (l.count && !(side3 && (l.kind == 5 || l.kind == 9)) && !(early_ok && (l.kind == 3 || l.kind == 7))) {
			last_launch = &l;
		}
	}
	std::vector<const launch_t *> seq;
	seq.reserve(wl.launches.size());
	if (early_big) {
		bool any = false;
		for (const launch_t &l : wl.launches) {
			if (l.count && (l.kind == 3 || l.kind == 7)) {
				seq.push_back(&l);
				any = true;
			}
		}
		if (any) {	/* (behind the cursors, which a MODE_BIG batch runs on the scan stream) */
			(void)hipEventRecord(ix->ev_fork3, ix->stream);
			(void)hipStreamWaitEvent(st_rp, ix->ev_fork3, 0);
		}
	}
	for (const launch_t &l : wl.launches) {
		if (!(early_big && l.count && (l.kind == 3 || l.kind == 7))) {
			seq.push_back(&l);
		}
	}
	for (const launch_t *lp : seq) {
		const launch_t &l = *lp;
		scan_args_t a = a0;
		const unsigned grid = l.count;
		/* mask path / dense-term class: top-k filter pass only; the exact passes
		 * (count, emit all) of these queries take the accumulator tiles */
		const bool topk64 = MODE == MODE_TOPK && a.k >= 1 && a.k <= WAVE;

		if (l.count == 0) {
			continue;
		}
		a.item_base = l.first;
		/* profiling: events around this class's scan kernels, on the stream they go to */
		const bool early = early_ok && (l.kind == 3 || l.kind == 7);
		const hipStream_t cls_stream = (side3 && (l.kind == 5 || l.kind == 9)) ? ix->stream3 : early ? early_stream : ix->stream;
		int pc = -1;
		if (psl && psl->ev_cls_ok && psl->n_cls < NXSGPU_PROF_CLS && MODE_FILTERS(MODE)) {
			pc = (int)psl->n_cls++;
			/* (bit 7: a launch of top levels sent ahead -- no queries end in it) */
			psl->cls_key[pc] = l.kind << 8 | l.nomask << 4 | l.nt_bucket | (l.q_count == 0 && ra ? 0x80u : 0u);
			psl->cls_post[pc] = l.postings;
			psl->cls_q[pc] = l.q_count;
			(void)hipEventRecord(psl->ev_cls[pc][0], cls_stream);
		}
		auto prof_stop = [&]() {
			if (pc >= 0) {
				(void)hipEventRecord(psl->ev_cls[pc][1], cls_stream);
			}
		};
		/* (k_scanb<.., DROP>: up to five tokens) */
		a.flags |= (l.kind == 5 && ix->cfg.drop_b && l.nt_bucket <= 5) ? 8u : 0u;
		/* this launch's retry list (mask path only) */
		const size_t li = (size_t)(&l - wl.launches.data());
		const bool retry = a0.retry_items && li < RETRY_LISTS && topk64 && (l.kind == 4 || l.kind == 5 || l.kind == 6 || l.kind == 8 || l.kind == 9);
		a.retry_count = retry ? a0.retry_count + li : NULL;
		a.retry_items = retry ? a0.retry_items + li * RETRY_CAP : NULL;
		a.retry_cap = retry ? RETRY_CAP : 0;
		/* the ranges whose pending list overflowed, once more on the accumulator
		 * tiles: a fixed, small grid whose wavefronts beyond the list's end return */
		bool retry_pending = false;
		auto launch_retry = [&](hipStream_t st) {
			if (retry) {
				scan_args_t a2 = a;
				a2.flags |= 2;
				nxs_launch_scan8(MODE_TOPK, l.nt_bucket, (l.kind == 5 || l.kind == 9 || l.nomask == 1) ? 1u : 0u, RETRY_CAP, st, a2);
			}
		};
		if (l.kind == 9) {
			a.flags |= 16u;		/* the class's second kernel is k_scans<.., DROP> */
		}
		if (side3 && (l.kind == 5 || l.kind == 9)) {
			replay_args_t r = *ra;
			r.qlist = d_qorder + l.q_first;
			if (!forked3) {
				(void)hipEventRecord(ix->ev_fork3, ix->stream);
				(void)hipStreamWaitEvent(ix->stream3, ix->ev_fork3, 0);
				if (ahead_done) {
					(void)hipStreamWaitEvent(ix->stream3, ahead_done, 0);	/* the class's top ranges (upload stream) */
				}
				forked3 = true;
			}
			if (ix->cfg.drop_tiles) {
				nxs_launch_scan8(MODE_TOPK, l.nt_bucket, 1u, grid, ix->stream3, a);
				prof_stop();
			} else {
				a.flags |= ix->cfg.drop_prio ? 1u : 0u;
				nxs_launch_drop_class(l.nt_bucket, grid, ix->stream3, a);
				prof_stop();
				launch_retry(ix->stream3);
			}
			if (l.q_count) {
				nxs_launch_replay(HEAP_REG, l.q_count, 0, ix->stream3, r);
			}
			continue;
		}
		if (early) {
			replay_args_t r = *ra;
			r.qlist = d_qorder + l.q_first;
			if (l.kind == 7 && (topk64 || MODE == MODE_BIG)) {
				nxs_launch_scanq(l.nt_bucket, grid, early_stream, a);
			} else {
				nxs_launch_scanr(MODE, l.nt_bucket, l.nomask == 1, grid, early_stream, a);
			}
			prof_stop();
			if (l.q_count) {
				nxs_launch_replay(heap, l.q_count, heap_lds, early_stream, r);
			}
			early_any = true;
			continue;
		}
		if (l.kind == 0) {
			nxs_launch_scan_generic(MODE, true, grid, ix->stream, a);
		} else if (ix->cfg.old_scan || ix->n_docs >= (1ull << 31)) {
			nxs_launch_scan_generic(MODE, false, grid, ix->stream, a);
		} else if (l.kind == 1) {
			if (l.nt_bucket == 1 && !ix->cfg.no_scan1) {
				nxs_launch_scan1(MODE, grid, ix->stream, a);
			} else {
				nxs_launch_scan8(MODE, l.nt_bucket, l.nt_bucket == 1 ? 0u : l.nomask, grid, ix->stream, a);
			}
		} else if (l.kind == 4 || l.kind == 6 || l.kind == 8) {
			if (topk64) {
				if (l.kind == 8) {
					nxs_launch_scans(l.nt_bucket, l.nomask != 1, grid, ix->stream, a);
				} else if (l.kind == 6) {
					nxs_launch_scanb(l.nt_bucket, l.nomask != 1, false, grid, ix->stream, a);
				} else {
					nxs_launch_scanm(l.nt_bucket, l.nomask != 1, grid, ix->stream, a);
				}
				/* (the second chance of its overflowed ranges: in front of the class's heap
				 * replay, on the replay's stream -- not in front of the next class's scan) */
				retry_pending = retry;
				if (!(ra && l.q_count)) {
					launch_retry(ix->stream);
					retry_pending = false;
				}
			} else {
				nxs_launch_scan8(MODE, l.nt_bucket, l.nomask == 1 ? 1u : 0u, grid, ix->stream, a);
			}
		} else if (l.kind == 5 || l.kind == 9) {
			/* sparse + dense pure OR: top-k pass with the dense lists dropped */
			if (topk64 && !ix->cfg.drop_tiles) {
				nxs_launch_drop_class(l.nt_bucket, grid, ix->stream, a);
				launch_retry(ix->stream);
			} else {
				nxs_launch_scan8(MODE, l.nt_bucket, 1u, grid, ix->stream, a);
			}
		} else if (l.kind == 3 || l.kind == 7) {
			if (l.kind == 7 && (topk64 || MODE == MODE_BIG)) {
				nxs_launch_scanq(l.nt_bucket, grid, ix->stream, a);
			} else {
				nxs_launch_scanr(MODE, l.nt_bucket, l.nomask == 1, grid, ix->stream, a);
			}
		}
		prof_stop();
		if (ra && l.q_count) {
			replay_args_t r = *ra;
			r.qlist = d_qorder + l.q_first;
			if (&l == last_launch && !replays_aside) {
				/* nothing left to run beside it: same stream, no event
				 * round trip (a single query has only this one) */
				if (retry_pending) {
					launch_retry(ix->stream);
				}
				if (scans_done) {
					(void)hipEventRecord(scans_done, ix->stream);
					scans_done = NULL;
				}
				nxs_launch_replay(heap, l.q_count, heap_lds, ix->stream, r);
			} else {
				(void)hipEventRecord(ix->ev_cls, ix->stream);
				(void)hipStreamWaitEvent(st_rp, ix->ev_cls, 0);
				if (retry_pending) {
					launch_retry(st_rp);
				}
				nxs_launch_replay(heap, l.q_count, heap_lds, st_rp, r);
				forked = true;
			}
			retry_pending = false;
		}
	}
	if (scans_done) {
		(void)hipEventRecord(scans_done, ix->stream);
	}
	if (early_any && !early_big) {
		(void)hipEventRecord(early_done, early_stream);
		(void)hipStreamWaitEvent(ix->stream, early_done, 0);
		if (replays_aside) {
			(void)hipStreamWaitEvent(st_rp, early_done, 0);		/* the batch ends there */
		}
	}
	/*
	 * replays_aside (MODE_BIG batches: a replay is thousands of heap insertions on one
	 * lane, milliseconds): every replay runs on the second stream and the scan
	 * stream does NOT wait for them -- the next batch's scans run beside this
	 * batch's replays; the caller takes the batch's end from the replay stream.
	 */
	if (forked && !replays_aside) {
		(void)hipEventRecord(ix->ev_join, st_rp);
		(void)hipStreamWaitEvent(ix->stream, ix->ev_join, 0);
	}
	if (forked3) {
		(void)hipEventRecord(ix->ev_join3, ix->stream3);
		(void)hipStreamWaitEvent(ix->stream, ix->ev_join3, 0);
		if (replays_aside) {
			(void)hipStreamWaitEvent(st_rp, ix->ev_join3, 0);	/* the batch ends there */
		}
	}
}

/*
 * Core of the search: fills device outputs.  If `d_out_*` are NULL the
 * results are copied to the host into `res`.
 */
/*
 * Device form of the batch's plans: posting ranges of the tokens, truth
 * table, required-token mask and k_scanr's slot order.  -1 on a bad plan.
 */
struct fill_job_t {
	const nxsgpu_index_t *ix;
	int		algo;
	const nxsgpu_query_t *queries;
	dev_query_t *	hq;
	bool		allow_drop;
	std::atomic<uint64_t> total_post;
	std::atomic<uint32_t> bad;	/* first bad query + 1 (0: none), and what is wrong with it */
	std::atomic<uint32_t> bad_term;
};

/* queries [lo, hi) of the batch: independent of each other (the caller's worker threads, fill_dev_queries) */
static void
fill_dev_chunk(void *arg, size_t lo, size_t hi)
{
	fill_job_t &J = *(fill_job_t *)arg;
	const nxsgpu_index_t *ix = J.ix;
	const int algo = J.algo;
	const bool allow_drop = J.allow_drop;
	const bool valid = (algo == NXSGPU_BM25) ? ix->bm25_valid : ix->tfidf_valid;
	const bool no_req = ix->cfg.no_req;
	uint64_t total_post = 0;

	for (uint32_t i = (uint32_t)lo; i < (uint32_t)hi; i++) {
		const nxsgpu_query_t &q = J.queries[i];
		dev_query_t &d = J.hq[i];
		/* (not the whole 900 bytes: posting ranges beyond the query's tokens and program
		 * bytes beyond prog_len are never read -- 0.7 MB less to write per 1024 queries) */
		d.nt = d.prog_len = 0;
		memset(d.pbeg, 0, 8 * sizeof(d.pbeg[0]));
		memset(d.pend, 0, 8 * sizeof(d.pend[0]));
		memset(d.truth, 0, offsetof(dev_query_t, prog) - offsetof(dev_query_t, truth));
		if (q.n_tokens > NXSGPU_MAX_TOKENS || q.prog_len > NXSGPU_MAX_PROG) {
			uint32_t none = 0;
			(void)J.bad.compare_exchange_strong(none, i + 1);
			continue;
		}
		/* invalid statistics => every pair is skipped (ranking.c:86-88,156-166) */
		d.nt = valid ? q.n_tokens : 0;
		d.prog_len = q.prog_len;
		memcpy(d.prog, q.prog, q.prog_len);
		memcpy(d.truth, q.truth, sizeof(d.truth));
		/* tokens common to every matching presence mask (<= 8 tokens) */
		d.req = 0;
		if (d.nt && d.nt <= 8 && !no_req) {
			uint32_t r = (1u << d.nt) - 1;
			for (uint32_t m = 1; m < (1u << d.nt); m++) {
				if ((d.truth[m >> 5] >> (m & 31)) & 1) {
					r &= m;
				}
			}
			d.req = r;
		}
		for (uint32_t t = 0; t < d.nt; t++) {
			const uint32_t tid = q.term_id[t];
			if (tid == 0) {
				uint32_t none = 0;
				if (J.bad.compare_exchange_strong(none, i + 1)) {
					J.bad_term.store(t + 1);
				}
				d.pbeg[t] = d.pend[t] = 0;
				continue;
			}
			if (tid > ix->n_terms) {
				/* a term the host dictionary has consumed but whose docs this snapshot
				 * does not hold yet (a refresh that stopped half way: partial sync,
				 * dtmap.c:527-535): no postings, like the reference's empty bitmap */
				d.pbeg[t] = d.pend[t] = 0;
				continue;
			}
			d.pbeg[t] = ix->h_post_off[tid];
			d.pend[t] = ix->h_post_off[tid + 1];
			total_post += d.pend[t] - d.pbeg[t];
			if (t < 8 && tid < ix->h_maximp[algo].size()) {
				d.tmax[t] = ix->h_maximp[algo][tid];
			}
		}
		/* block-presence bitmaps of the tokens that have one (k_scanq) */
		for (uint32_t t = 0; t < 8; t++) {
			d.bm_col[t] = 0xffffffffu;
		}
		if (d.nt <= 8 && !ix->bm_terms.empty()) {
			for (uint32_t t = 0; t < d.nt; t++) {
				const auto it = std::lower_bound(ix->bm_terms.begin(), ix->bm_terms.end(), q.term_id[t]);
				if (it != ix->bm_terms.end() && *it == q.term_id[t]) {
					d.bm_col[t] = (uint32_t)(it - ix->bm_terms.begin());
				}
			}
		}
		/* dense tokens (k_scanm<.., DROP>): lists above the mask path's density limit */
		/*
		 * (BM25's tf part saturates, so a term's largest impact says what the term
		 * typically adds.  TF-IDF's log(tf + 1) does not -- one posting with an outlier
		 * tf sets a ceiling that thresholds reach late: 3x slower than the accumulator
		 * tiles -- so there the ceiling is the term's CAP and the postings above it are
		 * scanned as the term's outlier list: nxsgpu_index::outl_off.)
		 */
		d.drop_mask = 0;
		d.outl_mask = 0;
		d.qflags = 0;
		for (uint32_t t = 0; t < 8; t++) {
			d.tcap[t] = d.tmax[t];
		}
		const bool cols = algo == NXSGPU_BM25 || (ix->cfg.tfidf_drop && ix->d_dense_col[algo] &&
		    ix->outl_cap.size() == ix->dense_terms.size());
		if (allow_drop && d.nt >= 2 && d.nt <= 8 && ix->cfg.use_drop && !ix->dense_terms.empty() && cols) {
			for (uint32_t t = 0; t < d.nt; t++) {
				const auto it = std::lower_bound(ix->dense_terms.begin(), ix->dense_terms.end(), q.term_id[t]);
				if (it != ix->dense_terms.end() && *it == q.term_id[t]) {
					const size_t c = (size_t)(it - ix->dense_terms.begin());
					d.drop_mask |= 1u << t;
					d.drop_col[t] = (uint32_t)c;
					if (algo == NXSGPU_TF_IDF) {
						d.tcap[t] = ix->outl_cap[c];
					}
				}
			}
		}
		if (d.drop_mask) {
			/* worth it only while the dense ceiling stays well below what one
			 * sparse posting can add */
			float u = 0.0f, smin = INFINITY;
			for (uint32_t t = 0; t < d.nt; t++) {
				if ((d.drop_mask >> t) & 1) {
					u += d.tcap[t];
				} else {
					smin = std::min(smin, d.tmax[t]);
				}
			}
			if (!(u <= 1.25f * smin)) {
				d.drop_mask = 0;
			}
		}
		/* (a dropped token's list as the scan sees it -- its outlier list, TF-IDF -- is put in
		 * its place by build_worklist, once the query is known to take the dense-term class) */
		d.outl_tfidf = (d.drop_mask && algo == NXSGPU_TF_IDF) ? 1u : 0u;
		if (!d.drop_mask) {
			for (uint32_t t = 0; t < 8; t++) {
				d.tcap[t] = d.tmax[t];
			}
		}
		/* k_scanr slot order: required tokens first, shortest list first */
		d.n_req = 0;
		if (!d.req && d.nt <= 8) {
			/* k_scanb slot order: ascending largest impact -- the commonest term first, so
			 * that the postings that are many meet the bound that is sharp */
			uint32_t ord[8];
			for (uint32_t t = 0; t < d.nt; t++) {
				ord[t] = t;
			}
			std::sort(ord, ord + d.nt, [&](uint32_t x, uint32_t y) {
				return d.tmax[x] != d.tmax[y] ? d.tmax[x] < d.tmax[y] : x < y;
			});
			for (uint32_t t = 0; t < d.nt; t++) {
				d.slot_tok[t] = (uint8_t)ord[t];
			}
		}
		if (d.req && d.nt <= 8) {
			uint32_t ord[8];
			for (uint32_t t = 0; t < d.nt; t++) {
				ord[t] = t;
			}
			std::sort(ord, ord + d.nt, [&](uint32_t x, uint32_t y) {
				const bool rx = (d.req >> x) & 1, ry = (d.req >> y) & 1;
				if (rx != ry) return rx;
				const uint64_t dx = d.pend[x] - d.pbeg[x], dy = d.pend[y] - d.pbeg[y];
				return dx != dy ? dx < dy : x < y;
			});
			for (uint32_t t = 0; t < d.nt; t++) {
				d.slot_tok[t] = (uint8_t)ord[t];
				d.n_req += (d.req >> t) & 1;
			}
		}
	}
	J.total_post.fetch_add(total_post);
}

/* the whole batch, on the caller's worker threads if it has handed any over */
static int
fill_dev_queries(const nxsgpu_index_t *ix, int algo, const nxsgpu_query_t *queries, uint32_t nq,
    dev_query_t *hq, uint64_t &total_post, bool allow_drop = true)
{
	fill_job_t J;

	J.ix = ix;
	J.algo = algo;
	J.queries = queries;
	J.hq = hq;
	J.allow_drop = allow_drop;
	J.total_post.store(0);
	J.bad.store(0);
	J.bad_term.store(0);
	/* (waking the workers costs 50-100 us: only for batches with enough tokens to pay for it -- a C2 batch of
	 * 1024 single-term queries is 20 us of this work on one thread) */
	uint64_t n_tok = 0;
	for (uint32_t i = 0; i < nq; i++) {
		n_tok += queries[i].n_tokens;
	}
	if (ix->par_run && nq >= 128 && n_tok >= 3 * (uint64_t)nq) {
		ix->par_run(ix->par_ctx, fill_dev_chunk, &J, nq, 32);
	} else if (nq) {
		fill_dev_chunk(&J, 0, nq);
	}
	total_post += J.total_post.load();
	if (J.bad.load()) {
		if (J.bad_term.load()) {
			set_error("query %u: bad term id 0", J.bad.load() - 1);
		} else {
			set_error("query %u exceeds the device limits", J.bad.load() - 1);
		}
		return -1;
	}
	return 0;
}

/* doc-sharded mode: the accepted-candidate log of every query (host arrays) */
struct cand_log_t {
	uint32_t	cap;
	uint64_t *	ids;	/* [nq * cap] */
	float *		sc;	/* [nq * cap] */
	uint32_t *	cnt;	/* [nq]; > cap = overflow */
};

static int
search_impl(nxsgpu_index_t *ix, int algo, uint64_t limit, const nxsgpu_query_t *queries,
    uint32_t nq, nxsgpu_results_t *res, cand_log_t *cl = NULL)
{
	/* the candidate filter pass: limits up to NXSGPU_BIG_K (MODE_BIG beyond 64) */
	const bool fast = limit <= NXSGPU_BIG_K;
	const bool big = fast && limit > NXSGPU_FAST_K;
	const uint32_t seg_cap = !big ? ix->cfg.seg_cap : ix->cfg.seg_cap_big ? ix->cfg.seg_cap_big :
	    (uint32_t)((6 * limit + 1023) & ~1023ull);
	std::vector<dev_query_t> hq(nq);
	std::vector<uint32_t> h_ovf, h_cnt;
	worklist_t wl;
	uint64_t total_post = 0;
	uint8_t *p;
	dev_query_t *d_q;
	qmeta_t *d_qmeta;
	item_t *d_items;
	uint32_t *d_seg_count, *d_cand_doc, *d_ovf, *d_cnt;
	uint64_t *d_ids;
	float *d_cand_sc, *d_sc;
	scan_args_t sa;
	replay_args_t ra;
	const uint32_t kfast = fast ? (uint32_t)limit : NXSGPU_FAST_K;

	if (algo != NXSGPU_BM25 && algo != NXSGPU_TF_IDF) {
		set_error("invalid algorithm");
		return -1;
	}
	if (limit == 0) {
		set_error("invalid limit");
		return -1;
	}
	if (hipSetDevice(ix->device) != hipSuccess) {
		set_error("hipSetDevice failed");
		return -1;
	}
	if (ensure_algo(ix, algo) != 0) {
		return -1;
	}
	if (res) {
		memset(res, 0, sizeof(*res));
		res->n_queries = nq;
	}
	if (nq == 0) {
		return 0;
	}

	uint64_t *d_log_ids = NULL;
	float *d_log_sc = NULL;
	uint32_t *d_log_cnt = NULL, *d_log_slot = NULL;
	struct log_guard_t {
		uint64_t *&a; float *&b; uint32_t *&c; uint32_t *&d;
		~log_guard_t() { (void)hipFree(a); (void)hipFree(b); (void)hipFree(c); (void)hipFree(d); }
	} log_guard{d_log_ids, d_log_sc, d_log_cnt, d_log_slot};
	if (cl) {
		if (hipMalloc((void **)&d_log_ids, (size_t)nq * cl->cap * 8 + 8) != hipSuccess ||
		    hipMalloc((void **)&d_log_sc, (size_t)nq * cl->cap * 4 + 4) != hipSuccess ||
		    hipMalloc((void **)&d_log_cnt, (size_t)nq * 4) != hipSuccess ||
		    hipMalloc((void **)&d_log_slot, (size_t)nq * 4) != hipSuccess ||
		    hipMemsetAsync(d_log_cnt, 0, (size_t)nq * 4, ix->stream) != hipSuccess) {
			set_error("hipMalloc for the candidate log failed");
			return -1;
		}
	}
	/* (this blocking path is also where queries land whose candidate lists
	 * overflowed in a batch: no sparse + dense class here -- its pending list is
	 * what overflows, and the tiles take such a query without emitting every match
	 * as the exact passes below would: 30 ms per query at 50M docs) */
	if (fill_dev_queries(ix, algo, queries, nq, hq.data(), total_post, false) != 0) {
		return -1;
	}
	build_worklist(ix, hq.data(), nq, wl, false, big ? (uint32_t)limit : 0);
	const uint64_t nseg = wl.n_segs;

	/* workspace: queries | meta | items | seg_count | overflow | candidates | outputs */
	{
		size_t need = 8192 + nq * 4 + nq * sizeof(dev_query_t) + nq * sizeof(qmeta_t)
		    + nseg * sizeof(item_t) + nseg * 4 + nq * 4
		    + (nseg + nq) * 4 * (1 + NXSGPU_MAX_TOKENS) + nseg * 4 + 1024
		    + nseg * seg_cap * 8 + (size_t)nq * kfast * 12 + nq * 4 + 16 * 256
		    + nseg * (16 * 4 + 64 * 4) + 1024 + RETRY_LISTS * (4 + RETRY_CAP * sizeof(item_t)) + 1024
		    + (big ? nseg * 32 + 256 : 0);
		if (!ensure_ws(ix, need)) {
			return -1;
		}
	}
	/*
	 * Everything the kernels read from the host is one contiguous block, staged
	 * in pinned memory and uploaded by ONE copy (the two zero-filled arrays
	 * included); the flags and the results are one block and ONE copy back.  A
	 * single query used to pay eleven small pageable copies + two memsets:
	 * most of its latency.
	 */
	p = (uint8_t *)ix->ws;
	uint8_t *const up0 = p;
	d_q = carve<dev_query_t>(p, nq);
	d_qmeta = carve<qmeta_t>(p, nq);
	d_items = carve<item_t>(p, nseg);
	uint32_t *d_bnd_q = carve<uint32_t>(p, nseg + nq);
	uint32_t *d_qorder = carve<uint32_t>(p, nq);
	float *d_pub = carve<float>(p, nseg);
	uint32_t *d_retry_cnt = carve<uint32_t>(p, RETRY_LISTS);
	d_ovf = carve<uint32_t>(p, nq);
	const size_t up_len = (size_t)(p - up0);
	uint8_t *const down0 = (uint8_t *)d_ovf;
	d_ids = carve<uint64_t>(p, (size_t)nq * kfast);
	d_sc = carve<float>(p, (size_t)nq * kfast);
	d_cnt = carve<uint32_t>(p, nq);
	const size_t down_len = (size_t)(p - down0);
	d_seg_count = carve<uint32_t>(p, nseg);
	uint32_t *d_cursors = carve<uint32_t>(p, (nseg + nq) * NXSGPU_MAX_TOKENS);
	d_cand_doc = carve<uint32_t>(p, nseg * seg_cap);
	d_cand_sc = carve<float>(p, nseg * seg_cap);
	uint32_t *d_cold_state = carve<uint32_t>(p, nseg * 16);
	float *d_cold_top = carve<float>(p, nseg * 64);
	item_t *d_retry_items = carve<item_t>(p, RETRY_LISTS * RETRY_CAP);
	float *d_pub_sk = carve<float>(p, big ? nseg * 8 : 0);

	if (!ensure_pin(ix, up_len + down_len + 512)) {
		return -1;
	}
	uint8_t *const h_up = (uint8_t *)ix->h_pin;
	uint8_t *const h_down = (uint8_t *)(((uintptr_t)h_up + up_len + 255) & ~(uintptr_t)255);
	memset(h_up, 0, up_len);
	memcpy(h_up + ((uint8_t *)d_q - up0), hq.data(), nq * sizeof(dev_query_t));
	memcpy(h_up + ((uint8_t *)d_qmeta - up0), wl.qmeta.data(), nq * sizeof(qmeta_t));
	memcpy(h_up + ((uint8_t *)d_items - up0), wl.items.data(), nseg * sizeof(item_t));
	memcpy(h_up + ((uint8_t *)d_bnd_q - up0), wl.bnd_q.data(), (nseg + nq) * 4);
	memcpy(h_up + ((uint8_t *)d_qorder - up0), wl.qorder.data(), nq * 4);
	if (hipMemcpyAsync(up0, h_up, up_len, hipMemcpyHostToDevice, ix->stream) != hipSuccess) {
		set_error("query upload failed");
		return -1;
	}

	memset(&sa, 0, sizeof(sa));
	sa.post = ix->d_post[algo];
	sa.dense_col = ix->d_dense_col[algo];
	sa.dense_stride = ix->n_docs;
	sa.dense_q8 = algo == NXSGPU_BM25 ? ix->d_dense_q8 : NULL;
	sa.dense_q8_stride = ix->dense_q8_stride;
	sa.blkmap = ix->d_blkmap;
	sa.bmrank = ix->d_bmrank;
	sa.bm_words = ix->bm_words;
	sa.queries = d_q;
	sa.n_docs = ix->n_docs;
	sa.qmeta = d_qmeta;
	sa.items = d_items;
	sa.k = kfast;
	sa.seg_cap = seg_cap;
	sa.seg_count = d_seg_count;
	sa.seg_off = NULL;
	sa.cand_doc = d_cand_doc;
	sa.cand_sc = d_cand_sc;
	sa.overflow = d_ovf;
	sa.cursors = d_cursors;
	sa.pub = d_pub;
	sa.cold_state = d_cold_state;
	sa.cold_top = d_cold_top;
	sa.retry_count = d_retry_cnt;
	sa.retry_items = d_retry_items;
	sa.pub_sk = d_pub_sk;
	if (big && hipMemsetAsync(d_pub_sk, 0, nseg * 32, ix->stream) != hipSuccess) {
		set_error("memset failed");
		return -1;
	}

	h_ovf.assign(nq, 0);
	if (fast) {
		if (ix->profiling) (void)hipEventRecord(ix->ev[0], ix->stream);
		launch_cursors(ix, sa, d_bnd_q, (uint32_t)(nseg + nq));
		memset(&ra, 0, sizeof(ra));
		ra.flags = ix->cfg.old_replay ? 1u : 0u;
		ra.qmeta = d_qmeta;
		ra.seg_cap = seg_cap;
		ra.seg_count = d_seg_count;
		ra.cand_doc = d_cand_doc;
		ra.cand_sc = d_cand_sc;
		ra.doc_ids = ix->d_doc_ids;
		ra.k = kfast;
		ra.out_ids = d_ids;
		ra.out_sc = d_sc;
		ra.out_count = d_cnt;
		ra.skip = d_ovf;
		if (cl) {
			ra.log_ids = d_log_ids;
			ra.log_sc = d_log_sc;
			ra.log_cnt = d_log_cnt;
			ra.log_cap = cl->cap;
		}
		if (ix->cfg.one_replay) {
			launch_scan(big ? MODE_BIG : MODE_TOPK, ix, sa, wl);
			if (ix->profiling) (void)hipEventRecord(ix->ev[1], ix->stream);
			nxs_launch_replay(big ? HEAP_LDS : HEAP_REG, nq, big ? (size_t)limit * 8 : 0, ix->stream, ra);
		} else {
			/* (profile: "replay" is then only what the last class's replay
			 * adds after the last scan) */
			launch_scan(big ? MODE_BIG : MODE_TOPK, ix, sa, wl, &ra, d_qorder, ix->profiling ? ix->ev[1] : NULL);
		}
		if (ix->profiling) (void)hipEventRecord(ix->ev[2], ix->stream);
		if (hipGetLastError() != hipSuccess) {
			set_error("kernel launch failed");
			return -1;
		}
		if (hipMemcpyAsync(h_down, down0, down_len, hipMemcpyDeviceToHost, ix->stream) != hipSuccess) {
			set_error("copy failed");
			return -1;
		}
	} else {
		std::fill(h_ovf.begin(), h_ovf.end(), 1u);
	}
	if (hipStreamSynchronize(ix->stream) != hipSuccess) {
		set_error("stream sync failed: %s", hipGetErrorString(hipGetLastError()));
		return -1;
	}

	/* host copy of the fast results */
	std::vector<uint64_t> f_ids;
	std::vector<float> f_sc;
	h_cnt.assign(nq, 0);
	if (fast) {
		f_ids.resize((size_t)nq * kfast);
		f_sc.resize((size_t)nq * kfast);
		memcpy(h_ovf.data(), h_down + ((uint8_t *)d_ovf - down0), nq * 4);
		memcpy(f_ids.data(), h_down + ((uint8_t *)d_ids - down0), f_ids.size() * 8);
		memcpy(f_sc.data(), h_down + ((uint8_t *)d_sc - down0), f_sc.size() * 4);
		memcpy(h_cnt.data(), h_down + ((uint8_t *)d_cnt - down0), nq * 4);
	}
	/* (a re-run beside batches in flight stays out of the per-launch averages: with
	 * it in, one overflowed query per step halved the "kernel_ms" bench.py prints) */
	if (fast && ix->profiling && !nxsgpu_batches_in_flight(ix)) {
		float a = 0, b = 0;
		(void)hipEventElapsedTime(&a, ix->ev[0], ix->ev[1]);
		(void)hipEventElapsedTime(&b, ix->ev[1], ix->ev[2]);
		ix->prof.launches++;
		ix->prof.scan_ms += a;
		ix->prof.replay_ms += b;
		ix->prof.postings += total_post;
	}

	/*
	 * Exact two-pass path for queries that overflowed their candidate
	 * segments or ask for more than NXSGPU_FAST_K results: count matches,
	 * emit them all, replay with the heap in global memory.
	 */
	std::vector<uint32_t> xq;	/* indices of such queries */
	for (uint32_t i = 0; i < nq; i++) {
		if (h_ovf[i]) {
			xq.push_back(i);
		}
	}
	std::vector<uint32_t> x_cnt;
	std::vector<uint64_t> x_off, x_ids;
	std::vector<float> x_sc;
	if (!xq.empty()) {
		const uint32_t nx = (uint32_t)xq.size();
		std::vector<dev_query_t> xhq(nx);
		worklist_t xwl;
		void *xws = NULL;
		uint8_t *xp;
		size_t xneed;
		int rc = -1;

		for (uint32_t j = 0; j < nx; j++) {
			xhq[j] = hq[xq[j]];
		}
		build_worklist(ix, xhq.data(), nx, xwl);
		const uint64_t xseg = xwl.n_segs;
		std::vector<uint32_t> sc_cnt(xseg);
		std::vector<uint64_t> sc_off(xseg + 1, 0), hp_off(nx + 1, 0), o_off(nx + 1, 0);

		/* device copies of the subset's queries / work list */
		void *xmeta = NULL;
		{
			const size_t mneed = 8192 + nx * sizeof(dev_query_t) + nx * sizeof(qmeta_t)
			    + xseg * sizeof(item_t) + xseg * 4
			    + (xseg + nx) * 4 * (1 + NXSGPU_MAX_TOKENS);
			if ((xmeta = xbuf_get(ix, 0, mneed)) == NULL) {
				set_error("hipMalloc(%zu) for the exact pass failed", mneed);
				return -1;
			}
		}
		uint8_t *mp = (uint8_t *)xmeta;
		dev_query_t *dx_q = carve<dev_query_t>(mp, nx);
		qmeta_t *dx_qmeta = carve<qmeta_t>(mp, nx);
		item_t *dx_items = carve<item_t>(mp, xseg);
		uint32_t *dx_seg_count = carve<uint32_t>(mp, xseg);
		uint32_t *dx_bnd_q = carve<uint32_t>(mp, xseg + nx);
		uint32_t *dx_cursors = carve<uint32_t>(mp, (xseg + nx) * NXSGPU_MAX_TOKENS);

		/* every failure of the pass leaves through here: nothing queued may still
		 * reference the buffers, and an oversized one is not kept */
		auto exact_fail = [&]() -> int {
			(void)hipStreamSynchronize(ix->stream);
			(void)hipGetLastError();
			xbuf_put(ix, 0);
			xbuf_put(ix, 1);
			return -1;
		};
		/* pass 1: count */
		if (hipMemcpyAsync(dx_q, xhq.data(), nx * sizeof(dev_query_t), hipMemcpyHostToDevice, ix->stream) != hipSuccess ||
		    hipMemcpyAsync(dx_qmeta, xwl.qmeta.data(), nx * sizeof(qmeta_t), hipMemcpyHostToDevice, ix->stream) != hipSuccess ||
		    hipMemcpyAsync(dx_items, xwl.items.data(), xseg * sizeof(item_t), hipMemcpyHostToDevice, ix->stream) != hipSuccess ||
		    hipMemcpyAsync(dx_bnd_q, xwl.bnd_q.data(), (xseg + nx) * 4, hipMemcpyHostToDevice, ix->stream) != hipSuccess) {
			set_error("query upload failed");
			return exact_fail();
		}
		sa.queries = dx_q;
		sa.qmeta = dx_qmeta;
		sa.items = dx_items;
		sa.seg_count = dx_seg_count;
		sa.cursors = dx_cursors;
		sa.k = 0xffffffffu;
		launch_cursors(ix, sa, dx_bnd_q, (uint32_t)(xseg + nx));
		launch_scan(MODE_COUNT, ix, sa, xwl);
		if (hipMemcpyAsync(sc_cnt.data(), dx_seg_count, xseg * 4, hipMemcpyDeviceToHost, ix->stream) != hipSuccess ||
		    hipStreamSynchronize(ix->stream) != hipSuccess) {
			set_error("count pass failed: %s", hipGetErrorString(hipGetLastError()));
			return exact_fail();
		}
		for (uint64_t sgi = 0; sgi < xseg; sgi++) {
			sc_off[sgi + 1] = sc_off[sgi] + sc_cnt[sgi];
		}
		for (uint32_t j = 0; j < nx; j++) {
			const qmeta_t &m = xwl.qmeta[j];
			const uint64_t matched = sc_off[(uint64_t)m.seg_first + m.n_groups] - sc_off[m.seg_first];
			const uint64_t hcap = std::min<uint64_t>(limit, matched);
			hp_off[j + 1] = hp_off[j] + hcap;
			o_off[j + 1] = o_off[j] + hcap;
		}
		const uint64_t tot_c = sc_off[xseg], tot_o = o_off[nx];
		xneed = 8192 + (xseg + 1) * 8 + tot_c * 8 + tot_o * 8 * 2 + tot_o * 12 + (nx + 1) * 16 + nx * 4;
		if ((xws = xbuf_get(ix, 1, xneed)) == NULL) {
			set_error("hipMalloc(%zu) for the exact pass failed", xneed);
			return exact_fail();
		}
		xp = (uint8_t *)xws;
		uint64_t *dx_seg_off = carve<uint64_t>(xp, xseg + 1);
		uint32_t *dx_cdoc = carve<uint32_t>(xp, tot_c + 1);
		float *dx_csc = carve<float>(xp, tot_c + 1);
		float *dx_hs = carve<float>(xp, tot_o + 1);
		uint32_t *dx_hd = carve<uint32_t>(xp, tot_o + 1);
		uint64_t *dx_hoff = carve<uint64_t>(xp, nx + 1);
		uint64_t *dx_ooff = carve<uint64_t>(xp, nx + 1);
		uint64_t *dx_ids = carve<uint64_t>(xp, tot_o + 1);
		float *dx_sc = carve<float>(xp, tot_o + 1);
		uint32_t *dx_cnt = carve<uint32_t>(xp, nx);

		x_cnt.assign(nx, 0);
		x_ids.resize(tot_o);
		x_sc.resize(tot_o);
		x_off = o_off;
		do {
			if (hipMemcpyAsync(dx_seg_off, sc_off.data(), (xseg + 1) * 8, hipMemcpyHostToDevice, ix->stream) != hipSuccess ||
			    hipMemcpyAsync(dx_hoff, hp_off.data(), (nx + 1) * 8, hipMemcpyHostToDevice, ix->stream) != hipSuccess ||
			    hipMemcpyAsync(dx_ooff, o_off.data(), (nx + 1) * 8, hipMemcpyHostToDevice, ix->stream) != hipSuccess) {
				set_error("upload failed");
				break;
			}
			/* pass 2: emit every match at its exact offset */
			scan_args_t sb = sa;
			sb.seg_off = dx_seg_off;
			sb.cand_doc = dx_cdoc;
			sb.cand_sc = dx_csc;
			launch_scan(MODE_ALL, ix, sb, xwl);
			memset(&ra, 0, sizeof(ra));
			ra.flags = ix->cfg.old_replay ? 1u : 0u;
			ra.qmeta = dx_qmeta;
			ra.seg_cap = 0;
			ra.seg_off = dx_seg_off;
			ra.cand_doc = dx_cdoc;
			ra.cand_sc = dx_csc;
			ra.doc_ids = ix->d_doc_ids;
			ra.k = (uint32_t)std::min<uint64_t>(limit, 0xffffffffu);
			ra.gheap_s = dx_hs;
			ra.gheap_d = dx_hd;
			ra.heap_off = dx_hoff;
			ra.out_ids = dx_ids;
			ra.out_sc = dx_sc;
			ra.out_count = dx_cnt;
			ra.out_off = dx_ooff;
			if (cl) {
				/* row of the log = the query's index in the whole batch */
				if (hipMemcpyAsync(d_log_slot, xq.data(), (size_t)nx * 4, hipMemcpyHostToDevice, ix->stream) != hipSuccess) {
					set_error("upload failed");
					break;
				}
				ra.log_ids = d_log_ids;
				ra.log_sc = d_log_sc;
				ra.log_cnt = d_log_cnt;
				ra.log_cap = cl->cap;
				ra.log_slot = d_log_slot;
			}
			if (ra.k <= REPLAY_LDS_K) {
				nxs_launch_replay(HEAP_LDS, nx, (size_t)ra.k * 8, ix->stream, ra);
			} else {
				nxs_launch_replay(HEAP_GLOBAL, nx, 0, ix->stream, ra);
			}
			if (hipGetLastError() != hipSuccess) {
				set_error("kernel launch failed");
				break;
			}
			if (hipMemcpyAsync(x_cnt.data(), dx_cnt, nx * 4, hipMemcpyDeviceToHost, ix->stream) != hipSuccess ||
			    (tot_o && hipMemcpyAsync(x_ids.data(), dx_ids, tot_o * 8, hipMemcpyDeviceToHost, ix->stream) != hipSuccess) ||
			    (tot_o && hipMemcpyAsync(x_sc.data(), dx_sc, tot_o * 4, hipMemcpyDeviceToHost, ix->stream) != hipSuccess) ||
			    hipStreamSynchronize(ix->stream) != hipSuccess) {
				set_error("exact pass failed: %s", hipGetErrorString(hipGetLastError()));
				break;
			}
			rc = 0;
		} while (0);
		xbuf_put(ix, 0);
		xbuf_put(ix, 1);
		if (rc != 0) {
			return -1;
		}
		if (res) {
			res->exact_requeries = nx;
		}
	}

	if (cl) {
		if (hipMemcpyAsync(cl->ids, d_log_ids, (size_t)nq * cl->cap * 8, hipMemcpyDeviceToHost, ix->stream) != hipSuccess ||
		    hipMemcpyAsync(cl->sc, d_log_sc, (size_t)nq * cl->cap * 4, hipMemcpyDeviceToHost, ix->stream) != hipSuccess ||
		    hipMemcpyAsync(cl->cnt, d_log_cnt, (size_t)nq * 4, hipMemcpyDeviceToHost, ix->stream) != hipSuccess ||
		    hipStreamSynchronize(ix->stream) != hipSuccess) {
			set_error("candidate log copy failed");
			return -1;
		}
	}

	/* assemble host results */
	if (res) {
		uint64_t total = 0;
		res->counts = (uint32_t *)calloc(nq, sizeof(uint32_t));
		res->offsets = (uint64_t *)calloc((size_t)nq + 1, sizeof(uint64_t));
		for (uint32_t i = 0, j = 0; i < nq; i++) {
			uint32_t c;
			if (h_ovf[i]) {
				c = x_cnt[j++];
			} else {
				c = h_cnt[i];
			}
			res->counts[i] = c;
			res->offsets[i + 1] = res->offsets[i] + c;
		}
		total = res->offsets[nq];
		res->doc_ids = (uint64_t *)malloc((total ? total : 1) * 8);
		res->scores = (float *)malloc((total ? total : 1) * 4);
		for (uint32_t i = 0, j = 0; i < nq; i++) {
			const uint64_t o = res->offsets[i];
			const uint32_t c = res->counts[i];
			if (h_ovf[i]) {
				memcpy(res->doc_ids + o, x_ids.data() + x_off[j], c * 8ull);
				memcpy(res->scores + o, x_sc.data() + x_off[j], c * 4ull);
				j++;
			} else {
				memcpy(res->doc_ids + o, f_ids.data() + (size_t)i * kfast, c * 8ull);
				memcpy(res->scores + o, f_sc.data() + (size_t)i * kfast, c * 4ull);
			}
		}
		res->postings = total_post;
	}
	return 0;
}

extern "C" int
nxsgpu_search(nxsgpu_index_t *ix, int algo, uint64_t limit, const nxsgpu_query_t *queries,
    uint32_t nq, nxsgpu_results_t *res)
{
	/* own workspace; with batches in flight, own streams too: a re-run of a few
	 * overflowed queries must not wait for the next batch's scans (19 ms at C5) */
	const bool busy = nxsgpu_batches_in_flight(ix) != 0;
	if (busy) {
		std::swap(ix->stream, ix->xstream[0]);
		std::swap(ix->stream2, ix->xstream[1]);
		std::swap(ix->stream3, ix->xstream[2]);
	}
	const int r = search_impl(ix, algo, limit, queries, nq, res);
	if (busy) {
		std::swap(ix->stream, ix->xstream[0]);
		std::swap(ix->stream2, ix->xstream[1]);
		std::swap(ix->stream3, ix->xstream[2]);
	}
	return r;
}


extern "C" int
nxsgpu_search_candidates(nxsgpu_index_t *ix, int algo, uint64_t limit, const nxsgpu_query_t *queries,
    uint32_t nq, uint32_t cap, uint64_t *ids, float *scores, uint32_t *counts)
{
	cand_log_t cl;
	nxsgpu_results_t res;
	int r;

	if (cap == 0) {
		set_error("nxsgpu_search_candidates: cap is 0");
		return -1;
	}
	cl.cap = cap;
	cl.ids = ids;
	cl.sc = scores;
	cl.cnt = counts;
	memset(counts, 0, (size_t)nq * 4);
	r = search_impl(ix, algo, limit, queries, nq, &res, &cl);
	if (r == 0) {
		nxsgpu_results_free(&res);
	}
	return r;
}

/*
 * Device-resident batches, two in flight.  _begin() plans on the host, stages
 * everything the kernels need in pinned memory, sends it up on its own stream
 * and queues cursors, scans and replays behind it; _end() waits for the oldest
 * batch and reports whether one of its queries overflowed its candidate
 * segments (1: the caller reruns the batch through nxsgpu_search(), which has
 * the exact two-pass path).  While batch i runs, the host prepares and uploads
 * batch i+1.  Outputs must be distinct per batch in flight.
 */
static int
slot_ensure(nxsgpu_index::dev_slot_t &sl, size_t ws_need, size_t stage_need)
{
	if (sl.ws_len < ws_need) {
		(void)hipFree(sl.ws);
		sl.ws = NULL;
		sl.ws_len = 0;
		ws_need = (ws_need + (size_t(1) << 20)) & ~((size_t(1) << 20) - 1);
		if (hipMalloc(&sl.ws, ws_need) != hipSuccess) {
			set_error("hipMalloc(%zu) for the query workspace failed", ws_need);
			return -1;
		}
		sl.ws_len = ws_need;
	}
	if (sl.h_stage_len < stage_need) {
		if (sl.h_stage) {
			(void)hipHostFree(sl.h_stage);
		}
		sl.h_stage = NULL;
		sl.h_stage_len = 0;
		stage_need = (stage_need + (size_t(1) << 20)) & ~((size_t(1) << 20) - 1);
		if (hipHostMalloc((void **)&sl.h_stage, stage_need, hipHostMallocDefault) != hipSuccess) {
			set_error("hipHostMalloc(%zu) failed", stage_need);
			return -1;
		}
		sl.h_stage_len = stage_need;
	}
	return 0;
}

/* what a batch writes its results to */
struct batch_out_t {
	/* caller's device arrays [nq][limit] / [nq] (nxsgpu_search_dev_begin) ... */
	uint64_t *	d_ids;
	float *		d_sc;
	uint32_t *	d_cnt;
	/* ... or record blocks (nxsgpu_batch_begin) */
	bool		records, gather;
	const uint32_t *slot_of_plan;
	const uint32_t *status;
	uint32_t	n_slots;
};

/* a failed _begin must not leave kernels queued over a slot it reports free */
static int
begin_fail(nxsgpu_index_t *ix)
{
	(void)hipStreamSynchronize(ix->stream_up);
	(void)hipStreamSynchronize(ix->stream);
	(void)hipStreamSynchronize(ix->stream2);
	(void)hipStreamSynchronize(ix->stream3);
	(void)hipStreamSynchronize(ix->stream_rp[1]);
	(void)hipStreamSynchronize(ix->stream_down);
	(void)hipGetLastError();
	return -1;
}


static int
batch_begin(nxsgpu_index_t *ix, int algo, uint32_t limit, const nxsgpu_query_t *queries,
    uint32_t nq, const batch_out_t &o)
{
	/* limits > 64 (the API's default is 1000, nxs_impl.h:39): the same pipeline with
	 * the histogram threshold (MODE_BIG), the heap in LDS, larger candidate segments */
	const bool big = limit > NXSGPU_FAST_K;
	const uint32_t seg_cap = !big ? ix->cfg.seg_cap : ix->cfg.seg_cap_big ? ix->cfg.seg_cap_big :
	    (uint32_t)((6 * (uint64_t)limit + 1023) & ~1023ull);
	nxsgpu_index::dev_slot_t *sl = NULL;
	uint64_t total_post = 0;
	const bool gather = o.records && o.gather && ix->comm;
	const uint32_t world = gather ? (uint32_t)nxsgpu_comm_world(ix->comm) : 1u;
	const int my_rank = gather ? nxsgpu_comm_rank(ix->comm) : 0;

	if (limit == 0 || limit > (o.records ? NXSGPU_BIG_K : NXSGPU_FAST_K)) {
		set_error("device batches take limit 1..%d", o.records ? NXSGPU_BIG_K : NXSGPU_FAST_K);
		return -1;
	}
	if (algo != NXSGPU_BM25 && algo != NXSGPU_TF_IDF) {
		set_error("invalid algorithm");
		return -1;
	}
	if (hipSetDevice(ix->device) != hipSuccess) {
		set_error("hipSetDevice failed");
		return -1;
	}
	if (ensure_algo(ix, algo) != 0) {
		return -1;
	}
	for (int i = 0; i < NXSGPU_INFLIGHT; i++) {
		if (!ix->slot[i].active) {
			sl = &ix->slot[i];
			break;
		}
	}
	if (!sl) {
		set_error("%d batches are already in flight", NXSGPU_INFLIGHT);
		return -1;
	}
	if (!sl->wl) {
		sl->wl = new worklist_t();
	}
	worklist_t &wl = *sl->wl;
	auto now_us = []() -> double {
		struct timespec ts;
		clock_gettime(CLOCK_MONOTONIC, &ts);
		return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
	};
	const double tb0 = now_us();
	double tb1 = 0, tb2 = 0, tb3 = 0, tc[6] = { 0, 0, 0, 0, 0, 0 };
	sl->nq = nq;
	sl->postings = 0;
	sl->records = o.records;
	sl->n_slots = o.n_slots;
	sl->k = limit;
	sl->world = world;
	sl->rec_bytes = NXSGPU_REC_BYTES(limit);
	sl->block_bytes = o.records ? NXSGPU_BLOCK_BYTES(o.n_slots, limit) : 0;
	if (nq == 0 && !o.records) {
		sl->seq = ++ix->slot_seq;
		sl->active = true;
		return 0;
	}

	/*
	 * A small batch with nothing else in flight (a single nxs_index_search())
	 * is latency-bound: everything goes down ONE stream -- no cross-stream event
	 * hops, each worth 10-20 us.  Otherwise plans go up and records come down on
	 * their own streams, beside the neighbouring batches' scans.
	 */
	bool others = false;
	for (int i = 0; i < NXSGPU_INFLIGHT; i++) {
		others = others || ix->slot[i].active;
	}
	const bool solo = nq <= 64 && !others && !gather;
	/* (limits > 64: the upload stream's hardware queue carries every third batch's replays --
	 * milliseconds --, so these batches' plans and cursors go up on the scan stream) */
	const bool big_b = limit > NXSGPU_FAST_K;
	hipStream_t s_up = (solo || big_b) ? ix->stream : ix->stream_up;
	/* the records come down on their own stream only when there is a collective
	 * to run beside the next batch's scans; a plain 135 KB copy rides the scan
	 * stream (a separate stream showed sporadic 5-20 ms host stalls in the copy
	 * submission, once or twice per process) */
	const bool own_down = !solo && gather && !ix->cfg.down_inline;
	hipStream_t s_down = own_down ? ix->stream_down : ix->stream;

	/* record blocks: pinned host copies of all ranks' blocks; on the device the
	 * own block is part of the uploaded workspace (one rank), or sits at its rank
	 * position of the all-gather's receive buffer (in-place send) */
	const size_t recs_len = (size_t)o.n_slots * sl->rec_bytes;
	if (o.records) {
		const size_t need = (size_t)world * sl->block_bytes + 256;
		if (gather && sl->d_blocks_len < need) {
			(void)hipFree(sl->d_blocks);
			sl->d_blocks = NULL;
			sl->d_blocks_len = 0;
			if (hipMalloc((void **)&sl->d_blocks, need) != hipSuccess) {
				set_error("hipMalloc(%zu) for the record blocks failed", need);
				return -1;
			}
			sl->d_blocks_len = need;
		}
		if (sl->h_blocks_len < need) {
			if (sl->h_blocks) {
				(void)hipHostFree(sl->h_blocks);
			}
			sl->h_blocks = NULL;
			sl->h_blocks_len = 0;
			if (hipHostMalloc((void **)&sl->h_blocks, need, hipHostMallocMapped) != hipSuccess ||
			    hipHostGetDevicePointer((void **)&sl->h_blocks_dev, sl->h_blocks, 0) != hipSuccess) {
				set_error("hipHostMalloc(%zu) failed", need);
				return -1;
			}
			sl->h_blocks_len = need;
		}
	}
	/* (a communicator of ONE rank still goes through the collective: the same
	 * code path as N ranks, and what the one-GPU tests exercise) */
	/*
	 * One rank, no collective: the heap replay writes the records STRAIGHT into
	 * the pinned host block (mapped into the device's address space) -- 135 KB of
	 * posted PCIe writes per batch instead of a copy command after the kernels
	 * (whose submission showed sporadic 5-20 ms host stalls).  The host zeroes the
	 * block and fills the status words itself before the launch.
	 */
	/* (the block as part of the batch's one upload -- zeroed and filled on the host, the all-gather's send buffer where
	 * it lies in the workspace: measured for sharded batches, 770k -> 650k queries/s, `_begin` 0.3 ms longer -- not the
	 * host's writes (pinned memory zeroes at 117 GB/s here, like pageable); the device-side memset stays) */
	const bool block_in_ws = false;
	const bool block_on_host = o.records && !gather;

	/* plans straight into the pinned staging area (room for the work list:
	 * <= target + nq ranges, see build_worklist) */
	const uint64_t wave_target = 4 * std::max(ix->cfg.wave_target, ix->cfg.wave_target_scans);	/* (build_worklist: target_eff) */
	const size_t seg_bound = (size_t)wave_target + 2 * (size_t)nq + 64;
	const size_t stage_need = 32768 + RETRY_LISTS * 4 + 256 + nq * (sizeof(dev_query_t) + sizeof(qmeta_t) + 16)
	    + seg_bound * (sizeof(item_t) + 8) + nq * 4 + NXSGPU_STATUS_WORDS(o.n_slots) * 4 + 4096
	    + (block_in_ws ? sl->block_bytes : 0);	/* (the record block itself lives in h_blocks) */
	if (slot_ensure(*sl, 0, stage_need) != 0) {
		return -1;
	}
	uint8_t *hp = sl->h_stage;
	dev_query_t *h_q = carve<dev_query_t>(hp, nq);
	if (fill_dev_queries(ix, algo, queries, nq, h_q, total_post, !big) != 0) {
		return -1;
	}
	const double tb_fill = now_us();
	build_worklist(ix, h_q, nq, wl, solo, big ? limit : 0);
	tb1 = now_us();
	const uint64_t nseg = wl.n_segs;
	if (nseg > seg_bound) {
		set_error("work list larger than its bound (%llu > %zu)", (unsigned long long)nseg, seg_bound);
		return -1;
	}
	/*
	 * Everything the kernels read from the host -- the zero-filled flag, threshold
	 * and record arrays included -- is ONE block and ONE copy up.
	 */
	qmeta_t *h_qmeta = carve<qmeta_t>(hp, nq);
	item_t *h_items = carve<item_t>(hp, nseg);
	uint32_t *h_bnd_q = carve<uint32_t>(hp, nseg + nq);
	uint32_t *h_qorder = carve<uint32_t>(hp, nq);
	uint32_t *h_recslot = carve<uint32_t>(hp, nq);
	uint32_t *h_ovf = carve<uint32_t>(hp, nq);
	float *h_pub = carve<float>(hp, nseg);
	uint32_t *h_retry_cnt = carve<uint32_t>(hp, RETRY_LISTS);
	uint8_t *h_block = carve<uint8_t>(hp, block_in_ws ? sl->block_bytes : 0);
	const size_t up_len = (size_t)(hp - sl->h_stage);
	uint32_t *h_status = block_in_ws ? (uint32_t *)(h_block + recs_len) : carve<uint32_t>(hp, NXSGPU_STATUS_WORDS(o.n_slots));
	if ((size_t)(hp - sl->h_stage) > sl->h_stage_len) {
		set_error("staging area too small (%zu > %zu)", (size_t)(hp - sl->h_stage), sl->h_stage_len);
		return -1;
	}
	if (nq) {
		memcpy(h_qmeta, wl.qmeta.data(), nq * sizeof(qmeta_t));
		memcpy(h_items, wl.items.data(), nseg * sizeof(item_t));
		memcpy(h_bnd_q, wl.bnd_q.data(), (nseg + nq) * 4);
		memcpy(h_qorder, wl.qorder.data(), nq * 4);
		memset(h_ovf, 0, nq * 4);
		memset(h_pub, 0, nseg * 4);
	}
	memset(h_retry_cnt, 0, RETRY_LISTS * 4);
	sl->h_ovf = h_ovf;
	if (o.records) {
		for (uint32_t i = 0; i < nq; i++) {
			if (o.slot_of_plan[i] >= o.n_slots) {
				set_error("plan %u: record slot %u out of range", i, o.slot_of_plan[i]);
				return -1;
			}
			h_recslot[i] = o.slot_of_plan[i];
		}
		if (block_in_ws) {
			memset(h_block, 0, sl->block_bytes);
		}
		if (block_on_host) {
			/* (the slot's previous batch was collected: nothing reads it any more) */
			if (sl->rec_bytes <= 1024) {
				memset(sl->h_blocks, 0, recs_len);
			} else {
				/* large records (12 KB at the default limit): count and flags only --
				 * nothing reads ids or scores beyond `count` */
				for (uint32_t i = 0; i < o.n_slots; i++) {
					*(uint64_t *)(sl->h_blocks + (size_t)i * sl->rec_bytes) = 0;
				}
			}
			h_status = (uint32_t *)(sl->h_blocks + recs_len);
		}
		if (o.status) {
			memcpy(h_status, o.status, NXSGPU_STATUS_WORDS(o.n_slots) * 4);
		} else {
			memset(h_status, 0, NXSGPU_STATUS_WORDS(o.n_slots) * 4);
		}
	}

	/* device workspace: the uploaded block first (same carve sequence => same
	 * offsets), then what only the kernels touch */
	const size_t ws_need = 32768 + up_len + nseg * 4
	    + (nseg + nq) * 4 * NXSGPU_MAX_TOKENS + nseg * (size_t)seg_cap * 8 + nseg * (16 * 4 + 64 * 4)
	    + RETRY_LISTS * RETRY_CAP * sizeof(item_t) + 1024 + (big ? nseg * 32 + 256 : 0);
	if (slot_ensure(*sl, ws_need, 0) != 0) {
		return -1;
	}
	uint8_t *p = (uint8_t *)sl->ws;
	dev_query_t *d_q = carve<dev_query_t>(p, nq);
	qmeta_t *d_qmeta = carve<qmeta_t>(p, nq);
	item_t *d_items = carve<item_t>(p, nseg);
	uint32_t *d_bnd_q = carve<uint32_t>(p, nseg + nq);
	uint32_t *d_qorder = carve<uint32_t>(p, nq);
	uint32_t *d_recslot = carve<uint32_t>(p, nq);
	uint32_t *d_ovf = carve<uint32_t>(p, nq);
	float *d_pub = carve<float>(p, nseg);
	uint32_t *d_retry_cnt = carve<uint32_t>(p, RETRY_LISTS);
	uint8_t *d_myblock = carve<uint8_t>(p, block_in_ws ? sl->block_bytes : 0);
	uint32_t *d_seg_count = carve<uint32_t>(p, nseg);
	uint32_t *d_cursors = carve<uint32_t>(p, (nseg + nq) * NXSGPU_MAX_TOKENS);
	uint32_t *d_cand_doc = carve<uint32_t>(p, nseg * (size_t)seg_cap);
	float *d_cand_sc = carve<float>(p, nseg * (size_t)seg_cap);
	uint32_t *d_cold_state = carve<uint32_t>(p, nseg * 16);
	float *d_cold_top = carve<float>(p, nseg * 64);
	item_t *d_retry_items = carve<item_t>(p, RETRY_LISTS * RETRY_CAP);
	float *d_pub_sk = carve<float>(p, big ? nseg * 8 : 0);
	if (block_on_host) {
		d_myblock = sl->h_blocks_dev;
	} else if (o.records && !block_in_ws) {
		d_myblock = sl->d_blocks + (size_t)my_rank * sl->block_bytes;
	}

	sl->seq = ++ix->slot_seq;
	tb2 = now_us();
	if (hipMemcpyAsync(sl->ws, sl->h_stage, up_len, hipMemcpyHostToDevice, s_up) != hipSuccess) {
		set_error("query upload failed");
		return begin_fail(ix);
	}
	if (o.records && !block_in_ws && !block_on_host) {
		if ((recs_len && hipMemsetAsync(d_myblock, 0, recs_len, s_up) != hipSuccess) ||
		    hipMemcpyAsync(d_myblock + recs_len, h_status, NXSGPU_STATUS_WORDS(o.n_slots) * 4,
		    hipMemcpyHostToDevice, s_up) != hipSuccess) {
			set_error("record block setup failed");
			return begin_fail(ix);
		}
	}

	tc[0] = now_us();
	scan_args_t sa;
	replay_args_t ra;
	memset(&sa, 0, sizeof(sa));
	sa.post = ix->d_post[algo];
	sa.dense_col = ix->d_dense_col[algo];
	sa.dense_stride = ix->n_docs;
	sa.dense_q8 = algo == NXSGPU_BM25 ? ix->d_dense_q8 : NULL;
	sa.dense_q8_stride = ix->dense_q8_stride;
	sa.blkmap = ix->d_blkmap;
	sa.bmrank = ix->d_bmrank;
	sa.bm_words = ix->bm_words;
	sa.queries = d_q;
	sa.n_docs = ix->n_docs;
	sa.qmeta = d_qmeta;
	sa.items = d_items;
	sa.k = limit;
	sa.seg_cap = seg_cap;
	sa.seg_count = d_seg_count;
	sa.cand_doc = d_cand_doc;
	sa.cand_sc = d_cand_sc;
	sa.overflow = d_ovf;
	sa.cursors = d_cursors;
	sa.pub = d_pub;
	sa.cold_state = d_cold_state;
	sa.cold_top = d_cold_top;
	sa.retry_count = d_retry_cnt;
	sa.retry_items = d_retry_items;
	sa.pub_sk = d_pub_sk;
	if (big && nseg && hipMemsetAsync(d_pub_sk, 0, nseg * 32, s_up) != hipSuccess) {
		set_error("memset failed");
		return begin_fail(ix);
	}
	memset(&ra, 0, sizeof(ra));
	ra.flags = ix->cfg.old_replay ? 1u : 0u;
	ra.qmeta = d_qmeta;
	ra.seg_cap = seg_cap;
	ra.seg_count = d_seg_count;
	ra.cand_doc = d_cand_doc;
	ra.cand_sc = d_cand_sc;
	ra.doc_ids = ix->d_doc_ids;
	ra.k = limit;
	ra.out_ids = o.d_ids;
	ra.out_sc = o.d_sc;
	ra.out_count = o.d_cnt;
	ra.skip = d_ovf;
	if (o.records) {
		ra.rec_base = d_myblock;
		ra.rec_slot = d_recslot;
		ra.rec_bytes = (uint32_t)sl->rec_bytes;
	}

	/*
	 * The range cursors depend on the uploaded plans only: k_cursors (a small,
	 * latency-bound grid of binary searches) runs on the upload stream, beside
	 * the previous batch's scans instead of in front of this batch's.
	 */
	if (nq && wl.need_cursors) {
		launch_cursors(ix, sa, d_bnd_q, (uint32_t)(nseg + nq), s_up);
	}
	tc[1] = now_us();
	if (s_up != ix->stream && (hipEventRecord(sl->ev_up, s_up) != hipSuccess ||
	    hipStreamWaitEvent(ix->stream, sl->ev_up, 0) != hipSuccess)) {
		set_error("query upload failed");
		return begin_fail(ix);
	}
	/*
	 * The top doc ranges of the sparse + dense class (build_worklist: a launch of their own,
	 * one wavefront per query -- 0.3-0.4 ms of pure latency: a cold phase, then a whole
	 * range on a cold threshold) depend on the plans and the cursors only: they go to the
	 * upload stream right here, i.e. beside the PREVIOUS batch's scans; when this batch's
	 * turn comes their thresholds are published and the class's other ranges start warm.
	 */
	sl->ahead = false;
	if (!solo && nq && !big && limit <= WAVE && ix->cfg.drop_early && ix->cfg.drop_side && !ix->cfg.drop_tiles &&
	    !ix->cfg.one_replay) {
		size_t n_l = 0;
		for (const launch_t &l : wl.launches) {
			n_l += l.count != 0;
		}
		for (launch_t &l : wl.launches) {
			if ((l.kind == 5 || l.kind == 9) && l.q_count == 0 && l.count && n_l > 2) {
				scan_args_t a = sa;
				a.item_base = l.first;
				a.flags |= ix->cfg.drop_prio ? 1u : 0u;
				a.flags |= (ix->cfg.drop_b && l.nt_bucket <= 5) ? 8u : 0u;
				a.retry_count = NULL;
				a.retry_items = NULL;
				a.retry_cap = 0;
				nxs_launch_drop_class(l.nt_bucket, l.count, s_up, a);
				l.count = 0;		/* (launch_scan skips it) */
				sl->ahead = true;
			}
		}
		if (sl->ahead && hipEventRecord(sl->ev_ahead, s_up) != hipSuccess) {
			set_error("hipEventRecord failed");
			return begin_fail(ix);
		}
	}
	tc[2] = now_us();
	/*
	 * MODE_BIG, one rank: the replays (milliseconds: thousands of heap insertions
	 * per query on one lane) all run on the second stream and the scan stream does
	 * not wait for them, so the NEXT batch's scans AND replays run beside them (a
	 * replay stream per batch slot); the batch ends when its replay stream has.  With a collective behind the replays the usual join stays.
	 */
	/* (limits <= 64: only where the tail is a visible share of the step -- short batches; a C5 batch scans
	 * for 35 ms, its host side is nearly as long, and ending it on the replay stream cost 7 %) */
	const bool short_batch = total_post < (1ull << 32);
	/* (sharded batches too: the all-gather, on its own stream, waits for the batch's replay stream instead of
	 * the scan stream waiting for the replays; MODE_BIG: the record stream shares hardware queue D with stream_rp[1], which
	 * these batches leave alone -- a collective queued behind another batch's 7 ms replay would end its batch late) */
	const bool aside = (big || (o.records && short_batch && !ix->cfg.replay_join)) && nq && (!gather || own_down) &&
	    !solo && !ix->cfg.one_replay;
	/* (limits <= 64: the replays of both slots share the replay stream -- they are short, and
	 * stream_rp[0] is the dense-term class's stream) */
	hipStream_t s_end = !aside ? ix->stream : big ? ix->stream_rp[gather ? (sl->seq & 1) * 2 : sl->seq % 3] : ix->stream2;
	if (ix->profiling) (void)hipEventRecord(sl->ev_t[0], ix->stream);
	sl->n_cls = 0;
	if (nq) {
		if (ix->cfg.one_replay) {
			launch_scan(big ? MODE_BIG : MODE_TOPK, ix, sa, wl);
			if (ix->profiling) (void)hipEventRecord(sl->ev_t[1], ix->stream);
			nxs_launch_replay(big ? HEAP_LDS : HEAP_REG, nq, big ? (size_t)limit * 8 : 0, ix->stream, ra);
		} else {
			/* (profile: "replay" is then only what the last class's replay adds
			 * after the last scan) */
			sl->n_cls = 0;
			launch_scan(big ? MODE_BIG : MODE_TOPK, ix, sa, wl, &ra, d_qorder, ix->profiling ? sl->ev_t[1] : NULL, aside, aside ? s_end : NULL,
			    ix->profiling ? sl : NULL, sl->ahead ? sl->ev_ahead : NULL,
			    (!solo && ix->cfg.and_early) ? s_up : NULL, sl->ev_early);
		}
	} else if (ix->profiling) {
		(void)hipEventRecord(sl->ev_t[1], ix->stream);
	}
	if (ix->profiling) (void)hipEventRecord(sl->ev_t[2], s_end);
	tc[3] = now_us();
	if (hipGetLastError() != hipSuccess) {
		set_error("kernel launch failed");
		return begin_fail(ix);
	}
	if (!o.records) {
		if (hipMemcpyAsync(h_ovf, d_ovf, nq * 4, hipMemcpyDeviceToHost, ix->stream) != hipSuccess ||
		    hipEventRecord(sl->ev_done, ix->stream) != hipSuccess) {
			set_error("copy failed");
			return begin_fail(ix);
		}
	} else {
		/*
		 * The records leave on their own stream: the all-gather (one collective
		 * per batch, sharded runs only) and the copy to pinned memory overlap the
		 * next batch's scans instead of sitting in front of them.
		 */
		if (own_down && (hipEventRecord(sl->ev_res, aside ? s_end : ix->stream) != hipSuccess ||
		    hipStreamWaitEvent(s_down, sl->ev_res, 0) != hipSuccess)) {
			set_error("event failed");
			return begin_fail(ix);
		}
		if (gather) {
			if (comm_allgather_dev(ix->comm, d_myblock, sl->d_blocks, sl->block_bytes, s_down) != 0) {
				return begin_fail(ix);
			}
			/* (a kernel, not a copy command: the gathered blocks go to the mapped pinned buffer as posted
			 * writes.  hipMemcpyAsync on this stream stalled the HOST for ~7 ms once or twice per run
			 * -- one batch in eight at worst -- which was the whole gap between a sharded and a plain step) */
			const uint64_t n8 = ((uint64_t)world * sl->block_bytes) / 8;
			hipLaunchKernelGGL(k_records_out, dim3((unsigned)std::min<uint64_t>((n8 + 255) / 256, 2048)), dim3(256), 0, s_down,
			    (const uint64_t *)sl->d_blocks, (uint64_t *)sl->h_blocks_dev, n8);
			if (hipGetLastError() != hipSuccess) {
				set_error("copy failed");
				return begin_fail(ix);
			}
		} else if (!block_on_host && sl->block_bytes && hipMemcpyAsync(sl->h_blocks, d_myblock, sl->block_bytes,
		    hipMemcpyDeviceToHost, s_down) != hipSuccess) {
			set_error("copy failed");
			return begin_fail(ix);
		}
		if (hipEventRecord(sl->ev_done, (aside && !gather) ? s_end : s_down) != hipSuccess) {
			set_error("event failed");
			return begin_fail(ix);
		}
	}
	sl->postings = total_post;
	sl->active = true;
	if (ix->cfg.debug_timing) {
		tb3 = now_us();
		fprintf(stderr, "[nxsgpu begin #%llu] (fill %.0f us) plan+worklist %.0f us, staging+alloc %.0f us, enqueue %.0f us "
		    "(upload %.0f, cursors %.0f, fork %.0f, scans+replays %.0f, tail %.0f)\n",
		    (unsigned long long)sl->seq, tb_fill - tb0, tb1 - tb0, tb2 - tb1, tb3 - tb2,
		    tc[0] - tb2, tc[1] - tc[0], tc[2] - tc[1], tc[3] - tc[2], tb3 - tc[3]);
	}
	return 0;
}

extern "C" int
nxsgpu_search_dev_begin(nxsgpu_index_t *ix, int algo, uint32_t limit, const nxsgpu_query_t *queries,
    uint32_t nq, uint64_t *d_doc_ids, float *d_scores, uint32_t *d_counts)
{
	batch_out_t o;

	memset(&o, 0, sizeof(o));
	if (!d_doc_ids || !d_scores || !d_counts) {
		set_error("nxsgpu_search_dev: outputs must be non-NULL");
		return -1;
	}
	o.d_ids = d_doc_ids;
	o.d_sc = d_scores;
	o.d_cnt = d_counts;
	return batch_begin(ix, algo, limit, queries, nq, o);
}

extern "C" int
nxsgpu_batch_begin(nxsgpu_index_t *ix, int algo, uint32_t limit, const nxsgpu_query_t *plans,
    uint32_t n_plans, const uint32_t *slot_of_plan, const uint32_t *status, uint32_t n_slots,
    int gather)
{
	batch_out_t o;

	memset(&o, 0, sizeof(o));
	if (n_plans && !slot_of_plan) {
		set_error("nxsgpu_batch_begin: slot_of_plan is NULL");
		return -1;
	}
	o.records = true;
	o.gather = gather != 0;
	o.slot_of_plan = slot_of_plan;
	o.status = status;
	o.n_slots = n_slots;
	return batch_begin(ix, algo, limit, plans, n_plans, o);
}

static nxsgpu_index::dev_slot_t *
oldest_slot(nxsgpu_index_t *ix)
{
	nxsgpu_index::dev_slot_t *sl = NULL;

	for (int i = 0; i < NXSGPU_INFLIGHT; i++) {
		if (ix->slot[i].active && (!sl || ix->slot[i].seq < sl->seq)) {
			sl = &ix->slot[i];
		}
	}
	return sl;
}

static int
slot_wait(nxsgpu_index_t *ix, nxsgpu_index::dev_slot_t *sl)
{
	if (hipEventSynchronize(sl->ev_done) != hipSuccess) {
		set_error("batch failed: %s", hipGetErrorString(hipGetLastError()));
		return -1;
	}
	if (ix->profiling && (sl->nq || sl->records)) {
		float a = 0, b = 0;
		(void)hipEventElapsedTime(&a, sl->ev_t[0], sl->ev_t[1]);
		(void)hipEventElapsedTime(&b, sl->ev_t[1], sl->ev_t[2]);
		ix->prof.launches++;
		ix->prof.scan_ms += a;
		ix->prof.replay_ms += b;
		ix->prof.postings += sl->postings;
		/* per class (the events sit on the classes' own streams; the batch is done: all have fired) */
		for (uint32_t c = 0; c < sl->n_cls; c++) {
			float ms = 0;
			uint32_t k = 0;
			if (hipEventElapsedTime(&ms, sl->ev_cls[c][0], sl->ev_cls[c][1]) != hipSuccess) {
				continue;
			}
			while (k < ix->prof.n_cls && ix->prof.cls_key[k] != sl->cls_key[c]) {
				k++;
			}
			if (k == ix->prof.n_cls) {
				if (k == NXSGPU_PROF_CLS) {
					continue;
				}
				ix->prof.cls_key[k] = sl->cls_key[c];
				ix->prof.n_cls++;
			}
			ix->prof.cls_launches[k]++;
			ix->prof.cls_ms[k] += ms;
			ix->prof.cls_postings[k] += sl->cls_post[c];
			ix->prof.cls_queries[k] += sl->cls_q[c];
		}
	}
	return 0;
}

extern "C" int
nxsgpu_search_dev_end(nxsgpu_index_t *ix)
{
	nxsgpu_index::dev_slot_t *sl = oldest_slot(ix);

	if (!sl) {
		set_error("nxsgpu_search_dev_end: no batch in flight");
		return -1;
	}
	if (sl->records) {
		set_error("nxsgpu_search_dev_end: the oldest batch in flight is a record batch (nxsgpu_batch_end)");
		return -1;
	}
	sl->active = false;
	if (sl->nq == 0) {
		return 0;
	}
	if (slot_wait(ix, sl) != 0) {
		return -1;
	}
	const uint32_t *h_ovf = sl->h_ovf;
	for (uint32_t i = 0; i < sl->nq; i++) {
		if (h_ovf[i]) {
			return 1;
		}
	}
	return 0;
}

extern "C" int
nxsgpu_batch_end(nxsgpu_index_t *ix, nxsgpu_batch_view_t *view)
{
	nxsgpu_index::dev_slot_t *sl = oldest_slot(ix);

	if (!sl) {
		set_error("nxsgpu_batch_end: no batch in flight");
		return -1;
	}
	if (!sl->records) {
		set_error("nxsgpu_batch_end: the oldest batch in flight is a device batch (nxsgpu_search_dev_end)");
		return -1;
	}
	sl->active = false;
	if (slot_wait(ix, sl) != 0) {
		return -1;
	}
	view->n_slots = sl->n_slots;
	view->k = sl->k;
	view->world = sl->world;
	view->rec_bytes = sl->rec_bytes;
	view->block_bytes = sl->block_bytes;
	/* with one rank the own block sits at position 0 of both copies */
	view->blocks = sl->h_blocks;
	return 0;
}

extern "C" int
nxsgpu_batches_in_flight(const nxsgpu_index_t *ix)
{
	int n = 0;

	for (int i = 0; i < NXSGPU_INFLIGHT; i++) {
		n += ix->slot[i].active ? 1 : 0;
	}
	return n;
}

extern "C" void
nxsgpu_index_set_parallel(nxsgpu_index_t *ix, nxsgpu_parallel_t run, void *ctx)
{
	ix->par_run = run;
	ix->par_ctx = ctx;
}

extern "C" void
nxsgpu_index_reconfigure(nxsgpu_index_t *ix)
{
	cfg_from_env(ix->cfg);
	if (ix->down_probe < 0) {
		ix->cfg.down_inline = true;	/* (pick_record_stream found no stream of its own for the records) */
	}
}

extern "C" int
nxsgpu_search_dev(nxsgpu_index_t *ix, int algo, uint32_t limit, const nxsgpu_query_t *queries,
    uint32_t nq, uint64_t *d_doc_ids, float *d_scores, uint32_t *d_counts)
{
	if (nxsgpu_batches_in_flight(ix)) {
		set_error("nxsgpu_search_dev: finish the batches in flight first (nxsgpu_search_dev_end)");
		return -1;
	}
	if (nxsgpu_search_dev_begin(ix, algo, limit, queries, nq, d_doc_ids, d_scores, d_counts) != 0) {
		return -1;
	}
	return nxsgpu_search_dev_end(ix);
}

/* ---- query sharding: slices and the RCCL communicator ---------------------------- */

extern "C" void
nxsgpu_shard_slice(uint64_t n, int rank, int world, uint64_t *lo, uint64_t *hi)
{
	if (world < 1) {
		world = 1;
	}
	*lo = n * (uint64_t)rank / (uint64_t)world;
	*hi = n * ((uint64_t)rank + 1) / (uint64_t)world;
}

extern "C" uint64_t
nxsgpu_shard_capacity(uint64_t n, int world)
{
	uint64_t cap = 0, lo, hi;

	for (int r = 0; r < (world < 1 ? 1 : world); r++) {
		nxsgpu_shard_slice(n, r, world, &lo, &hi);
		cap = std::max(cap, hi - lo);
	}
	return cap;
}

/*
 * nxs_gpu_scan_mask.hip -- mask path: k_scanm (quantised byte bounds), k_cold (cold phase of the sparse + dense class)
 * (MI355X / gfx950 query path of nxsearch; see nxs_gpu_int.h for the map of the files)
 */
#include "nxs_gpu_int.h"
#include "nxs_gpu_dev.h"

/*
 * k_scanm: pure-OR queries of sparse terms ("mask path").
 *
 * Why: k_scan8 pays a fixed price per (tile, term) visit, and a tile is only
 * 2048 docs wide because every doc needs an f32 accumulator in LDS.  A term
 * of rank 100..1000 has 5-50 postings per such tile: most lanes of a visit are
 * idle and a 64-posting window is visited in 2-14 tiles (measured: 0.7-1.2 TB/s
 * on all-sparse queries against 6.2 TB/s on all-dense ones).  But once a
 * candidate threshold exists, almost no doc of a sparse OR needs its sum.
 *
 * LDS holds one BYTE per doc (a tile is 4x wider for the same LDS): a
 * quantised upper bound of the doc's score so far.  A visit is one
 * fire-and-forget ds_add_rtn_u32 per window (integer DS atomics run at full
 * rate -- tools/lds_probe.hip -- unlike ds_add_f32); the old byte it returns
 * plus the posting's own quantised impact, compared with the quantised
 * threshold, is a necessary condition for "score > thr" (see `Quantisation`
 * below), for docs holding one term as for docs holding several.  Docs that
 * pass go to a pending list; after the tile they are sorted (descending doc),
 * deduplicated and scored EXACTLY from the register windows (see `flush`):
 * impacts added in token order from 0.0f, what the accumulator tile does.
 * They then take the common threshold filter; everything emitted carries its
 * exact score, in descending doc order, and every doc whose score beats the
 * heap root at its turn is emitted, so k_replay sees a superset in the right
 * order exactly as with the other scan kernels.
 *
 * Cold start: with thr = -inf every posting passes.  The tile width adapts: it
 * starts at 64 docs and doubles while a tile yields few candidates (halves
 * when it yields many), up to MT_W.  A tile that overflows the pending list
 * flags the query for the exact two-pass path.
 */
#ifndef MT_W
#define	MT_W		8192		/* max docs per mask tile (1 byte each) */
#endif
#ifndef MT_W_DROP
#define	MT_W_DROP	8192		/* ... of the sparse + dense class (k_scanm<.., DROP>), which runs beside the plain class
					 * on a stream of its own: 6144 (21 instead of 17 wavefronts per CU) ends IT 5 % sooner
					 * and the plain class 12 % later -- C3 step 1.24 -> 1.33 ms, measured in one session */
#endif
#define	MT_W0		64		/* cold-start tile width */
#ifndef MT_W_HINTED
#define	MT_W_HINTED	2048		/* first tile width when a higher range has published a threshold
					 * (1024 / 2048 / 8192 measured equal; a weak hint then costs two
					 * small tiles, not a pending-list overflow) */
#endif
#define	PEND_CAP	128
#ifndef DROP_PEND_MULT
#define	DROP_PEND_MULT	1		/* k_scanm<.., DROP>: pending list x1 (x2: -8 %, x4: -25 % on C3) */
#endif
#define	PEND_FLUSH	32		/* score the pending docs once this many wait */
/*
 * MT_FOLD = 1: two neighbouring docs share a byte -- a tile covers twice the docs
 * for the same LDS, at half the quantisation range (both docs' bounds must fit the
 * byte together).  A byte is then an upper bound of EITHER doc's score bound: what
 * the sibling adds can only make a doc a candidate that is not one (its exact
 * score decides, as for every candidate), never hide one.
 */
#ifndef MT_FOLD
#define	MT_FOLD		0
#endif
#ifndef MT_FOLD_DROP
#define	MT_FOLD_DROP	MT_FOLD		/* the same for the sparse + dense class */
#endif
#define	MT_DOCS		(MT_W << MT_FOLD)	/* docs per tile */
/* quantised score bound of a doc holding every term at its largest impact (+ 2 per term: <= 240, or 2 x 124) */
#define	QSUM_MAX_F(f)	((f) ? 108 : 224)

#ifdef NXS_STATS
/* diagnostic build only (make variant XFLAGS=-DNXS_STATS): k_scanm event counts
 * and cycle spans, read back with nxsgpu_debug_stats() */
__device__ unsigned long long g_stats[16];
#define	STAT_ADD(i, v)	do { if (lane == 0) atomicAdd(&g_stats[i], (unsigned long long)(v)); } while (0)
#define	STAT_CLK()	((unsigned long long)__builtin_amdgcn_s_memtime())
extern "C" void
nxsgpu_debug_stats(unsigned long long *out, int reset)
{
	unsigned long long z[16] = { 0 };
	(void)hipDeviceSynchronize();
	(void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stats), sizeof(z));
	if (reset) {
		(void)hipMemcpyToSymbol(HIP_SYMBOL(g_stats), z, sizeof(z));
	}
}
#else
#define	STAT_ADD(i, v)	do { } while (0)
#define	STAT_CLK()	0ull
#endif

/* No min-waves launch bound on kernels that own AGPRs by name (bpair_*): under
 * register pressure the compiler would spill VGPRs into accumulation registers,
 * possibly the ones with a prefetch in flight.  tests check .agpr_count. */
/*
 * DROP (queries that mix sparse terms with a DENSE one -- a list holding 8 % of
 * the docs or more): MaxScore's "non-essential lists", kept exact.  A doc that
 * holds only dense terms scores at most U = the token-order f32 sum of their
 * largest impacts (f32 rounding is monotone, so the bound survives the
 * reference's own summation).  Once the candidate threshold reaches U no such
 * doc can be emitted any more (a candidate needs score > thr), and thr never
 * falls: from then on the dense lists LEAVE the scan -- they are not streamed at
 * all.  A doc with sparse terms enters the pending list if its byte bound plus
 * the quantised U can beat the threshold, and its exact score takes the dense
 * terms' impacts from the lists by a 64-ary search (three dependent loads), in
 * token order like every other term.  Until the threshold gets there (cold
 * start: the first few hundred docs of a range whose higher ranges have not
 * published yet) the dense terms are scanned like any other; the wavefront
 * publishes its threshold the moment it drops them, so lower ranges start warm.
 */
template <int NT, bool GEN, bool DROP = false>	/* GEN: the expression is more than an OR: check the truth table */
__global__ void __launch_bounds__(WAVE)
k_scanm(const scan_args_t A)
{
	constexpr int RING = SCANM_RING;
	constexpr uint32_t MTW = DROP ? MT_W_DROP : MT_W;	/* bytes of the map */
	constexpr int FOLD = DROP ? MT_FOLD_DROP : MT_FOLD;
	constexpr uint32_t MTDOCS = MTW << FOLD;
	constexpr int QSUM_MAX = QSUM_MAX_F(FOLD);
	__shared__ __attribute__((aligned(16))) uint32_t s_mask[MTW / 4 + WAVE];	/* + one dummy word per lane */
	/* (DROP pushes on a ceiling and refines in parallel: a longer list, so that a
	 * burst of pushes does not send the query to the exact two-pass path) */
	constexpr uint32_t PCAP = DROP ? DROP_PEND_MULT * PEND_CAP : PEND_CAP;
	__shared__ uint32_t s_pend[PCAP];
	__shared__ uint32_t s_psum[DROP ? PCAP : 1];	/* DROP: the byte bound a doc was pushed with */
	__shared__ uint32_t s_truth[GEN ? 8 : 1];	/* which presence masks match the expression */

	const unsigned lane = threadIdx.x;
	const unsigned long long clk0 = STAT_CLK();
	(void)clk0;
	if constexpr (DROP) {
		/* few, latency-bound wavefronts beside the throughput-bound classes on
		 * the other stream: let the CU's arbiter prefer them */
		if (A.flags & 1) {
			__builtin_amdgcn_s_setprio(3);
		}
	}
	const item_t item = A.items[A.item_base + blockIdx.x];
	const uint32_t q = item.q, g = item.g;
	const qmeta_t qm = A.qmeta[q];
	const dev_query_t *Q = &A.queries[q];
	const uint32_t nt = Q->nt;
	const uint64_t seg = (uint64_t)qm.seg_first + g;

	for (uint32_t i = lane; i < MTW / 4 + WAVE; i += WAVE) {
		s_mask[i] = 0;
	}
	if (GEN && lane < 8) {	/* (the pure-OR instantiation never looks at it) */
		s_truth[lane] = Q->truth[lane];
	}
	WAVE_SYNC();

	/*
	 * Per term: set A (window at list index ab, being consumed) and set N (the
	 * window below it, already in registers), then RING windows in flight.
	 * vmA/vmN = lanes not consumed yet.  A tile may run from A into N but never
	 * past N, so every posting of a tile is in A or N when the tile is flushed.
	 */
	const posting_t *pt[NT];
	int32_t ab[NT], lo[NT], hi[NT], pdoc[NT];
	uint64_t vmA[NT], vmN[NT];
	uint32_t Ad[NT], Nd[NT];
	uint32_t rpc = 0, act = 0;	/* per slot: 2 bits of ring phase / 1 bit "has postings in this range" */
	float Ai[NT], Ni[NT], tmx[NT];
	int32_t ldocN[NT];	/* lowest doc of set N if a window lies below it, else 0 */
	ring_stamps<RING> rst[NT];
	uint32_t vseq = 0;
#pragma unroll
	for (int t = 0; t < NT; t++) {
#pragma unroll
		for (int r = 0; r < RING; r++) {
			rst[t].st[r] = 0;
		}
	}

	auto window_mask = [](int32_t wb, int32_t lo_, int32_t hi_) -> uint64_t {
		const int32_t a = max(lo_ - wb, 0), e = min(hi_ - wb, WAVE);
		if (e <= a) {
			return 0;
		}
		const uint64_t upto = e >= WAVE ? ~0ull : ((1ull << e) - 1);
		return upto & ~((1ull << a) - 1);
	};
	auto refresh_pdoc = [&](auto tc) {
		constexpr int t = decltype(tc)::value;
		pdoc[t] = vmA[t] ? __builtin_amdgcn_readlane((int)Ad[t], 63 - __builtin_clzll(vmA[t]))
		    : vmN[t] ? __builtin_amdgcn_readlane((int)Nd[t], 63 - __builtin_clzll(vmN[t])) : -1;
	};
	auto refresh_ldoc = [&](auto tc) {
		constexpr int t = decltype(tc)::value;
		ldocN[t] = (vmN[t] && ab[t] - WAVE > lo[t]) ? __builtin_amdgcn_readlane((int)Nd[t], 0) : 0;
	};
	/* set A is drained and a window lies below it: N becomes A, the oldest
	 * window in flight becomes N, the one RING windows further down is requested */
	auto shift = [&](auto tc) {
		constexpr int t = decltype(tc)::value;
		ab[t] -= WAVE;
		Ad[t] = Nd[t];
		Ai[t] = Ni[t];
		vmA[t] = vmN[t];
		if (ab[t] > lo[t]) {
			vmN[t] = window_mask(ab[t] - WAVE, lo[t], 0x7fffffff);
			const posting_t *np = &pt[t][max(ab[t] - (RING + 1) * WAVE + (int32_t)lane, lo[t])];
			/* ring position = takes so far, mod RING: from ab (one 2-bit constant per slot, rpc) instead
			 * of a loop-carried scalar per slot -- the kernel's scalar registers are what it runs out of */
			const uint32_t rpos = ((rpc >> (2 * t)) - (uint32_t)(ab[t] >> 6)) & (RING - 1);
			bring_take<t, RING>(rpos, rst[t].younger(vseq), Nd[t], Ni[t], np);
			rst[t].rotate(vseq++);
		} else {
			vmN[t] = 0;
			Nd[t] = 0xffffffffu;	/* no doc */
		}
		refresh_ldoc(tc);
	};

	/* DROP: the dense tokens are never streamed; their impacts come from the
	 * terms' columns (scan_args_t::dense_col) */
	const uint32_t dmask = DROP ? (uint32_t)__builtin_amdgcn_readfirstlane((int)Q->drop_mask) : 0u;
	const uint32_t omask = DROP ? (uint32_t)__builtin_amdgcn_readfirstlane((int)Q->outl_mask) : 0u;
	/* DROP: the range's cold phase (k_cold) stopped at doc cs_cur: only docs below
	 * it are left, and only for the sparse terms */
	const uint32_t *cs = A.cold_state + seg * 16;
	const uint32_t cs_left = DROP ? (uint32_t)__builtin_amdgcn_readfirstlane((int)cs[0]) : 0u;	/* 0: range used up */
	const uint32_t cs_nout = DROP ? (uint32_t)__builtin_amdgcn_readfirstlane((int)cs[1]) : 0u;
	const float cs_thr = DROP ? __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)cs[2])) : 0.0f;
	const bool cs_ovf = DROP && __builtin_amdgcn_readfirstlane((int)cs[3]) != 0;
	/*
	 * Set-up in three steps, each ONE memory round trip for all terms together
	 * (term by term -- plan words, wait, windows, wait, next term -- a wavefront spent
	 * ten dependent round trips before its first tile, a fifth of its life):
	 * the terms' plan words; then every term's two register windows and its ring;
	 * then the scalars that need the windows (highest / lowest doc).
	 */
	static_for<NT>([&](auto tc) {
		constexpr int t = decltype(tc)::value;
		pt[t] = A.post;
		lo[t] = hi[t] = ab[t] = 0;
		pdoc[t] = -1;
		vmA[t] = vmN[t] = 0;
		tmx[t] = 0.0f;
		Ad[t] = 0;
		Ai[t] = 0.0f;
		Nd[t] = 0xffffffffu;		/* no doc */
		Ni[t] = 0.0f;
		ldocN[t] = 0;
		if (t < (int)nt) {
			pt[t] = A.post + Q->pbeg[t];
			const uint64_t cb = ((uint64_t)qm.seg_first + q + g) * NXSGPU_MAX_TOKENS + t;
			lo[t] = (int32_t)A.cursors[cb];
			hi[t] = (int32_t)A.cursors[cb + NXSGPU_MAX_TOKENS];
			tmx[t] = Q->tmax[t];
		}
	});
	static_for<NT>([&](auto tc) {
		constexpr int t = decltype(tc)::value;
		if (DROP) {
			if ((((dmask & ~omask) >> t) & 1) || cs_left == 0) {
				hi[t] = lo[t];		/* no postings as far as the windows are concerned */
			} else if ((omask >> t) & 1) {
				/* a dropped term's OUTLIER list (TF-IDF: the postings above the term's
				 * cap, impact = the excess) is scanned like a sparse term's -- for the
				 * bounds only, the exact score takes the column; the cold phase never
				 * looked at it: what is left of it lies below the doc it stopped at */
				if (hi[t] > lo[t]) {
					hi[t] = wave_lower_bound(pt[t], lo[t], hi[t], cs_left);
				}
			} else if (t < (int)nt) {
				hi[t] = min(hi[t], (int32_t)__builtin_amdgcn_readfirstlane((int)cs[4 + t]));
			}
		}
		if (hi[t] > lo[t]) {
			ab[t] = ((hi[t] - 1) >> 6) << 6;
			/* the first take happens at ab - 64 (shift() decrements first) with ring position 0 */
			rpc |= (uint32_t)(((ab[t] >> 6) - 1) & (RING - 1)) << (2 * t);
			act |= 1u << t;
			/* clamped, unpredicated loads: validity lives in the masks */
			const int32_t ia = max(ab[t] + (int32_t)lane, lo[t]);
			const posting_t pa = pt[t][min(ia, hi[t] - 1)];
			Ad[t] = pa.doc; Ai[t] = pa.imp;
			vmA[t] = window_mask(ab[t], lo[t], hi[t]);
			if (ab[t] > lo[t]) {
				const int32_t in = max(ab[t] - WAVE + (int32_t)lane, lo[t]);
				const posting_t pn = pt[t][in];
				Nd[t] = pn.doc; Ni[t] = pn.imp;
				vmN[t] = window_mask(ab[t] - WAVE, lo[t], 0x7fffffff);
			}
			static_for<RING>([&](auto rc) {
				constexpr int r = decltype(rc)::value;
				const int32_t ir = max(ab[t] - (r + 2) * WAVE + (int32_t)lane, lo[t]);
				bpair_request<t * RING + r>(&pt[t][min(ir, hi[t] - 1)]);
				rst[t].st[r] = vseq++;
			});
		}
	});
	/* (the published thresholds of the higher ranges: in flight with the windows) */
	const float hint = range_hint(A, qm, g);	/* 0 = nothing published yet */
	static_for<NT>([&](auto tc) {
		constexpr int t = decltype(tc)::value;
		if (hi[t] > lo[t]) {
			refresh_pdoc(tc);
			refresh_ldoc(tc);
		}
	});

	float top = DROP ? A.cold_top[seg * 64 + lane] : -INFINITY;
	float thr = DROP ? fmaxf(hint, cs_thr) : hint;	/* scores are > 0: 0 passes everything */
	const uint32_t kidx = A.k - 1;			/* 1 <= k <= 64 (host) */
	uint32_t n_out = cs_nout;
	bool ovf = cs_ovf;
	const uint64_t out_base = seg * A.seg_cap;

	/*
	 * Quantisation: q(x) = floor(x * qs) + 2 with qs = QSUM_MAX / (sum of the
	 * terms' largest impacts), so a doc's byte never exceeds QSUM_MAX + 2*NT
	 * <= 240 (no carry into the neighbour doc) and  sum_i q(x_i) / qs  is an
	 * upper bound of the doc's score: floor(y) + 2 >= y + 1 covers the rounding
	 * of the f32 product (and of the reference's f32 additions) with a whole
	 * unit to spare.  A doc can only beat thr if its byte exceeds
	 * thr_q = floor(thr * qs) - 1.
	 */
	float tsum = 0.0f;
#pragma unroll
	for (int t = 0; t < NT; t++) {
		tsum += tmx[t];
	}
	const float qs = tsum > 0.0f ? (float)QSUM_MAX / tsum : 0.0f;
	/* (thr is wave-uniform but lives in a VGPR: hand the result to the scalar unit) */
	auto thr_quant = [&](float th) -> int32_t {
		return __builtin_amdgcn_readfirstlane(th > 0.0f ? (int32_t)min(th * qs, 1.0e6f) - 1 : -1);
	};
	int32_t thr_q = thr_quant(thr);

	/* DROP: what the dense tokens can add to a score (exactly: U; in byte-map
	 * units: qU -- part of every doc's bound from the start) and the largest
	 * share of one sparse posting */
	const uint32_t dropped = dmask;
	uint32_t qU = 0, q1max = 0;
	float U = 0.0f;
	if constexpr (DROP) {
#pragma unroll
		for (int t = 0; t < NT; t++) {
			if ((dmask >> t) & 1) {
				const float cap = Q->tcap[t];	/* (== tmx[t] unless the term has an outlier list) */
				U += cap;			/* token order, f32: see above */
				qU += (uint32_t)(cap * qs) + 2;
			} else {
				q1max = max(q1max, (uint32_t)(tmx[t] * qs) + 2);
			}
		}
		q1max = (uint32_t)__builtin_amdgcn_readfirstlane((int)q1max);
		qU = (uint32_t)__builtin_amdgcn_readfirstlane((int)qU);
		thr_q -= (int32_t)qU;
	}
	(void)U;

	uint32_t n_pend = 0;
	auto push = [&](uint64_t m, uint32_t doc, uint32_t sum) {
		const uint32_t n = __popcll(m);
		if (n_pend + n <= PCAP) {
			if (lane_of(m)) {
				s_pend[n_pend + lanes_below(m)] = doc;
				if (DROP) {
					s_psum[n_pend + lanes_below(m)] = sum;
				}
			}
		}
		n_pend += n;
	};

	auto rfl32 = [](uint32_t v) -> uint32_t {
		return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
	};
	auto rfl64 = [&](uint64_t v) -> uint64_t {
		return (uint64_t)rfl32((uint32_t)v) | ((uint64_t)rfl32((uint32_t)(v >> 32)) << 32);
	};
	/*
	 * Flush (after every tile that pushed something): sort the pending docs
	 * (descending), drop duplicates, score them exactly, emit what beats the
	 * threshold.  Element e = c*64 + lane of the list lives in pd[c].
	 *
	 * Scores come from registers: a tile never spans a rotation (below), so
	 * every posting of the tile sits in its term's set A, or in the window
	 * before it (Pd/Pi) if the term rotated when the tile drained its set.  A
	 * doc's impact in term t is found by comparing the doc with the 64 lanes;
	 * one doc at a time (wave-uniform), terms in token order, sum from 0.0f
	 * (results.c:134-136).  No memory access.
	 */
	auto flush = [&]() {
		constexpr int PC = PCAP / WAVE;
		n_pend = rfl32(n_pend);		/* (see the main loop) */
		n_out = rfl32(n_out);
		const uint32_t nch = (n_pend + WAVE - 1) / WAVE;
		uint32_t pd[PC], rk[PC], ps[PC];
		STAT_ADD(3, 1);
		STAT_ADD(4, n_pend);
#pragma unroll
		for (int c = 0; c < PC; c++) {
			const uint32_t e = c * WAVE + lane;
			pd[c] = e < n_pend ? s_pend[e] : 0;
			ps[c] = (DROP && e < n_pend) ? s_psum[e] : 0;
			rk[c] = 0;
		}
		WAVE_SYNC();
#pragma unroll
		for (int cj = 0; cj < PC; cj++) {
			if ((uint32_t)cj < nch) {
				const uint32_t nj = min(n_pend - cj * WAVE, (uint32_t)WAVE);
				for (uint32_t j = 0; j < nj; j++) {
					const uint32_t dj = __builtin_amdgcn_readlane((int)pd[cj], j);
#pragma unroll
					for (int c = 0; c < PC; c++) {
						if ((uint32_t)c < nch) {
							/* before me: larger doc, or the same doc pushed earlier */
							rk[c] += (c == cj) ? ((dj > pd[c]) || (dj == pd[c] && j < lane))
							    : ((dj > pd[c]) || (dj == pd[c] && cj < c));
						}
					}
				}
			}
		}
#pragma unroll
		for (int c = 0; c < PC; c++) {
			const uint32_t e = c * WAVE + lane;
			if (e < n_pend) {
				s_pend[rk[c]] = pd[c];
				if (DROP) {
					s_psum[rk[c]] = ps[c];
				}
			}
		}
		WAVE_SYNC();

		for (uint32_t off = 0; off < n_pend; off += WAVE) {
			const uint32_t e = off + lane;
			const bool valid = e < n_pend;
			const uint32_t d = valid ? s_pend[e] : 0;
			const bool dup = valid && e > 0 && s_pend[e - 1] == d;
			const bool live = valid && !dup;
			float sc = 0.0f;
			uint64_t todo = ballot64(live);
			/* DROP: every lane fetches its own doc's impacts in the dropped dense
			 * terms -- independent loads, one round trip for the whole chunk */
			uint32_t dcol[NT];
			if constexpr (DROP) {
				static_for<NT>([&](auto tc) {
					constexpr int t = decltype(tc)::value;
					dcol[t] = 0xffffffffu;
					if ((dropped >> t) & 1) {
						const uint64_t cbase = (uint64_t)__builtin_amdgcn_readfirstlane((int)Q->drop_col[t]) * A.dense_stride;
						dcol[t] = A.dense_col[cbase + (live ? d : 0u)];
					}
				});
				/*
				 * The docs were pushed on the ceiling of the dense terms (qU); now
				 * that their real dense impacts are here the bound is redone with
				 * them, all lanes at once: only what can still beat the threshold
				 * goes through the exact, one-doc-at-a-time scoring below.
				 */
				uint32_t qd = 0;
				static_for<NT>([&](auto tc) {
					constexpr int t = decltype(tc)::value;
					if (((dropped >> t) & 1) && dcol[t] != 0xffffffffu) {
						qd += (uint32_t)(__uint_as_float(dcol[t]) * qs) + 2;
					}
				});
				/* a doc is pushed once per visit that found it above the threshold
				 * (adjacent duplicates, at most one per term); pushes are not in
				 * visit order, so its complete byte bound is the LARGEST of them */
				uint32_t sumq = live ? s_psum[e] : 0u;
#pragma unroll
				for (int kk = 1; kk < NT; kk++) {
					if (live && e + kk < n_pend && s_pend[e + kk] == d) {
						sumq = max(sumq, s_psum[e + kk]);
					}
				}
				todo = ballot64(live && (int32_t)(sumq + qd) > thr_q + (int32_t)qU);
				STAT_ADD(11, __popcll(todo));
				STAT_ADD(12, __popcll(ballot64(live)));
			}
			(void)dcol;
			while (todo) {
				const int j = __builtin_ctzll(todo);
				todo &= todo - 1;
				const uint32_t dj = (uint32_t)__builtin_amdgcn_readlane((int)d, j);
				float acc = 0.0f;
				uint32_t pm = 0;	/* the tokens the doc holds (GEN) */
				static_for<NT>([&](auto tc) {
					constexpr int t = decltype(tc)::value;
					if (DROP && ((dropped >> t) & 1)) {
						/* a dense term that left the scan: its impact for this
						 * doc was fetched from the term's column above */
						const uint32_t xb = (uint32_t)__builtin_amdgcn_readlane((int)dcol[t], j);
						if (xb != 0xffffffffu) {
							acc += __uint_as_float(xb);
							pm |= 1u << t;
						}
					} else
					if ((act >> t) & 1) {
						const uint64_t ma = ballot64(Ad[t] == dj);
						if (ma) {
							acc += __builtin_bit_cast(float, __builtin_amdgcn_readlane(
							    __builtin_bit_cast(int, Ai[t]), __builtin_ctzll(ma)));
							if (GEN) {
								if (GEN) {
									pm |= 1u << t;
								}
							}
						} else {
							const uint64_t mp = ballot64(Nd[t] == dj);
							if (mp) {
								acc += __builtin_bit_cast(float, __builtin_amdgcn_readlane(
								    __builtin_bit_cast(int, Ni[t]), __builtin_ctzll(mp)));
								pm |= 1u << t;
							}
						}
					}
				});
				/* every token the doc holds counts towards its score, whatever
				 * its role in the expression (search.c:240-253); the doc is a
				 * result only if its presence mask satisfies the expression */
				if (GEN && !((s_truth[pm >> 5] >> (pm & 31)) & 1)) {
					acc = -INFINITY;
				}
				sc = (lane == (unsigned)j) ? acc : sc;
			}
			const bool cand = live && sc > thr;
			uint64_t bal = ballot64(cand);
			if (!bal) {
				continue;
			}
			const uint32_t ne = __popcll(bal);
			/*
			 * (Shape matters to the compiler's uniformity analysis: every phi at
			 * the join of a lane-dependent branch counts as divergent, so such a
			 * branch must not share its join with an assignment to wave-uniform
			 * state -- else `ovf`, and through the loop exit everything the main
			 * loop carries, ends up in VGPRs.)
			 */
			const bool room = n_out + ne <= A.seg_cap;
			if (!room) {
				ovf = true;
			}
			if (room && cand) {
				/* lanes are in descending doc order */
				const uint64_t o = out_base + n_out + lanes_below(bal);
				A.cand_doc[o] = d;
				A.cand_sc[o] = sc;
			}
			n_out += ne;
			while (bal) {
				const int L = __builtin_ctzll(bal);
				const float v = __shfl(sc, L);
				/* branch-free insert into the sorted top-k register */
				const bool ins = v > thr;
				const uint32_t pos = __popcll(ballot64(top >= v));
				const float up = __shfl_up(top, 1);
				const float ntop = (lane < pos) ? top : (lane == pos ? v : up);
				top = ins ? ntop : top;
				thr = ins ? fmaxf(__shfl(top, kidx), hint) : thr;
				bal &= bal - 1;
			}
		}
		WAVE_SYNC();
		thr_q = thr_quant(thr) - (int32_t)(DROP && dropped ? qU : 0u);
		n_pend = 0;
	};

	/* widest tile tried next: small while nothing is known about the threshold */
	uint32_t tw = thr_q >= 0 ? (uint32_t)MT_W_HINTED : (uint32_t)MT_W0;

	if constexpr (DROP) {
		/* (the cold phase -- while thr < U -- ran in k_cold; its threshold, top-k
		 * scores, output count and the sparse terms' cursors were taken over above) */
		if (dmask) {
			thr_q = thr_quant(thr) - (int32_t)qU;
			tw = thr_q >= (int32_t)q1max ? (uint32_t)MT_W_HINTED : (uint32_t)MT_W0;
		}
	}

	uint32_t ovf_u = 0;		/* `ovf` as the loop carries it */
	for (;;) {
		/*
		 * All of this is wave-uniform and lives in SGPRs; saying so explicitly
		 * (readfirstlane of an SGPR value folds away) stops the compiler's
		 * uniformity analysis from talking itself into a divergent loop, which
		 * put the whole loop state into VGPRs behind exec masks.
		 */
		n_pend = rfl32(n_pend);
		n_out = rfl32(n_out);
		tw = rfl32(tw);
		thr_q = (int32_t)rfl32((uint32_t)thr_q);
		ovf_u = rfl32(ovf_u | (ovf ? 1u : 0u));
		ovf = ovf_u != 0;
#pragma unroll
		for (int t = 0; t < NT; t++) {
			ab[t] = (int32_t)rfl32((uint32_t)ab[t]);
			pdoc[t] = (int32_t)rfl32((uint32_t)pdoc[t]);
			ldocN[t] = (int32_t)rfl32((uint32_t)ldocN[t]);
			vmA[t] = rfl64(vmA[t]);
			vmN[t] = rfl64(vmN[t]);
		}
		/*
		 * The tile: docs [base, md], md = highest unconsumed doc of any term.
		 * It must not reach past any term's set N: base is at least the lowest
		 * doc of every N that has a window below it.  The densest term thus
		 * brings one to two full windows to every tile.
		 */
		int32_t md = -1, low = 0;
#pragma unroll
		for (int t = 0; t < NT; t++) {
			md = max(md, pdoc[t]);
			low = max(low, pdoc[t] >= 0 ? ldocN[t] : 0);
		}
		/* (one loop exit only: with several the compiler treats the loop as
		 * divergent and keeps all its wave-uniform state in VGPRs) */
		if (md < 0 || ovf) {
			break;
		}
		const uint32_t base = (uint32_t)max(low, md - (int32_t)tw + 1);
		const uint32_t n_before = n_pend;
		STAT_ADD(1, 1);
		STAT_ADD(8, (uint32_t)md - base + 1);

		/*
		 * Add.  The old word a visit's atomic returns is looked at only after
		 * every term has been visited (or before the same term's second
		 * atomic, when the tile runs from A into N): up to NT atomics are in
		 * flight and no visit waits for LDS.
		 */
		uint32_t oldv[NT], qv[NT], vdoc[NT];
		uint64_t vis[NT];
		auto resolve = [&](auto tc) {
			constexpr int t = decltype(tc)::value;
			if (vis[t]) {
				/* qv = (q << 8) | shift: bound of the doc's score so far */
				const uint32_t sum = ((oldv[t] >> (qv[t] & 31)) & 0xffu) + (qv[t] >> 8);
				const uint64_t cm = vis[t] & ballot64((int32_t)sum > thr_q);
				if (cm) {
					push(cm, vdoc[t], sum);
				}
				vis[t] = 0;
			}
		};
		auto visit = [&](auto tc, uint64_t inm, uint32_t wd, float wi) {
			constexpr int t = decltype(tc)::value;
			const bool inl = lane_of(inm);
			const uint32_t dd = (wd - base) >> FOLD;
			const uint32_t sh = (dd & 3) * 8;
			const uint32_t w = inl ? (dd >> 2) : MTW / 4 + lane;
			/* floor + 2 >= the exact ceiling whatever the f32 product rounds to */
			const uint32_t qq = (uint32_t)(wi * qs) + 2;
			oldv[t] = __hip_atomic_fetch_add(&s_mask[w], inl ? (qq << sh) : 0u,
			    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
			qv[t] = (qq << 8) | sh;
			vdoc[t] = wd;
			vis[t] = inm;
			STAT_ADD(2, 1);
			STAT_ADD(9, __popcll(inm));
		};
		static_for<NT>([&](auto tc) {
			constexpr int t = decltype(tc)::value;
			vis[t] = 0;
			oldv[t] = 0;
			qv[t] = 0;
			vdoc[t] = 0;
			if (pdoc[t] >= (int32_t)base) {
				const uint64_t inA = vmA[t] & ballot64(Ad[t] >= base);
				if (inA) {
					visit(tc, inA, Ad[t], Ai[t]);
					vmA[t] ^= inA;
				}
				if (vmA[t] == 0 && vmN[t]) {
					const uint64_t inN = vmN[t] & ballot64(Nd[t] >= base);
					if (inN) {
						resolve(tc);
						visit(tc, inN, Nd[t], Ni[t]);
						vmN[t] ^= inN;
					}
				}
			}
		});
		static_for<NT>([&](auto tc) {
			resolve(tc);
		});
		WAVE_SYNC();

		/* wipe the tile's bytes (16 B per lane and store) */
		{
			/* (wave-uniform trip count: a lane-dependent one makes the compiler
			 * treat the enclosing loop's state as divergent) */
			const uint32_t words = ((((uint32_t)md - base) >> FOLD) + 4) >> 2;
			for (uint32_t i0 = 0; i0 < words; i0 += WAVE * 4) {
				*(uint4 *)&s_mask[i0 + lane * 4] = make_uint4(0, 0, 0, 0);
			}
		}

		const uint32_t n_tile = n_pend - n_before;
		if (n_pend > PCAP) {
			ovf = true;
		} else if (n_pend) {
			flush();		/* looks the docs up in A and N: before any shift */
		}
		static_for<NT>([&](auto tc) {
			constexpr int t = decltype(tc)::value;
			while (vmA[t] == 0 && ab[t] > lo[t]) {
				shift(tc);
			}
			refresh_pdoc(tc);
		});
		if (DROP && dropped) {
			/* pushes are cheap here (refined in parallel in the flush): as wide as
			 * the pending list takes */
			if (n_tile <= 36) {
				tw = min(tw * 2, (uint32_t)MTDOCS);
			} else if (n_tile > 88) {
				tw = max(tw / 2, (uint32_t)MT_W0);
			}
		} else
		if (n_tile <= 8) {
			tw = min(tw * 2, (uint32_t)MTDOCS);
		} else if (n_tile > 48) {
			tw = max(tw / 2, (uint32_t)MT_W0);
		}
	}

	STAT_ADD(0, 1);
	STAT_ADD(5, n_out);
	STAT_ADD(7, STAT_CLK() - clk0);
	STAT_ADD(10, ovf ? 1 : 0);
	if (!ovf) {
		range_publish(A, seg, __shfl(top, kidx));
	}
	if (lane == 0) {
		A.seg_count[seg] = ovf ? 0 : n_out;
		if (ovf) {
			/* once more on the accumulator tiles (scan_args_t::retry_items); a full
			 * retry list sends the query to the exact passes */
			const uint32_t ri = (A.retry_items && !(Q->qflags & 1)) ? atomicAdd(A.retry_count, 1u) : 0xffffffffu;
			if (ri < A.retry_cap) {
				A.retry_items[ri] = item;
			} else {
				A.overflow[q] = 1;
			}
		}
	}
}

/*
 * k_cold: the cold phase of the sparse + dense OR class (k_scanm<.., DROP>).
 * While the candidate threshold is below U -- what the dense terms can add to a
 * score -- a doc that holds dense terms only may still be emitted, so EVERY doc of
 * the range counts: the wavefront walks it from the top, 64 consecutive docs per
 * block, one per lane.  Dense impacts come from the terms' columns (one load per
 * dense term and block, four blocks' loads in flight), the sparse terms' few
 * postings of a block from a plain 64-posting window per term, summed in token
 * order like everywhere else (results.c:134-136).  It ends for good (thr never
 * falls) when k docs scoring >= U have been seen here or a higher range has
 * published such a threshold; the wavefront then publishes its own, and hands
 * threshold, top-k scores, output count and the sparse cursors to
 * k_scanm<.., DROP> (cold_state / cold_top), which scans what is left of the
 * range on the sparse terms alone.  A kernel of its own because it is light
 * (full occupancy) while the mask path is register-bound: fused into k_scanm it
 * cost that kernel three quarters of its occupancy.
 */
template <int NT, bool GEN>
__global__ void __launch_bounds__(WAVE)
k_cold(const scan_args_t A)
{
	__shared__ uint32_t s_truth[GEN ? 8 : 1];
	const unsigned lane = threadIdx.x;
	const item_t item = A.items[A.item_base + blockIdx.x];
	const uint32_t q = item.q, g = item.g;
	const qmeta_t qm = A.qmeta[q];
	const dev_query_t *Q = &A.queries[q];
	const uint32_t nt = Q->nt;
	const uint64_t seg = (uint64_t)qm.seg_first + g;
	const uint32_t dmask = (uint32_t)__builtin_amdgcn_readfirstlane((int)Q->drop_mask);
	uint32_t *cs = A.cold_state + seg * 16;

	if (GEN && lane < 8) {
		s_truth[lane] = Q->truth[lane];
	}
	WAVE_SYNC();

	auto rfl32 = [](uint32_t v) -> uint32_t {
		return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
	};
	auto rfl64 = [&](uint64_t v) -> uint64_t {
		return (uint64_t)rfl32((uint32_t)v) | ((uint64_t)rfl32((uint32_t)(v >> 32)) << 32);
	};

	/* sparse terms: a window of 64 postings [wb, wb + 64) clipped to [lo, hi);
	 * vm = lanes not consumed yet (always a prefix: docs are taken from the top) */
	const posting_t *pt[NT];
	int32_t lo[NT], wb[NT];
	uint64_t vm[NT];
	uint32_t wd[NT];
	float wi[NT], U = 0.0f;
	uint64_t colb[NT];
	static_for<NT>([&](auto tc) {
		constexpr int t = decltype(tc)::value;
		pt[t] = A.post;
		lo[t] = wb[t] = 0;
		vm[t] = 0;
		wd[t] = 0;
		wi[t] = 0.0f;
		colb[t] = 0;
		if (t < (int)nt) {
			if ((dmask >> t) & 1) {
				/* (the cap: docs above it are in the term's outlier list, which
				 * k_scanm<.., DROP> scans -- and this phase scores every doc anyway) */
				U += Q->tcap[t];		/* token order, f32 */
				colb[t] = (uint64_t)rfl32(Q->drop_col[t]) * A.dense_stride;
			} else {
				const uint64_t cb = ((uint64_t)qm.seg_first + q + g) * NXSGPU_MAX_TOKENS + t;
				const int32_t l = (int32_t)A.cursors[cb], h = (int32_t)A.cursors[cb + NXSGPU_MAX_TOKENS];
				pt[t] = A.post + Q->pbeg[t];
				lo[t] = l;
				if (h > l) {
					wb[t] = max(h - WAVE, l);
					const int32_t i = wb[t] + (int32_t)lane;
					const posting_t p = pt[t][min(i, h - 1)];
					wd[t] = p.doc;
					wi[t] = p.imp;
					const int32_t n = h - wb[t];
					vm[t] = n >= WAVE ? ~0ull : ((1ull << n) - 1);
				} else {
					wb[t] = l;
				}
			}
		}
	});

	float top = -INFINITY;
	const float hint = range_hint(A, qm, g);
	float thr = hint;
	const uint32_t kidx = A.k - 1;
	uint32_t n_out = 0, ovf = 0;
	const uint64_t out_base = seg * A.seg_cap;
	const uint32_t d_bot = (uint32_t)min((uint64_t)g * qm.group_docs, A.n_docs);
	const uint32_t d_top = (g + 1 == qm.n_groups) ? (uint32_t)A.n_docs :
	    (uint32_t)min((uint64_t)(g + 1) * qm.group_docs, A.n_docs);
	int32_t cur = (int32_t)d_top - 1;
	uint32_t cold = rfl32(thr >= U && thr > 0.0f ? 0u : 1u), rounds = 0;
	constexpr int CB = 4;

	while (cold && cur >= (int32_t)d_bot && !ovf) {
		cur = (int32_t)rfl32((uint32_t)cur);
		n_out = rfl32(n_out);
		rounds = rfl32(rounds);
		ovf = rfl32(ovf);
#pragma unroll
		for (int t = 0; t < NT; t++) {
			wb[t] = (int32_t)rfl32((uint32_t)wb[t]);
			vm[t] = rfl64(vm[t]);
		}
		uint32_t xd[CB][NT];
#pragma unroll
		for (int cb = 0; cb < CB; cb++) {
			const int32_t bcur = cur - cb * WAVE;
			const uint32_t bbase = (uint32_t)max(bcur - (WAVE - 1), (int32_t)d_bot);
			const uint32_t bdoc = bbase + lane;
			const bool binr = bcur >= (int32_t)d_bot && bdoc <= (uint32_t)max(bcur, 0);
			static_for<NT>([&](auto tc) {
				constexpr int t = decltype(tc)::value;
				xd[cb][t] = 0xffffffffu;
				if ((dmask >> t) & 1) {
					xd[cb][t] = A.dense_col[colb[t] + (binr ? bdoc : d_bot)];
				}
			});
		}
#pragma unroll
		for (int cb = 0; cb < CB; cb++) {
			if (cur < (int32_t)d_bot || ovf) {
				break;
			}
			const uint32_t base = (uint32_t)max(cur - (WAVE - 1), (int32_t)d_bot);
			const uint32_t doc = base + lane;
			const bool inr = doc <= (uint32_t)cur;
			float acc = 0.0f;
			uint32_t pm = 0;
			static_for<NT>([&](auto tc) {
				constexpr int t = decltype(tc)::value;
				if ((dmask >> t) & 1) {
					if (inr && xd[cb][t] != 0xffffffffu) {
						acc += __uint_as_float(xd[cb][t]);
						pm |= 1u << t;
					}
				} else if (t < (int)nt) {
					for (int guard = 0; guard < 4; guard++) {
						uint64_t in = rfl64(vm[t] & ballot64(wd[t] >= base));
						vm[t] ^= in;
						while (in) {
							const int j = __builtin_ctzll(in);
							in &= in - 1;
							const uint32_t pd = (uint32_t)__builtin_amdgcn_readlane((int)wd[t], j);
							const float pi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(
							    __builtin_bit_cast(int, wi[t]), j));
							if (doc == pd) {
								acc += pi;
								pm |= 1u << t;
							}
						}
						if (!(vm[t] == 0 && wb[t] > lo[t])) {
							break;
						}
						/* the window below */
						const int32_t nwb = max(wb[t] - WAVE, lo[t]);
						const int32_t n = wb[t] - nwb;
						const posting_t p = pt[t][min(nwb + (int32_t)lane, wb[t] - 1)];
						wd[t] = p.doc;
						wi[t] = p.imp;
						vm[t] = n >= WAVE ? ~0ull : ((1ull << n) - 1);
						wb[t] = nwb;
					}
				}
			});
			bool match = inr && pm != 0;
			if (GEN) {
				match = match && ((s_truth[pm >> 5] >> (pm & 31)) & 1);
			}
			const bool cand = match && acc > thr;
			uint64_t bal = ballot64(cand);
			if (bal) {
				const uint32_t ne = __popcll(bal);
				const bool room = n_out + ne <= A.seg_cap;
				if (!room) {
					ovf = 1;
				}
				if (room && cand) {
					/* lanes ascend with the doc: higher lanes are emitted first */
					const uint64_t above = (lane == 63) ? 0 : (bal >> (lane + 1));
					const uint64_t o = out_base + n_out + __popcll(above);
					A.cand_doc[o] = doc;
					A.cand_sc[o] = acc;
				}
				n_out += ne;
				while (bal) {
					const int L = 63 - __builtin_clzll(bal);
					const float v = __shfl(acc, L);
					const bool ins = v > thr;
					const uint32_t pos = __popcll(ballot64(top >= v));
					const float up = __shfl_up(top, 1);
					const float ntop = (lane < pos) ? top : (lane == pos ? v : up);
					top = ins ? ntop : top;
					thr = ins ? fmaxf(__shfl(top, kidx), hint) : thr;
					bal &= ~(1ull << L);
				}
			}
			cur = (int32_t)base - 1;
		}
		rounds++;
		if ((rounds & 3) == 0) {
			thr = fmaxf(thr, range_hint(A, qm, g));		/* a higher range may have published */
		}
		cold = rfl32(thr >= U && thr > 0.0f ? 0u : 1u);
	}

	/* lower ranges start warm */
	range_publish(A, seg, __shfl(top, kidx));
	A.cold_top[seg * 64 + lane] = top;
	static_for<NT>([&](auto tc) {
		constexpr int t = decltype(tc)::value;
		if (lane == 0 && t < (int)nt && !((dmask >> t) & 1)) {
			cs[4 + t] = (uint32_t)(wb[t] + (int32_t)__popcll(vm[t]));	/* the term's postings still to scan end here */
		}
	});
	if (lane == 0) {
		cs[0] = cur >= (int32_t)d_bot ? (uint32_t)cur + 1 : 0u;	/* docs below this are left (0: none) */
		cs[1] = n_out;
		cs[2] = __float_as_uint(thr);
		cs[3] = ovf;
	}
}

/* ---- launchers ------------------------------------------------------ */

/* k_scanm, top-k filter pass (1 <= k <= 64); gen: the expression is more than an OR */
void
nxs_launch_scanm(uint32_t nt_bucket, bool gen, unsigned grid_, hipStream_t st, const scan_args_t &a)
{
	const dim3 grid(grid_), block(WAVE);

	if (!gen) {
		switch (nt_bucket) {
		case 2:		/* two tokens: the third slot stays empty */
		case 3: hipLaunchKernelGGL((k_scanm<3, false>), grid, block, 0, st, a); break;
		case 5: hipLaunchKernelGGL((k_scanm<5, false>), grid, block, 0, st, a); break;
		default: hipLaunchKernelGGL((k_scanm<8, false>), grid, block, 0, st, a); break;
		}
	} else {
		switch (nt_bucket) {
		case 2:
		case 3: hipLaunchKernelGGL((k_scanm<3, true>), grid, block, 0, st, a); break;
		case 5: hipLaunchKernelGGL((k_scanm<5, true>), grid, block, 0, st, a); break;
		default: hipLaunchKernelGGL((k_scanm<8, true>), grid, block, 0, st, a); break;
		}
	}
}

/* the sparse + dense OR class: its cold phase (k_cold), then the mask path on
 * the sparse terms (k_scanm<.., DROP>), stream-ordered */
void
nxs_launch_drop_class(uint32_t nt_bucket, unsigned grid_, hipStream_t st, const scan_args_t &a)
{
	const dim3 grid(grid_), block(WAVE);

	switch (nt_bucket) {
	case 2:
	case 3:
		hipLaunchKernelGGL((k_cold<3, false>), grid, block, 0, st, a);
		if (a.flags & 16) { nxs_launch_scans_drop(nt_bucket, grid_, st, a); break; }
		if (a.flags & 8) { nxs_launch_scanb(nt_bucket, false, true, grid_, st, a); break; }
		hipLaunchKernelGGL((k_scanm<3, false, true>), grid, block, 0, st, a);
		break;
	case 5:
		hipLaunchKernelGGL((k_cold<5, false>), grid, block, 0, st, a);
		if (a.flags & 16) { nxs_launch_scans_drop(nt_bucket, grid_, st, a); break; }
		if (a.flags & 8) { nxs_launch_scanb(nt_bucket, false, true, grid_, st, a); break; }
		hipLaunchKernelGGL((k_scanm<5, false, true>), grid, block, 0, st, a);
		break;
	default:
		hipLaunchKernelGGL((k_cold<8, false>), grid, block, 0, st, a);
		if (a.flags & 16) { nxs_launch_scans_drop(nt_bucket, grid_, st, a); break; }
		hipLaunchKernelGGL((k_scanm<8, false, true>), grid, block, 0, st, a);
		break;
	}
}


/*
 * nxs_gpu_scan_bit.hip -- mask path, second form: k_scanb (one presence BIT per doc pair in LDS,
 * candidates scored one per lane)
 * (MI355X / gfx950 query path of nxsearch; see nxs_gpu_int.h for the map of the files)
 */
#include "nxs_gpu_int.h"
#include "nxs_gpu_dev.h"

/*
 * k_scanb: OR-like queries of sparse terms -- the class k_scanm serves (run_query_logic's
 * doc x token loop, search.c:240-253, for queries without a required token), with the two
 * fixed costs of that kernel taken out:
 *
 *  * k_scanm's LDS holds a quantised score bound per doc, a BYTE: a tile is 8192 docs, a
 *    64-posting window of a rank 100-1000 term spans 3-29 k docs, so a window is visited in
 *    2.4 tiles at 27 of 64 lanes and the per-(tile, term) bookkeeping -- scalar masks,
 *    ballots, which window the tile ends in -- is paid that often (160 issued instructions
 *    per 64 postings, PMC, rounds 2-3).  Here LDS holds one BIT per SB_FOLD docs: a tile is
 *    64k docs in 4 KB, a window almost always lies inside one tile, and a term's windows
 *    of a tile are consumed back to back.
 *  * k_scanm scores a candidate from the register windows, one doc at a time (60
 *    wave-instructions each), which also ties the tile to two windows per term.  Here a
 *    candidate is a LANE: its impact in every term comes from a lower-bound search in the
 *    term's postings of the wavefront's range (all lanes and all terms at once: ~11
 *    dependent L2 hits per 64 candidates), summed in token order from 0.0f like
 *    everywhere else (results.c:134-136).  A false candidate costs a lane, not 60
 *    instructions, so the filter may be blunt.
 *
 * The filter.  The slots are the query's tokens in ASCENDING order of their largest impact
 * (dev_query_t::slot_tok; the commonest term first) and a tile applies them in that order.
 * A posting of slot s with impact x is pushed if
 *    x > thr                                   (the doc may hold nothing else), or
 *    its bit was already set  and  x + E_s > thr,   E_s = sum of the largest impacts of
 *                                                   the slots before s
 * -- a necessary condition for score > thr at the doc's LAST posting in slot order: every
 * other posting of the doc is then in an earlier slot (all in this tile: a tile is a doc
 * range and every slot is consumed down to its lower edge), the bit is set, and the doc's
 * f32 sum is at most x + E_s up to rounding (margins below).  Bits shared by two docs or
 * set by a different doc only add candidates.  The commonest terms, which hold most of
 * the postings, meet the sharpest test: a coincidence of two common terms is not pushed
 * unless their two ceilings together reach the threshold.
 *
 * Candidates wait in an LDS list; when it is nearly full (or the tile is done) they are
 * scored, and the ones that beat the threshold (few, once it is warm) are kept as
 * survivors until the tile's end, sorted by descending doc, deduplicated, emitted and fed
 * to the wavefront's top-k register -- exact scores in descending doc order, a superset
 * of what the reference's heap would accept, as with every other scan kernel.
 */
#ifndef SB_WORDS
#define	SB_WORDS	1024		/* bitmap words per tile (4 KB) */
#endif
#ifndef SB_FOLD
#define	SB_FOLD		1		/* log2(docs per bit) */
#endif
#define	SB_DOCS		((SB_WORDS * 32u) << SB_FOLD)
#ifndef SB_PCAP
#define	SB_PCAP		256		/* pending candidates (scored when fewer than 64 slots are left) */
#endif
#ifndef SB_SCAP
#define	SB_SCAP		256		/* survivors of a tile */
#endif
#ifndef SB_RING
#define	SB_RING		4		/* windows in flight per term */
#endif
#ifndef SB_COLD_POST
#define	SB_COLD_POST	64		/* postings the first tile of a cold range aims at */
#endif

#ifdef NXS_STATS
extern __device__ unsigned long long g_stats[16];
#define	STAT_ADD(i, v)	do { if (lane == 0) atomicAdd(&g_stats[i], (unsigned long long)(v)); } while (0)
#else
#define	STAT_ADD(i, v)	do { } while (0)
#endif

template <int NT, bool GEN, bool DROP>
__global__ void __launch_bounds__(WAVE)
k_scanb(const scan_args_t A)
{
	constexpr int RING = SB_RING;
	__shared__ __attribute__((aligned(16))) uint32_t s_bits[SB_WORDS];
	__shared__ uint32_t s_pend[SB_PCAP];
	__shared__ uint32_t s_sdoc[SB_SCAP];
	__shared__ float s_ssc[SB_SCAP];
	__shared__ uint32_t s_truth[GEN ? 8 : 1];

	const unsigned lane = threadIdx.x;
	if constexpr (DROP) {
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

	for (uint32_t i = lane * 4; i < SB_WORDS; i += WAVE * 4) {
		*(uint4 *)&s_bits[i] = make_uint4(0, 0, 0, 0);
	}
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
	auto window_mask = [](int32_t wb, int32_t lo_, int32_t hi_) -> uint64_t {
		const int32_t a = max(lo_ - wb, 0), e = min(hi_ - wb, WAVE);
		if (e <= a) {
			return 0;
		}
		const uint64_t upto = e >= WAVE ? ~0ull : ((1ull << e) - 1);
		return upto & ~((1ull << a) - 1);
	};

	/*
	 * Per slot: the window being consumed (set A: list index ab, lanes not consumed yet
	 * vmA) in two VGPRs, RING windows below it in flight (AGPR pairs, bring_take).
	 * lo / hi: the slot's postings of this wavefront's doc range.
	 */
	const posting_t *pt[NT];
	int32_t ab[NT], lo[NT], hi[NT], pdoc[NT];
	uint64_t vmA[NT];
	uint32_t Ad[NT], rp[NT], stok[NT];
	float Ai[NT], tmx[NT];
	uint32_t sdrop = 0;		/* DROP: slots whose impact comes from the term's column */
	uint64_t colb[NT];
	float ecap[NT];			/* what a posting of the slot adds to a doc's bound at most */

	const uint32_t dmask = DROP ? rfl32(Q->drop_mask) : 0u;
	const uint32_t omask = DROP ? rfl32(Q->outl_mask) : 0u;
	const uint32_t *cs = A.cold_state + seg * 16;
	const uint32_t cs_left = DROP ? rfl32(cs[0]) : 0u;
	const uint32_t cs_nout = DROP ? rfl32(cs[1]) : 0u;
	const float cs_thr = DROP ? __uint_as_float(rfl32(cs[2])) : 0.0f;
	const bool cs_ovf = DROP && rfl32(cs[3]) != 0;

	static_for<NT>([&](auto sc_) {
		constexpr int s = decltype(sc_)::value;
		pt[s] = A.post;
		lo[s] = hi[s] = ab[s] = 0;
		pdoc[s] = -1;
		vmA[s] = 0;
		rp[s] = 0;
		tmx[s] = 0.0f;
		Ad[s] = 0;
		Ai[s] = 0.0f;
		stok[s] = s;
		colb[s] = 0;
		ecap[s] = 0.0f;
		if (s < (int)nt) {
			const uint32_t tok = rfl32(Q->slot_tok[s]);
			stok[s] = tok;
			pt[s] = A.post + Q->pbeg[tok];
			const uint64_t cb = ((uint64_t)qm.seg_first + q + g) * NXSGPU_MAX_TOKENS + tok;
			lo[s] = (int32_t)A.cursors[cb];
			hi[s] = (int32_t)A.cursors[cb + NXSGPU_MAX_TOKENS];
			tmx[s] = Q->tmax[tok];
			ecap[s] = tmx[s];
			if (DROP) {
				if ((dmask >> tok) & 1) {
					sdrop |= 1u << s;
					colb[s] = (uint64_t)rfl32(Q->drop_col[tok]) * A.dense_stride;
					ecap[s] = 0.0f;		/* (its ceiling is part of U) */
					if ((omask >> tok) & 1) {
						/* the dropped term's OUTLIER list (doc, impact - cap): scanned for
						 * the bounds, what the cold phase left of it lies below cs_left */
						ecap[s] = (tmx[s] - Q->tcap[tok]) * 1.000001f;
						if (hi[s] > lo[s] && cs_left) {
							hi[s] = wave_lower_bound(pt[s], lo[s], hi[s], cs_left);
						}
					} else {
						hi[s] = lo[s];
					}
				} else {
					hi[s] = min(hi[s], (int32_t)rfl32(cs[4 + tok]));
				}
				if (cs_left == 0) {
					hi[s] = lo[s];
				}
			}
		}
	});
	static_for<NT>([&](auto sc_) {
		constexpr int s = decltype(sc_)::value;
		if (hi[s] > lo[s]) {
			ab[s] = ((hi[s] - 1) >> 6) << 6;
			/* clamped, unpredicated loads: validity lives in the masks */
			const int32_t ia = max(ab[s] + (int32_t)lane, lo[s]);
			const posting_t pa = pt[s][min(ia, hi[s] - 1)];
			Ad[s] = pa.doc;
			Ai[s] = pa.imp;
			vmA[s] = window_mask(ab[s], lo[s], hi[s]);
			static_for<RING>([&](auto rc) {
				constexpr int r = decltype(rc)::value;
				const int32_t ir = max(ab[s] - (r + 1) * WAVE + (int32_t)lane, lo[s]);
				bpair_request<s * RING + r>(&pt[s][min(ir, hi[s] - 1)]);
			});
		}
	});
	const float hint = range_hint(A, qm, g);	/* 0 = nothing published yet */
	uint32_t maxlen = 0, npost = 0;
	static_for<NT>([&](auto sc_) {
		constexpr int s = decltype(sc_)::value;
		if (hi[s] > lo[s]) {
			pdoc[s] = __builtin_amdgcn_readlane((int)Ad[s], 63 - __builtin_clzll(vmA[s]));
			maxlen = max(maxlen, (uint32_t)(hi[s] - lo[s]));
			npost += (uint32_t)(hi[s] - lo[s]);
		}
	});
	/* steps of the lower-bound search over the longest slot (wave-uniform) */
	const uint32_t nsteps = rfl32(32u - (uint32_t)__builtin_clz(maxlen | 1u));

	float top = DROP ? A.cold_top[seg * 64 + lane] : -INFINITY;
	float thr = DROP ? fmaxf(hint, cs_thr) : hint;	/* scores are > 0: 0 passes everything */
	const uint32_t kidx = A.k - 1;			/* 1 <= k <= 64 (host) */
	uint32_t n_out = cs_nout;
	bool ovf = cs_ovf;
	const uint64_t out_base = seg * A.seg_cap;

	/*
	 * The per-slot thresholds.  A doc's reference score is the f32 sum of its impacts in
	 * token order: at most (the real sum) x (1 + 2^-24)^7.  E_s is summed in f32 too (it
	 * may come out low by as much); the margins DELTA = 2^-18 on both sides of
	 *    x > thr (1 - DELTA) - (E_s + U) (1 + DELTA)
	 * cover all of that 16 times over.  DROP: U = what the dense tokens (whose lists are
	 * not scanned) can add, part of every doc's bound.
	 */
	constexpr float DELTA = 1.0f / 262144.0f;
	float E[NT], U = 0.0f;
	{
		float e = 0.0f;
#pragma unroll
		for (int s = 0; s < NT; s++) {
			E[s] = e;
			e += ecap[s];
		}
	}
	if constexpr (DROP) {
#pragma unroll
		for (int s = 0; s < NT; s++) {
			if ((sdrop >> s) & 1) {
				U += Q->tcap[stok[s]];
			}
		}
	}
	float thr1, thrh[NT];
	auto set_thresholds = [&]() {
		const float tl = thr * (1.0f - DELTA);
		/* singles: the score of a doc holding only this posting is the impact itself
		 * (+ the dense tokens' share) */
		thr1 = DROP ? tl - U * (1.0f + DELTA) : thr;
#pragma unroll
		for (int s = 0; s < NT; s++) {
			thrh[s] = tl - (E[s] + U) * (1.0f + DELTA);
		}
	};
	set_thresholds();

	uint32_t n_pend = 0, n_surv = 0;
	auto push = [&](uint64_t m, uint32_t doc) {
		if (lane_of(m)) {
			s_pend[n_pend + lanes_below(m)] = doc;
		}
		n_pend += __popcll(m);
	};

	/*
	 * Score the pending docs, one per lane: a lower-bound search for the doc in every
	 * slot's postings of the range, all slots in step (one dependent load per step and
	 * slot, independent across slots and lanes).
	 */
	auto score_pending = [&]() {
		n_pend = rfl32(n_pend);
		n_surv = rfl32(n_surv);
		WAVE_SYNC();
		STAT_ADD(3, 1);
		STAT_ADD(4, n_pend);
		for (uint32_t off = 0; off < n_pend; off += WAVE) {
			const uint32_t e = off + lane;
			const bool valid = e < n_pend;
			const uint32_t d = valid ? s_pend[e] : 0xffffffffu;
			int32_t l[NT], h[NT];
#pragma unroll
			for (int s = 0; s < NT; s++) {
				l[s] = lo[s];
				h[s] = hi[s];
			}
			for (uint32_t i = 0; i < nsteps; i++) {
				static_for<NT>([&](auto sc_) {
					constexpr int s = decltype(sc_)::value;
					if (hi[s] > lo[s] && !(DROP && ((sdrop >> s) & 1))) {
						const int32_t mid = (l[s] + h[s]) >> 1;
						const uint32_t v = pt[s][min(mid, hi[s] - 1)].doc;
						const bool act = l[s] < h[s];
						const bool less = v < d;
						l[s] = (act && less) ? mid + 1 : l[s];
						h[s] = (act && !less) ? mid : h[s];
					}
				});
			}
			float imp[NT];
			bool fnd[NT];
			static_for<NT>([&](auto sc_) {
				constexpr int s = decltype(sc_)::value;
				imp[s] = 0.0f;
				fnd[s] = false;
				if (DROP && ((sdrop >> s) & 1)) {
					/* a dense term that left the scan: its impact from the term's column */
					const uint32_t xb = A.dense_col[colb[s] + (valid ? d : 0u)];
					fnd[s] = valid && xb != 0xffffffffu;
					imp[s] = __uint_as_float(xb);
				} else if (hi[s] > lo[s]) {
					const posting_t p = pt[s][min(l[s], hi[s] - 1)];
					fnd[s] = valid && l[s] < hi[s] && p.doc == d;
					imp[s] = p.imp;
				}
			});
			/* token order (results.c:134-136): slot of token tok by wave-uniform selects */
			float sc = 0.0f;
			uint32_t pm = 0;
#pragma unroll
			for (int tok = 0; tok < NT; tok++) {
				float x = 0.0f;
				bool f = false;
#pragma unroll
				for (int s = 0; s < NT; s++) {
					const bool me = stok[s] == (uint32_t)tok && s < (int)nt;
					x = me ? imp[s] : x;
					f = me ? fnd[s] : f;
				}
				sc = f ? sc + x : sc;
				pm |= f ? 1u << tok : 0u;
			}
			/* every token the doc holds counts towards its score, whatever its role in
			 * the expression (search.c:240-253); the doc is a result only if its
			 * presence mask satisfies the expression */
			bool match = valid && pm != 0;
			if (GEN) {
				match = match && ((s_truth[pm >> 5] >> (pm & 31)) & 1);
			}
			const bool cand = match && sc > thr;
			const uint64_t bal = ballot64(cand);
			if (bal) {
				const uint32_t ne = __popcll(bal);
				const bool room = n_surv + ne <= SB_SCAP;
				if (!room) {
					ovf = true;
				}
				if (room && cand) {
					const uint32_t o = n_surv + lanes_below(bal);
					s_sdoc[o] = d;
					s_ssc[o] = sc;
				}
				n_surv += ne;
			}
		}
		WAVE_SYNC();
		n_pend = 0;
	};

	/*
	 * The tile's survivors: sort (descending doc), drop duplicates (a doc is pushed
	 * once per posting that found its bit set), emit, feed the top-k register.
	 */
	auto emit_survivors = [&]() {
		constexpr int SC = SB_SCAP / WAVE;
		n_surv = rfl32(n_surv);
		n_out = rfl32(n_out);
		const uint32_t nch = (n_surv + WAVE - 1) / WAVE;
		uint32_t pd[SC], rk[SC];
		float ps[SC];
#pragma unroll
		for (int c = 0; c < SC; c++) {
			const uint32_t e = c * WAVE + lane;
			pd[c] = e < n_surv ? s_sdoc[e] : 0;
			ps[c] = e < n_surv ? s_ssc[e] : 0.0f;
			rk[c] = 0;
		}
		WAVE_SYNC();
#pragma unroll
		for (int cj = 0; cj < SC; cj++) {
			if ((uint32_t)cj < nch) {
				const uint32_t nj = min(n_surv - cj * WAVE, (uint32_t)WAVE);
				for (uint32_t j = 0; j < nj; j++) {
					const uint32_t dj = __builtin_amdgcn_readlane((int)pd[cj], j);
#pragma unroll
					for (int c = 0; c < SC; c++) {
						if ((uint32_t)c < nch) {
							/* before me: larger doc, or the same doc stored earlier */
							rk[c] += (c == cj) ? ((dj > pd[c]) || (dj == pd[c] && j < lane))
							    : ((dj > pd[c]) || (dj == pd[c] && cj < c));
						}
					}
				}
			}
		}
#pragma unroll
		for (int c = 0; c < SC; c++) {
			const uint32_t e = c * WAVE + lane;
			if (e < n_surv) {
				s_sdoc[rk[c]] = pd[c];
				s_ssc[rk[c]] = ps[c];
			}
		}
		WAVE_SYNC();
		for (uint32_t off = 0; off < n_surv; off += WAVE) {
			const uint32_t e = off + lane;
			const bool valid = e < n_surv;
			const uint32_t d = valid ? s_sdoc[e] : 0;
			const float sc = valid ? s_ssc[e] : 0.0f;
			const bool dup = valid && e > 0 && s_sdoc[e - 1] == d;
			const bool cand = valid && !dup && sc > thr;
			uint64_t bal = ballot64(cand);
			if (!bal) {
				continue;
			}
			const uint32_t ne = __popcll(bal);
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
		n_surv = 0;
		set_thresholds();
	};

	/* set A of slot s is drained and a window lies below it: the oldest window in flight
	 * becomes A, the one RING windows further down is requested */
	auto rotate = [&](auto sc_) {
		constexpr int s = decltype(sc_)::value;
		ab[s] -= WAVE;
		vmA[s] = window_mask(ab[s], lo[s], 0x7fffffff);
		const posting_t *np = &pt[s][max(ab[s] - RING * WAVE + (int32_t)lane, lo[s])];
		bring_take<s, RING>(rp[s], 0, Ad[s], Ai[s], np);
		rp[s] = (rp[s] + 1) & (RING - 1);
	};

	/* first tile: hinted -- an eighth of the bitmap; cold -- every posting passes until k
	 * scores are known, so the docs that hold ~SB_COLD_POST postings */
	const uint32_t d_bot = (uint32_t)min((uint64_t)g * qm.group_docs, A.n_docs);
	const uint32_t d_top = (g + 1 == qm.n_groups) ? (uint32_t)A.n_docs :
	    (uint32_t)min((uint64_t)(g + 1) * qm.group_docs, A.n_docs);
	uint32_t tw;
	{
		const uint32_t rdocs = (DROP && cs_left) ? cs_left - d_bot : d_top - d_bot;
		const uint64_t cold_w = (uint64_t)SB_COLD_POST * max(rdocs, 1u) / max(npost, 1u);
		const uint32_t cw = (uint32_t)min(max(cold_w, (uint64_t)64), (uint64_t)SB_DOCS);
		tw = rfl32(thr > 0.0f ? max(cw, (uint32_t)SB_DOCS / 8) : cw);
	}

	uint32_t ovf_u = 0;		/* `ovf` as the loop carries it */
	for (;;) {
		/* wave-uniform loop state, said so explicitly (see k_scanm) */
		n_pend = rfl32(n_pend);
		n_surv = rfl32(n_surv);
		n_out = rfl32(n_out);
		tw = rfl32(tw);
		ovf_u = rfl32(ovf_u | (ovf ? 1u : 0u));
		ovf = ovf_u != 0;
#pragma unroll
		for (int s = 0; s < NT; s++) {
			ab[s] = (int32_t)rfl32((uint32_t)ab[s]);
			pdoc[s] = (int32_t)rfl32((uint32_t)pdoc[s]);
			rp[s] = rfl32(rp[s]);
			vmA[s] = rfl64(vmA[s]);
		}
		int32_t md = -1;
#pragma unroll
		for (int s = 0; s < NT; s++) {
			md = max(md, pdoc[s]);
		}
		if (md < 0 || ovf) {
			break;
		}
		const uint32_t base = (uint32_t)max(0, md - (int32_t)tw + 1);
		uint32_t n_tile = 0;
		STAT_ADD(1, 1);
		STAT_ADD(8, (uint32_t)md - base + 1);

		/*
		 * The slots in order; a slot is consumed down to the tile's lower edge.  When
		 * the pending list is nearly full the walk stops where it is, the list is
		 * scored (ONE call site: the code is large) and the walk resumes at the same
		 * slot -- consumed lanes are gone from vmA.
		 */
		/* (ONE word of walk state -- next slot, bit 8: stopped -- : two variables that the slot
		 * lambdas assign end up behind a phi of their addresses and stay in scratch memory) */
		uint32_t ws = 0;
		do {
			ws = rfl32(ws) & 0xffu;
			static_for<NT>([&](auto sc_) {
				constexpr int s = decltype(sc_)::value;
				if (ws == (uint32_t)s) {
					bool full = false;
					if (pdoc[s] >= (int32_t)base) {
						for (;;) {
							const uint64_t inA = rfl64(vmA[s] & ballot64(Ad[s] >= base));
							if (inA) {
								const bool inl = lane_of(inA);
								const uint32_t bi = (Ad[s] - base) >> SB_FOLD;
								const uint32_t bit = 1u << (bi & 31);
								bool c = inl && Ai[s] > thr1;
								if constexpr (s == 0) {
									/* nothing before slot 0: no bit can be set */
									if (inl) {
										(void)__hip_atomic_fetch_or(&s_bits[bi >> 5], bit,
										    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
									}
								} else {
									uint32_t old = 0;
									if (inl) {
										old = __hip_atomic_fetch_or(&s_bits[bi >> 5], bit,
										    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
									}
									c = c || (inl && (old & bit) != 0 && Ai[s] > thrh[s]);
								}
								const uint64_t cm = ballot64(c);
								STAT_ADD(2, 1);
								STAT_ADD(9, __popcll(inA));
								if (cm) {
									push(cm, Ad[s]);
									n_tile += __popcll(cm);
								}
								vmA[s] ^= inA;
							}
							full = n_pend > SB_PCAP - WAVE;
							if (full || vmA[s] != 0 || ab[s] <= lo[s]) {
								break;
							}
							rotate(sc_);
						}
					}
					if (!full) {
						pdoc[s] = vmA[s] ? __builtin_amdgcn_readlane((int)Ad[s], 63 - __builtin_clzll(vmA[s])) : -1;
					}
					ws = full ? (uint32_t)s | 0x100u : (uint32_t)s + 1;
				}
			});
			if (n_pend) {
				score_pending();
			}
		} while ((ws & 0xffu) < (uint32_t)NT);

		/* wipe the tile's bits */
		{
			const uint32_t words = ((((uint32_t)md - base) >> SB_FOLD) >> 5) + 1;
			for (uint32_t i0 = 0; i0 < words; i0 += WAVE * 4) {
				*(uint4 *)&s_bits[i0 + lane * 4] = make_uint4(0, 0, 0, 0);
			}
		}
		if (n_surv && !ovf) {
			emit_survivors();
		}
		if (n_tile <= 64) {
			tw = min(tw * 2, (uint32_t)SB_DOCS);
		} else if (n_tile > 160) {
			tw = max(tw / 2, 64u);
		}
	}

	STAT_ADD(0, 1);
	STAT_ADD(5, n_out);
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

/* ---- launchers ------------------------------------------------------ */

/* k_scanb, top-k filter pass (1 <= k <= 64); gen: the expression is more than an OR */
void
nxs_launch_scanb(uint32_t nt_bucket, bool gen, bool drop, unsigned grid_, hipStream_t st, const scan_args_t &a)
{
	const dim3 grid(grid_), block(WAVE);

	if (drop) {
		switch (nt_bucket) {
		case 2:
		case 3: hipLaunchKernelGGL((k_scanb<3, false, true>), grid, block, 0, st, a); break;
		case 5: hipLaunchKernelGGL((k_scanb<5, false, true>), grid, block, 0, st, a); break;
		default: hipLaunchKernelGGL((k_scanb<8, false, true>), grid, block, 0, st, a); break;
		}
	} else if (!gen) {
		switch (nt_bucket) {
		case 2:		/* two tokens: the third slot stays empty */
		case 3: hipLaunchKernelGGL((k_scanb<3, false, false>), grid, block, 0, st, a); break;
		case 5: hipLaunchKernelGGL((k_scanb<5, false, false>), grid, block, 0, st, a); break;
		default: hipLaunchKernelGGL((k_scanb<8, false, false>), grid, block, 0, st, a); break;
		}
	} else {
		switch (nt_bucket) {
		case 2:
		case 3: hipLaunchKernelGGL((k_scanb<3, true, false>), grid, block, 0, st, a); break;
		case 5: hipLaunchKernelGGL((k_scanb<5, true, false>), grid, block, 0, st, a); break;
		default: hipLaunchKernelGGL((k_scanb<8, true, false>), grid, block, 0, st, a); break;
		}
	}
}

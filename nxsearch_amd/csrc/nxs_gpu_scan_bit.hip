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
#define	SB_WORDS	512		/* bitmap words per tile (2 KB) */
#endif
#ifndef SB_FOLD
#define	SB_FOLD		2		/* largest log2(docs per bit): 64k docs per tile.  The fold is chosen per WAVEFRONT (below):
					 * a range whose lists are dense fills the pool with a 16k-doc tile and keeps one bit per doc */
#endif
#define	SB_DOCS		((SB_WORDS * 32u) << SB_FOLD)
#ifndef SB_PCAP
#define	SB_PCAP		384		/* pending candidates of a tile */
#endif
#ifndef SB_SCAP
#define	SB_SCAP		128		/* survivors of a tile */
#endif
#ifndef SB_RING
#define	SB_RING		2		/* windows in flight per term (4: -3 % on the sparse probes, 20 registers more) */
#endif
#ifndef SB_POOL
#define	SB_POOL		32		/* staged 64-posting blocks per tile (2 bytes per posting); a power of two */
#endif
#define	SB_TW_MAX	(SB_DOCS < 65534u ? SB_DOCS : 65534u)	/* staged docs are 16-bit, 0 and 0xffff reserved */
#ifndef SB_COLD_POST
#define	SB_COLD_POST	64		/* postings the first tile of a cold range aims at */
#endif

#ifdef NXS_STATS
/* diagnostic build only (make variant SFX=stats XFLAGS=-DNXS_STATS): event counts of k_scanb */
__device__ unsigned long long g_stats_b[16];
extern "C" void
nxsgpu_debug_stats_bit(unsigned long long *out, int reset)
{
	unsigned long long z[16] = { 0 };
	(void)hipDeviceSynchronize();
	(void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stats_b), sizeof(z));
	if (reset) {
		(void)hipMemcpyToSymbol(HIP_SYMBOL(g_stats_b), z, sizeof(z));
	}
}
#define	STAT_ADD(i, v)	do { if (lane == 0) atomicAdd(&g_stats_b[i], (unsigned long long)(v)); } while (0)
#define	STAT_CLK()	((unsigned long long)__builtin_amdgcn_s_memtime())
#else
#define	STAT_ADD(i, v)	do { } while (0)
#define	STAT_CLK()	0ull
#endif

template <int NT, bool GEN, bool DROP>
__global__ void __launch_bounds__(WAVE)
k_scanb(const scan_args_t A)
{
	constexpr int RING = SB_RING;
	constexpr uint32_t CPC = WAVE / NT;		/* candidates scored per chunk: one lane per (candidate, slot) */
	__shared__ __attribute__((aligned(16))) uint32_t s_bits[SB_WORDS];
	__shared__ uint16_t s_pool[SB_POOL * WAVE];
	__shared__ uint32_t s_pend[SB_PCAP];
	__shared__ uint32_t s_sdoc[SB_SCAP];
	__shared__ float s_ssc[SB_SCAP];
	__shared__ uint32_t s_res[WAVE];		/* [candidate of the chunk][token]: impact bits or ~0 */
	__shared__ uint32_t s_slot[NT * 8];		/* per slot: list pointer (2), lo, hi, token | drop << 8, column base (2) */
	__shared__ uint32_t s_tile[NT * 4];		/* per slot and tile: staged bytes' offset in s_pool, their length, list index of element 0 */
	__shared__ uint32_t s_truth[GEN ? 8 : 1];

	const unsigned lane = threadIdx.x;
	const unsigned long long clk0 = STAT_CLK();
	(void)clk0;
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
	int32_t ab[NT], lo[NT];
	uint64_t vmA[NT];
	uint32_t Ad[NT];
	float Ai[NT];
	uint32_t sdrop = 0;		/* DROP: slots whose impact comes from the term's column */
	uint32_t rpc = 0;		/* 2 bits per slot: ring position = (this - (ab >> 6)) & (RING - 1) */
	uint32_t maxlen = 0, npost = 0;
	float E[NT], U = 0.0f;		/* E[s]: sum of the largest impacts of the slots before s */
	{
		const uint32_t dmask = DROP ? rfl32(Q->drop_mask) : 0u;
		const uint32_t omask = DROP ? rfl32(Q->outl_mask) : 0u;
		float esum = 0.0f;
		static_for<NT>([&](auto sc_) {
			constexpr int s = decltype(sc_)::value;
			int32_t hi_s = 0;
			uint32_t tok = s, drop = 0;
			uint64_t colb = 0;
			float ecap = 0.0f;	/* what a posting of the slot adds to a doc's bound at most */
			pt[s] = A.post;
			lo[s] = ab[s] = 0;
			vmA[s] = 0;
			Ad[s] = 0;
			Ai[s] = 0.0f;
			if (s < (int)nt) {
				tok = rfl32(Q->slot_tok[s]);
				pt[s] = A.post + Q->pbeg[tok];
				const uint64_t cb = ((uint64_t)qm.seg_first + q + g) * NXSGPU_MAX_TOKENS + tok;
				lo[s] = (int32_t)A.cursors[cb];
				hi_s = (int32_t)A.cursors[cb + NXSGPU_MAX_TOKENS];
				ecap = Q->tmax[tok];
				if (DROP) {
					const uint32_t *cs = A.cold_state + seg * 16;
					const uint32_t cs_left = rfl32(cs[0]);
					if ((dmask >> tok) & 1) {
						drop = 1;
						sdrop |= 1u << s;
						colb = (uint64_t)rfl32(Q->drop_col[tok]) * A.dense_stride;
						U += Q->tcap[tok];
						if ((omask >> tok) & 1) {
							/* the dropped term's OUTLIER list (doc, impact - cap): scanned for
							 * the bounds, what the cold phase left of it lies below cs_left */
							ecap = (ecap - Q->tcap[tok]) * 1.000001f;
							if (hi_s > lo[s] && cs_left) {
								hi_s = wave_lower_bound(pt[s], lo[s], hi_s, cs_left);
							}
						} else {
							ecap = 0.0f;		/* (its ceiling is part of U) */
							hi_s = lo[s];
						}
					} else {
						hi_s = min(hi_s, (int32_t)rfl32(cs[4 + tok]));
					}
					if (cs_left == 0) {
						hi_s = lo[s];
					}
				}
			}
			E[s] = esum;
			esum += ecap;
			if (lane == 0) {
				s_slot[s * 8 + 0] = (uint32_t)(uintptr_t)pt[s];
				s_slot[s * 8 + 1] = (uint32_t)((uintptr_t)pt[s] >> 32);
				s_slot[s * 8 + 2] = (uint32_t)lo[s];
				s_slot[s * 8 + 3] = (uint32_t)hi_s;
				s_slot[s * 8 + 4] = tok | drop << 8 | (s < (int)nt ? 0u : 1u << 9);
				s_slot[s * 8 + 5] = (uint32_t)colb;
				s_slot[s * 8 + 6] = (uint32_t)(colb >> 32);
			}
			if (hi_s > lo[s]) {
				ab[s] = ((hi_s - 1) >> 6) << 6;
				rpc |= (uint32_t)(((ab[s] >> 6) - 1) & (RING - 1)) << (2 * s);
				/* clamped, unpredicated loads: validity lives in the masks */
				const int32_t ia = max(ab[s] + (int32_t)lane, lo[s]);
				const posting_t pa = pt[s][min(ia, hi_s - 1)];
				Ad[s] = pa.doc;
				Ai[s] = pa.imp;
				vmA[s] = window_mask(ab[s], lo[s], hi_s);
				static_for<RING>([&](auto rc) {
					constexpr int r = decltype(rc)::value;
					const int32_t ir = max(ab[s] - (r + 1) * WAVE + (int32_t)lane, lo[s]);
					bpair_request<s * RING + r>(&pt[s][min(ir, hi_s - 1)]);
				});
				maxlen = max(maxlen, (uint32_t)(hi_s - lo[s]));
				npost += (uint32_t)(hi_s - lo[s]);
			}
		});
	}
	const float hint = range_hint(A, qm, g);	/* 0 = nothing published yet */
	/*
	 * The first windows were requested with ordinary loads: have them arrive HERE.  Their
	 * first use is inside the tile loop otherwise, and the wait the compiler puts in
	 * front of it -- s_waitcnt vmcnt(0), every time round -- also waits for every ring
	 * window in flight: each rotation then costs a full memory latency.
	 */
#pragma unroll
	for (int s = 0; s < NT; s++) {
		asm volatile("" : "+v"(Ad[s]), "+v"(Ai[s]));
	}
	WAVE_SYNC();
	/* steps of the fallback search (the postings themselves) over the longest slot */
	const uint32_t nsteps = rfl32(32u - (uint32_t)__builtin_clz(maxlen | 1u));

	const uint32_t *cs0 = A.cold_state + seg * 16;
	float top = DROP ? A.cold_top[seg * 64 + lane] : -INFINITY;
	float thr = DROP ? fmaxf(hint, __uint_as_float(rfl32(cs0[2]))) : hint;	/* scores are > 0: 0 passes everything */
	const uint32_t kidx = A.k - 1;			/* 1 <= k <= 64 (host) */
	uint32_t n_out = DROP ? rfl32(cs0[1]) : 0u;
	bool ovf = DROP && rfl32(cs0[3]) != 0;
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
#pragma unroll
	for (int s = 0; s < NT; s++) {
		E[s] = (E[s] + U) * (1.0f + DELTA);
	}
	const float U1 = U * (1.0f + DELTA);
	float thr1, thrh[NT];
	auto set_thresholds = [&]() {
		const float tl = thr * (1.0f - DELTA);
		/* singles: the score of a doc holding only this posting is the impact itself
		 * (+ the dense tokens' share) */
		thr1 = DROP ? tl - U1 : thr;
#pragma unroll
		for (int s = 0; s < NT; s++) {
			thrh[s] = tl - E[s];
		}
	};
	set_thresholds();

	uint32_t n_pend = 0, n_surv = 0;
	auto push = [&](uint64_t m, uint32_t doc) {
		const uint32_t n = __popcll(m);
		if (n_pend + n <= SB_PCAP) {
			if (lane_of(m)) {
				s_pend[n_pend + lanes_below(m)] = doc;
			}
		}
		n_pend += n;
	};

	/*
	 * Score the pending docs.  Where is the doc in slot s's list?  Every window a tile
	 * visits leaves its docs in LDS (s_pool: 16 bits each, relative to the tile's lower
	 * edge, + 1, clamped to [0, 0xffff]: 0 = below the tile).  The pool fills from the
	 * top and a slot's windows are visited highest docs first, so a slot's blocks form
	 * ONE ascending array.  One lane per (candidate, slot): a branch-free lower bound
	 * there (log2 steps of add / min / read / compare / select), then one global load
	 * for the impact of the posting found; the lanes of a candidate leave their impacts
	 * in s_res by TOKEN and the candidate's own lane sums them in token order from 0.0f
	 * (results.c:134-136).  (The first form searched the postings themselves in global
	 * memory, every lane in every slot: 11 dependent steps of 64 scattered requests per
	 * slot -- three quarters of a wavefront's life, the L2 request rate the limit.  That
	 * search, one slot per lane, is still the fallback (`glob`) when a tile outgrows the
	 * pool or its pending list has to be scored before every slot has been visited.)
	 */
	int32_t pblk = SB_POOL;		/* lowest block staged in this tile (the pool fills downwards; < 0: it wrapped) */
	uint32_t tbase = 0;		/* the tile's lower edge */
	uint32_t maxb = 0;		/* most blocks a slot has staged in this tile */
	const uint32_t my_c = lane / NT, my_s = lane - my_c * NT;
	const bool my_on = my_c < CPC;
	auto score_pending = [&](bool glob) {
		n_pend = rfl32(n_pend);
		n_surv = rfl32(n_surv);
		WAVE_SYNC();
		STAT_ADD(3, 1);
		STAT_ADD(4, n_pend);
		STAT_ADD(13, glob ? 1 : 0);
		/* my slot */
		const uint32_t fl = s_slot[my_s * 8 + 4];
		const posting_t *mpt = (const posting_t *)(uintptr_t)((uint64_t)s_slot[my_s * 8] | (uint64_t)s_slot[my_s * 8 + 1] << 32);
		const int32_t mlo = (int32_t)s_slot[my_s * 8 + 2], mhi = (int32_t)s_slot[my_s * 8 + 3];
		const uint32_t msb = s_tile[my_s * 4], mnb = s_tile[my_s * 4 + 1];
		const int32_t mabl = (int32_t)s_tile[my_s * 4 + 2];
		const bool mdrop = DROP && ((fl >> 8) & 1);
		const uint32_t mtok = fl & 0xff;
		/* top step of the lower bound (bytes): the smallest power of two >= the longest slot */
		const uint32_t top_st = rfl32(maxb ? 128u << (32u - (uint32_t)__builtin_clz(maxb) - ((maxb & (maxb - 1)) ? 0u : 1u)) : 0u);
		for (uint32_t off = 0; off < n_pend; off += CPC) {
			const uint32_t e = off + my_c;
			const bool valid = my_on && e < n_pend && !((fl >> 9) & 1);
			const uint32_t d = valid ? s_pend[e] : 0xffffffffu;
			bool fnd = false;
			uint32_t xb = 0xffffffffu;
			if (mdrop) {
				/* a dense term that left the scan: its impact from the term's column */
				const uint64_t colb = (uint64_t)s_slot[my_s * 8 + 5] | (uint64_t)s_slot[my_s * 8 + 6] << 32;
				if (valid) {
					xb = A.dense_col[colb + d];
				}
			} else if (!glob) {
				const uint32_t key = d - tbase + 1;
				uint32_t pos = 0;
				const uint32_t last = mnb ? mnb - 2 : 0u;
				for (uint32_t st = top_st; st >= 2; st >>= 1) {
					const uint32_t t = min(pos + st - 2, last);
					const uint32_t v = *(const uint16_t *)((const uint8_t *)s_pool + msb + t);
					pos = (v < key) ? pos + st : pos;
				}
				if (valid && mnb && pos < mnb) {
					fnd = *(const uint16_t *)((const uint8_t *)s_pool + msb + pos) == key;
				}
				if (fnd) {
					/* (lanes below lo hold copies of posting lo: the first copy is found) */
					xb = __float_as_uint(mpt[max(mabl + (int32_t)(pos >> 1), mlo)].imp);
				}
			} else {
				int32_t l = mlo, h = mhi;
				for (uint32_t i = 0; i < nsteps; i++) {
					const int32_t mid = (l + h) >> 1;
					const bool act = l < h;
					uint32_t v = 0;
					if (act && valid) {
						v = mpt[mid].doc;
					}
					const bool less = v < d;
					l = (act && less) ? mid + 1 : l;
					h = (act && !less) ? mid : h;
				}
				if (valid && l < mhi) {
					const posting_t p = mpt[l];
					if (p.doc == d) {
						xb = __float_as_uint(p.imp);
					}
				}
			}
			if (my_on) {
				s_res[my_c * NT + mtok] = xb;
			}
			WAVE_SYNC();
			/* the candidates' own lanes: token order */
			{
				const uint32_t e2 = off + lane;
				const bool v2 = lane < CPC && e2 < n_pend;
				const uint32_t d2 = v2 ? s_pend[e2] : 0;
				float sc = 0.0f;
				uint32_t pm = 0;
#pragma unroll
				for (int tok = 0; tok < NT; tok++) {
					const uint32_t x = lane < CPC ? s_res[lane * NT + tok] : 0xffffffffu;
					const bool f = x != 0xffffffffu;
					sc = f ? sc + __uint_as_float(x) : sc;
					pm |= f ? 1u << tok : 0u;
				}
				/* every token the doc holds counts towards its score, whatever its role in
				 * the expression (search.c:240-253); the doc is a result only if its
				 * presence mask satisfies the expression */
				bool match = v2 && pm != 0;
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
						s_sdoc[o] = d2;
						s_ssc[o] = sc;
					}
					n_surv += ne;
				}
			}
			WAVE_SYNC();
		}
		STAT_ADD(11, n_surv);
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
		/* (only a list's last window is clipped: the one that holds index lo) */
		vmA[s] = ab[s] >= lo[s] ? ~0ull : ~0ull << (lo[s] - ab[s]);
		const posting_t *np = &pt[s][max(ab[s] - RING * WAVE + (int32_t)lane, lo[s])];
		const uint32_t rp = ((rpc >> (2 * s)) - (uint32_t)(ab[s] >> 6)) & (RING - 1);
		bring_take<s, RING>(rp, 0, Ad[s], Ai[s], np);
		STAT_ADD(12, 1);
	};

	/*
	 * Tile width.  A tile's windows must fit the pool: three quarters of it by the
	 * range's density, adapted as the tiles come (a tile that does not fit is scored by
	 * the fallback search).  Hinted: as wide as that; cold: every posting passes until k
	 * scores are known, so the docs that hold ~SB_COLD_POST postings.
	 */
	const uint32_t d_bot = (uint32_t)min((uint64_t)g * qm.group_docs, A.n_docs);
	const uint32_t d_top = (g + 1 == qm.n_groups) ? (uint32_t)A.n_docs :
	    (uint32_t)min((uint64_t)(g + 1) * qm.group_docs, A.n_docs);
	uint32_t tw, fold, tw_max;
	{
		const uint32_t cs_left = DROP ? rfl32(cs0[0]) : 0u;
		const uint32_t rdocs = (DROP && cs_left) ? cs_left - d_bot : d_top - d_bot;
		const uint64_t cold_w = (uint64_t)SB_COLD_POST * max(rdocs, 1u) / max(npost, 1u);
		const uint64_t fit_w = (uint64_t)((SB_POOL - NT) * WAVE * 3 / 4) * max(rdocs, 1u) / max(npost, 1u);
		const uint64_t w = thr > 0.0f ? fit_w : min(cold_w, fit_w);
		/*
		 * Docs per bit: as few as the pool's tile needs.  The bitmap has SB_WORDS x 32 bits; a
		 * tile is never wider than what fills the pool (fit_w, with room to grow by half), so a
		 * range of dense lists -- narrow tiles -- keeps one bit per doc and a sparse one folds
		 * four docs into a bit: a bit that two docs share only adds candidates.
		 */
		fold = 0;
		while (fold < SB_FOLD && ((uint64_t)(SB_WORDS * 32u) << fold) < fit_w + fit_w / 2) {
			fold++;
		}
		fold = rfl32(fold);
		tw_max = rfl32(min((uint32_t)(SB_WORDS * 32u) << fold, 65534u));
		tw = rfl32((uint32_t)min(max(w, (uint64_t)64), (uint64_t)tw_max));
	}

	uint32_t ovf_u = 0;		/* `ovf` as the loop carries it */
#ifdef SB_PROBE_SETUP_ONLY	/* (instruction-count probes: results are wrong) */
	if (A.k) {
		ovf_u = 2;
	}
#endif
	for (;;) {
		/* wave-uniform loop state, said so explicitly (see k_scanm) */
		n_pend = rfl32(n_pend);
		n_surv = rfl32(n_surv);
		n_out = rfl32(n_out);
		tw = rfl32(tw);
		ovf_u = rfl32(ovf_u | (ovf ? 1u : 0u));
		ovf = ovf_u != 0;
		int32_t md = -1;
#pragma unroll
		for (int s = 0; s < NT; s++) {
			ab[s] = (int32_t)rfl32((uint32_t)ab[s]);
			vmA[s] = rfl64(vmA[s]);
		}
		/* highest unconsumed doc of any slot */
		static_for<NT>([&](auto sc_) {
			constexpr int s = decltype(sc_)::value;
			if (vmA[s]) {
				md = max(md, __builtin_amdgcn_readlane((int)Ad[s], 63 - __builtin_clzll(vmA[s])));
			}
		});
		if (md < 0 || ovf || ovf_u) {
			break;
		}
		const uint32_t base = (uint32_t)max(0, md - (int32_t)tw + 1);
		tbase = base;
		pblk = SB_POOL;
		maxb = 0;
		STAT_ADD(1, 1);
		STAT_ADD(8, (uint32_t)md - base + 1);

		/*
		 * The slots in order; a slot is consumed down to the tile's lower edge.  (A walk
		 * that could stop when the pending list fills up, score it and resume -- a state
		 * machine around the slots -- made the compiler fold the five window loops and
		 * the state machine into ONE loop whose every iteration runs through the
		 * dispatch: 250 M scalar instructions per C3 step.  The loops below are plain
		 * nests; a pending list that overflows sends the range to the retry list.)
		 */
		static_for<NT>([&](auto sc_) {
			constexpr int s = decltype(sc_)::value;
			const int32_t pstart = pblk;	/* the pool's level when the slot begins */
			const int32_t abf = ab[s];	/* ... and the slot's window then (its first staged one) */
			if (vmA[s]) {
				for (;;) {
					const uint64_t inA = rfl64(vmA[s] & ballot64(Ad[s] >= base));
					if (inA) {
						const uint32_t rel = Ad[s] - base;
						const uint32_t bi = rel >> fold;
						const uint32_t bit = 1u << (bi & 31);
						uint32_t old = 0;
						if (lane_of(inA)) {
							/* (slot 0: nothing before it, the old word is not looked at) */
							old = __hip_atomic_fetch_or(&s_bits[bi >> 5], bit,
							    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
						}
						/* stage the window's docs for the scoring (all 64 lanes: docs
						 * below the tile clamp to 0, docs above it stay above every key);
						 * the pool wraps round when a tile outgrows it (`staged`, below) */
						if (!(DROP && ((sdrop >> s) & 1))) {
							pblk--;
							const int32_t val = min(max((int32_t)rel, -1), 0xfffe) + 1;
							s_pool[(((uint32_t)pblk & (SB_POOL - 1)) << 6) + lane] = (uint16_t)val;
						}
						/* wave masks, no lane branches */
						uint64_t cm = ballot64(Ai[s] > thr1);
						if constexpr (s > 0) {
							cm |= ballot64((old & bit) != 0) & ballot64(Ai[s] > thrh[s]);
						}
						cm &= inA;
						STAT_ADD(2, 1);
						STAT_ADD(9, __popcll(inA));
						if (cm) {
							push(cm, Ad[s]);
						}
						vmA[s] ^= inA;
					}
					if (vmA[s] != 0 || ab[s] <= lo[s]) {
						break;
					}
					rotate(sc_);
				}
			}
			{
				/* the slot's staged array: blocks [pblk, pstart), ascending docs */
				const uint32_t nb = (uint32_t)(pstart - pblk);
				maxb = max(maxb, nb);
				if (lane == 0) {
					s_tile[s * 4 + 0] = (uint32_t)pblk << 7;
					s_tile[s * 4 + 1] = nb << 7;
					s_tile[s * 4 + 2] = (uint32_t)(abf - (int32_t)((nb ? nb - 1 : 0u) << 6));
				}
			}
		});
		const uint32_t n_tile = n_pend;
		const bool staged = pblk >= 0;	/* every visit of the tile fitted the pool */
		if (n_pend > SB_PCAP) {
			ovf = true;
		} else if (n_pend) {
			/* (with a visit that did not fit the pool: the postings themselves are searched) */
			score_pending(!staged);
		}

		/* wipe the tile's bits */
		{
			const uint32_t words = ((((uint32_t)md - base) >> fold) >> 5) + 1;
			for (uint32_t i0 = 0; i0 < words; i0 += WAVE * 4) {
				*(uint4 *)&s_bits[i0 + lane * 4] = make_uint4(0, 0, 0, 0);
			}
		}
		if (n_surv && !ovf) {
			emit_survivors();
		}
		{
			const uint32_t used = (uint32_t)(SB_POOL - pblk);
			if (!staged || used > SB_POOL * 7 / 8 || n_tile > SB_PCAP / 2) {
				tw = max(tw - tw / 4, 64u);
			} else if (n_tile <= 96 && used <= SB_POOL * 5 / 8) {
				tw = min(tw + tw / 2, tw_max);
			}
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

/* ---- launchers ------------------------------------------------------ */

/* k_scanb, top-k filter pass (1 <= k <= 64), 2..5 tokens (build_worklist routes nothing else
 * here); gen: the expression is more than an OR; drop: the sparse + dense class's second kernel
 * (dense terms from their columns, cold_state from k_cold), NXS_GPU_DROPB */
void
nxs_launch_scanb(uint32_t nt_bucket, bool gen, bool drop, unsigned grid_, hipStream_t st, const scan_args_t &a)
{
	const dim3 grid(grid_), block(WAVE);

	if (drop) {
		/* the sparse + dense class (behind k_cold, which nxs_launch_drop_class has queued) */
		switch (nt_bucket) {
		case 2:
		case 3: hipLaunchKernelGGL((k_scanb<3, false, true>), grid, block, 0, st, a); break;
		default: hipLaunchKernelGGL((k_scanb<5, false, true>), grid, block, 0, st, a); break;
		}
	} else if (!gen) {
		switch (nt_bucket) {
		case 2:		/* two tokens: the third slot stays empty */
		case 3: hipLaunchKernelGGL((k_scanb<3, false, false>), grid, block, 0, st, a); break;
		default: hipLaunchKernelGGL((k_scanb<5, false, false>), grid, block, 0, st, a); break;
		}
	} else {
		switch (nt_bucket) {
		case 2:
		case 3: hipLaunchKernelGGL((k_scanb<3, true, false>), grid, block, 0, st, a); break;
		default: hipLaunchKernelGGL((k_scanb<5, true, false>), grid, block, 0, st, a); break;
		}
	}
}

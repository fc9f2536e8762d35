/*
 * nxs_gpu_scan_req.hip -- k_cursors, k_scan1 (single token), k_scanr (required terms: intersect first), k_scanq (the same through block bitmaps)
 * (MI355X / gfx950 query path of nxsearch; see nxs_gpu_int.h for the map of the files)
 */
#include "nxs_gpu_int.h"
#include "nxs_gpu_dev.h"

/*
 * k_cursors: where every (query, range boundary) falls in every term's list.
 * Boundary b of query q is doc b * group_docs; wavefront g of the query then
 * owns postings [cur[g][t], cur[g+1][t]).  One thread per (boundary, token):
 * a plain binary search -- ~24 cache lines each, the top levels shared by all
 * boundaries of a list -- done once per batch instead of by every wavefront.
 */
__global__ void
k_cursors(const posting_t *__restrict__ post, const dev_query_t *__restrict__ queries,
    const qmeta_t *__restrict__ qmeta, const uint32_t *__restrict__ bnd_q,
    uint32_t n_bnd, uint64_t n_docs, uint32_t *__restrict__ cursors)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t b = i / NXSGPU_MAX_TOKENS, t = i % NXSGPU_MAX_TOKENS;
	if (b >= n_bnd) {
		return;
	}
	const uint32_t q = bnd_q[b];
	const qmeta_t qm = qmeta[q];
	const dev_query_t *Q = &queries[q];
	if (t >= Q->nt) {
		return;
	}
	const uint32_t g = b - (qm.seg_first + q);
	const uint64_t doc = min((uint64_t)g * qm.group_docs, n_docs);
	const uint64_t pb = Q->pbeg[t], pe = Q->pend[t];
	uint64_t lo = pb, hi = pe;
	if (doc >= n_docs) {
		lo = pe;
	} else if (doc == 0) {
		lo = pb;
	} else {
		while (lo < hi) {
			const uint64_t mid = lo + ((hi - lo) >> 1);
			if (post[mid].doc < doc) lo = mid + 1; else hi = mid;
		}
	}
	cursors[(uint64_t)b * NXSGPU_MAX_TOKENS + t] = (uint32_t)(lo - pb);
}

/*
 * k_scan1: single-token queries.  A doc's score is the posting's own impact
 * and it matches iff the one-token mask satisfies the expression, so nothing
 * is accumulated: the wavefront streams its slice of the list downwards, U
 * windows (U x 512 B) in flight, and compares impacts with the running
 * threshold in registers.  No LDS: full occupancy.
 */
template <int MODE>
__global__ void __launch_bounds__(WAVE)
k_scan1(const scan_args_t A)
{
	constexpr int U = 4;
	__shared__ __attribute__((aligned(16))) uint32_t s_hist[MODE == MODE_BIG ? BIGK_BUCKETS : 4];
	const unsigned lane = threadIdx.x;
	if constexpr (MODE == MODE_BIG) {
		for (uint32_t i = lane; i < BIGK_BUCKETS; i += WAVE) {
			s_hist[i] = 0;
		}
	}
	const item_t item = A.items[A.item_base + blockIdx.x];
	const uint32_t q = item.q, g = item.g;
	const qmeta_t qm = A.qmeta[q];
	const dev_query_t *Q = &A.queries[q];
	const uint64_t seg = (uint64_t)qm.seg_first + g;
	const uint64_t dlo = (uint64_t)g * qm.group_docs;
	const uint64_t dhi = min(A.n_docs, dlo + (uint64_t)qm.group_docs);
	const bool matches = Q->nt == 1 && ((Q->truth[0] >> 1) & 1);
	const posting_t *__restrict__ pt = A.post + Q->pbeg[0];
	const int32_t n = Q->nt ? (int32_t)(Q->pend[0] - Q->pbeg[0]) : 0;
	int32_t lo = 0, hi = 0;
	uint32_t n_out = 0;
	bool ovf = false;

	if (matches) {
		(void)dlo; (void)dhi;
		if (qm.pad) {
			/* ranges by posting index (no cursors): range g of G = [n g / G, n (g+1) / G) */
			lo = (int32_t)((uint64_t)n * g / qm.n_groups);
			hi = (int32_t)((uint64_t)n * (g + 1) / qm.n_groups);
		} else {
			const uint64_t cb = ((uint64_t)qm.seg_first + q + g) * NXSGPU_MAX_TOKENS;
			lo = (int32_t)A.cursors[cb];
			hi = (int32_t)A.cursors[cb + NXSGPU_MAX_TOKENS];
		}
	}
	if (MODE == MODE_COUNT) {
		if (lane == 0) {
			A.seg_count[seg] = (uint32_t)(hi - lo);
		}
		return;
	}

	float top = -INFINITY;
	float hint = (MODE == MODE_TOPK && A.k <= WAVE) ? range_hint(A, qm, g) :
	    MODE == MODE_BIG ? bigk_hint(A, qm, g) : -INFINITY;
	float thr = hint;
	const uint32_t kidx = (A.k >= 1 && A.k <= WAVE) ? A.k - 1 : WAVE - 1;
	const bool track = (MODE == MODE_TOPK) && A.k <= WAVE;
	/* MODE_BIG: candidates counted since the threshold was last read off the histogram */
	uint32_t big_since = 0;
	const uint32_t big_upd = bigk_update_every(A.k);
	const uint64_t out_base = (MODE == MODE_ALL) ? A.seg_off[seg] : seg * A.seg_cap;

	while (hi > lo) {
		uint32_t dv[U];
		float iv[U];
#pragma unroll
		for (int u = 0; u < U; u++) {
			const int32_t idx = hi - (u + 1) * WAVE + (int32_t)lane;
			dv[u] = 0;
			iv[u] = -INFINITY;
			if (idx >= lo) {
				const posting_t p = pt[idx];
				dv[u] = p.doc;
				iv[u] = p.imp;
			}
		}
#pragma unroll
		for (int u = 0; u < U; u++) {
			/* window u: descending doc = descending lane */
			const bool cand = iv[u] > thr;
			uint64_t bal = ballot64(cand);
			if (!bal) {
				continue;
			}
			const uint32_t ne = __popcll(bal);
			if (MODE_FILTERS(MODE) && n_out + ne > A.seg_cap) {
				ovf = true;
			} else {
				const uint64_t above = (lane == 63) ? 0 : (bal >> (lane + 1));
				if (cand) {
					const uint64_t o = out_base + n_out + __popcll(above);
					A.cand_doc[o] = dv[u];
					A.cand_sc[o] = iv[u];
				}
			}
			n_out += ne;
			if constexpr (MODE == MODE_BIG) {
				bigk_account(s_hist, A.k, big_upd, cand, iv[u], ne, big_since, hint, thr);
			}
			if (track) {
				while (bal) {
					const int L = 63 - __clzll(bal);
					bal &= ~(1ull << L);
					const float v = __shfl(iv[u], L);
					if (v > thr) {
						const uint32_t pos = __popcll(ballot64(top >= v));
						const float up = __shfl_up(top, 1);
						top = (lane < pos) ? top : (lane == pos ? v : up);
						thr = fmaxf(__shfl(top, kidx), hint);
					}
				}
			}
		}
		hi -= U * WAVE;
	}
	if (MODE == MODE_TOPK && track && !ovf) {
		range_publish(A, seg, __shfl(top, kidx));	/* k-th best of this range */
	}
	if constexpr (MODE == MODE_BIG) {
		if (!ovf) {
			bigk_publish(A, seg, s_hist, A.k);	/* lower bounds of this range's k-th, k/2-th ... best */
		}
	}
	if (lane == 0) {
		if (MODE != MODE_ALL) {
			A.seg_count[seg] = ovf ? 0 : n_out;
		}
		if (MODE_FILTERS(MODE) && ovf) {
			A.overflow[q] = 1;
		}
	}
}

/*
 * k_scanr: queries with required terms (`req` != 0: a doc can only match if
 * it holds every one of them), 2..8 tokens -- the conjunctive shapes.  The
 * reference intersects the terms' doc bitmaps before it scores anything
 * (search.c:118-174); this is the same idea on posting windows:
 *
 *  - slot 0 is the DRIVER: the required term with the fewest postings.  The
 *    other slots follow in ascending list length, required ones first
 *    (dev_query_t::slot_tok, host side);
 *  - a round takes the driver's unconsumed postings that fall into one
 *    RW-doc aligned span (at most one 64-posting window), marks their docs
 *    in an LDS byte map (value = driver lane + 1) and then lets the other
 *    slots look their own postings of that span up in the map.  A hit hands
 *    the posting's impact and presence bit to the driver lane's slot
 *    (s_imp / s_bits); there is no accumulator tile at all;
 *  - after each required slot the driver lanes that did not get its bit are
 *    unmarked.  When no lane is left the round is over: denser slots are not
 *    looked at, and what they hold above the driver's next posting is dropped
 *    unread at the start of the next round (mask, stream, or jump by a window
 *    probe / 64-ary search) -- their postings are mostly never loaded;
 *  - surviving lanes sum their slots' impacts in token order (the f32 order of
 *    results.c:134-136), test the truth table and go through the same
 *    threshold filter / candidate emission as the other scan kernels; lanes
 *    are in ascending doc order, so emission is by descending lane.
 */
#ifdef NXS_STATS
/* diagnostic build only: k_scanr event counts (tools/scanr_stats.py) */
__device__ unsigned long long g_stats_req[16];
#define	RSTAT(i, v)	do { if (lane == 0) atomicAdd(&g_stats_req[i], (unsigned long long)(v)); } while (0)
extern "C" void
nxsgpu_debug_stats_req(unsigned long long *out, int reset)
{
	unsigned long long z[16] = { 0 };
	(void)hipDeviceSynchronize();
	(void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stats_req), sizeof(z));
	if (reset) {
		(void)hipMemcpyToSymbol(HIP_SYMBOL(g_stats_req), z, sizeof(z));
	}
}
#else
#define	RSTAT(i, v)	do { } while (0)
#endif

#ifndef RW
#define	RW	4096		/* docs per round span (LDS byte map; SCANR_HASH 0) */
#endif
#ifndef SCANR_HT_BITS
#define	SCANR_HT_BITS	9
#endif
#define	SCANR_HT	(1 << SCANR_HT_BITS)

template <int MODE, int NT, bool HASHQ = false>
__global__ void __launch_bounds__(WAVE)
k_scanr(const scan_args_t A)
{
	/*
	 * HASH (queries with four required terms and more): a round takes the driver's WHOLE
	 * window, whatever doc span it covers, and keeps its docs in an open-addressing
	 * table (doc -> driver lane + 1) instead of a byte map over RW docs.  A sparse
	 * driver has a posting every few hundred docs: spans of RW docs held ~11 of
	 * them and a round's fixed scalar cost was paid 5-6 times per window.  A probe
	 * costs more than a byte-map read, though: where most of the work is looking
	 * the second list's postings up (2.1 -> 3.8 ms for 2-term ANDs, 0.99 -> 1.71
	 * for 3-term ones) the spans stay; with more required terms the later, denser
	 * lists are mostly never looked at and the rounds dominate (5-term AND 0.595
	 * -> 0.508 ms).  Decided per query on the host (the class key), compiled in
	 * per instantiation: with both forms in one kernel the 5-term AND took 0.65 ms.
	 */
	constexpr bool HASH = HASHQ;
	__shared__ uint32_t s_hdoc[HASH ? SCANR_HT : 1];
	__shared__ uint8_t s_hlane[HASH ? SCANR_HT : 1];
	__shared__ uint8_t s_mark[HASH ? 1 : RW + WAVE];	/* + one always-zero dummy slot per lane */
	__shared__ uint8_t s_bits[WAVE];		/* presence mask of the driver lane's doc */
	__shared__ float s_imp[NT][WAVE];		/* [token][driver lane] */
	__shared__ uint32_t s_truth[8];
	__shared__ __attribute__((aligned(16))) uint32_t s_hist[MODE == MODE_BIG ? BIGK_BUCKETS : 4];

	const unsigned lane = threadIdx.x;
	const item_t item = A.items[A.item_base + blockIdx.x];
	const uint32_t q = item.q, g = item.g;
	const qmeta_t qm = A.qmeta[q];
	const dev_query_t *Q = &A.queries[q];
	const uint32_t nt = Q->nt;
	const uint32_t n_req = Q->n_req;		/* slots [0, n_req) are required */
	const uint64_t seg = (uint64_t)qm.seg_first + g;

	if constexpr (HASH) {
		for (uint32_t i = lane; i < SCANR_HT; i += WAVE) {
			s_hdoc[i] = 0xffffffffu;
		}
	} else {
		for (uint32_t i = lane; i < RW + WAVE; i += WAVE) {
			s_mark[i] = 0;
		}
	}
	auto hash_of = [](uint32_t doc) -> uint32_t {
		return (doc * 2654435761u) >> (32 - SCANR_HT_BITS);
	};
	if constexpr (MODE == MODE_BIG) {
		for (uint32_t i = lane; i < BIGK_BUCKETS; i += WAVE) {
			s_hist[i] = 0;
		}
	}
	s_bits[lane] = 0;
	if (lane < 8) {
		s_truth[lane] = Q->truth[lane];
	}
	WAVE_SYNC();

	/* wave-uniform per-slot state as in k_scan8 (one window per set) */
	const posting_t *pt[NT];
	int32_t ab[NT], lo[NT], pdoc[NT];
	uint32_t tok[NT];
	uint64_t vm[NT];
	constexpr int RING = SCANR_RING;
	uint32_t rp[NT];		/* ring position of the oldest window in flight */
#pragma unroll
	for (int t = 0; t < NT; t++) {
		rp[t] = 0;
	}
	uint32_t Ad[NT];		/* set A; the windows in flight live in AGPRs (bpair_*) */
	float Ai[NT];

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
		pdoc[t] = vm[t] ? __builtin_amdgcn_readlane((int)Ad[t], 63 - __builtin_clzll(vm[t])) : -1;
	};
	auto load_sets = [&](auto tc, int32_t hi_) {
		constexpr int t = decltype(tc)::value;
		pdoc[t] = -1;
		ab[t] = 0;
		vm[t] = 0;
		if (hi_ > lo[t]) {
			ab[t] = ((hi_ - 1) >> 6) << 6;
			const int32_t ia = max(ab[t] + (int32_t)lane, lo[t]);
			const int32_t ib = max(ab[t] - WAVE + (int32_t)lane, lo[t]);
			const posting_t pa = pt[t][min(ia, hi_ - 1)];
			Ad[t] = pa.doc; Ai[t] = pa.imp;
			(void)ib;
			/* the RING windows below set A, oldest first (bring_take) */
			static_for<RING>([&](auto rc) {
				constexpr int r = decltype(rc)::value;
				const int32_t ir = max(ab[t] - (r + 1) * WAVE + (int32_t)lane, lo[t]);
				bpair_request<t * RING + r>(&pt[t][min(ir, hi_ - 1)]);
			});
			rp[t] = 0;
			vm[t] = window_mask(ab[t], lo[t], hi_);
			refresh_pdoc(tc);
		}
	};
	/* set A is drained: wait for the oldest window in flight, take it over,
	 * request the window RING below it */
	auto rotate_sets = [&](auto tc) {
		constexpr int t = decltype(tc)::value;
		ab[t] -= WAVE;
		vm[t] = window_mask(ab[t], lo[t], 0x7fffffff);
		const posting_t *np = &pt[t][max(ab[t] - RING * WAVE + (int32_t)lane, lo[t])];
		bring_take<t, RING>(rp[t], RING - 1, Ad[t], Ai[t], np);
		rp[t] = (rp[t] + 1) & (RING - 1);
	};

	static_for<NT>([&](auto tc) {
		constexpr int t = decltype(tc)::value;
		int32_t hi0 = 0;
		pt[t] = A.post;
		lo[t] = 0;
		tok[t] = 0;
		if (t < (int)nt) {
			tok[t] = Q->slot_tok[t];
			pt[t] = A.post + Q->pbeg[tok[t]];
			const uint64_t cb = ((uint64_t)qm.seg_first + q + g) * NXSGPU_MAX_TOKENS + tok[t];
			lo[t] = (int32_t)A.cursors[cb];
			hi0 = (int32_t)A.cursors[cb + NXSGPU_MAX_TOKENS];
		}
		load_sets(tc, hi0);
	});

	float top = -INFINITY;
	float hint = (MODE == MODE_TOPK && A.k <= WAVE) ? range_hint(A, qm, g) :
	    MODE == MODE_BIG ? bigk_hint(A, qm, g) : -INFINITY;
	float thr = hint;
	const uint32_t kidx = (A.k >= 1 && A.k <= WAVE) ? A.k - 1 : WAVE - 1;
	const bool track = (MODE == MODE_TOPK) && A.k <= WAVE;
	/* MODE_BIG: candidates counted since the threshold was last read off the histogram */
	uint32_t big_since = 0;
	const uint32_t big_upd = bigk_update_every(A.k);
	uint32_t n_out = 0;
	bool ovf = false;
	bool done = false;		/* a required slot ran out: nothing below can match */
	const uint64_t out_base = (MODE == MODE_ALL) ? A.seg_off[seg] : seg * A.seg_cap;

	RSTAT(0, 1);
	while (!done && pdoc[0] >= 0) {
		RSTAT(1, 1);
		const int32_t dtop = pdoc[0];
		const uint32_t base = HASH ? 0u : (uint32_t)dtop & ~(uint32_t)(RW - 1);
		/* the driver's postings of this round: its whole window (HASH), or those
		 * of one RW-doc span */
		const uint64_t inm0 = HASH ? vm[0] : (vm[0] & ballot64(Ad[0] >= base));
		const bool in0 = lane_of(inm0);
		const uint32_t dd0 = in0 ? Ad[0] - base : RW + lane;	/* (byte map) */
		uint32_t hslot = 0;
		if constexpr (HASH) {
			/* insert: claim an empty slot, verify, move on (no atomics: the lanes
			 * of one LDS instruction are serialised, exactly one claim survives) */
			uint32_t h = hash_of(Ad[0]);
			uint64_t todo = inm0;
			while (todo) {
				const bool mine = lane_of(todo);
				if (mine && s_hdoc[h] == 0xffffffffu) {
					s_hdoc[h] = Ad[0];
				}
				WAVE_SYNC();
				const bool got = mine && s_hdoc[h] == Ad[0];
				if (got) {
					s_hlane[h] = (uint8_t)(lane + 1);
					hslot = h;
				}
				if (mine && !got) {
					h = (h + 1) & (SCANR_HT - 1);
				}
				todo &= ~ballot64(got);
				WAVE_SYNC();
			}
		}
		const uint32_t rlo = (uint32_t)__builtin_amdgcn_readlane((int)Ad[0], __builtin_ctzll(inm0));
		uint64_t alive = inm0;

		RSTAT(2, __popcll(inm0));
		vm[0] ^= inm0;
		if constexpr (!HASH) {
			s_mark[dd0] = in0 ? (uint8_t)(lane + 1) : (uint8_t)0;
		}
		s_bits[lane] = (uint8_t)(1u << tok[0]);
		s_imp[tok[0]][lane] = Ai[0];
		WAVE_SYNC();

		static_for<NT - 1>([&](auto jc) {
			constexpr int j = decltype(jc)::value + 1;
			using JC = std::integral_constant<int, j>;
			if (j < (int)nt && alive && !done) {
				RSTAT(3 + (j > 1 ? 1 : 0), 1);
				/* nothing above the driver's top doc can match: drop it unread */
				if (pdoc[j] > dtop) {
					for (int tries = 0; ; tries++) {
						vm[j] &= ~ballot64(Ad[j] > (uint32_t)dtop);
						if (vm[j] || ab[j] <= lo[j]) {
							break;		/* the boundary is in this window / list exhausted */
						}
						if (tries < 2) {
							RSTAT(5, 1);
							rotate_sets(JC());	/* stream a little ... */
							continue;
						}
						/* ... then jump: lane l probes the first posting of the
						 * l-th window below; the boundary is in the first one
						 * that starts at or below the driver's doc */
						const int32_t li = ab[j];	/* postings [lo, li) are unseen */
						const int32_t pi = max(li - (int32_t)(lane + 1) * WAVE, lo[j]);
						const uint32_t pv = pt[j][pi].doc;
						const uint64_t pm = ballot64(pv <= (uint32_t)dtop);
						int32_t nh;
						if (pm) {
							nh = min(li, max(li - (int32_t)__builtin_ctzll(pm) * WAVE, lo[j] + 1));
						} else {
							const int32_t far = max(li - WAVE * WAVE, lo[j]);
							nh = far > lo[j] ? wave_lower_bound(pt[j], lo[j], far, (uint32_t)dtop + 1) : lo[j];
						}
						RSTAT(6, 1);
						load_sets(JC(), nh);
						tries = 2;
						if (nh <= lo[j]) {
							break;
						}
					}
					refresh_pdoc(JC());
				}
				if (pdoc[j] < 0 && j < (int)n_req) {
					done = true;
				}
				if (pdoc[j] >= (int32_t)rlo) {
					/* look this slot's postings of the span up in the map */
					const uint32_t tj = tok[j];
					uint64_t left = vm[j];
					for (;;) {
						const uint64_t inm = left & ballot64(Ad[j] >= rlo);
						left ^= inm;
						if (inm) {
							uint32_t m = 0;
							if constexpr (HASH) {
								uint32_t h = hash_of(Ad[j]);
								uint64_t todo = inm;
								while (todo) {
									const bool mine = lane_of(todo);
									const uint32_t v = s_hdoc[h];
									const bool hit = mine && v == Ad[j];
									const bool miss = mine && v == 0xffffffffu;
									if (hit) {
										m = s_hlane[h];
									}
									h = (h + 1) & (SCANR_HT - 1);
									todo &= ~ballot64(hit || miss);
								}
							} else {
								const bool inl = lane_of(inm);
								const uint32_t dd = inl ? Ad[j] - base : RW + lane;
								m = s_mark[dd];
							}
							if (ballot64(m != 0)) {
								if (m != 0) {
									s_imp[tj][m - 1] = Ai[j];
									s_bits[m - 1] = (uint8_t)(s_bits[m - 1] | (1u << tj));
								}
							}
						}
						if (left == 0 && ab[j] > lo[j]) {
							RSTAT(7, 1);
							rotate_sets(JC());
							left = vm[j];
							continue;
						}
						break;
					}
					vm[j] = left;
					refresh_pdoc(JC());
					WAVE_SYNC();
				}
				if (j < (int)n_req) {
					/* driver lanes whose doc lacks this required term are out */
					const uint32_t b = s_bits[lane];
					const uint64_t ok = alive & ballot64(((b >> tok[j]) & 1) != 0);
					if constexpr (!HASH) {
						if (lane_of(alive ^ ok)) {
							s_mark[dd0] = 0;
						}
					}
					alive = ok;
					WAVE_SYNC();
				}
			}
		});

		if (alive) {
			RSTAT(8, 1);
			RSTAT(9, __popcll(alive));
			const bool al = lane_of(alive);
			const uint32_t mask = al ? s_bits[lane] : 0;
			const bool match = al && ((s_truth[mask >> 5] >> (mask & 31)) & 1);
			if (MODE == MODE_COUNT) {
				n_out += __popcll(ballot64(match));
			} else {
				float sc = 0.0f;
				/* token order (results.c:134-136) */
#pragma unroll
				for (int k = 0; k < NT; k++) {
					if ((mask >> k) & 1) {
						sc += s_imp[k][lane];
					}
				}
				const bool cand = match && (MODE == MODE_ALL || sc > thr);
				uint64_t bal = ballot64(cand);
				if (bal) {
					const uint32_t ne = __popcll(bal);
					if (MODE_FILTERS(MODE) && n_out + ne > A.seg_cap) {
						ovf = true;
					} else {
						const uint64_t above = (lane == 63) ? 0 : (bal >> (lane + 1));
						if (cand) {
							const uint64_t o = out_base + n_out + __popcll(above);
							A.cand_doc[o] = Ad[0];
							A.cand_sc[o] = sc;
						}
					}
					n_out += ne;
					if constexpr (MODE == MODE_BIG) {
						bigk_account(s_hist, A.k, big_upd, cand, sc, ne, big_since, hint, thr);
					}
					if (track) {
						while (bal) {
							const int L = 63 - __clzll(bal);
							bal &= ~(1ull << L);
							const float v = __shfl(sc, L);
							if (v > thr) {
								const uint32_t pos = __popcll(ballot64(top >= v));
								const float up = __shfl_up(top, 1);
								top = (lane < pos) ? top : (lane == pos ? v : up);
								thr = fmaxf(__shfl(top, kidx), hint);
							}
						}
					}
				}
			}
			if constexpr (!HASH) {
				if (al) {
					s_mark[dd0] = 0;
				}
			}
		}
		if constexpr (HASH) {
			if (in0) {
				s_hdoc[hslot] = 0xffffffffu;	/* the table is empty again */
			}
		}
		WAVE_SYNC();

		if (vm[0] == 0 && ab[0] > lo[0]) {
			rotate_sets(std::integral_constant<int, 0>());
		}
		refresh_pdoc(std::integral_constant<int, 0>());
	}
	if (MODE == MODE_TOPK && track && !ovf) {
		range_publish(A, seg, __shfl(top, kidx));
	}
	if constexpr (MODE == MODE_BIG) {
		if (!ovf) {
			bigk_publish(A, seg, s_hist, A.k);	/* lower bounds of this range's k-th, k/2-th ... best */
		}
	}
	if (lane == 0) {
		if (MODE != MODE_ALL) {
			A.seg_count[seg] = ovf ? 0 : n_out;
		}
		if (MODE_FILTERS(MODE) && ovf) {
			A.overflow[q] = 1;
		}
	}
}

/*
 * k_scanq: conjunctions whose required terms all have a BLOCK-PRESENCE BITMAP (one bit per
 * 64-doc block, nxsgpu_index::d_blkmap).  The reference intersects the terms' roaring bitmaps
 * before it looks at a single posting (get_expr_bitmap, search.c:118-174); k_scanr does that
 * on posting windows and still streams the driver list (0.65 GB per C3 step for 512 five-term
 * ANDs that return one result in total).  Here the summaries are intersected first:
 *
 *  1. lane L ANDs word (top - L) of the required terms' bitmaps (64 blocks = 4096 docs per
 *     word, 8 B per term): the surviving blocks -- around 1 % of them for C3's ANDs -- go to
 *     an LDS list in descending order;
 *  2. one LANE per surviving block: the driver's (slot 0: the shortest required list)
 *     postings inside the block -- one or two -- are found by a lower-bound search inside the
 *     span of ONE bitmap word (the term's rank directory, d_bmrank), and each of their docs is
 *     looked up in the other required lists the same way, shortest list first, until one
 *     lacks it: nearly every surviving block ends there, after two searches;
 *  3. a doc that holds every required term (rare) also gets the optional tokens' impacts,
 *     sums them in token order (results.c:134-136), applies the truth table and the
 *     threshold, and waits in an LDS list; a round's matches are sorted by descending doc,
 *     emitted, and fed to the top-k register.
 * Every token counts towards a doc's score whatever its role (search.c:240-253); only the
 * REQUIRED ones take part in steps 1 and 2.  Filter passes only (top-k, k <= 64; limits above that
 * when a handful of matches is expected: BIG); the exact passes of these queries take k_scanr.
 */
#define	SQ_CAP	64		/* surviving blocks per round: one per lane */
#define	SQ_MCAP	128		/* matches of a round */

/* BIG (limits > 64): every match is emitted -- the host sends a query here only if it expects a handful
 * (build_worklist) --, nothing is tracked or published: the histogram threshold of the other MODE_BIG
 * kernels needs >= k emitted docs before it says anything */
template <int NT, bool BIG = false>
__global__ void __launch_bounds__(WAVE)
k_scanq(const scan_args_t A)
{
	__shared__ uint32_t s_blk[SQ_CAP];
	__shared__ float s_imp[NT][WAVE];	/* [token][lane]: the impacts of the doc the lane is looking at */
	__shared__ uint32_t s_mdoc[SQ_MCAP];
	__shared__ float s_msc[SQ_MCAP];
	__shared__ uint32_t s_truth[8];

	const unsigned lane = threadIdx.x;
	const item_t item = A.items[A.item_base + blockIdx.x];
	const uint32_t q = item.q, g = item.g;
	const qmeta_t qm = A.qmeta[q];
	const dev_query_t *Q = &A.queries[q];
	const uint32_t nt = Q->nt;
	const uint32_t n_req = Q->n_req;	/* slots [0, n_req) are the required tokens, shortest list first */
	const uint64_t seg = (uint64_t)qm.seg_first + g;
	const uint32_t d_bot = (uint32_t)min((uint64_t)g * qm.group_docs, A.n_docs);
	const uint32_t d_top = (g + 1 == qm.n_groups) ? (uint32_t)A.n_docs :
	    (uint32_t)min((uint64_t)(g + 1) * qm.group_docs, A.n_docs);

	if (lane < 8) {
		s_truth[lane] = Q->truth[lane];
	}
	/* per SLOT (k_scanr's order: dev_query_t::slot_tok) */
	const posting_t *pt[NT];
	int32_t lo[NT], hi[NT];
	uint32_t tok[NT];
	const unsigned long long *bm[NT];
	const uint32_t *rk[NT];
	static_for<NT>([&](auto sc_) {
		constexpr int s = decltype(sc_)::value;
		pt[s] = A.post;
		lo[s] = hi[s] = 0;
		tok[s] = 0;
		bm[s] = NULL;
		rk[s] = NULL;
		if (s < (int)nt) {
			tok[s] = (uint32_t)__builtin_amdgcn_readfirstlane((int)Q->slot_tok[s]);
			pt[s] = A.post + Q->pbeg[tok[s]];
			const uint64_t cb = ((uint64_t)qm.seg_first + q + g) * NXSGPU_MAX_TOKENS + tok[s];
			lo[s] = (int32_t)A.cursors[cb];
			hi[s] = (int32_t)A.cursors[cb + NXSGPU_MAX_TOKENS];
			const uint32_t col = (uint32_t)__builtin_amdgcn_readfirstlane((int)Q->bm_col[tok[s]]);
			if (col != 0xffffffffu) {
				rk[s] = A.bmrank + (uint64_t)col * (A.bm_words + 1);
				if (s < (int)n_req) {
					bm[s] = (const unsigned long long *)A.blkmap + (uint64_t)col * A.bm_words;
				}
			}
		}
	});
	WAVE_SYNC();

	/* where doc `d` would be in slot s's list: a lower bound inside the span of d's bitmap word
	 * (slots without a directory: inside the range's cursors); per lane */
	auto find = [&](auto sc_, uint32_t d) -> int32_t {
		constexpr int s = decltype(sc_)::value;
		int32_t l = lo[s], h = hi[s];
		if (rk[s]) {
			/* (the directory counts from the list's first posting, like the cursors) */
			const uint32_t *r = rk[s] + (d >> 12);
			l = max(l, (int32_t)r[0]);
			h = min(h, (int32_t)r[1]);
		}
		while (l < h) {
			const int32_t mid = (l + h) >> 1;
			if (pt[s][mid].doc < d) {
				l = mid + 1;
			} else {
				h = mid;
			}
		}
		return l;
	};

	float top = -INFINITY;
	const float hint = BIG ? 0.0f : range_hint(A, qm, g);	/* (0: scores are > 0, like the top range's bigk_hint) */
	float thr = hint;
	const uint32_t kidx = BIG ? 0u : A.k - 1;	/* 1 <= k <= 64 (host) */
	uint32_t n_out = 0;
	bool ovf = false;
	const uint64_t out_base = seg * A.seg_cap;

	if (d_top > d_bot && n_req >= 1 && hi[0] > lo[0]) {
		const uint32_t b0 = d_bot >> 6, b1 = (d_top - 1) >> 6;	/* first and last block of the range */
		const int32_t w_lo = (int32_t)(b0 >> 6), w_hi = (int32_t)(b1 >> 6);
		for (int32_t wtop = w_hi; wtop >= w_lo && !ovf; wtop -= WAVE) {
			/* 1. my word of the intersection */
			const int32_t w = wtop - (int32_t)lane;
			unsigned long long m = 0;
			if (w >= w_lo) {
				m = ~0ull;
				static_for<NT>([&](auto sc_) {
					constexpr int s = decltype(sc_)::value;
					if (bm[s]) {
						m &= bm[s][w];
					}
				});
				if (w == w_hi && (b1 & 63) != 63) {
					m &= (2ull << (b1 & 63)) - 1;
				}
				if (w == w_lo) {
					m &= ~0ull << (b0 & 63);
				}
			}
			/* rounds of up to SQ_CAP surviving blocks, highest first */
			while (ballot64(m != 0) && !ovf) {
				const uint32_t cnt = (uint32_t)__popcll(m);
				uint32_t pre = cnt;		/* inclusive prefix over the lanes (lane 0 = highest word) */
#pragma unroll
				for (int o = 1; o < WAVE; o <<= 1) {
					const uint32_t v = (uint32_t)__shfl_up((int)pre, o);
					if ((int)lane >= o) {
						pre += v;
					}
				}
				const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)pre, 63);
				const uint32_t excl = pre - cnt;
				const uint32_t take = excl >= SQ_CAP ? 0u : min(cnt, (uint32_t)SQ_CAP - excl);
				for (uint32_t i = 0; i < take; i++) {
					const uint32_t bit = 63u - (uint32_t)__builtin_clzll(m);
					s_blk[excl + i] = ((uint32_t)w << 6) | bit;
					m &= ~(1ull << bit);
				}
				const uint32_t nb = min(total, (uint32_t)SQ_CAP);
				WAVE_SYNC();

				/* 2. my block: the driver's postings in it, each doc through the other required lists */
				uint32_t n_match = 0;		/* wave-uniform: matches of this round in s_mdoc / s_msc */
				const bool have = lane < nb;
				const uint32_t base = have ? s_blk[lane] << 6 : 0u;
				int32_t ip = have ? find(std::integral_constant<int, 0>(), base) : hi[0];
				int32_t ie = ip;		/* [ip, ie): the driver's postings of my block, ascending docs */
				if (have) {
					while (ie < hi[0] && ie - ip < WAVE && pt[0][ie].doc < base + WAVE) {
						ie++;
					}
				}
				/* highest doc first (the emission order inside a block) */
				while (ballot64(ie > ip) && !ovf) {
					const bool act = ie > ip;
					uint32_t d = 0, pmask = 0;
					bool ok = act;
					if (act) {
						ie--;
						const posting_t p = pt[0][ie];
						d = p.doc;
						s_imp[tok[0]][lane] = p.imp;
						pmask = 1u << tok[0];
						ok = d >= d_bot && d < d_top;
					}
					static_for<NT - 1>([&](auto jc) {
						constexpr int s = decltype(jc)::value + 1;
						using SC = std::integral_constant<int, s>;
						if (s < (int)nt && ballot64(ok)) {
							if (ok) {
								const int32_t i = find(SC(), d);
								bool f = false;
								if (i < hi[s]) {
									const posting_t p = pt[s][i];
									f = p.doc == d;
									if (f) {
										s_imp[tok[s]][lane] = p.imp;
										pmask |= 1u << tok[s];
									}
								}
								if (s < (int)n_req && !f) {
									ok = false;	/* a required term is missing: the doc cannot match */
								}
							}
						}
					});
					/* 3. the docs that hold every required term */
					bool match = ok && ((s_truth[pmask >> 5] >> (pmask & 31)) & 1);
					float sc = 0.0f;
					if (match) {
						/* token order (results.c:134-136) */
#pragma unroll
						for (int k = 0; k < NT; k++) {
							if ((pmask >> k) & 1) {
								sc += s_imp[k][lane];
							}
						}
						match = sc > thr;
					}
					const uint64_t bal = ballot64(match);
					if (bal) {
						const uint32_t ne = __popcll(bal);
						if (n_match + ne > SQ_MCAP) {
							ovf = true;
						} else if (match) {
							const uint32_t o = n_match + lanes_below(bal);
							s_mdoc[o] = d;
							s_msc[o] = sc;
						}
						n_match += ne;
					}
				}
				WAVE_SYNC();
				/* the round's matches: descending doc, then the common filter */
				if (n_match && !ovf) {
					constexpr int MC = SQ_MCAP / WAVE;
					uint32_t pd[MC], rkk[MC];
					float ps[MC];
#pragma unroll
					for (int c = 0; c < MC; c++) {
						const uint32_t e = c * WAVE + lane;
						pd[c] = e < n_match ? s_mdoc[e] : 0;
						ps[c] = e < n_match ? s_msc[e] : 0.0f;
						rkk[c] = 0;
					}
					WAVE_SYNC();
#pragma unroll
					for (int cj = 0; cj < MC; cj++) {
						const uint32_t nj = n_match > (uint32_t)cj * WAVE ? min(n_match - cj * WAVE, (uint32_t)WAVE) : 0u;
						for (uint32_t j = 0; j < nj; j++) {
							const uint32_t dj = __builtin_amdgcn_readlane((int)pd[cj], j);
#pragma unroll
							for (int c = 0; c < MC; c++) {
								rkk[c] += dj > pd[c];		/* (docs are distinct) */
							}
						}
					}
#pragma unroll
					for (int c = 0; c < MC; c++) {
						const uint32_t e = c * WAVE + lane;
						if (e < n_match) {
							s_mdoc[rkk[c]] = pd[c];
							s_msc[rkk[c]] = ps[c];
						}
					}
					WAVE_SYNC();
					for (uint32_t off = 0; off < n_match && !ovf; off += WAVE) {
						const uint32_t e = off + lane;
						const bool valid = e < n_match;
						const uint32_t d = valid ? s_mdoc[e] : 0;
						const float sc = valid ? s_msc[e] : 0.0f;
						const bool cand = valid && sc > thr;
						uint64_t bal = ballot64(cand);
						if (!bal) {
							continue;
						}
						const uint32_t ne = __popcll(bal);
						if (n_out + ne > A.seg_cap) {
							ovf = true;
						} else if (cand) {
							/* lanes are in descending doc order */
							const uint64_t o = out_base + n_out + lanes_below(bal);
							A.cand_doc[o] = d;
							A.cand_sc[o] = sc;
						}
						n_out += ne;
						while (!BIG && bal) {
							const int L = __builtin_ctzll(bal);
							bal &= bal - 1;
							const float v = __shfl(sc, L);
							if (v > thr) {
								const uint32_t pos = __popcll(ballot64(top >= v));
								const float up = __shfl_up(top, 1);
								top = (lane < pos) ? top : (lane == pos ? v : up);
								thr = fmaxf(__shfl(top, kidx), hint);
							}
						}
					}
					WAVE_SYNC();
				}
			}
		}
	}
	if (!BIG && !ovf) {
		range_publish(A, seg, __shfl(top, kidx));
	}
	if (lane == 0) {
		A.seg_count[seg] = ovf ? 0 : n_out;
		if (ovf) {
			A.overflow[q] = 1;
		}
	}
}


/* ---- launchers ------------------------------------------------------ */

void
nxs_launch_cursors(const scan_args_t &a, const uint32_t *d_bnd_q, uint32_t n_bnd, hipStream_t st)
{
	const uint64_t threads = (uint64_t)n_bnd * NXSGPU_MAX_TOKENS;

	if (n_bnd) {
		hipLaunchKernelGGL(k_cursors, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st,
		    a.post, a.queries, a.qmeta, d_bnd_q, n_bnd, a.n_docs, (uint32_t *)a.cursors);
	}
}

void
nxs_launch_scan1(int mode, unsigned grid, hipStream_t st, const scan_args_t &a)
{
	switch (mode) {
	case MODE_TOPK: hipLaunchKernelGGL((k_scan1<MODE_TOPK>), dim3(grid), dim3(WAVE), 0, st, a); break;
	case MODE_BIG: hipLaunchKernelGGL((k_scan1<MODE_BIG>), dim3(grid), dim3(WAVE), 0, st, a); break;
	case MODE_COUNT: hipLaunchKernelGGL((k_scan1<MODE_COUNT>), dim3(grid), dim3(WAVE), 0, st, a); break;
	default: hipLaunchKernelGGL((k_scan1<MODE_ALL>), dim3(grid), dim3(WAVE), 0, st, a); break;
	}
}

void
nxs_launch_scanq(uint32_t nt_bucket, unsigned grid_, hipStream_t st, const scan_args_t &a)
{
	const dim3 grid(grid_), block(WAVE);

	if (a.k > WAVE) {
		switch (nt_bucket) {
		case 2: hipLaunchKernelGGL((k_scanq<2, true>), grid, block, 0, st, a); break;
		case 3: hipLaunchKernelGGL((k_scanq<3, true>), grid, block, 0, st, a); break;
		case 5: hipLaunchKernelGGL((k_scanq<5, true>), grid, block, 0, st, a); break;
		default: hipLaunchKernelGGL((k_scanq<8, true>), grid, block, 0, st, a); break;
		}
		return;
	}
	switch (nt_bucket) {
	case 2: hipLaunchKernelGGL((k_scanq<2>), grid, block, 0, st, a); break;
	case 3: hipLaunchKernelGGL((k_scanq<3>), grid, block, 0, st, a); break;
	case 5: hipLaunchKernelGGL((k_scanq<5>), grid, block, 0, st, a); break;
	default: hipLaunchKernelGGL((k_scanq<8>), grid, block, 0, st, a); break;
	}
}

template <int MODE>
static void
launch_scanr_mode(uint32_t nt_bucket, bool hash, const dim3 grid, hipStream_t st, const scan_args_t &a)
{
	const dim3 block(WAVE);

	switch (nt_bucket) {
	case 2: hipLaunchKernelGGL((k_scanr<MODE, 2>), grid, block, 0, st, a); break;
	case 3: hipLaunchKernelGGL((k_scanr<MODE, 3>), grid, block, 0, st, a); break;
	case 5:
		if (hash) { hipLaunchKernelGGL((k_scanr<MODE, 5, true>), grid, block, 0, st, a); }
		else { hipLaunchKernelGGL((k_scanr<MODE, 5>), grid, block, 0, st, a); }
		break;
	default:
		if (hash) { hipLaunchKernelGGL((k_scanr<MODE, 8, true>), grid, block, 0, st, a); }
		else { hipLaunchKernelGGL((k_scanr<MODE, 8>), grid, block, 0, st, a); }
		break;
	}
}

/* k_scanr; hash: rounds of whole driver windows (queries with >= 4 required terms) */
void
nxs_launch_scanr(int mode, uint32_t nt_bucket, bool hash, unsigned grid, hipStream_t st, const scan_args_t &a)
{
	switch (mode) {
	case MODE_TOPK: launch_scanr_mode<MODE_TOPK>(nt_bucket, hash, dim3(grid), st, a); break;
	case MODE_BIG: launch_scanr_mode<MODE_BIG>(nt_bucket, hash, dim3(grid), st, a); break;
	case MODE_COUNT: launch_scanr_mode<MODE_COUNT>(nt_bucket, hash, dim3(grid), st, a); break;
	default: launch_scanr_mode<MODE_ALL>(nt_bucket, hash, dim3(grid), st, a); break;
	}
}



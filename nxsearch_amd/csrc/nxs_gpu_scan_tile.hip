/*
 * nxs_gpu_scan_tile.hip -- accumulator-tile scan kernels: k_scan (generic, <= 32 tokens), k_scan8 (<= 8 tokens)
 * (MI355X / gfx950 query path of nxsearch; see nxs_gpu_int.h for the map of the files)
 */
#include "nxs_gpu_int.h"
#include "nxs_gpu_dev.h"

template <int NTMAX, typename MaskT, int MODE>
__global__ void __launch_bounds__(WAVE)
k_scan(const scan_args_t A)
{
	/* per-wavefront LDS tile */
	__shared__ float s_acc[TILE_W];
	__shared__ MaskT s_mask[TILE_W];
	__shared__ uint64_t s_hi[NTMAX], s_lo[NTMAX];
	__shared__ int64_t s_pdoc[NTMAX];	/* doc of posting hi-1, or -1 */
	__shared__ uint32_t s_truth[8];
	__shared__ uint8_t s_prog[NXSGPU_MAX_PROG];
	__shared__ __attribute__((aligned(16))) uint32_t s_hist[MODE == MODE_BIG ? BIGK_BUCKETS : 4];

	const unsigned lane = threadIdx.x;
	const item_t item = A.items[A.item_base + blockIdx.x];
	const uint32_t q = item.q, g = item.g;
	const qmeta_t qm = A.qmeta[q];
	const dev_query_t *Q = &A.queries[q];
	const uint32_t nt = Q->nt;
	const posting_t *__restrict__ post = A.post;
	const uint64_t seg = (uint64_t)qm.seg_first + g;

	for (uint32_t i = lane; i < TILE_W; i += WAVE) {
		s_acc[i] = 0.0f;
		s_mask[i] = 0;
	}
	if constexpr (MODE == MODE_BIG) {
		for (uint32_t i = lane; i < BIGK_BUCKETS; i += WAVE) {
			s_hist[i] = 0;
		}
	}
	if (lane < 8) {
		s_truth[lane] = Q->truth[lane];
	}
	if (sizeof(MaskT) > 1) {
		for (uint32_t i = lane; i < Q->prog_len; i += WAVE) {
			s_prog[i] = Q->prog[i];
		}
	}
	/* initial cursors of the group's doc range: lane t -> hi, lane 32+t -> lo */
	if (lane < nt || (lane >= 32 && lane - 32 < nt)) {
		const uint32_t t = lane & 31;
		const uint64_t pb = Q->pbeg[t], pe = Q->pend[t];
		const uint64_t cb = ((uint64_t)qm.seg_first + q + g) * NXSGPU_MAX_TOKENS + t;
		(void)pe;
		if (lane < 32) {
			s_hi[t] = pb + A.cursors[cb + NXSGPU_MAX_TOKENS];
		} else {
			s_lo[t] = pb + A.cursors[cb];
		}
	}
	__syncthreads();
	if (lane < nt) {
		const uint64_t h = s_hi[lane], l = s_lo[lane];
		s_pdoc[lane] = (h > l) ? (int64_t)post[h - 1].doc : -1;
	}
	__syncthreads();

	/* running top-k of the scores this wavefront has seen: lane i holds the
	 * i-th largest; thr = k-th largest (or -inf).  Everything the global
	 * heap replay could accept is > thr (see DESIGN.md "candidate filter"). */
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
	const uint64_t out_base = (MODE == MODE_ALL) ? A.seg_off[seg] : seg * A.seg_cap;

	for (;;) {
		/* next non-empty tile = tile of the largest unconsumed doc */
		int64_t md = -1;
		for (uint32_t t = 0; t < nt; t++) {
			md = max(md, s_pdoc[t]);
		}
		if (md < 0) {
			break;
		}
		const uint32_t base = (uint32_t)((uint64_t)md / TILE_W) * TILE_W;

		/* accumulate: tokens strictly in token-list order (results.c:134-136) */
		for (uint32_t t = 0; t < nt; t++) {
			if (s_pdoc[t] < (int64_t)base) {
				continue;
			}
			uint64_t hi = s_hi[t];
			const uint64_t lo = s_lo[t];
			int64_t pdoc = -1;
			while (hi > lo) {
				const int64_t i = (int64_t)hi - WAVE + lane;
				const bool valid = i >= (int64_t)lo;
				posting_t p;
				p.doc = 0; p.imp = 0.0f;
				if (valid) {
					p = post[i];
				}
				const bool in = valid && p.doc >= base;
				const uint64_t bal = ballot64(in);
				const uint32_t c = __popcll(bal);
				if (in) {
					const uint32_t d = p.doc - base;
					s_acc[d] += p.imp;
					if (sizeof(MaskT) == 1) {
						s_mask[mask_byte(d)] |= (MaskT)(1u << t);
					} else {
						s_mask[d] |= (MaskT)(1u << t);
					}
				}
				hi -= c;
				if (c < WAVE) {
					if (hi > lo) {
						pdoc = (int64_t)(uint32_t)__shfl((int)p.doc, WAVE - 1 - c);
					}
					break;
				}
			}
			__syncthreads();	/* single wavefront: orders the LDS updates */
			if (lane == 0) {
				s_hi[t] = hi;
				s_pdoc[t] = pdoc;
			}
			__syncthreads();
		}

		/* scan the tile in DESCENDING doc order (results.c:143-147 prepends,
		 * so the reference feeds its heap in descending doc id) */
		if (sizeof(MaskT) == 1) {
			uint32_t *mask32 = (uint32_t *)s_mask;
			for (int s = TILE_W / 256 - 1; s >= 0; s--) {
				const uint32_t mw = mask32[s * WAVE + lane];
				if (ballot64(mw != 0) == 0) {
					continue;
				}
				if (mw) {
					mask32[s * WAVE + lane] = 0;
				}
				for (int j = 3; j >= 0; j--) {
					const uint32_t m = (mw >> (8 * j)) & 0xff;
					if (ballot64(m != 0) == 0) {
						continue;
					}
					const uint32_t d = s * 256 + j * 64 + lane;
					float sc = 0.0f;
					if (m) {
						sc = s_acc[d];
						s_acc[d] = 0.0f;
					}
					const bool match = m && ((s_truth[m >> 5] >> (m & 31)) & 1);
					if (MODE == MODE_COUNT) {
						n_out += __popcll(ballot64(match));
						continue;
					}
					const bool cand = match && (sc > thr);
					uint64_t bal = ballot64(cand);
					if (!bal) {
						continue;
					}
					const uint32_t ne = __popcll(bal);
					if (MODE_FILTERS(MODE) && n_out + ne > A.seg_cap) {
						ovf = true;
					} else {
						/* slot = number of candidate lanes above me */
						const uint64_t above = (lane == 63) ? 0 : (bal >> (lane + 1));
						if (cand) {
							const uint64_t o = out_base + n_out + __popcll(above);
							A.cand_doc[o] = base + d;
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
		} else {
			for (int s = TILE_W / WAVE - 1; s >= 0; s--) {
				const uint32_t d = s * WAVE + lane;
				const uint32_t m = s_mask[d];
				if (ballot64(m != 0) == 0) {
					continue;
				}
				float sc = 0.0f;
				if (m) {
					sc = s_acc[d];
					s_acc[d] = 0.0f;
					s_mask[d] = 0;
				}
				const bool match = m && eval_prog(s_prog, Q->prog_len, m);
				if (MODE == MODE_COUNT) {
					n_out += __popcll(ballot64(match));
					continue;
				}
				const bool cand = match && (sc > thr);
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
						A.cand_doc[o] = base + d;
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
		__syncthreads();
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
 * k_scan8: the <= 8 token tile path.  Same contract as k_scan, restructured
 * for memory-level parallelism, sparse tiles and conjunctive queries:
 *  - every term streams its list through two register sets of K 64-posting
 *    windows: set A (the 64K-aligned slice holding posting hi-1, being
 *    consumed) and set B (the slice below it, K loads in flight).  Each
 *    posting is loaded from memory exactly once; there is no per-tile search;
 *  - LDS updates are plain read-add-write in token order: one wavefront's DS
 *    operations execute in issue order and a doc occurs once per term, so the
 *    f32 sum order is the reference's (DS atomics were measured 10x slower);
 *  - the old mask byte tells a doc's FIRST touch in the tile; first-touched
 *    docs go to a small LDS list, and a tile with few touched docs is scanned
 *    through that list (cost ~ touched docs, not tile width).  Its candidates
 *    are rank-sorted by doc before they are appended, so a segment still is in
 *    descending doc order.  Dense tiles (list overflow) and tiles with more
 *    than 64 candidates take the ordered full scan;
 *  - scores only grow while a tile is accumulated, so if no value written in
 *    the tile (to a doc holding every required term) beat the candidate
 *    threshold the tile is just wiped;
 *  - terms that every matching doc must contain (`req`, from the truth table)
 *    drive the tile choice: the next tile is that of the LOWEST of their
 *    highest remaining docs, everything above it is skipped with a 64-ary
 *    search instead of being streamed, and the wavefront stops as soon as one
 *    of them is exhausted -- the device analogue of intersecting the bitmaps
 *    before scoring (search.c:118-174).
 */

template <int MODE, int NT, int MM>
__global__ void __launch_bounds__(WAVE)
k_scan8(const scan_args_t A)
{
	/*
	 * MM = 0: general boolean query, presence-mask byte per doc.
	 * MM = 1: pure OR (every non-empty mask matches): a doc matches iff it was
	 *         touched, i.e. iff its score is > 0 -- no mask array.
	 * MM = 2: "a AND b" (exactly two tokens): no mask array either; token 0
	 *         stores +score, token 1 only updates docs with a positive entry
	 *         and stores -score (scores are positive; negation and fabs are
	 *         exact), a doc matches iff its entry is negative.  One sign bit
	 *         cannot chain three tokens, those use MM = 0.
	 */
	constexpr bool HASMASK = MM == 0;
	constexpr bool ANDM = MM == 2;
	__shared__ float s_acc[TILE_W + WAVE];		/* + one dummy slot per lane */
	/* HASMASK = false: pure-OR queries (every non-empty presence mask matches):
	 * a doc matches iff it was touched, i.e. iff its score is > 0; no mask array */
	__shared__ uint8_t s_mask8[HASMASK ? TILE_W + WAVE : 4];
	__shared__ uint16_t s_list[LIST_CAP + WAVE];	/* + slack: appends are clamped, not branched */
	__shared__ uint32_t s_cd[TCAND_CAP];
	__shared__ float s_cs[TCAND_CAP];
	__shared__ uint32_t s_truth[8];
	__shared__ __attribute__((aligned(16))) uint32_t s_hist[MODE == MODE_BIG ? BIGK_BUCKETS : 4];

	constexpr int KSH = NT <= 1 ? 3 : NT <= 2 ? 2 : 0;
	constexpr int K = 1 << KSH;
	constexpr int SW = WAVE * K;

	const unsigned lane = threadIdx.x;
	item_t item;
	if (A.flags & 2) {
		/* second chance of the ranges that overflowed on the mask path */
		if (blockIdx.x >= min(*A.retry_count, A.retry_cap)) {
			return;
		}
		item = A.retry_items[blockIdx.x];
	} else {
		item = A.items[A.item_base + blockIdx.x];
	}
	const uint32_t q = item.q, g = item.g;
	const qmeta_t qm = A.qmeta[q];
	const dev_query_t *Q = &A.queries[q];
	const uint32_t nt = Q->nt;
	/* terms every matching doc must contain; a pure-OR query (MM = 1) has none,
	 * which removes the whole skip logic from that instantiation */
	const uint32_t req = (MM == 1) ? 0 : Q->req;
	const uint64_t seg = (uint64_t)qm.seg_first + g;
	uint32_t *s_mask32 = (uint32_t *)s_mask8;

	for (uint32_t i = lane; i < TILE_W + WAVE; i += WAVE) {
		s_acc[i] = 0.0f;
		if (HASMASK) {
			s_mask8[i] = 0;
		}
	}
	if constexpr (MODE == MODE_BIG) {
		for (uint32_t i = lane; i < BIGK_BUCKETS; i += WAVE) {
			s_hist[i] = 0;
		}
	}
	if (lane < 8) {
		s_truth[lane] = Q->truth[lane];
	}
	WAVE_SYNC();

	/*
	 * Wave-uniform per-term state, kept scalar: ab = list index of lane 0 of
	 * set A's window 0; vm[k] = lanes of window k not consumed yet (a 64-bit
	 * mask: consuming the in-tile lanes is one s_andn2, the next highest doc
	 * one s_flbit + v_readlane); lo = first posting of this wavefront's doc
	 * range; pdoc = doc of the highest unconsumed posting or -1.
	 */
	const posting_t *pt[NT];
	int32_t ab[NT], lo[NT], pdoc[NT];
	uint64_t vm[NT][K];
	uint32_t Ad[NT][K], Bd[NT][K];
	float Ai[NT][K], Bi[NT][K];
	/* AP (one window per set): set B is the hidden prefetch register pair of
	 * bpair_request()/bpair_take(); Bd/Bi are unused then */
	constexpr bool AP = K == 1 && !ANDM;
	/* AP: windows in flight per term below set A, and the ring position of the
	 * oldest (bring_take) */
	constexpr int RING = MODE == MODE_BIG ? SCAN8_RING_BIG : SCAN8_RING_MAX;
	uint32_t rp[NT];
	ring_stamps<RING> rst[NT];	/* issue stamps of the ring loads (vm_wait_younger) */
	uint32_t vseq = 0;
#pragma unroll
	for (int t = 0; t < NT; t++) {
#pragma unroll
		for (int r = 0; r < RING; r++) {
			rst[t].st[r] = 0;
		}
	}

	/* lanes of the window starting at list index wb that lie in [lo_, hi_) */
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
		pdoc[t] = -1;
#pragma unroll
		for (int k = K - 1; k >= 0; k--) {
			if (pdoc[t] < 0 && vm[t][k]) {
				pdoc[t] = __builtin_amdgcn_readlane((int)Ad[t][k], 63 - __builtin_clzll(vm[t][k]));
			}
		}
	};
	/* (re)load both register sets of term t so that postings [lo, hi_) are
	 * the unconsumed ones */
	auto load_sets = [&](auto tc, int32_t hi_) {
		constexpr int t = decltype(tc)::value;
		pdoc[t] = -1;
		ab[t] = 0;
#pragma unroll
		for (int k = 0; k < K; k++) {
			vm[t][k] = 0;
		}
		if (hi_ > lo[t]) {
			ab[t] = ((hi_ - 1) >> (6 + KSH)) << (6 + KSH);
#pragma unroll
			for (int k = 0; k < K; k++) {
				/* clamped, unpredicated loads: validity lives in vm */
				const int32_t ia = max(ab[t] + k * WAVE + (int32_t)lane, lo[t]);
				const int32_t ib = max(ab[t] - SW + k * WAVE + (int32_t)lane, lo[t]);
				const posting_t pa = pt[t][min(ia, hi_ - 1)];
				Ad[t][k] = pa.doc; Ai[t][k] = pa.imp;
				if constexpr (AP) {
					/* the RING windows below set A, oldest first */
					static_for<RING>([&](auto rc) {
						constexpr int r = decltype(rc)::value;
						const int32_t ir = max(ab[t] - (r + 1) * WAVE + (int32_t)lane, lo[t]);
						bpair_request<t * RING + r>(&pt[t][min(ir, hi_ - 1)]);
						rst[t].st[r] = vseq++;
					});
					rp[t] = 0;
					(void)ib;
				} else {
					const posting_t pb = pt[t][min(ib, hi_ - 1)];
					Bd[t][k] = pb.doc; Bi[t][k] = pb.imp;
				}
				vm[t][k] = window_mask(ab[t] + k * WAVE, lo[t], hi_);
			}
			refresh_pdoc(tc);
		}
	};
	/* set A is drained: take over set B, put K new loads in flight */
	auto rotate_sets = [&](auto tc) {
		constexpr int t = decltype(tc)::value;
		if constexpr (AP) {
			ab[t] -= WAVE;
			vm[t][0] = window_mask(ab[t], lo[t], 0x7fffffff);
			const posting_t *np = &pt[t][max(ab[t] - RING * WAVE + (int32_t)lane, lo[t])];
			bring_take<t, RING>(rp[t], rst[t].younger(vseq), Ad[t][0], Ai[t][0], np);
			rst[t].rotate(vseq++);
			rp[t] = (rp[t] + 1) & (RING - 1);
			return;
		}
		ab[t] -= SW;
#pragma unroll
		for (int k = 0; k < K; k++) {
			Ad[t][k] = Bd[t][k];
			Ai[t][k] = Bi[t][k];
			vm[t][k] = window_mask(ab[t] + k * WAVE, lo[t], 0x7fffffff);
		}
#pragma unroll
		for (int k = 0; k < K; k++) {
			const int32_t ib = max(ab[t] - SW + k * WAVE + (int32_t)lane, lo[t]);
			const posting_t pb = pt[t][ib];
			Bd[t][k] = pb.doc; Bi[t][k] = pb.imp;
		}
	};

	static_for<NT>([&](auto tc) {
		constexpr int t = decltype(tc)::value;
		int32_t hi0 = 0;
		pt[t] = A.post;
		lo[t] = 0;
		if (t < (int)nt) {
			const int32_t n = (int32_t)(Q->pend[t] - Q->pbeg[t]);
			pt[t] = A.post + Q->pbeg[t];
			/* cursors of this wavefront's doc range [dlo, dhi): k_cursors */
			const uint64_t cb = ((uint64_t)qm.seg_first + q + g) * NXSGPU_MAX_TOKENS + t;
			(void)n;
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
	const uint64_t out_base = (MODE == MODE_ALL) ? A.seg_off[seg] : seg * A.seg_cap;

	uint32_t tile_no = 0;
	for (;;) {
		int32_t md = -1, rq = 0x7fffffff;
#pragma unroll
		for (int t = 0; t < NT; t++) {
			md = max(md, pdoc[t]);
			if (t < (int)nt && ((req >> t) & 1)) {
				rq = min(rq, pdoc[t]);
			}
		}
		if (md < 0 || rq < 0) {
			break;		/* all consumed, or a required term ran out */
		}
		if constexpr (MODE == MODE_BIG) {
			/* limits > 64: ranges are long (>= 32 x limit postings) and higher ranges
			 * finish meanwhile: look at what they have published every 64 tiles */
			if ((++tile_no & 63) == 0) {
				hint = fmaxf(hint, bigk_hint(A, qm, g));
				thr = fmaxf(thr, hint);
			}
		}
		const uint32_t base = ((uint32_t)(req ? rq : md) / TILE_W) * TILE_W;

		if (req && md >= (int32_t)(base + TILE_W)) {
			/* skip, unscored, everything above this tile: none of it can
			 * match (a required term has nothing up there) */
			const uint32_t bound = base + TILE_W;
			static_for<NT>([&](auto tc) {
				constexpr int t = decltype(tc)::value;
				if (t < (int)nt && pdoc[t] >= (int32_t)bound) {
					uint64_t left = 0;
#pragma unroll
					for (int k = 0; k < K; k++) {
						vm[t][k] &= ~ballot64(Ad[t][k] >= bound);
						left |= vm[t][k];
					}
					if (left) {
						refresh_pdoc(tc);	/* the boundary was inside set A */
					} else {
						/* the whole set is above it: jump */
						const int32_t li = max(lo[t], ab[t]);
						const int32_t nh = li > lo[t] ? wave_lower_bound(pt[t], lo[t], li, bound) : lo[t];
						load_sets(tc, nh);
					}
				}
			});
			/* the tile is worth scoring only if every required term
			 * still reaches it */
			bool reach = true;
#pragma unroll
			for (int t = 0; t < NT; t++) {
				if (t < (int)nt && ((req >> t) & 1) && pdoc[t] < (int32_t)base) {
					reach = false;
				}
			}
			if (!reach) {
				continue;
			}
		}

		uint32_t n_list = 0;
		/* largest accumulator value written in this tile, as its bit pattern:
		 * the values are sums of positive impacts (>= +0), for which unsigned
		 * order is float order and the max is one v_max_u32 */
		uint32_t tmax = 0;

		/* accumulate, tokens strictly in token-list order (results.c:134-136) */
		static_for<NT>([&](auto tc) {
			constexpr int t = decltype(tc)::value;
			if constexpr (K == 1 && !ANDM) {
				/*
				 * One window per set (3+ tokens): straight-line read-add-write.
				 * pdoc[t] >= base implies t < nt (unused terms keep pdoc = -1)
				 * and that the top unconsumed posting is in this tile.
				 */
				if (pdoc[t] >= (int32_t)base) {
					uint64_t left = vm[t][0];
					for (;;) {
						const uint64_t inm = left & ballot64(Ad[t][0] >= base);
						left ^= inm;
						if (inm) {
							const bool inl = lane_of(inm);
							/* lanes without an in-tile posting add 0 to a private
							 * dummy slot: no exec-mask juggling */
							const uint32_t dd = inl ? Ad[t][0] - base : TILE_W + lane;
							const float a0 = s_acc[dd];
							const uint32_t m0 = HASMASK ? s_mask8[dd] : 0;
							const float v = a0 + (inl ? Ai[t][0] : 0.0f);
							uint64_t fb;	/* lanes touching their doc first in this tile */
							s_acc[dd] = v;
							if (HASMASK) {
								const uint32_t bits = m0 | (inl ? (1u << t) : 0u);
								s_mask8[dd] = (uint8_t)bits;
								/* only docs that already hold every required
								 * term can become candidates */
								if ((bits & req) == req) {
									tmax = max(tmax, __float_as_uint(v));
								}
								fb = inm & ballot64(m0 == 0);
							} else {
								tmax = max(tmax, __float_as_uint(v));
								fb = inm & ballot64(a0 == 0.0f);
							}
							if (lane_of(fb)) {
								s_list[min(n_list, (uint32_t)LIST_CAP) + lanes_below(fb)] = (uint16_t)dd;
							}
							n_list += __popcll(fb);
						}
						if (left == 0 && ab[t] > lo[t]) {
							/* the whole window was in the tile and postings remain */
							rotate_sets(tc);
							left = vm[t][0];
							continue;
						}
						break;
					}
					vm[t][0] = left;
					refresh_pdoc(tc);
				}
			} else if (t < (int)nt && pdoc[t] >= (int32_t)base) {
				for (;;) {
					uint64_t inm[K];
					float a0[K];
					uint32_t m0[K], dd[K];
					bool inl[K];
					bool more = true;	/* windows below may still be in the tile */

					/*
					 * Read phase, top window first.  A doc occurs once per
					 * term, so the K windows touch distinct accumulators and
					 * their LDS reads can all be in flight together.  Lanes
					 * without an in-tile posting work on a private dummy slot
					 * (index TILE_W + lane) with impact 0: no exec-mask
					 * juggling, the scalar unit is the scarce resource here.
					 */
#pragma unroll
					for (int k = K - 1; k >= 0; k--) {
						inm[k] = 0;
						if (more && vm[t][k]) {
							const bool ge = Ad[t][k] >= base;
							inm[k] = vm[t][k] & ballot64(ge);
							vm[t][k] &= ~inm[k];
							more = vm[t][k] == 0;	/* else: the tile ends in this window */
							if (inm[k]) {
								inl[k] = lane_of(inm[k]);
								dd[k] = inl[k] ? Ad[t][k] - base : TILE_W + lane;
								a0[k] = s_acc[dd[k]];
								m0[k] = HASMASK ? s_mask8[dd[k]] : 0;
							}
						}
					}
					/* write phase */
#pragma unroll
					for (int k = K - 1; k >= 0; k--) {
						if (inm[k]) {
							float v;
							uint64_t fb;	/* lanes touching their doc first in this tile */
							if (ANDM) {
								/* alive: carries the previous token's parity */
								const bool alive = inl[k] && (t == 0 ||
								    (a0[k] != 0.0f && (a0[k] < 0.0f) == (((t - 1) & 1) != 0)));
								v = fabsf(a0[k]) + Ai[t][k];
								s_acc[dd[k]] = alive ? ((t & 1) ? -v : v) : a0[k];
								if (alive && t == (int)nt - 1) {
									tmax = max(tmax, __float_as_uint(v));
								}
								fb = t == 0 ? inm[k] : 0;
							} else {
								v = a0[k] + (inl[k] ? Ai[t][k] : 0.0f);
								s_acc[dd[k]] = v;
								if (HASMASK) {
									const uint32_t bits = m0[k] | (inl[k] ? (1u << t) : 0u);
									s_mask8[dd[k]] = (uint8_t)bits;
									/* only docs that already hold every required
									 * term can become candidates */
									if ((bits & req) == req) {
										tmax = max(tmax, __float_as_uint(v));
									}
								} else {
									tmax = max(tmax, __float_as_uint(v));
								}
								/* (masks of direct compares: no bool round trip) */
								fb = inm[k] & (HASMASK ? ballot64(m0[k] == 0) : ballot64(a0[k] == 0.0f));
							}
							if (n_list <= LIST_CAP) {
								const uint32_t nf = __popcll(fb);
								if (n_list + nf <= LIST_CAP && lane_of(fb)) {
									s_list[n_list + lanes_below(fb)] = (uint16_t)dd[k];
								}
								n_list += nf;
							}
						}
					}
					if (more && ab[t] > lo[t]) {
						/* the whole set was in the tile and postings remain */
						rotate_sets(tc);
						continue;
					}
					break;
				}
				refresh_pdoc(tc);
			}
		});
		WAVE_SYNC();

		bool full_scan = n_list > LIST_CAP;
		/*
		 * Scores only grow while a tile is accumulated (all impacts are
		 * positive), so a doc's final score is one of the values written.
		 * If none of them beats the threshold no doc of the tile can be a
		 * candidate: just wipe the accumulators.
		 */
		if (MODE_FILTERS(MODE) && ballot64(__uint_as_float(tmax) > thr) == 0) {
			if (full_scan) {
				for (uint32_t i = lane; i < TILE_W; i += WAVE) {
					s_acc[i] = 0.0f;
				}
				if (HASMASK) {
					for (uint32_t i = lane; i < TILE_W / 4; i += WAVE) {
						s_mask32[i] = 0;
					}
				}
			} else {
				for (uint32_t off = 0; off < n_list; off += WAVE) {
					const uint32_t i = off + lane;
					if (i < n_list) {
						const uint32_t d = s_list[i];
						s_acc[d] = 0.0f;
						if (HASMASK) {
							s_mask8[d] = 0;
						}
					}
				}
			}
			WAVE_SYNC();
			continue;
		}
		if (!full_scan) {
			/* sparse tile: visit only the touched docs */
			uint32_t ncand = 0;
			for (uint32_t off = 0; off < n_list && !full_scan; off += WAVE) {
				const uint32_t i = off + lane;
				const bool valid = i < n_list;
				uint32_t d = 0, m = 0;
				float sc = 0.0f;
				if (valid) {
					d = s_list[i];
					sc = s_acc[d];
					if (HASMASK) {
						m = s_mask8[d];
					} else if (ANDM) {
						/* matched iff the last token's parity is on it */
						m = sc != 0.0f && (sc < 0.0f) == (((nt - 1) & 1) != 0);
						sc = fabsf(sc);
					} else {
						m = 1;		/* listed => touched */
					}
				}
				if (MODE == MODE_COUNT) {
					const bool match = valid && (HASMASK ? ((s_truth[m >> 5] >> (m & 31)) & 1) : (m != 0));
					n_out += __popcll(ballot64(match));
					continue;
				}
				const bool pre = valid && (sc > thr);
				if (ballot64(pre) == 0) {
					continue;
				}
				const bool cand = pre && (HASMASK ? ((s_truth[m >> 5] >> (m & 31)) & 1) : (m != 0));
				const uint64_t bal = ballot64(cand);
				if (!bal) {
					continue;
				}
				const uint32_t ne = __popcll(bal);
				if (ncand + ne > TCAND_CAP) {
					full_scan = true;	/* nothing emitted or cleared yet */
					break;
				}
				if (cand) {
					const uint32_t slot = ncand + lanes_below(bal);
					s_cd[slot] = d;
					s_cs[slot] = sc;
				}
				ncand += ne;
			}
			if (!full_scan) {
				WAVE_SYNC();
				if (MODE != MODE_COUNT && ncand) {
					/* rank by doc (descending) so the segment stays ordered */
					uint32_t cd = 0, rank = 0;
					float cs = 0.0f;
					if (lane < ncand) {
						cd = s_cd[lane];
						cs = s_cs[lane];
					}
					for (uint32_t j = 0; j < ncand; j++) {
						const uint32_t dj = __builtin_amdgcn_readlane((int)cd, j);
						rank += dj > cd;
					}
					if (MODE_FILTERS(MODE) && n_out + ncand > A.seg_cap) {
						ovf = true;
					} else if (lane < ncand) {
						const uint64_t o = out_base + n_out + rank;
						A.cand_doc[o] = base + cd;
						A.cand_sc[o] = cs;
					}
					n_out += ncand;
					if constexpr (MODE == MODE_BIG) {
						bigk_account(s_hist, A.k, big_upd, lane < ncand, cs, ncand, big_since, hint, thr);
					}
					if (track) {
						for (uint32_t j = 0; j < ncand; j++) {
							const float v = __shfl(cs, (int)j);
							if (v > thr) {
								const uint32_t pos = __popcll(ballot64(top >= v));
								const float up = __shfl_up(top, 1);
								top = (lane < pos) ? top : (lane == pos ? v : up);
								thr = fmaxf(__shfl(top, kidx), hint);
							}
						}
					}
				}
				/* clear what this tile touched */
				for (uint32_t off = 0; off < n_list; off += WAVE) {
					const uint32_t i = off + lane;
					if (i < n_list) {
						const uint32_t d = s_list[i];
						s_acc[d] = 0.0f;
						if (HASMASK) {
							s_mask8[d] = 0;
						}
					}
				}
			}
		}
		if (full_scan) {
			/* dense tile: ordered scan, DESCENDING doc (results.c:143-147) */
			for (int sidx = TILE_W / WAVE - 1; sidx >= 0; sidx--) {
				const uint32_t d = sidx * WAVE + lane;
				float sc = 0.0f;
				uint32_t m;
				if (HASMASK) {
					m = s_mask8[d];
					if (ballot64(m != 0) == 0) {
						continue;
					}
					if (m) {
						sc = s_acc[d];
						s_acc[d] = 0.0f;
						s_mask8[d] = 0;
					}
				} else {
					sc = s_acc[d];
					m = sc != 0.0f;
					if (ballot64(m != 0) == 0) {
						continue;
					}
					if (m) {
						s_acc[d] = 0.0f;
					}
					if (ANDM) {
						m = m && (sc < 0.0f) == (((nt - 1) & 1) != 0);
						sc = fabsf(sc);
					}
				}
				if (MODE == MODE_COUNT) {
					const bool match = m && (HASMASK ? ((s_truth[m >> 5] >> (m & 31)) & 1) : (m != 0));
					n_out += __popcll(ballot64(match));
					continue;
				}
				const bool pre = m && (sc > thr);
				if (ballot64(pre) == 0) {
					continue;
				}
				const bool cand = pre && (HASMASK ? ((s_truth[m >> 5] >> (m & 31)) & 1) : (m != 0));
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
						A.cand_doc[o] = base + d;
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
		WAVE_SYNC();
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

/* ---- launchers ------------------------------------------------------ */

template <int MODE>
static void
launch_generic_mode(bool wide_mask, const dim3 grid, hipStream_t st, const scan_args_t &a)
{
	const dim3 block(WAVE);

	if (wide_mask) {
		hipLaunchKernelGGL((k_scan<NXSGPU_MAX_TOKENS, uint32_t, MODE>), grid, block, 0, st, a);
	} else {
		hipLaunchKernelGGL((k_scan<8, uint8_t, MODE>), grid, block, 0, st, a);
	}
}

/* k_scan: wide_mask = up to 32 tokens (u32 presence mask + postfix program),
 * else the <= 8 token form of NXS_GPU_OLDSCAN / >= 2^31 docs */
void
nxs_launch_scan_generic(int mode, bool wide_mask, unsigned grid, hipStream_t st, const scan_args_t &a)
{
	switch (mode) {
	case MODE_TOPK: launch_generic_mode<MODE_TOPK>(wide_mask, dim3(grid), st, a); break;
	case MODE_BIG: launch_generic_mode<MODE_BIG>(wide_mask, dim3(grid), st, a); break;
	case MODE_COUNT: launch_generic_mode<MODE_COUNT>(wide_mask, dim3(grid), st, a); break;
	default: launch_generic_mode<MODE_ALL>(wide_mask, dim3(grid), st, a); break;
	}
}

template <int MODE>
static void
launch_scan8_mode(uint32_t nt_bucket, uint32_t mm, const dim3 grid, hipStream_t st, const scan_args_t &a)
{
	const dim3 block(WAVE);

	switch (nt_bucket) {
	case 1:
		hipLaunchKernelGGL((k_scan8<MODE, 1, 0>), grid, block, 0, st, a);
		break;
	case 2:
		if (mm == 1) { hipLaunchKernelGGL((k_scan8<MODE, 2, 1>), grid, block, 0, st, a); }
		else { hipLaunchKernelGGL((k_scan8<MODE, 2, 0>), grid, block, 0, st, a); }
		break;
	case 3:
		if (mm == 1) { hipLaunchKernelGGL((k_scan8<MODE, 3, 1>), grid, block, 0, st, a); }
		else { hipLaunchKernelGGL((k_scan8<MODE, 3, 0>), grid, block, 0, st, a); }
		break;
	case 5:
		if (mm == 1) { hipLaunchKernelGGL((k_scan8<MODE, 5, 1>), grid, block, 0, st, a); }
		else { hipLaunchKernelGGL((k_scan8<MODE, 5, 0>), grid, block, 0, st, a); }
		break;
	default:
		if (mm == 1) { hipLaunchKernelGGL((k_scan8<MODE, 8, 1>), grid, block, 0, st, a); }
		else { hipLaunchKernelGGL((k_scan8<MODE, 8, 0>), grid, block, 0, st, a); }
		break;
	}
}

/* k_scan8: mm = 0 mask byte + truth table, 1 pure OR */
void
nxs_launch_scan8(int mode, uint32_t nt_bucket, uint32_t mm, unsigned grid, hipStream_t st, const scan_args_t &a)
{
	switch (mode) {
	case MODE_TOPK: launch_scan8_mode<MODE_TOPK>(nt_bucket, mm, dim3(grid), st, a); break;
	case MODE_BIG: launch_scan8_mode<MODE_BIG>(nt_bucket, mm, dim3(grid), st, a); break;
	case MODE_COUNT: launch_scan8_mode<MODE_COUNT>(nt_bucket, mm, dim3(grid), st, a); break;
	default: launch_scan8_mode<MODE_ALL>(nt_bucket, mm, dim3(grid), st, a); break;
	}
}


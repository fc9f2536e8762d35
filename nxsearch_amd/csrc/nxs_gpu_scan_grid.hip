/*
 * nxs_gpu_scan_grid.hip -- k_scang: the mask path (quantised byte bounds, see
 * nxs_gpu_scan_mask.hip) walked over a DOC GRID instead of per-term register windows
 * (MI355X / gfx950 query path of nxsearch; see nxs_gpu_int.h for the map of the files)
 *
 * Why.  k_scanm keeps two 64-posting windows per term in registers and visits
 * every term once per tile: a window of a rank 100..1000 term spans 3-29 k docs,
 * the byte map 8192, so a wavefront makes (tiles x terms) visits at 27 of 64
 * lanes, each with its scalar bookkeeping (which lanes, highest unconsumed doc,
 * window drained?, ring rotation): 160 issued instructions per 64 postings, more
 * scalar than vector (profiles/r3_pmc_summary.json).  Here the unit is a PART of
 * the doc space -- [b * bs, (b + 1) * bs), bs a power of two chosen per wavefront
 * so that a part holds a few hundred postings over all terms -- and the only
 * per-term state is where the part's postings begin in the term's list.  All
 * terms' postings of a part are one run of slots (term 0's, then term 1's ...):
 * slot s of group g sits in lane s - 64 g, so every lane of every group but the
 * last works, whatever the terms' densities, and a group costs the same ~30
 * vector instructions whether its postings come from one term or five.  No
 * windows, no masks of unconsumed lanes, no rotation.
 *
 * Where a part begins in a term's list is a lower bound by doc (what k_cursors
 * does for the range boundaries): lane t * CT + j searches boundary j of term t
 * for the next CT parts, one probe per part iteration, while the current CT
 * parts are processed -- the dependent loads of the search are never waited
 * for.  Only the first CT boundaries of a wavefront are searched blocking.
 *
 * Everything else is k_scanm's: one byte per doc in LDS, fire-and-forget
 * ds_add_rtn_u32, the old byte plus the posting's own quantised impact against
 * the quantised threshold, pending list, exact scores in token order for the
 * docs that pass (taken from the part's groups, which stay in registers),
 * descending doc order of what is emitted, threshold hand-down between the
 * ranges, the dense terms leaving the scan (DROP), the retry list.
 */
#include "nxs_gpu_int.h"
#include "nxs_gpu_dev.h"

#ifdef NXS_EXPERIMENTAL	/* opt-in build (round 3: built, bit-exact on the whole tier, level to 2 % behind k_scanm on C3: NOTES.md) */

#ifndef GT_W
#define	GT_W		8192		/* docs per byte map (and widest part); a power of two */
#endif
#define	MT_W		GT_W
#define	MT_W0		64		/* cold-start sub-tile width */
#define	MT_W_HINTED	(MT_W < 2048 ? MT_W : 2048)
#define	PEND_CAP	128
#ifndef DROP_PEND_MULT
#define	DROP_PEND_MULT	1
#endif
#define	QSUM_MAX	224
#ifndef GT_G
#define	GT_G		6		/* groups of 64 postings a part may hold (registers: 8 = 108 VGPRs and four
					 * wavefronts per SIMD, 6 = 90 and five; C3 1.42 -> 1.31 ms) */
#endif
#define	GT_CAP		(GT_G * WAVE)

#ifdef NXS_STATS
/* diagnostic build only: event counts (tools/scanm_stats.py) */
__device__ unsigned long long g_stats_grid[16];
#define	STAT_ADD(i, v)	do { if (lane == 0) atomicAdd(&g_stats_grid[i], (unsigned long long)(v)); } while (0)
extern "C" void
nxsgpu_debug_stats_grid(unsigned long long *out, int reset)
{
	unsigned long long z[16] = { 0 };
	(void)hipDeviceSynchronize();
	(void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stats_grid), sizeof(z));
	if (reset) {
		(void)hipMemcpyToSymbol(HIP_SYMBOL(g_stats_grid), z, sizeof(z));
	}
}
#define	STAT_CLK()	((unsigned long long)__builtin_amdgcn_s_memtime())
#else
#define	STAT_ADD(i, v)	do { } while (0)
#define	STAT_CLK()	0ull
#endif
#ifndef GT_FILL
#define	GT_FILL		260		/* postings per part the boundary stride aims at */
#endif

template <int NT, bool GEN, bool DROP = false>
__global__ void __launch_bounds__(WAVE)
k_scang(const scan_args_t A)
{
	constexpr int CT = WAVE / NT;		/* boundaries per term a chunk holds */
	__shared__ __attribute__((aligned(16))) uint32_t s_mask[MT_W / 4 + WAVE];	/* + one dummy word per lane */
	constexpr uint32_t PCAP = DROP ? DROP_PEND_MULT * PEND_CAP : PEND_CAP;
	__shared__ uint32_t s_pend[PCAP];
	__shared__ uint32_t s_psum[DROP ? PCAP : 1];
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

	for (uint32_t i = lane; i < MT_W / 4 + WAVE; i += WAVE) {
		s_mask[i] = 0;
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

	const uint32_t dmask = DROP ? rfl32(Q->drop_mask) : 0u;
	const uint32_t *cs = A.cold_state + seg * 16;
	const uint32_t cs_left = DROP ? rfl32(cs[0]) : 0u;	/* docs below this are left (0: none) */
	const uint32_t cs_nout = DROP ? rfl32(cs[1]) : 0u;
	const float cs_thr = DROP ? __uint_as_float(rfl32(cs[2])) : 0.0f;
	const bool cs_ovf = DROP && rfl32(cs[3]) != 0;

	/* the range's docs [d_bot, d_top) and, per term, its postings [lo, hi) in them */
	const uint32_t d_bot = (uint32_t)min((uint64_t)g * qm.group_docs, A.n_docs);
	uint32_t d_top = (g + 1 == qm.n_groups) ? (uint32_t)A.n_docs :
	    (uint32_t)min((uint64_t)(g + 1) * qm.group_docs, A.n_docs);
	if (DROP) {
		d_top = min(d_top, cs_left);		/* (0: nothing left, no part) */
	}
	uint64_t pb[NT];
	int32_t lo[NT], hi[NT];
	float tmx[NT];
	static_for<NT>([&](auto tc) {
		constexpr int t = decltype(tc)::value;
		pb[t] = 0;
		lo[t] = hi[t] = 0;
		tmx[t] = 0.0f;
		if (t < (int)nt) {
			const uint64_t cb = ((uint64_t)qm.seg_first + q + g) * NXSGPU_MAX_TOKENS + t;
			pb[t] = rfl64(Q->pbeg[t]);
			lo[t] = (int32_t)rfl32(A.cursors[cb]);
			hi[t] = (int32_t)rfl32(A.cursors[cb + NXSGPU_MAX_TOKENS]);
			tmx[t] = Q->tmax[t];
			if (DROP) {
				if (((dmask >> t) & 1) || cs_left == 0) {
					hi[t] = lo[t];
				} else {
					hi[t] = max(lo[t], min(hi[t], (int32_t)rfl32(cs[4 + t])));
				}
			}
		}
	});
	const float hint0 = range_hint(A, qm, g);	/* 0 = nothing published yet */

	/*
	 * Boundary stride: the largest power of two (<= the byte map) at which a part
	 * is expected to hold GT_FILL postings.  A part that holds more than the
	 * registers take (GT_CAP) is halved by doc until it fits (`split` below).
	 */
	uint32_t n_all = 0, maxlen = 0;
#pragma unroll
	for (int t = 0; t < NT; t++) {
		n_all += (uint32_t)(hi[t] - lo[t]);
		maxlen = max(maxlen, (uint32_t)(hi[t] - lo[t]));
	}
	static_assert((MT_W & (MT_W - 1)) == 0 && MT_W >= 1024, "the byte map is a power of two");
	uint32_t bshift = 31 - __builtin_clz((uint32_t)MT_W);
	{
		const uint64_t span = d_top > d_bot ? d_top - d_bot : 1;
		while (bshift > 6 && ((uint64_t)n_all << bshift) > (uint64_t)GT_FILL * span) {
			bshift--;
		}
	}
	/* boundary b is doc b << bshift; part = [max(b << bshift, d_bot), next part's base) from
	 * b_first (the highest boundary below d_top) down to b_last (at or below d_bot) */
	const int32_t b_first = d_top > d_bot ? (int32_t)((d_top - 1) >> bshift) : -1;
	const int32_t b_last = (int32_t)(d_bot >> bshift);
	int32_t parts_left = d_top > d_bot && n_all ? b_first - b_last + 1 : 0;
	const uint32_t nsteps = 32 - __builtin_clz(maxlen | 1);		/* probes a lower bound over maxlen postings takes */

	/* lane t * CT + j works for term t */
	const uint32_t lt = lane / CT, lj = lane - lt * CT;
	uint64_t l_pb = 0;
	int32_t l_lo = 0, l_hi = 0;
	static_for<NT>([&](auto tc) {
		constexpr int t = decltype(tc)::value;
		if (lt == t) {
			l_pb = pb[t];
			l_lo = lo[t];
			l_hi = hi[t];
		}
	});
	const posting_t *const l_pt = A.post + l_pb;

	/*
	 * The boundary search: lane (t, j) looks for the first posting of term t at or
	 * above doc (bc - j) << bshift.  srch_begin() sets the interval, srch_issue()
	 * sends the probe of the interval's middle, srch_take() narrows it.
	 */
	int32_t sl = 0, sh = 0, smid = 0;
	uint32_t sdoc = 0, sprobe = 0;
	bool sact = false;
	auto srch_begin = [&](int32_t bc) {
		const int32_t bl = bc - (int32_t)lj;
		sdoc = bl > 0 ? (uint32_t)bl << bshift : 0u;
		sl = l_lo;
		sh = l_hi;
		sact = false;
	};
	auto srch_issue = [&]() {
		sact = sl < sh;
		smid = sl + ((sh - sl) >> 1);
		sprobe = 0;
		if (sact) {
			sprobe = l_pt[smid].doc;
		}
	};
	auto srch_take = [&]() {
		if (sact) {
			if (sprobe < sdoc) {
				sl = smid + 1;
			} else {
				sh = smid;
			}
		}
		sact = false;
	};

	float top = DROP ? A.cold_top[seg * 64 + lane] : -INFINITY;
	float hint = hint0;
	float thr = DROP ? fmaxf(hint, cs_thr) : hint;
	const uint32_t kidx = A.k - 1;			/* 1 <= k <= 64 (host) */
	uint32_t n_out = cs_nout;
	bool ovf = cs_ovf;
	const uint64_t out_base = seg * A.seg_cap;

	/* Quantisation: see k_scanm */
	float tsum = 0.0f;
#pragma unroll
	for (int t = 0; t < NT; t++) {
		tsum += tmx[t];
	}
	const float qs = tsum > 0.0f ? (float)QSUM_MAX / tsum : 0.0f;
	auto thr_quant = [&](float th) -> int32_t {
		return __builtin_amdgcn_readfirstlane(th > 0.0f ? (int32_t)min(th * qs, 1.0e6f) - 1 : -1);
	};
	int32_t thr_q = thr_quant(thr);

	const uint32_t dropped = dmask;
	uint32_t qU = 0, q1max = 0;
	float U = 0.0f;
	if constexpr (DROP) {
#pragma unroll
		for (int t = 0; t < NT; t++) {
			if ((dmask >> t) & 1) {
				U += tmx[t];
				qU += (uint32_t)(tmx[t] * qs) + 2;
			} else {
				q1max = max(q1max, (uint32_t)(tmx[t] * qs) + 2);
			}
		}
		q1max = rfl32(q1max);
		qU = rfl32(qU);
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

	/* the part being processed: its groups (slot g * 64 + lane; invalid slots hold doc
	 * 0xffffffff), and the part whose loads are in flight */
	uint32_t cd[GT_G], nd[GT_G];
	float ci[GT_G], ni[GT_G];
#pragma unroll
	for (int gg = 0; gg < GT_G; gg++) {
		cd[gg] = nd[gg] = 0xffffffffu;
		ci[gg] = ni[gg] = 0.0f;
	}
	uint32_t p_base = 0, p_top = 0, p_total = 0, p_pre[NT];
	uint32_t f_base = 0, f_top = 0, f_total = 0, f_pre[NT];
#pragma unroll
	for (int t = 0; t < NT; t++) {
		p_pre[t] = f_pre[t] = 0;
	}

	/*
	 * Flush: k_scanm's, except that a doc's impacts are looked up in the part's
	 * groups.  Slots ascend with the term, so walking the groups and the matching
	 * lanes upwards adds the impacts in token order (results.c:134-136).
	 */
	auto flush = [&]() {
		constexpr int PC = PCAP / WAVE;
		n_pend = rfl32(n_pend);
		n_out = rfl32(n_out);
		const uint32_t nch = (n_pend + WAVE - 1) / WAVE;
		const uint32_t ng = (p_total + WAVE - 1) / WAVE;
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
			uint32_t dcol[NT];
			if constexpr (DROP) {
				static_for<NT>([&](auto tc) {
					constexpr int t = decltype(tc)::value;
					dcol[t] = 0xffffffffu;
					if ((dropped >> t) & 1) {
						const uint64_t cbase = (uint64_t)rfl32(Q->drop_col[t]) * A.dense_stride;
						dcol[t] = A.dense_col[cbase + (live ? d : 0u)];
					}
				});
				uint32_t qd = 0;
				static_for<NT>([&](auto tc) {
					constexpr int t = decltype(tc)::value;
					if (((dropped >> t) & 1) && dcol[t] != 0xffffffffu) {
						qd += (uint32_t)(__uint_as_float(dcol[t]) * qs) + 2;
					}
				});
				uint32_t sumq = live ? s_psum[e] : 0u;
#pragma unroll
				for (int kk = 1; kk < NT; kk++) {
					if (live && e + kk < n_pend && s_pend[e + kk] == d) {
						sumq = max(sumq, s_psum[e + kk]);
					}
				}
				todo = ballot64(live && (int32_t)(sumq + qd) > thr_q + (int32_t)qU);
			}
			(void)dcol;
			while (todo) {
				const int j = __builtin_ctzll(todo);
				todo &= todo - 1;
				const uint32_t dj = (uint32_t)__builtin_amdgcn_readlane((int)d, j);
				float acc = 0.0f;
				uint32_t pm = 0;	/* the tokens the doc holds */
				uint32_t cv = 0;	/* DROP: lane t = the doc's impact in token t (bits; +0.0f = absent) */
				static_for<GT_G>([&](auto gc) {
					constexpr int gg = decltype(gc)::value;
					if ((uint32_t)gg < ng) {
						uint64_t m = ballot64(cd[gg] == dj);
						while (m) {
							const int l = __builtin_ctzll(m);
							m &= m - 1;
							const int ib = __builtin_amdgcn_readlane(__builtin_bit_cast(int, ci[gg]), l);
							if constexpr (GEN || DROP) {
								/* whose slot: terms with no posting in the part share their
								 * first slot with the next term, which then owns it */
								const uint32_t slot = (uint32_t)gg * WAVE + (uint32_t)l;
								uint32_t tm = 0;
#pragma unroll
								for (int u = 1; u < NT; u++) {
									tm += slot >= p_pre[u] ? 1u : 0u;
								}
								pm |= 1u << tm;
								if constexpr (DROP) {
									cv = lane == tm ? (uint32_t)ib : cv;
								}
							}
							if constexpr (!DROP) {
								acc += __builtin_bit_cast(float, ib);
							}
						}
					}
				});
				if constexpr (DROP) {
					static_for<NT>([&](auto tc) {
						constexpr int t = decltype(tc)::value;
						if ((dropped >> t) & 1) {
							const uint32_t xb = (uint32_t)__builtin_amdgcn_readlane((int)dcol[t], j);
							if (xb != 0xffffffffu) {
								cv = lane == (unsigned)t ? xb : cv;
								pm |= 1u << t;
							}
						}
					});
					/* token order; x + 0.0f == x for the absent ones (impacts are > 0) */
					static_for<NT>([&](auto tc) {
						constexpr int t = decltype(tc)::value;
						acc += __builtin_bit_cast(float, __builtin_amdgcn_readlane((int)cv, t));
					});
				}
				if (GEN && !((s_truth[pm >> 5] >> (pm & 31)) & 1)) {
					acc = -INFINITY;
				}
				(void)pm;
				sc = (lane == (unsigned)j) ? acc : sc;
			}
			const bool cand = live && sc > thr;
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
				const uint64_t o = out_base + n_out + lanes_below(bal);
				A.cand_doc[o] = d;
				A.cand_sc[o] = sc;
			}
			n_out += ne;
			while (bal) {
				const int L = __builtin_ctzll(bal);
				const float v = __shfl(sc, L);
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

	/* widest sub-tile tried next: small while nothing is known about the threshold */
	uint32_t tw = thr_q >= 0 ? (uint32_t)MT_W_HINTED : (uint32_t)MT_W0;
	if constexpr (DROP) {
		if (dmask) {
			thr_q = thr_quant(thr) - (int32_t)qU;
			tw = thr_q >= (int32_t)q1max ? (uint32_t)MT_W_HINTED : (uint32_t)MT_W0;
		}
	}

	/*
	 * The walk.  cur_top / cur_ohi: where the part fetched next ends (doc,
	 * exclusive; list positions).  coff: lane (t, j) = where boundary bc - j cuts
	 * term t's list, for the chunk of CT boundaries the walk is in; cj = the next
	 * boundary of the chunk.
	 */
	uint32_t cur_top = d_top;
	int32_t cur_ohi[NT], f_olo[NT];
#pragma unroll
	for (int t = 0; t < NT; t++) {
		cur_ohi[t] = hi[t];
		f_olo[t] = hi[t];
	}
	int32_t bc = b_first, coff = 0;
	uint32_t cj = 0, steps_left = 0;
	if (parts_left > 0) {
		srch_begin(bc);
		for (uint32_t s = 0; s < nsteps; s++) {
			srch_issue();
			srch_take();
		}
		coff = sl;
		if (parts_left > CT) {
			srch_begin(bc - CT);
			steps_left = nsteps;
		}
	}
	bool inflight = false;		/* a probe of the next chunk's search is on its way */

	/*
	 * fetch: describe the next part (f_*) and send the loads of its groups into
	 * nd / ni.  Returns with f_total == 0 when the walk is over.
	 */
	auto fetch = [&]() {
		f_total = 0;
		/* (the probe sent at the end of the last call: taken BEFORE this call's loads go out --
		 * behind them the wait could not tell it from them and would wait for all) */
		if (inflight) {
			srch_take();
			inflight = false;
		}
		while (parts_left > 0 && f_total == 0) {
			parts_left = (int32_t)rfl32((uint32_t)parts_left);
			cj = rfl32(cj);
			bc = (int32_t)rfl32((uint32_t)bc);
			cur_top = rfl32(cur_top);
			if (cj == (uint32_t)CT) {
				/* next chunk: what is left of its search, blocking (none as a rule: one
				 * probe went out per part) */
				while (steps_left) {
					srch_issue();
					srch_take();
					steps_left--;
				}
				coff = sl;
				bc -= CT;
				cj = 0;
				if (parts_left > CT) {
					srch_begin(bc - CT);
					steps_left = nsteps;
				}
				/* a higher range may have published meanwhile */
				hint = fmaxf(hint, range_hint(A, qm, g));
				if (hint > thr) {
					thr = hint;
					thr_q = thr_quant(thr) - (int32_t)(DROP && dropped ? qU : 0u);
				}
			}
			const uint32_t gbase = max((uint32_t)(bc - (int32_t)cj) << bshift, d_bot);
			uint32_t total = 0;
			int32_t olo[NT];
			static_for<NT>([&](auto tc) {
				constexpr int t = decltype(tc)::value;
				/* (never above the position the walk is at: the lists ascend) */
				olo[t] = min(__builtin_amdgcn_readlane(coff, t * CT + (int)cj), cur_ohi[t]);
				total += (uint32_t)(cur_ohi[t] - olo[t]);
			});
			uint32_t base = gbase;
			/* too many postings for the registers: the upper half of the docs, until it fits
			 * (one doc holds at most NT postings, so this ends) */
			while (total > (uint32_t)GT_CAP) {
				STAT_ADD(11, 1);
				base = rfl32(base);
				const uint32_t mid = base + ((cur_top - base) >> 1);
				int32_t xl = 0, xh = 0;
				static_for<NT>([&](auto tc) {
					constexpr int t = decltype(tc)::value;
					if (lt == t) {
						xl = olo[t];
						xh = cur_ohi[t];
					}
				});
				while (ballot64(xl < xh)) {
					const int32_t xm = xl + ((xh - xl) >> 1);
					if (xl < xh) {
						if (l_pt[xm].doc < mid) {
							xl = xm + 1;
						} else {
							xh = xm;
						}
					}
				}
				total = 0;
				static_for<NT>([&](auto tc) {
					constexpr int t = decltype(tc)::value;
					olo[t] = __builtin_amdgcn_readlane(xl, t * CT);
					total += (uint32_t)(cur_ohi[t] - olo[t]);
				});
				base = mid;
			}
			/* the part: docs [base, cur_top), postings [olo, cur_ohi) of every term */
			f_base = base;
			f_top = cur_top;
			f_total = total;
			uint32_t pre = 0;
			/* posting index of slot s = kk[its term] + s, all of it in 32 bits (the host sends
			 * indexes with more than 2^32 postings to k_scanm); wrapping is fine */
			uint32_t kk[NT];
			static_for<NT>([&](auto tc) {
				constexpr int t = decltype(tc)::value;
				f_pre[t] = pre;
				kk[t] = (uint32_t)pb[t] + (uint32_t)olo[t] - pre;
				pre += (uint32_t)(cur_ohi[t] - olo[t]);
			});
			if (total) {
				static_for<GT_G>([&](auto gc) {
					constexpr int gg = decltype(gc)::value;
					if ((uint32_t)gg * WAVE < total) {
						const uint32_t s = min((uint32_t)gg * WAVE + lane, total - 1);
						/* (readfirstlane at the use: a select chain over the elements of a local
						 * array is otherwise turned into an indexed load from scratch memory) */
						uint32_t k = rfl32(kk[0]);
						static_for<NT - 1>([&](auto tc) {
							constexpr int t = decltype(tc)::value + 1;
							const uint32_t kt = rfl32(kk[t]);	/* (outside the select: convergent) */
							k = s >= f_pre[t] ? kt : k;
						});
						const posting_t p = A.post[k + s];
						nd[gg] = p.doc;
						ni[gg] = p.imp;
					}
				});
			}
#pragma unroll
			for (int t = 0; t < NT; t++) {
				cur_ohi[t] = olo[t];
			}
			cur_top = base;
			if (base == gbase) {
				cj++;
				parts_left--;
			}
		}
		/* one probe of the next chunk's search per call */
		if (steps_left) {
			srch_issue();
			steps_left--;
			inflight = true;
		}
	};

	fetch();
	uint32_t ovf_u = 0;
	for (;;) {
		n_pend = rfl32(n_pend);
		n_out = rfl32(n_out);
		tw = rfl32(tw);
		thr_q = (int32_t)rfl32((uint32_t)thr_q);
		ovf_u = rfl32(ovf_u | (ovf ? 1u : 0u));
		ovf = ovf_u != 0;
		f_total = rfl32(f_total);
		if (f_total == 0 || ovf) {
			break;
		}
		/* the fetched part becomes the current one (its loads land here) */
		p_base = rfl32(f_base);
		p_top = rfl32(f_top);
		p_total = f_total;
#pragma unroll
		for (int t = 0; t < NT; t++) {
			p_pre[t] = rfl32(f_pre[t]);
		}
#pragma unroll
		for (int gg = 0; gg < GT_G; gg++) {
			const bool v = (uint32_t)gg * WAVE + lane < p_total;
			cd[gg] = v ? nd[gg] : 0xffffffffu;
			ci[gg] = ni[gg];
		}
		const uint32_t ng = (p_total + WAVE - 1) / WAVE;
		STAT_ADD(1, 1);
		STAT_ADD(8, p_top - p_base);
		STAT_ADD(9, p_total);
		STAT_ADD(12, ng);
		fetch();		/* the part after it: in flight while this one is worked on */

		/* sub-tiles of the part, from the top: docs [sb, se] */
		int32_t se = (int32_t)p_top - 1;
		while (se >= (int32_t)p_base && !ovf) {
			se = (int32_t)rfl32((uint32_t)se);
			n_pend = rfl32(n_pend);
			tw = rfl32(tw);
			thr_q = (int32_t)rfl32((uint32_t)thr_q);
			const uint32_t sb = (uint32_t)max((int32_t)p_base, se - (int32_t)tw + 1);
			const uint32_t width = (uint32_t)se - sb;
			const uint32_t n_before = n_pend;
			STAT_ADD(2, 1);
			uint32_t oldv[GT_G], qv[GT_G];
			uint64_t vis[GT_G];
			static_for<GT_G>([&](auto gc) {
				constexpr int gg = decltype(gc)::value;
				vis[gg] = 0;
				oldv[gg] = 0;
				qv[gg] = 0;
				if ((uint32_t)gg < ng) {
					const uint32_t dd = cd[gg] - sb;
					const bool in = dd <= width;
					const uint64_t inm = ballot64(in);
					if (inm) {
						const uint32_t sh = (dd & 3) * 8;
						const uint32_t w = in ? (dd >> 2) : MT_W / 4 + lane;
						const uint32_t qq = (uint32_t)(ci[gg] * qs) + 2;
						oldv[gg] = __hip_atomic_fetch_add(&s_mask[w], in ? (qq << sh) : 0u,
						    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
						qv[gg] = (qq << 8) | sh;
						vis[gg] = inm;
					}
				}
			});
			static_for<GT_G>([&](auto gc) {
				constexpr int gg = decltype(gc)::value;
				if (vis[gg]) {
					const uint32_t sum = ((oldv[gg] >> (qv[gg] & 31)) & 0xffu) + (qv[gg] >> 8);
					const uint64_t cm = vis[gg] & ballot64((int32_t)sum > thr_q);
					if (cm) {
						push(cm, cd[gg], sum);
					}
				}
			});
			WAVE_SYNC();
			/* wipe the sub-tile's bytes (16 B per lane and store) */
			{
				const uint32_t words = (width + 4) >> 2;
				for (uint32_t i0 = 0; i0 < words; i0 += WAVE * 4) {
					*(uint4 *)&s_mask[i0 + lane * 4] = make_uint4(0, 0, 0, 0);
				}
			}
			const uint32_t n_tile = n_pend - n_before;
			/*
			 * More docs above the threshold than the pending list takes (a weak threshold
			 * and a wide sub-tile): nothing is lost -- the part is still in registers and
			 * the bytes are wiped -- so the same docs are walked again in narrower
			 * sub-tiles.  (The list is empty at every sub-tile's start: n_before == 0.)
			 */
			const bool over = n_pend > PCAP;
			const bool redo = over && tw > (uint32_t)MT_W0;
			if (over && !redo) {
				ovf = true;
			} else if (!over && n_pend) {
				flush();
			}
			if (redo) {
				STAT_ADD(6, 1);
				n_pend = 0;
				tw = max(min(tw, width + 1) >> 2, (uint32_t)MT_W0);
			} else {
				if (DROP && dropped) {
					if (n_tile <= 36) {
						tw = min(tw * 2, (uint32_t)MT_W);
					} else if (n_tile > 88) {
						tw = max(tw / 2, (uint32_t)MT_W0);
					}
				} else
				if (n_tile <= 8) {
					tw = min(tw * 2, (uint32_t)MT_W);
				} else if (n_tile > 48) {
					tw = max(tw / 2, (uint32_t)MT_W0);
				}
				se = (int32_t)sb - 1;
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

void
nxs_launch_scang(uint32_t nt_bucket, bool gen, bool drop, unsigned grid_, hipStream_t st, const scan_args_t &a)
{
	const dim3 grid(grid_), block(WAVE);

	if (drop) {
		switch (nt_bucket) {
		case 2:
		case 3: hipLaunchKernelGGL((k_scang<3, false, true>), grid, block, 0, st, a); break;
		case 5: hipLaunchKernelGGL((k_scang<5, false, true>), grid, block, 0, st, a); break;
		default: hipLaunchKernelGGL((k_scang<8, false, true>), grid, block, 0, st, a); break;
		}
	} else if (!gen) {
		switch (nt_bucket) {
		case 2:
		case 3: hipLaunchKernelGGL((k_scang<3, false>), grid, block, 0, st, a); break;
		case 5: hipLaunchKernelGGL((k_scang<5, false>), grid, block, 0, st, a); break;
		default: hipLaunchKernelGGL((k_scang<8, false>), grid, block, 0, st, a); break;
		}
	} else {
		switch (nt_bucket) {
		case 2:
		case 3: hipLaunchKernelGGL((k_scang<3, true>), grid, block, 0, st, a); break;
		case 5: hipLaunchKernelGGL((k_scang<5, true>), grid, block, 0, st, a); break;
		default: hipLaunchKernelGGL((k_scang<8, true>), grid, block, 0, st, a); break;
		}
	}
}
#endif /* NXS_EXPERIMENTAL */

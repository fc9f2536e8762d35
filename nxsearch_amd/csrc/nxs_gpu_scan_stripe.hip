/*
 * nxs_gpu_scan_stripe.hip -- mask path, third form: k_scans (doc stripes cut out of the lists by the rank
 * directories; all terms' postings of a stripe as ONE flat run of lanes; candidates scored lane-parallel)
 * (MI355X / gfx950 query path of nxsearch; see nxs_gpu_int.h for the map of the files)
 */
#include "nxs_gpu_int.h"
#include "nxs_gpu_dev.h"

/*
 * k_scans: OR-like queries without a required token whose terms all have a rank directory
 * (nxsgpu_index::d_bmrank: per 4096-doc word of the doc space, the list position of the first
 * posting at or above it -- built for the block bitmaps of k_scanq, terms holding >= n_docs / 1024 docs).
 *
 * Same filter as k_scanm -- one BYTE per doc in LDS, the quantised upper bound of the doc's score so far,
 * raised by fire-and-forget `ds_add_rtn_u32`; a posting whose returned byte plus its own share exceeds
 * the quantised threshold makes its doc a candidate; candidates are scored exactly (f32, token order,
 * results.c:134-136), take the common threshold filter and are emitted in descending doc order -- but
 * what k_scanm spends its time on is gone:
 *
 *   k_scanm keeps a 64-posting register window (+ the one below, + a prefetch ring) PER TERM and pays a
 *   visit per (tile x term) with its scalar bookkeeping -- which lanes, drained?, rotate, highest /
 *   lowest doc by readlane --: 155 issued instructions per 64 postings, more scalar than vector, at 27
 *   of 64 lanes (profiles/r4_pmc_summary.json).
 *
 *   Here the unit is a STRIPE: ST_WORDS consecutive 4096-doc words of the doc space.  Where a stripe
 *   begins in each list is a table lookup (one coalesced load serves WAVE / NT stripes of all terms), so
 *   a term needs no window state at all; the stripe's postings of all terms form one flat run
 *   [0, n) -- term 0's, then term 1's ... -- taken 64 lanes at a time whatever the terms' densities:
 *   per window a four-deep compare / select chain finds each lane's list, one global load, ~15 vector
 *   instructions for the byte map.  Nothing of the stripe has to stay in registers afterwards: a
 *   candidate's postings are found again by a lower-bound search inside its 4096-doc word (two directory
 *   loads + <= 9 probes, one lane per (candidate, term), L2-resident lines) -- so candidates are
 *   collected ACROSS stripes and scored a few dozen at a time, and loads are ordinary compiler-tracked
 *   loads (no AGPR-owned prefetch windows).
 *
 * Cold start (no threshold yet: every posting would be a candidate): a stripe is walked in doc
 * sub-ranges from the top, 64 docs first, doubling while a sub-range yields few candidates (k_scanm's
 * rule); a sub-range is a pass over the stripe's windows with a doc filter.  Once sub-ranges have
 * grown to whole stripes the windows become a STREAM: ST_RING of them in flight across stripe ends
 * (the loader walks the directory ahead of the consumer; a slot = 64 postings + how many of its lanes
 * count + "last window of its stripe", which is when the byte map is wiped), and the pending docs of
 * FINISHED stripes -- everything later is lower -- are scored ST_FLUSH at a time.  A range whose
 * pending list overflows goes to the retry list (accumulator tiles) like k_scanm's.
 */
#ifndef ST_WORDS
#define	ST_WORDS	2		/* 4096-doc words per stripe: 8192 docs, 8 KB of byte map */
#endif
#define	ST_DOCS		(4096 * ST_WORDS)
#define	ST_PEND		256		/* pending candidates (docs) */
#ifndef ST_FLUSH
#define	ST_FLUSH	48		/* score the pending docs of finished stripes once this many wait */
#endif
#ifndef ST_RING
#define	ST_RING		6		/* posting windows in flight once stripes are taken whole (across stripe ends) */
#endif
#ifndef ST_UNR
#define	ST_UNR		3		/* scoring rounds whose searches run side by side */
#endif
#ifndef ST_GO
#define	ST_GO		96		/* cold start ends when a whole stripe is expected to push at most this many docs */
#endif
#ifndef ST_FILL_MIN
#define	ST_FILL_MIN	48		/* DROP: a stripe's map is filled from the dense term's column if the stripe holds at least this many
					 * sparse postings (fewer: zeroed, and the postings pass on the term's ceiling -- a fill costs
					 * 8 KB, a memory latency and ~130 instructions) */
#endif
#define	ST_W0		64		/* cold-start sub-range */
#ifndef ST_W_HINTED
#define	ST_W_HINTED	ST_DOCS		/* first sub-range when a higher range has published a threshold: whole stripes at once */
#endif

#ifdef NXS_STATS
/* diagnostic build only (make variant SFX=stats XFLAGS=-DNXS_STATS): event counts and cycle spans,
 * read back with nxsgpu_debug_stats_stripe() (tools/scans_stats.py) */
__device__ unsigned long long g_stats_s[16];
#define	SSTAT_ADD(i, v)	do { if (lane == 0) atomicAdd(&g_stats_s[i], (unsigned long long)(v)); } while (0)
#define	SSTAT_CLK()	((unsigned long long)__builtin_amdgcn_s_memtime())
extern "C" void
nxsgpu_debug_stats_stripe(unsigned long long *out, int reset)
{
	unsigned long long z[16] = { 0 };
	(void)hipDeviceSynchronize();
	(void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stats_s), sizeof(z));
	if (reset) {
		(void)hipMemcpyToSymbol(HIP_SYMBOL(g_stats_s), z, sizeof(z));
	}
}
#else
#define	SSTAT_ADD(i, v)	do { } while (0)
#define	SSTAT_CLK()	0ull
#endif

/*
 * DROP (the sparse + dense class, see k_scanm<.., DROP> / k_cold in nxs_gpu_scan_mask.hip; opt-in,
 * NXS_GPU_SCANS_DROP; BM25): the dense terms are not streamed; k_cold has walked the range's top until the
 * threshold passed what they can add together (U) and hands over threshold, top-k scores, output count and
 * the sparse terms' cursors.  Here the stripes hold the SPARSE terms' postings only.  The first dropped term's
 * impacts are in the bytes from the start: a stripe with >= ST_FILL_MIN sparse postings has its map FILLED
 * from the term's byte column (global_load_lds_dwordx4, converted in place by a shift), so a doc's byte is the
 * bound of its whole score and is compared with the plain threshold; a sparser stripe is zeroed and its
 * postings are given the term's ceiling (flagged on the pending list: the flush redoes their bound with the
 * real impact before scoring).  Further dropped terms stay a ceiling on the threshold's side and are refined
 * in the flush.  In the scoring a dropped term's lane takes the candidate's impact from the f32 column (one
 * load, no search).  Queries whose dropped terms have outlier lists (TF-IDF) stay on k_scanm<.., DROP>: those
 * lists have no directory.
 */
#ifdef ST_OCC
#define	ST_OCC_ATTR	__attribute__((amdgpu_waves_per_eu(ST_OCC, ST_OCC)))
#else
#define	ST_OCC_ATTR
#endif
template <int NT, bool GEN, bool DROP = false>
__global__ void __launch_bounds__(WAVE) ST_OCC_ATTR
k_scans(const scan_args_t A)
{
	static_assert(!(GEN && DROP), "the sparse + dense class is pure OR");
	static_assert(ST_WORDS == 1 || ST_WORDS == 2 || ST_WORDS == 4, "stripe width");
	constexpr int NTP = NT <= 2 ? 2 : NT <= 3 ? 3 : NT <= 4 ? 4 : NT <= 5 ? 5 : 8;	/* lanes per candidate in the flush */
	constexpr int CPR = WAVE / NTP;				/* candidates per scoring round */
	constexpr int G = WAVE / NT;				/* stripes per directory fetch */
	constexpr int SH = ST_WORDS == 1 ? 12 : ST_WORDS == 2 ? 13 : 14;
	constexpr int QSUM_MAX = 224;
	constexpr int R = ST_RING, UNR = ST_UNR;
	__shared__ __attribute__((aligned(16))) uint32_t s_map[ST_DOCS / 4];
	__shared__ uint32_t s_pend[ST_PEND];
	__shared__ uint16_t s_psum[DROP ? ST_PEND : 4];		/* DROP: the byte bound a doc was pushed with; bit 8: in a stripe whose map was
								 * NOT filled (the bound holds the filled token's ceiling, not its impact) */
	__shared__ uint32_t s_truth[GEN ? 8 : 1];

	const unsigned lane = threadIdx.x;
	const unsigned long long clk0 = SSTAT_CLK();
	(void)clk0;
	const item_t item = A.items[A.item_base + blockIdx.x];
	const uint32_t q = item.q, g = item.g;
	const qmeta_t qm = A.qmeta[q];
	const dev_query_t *Q = &A.queries[q];
	const uint32_t nt = Q->nt;
	const uint64_t seg = (uint64_t)qm.seg_first + g;

	for (uint32_t i = lane * 4; i < ST_DOCS / 4; i += WAVE * 4) {
		*(uint4 *)&s_map[i] = make_uint4(0, 0, 0, 0);
	}
	if (GEN && lane < 8) {
		s_truth[lane] = Q->truth[lane];
	}
	WAVE_SYNC();

	auto rfl32 = [](uint32_t v) -> uint32_t {
		return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
	};

	/* DROP: what the range's cold phase (k_cold) left: docs below cs_left, for the sparse terms only */
	const uint32_t dmask = DROP ? rfl32(Q->drop_mask) : 0u;
	const uint32_t *cs = A.cold_state + seg * 16;
	const uint32_t cs_left = DROP ? rfl32(cs[0]) : 0u;		/* 0: range used up */
	const uint32_t cs_nout = DROP ? rfl32(cs[1]) : 0u;
	const float cs_thr = DROP ? __uint_as_float(rfl32(cs[2])) : 0.0f;
	const uint32_t cs_ovf = DROP ? rfl32(cs[3]) : 0u;
	/*
	 * Per term, in LANES (lane t = term t; the other lanes hold zeros): list start as a posting index,
	 * the range's slice [lo, hi) of the list, the upper end e of the stripe being described, the
	 * directory row.  The stripe descriptor is computed across these lanes (next_stripe): as scalars --
	 * five values per term -- it was ~135 instructions per stripe, most of them reloads of spilled SGPRs.
	 */
	uint32_t v_pb = 0, v_lo = 0, v_e = 0;
	float tsum = 0.0f;
	if (lane < nt && lane < (uint32_t)NT) {
		const uint64_t cb = ((uint64_t)qm.seg_first + q + g) * NXSGPU_MAX_TOKENS + lane;
		v_pb = (uint32_t)Q->pbeg[lane];
		v_lo = A.cursors[cb];
		v_e = A.cursors[cb + NXSGPU_MAX_TOKENS];
		if (DROP) {
			if (((dmask >> lane) & 1) || cs_left == 0) {
				v_e = v_lo;		/* no postings as far as the stripes are concerned */
			} else {
				v_e = max(min(v_e, cs[4 + lane]), v_lo);
			}
		}
	}
	/* (read here, so that the compiler waits for these loads HERE: it does not see the waits inside the
	 * ring's asm blocks, and a wait at the values' first use -- the descriptor, inside the loop -- is an
	 * s_waitcnt vmcnt(0) at every stripe: the ring of posting windows drained each time) */
	asm volatile("" :: "v"(v_pb), "v"(v_lo), "v"(v_e));
	static_for<NT>([&](auto tc) {
		constexpr int t = decltype(tc)::value;
		if (t < (int)nt) {
			tsum += Q->tmax[t];
		}
	});
	const uint32_t lg4 = (lane < (uint32_t)NT ? lane * G : 0u) << 2;	/* ds_bpermute address of (term lane, stripe 0) */
	/* the lanes' roles: directory fetch -- term lane / G, stripe lane % G of the group;
	 * flush -- term lane % NTP of candidate lane / NTP */
	const uint32_t ft = lane / G, fj = lane % G;
	const uint32_t f_row = (ft < nt && ft < (uint32_t)NT && !(DROP && ((dmask >> ft) & 1))) ?
	    Q->bm_col[ft] * (uint32_t)(A.bm_words + 1) : 0u;
	const uint32_t mc = lane / NTP, mt = lane % NTP;	/* (lanes >= CPR * NTP idle in the flush) */
	const bool m_on = mt < nt && mt < (uint32_t)NT && mc < (uint32_t)CPR;
	const bool m_drop = DROP && m_on && ((dmask >> mt) & 1);	/* my term's impacts come from its column */
	const uint32_t m_row = (m_on && !m_drop) ? Q->bm_col[mt] * (uint32_t)(A.bm_words + 1) : 0u;
	const uint32_t m_pb = m_on ? (uint32_t)Q->pbeg[mt] : 0u;
	const uint64_t m_col = m_drop ? (uint64_t)Q->drop_col[mt] * A.dense_stride : 0ull;

	const uint32_t d_bot = (uint32_t)min((uint64_t)g * qm.group_docs, A.n_docs);
	const uint32_t d_top = DROP ? cs_left : (g + 1 == qm.n_groups) ? (uint32_t)A.n_docs :
	    (uint32_t)min((uint64_t)(g + 1) * qm.group_docs, A.n_docs);
	const int32_t w_bot = (int32_t)(d_bot >> SH);
	const int32_t w_top = d_top > d_bot ? (int32_t)((d_top - 1) >> SH) : w_bot - 1;
#if defined(NXS_STATS) || defined(NXS_CHECK_HANDOVER)
	{	/* (diagnostic build: the hand-over from k_cold must describe a part of this range) */
		const uint32_t r_top = (g + 1 == qm.n_groups) ? (uint32_t)A.n_docs : (uint32_t)min((uint64_t)(g + 1) * qm.group_docs, A.n_docs);
		bool bad = d_top > r_top || (d_top && d_top < d_bot);
		bad = bad || ballot64(v_e < v_lo) != 0;
		if (bad) {
			if (lane == 0) {
				printf("k_scans<%d,%d>: bad hand-over q %u g %u: d_bot %u d_top %u r_top %u cs %u %u %u %u\n", NT, (int)DROP, q, g,
				    d_bot, d_top, r_top, cs[0], cs[1], cs[2], cs[3]);
			}
			return;
		}
	}
#endif

	/* the stripes' lower boundaries in every list: G stripes x NT terms per load */
	/*
	 * (Load and wait in ONE asm block: a load the compiler tracks gets its s_waitcnt vmcnt(0) where the value
	 * is used -- the descriptor of EVERY stripe, whether a row was fetched for it or not -- and that drains the
	 * ring of posting windows each time.  Lanes without a row load word 0 of the array and drop it.)
	 */
	auto fetch_dir = [&](int32_t wg) __attribute__((always_inline)) -> uint32_t {
		const int32_t wj = wg - (int32_t)fj;
		const bool on = ft < (uint32_t)NT && wj >= w_bot;
		const uint32_t *ap = A.bmrank + (on ? (uint64_t)f_row + (uint64_t)wj * ST_WORDS : 0ull);
		uint32_t v;
		asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(ap) : "memory");
		return on ? v : 0u;
	};

	float hint = range_hint(A, qm, g);		/* 0 = nothing published yet */
	float top = DROP ? A.cold_top[seg * 64 + lane] : -INFINITY;
	float thr = DROP ? fmaxf(hint, cs_thr) : hint;	/* scores are > 0: 0 passes everything */
	const uint32_t kidx = A.k - 1;			/* 1 <= k <= 64 (host) */
	uint32_t n_out = cs_nout, n_pend = 0, ovf = cs_ovf, flood = 0;
	const uint64_t out_base = seg * A.seg_cap;

	/* Quantisation: as k_scanm's (q(x) = floor(x * qs) + 2, a doc's byte <= QSUM_MAX + 2 NT <= 240;
	 * a doc can only beat thr if its byte exceeds floor(thr * qs) - 1) */
	float qs = tsum > 0.0f ? (float)QSUM_MAX / tsum : 0.0f;
	/*
	 * DROP: the FIRST dropped token's impacts are in every byte from the start -- a stripe's map is FILLED
	 * from the term's byte column (q8 = ceil(255 x impact / its largest impact), nxsgpu_index::d_dense_q8)
	 * instead of being zeroed.  The scale is bent so that the conversion is a shift: qs is the largest value
	 * <= the usual one (the term counted twice: room for the shift's rounding) with cap x qs / 255 = 2^-fk,
	 * and a byte of the column becomes (q8 >> fk) + 2 >= impact x qs + 1, the bound every posting's share
	 * obeys.  The other dropped tokens stay a ceiling on the threshold's side (qU) and are refined in the
	 * flush, as before; q1max = the largest share of one sparse posting.
	 */
	uint32_t qU = 0, q1max = 0, fk = 7, fmask = 0, qUf = 0;
	uint64_t f_col = 0;
	if constexpr (DROP) {
		const uint32_t td = (uint32_t)__builtin_ctz(dmask | 0x100u) & 7u;
		const float cap = Q->tcap[td];
		const float qs0 = tsum + cap > 0.0f ? (float)QSUM_MAX / (tsum + cap) : 0.0f;
		fmask = dmask & ~(1u << td);
		f_col = (uint64_t)rfl32(Q->drop_col[td]) * A.dense_q8_stride;
		qs = qs0;
		if (cap > 0.0f) {
			float p2 = 1.0f;
			fk = 0;
			while (fk < 7 && 255.0f * p2 > qs0 * cap) {
				p2 *= 0.5f;
				fk++;
			}
			if (255.0f * p2 <= qs0 * cap) {
				qs = 255.0f * p2 / cap;
			}		/* (else: a share below two units -- the shift by 7 bounds it with the usual scale) */
		}
		fk = rfl32(fk);
		qs = __uint_as_float(rfl32(__float_as_uint(qs)));
		qUf = rfl32((uint32_t)(cap * qs) + 2);	/* the filled token's ceiling: what a posting of an UNFILLED stripe is given */
		static_for<NT>([&](auto tc) {
			constexpr int t = decltype(tc)::value;
			if (t < (int)nt) {
				if ((fmask >> t) & 1) {
					qU += (uint32_t)(Q->tcap[t] * qs) + 2;
				} else if ((dmask >> t) & 1) {
					/* (the filled token: in the bytes) */
				} else {
					q1max = max(q1max, (uint32_t)(Q->tmax[t] * qs) + 2);
				}
			}
		});
		qU = rfl32(qU);
		q1max = rfl32(q1max);
	}
	auto thr_quant = [&](float th) -> int32_t {
		return __builtin_amdgcn_readfirstlane(th > 0.0f ? (int32_t)min(th * qs, 1.0e6f) - 1 : -1) - (int32_t)qU;
	};
	int32_t thr_q = thr_quant(thr);
	/* (DROP: mature once a single sparse posting no longer passes on its own) */
	uint32_t tw = (DROP ? thr_q >= (int32_t)q1max : thr_q >= 0) ? (uint32_t)ST_W_HINTED : (uint32_t)ST_W0;

	/*
	 * Flush of the first `np` pending docs (all of them lie above every doc that is still to come):
	 * sort (descending), drop duplicates (a doc is pushed once per posting that found it above the
	 * threshold), score -- one lane per (candidate, term): the term's postings inside the candidate's
	 * 4096-doc word by the directory, a lower-bound search there, the impact; UNR rounds of CPR
	 * candidates search side by side (one dependent chain of ~10 loads for all of them); the
	 * candidate's first lane sums its tokens in token order from 0.0f --, emit what beats the
	 * threshold in descending doc order and feed the top-k register.  The docs behind the first np
	 * move to the front.
	 */
	auto flush = [&](uint32_t np) __attribute__((always_inline)) {
		constexpr int PC = ST_PEND / WAVE;
		np = rfl32(np);
		n_pend = rfl32(n_pend);
		n_out = rfl32(n_out);
		const unsigned long long fclk = SSTAT_CLK();
		(void)fclk;
		SSTAT_ADD(4, 1);
		SSTAT_ADD(5, np);
		const uint32_t nch = (np + WAVE - 1) / WAVE;
		/* what was pushed after the np -- the stripes in progress -- goes to the front afterwards */
		const uint32_t rest = np + lane < n_pend ? s_pend[np + lane] : 0;
		const uint32_t rest2 = np + WAVE + lane < n_pend ? s_pend[np + WAVE + lane] : 0;
		const uint32_t rsum = (DROP && np + lane < n_pend) ? s_psum[np + lane] : 0;
		const uint32_t rsum2 = (DROP && np + WAVE + lane < n_pend) ? s_psum[np + WAVE + lane] : 0;
		if (n_pend - np > 2 * WAVE) {
			flood = 1;		/* (the caller starts the stripes in progress again, in sub-ranges) */
			SSTAT_ADD(14, 1);
		}
		uint32_t pd[PC], rk[PC], ps[PC];
#pragma unroll
		for (int c = 0; c < PC; c++) {
			const uint32_t ei = c * WAVE + lane;
			pd[c] = ei < np ? s_pend[ei] : 0;
			ps[c] = (DROP && ei < np) ? s_psum[ei] : 0;
			rk[c] = 0;
		}
		WAVE_SYNC();
#pragma unroll
		for (int cj = 0; cj < PC; cj++) {
			if ((uint32_t)cj < nch) {
				const uint32_t nj = min(np - cj * WAVE, (uint32_t)WAVE);
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
			const uint32_t ei = c * WAVE + lane;
			if (ei < np) {
				s_pend[rk[c]] = pd[c];
				if (DROP) {
					s_psum[rk[c]] = (uint16_t)ps[c];
				}
			}
		}
		WAVE_SYNC();
		/* unique docs to the front (stable) */
		uint32_t nu = 0, prev_last = 0xffffffffu;
		for (uint32_t off = 0; off < np; off += WAVE) {
			const uint32_t ei = off + lane;
			const bool valid = ei < np;
			const uint32_t d = valid ? s_pend[ei] : 0;
			const uint32_t before = (uint32_t)__shfl_up((int)d, 1);
			const bool dup = valid && (lane ? before == d : prev_last == d);
			bool keep = valid && !dup;
			if constexpr (DROP) {
				/*
				 * Docs of UNFILLED stripes were pushed on the ceiling of the filled token, all docs on the CEILING of the
				 * other dropped tokens (qU); with their real
				 * dense impacts -- one load per doc and dropped term, all lanes at once -- the
				 * bound is redone and only what can still beat the threshold is scored.  A doc
				 * is pushed once per posting that found it above the threshold (adjacent after
				 * the sort, at most one per term): its complete byte bound is the LARGEST of them.
				 */
				uint32_t sumq = keep ? s_psum[ei] : 0u;
#pragma unroll
				for (int kk = 1; kk < NT; kk++) {
					if (keep && ei + kk < np && s_pend[ei + kk] == d) {
						sumq = max(sumq, (uint32_t)s_psum[ei + kk]);
					}
				}
				const bool unfilled = (sumq >> 8) != 0;
				sumq &= 0xffu;
				uint32_t qd = 0;
				static_for<NT>([&](auto tc) {
					constexpr int t = decltype(tc)::value;
					if ((dmask >> t) & 1) {
						const bool mine = ((fmask >> t) & 1) || unfilled;	/* (the filled token: only where its impact is not in the byte) */
						const uint64_t cbase = (uint64_t)rfl32(Q->drop_col[t]) * A.dense_stride;
						const uint32_t xb = (keep && mine) ? A.dense_col[cbase + d] : 0xffffffffu;
						if (xb != 0xffffffffu) {
							qd += (uint32_t)(__uint_as_float(xb) * qs) + 2;
						}
					}
				});
				if (unfilled) {
					sumq -= qUf;
				}
				keep = keep && (int32_t)(sumq + qd) > thr_q + (int32_t)qU;
				SSTAT_ADD(14, __popcll(ballot64(keep)));
			}
			const uint64_t m = ballot64(keep);
			prev_last = (uint32_t)__builtin_amdgcn_readlane((int)d, 63);
			WAVE_SYNC();
			if (keep) {
				s_pend[nu + lanes_below(m)] = d;
			}
			nu += (uint32_t)__popcll(m);
			WAVE_SYNC();
		}
		nu = rfl32(nu);
		SSTAT_ADD(6, nu);

		for (uint32_t c0 = 0; c0 < nu; c0 += CPR * UNR) {
			uint32_t doc[UNR], sl[UNR], sh_[UNR], hdoc[UNR];
			float himp[UNR];
			bool inr[UNR];
			SSTAT_ADD(7, 1);
#pragma unroll
			for (int u = 0; u < UNR; u++) {
				const uint32_t ci = c0 + u * CPR + mc;
				inr[u] = ci < nu && mc < (uint32_t)CPR;
				doc[u] = s_pend[min(ci, nu - 1)];
				/* (lanes without a term or a candidate search an empty run of list 0) */
				const uint64_t ri = (uint64_t)m_row + (doc[u] >> 12);
				const uint32_t a = A.bmrank[ri], b = A.bmrank[ri + 1];
				const bool act = inr[u] && m_on && !m_drop;
				sl[u] = act ? a : 0u;
				sh_[u] = act ? b : 0u;
				hdoc[u] = 0xffffffffu;		/* the posting at sh_, once a probe has put sh_ there */
				himp[u] = 0.0f;
				if constexpr (DROP) {
					/* a dense term that left the scan: its impact for this doc from the term's column
					 * (lanes of other terms read a valid word of column 0) */
#ifdef NXS_DBG_CLAMP
					const uint32_t xb = A.dense_col[m_col + min(doc[u], (uint32_t)A.n_docs - 1)];
#else
					const uint32_t xb = A.dense_col[m_col + doc[u]];
#endif
					if (m_drop && inr[u] && xb != 0xffffffffu) {
						hdoc[u] = doc[u];
						himp[u] = __uint_as_float(xb);
					}
				}
			}
			/*
			 * Lower bound of the doc in [sl, sh_), four-ary: three probes per step, all rounds' probes
			 * in flight together -- a run of 330 postings (a term holding 8 % of the docs) takes 5
			 * dependent steps instead of 9, and the posting the bound lands on has been fetched by the
			 * probe that moved sh_ there (no load after the search).
			 */
			for (;;) {
				bool more = false;
#pragma unroll
				for (int u = 0; u < UNR; u++) {
					more = more || sl[u] < sh_[u];
				}
				if (!ballot64(more)) {
					break;
				}
				SSTAT_ADD(8, 1);
				posting_t pr[UNR][3];
				uint32_t qp[UNR][3];
#pragma unroll
				for (int u = 0; u < UNR; u++) {	/* (unconditional: a finished lane probes sl == sh_, a valid address) */
					const uint32_t len = sh_[u] - sl[u];
					qp[u][0] = sl[u] + (len >> 2);
					qp[u][1] = sl[u] + (len >> 1);
					qp[u][2] = sl[u] + (len >> 1) + (len >> 2);
#pragma unroll
					for (int k = 0; k < 3; k++) {
						pr[u][k] = A.post[(uint64_t)(m_pb + qp[u][k])];
					}
				}
#pragma unroll
				for (int u = 0; u < UNR; u++) {
					if (sl[u] < sh_[u]) {
						if (pr[u][0].doc >= doc[u]) {
							sh_[u] = qp[u][0]; hdoc[u] = pr[u][0].doc; himp[u] = pr[u][0].imp;
						} else if (pr[u][1].doc >= doc[u]) {
							sl[u] = qp[u][0] + 1;
							sh_[u] = qp[u][1]; hdoc[u] = pr[u][1].doc; himp[u] = pr[u][1].imp;
						} else if (pr[u][2].doc >= doc[u]) {
							sl[u] = qp[u][1] + 1;
							sh_[u] = qp[u][2]; hdoc[u] = pr[u][2].doc; himp[u] = pr[u][2].imp;
						} else {
							sl[u] = qp[u][2] + 1;
						}
					}
				}
			}
#pragma unroll
			for (int u = 0; u < UNR; u++) {
				const bool hit = hdoc[u] == doc[u];
				const float x = hit ? himp[u] : 0.0f;
				/* the candidate's tokens, token order, from 0.0f (an absent token adds +0.0f: the same bits) */
				const uint32_t gb = mc * NTP;
				const uint32_t pm = (uint32_t)(ballot64(hit) >> min(gb, 63u)) & ((1u << NTP) - 1);
				float sc = 0.0f;
#pragma unroll
				for (int t = 0; t < NT; t++) {
					sc += __shfl(x, (int)min(gb + t, 63u));
				}
				bool match = inr[u] && mt == 0 && pm != 0;	/* (a round beyond the last candidate: no lane matches) */
				if (GEN) {
					match = match && ((s_truth[pm >> 5] >> (pm & 31)) & 1);
				}
				const bool cand = match && sc > thr;
				uint64_t bal = ballot64(cand);
				if (!bal) {
					continue;
				}
				const uint32_t ne = __popcll(bal);
				const bool room = n_out + ne <= A.seg_cap;
				if (!room) {
					ovf = 1;
					SSTAT_ADD(15, 1);
				}
				if (room && cand) {
					/* lanes are in descending doc order */
					const uint64_t o = out_base + n_out + lanes_below(bal);
					A.cand_doc[o] = doc[u];
					A.cand_sc[o] = sc;
				}
				n_out += ne;
				while (bal) {
					const int L = __builtin_ctzll(bal);
					const float vv = __shfl(sc, L);
					/* branch-free insert into the sorted top-k register */
					const bool ins = vv > thr;
					const uint32_t pos = __popcll(ballot64(top >= vv));
					const float up = __shfl_up(top, 1);
					const float ntop = (lane < pos) ? top : (lane == pos ? vv : up);
					top = ins ? ntop : top;
					thr = ins ? fmaxf(__shfl(top, kidx), hint) : thr;
					bal &= bal - 1;
				}
			}
		}
		WAVE_SYNC();
		if (np + lane < n_pend) {
			s_pend[lane] = rest;
		}
		if (np + WAVE + lane < n_pend) {
			s_pend[WAVE + lane] = rest2;
		}
		if (DROP && np + lane < n_pend) {
			s_psum[lane] = (uint16_t)rsum;
		}
		if (DROP && np + WAVE + lane < n_pend) {
			s_psum[WAVE + lane] = (uint16_t)rsum2;
		}
		WAVE_SYNC();
		n_pend = min(n_pend - np, 2u * WAVE);
		thr_q = thr_quant(thr);
		SSTAT_ADD(10, SSTAT_CLK() - fclk);
	};

	/*
	 * The loader: the stripe whose windows are being requested -- its slice [s, e) of every list (e is
	 * the stripe above's s), as the flat run [0, n_ld): lane i of the run belongs to the last term t
	 * with c[t] <= i and is posting i + dsel[t] of the posting array.
	 */
	uint32_t c[NT], dsel0 = 0, n_ld = 0, i0_ld = 0;
	uint32_t ddv[NT];	/* dsel[t] - dsel[t - 1], in vector registers: the per-window select chain takes them as they are
				 * (from scalar registers every select needs a copy first: four per window) */
	uint32_t cnt_v = 0;	/* the described stripe's slice of list t, in lane t: [v_e, v_e + cnt_v) */
	int32_t w_ld = w_top + 1, j_ld = -1;
	uint32_t rkv = fetch_dir(w_top);
	static_for<NT>([&](auto tc) {
		constexpr int t = decltype(tc)::value;
		c[t] = 0;
		ddv[t] = 0;
	});
	auto next_stripe = [&]() __attribute__((always_inline)) -> bool {
		for (;;) {
			w_ld = (int32_t)rfl32((uint32_t)w_ld);
			j_ld = (int32_t)rfl32((uint32_t)j_ld);
			if (w_ld <= w_bot) {
				n_ld = i0_ld = 0;
				return false;
			}
			w_ld--;
			j_ld++;
			if (j_ld == G) {
				/*
				 * Loaded and used at once: one exposed load per G stripes.  (A copy requested G stripes
				 * ahead cost more: the compiler shuffled it between registers at EVERY stripe, behind an
				 * s_waitcnt vmcnt(0) that drained the window ring.  The asm keeps this a branch.)
				 */
				asm volatile("" ::: "memory");
				rkv = fetch_dir(w_ld);
				j_ld = 0;
			}
			/* lane t: the stripe's start in list t (directory, clamped to the range's slice), its count, the
			 * exclusive prefix sum over the terms (row-shift DPP: the terms sit in lanes 0 .. NT - 1 of row 0) */
			const uint32_t raw = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(lg4 + ((uint32_t)j_ld << 2)), (int)rkv);
			const uint32_t sl_ = min(max(raw, v_lo), v_e);
			const uint32_t cn_ = v_e - sl_;
			uint32_t inc = cn_;
			inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xf, 0xf, false);
			inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xf, 0xf, false);
			inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xf, 0xf, false);
			const uint32_t cl_ = inc - cn_;
			const uint32_t dl_ = v_pb + sl_ - cl_;
			/* (as an ADD of the negated neighbour: the compiler folds `x - dpp(x)` into v_subrev_u32_dpp, and what
			 * that instruction delivered on gfx950 was dpp(x) - x -- measured against the scalar computation; the folded adds above
			 * are commutative) */
			const uint32_t ndl = 0u - dl_;
			const uint32_t dd_ = dl_ + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ndl, 0x111, 0xf, 0xf, false);
			v_e = sl_;
			cnt_v = cn_;
			const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)inc, NT - 1);
			dsel0 = (uint32_t)__builtin_amdgcn_readlane((int)dl_, 0);
			static_for<NT>([&](auto tc) {
				constexpr int t = decltype(tc)::value;
				if (t > 0) {
					c[t] = (uint32_t)__builtin_amdgcn_readlane((int)cl_, t);
					ddv[t] = (uint32_t)__builtin_amdgcn_readlane((int)dd_, t);
					/* (the asm keeps the copy in a vector register) */
					asm volatile("" : "+v"(ddv[t]));
				}
			});
			SSTAT_ADD(1, 1);
			SSTAT_ADD(12, n);
			n_ld = n;
			i0_ld = 0;
			if (n) {
				return true;
			}
		}
	};
	uint32_t qadd = 0;	/* DROP: what a posting of the stripe in hand is given on top of its byte (prep, below) */
	auto push = [&](uint64_t cm, uint32_t doc, uint32_t sum) __attribute__((always_inline)) {
		const uint32_t np = (uint32_t)__popcll(cm);
		if (n_pend + np <= ST_PEND && lane_of(cm)) {
			s_pend[n_pend + lanes_below(cm)] = doc;
			if (DROP) {
				s_psum[n_pend + lanes_below(cm)] = (uint16_t)(sum | (qadd ? 0x100u : 0u));
			}
		}
		n_pend += np;
	};
	auto wipe = [&]() __attribute__((always_inline)) {
		WAVE_SYNC();
#pragma unroll
		for (uint32_t i = 0; i < ST_DOCS / 4; i += WAVE * 4) {
			*(uint4 *)&s_map[i + lane * 4] = make_uint4(0, 0, 0, 0);
		}
		WAVE_SYNC();
	};

	/*
	 * DROP: stripe w's bytes from the filled token's column -- eight 1 KB pieces straight into LDS
	 * (global_load_lds_dwordx4: no registers), waited for on the spot (they are the youngest vector memory
	 * operations: vmcnt(0), the ring's windows have landed by then too), then converted in place.
	 */
	uint32_t w_filled = 0;		/* stripe + 1 whose bytes the map holds (0: none) */
	auto fill = [&](int32_t w) __attribute__((always_inline)) {
		if constexpr (DROP) {
			const uint8_t *src = A.dense_q8 + f_col + (uint64_t)(uint32_t)w * ST_DOCS + lane * 16;
			const uint32_t fm = 0x01010101u * (0xffu >> fk);
			WAVE_SYNC();
#pragma unroll
			for (uint32_t i = 0; i < ST_DOCS / 1024; i++) {
				__builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + i * 1024),
				    (void __attribute__((address_space(3))) *)&s_map[i * 256], 16, 0, 0);
			}
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			WAVE_SYNC();
#pragma unroll 2
			for (uint32_t i = 0; i < ST_DOCS / 4; i += WAVE * 4) {
				uint4 x = *(uint4 *)&s_map[i + lane * 4];
				x.x = ((x.x >> fk) & fm) + 0x02020202u;
				x.y = ((x.y >> fk) & fm) + 0x02020202u;
				x.z = ((x.z >> fk) & fm) + 0x02020202u;
				x.w = ((x.w >> fk) & fm) + 0x02020202u;
				*(uint4 *)&s_map[i + lane * 4] = x;
			}
			WAVE_SYNC();
			w_filled = (uint32_t)w + 1;
			SSTAT_ADD(15, 1);
		}
	};
	/* DROP: stripe w's map before its first posting -- filled (then qadd = 0) or zeroed (qadd = the ceiling) */
	auto prep = [&](int32_t w, bool filled) __attribute__((always_inline)) {
		if (filled) {
			fill(w);
			qadd = 0;
		} else {
			wipe();
			w_filled = (uint32_t)w + 1;
			qadd = qUf;
		}
	};

	/*
	 * ---- cold start: stripes in doc sub-ranges, until a sub-range is a whole stripe ----
	 * A sub-range [rlo, rhi) of the stripe is, in every list, the postings right below the ones the
	 * sub-range above it took: a cursor per term (cur[t], moving down from the stripe's upper end) and
	 * plain 64-posting windows under it, taken while they still hold docs >= rlo -- the lanes at or
	 * above rlo are a suffix of the window (docs ascend), their count moves the cursor.  A sub-range
	 * costs what it holds, whatever the stripe holds (a pass over ALL the stripe's windows with a doc
	 * filter made a redone sub-range of a dense stripe a millisecond's work: the kernel's tail).
	 */
	for (;;) {
	while (tw < (uint32_t)ST_DOCS && !ovf) {
		tw = rfl32(tw);
		ovf = rfl32(ovf);
		if (!next_stripe()) {
			break;
		}
		if (DROP) {
			prep(w_ld, n_ld >= (uint32_t)ST_FILL_MIN);
		}
		/* the stripe's runs, list-relative: [low[t], cur[t]) */
		uint32_t low[NT], cur[NT], cur0[NT], pb[NT];
		static_for<NT>([&](auto tc) {
			constexpr int t = decltype(tc)::value;
			low[t] = (uint32_t)__builtin_amdgcn_readlane((int)v_e, t);
			cur[t] = low[t] + (uint32_t)__builtin_amdgcn_readlane((int)cnt_v, t);
			pb[t] = (uint32_t)__builtin_amdgcn_readlane((int)v_pb, t);
		});
		uint32_t rhi = ST_DOCS;		/* docs of the stripe still to take: rel < rhi */
		while (rhi > 0 && !ovf) {
			rhi = rfl32(rhi);
			n_pend = rfl32(n_pend);
			if (n_pend > ST_PEND - 64 && n_pend) {	/* (room for a sub-range's pushes: a burst is then the sub-range's own) */
				WAVE_SYNC();
				flush(n_pend);
			}
			tw = rfl32(tw);
			thr_q = (int32_t)rfl32((uint32_t)thr_q);
			const uint32_t rlo = rhi > tw ? rhi - tw : 0;
			const uint32_t n_before = n_pend;
			SSTAT_ADD(2, 1);
			/* (every term's first window under its cursor: requested together, one memory latency for all) */
			posting_t pw[NT];
			static_for<NT>([&](auto tc) {
				constexpr int t = decltype(tc)::value;
				cur[t] = rfl32(cur[t]);
				cur0[t] = cur[t];
				pw[t].doc = 0;
				pw[t].imp = 0.0f;
				if (cur[t] > low[t]) {
					const uint32_t wb = cur[t] > low[t] + WAVE ? cur[t] - WAVE : low[t];
					pw[t] = A.post[(uint64_t)(pb[t] + wb + min(lane, cur[t] - wb - 1))];
				}
			});
			static_for<NT>([&](auto tc) {
				constexpr int t = decltype(tc)::value;
				bool first = true;
				while (cur[t] > low[t]) {
					const uint32_t wb = cur[t] > low[t] + WAVE ? cur[t] - WAVE : low[t];
					const uint32_t cw = cur[t] - wb;
					posting_t p = pw[t];
					if (!first) {
						p = A.post[(uint64_t)(pb[t] + wb + min(lane, cw - 1))];
					}
					first = false;
					const uint32_t rel = p.doc & (ST_DOCS - 1);
					const bool valid = lane < cw && rel >= rlo;
					const uint32_t shv = (rel & 3) * 8;
					/* floor + 2 >= the exact ceiling whatever the f32 product rounds to */
					const uint32_t qq = (uint32_t)(p.imp * qs) + 2;
					const uint32_t oldv = __hip_atomic_fetch_add(&s_map[rel >> 2], valid ? (qq << shv) : 0u,
					    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
					const uint32_t sum = ((oldv >> shv) & 0xffu) + qq + (DROP ? qadd : 0u);
					const uint64_t vm = ballot64(valid);
					const uint64_t cm = ballot64(valid && (int32_t)sum > thr_q);
					SSTAT_ADD(3, 1);
					if (cm) {
						push(cm, p.doc, sum);
					}
					const uint32_t nv = (uint32_t)__popcll(vm);
					cur[t] = rfl32(cur[t] - nv);
					if (nv < cw) {
						break;		/* the rest of the window lies below the sub-range */
					}
				}
			});
			if (n_pend > ST_PEND && tw > (uint32_t)ST_W0) {
				/* a burst (the threshold is still weak for a sub-range this wide): the sub-range once
				 * more, a quarter as wide -- what it pushed is dropped, its bytes are wiped (the
				 * sub-ranges above it are done: their bytes are not needed any more) */
				n_pend = n_before;
				tw = max(tw / 4, (uint32_t)ST_W0);
				static_for<NT>([&](auto tc) {
					constexpr int t = decltype(tc)::value;
					cur[t] = cur0[t];
				});
				if (DROP) {
					prep(w_ld, n_ld >= (uint32_t)ST_FILL_MIN);
				} else {
					wipe();
				}
				SSTAT_ADD(13, 1 << 16);
				continue;
			}
			const uint32_t n_tile = n_pend - n_before, width = rhi - rlo;
			rhi = rlo;
			/*
			 * Whole stripes as soon as they look affordable: a threshold exists and, at this
			 * sub-range's rate, a stripe would push at most ST_GO docs.  (k_scanm's rule alone --
			 * double while a sub-range pushes <= 8 -- keeps a dense query whose sub-ranges push a
			 * dozen each in narrow sub-ranges for its whole range: one window per memory latency,
			 * a millisecond for 40 000 postings -- the kernel's tail.)
			 */
			if (thr_q >= (DROP ? (int32_t)q1max : 0) && n_tile * ((uint32_t)ST_DOCS / max(width, 1u)) <= (uint32_t)ST_GO) {
				tw = ST_DOCS;
			} else if (n_tile <= 8) {
				tw = min(tw * 2, (uint32_t)ST_DOCS);
			} else if (n_tile > 48) {
				tw = max(tw / 2, (uint32_t)ST_W0);
			}
			if (n_pend > ST_PEND) {
				ovf = 1;
			} else if (n_pend >= 24 || (n_pend && thr_q < (DROP ? (int32_t)q1max : 0))) {
				WAVE_SYNC();
				flush(n_pend);
			}
		}
		i0_ld = n_ld;		/* the stripe is used up */
		if (!DROP) {
			wipe();		/* (DROP: the next stripe's fill overwrites every byte) */
		}
	}

	/*
	 * ---- whole stripes: a stream of windows, R in flight across stripe ends ----
	 * Slot s of the ring is the AGPR pair s (bpair_request / bpair_take, nxs_gpu_dev.h): the compiler
	 * does not see these loads -- with ordinary loads it waits for vmcnt(0) in front of every slot (the
	 * loader's directory fetch and the flush are conditional loads inside the loop: it cannot count), i.e.
	 * one window per memory latency.  Every turn of a slot issues exactly ONE ring load (a dummy once
	 * the range is used up), in slot order, so when slot s is taken the R - 1 other slots' loads are
	 * younger than its own and vmcnt(R - 1) means it has landed; loads the compiler issues in between
	 * only make the wait longer than needed.
	 */
	{
		uint32_t cnt[R], last[R];
		const int32_t w_ld0 = (int32_t)rfl32((uint32_t)w_ld);	/* every stripe down to this one is done */
		/* the next window of the stream: its address, how many of its lanes count (0: nothing left),
		 * whether it ends its stripe (then: the stripe's number + 1) */
		auto advance = [&](const posting_t *&pa, uint32_t &pc, uint32_t &pl) __attribute__((always_inline)) {
			i0_ld = rfl32(i0_ld);
			n_ld = rfl32(n_ld);
			pc = pl = 0;
			pa = A.post;
			if (i0_ld >= n_ld && !next_stripe()) {
				return;
			}
			/*
			 * Lane `lane` takes flat index i0 + lane (clamped to the run).  Which list that is:
			 * the window starts inside the last term t with c[t] <= i0 -- a scalar selection --,
			 * and only the terms whose first flat index lies INSIDE the window (c[t] in (i0, i0 + 64):
			 * a wave-uniform test per term) cost vector instructions; a window inside one term's run
			 * -- most windows of the longer lists -- costs two.
			 */
			const uint32_t i0 = i0_ld, iend = min(i0 + WAVE, n_ld);
			const uint32_t ic = min(i0 + lane, iend - 1);
			uint32_t dv = ic + dsel0;
			static_for<NT>([&](auto tc) {
				constexpr int t = decltype(tc)::value;
				if (t > 0) {
					dv += ic >= c[t] ? ddv[t] : 0u;
				}
			});
			pa = A.post + dv;
			pc = iend - i0;
			if (DROP) {
				/* (every window names its stripe: the consumer fills the map at a stripe's FIRST window) */
				pl = (((uint32_t)w_ld + 1) << 2) | (n_ld >= (uint32_t)ST_FILL_MIN ? 2u : 0u) | (iend >= n_ld ? 1u : 0u);
			} else {
				pl = iend >= n_ld ? (uint32_t)w_ld + 1 : 0u;	/* (the window ends stripe w_ld) */
			}
			i0_ld = i0 + WAVE;
			SSTAT_ADD(3, 1);
		};
		static_for<R>([&](auto sc_) {
			constexpr int s = decltype(sc_)::value;
			const posting_t *pa;
			cnt[s] = last[s] = 0;
			pa = A.post;
			if (!ovf) {
				advance(pa, cnt[s], last[s]);
			}
			bpair_request<s>(pa);
		});
		uint32_t mark = n_pend;		/* pending docs of finished stripes */
		int32_t w_mark = w_ld0;		/* ... the last of which was this one */
		uint32_t rounds = 0;
		for (;;) {
			n_pend = rfl32(n_pend);
			mark = rfl32(mark);
			ovf = rfl32(ovf);
			thr_q = (int32_t)rfl32((uint32_t)thr_q);
			rounds = rfl32(rounds);
			uint32_t live = 0;
#pragma unroll
			for (int s = 0; s < R; s++) {
				cnt[s] = rfl32(cnt[s]);
				last[s] = rfl32(last[s]);
				live |= cnt[s];
			}
			if (!live || ovf) {
				break;
			}
			static_for<R>([&](auto sc_) {
				constexpr int s = decltype(sc_)::value;
				const posting_t *pa;
				uint32_t ncnt, nlast, pdoc;
				float pimp;
				advance(pa, ncnt, nlast);
				vm_wait_younger(R - 1);
				bpair_take<s, 63>(pdoc, pimp, pa);
				if (cnt[s]) {
					uint32_t ends = last[s];	/* the stripe's number + 1 if this window ends it */
					if (DROP) {
						const uint32_t ws = last[s] >> 2;
						ends = (last[s] & 1) ? ws : 0u;
						if (ws != w_filled) {
							prep((int32_t)ws - 1, (last[s] & 2) != 0);
						}
					}
					const uint32_t rel = pdoc & (ST_DOCS - 1);
					const uint32_t shv = (rel & 3) * 8;
					const uint32_t qq = (uint32_t)(pimp * qs) + 2;
					const bool valid = lane < cnt[s];	/* (not a mask built by 1 << cnt: cnt may be 64) */
					const uint32_t oldv = __hip_atomic_fetch_add(&s_map[rel >> 2], valid ? (qq << shv) : 0u,
					    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
					const uint32_t sum = ((oldv >> shv) & 0xffu) + qq + (DROP ? qadd : 0u);
					const uint64_t cm = ballot64(valid && (int32_t)sum > thr_q);
					if (cm) {
						push(cm, pdoc, sum);
					}
					if (ends) {
						if (!DROP) {
							wipe();
						}
						/* (a stripe whose pushes did not all fit is not finished: the flood check
						 * below sends the loader back to it -- its docs are not on the list) */
						if (n_pend <= (uint32_t)ST_PEND) {
							mark = n_pend;
							w_mark = (int32_t)ends - 1;
						}
					}
				}
				cnt[s] = ncnt;
				last[s] = nlast;
			});
			if (n_pend > ST_PEND) {
				flood = 1;
			} else if (mark >= ST_FLUSH) {
				WAVE_SYNC();
				flush(mark);
				mark = 0;
			}
			flood = rfl32(flood);
			if (flood) {
				break;
			}
			if ((++rounds & 7) == 0) {
				/* a higher range may have published since */
				hint = fmaxf(hint, range_hint(A, qm, g));
				if (hint > thr) {
					thr = hint;
					thr_q = thr_quant(thr);
				}
			}
		}
#ifdef NXS_DBG_NORESTART
		if (flood) {
			ovf = 1;
			flood = 0;
		}
#endif
		if (!flood) {
			break;
		}
		/*
		 * More docs pushed than the pending list takes (a weak hint, a threshold that is still
		 * immature): the stripes in progress -- everything below stripe w_mark, the last one whose docs
		 * are all on the list or scored -- start again in doc sub-ranges.  What they pushed is dropped,
		 * the loader goes back to the stripe below w_mark (its upper ends from the directory), the
		 * windows still in flight are simply requested over.
		 */
		SSTAT_ADD(13, 1);
		flood = 0;
		n_pend = min(n_pend, mark);
		{	/* (the range's upper ends and the directory rows again, in term lanes: not kept live for this) */
			uint32_t dv = 0, vh = 0;
			if (lane < nt && lane < (uint32_t)NT) {
				const uint64_t cb = ((uint64_t)qm.seg_first + q + g) * NXSGPU_MAX_TOKENS + lane;
				const bool dropped = DROP && ((dmask >> lane) & 1);
				vh = A.cursors[cb + NXSGPU_MAX_TOKENS];
				if (DROP) {
					vh = (dropped || cs_left == 0) ? v_lo : max(min(vh, cs[4 + lane]), v_lo);
				}
				if (w_mark <= w_top && !dropped) {
					dv = A.bmrank[(uint64_t)Q->bm_col[lane] * (uint32_t)(A.bm_words + 1) + (uint64_t)w_mark * ST_WORDS];
				}
			}
			v_e = w_mark > w_top ? vh : min(max(dv, v_lo), vh);
		}
		w_ld = w_mark;
		j_ld = -1;
		rkv = fetch_dir(w_mark - 1);
		n_ld = i0_ld = 0;
		tw = ST_DOCS / 4;
		w_filled = 0;
		wipe();
	}
	}
	if (n_pend && !ovf) {
		WAVE_SYNC();
		flush(n_pend);
	}

	SSTAT_ADD(0, 1);
	SSTAT_ADD(9, n_out);
	SSTAT_ADD(11, SSTAT_CLK() - clk0);
	SSTAT_ADD(13, ovf ? 1 : 0);
	if (!ovf) {
		range_publish(A, seg, __shfl(top, kidx));
	}
	if (lane == 0) {
		A.seg_count[seg] = ovf ? 0 : n_out;
		if (ovf) {
			/* once more on the accumulator tiles (scan_args_t::retry_items); a full retry list sends
			 * the query to the exact passes */
			const uint32_t ri = (A.retry_items && !(Q->qflags & 1)) ? atomicAdd(A.retry_count, 1u) : 0xffffffffu;
			if (ri < A.retry_cap) {
				A.retry_items[ri] = item;
			} else {
				A.overflow[q] = 1;
			}
		}
	}
}

/* the sparse + dense OR class on stripes: its cold phase (k_cold, nxs_gpu_scan_mask.hip) has run on this stream */
void
nxs_launch_scans_drop(uint32_t nt_bucket, unsigned grid_, hipStream_t st, const scan_args_t &a)
{
	const dim3 grid(grid_), block(WAVE);

	switch (nt_bucket) {
	case 2:
	case 3: hipLaunchKernelGGL((k_scans<3, false, true>), grid, block, 0, st, a); break;
	case 5: hipLaunchKernelGGL((k_scans<5, false, true>), grid, block, 0, st, a); break;
	default: hipLaunchKernelGGL((k_scans<8, false, true>), grid, block, 0, st, a); break;
	}
}

/* k_scans, top-k filter pass (1 <= k <= 64); gen: the expression is more than an OR */
void
nxs_launch_scans(uint32_t nt_bucket, bool gen, unsigned grid_, hipStream_t st, const scan_args_t &a)
{
	const dim3 grid(grid_), block(WAVE);

	if (!gen) {
		switch (nt_bucket) {
		case 2:		/* two tokens: the third slot stays empty */
		case 3: hipLaunchKernelGGL((k_scans<3, false>), grid, block, 0, st, a); break;
		case 5: hipLaunchKernelGGL((k_scans<5, false>), grid, block, 0, st, a); break;
		default: hipLaunchKernelGGL((k_scans<8, false>), grid, block, 0, st, a); break;
		}
	} else {
		switch (nt_bucket) {
		case 2:
		case 3: hipLaunchKernelGGL((k_scans<3, true>), grid, block, 0, st, a); break;
		case 5: hipLaunchKernelGGL((k_scans<5, true>), grid, block, 0, st, a); break;
		default: hipLaunchKernelGGL((k_scans<8, true>), grid, block, 0, st, a); break;
		}
	}
}

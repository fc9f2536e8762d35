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
#ifndef ST_CH
#define	ST_CH		4		/* posting windows in flight while a stripe is walked in sub-ranges (cold start) */
#endif
#ifndef ST_RING
#define	ST_RING		6		/* posting windows in flight once stripes are taken whole (across stripe ends) */
#endif
#ifndef ST_UNR
#define	ST_UNR		4		/* scoring rounds whose searches run side by side */
#endif
#ifndef ST_ADV_SCALAR
#define	ST_ADV_SCALAR	0		/* the window's lists by scalar tests (else: a compare / select per term and lane) */
#endif
#ifndef ST_REDO
#define	ST_REDO		1		/* a sub-range that overfills the pending list is walked again, narrower */
#endif
#ifndef ST_PROC_FAST
#define	ST_PROC_FAST	0		/* full windows without lane masks */
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

template <int NT, bool GEN>
__global__ void __launch_bounds__(WAVE)
k_scans(const scan_args_t A)
{
	static_assert(ST_WORDS == 1 || ST_WORDS == 2 || ST_WORDS == 4, "stripe width");
	constexpr int NTP = NT <= 2 ? 2 : NT <= 3 ? 3 : NT <= 4 ? 4 : NT <= 5 ? 5 : 8;	/* lanes per candidate in the flush */
	constexpr int CPR = WAVE / NTP;				/* candidates per scoring round */
	constexpr int G = WAVE / NT;				/* stripes per directory fetch */
	constexpr int SH = ST_WORDS == 1 ? 12 : ST_WORDS == 2 ? 13 : 14;
	constexpr int QSUM_MAX = 224;
	constexpr int R = ST_RING, UNR = ST_UNR;
	__shared__ __attribute__((aligned(16))) uint32_t s_map[ST_DOCS / 4];
	__shared__ uint32_t s_pend[ST_PEND];
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

	/* per term (wave-uniform): list start as a posting index, the range's slice [lo, hi) of the list */
	uint32_t pb[NT], lo[NT], hi[NT], e[NT];
	float tsum = 0.0f;
	static_for<NT>([&](auto tc) {
		constexpr int t = decltype(tc)::value;
		pb[t] = lo[t] = hi[t] = 0;
		if (t < (int)nt) {
			const uint64_t cb = ((uint64_t)qm.seg_first + q + g) * NXSGPU_MAX_TOKENS + t;
			pb[t] = rfl32((uint32_t)Q->pbeg[t]);
			lo[t] = rfl32(A.cursors[cb]);
			hi[t] = rfl32(A.cursors[cb + NXSGPU_MAX_TOKENS]);
			tsum += Q->tmax[t];
		}
		e[t] = hi[t];
	});
	/* the lanes' roles: directory fetch -- term lane / G, stripe lane % G of the group;
	 * flush -- term lane % NTP of candidate lane / NTP */
	const uint32_t ft = lane / G, fj = lane % G;
	const uint32_t f_row = (ft < nt && ft < (uint32_t)NT) ? Q->bm_col[ft] * (uint32_t)(A.bm_words + 1) : 0u;
	const uint32_t mc = lane / NTP, mt = lane % NTP;	/* (lanes >= CPR * NTP idle in the flush) */
	const bool m_on = mt < nt && mt < (uint32_t)NT && mc < (uint32_t)CPR;
	const uint32_t m_row = m_on ? Q->bm_col[mt] * (uint32_t)(A.bm_words + 1) : 0u;
	const uint32_t m_pb = m_on ? (uint32_t)Q->pbeg[mt] : 0u;

	const uint32_t d_bot = (uint32_t)min((uint64_t)g * qm.group_docs, A.n_docs);
	const uint32_t d_top = (g + 1 == qm.n_groups) ? (uint32_t)A.n_docs :
	    (uint32_t)min((uint64_t)(g + 1) * qm.group_docs, A.n_docs);
	const int32_t w_bot = (int32_t)(d_bot >> SH);
	const int32_t w_top = d_top > d_bot ? (int32_t)((d_top - 1) >> SH) : w_bot - 1;

	/* the stripes' lower boundaries in every list: G stripes x NT terms per load */
	auto fetch_dir = [&](int32_t wg) -> uint32_t {
		const int32_t wj = wg - (int32_t)fj;
		uint32_t v = 0;
		if (ft < (uint32_t)NT && wj >= w_bot) {
			v = A.bmrank[(uint64_t)f_row + (uint64_t)wj * ST_WORDS];
		}
		return v;
	};

	float hint = range_hint(A, qm, g);		/* 0 = nothing published yet */
	float top = -INFINITY;
	float thr = hint;				/* scores are > 0: 0 passes everything */
	const uint32_t kidx = A.k - 1;			/* 1 <= k <= 64 (host) */
	uint32_t n_out = 0, n_pend = 0, ovf = 0;
	const uint64_t out_base = seg * A.seg_cap;

	/* Quantisation: as k_scanm's (q(x) = floor(x * qs) + 2, a doc's byte <= QSUM_MAX + 2 NT <= 240;
	 * a doc can only beat thr if its byte exceeds floor(thr * qs) - 1) */
	const float qs = tsum > 0.0f ? (float)QSUM_MAX / tsum : 0.0f;
	auto thr_quant = [&](float th) -> int32_t {
		return __builtin_amdgcn_readfirstlane(th > 0.0f ? (int32_t)min(th * qs, 1.0e6f) - 1 : -1);
	};
	int32_t thr_q = thr_quant(thr);
	uint32_t tw = thr_q >= 0 ? (uint32_t)ST_W_HINTED : (uint32_t)ST_W0;

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
		/* what was pushed after the np: to the front afterwards (few: one or two stripes' worth) */
		const uint32_t rest = np + lane < n_pend ? s_pend[np + lane] : 0;
		if (n_pend - np > WAVE) {
			ovf = 1;
			SSTAT_ADD(14, 1);
		}
		uint32_t pd[PC], rk[PC];
#pragma unroll
		for (int c = 0; c < PC; c++) {
			const uint32_t ei = c * WAVE + lane;
			pd[c] = ei < np ? s_pend[ei] : 0;
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
			const uint64_t m = ballot64(valid && !dup);
			prev_last = (uint32_t)__builtin_amdgcn_readlane((int)d, 63);
			WAVE_SYNC();
			if (valid && !dup) {
				s_pend[nu + lanes_below(m)] = d;
			}
			nu += (uint32_t)__popcll(m);
			WAVE_SYNC();
		}
		nu = rfl32(nu);
		SSTAT_ADD(6, nu);

		for (uint32_t c0 = 0; c0 < nu; c0 += CPR * UNR) {
			uint32_t doc[UNR], sl[UNR], sh_[UNR], se[UNR];
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
				const bool act = inr[u] && m_on;
				sl[u] = act ? a : 0u;
				sh_[u] = se[u] = act ? b : 0u;
			}
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
				uint32_t v[UNR], mid[UNR];
#pragma unroll
				for (int u = 0; u < UNR; u++) {	/* (unconditional, clamped: the probes of all rounds in flight together) */
					mid[u] = sl[u] + ((sh_[u] - sl[u]) >> 1);
					v[u] = A.post[(uint64_t)(m_pb + mid[u])].doc;
				}
#pragma unroll
				for (int u = 0; u < UNR; u++) {
					if (sl[u] < sh_[u]) {
						if (v[u] < doc[u]) {
							sl[u] = mid[u] + 1;
						} else {
							sh_[u] = mid[u];
						}
					}
				}
			}
			posting_t fp[UNR];
#pragma unroll
			for (int u = 0; u < UNR; u++) {
				fp[u] = A.post[(uint64_t)(m_pb + sl[u])];
			}
#pragma unroll
			for (int u = 0; u < UNR; u++) {
				const bool hit = sl[u] < se[u] && fp[u].doc == doc[u];
				const float x = hit ? fp[u].imp : 0.0f;
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
		WAVE_SYNC();
		n_pend = min(n_pend - np, (uint32_t)WAVE);
		thr_q = thr_quant(thr);
		SSTAT_ADD(10, SSTAT_CLK() - fclk);
	};

	/*
	 * The loader: the stripe whose windows are being requested -- its slice [s, e) of every list (e is
	 * the stripe above's s), as the flat run [0, n_ld): lane i of the run belongs to the last term t
	 * with c[t] <= i and is posting i + dsel[t] of the posting array.
	 */
	uint32_t c[NT], dsel[NT], ddel[NT], n_ld = 0, i0_ld = 0;
	int32_t w_ld = w_top + 1, j_ld = -1;
	uint32_t rkv = fetch_dir(w_top), rk_next = fetch_dir(w_top - G);
	static_for<NT>([&](auto tc) {
		constexpr int t = decltype(tc)::value;
		c[t] = dsel[t] = ddel[t] = 0;
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
				rkv = rk_next;
				rk_next = fetch_dir(w_ld - G);
				j_ld = 0;
			}
			uint32_t n = 0;
			static_for<NT>([&](auto tc) {
				constexpr int t = decltype(tc)::value;
				const uint32_t raw = (uint32_t)__builtin_amdgcn_readlane((int)rkv, t * G + j_ld);
				const uint32_t et = rfl32(e[t]);
				const uint32_t s = min(max(raw, lo[t]), et);
				c[t] = n;
				dsel[t] = pb[t] + s - n;
				ddel[t] = t ? dsel[t] - dsel[t ? t - 1 : 0] : 0u;
				n += et - s;
				e[t] = s;
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
	/* window [i0, i0 + 64) of the loader's stripe (i0 < n_ld): clamped, unpredicated */
	auto load_window = [&](uint32_t i0) __attribute__((always_inline)) -> posting_t {
		const uint32_t ic = min(i0 + lane, n_ld - 1);
		/* (sums of selected differences: a chain of selects over dsel[] is turned into an indexed
		 * load from a stack array by the optimiser -- scratch memory) */
		uint32_t dv = ic + dsel[0];
		static_for<NT>([&](auto tc) {
			constexpr int t = decltype(tc)::value;
			if (t > 0) {
				dv += ic >= c[t] ? ddel[t] : 0u;
			}
		});
		return A.post[(uint64_t)dv];
	};
	auto push = [&](uint64_t cm, uint32_t doc) __attribute__((always_inline)) {
		const uint32_t np = (uint32_t)__popcll(cm);
		if (n_pend + np <= ST_PEND && lane_of(cm)) {
			s_pend[n_pend + lanes_below(cm)] = doc;
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

	/* ---- cold start: stripes in doc sub-ranges, until a sub-range is a whole stripe ---- */
	while (tw < (uint32_t)ST_DOCS && !ovf) {
		tw = rfl32(tw);
		ovf = rfl32(ovf);
		if (!next_stripe()) {
			break;
		}
		const uint32_t n = n_ld;
		uint32_t rhi = ST_DOCS;		/* docs of the stripe still to take: rel < rhi */
		while (rhi > 0 && !ovf) {
			rhi = rfl32(rhi);
			n_pend = rfl32(n_pend);
			if (ST_REDO && n_pend > ST_PEND - 64 && n_pend) {	/* (room for a sub-range's pushes: a burst is then the sub-range's own) */
				WAVE_SYNC();
				flush(n_pend);
			}
			tw = rfl32(tw);
			thr_q = (int32_t)rfl32((uint32_t)thr_q);
			const uint32_t rlo = rhi > tw ? rhi - tw : 0;
			const uint32_t n_before = n_pend;
			SSTAT_ADD(2, 1);
			SSTAT_ADD(3, (n + WAVE - 1) / WAVE);
			for (uint32_t i0 = 0; i0 < n; i0 += WAVE * ST_CH) {
				posting_t p[ST_CH];
#pragma unroll
				for (int k = 0; k < ST_CH; k++) {
					p[k].doc = 0;
					p[k].imp = 0.0f;
					if (i0 + k * WAVE < n) {	/* wave-uniform */
						p[k] = load_window(i0 + k * WAVE);
					}
				}
				uint32_t oldv[ST_CH], qq[ST_CH], shv[ST_CH];
				bool valid[ST_CH];
#pragma unroll
				for (int k = 0; k < ST_CH; k++) {
					valid[k] = false;
					if (i0 + k * WAVE < n) {
						const uint32_t rel = p[k].doc & (ST_DOCS - 1);
						valid[k] = i0 + k * WAVE + lane < n && rel - rlo < rhi - rlo;
						shv[k] = (rel & 3) * 8;
						/* floor + 2 >= the exact ceiling whatever the f32 product rounds to */
						qq[k] = (uint32_t)(p[k].imp * qs) + 2;
						oldv[k] = __hip_atomic_fetch_add(&s_map[rel >> 2], valid[k] ? (qq[k] << shv[k]) : 0u,
						    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
					}
				}
#pragma unroll
				for (int k = 0; k < ST_CH; k++) {
					if (i0 + k * WAVE < n) {
						const uint32_t sum = ((oldv[k] >> shv[k]) & 0xffu) + qq[k];
						const uint64_t cm = ballot64(valid[k] && (int32_t)sum > thr_q);
						if (cm) {
							push(cm, p[k].doc);
						}
					}
				}
			}
			if (ST_REDO && n_pend > ST_PEND && tw > (uint32_t)ST_W0) {
				/* a burst (the threshold is still weak for a sub-range this wide): the sub-range once
				 * more, a quarter as wide -- what it pushed is dropped, its bytes are wiped (the
				 * sub-ranges above it are done: their bytes are not needed any more) */
				n_pend = n_before;
				tw = max(tw / 4, (uint32_t)ST_W0);
				wipe();
				SSTAT_ADD(13, 1 << 16);
				continue;
			}
			rhi = rlo;
			const uint32_t n_tile = n_pend - n_before;
			if (n_tile <= 8) {
				tw = min(tw * 2, (uint32_t)ST_DOCS);
			} else if (n_tile > 48) {
				tw = max(tw / 2, (uint32_t)ST_W0);
			}
			if (n_pend > ST_PEND) {
				ovf = 1;
			} else if (n_pend >= 24 || (n_pend && thr_q < 0)) {
				WAVE_SYNC();
				flush(n_pend);
			}
		}
		i0_ld = n_ld;		/* the stripe is used up */
		wipe();
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
		/* the next window of the stream: its address, how many of its lanes count (0: nothing left),
		 * whether it ends its stripe */
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
#if ST_ADV_SCALAR
			uint32_t base = dsel[0];	/* (sums of differences, not selects over dsel[]: see load_window) */
			static_for<NT>([&](auto tc) {
				constexpr int t = decltype(tc)::value;
				if (t > 0) {
					base += c[t] <= i0 ? ddel[t] : 0u;
				}
			});
			uint32_t dv = min(i0 + lane, iend - 1) + base;
			static_for<NT>([&](auto tc) {
				constexpr int t = decltype(tc)::value;
				if (t > 0) {
					if (c[t] > i0 && c[t] < iend) {		/* wave-uniform */
						dv += i0 + lane >= c[t] ? ddel[t] : 0u;
					}
				}
			});
#else
			const uint32_t ic = min(i0 + lane, iend - 1);
			uint32_t dv = ic + dsel[0];
			static_for<NT>([&](auto tc) {
				constexpr int t = decltype(tc)::value;
				if (t > 0) {
					dv += ic >= c[t] ? ddel[t] : 0u;
				}
			});
#endif
			pa = A.post + dv;
			pc = iend - i0;
			pl = iend >= n_ld ? 1u : 0u;
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
					const uint32_t rel = pdoc & (ST_DOCS - 1);
					const uint32_t shv = (rel & 3) * 8;
					const uint32_t qq = (uint32_t)(pimp * qs) + 2;
					uint64_t cm;
					if (ST_PROC_FAST && cnt[s] == WAVE) {		/* a full window: no lane masks */
						const uint32_t oldv = __hip_atomic_fetch_add(&s_map[rel >> 2], qq << shv,
						    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
						const uint32_t sum = ((oldv >> shv) & 0xffu) + qq;
						cm = ballot64((int32_t)sum > thr_q);
					} else {
						const bool valid = lane < cnt[s];	/* (not a mask built by 1 << cnt: cnt may be 64) */
						const uint32_t oldv = __hip_atomic_fetch_add(&s_map[rel >> 2], valid ? (qq << shv) : 0u,
						    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
						const uint32_t sum = ((oldv >> shv) & 0xffu) + qq;
						cm = ballot64(valid && (int32_t)sum > thr_q);
					}
					if (cm) {
						push(cm, pdoc);
					}
					if (last[s]) {
						wipe();
						mark = min(n_pend, (uint32_t)ST_PEND);
					}
				}
				cnt[s] = ncnt;
				last[s] = nlast;
			});
			if (n_pend > ST_PEND) {
				ovf = 1;
			} else if (mark >= ST_FLUSH) {
				WAVE_SYNC();
				flush(mark);
				mark = 0;
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

/*
 * nxs_gpu_wide.hip -- k_scanw + nxsgpu_search_wide: queries beyond the fixed-size plan
 * (MI355X / gfx950 query path of nxsearch; see nxs_gpu_int.h for the map of the files)
 */
#include "nxs_gpu_int.h"
#include "nxs_gpu_dev.h"

/* ------------------------------------------------------------------ */
/* k_scanw: queries beyond the fixed-size plan (> 32 tokens, long or   */
/* deeply nested programs)                                             */
/* ------------------------------------------------------------------ */

/*
 * The reference puts no bound on the number of query terms
 * (run_query_logic, search.c:210-278, loops over a list).  Such queries are
 * rare; they take this generic kernel on the exact two-pass path (count, emit
 * all, global-memory heap replay): the same tile scheme as k_scan -- f32 sums
 * in token-list order, tiles visited from the highest doc down -- with a
 * presence BITSET of W words per doc instead of one mask word, and the
 * postfix program evaluated on a 128-deep bit stack (the nesting limit of 100,
 * search.c:70, bounds the stack at 101).
 */
#define	WTILE		512

struct wide_dev_t {
	uint32_t	nt, prog_len;
	uint64_t	tok_base;	/* into wtok: nt x (pbeg, pend) */
	uint64_t	prog_base;	/* into wprog */
};

struct wide_args_t {
	const posting_t *	post;
	const wide_dev_t *	wq;
	const uint64_t *	wtok;
	const uint16_t *	wprog;
	const qmeta_t *		qmeta;
	const item_t *		items;
	uint64_t		n_docs;
	uint32_t		W;		/* mask words per doc */
	uint32_t		nt_max, prog_max;
	uint32_t *		seg_count;
	const uint64_t *	seg_off;
	uint32_t *		cand_doc;
	float *			cand_sc;
};

__device__ static inline bool
eval_wide(const uint16_t *prog, uint32_t len, const uint32_t *mask)
{
	uint64_t lo = 0, hi = 0;	/* bit stack, top at bit 0 of lo */

	for (uint32_t i = 0; i < len; i++) {
		const uint32_t op = prog[i];
		if (op < 0x8000u || op == NXSGPU_WOP_EMPTY) {
			const uint64_t b = (op < 0x8000u) ? ((mask[op >> 5] >> (op & 31)) & 1u) : 0u;
			hi = (hi << 1) | (lo >> 63);
			lo = (lo << 1) | b;
		} else {
			const uint64_t b = lo & 1, a = (lo >> 1) & 1;
			uint64_t r;
			if (op == NXSGPU_WOP_AND) r = a & b;
			else if (op == NXSGPU_WOP_OR) r = a | b;
			else r = a & ~b & 1;
			lo = (lo >> 1) | (hi << 63);
			hi >>= 1;
			lo = (lo & ~1ull) | r;
		}
	}
	return lo & 1;
}

template <int MODE>
__global__ void __launch_bounds__(WAVE)
k_scanw(const wide_args_t A)
{
	extern __shared__ uint64_t smem_w[];
	const uint32_t W = A.W;
	uint64_t *s_hi = smem_w;
	uint64_t *s_lo = s_hi + A.nt_max;
	int64_t *s_pdoc = (int64_t *)(s_lo + A.nt_max);
	float *s_acc = (float *)(s_pdoc + A.nt_max);
	uint32_t *s_touch = (uint32_t *)(s_acc + WTILE);
	uint32_t *s_mask = s_touch + WTILE;
	uint16_t *s_prog = (uint16_t *)(s_mask + (size_t)WTILE * W);

	const unsigned lane = threadIdx.x;
	const item_t item = A.items[blockIdx.x];
	const uint32_t q = item.q, g = item.g;
	const qmeta_t qm = A.qmeta[q];
	const wide_dev_t Q = A.wq[q];
	const uint32_t nt = Q.nt;
	const posting_t *__restrict__ post = A.post;
	const uint64_t *tok = A.wtok + Q.tok_base;
	const uint64_t seg = (uint64_t)qm.seg_first + g;
	const uint64_t d_lo = min((uint64_t)g * qm.group_docs, A.n_docs);
	const uint64_t d_hi = (g + 1 == qm.n_groups) ? A.n_docs : min((uint64_t)(g + 1) * qm.group_docs, A.n_docs);

	for (uint32_t i = lane; i < WTILE; i += WAVE) {
		s_acc[i] = 0.0f;
		s_touch[i] = 0;
	}
	for (uint32_t i = lane; i < WTILE * W; i += WAVE) {
		s_mask[i] = 0;
	}
	for (uint32_t i = lane; i < Q.prog_len; i += WAVE) {
		s_prog[i] = A.wprog[Q.prog_base + i];
	}
	for (uint32_t t = lane; t < nt; t += WAVE) {
		const uint64_t pb = tok[2 * t], pe = tok[2 * t + 1];
		const uint64_t l = post_lower_bound(post, pb, pe, d_lo);
		const uint64_t h = (d_hi >= A.n_docs) ? pe : post_lower_bound(post, l, pe, d_hi);
		s_lo[t] = l;
		s_hi[t] = h;
		s_pdoc[t] = (h > l) ? (int64_t)post[h - 1].doc : -1;
	}
	__syncthreads();

	uint32_t n_out = 0;
	const uint64_t out_base = (MODE == MODE_ALL) ? A.seg_off[seg] : 0;

	for (;;) {
		int64_t md = -1;
		for (uint32_t t = lane; t < nt; t += WAVE) {
			md = max(md, s_pdoc[t]);
		}
		for (int o = 32; o; o >>= 1) {
			const int64_t other = ((int64_t)__shfl((int)(md >> 32), (int)(lane ^ o)) << 32) |
			    (uint32_t)__shfl((int)(uint32_t)md, (int)(lane ^ o));
			md = max(md, other);
		}
		if (md < 0) {
			break;
		}
		const uint32_t base = (uint32_t)((uint64_t)md / WTILE) * WTILE;

		/* tokens strictly in token-list order (results.c:134-136) */
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
				const uint32_t c = __popcll(ballot64(in));
				if (in) {
					const uint32_t d = p.doc - base;
					s_acc[d] += p.imp;
					s_mask[(size_t)d * W + (t >> 5)] |= 1u << (t & 31);
					s_touch[d] = 1;
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

		/* descending doc order (results.c:143-147) */
		for (int s = WTILE / WAVE - 1; s >= 0; s--) {
			const uint32_t d = s * WAVE + lane;
			const bool touched = s_touch[d] != 0;
			if (ballot64(touched) == 0) {
				continue;
			}
			float sc = 0.0f;
			bool match = false;
			if (touched) {
				sc = s_acc[d];
				match = eval_wide(s_prog, Q.prog_len, &s_mask[(size_t)d * W]);
				s_acc[d] = 0.0f;
				s_touch[d] = 0;
				for (uint32_t w = 0; w < W; w++) {
					s_mask[(size_t)d * W + w] = 0;
				}
			}
			const uint64_t bal = ballot64(match);
			if (MODE == MODE_ALL && match) {
				const uint64_t above = (lane == 63) ? 0 : (bal >> (lane + 1));
				const uint64_t o = out_base + n_out + __popcll(above);
				A.cand_doc[o] = base + d;
				A.cand_sc[o] = sc;
			}
			n_out += __popcll(bal);
		}
		__syncthreads();
	}
	if (MODE == MODE_COUNT && lane == 0) {
		A.seg_count[seg] = n_out;
	}
}

/* ------------------------------------------------------------------ */

extern "C" int
nxsgpu_search_wide(nxsgpu_index_t *ix, int algo, uint64_t limit, const nxsgpu_wide_query_t *queries,
    uint32_t nq, nxsgpu_results_t *res)
{
	const bool valid = (algo == NXSGPU_BM25) ? ix->bm25_valid : ix->tfidf_valid;
	std::vector<wide_dev_t> hq(nq);
	std::vector<uint64_t> wtok;
	std::vector<uint16_t> wprog;
	std::vector<qmeta_t> qmeta(nq);
	std::vector<item_t> items;
	uint32_t nt_max = 1, prog_max = 1, nseg = 0;
	void *ws = NULL, *ws2 = NULL;
	int rc = -1;

	memset(res, 0, sizeof(*res));
	res->n_queries = nq;
	if (algo != NXSGPU_BM25 && algo != NXSGPU_TF_IDF) {
		set_error("invalid algorithm");
		return -1;
	}
	if (limit == 0) {
		set_error("invalid limit");
		return -1;
	}
	if (nq == 0) {
		return 0;
	}
	if (hipSetDevice(ix->device) != hipSuccess) {
		set_error("hipSetDevice failed");
		return -1;
	}
	if (ensure_algo(ix, algo) != 0) {
		return -1;
	}
	const uint64_t tiles = std::max<uint64_t>(1, (ix->n_docs + WTILE - 1) / WTILE);
	for (uint32_t i = 0; i < nq; i++) {
		const nxsgpu_wide_query_t &q = queries[i];
		uint64_t work = 0;

		if (q.n_tokens > NXSGPU_WIDE_MAX_TOKENS || q.prog_len > 2 * NXSGPU_WIDE_MAX_TOKENS) {
			set_error("wide query %u exceeds %d tokens", i, NXSGPU_WIDE_MAX_TOKENS);
			return -1;
		}
		hq[i].nt = valid ? q.n_tokens : 0;	/* ranking.c:86-88,156-166 */
		hq[i].prog_len = q.prog_len;
		hq[i].tok_base = wtok.size();
		hq[i].prog_base = wprog.size();
		for (uint32_t t = 0; t < q.n_tokens; t++) {
			const uint32_t tid = q.term_id[t];
			if (tid == 0 || tid > ix->n_terms) {
				set_error("wide query %u: bad term id %u", i, tid);
				return -1;
			}
			wtok.push_back(ix->h_post_off[tid]);
			wtok.push_back(ix->h_post_off[tid + 1]);
			work += ix->h_post_off[tid + 1] - ix->h_post_off[tid];
		}
		for (uint32_t k = 0; k < q.prog_len; k++) {
			const uint16_t op = q.prog[k];
			if (op < 0x8000u && op >= q.n_tokens) {
				set_error("wide query %u: bad program", i);
				return -1;
			}
			wprog.push_back(op);
		}
		nt_max = std::max(nt_max, q.n_tokens);
		prog_max = std::max(prog_max, q.prog_len);
		/* one wavefront per ~64k postings */
		uint64_t g = std::max<uint64_t>(1, work / 65536);
		g = std::min<uint64_t>(std::min<uint64_t>(g, tiles), 4096);
		const uint64_t tiles_per = (tiles + g - 1) / g;
		g = (tiles + tiles_per - 1) / tiles_per;
		qmeta[i].seg_first = nseg;
		qmeta[i].n_groups = (uint32_t)g;
		qmeta[i].group_docs = (uint32_t)std::min<uint64_t>(tiles_per * WTILE, 0xffffffffu & ~(uint64_t)(WTILE - 1));
		qmeta[i].pad = 0;
		for (uint32_t gg = (uint32_t)g; gg-- > 0; ) {
			item_t it;
			it.q = i;
			it.g = gg;
			items.push_back(it);
		}
		nseg += (uint32_t)g;
	}
	if (wtok.empty()) wtok.push_back(0);
	if (wprog.empty()) wprog.push_back(0);
	const uint32_t W = (nt_max + 31) / 32;
	const size_t lds = (size_t)nt_max * 24 + (size_t)WTILE * 4 * (2 + W) + (size_t)prog_max * 2 + 16;
	if (lds > 160 * 1024 - 512) {
		set_error("wide query does not fit the LDS (%zu bytes)", lds);
		return -1;
	}

	std::vector<uint32_t> sc_cnt(nseg), x_cnt(nq, 0);
	std::vector<uint64_t> sc_off((size_t)nseg + 1, 0), hp_off((size_t)nq + 1, 0);
	std::vector<uint64_t> x_ids;
	std::vector<float> x_sc;
	do {
		const size_t need = 8192 + nq * sizeof(wide_dev_t) + wtok.size() * 8 + wprog.size() * 2
		    + nq * sizeof(qmeta_t) + nseg * sizeof(item_t) + nseg * 4;
		if (hipMalloc(&ws, need) != hipSuccess) {
			set_error("hipMalloc(%zu) failed", need);
			break;
		}
		uint8_t *p = (uint8_t *)ws;
		wide_dev_t *d_wq = carve<wide_dev_t>(p, nq);
		uint64_t *d_wtok = carve<uint64_t>(p, wtok.size());
		uint16_t *d_wprog = carve<uint16_t>(p, wprog.size());
		qmeta_t *d_qmeta = carve<qmeta_t>(p, nq);
		item_t *d_items = carve<item_t>(p, nseg);
		uint32_t *d_seg_count = carve<uint32_t>(p, nseg);
		if (hipMemcpyAsync(d_wq, hq.data(), nq * sizeof(wide_dev_t), hipMemcpyHostToDevice, ix->stream) != hipSuccess ||
		    hipMemcpyAsync(d_wtok, wtok.data(), wtok.size() * 8, hipMemcpyHostToDevice, ix->stream) != hipSuccess ||
		    hipMemcpyAsync(d_wprog, wprog.data(), wprog.size() * 2, hipMemcpyHostToDevice, ix->stream) != hipSuccess ||
		    hipMemcpyAsync(d_qmeta, qmeta.data(), nq * sizeof(qmeta_t), hipMemcpyHostToDevice, ix->stream) != hipSuccess ||
		    hipMemcpyAsync(d_items, items.data(), nseg * sizeof(item_t), hipMemcpyHostToDevice, ix->stream) != hipSuccess) {
			set_error("wide query upload failed");
			break;
		}
		wide_args_t wa;
		memset(&wa, 0, sizeof(wa));
		wa.post = ix->d_post[algo];
		wa.wq = d_wq;
		wa.wtok = d_wtok;
		wa.wprog = d_wprog;
		wa.qmeta = d_qmeta;
		wa.items = d_items;
		wa.n_docs = ix->n_docs;
		wa.W = W;
		wa.nt_max = nt_max;
		wa.prog_max = prog_max;
		wa.seg_count = d_seg_count;
		if (hipFuncSetAttribute((const void *)k_scanw<MODE_COUNT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
		    hipFuncSetAttribute((const void *)k_scanw<MODE_ALL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
			set_error("hipFuncSetAttribute(%zu bytes of LDS) failed", lds);
			break;
		}
		hipLaunchKernelGGL(k_scanw<MODE_COUNT>, dim3(nseg), dim3(WAVE), lds, ix->stream, wa);
		if (hipGetLastError() != hipSuccess ||
		    hipMemcpyAsync(sc_cnt.data(), d_seg_count, nseg * 4, hipMemcpyDeviceToHost, ix->stream) != hipSuccess ||
		    hipStreamSynchronize(ix->stream) != hipSuccess) {
			set_error("wide count pass failed: %s", hipGetErrorString(hipGetLastError()));
			break;
		}
		for (uint32_t s = 0; s < nseg; s++) {
			sc_off[s + 1] = sc_off[s] + sc_cnt[s];
		}
		for (uint32_t j = 0; j < nq; j++) {
			const uint64_t matched = sc_off[(size_t)qmeta[j].seg_first + qmeta[j].n_groups] - sc_off[qmeta[j].seg_first];
			hp_off[j + 1] = hp_off[j] + std::min<uint64_t>(limit, matched);
		}
		const uint64_t tot_c = sc_off[nseg], tot_o = hp_off[nq];
		const size_t need2 = 8192 + ((size_t)nseg + 1) * 8 + tot_c * 8 + tot_o * 8 + tot_o * 12 + ((size_t)nq + 1) * 8 + nq * 4;
		if (hipMalloc(&ws2, need2) != hipSuccess) {
			set_error("hipMalloc(%zu) failed", need2);
			break;
		}
		p = (uint8_t *)ws2;
		uint64_t *d_seg_off = carve<uint64_t>(p, (size_t)nseg + 1);
		uint32_t *d_cdoc = carve<uint32_t>(p, tot_c + 1);
		float *d_csc = carve<float>(p, tot_c + 1);
		float *d_hs = carve<float>(p, tot_o + 1);
		uint32_t *d_hd = carve<uint32_t>(p, tot_o + 1);
		uint64_t *d_hoff = carve<uint64_t>(p, (size_t)nq + 1);
		uint64_t *d_ids = carve<uint64_t>(p, tot_o + 1);
		float *d_sc = carve<float>(p, tot_o + 1);
		uint32_t *d_cnt = carve<uint32_t>(p, nq);
		if (hipMemcpyAsync(d_seg_off, sc_off.data(), ((size_t)nseg + 1) * 8, hipMemcpyHostToDevice, ix->stream) != hipSuccess ||
		    hipMemcpyAsync(d_hoff, hp_off.data(), ((size_t)nq + 1) * 8, hipMemcpyHostToDevice, ix->stream) != hipSuccess) {
			set_error("upload failed");
			break;
		}
		wa.seg_off = d_seg_off;
		wa.cand_doc = d_cdoc;
		wa.cand_sc = d_csc;
		hipLaunchKernelGGL(k_scanw<MODE_ALL>, dim3(nseg), dim3(WAVE), lds, ix->stream, wa);
		replay_args_t ra;
		memset(&ra, 0, sizeof(ra));
		ra.flags = ix->cfg.old_replay ? 1u : 0u;
		ra.qmeta = d_qmeta;
		ra.seg_cap = 0;
		ra.seg_off = d_seg_off;
		ra.cand_doc = d_cdoc;
		ra.cand_sc = d_csc;
		ra.doc_ids = ix->d_doc_ids;
		ra.k = (uint32_t)std::min<uint64_t>(limit, 0xffffffffu);
		ra.gheap_s = d_hs;
		ra.gheap_d = d_hd;
		ra.heap_off = d_hoff;
		ra.out_ids = d_ids;
		ra.out_sc = d_sc;
		ra.out_count = d_cnt;
		ra.out_off = d_hoff;
		if (ra.k <= REPLAY_LDS_K) {
			nxs_launch_replay(HEAP_LDS, nq, (size_t)ra.k * 8, ix->stream, ra);
		} else {
			nxs_launch_replay(HEAP_GLOBAL, nq, 0, ix->stream, ra);
		}
		x_ids.resize(tot_o);
		x_sc.resize(tot_o);
		if (hipGetLastError() != hipSuccess ||
		    hipMemcpyAsync(x_cnt.data(), d_cnt, nq * 4, hipMemcpyDeviceToHost, ix->stream) != hipSuccess ||
		    (tot_o && hipMemcpyAsync(x_ids.data(), d_ids, tot_o * 8, hipMemcpyDeviceToHost, ix->stream) != hipSuccess) ||
		    (tot_o && hipMemcpyAsync(x_sc.data(), d_sc, tot_o * 4, hipMemcpyDeviceToHost, ix->stream) != hipSuccess) ||
		    hipStreamSynchronize(ix->stream) != hipSuccess) {
			set_error("wide emit pass failed: %s", hipGetErrorString(hipGetLastError()));
			break;
		}
		rc = 0;
	} while (0);
	(void)hipFree(ws);
	(void)hipFree(ws2);
	if (rc != 0) {
		return -1;
	}
	res->counts = (uint32_t *)calloc(nq, sizeof(uint32_t));
	res->offsets = (uint64_t *)calloc((size_t)nq + 1, sizeof(uint64_t));
	for (uint32_t i = 0; i < nq; i++) {
		res->counts[i] = x_cnt[i];
		res->offsets[i + 1] = res->offsets[i] + x_cnt[i];
	}
	const uint64_t total = res->offsets[nq];
	res->doc_ids = (uint64_t *)malloc((total ? total : 1) * 8);
	res->scores = (float *)malloc((total ? total : 1) * 4);
	for (uint32_t i = 0; i < nq; i++) {
		memcpy(res->doc_ids + res->offsets[i], x_ids.data() + hp_off[i], (size_t)x_cnt[i] * 8);
		memcpy(res->scores + res->offsets[i], x_sc.data() + hp_off[i], (size_t)x_cnt[i] * 4);
	}
	res->exact_requeries = nq;
	return 0;
}

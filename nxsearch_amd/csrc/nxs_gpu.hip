/*
 * nxs_gpu.hip -- MI355X (gfx950, CDNA4) kernels of the nxsearch query path and
 * the C-ABI shim (include/nxs_gpu.h) the C11 host code calls.
 *
 * Kernels (all wave64, one independent wavefront per workgroup unless noted):
 *
 *  k_expand_pairs   index build: nxsdtmap image (big-endian forward index,
 *                   reference src/index/storage.h:67-97) -> (term, doc, tf)
 *                   triples; replaces dtmap_build_tdmap (dtmap.c:386-438).
 *  k_post_offsets   CSR row offsets of the (rocPRIM radix-)sorted triples.
 *  k_impacts        per-posting BM25 / TF-IDF scores in fp64 with the exact
 *                   operation order of src/algo/ranking.c:41-176; log() values
 *                   come from host libm tables so results are bit-identical.
 *  k_scan           THE hot kernel: posting-list iteration + f32 score
 *                   accumulation in token order in per-wavefront LDS tiles,
 *                   boolean filter, candidate pre-selection; replaces
 *                   run_query_logic + get_expr_bitmap (search.c:118-278) and
 *                   nxs_resp_addresult (results.c:128-150).
 *  k_scanm          pure-OR queries of sparse terms: a quantised score bound per
 *                   doc in LDS (integer DS atomics), exact f32 sums only for the
 *                   docs that can beat the threshold, from the register windows.
 *  k_replay         exact replay of the reference's capped min-heap + heapsort
 *                   (src/algo/heap.c:58-221; results.c:165-220) over the
 *                   candidates in descending doc order: bit-exact top-k order
 *                   including ties.
 *  k_bk_level       one BFS level of bktree_search (bktree.c:219-275), one
 *                   lane per (token, node): Levenshtein distance + child
 *                   range expansion; k_bk_* helpers around it.
 *
 * No MFMA anywhere: this is sparse gather/accumulate and byte/integer work,
 * bounded by HBM bandwidth and load latency.
 */
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdarg>
#include <vector>
#include <algorithm>
#include <thread>
#include <time.h>
#include <type_traits>

#include <dlfcn.h>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>		/* types only: the library is dlopen()ed (rccl_api) */
#include <rocprim/device/device_radix_sort.hpp>

#include "nxs_gpu.h"
#include "nxs_lev.h"

#define	WAVE		64
#ifndef TILE_W
#define	TILE_W		1024		/* docs per wavefront LDS tile (with the sparse OR queries on the mask
					 * path, 1024 beats 2048 by 3 % on C3 and is level elsewhere) */
#endif
#define	SEG_CAP_DEFAULT	1024		/* candidate slots per (query, group) */

/* ------------------------------------------------------------------ */
/* error handling                                                      */
/* ------------------------------------------------------------------ */

static thread_local char g_err[512];

static void
set_error(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}

#define	HIP_TRY(expr) do {						\
	hipError_t e_ = (expr);						\
	if (e_ != hipSuccess) {						\
		set_error("%s failed: %s (%s:%d)", #expr,		\
		    hipGetErrorString(e_), __FILE__, __LINE__);		\
		goto fail;						\
	}								\
} while (0)

extern "C" const char *
nxsgpu_last_error(void)
{
	return g_err;
}

extern "C" int
nxsgpu_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) {
		return 0;
	}
	return n;
}

/* ------------------------------------------------------------------ */
/* device-side data                                                    */
/* ------------------------------------------------------------------ */

struct posting_t {
	uint32_t	doc;	/* dense doc ordinal (rank in ascending doc id) */
	float		imp;	/* the reference's float score of this (term, doc) */
};

struct dev_query_t {
	uint32_t	nt;
	uint32_t	prog_len;
	uint64_t	pbeg[NXSGPU_MAX_TOKENS];
	uint64_t	pend[NXSGPU_MAX_TOKENS];
	uint32_t	truth[8];
	uint32_t	req;		/* tokens present in every matching mask */
	uint32_t	n_req;		/* k_scanr: slots [0, n_req) are the required tokens ... */
	uint8_t		slot_tok[8];	/* ... slot -> token, ascending list length within each group */
	uint32_t	drop_mask;	/* k_scanm<.., DROP>: dense tokens that leave the scan once the
					 * threshold exceeds what they can contribute together */
	uint32_t	drop_col[8];	/* ... and their impact columns (scan_args_t::dense_col) */
	float		tmax[8];	/* k_scanm: largest impact of tokens 0..7 */
	uint8_t		prog[NXSGPU_MAX_PROG];
};

/*
 * Tuning / A-B switches (DESIGN.md "Switches"), read from the environment ONCE
 * when the index is created (nxsgpu_index_reconfigure() re-reads them: tests
 * and tools/ab.sh only) and range-checked there; the query path never calls
 * getenv().
 */
struct gpu_cfg_t {
	uint64_t	wave_target;	/* NXS_GPU_WAVES */
	uint64_t	min_post;	/* NXS_GPU_MINPOST */
	uint64_t	min_post_solo;	/* NXS_GPU_MINPOST_SOLO: the same for a small batch with nothing else in flight */
	double		dense_thr;	/* NXS_GPU_DENSE (0 = posting-step path off) */
	double		scanm_dens;	/* NXS_GPU_SCANM_DENS */
	uint32_t	scanm_minnt, scanm_maxnt;
	uint32_t	rmin;		/* fewest tokens for k_scanr (NXS_GPU_NOSCANR2 => 3) */
	uint32_t	seg_cap;	/* NXS_GPU_SEGCAP */
	uint64_t	fuzzy_items;	/* NXS_GPU_FUZZY_ITEMS */
	bool		use_scanr, no_step, mask_off, by_level, use_scanm, scanm_general;
	bool		old_scan, no_scan1, no_req, one_replay, fuzzy_safe, fuzzy_noprune;
	bool		fuzzy_bfs;	/* NXS_GPU_FUZZY_BFS: level-by-level frontier search only */
	uint64_t	fuzzy_cand;	/* NXS_GPU_FUZZY_CAND: survivor queue of the match-first search (items) */
	bool		use_drop;	/* !NXS_GPU_NODROP: dense terms leave sparse OR scans (k_scanm<.., DROP>) */
	uint64_t	drop_minpost;	/* NXS_GPU_DROP_MINPOST: fewest sparse postings for that path */
	uint64_t	drop_workmul;	/* NXS_GPU_DROP_WORKMUL: range count multiplier of that class */
	bool		drop_prio, drop_side;	/* !NXS_GPU_DROP_NOPRIO / !NXS_GPU_DROP_NOSIDE */
	bool		debug_timing;	/* NXS_GPU_DEBUG_TIMING: per-batch host phases of _begin to stderr */
	bool		down_inline;	/* NXS_GPU_DOWN_INLINE: sharded runs also keep everything on one stream */
};

static void
cfg_from_env(gpu_cfg_t &c)
{
	auto u64 = [](const char *name, uint64_t dflt, uint64_t lo, uint64_t hi) -> uint64_t {
		const char *e = getenv(name);
		if (!e || !*e) {
			return dflt;
		}
		const uint64_t v = strtoull(e, NULL, 10);
		return v < lo ? lo : v > hi ? hi : v;
	};
	auto dbl = [](const char *name, double dflt) -> double {
		const char *e = getenv(name);
		return (e && *e) ? atof(e) : dflt;
	};
	auto on = [](const char *name) -> bool { return getenv(name) != NULL; };

	c.wave_target = u64("NXS_GPU_WAVES", 65536, 1, 1u << 22);
	c.min_post = u64("NXS_GPU_MINPOST", 4096, 1, ~0ull);
	c.min_post_solo = u64("NXS_GPU_MINPOST_SOLO", 512, 1, ~0ull);
	c.dense_thr = dbl("NXS_GPU_DENSE", 0.0);
	c.scanm_dens = dbl("NXS_GPU_SCANM_DENS", 0.08);
	c.scanm_minnt = (uint32_t)u64("NXS_GPU_SCANM_MINNT", 2, 2, 8);
	c.scanm_maxnt = (uint32_t)u64("NXS_GPU_SCANM_MAXNT", 8, 2, 8);
	c.rmin = on("NXS_GPU_NOSCANR2") ? 3u : 2u;
	c.seg_cap = (uint32_t)u64("NXS_GPU_SEGCAP", SEG_CAP_DEFAULT, 1, 1u << 16);
	c.fuzzy_items = u64("NXS_GPU_FUZZY_ITEMS", 256ull << 20, 1, 1ull << 32);
	c.use_scanr = !on("NXS_GPU_NOSCANR");
	c.no_step = on("NXS_GPU_NOSTEP");
	c.mask_off = !on("NXS_GPU_NOMASKOFF");
	c.by_level = !on("NXS_GPU_NOLEVELS");
	c.use_scanm = !on("NXS_GPU_NOSCANM");
	c.scanm_general = !on("NXS_GPU_SCANM_ORONLY");
	c.old_scan = on("NXS_GPU_OLDSCAN");
	c.no_scan1 = on("NXS_GPU_NOSCAN1");
	c.no_req = on("NXS_GPU_NOREQ");
	c.one_replay = on("NXS_GPU_ONEREPLAY");
	c.fuzzy_safe = on("NXS_GPU_FUZZY_SAFE");
	c.fuzzy_noprune = on("NXS_GPU_FUZZY_NOPRUNE");
	c.fuzzy_bfs = on("NXS_GPU_FUZZY_BFS");
	c.fuzzy_cand = u64("NXS_GPU_FUZZY_CAND", 32ull << 20, 1024, 1ull << 30);
	c.use_drop = !on("NXS_GPU_NODROP");
	c.drop_minpost = u64("NXS_GPU_DROP_MINPOST", 4096, 1, ~0ull);
	c.drop_workmul = u64("NXS_GPU_DROP_WORKMUL", 2, 1, 64);
	c.drop_prio = !on("NXS_GPU_DROP_NOPRIO");
	c.drop_side = !on("NXS_GPU_DROP_NOSIDE");
	c.debug_timing = on("NXS_GPU_DEBUG_TIMING");
	c.down_inline = on("NXS_GPU_DOWN_INLINE");
}

struct worklist_t;

struct nxsgpu_index {
	int		device;
	gpu_cfg_t	cfg;
	hipStream_t	stream;
	/* exact two-pass path: its device buffers are kept between calls (grow-only,
	 * up to X_KEEP_MAX: hipMalloc + hipFree of a few hundred MB cost milliseconds) */
	void *		xbuf[2];
	size_t		xbuf_len[2];
	hipStream_t	xstream[3];	/* the blocking search (nxsgpu_search: re-runs of overflowed queries,
					 * limits > 64) takes these in place of stream / stream2 / stream3
					 * while batches are in flight: beside them, not queued behind
					 * their scans */
	hipStream_t	stream2;	/* heap replay of a finished query class, beside the next class's scan */
	hipStream_t	stream3;	/* the sparse + dense OR class (k_scanm<.., DROP>): few, latency-bound
					 * wavefronts that run BESIDE the other classes, not in front of them */
	hipEvent_t	ev_cls, ev_join, ev_fork3, ev_join3;
	/* nxsgpu_search_dev_begin/_end: two batches in flight, each with its own
	 * device workspace and pinned staging; plans go up on their own stream */
	hipStream_t	stream_up;
	hipStream_t	stream_down;	/* record blocks: all-gather (sharded) + copy to pinned memory */
	hipStream_t	stream_fz;	/* BK-tree searches: beside the batches in flight, not behind them */
	struct nxsgpu_comm *comm;	/* attached communicator (query sharding) or NULL */
	struct dev_slot_t {
		void *		ws;
		size_t		ws_len;
		uint8_t *	h_stage;	/* pinned: uploads, then the overflow flags coming back */
		size_t		h_stage_len;
		hipEvent_t	ev_up, ev_done, ev_res, ev_t[3];
		bool		active;
		bool		records;	/* nxsgpu_batch_begin: results as record blocks */
		uint32_t	nq;
		uint64_t	postings;
		uint64_t	seq;
		/* record mode */
		uint8_t *	d_blocks;	/* device: world blocks (own block first when world == 1) */
		size_t		d_blocks_len;
		uint8_t *	h_blocks;	/* pinned: world blocks */
		uint8_t *	h_blocks_dev;	/* the same memory as the device sees it (zero-copy results) */
		size_t		h_blocks_len;
		uint32_t	n_slots, k, world;
		size_t		rec_bytes, block_bytes;
		uint32_t *	h_ovf;		/* overflow flags coming back (inside h_stage) */
		worklist_t *	wl;		/* the slot's work list: its vectors keep their capacity
						 * (several MB a batch: no mmap / page-fault churn) */
	}		slot[2];
	uint64_t	slot_seq;

	uint64_t	n_docs, n_post;
	uint32_t	n_terms;
	uint32_t	hdr_doc_count;
	uint64_t	hdr_token_count;
	uint64_t	first_bad;
	bool		bm25_valid, tfidf_valid;

	uint64_t *	d_doc_ids;	/* [D] */
	uint32_t *	d_doc_len;	/* [D] */
	uint64_t *	d_post_off;	/* [T+2] */
	uint64_t *	d_post_dt;	/* [P] doc<<32 | tf, sorted by (term, doc): the primary array;
					 * impacts are recomputed from it at every refresh (N1) */
	uint64_t	cap_post;	/* capacity of d_post[*] */
	uint64_t	cap_docs_ids, cap_docs_len;
	uint32_t	max_tf;
	posting_t *	d_post[2];	/* [P] per ranking algo */
	std::vector<uint64_t> h_post_off;
	std::vector<float> h_maximp[2];	/* [T+2] largest impact per term and ranking algo */
	std::vector<uint32_t> df_global;	/* [T+2] doc-sharded mode: collection-wide df, else empty */
	/*
	 * Dense terms (lists holding more than cfg.scanm_dens of the docs: a few
	 * dozen at most) also get a direct-access impact COLUMN per ranking
	 * function, [n_docs] f32: what a candidate needs from a dense list once
	 * k_scanm<.., DROP> no longer streams it is then one load, not a search.
	 */
	std::vector<uint32_t> dense_terms;	/* ascending term ids; column = position */
	uint32_t *	d_dense_col[2];
	uint64_t	dense_cap;		/* allocated words per algo */

	nxsgpu_bknode_t *d_bk;
	uint8_t *	d_bk_bytes;
	uint32_t	n_bk, bk_depth;
	/* match-first fuzzy search (k_fz_filter ...): per node the byte-set signature
	 * and length of its term, its parent and the slot it hangs in (k_bk_aux) */
	bool		fz_split;	/* nxsgpu_fuzzy is working on one half of a batch it split */
	uint32_t *	d_bk_parent;	/* [n_bk] */
	uint8_t *	d_bk_slot;	/* [n_bk] */
	uint32_t *	d_fz_node;	/* [n_fz] the nodes that can win, sorted by term length */
	uint32_t *	d_fz_sig;	/* [n_fz] byte-set signature */
	uint8_t *	d_fz_len;	/* [n_fz] */
	uint32_t	n_fz;

	/* reusable query workspaces */
	void *		ws;
	size_t		ws_len;
	void *		h_pin;
	size_t		h_pin_len;

	/* fuzzy workspaces */
	void *		fz;
	size_t		fz_len;

	bool		profiling;
	hipEvent_t	ev[4];
	nxsgpu_profile_t prof;
};

static inline uint32_t __device__ __host__
bswap32(uint32_t v)
{
	return (v >> 24) | ((v >> 8) & 0xff00) | ((v << 8) & 0xff0000) | (v << 24);
}

/* ------------------------------------------------------------------ */
/* index build kernels                                                 */
/* ------------------------------------------------------------------ */

/*
 * One wavefront walks 16 doc blocks; lanes stride over the (term_id, count)
 * pairs of a block (8 bytes each => coalesced).  Block layout:
 * u64 doc_id | u32 doc_len | u32 n | n x (u32 term_id, u32 count), all BE.
 */
__global__ void
k_expand_pairs(const uint8_t *__restrict__ img, const uint64_t *__restrict__ blk_off,
    const uint64_t *__restrict__ pair_base, uint64_t n_docs, uint32_t n_terms,
    const uint8_t *__restrict__ term_ok, uint32_t *__restrict__ keys,
    uint64_t *__restrict__ vals, uint32_t *__restrict__ doc_len,
    unsigned long long *__restrict__ first_bad, unsigned int *__restrict__ max_tf)
{
	const unsigned lane = threadIdx.x & 63;
	const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	uint32_t my_max = 0;

	for (unsigned k = 0; k < 16; k++) {
		const uint64_t ord = wave * 16 + k;
		if (ord >= n_docs) {
			break;
		}
		const uint64_t off = blk_off[ord];
		const uint32_t *blk = (const uint32_t *)(img + off);
		const uint32_t n = (uint32_t)(pair_base[ord + 1] - pair_base[ord]);
		const uint64_t base = pair_base[ord];

		if (lane == 0) {
			doc_len[ord] = bswap32(blk[2]);
		}
		for (uint32_t j = lane; j < n; j += WAVE) {
			const uint2 p = *(const uint2 *)(blk + 4 + 2 * (size_t)j);
			const uint32_t tid = bswap32(p.x), cnt = bswap32(p.y);
			bool ok = tid != 0 && tid <= n_terms;
			if (ok) {
				ok = term_ok[tid] != 0;
			}
			if (!ok) {
				atomicMin(first_bad, (unsigned long long)off);
			}
			keys[base + j] = ok ? tid : 0;
			vals[base + j] = (ord << 32) | cnt;
			my_max = max(my_max, cnt);
		}
	}
	for (int o = 32; o; o >>= 1) {
		my_max = max(my_max, (uint32_t)__shfl_xor((int)my_max, o));
	}
	if (lane == 0 && my_max) {
		atomicMax(max_tf, my_max);
	}
}

/* post_off[t] = first index i with keys[i] >= t, t in [0, n_terms+1] */
__global__ void
k_post_offsets(const uint32_t *__restrict__ keys, uint64_t n, uint32_t n_terms,
    uint64_t *__restrict__ post_off)
{
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t > (uint64_t)n_terms + 1) {
		return;
	}
	uint64_t lo = 0, hi = n;
	while (lo < hi) {
		const uint64_t mid = lo + ((hi - lo) >> 1);
		if (keys[mid] < t) lo = mid + 1; else hi = mid;
	}
	post_off[t] = lo;
}

/*
 * Per-posting scores.  fp64, -ffp-contract=off, operation order exactly as
 * written in the reference:
 *   bm25 (ranking.c:163-175):
 *	tf = log(term_freq + 1)                       [host libm table]
 *	tf_bm25 = tf / (tf + k * (1 - b + b * dl / adl))
 *	idf = log((N - df + 0.5) / (df + 0.5) + 1)    [host libm, per term]
 *	return (float)(tf_bm25 * idf)
 *   tf_idf (ranking.c:90-96):
 *	tf = (float)log(term_freq + 1); idf = (float)(log((float)N / df) + 1)
 *	return tf * idf                               [f32 multiply]
 */
__global__ void
k_impacts(const uint32_t *__restrict__ keys, const uint64_t *__restrict__ vals,
    uint64_t n, const uint32_t *__restrict__ doc_len,
    const double *__restrict__ logtf, const double *__restrict__ idf_bm25,
    const float *__restrict__ idf_tfidf, double adl, double kk, double bb,
    posting_t *__restrict__ out_bm25, posting_t *__restrict__ out_tfidf,
    uint32_t *__restrict__ max_bm25, uint32_t *__restrict__ max_tfidf)
{
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		const uint32_t t = keys[i];
		const uint64_t v = vals[i];
		const uint32_t doc = (uint32_t)(v >> 32), cnt = (uint32_t)v;
		const double tf = logtf[cnt];
		const double dl = (double)(int)doc_len[doc];
		const double one_b = 1 - bb;
		const double tf_bm25 = tf / (tf + kk * (one_b + bb * dl / adl));
		posting_t pb, pt;

		pb.doc = doc;
		pb.imp = (float)(tf_bm25 * idf_bm25[t]);
		pt.doc = doc;
		pt.imp = (float)tf * idf_tfidf[t];
		out_bm25[i] = pb;
		out_tfidf[i] = pt;
		/*
		 * Largest impact of the term (k_scanm's score bounds).  Impacts are
		 * >= +0, where unsigned order of the bit pattern is float order.  The
		 * plain read only saves atomics (a stale, smaller value just means one
		 * atomic more): dense terms converge after a few wavefronts.
		 */
		uint32_t bb_ = pb.imp > 0.0f ? __float_as_uint(pb.imp) : 0u;
		uint32_t bt_ = pt.imp > 0.0f ? __float_as_uint(pt.imp) : 0u;
		/* consecutive postings mostly belong to one term: reduce over the
		 * wavefront first when they all do (the loop bound is wave-uniform
		 * except in the last round, where the wavefront may be partial) */
		const bool whole = (i - (threadIdx.x & 63)) + 63 < n;
		if (whole && __builtin_amdgcn_ballot_w64(t != (uint32_t)__builtin_amdgcn_readfirstlane((int)t)) == 0) {
			for (int o = 32; o; o >>= 1) {
				bb_ = max(bb_, (uint32_t)__shfl_xor((int)bb_, o));
				bt_ = max(bt_, (uint32_t)__shfl_xor((int)bt_, o));
			}
			if ((threadIdx.x & 63) != 0) {
				bb_ = bt_ = 0;
			}
		}
		if (bb_ > max_bm25[t]) {
			atomicMax(&max_bm25[t], bb_);
		}
		if (bt_ > max_tfidf[t]) {
			atomicMax(&max_tfidf[t], bt_);
		}
	}
}

/* impact column of one dense term: col[doc] = impact bits of its postings */
__global__ void
k_dense_fill(const posting_t *__restrict__ post, uint64_t beg, uint64_t end, uint32_t *__restrict__ col)
{
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t i = beg + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < end; i += stride) {
		const posting_t p = post[i];
		col[p.doc] = __float_as_uint(p.imp);
	}
}

__global__ void
k_shift_ords(uint64_t *__restrict__ vals, uint64_t n, uint64_t first_ord)
{
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		vals[i] += first_ord << 32;
	}
}

/*
 * CSR walk shared by the refresh kernels: a wavefront takes SPAN consecutive
 * postings; the term of the first one is found by ONE binary search in the
 * row offsets (all lanes search the same key), then every lane advances its
 * own term index while its posting lies beyond the row's end -- rows are
 * ascending, so a lane's term only ever grows.
 */
#define	CSR_SPAN	(WAVE * 8)

__device__ static inline uint32_t
csr_row_of(const uint64_t *__restrict__ off, uint32_t n_rows, uint64_t i)
{
	/* largest t in [0, n_rows] with off[t] <= i */
	uint32_t lo = 0, hi = n_rows + 1;
	while (lo + 1 < hi) {
		const uint32_t mid = lo + ((hi - lo) >> 1);
		if (off[mid] <= i) lo = mid; else hi = mid;
	}
	return lo;
}

/*
 * Per-posting scores straight from the CSR form (term, doc, tf): the refreshable
 * twin of k_impacts (same arithmetic, same operation order; the term comes from
 * the row offsets instead of a key array).  off has n_terms + 2 entries.
 */
__global__ void
k_impacts_csr(const uint64_t *__restrict__ off, uint32_t n_terms, const uint64_t *__restrict__ vals,
    uint64_t n, const uint32_t *__restrict__ doc_len,
    const double *__restrict__ logtf, const double *__restrict__ idf_bm25,
    const float *__restrict__ idf_tfidf, double adl, double kk, double bb,
    posting_t *__restrict__ out_bm25, posting_t *__restrict__ out_tfidf,
    uint32_t *__restrict__ max_bm25, uint32_t *__restrict__ max_tfidf)
{
	const unsigned lane = threadIdx.x & 63;
	const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
	const uint64_t wave0 = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);

	for (uint64_t base = wave0 * CSR_SPAN; base < n; base += n_waves * CSR_SPAN) {
		uint32_t t = csr_row_of(off, n_terms, base);
		for (unsigned k = 0; k < CSR_SPAN / WAVE; k++) {
			const uint64_t i = base + k * WAVE + lane;
			const bool valid = i < n;
			posting_t pb, pt;
			uint32_t bb_ = 0, bt_ = 0;

			if (valid) {
				while (i >= off[t + 1]) {
					t++;
				}
				const uint64_t v = vals[i];
				const uint32_t doc = (uint32_t)(v >> 32), cnt = (uint32_t)v;
				const double tf = logtf[cnt];
				const double dl = (double)(int)doc_len[doc];
				const double one_b = 1 - bb;
				const double tf_bm25 = tf / (tf + kk * (one_b + bb * dl / adl));

				pb.doc = doc;
				pb.imp = (float)(tf_bm25 * idf_bm25[t]);
				pt.doc = doc;
				pt.imp = (float)tf * idf_tfidf[t];
				out_bm25[i] = pb;
				out_tfidf[i] = pt;
				bb_ = pb.imp > 0.0f ? __float_as_uint(pb.imp) : 0u;
				bt_ = pt.imp > 0.0f ? __float_as_uint(pt.imp) : 0u;
			}
			/* largest impact per term (k_scanm's bounds): one atomic per
			 * wavefront when all its postings belong to one term */
			const uint32_t t0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
			const bool whole = base + k * WAVE + 63 < n;
			if (whole && __builtin_amdgcn_ballot_w64(t != t0) == 0) {
				for (int o = 32; o; o >>= 1) {
					bb_ = max(bb_, (uint32_t)__shfl_xor((int)bb_, o));
					bt_ = max(bt_, (uint32_t)__shfl_xor((int)bt_, o));
				}
				if (lane != 0) {
					bb_ = bt_ = 0;
				}
			}
			if (valid) {
				if (bb_ > max_bm25[t]) {
					atomicMax(&max_bm25[t], bb_);
				}
				if (bt_ > max_tfidf[t]) {
					atomicMax(&max_tfidf[t], bt_);
				}
			}
		}
	}
}

/*
 * Incremental refresh, step 1: the surviving postings of the old CSR move to
 * their places in the new one.  Posting i of term t goes to
 *	i - (dead postings before i) + (new postings of terms < t)
 * (appended docs have the highest ordinals, so a term's new postings follow
 * its old ones).  dead_pos = ascending positions of the postings of removed
 * docs; new_off = row offsets of the sorted new postings.
 */
__global__ void
k_merge_old(const uint64_t *__restrict__ off, uint32_t n_terms_old, const uint64_t *__restrict__ vals,
    uint64_t n, const uint64_t *__restrict__ dead_pos, uint32_t n_dead,
    const uint64_t *__restrict__ new_off, uint64_t *__restrict__ out)
{
	const unsigned lane = threadIdx.x & 63;
	const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
	const uint64_t wave0 = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);

	for (uint64_t base = wave0 * CSR_SPAN; base < n; base += n_waves * CSR_SPAN) {
		uint32_t t = csr_row_of(off, n_terms_old, base);
		for (unsigned k = 0; k < CSR_SPAN / WAVE; k++) {
			const uint64_t i = base + k * WAVE + lane;
			if (i >= n) {
				continue;
			}
			while (i >= off[t + 1]) {
				t++;
			}
			/* dead postings at positions < i, and is i itself one? */
			uint32_t lo = 0, hi = n_dead;
			while (lo < hi) {
				const uint32_t mid = lo + ((hi - lo) >> 1);
				if (dead_pos[mid] < i) lo = mid + 1; else hi = mid;
			}
			if (lo < n_dead && dead_pos[lo] == i) {
				continue;
			}
			out[i - lo + new_off[t]] = vals[i];
		}
	}
}

/* step 2: the new postings (sorted by term, doc ascending inside a term) go to
 * the tail of their term's new row: row t ends at new_row_off[t + 1] */
__global__ void
k_place_new(const uint32_t *__restrict__ keys, const uint64_t *__restrict__ vals, uint64_t n_new,
    const uint64_t *__restrict__ new_off, const uint64_t *__restrict__ row_off_new,
    uint64_t *__restrict__ out)
{
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_new; j += stride) {
		const uint32_t t = keys[j];
		out[row_off_new[t + 1] - (new_off[t + 1] - j)] = vals[j];
	}
}

/* position of every (term, doc ordinal) of a removed doc in the old CSR
 * (~0 if the posting is not there: cannot happen on a consistent index) */
__global__ void
k_dead_positions(const uint64_t *__restrict__ off, const uint64_t *__restrict__ vals,
    const uint32_t *__restrict__ dead_term, const uint32_t *__restrict__ dead_ord, uint32_t n_dead,
    uint64_t *__restrict__ pos)
{
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n_dead) {
		return;
	}
	const uint32_t t = dead_term[j], ord = dead_ord[j];
	uint64_t lo = off[t], hi = off[t + 1];
	const uint64_t end = hi;
	while (lo < hi) {
		const uint64_t mid = lo + ((hi - lo) >> 1);
		if ((uint32_t)(vals[mid] >> 32) < ord) lo = mid + 1; else hi = mid;
	}
	pos[j] = (lo < end && (uint32_t)(vals[lo] >> 32) == ord) ? lo : ~0ull;
}

/* new row offsets: old row start minus the dead postings before it plus the
 * new postings of lower terms; rows of new terms start at the old end */
__global__ void
k_new_row_offsets(const uint64_t *__restrict__ off_old, uint32_t n_terms_old, uint64_t n_old,
    const uint64_t *__restrict__ dead_pos, uint32_t n_dead, const uint64_t *__restrict__ new_off,
    uint32_t n_terms_new, uint64_t *__restrict__ off_out)
{
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t > (uint64_t)n_terms_new + 1) {
		return;
	}
	const uint64_t o = t <= (uint64_t)n_terms_old + 1 ? off_old[t] : n_old;
	uint32_t lo = 0, hi = n_dead;
	while (lo < hi) {
		const uint32_t mid = lo + ((hi - lo) >> 1);
		if (dead_pos[mid] < o) lo = mid + 1; else hi = mid;
	}
	off_out[t] = o - lo + new_off[t];
}

/* ------------------------------------------------------------------ */
/* k_scan: posting iteration + LDS score accumulation + pre-selection   */
/* ------------------------------------------------------------------ */

enum { MODE_TOPK = 0, MODE_COUNT = 1, MODE_ALL = 2 };

/* a query's doc space is cut into n_groups ranges of group_docs docs; one
 * wavefront (work item) per range; its candidates go to segment seg_first+g */
/* pad: 1 = the ranges of this (single-token) query split its posting list by
 * INDEX, evenly -- no doc boundaries, no cursors (k_scan1) */
struct qmeta_t { uint32_t seg_first, n_groups, group_docs, pad; };
struct item_t { uint32_t q, g; };

struct scan_args_t {
	const posting_t *	post;
	const dev_query_t *	queries;
	uint64_t		n_docs;
	const qmeta_t *		qmeta;		/* [Q] */
	const item_t *		items;		/* (query, group) work items */
	uint32_t		item_base;	/* first item of this launch */
	const uint32_t *	cursors;	/* [(segments + Q)][MAX_TOKENS]: list position of each range boundary */
	uint32_t		k;		/* limit (<= 64 in MODE_TOPK) */
	uint32_t		seg_cap;
	uint32_t *		seg_count;	/* [segments] */
	const uint64_t *	seg_off;	/* [segments+1] (MODE_ALL) */
	uint32_t *		cand_doc;
	float *			cand_sc;
	uint32_t *		overflow;	/* [Q] */
	float *			pub;		/* [segments] k-th best score of a finished range (0 = none) */
	uint32_t		flags;		/* bit 0: raise the wavefronts' issue priority (side-stream class) */
	uint32_t *		cold_state;	/* [segments][16]: what k_cold hands to k_scanm<.., DROP> */
	float *			cold_top;	/* [segments][64]: its running top-k scores */
	const uint32_t *	dense_col;	/* impact columns of the dense terms: [col][n_docs] f32 bits,
						 * 0xffffffff = the doc does not hold the term */
	uint64_t		dense_stride;
};

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
 * Threshold hand-down between the wavefronts of one query.  A wavefront that
 * has finished its doc range publishes the k-th largest score it met (if it
 * met k).  The reference's heap is fed in descending doc id, so while range g
 * is being fed the heap root is at least the k-th largest score of ANY
 * higher range alone: a published value of a higher range is a valid
 * candidate threshold for range g from its very first doc.  Values only ever
 * make the filter tighter and a stale read (0 = nothing published yet, or an
 * older value in a non-coherent L2) is merely less tight -- correctness never
 * depends on visibility, so plain agent-scope relaxed accesses are enough.
 */
__device__ static inline float
range_hint(const scan_args_t &A, const qmeta_t &qm, uint32_t g)
{
	const unsigned lane = threadIdx.x & 63;
	float h = 0.0f;

	for (uint32_t g2 = g + 1 + lane; g2 < qm.n_groups; g2 += WAVE) {
		h = fmaxf(h, __hip_atomic_load(&A.pub[(uint64_t)qm.seg_first + g2],
		    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
	}
	for (int o = 32; o; o >>= 1) {
		h = fmaxf(h, __shfl_xor(h, o));
	}
	return h;
}

__device__ static inline void
range_publish(const scan_args_t &A, uint64_t seg, float kth)
{
	if ((threadIdx.x & 63) == 0 && kth > 0.0f) {
		__hip_atomic_store(&A.pub[seg], kth, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}

/* lower bound of `doc` in post[lo, hi) by doc ordinal */
__device__ static inline uint64_t
post_lower_bound(const posting_t *__restrict__ post, uint64_t lo, uint64_t hi, uint64_t doc)
{
	while (lo < hi) {
		const uint64_t mid = lo + ((hi - lo) >> 1);
		if (post[mid].doc < doc) lo = mid + 1; else hi = mid;
	}
	return lo;
}

/* compile-time loop: f(std::integral_constant<int, 0>{}) ... f(<N-1>) */
template <int I, int N, typename F>
__device__ __forceinline__ void
static_for_impl(F &&f)
{
	if constexpr (I < N) {
		f(std::integral_constant<int, I>{});
		static_for_impl<I + 1, N>(f);
	}
}
/*
 * Wave-level predicates without the int round trip of HIP's __ballot(): a lane
 * condition becomes a 64-bit scalar mask (the v_cmp result itself), a scalar
 * mask becomes a lane condition again (it is used as the select mask), and a
 * lane's rank inside a mask is the two v_mbcnt instructions.
 */
static __device__ __forceinline__ uint64_t
ballot64(bool p)
{
	return __builtin_amdgcn_ballot_w64(p);
}

static __device__ __forceinline__ bool
lane_of(uint64_t wave_uniform_mask)
{
	return __builtin_amdgcn_inverse_ballot_w64(wave_uniform_mask);
}

static __device__ __forceinline__ uint32_t
lanes_below(uint64_t m)
{
	return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
}

template <int N, typename F>
__device__ __forceinline__ void
static_for(F &&f)
{
	static_for_impl<0, N>(f);
}

/*
 * The prefetched posting window of term slot T ("set B" of the one-window scan
 * paths) lives in two accumulation registers that only these asm blocks name.
 *
 * Why not ordinary variables: a load the compiler tracks is waited for as
 * soon as its value is copied, and taking B over into A at a rotation is such
 * a copy -- the wavefront stalled for a memory latency every 64 postings.  A
 * load it does not track (inline asm into a C variable) is not safe either:
 * the register allocator is free to move that variable with v_mov while the
 * data is still in flight.  The kernels use no AGPRs otherwise, so these
 * registers are out of the compiler's reach: nothing can be scheduled into, copied out
 * of, or reallocated over a pending prefetch (bpair_request / bpair_take below).
 */

/*
 * Prefetch RING of the tile path: R windows per term in flight instead of one.
 * A dense term drains a 64-posting window in one visit (a few hundred cycles)
 * while a window takes a memory latency (1-2 us under load) to arrive, so with
 * one window in flight every rotation of a dense term exposed that latency.
 * Term slot T owns the AGPR pairs [T*R, T*R + R); pair p holds the window that
 * is p-th to be consumed (mod R, `ring position`).  Taking the oldest window:
 *
 *     s_waitcnt vmcnt(R - 1)
 *
 * is exact enough and needs no bookkeeping: vector memory operations retire in
 * issue order, the R - 1 other pairs of this term were requested after the
 * oldest one (every take re-requests the pair it has just read, load_ring()
 * requests all R in order, addresses are clamped instead of predicated so the
 * count never varies), hence at most R - 1 operations outstanding means the
 * oldest has landed.  Loads of other terms issued in between only make the
 * wait longer than necessary, never shorter.  The position is wave-uniform
 * (an SGPR): the switch below is a scalar branch tree, once per 64 postings.
 */
template <int I> __device__ __forceinline__ void bpair_request(const posting_t *np);
template <int PAIR, int N> struct bpair_take_impl;
#define	NXS_BPAIR(I, RD, RI, RP)							\
template <> __device__ __forceinline__ void						\
bpair_request<I>(const posting_t *np)							\
{											\
	asm volatile(									\
	    "global_load_dwordx2 " RP ", %0, off"					\
	    : : "v"(np) : "memory", RD, RI);						\
}											\
template <int N> struct bpair_take_impl<I, N> {					\
	static __device__ __forceinline__ void						\
	run(uint32_t &ad, float &ai, const posting_t *np)				\
	{										\
		asm volatile(								\
		    "s_waitcnt vmcnt(%3)\n\t"						\
		    "v_accvgpr_read_b32 %0, " RD "\n\t"					\
		    "v_accvgpr_read_b32 %1, " RI "\n\t"					\
		    "global_load_dwordx2 " RP ", %2, off"				\
		    : "=&v"(ad), "=&v"(ai) : "v"(np), "i"(N) : "memory", RD, RI);	\
	}										\
};
NXS_BPAIR(0, "a0", "a1", "a[0:1]")
NXS_BPAIR(1, "a2", "a3", "a[2:3]")
NXS_BPAIR(2, "a4", "a5", "a[4:5]")
NXS_BPAIR(3, "a6", "a7", "a[6:7]")
NXS_BPAIR(4, "a8", "a9", "a[8:9]")
NXS_BPAIR(5, "a10", "a11", "a[10:11]")
NXS_BPAIR(6, "a12", "a13", "a[12:13]")
NXS_BPAIR(7, "a14", "a15", "a[14:15]")
NXS_BPAIR(8, "a16", "a17", "a[16:17]")
NXS_BPAIR(9, "a18", "a19", "a[18:19]")
NXS_BPAIR(10, "a20", "a21", "a[20:21]")
NXS_BPAIR(11, "a22", "a23", "a[22:23]")
NXS_BPAIR(12, "a24", "a25", "a[24:25]")
NXS_BPAIR(13, "a26", "a27", "a[26:27]")
NXS_BPAIR(14, "a28", "a29", "a[28:29]")
NXS_BPAIR(15, "a30", "a31", "a[30:31]")
NXS_BPAIR(16, "a32", "a33", "a[32:33]")
NXS_BPAIR(17, "a34", "a35", "a[34:35]")
NXS_BPAIR(18, "a36", "a37", "a[36:37]")
NXS_BPAIR(19, "a38", "a39", "a[38:39]")
NXS_BPAIR(20, "a40", "a41", "a[40:41]")
NXS_BPAIR(21, "a42", "a43", "a[42:43]")
NXS_BPAIR(22, "a44", "a45", "a[44:45]")
NXS_BPAIR(23, "a46", "a47", "a[46:47]")
NXS_BPAIR(24, "a48", "a49", "a[48:49]")
NXS_BPAIR(25, "a50", "a51", "a[50:51]")
NXS_BPAIR(26, "a52", "a53", "a[52:53]")
NXS_BPAIR(27, "a54", "a55", "a[54:55]")
NXS_BPAIR(28, "a56", "a57", "a[56:57]")
NXS_BPAIR(29, "a58", "a59", "a[58:59]")
NXS_BPAIR(30, "a60", "a61", "a[60:61]")
NXS_BPAIR(31, "a62", "a63", "a[62:63]")
#undef NXS_BPAIR
template <int I, int N> __device__ __forceinline__ void
bpair_take(uint32_t &ad, float &ai, const posting_t *np)
{
	bpair_take_impl<I, N>::run(ad, ai, np);
}

/*
 * Exact wait for one ring load.  Vector memory operations retire in issue
 * order, so a load is done once at most `younger` operations are outstanding,
 * `younger` = operations issued after it.  The kernels stamp every ring load
 * with a per-wavefront issue counter, which gives a lower bound of that number
 * (operations the compiler issues are not counted: the wait can only be longer
 * than needed, never shorter).  Waiting for vmcnt(R - 1) instead made a term
 * whose window rotates right after another term's wait for that term's brand
 * new request: a full memory latency.  s_waitcnt takes an immediate, hence the
 * branch tree; `younger` is wave-uniform.
 */
#define	VM_WAIT(n)	asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
__device__ __forceinline__ void
vm_wait_younger(uint32_t younger)
{
	if (younger >= 8) {
		if (younger >= 16) {
			VM_WAIT(16);
		} else if (younger >= 12) {
			VM_WAIT(12);
		} else if (younger >= 10) {
			VM_WAIT(10);
		} else {
			VM_WAIT(8);
		}
	} else if (younger >= 4) {
		if (younger >= 6) {
			if (younger == 7) { VM_WAIT(7); } else { VM_WAIT(6); }
		} else {
			if (younger == 5) { VM_WAIT(5); } else { VM_WAIT(4); }
		}
	} else if (younger >= 2) {
		if (younger == 3) { VM_WAIT(3); } else { VM_WAIT(2); }
	} else {
		if (younger == 1) { VM_WAIT(1); } else { VM_WAIT(0); }
	}
}

/* request pair `pos` of term slot T (R pairs per term) */
template <int T, int R> __device__ __forceinline__ void
bring_request(uint32_t pos, const posting_t *np)
{
	static_assert(R == 1 || R == 2 || R == 4 || R == 8, "ring depth");
	static_assert(T * R + R <= 32, "AGPR pairs");
	if constexpr (R == 1) {
		bpair_request<T>(np);
	} else {
		static_for<R>([&](auto rc) {
			constexpr int r = decltype(rc)::value;
			if (pos == (uint32_t)r) {
				bpair_request<T * R + r>(np);
			}
		});
	}
}

/* wait for pair `pos` (the oldest of term slot T), read it, re-request it */
template <int T, int R> __device__ __forceinline__ void
bring_take(uint32_t pos, uint32_t younger, uint32_t &ad, float &ai, const posting_t *np)
{
#ifndef NXS_VMWAIT_STAMPS
	(void)younger;
	vm_wait_younger(R - 1);		/* the stamp-free wait: the R - 1 siblings are younger */
#else
	vm_wait_younger(younger);
#endif
	if constexpr (R == 1) {
		bpair_take<T, 63>(ad, ai, np);
	} else {
		static_for<R>([&](auto rc) {
			constexpr int r = decltype(rc)::value;
			if (pos == (uint32_t)r) {
				bpair_take<T * R + r, 63>(ad, ai, np);
			}
		});
	}
}

/* issue stamps of a term's R ring loads, oldest first (FIFO) */
template <int R> struct ring_stamps {
	uint32_t st[R];
	/* operations issued after the oldest load of this ring */
	__device__ __forceinline__ uint32_t younger(uint32_t seq) const { return seq - st[0] - 1; }
	/* the oldest was consumed and requested again with stamp `seq` */
	__device__ __forceinline__ void rotate(uint32_t seq)
	{
#pragma unroll
		for (int r = 0; r + 1 < R; r++) {
			st[r] = st[r + 1];
		}
		st[R - 1] = seq;
	}
};

/*
 * Wave-cooperative lower bound: first index in [lo, hi) (relative to pt) whose
 * doc is >= bound, or hi.  64-ary: each round the 64 lanes probe 64 evenly
 * spaced postings, so a 10M-entry list needs 4 dependent loads, not 24.
 * All arguments and the result are wave-uniform.
 */
__device__ static inline int32_t
wave_lower_bound(const posting_t *__restrict__ pt, int32_t lo, int32_t hi, uint32_t bound)
{
	const int32_t lane = (int32_t)(threadIdx.x & 63);

	while (hi - lo > WAVE) {
		const int32_t step = (hi - lo + WAVE - 1) / WAVE;
		const int32_t idx = lo + lane * step;
		const bool valid = idx < hi;
		uint32_t v = 0xffffffffu;
		if (valid) {
			v = pt[idx].doc;
		}
		/* lanes are monotone: the first lane whose probe is >= bound */
		const uint64_t m = ballot64(!valid || v >= bound);
		const int32_t L = m ? (int32_t)__ffsll((long long)m) - 1 : WAVE;
		if (L == 0) {
			return lo;
		}
		const int32_t nlo = lo + (L - 1) * step + 1;
		const int32_t nhi = (L < WAVE && lo + L * step < hi) ? lo + L * step : hi;
		lo = nlo;
		hi = nhi;
	}
	{
		const int32_t idx = lo + lane;
		const bool valid = idx < hi;
		uint32_t v = 0;
		if (valid) {
			v = pt[idx].doc;
		}
		const uint64_t m = ballot64(valid && v >= bound);
		return m ? lo + (int32_t)__ffsll((long long)m) - 1 : hi;
	}
}

/* byte index of doc d's mask inside a tile: a u32 read at word (s*64+lane)
 * yields the four docs s*256 + j*64 + lane, j = 0..3 */
__device__ static inline uint32_t
mask_byte(uint32_t d)
{
	return ((d >> 8) << 8) | ((d & 63) << 2) | ((d >> 6) & 3);
}

/* evaluate the postfix boolean program on a presence mask */
__device__ static inline bool
eval_prog(const uint8_t *prog, uint32_t len, uint32_t m)
{
	uint64_t st = 0;	/* bit stack, top at bit 0 */
	for (uint32_t i = 0; i < len; i++) {
		const uint8_t op = prog[i];
		if (op < NXSGPU_MAX_TOKENS) {
			st = (st << 1) | ((m >> op) & 1);
		} else if (op == NXSGPU_OP_EMPTY) {
			st <<= 1;
		} else {
			const uint64_t b = st & 1, a = (st >> 1) & 1;
			uint64_t r;
			if (op == NXSGPU_OP_AND) r = a & b;
			else if (op == NXSGPU_OP_OR) r = a | b;
			else r = a & ~b & 1;
			st = ((st >> 2) << 1) | r;
		}
	}
	return st & 1;
}

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
	const float hint = (MODE == MODE_TOPK && A.k <= WAVE) ? range_hint(A, qm, g) : -INFINITY;
	float thr = hint;
	const uint32_t kidx = (A.k >= 1 && A.k <= WAVE) ? A.k - 1 : WAVE - 1;
	const bool track = (MODE == MODE_TOPK) && A.k <= WAVE;
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
					if (MODE == MODE_TOPK && n_out + ne > A.seg_cap) {
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
				if (MODE == MODE_TOPK && n_out + ne > A.seg_cap) {
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
	if (lane == 0) {
		if (MODE != MODE_ALL) {
			A.seg_count[seg] = ovf ? 0 : n_out;
		}
		if (MODE == MODE_TOPK && ovf) {
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
/*
 * A workgroup here is ONE wavefront: its DS operations execute in issue order,
 * so cross-lane LDS hand-offs need no s_barrier -- and must not get one:
 * __syncthreads() also drains vmcnt(0), i.e. every posting prefetch in flight.
 * This only stops the compiler from moving memory operations across the point.
 */
#define	WAVE_SYNC()	__builtin_amdgcn_wave_barrier()

#define	LIST_CAP	512
#ifndef SCAN8_RING_MAX
#define	SCAN8_RING_MAX	2		/* prefetch ring depth of the one-window tile path */
#endif
#ifndef SCANR_RING
#define	SCANR_RING	4		/* prefetch ring depth of the required-term path (span rounds: 1, 2, 4 measured equal; whole-window rounds rotate the driver every round: 1 -> 4 = 1.22 -> 1.19 ms per C3 step) */
#endif
#ifndef SCANM_RING
#define	SCANM_RING	2		/* prefetch ring depth of the mask path */
#endif
#define	TCAND_CAP	64

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

	constexpr int KSH = NT <= 1 ? 3 : NT <= 2 ? 2 : 0;
	constexpr int K = 1 << KSH;
	constexpr int SW = WAVE * K;

	const unsigned lane = threadIdx.x;
	const item_t item = A.items[A.item_base + blockIdx.x];
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
	constexpr int RING = SCAN8_RING_MAX;
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
	const float hint = (MODE == MODE_TOPK && A.k <= WAVE) ? range_hint(A, qm, g) : -INFINITY;
	float thr = hint;
	const uint32_t kidx = (A.k >= 1 && A.k <= WAVE) ? A.k - 1 : WAVE - 1;
	const bool track = (MODE == MODE_TOPK) && A.k <= WAVE;
	uint32_t n_out = 0;
	bool ovf = false;
	const uint64_t out_base = (MODE == MODE_ALL) ? A.seg_off[seg] : seg * A.seg_cap;

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
		if (MODE == MODE_TOPK && ballot64(__uint_as_float(tmax) > thr) == 0) {
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
					if (MODE == MODE_TOPK && n_out + ncand > A.seg_cap) {
						ovf = true;
					} else if (lane < ncand) {
						const uint64_t o = out_base + n_out + rank;
						A.cand_doc[o] = base + cd;
						A.cand_sc[o] = cs;
					}
					n_out += ncand;
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
				if (MODE == MODE_TOPK && n_out + ne > A.seg_cap) {
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
	if (lane == 0) {
		if (MODE != MODE_ALL) {
			A.seg_count[seg] = ovf ? 0 : n_out;
		}
		if (MODE == MODE_TOPK && ovf) {
			A.overflow[q] = 1;
		}
	}
}

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
#define	QSUM_MAX	224		/* quantised score bound of a doc holding every term at its largest impact */

#ifdef NXS_STATS
/* diagnostic build only (make variant XFLAGS=-DNXS_STATS): k_scanm event counts
 * and cycle spans, read back with nxsgpu_debug_stats() */
__device__ unsigned long long g_stats[16];
__device__ unsigned long long g_rstats[8];	/* k_replay: queries, then 10 ns ticks per phase, candidates, inserts */
extern "C" void
nxsgpu_debug_rstats(unsigned long long *out, int reset)
{
	unsigned long long z[8] = { 0 };
	(void)hipDeviceSynchronize();
	(void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rstats), sizeof(z));
	if (reset) {
		(void)hipMemcpyToSymbol(HIP_SYMBOL(g_rstats), z, sizeof(z));
	}
}
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
	__shared__ __attribute__((aligned(16))) uint32_t s_mask[MT_W / 4 + WAVE];	/* + one dummy word per lane */
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

	for (uint32_t i = lane; i < MT_W / 4 + WAVE; i += WAVE) {
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
	uint32_t Ad[NT], Nd[NT], rp[NT];
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
			bring_take<t, RING>(rp[t], rst[t].younger(vseq), Nd[t], Ni[t], np);
			rst[t].rotate(vseq++);
			rp[t] = (rp[t] + 1) & (RING - 1);
		} else {
			vmN[t] = 0;
			Nd[t] = 0xffffffffu;	/* no doc */
		}
		refresh_ldoc(tc);
	};

	/* DROP: the dense tokens are never streamed; their impacts come from the
	 * terms' columns (scan_args_t::dense_col) */
	const uint32_t dmask = DROP ? (uint32_t)__builtin_amdgcn_readfirstlane((int)Q->drop_mask) : 0u;
	/* DROP: the range's cold phase (k_cold) stopped at doc cs_cur: only docs below
	 * it are left, and only for the sparse terms */
	const uint32_t *cs = A.cold_state + seg * 16;
	const uint32_t cs_left = DROP ? (uint32_t)__builtin_amdgcn_readfirstlane((int)cs[0]) : 0u;	/* 0: range used up */
	const uint32_t cs_nout = DROP ? (uint32_t)__builtin_amdgcn_readfirstlane((int)cs[1]) : 0u;
	const float cs_thr = DROP ? __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)cs[2])) : 0.0f;
	const bool cs_ovf = DROP && __builtin_amdgcn_readfirstlane((int)cs[3]) != 0;
	static_for<NT>([&](auto tc) {
		constexpr int t = decltype(tc)::value;
		pt[t] = A.post;
		lo[t] = hi[t] = ab[t] = 0;
		pdoc[t] = -1;
		vmA[t] = vmN[t] = 0;
		rp[t] = 0;
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
		if (DROP) {
			if (((dmask >> t) & 1) || cs_left == 0) {
				hi[t] = lo[t];		/* no postings as far as the windows are concerned */
			} else if (t < (int)nt) {
				hi[t] = min(hi[t], (int32_t)__builtin_amdgcn_readfirstlane((int)cs[4 + t]));
			}
		}
		if (hi[t] > lo[t]) {
			ab[t] = ((hi[t] - 1) >> 6) << 6;
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
			refresh_pdoc(tc);
			refresh_ldoc(tc);
		}
	});

	float top = DROP ? A.cold_top[seg * 64 + lane] : -INFINITY;
	const float hint = range_hint(A, qm, g);	/* 0 = nothing published yet */
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
				U += tmx[t];			/* token order, f32: see above */
				qU += (uint32_t)(tmx[t] * qs) + 2;
			} else {
				q1max = max(q1max, (uint32_t)(tmx[t] * qs) + 2);
			}
		}
		q1max = (uint32_t)__builtin_amdgcn_readfirstlane((int)q1max);
		qU = (uint32_t)__builtin_amdgcn_readfirstlane((int)qU);
		thr_q -= (int32_t)qU;
	}

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
					if (hi[t] > lo[t]) {
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
			rp[t] = rfl32(rp[t]);
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
			const uint32_t dd = wd - base;
			const uint32_t sh = (dd & 3) * 8;
			const uint32_t w = inl ? (dd >> 2) : MT_W / 4 + lane;
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
			const uint32_t words = ((uint32_t)md - base + 4) >> 2;
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
			A.overflow[q] = 1;
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
				U += Q->tmax[t];		/* token order, f32 */
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
	const unsigned lane = threadIdx.x;
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
	const float hint = (MODE == MODE_TOPK && A.k <= WAVE) ? range_hint(A, qm, g) : -INFINITY;
	float thr = hint;
	const uint32_t kidx = (A.k >= 1 && A.k <= WAVE) ? A.k - 1 : WAVE - 1;
	const bool track = (MODE == MODE_TOPK) && A.k <= WAVE;
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
			if (MODE == MODE_TOPK && n_out + ne > A.seg_cap) {
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
	if (lane == 0) {
		if (MODE != MODE_ALL) {
			A.seg_count[seg] = ovf ? 0 : n_out;
		}
		if (MODE == MODE_TOPK && ovf) {
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
#ifndef RW
#define	RW	4096		/* docs per round span (LDS byte map; SCANR_HASH 0) */
#endif
#ifndef SCANR_HASH
#define	SCANR_HASH	1		/* a round = a whole driver window, its docs in an LDS hash table */
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
	const float hint = (MODE == MODE_TOPK && A.k <= WAVE) ? range_hint(A, qm, g) : -INFINITY;
	float thr = hint;
	const uint32_t kidx = (A.k >= 1 && A.k <= WAVE) ? A.k - 1 : WAVE - 1;
	const bool track = (MODE == MODE_TOPK) && A.k <= WAVE;
	uint32_t n_out = 0;
	bool ovf = false;
	bool done = false;		/* a required slot ran out: nothing below can match */
	const uint64_t out_base = (MODE == MODE_ALL) ? A.seg_off[seg] : seg * A.seg_cap;

	while (!done && pdoc[0] >= 0) {
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
				/* nothing above the driver's top doc can match: drop it unread */
				if (pdoc[j] > dtop) {
					for (int tries = 0; ; tries++) {
						vm[j] &= ~ballot64(Ad[j] > (uint32_t)dtop);
						if (vm[j] || ab[j] <= lo[j]) {
							break;		/* the boundary is in this window / list exhausted */
						}
						if (tries < 2) {
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
					if (MODE == MODE_TOPK && n_out + ne > A.seg_cap) {
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
	if (lane == 0) {
		if (MODE != MODE_ALL) {
			A.seg_count[seg] = ovf ? 0 : n_out;
		}
		if (MODE == MODE_TOPK && ovf) {
			A.overflow[q] = 1;
		}
	}
}

#ifdef NXS_EXPERIMENTAL	/* opt-in build: measured not faster than the tiles (DESIGN.md "dead ends") */
/*
 * k_scanh: the posting-step path for queries without a very dense term.
 *
 * Every term streams its list through register sets as in k_scan8.  A STEP
 * picks base = the largest, over the terms, of the lowest doc still held in
 * the term's set A.  All unconsumed postings with doc >= base are then in
 * registers, for every term: the term that defines base drains its whole set
 * (so a query needs at most sum_t ceil(df_t / 64K) steps, however sparse it
 * is), the others contribute the part of their set above base.  The docs of a
 * step can span far more than an LDS tile, so scores accumulate in a small
 * LDS hash table keyed by doc (slot = low doc bits, linear probing, claims are
 * written then verified -- no atomics).  Terms are applied in token order, so
 * a doc's f32 sum order is the reference's (results.c:134-136).  Claimed slots
 * go to a list: the table is scanned and wiped through it.  Steps run in
 * descending doc ranges and a step's candidates are rank-sorted by doc, so the
 * segment is in descending doc order like k_scan8's.
 */
template <int MODE, int NT>
__global__ void __launch_bounds__(WAVE)
k_scanh(const scan_args_t A)
{
	constexpr int KSH = NT <= 2 ? 2 : NT <= 3 ? 1 : 0;
	constexpr int K = 1 << KSH;
	constexpr int SW = WAVE * K;
	constexpr int MAXE = WAVE * K * NT;		/* table entries per step */
	constexpr int TAB = MAXE <= 256 ? 512 : 1024;	/* load factor <= 1/2 */
	constexpr uint32_t EMPTY = 0xffffffffu;

	__shared__ uint32_t s_key[TAB];
	__shared__ float s_val[TAB];
	__shared__ uint8_t s_msk[TAB];
	__shared__ uint16_t s_list[MAXE];
	__shared__ uint32_t s_cd[MAXE];
	__shared__ float s_cs[MAXE];
	__shared__ uint32_t s_truth[8];
	__shared__ int64_t s_init[16];

	const unsigned lane = threadIdx.x;
	const item_t item = A.items[A.item_base + blockIdx.x];
	const uint32_t q = item.q, g = item.g;
	const qmeta_t qm = A.qmeta[q];
	const dev_query_t *Q = &A.queries[q];
	const uint32_t nt = Q->nt;
	const uint64_t seg = (uint64_t)qm.seg_first + g;
	const uint64_t dlo = (uint64_t)g * qm.group_docs;
	const uint64_t dhi = min(A.n_docs, dlo + (uint64_t)qm.group_docs);

	for (uint32_t i = lane; i < TAB; i += WAVE) {
		s_key[i] = EMPTY;
		s_val[i] = 0.0f;
		s_msk[i] = 0;
	}
	if (lane < 8) {
		s_truth[lane] = Q->truth[lane];
	}
	if (lane < 16) {
		const uint32_t t = lane & 7;
		int64_t v = 0;
		if (t < nt) {
			const uint64_t cb = ((uint64_t)qm.seg_first + q + g) * NXSGPU_MAX_TOKENS + t;
			(void)dlo; (void)dhi;
			v = (int64_t)A.cursors[cb + (lane < 8 ? NXSGPU_MAX_TOKENS : 0)];
		}
		s_init[lane] = v;
	}
	WAVE_SYNC();

	/* wave-uniform per-term state */
	const posting_t *pt[NT];
	int32_t hi[NT], lo[NT], pdoc[NT], lowdoc[NT];
	uint32_t Ad[NT][K], Bd[NT][K];
	float Ai[NT][K], Bi[NT][K];

#pragma unroll
	for (int t = 0; t < NT; t++) {
		pt[t] = A.post;
		hi[t] = lo[t] = 0;
		pdoc[t] = lowdoc[t] = -1;
#pragma unroll
		for (int k = 0; k < K; k++) {
			Ad[t][k] = Bd[t][k] = 0;
			Ai[t][k] = Bi[t][k] = 0.0f;
		}
		if (t < (int)nt) {
			pt[t] = A.post + Q->pbeg[t];
			hi[t] = __builtin_amdgcn_readfirstlane((int32_t)s_init[t]);
			lo[t] = __builtin_amdgcn_readfirstlane((int32_t)s_init[8 + t]);
			if (hi[t] > lo[t]) {
				const int32_t ab = ((hi[t] - 1) >> (6 + KSH)) << (6 + KSH);
#pragma unroll
				for (int k = 0; k < K; k++) {
					const int32_t ia = ab + k * WAVE + (int32_t)lane, ib = ia - SW;
					if (ia >= lo[t] && ia < hi[t]) {
						const posting_t p = pt[t][ia];
						Ad[t][k] = p.doc; Ai[t][k] = p.imp;
					}
					if (ib >= lo[t]) {
						const posting_t p = pt[t][ib];
						Bd[t][k] = p.doc; Bi[t][k] = p.imp;
					}
				}
				const int32_t kt = ((hi[t] - 1) >> 6) & (K - 1);
#pragma unroll
				for (int k = 0; k < K; k++) {
					if (k == kt) {
						pdoc[t] = __builtin_amdgcn_readlane((int)Ad[t][k], (hi[t] - 1) & 63);
					}
				}
				/* lowest doc held in set A, or -1 if the set reaches the
				 * start of this range's postings */
				if (ab > lo[t]) {
					lowdoc[t] = __builtin_amdgcn_readlane((int)Ad[t][0], 0);
				}
			}
		}
	}

	float top = -INFINITY;
	const float hint = (MODE == MODE_TOPK && A.k <= WAVE) ? range_hint(A, qm, g) : -INFINITY;
	float thr = hint;
	const uint32_t kidx = (A.k >= 1 && A.k <= WAVE) ? A.k - 1 : WAVE - 1;
	const bool track = (MODE == MODE_TOPK) && A.k <= WAVE;
	uint32_t n_out = 0;
	bool ovf = false;
	const uint64_t out_base = (MODE == MODE_ALL) ? A.seg_off[seg] : seg * A.seg_cap;

	for (;;) {
		int32_t md = -1, bs = -1;
#pragma unroll
		for (int t = 0; t < NT; t++) {
			md = max(md, pdoc[t]);
			bs = max(bs, lowdoc[t]);
		}
		if (md < 0) {
			break;		/* every list is consumed */
		}
		const uint32_t base = bs < 0 ? 0u : (uint32_t)bs;
		uint32_t n_list = 0;
		float tmax = -INFINITY;

#pragma unroll
		for (int t = 0; t < NT; t++) {
			if (t < (int)nt && pdoc[t] >= (int32_t)base) {
				const int32_t ab = ((hi[t] - 1) >> (6 + KSH)) << (6 + KSH);
				uint32_t ctot = 0;
				bool more = true;
#pragma unroll
				for (int k = K - 1; k >= 0; k--) {
					if (more && hi[t] > ab + k * WAVE) {
						const int32_t idx = ab + k * WAVE + (int32_t)lane;
						const uint32_t doc = Ad[t][k];
						const bool in = idx >= lo[t] && idx < hi[t] && doc >= base;
						const uint32_t c = __popcll(ballot64(in));
						const int32_t top_ = min(hi[t], ab + (k + 1) * WAVE);
						const int32_t bot_ = max(lo[t], ab + k * WAVE);
						ctot += c;
						if ((int32_t)c < top_ - bot_) {
							more = false;
						}
						if (c) {
							/* find or claim the doc's slot */
							/* volatile: the claim must really be re-read, not
							 * forwarded from this lane's own store */
							volatile uint32_t *vkey = s_key;
							uint32_t slot = doc & (TAB - 1);
							bool pending = in, isnew = false;
							while (ballot64(pending)) {
								uint32_t kk = 0;
								if (pending) {
									kk = vkey[slot];
									if (kk == EMPTY) {
										vkey[slot] = doc;
									}
								}
								if (pending) {
									if (kk == doc) {
										pending = false;
									} else if (kk == EMPTY) {
										/* several lanes may have written
										 * this slot: one value landed */
										if (vkey[slot] == doc) {
											pending = false;
											isnew = true;
										}
									} else {
										slot = (slot + 1) & (TAB - 1);
									}
								}
							}
							if (in) {
								const float v = s_val[slot] + Ai[t][k];
								s_val[slot] = v;
								s_msk[slot] = (uint8_t)(s_msk[slot] | (1u << t));
								tmax = fmaxf(tmax, v);
							}
							const uint64_t fb = ballot64(isnew);
							if (isnew) {
								s_list[n_list + lanes_below(fb)] = (uint16_t)slot;
							}
							n_list += __popcll(fb);
						}
					}
				}
				hi[t] = __builtin_amdgcn_readfirstlane(hi[t] - (int32_t)ctot);
				pdoc[t] = -1;
				if (hi[t] <= lo[t]) {
					lowdoc[t] = -1;
				} else if (hi[t] == ab) {
					/* set A drained: take over set B, K new loads in flight */
#pragma unroll
					for (int k = 0; k < K; k++) {
						Ad[t][k] = Bd[t][k];
						Ai[t][k] = Bi[t][k];
					}
#pragma unroll
					for (int k = 0; k < K; k++) {
						const int32_t ib = ab - 2 * SW + k * WAVE + (int32_t)lane;
						Bd[t][k] = 0; Bi[t][k] = 0.0f;
						if (ib >= lo[t]) {
							const posting_t p = pt[t][ib];
							Bd[t][k] = p.doc; Bi[t][k] = p.imp;
						}
					}
					lowdoc[t] = -1;
					if (ab - SW > lo[t]) {
						lowdoc[t] = __builtin_amdgcn_readlane((int)Ad[t][0], 0);
					}
					pdoc[t] = __builtin_amdgcn_readlane((int)Ad[t][K - 1], WAVE - 1);
				} else {
					const int32_t kt = ((hi[t] - 1) >> 6) & (K - 1);
#pragma unroll
					for (int k = 0; k < K; k++) {
						if (k == kt) {
							pdoc[t] = __builtin_amdgcn_readlane((int)Ad[t][k], (hi[t] - 1) & 63);
						}
					}
				}
			}
		}
		WAVE_SYNC();

		/* candidates of this step (skipped when nothing beat the threshold:
		 * scores only grow within a step, see k_scan8) */
		if (MODE == MODE_COUNT || ballot64(tmax > thr) != 0) {
			uint32_t ncand = 0;
			for (uint32_t off = 0; off < n_list; off += WAVE) {
				const uint32_t i = off + lane;
				const bool valid = i < n_list;
				uint32_t d = 0, m = 0;
				float sc = 0.0f;
				if (valid) {
					const uint32_t slot = s_list[i];
					d = s_key[slot];
					m = s_msk[slot];
					sc = s_val[slot];
				}
				if (MODE == MODE_COUNT) {
					const bool match = valid && ((s_truth[m >> 5] >> (m & 31)) & 1);
					n_out += __popcll(ballot64(match));
					continue;
				}
				const bool pre = valid && (sc > thr);
				if (ballot64(pre) == 0) {
					continue;
				}
				const bool cand = pre && ((s_truth[m >> 5] >> (m & 31)) & 1);
				const uint64_t bal = ballot64(cand);
				if (cand) {
					const uint32_t j = ncand + lanes_below(bal);
					s_cd[j] = d;
					s_cs[j] = sc;
				}
				ncand += __popcll(bal);
			}
			if (MODE != MODE_COUNT && ncand) {
				WAVE_SYNC();
				if (MODE == MODE_TOPK && n_out + ncand > A.seg_cap) {
					ovf = true;
				} else {
					/* rank by doc, descending: docs are distinct */
					for (uint32_t i0 = 0; i0 < ncand; i0 += WAVE) {
						const uint32_t i = i0 + lane;
						const uint32_t cd = i < ncand ? s_cd[i] : 0;
						uint32_t rank = 0;
						for (uint32_t j = 0; j < ncand; j++) {
							rank += s_cd[j] > cd;
						}
						if (i < ncand) {
							const uint64_t o = out_base + n_out + rank;
							A.cand_doc[o] = cd;
							A.cand_sc[o] = s_cs[i];
						}
					}
				}
				n_out += ncand;
				if (track) {
					for (uint32_t j = 0; j < ncand; j++) {
						const float v = s_cs[j];
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
		/* wipe the table through the list */
		for (uint32_t off = 0; off < n_list; off += WAVE) {
			const uint32_t i = off + lane;
			if (i < n_list) {
				const uint32_t slot = s_list[i];
				s_key[slot] = EMPTY;
				s_val[slot] = 0.0f;
				s_msk[slot] = 0;
			}
		}
		WAVE_SYNC();
	}

	if (MODE == MODE_TOPK && track && !ovf) {
		range_publish(A, seg, __shfl(top, kidx));	/* k-th best of this range */
	}
	if (lane == 0) {
		if (MODE != MODE_ALL) {
			A.seg_count[seg] = ovf ? 0 : n_out;
		}
		if (MODE == MODE_TOPK && ovf) {
			A.overflow[q] = 1;
		}
	}
}
#endif /* NXS_EXPERIMENTAL */

/* ------------------------------------------------------------------ */
/* k_replay: the reference's heap, replayed exactly                     */
/* ------------------------------------------------------------------ */

/* heap_remove_min: src/algo/heap.c:133-189 (comparator: score only) */
__device__ static void
heap_remove_min(float *hs, uint32_t *hd, uint32_t *nitems, float *os, uint32_t *od)
{
	uint32_t i = 0, max_, left;

	*os = hs[0];
	*od = hd[0];
	if ((max_ = --(*nitems)) == 0) {
		return;
	}
	hs[0] = hs[max_];
	hd[0] = hd[max_];
	while ((left = i * 2 + 1) < max_) {
		const float ps = hs[i];
		const uint32_t pd = hd[i];
		const uint32_t right = i * 2 + 2;
		uint32_t smallest = i;

		if (hs[left] < ps) {
			smallest = left;
		}
		if (right < max_ && hs[right] < hs[smallest]) {
			smallest = right;
		}
		if (smallest == i) {
			break;
		}
		hs[i] = hs[smallest];
		hd[i] = hd[smallest];
		hs[smallest] = ps;
		hd[smallest] = pd;
		i = smallest;
	}
}

/* heap_add: src/algo/heap.c:58-124; caller has checked acceptance */
__device__ static void
heap_add(float *hs, uint32_t *hd, uint32_t *nitems, uint32_t cap, float s, uint32_t d)
{
	uint32_t i;

	if (*nitems == cap) {
		float ts; uint32_t td;
		heap_remove_min(hs, hd, nitems, &ts, &td);
	}
	i = (*nitems)++;
	hs[i] = s;
	hd[i] = d;
	while (i) {
		const uint32_t parent = (i - 1) / 2;
		const float ps = hs[parent];
		const uint32_t pd = hd[parent];
		if (s >= ps) {		/* heap.c:103 */
			break;
		}
		hs[parent] = s;
		hd[parent] = d;
		hs[i] = ps;
		hd[i] = pd;
		i = parent;
	}
}

/*
 * The same heap with its array ACROSS THE LANES of the wavefront (element i in
 * lane i, capacity <= 64): an element is read with v_readlane and written with
 * v_writelane, a few cycles each, where the LDS array cost a full LDS round
 * trip per access on lane 0 (about 1 us per heap_add: the replay of a single
 * query took 60-80 us, most of nxs_index_search()'s latency).  Every lane runs
 * the same scalar control flow; indices and values are wave-uniform.  Line for
 * line the functions above.
 */
__device__ __forceinline__ float
rh_gets(float hs, uint32_t i)
{
	return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hs), (int)i));
}

__device__ __forceinline__ uint32_t
rh_getd(uint32_t hd, uint32_t i)
{
	return (uint32_t)__builtin_amdgcn_readlane((int)hd, (int)i);
}

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"	/* (M0 is reserved: nothing in these kernels uses it) */
__device__ __forceinline__ void
rh_set(float &hs, uint32_t &hd, uint32_t i, float s, uint32_t d)
{
	/* (no writelane builtin in this compiler; value and lane select are SGPRs) */
	const int sv = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s));
	const int dv = __builtin_amdgcn_readfirstlane((int)d);
	const int li = __builtin_amdgcn_readfirstlane((int)i);
	/* (one SGPR per VOP3 instruction on this target: the lane select goes through M0) */
	/* (s_nop: inline asm is outside the compiler's hazard recogniser; a scalar write
	 * of the lane select right in front of its vector use costs one idle cycle) */
	asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %4, m0"
	    : "+v"(hs), "+v"(hd) : "s"(sv), "s"(li), "s"(dv) : "m0");
}
#pragma clang diagnostic pop

/* heap_remove_min: src/algo/heap.c:133-189 */
__device__ static inline void
rheap_remove_min(float &hs, uint32_t &hd, uint32_t &nitems, float &os, uint32_t &od)
{
	uint32_t i = 0, max_, left;

	os = rh_gets(hs, 0);
	od = rh_getd(hd, 0);
	if ((max_ = --nitems) == 0) {
		return;
	}
	rh_set(hs, hd, 0, rh_gets(hs, max_), rh_getd(hd, max_));
	while ((left = i * 2 + 1) < max_) {
		const float ps = rh_gets(hs, i);
		const uint32_t pd = rh_getd(hd, i);
		const uint32_t right = i * 2 + 2;
		uint32_t smallest = i;

		if (rh_gets(hs, left) < ps) {
			smallest = left;
		}
		if (right < max_ && rh_gets(hs, right) < rh_gets(hs, smallest)) {
			smallest = right;
		}
		if (smallest == i) {
			break;
		}
		rh_set(hs, hd, i, rh_gets(hs, smallest), rh_getd(hd, smallest));
		rh_set(hs, hd, smallest, ps, pd);
		i = smallest;
	}
}

/* heap_add: src/algo/heap.c:58-124; caller has checked acceptance */
__device__ static inline void
rheap_add(float &hs, uint32_t &hd, uint32_t &nitems, uint32_t cap, float s, uint32_t d)
{
	uint32_t i;

	if (nitems == cap) {
		float ts; uint32_t td;
		rheap_remove_min(hs, hd, nitems, ts, td);
	}
	i = nitems++;
	rh_set(hs, hd, i, s, d);
	while (i) {
		const uint32_t parent = (i - 1) / 2;
		const float ps = rh_gets(hs, parent);
		const uint32_t pd = rh_getd(hd, parent);
		if (s >= ps) {		/* heap.c:103 */
			break;
		}
		rh_set(hs, hd, parent, s, d);
		rh_set(hs, hd, i, ps, pd);
		i = parent;
	}
}

/*
 * The same heap once more for 64 < k <= REPLAY_LDS_K (the API's default limit is
 * 1000), in dynamic LDS as (score, doc) PAIRS: both children of a node come with
 * one read, and the element on the move stays in registers ("hole" form of the
 * reference's swaps: the same comparisons, the same final array).  One level is
 * one dependent LDS read instead of half a dozen: a heap_add on the 1000-entry
 * heap took 2.4 us with separate score / doc arrays (global memory or LDS alike).
 */
#define	REPLAY_LDS_K	8000
__device__ static inline void
lheap_remove_min(uint2 *h, uint32_t &nitems, float &os, uint32_t &od)
{
	uint32_t i = 0, max_, left;

	os = __uint_as_float(h[0].x);
	od = h[0].y;
	if ((max_ = --nitems) == 0) {
		return;
	}
	const uint2 e = h[max_];			/* heap.c:146-147: the last item goes to the root ... */
	const float ps = __uint_as_float(e.x);
	while ((left = i * 2 + 1) < max_) {		/* ... and sinks (heap.c:149-187) */
		const uint32_t right = left + 1;
		const uint2 cl = h[left];
		const uint2 cr = h[right < max_ ? right : left];
		uint32_t smallest = i;
		float ss = ps;
		uint2 cs = e;

		if (__uint_as_float(cl.x) < ps) {
			smallest = left;
			ss = __uint_as_float(cl.x);
			cs = cl;
		}
		if (right < max_ && __uint_as_float(cr.x) < ss) {
			smallest = right;
			cs = cr;
		}
		if (smallest == i) {
			break;
		}
		h[i] = cs;
		i = smallest;
	}
	h[i] = e;
}

__device__ static inline void
lheap_add(uint2 *h, uint32_t &nitems, uint32_t cap, float s, uint32_t d)
{
	uint32_t i;

	if (nitems == cap) {
		float ts; uint32_t td;
		lheap_remove_min(h, nitems, ts, td);
	}
	i = nitems++;
	while (i) {					/* heap.c:96-122 */
		const uint32_t parent = (i - 1) / 2;
		const uint2 pe = h[parent];
		if (s >= __uint_as_float(pe.x)) {	/* heap.c:103 */
			break;
		}
		h[i] = pe;
		i = parent;
	}
	h[i] = make_uint2(__float_as_uint(s), d);
}

struct replay_args_t {
	const qmeta_t *		qmeta;
	uint32_t		seg_cap;	/* 0 => segments addressed by seg_off */
	const uint32_t *	seg_count;
	const uint64_t *	seg_off;
	const uint32_t *	cand_doc;
	const float *		cand_sc;
	const uint64_t *	doc_ids;
	uint32_t		k;		/* heap capacity (limit, clamped) */
	/* heap storage when it does not fit LDS: [Q] slices via heap_off */
	float *			gheap_s;
	uint32_t *		gheap_d;
	const uint64_t *	heap_off;	/* [Q+1] or NULL */
	/* outputs */
	uint64_t *		out_ids;
	float *			out_sc;
	uint32_t *		out_count;
	const uint64_t *	out_off;	/* [Q+1] or NULL => q * k */
	const uint32_t *	skip;		/* [Q] nonzero => leave untouched */
	const uint32_t *	qlist;		/* NULL, or the queries this launch replays (blockIdx -> query) */
	/* record mode (nxsgpu_batch_begin): the result of query q goes to the
	 * fixed-size record rec_base + rec_slot[q] * rec_bytes instead of out_* */
	uint8_t *		rec_base;
	const uint32_t *	rec_slot;
	uint32_t		rec_bytes;
	/*
	 * Doc-sharded mode (N4): every item the heap ACCEPTS, in feed order, is
	 * also written to log_*[q * log_cap ...] -- the exact sequence the
	 * reference's heap would take from this shard's docs if they were fed alone;
	 * the sequence it takes from them inside the global feed is a subsequence
	 * (the global root is never below the local one).  log_cnt[q] keeps counting
	 * past log_cap (overflow).  cand_doc == NULL: the candidate's index is its
	 * doc handle (doc_ids[] is then indexed like cand_sc[]).
	 */
	uint64_t *		log_ids;
	float *			log_sc;
	uint32_t *		log_cnt;
	uint32_t		log_cap;
	const uint32_t *	log_slot;	/* [Q] row of the log per query (NULL: q) */
};

/* where the heap lives: global memory (any k), across the lanes (k <= 64), or in
 * dynamic LDS as pairs (k <= REPLAY_LDS_K) */
#define	HEAP_GLOBAL	0
#define	HEAP_REG	1
#define	HEAP_LDS	2
template <int HEAP>
__global__ void __launch_bounds__(WAVE)
k_replay(const replay_args_t A)
{
	constexpr bool LDS_HEAP = HEAP == HEAP_REG;	/* (historic name: the k <= 64 heap) */
	extern __shared__ uint2 dyn_heap[];		/* HEAP_LDS: [k] */
	__shared__ uint32_t s_n;	/* (global-memory heap only) */
	__shared__ float s_min;

	const unsigned lane = threadIdx.x;
#ifdef NXS_STATS
	const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
	unsigned long long rt1 = rt0, rt2 = rt0, rt3 = rt0;
	unsigned long long n_ins = 0, n_cand = 0;
#define	RSTAT(x)	x
#else
#define	RSTAT(x)
#endif
	const uint32_t q = A.qlist ? A.qlist[blockIdx.x] : blockIdx.x;
	float *hs = NULL;
	uint32_t *hd = NULL, cap;
	/* LDS_HEAP (k <= 64): the heap lives in these two registers, element i in lane i */
	float rhs = 0.0f;
	uint32_t rhd = 0, rn = 0;
	float rmin = 0.0f;

	if (A.skip && A.skip[q]) {
		/* the query overflowed its candidate segments: its record says so (the
		 * owner re-runs it on the exact path; with sharding every rank sees it) */
		if (A.rec_base && lane == 0) {
			((uint32_t *)(A.rec_base + (size_t)A.rec_slot[q] * A.rec_bytes))[1] = NXSGPU_REC_INEXACT;
		}
		return;
	}
	if (LDS_HEAP) {
		cap = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(A.k, (uint32_t)WAVE));
	} else if (HEAP == HEAP_LDS) {
		cap = (uint32_t)min((uint64_t)A.k, A.heap_off[q + 1] - A.heap_off[q]);
	} else {
		hs = A.gheap_s + A.heap_off[q];
		hd = A.gheap_d + A.heap_off[q];
		cap = (uint32_t)min((uint64_t)A.k, A.heap_off[q + 1] - A.heap_off[q]);
	}
	if (lane == 0) {
		s_n = 0;
		s_min = 0.0f;
	}
	__syncthreads();

	/* candidates: groups in descending doc range, each already descending.
	 * Segment counts are fetched 64 at a time (one lane each) and empty
	 * segments -- most of them, once thresholds have warmed up -- are skipped
	 * without a memory round trip. */
	const qmeta_t qm = A.qmeta[q];
	for (int g0 = (int)qm.n_groups; g0 > 0 && cap; g0 -= WAVE) {
		const int gi = g0 - 1 - (int)lane;
		uint32_t cnt_l = 0;
		uint64_t sb_l = 0;
		if (gi >= 0) {
			const uint64_t seg_l = (uint64_t)qm.seg_first + gi;
			if (A.seg_cap) {
				cnt_l = A.seg_count[seg_l];
				sb_l = seg_l * A.seg_cap;
			} else {
				sb_l = A.seg_off[seg_l];
				cnt_l = (uint32_t)(A.seg_off[seg_l + 1] - sb_l);
			}
		}
		/*
		 * The candidates of these 64 segments, in feed order (segment lane
		 * ascending = doc range descending, then position), are packed 64
		 * to a load by a prefix sum over the counts; RU chunks are in flight.
		 */
		uint32_t incl = cnt_l;
		for (int o = 1; o < WAVE; o <<= 1) {
			const uint32_t v = (uint32_t)__shfl_up((int)incl, o);
			if (lane >= (unsigned)o) {
				incl += v;
			}
		}
		const uint32_t total = (uint32_t)__shfl((int)incl, WAVE - 1);
		RSTAT(n_cand += total; if (rt1 == rt0) rt1 = __builtin_amdgcn_s_memrealtime();)
		constexpr int RU = 4;
		/* heap.c:68-74 on one round of RU x 64 candidates (valid: index < bound) */
		auto consume = [&](uint32_t c0, uint32_t bound, const float (&scv)[RU], const uint32_t (&dcv)[RU]) {
#pragma unroll
			for (int u = 0; u < RU; u++) {
				const bool valid = c0 + u * WAVE + lane < bound;
				const float sc = scv[u];
				const uint32_t dc = dcv[u];
				/* heap.c:68-74: when full, an item <= the root is dropped
				 * without touching the heap */
				if constexpr (LDS_HEAP) {
					uint64_t pend = ballot64(valid && (rn < cap || sc > rmin));
					while (pend) {
						const int L = __builtin_ctzll(pend);
						const float v = rh_gets(sc, (uint32_t)L);
						const uint32_t dv = rh_getd(dc, (uint32_t)L);
						rheap_add(rhs, rhd, rn, cap, v, dv);
						RSTAT(n_ins++;)
						rn = (uint32_t)__builtin_amdgcn_readfirstlane((int)rn);
						if (A.log_cnt && lane == 0) {
							const uint32_t row = A.log_slot ? A.log_slot[q] : q;
							const uint32_t nl = A.log_cnt[row];
							if (nl < A.log_cap) {
								A.log_ids[(uint64_t)row * A.log_cap + nl] = A.doc_ids[dv];
								A.log_sc[(uint64_t)row * A.log_cap + nl] = v;
							}
							A.log_cnt[row] = nl + 1;
						}
						rmin = rh_gets(rhs, 0);
						pend &= pend - 1;
						pend &= ballot64(valid && (rn < cap || sc > rmin));
					}
					continue;
				}
				uint32_t nn = s_n;
				float mn = s_min;
				uint64_t pend = ballot64(valid && (nn < cap || sc > mn));
				while (pend) {
					const int L = __ffsll((long long)pend) - 1;
					const float v = __shfl(sc, L);
					const uint32_t dv = (uint32_t)__shfl((int)dc, L);
					if (lane == 0) {
						uint32_t cnt = s_n;
						if constexpr (HEAP == HEAP_LDS) {
							lheap_add(dyn_heap, cnt, cap, v, dv);
						} else {
							heap_add(hs, hd, &cnt, cap, v, dv);
						}
						if (A.log_cnt) {
							const uint32_t row = A.log_slot ? A.log_slot[q] : q;
							const uint32_t nl = A.log_cnt[row];
							if (nl < A.log_cap) {
								A.log_ids[(uint64_t)row * A.log_cap + nl] = A.doc_ids[dv];
								A.log_sc[(uint64_t)row * A.log_cap + nl] = v;
							}
							A.log_cnt[row] = nl + 1;
						}
						s_n = cnt;
						s_min = HEAP == HEAP_LDS ? __uint_as_float(dyn_heap[0].x) : hs[0];
					}
					__syncthreads();
					nn = s_n;
					mn = s_min;
					pend &= pend - 1;
					pend &= ballot64(valid && (nn < cap || sc > mn));
				}
			}
		};
		if constexpr (!LDS_HEAP) {
			/*
			 * The exact passes (limit > 64) stream EVERY match of the query through
			 * here, thousands per segment: segment by segment, plain coalesced loads.
			 * (Packing the candidates of 64 segments by a prefix sum -- below, what
			 * the top-k pass needs for its many near-empty segments -- costs a 6-step
			 * cross-lane search per 64 candidates: 28 of the 30 ms of a default-limit
			 * batch.)
			 */
			uint64_t nonempty = ballot64(cnt_l != 0);
			while (nonempty) {
				const int sl = __builtin_ctzll(nonempty);
				nonempty &= nonempty - 1;
				const uint32_t scnt = (uint32_t)__builtin_amdgcn_readlane((int)cnt_l, sl);
				const uint64_t ssb = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(sb_l >> 32), sl) << 32) |
				    (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)sb_l, sl);
				for (uint32_t c0 = 0; c0 < scnt; c0 += WAVE * RU) {
					float scv[RU];
					uint32_t dcv[RU];
#pragma unroll
					for (int u = 0; u < RU; u++) {
						const uint32_t j = c0 + u * WAVE + lane;
						scv[u] = 0.0f;
						dcv[u] = 0;
						if (j < scnt) {
							const uint64_t at = ssb + j;
							scv[u] = A.cand_sc[at];
							dcv[u] = A.cand_doc ? A.cand_doc[at] : (uint32_t)at;
						}
					}
					consume(c0, scnt, scv, dcv);
				}
			}
			continue;
		}
		for (uint32_t c0 = 0; c0 < total; c0 += WAVE * RU) {
			float scv[RU];
			uint32_t dcv[RU];
#pragma unroll
			for (int u = 0; u < RU; u++) {
				const uint32_t j = c0 + u * WAVE + lane;
				scv[u] = 0.0f;
				dcv[u] = 0;
				/* segment lane sl = first lane with incl > j (binary search
				 * over the lanes' inclusive sums) */
				uint32_t sl = 0;
#pragma unroll
				for (int step = 32; step >= 1; step >>= 1) {
					const uint32_t probe = (uint32_t)__shfl((int)incl, (int)(sl + step - 1));
					if (probe <= j) {
						sl += step;
					}
				}
				sl = min(sl, (uint32_t)WAVE - 1);
				const uint32_t s_incl = (uint32_t)__shfl((int)incl, (int)sl);
				const uint32_t s_cnt = (uint32_t)__shfl((int)cnt_l, (int)sl);
				const uint64_t s_sb = ((uint64_t)(uint32_t)__shfl((int)(sb_l >> 32), (int)sl) << 32) |
				    (uint32_t)__shfl((int)(uint32_t)sb_l, (int)sl);
				if (j < total) {
					const uint64_t at = s_sb + (j - (s_incl - s_cnt));
					scv[u] = A.cand_sc[at];
					dcv[u] = A.cand_doc ? A.cand_doc[at] : (uint32_t)at;
				}
			}
			consume(c0, total, scv, dcv);
		}
	}
	__syncthreads();

	if constexpr (LDS_HEAP) {
		RSTAT(rt2 = __builtin_amdgcn_s_memrealtime();)
		/* heap_sort (heap.c:197-221): repeated remove-min, placed from the back */
		const uint32_t cnt = rn;
		uint32_t n = rn;
		while (n) {
			const uint32_t last = n - 1;
			float ms; uint32_t mdoc;
			rheap_remove_min(rhs, rhd, n, ms, mdoc);
			n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
			rh_set(rhs, rhd, last, ms, mdoc);
		}
		/* lane i holds result i */
		if (A.rec_base) {
			/* u32 count | u32 flags | u64 ids[k] | f32 scores[k]  (nxs_gpu.h) */
			uint8_t *rec = A.rec_base + (size_t)A.rec_slot[q] * A.rec_bytes;
			uint64_t *r_ids = (uint64_t *)(rec + 8);
			float *r_sc = (float *)(rec + 8 + 8 * (size_t)A.k);
			RSTAT(rt3 = __builtin_amdgcn_s_memrealtime();)
			if (lane < cnt) {
				r_ids[lane] = A.doc_ids[rhd];
				r_sc[lane] = rhs;
			}
			if (lane == 0) {
				((uint32_t *)rec)[0] = cnt;
			}
#ifdef NXS_STATS
			if (lane == 0) {
				const unsigned long long rt4 = __builtin_amdgcn_s_memrealtime();
				atomicAdd(&g_rstats[0], 1ull);
				atomicAdd(&g_rstats[1], rt1 - rt0);
				atomicAdd(&g_rstats[2], rt2 - rt1);
				atomicAdd(&g_rstats[3], rt3 - rt2);
				atomicAdd(&g_rstats[4], rt4 - rt3);
				atomicAdd(&g_rstats[5], n_cand);
				atomicAdd(&g_rstats[6], n_ins);
			}
#endif
			return;
		}
		const uint64_t ob = A.out_off ? A.out_off[q] : (uint64_t)q * A.k;
		if (lane < cnt) {
			A.out_ids[ob + lane] = A.doc_ids[rhd];
			A.out_sc[ob + lane] = rhs;
		}
		if (lane == 0) {
			A.out_count[q] = cnt;
		}
		return;
	}
	/* heap_sort (heap.c:197-221): repeated remove-min, placed from the back */
	const uint32_t cnt = s_n;
	if constexpr (HEAP == HEAP_LDS) {
		if (lane == 0) {
			uint32_t n = cnt;
			while (n) {
				const uint32_t last = n - 1;
				float ms; uint32_t mdoc;
				lheap_remove_min(dyn_heap, n, ms, mdoc);
				dyn_heap[last] = make_uint2(__float_as_uint(ms), mdoc);
			}
		}
		__syncthreads();
		if (A.rec_base) {
			uint8_t *rec = A.rec_base + (size_t)A.rec_slot[q] * A.rec_bytes;
			uint64_t *r_ids = (uint64_t *)(rec + 8);
			float *r_sc = (float *)(rec + 8 + 8 * (size_t)A.k);
			for (uint32_t i = lane; i < cnt; i += WAVE) {
				r_ids[i] = A.doc_ids[dyn_heap[i].y];
				r_sc[i] = __uint_as_float(dyn_heap[i].x);
			}
			if (lane == 0) {
				((uint32_t *)rec)[0] = cnt;
			}
			return;
		}
		const uint64_t ob2 = A.out_off ? A.out_off[q] : (uint64_t)q * A.k;
		for (uint32_t i = lane; i < cnt; i += WAVE) {
			A.out_ids[ob2 + i] = A.doc_ids[dyn_heap[i].y];
			A.out_sc[ob2 + i] = __uint_as_float(dyn_heap[i].x);
		}
		if (lane == 0) {
			A.out_count[q] = cnt;
		}
		return;
	}
	if (lane == 0) {
		uint32_t n = cnt;
		while (n) {
			const uint32_t last = n - 1;
			float ms; uint32_t mdoc;
			heap_remove_min(hs, hd, &n, &ms, &mdoc);
			hs[last] = ms;
			hd[last] = mdoc;
		}
	}
	__syncthreads();
	if (A.rec_base) {
		/* u32 count | u32 flags | u64 ids[k] | f32 scores[k]  (nxs_gpu.h) */
		uint8_t *rec = A.rec_base + (size_t)A.rec_slot[q] * A.rec_bytes;
		uint64_t *r_ids = (uint64_t *)(rec + 8);
		float *r_sc = (float *)(rec + 8 + 8 * (size_t)A.k);
		for (uint32_t i = lane; i < cnt; i += WAVE) {
			r_ids[i] = A.doc_ids[hd[i]];
			r_sc[i] = hs[i];
		}
		if (lane == 0) {
			((uint32_t *)rec)[0] = cnt;
		}
		return;
	}
	const uint64_t ob = A.out_off ? A.out_off[q] : (uint64_t)q * A.k;
	for (uint32_t i = lane; i < cnt; i += WAVE) {
		A.out_ids[ob + i] = A.doc_ids[hd[i]];
		A.out_sc[ob + i] = hs[i];
	}
	if (lane == 0) {
		A.out_count[q] = cnt;
	}
}

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
/* fuzzy: BK-tree BFS on the device                                    */
/* ------------------------------------------------------------------ */

struct fz_item_t { uint32_t tok, node; };

struct fz_args_t {
	const nxsgpu_bknode_t *	bk;
	const uint8_t *		bk_bytes;
	const uint8_t *		tok_bytes;
	const uint32_t *	tok_off;
	const uint64_t *	peq;		/* [n_tok][256] */
	const fz_item_t *	cur;
	fz_item_t *		next;
	const uint32_t *	cur_count;
	uint32_t *		next_count;
	uint32_t		cap;
	uint32_t *		best;		/* [n_tok] min BFS index of a usable match */
	unsigned long long *	visited;	/* [n_tok] or NULL */
	uint16_t *		dp_rows;	/* scratch for tokens > 64 bytes */
	uint32_t		dp_stride;
	uint32_t *		overflow;
	uint32_t		prune;		/* drop (token, node) pairs that can no longer win */
	unsigned long long *	evals;		/* distance evaluations (profiling) or NULL */
};

/* distance between token `tok` and the node's term */
__device__ static inline int
fz_distance(const fz_args_t &A, uint32_t tok, const nxsgpu_bknode_t &nd, uint64_t slot)
{
	const uint32_t qoff = A.tok_off[tok], m = A.tok_off[tok + 1] - qoff;
	const uint32_t n = nd.str_len;

	if (m == 0) {
		return (int)n;
	}
	if (m <= NXS_MYERS_MAXPAT) {
		/* Myers bit-vector, pattern = query token */
		const uint64_t *peq = A.peq + (uint64_t)tok * 256;
		nxs_myers_t s;
		nxs_myers_init(&s, m);
		/*
		 * Eight term bytes at a time: their Peq words are eight independent
		 * gathers issued together (one L2 round trip), then the dependent
		 * bit-vector steps.  Fetched inside the step loop they cost one round
		 * trip per byte -- the kernel was bound by exactly that latency.  Bytes
		 * past the term's end index a valid table row and are not stepped.
		 */
		uint64_t w;
		memcpy(&w, nd.inl, 8);
		const uint8_t *rest = A.bk_bytes + nd.str_off;
		for (uint32_t i0 = 0; i0 < n; i0 += 8) {
			uint64_t e[8];
			if (i0) {
				/* (the byte pool carries 16 bytes of slack behind its end) */
				uint32_t lo32, hi32;
				__builtin_memcpy(&lo32, rest + i0, 4);
				__builtin_memcpy(&hi32, rest + i0 + 4, 4);
				w = (uint64_t)lo32 | ((uint64_t)hi32 << 32);
			}
#pragma unroll
			for (int i = 0; i < 8; i++) {
				e[i] = peq[(w >> (8 * i)) & 0xff];
			}
#pragma unroll
			for (int i = 0; i < 8; i++) {
				if (i0 + i < n) {
					nxs_myers_step(&s, e[i]);
				}
			}
		}
		return s.score;
	}
	/* long token: row DP (levdist.c:67-150) in global scratch */
	{
		const uint8_t *a = A.tok_bytes + qoff;		/* length m */
		const uint8_t *b = A.bk_bytes + nd.str_off;	/* length n */
		uint16_t *row = A.dp_rows + slot * A.dp_stride;
		uint32_t la = m, lb = n;
		if (la < lb) {
			const uint8_t *t = a; a = b; b = t;
			const uint32_t tl = la; la = lb; lb = tl;
		}
		if (lb == 0) {
			return (int)la;
		}
		for (uint32_t j = 0; j <= lb; j++) {
			row[j] = (uint16_t)j;
		}
		for (uint32_t i = 0; i < la; i++) {
			uint32_t diag = i, above;
			row[0] = (uint16_t)(i + 1);
			for (uint32_t j = 1; j <= lb; j++) {
				above = row[j];
				uint32_t v = diag + (a[i] != b[j - 1]);
				v = min(v, (uint32_t)row[j - 1] + 1);
				v = min(v, above + 1);
				row[j] = (uint16_t)v;
				diag = above;
			}
		}
		return (int)row[lb];
	}
}

template <bool LONG>
__global__ void
k_bk_level(const fz_args_t A)
{
	__shared__ uint32_t s_wtot[16], s_base;
	const unsigned lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
	const uint32_t count = min(*A.cur_count, A.cap);
	const uint32_t nthreads = gridDim.x * blockDim.x;
	const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t rounds = (count + nthreads - 1) / nthreads;
	uint32_t n_eval = 0;

	for (uint32_t r = 0; r < rounds; r++) {
		const uint32_t i = r * nthreads + tid;
		uint32_t nkids = 0, first = 0, tok = 0;
		uint64_t bm = 0, full = 0;

		if (i < count) {
			const fz_item_t it = A.cur[i];
			const uint32_t m = A.tok_off[it.tok + 1] - A.tok_off[it.tok];
			tok = it.tok;
			/*
			 * Exact pruning.  The answer is the match of LOWEST BFS rank
			 * (idxterm.c:238-242: first pushed with total > 0; Q7), a node's
			 * descendants all have higher ranks than the node itself (BFS
			 * numbering), and best[tok] only ever decreases.  So once a usable
			 * match of rank r is known, a pair whose node has rank > r can
			 * neither be nor lead to the winner: it is dropped without a distance
			 * computation and without children.  Any value read here -- stale or
			 * written by a concurrent lane of this very level -- is the rank of a
			 * real match, hence a valid bound.  (Off when the caller wants the
			 * reference's visit counts.)
			 */
			const bool dead = A.prune && __hip_atomic_load(&A.best[it.tok], __ATOMIC_RELAXED,
			    __HIP_MEMORY_SCOPE_AGENT) < it.node;
			if (!dead && LONG == (m > NXS_MYERS_MAXPAT)) {
				const nxsgpu_bknode_t nd = A.bk[it.node];
				const int d = fz_distance(A, it.tok, nd, tid);
				n_eval++;
				if (A.visited) {
					atomicAdd(&A.visited[it.tok], 1ull);
				}
				/* match: bktree.c:252-254; winner = first pushed with
				 * total > 0 (idxterm.c:238-242) = min BFS index */
				if (d <= 2 && (nd.flags & 1)) {
					atomicMin(&A.best[it.tok], it.node);
				}
				/* children in slots [max(d-2,0), min(d+2,63)):
				 * bktree.c:150-156,260-264 (x86 shift semantics) */
				const unsigned min_d = d > 2 ? (unsigned)d - 2 : 0;
				const unsigned max_d = min((unsigned)d + 2, 63u);
				const uint64_t lo_mask = ~0ull << (min_d & 63);
				const uint64_t hi_mask = ~0ull >> ((64 - max_d) & 63);
				full = nd.bitmap;
				bm = full & lo_mask & hi_mask;
				nkids = __popcll(bm);
				first = nd.first_child;
			}
		}
		/* wave-level inclusive scan of nkids, one atomic per wavefront */
		uint32_t incl = nkids;
		for (int o = 1; o < WAVE; o <<= 1) {
			const uint32_t v = __shfl_up((int)incl, o);
			if (lane >= (unsigned)o) incl += v;
		}
		const uint32_t total = __shfl((int)incl, WAVE - 1);
		/*
		 * One returning atomic per WORKGROUP and round, not per wavefront: a
		 * single counter word takes ~88 M atomics/s (MI355X_MICROARCH.md,
		 * `dequeue`), and with one per 64 candidates that ceiling -- not memory,
		 * not the DP -- was the 5.4 G candidates/s this kernel ran at.  `rounds`
		 * is uniform over the grid, so every wavefront reaches the barriers.
		 */
		s_wtot[wid] = total;		/* (all lanes write the same value) */
		__syncthreads();
		if (threadIdx.x == 0) {
			uint32_t sum = 0;
			for (unsigned w = 0; w < nw; w++) {
				const uint32_t tw = s_wtot[w];
				s_wtot[w] = sum;
				sum += tw;
			}
			s_base = sum ? atomicAdd(A.next_count, sum) : 0;
		}
		__syncthreads();
		const uint32_t wbase = s_base + s_wtot[wid];
		__syncthreads();		/* s_wtot is rewritten next round */
		uint32_t o = wbase + incl - nkids;
		/* ascending slot order = the order bktree_search pushes children */
		while (bm) {
			const int slot = __ffsll((long long)bm) - 1;
			bm &= bm - 1;
			const uint32_t child = first + __popcll(full & ((1ull << slot) - 1));
			if (o < A.cap) {
				fz_item_t ni;
				ni.tok = tok;
				ni.node = child;
				A.next[o] = ni;
			} else {
				*A.overflow = 1;
			}
			o++;
		}
	}
	if (A.evals) {
		for (int o = 32; o; o >>= 1) {
			n_eval += (uint32_t)__shfl_xor((int)n_eval, o);
		}
		if (lane == 0 && n_eval) {
			atomicAdd(A.evals, (unsigned long long)n_eval);
		}
	}
}

__global__ void
k_bk_seed(fz_item_t *items, uint32_t *count0, uint32_t n_tok, uint32_t *best,
    unsigned long long *visited)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n_tok) {
		fz_item_t it;
		it.tok = i;
		it.node = 0;
		items[i] = it;
		best[i] = 0xffffffffu;
		if (visited) {
			visited[i] = 0;
		}
	}
	if (i == 0) {
		*count0 = n_tok;
	}
}

__global__ void
k_bk_peq(const uint8_t *tok_bytes, const uint32_t *tok_off, uint32_t n_tok, uint64_t *peq, uint2 *tokf,
    const uint32_t *tok_rank)
{
	const uint32_t tok = blockIdx.x;
	const uint32_t off = tok_off[tok], m = tok_off[tok + 1] - off;
	if (tokf && threadIdx.x == 0) {
		/* what k_fz_filter compares: the token's byte set, at the token's place in
		 * the length-sorted order */
		uint32_t sg = 0;
		for (uint32_t j = 0; j < m; j++) {
			sg |= 1u << (tok_bytes[off + j] & 31);
		}
		tokf[tok_rank[tok]] = make_uint2(sg, tok);
	}
	for (uint32_t c = threadIdx.x; c < 256; c += blockDim.x) {
		uint64_t bits = 0;
		if (m <= NXS_MYERS_MAXPAT) {
			for (uint32_t j = 0; j < m; j++) {
				if (tok_bytes[off + j] == c) {
					bits |= 1ull << j;
				}
			}
		}
		peq[(uint64_t)tok * 256 + c] = bits;
	}
}

__global__ void
k_bk_finish(const nxsgpu_bknode_t *bk, const uint32_t *best, uint32_t n_tok, uint32_t *term_ids)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n_tok) {
		const uint32_t b = best[i];
		term_ids[i] = (b == 0xffffffffu) ? 0 : bk[b].term_id;
	}
}

/* ---- match-first fuzzy search ---------------------------------------- */
/*
 * The level-by-level search above spends its time on the frontier: the winner
 * sits 9-12 levels deep and everything above it has to be expanded (65 M pairs
 * for 1024 tokens over a 1M-term tree).  But the winner has a closed form:
 *
 *   the node of LOWEST BFS rank among those that (1) are a match -- distance
 *   <= 2, on-disk total > 0 -- and (2) bktree_search reaches: at EVERY ancestor
 *   a the slot of the path's child lies in [max(d(q,a)-2, 0), min(d(q,a)+2, 63))
 *   (bktree.c:150-156,260-264; Q8: the range is half-open, so a match is not
 *   always reached).
 *
 * (1) needs no tree: all (token, term) pairs are screened with a necessary
 * condition -- |len difference| <= 2 and, on the sets of bytes the strings
 * contain (hashed to 64 bits), at most 2 bytes on either side that the other
 * string lacks: an edit removes at most one such byte per side -- 0.03-0.6 % of
 * the pairs survive on the synthetic vocabulary and take the exact bit-vector
 * distance.  (2) walks the few real matches up to the root.  The three steps
 * are three launches over flat queues; their result is the same min-rank node
 * (tests and bench.py compare against the level-by-level search with and
 * without pruning, and against the oracle).
 */
#define	FZ_NOPARENT	0xffffffffu
#define	FZF_BUF		192		/* survivors a wavefront stages in LDS */
#define	FZ_MAXLEN	(NXS_MYERS_MAXPAT + 2)	/* longest term that can be within 2 of a token */
#define	FZ_NQ		64		/* survivor sub-queues: a single counter word takes ~88 M atomics/s */
#define	FZ_CSTRIDE	16		/* their counters, one per 64 bytes */

/* per node: its parent and the slot it hangs in */
__global__ void __launch_bounds__(256)
k_bk_aux(const nxsgpu_bknode_t *bk, uint32_t n, uint32_t *parent, uint8_t *slot)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) {
		return;
	}
	if (i == 0) {
		parent[0] = FZ_NOPARENT;
		slot[0] = 0;
	}
	uint64_t bm = bk[i].bitmap;
	uint32_t c = bk[i].first_child;
	while (bm) {
		const int sl = __ffsll((long long)bm) - 1;
		bm &= bm - 1;
		parent[c] = i;
		slot[c] = (uint8_t)sl;
		c++;
	}
}

/* per candidate (the nodes that can win, sorted by term length on the host): the
 * set of bytes its term contains, hashed to 32 bits, and the length */
__global__ void __launch_bounds__(256)
k_fz_sigs(const nxsgpu_bknode_t *bk, const uint8_t *bytes, const uint32_t *cand_node, uint32_t n_c,
    uint32_t *sig, uint8_t *len8)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_c) {
		return;
	}
	const nxsgpu_bknode_t nd = bk[cand_node[i]];
	const uint8_t *str = bytes + nd.str_off;
	uint32_t sg = 0;
	for (uint32_t j = 0; j < nd.str_len; j++) {
		sg |= 1u << (str[j] & 31);
	}
	sig[i] = sg;
	len8[i] = (uint8_t)nd.str_len;
}

/*
 * Screen.  lane = one candidate node (sorted by length: a workgroup's 256 terms
 * span lengths [Lmin, Lmax]), loop = the tokens of length Lmin-2 .. Lmax+2
 * (tokens sorted by length too; their features are wave-uniform and come
 * through the scalar unit, four tokens per round).  grid.y slices the token
 * range.  A pair that passes wrongly (|length difference| = 3 across a length
 * boundary of the workgroup, hash collisions) is dropped by the exact distance.
 */
__global__ void __launch_bounds__(256)
k_fz_filter(const uint32_t *__restrict__ sig, const uint32_t *__restrict__ cand_node,
    const uint8_t *__restrict__ len8, uint32_t n_c, const uint2 *__restrict__ tokf,
    const uint32_t *__restrict__ tok_len_off, fz_item_t *out, uint32_t *out_count, uint32_t qcap,
    uint32_t *overflow)
{
	__shared__ fz_item_t s_buf[4][FZF_BUF];
	/* this workgroup's sub-queue: [sq * qcap, (sq + 1) * qcap) */
	const uint32_t sq = (blockIdx.x + 5 * blockIdx.y) & (FZ_NQ - 1);
	fz_item_t *const sq_out = out + (uint64_t)sq * qcap;
	uint32_t *const sq_count = out_count + sq * FZ_CSTRIDE;
	const unsigned lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	const uint32_t first = blockIdx.x * 256, i = first + threadIdx.x;
	const bool valid = i < n_c;
	const uint32_t ts = valid ? sig[i] : 0, nts = ~ts;
	const uint32_t node = valid ? cand_node[i] : 0;
	const uint64_t vmask = ballot64(valid);
	const uint32_t lmin = len8[first], lmax = len8[min(first + 255, n_c - 1)];
	const uint32_t ta = tok_len_off[lmin > 2 ? lmin - 2 : 0];
	const uint32_t tb = tok_len_off[min(lmax + 2, (uint32_t)NXS_MYERS_MAXPAT) + 1];
	const uint32_t per = (tb - ta + gridDim.y - 1) / gridDim.y;
	const uint32_t t0 = ta + blockIdx.y * per, t1 = min(tb, t0 + per);
	uint32_t nb = 0;

	auto flush = [&]() {
		uint32_t base = 0;
		if (lane == 0) {
			base = atomicAdd(sq_count, nb);
		}
		base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
		WAVE_SYNC();
		for (uint32_t e = lane; e < nb; e += WAVE) {
			if (base + e < qcap) {
				sq_out[base + e] = s_buf[wid][e];
			} else {
				*overflow = 1;
			}
		}
		WAVE_SYNC();
		nb = 0;
	};
	auto check = [&](const uint2 f, bool in_range) {
		/* bytes only the term has / only the token has: an edit removes at
		 * most one of each */
		const uint32_t a = __popc(ts & ~f.x), b = __popc(nts & f.x);
		const uint64_t m = in_range ? (ballot64(max(a, b) <= 2u) & vmask) : 0ull;
		if (m) {
			if (lane_of(m)) {
				fz_item_t it;
				it.tok = f.y;
				it.node = node;
				s_buf[wid][nb + lanes_below(m)] = it;
			}
			nb += __popcll(m);
			if (nb > FZF_BUF - WAVE) {
				flush();
			}
		}
	};
	for (uint32_t j = t0; j < t1; j += 4) {
		/* (the array carries four entries of slack behind its end) */
		const uint2 f0 = tokf[j], f1 = tokf[j + 1], f2 = tokf[j + 2], f3 = tokf[j + 3];
		check(f0, true);
		check(f1, j + 1 < t1);
		check(f2, j + 2 < t1);
		check(f3, j + 3 < t1);
	}
	if (nb) {
		flush();
	}
}

/*
 * Exact distance of the screened pairs (grid.y = sub-queue).  A match at
 * distance <= 1 is always reached: every node of a child's subtree is at the
 * child's slot distance s from the ancestor a (slot-63 subtrees, which no search
 * ever enters, are not candidates), so |d(q,a) - s| <= 1 and s lies inside
 * [d(q,a)-2, d(q,a)+2) at every ancestor -- it lowers best[] right here.  A
 * match at distance 2 misses exactly when some ancestor has d(q,a) = s - 2: it
 * goes to the next queue for the walk (one returning atomic per workgroup and
 * round, as in k_bk_level).
 */
__global__ void __launch_bounds__(1024)
k_fz_dist(const fz_args_t A, const fz_item_t *cand, const uint32_t *cand_count, uint32_t qcap, fz_item_t *match,
    uint32_t *match_count, uint32_t mcap)
{
	__shared__ uint32_t s_wtot[16], s_base;
	const unsigned lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
	const uint32_t count = min(cand_count[blockIdx.y * FZ_CSTRIDE], qcap);
	const uint32_t nthreads = gridDim.x * blockDim.x;
	const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t rounds = (count + nthreads - 1) / nthreads;
	uint32_t n_eval = 0;

	cand += (uint64_t)blockIdx.y * qcap;
	for (uint32_t r = 0; r < rounds; r++) {
		const uint32_t i = r * nthreads + tid;
		fz_item_t it;
		bool hit = false;

		it.tok = it.node = 0;
		if (i < count) {
			it = cand[i];
			const nxsgpu_bknode_t nd = A.bk[it.node];
			const int d = fz_distance(A, it.tok, nd, 0);
			n_eval++;
			if (d <= 1) {
				atomicMin(&A.best[it.tok], it.node);
			} else if (d == 2) {	/* bktree.c:252-254 */
				hit = __hip_atomic_load(&A.best[it.tok], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > it.node;
			}
		}
		const uint64_t m = ballot64(hit);
		s_wtot[wid] = __popcll(m);
		__syncthreads();
		if (threadIdx.x == 0) {
			uint32_t sum = 0;
			for (unsigned w = 0; w < nw; w++) {
				const uint32_t tw = s_wtot[w];
				s_wtot[w] = sum;
				sum += tw;
			}
			s_base = sum ? atomicAdd(match_count, sum) : 0;
		}
		__syncthreads();
		const uint32_t o = s_base + s_wtot[wid] + lanes_below(m);
		__syncthreads();
		if (hit) {
			if (o < mcap) {
				match[o] = it;
			} else {
				*A.overflow = 1;
			}
		}
	}
	if (A.evals) {
		for (int o = 32; o; o >>= 1) {
			n_eval += (uint32_t)__shfl_xor((int)n_eval, o);
		}
		if (lane == 0 && n_eval) {
			atomicAdd(A.evals, (unsigned long long)n_eval);
		}
	}
}

/* does bktree_search reach the match?  Walk to the root; every ancestor's child
 * range must hold the slot the path leaves it through. */
__global__ void __launch_bounds__(256)
k_fz_chain(const fz_args_t A, const uint32_t *__restrict__ parent, const uint8_t *__restrict__ slot,
    const fz_item_t *match, const uint32_t *match_count, uint32_t mcap)
{
	const unsigned lane = threadIdx.x & 63;
	const uint32_t count = min(*match_count, mcap);
	const uint32_t nthreads = gridDim.x * blockDim.x;
	uint32_t n_eval = 0;

	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += nthreads) {
		const fz_item_t it = match[i];
		uint32_t c = it.node;
		bool ok = true;

		for (;;) {
			/* (a match of lower rank is known: this one cannot win -- any value
			 * read is the rank of a reachable match) */
			if (__hip_atomic_load(&A.best[it.tok], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < it.node) {
				ok = false;
				break;
			}
			const uint32_t p = parent[c];
			if (p == FZ_NOPARENT) {
				break;
			}
			const unsigned sl = slot[c];
			const nxsgpu_bknode_t nd = A.bk[p];
			const int d = fz_distance(A, it.tok, nd, 0);
			n_eval++;
			/* bktree.c:150-156,260-264 (x86 shift semantics), as k_bk_level */
			const unsigned min_d = d > 2 ? (unsigned)d - 2 : 0;
			const unsigned max_d = min((unsigned)d + 2, 63u);
			const uint64_t lo_mask = ~0ull << (min_d & 63);
			const uint64_t hi_mask = ~0ull >> ((64 - max_d) & 63);
			if (!(((lo_mask & hi_mask) >> sl) & 1)) {
				ok = false;
				break;
			}
			c = p;
		}
		if (ok) {
			atomicMin(&A.best[it.tok], it.node);
		}
	}
	if (A.evals) {
		for (int o = 32; o; o >>= 1) {
			n_eval += (uint32_t)__shfl_xor((int)n_eval, o);
		}
		if (lane == 0 && n_eval) {
			atomicAdd(A.evals, (unsigned long long)n_eval);
		}
	}
}

/* ------------------------------------------------------------------ */
/* host side of the shim                                               */
/* ------------------------------------------------------------------ */

static void
bk_aux_free(nxsgpu_index_t *ix)
{
	(void)hipFree(ix->d_bk_parent);
	(void)hipFree(ix->d_bk_slot);
	(void)hipFree(ix->d_fz_node);
	(void)hipFree(ix->d_fz_sig);
	(void)hipFree(ix->d_fz_len);
	ix->d_bk_parent = NULL;
	ix->d_bk_slot = NULL;
	ix->d_fz_node = NULL;
	ix->d_fz_sig = NULL;
	ix->d_fz_len = NULL;
	ix->n_fz = 0;
}

/* the match-first search's view of the tree (after d_bk / d_bk_bytes are in place) */
static int
bk_aux_build(nxsgpu_index_t *ix, const nxsgpu_bknode_t *nodes, uint32_t n)
{
	bk_aux_free(ix);
	if (n == 0) {
		return 0;
	}
	/* candidates: nodes with postings on disk whose term can be within 2 of a
	 * token of <= 64 bytes; counting sort by length, BFS rank inside a length */
	std::vector<uint32_t> start(FZ_MAXLEN + 2, 0), perm;
	/* (a child in slot 63 is never visited -- the range's upper end is at most 63,
	 * exclusive: bktree.c:150-156 -- and neither is anything below it; BFS
	 * numbering: a parent precedes its children) */
	std::vector<uint8_t> cut(n, 0);
	uint32_t n_c = 0;
	for (uint32_t i = 0; i < n; i++) {
		uint64_t bm = nodes[i].bitmap;
		uint32_t c = nodes[i].first_child;
		while (bm) {
			const int sl = __builtin_ctzll(bm);
			bm &= bm - 1;
			if (c < n) {
				cut[c] = cut[i] | (sl >= 63);
			}
			c++;
		}
	}
	auto is_cand = [&](uint32_t i) { return (nodes[i].flags & 1) && nodes[i].str_len <= FZ_MAXLEN && !cut[i]; };
	for (uint32_t i = 0; i < n; i++) {
		if (is_cand(i)) {
			start[nodes[i].str_len + 1]++;
			n_c++;
		}
	}
	for (uint32_t l = 0; l <= FZ_MAXLEN; l++) {
		start[l + 1] += start[l];
	}
	perm.resize(std::max<uint32_t>(n_c, 1));
	for (uint32_t i = 0; i < n; i++) {
		if (is_cand(i)) {
			perm[start[nodes[i].str_len]++] = i;
		}
	}
	if (hipMalloc(&ix->d_bk_parent, (size_t)n * 4) != hipSuccess ||
	    hipMalloc(&ix->d_bk_slot, (size_t)n + 16) != hipSuccess ||
	    hipMalloc(&ix->d_fz_node, (size_t)std::max<uint32_t>(n_c, 1) * 4) != hipSuccess ||
	    hipMalloc(&ix->d_fz_sig, (size_t)std::max<uint32_t>(n_c, 1) * 4) != hipSuccess ||
	    hipMalloc(&ix->d_fz_len, (size_t)n_c + 16) != hipSuccess ||
	    hipMemcpyAsync(ix->d_fz_node, perm.data(), (size_t)n_c * 4, hipMemcpyHostToDevice, ix->stream_fz) != hipSuccess) {
		bk_aux_free(ix);
		set_error("BK-tree side arrays: out of device memory");
		return -1;
	}
	hipLaunchKernelGGL(k_bk_aux, dim3((n + 255) / 256), dim3(256), 0, ix->stream_fz,
	    ix->d_bk, n, ix->d_bk_parent, ix->d_bk_slot);
	if (n_c) {
		hipLaunchKernelGGL(k_fz_sigs, dim3((n_c + 255) / 256), dim3(256), 0, ix->stream_fz,
		    ix->d_bk, ix->d_bk_bytes, ix->d_fz_node, n_c, ix->d_fz_sig, ix->d_fz_len);
	}
	if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ix->stream_fz) != hipSuccess) {
		bk_aux_free(ix);
		set_error("k_bk_aux failed");
		return -1;
	}
	ix->n_fz = n_c;
	return 0;
}

#define	X_KEEP_MAX	(4ull << 30)
/* buffer `which` of the exact path, at least `need` bytes (NULL: out of memory) */
static void *
xbuf_get(nxsgpu_index_t *ix, int which, size_t need)
{
	if (ix->xbuf_len[which] < need) {
		(void)hipFree(ix->xbuf[which]);
		ix->xbuf[which] = NULL;
		ix->xbuf_len[which] = 0;
		const size_t len = need + need / 8;
		if (hipMalloc(&ix->xbuf[which], len) != hipSuccess) {
			return NULL;
		}
		ix->xbuf_len[which] = len;
	}
	return ix->xbuf[which];
}

/* after the pass: an oversized buffer is not kept */
static void
xbuf_put(nxsgpu_index_t *ix, int which)
{
	if (ix->xbuf_len[which] > X_KEEP_MAX) {
		(void)hipFree(ix->xbuf[which]);
		ix->xbuf[which] = NULL;
		ix->xbuf_len[which] = 0;
	}
}

static bool
ensure_ws(nxsgpu_index_t *ix, size_t need)
{
	if (ix->ws_len >= need) {
		return true;
	}
	if (ix->ws) {
		(void)hipFree(ix->ws);
		ix->ws = NULL;
		ix->ws_len = 0;
	}
	need = (need + (size_t(1) << 20)) & ~((size_t(1) << 20) - 1);
	if (hipMalloc(&ix->ws, need) != hipSuccess) {
		set_error("hipMalloc(%zu) for the query workspace failed", need);
		return false;
	}
	ix->ws_len = need;
	return true;
}

static bool
ensure_pin(nxsgpu_index_t *ix, size_t need)
{
	if (ix->h_pin_len >= need) {
		return true;
	}
	if (ix->h_pin) {
		(void)hipHostFree(ix->h_pin);
		ix->h_pin = NULL;
		ix->h_pin_len = 0;
	}
	need = (need + 65535) & ~(size_t)65535;
	if (hipHostMalloc(&ix->h_pin, need, hipHostMallocDefault) != hipSuccess) {
		set_error("hipHostMalloc(%zu) failed", need);
		return false;
	}
	ix->h_pin_len = need;
	return true;
}

template <typename T>
static T *
carve(uint8_t *&p, size_t n)
{
	uintptr_t a = ((uintptr_t)p + 255) & ~(uintptr_t)255;
	T *r = (T *)a;
	p = (uint8_t *)(a + n * sizeof(T));
	return r;
}

static void delete_worklist(worklist_t *);

extern "C" void
nxsgpu_index_destroy(nxsgpu_index_t *ix)
{
	if (!ix) {
		return;
	}
	(void)hipSetDevice(ix->device);
	if (ix->stream) {
		(void)hipStreamSynchronize(ix->stream);
	}
	(void)hipFree(ix->d_doc_ids);
	(void)hipFree(ix->d_doc_len);
	(void)hipFree(ix->d_post_off);
	(void)hipFree(ix->d_post_dt);
	(void)hipFree(ix->d_post[0]);
	(void)hipFree(ix->d_post[1]);
	(void)hipFree(ix->d_dense_col[0]);
	(void)hipFree(ix->d_dense_col[1]);
	(void)hipFree(ix->d_bk);
	(void)hipFree(ix->d_bk_bytes);
	bk_aux_free(ix);
	(void)hipFree(ix->ws);
	(void)hipFree(ix->fz);
	if (ix->h_pin) {
		(void)hipHostFree(ix->h_pin);
	}
	for (int i = 0; i < 4; i++) {
		if (ix->ev[i]) {
			(void)hipEventDestroy(ix->ev[i]);
		}
	}
	for (int i = 0; i < 2; i++) {
		nxsgpu_index::dev_slot_t &sl = ix->slot[i];
		if (sl.active && sl.ev_done) {
			(void)hipEventSynchronize(sl.ev_done);
		}
		(void)hipFree(sl.ws);
		if (sl.h_stage) {
			(void)hipHostFree(sl.h_stage);
		}
		if (sl.ev_up) (void)hipEventDestroy(sl.ev_up);
		if (sl.ev_done) (void)hipEventDestroy(sl.ev_done);
		if (sl.ev_res) (void)hipEventDestroy(sl.ev_res);
		delete_worklist(sl.wl);
		(void)hipFree(sl.d_blocks);
		if (sl.h_blocks) {
			(void)hipHostFree(sl.h_blocks);
		}
		for (int j = 0; j < 3; j++) {
			if (sl.ev_t[j]) (void)hipEventDestroy(sl.ev_t[j]);
		}
	}
	if (ix->stream_up) {
		(void)hipStreamDestroy(ix->stream_up);
	}
	if (ix->stream_down) {
		(void)hipStreamDestroy(ix->stream_down);
	}
	if (ix->stream_fz) {
		(void)hipStreamSynchronize(ix->stream_fz);
		(void)hipStreamDestroy(ix->stream_fz);
	}
	if (ix->ev_cls) {
		(void)hipEventDestroy(ix->ev_cls);
	}
	if (ix->ev_join) {
		(void)hipEventDestroy(ix->ev_join);
	}
	if (ix->stream2) {
		(void)hipStreamDestroy(ix->stream2);
	}
	if (ix->stream3) {
		(void)hipStreamDestroy(ix->stream3);
	}
	for (int i = 0; i < 3; i++) {
		if (ix->xstream[i]) {
			(void)hipStreamDestroy(ix->xstream[i]);
		}
	}
	(void)hipFree(ix->xbuf[0]);
	(void)hipFree(ix->xbuf[1]);
	if (ix->ev_fork3) {
		(void)hipEventDestroy(ix->ev_fork3);
	}
	if (ix->ev_join3) {
		(void)hipEventDestroy(ix->ev_join3);
	}
	if (ix->stream) {
		(void)hipStreamDestroy(ix->stream);
	}
	delete ix;
}

/*
 * Impacts of every posting from the CSR form (d_post_off, d_post_dt) and the
 * header statistics: host libm tables (the device only does IEEE + - * / on
 * them), one k_impacts_csr pass, the per-term maxima back to the host.  Used by
 * the first build and by every refresh (N, adl and df move every idf).
 */
static int
rebuild_impacts(nxsgpu_index_t *ix)
{
	const uint32_t T = ix->n_terms;
	const uint64_t P = ix->n_post;
	const unsigned long N = ix->hdr_doc_count;
	static const double kk = 1.2f;		/* ranking.c:141 */
	static const double bb = 0.75f;		/* ranking.c:142 */
	std::vector<double> logtf((size_t)ix->max_tf + 2), idf_b((size_t)T + 2, 0.0);
	std::vector<float> idf_t((size_t)T + 2, 0.0f);
	double *d_logtf = NULL, *d_idf_bm25 = NULL;
	float *d_idf_tfidf = NULL;
	uint32_t *d_maximp = NULL;
	double adl = 0.0;
	int rc = -1;

	for (size_t c = 0; c < logtf.size(); c++) {
		logtf[c] = log((double)((int)c + 1));	/* ranking.c:90,168 */
	}
	/* two log() per term: spread over a few host threads (1M terms ~ 40 ms on one) */
	{
		const unsigned nthr = T > 65536 ? 8u : 1u;
		std::vector<std::thread> thr;
		auto work = [&](uint32_t lo, uint32_t hi) {
			for (uint32_t t = lo; t < hi; t++) {
				/* doc-sharded (N4): df of the WHOLE collection, not of this shard */
				const unsigned long df = !ix->df_global.empty() ? ix->df_global[t] :
				    ix->h_post_off[t + 1] - ix->h_post_off[t];
				if (df == 0 || N == 0) {
					continue;
				}
				idf_b[t] = log(((N - df + 0.5) / (df + 0.5)) + 1);	/* ranking.c:172 */
				/* ranking.c:91: f32 division, double log, f32 result */
				float idf = log((double)((float)N / (float)df)) + 1;
				idf_t[t] = idf;
			}
		};
		for (unsigned k = 1; k < nthr; k++) {
			const uint32_t lo = 1 + (uint32_t)((uint64_t)T * k / nthr), hi = 1 + (uint32_t)((uint64_t)T * (k + 1) / nthr);
			thr.emplace_back(work, lo, hi);
		}
		work(1, 1 + (uint32_t)((uint64_t)T / nthr));
		for (auto &th : thr) {
			th.join();
		}
	}
	ix->tfidf_valid = N != 0;
	ix->bm25_valid = false;
	if (N != 0) {
		adl = (double)(ix->hdr_token_count / N);	/* ranking.c:163 */
		ix->bm25_valid = !(adl < 1);
	}
	ix->h_maximp[NXSGPU_BM25].assign((size_t)T + 2, 0.0f);
	ix->h_maximp[NXSGPU_TF_IDF].assign((size_t)T + 2, 0.0f);
	if (P == 0) {
		return 0;
	}
	HIP_TRY(hipMalloc(&d_logtf, logtf.size() * 8));
	HIP_TRY(hipMalloc(&d_idf_bm25, idf_b.size() * 8));
	HIP_TRY(hipMalloc(&d_idf_tfidf, idf_t.size() * 4));
	HIP_TRY(hipMemcpyAsync(d_logtf, logtf.data(), logtf.size() * 8, hipMemcpyHostToDevice, ix->stream));
	HIP_TRY(hipMemcpyAsync(d_idf_bm25, idf_b.data(), idf_b.size() * 8, hipMemcpyHostToDevice, ix->stream));
	HIP_TRY(hipMemcpyAsync(d_idf_tfidf, idf_t.data(), idf_t.size() * 4, hipMemcpyHostToDevice, ix->stream));
	HIP_TRY(hipMalloc(&d_maximp, ((size_t)T + 2) * 4 * 2));
	HIP_TRY(hipMemsetAsync(d_maximp, 0, ((size_t)T + 2) * 4 * 2, ix->stream));
	hipLaunchKernelGGL(k_impacts_csr, dim3(4096), dim3(256), 0, ix->stream,
	    ix->d_post_off, T, ix->d_post_dt, P, ix->d_doc_len, d_logtf, d_idf_bm25,
	    d_idf_tfidf, adl >= 1 ? adl : 1.0, kk, bb,
	    ix->d_post[NXSGPU_BM25], ix->d_post[NXSGPU_TF_IDF],
	    d_maximp, d_maximp + (size_t)T + 2);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(ix->h_maximp[NXSGPU_BM25].data(), d_maximp, ((size_t)T + 2) * 4,
	    hipMemcpyDeviceToHost, ix->stream));
	HIP_TRY(hipMemcpyAsync(ix->h_maximp[NXSGPU_TF_IDF].data(), d_maximp + (size_t)T + 2, ((size_t)T + 2) * 4,
	    hipMemcpyDeviceToHost, ix->stream));
	HIP_TRY(hipStreamSynchronize(ix->stream));
	/* impact columns of the dense terms (at most 64, densest first) */
	{
		std::vector<std::pair<uint64_t, uint32_t>> dn;
		for (uint32_t t = 1; t <= T; t++) {
			const uint64_t df = ix->h_post_off[t + 1] - ix->h_post_off[t];
			if ((double)df > ix->cfg.scanm_dens * (double)ix->n_docs && df >= 1024) {
				dn.push_back(std::make_pair(df, t));
			}
		}
		std::sort(dn.begin(), dn.end(), [](const std::pair<uint64_t, uint32_t> &x, const std::pair<uint64_t, uint32_t> &y) {
			return x.first != y.first ? x.first > y.first : x.second < y.second;
		});
		if (dn.size() > 64) {
			dn.resize(64);
		}
		ix->dense_terms.clear();
		for (auto &e : dn) {
			ix->dense_terms.push_back(e.second);
		}
		std::sort(ix->dense_terms.begin(), ix->dense_terms.end());
		const uint64_t words = (uint64_t)ix->dense_terms.size() * ix->n_docs;
		if (words > ix->dense_cap || (!words && ix->dense_cap)) {
			(void)hipFree(ix->d_dense_col[0]);
			(void)hipFree(ix->d_dense_col[1]);
			ix->d_dense_col[0] = ix->d_dense_col[1] = NULL;
			ix->dense_cap = 0;
			if (words) {
				const uint64_t cap = words + words / 16 + 1024;
				HIP_TRY(hipMalloc((void **)&ix->d_dense_col[0], cap * 4));
				HIP_TRY(hipMalloc((void **)&ix->d_dense_col[1], cap * 4));
				ix->dense_cap = cap;
			}
		}
		if (words) {
			for (int a = 0; a < 2; a++) {
				HIP_TRY(hipMemsetAsync(ix->d_dense_col[a], 0xff, words * 4, ix->stream));
				for (size_t c = 0; c < ix->dense_terms.size(); c++) {
					const uint32_t t = ix->dense_terms[c];
					hipLaunchKernelGGL(k_dense_fill, dim3(1024), dim3(256), 0, ix->stream,
					    ix->d_post[a], ix->h_post_off[t], ix->h_post_off[t + 1],
					    ix->d_dense_col[a] + c * ix->n_docs);
				}
			}
			HIP_TRY(hipGetLastError());
			HIP_TRY(hipStreamSynchronize(ix->stream));
		}
	}
	rc = 0;
fail:
	(void)hipFree(d_logtf);
	(void)hipFree(d_idf_bm25);
	(void)hipFree(d_idf_tfidf);
	(void)hipFree(d_maximp);
	return rc;
}

static void warm_streams(nxsgpu_index_t *);

extern "C" nxsgpu_index_t *
nxsgpu_index_create(int device, const nxsgpu_index_src_t *src)
{
	nxsgpu_index_t *ix = new nxsgpu_index_t();
	const uint64_t D = src->n_docs;
	const uint64_t P = D ? src->pair_base[D] : 0;
	const uint32_t T = src->n_terms;
	uint8_t *d_img = NULL, *d_term_ok = NULL;
	uint64_t *d_blk_off = NULL, *d_pair_base = NULL, *d_vals_in = NULL;
	uint32_t *d_keys_in = NULL, *d_keys = NULL;
	unsigned long long *d_first_bad = NULL;
	unsigned int *d_max_tf = NULL;
	void *d_tmp = NULL;
	size_t tmp_bytes = 0;
	unsigned long long h_first_bad = ~0ull;
	unsigned int h_max_tf = 0;

	ix->device = device;
	cfg_from_env(ix->cfg);
	ix->n_docs = D;
	ix->n_post = P;
	ix->n_terms = T;
	ix->hdr_doc_count = src->hdr_doc_count;
	ix->hdr_token_count = src->hdr_token_count;
	ix->first_bad = ~0ull;
	ix->n_bk = src->n_bk;
	ix->bk_depth = src->bk_depth;
	memset(&ix->prof, 0, sizeof(ix->prof));

	if (D >= (1ull << 32) || P >= (1ull << 40)) {
		set_error("index too large for 32-bit doc ordinals");
		goto fail;
	}
	HIP_TRY(hipSetDevice(device));
	HIP_TRY(hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking));
	for (int i = 0; i < 3; i++) {
		HIP_TRY(hipStreamCreateWithFlags(&ix->xstream[i], hipStreamNonBlocking));
	}
	HIP_TRY(hipStreamCreateWithFlags(&ix->stream2, hipStreamNonBlocking));
	HIP_TRY(hipStreamCreateWithFlags(&ix->stream3, hipStreamNonBlocking));
	HIP_TRY(hipEventCreateWithFlags(&ix->ev_fork3, hipEventDisableTiming));
	HIP_TRY(hipEventCreateWithFlags(&ix->ev_join3, hipEventDisableTiming));
	HIP_TRY(hipStreamCreateWithFlags(&ix->stream_up, hipStreamNonBlocking));
	HIP_TRY(hipStreamCreateWithFlags(&ix->stream_down, hipStreamNonBlocking));
	/* (at the highest stream priority the fuzzy passes finish sooner -- the host
	 * waits 8-10 instead of 27-32 ms per C5 step for them -- but that wait is
	 * hidden behind the device's 38 ms anyway, and the changed timing made one
	 * query per step overflow its candidate lists: plain priority) */
	HIP_TRY(hipStreamCreateWithFlags(&ix->stream_fz, hipStreamNonBlocking));
	for (int i = 0; i < 2; i++) {
		HIP_TRY(hipEventCreateWithFlags(&ix->slot[i].ev_up, hipEventDisableTiming));
		HIP_TRY(hipEventCreateWithFlags(&ix->slot[i].ev_done, hipEventDisableTiming));
		HIP_TRY(hipEventCreateWithFlags(&ix->slot[i].ev_res, hipEventDisableTiming));
		for (int j = 0; j < 3; j++) {
			HIP_TRY(hipEventCreate(&ix->slot[i].ev_t[j]));
		}
	}
	HIP_TRY(hipEventCreateWithFlags(&ix->ev_cls, hipEventDisableTiming));
	HIP_TRY(hipEventCreateWithFlags(&ix->ev_join, hipEventDisableTiming));
	for (int i = 0; i < 4; i++) {
		HIP_TRY(hipEventCreate(&ix->ev[i]));
	}

	/* room for appended docs (N1) without moving the tables */
	ix->cap_docs_ids = ix->cap_docs_len = D + D / 8 + 4096;
	HIP_TRY(hipMalloc(&ix->d_doc_ids, ix->cap_docs_ids * 8));
	HIP_TRY(hipMalloc(&ix->d_doc_len, ix->cap_docs_len * 4));
	HIP_TRY(hipMalloc(&ix->d_post_off, ((size_t)T + 2) * 8));
	HIP_TRY(hipMalloc(&ix->d_post_dt, std::max<uint64_t>(P, 1) * 8));
	HIP_TRY(hipMalloc(&ix->d_post[0], std::max<uint64_t>(P, 1) * sizeof(posting_t)));
	HIP_TRY(hipMalloc(&ix->d_post[1], std::max<uint64_t>(P, 1) * sizeof(posting_t)));
	ix->h_post_off.assign((size_t)T + 2, 0);

	if (D) {
		/* stage the forward index and transpose it on the device */
		HIP_TRY(hipMalloc(&d_img, src->dtmap_len));
		HIP_TRY(hipMalloc(&d_blk_off, D * 8));
		HIP_TRY(hipMalloc(&d_pair_base, (D + 1) * 8));
		HIP_TRY(hipMalloc(&d_term_ok, (size_t)T + 1));
		HIP_TRY(hipMalloc(&d_keys_in, std::max<uint64_t>(P, 1) * 4));
		HIP_TRY(hipMalloc(&d_keys, std::max<uint64_t>(P, 1) * 4));
		HIP_TRY(hipMalloc(&d_vals_in, std::max<uint64_t>(P, 1) * 8));
		HIP_TRY(hipMalloc(&d_first_bad, 8));
		HIP_TRY(hipMalloc(&d_max_tf, 4));
		HIP_TRY(hipMemcpyAsync(d_img, src->dtmap_img, src->dtmap_len, hipMemcpyHostToDevice, ix->stream));
		HIP_TRY(hipMemcpyAsync(d_blk_off, src->blk_off, D * 8, hipMemcpyHostToDevice, ix->stream));
		HIP_TRY(hipMemcpyAsync(d_pair_base, src->pair_base, (D + 1) * 8, hipMemcpyHostToDevice, ix->stream));
		HIP_TRY(hipMemcpyAsync(d_term_ok, src->term_ok, (size_t)T + 1, hipMemcpyHostToDevice, ix->stream));
		HIP_TRY(hipMemcpyAsync(ix->d_doc_ids, src->doc_ids, D * 8, hipMemcpyHostToDevice, ix->stream));
		HIP_TRY(hipMemcpyAsync(d_first_bad, &h_first_bad, 8, hipMemcpyHostToDevice, ix->stream));
		HIP_TRY(hipMemsetAsync(d_max_tf, 0, 4, ix->stream));
		{
			const uint64_t waves = (D + 15) / 16;
			const unsigned blocks = (unsigned)((waves + 3) / 4);
			hipLaunchKernelGGL(k_expand_pairs, dim3(blocks), dim3(256), 0, ix->stream,
			    d_img, d_blk_off, d_pair_base, D, T, d_term_ok, d_keys_in, d_vals_in,
			    ix->d_doc_len, d_first_bad, d_max_tf);
			HIP_TRY(hipGetLastError());
		}
		HIP_TRY(hipMemcpyAsync(&h_first_bad, d_first_bad, 8, hipMemcpyDeviceToHost, ix->stream));
		HIP_TRY(hipMemcpyAsync(&h_max_tf, d_max_tf, 4, hipMemcpyDeviceToHost, ix->stream));
		HIP_TRY(hipStreamSynchronize(ix->stream));
		(void)hipFree(d_img); d_img = NULL;
		(void)hipFree(d_blk_off); d_blk_off = NULL;
		(void)hipFree(d_pair_base); d_pair_base = NULL;
		(void)hipFree(d_term_ok); d_term_ok = NULL;
		ix->first_bad = h_first_bad;
		if (h_first_bad != ~0ull) {
			/* the caller truncates at this doc and rebuilds (partial sync) */
			goto done_partial;
		}
		if (h_max_tf >= (1u << 24)) {
			set_error("term frequency %u exceeds the supported 2^24", h_max_tf);
			goto fail;
		}
		if (P) {
			unsigned bits = 1;
			while (bits < 32 && (1ull << bits) <= T) {
				bits++;
			}
			/* stable LSD radix sort by term id keeps docs ascending inside a term */
			HIP_TRY(rocprim::radix_sort_pairs(NULL, tmp_bytes, d_keys_in, d_keys,
			    d_vals_in, ix->d_post_dt, (size_t)P, 0, bits, ix->stream));
			HIP_TRY(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 8));
			HIP_TRY(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_keys_in, d_keys,
			    d_vals_in, ix->d_post_dt, (size_t)P, 0, bits, ix->stream));
		}
	}
	{
		const unsigned blocks = (unsigned)(((uint64_t)T + 2 + 255) / 256);
		hipLaunchKernelGGL(k_post_offsets, dim3(blocks), dim3(256), 0, ix->stream,
		    d_keys, P, T, ix->d_post_off);
		HIP_TRY(hipGetLastError());
	}
	HIP_TRY(hipMemcpyAsync(ix->h_post_off.data(), ix->d_post_off, ((size_t)T + 2) * 8,
	    hipMemcpyDeviceToHost, ix->stream));
	HIP_TRY(hipStreamSynchronize(ix->stream));

	ix->max_tf = h_max_tf;
	ix->cap_post = std::max<uint64_t>(P, 1);
	if (rebuild_impacts(ix) != 0) {
		goto fail;
	}

	/* BK-tree image */
	if (src->n_bk) {
		HIP_TRY(hipMalloc(&ix->d_bk, (size_t)src->n_bk * sizeof(nxsgpu_bknode_t)));
		HIP_TRY(hipMalloc(&ix->d_bk_bytes, src->bk_bytes_len + 16));
		HIP_TRY(hipMemcpy(ix->d_bk, src->bk_nodes, (size_t)src->n_bk * sizeof(nxsgpu_bknode_t), hipMemcpyHostToDevice));
		HIP_TRY(hipMemcpy(ix->d_bk_bytes, src->bk_bytes, src->bk_bytes_len, hipMemcpyHostToDevice));
		if (bk_aux_build(ix, src->bk_nodes, src->n_bk) != 0) {
			goto fail;
		}
	}

	warm_streams(ix);
done_partial:
	(void)hipFree(d_keys_in);
	(void)hipFree(d_keys);
	(void)hipFree(d_vals_in);
	(void)hipFree(d_first_bad);
	(void)hipFree(d_max_tf);
	(void)hipFree(d_tmp);
	return ix;
fail:
	(void)hipFree(d_img);
	(void)hipFree(d_blk_off);
	(void)hipFree(d_pair_base);
	(void)hipFree(d_term_ok);
	(void)hipFree(d_keys_in);
	(void)hipFree(d_keys);
	(void)hipFree(d_vals_in);
	(void)hipFree(d_first_bad);
	(void)hipFree(d_max_tf);
	(void)hipFree(d_tmp);
	nxsgpu_index_destroy(ix);
	return NULL;
}

/* device array with room to grow: keeps `keep` elements when it has to move */
template <typename T>
static int
grow_array(T *&p, uint64_t &cap, uint64_t need, uint64_t keep, hipStream_t stream)
{
	if (need <= cap && p) {
		return 0;
	}
	const uint64_t ncap = need + need / 8 + 4096;
	T *np = NULL;
	if (hipMalloc((void **)&np, ncap * sizeof(T)) != hipSuccess) {
		set_error("hipMalloc(%llu) failed", (unsigned long long)(ncap * sizeof(T)));
		return -1;
	}
	if (p && keep && hipMemcpyAsync(np, p, keep * sizeof(T), hipMemcpyDeviceToDevice, stream) != hipSuccess) {
		(void)hipFree(np);
		set_error("device copy failed");
		return -1;
	}
	if (p) {
		(void)hipStreamSynchronize(stream);
		(void)hipFree(p);
	}
	p = np;
	cap = ncap;
	return 0;
}

/*
 * N1 -- incremental refresh (idx_terms_sync + idx_dtmap_sync on an open index,
 * src/index/terms.c:320-414, src/index/dtmap.c:440-544, called before every
 * search: src/query/search.c:309-312).  The delta -- appended doc blocks,
 * the postings of removed docs, new term ids, the header counters -- is merged
 * into the device CSR in one streaming pass and every impact is recomputed
 * (N, adl and df changed): O(postings) of device bandwidth, a few ms at 10M
 * docs, instead of re-reading and re-sorting the whole forward index.
 */
extern "C" int
nxsgpu_index_apply(nxsgpu_index_t *ix, const nxsgpu_index_delta_t *d)
{
	const uint32_t T_old = ix->n_terms, T_new = d->n_terms;
	const uint64_t P_old = ix->n_post, D_old = ix->n_docs;
	const uint64_t n_newdocs = d->n_new, n_newp = n_newdocs ? d->pair_base[n_newdocs] : 0;
	const uint32_t n_dead = (uint32_t)d->n_dead_pairs;
	uint8_t *d_img = NULL, *d_term_ok = NULL;
	uint64_t *d_blk_off = NULL, *d_pair_base = NULL, *d_vals_in = NULL, *d_vals = NULL;
	uint32_t *d_keys_in = NULL, *d_keys = NULL, *d_dead_term = NULL, *d_dead_ord = NULL;
	uint64_t *d_dead_pos = NULL, *d_dead_sorted = NULL, *d_new_off = NULL, *d_off_new = NULL, *d_out = NULL;
	unsigned long long *d_first_bad = NULL;
	unsigned int *d_max_tf = NULL;
	void *d_tmp = NULL;
	size_t tmp_bytes = 0;
	unsigned long long h_first_bad = ~0ull;
	unsigned int h_max_tf = 0;
	int rc = -1;

	if (ix->slot[0].active || ix->slot[1].active) {
		set_error("nxsgpu_index_apply: batches are in flight");
		return -1;
	}
	if (T_new < T_old || D_old + n_newdocs >= (1ull << 32)) {
		set_error("nxsgpu_index_apply: bad delta");
		return -1;
	}
	if (hipSetDevice(ix->device) != hipSuccess) {
		set_error("hipSetDevice failed");
		return -1;
	}
	uint64_t min_off = ~0ull, max_end = 0;
	for (uint64_t i = 0; i < n_newdocs; i++) {
		min_off = std::min(min_off, d->blk_off[i]);
		max_end = std::max(max_end, d->blk_off[i] + 16 + 8 * (d->pair_base[i + 1] - d->pair_base[i]));
	}
	if (n_newdocs && max_end > d->dtmap_len) {
		set_error("nxsgpu_index_apply: block beyond the image");
		return -1;
	}

	/* the doc tables grow at the end (appended docs take the highest ordinals) */
	if (grow_array(ix->d_doc_ids, ix->cap_docs_ids, D_old + n_newdocs, D_old, ix->stream) != 0 ||
	    grow_array(ix->d_doc_len, ix->cap_docs_len, D_old + n_newdocs, D_old, ix->stream) != 0) {
		return -1;
	}
	HIP_TRY(hipMalloc(&d_new_off, ((size_t)T_new + 2) * 8));
	HIP_TRY(hipMalloc(&d_off_new, ((size_t)T_new + 2) * 8));
	if (n_newdocs) {
		std::vector<uint64_t> rel(n_newdocs);
		for (uint64_t i = 0; i < n_newdocs; i++) {
			rel[i] = d->blk_off[i] - min_off;
		}
		HIP_TRY(hipMalloc(&d_img, max_end - min_off));
		HIP_TRY(hipMalloc(&d_blk_off, n_newdocs * 8));
		HIP_TRY(hipMalloc(&d_pair_base, (n_newdocs + 1) * 8));
		HIP_TRY(hipMalloc(&d_term_ok, (size_t)T_new + 1));
		HIP_TRY(hipMalloc(&d_keys_in, std::max<uint64_t>(n_newp, 1) * 4));
		HIP_TRY(hipMalloc(&d_keys, std::max<uint64_t>(n_newp, 1) * 4));
		HIP_TRY(hipMalloc(&d_vals_in, std::max<uint64_t>(n_newp, 1) * 8));
		HIP_TRY(hipMalloc(&d_vals, std::max<uint64_t>(n_newp, 1) * 8));
		HIP_TRY(hipMalloc(&d_first_bad, 8));
		HIP_TRY(hipMalloc(&d_max_tf, 4));
		HIP_TRY(hipMemcpyAsync(d_img, d->dtmap_img + min_off, max_end - min_off, hipMemcpyHostToDevice, ix->stream));
		HIP_TRY(hipMemcpyAsync(d_blk_off, rel.data(), n_newdocs * 8, hipMemcpyHostToDevice, ix->stream));
		HIP_TRY(hipMemcpyAsync(d_pair_base, d->pair_base, (n_newdocs + 1) * 8, hipMemcpyHostToDevice, ix->stream));
		HIP_TRY(hipMemcpyAsync(d_term_ok, d->term_ok, (size_t)T_new + 1, hipMemcpyHostToDevice, ix->stream));
		HIP_TRY(hipMemcpyAsync(ix->d_doc_ids + D_old, d->doc_ids, n_newdocs * 8, hipMemcpyHostToDevice, ix->stream));
		HIP_TRY(hipMemcpyAsync(d_first_bad, &h_first_bad, 8, hipMemcpyHostToDevice, ix->stream));
		HIP_TRY(hipMemsetAsync(d_max_tf, 0, 4, ix->stream));
		{
			const uint64_t waves = (n_newdocs + 15) / 16;
			const unsigned blocks = (unsigned)((waves + 3) / 4);
			/* ordinals of the appended docs start at D_old: the kernel numbers
			 * docs from 0, so it gets shifted views of the doc tables */
			hipLaunchKernelGGL(k_expand_pairs, dim3(blocks), dim3(256), 0, ix->stream,
			    d_img, d_blk_off, d_pair_base, n_newdocs, T_new, d_term_ok, d_keys_in, d_vals_in,
			    ix->d_doc_len + D_old, d_first_bad, d_max_tf);
			HIP_TRY(hipGetLastError());
		}
		HIP_TRY(hipMemcpyAsync(&h_first_bad, d_first_bad, 8, hipMemcpyDeviceToHost, ix->stream));
		HIP_TRY(hipMemcpyAsync(&h_max_tf, d_max_tf, 4, hipMemcpyDeviceToHost, ix->stream));
		HIP_TRY(hipStreamSynchronize(ix->stream));
		if (h_first_bad != ~0ull) {
			set_error("nxsgpu_index_apply: an appended block names an unknown term");
			goto fail;	/* (the host validates the delta first) */
		}
		if (h_max_tf >= (1u << 24)) {
			set_error("term frequency %u exceeds the supported 2^24", h_max_tf);
			goto fail;
		}
		if (n_newp) {
			unsigned bits = 1;
			while (bits < 32 && (1ull << bits) <= T_new) {
				bits++;
			}
			/* ordinals: + D_old (k_expand_pairs wrote 0-based ones) */
			hipLaunchKernelGGL(k_shift_ords, dim3((unsigned)std::min<uint64_t>((n_newp + 255) / 256, 65535)), dim3(256), 0,
			    ix->stream, d_vals_in, n_newp, D_old);
			HIP_TRY(rocprim::radix_sort_pairs(NULL, tmp_bytes, d_keys_in, d_keys, d_vals_in, d_vals,
			    (size_t)n_newp, 0, bits, ix->stream));
			HIP_TRY(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 8));
			HIP_TRY(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_keys_in, d_keys, d_vals_in, d_vals,
			    (size_t)n_newp, 0, bits, ix->stream));
		}
	}
	/* row offsets of the new postings (all zero when there are none) */
	if (n_newp) {
		const unsigned blocks = (unsigned)(((uint64_t)T_new + 2 + 255) / 256);
		hipLaunchKernelGGL(k_post_offsets, dim3(blocks), dim3(256), 0, ix->stream, d_keys, n_newp, T_new, d_new_off);
	} else {
		HIP_TRY(hipMemsetAsync(d_new_off, 0, ((size_t)T_new + 2) * 8, ix->stream));
	}
	/* where the postings of the removed docs sit */
	if (n_dead) {
		HIP_TRY(hipMalloc(&d_dead_term, (size_t)n_dead * 4));
		HIP_TRY(hipMalloc(&d_dead_ord, (size_t)n_dead * 4));
		HIP_TRY(hipMalloc(&d_dead_pos, (size_t)n_dead * 8));
		HIP_TRY(hipMalloc(&d_dead_sorted, (size_t)n_dead * 8));
		HIP_TRY(hipMemcpyAsync(d_dead_term, d->dead_term, (size_t)n_dead * 4, hipMemcpyHostToDevice, ix->stream));
		HIP_TRY(hipMemcpyAsync(d_dead_ord, d->dead_ord, (size_t)n_dead * 4, hipMemcpyHostToDevice, ix->stream));
		hipLaunchKernelGGL(k_dead_positions, dim3((n_dead + 255) / 256), dim3(256), 0, ix->stream,
		    ix->d_post_off, ix->d_post_dt, d_dead_term, d_dead_ord, n_dead, d_dead_pos);
		size_t tb = 0;
		void *tmp2 = NULL;
		HIP_TRY(rocprim::radix_sort_keys(NULL, tb, d_dead_pos, d_dead_sorted, (size_t)n_dead, 0, 64, ix->stream));
		HIP_TRY(hipMalloc(&tmp2, tb ? tb : 8));
		if (rocprim::radix_sort_keys(tmp2, tb, d_dead_pos, d_dead_sorted, (size_t)n_dead, 0, 64, ix->stream) != hipSuccess) {
			(void)hipFree(tmp2);
			set_error("sort failed");
			goto fail;
		}
		HIP_TRY(hipStreamSynchronize(ix->stream));
		(void)hipFree(tmp2);
		/* a posting that was not found would corrupt the merge */
		uint64_t last = 0;
		HIP_TRY(hipMemcpy(&last, d_dead_sorted + (n_dead - 1), 8, hipMemcpyDeviceToHost));
		if (last == ~0ull) {
			set_error("nxsgpu_index_apply: a removed doc's posting is not in the index");
			goto fail;
		}
	}
	{
		const uint64_t P_new = P_old - n_dead + n_newp;
		const unsigned blocks = (unsigned)(((uint64_t)T_new + 2 + 255) / 256);

		HIP_TRY(hipMalloc(&d_out, std::max<uint64_t>(P_new + P_new / 16 + 4096, 1) * 8));
		hipLaunchKernelGGL(k_new_row_offsets, dim3(blocks), dim3(256), 0, ix->stream,
		    ix->d_post_off, T_old, P_old, d_dead_sorted, n_dead, d_new_off, T_new, d_off_new);
		if (P_old) {
			hipLaunchKernelGGL(k_merge_old, dim3(4096), dim3(256), 0, ix->stream,
			    ix->d_post_off, T_old, ix->d_post_dt, P_old, d_dead_sorted, n_dead, d_new_off, d_out);
		}
		if (n_newp) {
			hipLaunchKernelGGL(k_place_new, dim3((unsigned)std::min<uint64_t>((n_newp + 255) / 256, 65535)), dim3(256), 0,
			    ix->stream, d_keys, d_vals, n_newp, d_new_off, d_off_new, d_out);
		}
		HIP_TRY(hipGetLastError());
		ix->h_post_off.assign((size_t)T_new + 2, 0);
		HIP_TRY(hipMemcpyAsync(ix->h_post_off.data(), d_off_new, ((size_t)T_new + 2) * 8, hipMemcpyDeviceToHost, ix->stream));
		HIP_TRY(hipStreamSynchronize(ix->stream));
		if (ix->h_post_off[(size_t)T_new + 1] != P_new) {
			set_error("nxsgpu_index_apply: merged %llu postings, expected %llu",
			    (unsigned long long)ix->h_post_off[(size_t)T_new + 1], (unsigned long long)P_new);
			goto fail;
		}
		/* swap in the new CSR */
		(void)hipFree(ix->d_post_dt);
		ix->d_post_dt = d_out;
		d_out = NULL;
		(void)hipFree(ix->d_post_off);
		ix->d_post_off = d_off_new;
		d_off_new = NULL;
		ix->n_post = P_new;
		ix->n_terms = T_new;
		ix->n_docs = D_old + n_newdocs;
		ix->max_tf = std::max(ix->max_tf, h_max_tf);
		ix->hdr_doc_count = d->hdr_doc_count;
		ix->hdr_token_count = d->hdr_token_count;
		/* impact arrays follow the posting count */
		if (P_new > ix->cap_post) {
			(void)hipFree(ix->d_post[0]);
			(void)hipFree(ix->d_post[1]);
			ix->d_post[0] = ix->d_post[1] = NULL;
			ix->cap_post = P_new + P_new / 16 + 4096;
			HIP_TRY(hipMalloc(&ix->d_post[0], ix->cap_post * sizeof(posting_t)));
			HIP_TRY(hipMalloc(&ix->d_post[1], ix->cap_post * sizeof(posting_t)));
		}
	}
	if (rebuild_impacts(ix) != 0) {
		goto fail;
	}
	rc = 0;
fail:
	(void)hipFree(d_img);
	(void)hipFree(d_blk_off);
	(void)hipFree(d_pair_base);
	(void)hipFree(d_term_ok);
	(void)hipFree(d_keys_in);
	(void)hipFree(d_keys);
	(void)hipFree(d_vals_in);
	(void)hipFree(d_vals);
	(void)hipFree(d_first_bad);
	(void)hipFree(d_max_tf);
	(void)hipFree(d_tmp);
	(void)hipFree(d_dead_term);
	(void)hipFree(d_dead_ord);
	(void)hipFree(d_dead_pos);
	(void)hipFree(d_dead_sorted);
	(void)hipFree(d_new_off);
	(void)hipFree(d_off_new);
	(void)hipFree(d_out);
	return rc;
}

/* replace the BK-tree image (new terms were inserted on the host) */
extern "C" int
nxsgpu_index_set_bk(nxsgpu_index_t *ix, const nxsgpu_bknode_t *nodes, uint32_t n, uint32_t depth,
    const uint8_t *bytes, uint64_t bytes_len)
{
	if (hipSetDevice(ix->device) != hipSuccess) {
		set_error("hipSetDevice failed");
		return -1;
	}
	(void)hipStreamSynchronize(ix->stream_fz);
	(void)hipFree(ix->d_bk);
	(void)hipFree(ix->d_bk_bytes);
	bk_aux_free(ix);
	ix->d_bk = NULL;
	ix->d_bk_bytes = NULL;
	ix->n_bk = 0;
	ix->bk_depth = 0;
	if (n == 0) {
		return 0;
	}
	if (hipMalloc(&ix->d_bk, (size_t)n * sizeof(nxsgpu_bknode_t)) != hipSuccess ||
	    hipMalloc(&ix->d_bk_bytes, bytes_len + 16) != hipSuccess ||
	    hipMemcpy(ix->d_bk, nodes, (size_t)n * sizeof(nxsgpu_bknode_t), hipMemcpyHostToDevice) != hipSuccess ||
	    hipMemcpy(ix->d_bk_bytes, bytes, bytes_len, hipMemcpyHostToDevice) != hipSuccess) {
		set_error("BK-tree upload failed");
		return -1;
	}
	if (bk_aux_build(ix, nodes, n) != 0) {
		return -1;
	}
	ix->n_bk = n;
	ix->bk_depth = depth;
	return 0;
}

extern "C" int
nxsgpu_index_df(nxsgpu_index_t *ix, uint32_t *df)
{
	df[0] = 0;
	for (uint32_t t = 1; t <= ix->n_terms; t++) {
		df[t] = (uint32_t)(ix->h_post_off[t + 1] - ix->h_post_off[t]);
	}
	return 0;
}

extern "C" uint64_t nxsgpu_index_postings(const nxsgpu_index_t *ix) { return ix->n_post; }
extern "C" uint64_t nxsgpu_index_docs(const nxsgpu_index_t *ix) { return ix->n_docs; }
extern "C" uint64_t nxsgpu_index_first_bad_doc(const nxsgpu_index_t *ix) { return ix->first_bad; }

extern "C" void
nxsgpu_set_profiling(nxsgpu_index_t *ix, int on)
{
	ix->profiling = on != 0;
}

extern "C" void
nxsgpu_get_profile(nxsgpu_index_t *ix, nxsgpu_profile_t *p, int reset)
{
	*p = ix->prof;
	if (reset) {
		memset(&ix->prof, 0, sizeof(ix->prof));
	}
}

extern "C" void
nxsgpu_synchronize(nxsgpu_index_t *ix)
{
	(void)hipSetDevice(ix->device);
	(void)hipStreamSynchronize(ix->stream);
}

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
struct launch_t { uint32_t first, count, nt_bucket, kind, nomask, q_first, q_count; };	/* kind: 0 wide, 1 tile, 2 step */

struct worklist_t {
	std::vector<qmeta_t>	qmeta;
	std::vector<item_t>	items;
	std::vector<launch_t>	launches;
	std::vector<uint32_t>	bnd_q;		/* boundary -> query, n_segs + nq entries */
	std::vector<uint32_t>	qorder;		/* queries in launch order; launch_t::q_first/q_count index it */
	uint32_t		n_segs;
	bool			need_cursors;	/* some query's ranges are doc ranges (k_cursors has work) */
};

static void
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
build_worklist(const nxsgpu_index_t *ix, const dev_query_t *hq, uint32_t nq, worklist_t &wl, bool solo = false)
{
	const uint64_t tiles = std::max<uint64_t>(1, (ix->n_docs + TILE_W - 1) / TILE_W);
	const gpu_cfg_t &cf = ix->cfg;
	/* (a batch that has the GPU to itself is latency-bound: shorter ranges, more of them) */
	const uint64_t target = cf.wave_target, min_post = solo ? std::min(cf.min_post, cf.min_post_solo) : cf.min_post;
	/* densest term has >= this many postings per tile => tile path (step path
	 * off by default: the tile path is at least as fast) */
#ifdef NXS_EXPERIMENTAL
	const double dense_thr = cf.dense_thr;
#else
	const double dense_thr = 0.0;	/* k_scanh is an opt-in build */
#endif
	const bool use_scanr = cf.use_scanr && ix->n_docs < (1ull << 31);
	const bool no_step = cf.no_step, mask_off = cf.mask_off;
	const uint32_t rmin = cf.rmin;	/* 3: "a AND b" takes k_scan8's sign-bit path */
	const bool by_level = cf.by_level;
	const bool use_scanm = cf.use_scanm && ix->n_docs < (1ull << 31);
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
			const double per_tile = (double)wmax * TILE_W / (double)std::max<uint64_t>(ix->n_docs, 1);
			const bool tile = dense_thr <= 0.0 || per_tile >= dense_thr || hq[i].nt <= 1 ||
			    ix->n_docs >= (1ull << 31) || no_step;
			/* pure OR: every non-empty presence mask matches => no mask array */
			bool or_only = hq[i].nt >= 2 && mask_off;
			for (uint32_t m = 1; or_only && m < (1u << hq[i].nt); m++) {
				or_only = (hq[i].truth[m >> 5] >> (m & 31)) & 1;
			}
			/* pure AND of exactly two tokens: only the full mask matches.
			 * (The sign-parity scheme of MM = 2 cannot tell "stuck at token
			 * t-2" from "updated by token t" for three tokens or more.) */
			bool and_only = hq[i].nt == 2 && mask_off;
			for (uint32_t m = 1; and_only && m < (1u << hq[i].nt); m++) {
				const bool hit = (hq[i].truth[m >> 5] >> (m & 31)) & 1;
				and_only = hit == (m == (1u << hq[i].nt) - 1);
			}
#ifndef NXS_EXPERIMENTAL
			and_only = false;	/* the sign-bit AND path (MM = 2) is an opt-in build */
#endif
			const uint32_t mm = !tile ? 0u : or_only ? 1u : and_only ? 2u : 0u;
			cls[i] = (tile ? 1u : 2u) * 64 + mm * 16 + nt_bucket(hq[i].nt);
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
			if (tile && (or_only || (no_req && scanm_general)) && use_scanm &&
			    hq[i].nt >= scanm_minnt && hq[i].nt <= scanm_maxnt &&
			    (double)wmax <= scanm_dens * (double)ix->n_docs) {
				cls[i] = 4u * 64 + (or_only ? 16u : 0u) + nt_bucket(hq[i].nt);
			} else if (tile && or_only && use_scanm && cf.use_drop && hq[i].drop_mask &&
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
					}
				}
				if (n_sparse && ws >= cf.drop_minpost) {
					total -= work[i];
					work[i] = cf.drop_workmul * (ws + 16384);	/* latency-bound wavefronts: more, shorter ranges */
					total += work[i];
					cls[i] = 5u * 64 + 16u + nt_bucket(hq[i].nt);
				}
			}
			/* required terms: intersect first (k_scanr).  Its work is set by
			 * the shortest required list; longer lists are mostly skipped */
			if (tile && hq[i].n_req && hq[i].nt >= rmin && use_scanr) {
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
			}
		}
	}
	const uint64_t per_wave = std::max<uint64_t>(min_post, total / std::max<uint64_t>(target, 1) + 1);
	/* launch order of the classes: the mask path first -- a class's heap replay
	 * runs beside the NEXT class's scan, and the last class (required-term
	 * queries: few candidates, short replay) is the one left exposed */
	/* (the sparse + dense class leads: it runs on a stream of its own, beside the rest) */
	auto cls_key = [&](uint32_t c) -> uint32_t { return (c >> 6) == 5 ? (c & 63) : (c >> 6) == 4 ? 64 + (c & 63) : c + 256; };
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
	wl.bnd_q.clear();
	wl.bnd_q.reserve((size_t)wl.n_segs + nq);
	for (uint32_t i = 0; i < nq; i++) {
		/* query i owns boundaries seg_first + i ... + n_groups (inclusive) */
		for (uint32_t g = 0; g <= wl.qmeta[i].n_groups; g++) {
			wl.bnd_q.push_back(i);
		}
	}
	wl.items.reserve(wl.n_segs);
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
		l.first = (uint32_t)wl.items.size();
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
						wl.items.push_back(it);
					}
				}
			}
		} else {
			for (uint32_t oi = o0; oi < o1; oi++) {
				const uint32_t i = order[oi];
				for (uint32_t g = wl.qmeta[i].n_groups; g-- > 0; ) {
					item_t it;
					it.q = i;
					it.g = g;
					wl.items.push_back(it);
				}
			}
		}
		l.count = (uint32_t)wl.items.size() - l.first;
		l.q_first = o0;
		l.q_count = o1 - o0;
		wl.launches.push_back(l);
		o0 = o1;
	}
}

static void
launch_cursors(nxsgpu_index_t *ix, const scan_args_t &a, const uint32_t *d_bnd_q, uint32_t n_bnd,
    hipStream_t stream = NULL)
{
	const uint64_t threads = (uint64_t)n_bnd * NXSGPU_MAX_TOKENS;
	if (n_bnd) {
		hipLaunchKernelGGL(k_cursors, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
		    stream ? stream : ix->stream,
		    a.post, a.queries, a.qmeta, d_bnd_q, n_bnd, a.n_docs, (uint32_t *)a.cursors);
	}
}

/* the sparse + dense OR class: its cold phase (k_cold), then the mask path on
 * the sparse terms (k_scanm<.., DROP>), stream-ordered */
static void
launch_drop_class(const dim3 grid, const scan_args_t &a, uint32_t nt_bucket, hipStream_t st)
{
	const dim3 block(WAVE);

	switch (nt_bucket) {
	case 2:
	case 3:
		hipLaunchKernelGGL((k_cold<3, false>), grid, block, 0, st, a);
		hipLaunchKernelGGL((k_scanm<3, false, true>), grid, block, 0, st, a);
		break;
	case 5:
		hipLaunchKernelGGL((k_cold<5, false>), grid, block, 0, st, a);
		hipLaunchKernelGGL((k_scanm<5, false, true>), grid, block, 0, st, a);
		break;
	default:
		hipLaunchKernelGGL((k_cold<8, false>), grid, block, 0, st, a);
		hipLaunchKernelGGL((k_scanm<8, false, true>), grid, block, 0, st, a);
		break;
	}
}

/*
 * One scan launch per query class.  With `ra` (top-k filter pass) the heap
 * replay of a class is queued on the second stream as soon as the class's
 * scan is: the replay is a few latency-bound wavefronts (one per query) and
 * runs beside the next class's scan instead of after all of them.
 */
template <int MODE>
static void
launch_scan(nxsgpu_index_t *ix, const scan_args_t &a0, const worklist_t &wl,
    const replay_args_t *ra = NULL, const uint32_t *d_qorder = NULL, hipEvent_t scans_done = NULL)
{
	bool forked = false, forked3 = false;
	const launch_t *last_launch = NULL;
	size_t n_launches = 0;

	for (const launch_t &l : wl.launches) {
		n_launches += l.count != 0;
	}
	/* the sparse + dense class goes to its own stream when there is something to
	 * run it beside (top-k pass only: its replay follows it there) */
	const bool side3 = MODE == MODE_TOPK && ra && n_launches > 1 && a0.k >= 1 && a0.k <= WAVE && ix->cfg.drop_side;
	for (const launch_t &l : wl.launches) {
		if (l.count && !(side3 && l.kind == 5)) {
			last_launch = &l;
		}
	}
	for (const launch_t &l : wl.launches) {
		scan_args_t a = a0;
		const dim3 grid(l.count), block(WAVE);

		if (l.count == 0) {
			continue;
		}
		a.item_base = l.first;
		if (side3 && l.kind == 5) {
			replay_args_t r = *ra;
			r.qlist = d_qorder + l.q_first;
			if (!forked3) {
				(void)hipEventRecord(ix->ev_fork3, ix->stream);
				(void)hipStreamWaitEvent(ix->stream3, ix->ev_fork3, 0);
				forked3 = true;
			}
			a.flags = ix->cfg.drop_prio ? 1u : 0u;
			launch_drop_class(grid, a, l.nt_bucket, ix->stream3);
			hipLaunchKernelGGL(k_replay<HEAP_REG>, dim3(l.q_count), dim3(WAVE), 0, ix->stream3, r);
			continue;
		}
		if (l.kind == 0) {
			hipLaunchKernelGGL((k_scan<NXSGPU_MAX_TOKENS, uint32_t, MODE>), grid, block, 0, ix->stream, a);
		} else if (ix->cfg.old_scan || ix->n_docs >= (1ull << 31)) {
			hipLaunchKernelGGL((k_scan<8, uint8_t, MODE>), grid, block, 0, ix->stream, a);
		} else if (l.kind == 1) {
			switch (l.nt_bucket) {
			case 1:
				if (ix->cfg.no_scan1) {
					hipLaunchKernelGGL((k_scan8<MODE, 1, 0>), grid, block, 0, ix->stream, a);
				} else {
					hipLaunchKernelGGL((k_scan1<MODE>), grid, block, 0, ix->stream, a);
				}
				break;
			case 2: if (l.nomask == 1) { hipLaunchKernelGGL((k_scan8<MODE, 2, 1>), grid, block, 0, ix->stream, a); }
#ifdef NXS_EXPERIMENTAL
			else if (l.nomask == 2) { hipLaunchKernelGGL((k_scan8<MODE, 2, 2>), grid, block, 0, ix->stream, a); }
#endif
			else { hipLaunchKernelGGL((k_scan8<MODE, 2, 0>), grid, block, 0, ix->stream, a); } break;
			case 3: if (l.nomask == 1) { hipLaunchKernelGGL((k_scan8<MODE, 3, 1>), grid, block, 0, ix->stream, a); } else { hipLaunchKernelGGL((k_scan8<MODE, 3, 0>), grid, block, 0, ix->stream, a); } break;
			case 5: if (l.nomask == 1) { hipLaunchKernelGGL((k_scan8<MODE, 5, 1>), grid, block, 0, ix->stream, a); } else { hipLaunchKernelGGL((k_scan8<MODE, 5, 0>), grid, block, 0, ix->stream, a); } break;
			default: if (l.nomask == 1) { hipLaunchKernelGGL((k_scan8<MODE, 8, 1>), grid, block, 0, ix->stream, a); } else { hipLaunchKernelGGL((k_scan8<MODE, 8, 0>), grid, block, 0, ix->stream, a); } break;
			}
		} else if (l.kind == 4) {
			/* mask path: top-k filter pass only; the exact passes (count, emit
			 * all) of these queries take the pure-OR accumulator tiles */
			if (MODE == MODE_TOPK && a.k >= 1 && a.k <= WAVE) {
				if (l.nomask == 1) {
					switch (l.nt_bucket) {
					case 2:		/* two tokens: the third slot stays empty */
					case 3: hipLaunchKernelGGL((k_scanm<3, false>), grid, block, 0, ix->stream, a); break;
					case 5: hipLaunchKernelGGL((k_scanm<5, false>), grid, block, 0, ix->stream, a); break;
					default: hipLaunchKernelGGL((k_scanm<8, false>), grid, block, 0, ix->stream, a); break;
					}
				} else {
					switch (l.nt_bucket) {
					case 2:
					case 3: hipLaunchKernelGGL((k_scanm<3, true>), grid, block, 0, ix->stream, a); break;
					case 5: hipLaunchKernelGGL((k_scanm<5, true>), grid, block, 0, ix->stream, a); break;
					default: hipLaunchKernelGGL((k_scanm<8, true>), grid, block, 0, ix->stream, a); break;
					}
				}
			} else {
				if (l.nomask == 1) {
					switch (l.nt_bucket) {
					case 2: hipLaunchKernelGGL((k_scan8<MODE, 2, 1>), grid, block, 0, ix->stream, a); break;
					case 3: hipLaunchKernelGGL((k_scan8<MODE, 3, 1>), grid, block, 0, ix->stream, a); break;
					case 5: hipLaunchKernelGGL((k_scan8<MODE, 5, 1>), grid, block, 0, ix->stream, a); break;
					default: hipLaunchKernelGGL((k_scan8<MODE, 8, 1>), grid, block, 0, ix->stream, a); break;
					}
				} else {
					switch (l.nt_bucket) {
					case 2: hipLaunchKernelGGL((k_scan8<MODE, 2, 0>), grid, block, 0, ix->stream, a); break;
					case 3: hipLaunchKernelGGL((k_scan8<MODE, 3, 0>), grid, block, 0, ix->stream, a); break;
					case 5: hipLaunchKernelGGL((k_scan8<MODE, 5, 0>), grid, block, 0, ix->stream, a); break;
					default: hipLaunchKernelGGL((k_scan8<MODE, 8, 0>), grid, block, 0, ix->stream, a); break;
					}
				}
			}
		} else if (l.kind == 5) {
			/* sparse + dense pure OR: top-k pass with the dense lists dropped;
			 * the exact passes take the accumulator tiles */
			if (MODE == MODE_TOPK && a.k >= 1 && a.k <= WAVE) {
				launch_drop_class(grid, a, l.nt_bucket, ix->stream);
			} else {
				switch (l.nt_bucket) {
				case 2: hipLaunchKernelGGL((k_scan8<MODE, 2, 1>), grid, block, 0, ix->stream, a); break;
				case 3: hipLaunchKernelGGL((k_scan8<MODE, 3, 1>), grid, block, 0, ix->stream, a); break;
				case 5: hipLaunchKernelGGL((k_scan8<MODE, 5, 1>), grid, block, 0, ix->stream, a); break;
				default: hipLaunchKernelGGL((k_scan8<MODE, 8, 1>), grid, block, 0, ix->stream, a); break;
				}
			}
		} else if (l.kind == 3) {
			switch (l.nt_bucket) {
			case 2: hipLaunchKernelGGL((k_scanr<MODE, 2>), grid, block, 0, ix->stream, a); break;
			case 3: hipLaunchKernelGGL((k_scanr<MODE, 3>), grid, block, 0, ix->stream, a); break;
			case 5:
				if (l.nomask == 1) { hipLaunchKernelGGL((k_scanr<MODE, 5, true>), grid, block, 0, ix->stream, a); }
				else { hipLaunchKernelGGL((k_scanr<MODE, 5>), grid, block, 0, ix->stream, a); }
				break;
			default:
				if (l.nomask == 1) { hipLaunchKernelGGL((k_scanr<MODE, 8, true>), grid, block, 0, ix->stream, a); }
				else { hipLaunchKernelGGL((k_scanr<MODE, 8>), grid, block, 0, ix->stream, a); }
				break;
			}
		} else {
#ifdef NXS_EXPERIMENTAL
			switch (l.nt_bucket) {
			case 2: hipLaunchKernelGGL((k_scanh<MODE, 2>), grid, block, 0, ix->stream, a); break;
			case 3: hipLaunchKernelGGL((k_scanh<MODE, 3>), grid, block, 0, ix->stream, a); break;
			case 5: hipLaunchKernelGGL((k_scanh<MODE, 5>), grid, block, 0, ix->stream, a); break;
			default: hipLaunchKernelGGL((k_scanh<MODE, 8>), grid, block, 0, ix->stream, a); break;
			}
#endif
		}
		if (ra && l.q_count) {
			replay_args_t r = *ra;
			r.qlist = d_qorder + l.q_first;
			if (&l == last_launch) {
				/* nothing left to run beside it: same stream, no event
				 * round trip (a single query has only this one) */
				if (scans_done) {
					(void)hipEventRecord(scans_done, ix->stream);
					scans_done = NULL;
				}
				hipLaunchKernelGGL(k_replay<HEAP_REG>, dim3(l.q_count), dim3(WAVE), 0, ix->stream, r);
			} else {
				(void)hipEventRecord(ix->ev_cls, ix->stream);
				(void)hipStreamWaitEvent(ix->stream2, ix->ev_cls, 0);
				hipLaunchKernelGGL(k_replay<HEAP_REG>, dim3(l.q_count), dim3(WAVE), 0, ix->stream2, r);
				forked = true;
			}
		}
	}
	if (scans_done) {
		(void)hipEventRecord(scans_done, ix->stream);
	}
	if (forked) {
		(void)hipEventRecord(ix->ev_join, ix->stream2);
		(void)hipStreamWaitEvent(ix->stream, ix->ev_join, 0);
	}
	if (forked3) {
		(void)hipEventRecord(ix->ev_join3, ix->stream3);
		(void)hipStreamWaitEvent(ix->stream, ix->ev_join3, 0);
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
static int
fill_dev_queries(const nxsgpu_index_t *ix, int algo, const nxsgpu_query_t *queries, uint32_t nq,
    dev_query_t *hq, uint64_t &total_post, bool allow_drop = true)
{
	const bool valid = (algo == NXSGPU_BM25) ? ix->bm25_valid : ix->tfidf_valid;
	const bool no_req = ix->cfg.no_req;

	for (uint32_t i = 0; i < nq; i++) {
		const nxsgpu_query_t &q = queries[i];
		dev_query_t &d = hq[i];
		memset(&d, 0, sizeof(d));
		if (q.n_tokens > NXSGPU_MAX_TOKENS || q.prog_len > NXSGPU_MAX_PROG) {
			set_error("query %u exceeds the device limits", i);
			return -1;
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
			if (tid == 0 || tid > ix->n_terms) {
				set_error("query %u: bad term id %u", i, tid);
				return -1;
			}
			d.pbeg[t] = ix->h_post_off[tid];
			d.pend[t] = ix->h_post_off[tid + 1];
			total_post += d.pend[t] - d.pbeg[t];
			if (t < 8 && tid < ix->h_maximp[algo].size()) {
				d.tmax[t] = ix->h_maximp[algo][tid];
			}
		}
		/* dense tokens (k_scanm<.., DROP>): lists above the mask path's density limit */
		/*
		 * (BM25 only: its tf part saturates, so a term's largest impact says what
		 * the term typically adds.  TF-IDF's log(tf + 1) does not: one posting with
		 * an outlier tf sets a ceiling that thresholds reach late -- measured 3x
		 * slower than the accumulator tiles there.)
		 */
		d.drop_mask = 0;
		if (allow_drop && d.nt >= 2 && d.nt <= 8 && ix->cfg.use_drop && !ix->dense_terms.empty() && algo == NXSGPU_BM25) {
			for (uint32_t t = 0; t < d.nt; t++) {
				const auto it = std::lower_bound(ix->dense_terms.begin(), ix->dense_terms.end(), q.term_id[t]);
				if (it != ix->dense_terms.end() && *it == q.term_id[t]) {
					d.drop_mask |= 1u << t;
					d.drop_col[t] = (uint32_t)(it - ix->dense_terms.begin());
				}
			}
		}
		if (d.drop_mask) {
			/* worth it only while the dense ceiling stays well below what one
			 * sparse posting can add */
			float u = 0.0f, smin = INFINITY;
			for (uint32_t t = 0; t < d.nt; t++) {
				if ((d.drop_mask >> t) & 1) {
					u += d.tmax[t];
				} else {
					smin = std::min(smin, d.tmax[t]);
				}
			}
			if (!(u <= 1.25f * smin)) {
				d.drop_mask = 0;
			}
		}
		/* k_scanr slot order: required tokens first, shortest list first */
		d.n_req = 0;
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
	const bool fast = limit <= NXSGPU_FAST_K;
	const uint32_t seg_cap = ix->cfg.seg_cap;
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
	build_worklist(ix, hq.data(), nq, wl);
	const uint64_t nseg = wl.n_segs;

	/* workspace: queries | meta | items | seg_count | overflow | candidates | outputs */
	{
		size_t need = 8192 + nq * 4 + nq * sizeof(dev_query_t) + nq * sizeof(qmeta_t)
		    + nseg * sizeof(item_t) + nseg * 4 + nq * 4
		    + (nseg + nq) * 4 * (1 + NXSGPU_MAX_TOKENS) + nseg * 4 + 1024
		    + nseg * seg_cap * 8 + (size_t)nq * kfast * 12 + nq * 4 + 16 * 256
		    + nseg * (16 * 4 + 64 * 4) + 1024;
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

	h_ovf.assign(nq, 0);
	if (fast) {
		if (ix->profiling) (void)hipEventRecord(ix->ev[0], ix->stream);
		launch_cursors(ix, sa, d_bnd_q, (uint32_t)(nseg + nq));
		memset(&ra, 0, sizeof(ra));
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
			launch_scan<MODE_TOPK>(ix, sa, wl);
			if (ix->profiling) (void)hipEventRecord(ix->ev[1], ix->stream);
			hipLaunchKernelGGL(k_replay<HEAP_REG>, dim3(nq), dim3(WAVE), 0, ix->stream, ra);
		} else {
			/* (profile: "replay" is then only what the last class's replay
			 * adds after the last scan) */
			launch_scan<MODE_TOPK>(ix, sa, wl, &ra, d_qorder, ix->profiling ? ix->ev[1] : NULL);
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
	if (fast && ix->profiling && !(ix->slot[0].active || ix->slot[1].active)) {
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

		/* pass 1: count */
		if (hipMemcpyAsync(dx_q, xhq.data(), nx * sizeof(dev_query_t), hipMemcpyHostToDevice, ix->stream) != hipSuccess ||
		    hipMemcpyAsync(dx_qmeta, xwl.qmeta.data(), nx * sizeof(qmeta_t), hipMemcpyHostToDevice, ix->stream) != hipSuccess ||
		    hipMemcpyAsync(dx_items, xwl.items.data(), xseg * sizeof(item_t), hipMemcpyHostToDevice, ix->stream) != hipSuccess ||
		    hipMemcpyAsync(dx_bnd_q, xwl.bnd_q.data(), (xseg + nx) * 4, hipMemcpyHostToDevice, ix->stream) != hipSuccess) {
			set_error("query upload failed");
			return -1;
		}
		sa.queries = dx_q;
		sa.qmeta = dx_qmeta;
		sa.items = dx_items;
		sa.seg_count = dx_seg_count;
		sa.cursors = dx_cursors;
		sa.k = 0xffffffffu;
		launch_cursors(ix, sa, dx_bnd_q, (uint32_t)(xseg + nx));
		launch_scan<MODE_COUNT>(ix, sa, xwl);
		if (hipMemcpyAsync(sc_cnt.data(), dx_seg_count, xseg * 4, hipMemcpyDeviceToHost, ix->stream) != hipSuccess ||
		    hipStreamSynchronize(ix->stream) != hipSuccess) {
			set_error("count pass failed: %s", hipGetErrorString(hipGetLastError()));
			return -1;
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
			return -1;
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
			launch_scan<MODE_ALL>(ix, sb, xwl);
			memset(&ra, 0, sizeof(ra));
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
				hipLaunchKernelGGL(k_replay<HEAP_LDS>, dim3(nx), dim3(WAVE), (size_t)ra.k * 8, ix->stream, ra);
			} else {
				hipLaunchKernelGGL(k_replay<HEAP_GLOBAL>, dim3(nx), dim3(WAVE), 0, ix->stream, ra);
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
	const bool busy = ix->slot[0].active || ix->slot[1].active;
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

/* ---- N4: doc-sharded mode ------------------------------------------------------------ */

extern "C" int
nxsgpu_index_set_global_df(nxsgpu_index_t *ix, const uint32_t *df, uint32_t n_terms)
{
	if (ix->slot[0].active || ix->slot[1].active) {
		set_error("nxsgpu_index_set_global_df: batches are in flight");
		return -1;
	}
	if (n_terms != ix->n_terms) {
		set_error("nxsgpu_index_set_global_df: %u terms, the index has %u", n_terms, ix->n_terms);
		return -1;
	}
	if (hipSetDevice(ix->device) != hipSuccess) {
		set_error("hipSetDevice failed");
		return -1;
	}
	ix->df_global.assign((size_t)n_terms + 2, 0);
	for (uint32_t t = 1; t <= n_terms; t++) {
		ix->df_global[t] = df[t];
	}
	return rebuild_impacts(ix);
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
 * The merge step of the doc-sharded mode: the shards' accepted-candidate logs
 * of every query, highest shard (highest doc ids) first, are fed to the
 * reference's heap again (k_replay) -- by induction its state is the global
 * one (results.c:182-220 feeds descending doc id; heap.c:58-221).  Layout:
 * ids/scores [nq][n_shards][cap], counts [nq][n_shards], shard 0 = LOWEST docs.
 * Output: [nq][limit] + counts.  Runs on `device`, blocking.
 */
extern "C" int
nxsgpu_merge_candidates(int device, uint32_t limit, uint32_t nq, uint32_t n_shards, uint32_t cap,
    const uint64_t *ids, const float *scores, const uint32_t *counts,
    uint64_t *out_ids, float *out_scores, uint32_t *out_counts)
{
	const size_t nseg = (size_t)nq * n_shards, ncand = nseg * cap;
	std::vector<qmeta_t> qm(nq);
	void *ws = NULL;
	hipStream_t st = NULL;
	int rc = -1;

	if (limit == 0 || limit > NXSGPU_FAST_K) {
		set_error("nxsgpu_merge_candidates: limit must be 1..%d", NXSGPU_FAST_K);
		return -1;
	}
	if (nq == 0) {
		return 0;
	}
	for (size_t i = 0; i < nseg; i++) {
		if (counts[i] > cap) {
			set_error("nxsgpu_merge_candidates: a candidate log overflowed (%u > %u)", counts[i], cap);
			return -1;
		}
	}
	for (uint32_t q = 0; q < nq; q++) {
		qm[q].seg_first = q * n_shards;
		qm[q].n_groups = n_shards;
		qm[q].group_docs = 0;
		qm[q].pad = 0;
	}
	do {
		if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
			set_error("device setup failed");
			break;
		}
		const size_t need = 8192 + nq * sizeof(qmeta_t) + nseg * 4 + ncand * 12 + (size_t)nq * limit * 12 + nq * 4;
		if (hipMalloc(&ws, need) != hipSuccess) {
			set_error("hipMalloc(%zu) failed", need);
			break;
		}
		uint8_t *p = (uint8_t *)ws;
		qmeta_t *d_qm = carve<qmeta_t>(p, nq);
		uint32_t *d_cnt = carve<uint32_t>(p, nseg);
		uint64_t *d_ids = carve<uint64_t>(p, ncand);
		float *d_sc = carve<float>(p, ncand);
		uint64_t *d_oid = carve<uint64_t>(p, (size_t)nq * limit);
		float *d_osc = carve<float>(p, (size_t)nq * limit);
		uint32_t *d_ocnt = carve<uint32_t>(p, nq);
		replay_args_t ra;

		if (hipMemcpyAsync(d_qm, qm.data(), nq * sizeof(qmeta_t), hipMemcpyHostToDevice, st) != hipSuccess ||
		    hipMemcpyAsync(d_cnt, counts, nseg * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
		    hipMemcpyAsync(d_ids, ids, ncand * 8, hipMemcpyHostToDevice, st) != hipSuccess ||
		    hipMemcpyAsync(d_sc, scores, ncand * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
		    hipMemsetAsync(d_ocnt, 0, nq * 4, st) != hipSuccess) {
			set_error("upload failed");
			break;
		}
		memset(&ra, 0, sizeof(ra));
		ra.qmeta = d_qm;
		ra.seg_cap = cap;
		ra.seg_count = d_cnt;
		ra.cand_doc = NULL;		/* the candidate's index is its handle */
		ra.cand_sc = d_sc;
		ra.doc_ids = d_ids;
		ra.k = limit;
		ra.out_ids = d_oid;
		ra.out_sc = d_osc;
		ra.out_count = d_ocnt;
		hipLaunchKernelGGL(k_replay<HEAP_REG>, dim3(nq), dim3(WAVE), 0, st, ra);
		if (hipGetLastError() != hipSuccess ||
		    hipMemcpyAsync(out_ids, d_oid, (size_t)nq * limit * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
		    hipMemcpyAsync(out_scores, d_osc, (size_t)nq * limit * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
		    hipMemcpyAsync(out_counts, d_ocnt, nq * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
		    hipStreamSynchronize(st) != hipSuccess) {
			set_error("merge failed: %s", hipGetErrorString(hipGetLastError()));
			break;
		}
		rc = 0;
	} while (0);
	(void)hipFree(ws);
	if (st) {
		(void)hipStreamDestroy(st);
	}
	return rc;
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
	(void)hipStreamSynchronize(ix->stream_down);
	(void)hipGetLastError();
	return -1;
}

static int comm_allgather_dev(nxsgpu_comm_t *, const void *, void *, size_t, hipStream_t);

static int
batch_begin(nxsgpu_index_t *ix, int algo, uint32_t limit, const nxsgpu_query_t *queries,
    uint32_t nq, const batch_out_t &o)
{
	const uint32_t seg_cap = ix->cfg.seg_cap;
	nxsgpu_index::dev_slot_t *sl = NULL;
	uint64_t total_post = 0;
	const bool gather = o.records && o.gather && ix->comm;
	const uint32_t world = gather ? (uint32_t)nxsgpu_comm_world(ix->comm) : 1u;
	const int my_rank = gather ? nxsgpu_comm_rank(ix->comm) : 0;

	if (limit == 0 || limit > NXSGPU_FAST_K) {
		set_error("device batches take limit 1..%d", NXSGPU_FAST_K);
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
	for (int i = 0; i < 2; i++) {
		if (!ix->slot[i].active) {
			sl = &ix->slot[i];
			break;
		}
	}
	if (!sl) {
		set_error("two batches are already in flight");
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
	for (int i = 0; i < 2; i++) {
		others = others || ix->slot[i].active;
	}
	const bool solo = nq <= 64 && !others && !gather;
	hipStream_t s_up = solo ? ix->stream : ix->stream_up;
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
	const bool block_in_ws = false;
	const bool block_on_host = o.records && !gather;

	/* plans straight into the pinned staging area (room for the work list:
	 * <= target + nq ranges, see build_worklist) */
	const uint64_t wave_target = ix->cfg.wave_target;
	const size_t seg_bound = (size_t)wave_target + 2 * (size_t)nq + 64;
	const size_t stage_need = 32768 + nq * (sizeof(dev_query_t) + sizeof(qmeta_t) + 16)
	    + seg_bound * (sizeof(item_t) + 8) + nq * 4 + sl->block_bytes;
	if (slot_ensure(*sl, 0, stage_need) != 0) {
		return -1;
	}
	uint8_t *hp = sl->h_stage;
	dev_query_t *h_q = carve<dev_query_t>(hp, nq);
	if (fill_dev_queries(ix, algo, queries, nq, h_q, total_post) != 0) {
		return -1;
	}
	build_worklist(ix, h_q, nq, wl, solo);
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
	uint8_t *h_block = carve<uint8_t>(hp, block_in_ws ? sl->block_bytes : 0);
	const size_t up_len = (size_t)(hp - sl->h_stage);
	uint32_t *h_status = block_in_ws ? (uint32_t *)(h_block + recs_len) : carve<uint32_t>(hp, o.n_slots);
	if (nq) {
		memcpy(h_qmeta, wl.qmeta.data(), nq * sizeof(qmeta_t));
		memcpy(h_items, wl.items.data(), nseg * sizeof(item_t));
		memcpy(h_bnd_q, wl.bnd_q.data(), (nseg + nq) * 4);
		memcpy(h_qorder, wl.qorder.data(), nq * 4);
		memset(h_ovf, 0, nq * 4);
		memset(h_pub, 0, nseg * 4);
	}
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
			memset(sl->h_blocks, 0, recs_len);
			h_status = (uint32_t *)(sl->h_blocks + recs_len);
		}
		if (o.status) {
			memcpy(h_status, o.status, (size_t)o.n_slots * 4);
		} else {
			memset(h_status, 0, (size_t)o.n_slots * 4);
		}
	}

	/* device workspace: the uploaded block first (same carve sequence => same
	 * offsets), then what only the kernels touch */
	const size_t ws_need = 32768 + up_len + nseg * 4
	    + (nseg + nq) * 4 * NXSGPU_MAX_TOKENS + nseg * (size_t)seg_cap * 8 + nseg * (16 * 4 + 64 * 4);
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
	uint8_t *d_myblock = carve<uint8_t>(p, block_in_ws ? sl->block_bytes : 0);
	uint32_t *d_seg_count = carve<uint32_t>(p, nseg);
	uint32_t *d_cursors = carve<uint32_t>(p, (nseg + nq) * NXSGPU_MAX_TOKENS);
	uint32_t *d_cand_doc = carve<uint32_t>(p, nseg * (size_t)seg_cap);
	float *d_cand_sc = carve<float>(p, nseg * (size_t)seg_cap);
	uint32_t *d_cold_state = carve<uint32_t>(p, nseg * 16);
	float *d_cold_top = carve<float>(p, nseg * 64);
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
		    (o.n_slots && hipMemcpyAsync(d_myblock + recs_len, h_status, (size_t)o.n_slots * 4,
		    hipMemcpyHostToDevice, s_up) != hipSuccess)) {
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
	memset(&ra, 0, sizeof(ra));
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
	if (!solo && (hipEventRecord(sl->ev_up, ix->stream_up) != hipSuccess ||
	    hipStreamWaitEvent(ix->stream, sl->ev_up, 0) != hipSuccess)) {
		set_error("query upload failed");
		return begin_fail(ix);
	}
	tc[2] = now_us();
	if (ix->profiling) (void)hipEventRecord(sl->ev_t[0], ix->stream);
	if (nq) {
		if (ix->cfg.one_replay) {
			launch_scan<MODE_TOPK>(ix, sa, wl);
			if (ix->profiling) (void)hipEventRecord(sl->ev_t[1], ix->stream);
			hipLaunchKernelGGL(k_replay<HEAP_REG>, dim3(nq), dim3(WAVE), 0, ix->stream, ra);
		} else {
			/* (profile: "replay" is then only what the last class's replay adds
			 * after the last scan) */
			launch_scan<MODE_TOPK>(ix, sa, wl, &ra, d_qorder, ix->profiling ? sl->ev_t[1] : NULL);
		}
	} else if (ix->profiling) {
		(void)hipEventRecord(sl->ev_t[1], ix->stream);
	}
	if (ix->profiling) (void)hipEventRecord(sl->ev_t[2], ix->stream);
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
		if (own_down && (hipEventRecord(sl->ev_res, ix->stream) != hipSuccess ||
		    hipStreamWaitEvent(s_down, sl->ev_res, 0) != hipSuccess)) {
			set_error("event failed");
			return begin_fail(ix);
		}
		if (gather) {
			if (comm_allgather_dev(ix->comm, d_myblock, sl->d_blocks, sl->block_bytes, s_down) != 0) {
				return begin_fail(ix);
			}
			if (hipMemcpyAsync(sl->h_blocks, sl->d_blocks, (size_t)world * sl->block_bytes,
			    hipMemcpyDeviceToHost, s_down) != hipSuccess) {
				set_error("copy failed");
				return begin_fail(ix);
			}
		} else if (!block_on_host && sl->block_bytes && hipMemcpyAsync(sl->h_blocks, d_myblock, sl->block_bytes,
		    hipMemcpyDeviceToHost, s_down) != hipSuccess) {
			set_error("copy failed");
			return begin_fail(ix);
		}
		if (hipEventRecord(sl->ev_done, s_down) != hipSuccess) {
			set_error("event failed");
			return begin_fail(ix);
		}
	}
	sl->postings = total_post;
	sl->active = true;
	if (ix->cfg.debug_timing) {
		tb3 = now_us();
		fprintf(stderr, "[nxsgpu begin #%llu] plan+worklist %.0f us, staging+alloc %.0f us, enqueue %.0f us "
		    "(upload %.0f, cursors %.0f, fork %.0f, scans+replays %.0f, tail %.0f)\n",
		    (unsigned long long)sl->seq, tb1 - tb0, tb2 - tb1, tb3 - tb2,
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

	for (int i = 0; i < 2; i++) {
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
	return (ix->slot[0].active ? 1 : 0) + (ix->slot[1].active ? 1 : 0);
}

extern "C" void
nxsgpu_index_reconfigure(nxsgpu_index_t *ix)
{
	cfg_from_env(ix->cfg);
}

extern "C" int
nxsgpu_search_dev(nxsgpu_index_t *ix, int algo, uint32_t limit, const nxsgpu_query_t *queries,
    uint32_t nq, uint64_t *d_doc_ids, float *d_scores, uint32_t *d_counts)
{
	if (ix->slot[0].active || ix->slot[1].active) {
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

/*
 * RCCL is loaded at first use (dlopen), so that a single-GPU consumer has no
 * link-time dependency on it and a process that already holds an RCCL (PyTorch
 * ships its own copy) keeps using that one.
 */
struct rccl_api_t {
	void *		handle;
	ncclResult_t	(*GetUniqueId)(ncclUniqueId *);
	ncclResult_t	(*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
	ncclResult_t	(*CommDestroy)(ncclComm_t);
	ncclResult_t	(*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
	const char *	(*GetErrorString)(ncclResult_t);
};

static rccl_api_t *
rccl_api(void)
{
	static rccl_api_t api;
	static bool tried = false;

	if (tried) {
		return api.handle ? &api : NULL;
	}
	tried = true;
	static const char *const names[] = { "librccl.so.1", "librccl.so" };
	void *h = NULL;
	for (int pass = 0; pass < 2 && !h; pass++) {
		for (size_t i = 0; i < 2 && !h; i++) {
			h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
		}
	}
	if (!h) {
		h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
	}
	if (!h) {
		set_error("cannot load librccl: %s", dlerror());
		return NULL;
	}
	api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
	api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
	api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
	api.AllGather = (decltype(api.AllGather))dlsym(h, "ncclAllGather");
	api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
	if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather) {
		set_error("librccl lacks an expected symbol");
		return NULL;
	}
	api.handle = h;
	return &api;
}

struct nxsgpu_comm {
	int		device, rank, world;
	ncclComm_t	comm;
	hipStream_t	stream;		/* blocking helper's own stream */
	void *		d_buf;
	size_t		d_len;
};

static const char *
rccl_err(rccl_api_t *R, ncclResult_t r)
{
	return R->GetErrorString ? R->GetErrorString(r) : "rccl error";
}

extern "C" int
nxsgpu_comm_unique_id(uint8_t uid[NXSGPU_UID_BYTES])
{
	rccl_api_t *R = rccl_api();
	ncclUniqueId id;
	ncclResult_t r;

	static_assert(sizeof(ncclUniqueId) == NXSGPU_UID_BYTES, "uid size");
	if (!R) {
		return -1;
	}
	if ((r = R->GetUniqueId(&id)) != ncclSuccess) {
		set_error("ncclGetUniqueId: %s", rccl_err(R, r));
		return -1;
	}
	memcpy(uid, &id, NXSGPU_UID_BYTES);
	return 0;
}

extern "C" nxsgpu_comm_t *
nxsgpu_comm_create(int device, int rank, int world, const uint8_t uid[NXSGPU_UID_BYTES])
{
	rccl_api_t *R = rccl_api();
	nxsgpu_comm_t *c;
	ncclUniqueId id;
	ncclResult_t r;

	if (!R) {
		return NULL;
	}
	if (world < 1 || rank < 0 || rank >= world) {
		set_error("bad rank %d of %d", rank, world);
		return NULL;
	}
	if (hipSetDevice(device) != hipSuccess) {
		set_error("hipSetDevice(%d) failed", device);
		return NULL;
	}
	c = new nxsgpu_comm();
	memset(c, 0, sizeof(*c));
	c->device = device;
	c->rank = rank;
	c->world = world;
	memcpy(&id, uid, NXSGPU_UID_BYTES);
	if ((r = R->CommInitRank(&c->comm, world, id, rank)) != ncclSuccess) {
		set_error("ncclCommInitRank(rank %d of %d): %s", rank, world, rccl_err(R, r));
		delete c;
		return NULL;
	}
	if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
		set_error("hipStreamCreate failed");
		(void)R->CommDestroy(c->comm);
		delete c;
		return NULL;
	}
	return c;
}

extern "C" void
nxsgpu_comm_destroy(nxsgpu_comm_t *c)
{
	rccl_api_t *R = rccl_api();

	if (!c) {
		return;
	}
	(void)hipSetDevice(c->device);
	if (c->stream) {
		(void)hipStreamSynchronize(c->stream);
		(void)hipStreamDestroy(c->stream);
	}
	(void)hipFree(c->d_buf);
	if (R && c->comm) {
		(void)R->CommDestroy(c->comm);
	}
	delete c;
}

extern "C" int nxsgpu_comm_rank(const nxsgpu_comm_t *c) { return c ? c->rank : 0; }
extern "C" int nxsgpu_comm_world(const nxsgpu_comm_t *c) { return c ? c->world : 1; }

/* device buffers, asynchronous on `stream`; recv holds world x bytes */
static int
comm_allgather_dev(nxsgpu_comm_t *c, const void *send, void *recv, size_t bytes, hipStream_t stream)
{
	rccl_api_t *R = rccl_api();
	ncclResult_t r;

	if (!R || !c) {
		set_error("no communicator");
		return -1;
	}
	if ((r = R->AllGather(send, recv, bytes, ncclChar, c->comm, stream)) != ncclSuccess) {
		set_error("ncclAllGather: %s", rccl_err(R, r));
		return -1;
	}
	return 0;
}

extern "C" int
nxsgpu_comm_allgather(nxsgpu_comm_t *c, const void *send, void *recv, size_t bytes)
{
	const size_t need = (size_t)(c->world + 1) * bytes + 512;
	uint8_t *d_send, *d_recv;

	if (hipSetDevice(c->device) != hipSuccess) {
		set_error("hipSetDevice failed");
		return -1;
	}
	if (bytes == 0) {
		return 0;
	}
	if (c->d_len < need) {
		(void)hipFree(c->d_buf);
		c->d_buf = NULL;
		c->d_len = 0;
		if (hipMalloc(&c->d_buf, need) != hipSuccess) {
			set_error("hipMalloc(%zu) failed", need);
			return -1;
		}
		c->d_len = need;
	}
	d_send = (uint8_t *)c->d_buf;
	d_recv = d_send + ((bytes + 255) & ~(size_t)255);
	g_err[0] = '\0';
	if (hipMemcpyAsync(d_send, send, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
	    comm_allgather_dev(c, d_send, d_recv, bytes, c->stream) != 0 ||
	    hipMemcpyAsync(recv, d_recv, (size_t)c->world * bytes, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
	    hipStreamSynchronize(c->stream) != hipSuccess) {
		if (!g_err[0]) {
			set_error("all-gather failed");
		}
		return -1;
	}
	return 0;
}

extern "C" int
nxsgpu_index_set_comm(nxsgpu_index_t *ix, nxsgpu_comm_t *c)
{
	if (ix->slot[0].active || ix->slot[1].active) {
		set_error("nxsgpu_index_set_comm: batches are in flight");
		return -1;
	}
	if (c && c->device != ix->device) {
		set_error("communicator and index live on different devices (%d, %d)", c->device, ix->device);
		return -1;
	}
	ix->comm = c;
	return 0;
}

extern "C" void
nxsgpu_results_free(nxsgpu_results_t *res)
{
	free(res->counts);
	free(res->offsets);
	free(res->doc_ids);
	free(res->scores);
	memset(res, 0, sizeof(*res));
}

/* ---- measured HBM read bandwidth ----------------------------------------------------- */

typedef uint32_t v4u_t __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256)
k_hbm_read(const v4u_t *__restrict__ src, uint64_t n16, uint32_t *__restrict__ sink)
{
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	uint32_t acc = 0;
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;

	/* four independent 16-byte loads in flight per lane */
	for (; i + 3 * stride < n16; i += 4 * stride) {
		const v4u_t a = __builtin_nontemporal_load(&src[i]);
		const v4u_t b = __builtin_nontemporal_load(&src[i + stride]);
		const v4u_t c = __builtin_nontemporal_load(&src[i + 2 * stride]);
		const v4u_t d = __builtin_nontemporal_load(&src[i + 3 * stride]);
		acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
	}
	for (; i < n16; i += stride) {
		const v4u_t a = __builtin_nontemporal_load(&src[i]);
		acc ^= a.x ^ a.y ^ a.z ^ a.w;
	}
	if (acc == 0x9e3779b9u) {	/* keeps the loads alive; practically never taken */
		atomicAdd(sink, 1u);
	}
}

/* the scan kernels' own access width: one 8-byte posting per lane and load
 * (global_load_dwordx2) -- the PMC calibration case (tools/pmc_calib.py) */
__global__ void __launch_bounds__(256)
k_hbm_read_x2(const uint2 *__restrict__ src, uint64_t n8, uint32_t *__restrict__ sink)
{
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	uint32_t acc = 0;
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;

	for (; i + 3 * stride < n8; i += 4 * stride) {
		const uint2 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
		acc ^= a.x ^ a.y ^ b.x ^ b.y ^ c.x ^ c.y ^ d.x ^ d.y;
	}
	for (; i < n8; i += stride) {
		const uint2 a = src[i];
		acc ^= a.x ^ a.y;
	}
	if (acc == 0x9e3779b9u) {
		atomicAdd(sink, 1u);
	}
}

/* one launch of each probe kernel over exactly `*bytes` bytes (returned): the
 * known byte count FETCH_SIZE is calibrated against */
extern "C" int
nxsgpu_hbm_calibrate(nxsgpu_index_t *ix, uint64_t *bytes_out)
{
	const uint64_t bytes = std::min<uint64_t>(ix->n_post * sizeof(posting_t), 2ull << 30) & ~(uint64_t)4095;
	uint32_t *d_sink = NULL;

	*bytes_out = bytes;
	if (bytes == 0 || hipSetDevice(ix->device) != hipSuccess || hipMalloc((void **)&d_sink, 4) != hipSuccess) {
		return -1;
	}
	(void)hipMemsetAsync(d_sink, 0, 4, ix->stream);
	hipLaunchKernelGGL(k_hbm_read, dim3(256 * 16), dim3(256), 0, ix->stream,
	    (const v4u_t *)ix->d_post[NXSGPU_BM25], bytes / 16, d_sink);
	hipLaunchKernelGGL(k_hbm_read_x2, dim3(256 * 16), dim3(256), 0, ix->stream,
	    (const uint2 *)ix->d_post[NXSGPU_TF_IDF], bytes / 8, d_sink);
	(void)hipStreamSynchronize(ix->stream);
	(void)hipFree(d_sink);
	return 0;
}

/* (its own kernel name: the FETCH_SIZE calibration sums k_hbm_read*'s counters) */
__global__ void __launch_bounds__(256)
k_stream_warm(const v4u_t *__restrict__ src, uint64_t n16, uint32_t *__restrict__ sink)
{
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	uint32_t acc = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
		const v4u_t a = __builtin_nontemporal_load(&src[i]);
		acc ^= a.x ^ a.y ^ a.z ^ a.w;
	}
	if (acc == 0x9e3779b9u) {
		atomicAdd(sink, 1u);
	}
}

/*
 * The HIP runtime creates its hardware queues lazily, the first time several
 * of a process's streams are busy at once -- a one-time stall of ~16 ms that
 * otherwise lands in whichever early batch first overlaps its neighbours
 * (measured: begin #3 or #6).  Pay it at index create: every stream of the
 * index gets a real kernel, all in flight together, twice.
 */
static void
warm_streams(nxsgpu_index_t *ix)
{
	hipStream_t st[] = { ix->stream, ix->stream2, ix->stream3, ix->stream_up, ix->stream_down, ix->stream_fz,
	    ix->xstream[0], ix->xstream[1], ix->xstream[2] };
	const uint64_t bytes = std::min<uint64_t>(ix->n_post * sizeof(posting_t), 512ull << 20) & ~(uint64_t)15;
	const size_t cb = 4u << 20;
	uint32_t *d_sink = NULL;
	uint8_t *h_buf = NULL, *d_buf = NULL;

	if (bytes < 4096 || hipMalloc((void **)&d_sink, 4) != hipSuccess) {
		return;
	}
	(void)hipMemset(d_sink, 0, 4);
	/* (the copy engines' queues are lazy too: uploads and downloads in flight
	 * together, on the streams that carry them later) */
	if (hipHostMalloc((void **)&h_buf, 2 * cb, hipHostMallocDefault) != hipSuccess ||
	    hipMalloc((void **)&d_buf, 2 * cb) != hipSuccess) {
		h_buf = NULL;
	}
	for (int round = 0; round < 3; round++) {
		for (hipStream_t s : st) {
			hipLaunchKernelGGL(k_stream_warm, dim3(1024), dim3(256), 0, s,
			    (const v4u_t *)ix->d_post[NXSGPU_BM25], bytes / 16, d_sink);
		}
		if (h_buf && d_buf) {
			(void)hipMemcpyAsync(d_buf, h_buf, cb, hipMemcpyHostToDevice, ix->stream_up);
			(void)hipMemcpyAsync(h_buf + cb, d_buf + cb, cb, hipMemcpyDeviceToHost, ix->stream_down);
			(void)hipMemcpyAsync(h_buf + cb, d_buf + cb, 4096, hipMemcpyDeviceToHost, ix->stream);
			(void)hipMemsetAsync(d_buf, 0, 4096, ix->stream_up);
		}
	}
	/* ... and so are the runtime's pools of completion signals: a few thousand
	 * event records / cross-stream waits / small copies queued without a sync in
	 * between, the depth two batches in flight reach */
	{
		hipEvent_t ev[8];
		int n_ev = 0;
		for (; n_ev < 8; n_ev++) {
			if (hipEventCreateWithFlags(&ev[n_ev], hipEventDisableTiming) != hipSuccess) {
				break;
			}
		}
		for (int i = 0; n_ev == 8 && i < 512; i++) {
			hipStream_t sa = st[i % 6], sb = st[(i + 1 + i / 6) % 6];
			(void)hipEventRecord(ev[i & 7], sa);
			(void)hipStreamWaitEvent(sb, ev[i & 7], 0);
			if (h_buf && d_buf) {
				(void)hipMemcpyAsync(h_buf + cb + (size_t)(i & 63) * 4096, d_buf + cb, 4096,
				    hipMemcpyDeviceToHost, sb);
			}
			if ((i & 63) == 63) {
				hipLaunchKernelGGL(k_stream_warm, dim3(64), dim3(256), 0, sb,
				    (const v4u_t *)ix->d_post[NXSGPU_BM25], (uint64_t)4096, d_sink);
			}
		}
		for (hipStream_t s2 : st) {
			(void)hipStreamSynchronize(s2);
		}
		for (int i = 0; i < n_ev; i++) {
			(void)hipEventDestroy(ev[i]);
		}
	}
	for (hipStream_t s : st) {
		(void)hipStreamSynchronize(s);
	}
	(void)hipGetLastError();
	(void)hipFree(d_sink);
	(void)hipFree(d_buf);
	if (h_buf) {
		(void)hipHostFree(h_buf);
	}
}

extern "C" double
nxsgpu_hbm_read_gbs(nxsgpu_index_t *ix, int reps)
{
	/* at most 4 GiB of the BM25 posting array: far beyond the 256 MiB Infinity Cache */
	const uint64_t bytes = std::min<uint64_t>(ix->n_post * sizeof(posting_t), 4ull << 30) & ~(uint64_t)15;
	uint32_t *d_sink = NULL;
	hipEvent_t e0 = NULL, e1 = NULL;
	double best = 0.0;

	if (bytes < (64u << 20) || hipSetDevice(ix->device) != hipSuccess) {
		return 0.0;
	}
	if (hipMalloc((void **)&d_sink, 4) != hipSuccess || hipEventCreate(&e0) != hipSuccess ||
	    hipEventCreate(&e1) != hipSuccess) {
		goto out;
	}
	(void)hipMemsetAsync(d_sink, 0, 4, ix->stream);
	for (int r = 0; r < (reps < 1 ? 1 : reps) + 1; r++) {
		float ms = 0;
		(void)hipEventRecord(e0, ix->stream);
		hipLaunchKernelGGL(k_hbm_read, dim3(256 * 16), dim3(256), 0, ix->stream,
		    (const v4u_t *)ix->d_post[NXSGPU_BM25], bytes / 16, d_sink);
		(void)hipEventRecord(e1, ix->stream);
		if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) {
			break;
		}
		if (r && ms > 0) {	/* first launch warms up */
			best = std::max(best, (double)bytes / (ms * 1e-3) / 1e9);
		}
	}
out:
	(void)hipFree(d_sink);
	if (e0) (void)hipEventDestroy(e0);
	if (e1) (void)hipEventDestroy(e1);
	return best;
}

/* ---- wide queries (beyond nxsgpu_query_t) ------------------------------------------ */

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
			hipLaunchKernelGGL(k_replay<HEAP_LDS>, dim3(nq), dim3(WAVE), (size_t)ra.k * 8, ix->stream, ra);
		} else {
			hipLaunchKernelGGL(k_replay<HEAP_GLOBAL>, dim3(nq), dim3(WAVE), 0, ix->stream, ra);
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

/* ---- fuzzy ----------------------------------------------------------- */

/*
 * Match-first search of all tokens at once (tokens of <= 64 bytes only).
 * 0 = term_ids filled, 1 = a queue overflowed (the caller takes the
 * level-by-level search), -1 = error.
 */
static int
fuzzy_match_first(nxsgpu_index_t *ix, const uint8_t *tok_bytes, const uint32_t *tok_off, uint32_t n_tok,
    uint32_t *term_ids)
{
	const uint32_t n_c = ix->n_fz;
	const uint32_t blen = tok_off[n_tok] - tok_off[0];
	/* FZ_NQ sub-queues of qcap survivors each */
	/* (what a sub-queue can receive at most: its workgroups x 256 nodes x the
	 * tokens of a grid.y slice -- a small tree fills few sub-queues) */
	const uint32_t gy = std::max<uint32_t>(1, std::min<uint32_t>(8, n_tok / 128));
	const uint64_t q_most = (uint64_t)gy * (((n_c + 255) / 256 + FZ_NQ - 1) / FZ_NQ) * 256 * ((n_tok + gy - 1) / gy);
	const uint64_t qcap = std::min<uint64_t>(0xffffffffu, std::max<uint64_t>((n_tok + FZ_NQ - 1) / FZ_NQ + 1,
	    std::min<uint64_t>(ix->cfg.fuzzy_cand / FZ_NQ, q_most)));
	const uint64_t ccap = qcap * FZ_NQ;
	const uint64_t mcap = std::max<uint64_t>(1024, ccap / 4);
	const size_t need = 16384 + FZ_NQ * FZ_CSTRIDE * 4 + (ccap + mcap) * sizeof(fz_item_t) + (size_t)n_tok * (256 * 8 + 8 + 4 + 4 + 4) + 64 + blen + 16 +
	    ((size_t)n_tok + 1) * 4 + (NXS_MYERS_MAXPAT + 4) * 4 + 16 * 256;
	uint32_t h_cnt[4] = { 0, 0, 0, 0 };
	std::vector<uint32_t> h_qcnt(FZ_NQ * FZ_CSTRIDE);
	unsigned long long h_evals = 0;
	/* one upload: token offsets, rank of every token in the length-sorted order,
	 * first rank of every length */
	std::vector<uint32_t> up((size_t)n_tok + 1 + n_tok + NXS_MYERS_MAXPAT + 2);
	uint32_t *roff = up.data(), *rank = roff + n_tok + 1, *len_off = rank + n_tok;

	if (ix->fz_len < need) {
		if (ix->fz) {
			(void)hipFree(ix->fz);
			ix->fz = NULL;
			ix->fz_len = 0;
		}
		if (hipMalloc(&ix->fz, need) != hipSuccess) {
			set_error("hipMalloc(%zu) for the fuzzy workspace failed", need);
			return -1;
		}
		ix->fz_len = need;
	}
	uint8_t *p = (uint8_t *)ix->fz;
	fz_item_t *d_cand = carve<fz_item_t>(p, ccap);
	fz_item_t *d_match = carve<fz_item_t>(p, mcap);
	uint32_t *d_cnt = carve<uint32_t>(p, 4);		/* -, matches, overflow, (seed's count) */
	uint32_t *d_qcnt = carve<uint32_t>(p, FZ_NQ * FZ_CSTRIDE);	/* survivors per sub-queue */
	unsigned long long *d_evals = carve<unsigned long long>(p, 1);
	uint64_t *d_peq = carve<uint64_t>(p, (size_t)n_tok * 256);
	uint2 *d_tokf = carve<uint2>(p, (size_t)n_tok + 4);
	uint32_t *d_best = carve<uint32_t>(p, n_tok);
	uint32_t *d_tids = carve<uint32_t>(p, n_tok);
	uint8_t *d_bytes = carve<uint8_t>(p, blen + 16);
	uint32_t *d_up = carve<uint32_t>(p, up.size());
	uint32_t *d_off = d_up, *d_rank = d_up + n_tok + 1, *d_len_off = d_rank + n_tok;
	hipStream_t st = ix->stream_fz;
	fz_args_t fa;

	for (uint32_t i = 0; i <= n_tok; i++) {
		roff[i] = tok_off[i] - tok_off[0];
	}
	for (uint32_t l = 0; l <= NXS_MYERS_MAXPAT + 1; l++) {
		len_off[l] = 0;
	}
	for (uint32_t i = 0; i < n_tok; i++) {
		len_off[roff[i + 1] - roff[i] + 1]++;		/* (every token is <= 64 bytes here) */
	}
	for (uint32_t l = 0; l <= NXS_MYERS_MAXPAT; l++) {
		len_off[l + 1] += len_off[l];
	}
	{
		uint32_t next[NXS_MYERS_MAXPAT + 2];
		memcpy(next, len_off, sizeof(next));
		for (uint32_t i = 0; i < n_tok; i++) {
			rank[i] = next[roff[i + 1] - roff[i]]++;
		}
	}
	if (hipMemcpyAsync(d_bytes, tok_bytes + tok_off[0], blen, hipMemcpyHostToDevice, st) != hipSuccess ||
	    hipMemcpyAsync(d_up, up.data(), up.size() * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
	    hipMemsetAsync(d_cnt, 0, 16, st) != hipSuccess ||
	    hipMemsetAsync(d_qcnt, 0, FZ_NQ * FZ_CSTRIDE * 4, st) != hipSuccess ||
	    hipMemsetAsync(d_evals, 0, 8, st) != hipSuccess ||
	    hipMemsetAsync(d_tokf + n_tok, 0xff, 4 * sizeof(uint2), st) != hipSuccess) {
		set_error("fuzzy upload failed");
		return -1;
	}
	if (ix->profiling) (void)hipEventRecord(ix->ev[0], st);
	hipLaunchKernelGGL(k_bk_peq, dim3(n_tok), dim3(256), 0, st, d_bytes, d_off, n_tok, d_peq, d_tokf, d_rank);
	hipLaunchKernelGGL(k_bk_seed, dim3((n_tok + 255) / 256), dim3(256), 0, st, d_cand, d_cnt + 3, n_tok, d_best,
	    (unsigned long long *)NULL);
	if (n_c) {
		hipLaunchKernelGGL(k_fz_filter, dim3((n_c + 255) / 256, gy), dim3(256), 0, st,
		    ix->d_fz_sig, ix->d_fz_node, ix->d_fz_len, n_c, d_tokf, d_len_off, d_cand, d_qcnt,
		    (uint32_t)qcap, d_cnt + 2);
	}
	memset(&fa, 0, sizeof(fa));
	fa.bk = ix->d_bk;
	fa.bk_bytes = ix->d_bk_bytes;
	fa.tok_bytes = d_bytes;
	fa.tok_off = d_off;
	fa.peq = d_peq;
	fa.cap = (uint32_t)std::min<uint64_t>(ccap, 0xffffffffu);
	fa.best = d_best;
	fa.overflow = d_cnt + 2;
	fa.prune = 1;
	fa.evals = ix->profiling ? d_evals : NULL;
	hipLaunchKernelGGL(k_fz_dist, dim3(8, FZ_NQ), dim3(1024), 0, st, fa, d_cand, d_qcnt, (uint32_t)qcap, d_match, d_cnt + 1,
	    (uint32_t)std::min<uint64_t>(mcap, 0xffffffffu));
	hipLaunchKernelGGL(k_fz_chain, dim3(1024), dim3(256), 0, st, fa, ix->d_bk_parent, ix->d_bk_slot, d_match, d_cnt + 1,
	    (uint32_t)std::min<uint64_t>(mcap, 0xffffffffu));
	hipLaunchKernelGGL(k_bk_finish, dim3((n_tok + 255) / 256), dim3(256), 0, st, ix->d_bk, d_best, n_tok, d_tids);
	if (ix->profiling) (void)hipEventRecord(ix->ev[1], st);
	if (hipGetLastError() != hipSuccess) {
		set_error("fuzzy kernel launch failed");
		return -1;
	}
	if (hipMemcpyAsync(term_ids, d_tids, (size_t)n_tok * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
	    hipMemcpyAsync(h_cnt, d_cnt, 16, hipMemcpyDeviceToHost, st) != hipSuccess ||
	    (ix->profiling && hipMemcpyAsync(h_qcnt.data(), d_qcnt, h_qcnt.size() * 4, hipMemcpyDeviceToHost, st) != hipSuccess) ||
	    hipMemcpyAsync(&h_evals, d_evals, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
	    hipStreamSynchronize(st) != hipSuccess) {
		set_error("fuzzy pass failed: %s", hipGetErrorString(hipGetLastError()));
		return -1;
	}
	if (ix->profiling) {
		float ms = 0;
		(void)hipEventElapsedTime(&ms, ix->ev[0], ix->ev[1]);
		ix->prof.fuzzy_ms += ms;
	}
	if (h_cnt[2]) {
		return 1;
	}
	if (ix->profiling) {
		/* distance evaluations; "pairs" = what the queues carried; levels: pairs
		 * screened, survivors, matches */
		uint64_t surv = 0;
		for (uint32_t q = 0; q < FZ_NQ; q++) {
			surv += h_qcnt[q * FZ_CSTRIDE];
		}
		ix->prof.fuzzy_visits += h_evals;
		ix->prof.fuzzy_pairs += surv + h_cnt[1];
		ix->prof.fuzzy_level[0] += (uint64_t)n_tok * n_c;
		ix->prof.fuzzy_level[1] += surv;
		ix->prof.fuzzy_level[2] += h_cnt[1];
	}
	return 0;
}

extern "C" int
nxsgpu_fuzzy(nxsgpu_index_t *ix, const uint8_t *tok_bytes, const uint32_t *tok_off,
    uint32_t n_tok, uint32_t *term_ids, uint64_t *visited)
{
	const uint64_t budget = ix->cfg.fuzzy_items;
	const uint32_t n_bk = ix->n_bk;
	uint32_t chunk, max_len = 0;
	bool any_long = false;

	if (n_tok == 0) {
		return 0;
	}
	if (n_bk == 0) {
		memset(term_ids, 0, n_tok * sizeof(uint32_t));
		if (visited) memset(visited, 0, n_tok * sizeof(uint64_t));
		return 0;
	}
	if (hipSetDevice(ix->device) != hipSuccess) {
		set_error("hipSetDevice failed");
		return -1;
	}
	for (uint32_t i = 0; i < n_tok; i++) {
		const uint32_t m = tok_off[i + 1] - tok_off[i];
		max_len = std::max(max_len, m);
		if (m > NXS_MYERS_MAXPAT) {
			any_long = true;
		}
	}
	/*
	 * Worst case one token visits every node: with `safe_chunk` tokens per pass
	 * the frontier queues can never overflow.  A d <= 2 search visits ~10 % of a
	 * large tree, though, so the whole batch is tried in ONE pass first (29
	 * level launches instead of 29 per chunk, and fuller levels); a pass that
	 * does overflow the queues is repeated with a quarter of the tokens, down to
	 * the safe size.
	 */
	/* the usual case: no visit counts wanted, every token fits the bit-vector
	 * distance -- match first, then reachability; the frontier search below is
	 * what the reference does, step for step, and the fallback */
	if (!visited && !ix->cfg.fuzzy_bfs && !ix->cfg.fuzzy_noprune && ix->d_bk_parent) {
		if (!any_long) {
			const int r = fuzzy_match_first(ix, tok_bytes, tok_off, n_tok, term_ids);
			if (r <= 0) {
				return r;
			}
		} else if (!ix->fz_split) {
			/* tokens beyond the bit-vector distance (> 64 bytes) take the frontier
			 * search with its row DP, the others the match-first search */
			std::vector<uint32_t> sel[2], off[2], ids[2];
			std::vector<uint8_t> bytes[2];
			int rc = 0;
			for (uint32_t i = 0; i < n_tok; i++) {
				const uint32_t m = tok_off[i + 1] - tok_off[i];
				const int w = m > NXS_MYERS_MAXPAT;
				if (sel[w].empty()) {
					off[w].push_back(0);
				}
				sel[w].push_back(i);
				bytes[w].insert(bytes[w].end(), tok_bytes + tok_off[i], tok_bytes + tok_off[i + 1]);
				off[w].push_back((uint32_t)bytes[w].size());
			}
			ix->fz_split = true;
			for (int w = 0; w < 2 && rc == 0; w++) {
				if (!sel[w].empty()) {
					ids[w].resize(sel[w].size());
					bytes[w].resize(bytes[w].size() + 16);
					rc = nxsgpu_fuzzy(ix, bytes[w].data(), off[w].data(), (uint32_t)sel[w].size(), ids[w].data(), NULL);
					for (size_t j = 0; j < sel[w].size(); j++) {
						term_ids[sel[w][j]] = ids[w][j];
					}
				}
			}
			ix->fz_split = false;
			return rc;
		}
	}
	const uint32_t safe_chunk = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n_tok, budget / n_bk));
	chunk = ix->cfg.fuzzy_safe ? safe_chunk : n_tok;

	const uint32_t LONG_THREADS = 64 * 64;
	const uint64_t cap = std::max<uint64_t>((uint64_t)safe_chunk * n_bk, std::min<uint64_t>(budget, (uint64_t)chunk * n_bk));
	const size_t levels = (size_t)ix->bk_depth + 2;
	size_t need = 4096 + cap * sizeof(fz_item_t) * 2 + levels * 4 + 256
	    + (size_t)chunk * (256 * 8 + 4 + 8 + 4) + tok_off[n_tok] + 16 + ((size_t)chunk + 1) * 4 + 512
	    + (any_long ? (size_t)LONG_THREADS * ((size_t)max_len + 2) * 2 : 0) + 16 * 256;
	if (ix->fz_len < need) {
		if (ix->fz) {
			(void)hipFree(ix->fz);
			ix->fz = NULL;
			ix->fz_len = 0;
		}
		if (hipMalloc(&ix->fz, need) != hipSuccess) {
			set_error("hipMalloc(%zu) for the fuzzy workspace failed", need);
			return -1;
		}
		ix->fz_len = need;
	}

	for (uint32_t c0 = 0; c0 < n_tok; ) {
		const uint32_t nc = std::min(chunk, n_tok - c0);
		const uint32_t boff = tok_off[c0], blen = tok_off[c0 + nc] - boff;
		std::vector<uint32_t> roff(nc + 1);
		uint8_t *p = (uint8_t *)ix->fz;
		fz_item_t *qa = carve<fz_item_t>(p, cap);
		fz_item_t *qb = carve<fz_item_t>(p, cap);
		uint32_t *counts = carve<uint32_t>(p, levels);
		uint32_t *d_ovf = carve<uint32_t>(p, 1);
		uint64_t *d_peq = carve<uint64_t>(p, (size_t)nc * 256);
		uint32_t *d_best = carve<uint32_t>(p, nc);
		unsigned long long *d_vis = carve<unsigned long long>(p, nc);
		unsigned long long *d_evals = carve<unsigned long long>(p, 1);
		uint32_t *d_tids = carve<uint32_t>(p, nc);
		uint8_t *d_bytes = carve<uint8_t>(p, blen + 16);
		uint32_t *d_off = carve<uint32_t>(p, nc + 1);
		uint16_t *d_rows = any_long ? carve<uint16_t>(p, (size_t)LONG_THREADS * (max_len + 2)) : NULL;
		fz_args_t fa;
		uint32_t h_ovf = 0;

		for (uint32_t i = 0; i <= nc; i++) {
			roff[i] = tok_off[c0 + i] - boff;
		}
		if (hipMemcpyAsync(d_bytes, tok_bytes + boff, blen, hipMemcpyHostToDevice, ix->stream_fz) != hipSuccess ||
		    hipMemcpyAsync(d_off, roff.data(), (nc + 1) * 4, hipMemcpyHostToDevice, ix->stream_fz) != hipSuccess ||
		    hipMemsetAsync(counts, 0, levels * 4, ix->stream_fz) != hipSuccess ||
		    hipMemsetAsync(d_ovf, 0, 4, ix->stream_fz) != hipSuccess ||
		    hipMemsetAsync(d_evals, 0, 8, ix->stream_fz) != hipSuccess) {
			set_error("fuzzy upload failed");
			return -1;
		}
		if (ix->profiling) (void)hipEventRecord(ix->ev[0], ix->stream_fz);
		hipLaunchKernelGGL(k_bk_peq, dim3(nc), dim3(256), 0, ix->stream_fz, d_bytes, d_off, nc, d_peq, (uint2 *)NULL, (const uint32_t *)NULL);
		hipLaunchKernelGGL(k_bk_seed, dim3((nc + 255) / 256), dim3(256), 0, ix->stream_fz,
		    qa, counts, nc, d_best, visited ? d_vis : (unsigned long long *)NULL);

		memset(&fa, 0, sizeof(fa));
		fa.bk = ix->d_bk;
		fa.bk_bytes = ix->d_bk_bytes;
		fa.tok_bytes = d_bytes;
		fa.tok_off = d_off;
		fa.peq = d_peq;
		fa.cap = (uint32_t)std::min<uint64_t>(cap, 0xffffffffu);
		fa.best = d_best;
		fa.visited = visited ? d_vis : NULL;
		fa.dp_rows = d_rows;
		fa.dp_stride = max_len + 2;
		fa.overflow = d_ovf;
		fa.prune = (!visited && !ix->cfg.fuzzy_noprune) ? 1u : 0u;
		fa.evals = ix->profiling ? d_evals : NULL;
		for (uint32_t lvl = 0; lvl < ix->bk_depth; lvl++) {
			fa.cur = (lvl & 1) ? qb : qa;
			fa.next = (lvl & 1) ? qa : qb;
			fa.cur_count = counts + lvl;
			fa.next_count = counts + lvl + 1;
			hipLaunchKernelGGL(k_bk_level<false>, dim3(512), dim3(1024), 0, ix->stream_fz, fa);
			if (any_long) {
				/* tokens longer than 64 bytes: row DP, bounded scratch */
				hipLaunchKernelGGL(k_bk_level<true>, dim3(LONG_THREADS / 64), dim3(64), 0, ix->stream_fz, fa);
			}
		}
		hipLaunchKernelGGL(k_bk_finish, dim3((nc + 255) / 256), dim3(256), 0, ix->stream_fz,
		    ix->d_bk, d_best, nc, d_tids);
		if (ix->profiling) (void)hipEventRecord(ix->ev[1], ix->stream_fz);
		if (hipGetLastError() != hipSuccess) {
			set_error("fuzzy kernel launch failed");
			return -1;
		}
		std::vector<uint32_t> h_counts(levels);
		unsigned long long h_evals = 0;
		if (hipMemcpyAsync(term_ids + c0, d_tids, nc * 4, hipMemcpyDeviceToHost, ix->stream_fz) != hipSuccess ||
		    hipMemcpyAsync(&h_evals, d_evals, 8, hipMemcpyDeviceToHost, ix->stream_fz) != hipSuccess ||
		    (visited && hipMemcpyAsync(visited + c0, d_vis, nc * 8, hipMemcpyDeviceToHost, ix->stream_fz) != hipSuccess) ||
		    hipMemcpyAsync(&h_ovf, d_ovf, 4, hipMemcpyDeviceToHost, ix->stream_fz) != hipSuccess ||
		    hipMemcpyAsync(h_counts.data(), counts, levels * 4, hipMemcpyDeviceToHost, ix->stream_fz) != hipSuccess ||
		    hipStreamSynchronize(ix->stream_fz) != hipSuccess) {
			set_error("fuzzy pass failed: %s", hipGetErrorString(hipGetLastError()));
			return -1;
		}
		if (ix->profiling) {
			float ms = 0;
			(void)hipEventElapsedTime(&ms, ix->ev[0], ix->ev[1]);
			ix->prof.fuzzy_ms += ms;		/* a repeated pass is time spent too */
		}
		if (h_ovf) {
			if (chunk <= safe_chunk) {
				set_error("fuzzy frontier overflow (internal error)");
				return -1;
			}
			chunk = std::max(safe_chunk, chunk / 4);
			continue;		/* same tokens again, fewer at a time */
		}
		if (ix->profiling) {
			/* distance evaluations; (token, node) pairs dequeued, pruned ones
			 * included, are the level counts */
			ix->prof.fuzzy_visits += h_evals;
			for (size_t l = 0; l < levels; l++) {
				ix->prof.fuzzy_pairs += h_counts[l];
				if (l < 40) {
					ix->prof.fuzzy_level[l] += h_counts[l];
				}
			}
		}
		c0 += nc;
	}
	return 0;
}

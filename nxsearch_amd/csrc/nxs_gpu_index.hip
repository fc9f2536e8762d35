/*
 * nxs_gpu_index.hip -- device index: build / refresh kernels, create, apply (N1), destroy, HBM probes
 * (MI355X / gfx950 query path of nxsearch; see nxs_gpu_int.h for the map of the files)
 */
#include "nxs_gpu_int.h"

/* ------------------------------------------------------------------ */
/* error handling                                                      */
/* ------------------------------------------------------------------ */

static thread_local char g_err[512];

void
set_error(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}

void
clear_error(void)
{
	g_err[0] = '\0';
}

bool
have_error(void)
{
	return g_err[0] != '\0';
}

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

void
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
	c.max_post = u64("NXS_GPU_MAXPOST", 65536, 1024, 1ull << 40);
	c.min_post = u64("NXS_GPU_MINPOST", 4096, 1, ~0ull);
	c.min_post_solo = u64("NXS_GPU_MINPOST_SOLO", 512, 1, ~0ull);
	c.scanm_dens = dbl("NXS_GPU_SCANM_DENS", 0.08);
	c.scanm_minnt = (uint32_t)u64("NXS_GPU_SCANM_MINNT", 2, 2, 8);
	c.scanm_maxnt = (uint32_t)u64("NXS_GPU_SCANM_MAXNT", 8, 2, 8);
	c.rmin = on("NXS_GPU_NOSCANR2") ? 3u : 2u;
	c.seg_cap = (uint32_t)u64("NXS_GPU_SEGCAP", SEG_CAP_DEFAULT, 1, 1u << 16);
	c.scan1_split = (uint32_t)u64("NXS_GPU_SCAN1_SPLIT", 64, 1, 1u << 30);
	c.seg_cap_big = (uint32_t)u64("NXS_GPU_SEGCAP_BIG", 0, 0, 1u << 20);
	c.big_minpost = u64("NXS_GPU_BIG_MINPOST", 16, 0, 1u << 20);
	c.fuzzy_items = u64("NXS_GPU_FUZZY_ITEMS", 256ull << 20, 1, 1ull << 32);
	c.use_scanr = !on("NXS_GPU_NOSCANR");
	c.mask_off = !on("NXS_GPU_NOMASKOFF");
	c.by_level = !on("NXS_GPU_NOLEVELS");
	c.use_scanm = !on("NXS_GPU_NOSCANM");
	c.use_blkmap = !on("NXS_GPU_NOBLKMAP");
	c.bm_share = u64("NXS_GPU_BM_SHARE", 1024, 1, 1u << 30);
	c.bm_gain = dbl("NXS_GPU_BM_GAIN", 16.0);
	c.bigq_em = dbl("NXS_GPU_BIGQ_EM", 16.0);
	c.use_scans = !on("NXS_GPU_NOSCANS");
	c.use_scans_drop = on("NXS_GPU_SCANS_DROP");
	c.scans_workpct = u64("NXS_GPU_SCANS_WORKPCT", 70, 10, 400);
	c.wave_target_scans = u64("NXS_GPU_WAVES_SCANS", std::min<uint64_t>(c.wave_target, 57344), 1, 1u << 22);
	c.use_scanb = !on("NXS_GPU_NOSCANB");
	c.scanb_dens = dbl("NXS_GPU_SCANB_DENS", 0.01);
	c.replay_join = on("NXS_GPU_REPLAY_JOIN");
	c.tfidf_drop = !on("NXS_GPU_TFIDF_NODROP");
	c.outl_share = (uint32_t)u64("NXS_GPU_OUTL_SHARE", 8, 2, 1u << 20);
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
	c.drop_tiles = on("NXS_GPU_DROP_TILES");
	c.no_straggler = on("NXS_GPU_NOSTRAGGLER");
	c.drop_split = (uint32_t)u64("NXS_GPU_DROP_SPLIT", 1, 0, 64);
	c.drop_early = !on("NXS_GPU_DROP_NOEARLY");
	c.drop_b = on("NXS_GPU_DROPB");
	c.and_early = !on("NXS_GPU_AND_NOEARLY");
	c.debug_timing = on("NXS_GPU_DEBUG_TIMING");
	c.down_inline = on("NXS_GPU_DOWN_INLINE");
	c.old_replay = on("NXS_GPU_OLDREPLAY");
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
		if (out_bm25) {
			out_bm25[i] = pb;
		}
		if (out_tfidf) {
			out_tfidf[i] = pt;
		}
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

/* ... and as a byte: ceil(255 x impact / the term's largest impact), never 0 for a posting (the margin
 * keeps q8 x max / 255 >= impact whatever the divisions round to) */
__global__ void
k_dense_fill_q8(const posting_t *__restrict__ post, uint64_t beg, uint64_t end, float scale, uint8_t *__restrict__ col)
{
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t i = beg + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < end; i += stride) {
		const posting_t p = post[i];
		const float q = ceilf(p.imp * scale * 1.00001f);
		col[p.doc] = (uint8_t)fminf(fmaxf(q, 1.0f), 255.0f);
	}
}

/*
 * Block-presence bitmap + rank directory of one list (d_blkmap / d_bmrank rows).  A wavefront
 * reads 64 consecutive postings (coalesced); docs ascend, so the postings of one 4096-doc word
 * are a RUN of lanes: the run's bits are OR-ed together by a segmented scan across the lanes
 * and its last lane issues the one atomic (runs continue in the neighbouring wavefronts).  A
 * posting whose predecessor lies in an earlier word records its position for its word and for
 * the empty words in between (the rank directory).
 */
__global__ void
k_blkmap_fill(const uint64_t *__restrict__ post_dt, const uint64_t *__restrict__ rows_beg,
    const uint64_t *__restrict__ rows_end, uint64_t words, unsigned long long *__restrict__ blkmap,
    uint32_t *__restrict__ bmrank)
{
	const uint32_t row = blockIdx.y;
	const unsigned lane = threadIdx.x & 63;
	const uint64_t beg = rows_beg[row], end = rows_end[row];
	unsigned long long *bm = blkmap + (uint64_t)row * words;
	uint32_t *rk = bmrank + (uint64_t)row * (words + 1);
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;

	for (uint64_t i0 = beg + ((uint64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u)); i0 < end; i0 += stride) {
		const uint64_t i = i0 + lane;
		const bool valid = i < end;
		const uint32_t doc = valid ? (uint32_t)(post_dt[i] >> 32) : 0xffffffffu;
		const uint64_t w = valid ? doc >> 12 : ~0ull;
		/* word of the predecessor (lane 0: the posting before the window) */
		uint64_t pw = (uint64_t)__shfl_up((long long)w, 1);
		if (lane == 0) {
			pw = i0 > beg ? (post_dt[i0 - 1] >> 32) >> 12 : ~0ull;
		}
		unsigned long long acc = valid ? 1ull << ((doc >> 6) & 63) : 0;
		/* segmented inclusive OR over the run of equal words (runs are contiguous) */
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const unsigned long long v = (unsigned long long)__shfl_up((long long)acc, o);
			const uint64_t wv = (uint64_t)__shfl_up((long long)w, o);
			if ((int)lane >= o && wv == w) {
				acc |= v;
			}
		}
		const uint64_t nw = (uint64_t)__shfl_down((long long)w, 1);
		if (valid && (lane == 63 || nw != w)) {
			atomicOr(&bm[w], acc);		/* the run's last lane */
		}
		if (valid && w != pw) {
			/* first posting of word w: it starts w and every empty word after pw */
			for (uint64_t e = (pw == ~0ull ? 0 : pw + 1); e <= w; e++) {
				rk[e] = (uint32_t)(i - beg);
			}
		}
		if (valid && i + 1 == end) {
			for (uint64_t e = w + 1; e <= words; e++) {
				rk[e] = (uint32_t)(end - beg);
			}
		}
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
				if (out_bm25) {
					out_bm25[i] = pb;
				}
				if (out_tfidf) {
					out_tfidf[i] = pt;
				}
				bb_ = (out_bm25 && pb.imp > 0.0f) ? __float_as_uint(pb.imp) : 0u;
				bt_ = (out_tfidf && pt.imp > 0.0f) ? __float_as_uint(pt.imp) : 0u;
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
				if (bb_ && bb_ > max_bm25[t]) {
					atomicMax(&max_bm25[t], bb_);
				}
				if (bt_ && bt_ > max_tfidf[t]) {
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

/* buffer `which` of the exact path, at least `need` bytes (NULL: out of memory) */
void *
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
void
xbuf_put(nxsgpu_index_t *ix, int which)
{
	if (ix->xbuf_len[which] > X_KEEP_MAX) {
		(void)hipFree(ix->xbuf[which]);
		ix->xbuf[which] = NULL;
		ix->xbuf_len[which] = 0;
	}
}

bool
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

bool
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
	(void)hipFree(ix->d_post_dt_spare);
	(void)hipFree(ix->d_post[0]);
	(void)hipFree(ix->d_post[1]);
	(void)hipFree(ix->d_dense_col[0]);
	(void)hipFree(ix->d_dense_col[1]);
	(void)hipFree(ix->d_dense_q8);
	(void)hipFree(ix->d_blkmap);
	(void)hipFree(ix->d_bmrank);
	(void)hipFree(ix->d_bk);
	(void)hipFree(ix->d_bk_bytes);
	bk_aux_free(ix);
	(void)hipFree(ix->ws);
	(void)hipFree(ix->fz);
	if (ix->h_pin) {
		(void)hipHostFree(ix->h_pin);
	}
	for (int i = 0; i < NXSGPU_FZ_SLOTS; i++) {
		if (ix->fzs[i].pin) {
			(void)hipHostFree(ix->fzs[i].pin);
		}
		if (ix->fzs[i].ws) {
			(void)hipFree(ix->fzs[i].ws);
		}
		for (int j = 0; j < 4; j++) {
			if (ix->fzs[i].ev[j]) {
				(void)hipEventDestroy(ix->fzs[i].ev[j]);
			}
		}
		if (ix->fzs[i].ev_done) {
			(void)hipEventDestroy(ix->fzs[i].ev_done);
		}
	}
	for (int i = 0; i < 4; i++) {
		if (ix->ev[i]) {
			(void)hipEventDestroy(ix->ev[i]);
		}
	}
	for (int i = 0; i < NXSGPU_INFLIGHT; i++) {
		nxsgpu_index::dev_slot_t &sl = ix->slot[i];
		if (sl.active && sl.ev_done) {
			(void)hipEventSynchronize(sl.ev_done);
		}
		(void)hipFree(sl.ws);
		if (sl.h_stage) {
			(void)hipHostFree(sl.h_stage);
		}
		if (sl.ev_up) (void)hipEventDestroy(sl.ev_up);
		if (sl.ev_ahead) (void)hipEventDestroy(sl.ev_ahead);
		if (sl.ev_early) (void)hipEventDestroy(sl.ev_early);
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
	for (int i = 0; i < 3; i++) {
		if (ix->down_spare[i]) {
			(void)hipStreamDestroy(ix->down_spare[i]);
		}
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
	if (ix->stream_rp[1]) {	/* ([0] is stream3, [2] is stream_up) */
		(void)hipStreamDestroy(ix->stream_rp[1]);
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

/* elements of d_post[a]: the postings, then (TF-IDF) room for the dense terms' outlier lists */
static size_t
post_elems(const nxsgpu_index_t *ix, int a)
{
	return (size_t)ix->cap_post + (a == NXSGPU_TF_IDF ? (size_t)(ix->cap_post / 8 + 4096) : 0);
}

/*
 * TF-IDF outlier lists (nxsgpu_index::outl_off): tf histogram of one dense term's
 * postings, then an order-preserving compaction of the postings above the cap --
 * per-block counts, one scan, per-block writes.
 */
#define	OUTL_CHUNK	4096		/* postings per block (256 threads x 16) */
__global__ void __launch_bounds__(256)
k_tf_hist(const uint64_t *__restrict__ dt, uint64_t n, uint32_t *__restrict__ hist)
{
	__shared__ uint32_t s_h[64];
	if (threadIdx.x < 64) {
		s_h[threadIdx.x] = 0;
	}
	__syncthreads();
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		atomicAdd(&s_h[min((uint32_t)dt[i], 63u)], 1u);
	}
	__syncthreads();
	if (threadIdx.x < 64 && s_h[threadIdx.x]) {
		atomicAdd(&hist[threadIdx.x], s_h[threadIdx.x]);
	}
}

__global__ void __launch_bounds__(256)
k_outl_count(const uint64_t *__restrict__ dt, uint64_t n, uint32_t tf_cap, uint32_t *__restrict__ cnt)
{
	__shared__ uint32_t s_n;
	if (threadIdx.x == 0) {
		s_n = 0;
	}
	__syncthreads();
	const uint64_t b0 = (uint64_t)blockIdx.x * OUTL_CHUNK + (uint64_t)threadIdx.x * 16;
	uint32_t c = 0;
	for (int k = 0; k < 16; k++) {
		c += (b0 + k < n && (uint32_t)dt[b0 + k] > tf_cap) ? 1u : 0u;
	}
	if (c) {
		atomicAdd(&s_n, c);
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		cnt[blockIdx.x] = s_n;
	}
}

/* exclusive scan of cnt[0..nb) in place, total to cnt[nb]: one block */
__global__ void __launch_bounds__(1024)
k_outl_scan(uint32_t *__restrict__ cnt, uint32_t nb)
{
	__shared__ uint32_t s_part[1024];
	const uint32_t per = (nb + 1023) / 1024;
	const uint32_t lo = threadIdx.x * per, hi = min(lo + per, nb);
	uint32_t sum = 0;
	for (uint32_t i = lo; i < hi; i++) {
		sum += cnt[i];
	}
	s_part[threadIdx.x] = sum;
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t run = 0;
		for (int i = 0; i < 1024; i++) {
			const uint32_t v = s_part[i];
			s_part[i] = run;
			run += v;
		}
		cnt[nb] = run;
	}
	__syncthreads();
	uint32_t run = s_part[threadIdx.x];
	for (uint32_t i = lo; i < hi; i++) {
		const uint32_t v = cnt[i];
		cnt[i] = run;
		run += v;
	}
}

__global__ void __launch_bounds__(256)
k_outl_write(const uint64_t *__restrict__ dt, const posting_t *__restrict__ imp, uint64_t n, uint32_t tf_cap,
    float cap_imp, const uint32_t *__restrict__ cnt, posting_t *__restrict__ out, uint32_t *__restrict__ max_excess)
{
	__shared__ uint32_t s_c[256];
	const uint64_t b0 = (uint64_t)blockIdx.x * OUTL_CHUNK + (uint64_t)threadIdx.x * 16;
	uint32_t c = 0;
	for (int k = 0; k < 16; k++) {
		c += (b0 + k < n && (uint32_t)dt[b0 + k] > tf_cap) ? 1u : 0u;
	}
	s_c[threadIdx.x] = c;
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t run = cnt[blockIdx.x];
		for (int i = 0; i < 256; i++) {
			const uint32_t v = s_c[i];
			s_c[i] = run;
			run += v;
		}
	}
	__syncthreads();
	uint32_t o = s_c[threadIdx.x], mx = 0;
	for (int k = 0; k < 16; k++) {
		if (b0 + k < n && (uint32_t)dt[b0 + k] > tf_cap) {
			posting_t p = imp[b0 + k];
			p.imp = p.imp - cap_imp;	/* > 0: the impact grows with tf */
			mx = max(mx, __float_as_uint(p.imp));
			out[o++] = p;
		}
	}
	if (mx) {
		atomicMax(max_excess, mx);
	}
}

/*
 * Impacts of every posting from the CSR form (d_post_off, d_post_dt) and the
 * header statistics: host libm tables (the device only does IEEE + - * / on
 * them), one k_impacts_csr pass, the per-term maxima back to the host.  Used by
 * the first build and by every refresh (N, adl and df move every idf).
 */
int
rebuild_impacts(nxsgpu_index_t *ix, unsigned only)
{
	/* the ranking functions to (re)compute: materialised ones only */
	const bool do_b = (only & (1u << NXSGPU_BM25)) && ix->algo_on[NXSGPU_BM25];
	const bool do_t = (only & (1u << NXSGPU_TF_IDF)) && ix->algo_on[NXSGPU_TF_IDF];
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
	if (do_b || !ix->algo_on[NXSGPU_BM25]) {
		ix->h_maximp[NXSGPU_BM25].assign((size_t)T + 2, 0.0f);
	}
	if (do_t || !ix->algo_on[NXSGPU_TF_IDF]) {
		ix->h_maximp[NXSGPU_TF_IDF].assign((size_t)T + 2, 0.0f);
	}
	if (P == 0 || (!do_b && !do_t)) {
		if (only == 3) {
			ix->bm_terms.clear();
		}
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
	    do_b ? ix->d_post[NXSGPU_BM25] : (posting_t *)NULL, do_t ? ix->d_post[NXSGPU_TF_IDF] : (posting_t *)NULL,
	    d_maximp, d_maximp + (size_t)T + 2);
	HIP_TRY(hipGetLastError());
	if (do_b) {
		HIP_TRY(hipMemcpyAsync(ix->h_maximp[NXSGPU_BM25].data(), d_maximp, ((size_t)T + 2) * 4,
		    hipMemcpyDeviceToHost, ix->stream));
	}
	if (do_t) {
		HIP_TRY(hipMemcpyAsync(ix->h_maximp[NXSGPU_TF_IDF].data(), d_maximp + (size_t)T + 2, ((size_t)T + 2) * 4,
		    hipMemcpyDeviceToHost, ix->stream));
	}
	HIP_TRY(hipStreamSynchronize(ix->stream));
	/* impact columns of the dense terms (at most 64, densest first), per materialised
	 * ranking function -- the class that reads them is (k_scanm<.., DROP>, fill_dev_queries) */
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
		/* (the same list whichever function is rebuilt: it depends on the df only) */
		ix->dense_terms.clear();
		for (auto &e : dn) {
			ix->dense_terms.push_back(e.second);
		}
		std::sort(ix->dense_terms.begin(), ix->dense_terms.end());
		const uint64_t words = (uint64_t)ix->dense_terms.size() * ix->n_docs;
		for (int a = 0; a < 2; a++) {
			if (!(a == NXSGPU_BM25 ? do_b : do_t) || (a == NXSGPU_TF_IDF && !ix->cfg.tfidf_drop)) {
				continue;
			}
			if (words > ix->dense_cap[a] || (!words && ix->dense_cap[a])) {
				(void)hipFree(ix->d_dense_col[a]);
				ix->d_dense_col[a] = NULL;
				ix->dense_cap[a] = 0;
				if (words) {
					const uint64_t cap = words + words / 16 + 1024;
					HIP_TRY(hipMalloc((void **)&ix->d_dense_col[a], cap * 4));
					ix->dense_cap[a] = cap;
				}
			}
			if (words) {
				HIP_TRY(hipMemsetAsync(ix->d_dense_col[a], 0xff, words * 4, ix->stream));
				for (size_t c = 0; c < ix->dense_terms.size(); c++) {
					const uint32_t t = ix->dense_terms[c];
					hipLaunchKernelGGL(k_dense_fill, dim3(1024), dim3(256), 0, ix->stream,
					    ix->d_post[a], ix->h_post_off[t], ix->h_post_off[t + 1],
					    ix->d_dense_col[a] + c * ix->n_docs);
				}
				HIP_TRY(hipGetLastError());
				HIP_TRY(hipStreamSynchronize(ix->stream));
			}
			if (a == NXSGPU_BM25 && ix->cfg.use_scans_drop) {
				/* the byte form (k_scans<.., DROP>'s map fill: opt-in, NXS_GPU_SCANS_DROP) */
				const uint64_t row = ((ix->n_docs + 16383) & ~(uint64_t)16383) + 16384;
				const uint64_t bytes = (uint64_t)ix->dense_terms.size() * row;
				if (bytes > ix->dense_q8_cap || (!bytes && ix->dense_q8_cap)) {
					(void)hipFree(ix->d_dense_q8);
					ix->d_dense_q8 = NULL;
					ix->dense_q8_cap = 0;
					if (bytes) {
						const uint64_t cap = bytes + bytes / 16 + 65536;
						HIP_TRY(hipMalloc((void **)&ix->d_dense_q8, cap));
						ix->dense_q8_cap = cap;
					}
				}
				ix->dense_q8_stride = row;
				if (bytes) {
					HIP_TRY(hipMemsetAsync(ix->d_dense_q8, 0, bytes, ix->stream));
					for (size_t c = 0; c < ix->dense_terms.size(); c++) {
						const uint32_t t = ix->dense_terms[c];
						const float mx = ix->h_maximp[NXSGPU_BM25][t];
						hipLaunchKernelGGL(k_dense_fill_q8, dim3(1024), dim3(256), 0, ix->stream,
						    ix->d_post[a], ix->h_post_off[t], ix->h_post_off[t + 1],
						    mx > 0.0f ? 255.0f / mx : 0.0f, ix->d_dense_q8 + c * row);
					}
					HIP_TRY(hipGetLastError());
					HIP_TRY(hipStreamSynchronize(ix->stream));
				}
			}
		}
	}
	/* block-presence bitmaps + rank directories of the longer lists (they depend on the
	 * postings alone: rebuilt at build and refresh, not when a second ranking function is
	 * materialised) */
	if (only == 3) {
		const uint64_t words = (ix->n_docs + 4095) / 4096;
		const uint64_t min_df = std::max<uint64_t>(1, ix->n_docs / std::max<uint64_t>(ix->cfg.bm_share, 1));
		std::vector<std::pair<uint64_t, uint32_t>> bt;
		ix->bm_terms.clear();
		ix->bm_words = words;
		if (ix->cfg.use_blkmap && ix->n_docs < (1ull << 31) && ix->n_post < (1ull << 32)) {
			for (uint32_t t = 1; t <= T; t++) {
				const uint64_t df = ix->h_post_off[t + 1] - ix->h_post_off[t];
				if (df >= min_df) {
					bt.push_back(std::make_pair(df, t));
				}
			}
			std::sort(bt.begin(), bt.end(), [](const std::pair<uint64_t, uint32_t> &x, const std::pair<uint64_t, uint32_t> &y) {
				return x.first != y.first ? x.first > y.first : x.second < y.second;
			});
			if (bt.size() > 8192) {
				bt.resize(8192);
			}
			for (auto &e : bt) {
				ix->bm_terms.push_back(e.second);
			}
			std::sort(ix->bm_terms.begin(), ix->bm_terms.end());
		}
		const size_t rows = ix->bm_terms.size();
		if (rows) {
			const uint64_t need = (uint64_t)rows * (words + 1);
			uint64_t *d_rows = NULL;
			if (need > ix->bm_cap) {
				(void)hipFree(ix->d_blkmap);
				(void)hipFree(ix->d_bmrank);
				ix->d_blkmap = NULL;
				ix->d_bmrank = NULL;
				ix->bm_cap = 0;
				const uint64_t cap = need + need / 8 + 64;
				HIP_TRY(hipMalloc((void **)&ix->d_blkmap, cap * 8));
				HIP_TRY(hipMalloc((void **)&ix->d_bmrank, cap * 4));
				ix->bm_cap = cap;
			}
			std::vector<uint64_t> rb(2 * rows);
			for (size_t r = 0; r < rows; r++) {
				rb[r] = ix->h_post_off[ix->bm_terms[r]];
				rb[rows + r] = ix->h_post_off[ix->bm_terms[r] + 1];
			}
			HIP_TRY(hipMalloc((void **)&d_rows, rb.size() * 8));
			if (hipMemcpyAsync(d_rows, rb.data(), rb.size() * 8, hipMemcpyHostToDevice, ix->stream) != hipSuccess ||
			    hipMemsetAsync(ix->d_blkmap, 0, (uint64_t)rows * words * 8, ix->stream) != hipSuccess) {
				(void)hipFree(d_rows);
				set_error("block bitmaps: upload failed");
				goto fail;
			}
			hipLaunchKernelGGL(k_blkmap_fill, dim3(64, (unsigned)rows), dim3(256), 0, ix->stream,
			    ix->d_post_dt, d_rows, d_rows + rows, words, (unsigned long long *)ix->d_blkmap, ix->d_bmrank);
			const hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(ix->stream);
			(void)hipFree(d_rows);
			if (e1 != hipSuccess || e2 != hipSuccess) {
				set_error("k_blkmap_fill failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
				goto fail;
			}
		}
	}
	/* TF-IDF: cap + outlier list per dense term (nxsgpu_index::outl_off).  Two host
	 * round trips for all terms together: the tf histograms, then the lists. */
	if (do_t) {
		const size_t nc = ix->cfg.tfidf_drop ? ix->dense_terms.size() : 0;
		const uint64_t room = post_elems(ix, NXSGPU_TF_IDF) - ix->cap_post;
		uint32_t *d_hist = NULL, *d_cnt = NULL, *d_res = NULL;	/* d_res: [nc] largest excess, [nc] postings written */
		std::vector<uint32_t> hist(nc * 64), res(2 * nc, 0), tf_cap(nc, 0);
		std::vector<uint64_t> above(nc, 0);
		uint64_t at = ix->cap_post, max_nb = 1;
		bool ok = true;

		ix->outl_off.assign(nc + 1, at);
		ix->outl_cap.assign(nc, 0.0f);
		ix->outl_max.assign(nc, 0.0f);
		for (size_t c = 0; c < nc; c++) {
			const uint32_t t = ix->dense_terms[c];
			max_nb = std::max<uint64_t>(max_nb, (ix->h_post_off[t + 1] - ix->h_post_off[t] + OUTL_CHUNK - 1) / OUTL_CHUNK);
		}
		if (nc) {
			ok = hipMalloc((void **)&d_hist, nc * 64 * 4) == hipSuccess &&
			    hipMalloc((void **)&d_cnt, (max_nb + 1) * 4) == hipSuccess &&
			    hipMalloc((void **)&d_res, 2 * nc * 4) == hipSuccess &&
			    hipMemsetAsync(d_hist, 0, nc * 64 * 4, ix->stream) == hipSuccess &&
			    hipMemsetAsync(d_res, 0, 2 * nc * 4, ix->stream) == hipSuccess;
		}
		for (size_t c = 0; c < nc && ok; c++) {
			const uint32_t t = ix->dense_terms[c];
			const uint64_t p0 = ix->h_post_off[t], n = ix->h_post_off[t + 1] - p0;
			const uint32_t nb = (uint32_t)((n + OUTL_CHUNK - 1) / OUTL_CHUNK);
			hipLaunchKernelGGL(k_tf_hist, dim3(std::min<uint32_t>(nb, 1024)), dim3(256), 0, ix->stream,
			    ix->d_post_dt + p0, n, d_hist + c * 64);
		}
		if (nc && ok) {
			ok = hipMemcpyAsync(hist.data(), d_hist, nc * 64 * 4, hipMemcpyDeviceToHost, ix->stream) == hipSuccess &&
			    hipStreamSynchronize(ix->stream) == hipSuccess;
		}
		for (size_t c = 0; c < nc && ok; c++) {
			const uint32_t t = ix->dense_terms[c];
			const uint64_t p0 = ix->h_post_off[t], n = ix->h_post_off[t + 1] - p0;
			const uint32_t nb = (uint32_t)((n + OUTL_CHUNK - 1) / OUTL_CHUNK);
			const uint32_t *h = &hist[c * 64];
			uint64_t ab = 0;
			uint32_t cap = 1;

			ix->outl_cap[c] = ix->h_maximp[NXSGPU_TF_IDF][t];
			ix->outl_off[c] = at;
			/* the smallest tf that all but 1/outl_share of the postings stay at or below */
			for (int k = 63; k >= 1; k--) {
				ab += h[k];
			}
			for (; cap < 62 && ab - h[cap] > n / ix->cfg.outl_share; cap++) {
				ab -= h[cap];
			}
			ab -= h[cap];			/* postings with tf > cap */
			if (ab == 0 || cap >= 62 || at + ab > ix->cap_post + room) {
				continue;		/* no outliers (or no room): the cap is the largest impact */
			}
			const float cap_imp = (float)logtf[cap] * idf_t[t];	/* k_impacts_csr's own expression */
			tf_cap[c] = cap;
			above[c] = ab;
			hipLaunchKernelGGL(k_outl_count, dim3(nb), dim3(256), 0, ix->stream, ix->d_post_dt + p0, n, cap, d_cnt);
			hipLaunchKernelGGL(k_outl_scan, dim3(1), dim3(1024), 0, ix->stream, d_cnt, nb);
			hipLaunchKernelGGL(k_outl_write, dim3(nb), dim3(256), 0, ix->stream, ix->d_post_dt + p0,
			    ix->d_post[NXSGPU_TF_IDF] + p0, n, cap, cap_imp, d_cnt, ix->d_post[NXSGPU_TF_IDF] + at, d_res + c);
			ok = hipMemcpyAsync(d_res + nc + c, d_cnt + nb, 4, hipMemcpyDeviceToDevice, ix->stream) == hipSuccess;
			ix->outl_cap[c] = cap_imp;
			at += ab;
		}
		ix->outl_off[nc] = at;
		if (nc && ok) {
			ok = hipMemcpyAsync(res.data(), d_res, 2 * nc * 4, hipMemcpyDeviceToHost, ix->stream) == hipSuccess &&
			    hipStreamSynchronize(ix->stream) == hipSuccess;
		}
		for (size_t c = 0; c < nc && ok; c++) {
			memcpy(&ix->outl_max[c], &res[c], 4);
			ok = res[nc + c] == above[c];		/* the compaction wrote what the histogram promised */
		}
		(void)tf_cap;
		(void)hipFree(d_hist);
		(void)hipFree(d_cnt);
		(void)hipFree(d_res);
		if (!ok) {
			ix->outl_off.assign(nc + 1, ix->cap_post);	/* no outlier lists: the caps must not be used */
			ix->outl_cap.clear();
			set_error("outlier lists of the dense terms failed");
			goto fail;
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


/*
 * An index has ONE default ranking function (params.db "algo"); the other one's
 * impacts (8 B per posting: 2.6 GB at 10M docs) are computed on the first search
 * that names it: one k_impacts_csr pass into a new array, beside whatever batches
 * are in flight (nothing they read is touched).
 */
int
ensure_algo(nxsgpu_index_t *ix, int algo)
{
	if (algo != NXSGPU_BM25 && algo != NXSGPU_TF_IDF) {
		set_error("invalid algorithm");
		return -1;
	}
	if (ix->algo_on[algo]) {
		return 0;
	}
	if (hipSetDevice(ix->device) != hipSuccess) {
		set_error("hipSetDevice failed");
		return -1;
	}
	const uint64_t cap = post_elems(ix, algo);
	if (hipMalloc(&ix->d_post[algo], cap * sizeof(posting_t)) != hipSuccess) {
		ix->d_post[algo] = NULL;
		set_error("hipMalloc(%llu) for the impacts of ranking function %d failed",
		    (unsigned long long)(cap * sizeof(posting_t)), algo);
		return -1;
	}
	ix->algo_on[algo] = true;
	if (rebuild_impacts(ix, 1u << algo) != 0) {
		(void)hipFree(ix->d_post[algo]);
		ix->d_post[algo] = NULL;
		ix->algo_on[algo] = false;
		return -1;
	}
	return 0;
}

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
	/*
	 * HIP deals its hardware queues (four by default) out round-robin in stream
	 * creation order, and work on streams that share a queue runs one after the
	 * other.  Which streams must NOT share one (measured: a C3 step took 2.0 ms
	 * instead of 1.2 when the dense-term class's stream sat on the upload stream's
	 * queue; with 8 or 16 queues -- GPU_MAX_HW_QUEUES -- every stream has its own and
	 * the step takes 1.7: some serialisation helps) decides the order here:
	 *   queue A: stream (scans), stream2 (replays of a class: short), stream_fz
	 *   queue B: stream3 (dense-term class; limits > 64: replays of batch slot 0),
	 *            xstream[0], xstream[2]
	 *   queue C: stream_up (uploads + k_cursors of the NEXT batch; limits > 64: replays of every
	 *            third batch -- such batches upload on the scan stream), xstream[1]
	 *   queue D: stream_rp[1] (limits > 64: replays of batch slot 1), stream_down (record
	 *            blocks of a sharded batch: all-gather + copy, beside the next batch)
	 */
	HIP_TRY(hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking));
	HIP_TRY(hipStreamCreateWithFlags(&ix->stream3, hipStreamNonBlocking));
	HIP_TRY(hipStreamCreateWithFlags(&ix->stream_up, hipStreamNonBlocking));
	HIP_TRY(hipStreamCreateWithFlags(&ix->stream_rp[1], hipStreamNonBlocking));
	HIP_TRY(hipStreamCreateWithFlags(&ix->stream2, hipStreamNonBlocking));
	HIP_TRY(hipStreamCreateWithFlags(&ix->xstream[0], hipStreamNonBlocking));
	HIP_TRY(hipStreamCreateWithFlags(&ix->xstream[1], hipStreamNonBlocking));
	/* (queue D: a sharded batch's all-gather waits there for the batch's last replay -- on
	 * queue B it would hold up the NEXT batch's dense-term class, queued behind it) */
	HIP_TRY(hipStreamCreateWithFlags(&ix->stream_down, hipStreamNonBlocking));
	/*
	 * The fuzzy passes of batch i + 1 run beside batch i's scans, and the HOST waits for them in _begin
	 * (the plans need the resolved terms): at the highest stream priority their workgroups are dispatched
	 * ahead of the scans' as CUs free up.  (Round 2 measured 8-10 instead of 27-32 ms of waiting per C5
	 * step and left it off -- the wait was hidden behind 38 ms of device work then; since round 4 a C5 step
	 * is 30 ms of which the host waits 24 for its fuzzy tokens.  NXS_GPU_FZ_NOPRIO: plain priority.)
	 */
	{
		int lo_p = 0, hi_p = 0;
		if (getenv("NXS_GPU_FZ_NOPRIO") || hipDeviceGetStreamPriorityRange(&lo_p, &hi_p) != hipSuccess ||
		    hipStreamCreateWithPriority(&ix->stream_fz, hipStreamNonBlocking, hi_p) != hipSuccess) {
			(void)hipGetLastError();
			HIP_TRY(hipStreamCreateWithFlags(&ix->stream_fz, hipStreamNonBlocking));
		}
	}
	HIP_TRY(hipStreamCreateWithFlags(&ix->xstream[2], hipStreamNonBlocking));
	ix->stream_rp[0] = ix->stream3;		/* (the dense-term class does not exist for limits > 64) */
	/* (queue C: MODE_BIG batches send their plans up on the scan stream, so nothing of theirs waits
	 * behind a replay there.  A stream created here would land on a queue of the runtime's choosing --
	 * measured: B, behind slot 0's replays) */
	ix->stream_rp[2] = ix->stream_up;
	HIP_TRY(hipEventCreateWithFlags(&ix->ev_fork3, hipEventDisableTiming));
	HIP_TRY(hipEventCreateWithFlags(&ix->ev_join3, hipEventDisableTiming));
	for (int i = 0; i < NXSGPU_INFLIGHT; i++) {
		HIP_TRY(hipEventCreateWithFlags(&ix->slot[i].ev_up, hipEventDisableTiming));
		HIP_TRY(hipEventCreateWithFlags(&ix->slot[i].ev_ahead, hipEventDisableTiming));
		HIP_TRY(hipEventCreateWithFlags(&ix->slot[i].ev_early, hipEventDisableTiming));
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
		for (int j = 0; j < NXSGPU_FZ_SLOTS; j++) {
			HIP_TRY(hipEventCreate(&ix->fzs[j].ev[i]));
		}
	}
	for (int j = 0; j < NXSGPU_FZ_SLOTS; j++) {
		HIP_TRY(hipEventCreateWithFlags(&ix->fzs[j].ev_done, hipEventDisableTiming));
	}

	/* room for appended docs (N1) without moving the tables */
	ix->cap_docs_ids = ix->cap_docs_len = D + D / 8 + 4096;
	HIP_TRY(hipMalloc(&ix->d_doc_ids, ix->cap_docs_ids * 8));
	HIP_TRY(hipMalloc(&ix->d_doc_len, ix->cap_docs_len * 4));
	HIP_TRY(hipMalloc(&ix->d_post_off, ((size_t)T + 2) * 8));
	/* (room to grow: the first appended document must not pay for new arrays) */
	ix->cap_post = P + P / 16 + 4096;
	HIP_TRY(hipMalloc(&ix->d_post_dt, ix->cap_post * 8));
	ix->cap_post_dt = ix->cap_post;
	/* the buffer the first refresh merges into: taken now, so that no refresh pays for a
	 * multi-GB allocation (not fatal if it cannot be had: nxsgpu_index_apply tries again) */
	if (hipMalloc(&ix->d_post_dt_spare, ix->cap_post * 8) == hipSuccess) {
		ix->cap_post_dt_spare = ix->cap_post;
	} else {
		(void)hipGetLastError();
		ix->d_post_dt_spare = NULL;
	}
	for (int a = 0; a < 2; a++) {
		/* (the other ranking function's impacts: on its first search, ensure_algo) */
		ix->algo_on[a] = src->default_algo < 0 || src->default_algo == a;
		if (ix->algo_on[a]) {
			HIP_TRY(hipMalloc(&ix->d_post[a], post_elems(ix, a) * sizeof(posting_t)));
		}
	}
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
	pick_record_stream(ix);
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

	auto now_ms = []() -> double {
		struct timespec ts;
		clock_gettime(CLOCK_MONOTONIC, &ts);
		return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec;
	};
	const double t_a0 = now_ms();
	double t_a1 = t_a0, t_a2 = t_a0;
	if (nxsgpu_batches_in_flight(ix)) {
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
		{
			const hipError_t se = hipStreamSynchronize(ix->stream);
			(void)hipFree(tmp2);
			HIP_TRY(se);
		}
		/* a posting that was not found would corrupt the merge */
		uint64_t last = 0;
		HIP_TRY(hipMemcpy(&last, d_dead_sorted + (n_dead - 1), 8, hipMemcpyDeviceToHost));
		if (last == ~0ull) {
			set_error("nxsgpu_index_apply: a removed doc's posting is not in the index");
			goto fail;
		}
	}
	t_a1 = now_ms();
	{
		const uint64_t P_new = P_old - n_dead + n_newp;
		const unsigned blocks = (unsigned)(((uint64_t)T_new + 2 + 255) / 256);

		/* the merge writes the other CSR buffer: kept between refreshes (a fresh 2.6 GB
		 * hipMalloc at 10M docs costs more than the merge itself) */
		uint64_t cap_out = ix->cap_post_dt_spare;
		d_out = ix->d_post_dt_spare;
		ix->d_post_dt_spare = NULL;
		ix->cap_post_dt_spare = 0;
		if (!d_out || cap_out < P_new) {
			(void)hipFree(d_out);
			d_out = NULL;
			cap_out = P_new + P_new / 16 + 4096;
			HIP_TRY(hipMalloc(&d_out, cap_out * 8));
		}
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
		/* swap in the new CSR; the old buffer is the next refresh's target */
		ix->d_post_dt_spare = ix->d_post_dt;
		ix->cap_post_dt_spare = ix->cap_post_dt;
		ix->d_post_dt = d_out;
		ix->cap_post_dt = cap_out;
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
			for (int a = 0; a < 2; a++) {
				if (ix->algo_on[a]) {
					HIP_TRY(hipMalloc(&ix->d_post[a], post_elems(ix, a) * sizeof(posting_t)));
				}
			}
		}
	}
	t_a2 = now_ms();
	if (rebuild_impacts(ix) != 0) {
		goto fail;
	}
	if (ix->cfg.debug_timing) {
		fprintf(stderr, "[nxsgpu apply] delta %.1f ms, merge %.1f ms, impacts + columns %.1f ms\n",
		    t_a1 - t_a0, t_a2 - t_a1, now_ms() - t_a2);
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
	/* the per-class events of both batch slots: created here, not inside a timed batch */
	for (int i = 0; on && i < NXSGPU_INFLIGHT; i++) {
		nxsgpu_index::dev_slot_t &sl = ix->slot[i];
		if (!sl.ev_cls_ok && hipSetDevice(ix->device) == hipSuccess) {
			bool ok = true;
			for (int c = 0; c < NXSGPU_PROF_CLS && ok; c++) {
				ok = hipEventCreate(&sl.ev_cls[c][0]) == hipSuccess && hipEventCreate(&sl.ev_cls[c][1]) == hipSuccess;
			}
			sl.ev_cls_ok = ok;
		}
	}
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

/* ---- N4: doc-sharded mode ------------------------------------------------------------ */

extern "C" int
nxsgpu_index_set_global_df(nxsgpu_index_t *ix, const uint32_t *df, uint32_t n_terms)
{
	if (nxsgpu_batches_in_flight(ix)) {
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
	    (const v4u_t *)ix->d_post_dt, bytes / 16, d_sink);
	hipLaunchKernelGGL(k_hbm_read_x2, dim3(256 * 16), dim3(256), 0, ix->stream,
	    (const uint2 *)ix->d_post_dt, bytes / 8, d_sink);
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
void
warm_streams(nxsgpu_index_t *ix)
{
	hipStream_t st[] = { ix->stream, ix->stream2, ix->stream3, ix->stream_up, ix->stream_down, ix->stream_fz,
	    ix->xstream[0], ix->xstream[1], ix->xstream[2], ix->stream_rp[1] };
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
			    (const v4u_t *)ix->d_post_dt, bytes / 16, d_sink);
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
				    (const v4u_t *)ix->d_post_dt, (uint64_t)4096, d_sink);
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

/*
 * Which hardware queue a stream lands on is the runtime's choice (the creation order only decides it for the
 * first four streams: traced, stream_down shared the SCAN stream's queue).  For the record stream of sharded
 * batches that matters: it holds a device-side wait for the batch's last replay, and a wait at the head of a
 * hardware queue blocks every stream behind it -- the next batch's scans.  So: measure.  A bounded spin kernel
 * (400 us) on a reference stream, a trivial kernel on the candidate; if the candidate's kernel is done while
 * the spin still runs, the two do not share a queue.  The first candidate -- stream_down itself, then up to
 * three spare streams -- that runs beside the scan stream, the dense-term stream AND the upload stream
 * becomes stream_down; with none, sharded batches keep the records on the scan stream (down_inline).
 */
__global__ void __launch_bounds__(64)
k_queue_probe(uint32_t ticks, uint32_t *sink)
{
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();	/* 100 MHz */
	while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
		__builtin_amdgcn_s_sleep(32);
	}
	if (ticks == 0xffffffffu) {
		*sink = 1;
	}
}

static bool
runs_beside(hipStream_t ref, hipStream_t x, hipEvent_t e_ref, hipEvent_t e_x, uint32_t *d_sink)
{
	hipLaunchKernelGGL(k_queue_probe, dim3(1), dim3(64), 0, ref, 40000u, d_sink);	/* 400 us: a busy host must not make the candidate look late */
	(void)hipEventRecord(e_ref, ref);
	hipLaunchKernelGGL(k_queue_probe, dim3(1), dim3(64), 0, x, 0u, d_sink);
	(void)hipEventRecord(e_x, x);
	(void)hipEventSynchronize(e_x);
	const bool beside = hipEventQuery(e_ref) == hipErrorNotReady;
	(void)hipEventSynchronize(e_ref);
	return beside;
}

void
pick_record_stream(nxsgpu_index_t *ix)
{
	hipEvent_t e_ref = NULL, e_x = NULL;
	uint32_t *d_sink = NULL;
	const hipStream_t refs[3] = { ix->stream, ix->stream3, ix->stream_up };

	if (ix->cfg.down_inline || hipMalloc((void **)&d_sink, 4) != hipSuccess) {
		return;
	}
	if (hipEventCreateWithFlags(&e_ref, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&e_x, hipEventDisableTiming) != hipSuccess) {
		goto out;
	}
	for (int c = 0; c <= 3; c++) {
		hipStream_t cand = ix->stream_down;
		bool ok = true;

		if (c > 0) {
			if (hipStreamCreateWithFlags(&ix->down_spare[c - 1], hipStreamNonBlocking) != hipSuccess) {
				ix->down_spare[c - 1] = NULL;
				break;
			}
			cand = ix->down_spare[c - 1];
			hipLaunchKernelGGL(k_queue_probe, dim3(1), dim3(64), 0, cand, 0u, d_sink);	/* (binds its queue) */
			(void)hipStreamSynchronize(cand);
		}
		for (int r = 0; r < 3 && ok; r++) {
			/* (twice: a slow first launch must not pass for "beside") */
			ok = runs_beside(refs[r], cand, e_ref, e_x, d_sink) && runs_beside(refs[r], cand, e_ref, e_x, d_sink);
		}
		if (ok) {
			if (c > 0) {
				std::swap(ix->stream_down, ix->down_spare[c - 1]);
			}
			ix->down_probe = c;
			goto out;
		}
	}
	ix->down_probe = -1;
	ix->cfg.down_inline = true;	/* no stream runs beside all three: the records ride the scan stream */
out:
	(void)hipGetLastError();
	if (e_ref) (void)hipEventDestroy(e_ref);
	if (e_x) (void)hipEventDestroy(e_x);
	(void)hipFree(d_sink);
	if (ix->cfg.debug_timing) {
		fprintf(stderr, "[nxsgpu] record stream: candidate %d\n", ix->down_probe);
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
		    (const v4u_t *)ix->d_post_dt, bytes / 16, d_sink);
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


/*
 * nxs_gpu_int.h -- internal declarations shared by the translation units of the
 * HIP (gfx950 / MI355X) side of the query path.  include/nxs_gpu.h is the C ABI
 * the C11 host code sees; nothing here crosses it.
 *
 *  nxs_gpu_index.hip      k_expand_pairs, k_post_offsets, k_impacts[_csr], the
 *                         refresh kernels (k_merge_old ...): the device index,
 *                         built from the nxsdtmap image and refreshed in place
 *  nxs_gpu_scan_tile.hip  k_scan / k_scan8: posting iteration + f32 sums in token
 *                         order in per-wavefront LDS tiles (run_query_logic +
 *                         get_expr_bitmap, search.c:118-278; results.c:128-150)
 *  nxs_gpu_scan_mask.hip  k_scanm / k_cold: OR-like queries of sparse terms, a
 *                         quantised score bound per doc in LDS, exact sums only
 *                         for the docs that can beat the threshold
 *  nxs_gpu_scan_stripe.hip k_scans: the same class on doc STRIPES cut out of the lists by the rank
 *                         directories -- no per-term window state, all terms' postings of a stripe as
 *                         one flat run of lanes, candidates scored lane-parallel (the default where
 *                         every term has a directory)
 *  nxs_gpu_scan_bit.hip   k_scanb: the same class on one presence BIT per doc pair (64k-doc tiles),
 *                         candidates scored one per lane by lower-bound searches (the default)
 *  nxs_gpu_scan_req.hip   k_cursors, k_scan1 (one token), k_scanr (required
 *                         terms: intersect first), k_scanq (the same through
 *                         block-presence bitmaps: postings of surviving blocks only)
 *  nxs_gpu_replay.hip     k_replay: the reference's capped min-heap + heapsort
 *                         (heap.c:58-221; results.c:165-220) replayed exactly
 *  nxs_gpu_wide.hip       k_scanw: queries beyond the fixed-size plan
 *  nxs_gpu_fuzzy.hip      BK-tree search (bktree.c:219-275) + Levenshtein
 *  nxs_gpu_search.hip     work list, kernel dispatch, batches in flight
 *  nxs_gpu_comm.hip       RCCL all-gather of the record blocks (query sharding)
 *
 * No MFMA anywhere: this is sparse gather/accumulate and byte/integer work,
 * bounded by HBM bandwidth and load latency.
 */
#ifndef NXS_GPU_INT_H
#define NXS_GPU_INT_H

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdarg>
#include <vector>
#include <algorithm>
#include <atomic>
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

/* last error of the calling thread (nxsgpu_last_error): nxs_gpu_index.hip */
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
void clear_error(void);
bool have_error(void);

#define	HIP_TRY(expr) do {						\
	hipError_t e_ = (expr);						\
	if (e_ != hipSuccess) {						\
		set_error("%s failed: %s (%s:%d)", #expr,		\
		    hipGetErrorString(e_), __FILE__, __LINE__);		\
		goto fail;						\
	}								\
} while (0)

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
	float		tcap[8];	/* dropped tokens: what the term adds to a doc that is NOT in its outlier list at
					 * most (== tmax when it has none) -- the ceiling the dropping rests on */
	uint32_t	outl_mask;	/* dropped tokens whose pbeg / pend name the term's OUTLIER list (the postings
					 * above tcap, impact = the excess over it): scanned like a sparse term for the
					 * bounds, never for the exact score (that comes from the column) */
	uint32_t	outl_tfidf;	/* host only: the dropped tokens have outlier lists to put in place (build_worklist) */
	uint32_t	bm_col[8];	/* k_scanq: the token's block-presence bitmap (nxsgpu_index::d_blkmap row), ~0 = none */
	uint32_t	qflags;		/* bit 0: no second chance on the accumulator tiles (pbeg / pend are not the
					 * terms' lists) -- an overflowing range flags the query for the exact passes */
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
	uint64_t	max_post;	/* NXS_GPU_MAXPOST (65536): a batch too big for that many postings per wavefront at the target gets more wavefronts (<= 4x) */
	uint64_t	wave_target_scans;	/* NXS_GPU_WAVES_SCANS: ... of a batch that holds the stripe class (default: min(that, 57344)) */
	uint64_t	min_post;	/* NXS_GPU_MINPOST */
	uint64_t	min_post_solo;	/* NXS_GPU_MINPOST_SOLO: the same for a small batch with nothing else in flight */
	double		scanm_dens;	/* NXS_GPU_SCANM_DENS */
	uint32_t	scanm_minnt, scanm_maxnt;
	uint32_t	rmin;		/* fewest tokens for k_scanr (NXS_GPU_NOSCANR2 => 3) */
	uint32_t	seg_cap;	/* NXS_GPU_SEGCAP */
	uint32_t	scan1_split;	/* NXS_GPU_SCAN1_SPLIT: single-token classes with a query of this many ranges
					 * send every query's top range ahead in a launch of its own */
	uint32_t	seg_cap_big;	/* NXS_GPU_SEGCAP_BIG: the same for limits > 64 (0: 6 x limit) */
	uint64_t	big_minpost;	/* NXS_GPU_BIG_MINPOST: postings per range and unit of limit, limits > 64 */
	uint64_t	fuzzy_items;	/* NXS_GPU_FUZZY_ITEMS */
	bool		use_scanr, mask_off, by_level, use_scanm, scanm_general;
	bool		old_scan, no_scan1, no_req, one_replay, fuzzy_safe, fuzzy_noprune;
	bool		fuzzy_bfs;	/* NXS_GPU_FUZZY_BFS: level-by-level frontier search only */
	uint64_t	fuzzy_cand;	/* NXS_GPU_FUZZY_CAND: survivor queue of the match-first search (items) */
	bool		use_drop;	/* !NXS_GPU_NODROP: dense terms leave sparse OR scans (k_scanm<.., DROP>) */
	uint64_t	drop_minpost;	/* NXS_GPU_DROP_MINPOST: fewest sparse postings for that path */
	uint64_t	drop_workmul;	/* NXS_GPU_DROP_WORKMUL: range count multiplier of that class */
	bool		drop_prio, drop_side;	/* !NXS_GPU_DROP_NOPRIO / !NXS_GPU_DROP_NOSIDE */
	uint32_t	drop_split;	/* NXS_GPU_DROP_SPLIT (1): this many top levels of the sparse + dense class go ahead in a launch of their own (0: none) */
	bool		drop_b;		/* NXS_GPU_DROPB: the sparse + dense class's second kernel is k_scanb<.., DROP> (64k-doc tiles:
					 * a handful of flushes per wavefront, each gathering the dense impacts of ~100 candidates at once) */
	bool		and_early;	/* !NXS_GPU_AND_NOEARLY: the conjunctive classes (k_scanr, k_scanq) run on the upload stream,
					 * beside the previous batch's scans */
	bool		drop_early;	/* !NXS_GPU_DROP_NOEARLY: ... on the upload stream, right behind k_cursors: beside the PREVIOUS batch */
	bool		no_straggler;	/* NXS_GPU_NOSTRAGGLER: tiny tile-path OR classes stay launches of their own */
	bool		drop_tiles;	/* NXS_GPU_DROP_TILES: the sparse + dense class streams its dense lists on the accumulator
					 * tiles (k_scan8) -- still beside the other classes, on the side stream (experiment) */
	bool		debug_timing;	/* NXS_GPU_DEBUG_TIMING: per-batch host phases of _begin to stderr */
	bool		old_replay;	/* NXS_GPU_OLDREPLAY: the LDS heap on one lane (k_replay<HEAP_LDS>) */
	bool		tfidf_drop;	/* !NXS_GPU_TFIDF_NODROP: dense terms leave TF-IDF scans too (capped ceiling + outlier lists) */
	uint32_t	outl_share;	/* NXS_GPU_OUTL_SHARE (8): at most 1/this of a dense term's postings are outliers */
	bool		replay_join;	/* NXS_GPU_REPLAY_JOIN: the scan stream waits for a batch's last heap replay (limits <= 64) */
	bool		down_inline;	/* NXS_GPU_DOWN_INLINE: sharded runs also keep everything on one stream */
	bool		use_blkmap;	/* !NXS_GPU_NOBLKMAP: block-presence bitmaps for the longer lists; conjunctions whose
					 * required terms all have one intersect THOSE first (k_scanq) */
	uint64_t	bm_share;	/* NXS_GPU_BM_SHARE (1024): a term gets a bitmap if it holds >= n_docs / this docs */
	double		bigq_em;	/* NXS_GPU_BIGQ_EM (16): limits > 64 take k_scanq only for queries that expect fewer matches (it emits them all) */
	double		bm_gain;	/* NXS_GPU_BM_GAIN (16): k_scanq if (expected surviving blocks) x this < the driver's postings */
	bool		use_scans;	/* !NXS_GPU_NOSCANS: the mask path on doc stripes cut out of the lists by the rank directories
					 * (k_scans) for the queries whose terms all have one */
	uint64_t	scans_workpct;	/* NXS_GPU_SCANS_WORKPCT (70): the stripe class's share of the work list's wavefronts, per cent of what its postings would get */
	bool		use_scans_drop;	/* NXS_GPU_SCANS_DROP: ... also as the sparse + dense class's second kernel (k_scans<.., DROP>, BM25: the
					 * stripes' byte maps FILLED from the dense term's byte column, built only with this switch):
					 * opt-in -- 12-30 % faster than k_scanm<.., DROP> on the kprobe sets, no gain in a C3 / C5 step
					 * (NOTES.md, round 5) */
	bool		use_scanb;	/* !NXS_GPU_NOSCANB: the mask path's sparsest queries on the presence-bit kernel (k_scanb) */
	double		scanb_dens;	/* NXS_GPU_SCANB_DENS: ... those whose lists together hold at most this fraction of the docs */
};

void cfg_from_env(gpu_cfg_t &c);

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
	hipStream_t	stream_rp[3];	/* MODE_BIG batches (limit > 64): the replays of a batch -- milliseconds of heap
					 * insertions -- run here (batch seq mod 3), beside the next batches' scans AND replays */
	hipStream_t	stream3;	/* the sparse + dense OR class (k_scanm<.., DROP>): few, latency-bound
					 * wavefronts that run BESIDE the other classes, not in front of them */
	hipEvent_t	ev_cls, ev_join, ev_fork3, ev_join3;
	/* nxsgpu_search_dev_begin/_end: up to NXSGPU_INFLIGHT batches in flight, each with its own
	 * device workspace and pinned staging; plans go up on their own stream */
	hipStream_t	stream_up;
	hipStream_t	stream_down;	/* record blocks: all-gather (sharded) + copy to pinned memory */
	hipStream_t	down_spare[3];	/* pick_record_stream: candidates that were not taken (or the original stream_down) */
	int		down_probe;	/* ... which candidate was (0 = stream_down as created, -1 = none: down_inline) */
	hipStream_t	stream_fz;	/* BK-tree searches: beside the batches in flight, not behind them */
	struct nxsgpu_comm *comm;	/* attached communicator (query sharding) or NULL */
	nxsgpu_parallel_t par_run;	/* the caller's worker threads (nxsgpu_index_set_parallel) or NULL */
	void *		par_ctx;
	struct dev_slot_t {
		void *		ws;
		size_t		ws_len;
		uint8_t *	h_stage;	/* pinned: uploads, then the overflow flags coming back */
		size_t		h_stage_len;
		hipEvent_t	ev_up, ev_done, ev_res, ev_t[3];
		hipEvent_t	ev_early;	/* the conjunctive classes, run early on the upload stream, and their replays are done */
		hipEvent_t	ev_ahead;	/* the sparse + dense class's top ranges, run ahead on the upload stream, are done */
		bool		ahead;		/* ... this batch has such a launch */
		/* profiling: events around each class's scan kernels, on the class's stream (created on demand) */
		hipEvent_t	ev_cls[NXSGPU_PROF_CLS][2];
		uint32_t	cls_key[NXSGPU_PROF_CLS];
		uint64_t	cls_post[NXSGPU_PROF_CLS], cls_q[NXSGPU_PROF_CLS];
		uint32_t	n_cls;
		bool		ev_cls_ok;
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
	}		slot[NXSGPU_INFLIGHT];
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
	uint64_t *	d_post_dt_spare;	/* the CSR buffer the next refresh merges into (NULL until the first one) */
	uint64_t	cap_post_dt, cap_post_dt_spare;
	uint64_t	cap_docs_ids, cap_docs_len;
	uint32_t	max_tf;
	posting_t *	d_post[2];	/* [P] per ranking algo; NULL until the algo is first used (algo_on) */
	bool		algo_on[2];	/* impacts of this ranking function are materialised */
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
	uint64_t	dense_cap[2];		/* allocated words per algo */
	/* BM25: the same columns as bytes, q8[doc] = ceil(255 x impact / the term's largest impact) (0: the doc
	 * does not hold the term), rows of dense_q8_stride bytes (a multiple of 16 KB beyond n_docs: a doc stripe
	 * is read whole) -- what k_scans<.., DROP> fills its byte map with (nxs_gpu_scan_stripe.hip) */
	uint8_t *	d_dense_q8;
	uint64_t	dense_q8_stride, dense_q8_cap;
	/*
	 * TF-IDF: log(tf + 1) does not saturate, so one posting with an outlier tf sets a
	 * ceiling far above what the term typically adds and the dense terms could never
	 * leave a scan.  Each dense term's list is therefore split by value: the column
	 * holds every impact, a CAP (the impact of the largest tf that all but 1/outl_share
	 * of the postings stay at or below) bounds the ordinary ones, and the postings
	 * above it form the term's OUTLIER list -- (doc, impact - cap), in doc order, behind
	 * the regular postings in d_post[TF_IDF] -- which a query scans like a sparse term.
	 */
	std::vector<uint64_t> outl_off;		/* [columns + 1] positions in d_post[TF_IDF] (from cap_post on) */
	std::vector<float> outl_cap;		/* [columns] the cap (== the largest impact: no outlier list) */
	std::vector<float> outl_max;		/* [columns] largest excess over the cap */

	/*
	 * Block-presence bitmaps (the reference intersects roaring bitmaps before it looks at
	 * a posting, search.c:118-174): per term holding >= n_docs / cfg.bm_share docs, one bit
	 * per 64-doc block -- is there a posting in [64 b, 64 b + 64)? -- and, per 4096-doc word
	 * of that map, the list position of the first posting at or above the word's first doc
	 * (a rank directory: a block's postings are found by a search inside one word's span).
	 * Rebuilt with the impacts at every refresh (one pass over the lists that have one).
	 */
	std::vector<uint32_t> bm_terms;		/* ascending term ids; row = position */
	uint64_t *	d_blkmap;		/* [rows][bm_words] */
	uint32_t *	d_bmrank;		/* [rows][bm_words + 1] */
	uint64_t	bm_words;		/* ceil(n_docs / 4096) */
	uint64_t	bm_cap;			/* allocated rows x words (grow-only) */

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
	std::vector<uint32_t> fz_len_start;	/* [FZ_MAXLEN + 2] first candidate of every term length (profiling) */

	/* reusable query workspaces */
	void *		ws;
	size_t		ws_len;
	void *		h_pin;
	size_t		h_pin_len;

	/* fuzzy workspaces */
	void *		fz;
	size_t		fz_len;
	/*
	 * Match-first passes (nxs_gpu_fuzzy.hip mf_launch / mf_finish): NXSGPU_FZ_SLOTS of them can be queued on
	 * stream_fz, each with its own device workspace, pinned staging (one copy up, one down) and profiling
	 * events.  state: 0 free, 1 a pass is queued (nxsgpu_fuzzy_begin; its results land in pin), 2 begun but
	 * nothing queued (the batch does not qualify for a single pass: _end runs the whole search).  `fz`
	 * above is the level-by-level search's workspace.
	 */
	struct fz_slot_t {
		void *		ws;
		size_t		ws_len;
		uint8_t *	pin;
		size_t		pin_len;
		int		state;
		uint32_t	n;
		hipEvent_t	ev[4];
		hipEvent_t	ev_done;	/* after the copy back */
	} fzs[NXSGPU_FZ_SLOTS];

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
/* k_scan: posting iteration + LDS score accumulation + pre-selection   */
/* ------------------------------------------------------------------ */

/*
 * MODE_TOPK: candidate filter pass, 1 <= k <= 64 (the wavefront's k best scores
 *            in one VGPR, lane i = i-th largest);
 * MODE_BIG:  the same filter for 64 < k <= NXSGPU_BIG_K (the API's default limit
 *            is 1000, nxs_impl.h:39): the threshold is a LOWER BOUND of the k-th
 *            largest score the wavefront has emitted, read off a histogram of
 *            score buckets in LDS (bigk_*, nxs_gpu_dev.h);
 * MODE_COUNT / MODE_ALL: the exact two-pass path (count matches, emit them all).
 */
enum { MODE_TOPK = 0, MODE_COUNT = 1, MODE_ALL = 2, MODE_BIG = 3 };
#define	MODE_FILTERS(m)	((m) == MODE_TOPK || (m) == MODE_BIG)

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
	float *			pub_sk;		/* MODE_BIG: [segments][8] lower bounds of a finished range's ceil(k / 2^j)-th
						 * best score, j = 0..5 (bigk_publish / bigk_hint) */
	uint32_t		flags;		/* bit 0: raise the wavefronts' issue priority (side-stream class);
						 * bit 1 (k_scan8): the work items are the retry list's;
						 * bit 4: the sparse + dense class's second kernel is k_scans<.., DROP> */
	/*
	 * Ranges whose pending list overflowed on the mask path (k_scanm: a burst of docs
	 * above a still-weak threshold -- it depends on when higher ranges publish theirs,
	 * i.e. on timing) are not a reason to re-run the query: the range is queued here
	 * and scanned once more on the accumulator tiles (k_scan8), which have no such
	 * list, right behind the class's scan and in front of its heap replay.
	 */
	uint32_t *		retry_count;	/* entries queued (may exceed retry_cap: those flag the query instead) */
	item_t *		retry_items;	/* [retry_cap] */
	uint32_t		retry_cap;
	uint32_t *		cold_state;	/* [segments][16]: what k_cold hands to k_scanm<.., DROP> */
	float *			cold_top;	/* [segments][64]: its running top-k scores */
	const uint64_t *	blkmap;		/* k_scanq: block-presence bitmaps [row][bm_words] and ... */
	const uint32_t *	bmrank;		/* ... their rank directories [row][bm_words + 1] */
	uint64_t		bm_words;
	const uint32_t *	dense_col;	/* impact columns of the dense terms: [col][n_docs] f32 bits,
						 * 0xffffffff = the doc does not hold the term */
	uint64_t		dense_stride;
	const uint8_t *		dense_q8;	/* BM25: the columns as bytes (nxsgpu_index::d_dense_q8), or NULL */
	uint64_t		dense_q8_stride;
};

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
	uint32_t		flags;		/* bit 0: 64 < k <= REPLAY_LDS_K on the one-lane kernel (NXS_GPU_OLDREPLAY) */
};

/* where the heap lives: global memory (any k), across the lanes (k <= 64), or in
 * dynamic LDS as pairs (k <= REPLAY_LDS_K) */
#define	HEAP_GLOBAL	0
#define	HEAP_REG	1
#define	HEAP_LDS	2
#define	REPLAY_LDS_K	8000

struct launch_t { uint32_t first, count, nt_bucket, kind, nomask, q_first, q_count; uint64_t postings; };	/* kind: 0 generic (9..32 tokens), 1 accumulator tiles / k_scan1, 3 k_scanr, 4 k_scanm, 5 k_cold + k_scanm<DROP>,
 * 6 k_scanb, 7 k_scanq, 8 k_scans, 9 k_cold + k_scans<DROP> */

struct worklist_t {
	std::vector<qmeta_t>	qmeta;
	std::vector<item_t>	items;
	std::vector<launch_t>	launches;
	std::vector<uint32_t>	bnd_q;		/* boundary -> query, n_segs + nq entries */
	std::vector<uint32_t>	qorder;		/* queries in launch order; launch_t::q_first/q_count index it */
	uint32_t		n_segs;
	bool			need_cursors;	/* some query's ranges are doc ranges (k_cursors has work) */
};

template <typename T>
static T *
carve(uint8_t *&p, size_t n)
{
	uintptr_t a = ((uintptr_t)p + 255) & ~(uintptr_t)255;
	T *r = (T *)a;
	p = (uint8_t *)(a + n * sizeof(T));
	return r;
}

/* k_scanr: queries with >= 4 required terms take rounds of whole driver windows
 * over an LDS hash table (a class of their own: build_worklist) */
#ifndef SCANR_HASH
#define	SCANR_HASH	1		/* a round = a whole driver window, its docs in an LDS hash table */
#endif

/* retry lists of the mask path (scan_args_t::retry_items): one per scan launch */
#define	RETRY_LISTS	64
#define	RETRY_CAP	64

/* ---- nxs_gpu_index.hip ---- */
#define	X_KEEP_MAX	(4ull << 30)
void *	xbuf_get(nxsgpu_index_t *ix, int which, size_t need);
void	xbuf_put(nxsgpu_index_t *ix, int which);
bool	ensure_ws(nxsgpu_index_t *ix, size_t need);
bool	ensure_pin(nxsgpu_index_t *ix, size_t need);
int	rebuild_impacts(nxsgpu_index_t *ix, unsigned only = 3);	/* bit a: ranking function a */
/* materialise the impacts of `algo` (first search with the non-default function) */
int	ensure_algo(nxsgpu_index_t *ix, int algo);
void	warm_streams(nxsgpu_index_t *ix);
void	pick_record_stream(nxsgpu_index_t *ix);

/* ---- nxs_gpu_fuzzy.hip ---- */
void	bk_aux_free(nxsgpu_index_t *ix);
int	bk_aux_build(nxsgpu_index_t *ix, const nxsgpu_bknode_t *nodes, uint32_t n);

/* ---- nxs_gpu_search.hip ---- */
void	delete_worklist(worklist_t *);

/* ---- nxs_gpu_comm.hip ---- */
int	comm_allgather_dev(nxsgpu_comm_t *, const void *send, void *recv, size_t bytes, hipStream_t);

/*
 * Kernel launchers, one per kernel family (the kernels are templates local to
 * their translation unit).  `mode` is MODE_*; nt_bucket 1 / 2 / 3 / 5 / 8.
 */
void	nxs_launch_cursors(const scan_args_t &a, const uint32_t *d_bnd_q, uint32_t n_bnd, hipStream_t st);
void	nxs_launch_scan_generic(int mode, bool wide_mask, unsigned grid, hipStream_t st, const scan_args_t &a);
void	nxs_launch_scan8(int mode, uint32_t nt_bucket, uint32_t mm, unsigned grid, hipStream_t st, const scan_args_t &a);
void	nxs_launch_scanm(uint32_t nt_bucket, bool gen, unsigned grid, hipStream_t st, const scan_args_t &a);
void	nxs_launch_drop_class(uint32_t nt_bucket, unsigned grid, hipStream_t st, const scan_args_t &a);
/* the mask path on doc stripes (k_scans: rank directories instead of per-term register windows) */
void	nxs_launch_scans(uint32_t nt_bucket, bool gen, unsigned grid, hipStream_t st, const scan_args_t &a);
/* ... the sparse + dense class's second kernel on stripes (k_scans<.., DROP>; scan_args_t::flags bit 4 makes
 * nxs_launch_drop_class launch it behind k_cold) */
void	nxs_launch_scans_drop(uint32_t nt_bucket, unsigned grid, hipStream_t st, const scan_args_t &a);
/* the mask path on presence bits, candidates scored one per lane (k_scanb) */
void	nxs_launch_scanb(uint32_t nt_bucket, bool gen, bool drop, unsigned grid, hipStream_t st, const scan_args_t &a);
void	nxs_launch_scan1(int mode, unsigned grid, hipStream_t st, const scan_args_t &a);
void	nxs_launch_scanr(int mode, uint32_t nt_bucket, bool hash, unsigned grid, hipStream_t st, const scan_args_t &a);
/* conjunctions whose required terms all have a block bitmap: intersect the bitmaps, look only at surviving blocks */
void	nxs_launch_scanq(uint32_t nt_bucket, unsigned grid, hipStream_t st, const scan_args_t &a);
void	nxs_launch_replay(int heap, unsigned grid, size_t dyn_lds, hipStream_t st, const replay_args_t &r);

#endif /* NXS_GPU_INT_H */


/*
 * nxs_gpu.h -- C-ABI of the HIP (gfx950 / MI355X) side of the query path.
 *
 * This is the thin shim the C11 host code (nxsearch_amd/csrc/nxs_*.c) calls;
 * no HIP or C++ types cross it: plain pointers, sizes and PODs.  Each entry
 * names the reference seam it replaces.
 *
 *   nxsgpu_index_create  <- idx_dtmap_sync + dtmap_build_tdmap
 *                           (src/index/dtmap.c:386-544): builds the reverse
 *                           index (here: CSR posting arrays in HBM, transposed
 *                           from the nxsdtmap image on the device) and the
 *                           BK-tree image (src/index/idxterm.c:157-187).
 *   nxsgpu_search        <- run_query_logic + get_expr_bitmap
 *                           (src/query/search.c:118-278) with the
 *                           ranking_func_t seam (src/core/nxs_impl.h:52-53;
 *                           src/algo/ranking.c:41-176) and
 *                           nxs_resp_addresult/nxs_resp_build
 *                           (src/core/results.c:128-220; src/algo/heap.c).
 *   nxsgpu_fuzzy         <- idxterm_fuzzysearch (src/index/idxterm.c:210-249)
 *                           = bktree_search (src/algo/bktree.c:219-275) with
 *                           the bktree_distfunc_t seam bound to levdist
 *                           (src/algo/levdist.c:67-150).
 */
#ifndef NXS_GPU_H
#define NXS_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define	NXSGPU_MAX_TOKENS	32	/* tokens per query on the device path */
#define	NXSGPU_MAX_PROG		256	/* postfix program bytes */
#define	NXSGPU_FAST_K		64	/* limit up to which the heap lives across the lanes of a wavefront */
#define	NXSGPU_BIG_K		8000	/* limit up to which the candidate filter applies (heap in LDS);
					 * beyond it: the exact two-pass path */

/* ranking_algo_t (reference src/index/index.h:30-34) */
#define	NXSGPU_TF_IDF		0
#define	NXSGPU_BM25		1

/* postfix opcodes: 0..31 push presence bit of token i */
#define	NXSGPU_OP_EMPTY		0x40	/* push the empty set (search.c:140) */
#define	NXSGPU_OP_AND		0x80	/* roaring64_bitmap_and_inplace    */
#define	NXSGPU_OP_OR		0x81	/* roaring64_bitmap_or_inplace     */
#define	NXSGPU_OP_ANDNOT	0x82	/* roaring64_bitmap_andnot_inplace */

typedef struct nxsgpu_index nxsgpu_index_t;

/*
 * One flattened BK-tree node (BFS numbering: index = BFS rank, the children of
 * a node are contiguous and in ascending slot order -- bktree.c:54-58,79-98).
 */
typedef struct {
	uint64_t	bitmap;		/* child slots (distance 1..63)       */
	uint32_t	first_child;	/* BFS index of the lowest-slot child */
	uint32_t	term_id;
	uint32_t	str_off;	/* term bytes in the byte pool        */
	uint16_t	str_len;
	uint16_t	flags;		/* bit 0: on-disk total count > 0     */
	uint8_t		inl[8];		/* first 8 term bytes (zero padded)   */
} nxsgpu_bknode_t;

typedef struct {
	/* nxsdtmap image (host memory) and the live doc blocks in it */
	const uint8_t *	dtmap_img;
	uint64_t	dtmap_len;
	const uint64_t *blk_off;	/* [n_docs] block offsets, ascending doc id */
	const uint64_t *doc_ids;	/* [n_docs] ascending                       */
	const uint64_t *pair_base;	/* [n_docs+1] prefix sum of per-doc n       */
	uint64_t	n_docs;
	/* term-id space of nxsterms: ids 1..n_terms; term_ok[id]=1 if live */
	uint32_t	n_terms;
	const uint8_t *	term_ok;	/* [n_terms+1] */
	/* header counters (dtmap.c:660-677) */
	uint32_t	hdr_doc_count;
	uint64_t	hdr_token_count;
	/* flattened BK-tree */
	const nxsgpu_bknode_t *bk_nodes;
	uint32_t	n_bk;
	uint32_t	bk_depth;	/* number of BFS levels */
	const uint8_t *	bk_bytes;
	uint64_t	bk_bytes_len;
	/* the index's ranking function (params.db "algo"): its impacts are built with
	 * the index, the other function's on the first search that asks for it; -1: both */
	int		default_algo;
} nxsgpu_index_src_t;

/* one resolved query */
typedef struct nxsgpu_query {
	uint32_t	n_tokens;			/* token-list order */
	uint32_t	term_id[NXSGPU_MAX_TOKENS];
	uint32_t	prog_len;
	uint8_t		prog[NXSGPU_MAX_PROG];
	uint32_t	truth[8];	/* 256-bit truth table, valid if n_tokens <= 8 */
} nxsgpu_query_t;

typedef struct {
	uint32_t	n_queries;
	uint32_t *	counts;		/* [n]   results per query          */
	uint64_t *	offsets;	/* [n+1] into doc_ids / scores      */
	uint64_t *	doc_ids;
	float *		scores;
	/* work counters of the call */
	uint64_t	postings;	/* postings streamed                */
	uint64_t	candidates;	/* (doc,score) handed to the replay */
	uint32_t	exact_requeries;/* queries re-run through the two-pass path */
} nxsgpu_results_t;

#define	NXSGPU_PROF_CLS	16
typedef struct {
	uint64_t	launches;	/* launches of the dominant scan kernel */
	double		scan_ms;	/* summed HIP-event time of those      */
	double		replay_ms;
	double		fuzzy_ms;
	uint64_t	postings;	/* summed algorithmic postings         */
	uint64_t	fuzzy_visits;	/* Levenshtein distance evaluations    */
	uint64_t	fuzzy_pairs;	/* (token, node) pairs dequeued        */
	uint64_t	fuzzy_level[40];/* ... per BFS level                   */
	/* match-first search, per kernel (HIP events on the fuzzy stream) */
	double		fuzzy_filter_ms;/* k_bk_peq + k_bk_seed + k_fz_filter   */
	double		fuzzy_dist_ms;	/* k_fz_dist                           */
	double		fuzzy_chain_ms;	/* k_fz_chain + k_bk_finish            */
	uint64_t	fuzzy_checked;	/* (token, term) pairs k_fz_filter compared */
	/*
	 * Per query CLASS (= one scan launch per batch): HIP events recorded on the stream
	 * the class's kernels are launched on, around them.  cls_key = kind << 8 | shape << 4
	 * bits | token bucket (nxs_gpu_search.hip: build_worklist); kind 1 k_scan1 / k_scan8,
	 * 3 k_scanr, 4 k_scanm, 5 k_cold + k_scanm<.., DROP>, 6 k_scanb, 7 k_scanq, 0 k_scan.
	 */
	uint32_t	n_cls;
	uint32_t	cls_key[NXSGPU_PROF_CLS];
	uint64_t	cls_launches[NXSGPU_PROF_CLS];
	double		cls_ms[NXSGPU_PROF_CLS];
	uint64_t	cls_postings[NXSGPU_PROF_CLS];	/* algorithmic postings of the class's queries */
	uint64_t	cls_queries[NXSGPU_PROF_CLS];
} nxsgpu_profile_t;

int		nxsgpu_device_count(void);
const char *	nxsgpu_last_error(void);

nxsgpu_index_t *nxsgpu_index_create(int device, const nxsgpu_index_src_t *);
void		nxsgpu_index_destroy(nxsgpu_index_t *);

/*
 * N1 -- incremental refresh: what idx_terms_sync / idx_dtmap_sync consumed since
 * the last snapshot (src/index/terms.c:320-414, src/index/dtmap.c:440-544),
 * validated by the host: appended docs carry ids above every loaded one (they
 * take the highest ordinals) and name known terms only.
 */
typedef struct {
	uint32_t	n_terms;	/* new last term id (>= the old one) */
	const uint8_t *	term_ok;	/* [n_terms+1] */
	const uint8_t *	dtmap_img;	/* the mapped nxsdtmap image */
	uint64_t	dtmap_len;
	const uint64_t *blk_off;	/* [n_new] appended doc blocks, ascending doc id */
	const uint64_t *doc_ids;	/* [n_new] */
	const uint64_t *pair_base;	/* [n_new+1] */
	uint64_t	n_new;
	const uint32_t *dead_term;	/* (term, ordinal) of every posting of a removed doc */
	const uint32_t *dead_ord;
	uint64_t	n_dead_pairs;
	uint32_t	hdr_doc_count;	/* header counters now (dtmap.c:660-677) */
	uint64_t	hdr_token_count;
} nxsgpu_index_delta_t;

/* merge the delta into the device CSR and recompute every impact; 0 / -1 (the
 * index is unchanged on failure unless the error says otherwise) */
int		nxsgpu_index_apply(nxsgpu_index_t *, const nxsgpu_index_delta_t *);
/* replace the BK-tree image (after terms were inserted on the host) */
int		nxsgpu_index_set_bk(nxsgpu_index_t *, const nxsgpu_bknode_t *nodes, uint32_t n,
		    uint32_t depth, const uint8_t *bytes, uint64_t bytes_len);

/*
 * N4 -- doc-sharded mode (collections beyond one GPU's HBM): every shard holds
 * the postings of a contiguous range of docs (ascending doc id) but scores with
 * COLLECTION-WIDE statistics: N and the token count come from the file header,
 * df is handed in (sum of the shards' nxsgpu_index_df) and all impacts are
 * recomputed.  A query then runs on every shard, each returns the exact
 * sequence of candidates its local heap accepted, and the sequences -- highest
 * shard first -- are replayed through the reference's heap once more
 * (nxsgpu_merge_candidates): identical top-k, ties included.
 */
int		nxsgpu_index_set_global_df(nxsgpu_index_t *, const uint32_t *df, uint32_t n_terms);
/* ids/scores [n_queries][cap], counts [n_queries] (host); counts[q] > cap = overflow */
int		nxsgpu_search_candidates(nxsgpu_index_t *, int algo, uint64_t limit,
		    const nxsgpu_query_t *queries, uint32_t n_queries, uint32_t cap,
		    uint64_t *ids, float *scores, uint32_t *counts);
/* ids/scores [nq][n_shards][cap], counts [nq][n_shards], shard 0 = lowest docs;
 * out_* [nq][limit] / [nq]; limit <= NXSGPU_FAST_K */
int		nxsgpu_merge_candidates(int device, uint32_t limit, uint32_t nq, uint32_t n_shards,
		    uint32_t cap, const uint64_t *ids, const float *scores, const uint32_t *counts,
		    uint64_t *out_ids, float *out_scores, uint32_t *out_counts);

/* document frequency per term id [n_terms+1] (host buffer) */
int		nxsgpu_index_df(nxsgpu_index_t *, uint32_t *df);
uint64_t	nxsgpu_index_postings(const nxsgpu_index_t *);
uint64_t	nxsgpu_index_docs(const nxsgpu_index_t *);
/* first live doc (file order) whose block names an unknown term, or ~0 */
uint64_t	nxsgpu_index_first_bad_doc(const nxsgpu_index_t *);

/*
 * Threading: an nxsgpu_index_t belongs to ONE host thread at a time, like the
 * reference's nxs_t (docs/c-api.md:5-8).  A blocking nxsgpu_search() while batches
 * are in flight runs on stream and event sets of its own (it swaps them into the
 * index for the call): another thread calling into the same index meanwhile would
 * enqueue on the wrong streams.
 */
int		nxsgpu_search(nxsgpu_index_t *, int algo, uint64_t limit,
		    const nxsgpu_query_t *queries, uint32_t n_queries,
		    nxsgpu_results_t *res);
void		nxsgpu_results_free(nxsgpu_results_t *);

/*
 * Queries that do not fit nxsgpu_query_t (more than NXSGPU_MAX_TOKENS live
 * tokens, a program of more than NXSGPU_MAX_PROG items, or an evaluation
 * stack deeper than 64): run_query_logic (search.c:210-278) has no such
 * bounds.  Variable-size plan, generic kernel, always the exact two-pass path.
 */
#define	NXSGPU_WIDE_MAX_TOKENS	1024
#define	NXSGPU_WOP_AND		0xfff0u
#define	NXSGPU_WOP_OR		0xfff1u
#define	NXSGPU_WOP_ANDNOT	0xfff2u
#define	NXSGPU_WOP_EMPTY	0xffffu	/* push the empty set (search.c:140) */

typedef struct {
	uint32_t	n_tokens;	/* token-list order */
	const uint32_t *term_id;
	uint32_t	prog_len;
	const uint16_t *prog;		/* postfix: < 0x8000 pushes token i, else NXSGPU_WOP_* */
} nxsgpu_wide_query_t;

int		nxsgpu_search_wide(nxsgpu_index_t *, int algo, uint64_t limit,
		    const nxsgpu_wide_query_t *queries, uint32_t n_queries,
		    nxsgpu_results_t *res);

/*
 * Device-resident variant for multi-GPU gathers: limit <= NXSGPU_FAST_K,
 * outputs are DEVICE pointers laid out [n_queries][limit] / [n_queries],
 * left on the device (no host copy).  Returns 0, or 1 if some query needs the
 * exact two-pass path (then call nxsgpu_search for it), -1 on error.
 */
int		nxsgpu_search_dev(nxsgpu_index_t *, int algo, uint32_t limit,
		    const nxsgpu_query_t *queries, uint32_t n_queries,
		    uint64_t *d_doc_ids, float *d_scores, uint32_t *d_counts);

/*
 * The same, split for pipelining: _begin() plans the batch on the host, sends
 * the plans up and queues the kernels, then returns; _end() waits for the
 * OLDEST batch in flight and returns its status (0 / 1 / -1 as above).  Up to
 * NXSGPU_INFLIGHT batches may be in flight, so the host prepares batch i+1 while
 * batch i runs (two in flight hide the host side of a top-10 batch; limits in the
 * hundreds end in a heap replay of milliseconds -- one wavefront per query, the
 * chip nearly idle -- and take three or four to fill the GPU); each needs its own
 * output buffers until its _end().  nxsgpu_search() and nxsgpu_search_dev()
 * refuse to run while a batch is in flight.
 */
#define	NXSGPU_INFLIGHT	4
int		nxsgpu_search_dev_begin(nxsgpu_index_t *, int algo, uint32_t limit,
		    const nxsgpu_query_t *queries, uint32_t n_queries,
		    uint64_t *d_doc_ids, float *d_scores, uint32_t *d_counts);
int		nxsgpu_search_dev_end(nxsgpu_index_t *);

int		nxsgpu_fuzzy(nxsgpu_index_t *, const uint8_t *tok_bytes,
		    const uint32_t *tok_off, uint32_t n_tokens,
		    uint32_t *term_ids, uint64_t *visited);
/*
 * The same search split in two (no visit counts): _begin queues the device pass on the fuzzy stream and
 * returns a slot (>= 0; -1 error; -2 all NXSGPU_FZ_SLOTS slots taken), _end -- with the slot and the same
 * tokens -- waits and delivers.  The host's planning of the NEXT batch and the scans of the batches in
 * flight run meanwhile (nxs_api.c: a batch with misses is finished by the next
 * nxs_index_search_batch_begin).  Passes run, and are to be ended, in the order they were begun;
 * nxsgpu_fuzzy() refuses to run while one is in flight.
 */
#define	NXSGPU_FZ_SLOTS	2
int		nxsgpu_fuzzy_begin(nxsgpu_index_t *, const uint8_t *tok_bytes,
		    const uint32_t *tok_off, uint32_t n_tokens);
int		nxsgpu_fuzzy_end(nxsgpu_index_t *, int slot, const uint8_t *tok_bytes,
		    const uint32_t *tok_off, uint32_t n_tokens, uint32_t *term_ids);

/*
 * ---- host batches as fixed-size records; query sharding over several GPUs ----
 *
 * The reference scales out by running independent worker processes
 * (compose/nginx.conf:2); here a batch shards BY QUERY over the GPUs of a node
 * (SURVEY.md 8e): every rank holds a replica of the device index, takes a
 * contiguous slice of the batch, and ONE RCCL all-gather of fixed-size
 * per-query records reassembles the batch on every rank.
 *
 * Record of one query (limit k <= NXSGPU_BIG_K), NXSGPU_REC_BYTES(k) bytes:
 *	u32 count | u32 flags | u64 doc_id[k] | f32 score[k] | pad to 8
 * A rank's BLOCK = n_slots records followed by n_slots + 1 u32 status words:
 * one per slot (the nxs_err_t of a query that never reached the device) and a
 * last word of per-rank flags every rank reads after the all-gather
 * (NXSGPU_BLOCK_CHANGED), padded to 8 bytes.
 * With one rank there is no collective and the block is simply the batch's
 * host copy.
 */
#define	NXSGPU_REC_BYTES(k)	((8 + 12 * (size_t)(k) + 7) & ~(size_t)7)
#define	NXSGPU_STATUS_WORDS(n_slots)	((size_t)(n_slots) + 1)
#define	NXSGPU_BLOCK_BYTES(n_slots, k) \
	((size_t)(n_slots) * NXSGPU_REC_BYTES(k) + ((NXSGPU_STATUS_WORDS(n_slots) * 4 + 7) & ~(size_t)7))
/* block flags word: this rank saw the index files move when it planned the batch --
 * all ranks then re-sync at the same later _begin (nxs.h, "sharded mode") */
#define	NXSGPU_BLOCK_CHANGED	1u
#define	NXSGPU_REC_INEXACT	1u	/* flags: the query needs the exact two-pass path (nxsgpu_search) */
#define	NXSGPU_UID_BYTES	128	/* = NCCL_UNIQUE_ID_BYTES */

typedef struct nxsgpu_comm nxsgpu_comm_t;

/* contiguous slice [lo, hi) of an n-query batch owned by `rank`, and the
 * largest slice (= n_slots of every rank's block) */
void		nxsgpu_shard_slice(uint64_t n, int rank, int world, uint64_t *lo, uint64_t *hi);
uint64_t	nxsgpu_shard_capacity(uint64_t n, int world);

/*
 * One RCCL communicator over the ranks' GPUs.  Rank 0 obtains the unique id,
 * the application hands its bytes to the other ranks (MPI, a file, a socket,
 * torch.distributed ...), then EVERY rank calls nxsgpu_comm_create (collective).
 * librccl is loaded on first use; single-GPU users never touch it.
 */
int		nxsgpu_comm_unique_id(uint8_t uid[NXSGPU_UID_BYTES]);
nxsgpu_comm_t *	nxsgpu_comm_create(int device, int rank, int world,
		    const uint8_t uid[NXSGPU_UID_BYTES]);
void		nxsgpu_comm_destroy(nxsgpu_comm_t *);
int		nxsgpu_comm_rank(const nxsgpu_comm_t *);
int		nxsgpu_comm_world(const nxsgpu_comm_t *);
/* what RCCL itself reports for the communicator (ncclCommCount; -1: unknown), and how much has gone through
 * it: out[0] all-gathers queued, out[1] bytes this rank contributed -- evidence for scaling records */
int		nxsgpu_comm_rccl_count(const nxsgpu_comm_t *);
void		nxsgpu_comm_stats(const nxsgpu_comm_t *, uint64_t out[2]);
/* blocking all-gather of host buffers (staged through the device): the rare
 * fix-up round of a sharded batch, barriers */
int		nxsgpu_comm_allgather(nxsgpu_comm_t *, const void *send, void *recv,
		    size_t bytes_per_rank);
/* batches of this index all-gather their record blocks over `comm` (NULL: detach) */
int		nxsgpu_index_set_comm(nxsgpu_index_t *, nxsgpu_comm_t *);

typedef struct {
	uint32_t	n_slots;	/* records per block */
	uint32_t	k;
	uint32_t	world;		/* blocks */
	size_t		rec_bytes, block_bytes;
	const uint8_t *	blocks;		/* host memory, valid until NXSGPU_INFLIGHT more batches have begun */
} nxsgpu_batch_view_t;

/*
 * Pipelined host batches (up to NXSGPU_INFLIGHT in flight, shared with
 * nxsgpu_search_dev_begin): the result of plans[i] goes to record
 * slot_of_plan[i] of this rank's block; status[NXSGPU_STATUS_WORDS(n_slots)]
 * (NULL: zeros) travels with it.
 * _begin() queues upload, cursors, scans, heap replays, the all-gather (if
 * `gather` is set and a communicator is attached: every rank must then call
 * with the same n_slots and limit) and the copy to pinned host memory, then
 * returns; _end() waits for
 * the OLDEST batch in flight and hands out the blocks.  0 / -1.
 */
int		nxsgpu_batch_begin(nxsgpu_index_t *, int algo, uint32_t limit,
		    const nxsgpu_query_t *plans, uint32_t n_plans,
		    const uint32_t *slot_of_plan, const uint32_t *status,
		    uint32_t n_slots, int gather);
int		nxsgpu_batch_end(nxsgpu_index_t *, nxsgpu_batch_view_t *);
/* number of batches in flight (either API) */
int		nxsgpu_batches_in_flight(const nxsgpu_index_t *);

/*
 * The caller's worker threads for the host side of a batch (per-query plan -> device form:
 * embarrassingly parallel, 0.15 ms on one thread for 1024 five-term queries): `run` executes
 * body(arg, lo, hi) over [0, n) in chunks on whatever threads it has and returns when all of it
 * is done.  NULL: one thread.  nxs_api.c hands over the nxs_t's pool (the one that parses).
 */
typedef void (*nxsgpu_body_t)(void *arg, size_t lo, size_t hi);
typedef void (*nxsgpu_parallel_t)(void *ctx, nxsgpu_body_t body, void *arg, size_t n, size_t chunk);
void		nxsgpu_index_set_parallel(nxsgpu_index_t *, nxsgpu_parallel_t run, void *ctx);

/* re-read the NXS_GPU_* switches (they are parsed once at index create);
 * tests and A/B tools only */
void		nxsgpu_index_reconfigure(nxsgpu_index_t *);

/*
 * Measured HBM roofline on THIS device: streams the index's own posting array
 * (16 B per lane, every CU busy) `reps` times and reports the best read rate in
 * GB/s, the denominator bench.py prints beside the nominal 8 TB/s.  0 on error.
 */
double		nxsgpu_hbm_read_gbs(nxsgpu_index_t *, int reps);
/* one launch of k_hbm_read (16 B/lane) and of k_hbm_read_x2 (8 B/lane, the scan
 * kernels' width) over *bytes bytes each: known byte counts to calibrate the
 * FETCH_SIZE counter against (tools/pmc_calib.py) */
int		nxsgpu_hbm_calibrate(nxsgpu_index_t *, uint64_t *bytes);

void		nxsgpu_set_profiling(nxsgpu_index_t *, int on);
void		nxsgpu_get_profile(nxsgpu_index_t *, nxsgpu_profile_t *, int reset);
void		nxsgpu_synchronize(nxsgpu_index_t *);

#ifdef __cplusplus
}
#endif
#endif

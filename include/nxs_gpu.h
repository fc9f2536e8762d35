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
#define	NXSGPU_FAST_K		64	/* limit up to which top-k stays in LDS */

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

typedef struct {
	uint64_t	launches;	/* launches of the dominant scan kernel */
	double		scan_ms;	/* summed HIP-event time of those      */
	double		replay_ms;
	double		fuzzy_ms;
	uint64_t	postings;	/* summed algorithmic postings         */
	uint64_t	fuzzy_visits;
} nxsgpu_profile_t;

int		nxsgpu_device_count(void);
const char *	nxsgpu_last_error(void);

nxsgpu_index_t *nxsgpu_index_create(int device, const nxsgpu_index_src_t *);
void		nxsgpu_index_destroy(nxsgpu_index_t *);

/* document frequency per term id [n_terms+1] (host buffer) */
int		nxsgpu_index_df(nxsgpu_index_t *, uint32_t *df);
uint64_t	nxsgpu_index_postings(const nxsgpu_index_t *);
uint64_t	nxsgpu_index_docs(const nxsgpu_index_t *);
/* first live doc (file order) whose block names an unknown term, or ~0 */
uint64_t	nxsgpu_index_first_bad_doc(const nxsgpu_index_t *);

int		nxsgpu_search(nxsgpu_index_t *, int algo, uint64_t limit,
		    const nxsgpu_query_t *queries, uint32_t n_queries,
		    nxsgpu_results_t *res);
void		nxsgpu_results_free(nxsgpu_results_t *);

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
 * two batches may be in flight, so the host prepares batch i+1 while batch i
 * runs; each needs its own output buffers until its _end().  nxsgpu_search()
 * and nxsgpu_search_dev() refuse to run while a batch is in flight.
 */
int		nxsgpu_search_dev_begin(nxsgpu_index_t *, int algo, uint32_t limit,
		    const nxsgpu_query_t *queries, uint32_t n_queries,
		    uint64_t *d_doc_ids, float *d_scores, uint32_t *d_counts);
int		nxsgpu_search_dev_end(nxsgpu_index_t *);

int		nxsgpu_fuzzy(nxsgpu_index_t *, const uint8_t *tok_bytes,
		    const uint32_t *tok_off, uint32_t n_tokens,
		    uint32_t *term_ids, uint64_t *visited);

void		nxsgpu_set_profiling(nxsgpu_index_t *, int on);
void		nxsgpu_get_profile(nxsgpu_index_t *, nxsgpu_profile_t *, int reset);
void		nxsgpu_synchronize(nxsgpu_index_t *);

#ifdef __cplusplus
}
#endif
#endif

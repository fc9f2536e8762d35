/*
 * nxs_impl.h -- private declarations of the C11 host side.
 *
 * Host responsibilities (everything that is not per-posting or per-BK-node
 * work): mapping and validating the two index files, the term dictionary,
 * building and flattening the BK-tree image, query lexing/parsing, token
 * resolution, compiling a query into the device plan, and the nxs_resp_t
 * object.  All per-posting and per-node work happens in nxs_gpu.hip.
 */
#ifndef NXS_IMPL_H
#define NXS_IMPL_H

#include <stddef.h>
#include <stdint.h>
#include <stdbool.h>

#include "nxs.h"
#include "nxs_gpu.h"

#define	NXS_DEFAULT_RESULTS_LIMIT	1000	/* nxs_impl.h:39 of the reference */
#define	NXS_QUERY_RLIMIT		100	/* search.c:70 */
#define	LEVDIST_TOLERANCE		2	/* index.h:26 */

struct nxs_pool;

struct nxs {
	char *		basedir;
	char *		errmsg;
	nxs_err_t	errcode;
	nxs_index_t **	indexes;
	size_t		n_indexes;
	struct nxs_pool *pool;		/* host workers of the batch front half */
	bool		pool_tried;
};

void	nxs_clear_error(nxs_t *);
void	nxs_decl_err(nxs_t *, nxs_err_t, const char *fmt, ...)
	    __attribute__((format(printf, 3, 4)));

/* ---- params ---------------------------------------------------------- */

typedef enum { PV_STR, PV_UINT, PV_BOOL, PV_NONE } pv_type_t;	/* PV_NONE: a JSON member of a kind no getter reads */

typedef struct {
	char *		key;
	pv_type_t	type;
	char *		s;
	uint64_t	u;
	bool		b;
} param_kv_t;

struct nxs_params {
	param_kv_t *	kv;
	size_t		n;
};

const char *nxs_params_get_str(const nxs_params_t *, const char *);
int	nxs_params_get_uint(const nxs_params_t *, const char *, uint64_t *);
int	nxs_params_get_bool(const nxs_params_t *, const char *, bool *);

/* ---- index ----------------------------------------------------------- */

typedef struct {
	const uint8_t *	val;		/* into the mapped nxsterms image */
	uint32_t	tot_off;	/* offset of the u64 BE total counter; 0 = dead id */
	uint16_t	len;
} hterm_t;

struct qprep;

/* one batch between nxs_index_search_batch_begin and _end */
typedef struct nxs_pend {
	bool		active;
	bool		on_device;	/* queued through nxsgpu_batch_begin */
	size_t		n;		/* queries of the whole batch */
	size_t		lo, hi;		/* this rank's slice */
	uint32_t	cap;		/* record slots per rank */
	int		rank, world;
	uint64_t	limit;
	int		algo;
	struct qprep *	prep;		/* [hi - lo] */
	uint64_t	seq;
	/*
	 * The batch's misses are being resolved on the device (nxsgpu_fuzzy_begin) and _begin has returned:
	 * the rest of its front half -- winners into the token lists, compile, queueing on the device -- is
	 * done by the next _begin (after ITS parse) or by this batch's _end, whichever comes first
	 * (nxs_api.c: late_finish).  NULL otherwise.
	 */
	struct late_half *late;
	/*
	 * Collected early: the index files changed while this batch was in flight and
	 * a later _begin had to finish it to re-sync (search.c:309-312).  Its
	 * responses wait here for the caller's _end.
	 */
	bool		stashed;
	nxs_resp_t **	st_resps;	/* [n] */
	nxs_err_t *	st_errs;	/* [n] */
	int		st_ret;
	nxs_err_t	st_errcode;
	char *		st_errmsg;
} nxs_pend_t;

struct nxs_index {
	nxs_t *		nxs;
	char *		name;
	int		algo;		/* NXSGPU_BM25 / NXSGPU_TF_IDF */
	bool		lowercase;	/* open_files(): the pipeline is { normalizer } */
	struct nxs_filters *filters;	/* filter pipeline of query tokens, or NULL */

	uint8_t *	tmap;	size_t tmap_len;
	uint8_t *	dmap;	size_t dmap_len;

	/* term dictionary: ids 1..last_id (file order, terms.c:404) */
	hterm_t *	terms;
	uint32_t	last_id;
	uint32_t	term_count;
	uint32_t *	thash;		/* open addressing: term ids */
	size_t		thash_cap;

	uint64_t	n_docs;
	nxsgpu_index_t *dev;
	int		device;		/* HIP device of the index (NXS_GPU_DEVICE at open) */

	/* query sharding (nxs_index_shard) and the batches in flight */
	nxsgpu_comm_t *	comm;
	struct nxs_pend	pend[NXSGPU_INFLIGHT];
	/* tests: the n-th next _begin / exact fix-up of this index fails (0: off) */
	unsigned	test_fail_begin, test_fail_fixup, test_fail_fixup_recv, test_fail_late;
	bool		resync_pending;	/* sharded: a rank's block flags said its files moved */
	struct plan_cache *pcache;	/* query string -> compiled plan (nxs_api.c: plan_batch) */
	uint64_t	pend_seq;
	/* host-side phase times of the batches, seconds (nxs_index_host_profile) */
	double		hp_plan, hp_queue, hp_wait, hp_resps, hp_begin, hp_end, hp_fzwait, hp_front, hp_fzlaunch, hp_back;
	uint64_t	hp_batches, hp_inexact;
	/* doc-sharded mode (N4): this index is shard `shard` of `n_shards` (0 = whole) */
	unsigned	shard, n_shards;
	int		want_device;	/* explicit device + 1, or 0: NXS_GPU_DEVICE / device 0 */
	bool		global_df_set;
	bool		shard_local;	/* nxs_index_shard_local: responses of the own slice only */
	int8_t		late_mode;	/* 0 not read yet, 1 a batch's fuzzy pass is left running (late_finish), 2 NXS_LATE_FUZZY=0:
					 * _begin waits for it itself */
	/* tests: play one rank of emu_world (nxs_test_shard_emulate) */
	int		emu_rank, emu_world;
	uint8_t *	emu_block;
	size_t		emu_block_len;

	/* what idx_terms_sync / idx_dtmap_sync have consumed (bytes of the data areas)
	 * and the header counters the device statistics were computed from */
	char *		terms_path;
	char *		dtmap_path;
	uint64_t	terms_consumed, dt_consumed;
	uint64_t	hdr_docs_seen, hdr_tokens_seen;
	size_t		terms_cap;
	/* persistent doc table, ordinal order (= ascending doc id) */
	uint64_t *	h_doc_ids;
	uint64_t *	h_blk_off;	/* block offset in the nxsdtmap image */
	uint32_t *	h_npairs;
	uint8_t *	h_alive;
	uint64_t	n_ord, cap_ord;
	/* host BK-tree (terms inserted up to bk_upto) behind the device image */
	struct nxs_bktree *bktree;
	uint32_t	bk_upto;
	bool		bk_flags_stale;
	uint64_t	n_incremental, n_rebuilds;
};

/* nxs_index.c */
int	nxs_index_load(nxs_index_t *, const char *terms_path, const char *dtmap_path);
void	nxs_index_unload(nxs_index_t *);
int	nxs_index_refresh(nxs_index_t *);
bool	nxs_index_changed(const nxs_index_t *);
int	nxs_index_bk_sync(nxs_index_t *);
void	nxs_index_refresh_stats(const nxs_index_t *, uint64_t out[2]);
uint32_t nxs_term_lookup(const nxs_index_t *, const uint8_t *val, size_t len);

/* flattened BK-tree built on the host (exported for the CPU-side tests) */
typedef struct {
	nxsgpu_bknode_t *nodes;
	uint32_t	n;
	uint32_t	depth;
	uint8_t *	bytes;
	uint64_t	bytes_len;
} nxs_bkimage_t;

typedef struct nxs_bktree nxs_bktree_t;

nxs_bktree_t *nxs_bktree_create(void);
void	nxs_bktree_destroy(nxs_bktree_t *);
int	nxs_bktree_insert(nxs_bktree_t *, const hterm_t *terms, uint32_t from_id, uint32_t last_id);
int	nxs_bk_flatten(const nxs_bktree_t *, const hterm_t *terms, const uint8_t *tmap,
	    nxs_bkimage_t *out);
int	nxs_bk_build(const hterm_t *terms, uint32_t last_id, const uint8_t *tmap,
	    nxs_bkimage_t *out);
void	nxs_bk_free(nxs_bkimage_t *);

/* ---- filter pipeline of query tokens (nxs_filters.c) ------------------------ */

typedef struct nxs_filters nxs_filters_t;

nxs_filters_t *nxs_filters_create(const char *basedir, const char *const *names, size_t n,
	    const char *lang, const char **err);
void	nxs_filters_destroy(nxs_filters_t *);
int	nxs_filters_run(nxs_filters_t *, char **val, size_t *len);
/* the English Snowball stemmer (Porter2) on a UTF-8 token, in place -> new length (nxs_stem_en.c) */
size_t	nxs_stem_en(char *w, size_t len);

/* ---- query ----------------------------------------------------------- */

typedef enum {
	QTK_EOF = 0, QTK_AND, QTK_OR, QTK_NOT, QTK_BR_OPEN, QTK_BR_CLOSE,
	QTK_FF_STRING, QTK_QUOTED_STRING,
} qtoken_t;

/* postfix item: a leaf string or an operator */
typedef struct {
	uint8_t		op;	/* 0 = leaf, else NXSGPU_OP_AND/OR/ANDNOT */
	char *		str;	/* leaf value (owned) */
	int		token;	/* leaf: index into the token list, -1 = none */
} qitem_t;

typedef struct {
	qitem_t *	items;
	size_t		n;
	bool		error;
	char *		errmsg;
	/* one block for everything the front half of a query allocates */
	char *		arena;
	size_t		arena_used, arena_cap;
} qparse_t;

int	nxs_query_lex(const char *query, int *kinds, size_t cap);
void	nxs_query_parse(const char *query, qparse_t *out);
void	nxs_query_free(qparse_t *);
char *	nxs_query_repr(const qparse_t *);

typedef struct {
	char *		value;
	size_t		len;
	uint32_t	term_id;	/* 0 = unresolved */
} qtok_t;

typedef struct qprep {
	qparse_t	parse;
	qtok_t *	tokens;		/* token-list order (query.c:89-95) */
	size_t		n_tokens;
	nxs_err_t	errcode;	/* set when the query cannot run */
	char *		errmsg;
	bool		empty;		/* no live tokens: empty result */
	bool		wide;		/* does not fit nxsgpu_query_t: wplan is its plan */
	bool		cached;		/* the plan came from the index's plan cache: nothing was parsed */
	bool		compiled;	/* plan_batch: compiled in the first pass already (no token waits for the fuzzy search) */
	nxsgpu_query_t	plan;
	nxsgpu_wide_query_t wplan;	/* arrays owned by this object */
	char **		heap_vals;	/* token values a filter grew beyond the arena's reserve */
	size_t		n_heap_vals;
} qprep_t;

void	nxs_query_prepare(const nxs_index_t *, const char *query, qprep_t *out);
int	nxs_query_compile(qprep_t *);	/* after term ids are final */
void	nxs_query_release(qprep_t *);
void	nxs_query_release_scratch(qprep_t *);	/* parse + token list; keeps errors and plans */

#endif

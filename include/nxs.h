/*
 * nxs.h -- query-side C API of the MI355X-native nxsearch ranking engine.
 *
 * Drop-in for the query path of the reference's public header
 * (reference src/core/nxs.h:21-101, docs/c-api.md:113-148): same names,
 * signatures, ownership and error conventions, so that a program linked
 * against libnxsearch can be relinked against libnxsearch_gpu.so for
 * searching an index that the reference (or anything writing the same
 * on-disk format, src/index/storage.h:13-134) has produced.
 *
 * Not provided (out of scope): index creation and
 * mutation (nxs_index_create/add/remove/destroy), Lua filters.
 *
 * Added: nxs_index_search_batch() -- the reference API is one query per call;
 * a GPU wants many (SURVEY.md 8b last row).
 */
#ifndef NXS_GPU_PUBLIC_H
#define NXS_GPU_PUBLIC_H

#include <stddef.h>
#include <stdint.h>
#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint64_t nxs_doc_id_t;			/* nxs.h:21 */

struct nxs;
typedef struct nxs nxs_t;

nxs_t *		nxs_open(const char *basedir);	/* nxs.h:26, nxs.c:91-133 */
void		nxs_close(nxs_t *);		/* nxs.h:27 */

/* nxs.h:33-46 -- ABI-frozen codes */
typedef enum {
	NXS_ERR_SUCCESS		= 0,
	NXS_ERR_FATAL,
	NXS_ERR_SYSTEM,
	NXS_ERR_INVALID,
	NXS_ERR_EXISTS,
	NXS_ERR_MISSING,
	NXS_ERR_LIMIT,
} nxs_err_t;

nxs_err_t	nxs_get_error(const nxs_t *, const char **);	/* nxs.h:48 */

/* Parameters (nxs.h:54-67); the query path reads limit / algo / fuzzymatch */
struct nxs_params;
typedef struct nxs_params nxs_params_t;

nxs_params_t *	nxs_params_create(void);
/* params.c:201-208: a JSON object of string / unsigned / bool members (how the
 * Lua binding passes limit, algo, fuzzymatch: lua.c:99-110); NULL + NXS_ERR_SYSTEM
 * "params parsing failed: ..." on a syntax error */
nxs_params_t *	nxs_params_fromjson(nxs_t *, const char *, size_t);
int		nxs_params_set_str(nxs_params_t *, const char *, const char *);
int		nxs_params_set_uint(nxs_params_t *, const char *, uint64_t);
int		nxs_params_set_bool(nxs_params_t *, const char *, bool);
void		nxs_params_release(nxs_params_t *);

/* Index handles (nxs.h:73-85): open/close only */
struct nxs_index;
typedef struct nxs_index nxs_index_t;

nxs_index_t *	nxs_index_open(nxs_t *, const char *name);
void		nxs_index_close(nxs_index_t *);

/* Query and response API (nxs.h:87-101) */
struct nxs_resp;
typedef struct nxs_resp nxs_resp_t;

nxs_resp_t *	nxs_index_search(nxs_index_t *, nxs_params_t *,
		    const char *query, size_t len);

void		nxs_resp_iter_reset(nxs_resp_t *);
bool		nxs_resp_iter_result(nxs_resp_t *, nxs_doc_id_t *, float *);
unsigned	nxs_resp_resultcount(const nxs_resp_t *);
char *		nxs_resp_tojson(nxs_resp_t *, size_t *);
void		nxs_resp_release(nxs_resp_t *);

/*
 * Batch entry point (new).  Runs `n` queries with one set of params as one
 * device batch.  resps[i] receives a response object or NULL if query i
 * failed (its code/message are then in errs[i]/the nxs_t error slot for the
 * last failure).  Returns the number of failed queries, or -1 if the batch as
 * a whole could not run (nxs_get_error() tells why).
 */
int		nxs_index_search_batch(nxs_index_t *, nxs_params_t *,
		    const char *const *queries, size_t n,
		    nxs_resp_t **resps, nxs_err_t *errs);

/*
 * The same, split for pipelining (new): _begin() parses, resolves and plans
 * the batch on the host (worker threads), queues it on the device and returns;
 * _end() waits for the OLDEST batch in flight and builds its responses.  Up to
 * NXS_BATCHES_INFLIGHT batches may be in flight, so the host prepares batch i+1
 * while the GPU runs batch i (two hide the host side of a top-10 batch; at the
 * default limit of 1000 a batch ends in milliseconds of heap replay on a nearly
 * idle chip, and three or four in flight fill it).  `queries` need not outlive _begin().  The index re-syncs with
 * the files (search.c:309-312) in every _begin: when they have moved while batches
 * are in flight, _begin finishes those first (their responses wait for _end).
 */
#define	NXS_BATCHES_INFLIGHT	4
int		nxs_index_search_batch_begin(nxs_index_t *, nxs_params_t *,
		    const char *const *queries, size_t n);
int		nxs_index_search_batch_end(nxs_index_t *, nxs_resp_t **resps,
		    nxs_err_t *errs);

/*
 * Query sharding over the GPUs of one node (new; the reference scales out with
 * independent worker processes, compose/nginx.conf:2).  One process per GPU,
 * each with a replica of the index (NXS_GPU_DEVICE selects the device at open).
 * Rank 0 calls nxs_shard_unique_id(), the application hands the
 * NXS_SHARD_UID_BYTES bytes to the other ranks, then EVERY rank calls
 * nxs_index_shard() (collective: it builds an RCCL communicator).  From then on
 * every rank passes the SAME batch to nxs_index_search_batch[_begin]; each
 * rank runs its contiguous slice, one RCCL all-gather of fixed-size per-query
 * records over xGMI reassembles the batch, and every rank receives all n
 * responses.  Applies to limit <= 8000 (NXSGPU_BIG_K: fixed-size records); larger
 * limits run replicated (every rank computes the whole batch).  nxs_index_search()
 * (one query) never shards.  Re-sync with a communicator attached: a rank that sees
 * the files move while batches are in flight says so in its record block; all ranks
 * read that at the batch's _end and finish their batches in flight + re-read the
 * files in their next _begin (the same one on every rank: two batches after the
 * change was noticed in a fully pipelined loop).  A rank that cannot do its share of a batch still contributes an
 * "aborted" block, so every rank fails that batch together and the next one is in
 * step -- only a rank that cannot reach the collective at all (device memory for
 * the staging buffer, a dead process) stalls the group, as with any collective.
 * nxs_index_shard(idx, 0, 1, NULL) detaches.
 */
#define	NXS_SHARD_UID_BYTES	128
int		nxs_shard_unique_id(nxs_t *, uint8_t *uid);
int		nxs_index_shard(nxs_index_t *, int rank, int world, const uint8_t *uid);
/*
 * nxs_index_shard_local(idx, true): like the reference's worker processes, which answer only their own
 * requests (compose/nginx.conf:2), a rank then materialises the responses of ITS slice only: after _end
 * resps[i] is NULL and errs[i] untouched for the queries other ranks own, and the return value counts the
 * own slice's failures -- O(n / world) host work per rank and batch.  nxs_index_shard_slice() tells which
 * part [*lo, *hi) of an n-query batch that is (the whole batch when the mode is off or nothing is attached).
 */
int		nxs_index_shard_local(nxs_index_t *, bool on);
void		nxs_index_shard_slice(const nxs_index_t *, size_t n, size_t *lo, size_t *hi);

/*
 * Front half of a batch only: parse + resolve (fuzzy misses on the device) +
 * compile into device plans (struct nxsgpu_query = nxsgpu_query_t of
 * nxs_gpu.h), for callers that keep the results on the device
 * (nxsgpu_search_dev).  plans[i].n_tokens == 0 when query i matches nothing
 * or failed (errs[i] then holds its code).  Returns #failed or -1.
 */
struct nxsgpu_query;
int		nxs_index_plan_batch(nxs_index_t *, nxs_params_t *,
		    const char *const *queries, size_t n,
		    struct nxsgpu_query *plans, nxs_err_t *errs);

/*
 * Opens an index straight from the two files (no basedir/params.db): the
 * entry used by the bench and tests for synthetic corpora.  `algo` is the
 * index default ("BM25" / "TF-IDF"); `lowercase` enables the ASCII part of
 * the reference's "normalizer" filter for query tokens.
 */
nxs_index_t *	nxs_index_open_files(nxs_t *, const char *terms_path,
		    const char *dtmap_path, const char *algo, bool lowercase);

/*
 * Doc-sharded mode (new): for collections beyond one GPU's
 * memory.  Shard s of S holds the docs of rank [D*s/S, D*(s+1)/S) in ascending
 * doc id, scored with collection-wide statistics; nxs_docshard_search_batch()
 * runs a batch on every shard (each on its own device, `device` < 0 = the
 * NXS_GPU_DEVICE default) and merges the shards' candidates through one more
 * exact heap replay: the responses equal those of the unsharded index, ties
 * included.  limit <= 8000, at most 32 query terms; a shard is a static snapshot (no re-sync).  The
 * shards' passes run concurrently (every shard has its own device / streams).
 *
 * One process per shard (one GPU each): rank r opens shard r of W, attaches a
 * communicator of W ranks (nxs_index_shard), joins the collection
 * (nxs_docshard_attach: all-gather + sum of the shards' df, collective) and
 * then every rank calls nxs_docshard_search_batch_rank() with the SAME batch:
 * each runs its shard, ONE all-gather of the ranks' candidate blocks, every
 * rank merges and holds all responses.
 */
nxs_index_t *	nxs_index_open_shard(nxs_t *, const char *terms_path,
		    const char *dtmap_path, const char *algo, bool lowercase,
		    unsigned shard, unsigned n_shards, int device);
int		nxs_docshard_search_batch(nxs_index_t *const *shards, unsigned n_shards,
		    nxs_params_t *, const char *const *queries, size_t n,
		    nxs_resp_t **resps, nxs_err_t *errs);
int		nxs_docshard_attach(nxs_index_t *shard);
int		nxs_docshard_search_batch_rank(nxs_index_t *shard, nxs_params_t *,
		    const char *const *queries, size_t n,
		    nxs_resp_t **resps, nxs_err_t *errs);

/* (test hooks and the bench's accessors -- nxs_index_host_profile, nxs_index_device, nxs_test_* -- are
 * not part of this ABI: nxsearch_amd/csrc/nxs_hooks.h, builds with -DNXS_TEST_HOOKS only) */

#ifdef __cplusplus
}
#endif
#endif

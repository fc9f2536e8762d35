/*
 * nxs_hooks.h -- NOT part of the library's ABI: test hooks (nxs_test_*) and the bench's
 * accessors.  They exist only in builds with -DNXS_TEST_HOOKS (the Makefile's default,
 * which is what tests/ and bench.py load; `make HOOKS=` builds the library without them:
 * the drop-in a consumer of include/nxs.h links).  Python reaches them through ctypes
 * (nxsearch_amd/__init__.py, nxsearch_amd/multi.py).
 */
#ifndef NXS_HOOKS_H
#define NXS_HOOKS_H
#ifdef NXS_TEST_HOOKS

#include "nxs_impl.h"

/*
 * Host-side phase times of the batches since the last call, in seconds:
 * out[0] parse/resolve/compile, out[1] queueing on the device, out[2] waiting
 * for the device, out[3] building responses, out[4] number of batches, out[5]
 * queries that had to be re-run on the exact two-pass path, out[6] / out[7] the
 * whole _begin() / _end() calls.
 */
void		nxs_index_host_profile(nxs_index_t *, double out[12]);
/* out[0] ncclCommCount of the attached communicator (-1: none / unknown), out[1] world, out[2] all-gathers queued,
 * out[3] bytes this rank contributed to them */
void		nxs_index_shard_info(nxs_index_t *, uint64_t out[4]);
/* the device-side handle behind an index (nxs_gpu.h): pre-resolved plans, results left in HBM */
struct nxsgpu_index;
struct nxsgpu_index *nxs_index_device(nxs_index_t *);
/* the plan cache (query string -> compiled plan) on / off at run time */
void		nxs_index_set_plan_cache(nxs_index_t *, int on);

/* worker pool: every item of every run worked on exactly once */
size_t		nxs_test_pool(unsigned n_thr, size_t n, unsigned rounds, size_t chunk);
/* parser / plan compiler without an index */
char *		nxs_test_query_repr(const char *query, char **errmsg);
int		nxs_test_compile(const char *query, const char *const *words, uint32_t n_words,
		    bool lowercase, nxsgpu_query_t *plan, int *empty, char *err, size_t errlen);
int		nxs_test_compile_wide(const char *query, const char *const *words, uint32_t n_words,
		    int *wide, uint32_t *n_tokens, uint32_t *term_ids, uint32_t cap_t,
		    uint32_t *prog_len, uint16_t *prog, uint32_t cap_p);
char *		nxs_test_filter(const char *basedir, int stages, const char *s, int *act);
int		nxs_test_bk_image(const char *const *words, uint32_t n_words, nxs_bkimage_t *out);
int		nxs_test_levdist(const uint8_t *a, size_t n, const uint8_t *b, size_t m);
/* query sharding without a second GPU: one emulated rank, record blocks, the fix-up protocol */
void		nxs_test_shard_emulate(nxs_index_t *, int rank, int world);
size_t		nxs_test_shard_block(nxs_index_t *, uint8_t *out, size_t cap);
void		nxs_test_pack_record(uint8_t *block, uint32_t n_slots, uint32_t k, uint32_t slot,
		    uint32_t count, const uint64_t *ids, const float *scores, uint32_t status);
void		nxs_test_mark_inexact(uint8_t *block, uint32_t n_slots, uint32_t k, uint32_t slot);
void		nxs_test_mark_changed(uint8_t *block, uint32_t n_slots, uint32_t k);
int		nxs_test_blocks_changed(const uint8_t *blocks, uint32_t world, uint32_t n_slots, uint32_t k);
void		nxs_test_pack_abort(uint8_t *block, uint32_t n_slots, uint32_t k, uint32_t code);
int		nxs_test_fixup_scan(const uint8_t *blocks, uint32_t world, uint32_t n_slots, uint32_t k,
		    size_t n, int rank, uint32_t *which, size_t *nw);
int		nxs_test_fixup_verify(const uint8_t *blocks, uint32_t world, uint32_t n_slots, uint32_t k, size_t n);
int		nxs_test_assemble(const uint8_t *blocks, uint32_t world, uint32_t n_slots, uint32_t k,
		    size_t n, nxs_resp_t **resps, nxs_err_t *errs, int only_rank);
void		nxs_test_inject_failure(nxs_index_t *, int which, unsigned nth);
/* doc shards: the two halves of the rank form */
int		nxs_test_docshard_block(nxs_index_t *shard, nxs_params_t *, const char *const *queries, size_t n,
		    uint32_t cap, uint8_t **block, size_t *len);
int		nxs_test_docshard_finish(nxs_index_t *shard, nxs_params_t *, const char *const *queries, size_t n,
		    uint32_t cap, const uint8_t *gathered, nxs_resp_t **resps, nxs_err_t *errs);
int		nxs_test_docshard_set_df(nxs_index_t *const *shards, unsigned n_shards);

#endif /* NXS_TEST_HOOKS */
#endif /* NXS_HOOKS_H */

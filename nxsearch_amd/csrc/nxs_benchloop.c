/*
 * nxs_benchloop.c -- the C consumer bench.py times (libnxsbench.so).
 *
 * The reference's own measuring tool is a C program that calls the public API
 * in a loop and walks the results (src/utils/benchmark.c:199-215); this is the
 * same thing for the batch entry points, so that the timed region holds no
 * Python: strings in, nxs_resp_t out, every result read, every response
 * released.  Links against libnxsearch_gpu.so only through include/nxs.h.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "nxs.h"

typedef struct {
	double		seconds;
	uint64_t	results;	/* (doc, score) pairs read */
	uint64_t	checksum;	/* over doc ids and score bits */
	uint64_t	failed;		/* queries without a response */
} nxs_bench_out_t;

static double
now_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* (a rank of a sharded index in nxs_index_shard_local mode delivers -- and a worker walks -- its own slice only) */
static size_t own_lo, own_hi;

static void
consume_range(nxs_resp_t **resps, size_t lo, size_t hi, nxs_bench_out_t *o);

static void
consume(nxs_resp_t **resps, size_t n, nxs_bench_out_t *o)
{
	if (own_hi > own_lo && own_hi <= n) {
		consume_range(resps, own_lo, own_hi, o);
	} else {
		consume_range(resps, 0, n, o);
	}
}

static void
consume_range(nxs_resp_t **resps, size_t lo, size_t hi, nxs_bench_out_t *o)
{
	for (size_t i = lo; i < hi; i++) {
		nxs_doc_id_t id;
		float sc;

		if (!resps[i]) {
			o->failed++;
			continue;
		}
		nxs_resp_iter_reset(resps[i]);
		while (nxs_resp_iter_result(resps[i], &id, &sc)) {
			uint32_t b;
			memcpy(&b, &sc, 4);
			o->checksum = (o->checksum * 0x9e3779b97f4a7c15ULL) ^ id ^ ((uint64_t)b << 32);
			o->results++;
		}
		nxs_resp_release(resps[i]);
	}
}

/*
 * `steps` batches of n query strings; step s takes batch s mod n_sets of the
 * n_sets * n strings (distinct batches: nothing a step warms up -- term hash
 * lines, cursor searches, dense columns -- is what the next one reads).  depth 1:
 * the blocking nxs_index_search_batch(); depth d >= 2: nxs_index_search_batch_begin /
 * _end with d batches in flight (the host plans batch i+1 while the GPU runs
 * batch i).  All steps have completed and all responses are consumed on return.
 */
int
nxs_bench_batches_rot(nxs_index_t *idx, nxs_params_t *params, const char *const *all_queries,
    size_t n, unsigned n_sets, unsigned steps, int depth, nxs_bench_out_t *out)
{
	nxs_resp_t **resps = calloc(n ? n : 1, sizeof(*resps));
	nxs_err_t *errs = calloc(n ? n : 1, sizeof(*errs));
	int ret = -1;
	double t0;

	memset(out, 0, sizeof(*out));
	if (!resps || !errs || n_sets == 0) {
		goto out;
	}
	nxs_index_shard_slice(idx, n, &own_lo, &own_hi);
	t0 = now_s();
	if (depth <= 1) {
		for (unsigned s = 0; s < steps; s++) {
			const char *const *queries = all_queries + (size_t)(s % n_sets) * n;
			if (nxs_index_search_batch(idx, params, queries, n, resps, errs) < 0) {
				goto out;
			}
			consume(resps, n, out);
		}
	} else {
		/*
		 * depth d: d batches in flight.  The responses of batch s are consumed AFTER
		 * batch s + d has been queued (a server hands the next batch to the GPU before
		 * it walks the previous one's results): at the default limit that walk is
		 * half a million results per batch and would otherwise sit between two
		 * batches' device work.
		 */
		nxs_resp_t **held = calloc(n ? n : 1, sizeof(*held));
		unsigned ended = 0;
		bool have = false;

		if (!held) {
			goto out;
		}
		if (depth > NXS_BATCHES_INFLIGHT) {
			depth = NXS_BATCHES_INFLIGHT;
		}
		for (unsigned s = 0; s < steps; s++) {
			const char *const *queries = all_queries + (size_t)(s % n_sets) * n;
			if (nxs_index_search_batch_begin(idx, params, queries, n) != 0) {
				free(held);
				goto out;
			}
			if (have) {
				consume(held, n, out);
				have = false;
			}
			if (s + 1 >= (unsigned)depth) {
				if (nxs_index_search_batch_end(idx, held, errs) < 0) {
					free(held);
					goto out;
				}
				ended++;
				have = true;
			}
		}
		while (ended < steps) {
			if (have) {
				consume(held, n, out);
			}
			if (nxs_index_search_batch_end(idx, held, errs) < 0) {
				free(held);
				goto out;
			}
			ended++;
			have = true;
		}
		if (have) {
			consume(held, n, out);
		}
		free(held);
	}
	out->seconds = now_s() - t0;
	ret = 0;
out:
	free(resps);
	free(errs);
	return ret;
}

int
nxs_bench_batches(nxs_index_t *idx, nxs_params_t *params, const char *const *queries,
    size_t n, unsigned steps, int depth, nxs_bench_out_t *out)
{
	return nxs_bench_batches_rot(idx, params, queries, n, 1, steps, depth, out);
}

/* one nxs_index_search() per query string; per-call wall time in microseconds */
int
nxs_bench_singles(nxs_index_t *idx, nxs_params_t *params, const char *const *queries,
    size_t n, double *lat_us, nxs_bench_out_t *out)
{
	memset(out, 0, sizeof(*out));
	for (size_t i = 0; i < n; i++) {
		const double t0 = now_s();
		nxs_resp_t *r = nxs_index_search(idx, params, queries[i], strlen(queries[i]));
		lat_us[i] = 1e6 * (now_s() - t0);
		out->seconds += 1e-6 * lat_us[i];
		consume(&r, 1, out);
	}
	return 0;
}

/* `steps` doc-sharded batches (N4): every shard's pass concurrently, one merge */
int
nxs_bench_docshard(nxs_index_t *const *shards, unsigned n_shards, nxs_params_t *params,
    const char *const *all_queries, size_t n, unsigned n_sets, unsigned steps, nxs_bench_out_t *out)
{
	nxs_resp_t **resps = calloc(n ? n : 1, sizeof(*resps));
	nxs_err_t *errs = calloc(n ? n : 1, sizeof(*errs));
	int ret = -1;
	double t0;

	memset(out, 0, sizeof(*out));
	if (!resps || !errs || n_sets == 0) {
		goto out;
	}
	t0 = now_s();
	for (unsigned s = 0; s < steps; s++) {
		const char *const *queries = all_queries + (size_t)(s % n_sets) * n;
		if (nxs_docshard_search_batch(shards, n_shards, params, queries, n, resps, errs) < 0) {
			goto out;
		}
		consume(resps, n, out);
	}
	out->seconds = now_s() - t0;
	ret = 0;
out:
	free(resps);
	free(errs);
	return ret;
}

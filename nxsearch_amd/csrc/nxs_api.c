/*
 * nxs_api.c -- the public C API (include/nxs.h): library instance and error
 * slot, params, index open/close, nxs_index_search(), nxs_resp_*.
 *
 * Mirrors the reference's conventions for this path:
 *   error slot            src/core/nxs.c:154-217, nxs_impl.h:84-90
 *   search params         src/query/search.c:78-112  (limit / algo / fuzzymatch)
 *   nxs_index_search      src/query/search.c:285-342
 *   token resolution      src/core/tokenizer.c:160-199 (exact, else fuzzy)
 *   response object       src/core/results.c:46-246
 * All scoring, boolean filtering, top-k and fuzzy matching run on the GPU
 * through include/nxs_gpu.h; there is no CPU fallback: without a HIP device
 * nxs_index_open() fails.
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
#include <string.h>
#include <strings.h>
#include <limits.h>
#include <errno.h>
#include <sys/stat.h>
#include <time.h>
#include <pthread.h>
#include <stdatomic.h>
#include <unistd.h>

#include "nxs_impl.h"
#include "nxs_hooks.h"

struct plan_cache;
static void plan_cache_destroy(struct plan_cache *);

/* ---- host worker pool ------------------------------------------------------ */

/*
 * The front half of a batch -- lexing, parsing, token sets, dictionary lookups,
 * plan compilation -- is independent per query (query.c:75-115 works on one
 * query_t).  A small persistent pool spreads it over the host cores the process
 * may use; the pool belongs to the nxs_t (one per thread/process in the
 * reference's threading model, docs/c-api.md:5-8) and is created on the first
 * batch that is large enough to pay for a wake-up.
 */
typedef void (*pool_fn_t)(void *arg, size_t lo, size_t hi);

struct nxs_pool {
	pthread_t *	thr;
	unsigned	n_thr;
	pthread_mutex_t	mu;
	pthread_cond_t	cv_work, cv_done;
	uint64_t	gen;		/* run number (under mu) */
	bool		stop;
	bool		waiting;	/* the caller sleeps on cv_done */
	pool_fn_t	fn;
	void *		arg;
	size_t		n, chunk;
	/*
	 * next item to hand out, tagged with the run it belongs to: (gen << 40) | index.
	 * A worker that wakes up late -- after its run has ended, maybe inside the next
	 * one -- draws a ticket of another run and leaves without touching anything:
	 * a run therefore never waits for its slowest sleeper, only for its items (on a
	 * busy host waking 15 threads took 0.3-0.7 ms, twice per batch: the whole front
	 * half of a C3 step is 0.2 ms of work).
	 */
	_Atomic uint64_t next;
	atomic_size_t	done;		/* items of the current run completed */
	atomic_flag	busy;		/* a run is under way (pool_run takes one caller at a time) */
	/*
	 * gen as the workers may read it without the lock: a worker that has just finished a run
	 * polls it for POOL_SPIN_NS before it goes to sleep -- a batch's front half is two runs
	 * (parse, compile) a few dozen microseconds apart, and a pipelined server's next batch is
	 * a millisecond away: the second run finds its workers awake instead of paying the wake-up.
	 */
	_Atomic uint64_t gen_pub;
	long long	spin_ns;	/* NXS_POOL_SPIN_US (120; 0: sleep at once), read when the pool is created */
};
#define	POOL_GEN_SHIFT	40

static void
pool_work(struct nxs_pool *p, uint64_t gen, pool_fn_t fn, void *arg, size_t n, size_t chunk)
{
	for (;;) {
		/* draw a ticket of THIS run only (compare-and-swap: a latecomer of an earlier
		 * run must not take items away from the current one) */
		uint64_t tk = atomic_load(&p->next);
		size_t i;

		for (;;) {
			i = (size_t)(tk & ((1ull << POOL_GEN_SHIFT) - 1));
			if ((tk >> POOL_GEN_SHIFT) != (gen & 0xffffff) || i >= n) {
				return;
			}
			if (atomic_compare_exchange_weak(&p->next, &tk, tk + (uint64_t)chunk)) {
				break;
			}
		}
		const size_t hi = i + chunk < n ? i + chunk : n;
		fn(arg, i, hi);
		if (atomic_fetch_add(&p->done, hi - i) + (hi - i) == n) {
			/* the last items of the run: wake the caller if it went to sleep */
			pthread_mutex_lock(&p->mu);
			if (p->waiting) {
				pthread_cond_signal(&p->cv_done);
			}
			pthread_mutex_unlock(&p->mu);
		}
	}
}

static void *
pool_main(void *arg)
{
	struct nxs_pool *p = arg;
	uint64_t seen = 0;

	pthread_mutex_lock(&p->mu);
	for (;;) {
		if (seen && p->spin_ns && p->gen == seen && !p->stop) {
			struct timespec t0, t1;

			pthread_mutex_unlock(&p->mu);
			clock_gettime(CLOCK_MONOTONIC, &t0);
			while (atomic_load_explicit(&p->gen_pub, memory_order_acquire) == seen) {
				for (int i = 0; i < 64; i++) {
					__builtin_ia32_pause();
				}
				clock_gettime(CLOCK_MONOTONIC, &t1);
				if ((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec) > p->spin_ns) {
					break;
				}
			}
			pthread_mutex_lock(&p->mu);
		}
		while (p->gen == seen && !p->stop) {
			pthread_cond_wait(&p->cv_work, &p->mu);
		}
		if (p->stop) {
			break;
		}
		seen = p->gen;
		/* (the run's description, read under the lock; a stale one is harmless: its
		 * tickets do not match) */
		const pool_fn_t fn = p->fn;
		void *const farg = p->arg;
		const size_t n = p->n, chunk = p->chunk;
		pthread_mutex_unlock(&p->mu);
		pool_work(p, seen, fn, farg, n, chunk);
		pthread_mutex_lock(&p->mu);
	}
	pthread_mutex_unlock(&p->mu);
	return NULL;
}

static struct nxs_pool *
pool_create(unsigned n_thr)
{
	struct nxs_pool *p = calloc(1, sizeof(*p));

	if (!p) {
		return NULL;
	}
	{
		const char *e = getenv("NXS_POOL_SPIN_US");
		const long v = e ? strtol(e, NULL, 10) : 120;
		p->spin_ns = (v < 0 ? 0 : v > 5000 ? 5000 : v) * 1000ll;
	}
	pthread_mutex_init(&p->mu, NULL);
	pthread_cond_init(&p->cv_work, NULL);
	pthread_cond_init(&p->cv_done, NULL);
	p->thr = calloc(n_thr ? n_thr : 1, sizeof(pthread_t));
	for (unsigned i = 0; p->thr && i < n_thr; i++) {
		if (pthread_create(&p->thr[p->n_thr], NULL, pool_main, p) != 0) {
			break;
		}
		p->n_thr++;
	}
	return p;
}

static void
pool_destroy(struct nxs_pool *p)
{
	if (!p) {
		return;
	}
	pthread_mutex_lock(&p->mu);
	p->stop = true;
	atomic_store_explicit(&p->gen_pub, ~0ull, memory_order_release);
	pthread_cond_broadcast(&p->cv_work);
	pthread_mutex_unlock(&p->mu);
	for (unsigned i = 0; i < p->n_thr; i++) {
		pthread_join(p->thr[i], NULL);
	}
	pthread_mutex_destroy(&p->mu);
	pthread_cond_destroy(&p->cv_work);
	pthread_cond_destroy(&p->cv_done);
	free(p->thr);
	free(p);
}

/* fn over [0, n) in chunks, on the pool's threads and the caller */
static void
pool_run(struct nxs_pool *p, pool_fn_t fn, void *arg, size_t n, size_t chunk)
{
	uint64_t gen;

	/* (one run at a time: a second caller -- there should be none -- works its items itself) */
	if (!p || p->n_thr == 0 || n <= chunk || n >= (1ull << POOL_GEN_SHIFT) ||
	    atomic_flag_test_and_set_explicit(&p->busy, memory_order_acquire)) {
		if (n) {
			fn(arg, 0, n);
		}
		return;
	}
	pthread_mutex_lock(&p->mu);
	p->fn = fn;
	p->arg = arg;
	p->n = n;
	p->chunk = chunk;
	gen = ++p->gen;
	atomic_store(&p->done, 0);
	atomic_store(&p->next, (gen & 0xffffff) << POOL_GEN_SHIFT);
	atomic_store_explicit(&p->gen_pub, gen, memory_order_release);
	pthread_cond_broadcast(&p->cv_work);
	pthread_mutex_unlock(&p->mu);
	pool_work(p, gen, fn, arg, n, chunk);
	/* every item has been handed out; the last ones are still being worked on by
	 * whoever drew them: a short spin, then sleep */
	for (int spin = 0; spin < 4000 && atomic_load(&p->done) < n; spin++) {
		__builtin_ia32_pause();
	}
	if (atomic_load(&p->done) < n) {
		pthread_mutex_lock(&p->mu);
		p->waiting = true;
		while (atomic_load(&p->done) < n) {
			pthread_cond_wait(&p->cv_done, &p->mu);
		}
		p->waiting = false;
		pthread_mutex_unlock(&p->mu);
	}
	atomic_flag_clear_explicit(&p->busy, memory_order_release);
}

#ifdef NXS_TEST_HOOKS	/* (nxs_hooks.h: test hooks and bench accessors are not part of the production ABI) */
/* tests: `rounds` runs of `n` items each on a pool of `n_thr` threads; every item of
 * every run must be worked on exactly once.  Returns the number of items that were not. */
static void
pool_test_fn(void *arg, size_t lo, size_t hi)
{
	_Atomic unsigned char *hits = arg;

	for (size_t i = lo; i < hi; i++) {
		atomic_fetch_add(&hits[i], 1);
	}
}

size_t
nxs_test_pool(unsigned n_thr, size_t n, unsigned rounds, size_t chunk)
{
	struct nxs_pool *p = pool_create(n_thr);
	_Atomic unsigned char *hits = calloc(n ? n : 1, 1);
	size_t bad = 0;

	for (unsigned r = 0; p && hits && r < rounds; r++) {
		memset((void *)hits, 0, n);
		pool_run(p, pool_test_fn, (void *)hits, n, chunk);
		for (size_t i = 0; i < n; i++) {
			bad += hits[i] != 1;
		}
	}
	pool_destroy(p);
	free((void *)hits);
	return (p && hits) ? bad : (size_t)-1;
}

#endif /* NXS_TEST_HOOKS */

/* the pool of an instance: NXS_HOST_THREADS (read once), else min(cores, 16) */
static struct nxs_pool *nxs_pool_get(nxs_t *nxs);

/* nxsgpu_parallel_t: the device layer's per-query host work on this nxs_t's pool */
static void
api_parallel(void *ctx, nxsgpu_body_t body, void *arg, size_t n, size_t chunk)
{
	pool_run(nxs_pool_get((nxs_t *)ctx), body, arg, n, chunk);
}

static struct nxs_pool *
nxs_pool_get(nxs_t *nxs)
{
	if (!nxs->pool_tried) {
		const char *e = getenv("NXS_HOST_THREADS");
		long n = e ? atol(e) : sysconf(_SC_NPROCESSORS_ONLN);

		nxs->pool_tried = true;
		if (n > 16 && !e) {
			n = 16;
		}
		if (n > 64) {
			n = 64;
		}
		if (n > 1) {
			nxs->pool = pool_create((unsigned)n - 1);	/* the caller works too */
		}
	}
	return nxs->pool;
}

/* ---- instance + errors --------------------------------------------------- */

nxs_t *
nxs_open(const char *basedir)
{
	nxs_t *nxs = calloc(1, sizeof(nxs_t));
	const char *s = basedir ? basedir : getenv("NXS_BASEDIR");	/* nxs.c:109 */

	if (!nxs) {
		return NULL;
	}
	if (s == NULL || (nxs->basedir = realpath(s, NULL)) == NULL) {
		free(nxs);
		return NULL;
	}
	return nxs;
}

void
nxs_close(nxs_t *nxs)
{
	while (nxs->n_indexes) {
		nxs_index_close(nxs->indexes[nxs->n_indexes - 1]);
	}
	pool_destroy(nxs->pool);
	free(nxs->indexes);
	free(nxs->basedir);
	free(nxs->errmsg);
	free(nxs);
}

void
nxs_clear_error(nxs_t *nxs)
{
	free(nxs->errmsg);
	nxs->errmsg = NULL;
	nxs->errcode = NXS_ERR_SUCCESS;
}

void
nxs_decl_err(nxs_t *nxs, nxs_err_t code, const char *fmt, ...)
{
	char *msg = NULL;
	va_list ap;

	va_start(ap, fmt);
	if (vasprintf(&msg, fmt, ap) == -1) {
		msg = NULL;
	}
	va_end(ap);
	free(nxs->errmsg);
	nxs->errmsg = msg;
	nxs->errcode = code;
}

nxs_err_t
nxs_get_error(const nxs_t *nxs, const char **errmsg)
{
	if (errmsg) {
		*errmsg = nxs->errmsg;
	}
	return nxs->errcode;
}

/* ---- params ---------------------------------------------------------------- */

nxs_params_t *
nxs_params_create(void)
{
	return calloc(1, sizeof(nxs_params_t));
}

void
nxs_params_release(nxs_params_t *p)
{
	for (size_t i = 0; i < p->n; i++) {
		free(p->kv[i].key);
		free(p->kv[i].s);
	}
	free(p->kv);
	free(p);
}

static param_kv_t *
params_slot(nxs_params_t *p, const char *key)
{
	param_kv_t *kv;

	for (size_t i = 0; i < p->n; i++) {
		if (strcmp(p->kv[i].key, key) == 0) {
			free(p->kv[i].s);
			p->kv[i].s = NULL;
			return &p->kv[i];
		}
	}
	if ((kv = realloc(p->kv, (p->n + 1) * sizeof(param_kv_t))) == NULL) {
		return NULL;
	}
	p->kv = kv;
	kv = &p->kv[p->n++];
	memset(kv, 0, sizeof(*kv));
	kv->key = strdup(key);
	return kv;
}

int
nxs_params_set_str(nxs_params_t *p, const char *key, const char *val)
{
	param_kv_t *kv = params_slot(p, key);
	if (!kv) return -1;
	kv->type = PV_STR;
	kv->s = strdup(val);
	return 0;
}

int
nxs_params_set_uint(nxs_params_t *p, const char *key, uint64_t val)
{
	param_kv_t *kv = params_slot(p, key);
	if (!kv) return -1;
	kv->type = PV_UINT;
	kv->u = val;
	return 0;
}

int
nxs_params_set_bool(nxs_params_t *p, const char *key, bool val)
{
	param_kv_t *kv = params_slot(p, key);
	if (!kv) return -1;
	kv->type = PV_BOOL;
	kv->b = val;
	return 0;
}

static const param_kv_t *
params_find(const nxs_params_t *p, const char *key, pv_type_t type)
{
	for (size_t i = 0; p && i < p->n; i++) {
		if (strcmp(p->kv[i].key, key) == 0 && p->kv[i].type == type) {
			return &p->kv[i];
		}
	}
	return NULL;
}

const char *
nxs_params_get_str(const nxs_params_t *p, const char *key)
{
	const param_kv_t *kv = params_find(p, key, PV_STR);
	return kv ? kv->s : NULL;
}

int
nxs_params_get_uint(const nxs_params_t *p, const char *key, uint64_t *val)
{
	const param_kv_t *kv = params_find(p, key, PV_UINT);
	if (!kv) return -1;
	*val = kv->u;
	return 0;
}

int
nxs_params_get_bool(const nxs_params_t *p, const char *key, bool *val)
{
	const param_kv_t *kv = params_find(p, key, PV_BOOL);
	if (!kv) return -1;
	*val = kv->b;
	return 0;
}

/*
 * nxs_params_fromjson (params.c:201-208): how the reference's Lua / HTTP tier
 * hands `limit`, `algo` and `fuzzymatch` to nxs_index_search (lua.c:99-110).  The
 * reference parses with yyjson into a mutable document and the getters look a key
 * up in the ROOT OBJECT by type; here: a strict JSON scanner (RFC 8259, no
 * trailing content -- yyjson's default flags) that keeps the root object's
 * string / unsigned-integer / bool members and validates and skips everything
 * else (negative or fractional numbers, null, arrays, nested objects: no getter of
 * the query path reads those).  A syntax error is NXS_ERR_SYSTEM "params parsing
 * failed: ... at <offset>", as there.
 */
typedef struct {
	const char *	p;
	const char *	end;
	const char *	beg;
	const char *	err;
} jscan_t;

static void
js_ws(jscan_t *j)
{
	while (j->p < j->end && (*j->p == ' ' || *j->p == '\t' || *j->p == '\n' || *j->p == '\r')) {
		j->p++;
	}
}

static int
js_fail(jscan_t *j, const char *msg)
{
	if (!j->err) {
		j->err = msg;
	}
	return -1;
}

static int
js_hex4(jscan_t *j, unsigned *out)
{
	unsigned v = 0;

	if (j->end - j->p < 4) {
		return js_fail(j, "invalid escaped sequence in string");
	}
	for (int i = 0; i < 4; i++) {
		const char c = *j->p++;
		v <<= 4;
		if (c >= '0' && c <= '9') v |= (unsigned)(c - '0');
		else if (c >= 'a' && c <= 'f') v |= (unsigned)(c - 'a' + 10);
		else if (c >= 'A' && c <= 'F') v |= (unsigned)(c - 'A' + 10);
		else return js_fail(j, "invalid escaped sequence in string");
	}
	*out = v;
	return 0;
}

/* a string; *out (if wanted) = malloc'ed, unescaped, NUL-terminated copy */
static int
js_string(jscan_t *j, char **out)
{
	char *buf = NULL;
	size_t n = 0;

	if (j->p >= j->end || *j->p != '"') {
		return js_fail(j, "unexpected character");
	}
	j->p++;
	if (out && (buf = malloc((size_t)(j->end - j->p) + 1)) == NULL) {
		return js_fail(j, "out of memory");
	}
	while (j->p < j->end && *j->p != '"') {
		unsigned char c = (unsigned char)*j->p++;

		if (c < 0x20) {
			free(buf);
			return js_fail(j, "unexpected control character in string");
		}
		if (c == '\\') {
			unsigned cp;

			if (j->p >= j->end) {
				break;
			}
			c = (unsigned char)*j->p++;
			switch (c) {
			case '"': case '\\': case '/': cp = c; break;
			case 'b': cp = '\b'; break;
			case 'f': cp = '\f'; break;
			case 'n': cp = '\n'; break;
			case 'r': cp = '\r'; break;
			case 't': cp = '\t'; break;
			case 'u':
				if (js_hex4(j, &cp) == -1) {
					free(buf);
					return -1;
				}
				if (cp >= 0xd800 && cp <= 0xdbff) {	/* surrogate pair */
					unsigned lo;
					if (j->end - j->p < 6 || j->p[0] != '\\' || j->p[1] != 'u') {
						free(buf);
						return js_fail(j, "no low surrogate in string");
					}
					j->p += 2;
					if (js_hex4(j, &lo) == -1 || lo < 0xdc00 || lo > 0xdfff) {
						free(buf);
						return js_fail(j, "invalid low surrogate in string");
					}
					cp = 0x10000 + ((cp - 0xd800) << 10) + (lo - 0xdc00);
				} else if (cp >= 0xdc00 && cp <= 0xdfff) {
					free(buf);
					return js_fail(j, "invalid high surrogate in string");
				}
				break;
			default:
				free(buf);
				return js_fail(j, "invalid escaped character in string");
			}
			if (buf) {
				if (cp < 0x80) {
					buf[n++] = (char)cp;
				} else if (cp < 0x800) {
					buf[n++] = (char)(0xc0 | (cp >> 6));
					buf[n++] = (char)(0x80 | (cp & 0x3f));
				} else if (cp < 0x10000) {
					buf[n++] = (char)(0xe0 | (cp >> 12));
					buf[n++] = (char)(0x80 | ((cp >> 6) & 0x3f));
					buf[n++] = (char)(0x80 | (cp & 0x3f));
				} else {
					buf[n++] = (char)(0xf0 | (cp >> 18));
					buf[n++] = (char)(0x80 | ((cp >> 12) & 0x3f));
					buf[n++] = (char)(0x80 | ((cp >> 6) & 0x3f));
					buf[n++] = (char)(0x80 | (cp & 0x3f));
				}
			}
			continue;
		}
		if (buf) {
			buf[n++] = (char)c;
		}
	}
	if (j->p >= j->end) {
		free(buf);
		return js_fail(j, "unclosed string");
	}
	j->p++;		/* the closing quote */
	if (buf) {
		buf[n] = '\0';
		*out = buf;
	}
	return 0;
}

/* a number; *is_uint: a non-negative integer without fraction / exponent that fits u64 */
static int
js_number(jscan_t *j, bool *is_uint, uint64_t *u)
{
	const char *s = j->p;
	bool neg = false, integral = true, fits = true;
	uint64_t v = 0;

	if (j->p < j->end && *j->p == '-') {
		neg = true;
		j->p++;
	}
	if (j->p >= j->end || *j->p < '0' || *j->p > '9') {
		j->p = s;
		return js_fail(j, "unexpected character");
	}
	if (*j->p == '0') {
		j->p++;
		if (j->p < j->end && *j->p >= '0' && *j->p <= '9') {
			return js_fail(j, "number with leading zero is not allowed");
		}
	} else {
		while (j->p < j->end && *j->p >= '0' && *j->p <= '9') {
			const unsigned d = (unsigned)(*j->p++ - '0');
			if (v > (UINT64_MAX - d) / 10) {
				fits = false;
			} else {
				v = v * 10 + d;
			}
		}
	}
	if (j->p < j->end && *j->p == '.') {
		integral = false;
		j->p++;
		if (j->p >= j->end || *j->p < '0' || *j->p > '9') {
			return js_fail(j, "no digit after decimal point");
		}
		while (j->p < j->end && *j->p >= '0' && *j->p <= '9') {
			j->p++;
		}
	}
	if (j->p < j->end && (*j->p == 'e' || *j->p == 'E')) {
		integral = false;
		j->p++;
		if (j->p < j->end && (*j->p == '+' || *j->p == '-')) {
			j->p++;
		}
		if (j->p >= j->end || *j->p < '0' || *j->p > '9') {
			return js_fail(j, "no digit after exponent sign");
		}
		while (j->p < j->end && *j->p >= '0' && *j->p <= '9') {
			j->p++;
		}
	}
	*is_uint = !neg && integral && fits;
	*u = v;
	return 0;
}

static int js_value(jscan_t *j, nxs_params_t *into, const char *key, unsigned depth);

static int
js_literal(jscan_t *j, const char *word)
{
	const size_t n = strlen(word);

	if ((size_t)(j->end - j->p) < n || memcmp(j->p, word, n) != 0) {
		return js_fail(j, "invalid literal");
	}
	j->p += n;
	return 0;
}

/* one value; if `into` (the root object's member `key`), keep what the getters read */
static int
js_value(jscan_t *j, nxs_params_t *into, const char *key, unsigned depth)
{
	js_ws(j);
	if (j->p >= j->end) {
		return js_fail(j, "unexpected end of data");
	}
	if (depth > 512) {
		return js_fail(j, "nesting too deep");
	}
	switch (*j->p) {
	case '"': {
		char *str = NULL;

		if (js_string(j, into ? &str : NULL) == -1) {
			return -1;
		}
		if (into) {
			const int r = nxs_params_set_str(into, key, str);
			free(str);
			return r == 0 ? 0 : js_fail(j, "out of memory");
		}
		return 0;
	}
	case 't':
		if (js_literal(j, "true") == -1) return -1;
		return into && nxs_params_set_bool(into, key, true) != 0 ? js_fail(j, "out of memory") : 0;
	case 'f':
		if (js_literal(j, "false") == -1) return -1;
		return into && nxs_params_set_bool(into, key, false) != 0 ? js_fail(j, "out of memory") : 0;
	case 'n':
		return js_literal(j, "null");
	case '[':
		j->p++;
		js_ws(j);
		if (j->p < j->end && *j->p == ']') {
			j->p++;
			return 0;
		}
		for (;;) {
			if (js_value(j, NULL, NULL, depth + 1) == -1) {
				return -1;
			}
			js_ws(j);
			if (j->p < j->end && *j->p == ',') {
				j->p++;
				continue;
			}
			if (j->p < j->end && *j->p == ']') {
				j->p++;
				return 0;
			}
			return js_fail(j, j->p < j->end ? "unexpected character" : "unclosed array");
		}
	case '{': {
		/* only the ROOT object's members are parameters */
		nxs_params_t *const members = (depth == 0) ? into : NULL;

		j->p++;
		js_ws(j);
		if (j->p < j->end && *j->p == '}') {
			j->p++;
			return 0;
		}
		for (;;) {
			char *k = NULL;
			int r;

			js_ws(j);
			if (js_string(j, &k) == -1) {
				return -1;
			}
			js_ws(j);
			if (j->p >= j->end || *j->p != ':') {
				free(k);
				return js_fail(j, "unexpected character");
			}
			j->p++;
			/* (an embedded NUL would truncate the key: such a key is no parameter) */
			/*
			 * The reference's getters look a key up with yyjson_mut_obj_get(): the FIRST
			 * member of that name, whatever its kind -- a later duplicate is never seen,
			 * and a first member of a kind the getter does not read (null, a negative
			 * number, an array ...) hides a later usable one.  So only the first
			 * occurrence is kept, as a typeless entry if need be.
			 */
			{
				nxs_params_t *dst = members;
				bool first = false;

				if (members && k) {
					first = true;
					for (size_t i = 0; i < members->n; i++) {
						if (strcmp(members->kv[i].key, k) == 0) {
							first = false;
							break;
						}
					}
					if (!first) {
						dst = NULL;
					}
				}
				const size_t n_before = members ? members->n : 0;
				r = js_value(j, dst, k, depth + 1);
				if (r == 0 && first && members->n == n_before) {
					param_kv_t *kv = params_slot(members, k);
					if (!kv) {
						r = js_fail(j, "out of memory");
					} else {
						kv->type = PV_NONE;
					}
				}
			}
			free(k);
			if (r == -1) {
				return -1;
			}
			js_ws(j);
			if (j->p < j->end && *j->p == ',') {
				j->p++;
				continue;
			}
			if (j->p < j->end && *j->p == '}') {
				j->p++;
				return 0;
			}
			return js_fail(j, j->p < j->end ? "unexpected character" : "unclosed object");
		}
	}
	default: {
		bool is_uint = false;
		uint64_t u = 0;

		if (js_number(j, &is_uint, &u) == -1) {
			return -1;
		}
		if (into && is_uint && nxs_params_set_uint(into, key, u) != 0) {
			return js_fail(j, "out of memory");
		}
		return 0;
	}
	}
}

nxs_params_t *
nxs_params_fromjson(nxs_t *nxs, const char *json, size_t len)
{
	nxs_params_t *params;
	jscan_t j = { json, json + len, json, NULL };

	if ((params = nxs_params_create()) == NULL) {
		return NULL;
	}
	/* the root: its members if it is an object, nothing otherwise (the getters
	 * look keys up in the root object) */
	js_ws(&j);
	if (j.p < j.end && *j.p == '{') {
		if (js_value(&j, params, NULL, 0) == -1) {
			goto fail;
		}
	} else if (js_value(&j, NULL, NULL, 1) == -1) {
		goto fail;
	}
	js_ws(&j);
	if (j.p < j.end) {
		js_fail(&j, "unexpected content after document");
		goto fail;
	}
	return params;
fail:
	nxs_params_release(params);
	nxs_decl_err(nxs, NXS_ERR_SYSTEM, "params parsing failed: %s at %u",
	    j.err ? j.err : "invalid JSON", (unsigned)(j.p - j.beg));
	return NULL;
}

/* ranking.c:182-192 */
static int
get_ranking_func_id(const char *name)
{
	if (strcasecmp(name, "TF-IDF") == 0) {
		return NXSGPU_TF_IDF;
	}
	if (strcasecmp(name, "BM25") == 0) {
		return NXSGPU_BM25;
	}
	return -1;
}

/* ---- index open / close ---------------------------------------------------- */

/* str_isalnumdu(): index names are [A-Za-z0-9_-]+ (nxs.c:236-240) */
static bool
name_ok(const char *s)
{
	if (!*s) {
		return false;
	}
	for (; *s; s++) {
		const char c = *s;
		if (!((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') ||
		    (c >= '0' && c <= '9') || c == '-' || c == '_')) {
			return false;
		}
	}
	return true;
}

static nxs_index_t *
index_open_common(nxs_t *nxs, const char *name, const char *terms_path,
    const char *dtmap_path, int algo, const char *const *filters, size_t n_filters,
    const char *lang)
{
	nxs_index_t *idx = calloc(1, sizeof(nxs_index_t)), **list;
	const char *ferr = NULL;

	if (!idx) {
		nxs_decl_err(nxs, NXS_ERR_SYSTEM, "out of memory");
		return NULL;
	}
	idx->nxs = nxs;
	idx->algo = algo;
	idx->name = strdup(name);
	/* filter_pipeline_create (nxs.c:412-419): the query side of it */
	if (n_filters) {
		idx->filters = nxs_filters_create(nxs->basedir, filters, n_filters, lang, &ferr);
		if (!idx->filters) {
			nxs_decl_err(nxs, NXS_ERR_INVALID, "%s", ferr ? ferr : "filter pipeline failed");
			free(idx->name);
			free(idx);
			return NULL;
		}
		for (size_t i = 0; i < n_filters; i++) {
			idx->lowercase = idx->lowercase || strcmp(filters[i], "normalizer") == 0;
		}
	}
	if (nxs_index_load(idx, terms_path, dtmap_path) == -1) {
		nxs_index_unload(idx);
		nxs_filters_destroy(idx->filters);
		free(idx->name);
		free(idx);
		return NULL;
	}
	list = realloc(nxs->indexes, (nxs->n_indexes + 1) * sizeof(void *));
	nxs->indexes = list;
	nxs->indexes[nxs->n_indexes++] = idx;
	return idx;
}

/* minimal reader for the JSON params.db the reference writes (nxs.c:282-288) */
static char *
json_get_str(const char *json, const char *key)
{
	char pat[64];
	const char *p, *e;

	snprintf(pat, sizeof(pat), "\"%s\"", key);
	if ((p = strstr(json, pat)) == NULL) {
		return NULL;
	}
	p += strlen(pat);
	while (*p == ' ' || *p == ':' || *p == '\t' || *p == '\n') p++;
	if (*p != '"') {
		return NULL;
	}
	p++;
	if ((e = strchr(p, '"')) == NULL) {
		return NULL;
	}
	return strndup(p, e - p);
}

/* the strings of a JSON array of strings: "key": ["a", "b"] -> count */
static size_t
json_get_strlist(const char *json, const char *key, char **out, size_t cap)
{
	char pat[64];
	const char *p, *end;
	size_t n = 0;

	snprintf(pat, sizeof(pat), "\"%s\"", key);
	if ((p = strstr(json, pat)) == NULL) {
		return 0;
	}
	p += strlen(pat);
	while (*p == ' ' || *p == ':' || *p == '\t' || *p == '\n') p++;
	if (*p != '[' || (end = strchr(p, ']')) == NULL) {
		return 0;
	}
	while (n < cap) {
		const char *q = memchr(p, '"', (size_t)(end - p)), *e;
		if (!q || (e = memchr(q + 1, '"', (size_t)(end - q - 1))) == NULL) {
			break;
		}
		out[n++] = strndup(q + 1, (size_t)(e - q - 1));
		p = e + 1;
	}
	return n;
}

nxs_index_t *
nxs_index_open(nxs_t *nxs, const char *name)
{
	char *ppath = NULL, *tpath = NULL, *dpath = NULL, *json = NULL, *algo_name = NULL;
	char *filters[8] = { NULL }, *lang = NULL;
	size_t n_filters = 0;
	nxs_index_t *idx = NULL;
	struct stat sb;
	FILE *fp;
	int algo;

	nxs_clear_error(nxs);
	if (!name_ok(name)) {
		nxs_decl_err(nxs, NXS_ERR_INVALID, "invalid characters in index name");
		return NULL;
	}
	for (size_t i = 0; i < nxs->n_indexes; i++) {
		if (strcmp(nxs->indexes[i]->name, name) == 0) {
			nxs_decl_err(nxs, NXS_ERR_EXISTS, "index `%s' is already open", name);
			return NULL;
		}
	}
	if (asprintf(&ppath, "%s/data/%s/params.db", nxs->basedir, name) == -1 ||
	    asprintf(&tpath, "%s/data/%s/nxsterms", nxs->basedir, name) == -1 ||
	    asprintf(&dpath, "%s/data/%s/nxsdtmap", nxs->basedir, name) == -1) {
		goto out;
	}
	if (stat(ppath, &sb) == -1 && errno == ENOENT) {
		nxs_decl_err(nxs, NXS_ERR_MISSING, "index `%s' does not exist", name);
		goto out;
	}
	if ((fp = fopen(ppath, "r")) == NULL) {
		nxs_decl_err(nxs, NXS_ERR_SYSTEM, "could not open %s: %s", ppath, strerror(errno));
		goto out;
	}
	json = calloc(1, sb.st_size + 1);
	if (fread(json, 1, sb.st_size, fp) != (size_t)sb.st_size) {
		fclose(fp);
		nxs_decl_err(nxs, NXS_ERR_FATAL, "corrupted index params");
		goto out;
	}
	fclose(fp);
	if ((algo_name = json_get_str(json, "algo")) == NULL) {
		nxs_decl_err(nxs, NXS_ERR_FATAL, "corrupted index params");	/* nxs.c:405-409 */
		goto out;
	}
	algo = get_ranking_func_id(algo_name);
	if (algo < 0) {
		nxs_decl_err(nxs, NXS_ERR_FATAL, "corrupted index params");
		goto out;
	}
	/* "filters": [...] in list order, "lang" (nxs.c:263-266; params.db) */
	n_filters = json_get_strlist(json, "filters", filters, 8);
	lang = json_get_str(json, "lang");
	idx = index_open_common(nxs, name, tpath, dpath, algo,
	    (const char *const *)filters, n_filters, lang);
out:
	for (size_t i = 0; i < n_filters; i++) {
		free(filters[i]);
	}
	free(lang);
	free(ppath);
	free(tpath);
	free(dpath);
	free(json);
	free(algo_name);
	return idx;
}

nxs_index_t *
nxs_index_open_files(nxs_t *nxs, const char *terms_path, const char *dtmap_path,
    const char *algo_name, bool lowercase)
{
	const int algo = get_ranking_func_id(algo_name ? algo_name : "BM25");

	nxs_clear_error(nxs);
	if (algo < 0) {
		nxs_decl_err(nxs, NXS_ERR_INVALID, "invalid algorithm");
		return NULL;
	}
	{
		static const char *const norm_only[] = { "normalizer" };
		return index_open_common(nxs, terms_path, terms_path, dtmap_path, algo,
		    norm_only, lowercase ? 1 : 0, "en");
	}
}

static void index_drain(nxs_index_t *);

/* N4: one shard of a doc-sharded collection (include/nxs.h) */
nxs_index_t *
nxs_index_open_shard(nxs_t *nxs, const char *terms_path, const char *dtmap_path,
    const char *algo_name, bool lowercase, unsigned shard, unsigned n_shards, int device)
{
	static const char *const norm_only[] = { "normalizer" };
	const int algo = get_ranking_func_id(algo_name ? algo_name : "BM25");
	nxs_index_t *idx, **list;
	const char *ferr = NULL;

	nxs_clear_error(nxs);
	if (algo < 0 || n_shards == 0 || shard >= n_shards) {
		nxs_decl_err(nxs, NXS_ERR_INVALID, algo < 0 ? "invalid algorithm" : "invalid shard");
		return NULL;
	}
	if ((idx = calloc(1, sizeof(nxs_index_t))) == NULL) {
		nxs_decl_err(nxs, NXS_ERR_SYSTEM, "out of memory");
		return NULL;
	}
	idx->nxs = nxs;
	idx->algo = algo;
	idx->lowercase = lowercase;
	idx->shard = shard;
	idx->n_shards = n_shards;
	idx->want_device = device >= 0 ? device + 1 : 0;
	idx->name = strdup(terms_path);
	if (lowercase && (idx->filters = nxs_filters_create(nxs->basedir, norm_only, 1, "en", &ferr)) == NULL) {
		nxs_decl_err(nxs, NXS_ERR_INVALID, "%s", ferr ? ferr : "filter pipeline failed");
		free(idx->name);
		free(idx);
		return NULL;
	}
	if (nxs_index_load(idx, terms_path, dtmap_path) == -1) {
		nxs_index_unload(idx);
		nxs_filters_destroy(idx->filters);
		free(idx->name);
		free(idx);
		return NULL;
	}
	list = realloc(nxs->indexes, (nxs->n_indexes + 1) * sizeof(void *));
	nxs->indexes = list;
	nxs->indexes[nxs->n_indexes++] = idx;
	return idx;
}

void
nxs_index_close(nxs_index_t *idx)
{
	nxs_t *nxs = idx->nxs;

	for (size_t i = 0; i < nxs->n_indexes; i++) {
		if (nxs->indexes[i] == idx) {
			nxs->indexes[i] = nxs->indexes[--nxs->n_indexes];
			break;
		}
	}
	index_drain(idx);
	if (idx->comm) {
		if (idx->dev) {
			(void)nxsgpu_index_set_comm(idx->dev, NULL);
		}
		nxsgpu_comm_destroy(idx->comm);
		idx->comm = NULL;
	}
	free(idx->emu_block);
	nxs_filters_destroy(idx->filters);
	nxs_index_unload(idx);
	plan_cache_destroy(idx->pcache);
	free(idx->name);
	free(idx);
}

#ifdef NXS_TEST_HOOKS
struct nxsgpu_index *
nxs_index_device(nxs_index_t *idx)
{
	return idx->dev;
}

/* bench: plan cache on / off at run time (the environment decides it otherwise, once) */
void
nxs_index_set_plan_cache(nxs_index_t *idx, int on)
{
	extern void nxs_plan_cache_switch(nxs_index_t *, int);
	nxs_plan_cache_switch(idx, on);
}
#endif

/* ---- response object --------------------------------------------------------- */

/*
 * The responses of one batch live in ONE allocation (the reference's
 * nxs_resp_create mallocs a map, a heap and a JSON document per query,
 * results.c:46-85): header + n response structs + all (id, score) pairs.  Each
 * nxs_resp_t stays individually releasable; the slab goes with the last one.
 */
struct resp_slab {
	size_t		refs;
};

struct nxs_resp {
	nxs_doc_id_t *	ids;
	float *		scores;
	unsigned	count;
	unsigned	iter;
	struct resp_slab *slab;		/* NULL: ids/scores are this response's own */
};

typedef struct {
	struct resp_slab *slab;
	nxs_resp_t *	resps;		/* [n] */
	nxs_doc_id_t *	ids;		/* [total] */
	float *		scores;		/* [total] */
	size_t		used;
} slab_builder_t;

static int
slab_begin(slab_builder_t *b, size_t n, size_t total)
{
	const size_t hdr = (sizeof(struct resp_slab) + 15) & ~(size_t)15;
	const size_t rs = (n * sizeof(nxs_resp_t) + 15) & ~(size_t)15;
	uint8_t *m = malloc(hdr + rs + total * sizeof(nxs_doc_id_t) + total * sizeof(float) + 16);

	if (!m) {
		return -1;
	}
	b->slab = (struct resp_slab *)m;
	b->slab->refs = 0;
	b->resps = (nxs_resp_t *)(m + hdr);
	b->ids = (nxs_doc_id_t *)(m + hdr + rs);
	b->scores = (float *)(b->ids + total);
	b->used = 0;
	return 0;
}

/* response i of the slab: `count` results to be filled in by the caller */
static nxs_resp_t *
slab_resp(slab_builder_t *b, size_t i, unsigned count)
{
	nxs_resp_t *r = &b->resps[i];

	r->ids = b->ids + b->used;
	r->scores = b->scores + b->used;
	r->count = count;
	r->iter = 0;
	r->slab = b->slab;
	b->used += count;
	b->slab->refs++;
	return r;
}

void
nxs_resp_release(nxs_resp_t *r)
{
	if (r->slab) {
		if (--r->slab->refs == 0) {
			free(r->slab);
		}
		return;
	}
	free(r->ids);
	free(r->scores);
	free(r);
}

void
nxs_resp_iter_reset(nxs_resp_t *r)
{
	r->iter = 0;
}

bool
nxs_resp_iter_result(nxs_resp_t *r, nxs_doc_id_t *doc_id, float *score)
{
	if (r->iter >= r->count) {
		return false;
	}
	*doc_id = r->ids[r->iter];
	*score = r->scores[r->iter];	/* float -> JSON double -> float is exact */
	r->iter++;
	return true;
}

unsigned
nxs_resp_resultcount(const nxs_resp_t *r)
{
	return r->count;
}

/*
 * JSON real: shortest decimal that round-trips (double)score, always with a
 * fraction digit -- what yyjson's writer produces for results.c:158.  Pinned
 * by the reference only for 3.0 and 1.5 (t_misc.c:115-117).
 */
static size_t
fmt_real(char *out, double v)
{
	char e[40], digs[24];
	int nd = 0, x, prec;
	size_t o = 0;
	const char *p, *ep;

	for (prec = 1; prec <= 17; prec++) {
		snprintf(e, sizeof(e), "%.*e", prec - 1, v);
		if (strtod(e, NULL) == v) {
			break;
		}
	}
	p = e;
	if (*p == '-') {
		out[o++] = '-';
		p++;
	}
	ep = strchr(p, 'e');
	for (; p < ep; p++) {
		if (*p != '.') {
			digs[nd++] = *p;
		}
	}
	while (nd > 1 && digs[nd - 1] == '0') {
		nd--;
	}
	x = atoi(ep + 1);
	if (x >= -6 && x < 21) {
		if (x < 0) {
			out[o++] = '0';
			out[o++] = '.';
			for (int i = 0; i < -x - 1; i++) out[o++] = '0';
			for (int i = 0; i < nd; i++) out[o++] = digs[i];
		} else {
			for (int i = 0; i <= x; i++) out[o++] = i < nd ? digs[i] : '0';
			out[o++] = '.';
			if (nd > x + 1) {
				for (int i = x + 1; i < nd; i++) out[o++] = digs[i];
			} else {
				out[o++] = '0';
			}
		}
	} else {
		out[o++] = digs[0];
		if (nd > 1) {
			out[o++] = '.';
			for (int i = 1; i < nd; i++) out[o++] = digs[i];
		}
		o += sprintf(out + o, "e%d", x);
	}
	out[o] = '\0';
	return o;
}

/* {"results":[{"doc_id":N,"score":X},...],"count":K}  (results.c:80-82,153-161,218) */
char *
nxs_resp_tojson(nxs_resp_t *r, size_t *len)
{
	const size_t cap = 48 + (size_t)r->count * 88;
	char *s = malloc(cap);
	size_t o = 0;

	if (!s) {
		return NULL;
	}
	o += sprintf(s + o, "{\"results\":[");
	for (unsigned i = 0; i < r->count; i++) {
		o += sprintf(s + o, "%s{\"doc_id\":%llu,\"score\":", i ? "," : "",
		    (unsigned long long)r->ids[i]);
		o += fmt_real(s + o, (double)r->scores[i]);
		s[o++] = '}';
	}
	o += sprintf(s + o, "],\"count\":%u}", r->count);
	if (len) {
		*len = o;
	}
	return s;
}

/* ---- search -------------------------------------------------------------------- */

typedef struct {
	uint64_t	limit;
	int		algo;
	bool		fuzzymatch;
} search_params_t;

/* get_search_params: search.c:78-112 */
static int
get_search_params(nxs_index_t *idx, nxs_params_t *params, search_params_t *sp)
{
	const char *s;
	bool fl;

	sp->limit = NXS_DEFAULT_RESULTS_LIMIT;
	sp->fuzzymatch = true;
	sp->algo = idx->algo;
	if (!params) {
		return 0;
	}
	if (nxs_params_get_uint(params, "limit", &sp->limit) == 0 &&
	    (sp->limit == 0 || sp->limit > UINT_MAX)) {
		nxs_decl_err(idx->nxs, NXS_ERR_INVALID, "invalid limit");
		return -1;
	}
	if ((s = nxs_params_get_str(params, "algo")) != NULL &&
	    (sp->algo = get_ranking_func_id(s)) < 0) {
		nxs_decl_err(idx->nxs, NXS_ERR_INVALID, "invalid algorithm");
		return -1;
	}
	if (nxs_params_get_bool(params, "fuzzymatch", &fl) == 0 && !fl) {
		sp->fuzzymatch = false;
	}
	return 0;
}

/*
 * Plan cache.  The reference builds a query_t per call (construct_query, search.c:176-208);
 * what that yields for a given query string -- tokens, their term ids, the boolean program --
 * depends only on the string, the `fuzzymatch` flag and the index's dictionary, so the
 * compiled plan of a string is kept until the index changes (any refresh clears the cache:
 * new terms change lookups and fuzzy winners).  A server's head queries then cost a hash
 * lookup and a 424-byte copy instead of lex + parse + resolve + compile (C2: planning was
 * 60 % of a 1024-query step).  Lookups run on the worker threads (read-only); the batch's
 * misses are inserted by the caller's thread afterwards.  Only plans that fit
 * nxsgpu_query_t and queries without errors are kept.  NXS_PLAN_CACHE=0 turns it off.
 */
typedef struct {
	uint64_t	h;
	char *		key;		/* NULL = empty slot */
	uint32_t	klen;
	uint8_t		fuzzy, empty;
	nxsgpu_query_t	plan;
} pc_ent_t;

struct plan_cache {
	pc_ent_t *	e;
	size_t		cap, n;		/* cap: a power of two */
	uint64_t	gen;		/* refreshes of the index when the entries were made */
	bool		off;
};

#define	PLAN_CACHE_CAP	(1u << 15)

static uint64_t
pc_hash(const char *s, size_t n, bool fuzzy)
{
	uint64_t h = 1469598103934665603ull ^ (fuzzy ? 0x9e3779b97f4a7c15ull : 0);

	for (size_t i = 0; i < n; i++) {
		h = (h ^ (uint8_t)s[i]) * 1099511628211ull;
	}
	return h ? h : 1;
}

static void
plan_cache_clear(struct plan_cache *pc)
{
	for (size_t i = 0; pc && pc->e && i < pc->cap; i++) {
		free(pc->e[i].key);
		pc->e[i].key = NULL;
	}
	if (pc) {
		pc->n = 0;
	}
}

static void
plan_cache_destroy(struct plan_cache *pc)
{
	plan_cache_clear(pc);
	if (pc) {
		free(pc->e);
		free(pc);
	}
}

/* the cache of the index, valid for its current snapshot (NULL: off / out of memory) */
static struct plan_cache *
plan_cache_get(nxs_index_t *idx)
{
	/* (the dictionary can move without either refresh counter moving -- a refresh whose device half fails
	 * after sync_terms has consumed new terms --: what a lookup yields depends on the terms consumed) */
	const uint64_t gen = idx->n_incremental + idx->n_rebuilds + ((uint64_t)idx->last_id << 20);
	struct plan_cache *pc = idx->pcache;

	if (!pc) {
		const char *e = getenv("NXS_PLAN_CACHE");

		if ((pc = calloc(1, sizeof(*pc))) == NULL) {
			return NULL;
		}
		pc->off = e && atoi(e) == 0;
		pc->cap = PLAN_CACHE_CAP;
		if (!pc->off && (pc->e = calloc(pc->cap, sizeof(pc_ent_t))) == NULL) {
			pc->off = true;
		}
		pc->gen = gen;
		idx->pcache = pc;
	}
	if (pc->off) {
		return NULL;
	}
	if (pc->gen != gen || pc->n >= pc->cap / 2) {
		/* (half full: start over -- plan_cache_put refuses inserts from there on, so without this the table
		 * would stay frozen at its first 16 384 strings) */
		plan_cache_clear(pc);
		pc->gen = gen;
	}
	return pc;
}

void
nxs_plan_cache_switch(nxs_index_t *idx, int on)
{
	(void)plan_cache_get(idx);
	if (idx->pcache) {
		plan_cache_clear(idx->pcache);
		idx->pcache->off = !on;
		if (on && !idx->pcache->e && (idx->pcache->e = calloc(idx->pcache->cap, sizeof(pc_ent_t))) == NULL) {
			idx->pcache->off = true;
		}
	}
}

static const pc_ent_t *
plan_cache_find(const struct plan_cache *pc, const char *q, size_t n, bool fuzzy)
{
	const uint64_t h = pc_hash(q, n, fuzzy);

	for (size_t i = h & (pc->cap - 1); pc->e[i].key; i = (i + 1) & (pc->cap - 1)) {
		const pc_ent_t *e = &pc->e[i];
		if (e->h == h && e->klen == n && e->fuzzy == (uint8_t)fuzzy && memcmp(e->key, q, n) == 0) {
			return e;
		}
	}
	return NULL;
}

static void
plan_cache_put(struct plan_cache *pc, const char *q, size_t n, bool fuzzy, const qprep_t *p)
{
	const uint64_t h = pc_hash(q, n, fuzzy);
	size_t i = h & (pc->cap - 1);

	if (pc->n >= pc->cap / 2 || n > 4096) {
		return;
	}
	for (; pc->e[i].key; i = (i + 1) & (pc->cap - 1)) {
		if (pc->e[i].h == h && pc->e[i].klen == n && pc->e[i].fuzzy == (uint8_t)fuzzy &&
		    memcmp(pc->e[i].key, q, n) == 0) {
			return;		/* (twice in one batch) */
		}
	}
	if ((pc->e[i].key = malloc(n + 1)) == NULL) {
		return;
	}
	memcpy(pc->e[i].key, q, n);
	pc->e[i].key[n] = 0;
	pc->e[i].h = h;
	pc->e[i].klen = (uint32_t)n;
	pc->e[i].fuzzy = (uint8_t)fuzzy;
	pc->e[i].empty = (uint8_t)p->empty;
	pc->e[i].plan = p->plan;
	pc->n++;
}

/*
 * Front half of a batch: parse, build the token sets, resolve (exact on the
 * host, misses through one device BK-tree pass), compile the device plans.
 * prep[i].errcode / .empty tell how query i ended.  Parsing + lookups and the
 * compilation are spread over the instance's worker pool.
 */
typedef struct {
	const nxs_index_t *	idx;
	const search_params_t *	sp;
	const char *const *	queries;
	qprep_t *		prep;
	const struct plan_cache *pc;	/* read-only while the workers run */
} plan_job_t;

static void
plan_parse_chunk(void *arg, size_t lo, size_t hi)
{
	const plan_job_t *j = arg;

	for (size_t i = lo; i < hi; i++) {
		qprep_t *q = &j->prep[i];

		if (j->pc) {
			const pc_ent_t *e = plan_cache_find(j->pc, j->queries[i], strlen(j->queries[i]), j->sp->fuzzymatch);
			if (e) {
				memset(q, 0, sizeof(*q));
				q->plan = e->plan;
				q->empty = e->empty != 0;
				q->cached = true;
				continue;
			}
		}
		nxs_query_prepare(j->idx, j->queries[i], q);
		if (q->errcode) {
			nxs_query_release_scratch(q);	/* (what the second pass would do for it) */
			q->compiled = true;
			continue;
		}
		/* idxterm_lookup for every token (tokenizer.c:171-176) */
		bool miss = false;
		for (size_t k = 0; k < q->n_tokens; k++) {
			qtok_t *t = &q->tokens[k];
			t->term_id = nxs_term_lookup(j->idx, (const uint8_t *)t->value, t->len);
			miss = miss || !t->term_id;
		}
		/* nothing of this query waits for the fuzzy search: compile it here and now -- a batch without
		 * misses (or with fuzzymatch off) is ONE run over the worker threads, not two */
		if (!miss || !j->sp->fuzzymatch) {
			(void)nxs_query_compile(q);
			nxs_query_release_scratch(q);
			q->compiled = true;
		}
	}
}

static void
plan_compile_chunk(void *arg, size_t lo, size_t hi)
{
	const plan_job_t *j = arg;

	for (size_t i = lo; i < hi; i++) {
		if (j->prep[i].cached || j->prep[i].compiled) {
			continue;
		}
		if (!j->prep[i].errcode) {
			(void)nxs_query_compile(&j->prep[i]);
		}
		/* the parse and the token list have done their job: freed here, on the
		 * worker, not on the caller's critical path */
		nxs_query_release_scratch(&j->prep[i]);
	}
}

/* the tokens of a batch that missed the dictionary, as one byte string (tokenizer.c:177-180) */
typedef struct {
	uint32_t *	q, *t, *off, *ids;	/* [n]: query, token index, byte offset, winner */
	uint8_t *	bytes;
	size_t		n;
} fz_set_t;

static void
fz_set_free(fz_set_t *fz)
{
	free(fz->q);
	free(fz->t);
	free(fz->off);
	free(fz->ids);
	free(fz->bytes);
	memset(fz, 0, sizeof(*fz));
}

/* parse + lookups (+ compile for the queries without misses) on the worker threads; the misses into *fz */
static int
plan_front(nxs_index_t *idx, const search_params_t *sp, const char *const *queries,
    size_t n, qprep_t *prep, fz_set_t *fz)
{
	nxs_t *nxs = idx->nxs;
	struct nxs_pool *pool = n >= 64 ? nxs_pool_get(nxs) : NULL;
	plan_job_t job = { .idx = idx, .sp = sp, .queries = queries, .prep = prep, .pc = plan_cache_get(idx) };
	size_t n_fz = 0, fz_len = 0, k = 0, o = 0;

	memset(fz, 0, sizeof(*fz));
	pool_run(pool, plan_parse_chunk, &job, n, 16);

	for (size_t i = 0; sp->fuzzymatch && i < n; i++) {
		const qprep_t *q = &prep[i];

		for (size_t j = 0; !q->errcode && !q->compiled && !q->cached && j < q->n_tokens; j++) {
			if (!q->tokens[j].term_id) {
				n_fz++;
				fz_len += q->tokens[j].len;
			}
		}
	}
	if (!n_fz) {
		return 0;
	}
	fz->q = malloc(n_fz * sizeof(uint32_t));
	fz->t = malloc(n_fz * sizeof(uint32_t));
	fz->off = malloc((n_fz + 1) * sizeof(uint32_t));
	fz->ids = calloc(n_fz, sizeof(uint32_t));
	fz->bytes = malloc(fz_len + 16);
	if (!fz->q || !fz->t || !fz->off || !fz->ids || !fz->bytes) {
		nxs_decl_err(nxs, NXS_ERR_SYSTEM, "out of memory");
		fz_set_free(fz);
		return -1;
	}
	for (size_t i = 0; i < n; i++) {
		qprep_t *q = &prep[i];
		if (q->errcode || q->compiled || q->cached) {
			continue;
		}
		for (size_t j = 0; j < q->n_tokens; j++) {
			const qtok_t *t = &q->tokens[j];
			if (t->term_id) {
				continue;
			}
			fz->q[k] = (uint32_t)i;
			fz->t[k] = (uint32_t)j;
			fz->off[k] = (uint32_t)o;
			memcpy(fz->bytes + o, t->value, t->len);
			o += t->len;
			k++;
		}
	}
	fz->off[k] = (uint32_t)o;
	fz->n = n_fz;
	return 0;
}

/* the winners into the token lists, the remaining queries compiled, the batch's new plans into the cache */
static void
plan_back(nxs_index_t *idx, const search_params_t *sp, const char *const *queries,
    size_t n, qprep_t *prep, const fz_set_t *fz)
{
	struct nxs_pool *pool = n >= 64 ? nxs_pool_get(idx->nxs) : NULL;
	struct plan_cache *pc = plan_cache_get(idx);
	plan_job_t job = { .idx = idx, .sp = sp, .queries = queries, .prep = prep, .pc = pc };
	bool second = fz->n != 0;

	for (size_t k = 0; k < fz->n; k++) {
		prep[fz->q[k]].tokens[fz->t[k]].term_id = fz->ids[k];
	}
	for (size_t i = 0; !second && i < n; i++) {
		second = !prep[i].cached && !prep[i].compiled;
	}
	if (second) {
		pool_run(pool, plan_compile_chunk, &job, n, 32);
	}
	/* (this thread only; a query whose string is not at hand -- the late half keeps only the strings it
	 * still has to compile -- was put there by the first half's caller or is not cached) */
	for (size_t i = 0; pc && i < n; i++) {
		const qprep_t *q = &prep[i];
		if (queries[i] && !q->cached && !q->errcode && !q->wide) {
			plan_cache_put(pc, queries[i], strlen(queries[i]), sp->fuzzymatch, q);
		}
	}
}

static int late_finish(nxs_index_t *);

static int
plan_batch(nxs_index_t *idx, const search_params_t *sp, const char *const *queries,
    size_t n, qprep_t *prep)
{
	fz_set_t fz;
	int ret = -1;

	if (plan_front(idx, sp, queries, n, prep, &fz) == -1) {
		return -1;
	}
	if (fz.n) {
		/* one device BK-tree pass for every token that missed */
		(void)late_finish(idx);		/* (a batch whose own pass is still on the device: one pass at a time) */
		if (nxs_index_bk_sync(idx) == -1) {
			goto out;
		}
		if (nxsgpu_fuzzy(idx->dev, fz.bytes, fz.off, (uint32_t)fz.n, fz.ids, NULL) != 0) {
			nxs_decl_err(idx->nxs, NXS_ERR_FATAL, "device fuzzy search failed: %s",
			    nxsgpu_last_error());
			goto out;
		}
	}
	plan_back(idx, sp, queries, n, prep, &fz);
	ret = 0;
out:
	fz_set_free(&fz);
	return ret;
}

/*
 * tokenizer.c:177-180 resolves a miss where it meets it; here a batch's misses are one device pass, and
 * _begin does not wait for it: it returns once the pass is queued, and the batch's second half runs
 * when the host comes by again (late_finish) -- the pass has had the caller's work on the previous
 * responses and the next batch's parse to finish in.  What the second half needs is kept here.
 */
struct late_half {
	search_params_t	sp;
	fz_set_t	fz;
	int		slot;		/* nxsgpu_fuzzy_begin's */
	bool		collected;	/* the pass is over, fz.ids hold the winners */
	char *		qbuf;		/* the strings still to compile (the caller's may be gone by then) */
	const char **	queries;	/* [hi - lo]: into qbuf; NULL = nothing left to do for that query */
};

static void
late_free(struct late_half *lh)
{
	if (lh) {
		fz_set_free(&lh->fz);
		free(lh->qbuf);
		free(lh->queries);
		free(lh);
	}
}

static struct late_half *
late_make(const search_params_t *sp, fz_set_t *fz, const char *const *queries, size_t n, const qprep_t *prep)
{
	struct late_half *lh = calloc(1, sizeof(*lh));
	size_t total = 0, o = 0;

	if (!lh) {
		return NULL;
	}
	lh->sp = *sp;
	for (size_t i = 0; i < n; i++) {
		if (!prep[i].cached && !prep[i].compiled && !prep[i].errcode) {
			total += strlen(queries[i]) + 1;
		}
	}
	lh->queries = calloc(n ? n : 1, sizeof(*lh->queries));
	lh->qbuf = malloc(total ? total : 1);
	if (!lh->queries || !lh->qbuf) {
		late_free(lh);
		return NULL;
	}
	for (size_t i = 0; i < n; i++) {
		if (!prep[i].cached && !prep[i].compiled && !prep[i].errcode) {
			const size_t l = strlen(queries[i]) + 1;
			memcpy(lh->qbuf + o, queries[i], l);
			lh->queries[i] = lh->qbuf + o;
			o += l;
		}
	}
	lh->fz = *fz;			/* (moved) */
	memset(fz, 0, sizeof(*fz));
	return lh;
}

int
nxs_index_plan_batch(nxs_index_t *idx, nxs_params_t *params,
    const char *const *queries, size_t n, struct nxsgpu_query *plans_out,
    nxs_err_t *errs)
{
	nxsgpu_query_t *plans = (nxsgpu_query_t *)plans_out;
	search_params_t sp;
	qprep_t *prep;
	int failed = 0;

	nxs_clear_error(idx->nxs);
	if (get_search_params(idx, params, &sp) == -1) {
		return -1;
	}
	if (nxs_index_refresh(idx) == -1) {	/* search.c:309-312 */
		return -1;
	}
	if ((prep = calloc(n ? n : 1, sizeof(qprep_t))) == NULL) {
		nxs_decl_err(idx->nxs, NXS_ERR_SYSTEM, "out of memory");
		return -1;
	}
	if (plan_batch(idx, &sp, queries, n, prep) == -1) {
		failed = -1;
	}
	for (size_t i = 0; i < n; i++) {
		memset(&plans[i], 0, sizeof(plans[i]));
		if (failed != -1) {
			nxs_err_t code = prep[i].errcode;

			if (!code && prep[i].wide) {
				/* a fixed-size nxsgpu_query_t cannot hold it */
				code = NXS_ERR_LIMIT;
				nxs_decl_err(idx->nxs, code, "query %zu has more than %u terms: "
				    "use nxs_index_search_batch", i, NXSGPU_MAX_TOKENS);
			} else if (code) {
				nxs_decl_err(idx->nxs, code, "%s",
				    prep[i].errmsg ? prep[i].errmsg : "");
			} else if (!prep[i].empty) {
				plans[i] = prep[i].plan;
			}
			failed += code != 0;
			if (errs) {
				errs[i] = code;
			}
		}
		nxs_query_release(&prep[i]);
	}
	free(prep);
	return failed;
}

/* ---- batches: begin / end ------------------------------------------------------- */

static inline double
now_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

#ifdef NXS_TEST_HOOKS
/*
 * Where the host's time goes, summed over the batches so far: out[0] parse +
 * resolve + compile (worker pool), out[1] queueing the batch on the device
 * (work list, staging, launches), out[2] waiting for the device, out[3]
 * building the responses; out[4] = batches; out[8] = the part of out[0] spent waiting for the device's
 * fuzzy pass.  Reset on read.
 */
void
nxs_index_host_profile(nxs_index_t *idx, double out[12])
{
	out[8] = idx->hp_fzwait;	/* of out[0]: waiting for the device's fuzzy pass, */
	out[9] = idx->hp_front;		/* parse + lookups (+ compile of the queries without misses), */
	out[10] = idx->hp_fzlaunch;	/* queueing the fuzzy pass, */
	out[11] = idx->hp_back;		/* winners into the plans + compile of the rest */
	idx->hp_fzwait = idx->hp_front = idx->hp_fzlaunch = idx->hp_back = 0;
	out[6] = idx->hp_begin;		/* whole _begin() / _end() calls */
	out[7] = idx->hp_end;
	idx->hp_begin = idx->hp_end = 0;
	out[5] = (double)idx->hp_inexact;	/* queries re-run on the exact path */
	idx->hp_inexact = 0;
	out[0] = idx->hp_plan;
	out[1] = idx->hp_queue;
	out[2] = idx->hp_wait;
	out[3] = idx->hp_resps;
	out[4] = (double)idx->hp_batches;
	idx->hp_plan = idx->hp_queue = idx->hp_wait = idx->hp_resps = 0;
	idx->hp_batches = 0;
}

void
nxs_index_shard_info(nxs_index_t *idx, uint64_t out[4])
{
	uint64_t st[2] = { 0, 0 };

	out[0] = (uint64_t)(int64_t)(idx->comm ? nxsgpu_comm_rccl_count(idx->comm) : -1);
	out[1] = idx->comm ? (uint64_t)nxsgpu_comm_world(idx->comm) : 1;
	nxsgpu_comm_stats(idx->comm, st);
	out[2] = st[0];
	out[3] = st[1];
}

#endif /* NXS_TEST_HOOKS */

/* status word of a record slot: 0, an nxs_err_t, or ... */
#define	STATUS_HOSTPATH	0x100u	/* the owner evaluates it on the exact path (fix-up round) */
/*
 * A rank that cannot do its share of a sharded batch (planning failed, out of
 * memory, its exact fix-up failed) must not leave its peers waiting in the
 * all-gather: it still contributes a block, every status word of which carries
 * STATUS_ABORT | its error code.  All ranks see all blocks, so all fail the batch
 * together -- the collectives of every rank stay in step.
 */
#define	STATUS_ABORT	0x200u

/* the first rank whose block says "aborted" (and its error code), or -1 */
static int
blocks_aborted(const uint8_t *blocks, uint32_t world, uint32_t n_slots, uint32_t k, nxs_err_t *code)
{
	const size_t rec_bytes = NXSGPU_REC_BYTES(k), block_bytes = NXSGPU_BLOCK_BYTES(n_slots, k);

	for (uint32_t r = 0; n_slots && r < world; r++) {
		const uint32_t *st = (const uint32_t *)(blocks + (size_t)r * block_bytes + (size_t)n_slots * rec_bytes);
		if (st[0] & STATUS_ABORT) {
			*code = (nxs_err_t)(st[0] & 0xff);
			return (int)r;
		}
	}
	return -1;
}

static nxs_pend_t *
pend_oldest(nxs_index_t *idx)
{
	nxs_pend_t *p = NULL;

	for (int i = 0; i < NXSGPU_INFLIGHT; i++) {
		if (idx->pend[i].active && (!p || idx->pend[i].seq < p->seq)) {
			p = &idx->pend[i];
		}
	}
	return p;
}

static void
pend_release(nxs_pend_t *p)
{
	for (size_t i = 0; p->prep && i < p->hi - p->lo; i++) {
		nxs_query_release(&p->prep[i]);
	}
	free(p->prep);
	late_free(p->late);
	for (size_t i = 0; p->st_resps && i < p->n; i++) {
		if (p->st_resps[i]) {		/* stashed and never collected */
			nxs_resp_release(p->st_resps[i]);
		}
	}
	free(p->st_resps);
	free(p->st_errs);
	free(p->st_errmsg);
	memset(p, 0, sizeof(*p));
}

static int batch_end_core(nxs_index_t *, nxs_pend_t *, nxs_resp_t **, nxs_err_t *);

/*
 * search.c:309-312: the reference syncs with the index files before EVERY search.
 * A refresh swaps device arrays the batches in flight read, so when the files
 * have moved (nxs_index_changed: four loads) the batches in flight are finished
 * here, oldest first, their responses kept for the caller's _end -- then the
 * index is refreshed and the new batch sees the change.  In the steady state of
 * a pipelined server (one batch always in flight) nothing else ever would.
 */
/*
 * Finish the batches in flight, oldest first, and keep their outcome (responses, error
 * slot) for the caller's _end.
 */
static int
stash_inflight(nxs_index_t *idx)
{
	int failed = 0;

	(void)late_finish(idx);		/* (a failure is that batch's: kept in its slot) */

	for (;;) {
		nxs_pend_t *pd = NULL;

		for (int i = 0; i < NXSGPU_INFLIGHT; i++) {
			nxs_pend_t *c = &idx->pend[i];
			if (c->active && !c->stashed && (!pd || c->seq < pd->seq)) {
				pd = c;
			}
		}
		if (!pd) {
			if (failed) {
				nxs_decl_err(idx->nxs, NXS_ERR_SYSTEM, "out of memory");
			}
			return failed;
		}
		pd->st_resps = calloc(pd->n ? pd->n : 1, sizeof(*pd->st_resps));
		pd->st_errs = calloc(pd->n ? pd->n : 1, sizeof(*pd->st_errs));
		if (!pd->st_resps || !pd->st_errs) {
			/*
			 * No memory to keep the batch's outcome: the batch is given up (its _end reports the
			 * error) but its device slot is still handed back HERE, in order -- a caller that goes on
			 * to end a younger slot (abort_collective) must not find this one the oldest.
			 */
			free(pd->st_resps);
			free(pd->st_errs);
			pd->st_resps = NULL;
			pd->st_errs = NULL;
			if (pd->on_device) {
				nxsgpu_batch_view_t v;
				(void)nxsgpu_batch_end(idx->dev, &v);
				pd->on_device = false;
			}
			pd->st_ret = -1;
			pd->st_errcode = NXS_ERR_SYSTEM;
			pd->st_errmsg = strdup("out of memory");
			pd->stashed = true;
			failed = -1;
			continue;
		}
		pd->st_ret = batch_end_core(idx, pd, pd->st_resps, pd->st_errs);
		pd->st_errcode = idx->nxs->errcode;
		pd->st_errmsg = idx->nxs->errmsg ? strdup(idx->nxs->errmsg) : NULL;
		pd->stashed = true;
		nxs_clear_error(idx->nxs);
	}
}

static int
resync_before_batch(nxs_index_t *idx)
{
	/*
	 * Sharded: the fix-up round of a batch in flight is a collective, so the ranks have to
	 * agree on WHICH _begin finishes the batches in flight.  Each rank says in the flags
	 * word of its record block whether it saw the files move (NXSGPU_BLOCK_CHANGED, set in
	 * _begin); every rank reads all flags after the all-gather (batch_end_core) and, if any
	 * is set, drains at its next _begin -- the same one on every rank, since all of them
	 * make the same calls in the same order.
	 */
	if (pend_oldest(idx)) {
		if (idx->comm ? !idx->resync_pending : !nxs_index_changed(idx)) {
			return idx->comm ? 0 : nxs_index_refresh(idx);
		}
		if (stash_inflight(idx) == -1) {
			return -1;
		}
	}
	idx->resync_pending = false;
	return nxs_index_refresh(idx);
}

/* any rank's block flags say "my index files moved" (all W blocks present) */
static bool
blocks_changed(const uint8_t *blocks, uint32_t world, uint32_t n_slots, uint32_t k)
{
	const size_t rec_bytes = NXSGPU_REC_BYTES(k), block_bytes = NXSGPU_BLOCK_BYTES(n_slots, k);

	for (uint32_t r = 0; r < world; r++) {
		const uint32_t *st = (const uint32_t *)(blocks + (size_t)r * block_bytes + (size_t)n_slots * rec_bytes);
		if (st[n_slots] & NXSGPU_BLOCK_CHANGED) {
			return true;
		}
	}
	return false;
}

/* batches never collected (the caller closes the index instead): wait, drop */
static void
index_drain(nxs_index_t *idx)
{
	nxs_pend_t *pd;

	if (idx->dev) {
		(void)late_finish(idx);
	}
	while ((pd = pend_oldest(idx)) != NULL) {
		nxsgpu_batch_view_t v;

		if (pd->on_device && !pd->stashed && idx->dev) {
			(void)nxsgpu_batch_end(idx->dev, &v);
		}
		pend_release(pd);
	}
}

/* the planned batch (record path: limit <= NXSGPU_BIG_K) onto the device; 0, or -1 with the error declared */
static int
queue_on_device(nxs_index_t *idx, nxs_pend_t *pd, const search_params_t *sp, bool collective)
{
	nxs_t *nxs = idx->nxs;
	const size_t nl = pd->hi - pd->lo;
	nxsgpu_query_t *plans = malloc((nl ? nl : 1) * sizeof(nxsgpu_query_t));
	uint32_t *slot_of = malloc((nl ? nl : 1) * sizeof(uint32_t));
	uint32_t *status = calloc(NXSGPU_STATUS_WORDS(pd->cap), sizeof(uint32_t));
	size_t n_plans = 0;
	int ret = -1;

	if (!plans || !slot_of || !status) {
		nxs_decl_err(nxs, NXS_ERR_SYSTEM, "out of memory");
		goto out;
	}
	for (size_t i = 0; i < nl; i++) {
		const qprep_t *q = &pd->prep[i];

		if (q->errcode) {
			status[i] = q->errcode;
		} else if (q->wide) {
			status[i] = STATUS_HOSTPATH;
		} else if (!q->empty) {
			slot_of[n_plans] = (uint32_t)i;
			plans[n_plans++] = q->plan;
		}
	}
	if (collective && nxs_index_changed(idx)) {
		/* a batch is in flight (else resync_before_batch refreshed just now): tell
		 * the peers, all ranks drain and re-sync together */
		status[pd->cap] = NXSGPU_BLOCK_CHANGED;
	}
	/* the worker threads are lent for THIS call only (the pool takes one run at a time: the doc-shard
	 * entry runs a host thread per shard through the same device layer and must never find it set) */
	nxsgpu_index_set_parallel(idx->dev, api_parallel, nxs);
	const int brc = nxsgpu_batch_begin(idx->dev, sp->algo, (uint32_t)sp->limit, plans,
	    (uint32_t)n_plans, slot_of, status, pd->cap,
	    idx->comm != NULL && pd->world >= 1 && !idx->emu_world);
	nxsgpu_index_set_parallel(idx->dev, NULL, NULL);
	if (brc != 0) {
		nxs_decl_err(nxs, NXS_ERR_FATAL, "device search failed: %s", nxsgpu_last_error());
		goto out;
	}
	pd->on_device = true;
	ret = 0;
out:
	free(plans);
	free(slot_of);
	free(status);
	return ret;
}

/*
 * The second half of a batch whose fuzzy pass was left running (struct late_half), in two steps: COLLECT
 * waits for the pass and takes its winners (the fuzzy workspaces are free again: the next batch's pass can
 * be queued), COMPLETE finishes the plans and queues the batch.  The batch's _begin has long returned
 * success, so a failure here is kept in the batch's slot for its _end (like a batch finished early by a
 * re-sync).
 */
static nxs_pend_t *
late_oldest(nxs_index_t *idx)
{
	nxs_pend_t *pd = NULL;

	for (int i = 0; i < NXSGPU_INFLIGHT; i++) {
		nxs_pend_t *c = &idx->pend[i];
		if (c->active && c->late && (!pd || c->seq < pd->seq)) {
			pd = c;
		}
	}
	return pd;
}

static void
late_failed(nxs_index_t *idx, nxs_pend_t *pd)
{
	nxs_t *nxs = idx->nxs;

	pd->st_ret = -1;
	pd->st_errcode = nxs->errcode ? nxs->errcode : NXS_ERR_FATAL;
	pd->st_errmsg = nxs->errmsg ? strdup(nxs->errmsg) : NULL;
	pd->stashed = true;
	nxs_clear_error(nxs);
	late_free(pd->late);
	pd->late = NULL;
}

static int
late_collect(nxs_index_t *idx, nxs_pend_t *pd)
{
	struct late_half *lh = pd->late;
	const double t0 = now_s();

	if (lh->collected) {
		return 0;
	}
	if (nxsgpu_fuzzy_end(idx->dev, lh->slot, lh->fz.bytes, lh->fz.off, (uint32_t)lh->fz.n, lh->fz.ids) != 0) {
		nxs_decl_err(idx->nxs, NXS_ERR_FATAL, "device fuzzy search failed: %s", nxsgpu_last_error());
		late_failed(idx, pd);
		return -1;
	}
	lh->collected = true;
	idx->hp_fzwait += now_s() - t0;
	idx->hp_plan += now_s() - t0;
	return 0;
}

static int
late_complete(nxs_index_t *idx, nxs_pend_t *pd)
{
	struct late_half *lh = pd->late;
	const double t0 = now_s();
	double t1;

	plan_back(idx, &lh->sp, lh->queries, pd->hi - pd->lo, pd->prep, &lh->fz);
	t1 = now_s();
	idx->hp_back += t1 - t0;
	idx->hp_plan += t1 - t0;
	if ((idx->test_fail_late && idx->test_fail_late-- == 1 &&
	    (nxs_decl_err(idx->nxs, NXS_ERR_SYSTEM, "injected failure in the late half (test)"), true)) ||
	    queue_on_device(idx, pd, &lh->sp, false) != 0) {
		late_failed(idx, pd);
		return -1;
	}
	idx->hp_queue += now_s() - t1;
	late_free(lh);
	pd->late = NULL;
	return 0;
}

/* every late batch, oldest first (0: nothing to do, or all went well) */
static int
late_finish(nxs_index_t *idx)
{
	nxs_pend_t *pd;
	int ret = 0;

	while ((pd = late_oldest(idx)) != NULL) {
		if (late_collect(idx, pd) != 0 || late_complete(idx, pd) != 0) {
			ret = -1;
		}
	}
	return ret;
}

int
nxs_index_search_batch_begin(nxs_index_t *idx, nxs_params_t *params,
    const char *const *queries, size_t n)
{
	nxs_t *nxs = idx->nxs;
	nxs_pend_t *pd = NULL;
	search_params_t sp;
	uint32_t *status = NULL;
	uint64_t lo = 0, hi = n;
	size_t nl;
	double t0, t1 = 0;
	const double t_in = now_s();
	fz_set_t fz = { 0 };
	nxs_pend_t *old;
	int ret = -1;

	nxs_clear_error(nxs);
	if (get_search_params(idx, params, &sp) == -1) {
		return -1;
	}
	for (int i = 0; i < NXSGPU_INFLIGHT; i++) {
		if (!idx->pend[i].active) {
			pd = &idx->pend[i];
			break;
		}
	}
	if (!pd) {
		nxs_decl_err(nxs, NXS_ERR_INVALID, "%d batches are already in flight", NXSGPU_INFLIGHT);
		return -1;
	}
	/* search.c:309-312: pick up what other processes appended or removed */
	if (resync_before_batch(idx) == -1) {
		return -1;
	}
	if (n > UINT32_MAX / 2) {
		nxs_decl_err(nxs, NXS_ERR_LIMIT, "batch too large");
		return -1;
	}
	memset(pd, 0, sizeof(*pd));
	pd->n = n;
	pd->limit = sp.limit;
	pd->algo = sp.algo;
	pd->world = 1;
	/* query sharding: fixed-size records, limit <= NXSGPU_BIG_K;
	 * larger limits run replicated -- every rank computes the whole batch */
	if (idx->comm && sp.limit <= NXSGPU_BIG_K) {
		pd->rank = nxsgpu_comm_rank(idx->comm);
		pd->world = nxsgpu_comm_world(idx->comm);
		nxsgpu_shard_slice(n, pd->rank, pd->world, &lo, &hi);
	} else if (idx->emu_world > 1 && sp.limit <= NXSGPU_BIG_K) {
		/* tests: this process plays ONE rank of a W-rank run, no collective */
		pd->rank = idx->emu_rank;
		pd->world = idx->emu_world;
		nxsgpu_shard_slice(n, pd->rank, pd->world, &lo, &hi);
	}
	pd->lo = lo;
	pd->hi = hi;
	pd->cap = (uint32_t)nxsgpu_shard_capacity(n, pd->world);
	nl = hi - lo;
	/* (a rank of a real communicator: its peers queue an all-gather for this batch, so from
	 * here on a failure of this rank still has to contribute a block: abort_collective) */
	const bool collective = idx->comm != NULL && !idx->emu_world && sp.limit <= NXSGPU_BIG_K;
	pd->prep = calloc(nl ? nl : 1, sizeof(qprep_t));
	if (!pd->prep) {
		nxs_decl_err(nxs, NXS_ERR_SYSTEM, "out of memory");
		if (collective) {
			goto abort_collective;
		}
		goto out;
	}
	t0 = now_s();
	if (plan_front(idx, &sp, queries + lo, nl, pd->prep, &fz) == -1 ||
	    (idx->test_fail_begin && idx->test_fail_begin-- == 1 &&
	    (nxs_decl_err(nxs, NXS_ERR_SYSTEM, "injected failure (test)"), true))) {
		(void)late_finish(idx);
		if (collective) {
			goto abort_collective;
		}
		goto out;
	}
	t1 = now_s();
	idx->hp_front += t1 - t0;
	idx->hp_plan += t1 - t0;
	/*
	 * The batch before this one may still lack its second half: its fuzzy pass has had the time since its
	 * _begin returned (the caller's work, this batch's parse).  THIS batch's pass is queued first (the
	 * device layer has two sets of fuzzy workspaces; the passes run in order), then the older batch's
	 * winners are collected, its plans compiled and the batch sent to the device -- it still goes there
	 * before this one.  A failure of the older batch is its own (reported by its _end).
	 */
	old = late_oldest(idx);
	t0 = now_s();
	if (fz.n) {
		if (!idx->late_mode) {
			const char *e = getenv("NXS_LATE_FUZZY");	/* (once per index: the query path reads no environment) */
			idx->late_mode = e && atoi(e) == 0 ? 2 : 1;
		}
		/* (sharded batches, doc shards and limits beyond the record path wait for the pass here: their
		 * failure paths are collectives of their own) */
		const bool late = idx->late_mode == 1 && !collective && !idx->comm && !idx->emu_world &&
		    !idx->n_shards && sp.limit <= NXSGPU_BIG_K;
		struct late_half *lh = NULL;
		bool failed;

		if (old && (!late || idx->bk_upto != idx->last_id || idx->bk_flags_stale)) {
			/* (the BK-tree image is about to be replaced, or this batch's pass runs at once:
			 * nothing of the older batch's may be on the device then) */
			(void)late_finish(idx);
			old = NULL;
			t0 = now_s();
		}
		failed = nxs_index_bk_sync(idx) == -1;
		if (!failed && late) {
			if ((lh = late_make(&sp, &fz, queries + lo, nl, pd->prep)) == NULL) {
				nxs_decl_err(nxs, NXS_ERR_SYSTEM, "out of memory");
				failed = true;
			} else if ((lh->slot = nxsgpu_fuzzy_begin(idx->dev, lh->fz.bytes, lh->fz.off, (uint32_t)lh->fz.n)) < 0) {
				nxs_decl_err(nxs, NXS_ERR_FATAL, "device fuzzy search failed: %s", nxsgpu_last_error());
				late_free(lh);
				lh = NULL;
				failed = true;
			}
		}
		idx->hp_fzlaunch += now_s() - t0;
		idx->hp_plan += now_s() - t0;
		if (old) {
			/* (its failure would wipe this batch's error slot: kept aside) */
			const nxs_err_t code = nxs->errcode;
			char *msg = failed && nxs->errmsg ? strdup(nxs->errmsg) : NULL;

			if (late_collect(idx, old) == 0) {
				(void)late_complete(idx, old);
			}
			old = NULL;
			if (failed) {
				nxs_decl_err(nxs, code ? code : NXS_ERR_FATAL, "%s", msg ? msg : "");
			}
			free(msg);
		}
		if (failed) {
			if (collective) {
				goto abort_collective;
			}
			goto out;
		}
		if (lh) {
			pd->late = lh;
			idx->hp_batches++;
			idx->hp_begin += now_s() - t_in;
			pd->seq = ++idx->pend_seq;
			pd->active = true;
			ret = 0;
			goto out;
		}
		t0 = now_s();
		if (nxsgpu_fuzzy(idx->dev, fz.bytes, fz.off, (uint32_t)fz.n, fz.ids, NULL) != 0) {
			nxs_decl_err(nxs, NXS_ERR_FATAL, "device fuzzy search failed: %s", nxsgpu_last_error());
			if (collective) {
				goto abort_collective;
			}
			goto out;
		}
		idx->hp_fzwait += now_s() - t0;
	}
	if (old && late_collect(idx, old) == 0) {
		(void)late_complete(idx, old);
	}
	t1 = now_s();
	plan_back(idx, &sp, queries + lo, nl, pd->prep, &fz);
	idx->hp_back += now_s() - t1;
	t1 = now_s();
	idx->hp_plan += t1 - t0;
	if (sp.limit <= NXSGPU_BIG_K) {
		const int qrc = queue_on_device(idx, pd, &sp, collective);

		if (qrc != 0) {
			if (collective) {
				goto abort_collective;	/* (an empty block may still go up) */
			}
			goto out;
		}
	}
	idx->hp_queue += now_s() - t1;
	idx->hp_batches++;
	idx->hp_begin += now_s() - t_in;
	pd->seq = ++idx->pend_seq;
	pd->active = true;
	ret = 0;
out:
	fz_set_free(&fz);
	free(status);
	if (ret != 0) {
		pend_release(pd);
	}
	return ret;

abort_collective:
	/*
	 * This rank cannot do its share, but its peers have queued (or will queue) the
	 * batch's all-gather: contribute a block that says so and wait for the
	 * collective, so that every rank fails this batch and the next one starts in
	 * step.  The error of this rank stays in its slot.
	 */
	{
		const nxs_err_t code = nxs->errcode ? nxs->errcode : NXS_ERR_FATAL;
		char *msg = nxs->errmsg ? strdup(nxs->errmsg) : NULL;
		nxsgpu_batch_view_t v;

		free(status);
		status = calloc(NXSGPU_STATUS_WORDS(pd->cap), sizeof(uint32_t));
		for (uint32_t i = 0; status && i < pd->cap; i++) {
			status[i] = STATUS_ABORT | (uint32_t)code;
		}
		/*
		 * Order: the block's all-gather is queued FIRST (the peers queued theirs in their
		 * _begin), then the batches this rank still has in flight are finished -- the
		 * device hands its slots back oldest first, and an older batch's fix-up round is
		 * a collective the peers enter in their _end, after this batch's all-gather --,
		 * their outcome kept for the caller's _end; only then is the abort slot the
		 * oldest one.  (Ending it at once took the OLDER batch's slot: that batch's
		 * _end then read the abort block as its own.)
		 */
		if (nxsgpu_batch_begin(idx->dev, sp.algo, (uint32_t)sp.limit, NULL, 0, NULL, status, pd->cap, 1) == 0) {
			(void)stash_inflight(idx);
			(void)nxsgpu_batch_end(idx->dev, &v);
		}
		/* (if even the empty block cannot go up the communicator is unusable: the peers'
		 * collective never completes -- fatal for the sharded group) */
		nxs_decl_err(nxs, code, "%s", msg ? msg : "this rank aborted the sharded batch");
		free(msg);
	}
	goto out;
}

/* exact path (nxsgpu_search / nxsgpu_search_wide) for the given local queries */
static int
run_exact(nxs_index_t *idx, const nxs_pend_t *pd, const uint32_t *which, size_t nw,
    nxsgpu_results_t *res, nxsgpu_results_t *wres, uint32_t *pos)
{
	nxsgpu_query_t *plans = NULL;
	nxsgpu_wide_query_t *wplans = NULL;
	size_t np = 0, nwd = 0;
	int ret = -1;

	memset(res, 0, sizeof(*res));
	memset(wres, 0, sizeof(*wres));
	plans = malloc((nw ? nw : 1) * sizeof(nxsgpu_query_t));
	wplans = malloc((nw ? nw : 1) * sizeof(nxsgpu_wide_query_t));
	if (!plans || !wplans) {
		nxs_decl_err(idx->nxs, NXS_ERR_SYSTEM, "out of memory");
		goto out;
	}
	for (size_t j = 0; j < nw; j++) {
		const qprep_t *q = &pd->prep[which[j]];

		if (q->wide) {
			pos[j] = (uint32_t)nwd | 0x80000000u;
			wplans[nwd++] = q->wplan;
		} else {
			pos[j] = (uint32_t)np;
			plans[np++] = q->plan;
		}
	}
	if (np && nxsgpu_search(idx->dev, pd->algo, pd->limit, plans, (uint32_t)np, res) != 0) {
		nxs_decl_err(idx->nxs, NXS_ERR_FATAL, "device search failed: %s", nxsgpu_last_error());
		goto out;
	}
	if (nwd && nxsgpu_search_wide(idx->dev, pd->algo, pd->limit, wplans, (uint32_t)nwd, wres) != 0) {
		nxs_decl_err(idx->nxs, NXS_ERR_FATAL, "device search failed: %s", nxsgpu_last_error());
		goto out;
	}
	ret = 0;
out:
	free(plans);
	free(wplans);
	return ret;
}

static inline const nxsgpu_results_t *
exact_pick(const nxsgpu_results_t *res, const nxsgpu_results_t *wres, uint32_t pos, uint32_t *at)
{
	*at = pos & 0x7fffffffu;
	return (pos & 0x80000000u) ? wres : res;
}

/*
 * Responses of a whole batch from the ranks' record blocks, in query order
 * (rank r owns the contiguous slice nxsgpu_shard_slice(n, r, world)).  A slot
 * with a status word is a failed query: no response, its code in errs[].
 */
static int
resps_from_blocks(nxs_t *nxs, const nxs_pend_t *pd, size_t n, uint32_t world, uint32_t n_slots,
    uint32_t k, const uint8_t *blocks, nxs_resp_t **resps, nxs_err_t *errs, slab_builder_t *sb,
    int *failed, int only_rank)
{
	const size_t rec_bytes = NXSGPU_REC_BYTES(k), block_bytes = NXSGPU_BLOCK_BYTES(n_slots, k);
	size_t total = 0;

	/* only_rank >= 0 (nxs_index_shard_local): that rank's slice alone -- the other slices' responses stay
	 * NULL and their errs[] untouched: O(n / world) host work per rank and batch instead of O(n) */
	for (uint32_t r = 0; r < world; r++) {
		const uint8_t *blk = blocks + (size_t)r * block_bytes;
		uint64_t rlo, rhi;

		if (only_rank >= 0 && (int)r != only_rank) {
			continue;
		}
		nxsgpu_shard_slice(n, (int)r, (int)world, &rlo, &rhi);
		for (uint64_t i = 0; i < rhi - rlo; i++) {
			const uint32_t c = ((const uint32_t *)(blk + i * rec_bytes))[0];
			if (c > k) {
				nxs_decl_err(nxs, NXS_ERR_FATAL, "corrupted result record");
				return -1;
			}
			total += c;
		}
	}
	if (slab_begin(sb, n, total) == -1) {
		nxs_decl_err(nxs, NXS_ERR_SYSTEM, "out of memory");
		return -1;
	}
	for (uint32_t r = 0; r < world; r++) {
		const uint8_t *blk = blocks + (size_t)r * block_bytes;
		const uint32_t *st = (const uint32_t *)(blk + (size_t)n_slots * rec_bytes);
		uint64_t rlo, rhi;

		if (only_rank >= 0 && (int)r != only_rank) {
			continue;
		}
		nxsgpu_shard_slice(n, (int)r, (int)world, &rlo, &rhi);
		for (uint64_t i = 0; i < rhi - rlo; i++) {
			const uint8_t *rec = blk + i * rec_bytes;
			const uint32_t c = ((const uint32_t *)rec)[0];
			nxs_resp_t *rp;

			if (st[i]) {
				(*failed)++;
				if (errs) {
					errs[rlo + i] = (nxs_err_t)st[i];
				}
				if (pd && (int)r == pd->rank) {
					nxs_decl_err(nxs, (nxs_err_t)st[i], "%s",
					    pd->prep[i].errmsg ? pd->prep[i].errmsg : "");
				} else {
					nxs_decl_err(nxs, (nxs_err_t)st[i], "query %llu failed on rank %u",
					    (unsigned long long)(rlo + i), r);
				}
				continue;
			}
			rp = slab_resp(sb, rlo + i, c);
			memcpy(rp->ids, rec + 8, (size_t)c * 8);
			memcpy(rp->scores, rec + 8 + 8 * (size_t)k, (size_t)c * 4);
			resps[rlo + i] = rp;
		}
	}
	return 0;
}

int
nxs_index_search_batch_end(nxs_index_t *idx, nxs_resp_t **resps, nxs_err_t *errs)
{
	nxs_pend_t *pd = pend_oldest(idx);
	int ret;

	if (!pd) {
		nxs_clear_error(idx->nxs);
		nxs_decl_err(idx->nxs, NXS_ERR_INVALID, "no batch in flight");
		return -1;
	}
	if (pd->late) {
		(void)late_finish(idx);		/* (no later _begin came by: the second half runs here) */
	}
	if (pd->stashed) {
		/* finished early by a later _begin (resync_before_batch): hand over */
		nxs_clear_error(idx->nxs);
		for (size_t i = 0; i < pd->n; i++) {
			resps[i] = pd->st_resps ? pd->st_resps[i] : NULL;
			if (pd->st_resps) {
				pd->st_resps[i] = NULL;
			}
			if (errs) {
				errs[i] = pd->st_errs ? pd->st_errs[i] : pd->st_errcode;
			}
		}
		if (pd->st_errcode) {
			nxs_decl_err(idx->nxs, pd->st_errcode, "%s", pd->st_errmsg ? pd->st_errmsg : "");
		}
		ret = pd->st_ret;
	} else {
		ret = batch_end_core(idx, pd, resps, errs);
	}
	pend_release(pd);
	return ret;
}

/*
 * The fix-up round of a sharded batch, as every rank decides it from the gathered blocks:
 * a record marked inexact (candidate overflow) or a host-path query (wide plan) anywhere
 * means ALL ranks take a second all-gather, after each owner has re-run its own such
 * queries on the exact path (`which`: the owner's, local indexes).  `all` = the blocks of
 * all W ranks are present (else: this rank's block only -- one emulated rank, tests).
 */
static bool
fixup_scan(const uint8_t *blocks, bool all, uint32_t W, int rank, uint32_t n_slots, uint32_t k,
    size_t n, uint32_t *which, size_t *nw)
{
	const size_t rec_bytes = NXSGPU_REC_BYTES(k), block_bytes = NXSGPU_BLOCK_BYTES(n_slots, k);
	bool fixup = false;

	for (uint32_t r = 0; r < W; r++) {
		const uint8_t *blk = all ? blocks + (size_t)r * block_bytes : blocks;
		const uint32_t *st = (const uint32_t *)(blk + (size_t)n_slots * rec_bytes);
		uint64_t rlo, rhi;

		if (!all && (int)r != rank) {
			continue;
		}
		nxsgpu_shard_slice(n, (int)r, (int)W, &rlo, &rhi);
		for (uint64_t i = 0; i < rhi - rlo; i++) {
			const uint32_t *rec = (const uint32_t *)(blk + i * rec_bytes);
			if (rec[1] == NXSGPU_REC_INEXACT || st[i] == STATUS_HOSTPATH) {
				fixup = true;
				if ((int)r == rank) {
					which[(*nw)++] = (uint32_t)i;
				}
			}
		}
	}
	return fixup;
}

/* after the second all-gather: the rank that aborted in the fix-up round, or one that left a
 * record unpatched (still marked) -- every rank fails the batch then --, else -1 */
static int
fixup_verify(const uint8_t *blocks, uint32_t W, uint32_t n_slots, uint32_t k, size_t n, nxs_err_t *acode)
{
	const size_t rec_bytes = NXSGPU_REC_BYTES(k), block_bytes = NXSGPU_BLOCK_BYTES(n_slots, k);
	int ar = blocks_aborted(blocks, W, n_slots, k, acode);

	for (uint32_t r = 0; ar < 0 && r < W; r++) {
		const uint8_t *blk = blocks + (size_t)r * block_bytes;
		const uint32_t *st = (const uint32_t *)(blk + (size_t)n_slots * rec_bytes);
		uint64_t rlo, rhi;

		nxsgpu_shard_slice(n, (int)r, (int)W, &rlo, &rhi);
		for (uint64_t i = 0; i < rhi - rlo; i++) {
			if (((const uint32_t *)(blk + i * rec_bytes))[1] == NXSGPU_REC_INEXACT || st[i] == STATUS_HOSTPATH) {
				ar = (int)r;
				*acode = NXS_ERR_FATAL;
			}
		}
	}
	return ar;
}

static int
batch_end_core(nxs_index_t *idx, nxs_pend_t *pd, nxs_resp_t **resps, nxs_err_t *errs)
{
	nxs_t *nxs = idx->nxs;
	nxsgpu_results_t res, wres;
	slab_builder_t sb = { 0 };
	uint8_t *patched = NULL;
	bool patched_own = true;	/* `patched` is malloc()ed (not the slot's pinned blocks) */
	uint32_t *which, *pos;
	size_t nw = 0, total = 0, n, nl;
	double t0;
	const double t_in = now_s();
	int failed = 0, ret = -1;

	memset(&res, 0, sizeof(res));
	memset(&wres, 0, sizeof(wres));
	nxs_clear_error(nxs);
	n = pd->n;
	nl = pd->hi - pd->lo;
	for (size_t i = 0; i < n; i++) {
		resps[i] = NULL;
		if (errs) {
			errs[i] = NXS_ERR_SUCCESS;
		}
	}
	which = calloc(nl ? nl : 1, sizeof(uint32_t));
	pos = calloc(nl ? nl : 1, sizeof(uint32_t));
	if (!which || !pos) {
		nxs_decl_err(nxs, NXS_ERR_SYSTEM, "out of memory");
		goto out;
	}

	if (pd->on_device) {
		nxsgpu_batch_view_t v;
		const uint8_t *blocks;
		const uint32_t W = (uint32_t)pd->world;
		/* all W blocks are present after the all-gather; a single rank -- or the
		 * emulation of one rank of W (tests) -- holds its own block only */
		const bool all = W == 1 || !idx->emu_world;
		bool fixup = false;

		t0 = now_s();
		if (nxsgpu_batch_end(idx->dev, &v) != 0) {
			nxs_decl_err(nxs, NXS_ERR_FATAL, "device search failed: %s", nxsgpu_last_error());
			goto out;
		}
		idx->hp_wait += now_s() - t0;
		t0 = now_s();
		blocks = v.blocks;
		if (all && v.world != W) {
			nxs_decl_err(nxs, NXS_ERR_FATAL, "sharded batch came back with %u blocks, not %u",
			    v.world, W);
			goto out;
		}
		if (all && idx->comm && blocks_changed(blocks, W, v.n_slots, v.k)) {
			idx->resync_pending = true;	/* (every rank reads the same flags) */
		}
		if (all) {
			nxs_err_t acode;
			const int ar = blocks_aborted(blocks, W, v.n_slots, v.k, &acode);
			if (ar >= 0) {
				/* every rank sees it: all of them fail here, none enters a fix-up round */
				nxs_decl_err(nxs, acode ? acode : NXS_ERR_FATAL, "rank %d aborted the sharded batch", ar);
				goto out;
			}
		}
		/* records that need the exact path: every rank sees the same flags, so
		 * every rank takes (or skips) the fix-up round together */
		fixup = fixup_scan(blocks, all, W, pd->rank, v.n_slots, v.k, n, which, &nw);
		if (fixup) {
			const size_t len = (all ? (size_t)W : 1) * v.block_bytes;
			uint8_t *mine;

			bool fix_failed = false;
			char *fix_msg = NULL;
			nxs_err_t fix_code = NXS_ERR_SUCCESS;

			if ((patched = malloc(len ? len : 1)) == NULL) {
				nxs_decl_err(nxs, NXS_ERR_SYSTEM, "out of memory");
				if (!(all && W > 1)) {
					goto out;
				}
				/* (the peers are on their way into the fix-up all-gather: send the
				 * unpatched block -- records still marked inexact fail the batch on
				 * every rank, below) */
				fix_failed = true;
				fix_code = NXS_ERR_SYSTEM;
			} else {
				memcpy(patched, blocks, len);
			}
			mine = patched ? patched + (all ? (size_t)pd->rank * v.block_bytes : 0) :
			    (uint8_t *)(uintptr_t)(blocks + (size_t)pd->rank * v.block_bytes);
			if (!fix_failed && (run_exact(idx, pd, which, nw, &res, &wres, pos) != 0 ||
			    (idx->test_fail_fixup && idx->test_fail_fixup-- == 1 &&
			    (nxs_decl_err(nxs, NXS_ERR_SYSTEM, "injected failure (test)"), true)))) {
				if (!(all && W > 1)) {
					goto out;
				}
				/* this rank's exact pass failed: say so in its block and still take
				 * part in the collective */
				uint32_t *st = (uint32_t *)(mine + (size_t)v.n_slots * v.rec_bytes);
				fix_failed = true;
				fix_code = nxs->errcode ? nxs->errcode : NXS_ERR_FATAL;
				fix_msg = nxs->errmsg ? strdup(nxs->errmsg) : NULL;
				for (uint32_t i = 0; i < v.n_slots; i++) {
					st[i] = STATUS_ABORT | (uint32_t)fix_code;
				}
			}
			idx->hp_inexact += nw;
			for (size_t j = 0; !fix_failed && j < nw; j++) {
				uint8_t *rec = mine + (size_t)which[j] * v.rec_bytes;
				uint32_t *st = (uint32_t *)(mine + (size_t)v.n_slots * v.rec_bytes);
				uint32_t at;
				const nxsgpu_results_t *rs = exact_pick(&res, &wres, pos[j], &at);
				const uint32_t c = rs->counts[at];

				((uint32_t *)rec)[0] = c;
				((uint32_t *)rec)[1] = 0;
				memcpy(rec + 8, rs->doc_ids + rs->offsets[at], (size_t)c * 8);
				memcpy(rec + 8 + 8 * (size_t)v.k, rs->scores + rs->offsets[at], (size_t)c * 4);
				st[which[j]] = 0;
			}
			if (all && W > 1) {
				uint8_t *gathered = malloc(len);
				bool g_own = gathered != NULL;

				/*
				 * No memory to receive into: the slot's own pinned blocks (the first
				 * round's result: W blocks, the size this round needs) are always
				 * there -- the all-gather stages through device memory, so receiving
				 * over the block that is being sent is safe -- and what this rank
				 * needs of the first round is in `patched` (or it sends its block
				 * unpatched, which fails the batch everywhere).  A rank that is out
				 * of memory no longer strands its peers.
				 */
				if (!gathered || (idx->test_fail_fixup_recv && idx->test_fail_fixup_recv-- == 1)) {
					free(gathered);
					gathered = (uint8_t *)(uintptr_t)v.blocks;
					g_own = false;
				}
				if (nxsgpu_comm_allgather(idx->comm, mine, gathered, v.block_bytes) != 0) {
					nxs_decl_err(nxs, NXS_ERR_FATAL, "all-gather failed: %s", nxsgpu_last_error());
					if (g_own) {
						free(gathered);
					}
					free(fix_msg);
					goto out;
				}
				free(patched);
				patched = gathered;
				patched_own = g_own;
				{
					nxs_err_t acode;
					const int ar = fixup_verify(patched, W, v.n_slots, v.k, n, &acode);
					if (ar >= 0) {
						if (ar == pd->rank && fix_code) {
							nxs_decl_err(nxs, fix_code, "%s", fix_msg ? fix_msg : "exact pass failed");
						} else {
							nxs_decl_err(nxs, acode ? acode : NXS_ERR_FATAL,
							    "rank %d aborted the sharded batch (exact fix-up round)", ar);
						}
						free(fix_msg);
						goto out;
					}
				}
			}
			free(fix_msg);
			blocks = patched;
		}
		if (!all) {
			/* one rank of an emulated W-rank run: hand the block to the test */
			free(idx->emu_block);
			idx->emu_block = malloc(v.block_bytes ? v.block_bytes : 1);
			if (!idx->emu_block) {
				nxs_decl_err(nxs, NXS_ERR_SYSTEM, "out of memory");
				goto out;
			}
			memcpy(idx->emu_block, blocks, v.block_bytes);
			idx->emu_block_len = v.block_bytes;
			ret = 0;
			goto out;
		}
		if (resps_from_blocks(nxs, pd, n, W, v.n_slots, v.k, blocks, resps, errs, &sb, &failed,
		    (idx->shard_local && W > 1) ? pd->rank : -1) == -1) {
			goto out;
		}
		idx->hp_resps += now_s() - t0;
	} else {
		/* limit > NXSGPU_BIG_K: the exact two-pass path for the whole batch */
		for (size_t i = 0; i < nl; i++) {
			const qprep_t *q = &pd->prep[i];
			if (!q->errcode && !q->empty) {
				which[nw++] = (uint32_t)i;
			}
		}
		if (run_exact(idx, pd, which, nw, &res, &wres, pos) != 0) {
			goto out;
		}
		for (size_t j = 0; j < nw; j++) {
			uint32_t at;
			const nxsgpu_results_t *rs = exact_pick(&res, &wres, pos[j], &at);
			total += rs->counts[at];
		}
		if (slab_begin(&sb, n, total) == -1) {
			nxs_decl_err(nxs, NXS_ERR_SYSTEM, "out of memory");
			goto out;
		}
		for (size_t i = 0, j = 0; i < nl; i++) {
			const qprep_t *q = &pd->prep[i];
			nxs_resp_t *rp;

			if (q->errcode) {
				failed++;
				if (errs) {
					errs[i] = q->errcode;
				}
				nxs_decl_err(nxs, q->errcode, "%s", q->errmsg ? q->errmsg : "");
				continue;
			}
			if (q->empty) {
				resps[i] = slab_resp(&sb, i, 0);
				continue;
			}
			{
				uint32_t at;
				const nxsgpu_results_t *rs = exact_pick(&res, &wres, pos[j], &at);
				const uint32_t c = rs->counts[at];

				rp = slab_resp(&sb, i, c);
				memcpy(rp->ids, rs->doc_ids + rs->offsets[at], (size_t)c * 8);
				memcpy(rp->scores, rs->scores + rs->offsets[at], (size_t)c * 4);
				resps[i] = rp;
				j++;
			}
		}
	}
	if (sb.slab && sb.slab->refs == 0) {
		free(sb.slab);		/* every query failed */
	}
	ret = failed;
out:
	if (ret == -1 && sb.slab) {
		for (size_t i = 0; i < n; i++) {
			resps[i] = NULL;
		}
		free(sb.slab);
	}
	if (res.counts) {
		nxsgpu_results_free(&res);
	}
	if (wres.counts) {
		nxsgpu_results_free(&wres);
	}
	if (patched_own) {
		free(patched);
	}
	free(which);
	free(pos);
	idx->hp_end += now_s() - t_in;
	return ret;
}

int
nxs_index_search_batch(nxs_index_t *idx, nxs_params_t *params,
    const char *const *queries, size_t n, nxs_resp_t **resps, nxs_err_t *errs)
{
	if (pend_oldest(idx)) {
		nxs_clear_error(idx->nxs);
		nxs_decl_err(idx->nxs, NXS_ERR_INVALID,
		    "finish the batches in flight first (nxs_index_search_batch_end)");
		return -1;
	}
	for (size_t i = 0; i < n; i++) {
		resps[i] = NULL;
		if (errs) {
			errs[i] = NXS_ERR_SUCCESS;
		}
	}
	if (nxs_index_search_batch_begin(idx, params, queries, n) != 0) {
		return -1;
	}
	return nxs_index_search_batch_end(idx, resps, errs);
}

/* nxs_index_search: search.c:285-342 (one query = a batch of one) */
nxs_resp_t *
nxs_index_search(nxs_index_t *idx, nxs_params_t *params, const char *query, size_t len)
{
	nxs_resp_t *resp = NULL;
	const char *qv[1] = { query };
	nxsgpu_comm_t *comm = idx->comm;
	int r;

	(void)len;	/* the reference's lexer stops at the NUL byte too (search.c:177) */
	idx->comm = NULL;	/* a single query is never sharded */
	r = nxs_index_search_batch(idx, params, qv, 1, &resp, NULL);
	idx->comm = comm;
	if (r != 0) {
		if (resp) {
			nxs_resp_release(resp);
		}
		return NULL;
	}
	return resp;
}

/* ---- N4: doc-sharded collections --------------------------------------------------- */

/* collection-wide df = sum of the shards' (in a multi-process deployment: an
 * all-reduce of the same arrays); every shard then recomputes its impacts */
static int
docshard_set_global_df(nxs_index_t *const *shards, unsigned n_shards)
{
	nxs_t *nxs = shards[0]->nxs;
	const uint32_t T = shards[0]->last_id;
	uint32_t *sum = calloc((size_t)T + 2, sizeof(uint32_t));
	uint32_t *df = calloc((size_t)T + 2, sizeof(uint32_t));
	int ret = -1;

	if (!sum || !df) {
		nxs_decl_err(nxs, NXS_ERR_SYSTEM, "out of memory");
		goto out;
	}
	for (unsigned s = 0; s < n_shards; s++) {
		if (shards[s]->last_id != T || shards[s]->n_shards != n_shards || shards[s]->shard != s) {
			nxs_decl_err(nxs, NXS_ERR_INVALID, "the indexes are not shards 0..%u of one collection",
			    n_shards - 1);
			goto out;
		}
		(void)nxsgpu_index_df(shards[s]->dev, df);
		for (uint32_t t = 1; t <= T; t++) {
			sum[t] += df[t];
		}
	}
	for (unsigned s = 0; s < n_shards; s++) {
		if (nxsgpu_index_set_global_df(shards[s]->dev, sum, T) != 0) {
			nxs_decl_err(nxs, NXS_ERR_SYSTEM, "%s", nxsgpu_last_error());
			goto out;
		}
		shards[s]->global_df_set = true;
	}
	ret = 0;
out:
	free(sum);
	free(df);
	return ret;
}

/*
 * One shard's pass: the candidates its heap accepts for every plan of the batch
 * (nxsgpu_search_candidates), on the shard's own device and streams.  Shards of
 * one process run these side by side, one host thread each (the HIP side keeps
 * its error text per thread).
 */
typedef struct {
	nxs_index_t *	shard;
	int		algo;
	uint64_t	limit;
	const nxsgpu_query_t *plans;
	uint32_t	np, cap;
	uint64_t *	ids;	/* [np][cap] */
	float *		sc;
	uint32_t *	cnt;	/* [np] */
	int		ret;
	char		err[256];
} ds_job_t;

static void *
ds_job_run(void *arg)
{
	ds_job_t *j = arg;

	j->ret = nxsgpu_search_candidates(j->shard->dev, j->algo, j->limit, j->plans, j->np, j->cap,
	    j->ids, j->sc, j->cnt);
	if (j->ret != 0) {
		snprintf(j->err, sizeof(j->err), "%s", nxsgpu_last_error());
	}
	return NULL;
}

/*
 * The doc-sharded search (N4).  Two forms share everything but where the other
 * shards' candidates come from:
 *  - in-process (nxs_docshard_search_batch): `local` holds ALL n_shards shard
 *    indexes, each on its own device / streams; their passes run concurrently
 *    (one host thread per shard), then the merge;
 *  - one process per shard (nxs_docshard_search_batch_rank): `local` is this
 *    rank's shard; the ranks all-gather their candidate blocks
 *    (u32 abort | u32 cnt[np] | u64 ids[np][cap] | f32 sc[np][cap]) through the
 *    communicator attached with nxs_index_shard() and EVERY rank merges -- the
 *    query-sharded mode's rule: one collective per step, all ranks hold all
 *    responses.  `gathered` (tests): the blocks of all ranks, instead of a
 *    communicator; `my_block` (tests): hand out this rank's block and stop.
 * The merge feeds the shards' accepted-candidate logs, highest doc ids first,
 * through the reference's heap once more (nxsgpu_merge_candidates).
 */
static size_t
ds_block_bytes(size_t np, uint32_t cap)
{
	return 8 + ((np * 4 + 7) & ~(size_t)7) + np * cap * 8 + np * cap * 4;
}

static int
docshard_search(nxs_index_t *const *local, unsigned n_local, unsigned n_shards, unsigned my_shard,
    nxs_params_t *params, const char *const *queries, size_t n, nxs_resp_t **resps, nxs_err_t *errs,
    uint32_t cap0, const uint8_t *gathered, uint8_t **my_block, size_t *my_block_len)
{
	nxs_index_t *idx0 = local[0];
	nxs_t *nxs = idx0->nxs;
	const bool ranks = n_local == 1 && n_shards > 1;	/* one process per shard */
	search_params_t sp;
	qprep_t *prep = NULL;
	nxsgpu_query_t *plans = NULL;
	uint32_t *plan_of = NULL, *cnt_all = NULL, *o_cnt = NULL;
	uint64_t *ids_all = NULL, *o_ids = NULL;
	float *sc_all = NULL, *o_sc = NULL;
	ds_job_t *jobs = NULL;
	pthread_t *thr = NULL;
	uint8_t *sendb = NULL, *recvb = NULL;
	slab_builder_t sb = { 0 };
	size_t np = 0, total = 0;
	uint32_t cap = cap0 ? cap0 : 512;
	int failed = 0, ret = -1;

	nxs_clear_error(nxs);
	for (size_t i = 0; i < n; i++) {
		resps[i] = NULL;
		if (errs) {
			errs[i] = NXS_ERR_SUCCESS;
		}
	}
	if (get_search_params(idx0, params, &sp) == -1) {
		return -1;
	}
	if (sp.limit > NXSGPU_BIG_K) {
		nxs_decl_err(nxs, NXS_ERR_LIMIT, "doc-sharded search takes limit <= %d", NXSGPU_BIG_K);
		return -1;
	}
	/* a shard's heap accepts ~ k (1 + ln(matches / k)) items: room for that from the start (a log that
	 * overflows costs a second pass over every shard) */
	if (!cap0 && sp.limit > NXSGPU_FAST_K) {
		cap = (uint32_t)(sp.limit * 8 < 32768 ? sp.limit * 8 : 32768);
	}
	if (!idx0->global_df_set) {
		if (!ranks && docshard_set_global_df(local, n_shards) == -1) {
			return -1;
		}
		if (ranks) {
			nxs_decl_err(nxs, NXS_ERR_INVALID, "nxs_docshard_attach() the shard first (collection-wide df)");
			return -1;
		}
	}
	if (n == 0) {
		return 0;
	}
	prep = calloc(n, sizeof(qprep_t));
	plans = calloc(n, sizeof(nxsgpu_query_t));
	plan_of = calloc(n, sizeof(uint32_t));
	jobs = calloc(n_local, sizeof(ds_job_t));
	thr = calloc(n_local, sizeof(pthread_t));
	if (!prep || !plans || !plan_of || !jobs || !thr) {
		nxs_decl_err(nxs, NXS_ERR_SYSTEM, "out of memory");
		goto out;
	}
	/* the term dictionary and the BK-tree are the same on every shard */
	if (plan_batch(idx0, &sp, queries, n, prep) == -1) {
		goto out;
	}
	for (size_t i = 0; i < n; i++) {
		if (!prep[i].errcode && prep[i].wide) {
			prep[i].errcode = NXS_ERR_LIMIT;
			prep[i].errmsg = strdup("doc-sharded search takes at most 32 query terms");
		}
		if (!prep[i].errcode && !prep[i].empty) {
			plan_of[i] = (uint32_t)np;
			plans[np++] = prep[i].plan;
		}
	}
	o_ids = malloc((np ? np : 1) * sp.limit * sizeof(uint64_t));
	o_sc = malloc((np ? np : 1) * sp.limit * sizeof(float));
	o_cnt = calloc(np ? np : 1, sizeof(uint32_t));
	for (;;) {
		bool overflow = false;
		const size_t per = (np ? np : 1) * (size_t)cap;
		const size_t bb = ds_block_bytes(np, cap);

		free(ids_all); free(sc_all); free(cnt_all);
		ids_all = malloc(per * n_shards * sizeof(uint64_t));
		sc_all = malloc(per * n_shards * sizeof(float));
		cnt_all = calloc((np ? np : 1) * (size_t)n_shards, sizeof(uint32_t));
		for (unsigned s = 0; s < n_local; s++) {
			free(jobs[s].ids); free(jobs[s].sc); free(jobs[s].cnt);
			jobs[s].ids = malloc(per * sizeof(uint64_t));
			jobs[s].sc = malloc(per * sizeof(float));
			jobs[s].cnt = calloc(np ? np : 1, sizeof(uint32_t));
			if (!jobs[s].ids || !jobs[s].sc || !jobs[s].cnt) {
				o_cnt = (free(o_cnt), NULL);
			}
			jobs[s].shard = local[s];
			jobs[s].algo = sp.algo;
			jobs[s].limit = sp.limit;
			jobs[s].plans = plans;
			jobs[s].np = (uint32_t)np;
			jobs[s].cap = cap;
			jobs[s].ret = 0;
		}
		if (!o_ids || !o_sc || !o_cnt || !ids_all || !sc_all || !cnt_all) {
			nxs_decl_err(nxs, NXS_ERR_SYSTEM, "out of memory");
			goto out;
		}
		/* every local shard's pass is queued before any is waited for */
		if (np) {
			unsigned started = 0;
			for (unsigned s = 1; s < n_local; s++) {
				if (pthread_create(&thr[s], NULL, ds_job_run, &jobs[s]) != 0) {
					break;
				}
				started = s;
			}
			ds_job_run(&jobs[0]);
			for (unsigned s = 1; s <= started; s++) {
				(void)pthread_join(thr[s], NULL);
			}
			for (unsigned s = started + 1; s < n_local; s++) {
				ds_job_run(&jobs[s]);	/* (no thread: in line) */
			}
		}
		for (unsigned s = 0; s < n_local; s++) {
			if (jobs[s].ret != 0) {
				nxs_decl_err(nxs, NXS_ERR_FATAL, "device search failed: %s", jobs[s].err);
				if (!ranks) {
					goto out;
				}
			}
		}
		if (!ranks) {
			/* [query][shard][cap]: what the merge works on */
			for (unsigned s = 0; s < n_local; s++) {
				for (size_t q = 0; q < np; q++) {
					const size_t at = (q * n_shards + s) * cap;
					overflow = overflow || jobs[s].cnt[q] > cap;
					cnt_all[q * n_shards + s] = jobs[s].cnt[q];
					memcpy(ids_all + at, jobs[s].ids + q * cap, (size_t)cap * sizeof(uint64_t));
					memcpy(sc_all + at, jobs[s].sc + q * cap, (size_t)cap * sizeof(float));
				}
			}
		} else {
			/* this rank's block; a rank whose pass failed says so in the first word
			 * and still takes part in the collective */
			free(sendb); free(recvb);
			sendb = calloc(1, bb);
			recvb = malloc(bb * n_shards);
			if (!sendb || !recvb) {
				nxs_decl_err(nxs, NXS_ERR_SYSTEM, "out of memory");
				goto out;
			}
			const size_t cnt_len = (np * 4 + 7) & ~(size_t)7;
			((uint32_t *)sendb)[0] = jobs[0].ret != 0 ? (uint32_t)NXS_ERR_FATAL : 0u;
			memcpy(sendb + 8, jobs[0].cnt, np * 4);
			memcpy(sendb + 8 + cnt_len, jobs[0].ids, np * (size_t)cap * 8);
			memcpy(sendb + 8 + cnt_len + np * (size_t)cap * 8, jobs[0].sc, np * (size_t)cap * 4);
			if (my_block) {			/* tests: one rank at a time, no collective */
				*my_block = sendb;
				*my_block_len = bb;
				sendb = NULL;
				ret = 0;
				goto out;
			}
			if (gathered) {
				memcpy(recvb, gathered, bb * n_shards);
			} else if (nxsgpu_comm_allgather(idx0->comm, sendb, recvb, bb) != 0) {
				nxs_decl_err(nxs, NXS_ERR_FATAL, "all-gather failed: %s", nxsgpu_last_error());
				goto out;
			}
			for (unsigned s = 0; s < n_shards; s++) {
				const uint8_t *blk = recvb + (size_t)s * bb;
				const uint32_t *bc = (const uint32_t *)(blk + 8);

				if (((const uint32_t *)blk)[0]) {
					if (s != my_shard || !nxs->errcode) {
						nxs_decl_err(nxs, NXS_ERR_FATAL, "shard %u failed its pass of the batch", s);
					}
					goto out;	/* every rank sees it: all fail together */
				}
				for (size_t q = 0; q < np; q++) {
					const size_t at = (q * n_shards + s) * cap;
					overflow = overflow || bc[q] > cap;
					cnt_all[q * n_shards + s] = bc[q];
					memcpy(ids_all + at, blk + 8 + cnt_len + q * (size_t)cap * 8, (size_t)cap * 8);
					memcpy(sc_all + at, blk + 8 + cnt_len + np * (size_t)cap * 8 + q * (size_t)cap * 4, (size_t)cap * 4);
				}
			}
		}
		if (!overflow) {
			break;
		}
		if (cap >= (1u << 16) || gathered || my_block) {
			nxs_decl_err(nxs, NXS_ERR_LIMIT, "candidate log overflow");
			goto out;
		}
		cap *= 8;	/* rare: adversarial score orders; try again with room (every rank
				 * sees every count: all ranks retry together) */
	}
	if (np && nxsgpu_merge_candidates(idx0->device, (uint32_t)sp.limit, (uint32_t)np, n_shards, cap,
	    ids_all, sc_all, cnt_all, o_ids, o_sc, o_cnt) != 0) {
		nxs_decl_err(nxs, NXS_ERR_FATAL, "merge failed: %s", nxsgpu_last_error());
		goto out;
	}
	for (size_t q = 0; q < np; q++) {
		total += o_cnt[q];
	}
	if (slab_begin(&sb, n, total) == -1) {
		nxs_decl_err(nxs, NXS_ERR_SYSTEM, "out of memory");
		goto out;
	}
	for (size_t i = 0; i < n; i++) {
		const qprep_t *q = &prep[i];
		nxs_resp_t *rp;

		if (q->errcode) {
			failed++;
			if (errs) {
				errs[i] = q->errcode;
			}
			nxs_decl_err(nxs, q->errcode, "%s", q->errmsg ? q->errmsg : "");
			continue;
		}
		if (q->empty) {
			resps[i] = slab_resp(&sb, i, 0);
			continue;
		}
		rp = slab_resp(&sb, i, o_cnt[plan_of[i]]);
		memcpy(rp->ids, o_ids + (size_t)plan_of[i] * sp.limit, (size_t)rp->count * sizeof(uint64_t));
		memcpy(rp->scores, o_sc + (size_t)plan_of[i] * sp.limit, (size_t)rp->count * sizeof(float));
		resps[i] = rp;
	}
	if (sb.slab && sb.slab->refs == 0) {
		free(sb.slab);
	}
	ret = failed;
out:
	for (size_t i = 0; prep && i < n; i++) {
		nxs_query_release(&prep[i]);
	}
	for (unsigned s = 0; jobs && s < n_local; s++) {
		free(jobs[s].ids); free(jobs[s].sc); free(jobs[s].cnt);
	}
	free(jobs); free(thr); free(sendb); free(recvb);
	free(prep); free(plans); free(plan_of);
	free(ids_all); free(sc_all); free(cnt_all);
	free(o_ids); free(o_sc); free(o_cnt);
	return ret;
}

int
nxs_docshard_search_batch(nxs_index_t *const *shards, unsigned n_shards, nxs_params_t *params,
    const char *const *queries, size_t n, nxs_resp_t **resps, nxs_err_t *errs)
{
	return docshard_search(shards, n_shards, n_shards, 0, params, queries, n, resps, errs, 0, NULL, NULL, NULL);
}

/*
 * One process per shard: make this rank's shard part of the collection.  The
 * communicator is the one nxs_index_shard() attached (rank r holds shard r of
 * `world`); collection-wide df = the all-gathered shards' df arrays, summed, and
 * every impact of the shard is recomputed with it.  Collective.
 */
int
nxs_docshard_attach(nxs_index_t *shard)
{
	nxs_t *nxs = shard->nxs;
	const uint32_t T = shard->last_id;
	const unsigned W = shard->n_shards;
	uint32_t *df = NULL, *all = NULL;
	int ret = -1;

	nxs_clear_error(nxs);
	if (W > 1 && (!shard->comm || nxsgpu_comm_world(shard->comm) != (int)W ||
	    nxsgpu_comm_rank(shard->comm) != (int)shard->shard)) {
		nxs_decl_err(nxs, NXS_ERR_INVALID, "shard %u of %u needs a communicator of %u ranks with itself as "
		    "rank %u (nxs_index_shard)", shard->shard, W, W, shard->shard);
		return -1;
	}
	df = calloc((size_t)T + 2, sizeof(uint32_t));
	all = calloc(((size_t)T + 2) * W, sizeof(uint32_t));
	if (!df || !all) {
		nxs_decl_err(nxs, NXS_ERR_SYSTEM, "out of memory");
		goto out;
	}
	(void)nxsgpu_index_df(shard->dev, df);
	if (W > 1) {
		if (nxsgpu_comm_allgather(shard->comm, df, all, ((size_t)T + 2) * 4) != 0) {
			nxs_decl_err(nxs, NXS_ERR_FATAL, "all-gather of the shards' df failed: %s", nxsgpu_last_error());
			goto out;
		}
		memset(df, 0, ((size_t)T + 2) * 4);
		for (unsigned r = 0; r < W; r++) {
			for (uint32_t t = 1; t <= T; t++) {
				df[t] += all[(size_t)r * (T + 2) + t];
			}
		}
	}
	if (nxsgpu_index_set_global_df(shard->dev, df, T) != 0) {
		nxs_decl_err(nxs, NXS_ERR_SYSTEM, "%s", nxsgpu_last_error());
		goto out;
	}
	shard->global_df_set = true;
	ret = 0;
out:
	free(df);
	free(all);
	return ret;
}

int
nxs_docshard_search_batch_rank(nxs_index_t *shard, nxs_params_t *params,
    const char *const *queries, size_t n, nxs_resp_t **resps, nxs_err_t *errs)
{
	nxs_index_t *local[1] = { shard };

	if (shard->n_shards > 1 && !shard->comm) {
		nxs_clear_error(shard->nxs);
		nxs_decl_err(shard->nxs, NXS_ERR_INVALID, "no communicator attached (nxs_index_shard)");
		return -1;
	}
	return docshard_search(local, 1, shard->n_shards, shard->shard, params, queries, n, resps, errs, 0, NULL, NULL, NULL);
}

#ifdef NXS_TEST_HOOKS
/*
 * Tests (one GPU, no second rank to talk to): the two halves of the rank form.
 * nxs_test_docshard_block() = this rank's candidate block (malloc'ed);
 * nxs_test_docshard_finish() = what every rank does once it holds all blocks.
 * nxs_test_docshard_set_df() stands in for nxs_docshard_attach()'s collective.
 */
int
nxs_test_docshard_block(nxs_index_t *shard, nxs_params_t *params, const char *const *queries, size_t n,
    uint32_t cap, uint8_t **block, size_t *len)
{
	nxs_index_t *local[1] = { shard };
	nxs_resp_t **resps = calloc(n ? n : 1, sizeof(*resps));
	int r;

	if (!resps) {
		return -1;
	}
	*block = NULL;
	*len = 0;
	r = docshard_search(local, 1, shard->n_shards, shard->shard, params, queries, n,
	    resps, NULL, cap, NULL, block, len);
	free(resps);
	return r;
}

int
nxs_test_docshard_finish(nxs_index_t *shard, nxs_params_t *params, const char *const *queries, size_t n,
    uint32_t cap, const uint8_t *gathered, nxs_resp_t **resps, nxs_err_t *errs)
{
	nxs_index_t *local[1] = { shard };

	return docshard_search(local, 1, shard->n_shards, shard->shard, params, queries, n, resps, errs, cap,
	    gathered, NULL, NULL);
}

int
nxs_test_docshard_set_df(nxs_index_t *const *shards, unsigned n_shards)
{
	return docshard_set_global_df(shards, n_shards);
}

#endif /* NXS_TEST_HOOKS */

/* ---- query sharding over the GPUs of a node ---------------------------------------- */

int
nxs_shard_unique_id(nxs_t *nxs, uint8_t *uid)
{
	nxs_clear_error(nxs);
	if (nxsgpu_comm_unique_id(uid) != 0) {
		nxs_decl_err(nxs, NXS_ERR_SYSTEM, "%s", nxsgpu_last_error());
		return -1;
	}
	return 0;
}

int
nxs_index_shard(nxs_index_t *idx, int rank, int world, const uint8_t *uid)
{
	nxs_t *nxs = idx->nxs;

	nxs_clear_error(nxs);
	if (pend_oldest(idx)) {
		nxs_decl_err(nxs, NXS_ERR_INVALID, "batches are in flight");
		return -1;
	}
	if (idx->comm) {
		(void)nxsgpu_index_set_comm(idx->dev, NULL);
		nxsgpu_comm_destroy(idx->comm);
		idx->comm = NULL;
	}
	if (world <= 1 && !uid) {
		return 0;	/* detach */
	}
	idx->comm = nxsgpu_comm_create(idx->device, rank, world, uid);
	if (!idx->comm || nxsgpu_index_set_comm(idx->dev, idx->comm) != 0) {
		nxs_decl_err(nxs, NXS_ERR_SYSTEM, "sharding setup failed: %s", nxsgpu_last_error());
		nxsgpu_comm_destroy(idx->comm);
		idx->comm = NULL;
		return -1;
	}
	return 0;
}

/*
 * The reference scales out by independent worker processes, each answering only ITS OWN requests
 * (compose/nginx.conf:2).  nxs_index_shard_local(idx, true): a rank of a sharded index materialises the
 * responses of its own slice only -- resps[i] stays NULL and errs[i] is left alone for the queries the other
 * ranks own; the return value counts the failures of the own slice.  The collective is unchanged (every rank
 * still sees every block: aborts, the fix-up round and re-sync agreement read all status words), but the
 * per-rank host work per batch is O(n / world) instead of O(n).
 */
int
nxs_index_shard_local(nxs_index_t *idx, bool on)
{
	nxs_clear_error(idx->nxs);
	if (pend_oldest(idx)) {
		nxs_decl_err(idx->nxs, NXS_ERR_INVALID, "batches are in flight");
		return -1;
	}
	idx->shard_local = on;
	return 0;
}

/* the part [*lo, *hi) of an n-query batch whose responses this index delivers (everything unless
 * nxs_index_shard_local is on and a communicator of more than one rank -- or its emulation -- is attached) */
void
nxs_index_shard_slice(const nxs_index_t *idx, size_t n, size_t *lo, size_t *hi)
{
	const int world = idx->comm ? nxsgpu_comm_world(idx->comm) : idx->emu_world;
	const int rank = idx->comm ? nxsgpu_comm_rank(idx->comm) : idx->emu_rank;
	uint64_t a = 0, b = n;

	if (idx->shard_local && world > 1) {
		nxsgpu_shard_slice(n, rank, world, &a, &b);
	}
	*lo = (size_t)a;
	*hi = (size_t)b;
}

/* ---- test hooks (host-only pieces, exercised without a GPU) ------------------------ */
#ifdef NXS_TEST_HOOKS

char *
nxs_test_query_repr(const char *query, char **errmsg)
{
	qparse_t q;
	char *r;

	nxs_query_parse(query, &q);
	r = nxs_query_repr(&q);
	if (errmsg) {
		*errmsg = q.errmsg ? strdup(q.errmsg) : NULL;
	}
	nxs_query_free(&q);
	return r;
}

/*
 * Compile a query against a caller-supplied dictionary (words[i] has term id
 * i+1); unknown words stay unresolved.  Writes the plan; returns the error
 * code (0 = ok), *empty = no live tokens.
 */
int
nxs_test_compile(const char *query, const char *const *words, uint32_t n_words,
    bool lowercase, nxsgpu_query_t *plan, int *empty, char *err, size_t errlen)
{
	nxs_index_t fake = { .lowercase = lowercase };
	qprep_t q;
	int code;

	nxs_query_prepare(&fake, query, &q);
	if (!q.errcode) {
		for (size_t j = 0; j < q.n_tokens; j++) {
			for (uint32_t w = 0; w < n_words; w++) {
				if (strlen(words[w]) == q.tokens[j].len &&
				    memcmp(words[w], q.tokens[j].value, q.tokens[j].len) == 0) {
					q.tokens[j].term_id = w + 1;
					break;
				}
			}
		}
		(void)nxs_query_compile(&q);
	}
	code = q.errcode;
	if (err && errlen) {
		snprintf(err, errlen, "%s", q.errmsg ? q.errmsg : "");
	}
	*plan = q.plan;
	*empty = q.empty;
	nxs_query_release(&q);
	return code;
}

/*
 * The same for queries that take the wide plan: *wide = 1 and the plan's
 * arrays are copied out (term_ids[cap_t], prog[cap_p]); returns the error code.
 */
int
nxs_test_compile_wide(const char *query, const char *const *words, uint32_t n_words,
    int *wide, uint32_t *n_tokens, uint32_t *term_ids, uint32_t cap_t,
    uint32_t *prog_len, uint16_t *prog, uint32_t cap_p)
{
	nxs_index_t fake = { .lowercase = false };
	qprep_t q;
	int code;

	nxs_query_prepare(&fake, query, &q);
	if (!q.errcode) {
		for (size_t j = 0; j < q.n_tokens; j++) {
			/* words are "w<id>" here: resolve by number, not by search */
			const char *v = q.tokens[j].value;
			if (v[0] == 'w') {
				const unsigned long id = strtoul(v + 1, NULL, 10);
				if (id >= 1 && id <= n_words && strcmp(words[id - 1], v) == 0) {
					q.tokens[j].term_id = (uint32_t)id;
				}
			}
		}
		(void)nxs_query_compile(&q);
	}
	code = q.errcode;
	*wide = q.wide;
	*n_tokens = q.wide ? q.wplan.n_tokens : q.plan.n_tokens;
	*prog_len = q.wide ? q.wplan.prog_len : q.plan.prog_len;
	if (q.wide && q.wplan.n_tokens <= cap_t && q.wplan.prog_len <= cap_p) {
		memcpy(term_ids, q.wplan.term_id, q.wplan.n_tokens * sizeof(uint32_t));
		memcpy(prog, q.wplan.prog, q.wplan.prog_len * sizeof(uint16_t));
	}
	nxs_query_release(&q);
	return code;
}

/*
 * Sharding without a second GPU.  nxs_test_shard_emulate(idx, r, W) makes the
 * index play rank r of a W-rank run with the collective left out: the next
 * batch plans and runs rank r's slice, and nxs_test_shard_block() hands out
 * the record block it would have contributed to the all-gather (W = 0: off).
 * nxs_test_pack_record() writes one record + status word into a block (a
 * CPU-side stand-in for the device in the gloo test), and
 * nxs_test_assemble() is the reassembly every rank runs on the gathered blocks.
 */
void
nxs_test_shard_emulate(nxs_index_t *idx, int rank, int world)
{
	idx->emu_rank = rank;
	idx->emu_world = world;
}

size_t
nxs_test_shard_block(nxs_index_t *idx, uint8_t *out, size_t cap)
{
	if (out && idx->emu_block && idx->emu_block_len <= cap) {
		memcpy(out, idx->emu_block, idx->emu_block_len);
	}
	return idx->emu_block ? idx->emu_block_len : 0;
}

void
nxs_test_pack_record(uint8_t *block, uint32_t n_slots, uint32_t k, uint32_t slot,
    uint32_t count, const uint64_t *ids, const float *scores, uint32_t status)
{
	uint8_t *rec = block + (size_t)slot * NXSGPU_REC_BYTES(k);
	uint32_t *st = (uint32_t *)(block + (size_t)n_slots * NXSGPU_REC_BYTES(k));

	((uint32_t *)rec)[0] = count;
	((uint32_t *)rec)[1] = 0;
	memcpy(rec + 8, ids, (size_t)count * 8);
	memcpy(rec + 8 + 8 * (size_t)k, scores, (size_t)count * 4);
	st[slot] = status;
}

/* mark a record "inexact" (candidate overflow: the owner re-runs the query in the fix-up round) */
void
nxs_test_mark_inexact(uint8_t *block, uint32_t n_slots, uint32_t k, uint32_t slot)
{
	(void)n_slots;
	((uint32_t *)(block + (size_t)slot * NXSGPU_REC_BYTES(k)))[1] = NXSGPU_REC_INEXACT;
}

/* the block's flags word says "this rank saw the index files move" ... */
void
nxs_test_mark_changed(uint8_t *block, uint32_t n_slots, uint32_t k)
{
	((uint32_t *)(block + (size_t)n_slots * NXSGPU_REC_BYTES(k)))[n_slots] |= NXSGPU_BLOCK_CHANGED;
}

/* ... and what every rank reads off the gathered blocks: re-sync at the next _begin? */
int
nxs_test_blocks_changed(const uint8_t *blocks, uint32_t world, uint32_t n_slots, uint32_t k)
{
	return blocks_changed(blocks, world, n_slots, k) ? 1 : 0;
}

/* what every rank reads off the gathered blocks: does the batch need a fix-up round, and which
 * of `rank`'s own queries (local indexes) have to be re-run?  -> 1 / 0, *nw set */
int
nxs_test_fixup_scan(const uint8_t *blocks, uint32_t world, uint32_t n_slots, uint32_t k, size_t n,
    int rank, uint32_t *which, size_t *nw)
{
	*nw = 0;
	return fixup_scan(blocks, true, world, rank, n_slots, k, n, which, nw) ? 1 : 0;
}

/* ... and off the blocks of the second all-gather: -1 = fine, else the rank that failed the batch */
int
nxs_test_fixup_verify(const uint8_t *blocks, uint32_t world, uint32_t n_slots, uint32_t k, size_t n)
{
	nxs_err_t acode = NXS_ERR_SUCCESS;
	return fixup_verify(blocks, world, n_slots, k, n, &acode);
}

/* what a rank that cannot do its share contributes instead (STATUS_ABORT) */
void
nxs_test_pack_abort(uint8_t *block, uint32_t n_slots, uint32_t k, uint32_t code)
{
	uint32_t *st = (uint32_t *)(block + (size_t)n_slots * NXSGPU_REC_BYTES(k));

	memset(block, 0, NXSGPU_BLOCK_BYTES(n_slots, k));
	for (uint32_t i = 0; i < n_slots; i++) {
		st[i] = STATUS_ABORT | code;
	}
}

/* the n-th next _begin (which = 0) / exact fix-up round (1) of the index fails; 2: the n-th next
 * fix-up round finds no memory for its receive buffer; 3: the n-th next late second half (late_complete) fails */
void
nxs_test_inject_failure(nxs_index_t *idx, int which, unsigned nth)
{
	if (which == 0) {
		idx->test_fail_begin = nth;
	} else if (which == 1) {
		idx->test_fail_fixup = nth;
	} else if (which == 3) {
		idx->test_fail_late = nth;
	} else {
		idx->test_fail_fixup_recv = nth;
	}
}

/* >= 0: failed queries; -1: error; <= -2: rank (-2 - ret) aborted the batch --
 * every rank sees that and fails the batch, none is left in a collective */
int
nxs_test_assemble(const uint8_t *blocks, uint32_t world, uint32_t n_slots, uint32_t k,
    size_t n, nxs_resp_t **resps, nxs_err_t *errs, int only_rank)
{
	nxs_t fake;
	slab_builder_t sb = { 0 };
	int failed = 0, ar;
	nxs_err_t acode;

	memset(&fake, 0, sizeof(fake));
	for (size_t i = 0; i < n; i++) {
		resps[i] = NULL;
		errs[i] = NXS_ERR_SUCCESS;
	}
	if ((ar = blocks_aborted(blocks, world, n_slots, k, &acode)) >= 0) {
		for (size_t i = 0; i < n; i++) {
			errs[i] = acode;
		}
		return -2 - ar;
	}
	if (resps_from_blocks(&fake, NULL, n, world, n_slots, k, blocks, resps, errs, &sb, &failed, only_rank) == -1) {
		free(fake.errmsg);
		return -1;
	}
	if (sb.slab && sb.slab->refs == 0) {
		free(sb.slab);
	}
	free(fake.errmsg);
	return failed;
}

/* the "normalizer" stage (+ stop words from `basedir`: bit 0 of `stages`, + the
 * English stemmer: bit 1) on one string:
 * -> malloc'd result, NULL if discarded or on error (*act tells which) */
char *
nxs_test_filter(const char *basedir, int stages, const char *s, int *act)
{
	const char *names[3] = { "normalizer" };
	size_t n = 1;
	const char *err = NULL;
	nxs_filters_t *f;
	if (stages & 1) names[n++] = "stopwords";
	if (stages & 2) names[n++] = "stemmer";
	f = nxs_filters_create(basedir, names, n, "en", &err);
	char *val = strdup(s);
	size_t len = strlen(s);

	*act = -2;
	if (!f) {
		free(val);
		return NULL;
	}
	*act = nxs_filters_run(f, &val, &len);
	nxs_filters_destroy(f);
	if (*act != 1) {
		free(val);
		return NULL;
	}
	return val;
}

/* host BK-tree image over a word list (ids 1..n), for structure tests */
int
nxs_test_bk_image(const char *const *words, uint32_t n_words, nxs_bkimage_t *out)
{
	hterm_t *terms = calloc((size_t)n_words + 2, sizeof(hterm_t));
	/* fake "nxsterms" bytes: every term points at a non-zero u64 total */
	static const uint8_t one[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1 };
	int r;

	for (uint32_t i = 0; i < n_words; i++) {
		bool dup = false;
		for (uint32_t j = 0; j < i && !dup; j++) {
			dup = strcmp(words[i], words[j]) == 0;
		}
		terms[i + 1].val = (const uint8_t *)words[i];
		terms[i + 1].len = (uint16_t)strlen(words[i]);
		terms[i + 1].tot_off = dup ? 0 : 8;
	}
	r = nxs_bk_build(terms, n_words, one, out);
	free(terms);
	return r;
}

int
nxs_test_levdist(const uint8_t *a, size_t n, const uint8_t *b, size_t m)
{
	extern int nxs_levdist_export(const uint8_t *, size_t, const uint8_t *, size_t);
	return nxs_levdist_export(a, n, b, m);
}
#endif /* NXS_TEST_HOOKS */

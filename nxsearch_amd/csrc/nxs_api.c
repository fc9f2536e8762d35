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

#include "nxs_impl.h"

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
    const char *dtmap_path, int algo, bool lowercase)
{
	nxs_index_t *idx = calloc(1, sizeof(nxs_index_t)), **list;

	if (!idx) {
		nxs_decl_err(nxs, NXS_ERR_SYSTEM, "out of memory");
		return NULL;
	}
	idx->nxs = nxs;
	idx->algo = algo;
	idx->lowercase = lowercase;
	idx->name = strdup(name);
	if (nxs_index_load(idx, terms_path, dtmap_path) == -1) {
		nxs_index_unload(idx);
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

nxs_index_t *
nxs_index_open(nxs_t *nxs, const char *name)
{
	char *ppath = NULL, *tpath = NULL, *dpath = NULL, *json = NULL, *algo_name = NULL;
	nxs_index_t *idx = NULL;
	struct stat sb;
	bool lowercase;
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
	/* only the ASCII lower-casing of the "normalizer" filter is provided */
	lowercase = strstr(json, "\"normalizer\"") != NULL;
	idx = index_open_common(nxs, name, tpath, dpath, algo, lowercase);
out:
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
	return index_open_common(nxs, terms_path, terms_path, dtmap_path, algo, lowercase);
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
	nxs_index_unload(idx);
	free(idx->name);
	free(idx);
}

struct nxsgpu_index *
nxs_index_device(nxs_index_t *idx)
{
	return idx->dev;
}

/* ---- response object --------------------------------------------------------- */

struct nxs_resp {
	nxs_doc_id_t *	ids;
	float *		scores;
	unsigned	count;
	unsigned	iter;
};

static nxs_resp_t *
resp_create(const uint64_t *ids, const float *scores, unsigned count)
{
	nxs_resp_t *r = calloc(1, sizeof(nxs_resp_t));

	if (!r) {
		return NULL;
	}
	r->ids = malloc((count ? count : 1) * sizeof(nxs_doc_id_t));
	r->scores = malloc((count ? count : 1) * sizeof(float));
	if (count) {
		memcpy(r->ids, ids, count * sizeof(nxs_doc_id_t));
		memcpy(r->scores, scores, count * sizeof(float));
	}
	r->count = count;
	return r;
}

void
nxs_resp_release(nxs_resp_t *r)
{
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
 * Front half of a batch: parse, build the token sets, resolve (exact on the
 * host, misses through one device BK-tree pass), compile the device plans.
 * prep[i].errcode / .empty tell how query i ended.
 */
static int
plan_batch(nxs_index_t *idx, const search_params_t *sp, const char *const *queries,
    size_t n, qprep_t *prep)
{
	nxs_t *nxs = idx->nxs;
	uint32_t *fz_q = NULL, *fz_t = NULL, *fz_off = NULL, *fz_ids = NULL;
	uint8_t *fz_bytes = NULL;
	size_t n_fz = 0, fz_len = 0;
	int ret = -1;

	/* idxterm_lookup for every token */
	for (size_t i = 0; i < n; i++) {
		qprep_t *q = &prep[i];

		nxs_query_prepare(idx, queries[i], q);
		if (q->errcode) {
			continue;
		}
		for (size_t j = 0; j < q->n_tokens; j++) {
			qtok_t *t = &q->tokens[j];
			t->term_id = nxs_term_lookup(idx, (const uint8_t *)t->value, t->len);
			if (!t->term_id && sp->fuzzymatch) {
				n_fz++;
				fz_len += t->len;
			}
		}
	}

	/* one device BK-tree pass for every token that missed (tokenizer.c:177-180) */
	if (n_fz) {
		size_t k = 0, o = 0;

		fz_q = malloc(n_fz * sizeof(uint32_t));
		fz_t = malloc(n_fz * sizeof(uint32_t));
		fz_off = malloc((n_fz + 1) * sizeof(uint32_t));
		fz_ids = calloc(n_fz, sizeof(uint32_t));
		fz_bytes = malloc(fz_len + 1);
		for (size_t i = 0; i < n; i++) {
			qprep_t *q = &prep[i];
			if (q->errcode) {
				continue;
			}
			for (size_t j = 0; j < q->n_tokens; j++) {
				const qtok_t *t = &q->tokens[j];
				if (t->term_id) {
					continue;
				}
				fz_q[k] = (uint32_t)i;
				fz_t[k] = (uint32_t)j;
				fz_off[k] = (uint32_t)o;
				memcpy(fz_bytes + o, t->value, t->len);
				o += t->len;
				k++;
			}
		}
		fz_off[k] = (uint32_t)o;
		if (nxsgpu_fuzzy(idx->dev, fz_bytes, fz_off, (uint32_t)n_fz, fz_ids, NULL) != 0) {
			nxs_decl_err(nxs, NXS_ERR_FATAL, "device fuzzy search failed: %s",
			    nxsgpu_last_error());
			goto out;
		}
		for (k = 0; k < n_fz; k++) {
			prep[fz_q[k]].tokens[fz_t[k]].term_id = fz_ids[k];
		}
	}
	for (size_t i = 0; i < n; i++) {
		if (!prep[i].errcode) {
			(void)nxs_query_compile(&prep[i]);
		}
	}
	ret = 0;
out:
	free(fz_q);
	free(fz_t);
	free(fz_off);
	free(fz_ids);
	free(fz_bytes);
	return ret;
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
			if (prep[i].errcode) {
				failed++;
				nxs_decl_err(idx->nxs, prep[i].errcode, "%s",
				    prep[i].errmsg ? prep[i].errmsg : "");
			} else if (!prep[i].empty) {
				plans[i] = prep[i].plan;
			}
			if (errs) {
				errs[i] = prep[i].errcode;
			}
		}
		nxs_query_release(&prep[i]);
	}
	free(prep);
	return failed;
}

int
nxs_index_search_batch(nxs_index_t *idx, nxs_params_t *params,
    const char *const *queries, size_t n, nxs_resp_t **resps, nxs_err_t *errs)
{
	nxs_t *nxs = idx->nxs;
	search_params_t sp;
	qprep_t *prep = NULL;
	nxsgpu_query_t *plans = NULL;
	uint32_t *plan_of = NULL;
	size_t n_plans = 0;
	nxsgpu_results_t res;
	int failed = 0, ret = -1;

	nxs_clear_error(nxs);
	memset(&res, 0, sizeof(res));
	for (size_t i = 0; i < n; i++) {
		resps[i] = NULL;
		if (errs) {
			errs[i] = NXS_ERR_SUCCESS;
		}
	}
	if (get_search_params(idx, params, &sp) == -1) {
		return -1;
	}
	if (nxs_index_refresh(idx) == -1) {	/* search.c:309-312 */
		return -1;
	}
	if (n == 0) {
		return 0;
	}
	if (n > UINT32_MAX / 2) {
		nxs_decl_err(nxs, NXS_ERR_LIMIT, "batch too large");
		return -1;
	}
	prep = calloc(n, sizeof(qprep_t));
	plans = calloc(n, sizeof(nxsgpu_query_t));
	plan_of = calloc(n, sizeof(uint32_t));
	if (!prep || !plans || !plan_of) {
		nxs_decl_err(nxs, NXS_ERR_SYSTEM, "out of memory");
		goto out;
	}
	if (plan_batch(idx, &sp, queries, n, prep) == -1) {
		goto out;
	}
	for (size_t i = 0; i < n; i++) {
		const qprep_t *q = &prep[i];

		if (!q->errcode && !q->empty) {
			plan_of[i] = (uint32_t)n_plans;
			plans[n_plans++] = q->plan;
		}
	}
	if (n_plans && nxsgpu_search(idx->dev, sp.algo, sp.limit, plans,
	    (uint32_t)n_plans, &res) != 0) {
		nxs_decl_err(nxs, NXS_ERR_FATAL, "device search failed: %s",
		    nxsgpu_last_error());
		goto out;
	}

	for (size_t i = 0; i < n; i++) {
		qprep_t *q = &prep[i];

		if (q->errcode) {
			failed++;
			if (errs) {
				errs[i] = q->errcode;
			}
			nxs_decl_err(nxs, q->errcode, "%s", q->errmsg ? q->errmsg : "");
			continue;
		}
		if (q->empty) {
			resps[i] = resp_create(NULL, NULL, 0);
		} else {
			const uint32_t p = plan_of[i];
			resps[i] = resp_create(res.doc_ids + res.offsets[p],
			    res.scores + res.offsets[p], res.counts[p]);
		}
		if (!resps[i]) {
			failed++;
			if (errs) {
				errs[i] = NXS_ERR_SYSTEM;
			}
			nxs_decl_err(nxs, NXS_ERR_SYSTEM, "out of memory");
		}
	}
	ret = failed;
out:
	if (res.counts) {
		nxsgpu_results_free(&res);
	}
	for (size_t i = 0; prep && i < n; i++) {
		nxs_query_release(&prep[i]);
	}
	if (ret == -1) {
		for (size_t i = 0; i < n; i++) {
			if (resps[i]) {
				nxs_resp_release(resps[i]);
				resps[i] = NULL;
			}
		}
	}
	free(prep);
	free(plans);
	free(plan_of);
	return ret;
}

/* nxs_index_search: search.c:285-342 (one query = a batch of one) */
nxs_resp_t *
nxs_index_search(nxs_index_t *idx, nxs_params_t *params, const char *query, size_t len)
{
	nxs_resp_t *resp = NULL;
	const char *qv[1] = { query };

	(void)len;	/* the reference's lexer stops at the NUL byte too (search.c:177) */
	if (nxs_index_search_batch(idx, params, qv, 1, &resp, NULL) != 0) {
		if (resp) {
			nxs_resp_release(resp);
		}
		return NULL;
	}
	return resp;
}

/* ---- test hooks (host-only pieces, exercised without a GPU) ------------------------ */

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

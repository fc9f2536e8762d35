/*
 * nxs_oracle.h -- CPU oracle for the nxsearch query/ranking hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference
 * algorithm (rmind/nxsearch @ 2024-11-15) used as the parity checker by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing
 * in the product (nxsearch_amd/, include/) may include, link or call it.
 *
 * Parity pinning: checked against the reference's own golden vectors
 * (t_levdist.c, t_bktree.c, t_scoring.c, t_querylogic.c, t_queryparser.c,
 * t_misc.c, t_index_terms.c, t_index_dtmap.c; see tests/golden/) and against
 * the genuine reference algo/{levdist,bktree,deque,heap}.c compiled into
 * oracle/_ref/ (see oracle/Makefile).  ranking.c / search.c / results.c
 * cannot be built here (they need CRoaring, rhashmap, yyjson, lemon, re2c,
 * which are absent), so those are pinned by the known-answer tables only.
 */
#ifndef NXS_ORACLE_H
#define NXS_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ranking_algo_t values: reference src/index/index.h:30-34 */
enum { ORC_TF_IDF = 0, ORC_BM25 = 1 };

/* nxs_err_t values: reference src/core/nxs.h:35-46 */
enum {
	ORC_ERR_SUCCESS = 0, ORC_ERR_FATAL, ORC_ERR_SYSTEM, ORC_ERR_INVALID,
	ORC_ERR_EXISTS, ORC_ERR_MISSING, ORC_ERR_LIMIT,
};

typedef struct { uint64_t doc_id; float score; } orc_result_t;

/* -- pure functions ---------------------------------------------------- */

/* levdist(): reference src/algo/levdist.c:67-150 */
int	orc_levdist(const char *s1, size_t n, const char *s2, size_t m);

/* bm25(): reference src/algo/ranking.c:99-176 (returns -1 to skip) */
float	orc_bm25(int term_freq, int doc_len, uint32_t doc_count,
	    uint64_t token_count, uint64_t doc_freq);

/* tf_idf(): reference src/algo/ranking.c:41-97 */
float	orc_tf_idf(int term_freq, uint32_t doc_count, uint64_t doc_freq);

/*
 * Top-k with the reference's capped min-heap + heapsort (src/algo/heap.c:58-221)
 * and comparator results.c:165-176.  Items are fed in the given order (the
 * reference feeds descending doc id, results.c:128-150,182-193).
 * Returns the number of results written (min(cap, n)).
 */
size_t	orc_topk(const uint64_t *ids, const float *scores, size_t n,
	    size_t cap, uint64_t *out_ids, float *out_scores);

/*
 * Query parser: hand restatement of src/query/scan.re:43-121 (lexer) and
 * src/query/grammar.y:66-120 (grammar).  Returns the IR dump in the format of
 * src/tests/t_queryparser.c:146-169, or NULL on syntax error (*errmsg set,
 * query.c:46-58 format).  Both strings are malloc'ed.
 */
char *	orc_query_repr(const char *query, char **errmsg);

/* Lexer only: writes token kinds (1=AND 2=OR 3=NOT 4=( 5=) 6=FF 7=QUOTED). */
int	orc_query_lex(const char *query, int *kinds, size_t cap);

/* -- standalone BK-tree over words (src/algo/bktree.c:160-275) ---------- */

typedef struct orc_bkt orc_bkt_t;

orc_bkt_t *orc_bkt_create(void);
void	orc_bkt_destroy(orc_bkt_t *);
/* returns 0, or -1 if rejected (duplicate, bktree.c:182-189) */
int	orc_bkt_insert(orc_bkt_t *, const char *word, size_t len);
/*
 * BFS search; writes the insertion indices (0-based, counting accepted
 * and rejected inserts alike) of matches in deque push order.  *ndist gets
 * the number of distance evaluations (visited nodes).
 */
size_t	orc_bkt_search(orc_bkt_t *, unsigned tolerance, const char *word,
	    size_t len, uint32_t *out, size_t cap, uint64_t *ndist);

/* -- index + search ------------------------------------------------------ */

typedef struct orc_index orc_index_t;

/* idx_terms_open/sync + idx_dtmap_open/sync: terms.c:83-135,320-414; dtmap.c:94-147,440-544 */
orc_index_t *orc_index_load(const char *terms_path, const char *dtmap_path,
	    char *err, size_t errlen);
void	orc_index_free(orc_index_t *);
/* ASCII lower-casing of query tokens (stand-in for the "normalizer" filter) */
void	orc_index_set_lowercase(orc_index_t *, bool);
/* the `stemmer' filter (filters_builtin.c:203-245) on query tokens, lang "en": orc_stem_en.c */
void	orc_index_set_stemmer(orc_index_t *, bool);
/* sb_stemmer_stem() of libstemmer's English stemmer, restated (orc_stem_en.c); `out` holds len + 2 bytes */
size_t	orc_stem_en(const char *in, size_t len, char *out, size_t cap);

uint32_t orc_index_term_count(const orc_index_t *);
uint64_t orc_index_dt_count(const orc_index_t *);	/* live docs loaded */
uint32_t orc_index_doc_count(const orc_index_t *);	/* header doc_count */
uint64_t orc_index_token_count(const orc_index_t *);	/* header token_count */

/* idxterm_lookup (idxterm.c:192-196): term id or 0 */
uint32_t orc_index_lookup(const orc_index_t *, const char *tok, size_t len);
/* idxterm_fuzzysearch (idxterm.c:210-249): term id or 0; *visited = BK nodes evaluated */
uint32_t orc_index_fuzzy(const orc_index_t *, const char *tok, size_t len,
	    uint64_t *visited);
uint64_t orc_index_df(const orc_index_t *, uint32_t term_id);
/* term bytes by id (NULL if unknown) */
const char *orc_index_term(const orc_index_t *, uint32_t term_id, size_t *len);
/* per (term, doc) score straight through idxdoc_get_termcount + ranking func */
float	orc_index_score(const orc_index_t *, int algo, uint32_t term_id,
	    uint64_t doc_id);

/*
 * nxs_index_search(): search.c:285-342 with run_query_logic (210-278),
 * get_expr_bitmap (118-174), nxs_resp_addresult/build (results.c:128-220).
 * Returns 0 and fills out[0..*count) or -1 with *errcode/errmsg set.
 * `limit` as the "limit" param (0 or > UINT_MAX is NXS_ERR_INVALID).
 */
int	orc_search(orc_index_t *, const char *query, int algo, uint64_t limit,
	    bool fuzzymatch, orc_result_t *out, size_t cap, uint32_t *count,
	    int *errcode, char *errmsg, size_t errlen);

/* number of (doc, term) pairs scored by the last orc_search (work measure) */
uint64_t orc_last_pairs(const orc_index_t *);

/* nxs_resp_tojson(): results.c:118-122,153-161,218 */
char *	orc_results_json(const orc_result_t *res, size_t n);

#ifdef __cplusplus
}
#endif
#endif

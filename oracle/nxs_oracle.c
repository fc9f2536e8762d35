/*
 * nxs_oracle.c -- CPU oracle for the nxsearch query/ranking hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see nxs_oracle.h).  Plain C11 restatement of
 * the reference algorithm, function by function, keeping the reference's
 * per-query structure (result-set iteration in ascending doc id, per-pair
 * membership test, hash lookup of the doc, binary search for tf inside the
 * doc block, two log() calls, hash-map score accumulation, pointer min-heap
 * top-k; BFS BK-tree walk with a row-DP Levenshtein per visited node).
 * Every function cites the reference file:line it follows.  The roaring64
 * bitmaps of the reference are replaced by sorted uint64_t vectors (pure set
 * semantics + ascending iteration: search.c:138-171,235-276) and rhashmap by
 * a local open-addressing map (lookup only, never iterated).
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <stdarg.h>
#include <limits.h>
#include <errno.h>
#include <math.h>
#include <endian.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/stat.h>

#include "nxs_oracle.h"

#define	MIN(x, y)	((x) < (y) ? (x) : (y))
#define	MAX(x, y)	((x) > (y) ? (x) : (y))

/* ------------------------------------------------------------------ */
/* Small containers (replacing rhashmap / deque; order-only semantics) */
/* ------------------------------------------------------------------ */

static uint64_t
hash_bytes(const void *key, size_t len)
{
	const uint8_t *p = key;
	uint64_t h = 0xcbf29ce484222325ULL;

	for (size_t i = 0; i < len; i++) {
		h = (h ^ p[i]) * 0x100000001b3ULL;
	}
	h ^= h >> 32;
	h *= 0xd6e8feb86659fd93ULL;
	h ^= h >> 32;
	return h;
}

/* byte-string keyed map; keys are borrowed (RHM_NOCOPY semantics) */
typedef struct { const void *key; size_t len; void *val; } smap_ent_t;
typedef struct { smap_ent_t *e; size_t cap, n; } smap_t;

static void
smap_init(smap_t *m)
{
	m->cap = 64;
	m->n = 0;
	m->e = calloc(m->cap, sizeof(smap_ent_t));
}

static void *
smap_get(const smap_t *m, const void *key, size_t len)
{
	size_t i = hash_bytes(key, len) & (m->cap - 1);

	while (m->e[i].key) {
		if (m->e[i].len == len && memcmp(m->e[i].key, key, len) == 0) {
			return m->e[i].val;
		}
		i = (i + 1) & (m->cap - 1);
	}
	return NULL;
}

/* returns the already-present value if the key exists, else val (rhashmap_put) */
static void *
smap_put(smap_t *m, const void *key, size_t len, void *val)
{
	size_t i;

	if ((m->n + 1) * 2 > m->cap) {
		smap_t nm = { .cap = m->cap * 2, .n = 0 };
		nm.e = calloc(nm.cap, sizeof(smap_ent_t));
		for (size_t j = 0; j < m->cap; j++) {
			if (m->e[j].key) {
				smap_put(&nm, m->e[j].key, m->e[j].len,
				    m->e[j].val);
			}
		}
		free(m->e);
		*m = nm;
	}
	i = hash_bytes(key, len) & (m->cap - 1);
	while (m->e[i].key) {
		if (m->e[i].len == len && memcmp(m->e[i].key, key, len) == 0) {
			return m->e[i].val;
		}
		i = (i + 1) & (m->cap - 1);
	}
	m->e[i].key = key;
	m->e[i].len = len;
	m->e[i].val = val;
	m->n++;
	return val;
}

/* u64 -> u64 map; key 0 is reserved (doc id 0 never exists: nxs.c:498-502) */
typedef struct { uint64_t *k, *v; size_t cap, n; } umap_t;

static inline uint64_t
mix64(uint64_t x)
{
	x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
	x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
	x ^= x >> 33;
	return x;
}

static void
umap_init(umap_t *m, size_t hint)
{
	m->cap = 64;
	while (m->cap < hint * 2) {
		m->cap <<= 1;
	}
	m->n = 0;
	m->k = calloc(m->cap, sizeof(uint64_t));
	m->v = calloc(m->cap, sizeof(uint64_t));
}

static void
umap_free(umap_t *m)
{
	free(m->k);
	free(m->v);
	m->k = m->v = NULL;
}

static bool
umap_get(const umap_t *m, uint64_t key, uint64_t *val)
{
	size_t i = mix64(key) & (m->cap - 1);

	while (m->k[i]) {
		if (m->k[i] == key) {
			*val = m->v[i];
			return true;
		}
		i = (i + 1) & (m->cap - 1);
	}
	return false;
}

static uint64_t *
umap_slot(umap_t *m, uint64_t key, bool *isnew)
{
	size_t i;

	if ((m->n + 1) * 2 > m->cap) {
		umap_t nm;
		bool dummy;
		umap_init(&nm, m->cap);
		for (size_t j = 0; j < m->cap; j++) {
			if (m->k[j]) {
				*umap_slot(&nm, m->k[j], &dummy) = m->v[j];
			}
		}
		umap_free(m);
		*m = nm;
	}
	i = mix64(key) & (m->cap - 1);
	while (m->k[i]) {
		if (m->k[i] == key) {
			*isnew = false;
			return &m->v[i];
		}
		i = (i + 1) & (m->cap - 1);
	}
	m->k[i] = key;
	m->n++;
	*isnew = true;
	return &m->v[i];
}

static void
umap_del(umap_t *m, uint64_t key)
{
	size_t i = mix64(key) & (m->cap - 1), j;

	while (m->k[i] && m->k[i] != key) {
		i = (i + 1) & (m->cap - 1);
	}
	if (!m->k[i]) {
		return;
	}
	/* backward-shift deletion */
	m->k[i] = 0;
	m->n--;
	j = i;
	for (;;) {
		j = (j + 1) & (m->cap - 1);
		if (!m->k[j]) {
			break;
		}
		size_t home = mix64(m->k[j]) & (m->cap - 1);
		if ((i <= j) ? (home <= i || home > j) : (home <= i && home > j)) {
			m->k[i] = m->k[j];
			m->v[i] = m->v[j];
			m->k[j] = 0;
			i = j;
		}
	}
}

/* FIFO/LIFO pointer queue: only the order of deque.c:25-132 is semantic */
typedef struct { void **e; size_t head, tail, cap; } pq_t;

static void
pq_push(pq_t *q, void *p)
{
	if (q->tail == q->cap) {
		q->cap = q->cap ? q->cap * 2 : 64;
		q->e = realloc(q->e, q->cap * sizeof(void *));
	}
	q->e[q->tail++] = p;
}

static void *
pq_pop_front(pq_t *q)
{
	return (q->head < q->tail) ? q->e[q->head++] : NULL;
}

static void *
pq_pop_back(pq_t *q)
{
	return (q->head < q->tail) ? q->e[--q->tail] : NULL;
}

/* ------------------------------------------------------------------ */
/* Levenshtein distance: reference src/algo/levdist.c:67-150           */
/* ------------------------------------------------------------------ */

static __thread uint16_t *lev_row;
static __thread unsigned lev_rlen;

int
orc_levdist(const char *s1, size_t n, const char *s2, size_t m)
{
	unsigned rlen, prev_diag, prev_above;
	uint16_t *row;

	/* levdist.c:73-79: longer string is the outer loop; empty shortcuts */
	if (n < m) {
		return orc_levdist(s2, m, s1, n);
	}
	if (n == 0)
		return m;
	if (m == 0)
		return n;

	rlen = m + 1;	/* levdist.c:86 */
	if (rlen > lev_rlen) {
		lev_row = realloc(lev_row, sizeof(uint16_t) * rlen);
		lev_rlen = rlen;
	}
	row = lev_row;

	for (unsigned j = 0; j < rlen; j++) {	/* levdist.c:109-111 */
		row[j] = j;
	}
	for (unsigned i = 0; i < n; i++) {	/* levdist.c:113-147 */
		const char s1c = s1[i];

		row[0] = i + 1;
		prev_above = i;

		for (unsigned j = 1; j <= m; j++) {
			const char s2c = s2[j - 1];
			const unsigned cost = !(s1c == s2c);
			unsigned a, b, c, v;

			prev_diag = prev_above;
			prev_above = row[j];

			a = row[j - 1] + 1;
			b = prev_above + 1;
			c = prev_diag + cost;
			v = MIN(MIN(a, b), c);
			row[j] = v;	/* uint16_t store, as in the reference */
		}
	}
	return row[m];
}

/* ------------------------------------------------------------------ */
/* Ranking: reference src/algo/ranking.c                               */
/* ------------------------------------------------------------------ */

/* ranking.c:41-97 */
float
orc_tf_idf(int term_freq, uint32_t doc_count_, uint64_t doc_freq_)
{
	unsigned long doc_freq = doc_freq_, doc_count = doc_count_;
	float tf, idf;

	if (term_freq <= 0 || doc_count == 0) {	/* ranking.c:86-88 */
		return -1;
	}
	tf = log(term_freq + 1);			/* ranking.c:90 */
	idf = log((float)doc_count / doc_freq) + 1;	/* ranking.c:91 */
	return tf * idf;				/* ranking.c:96 */
}

/* ranking.c:99-176 */
float
orc_bm25(int term_freq, int doc_len, uint32_t doc_count_,
    uint64_t token_count, uint64_t doc_freq_)
{
	static const double k = 1.2f;	/* ranking.c:141 */
	static const double b = 0.75f;	/* ranking.c:142 */
	unsigned long doc_freq = doc_freq_, doc_count = doc_count_;
	double tf, dl, adl, tf_bm25, idf_bm25;

	if (term_freq <= 0 || doc_count == 0) {	/* ranking.c:156-158 */
		return -1;
	}
	adl = token_count / doc_count;		/* ranking.c:163: INTEGER division */
	if (adl < 1) {
		return -1;
	}
	tf = log(term_freq + 1);		/* ranking.c:168 */
	dl = doc_len;				/* ranking.c:169 */
	tf_bm25 = tf / (tf + k * (1 - b + b * dl / adl));	/* ranking.c:170 */
	idf_bm25 = log(((doc_count - doc_freq + 0.5) / (doc_freq + 0.5)) + 1);
	return tf_bm25 * idf_bm25;		/* ranking.c:175 */
}

/* ------------------------------------------------------------------ */
/* Capped min-heap: reference src/algo/heap.c                          */
/* ------------------------------------------------------------------ */

typedef struct result_entry {
	uint64_t		doc_id;
	float			score;
	struct result_entry *	next;
} result_entry_t;	/* results.c:22-26 */

/* results.c:165-176 */
static int
result_entry_cmp(const result_entry_t *e1, const result_entry_t *e2)
{
	if (e1->score < e2->score)
		return -1;
	if (e1->score > e2->score)
		return 1;
	return 0;
}

typedef struct {
	size_t		cap, nitems;
	result_entry_t **items;
} heap_t;

#define	HEAP_PARENT(i)		(((i) - 1) / 2)
#define	HEAP_LEFT_NODE(i)	((i) * 2 + 1)
#define	HEAP_RIGHT_NODE(i)	((i) * 2 + 2)

/* heap.c:133-189 */
static result_entry_t *
heap_remove_min(heap_t *h)
{
	size_t i, max, left_idx;
	result_entry_t *item;

	if (h->nitems == 0) {
		return NULL;
	}
	i = 0;
	item = h->items[i];
	if ((max = --h->nitems) == 0) {
		h->items[i] = NULL;
		return item;
	}
	h->items[i] = h->items[max];
	h->items[max] = NULL;

	while ((left_idx = HEAP_LEFT_NODE(i)) < max) {
		result_entry_t *parent = h->items[i];
		const size_t right_idx = HEAP_RIGHT_NODE(i);
		size_t smallest_idx = i;

		if (result_entry_cmp(h->items[left_idx], parent) < 0) {
			smallest_idx = left_idx;
		}
		if (right_idx < max) {
			const result_entry_t *smallest = h->items[smallest_idx];

			if (result_entry_cmp(h->items[right_idx], smallest) < 0) {
				smallest_idx = right_idx;
			}
		}
		if (smallest_idx == i) {
			break;
		}
		h->items[i] = h->items[smallest_idx];
		h->items[smallest_idx] = parent;
		i = smallest_idx;
	}
	return item;
}

/* heap.c:58-124 */
static bool
heap_add(heap_t *h, result_entry_t *item)
{
	size_t i;

	if (h->nitems == h->cap) {
		result_entry_t *root = h->items[0];

		if (result_entry_cmp(item, root) <= 0) {	/* heap.c:72 */
			return false;
		}
		heap_remove_min(h);
	}
	i = h->nitems++;
	h->items[i] = item;

	while (i) {
		const size_t parent_idx = HEAP_PARENT(i);
		result_entry_t *parent = h->items[parent_idx];

		if (result_entry_cmp(item, parent) >= 0) {	/* heap.c:103 */
			break;
		}
		h->items[parent_idx] = item;
		h->items[i] = parent;
		i = parent_idx;
	}
	return true;
}

/* heap.c:197-221 */
static result_entry_t **
heap_sort(heap_t *h, size_t *nitems)
{
	*nitems = h->nitems;
	while (h->nitems) {
		const size_t last_idx = h->nitems - 1;
		result_entry_t *min_item = heap_remove_min(h);
		h->items[last_idx] = min_item;
	}
	return h->items;
}

size_t
orc_topk(const uint64_t *ids, const float *scores, size_t n, size_t cap,
    uint64_t *out_ids, float *out_scores)
{
	result_entry_t *ents = calloc(n ? n : 1, sizeof(result_entry_t));
	heap_t h = { .cap = cap, .nitems = 0 };
	result_entry_t **top;
	size_t cnt;

	h.items = calloc(MIN(cap, n) + 1, sizeof(void *));
	for (size_t i = 0; i < n; i++) {
		ents[i].doc_id = ids[i];
		ents[i].score = scores[i];
		heap_add(&h, &ents[i]);
	}
	top = heap_sort(&h, &cnt);
	for (size_t i = 0; i < cnt; i++) {
		out_ids[i] = top[i]->doc_id;
		out_scores[i] = top[i]->score;
	}
	free(h.items);
	free(ents);
	return cnt;
}

/* ------------------------------------------------------------------ */
/* BK-tree: reference src/algo/bktree.c                                */
/* ------------------------------------------------------------------ */

#define	BKT_DIST_LIMIT	63	/* bktree.h:11 */

typedef int (*bk_distfunc_t)(void *, const void *, const void *);

typedef struct bknode {
	const void *	obj;
	uint64_t	bitmap;
	struct bknode *	map[];
} bknode_t;	/* bktree.c:54-58 */

typedef struct {
	bknode_t *	root;
	bk_distfunc_t	distfunc;
	void *		distctx;
	uint64_t	ndist;		/* instrumentation only */
} bktree_t;

/* bktree.c:79-98 */
static bknode_t *
bknode_get(bknode_t *node, unsigned i)
{
	const uint64_t bitmap = node->bitmap;
	const uint64_t bit = UINT64_C(1) << i;

	if ((bitmap & bit) == 0) {
		return NULL;
	}
	return node->map[__builtin_popcountll(bitmap & (bit - 1))];
}

/* bktree.c:123-148: the node is re-allocated with one more child slot */
static bknode_t *
bknode_set(bknode_t *cur, unsigned i, bknode_t *val)
{
	const uint64_t bit = UINT64_C(1) << i;
	const unsigned nitems = __builtin_popcountll(cur->bitmap);
	const unsigned slot = __builtin_popcountll(cur->bitmap & (bit - 1));
	bknode_t *node;

	node = malloc(sizeof(bknode_t) + sizeof(bknode_t *) * (nitems + 1));
	node->obj = cur->obj;
	node->bitmap = cur->bitmap | bit;
	memcpy(&node->map[0], &cur->map[0], sizeof(bknode_t *) * slot);
	node->map[slot] = val;
	memcpy(&node->map[slot + 1], &cur->map[slot],
	    sizeof(bknode_t *) * (nitems - slot));
	free(cur);
	return node;
}

/* bktree.c:150-156 (x86 shift semantics for the out-of-range UB case) */
static uint64_t
bknode_get_range(const bknode_t *node, unsigned start, unsigned end)
{
	const uint64_t lo_mask = ~UINT64_C(0) << (start & 63);
	const uint64_t hi_mask = ~UINT64_C(0) >> ((64 - end) & 63);
	return node->bitmap & (lo_mask & hi_mask);
}

/* bktree.c:160-217 */
static int
bktree_insert(bktree_t *bkt, const void *obj)
{
	bknode_t *new_node, *node, *child, **pp;
	int d;

	new_node = calloc(1, sizeof(bknode_t));
	new_node->obj = obj;
	if ((node = bkt->root) == NULL) {
		bkt->root = new_node;
		return 0;
	}
	pp = &bkt->root;
desc:
	d = bkt->distfunc(bkt->distctx, obj, node->obj);
	if (d <= 0) {		/* bktree.c:182-189: duplicate => rejected */
		free(new_node);
		return -1;
	}
	d = MIN((unsigned)d, BKT_DIST_LIMIT);	/* bktree.c:196 */

	if ((child = bknode_get(node, d)) != NULL) {
		const uint64_t bit = UINT64_C(1) << d;
		pp = &node->map[__builtin_popcountll(node->bitmap & (bit - 1))];
		node = child;
		goto desc;
	}
	node = bknode_set(node, d, new_node);
	*pp = node;
	return 0;
}

/* bktree.c:219-275; results receive node->obj in deque push order */
static int
bktree_search(bktree_t *bkt, unsigned tolerance, const void *obj, pq_t *results)
{
	bknode_t *node;
	pq_t dq = { 0 };

	if ((node = bkt->root) == NULL) {
		return 0;
	}
	pq_push(&dq, node);

	while ((node = pq_pop_front(&dq)) != NULL) {	/* FIFO: bktree.c:241 */
		unsigned i, min_d, max_d;
		uint64_t bitmap;
		int d;

		d = bkt->distfunc(bkt->distctx, obj, node->obj);
		bkt->ndist++;
		if (d < 0) {
			free(dq.e);
			return -1;
		}
		if ((unsigned)d <= tolerance) {
			pq_push(results, (void *)(uintptr_t)node->obj);
		}
		min_d = MAX((int)d - (int)tolerance, 0);	/* bktree.c:260 */
		max_d = MIN(d + tolerance, BKT_DIST_LIMIT);	/* bktree.c:261 */

		/* half-open [min_d, max_d): bktree.c:150-156,264 */
		bitmap = bknode_get_range(node, min_d, max_d);
		while ((i = __builtin_ffsll(bitmap)) != 0) {
			bknode_t *child = bknode_get(node, --i);
			pq_push(&dq, child);
			bitmap &= ~(UINT64_C(1) << i);
		}
	}
	free(dq.e);
	return 0;
}

static void
bktree_free_nodes(bktree_t *bkt)
{
	pq_t dq = { 0 };
	bknode_t *node;

	if (bkt->root) {
		pq_push(&dq, bkt->root);
	}
	while ((node = pq_pop_front(&dq)) != NULL) {
		const unsigned nitems = __builtin_popcountll(node->bitmap);
		for (unsigned i = 0; i < nitems; i++) {
			pq_push(&dq, node->map[i]);
		}
		free(node);
	}
	free(dq.e);
	bkt->root = NULL;
}

/* standalone word tree (t_bktree.c:15-19 style distance callback) */
typedef struct { char *s; size_t len; uint32_t idx; } bkword_t;

struct orc_bkt {
	bktree_t	bkt;
	bkword_t **	words;
	size_t		nwords, cap;
};

static int
bkword_levdist(void *ctx, const void *a, const void *b)
{
	const bkword_t *wa = a, *wb = b;
	(void)ctx;
	return orc_levdist(wa->s, wa->len, wb->s, wb->len);
}

orc_bkt_t *
orc_bkt_create(void)
{
	orc_bkt_t *t = calloc(1, sizeof(orc_bkt_t));
	t->bkt.distfunc = bkword_levdist;
	return t;
}

void
orc_bkt_destroy(orc_bkt_t *t)
{
	bktree_free_nodes(&t->bkt);
	for (size_t i = 0; i < t->nwords; i++) {
		free(t->words[i]->s);
		free(t->words[i]);
	}
	free(t->words);
	free(t);
}

int
orc_bkt_insert(orc_bkt_t *t, const char *word, size_t len)
{
	bkword_t *w = calloc(1, sizeof(bkword_t));

	w->s = malloc(len + 1);
	memcpy(w->s, word, len);
	w->s[len] = '\0';
	w->len = len;
	w->idx = t->nwords;
	if (t->nwords == t->cap) {
		t->cap = t->cap ? t->cap * 2 : 64;
		t->words = realloc(t->words, t->cap * sizeof(void *));
	}
	t->words[t->nwords++] = w;
	return bktree_insert(&t->bkt, w);
}

size_t
orc_bkt_search(orc_bkt_t *t, unsigned tolerance, const char *word, size_t len,
    uint32_t *out, size_t cap, uint64_t *ndist)
{
	bkword_t q = { .s = (char *)(uintptr_t)word, .len = len };
	pq_t results = { 0 };
	size_t n = 0;
	bkword_t *w;

	t->bkt.ndist = 0;
	bktree_search(&t->bkt, tolerance, &q, &results);
	while ((w = pq_pop_front(&results)) != NULL) {
		if (n < cap) {
			out[n] = w->idx;
		}
		n++;
	}
	free(results.e);
	if (ndist) {
		*ndist = t->bkt.ndist;
	}
	return n;
}

/* ------------------------------------------------------------------ */
/* Index: terms.c, dtmap.c, idxterm.c, idxdoc.c                        */
/* ------------------------------------------------------------------ */

typedef struct oterm {
	uint32_t	id;
	uint32_t	offset;		/* of the u64 total counter in nxsterms */
	uint16_t	value_len;
	uint64_t *	docs;		/* ascending doc ids (the "doc_bitmap") */
	size_t		ndocs, capdocs;
	char		value[];
} oterm_t;	/* index.h:42-52 */

struct orc_index {
	uint8_t *	tmap;	size_t tmap_len;	/* nxsterms image */
	uint8_t *	dmap;	size_t dmap_len;	/* nxsdtmap image */
	size_t		terms_consumed, dt_consumed;
	uint32_t	terms_last_id;

	smap_t		term_map;	/* value bytes -> oterm_t* */
	oterm_t **	td_map;		/* id -> oterm_t* (NULL: duplicate/lost) */
	size_t		td_cap;
	uint32_t	term_count;
	bktree_t	term_bkt;

	umap_t		dt_map;		/* doc id -> block offset in nxsdtmap */
	uint64_t	dt_count;

	bool		lowercase;	/* "normalizer" stand-in: ASCII only */
	bool		stemmer;	/* the `stemmer' filter, lang "en" */
	uint64_t	last_pairs;
};

static inline uint16_t rd16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return be16toh(v); }
static inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return be32toh(v); }
static inline uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return be64toh(v); }

/* storage.h:60-65 */
#define	IDXTERMS_PAD_LEN(len)	((((2UL + 1 + (len)) + 7) & ~7UL) - (2 + 1 + (len)))
#define	IDXTERMS_HDR_LEN	16
#define	IDXDT_HDR_LEN		32

/* idxterm.c:92-101 */
static int
idxterm_levdist(void *ctx, const void *a, const void *b)
{
	const oterm_t *ta = a, *tb = b;
	(void)ctx;
	return orc_levdist(ta->value, ta->value_len, tb->value, tb->value_len);
}

static void *
map_file(const char *path, size_t *len)
{
	struct stat sb;
	void *p;
	int fd;

	if ((fd = open(path, O_RDONLY)) == -1) {
		return NULL;
	}
	if (fstat(fd, &sb) == -1 || sb.st_size == 0) {
		close(fd);
		return NULL;
	}
	p = mmap(NULL, sb.st_size, PROT_READ, MAP_SHARED, fd, 0);
	close(fd);
	if (p == MAP_FAILED) {
		return NULL;
	}
	*len = sb.st_size;
	return p;
}

/* terms.c:320-414 (idx_terms_sync) + idxterm.c:157-187 (idxterm_insert) */
static int
load_terms(orc_index_t *idx, char *err, size_t errlen)
{
	const uint8_t *hdr = idx->tmap;
	size_t seen_data_len, off;

	if (idx->tmap_len < IDXTERMS_HDR_LEN ||
	    memcmp(hdr, "NXS_T", 5) != 0) {	/* terms.c:65-72 */
		snprintf(err, errlen, "corrupted terms index header");
		return -1;
	}
	if (hdr[5] != 1) {			/* terms.c:73-77 */
		snprintf(err, errlen, "incompatible nxsearch index version");
		return -1;
	}
	seen_data_len = rd32(hdr + 8);		/* storage.h:50 */
	if (IDXTERMS_HDR_LEN + seen_data_len > idx->tmap_len) {
		snprintf(err, errlen, "terms mapping failed");
		return -1;
	}
	off = 0;
	while (off < seen_data_len) {		/* terms.c:367-407 */
		const uint8_t *p = hdr + IDXTERMS_HDR_LEN + off;
		size_t remaining = seen_data_len - off, adv;
		uint16_t len;
		oterm_t *term, *res;
		uint32_t id;

		if (remaining < 2 || (len = rd16(p)) == 0) {
			snprintf(err, errlen, "corrupted terms index");
			return -1;
		}
		adv = 2 + (size_t)len + 1 + IDXTERMS_PAD_LEN(len);
		if (adv + 8 > remaining) {
			snprintf(err, errlen, "corrupted terms index");
			return -1;
		}
		/* idxterm_create: idxterm.c:103-127 */
		term = calloc(1, sizeof(oterm_t) + len + 1);
		memcpy(term->value, p + 2, len);
		term->value_len = len;
		term->offset = IDXTERMS_HDR_LEN + off + adv;

		id = ++idx->terms_last_id;	/* terms.c:404: 1-based file order */
		if (id >= idx->td_cap) {
			size_t ncap = idx->td_cap ? idx->td_cap * 2 : 1024;
			idx->td_map = realloc(idx->td_map, ncap * sizeof(void *));
			memset(idx->td_map + idx->td_cap, 0,
			    (ncap - idx->td_cap) * sizeof(void *));
			idx->td_cap = ncap;
		}
		/* idxterm_insert: duplicate => id is consumed, term dropped */
		res = smap_put(&idx->term_map, term->value, len, term);
		if (res != term) {
			free(term);
		} else if (bktree_insert(&idx->term_bkt, term) == -1) {
			/* idxterm.c:173-177 (unreachable after the dedupe) */
			free(term);
		} else {
			term->id = id;
			idx->td_map[id] = term;
			idx->term_count++;
		}
		off += adv + 8;
	}
	idx->terms_consumed = off;
	return 0;
}

static void
term_add_doc(oterm_t *t, uint64_t doc_id)
{
	if (t->ndocs == t->capdocs) {
		t->capdocs = t->capdocs ? t->capdocs * 2 : 4;
		t->docs = realloc(t->docs, t->capdocs * sizeof(uint64_t));
	}
	t->docs[t->ndocs++] = doc_id;
}

static int
cmp_u64(const void *a, const void *b)
{
	const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
	return (x > y) - (x < y);
}

/* dtmap.c:440-544 (idx_dtmap_sync, DTMAP_PARTIAL_SYNC as at open: dtmap.c:143) */
static int
load_dtmap(orc_index_t *idx, char *err, size_t errlen)
{
	const uint8_t *hdr = idx->dmap;
	size_t seen_data_len, off;

	if (idx->dmap_len < IDXDT_HDR_LEN ||
	    memcmp(hdr, "NXS_D", 5) != 0) {	/* dtmap.c:76-83 */
		snprintf(err, errlen, "corrupted dtmap index header");
		return -1;
	}
	if (hdr[5] != 1) {
		snprintf(err, errlen, "incompatible nxsearch index version");
		return -1;
	}
	seen_data_len = rd64(hdr + 8);		/* storage.h:106-110 */
	if (IDXDT_HDR_LEN + seen_data_len > idx->dmap_len) {
		snprintf(err, errlen, "dtmap mapping failed");
		return -1;
	}
	umap_init(&idx->dt_map, 1024);

	off = 0;
	while (off < seen_data_len) {		/* dtmap.c:483-536 */
		const uint8_t *p = hdr + IDXDT_HDR_LEN + off;
		const size_t remaining = seen_data_len - off;
		uint64_t doc_id, *slot;
		uint32_t doc_total_len, n;
		bool isnew, ok = true;
		unsigned i;

		if (remaining < 16) {
			snprintf(err, errlen, "corrupted dtmap index");
			return -1;
		}
		doc_id = rd64(p);
		doc_total_len = rd32(p + 8);
		n = rd32(p + 12);
		if ((size_t)n * 8 > remaining - 16) {
			snprintf(err, errlen, "corrupted dtmap index");
			return -1;
		}
		/* dtmap_deletion: dtmap.c:357-384 */
		if (doc_id == 0) {
			off += 16 + (size_t)n * 8;
			continue;
		}
		if (doc_total_len == 0) {
			uint64_t v;
			if (umap_get(&idx->dt_map, doc_id, &v)) {
				umap_del(&idx->dt_map, doc_id);
				idx->dt_count--;
				/* NB: the term bitmaps keep the doc (reference
				 * does not unlink them on this path either). */
			}
			off += 16 + (size_t)n * 8;
			continue;
		}
		/* idxdoc_create: idxdoc.c:28-50 */
		slot = umap_slot(&idx->dt_map, doc_id, &isnew);
		if (!isnew) {
			snprintf(err, errlen, "idxdoc_create failed");
			return -1;
		}
		*slot = IDXDT_HDR_LEN + off;
		idx->dt_count++;

		/* dtmap_build_tdmap: dtmap.c:386-438 */
		for (i = 0; i < n; i++) {
			const uint32_t term_id = rd32(p + 16 + i * 8);
			oterm_t *term = (term_id < idx->td_cap) ?
			    idx->td_map[term_id] : NULL;
			if (term == NULL) {
				ok = false;
				break;
			}
			term_add_doc(term, doc_id);
		}
		if (!ok) {
			/* revert and stop consuming (partial sync: ret 0) */
			while (i--) {
				const uint32_t term_id = rd32(p + 16 + i * 8);
				oterm_t *term = idx->td_map[term_id];
				term->ndocs--;
			}
			umap_del(&idx->dt_map, doc_id);
			idx->dt_count--;
			break;
		}
		off += 16 + (size_t)n * 8;
	}
	idx->dt_consumed = off;

	/* roaring set semantics: ascending + unique */
	for (uint32_t id = 1; id <= idx->terms_last_id; id++) {
		oterm_t *t = idx->td_map[id];
		bool sorted = true;
		size_t w = 0;

		if (!t || t->ndocs < 2) {
			continue;
		}
		for (size_t j = 1; j < t->ndocs; j++) {
			if (t->docs[j] <= t->docs[j - 1]) {
				sorted = false;
				break;
			}
		}
		if (sorted) {
			continue;
		}
		qsort(t->docs, t->ndocs, sizeof(uint64_t), cmp_u64);
		for (size_t j = 0; j < t->ndocs; j++) {
			if (w == 0 || t->docs[j] != t->docs[w - 1]) {
				t->docs[w++] = t->docs[j];
			}
		}
		t->ndocs = w;
	}
	return 0;
}

orc_index_t *
orc_index_load(const char *terms_path, const char *dtmap_path,
    char *err, size_t errlen)
{
	orc_index_t *idx = calloc(1, sizeof(orc_index_t));
	char ebuf[64];

	if (!err) {
		err = ebuf;
		errlen = sizeof(ebuf);
	}
	smap_init(&idx->term_map);
	idx->term_bkt.distfunc = idxterm_levdist;
	idx->lowercase = false;
	idx->stemmer = false;

	if ((idx->tmap = map_file(terms_path, &idx->tmap_len)) == NULL) {
		snprintf(err, errlen, "could not open terms index");
		goto fail;
	}
	if ((idx->dmap = map_file(dtmap_path, &idx->dmap_len)) == NULL) {
		snprintf(err, errlen, "could not open dtmap index");
		goto fail;
	}
	if (load_terms(idx, err, errlen) == -1 ||
	    load_dtmap(idx, err, errlen) == -1) {
		goto fail;
	}
	return idx;
fail:
	orc_index_free(idx);
	return NULL;
}

void
orc_index_free(orc_index_t *idx)
{
	if (!idx) {
		return;
	}
	bktree_free_nodes(&idx->term_bkt);
	for (uint32_t id = 1; idx->td_map && id <= idx->terms_last_id; id++) {
		if (idx->td_map[id]) {
			free(idx->td_map[id]->docs);
			free(idx->td_map[id]);
		}
	}
	free(idx->td_map);
	free(idx->term_map.e);
	umap_free(&idx->dt_map);
	if (idx->tmap) munmap(idx->tmap, idx->tmap_len);
	if (idx->dmap) munmap(idx->dmap, idx->dmap_len);
	free(idx);
}

uint32_t orc_index_term_count(const orc_index_t *idx) { return idx->term_count; }
uint64_t orc_index_dt_count(const orc_index_t *idx) { return idx->dt_count; }
/* dtmap.c:660-677: header counters */
uint32_t orc_index_doc_count(const orc_index_t *idx) { return rd32(idx->dmap + 24); }
uint64_t orc_index_token_count(const orc_index_t *idx) { return rd64(idx->dmap + 16); }

void
orc_index_set_lowercase(orc_index_t *idx, bool on)
{
	idx->lowercase = on;
}

void
orc_index_set_stemmer(orc_index_t *idx, bool on)
{
	idx->stemmer = on;
}

static oterm_t *
idxterm_lookup(const orc_index_t *idx, const char *value, size_t len)
{
	return smap_get(&idx->term_map, value, len);	/* idxterm.c:192-196 */
}

/* idxterm.c:251-260 */
static uint64_t
idxterm_get_total(const orc_index_t *idx, const oterm_t *term)
{
	return rd64(idx->tmap + term->offset);
}

/* idxterm.c:210-249 */
static oterm_t *
idxterm_fuzzysearch(const orc_index_t *idx, const char *value, size_t len,
    uint64_t *visited)
{
	oterm_t *search_token, *term = NULL, *iterm;
	bktree_t *bkt = (bktree_t *)(uintptr_t)&idx->term_bkt;
	pq_t results = { 0 };
	uint64_t term_total = 0;	/* never updated: idxterm.c:215,238-242 */

	search_token = calloc(1, sizeof(oterm_t) + len + 1);
	memcpy(search_token->value, value, len);
	search_token->value_len = len;	/* uint16_t, as in the reference */

	bkt->ndist = 0;
	bktree_search(bkt, 2 /* LEVDIST_TOLERANCE, index.h:26 */,
	    search_token, &results);
	while ((iterm = pq_pop_back(&results)) != NULL) {
		if (idxterm_get_total(idx, iterm) > term_total) {
			term = iterm;
		}
	}
	if (visited) {
		*visited = bkt->ndist;
	}
	free(results.e);
	free(search_token);
	return term;
}

uint32_t
orc_index_lookup(const orc_index_t *idx, const char *tok, size_t len)
{
	const oterm_t *t = idxterm_lookup(idx, tok, len);
	return t ? t->id : 0;
}

uint32_t
orc_index_fuzzy(const orc_index_t *idx, const char *tok, size_t len,
    uint64_t *visited)
{
	const oterm_t *t = idxterm_fuzzysearch(idx, tok, len, visited);
	return t ? t->id : 0;
}

uint64_t
orc_index_df(const orc_index_t *idx, uint32_t term_id)
{
	const oterm_t *t = (term_id < idx->td_cap) ? idx->td_map[term_id] : NULL;
	return t ? t->ndocs : 0;
}

const char *
orc_index_term(const orc_index_t *idx, uint32_t term_id, size_t *len)
{
	const oterm_t *t = (term_id < idx->td_cap) ? idx->td_map[term_id] : NULL;
	if (!t) {
		return NULL;
	}
	*len = t->value_len;
	return t->value;
}

/* idxdoc.c:78-94 */
static int
idxdoc_get_doclen(const orc_index_t *idx, uint64_t doc_off)
{
	return (int)rd32(idx->dmap + doc_off + 8);
}

/* idxdoc.c:100-142 */
static int
idxdoc_get_termcount(const orc_index_t *idx, uint64_t doc_off, uint32_t term_id)
{
	const uint8_t *termblocks = idx->dmap + doc_off + 16;
	unsigned n = rd32(idx->dmap + doc_off + 12), off = 0;

	while (n) {
		unsigned i = off + (n >> 1);
		uint32_t target = rd32(termblocks + (size_t)i * 8);

		if (term_id == target) {
			return rd32(termblocks + (size_t)i * 8 + 4);
		}
		if (term_id > target) {
			off = i + 1;
			n--;
		}
		n >>= 1;
	}
	return -1;
}

/* the ranking_func_t of nxs_impl.h:52-53 applied to (term, doc) */
static float
rank_pair(const orc_index_t *idx, int algo, const oterm_t *term, uint64_t doc_off)
{
	const int term_freq = idxdoc_get_termcount(idx, doc_off, term->id);
	const uint32_t doc_count = orc_index_doc_count(idx);
	const uint64_t doc_freq = term->ndocs;	/* bitmap cardinality */

	if (algo == ORC_TF_IDF) {
		return orc_tf_idf(term_freq, doc_count, doc_freq);
	}
	return orc_bm25(term_freq, idxdoc_get_doclen(idx, doc_off), doc_count,
	    orc_index_token_count(idx), doc_freq);
}

float
orc_index_score(const orc_index_t *idx, int algo, uint32_t term_id,
    uint64_t doc_id)
{
	const oterm_t *t = (term_id < idx->td_cap) ? idx->td_map[term_id] : NULL;
	uint64_t off;

	if (!t || !umap_get(&idx->dt_map, doc_id, &off)) {
		return -2;
	}
	return rank_pair(idx, algo, t, off);
}

/* ------------------------------------------------------------------ */
/* Query lexer + parser: scan.re:43-121, grammar.y:66-140              */
/* ------------------------------------------------------------------ */

enum {
	TK_EOF = 0, TK_AND, TK_OR, TK_NOT, TK_BR_OPEN, TK_BR_CLOSE,
	TK_FF_STRING, TK_QUOTED_STRING,
};

typedef struct {
	const char *	cursor;
	const char *	token;
	const char *	cur_line;
	unsigned	line;
	/* value of the last string token */
	char *		str;
} lexer_t;

static inline bool
is_sp(unsigned char c)
{
	/* SP = [ \t\v\f\r\n]: scan.re:59 */
	return c == ' ' || c == '\t' || c == '\v' || c == '\f' ||
	    c == '\r' || c == '\n';
}

/* one token; re2c semantics = longest match, earlier rule wins ties */
static int
lex(lexer_t *ctx)
{
loop:
	ctx->token = ctx->cursor;
	const unsigned char *p = (const unsigned char *)ctx->cursor;
	size_t ff_len = 0, kw_len = 0, str_len = 0;
	int kw = 0;

	if (*p == 0) {
		return TK_EOF;				/* scan.re:83 */
	}
	if (is_sp(*p)) {
		size_t n = 1;
		while (is_sp(p[n])) {
			n++;
		}
		if (*p == '\n' && n == 1) {		/* EOL rule: scan.re:89 */
			ctx->cur_line = ctx->token;
			ctx->line++;
		}
		ctx->cursor += n;			/* WSP rule: scan.re:90 */
		goto loop;
	}
	if (*p == '(') {
		ctx->cursor++;
		return TK_BR_OPEN;
	}
	if (*p == ')') {
		ctx->cursor++;
		return TK_BR_CLOSE;
	}
	/* FF_STR = ([^\x00] \ SP \ "(" \ ")")+ : scan.re:76 */
	while (p[ff_len] && !is_sp(p[ff_len]) &&
	    p[ff_len] != '(' && p[ff_len] != ')') {
		ff_len++;
	}
	/* AND = '&' | 'AND'; OR = '|' | 'OR'; NOT = 'NOT' (case-insens.) */
	if (*p == '&') {
		kw = TK_AND; kw_len = 1;
	} else if (*p == '|') {
		kw = TK_OR; kw_len = 1;
	} else if (strncasecmp((const char *)p, "AND", 3) == 0) {
		kw = TK_AND; kw_len = 3;
	} else if (strncasecmp((const char *)p, "NOT", 3) == 0) {
		kw = TK_NOT; kw_len = 3;
	} else if (strncasecmp((const char *)p, "OR", 2) == 0) {
		kw = TK_OR; kw_len = 2;
	}
	/* SQ_STR / DQ_STR: scan.re:72-74 */
	if (*p == '\'' || *p == '"') {
		const unsigned char q = *p;
		size_t i = 1;
		for (;;) {
			if (p[i] == 0) {
				break;
			}
			if (p[i] == '\\') {
				if (p[i + 1] == 0) {
					break;
				}
				i += 2;
				continue;
			}
			if (p[i] == q) {
				str_len = i + 1;
				break;
			}
			i++;
		}
	}
	/* rule order: AND, OR, NOT, "(", ")", STR, FF_STR */
	if (kw && kw_len >= str_len && kw_len >= ff_len) {
		ctx->cursor += kw_len;
		return kw;
	}
	if (str_len && str_len >= ff_len) {
		ctx->cursor += str_len;
		ctx->str = strndup(ctx->token + 1, str_len - 2);	/* scan.re:108 */
		return TK_QUOTED_STRING;
	}
	ctx->cursor += ff_len;
	ctx->str = strndup(ctx->token, ff_len);		/* scan.re:115 */
	return TK_FF_STRING;
}

int
orc_query_lex(const char *query, int *kinds, size_t cap)
{
	lexer_t lx = { .cursor = query, .cur_line = query, .line = 1 };
	int n = 0, tk;

	while ((tk = lex(&lx)) > 0) {
		if (tk >= TK_FF_STRING) {
			free(lx.str);
		}
		if ((size_t)n < cap) {
			kinds[n] = tk;
		}
		n++;
	}
	return n;
}

typedef enum { EXPR_VAL_TOKEN, EXPR_OP_AND, EXPR_OP_OR, EXPR_OP_NOT } expr_type_t;

struct otoken;

typedef struct expr {
	expr_type_t	type;
	char *		value;
	struct otoken *	token;
	struct expr *	elements[2];
} expr_t;	/* expr.h:24-34 (binary only: grammar.y:81-99) */

typedef struct {
	lexer_t		lx;
	int		tk;		/* lookahead */
	char *		tkstr;
	bool		error;
	char *		errmsg;
} parser_t;

static void
expr_destroy(expr_t *e)
{
	if (!e) {
		return;
	}
	if (e->type != EXPR_VAL_TOKEN) {
		expr_destroy(e->elements[0]);
		expr_destroy(e->elements[1]);
	}
	free(e->value);
	free(e);
}

static void
parser_advance(parser_t *ps)
{
	ps->tk = lex(&ps->lx);
	ps->tkstr = (ps->tk >= TK_FF_STRING) ? ps->lx.str : NULL;
}

/* query_set_error: query.c:46-58 */
static void
parser_error(parser_t *ps)
{
	if (!ps->error) {
		const unsigned offset = (uintptr_t)ps->lx.token -
		    (uintptr_t)ps->lx.cur_line;
		if (asprintf(&ps->errmsg, "syntax error near %u:%u: \"%.50s ...\"",
		    ps->lx.line, offset, ps->lx.token) == -1) {
			ps->errmsg = NULL;
		}
		ps->error = true;
	}
}

static expr_t *parse_expr(parser_t *ps, int min_prec);

static expr_t *
expr_op(expr_type_t type, expr_t *l, expr_t *r)
{
	expr_t *e = calloc(1, sizeof(expr_t));
	e->type = type;
	e->elements[0] = l;
	e->elements[1] = r;
	return e;
}

static expr_t *
parse_primary(parser_t *ps)
{
	expr_t *e;

	if (ps->tk == TK_FF_STRING || ps->tk == TK_QUOTED_STRING) {
		e = calloc(1, sizeof(expr_t));	/* grammar.y:106-110 */
		e->type = EXPR_VAL_TOKEN;
		e->value = ps->tkstr;
		parser_advance(ps);
		return e;
	}
	if (ps->tk == TK_BR_OPEN) {		/* grammar.y:101-104 */
		parser_advance(ps);
		if ((e = parse_expr(ps, 1)) == NULL) {
			return NULL;
		}
		if (ps->tk != TK_BR_CLOSE) {
			parser_error(ps);
			expr_destroy(e);
			return NULL;
		}
		parser_advance(ps);
		return e;
	}
	parser_error(ps);
	return NULL;
}

/*
 * %left OR. %left AND. %left NOT.  (grammar.y:66-69).  The "expr AND NOT
 * expr" rule takes the precedence of its left-most terminal (AND), so it
 * binds exactly like AND; all three are left-associative.
 */
static expr_t *
parse_expr(parser_t *ps, int min_prec)
{
	expr_t *lhs, *rhs;

	if ((lhs = parse_primary(ps)) == NULL) {
		return NULL;
	}
	for (;;) {
		if (ps->tk == TK_OR && min_prec <= 1) {
			parser_advance(ps);
			if ((rhs = parse_expr(ps, 2)) == NULL) {
				expr_destroy(lhs);
				return NULL;
			}
			lhs = expr_op(EXPR_OP_OR, lhs, rhs);	/* grammar.y:91-94 */
			continue;
		}
		if (ps->tk == TK_AND && min_prec <= 2) {
			expr_type_t type = EXPR_OP_AND;		/* grammar.y:86-89 */
			parser_advance(ps);
			if (ps->tk == TK_NOT) {
				type = EXPR_OP_NOT;		/* grammar.y:96-99 */
				parser_advance(ps);
			}
			if ((rhs = parse_expr(ps, 3)) == NULL) {
				expr_destroy(lhs);
				return NULL;
			}
			lhs = expr_op(type, lhs, rhs);
			continue;
		}
		break;
	}
	return lhs;
}

/* query ::= expr_list; juxtaposition = OR at the top level only (grammar.y:71-84) */
static expr_t *
query_parse(const char *query, char **errmsg)
{
	parser_t ps = { .lx = { .cursor = query, .cur_line = query, .line = 1 } };
	expr_t *root, *e;

	parser_advance(&ps);
	if ((root = parse_expr(&ps, 1)) == NULL) {
		goto err;
	}
	while (ps.tk != TK_EOF) {
		if (ps.tk != TK_FF_STRING && ps.tk != TK_QUOTED_STRING &&
		    ps.tk != TK_BR_OPEN) {
			parser_error(&ps);
			goto err;
		}
		if ((e = parse_expr(&ps, 1)) == NULL) {
			goto err;
		}
		root = expr_op(EXPR_OP_OR, root, e);
	}
	*errmsg = NULL;
	return root;
err:
	expr_destroy(root);
	if (ps.tk >= TK_FF_STRING) {
		free(ps.tkstr);
	}
	*errmsg = ps.errmsg ? ps.errmsg : strdup("out of memory");
	return NULL;
}

/* t_queryparser.c:146-169 */
static char *
expr_string_dump(const expr_t *expr)
{
	static const char *op[] = {
		[EXPR_OP_AND] = "AND", [EXPR_OP_OR] = "OR", [EXPR_OP_NOT] = "NOT",
	};
	char *buf = NULL;

	if (expr->type == EXPR_VAL_TOKEN) {
		if (asprintf(&buf, "`%s`", expr->value) == -1) return NULL;
	} else {
		char *e1 = expr_string_dump(expr->elements[0]);
		char *e2 = expr_string_dump(expr->elements[1]);
		if (asprintf(&buf, "(%s %s %s)", op[expr->type], e1, e2) == -1) buf = NULL;
		free(e1);
		free(e2);
	}
	return buf;
}

char *
orc_query_repr(const char *query, char **errmsg)
{
	char *em = NULL, *repr;
	expr_t *root = query_parse(query, &em);

	if (!root) {
		if (errmsg) *errmsg = em; else free(em);
		return NULL;
	}
	if (errmsg) *errmsg = NULL;
	repr = expr_string_dump(root);
	expr_destroy(root);
	return repr;
}

/* ------------------------------------------------------------------ */
/* Token set: tokenizer.c:94-199                                       */
/* ------------------------------------------------------------------ */

typedef struct otoken {
	char *		value;
	size_t		len;
	oterm_t *	idxterm;
	bool		removed;	/* TOKENSET_TRIM'ed */
} otoken_t;

typedef struct {
	otoken_t **	list;		/* first-seen order */
	size_t		n, cap;
} tokenset_t;

static otoken_t *
tokenset_add(tokenset_t *ts, const char *val, size_t len)
{
	otoken_t *t;

	for (size_t i = 0; i < ts->n; i++) {	/* tokenizer.c:100-107 */
		if (ts->list[i]->len == len &&
		    memcmp(ts->list[i]->value, val, len) == 0) {
			return ts->list[i];
		}
	}
	t = calloc(1, sizeof(otoken_t));
	t->value = malloc(len + 1);
	memcpy(t->value, val, len);
	t->value[len] = '\0';
	t->len = len;
	if (ts->n == ts->cap) {
		ts->cap = ts->cap ? ts->cap * 2 : 16;
		ts->list = realloc(ts->list, ts->cap * sizeof(void *));
	}
	ts->list[ts->n++] = t;		/* TAILQ_INSERT_TAIL */
	return t;
}

static void
tokenset_destroy(tokenset_t *ts)
{
	for (size_t i = 0; i < ts->n; i++) {
		free(ts->list[i]->value);
		free(ts->list[i]);
	}
	free(ts->list);
}

/* ------------------------------------------------------------------ */
/* Document sets (roaring64 stand-in): sorted unique uint64_t vectors   */
/* ------------------------------------------------------------------ */

typedef struct { uint64_t *v; size_t n; } docset_t;

static docset_t *
docset_copy(const uint64_t *v, size_t n)
{
	docset_t *s = malloc(sizeof(docset_t));
	s->v = malloc((n ? n : 1) * sizeof(uint64_t));
	if (n) memcpy(s->v, v, n * sizeof(uint64_t));
	s->n = n;
	return s;
}

static void
docset_free(docset_t *s)
{
	free(s->v);
	free(s);
}

static void
docset_and(docset_t *a, const docset_t *b)
{
	size_t i = 0, j = 0, w = 0;
	while (i < a->n && j < b->n) {
		if (a->v[i] < b->v[j]) i++;
		else if (a->v[i] > b->v[j]) j++;
		else { a->v[w++] = a->v[i]; i++; j++; }
	}
	a->n = w;
}

static void
docset_or(docset_t *a, const docset_t *b)
{
	uint64_t *out = malloc((a->n + b->n + 1) * sizeof(uint64_t));
	size_t i = 0, j = 0, w = 0;
	while (i < a->n || j < b->n) {
		if (j >= b->n || (i < a->n && a->v[i] < b->v[j])) out[w++] = a->v[i++];
		else if (i >= a->n || b->v[j] < a->v[i]) out[w++] = b->v[j++];
		else { out[w++] = a->v[i]; i++; j++; }
	}
	free(a->v);
	a->v = out;
	a->n = w;
}

static void
docset_andnot(docset_t *a, const docset_t *b)
{
	size_t i = 0, j = 0, w = 0;
	while (i < a->n) {
		while (j < b->n && b->v[j] < a->v[i]) j++;
		if (j < b->n && b->v[j] == a->v[i]) { i++; continue; }
		a->v[w++] = a->v[i++];
	}
	a->n = w;
}

static bool
docs_contains(const uint64_t *v, size_t n, uint64_t key)
{
	size_t lo = 0, hi = n;
	while (lo < hi) {
		size_t mid = lo + (hi - lo) / 2;
		if (v[mid] < key) lo = mid + 1; else hi = mid;
	}
	return lo < n && v[lo] == key;
}

/* ------------------------------------------------------------------ */
/* Search: search.c:78-342, query.c:75-115, results.c:128-220          */
/* ------------------------------------------------------------------ */

#define	NXS_QUERY_RLIMIT	100	/* search.c:70 */

typedef struct {
	int	code;
	char *	buf;
	size_t	len;
} errslot_t;

static void
set_err(errslot_t *es, int code, const char *fmt, ...)
{
	va_list ap;
	es->code = code;
	if (es->buf && es->len) {
		va_start(ap, fmt);
		vsnprintf(es->buf, es->len, fmt, ap);
		va_end(ap);
	}
}

/* get_expr_bitmap: search.c:118-174 */
static docset_t *
get_expr_bitmap(expr_t *expr, unsigned r, errslot_t *es)
{
	docset_t *result, *elm;

	if (r > NXS_QUERY_RLIMIT) {
		set_err(es, ORC_ERR_LIMIT,
		    "query nesting limit reached (%u levels)", NXS_QUERY_RLIMIT);
		return NULL;
	}
	if (expr->type == EXPR_VAL_TOKEN) {
		const otoken_t *token = expr->token;
		/*
		 * Q14: a trimmed token is a dangling pointer in the reference
		 * (tokenizer.c:188-192 vs search.c:133-139); the defined
		 * stand-in is the empty set, as for a NULL token (search.c:140).
		 */
		if (token && !token->removed) {
			const oterm_t *term = token->idxterm;
			return docset_copy(term->docs, term->ndocs);
		}
		return docset_copy(NULL, 0);
	}
	if ((result = get_expr_bitmap(expr->elements[0], r + 1, es)) == NULL) {
		return NULL;
	}
	if ((elm = get_expr_bitmap(expr->elements[1], r + 1, es)) == NULL) {
		docset_free(result);
		return NULL;
	}
	switch (expr->type) {
	case EXPR_OP_AND: docset_and(result, elm); break;
	case EXPR_OP_OR:  docset_or(result, elm); break;
	case EXPR_OP_NOT: docset_andnot(result, elm); break;
	default: abort();
	}
	docset_free(elm);
	return result;
}

uint64_t
orc_last_pairs(const orc_index_t *idx)
{
	return idx->last_pairs;
}

int
orc_search(orc_index_t *idx, const char *query, int algo, uint64_t limit,
    bool fuzzymatch, orc_result_t *out, size_t cap, uint32_t *count,
    int *errcode, char *errmsg, size_t errlen)
{
	errslot_t es = { .code = 0, .buf = errmsg, .len = errlen };
	tokenset_t tokens = { 0 };
	result_entry_t *results = NULL, *entry;
	expr_t *root = NULL;
	docset_t *doc_bitmap = NULL;
	umap_t doc_map = { 0 };
	heap_t heap = { 0 };
	char *perr = NULL;
	size_t nresults = 0;
	int ret = -1;

	*count = 0;
	idx->last_pairs = 0;
	if (errmsg && errlen) {
		errmsg[0] = '\0';
	}

	/* get_search_params: search.c:96-107 */
	if (limit == 0 || limit > UINT_MAX) {
		set_err(&es, ORC_ERR_INVALID, "invalid limit");
		goto out;
	}
	if (algo != ORC_TF_IDF && algo != ORC_BM25) {
		set_err(&es, ORC_ERR_INVALID, "invalid algorithm");
		goto out;
	}

	/* construct_query: search.c:176-208 */
	if ((root = query_parse(query, &perr)) == NULL) {
		set_err(&es, ORC_ERR_INVALID, "query failed with %s", perr);
		goto out;
	}

	/*
	 * query_prepare: query.c:75-115.  Explicit stack, children pushed
	 * left to right and popped from the back => leaves right-to-left.
	 */
	{
		pq_t iter = { 0 };
		expr_t *expr;

		pq_push(&iter, root);
		while ((expr = pq_pop_back(&iter)) != NULL) {
			if (expr->type != EXPR_VAL_TOKEN) {
				pq_push(&iter, expr->elements[0]);
				pq_push(&iter, expr->elements[1]);
				continue;
			}
			/* tokenize_value: tokenizer.c:205-227; filter pipeline
			 * reduced to the ASCII part of the normalizer */
			size_t len = strlen(expr->value);
			char *val = malloc(len + 2);
			memcpy(val, expr->value, len + 1);
			if (idx->lowercase) {
				for (size_t i = 0; i < len; i++) {
					if (val[i] >= 'A' && val[i] <= 'Z') {
						val[i] += 'a' - 'A';
					}
				}
			}
			if (idx->stemmer) {	/* stemmer_filter: filters_builtin.c:219-238 */
				char *st = malloc(len + 2);
				len = orc_stem_en(val, len, st, len + 2);
				free(val);
				val = st;
			}
			expr->token = tokenset_add(&tokens, val, len);
			free(val);
		}
		free(iter.e);
	}
	/* tokenset_resolve(TRIM | flags): tokenizer.c:160-199 */
	size_t live_tokens = 0;
	for (size_t i = 0; i < tokens.n; i++) {
		otoken_t *token = tokens.list[i];
		oterm_t *term = idxterm_lookup(idx, token->value, token->len);
		if (!term && fuzzymatch) {
			term = idxterm_fuzzysearch(idx, token->value, token->len, NULL);
		}
		if (!term) {
			token->removed = true;
		} else {
			token->idxterm = term;
			live_tokens++;
		}
	}

	/* run_query_logic: search.c:210-278 */
	if (live_tokens == 0) {		/* search.c:224-226 */
		ret = 0;
		goto out;
	}
	if ((doc_bitmap = get_expr_bitmap(root, 0, &es)) == NULL) {
		goto out;
	}
	umap_init(&doc_map, 1024);
	for (size_t di = 0; di < doc_bitmap->n; di++) {
		const uint64_t doc_id = doc_bitmap->v[di];

		for (size_t ti = 0; ti < tokens.n; ti++) {
			const otoken_t *token = tokens.list[ti];
			const oterm_t *term;
			uint64_t doc_off, *slot;
			float score;
			bool isnew;

			if (token->removed) {
				continue;
			}
			term = token->idxterm;
			if (!docs_contains(term->docs, term->ndocs, doc_id)) {
				continue;	/* search.c:240-243 */
			}
			if (!umap_get(&idx->dt_map, doc_id, &doc_off)) {
				set_err(&es, ORC_ERR_FATAL, "internal error");
				goto out;	/* search.c:248-250 */
			}
			idx->last_pairs++;
			if ((score = rank_pair(idx, algo, term, doc_off)) < 0) {
				continue;	/* search.c:251-256 */
			}
			/* nxs_resp_addresult: results.c:128-150 */
			slot = umap_slot(&doc_map, doc_id, &isnew);
			if (!isnew) {
				entry = (result_entry_t *)(uintptr_t)*slot;
				entry->score += score;
				continue;
			}
			entry = calloc(1, sizeof(result_entry_t));
			*slot = (uintptr_t)entry;
			entry->doc_id = doc_id;
			entry->score = score;
			entry->next = results;	/* prepend => descending doc id */
			results = entry;
			nresults++;
		}
	}

	/* nxs_resp_build: results.c:182-220 */
	heap.cap = limit;
	heap.items = calloc(MIN((size_t)limit, nresults) + 1, sizeof(void *));
	for (entry = results; entry; entry = entry->next) {
		heap_add(&heap, entry);
	}
	{
		size_t cnt;
		result_entry_t **top = heap_sort(&heap, &cnt);
		for (size_t i = 0; i < cnt && i < cap; i++) {
			out[i].doc_id = top[i]->doc_id;
			out[i].score = top[i]->score;
		}
		*count = cnt;
	}
	ret = 0;
out:
	while (results) {
		entry = results->next;
		free(results);
		results = entry;
	}
	free(heap.items);
	if (doc_map.k) umap_free(&doc_map);
	if (doc_bitmap) docset_free(doc_bitmap);
	tokenset_destroy(&tokens);
	expr_destroy(root);
	free(perr);
	if (errcode) {
		*errcode = es.code;
	}
	return ret;
}

/* ------------------------------------------------------------------ */
/* Response JSON: results.c:118-122,153-161,218                        */
/* ------------------------------------------------------------------ */

/*
 * yyjson writes a real as the shortest decimal that round-trips the double,
 * always keeping a fraction ("3.0").  PARITY UNPINNED beyond the two values
 * the reference tests hold (t_misc.c:115-117: 3.0 and 1.5); yyjson's source
 * is an absent submodule.  Decimal notation for 1e-6 <= |x| < 1e21,
 * exponent notation otherwise.
 */
static void
fmt_real(char *buf, size_t len, double v)
{
	char tmp[40];
	int prec;

	for (prec = 1; prec <= 17; prec++) {
		snprintf(tmp, sizeof(tmp), "%.*e", prec - 1, v);
		if (strtod(tmp, NULL) == v) {
			break;
		}
	}
	/* tmp = d[.ddd]e[+-]XX */
	char digits[24];
	int nd = 0, exp10;
	const char *e = strchr(tmp, 'e');
	bool neg = tmp[0] == '-';
	for (const char *p = tmp + neg; p < e; p++) {
		if (*p != '.') digits[nd++] = *p;
	}
	while (nd > 1 && digits[nd - 1] == '0') nd--;
	digits[nd] = '\0';
	exp10 = atoi(e + 1);	/* value = d.ddd * 10^exp10 */

	char *o = buf;
	const char *end = buf + len - 1;
#define PUT(c) do { if (o < end) *o++ = (c); } while (0)
	if (neg) PUT('-');
	if (exp10 >= -6 && exp10 < 21) {
		if (exp10 < 0) {
			PUT('0'); PUT('.');
			for (int i = 0; i < -exp10 - 1; i++) PUT('0');
			for (int i = 0; i < nd; i++) PUT(digits[i]);
		} else {
			for (int i = 0; i <= exp10; i++) PUT(i < nd ? digits[i] : '0');
			PUT('.');
			if (nd > exp10 + 1) {
				for (int i = exp10 + 1; i < nd; i++) PUT(digits[i]);
			} else {
				PUT('0');
			}
		}
	} else {
		PUT(digits[0]);
		if (nd > 1) {
			PUT('.');
			for (int i = 1; i < nd; i++) PUT(digits[i]);
		}
		o += snprintf(o, end - o, "e%d", exp10);
	}
	*o = '\0';
#undef PUT
}

char *
orc_results_json(const orc_result_t *res, size_t n)
{
	size_t cap = 64 + n * 80, len = 0;
	char *s = malloc(cap), num[48];

	len += snprintf(s + len, cap - len, "{\"results\":[");
	for (size_t i = 0; i < n; i++) {
		fmt_real(num, sizeof(num), (double)res[i].score);
		len += snprintf(s + len, cap - len,
		    "%s{\"doc_id\":%llu,\"score\":%s}", i ? "," : "",
		    (unsigned long long)res[i].doc_id, num);
	}
	len += snprintf(s + len, cap - len, "],\"count\":%zu}", n);
	return s;
}

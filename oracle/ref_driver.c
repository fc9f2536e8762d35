/*
 * ref_driver.c -- thin driver around the GENUINE reference sources
 * src/algo/{levdist,bktree,deque,heap}.c, which oracle/Makefile compiles in
 * place from /root/reference into oracle/_ref/libnxsref.so (never copied).
 *
 * TEST INFRASTRUCTURE ONLY.  It exposes ctypes-friendly entry points so the
 * tests can validate the oracle restatement (nxs_oracle.c) against the real
 * reference on random inputs, and bench.py can time the reference's own
 * BK-tree/Levenshtein as the fuzzy CPU baseline (cpu_baseline.kind =
 * "reference").  This file contains no reference code: only calls into the
 * reference's public algo API (levdist.h, bktree.h, deque.h, heap.h).
 */
#include <stdlib.h>
#include <stdint.h>
#include <stdbool.h>
#include <string.h>

#include "levdist.h"
#include "deque.h"
#include "bktree.h"
#include "heap.h"

typedef struct { const char *s; size_t len; uint32_t idx; } word_t;

typedef struct {
	bktree_t *	bkt;
	levdist_t *	lev;
	word_t *	words;
	size_t		n;
	uint64_t	ndist;
} ref_bkt_t;

int
ref_levdist(const char *s1, size_t n, const char *s2, size_t m)
{
	levdist_t *ctx = levdist_create();
	int d = levdist(ctx, s1, n, s2, m);
	levdist_destroy(ctx);
	return d;
}

static int
word_dist(void *ctx, const void *a, const void *b)
{
	ref_bkt_t *t = ctx;
	const word_t *wa = a, *wb = b;
	t->ndist++;
	return levdist(t->lev, wa->s, wa->len, wb->s, wb->len);
}

/* words: concatenated bytes; offs[n+1] */
ref_bkt_t *
ref_bkt_build(const char *bytes, const uint32_t *offs, size_t n)
{
	ref_bkt_t *t = calloc(1, sizeof(ref_bkt_t));

	t->lev = levdist_create();
	t->bkt = bktree_create(word_dist, t);
	t->words = calloc(n ? n : 1, sizeof(word_t));
	t->n = n;
	for (size_t i = 0; i < n; i++) {
		t->words[i].s = bytes + offs[i];
		t->words[i].len = offs[i + 1] - offs[i];
		t->words[i].idx = i;
		(void)bktree_insert(t->bkt, &t->words[i]);
	}
	return t;
}

uint64_t
ref_bkt_ndist(const ref_bkt_t *t)
{
	return t->ndist;
}

/* returns #matches; out[] = word indices in deque push order */
size_t
ref_bkt_search(ref_bkt_t *t, unsigned tol, const char *q, size_t qlen,
    uint32_t *out, size_t cap, uint64_t *ndist)
{
	word_t qw = { .s = q, .len = qlen };
	deque_t *dq = deque_create(0, 0);
	size_t n = 0;
	word_t *w;

	t->ndist = 0;
	bktree_search(t->bkt, tol, &qw, dq);
	while ((w = deque_pop_front(dq)) != NULL) {
		if (n < cap) {
			out[n] = w->idx;
		}
		n++;
	}
	deque_destroy(dq);
	if (ndist) {
		*ndist = t->ndist;
	}
	return n;
}

void
ref_bkt_destroy(ref_bkt_t *t)
{
	bktree_destroy(t->bkt);
	levdist_destroy(t->lev);
	free(t->words);
	free(t);
}

typedef struct { uint64_t id; float score; } ent_t;

/* the comparator of reference src/core/results.c:165-176, restated */
static int
ent_cmp(const void *p1, const void *p2)
{
	const ent_t *e1 = p1, *e2 = p2;
	if (e1->score < e2->score) return -1;
	if (e1->score > e2->score) return 1;
	return 0;
}

/* feed (id,score) in the given order through the reference heap */
size_t
ref_topk(const uint64_t *ids, const float *scores, size_t n, size_t cap,
    uint64_t *out_ids, float *out_scores)
{
	ent_t *ents = calloc(n ? n : 1, sizeof(ent_t));
	heap_t *h = heap_create(cap, ent_cmp);
	ent_t **top;
	size_t cnt;

	for (size_t i = 0; i < n; i++) {
		ents[i].id = ids[i];
		ents[i].score = scores[i];
		heap_add(h, &ents[i]);
	}
	top = heap_sort(h, &cnt);
	for (size_t i = 0; i < cnt; i++) {
		out_ids[i] = top[i]->id;
		out_scores[i] = top[i]->score;
	}
	heap_destroy(h);
	free(ents);
	return cnt;
}

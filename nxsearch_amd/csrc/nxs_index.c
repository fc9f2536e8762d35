/*
 * nxs_index.c -- host side of index open: maps the reference's two on-disk
 * files, validates them, builds the term dictionary and the BK-tree image,
 * walks the doc blocks, and hands everything to the device builder.
 *
 * Reference read path mirrored here (behaviour, not code):
 *   nxsterms: idx_terms_verify + idx_terms_sync   src/index/terms.c:65-81,320-414
 *             idxterm_insert (dedupe, BK insert)  src/index/idxterm.c:157-187
 *   nxsdtmap: idx_dtmap_verify + idx_dtmap_sync   src/index/dtmap.c:76-92,440-544
 *             dtmap_deletion                      src/index/dtmap.c:357-384
 *   on-disk ABI                                   src/index/storage.h:13-134
 *   BK-tree shape (insert in term-id order)       src/algo/bktree.c:160-217
 * The reverse index itself (dtmap_build_tdmap, dtmap.c:386-438) is built on
 * the GPU from the raw nxsdtmap image: see nxs_gpu.hip.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <errno.h>
#include <endian.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>

#include "nxs_impl.h"
#include "nxs_lev.h"

static inline uint16_t rd16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return be16toh(v); }
static inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return be32toh(v); }
static inline uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return be64toh(v); }

#define	TERMS_HDR_LEN	16	/* storage.h:40-53 */
#define	DTMAP_HDR_LEN	32	/* storage.h:100-123 */

static uint8_t *
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
		errno = EINVAL;
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

/* ---- term dictionary --------------------------------------------------- */

static uint64_t
hash_bytes(const uint8_t *p, size_t len)
{
	uint64_t h = 0x9e3779b97f4a7c15ULL ^ len;

	while (len >= 8) {
		uint64_t w;
		memcpy(&w, p, 8);
		h = (h ^ w) * 0xff51afd7ed558ccdULL;
		h ^= h >> 32;
		p += 8;
		len -= 8;
	}
	if (len) {
		uint64_t w = 0;
		memcpy(&w, p, len);
		h = (h ^ w) * 0xc4ceb9fe1a85ec53ULL;
		h ^= h >> 29;
	}
	h *= 0x9fb21c651e98df25ULL;
	return h ^ (h >> 32);
}

uint32_t
nxs_term_lookup(const nxs_index_t *idx, const uint8_t *val, size_t len)
{
	size_t i;

	if (!idx->thash_cap || len > UINT16_MAX) {
		return 0;
	}
	i = hash_bytes(val, len) & (idx->thash_cap - 1);
	while (idx->thash[i]) {
		const hterm_t *t = &idx->terms[idx->thash[i]];
		if (t->len == len && memcmp(t->val, val, len) == 0) {
			return idx->thash[i];
		}
		i = (i + 1) & (idx->thash_cap - 1);
	}
	return 0;
}

/* re-insert every live term into a table of `cap` slots */
static int
thash_rebuild(nxs_index_t *idx, size_t cap)
{
	uint32_t *h = calloc(cap, sizeof(uint32_t));

	if (!h) {
		return -1;
	}
	for (uint32_t id = 1; id <= idx->last_id; id++) {
		const hterm_t *t = &idx->terms[id];
		size_t i;

		if (t->tot_off == 0) {
			continue;
		}
		i = hash_bytes(t->val, t->len) & (cap - 1);
		while (h[i]) {
			i = (i + 1) & (cap - 1);
		}
		h[i] = id;
	}
	free(idx->thash);
	idx->thash = h;
	idx->thash_cap = cap;
	return 0;
}

/*
 * idx_terms_sync (terms.c:320-414): consume the term blocks in
 * [idx->terms_consumed, data_len) of the mapped nxsterms image.  Ids are file
 * order (terms.c:404); a duplicate string keeps its id unused
 * (idxterm_insert => EEXIST, idxterm.c:166-171).  Used by the first load and by
 * every refresh.
 */
static int
sync_terms(nxs_index_t *idx)
{
	const uint8_t *hdr = idx->tmap;
	size_t data_len, off;

	if (idx->tmap_len < TERMS_HDR_LEN || memcmp(hdr, "NXS_T", 5) != 0) {
		nxs_decl_err(idx->nxs, NXS_ERR_FATAL, "corrupted terms index header");
		return -1;
	}
	if (hdr[5] != 1) {
		nxs_decl_err(idx->nxs, NXS_ERR_FATAL,
		    "incompatible nxsearch index version");
		return -1;
	}
	data_len = be32toh(__atomic_load_n((const uint32_t *)(hdr + 8), __ATOMIC_ACQUIRE));
	if (TERMS_HDR_LEN + data_len > idx->tmap_len) {
		/* the reference maps in 32 KiB steps and fails likewise
		 * (idxmap.c:119-148) */
		nxs_decl_err(idx->nxs, NXS_ERR_SYSTEM, "terms mapping failed");
		return -1;
	}
	for (off = idx->terms_consumed; off < data_len; ) {
		const uint8_t *p = hdr + TERMS_HDR_LEN + off;
		const size_t remaining = data_len - off;
		size_t blk, i;
		uint16_t len;
		uint32_t id;
		bool dup = false;

		if (remaining < 2 || (len = rd16(p)) == 0) {
			nxs_decl_err(idx->nxs, NXS_ERR_FATAL, "corrupted terms index");
			return -1;
		}
		blk = ((2 + (size_t)len + 1 + 7) & ~(size_t)7) + 8;	/* storage.h:60-65 */
		if (blk > remaining) {
			nxs_decl_err(idx->nxs, NXS_ERR_FATAL, "corrupted terms index");
			return -1;
		}
		if (idx->last_id >= UINT32_MAX - 2) {
			nxs_decl_err(idx->nxs, NXS_ERR_LIMIT, "reached the term limit");
			return -1;
		}
		if ((size_t)idx->last_id + 2 >= idx->terms_cap) {
			const size_t ncap = idx->terms_cap ? idx->terms_cap * 2 : 1024;
			hterm_t *nt = realloc(idx->terms, ncap * sizeof(hterm_t));
			if (!nt) {
				nxs_decl_err(idx->nxs, NXS_ERR_SYSTEM, "out of memory");
				return -1;
			}
			memset(nt + idx->terms_cap, 0, (ncap - idx->terms_cap) * sizeof(hterm_t));
			idx->terms = nt;
			idx->terms_cap = ncap;
		}
		if (((size_t)idx->term_count + 2) * 2 > idx->thash_cap) {
			size_t cap = idx->thash_cap ? idx->thash_cap * 2 : 64;
			while (cap < ((size_t)idx->term_count + 2) * 2) {
				cap <<= 1;
			}
			if (thash_rebuild(idx, cap) == -1) {
				nxs_decl_err(idx->nxs, NXS_ERR_SYSTEM, "out of memory");
				return -1;
			}
		}
		id = ++idx->last_id;
		memset(&idx->terms[id], 0, sizeof(hterm_t));
		i = hash_bytes(p + 2, len) & (idx->thash_cap - 1);
		while (idx->thash[i]) {
			const hterm_t *t = &idx->terms[idx->thash[i]];
			if (t->len == len && memcmp(t->val, p + 2, len) == 0) {
				dup = true;
				break;
			}
			i = (i + 1) & (idx->thash_cap - 1);
		}
		if (!dup) {
			idx->thash[i] = id;
			idx->terms[id].val = p + 2;
			idx->terms[id].len = len;
			idx->terms[id].tot_off = (uint32_t)(TERMS_HDR_LEN + off + blk - 8);
			idx->term_count++;
		}
		off += blk;
	}
	idx->terms_consumed = off;
	return 0;
}

typedef struct {
	uint32_t	term;		/* term id */
	uint32_t	child;		/* head of the child list (node index + 1), ascending slot */
	uint32_t	sibling;	/* next sibling (node index + 1) */
	uint32_t	slot;		/* distance slot under the parent */
} bkn_t;

/* ---- BK-tree: persistent host tree + flattened image ---------------------- */

/*
 * The tree lives on the host as an array of nodes with child/sibling links so
 * that terms appended later can be inserted (bktree_insert in term-id order,
 * terms.c:404-405; bktree.c:160-217); children are kept in ascending slot
 * order.  The device works on a flattened BFS image (nxs_bk_flatten): "first
 * match in BFS push order" (Q7) = lowest image index.
 */
struct nxs_bktree {
	bkn_t *		nodes;
	uint32_t	n, cap;
};

nxs_bktree_t *
nxs_bktree_create(void)
{
	return calloc(1, sizeof(nxs_bktree_t));
}

void
nxs_bktree_destroy(nxs_bktree_t *bt)
{
	if (bt) {
		free(bt->nodes);
		free(bt);
	}
}

/* insert terms (from_id .. last_id]; 0, or -1 on out of memory */
int
nxs_bktree_insert(nxs_bktree_t *bt, const hterm_t *terms, uint32_t from_id, uint32_t last_id)
{
	uint64_t peq[256];

	memset(peq, 0, sizeof(peq));
	for (uint32_t id = from_id + 1; id <= last_id; id++) {
		const hterm_t *t = &terms[id];
		const bool bitpar = t->len <= NXS_MYERS_MAXPAT;
		uint32_t cur;

		if (t->tot_off == 0) {
			continue;	/* duplicate string: never inserted */
		}
		if (bt->n == bt->cap) {
			const uint32_t ncap = bt->cap ? bt->cap * 2 : 1024;
			bkn_t *nn = realloc(bt->nodes, (size_t)ncap * sizeof(bkn_t));
			if (!nn) {
				return -1;
			}
			memset(nn + bt->cap, 0, (size_t)(ncap - bt->cap) * sizeof(bkn_t));
			bt->nodes = nn;
			bt->cap = ncap;
		}
		if (bt->n == 0) {
			memset(&bt->nodes[0], 0, sizeof(bkn_t));
			bt->nodes[bt->n++].term = id;
			continue;
		}
		if (bitpar) {
			for (unsigned j = 0; j < t->len; j++) {
				peq[t->val[j]] |= UINT64_C(1) << j;
			}
		}
		cur = 0;
		for (;;) {
			bkn_t *nodes = bt->nodes;
			const hterm_t *o = &terms[nodes[cur].term];
			uint32_t c, prev = 0;
			int d;

			if (bitpar) {
				nxs_myers_t ms;
				nxs_myers_init(&ms, t->len);
				for (unsigned j = 0; j < o->len; j++) {
					nxs_myers_step(&ms, peq[o->val[j]]);
				}
				d = ms.score;
			} else {
				d = nxs_levdist_host(t->val, t->len, o->val, o->len);
			}
			if (d <= 0) {
				break;		/* EEXIST: bktree.c:182-189 */
			}
			if (d > 63) {
				d = 63;		/* BKT_DIST_LIMIT: bktree.c:196 */
			}
			/* children in ascending slot order */
			for (c = nodes[cur].child; c && nodes[c - 1].slot < (uint32_t)d; c = nodes[c - 1].sibling) {
				prev = c;
			}
			if (c && nodes[c - 1].slot == (uint32_t)d) {
				cur = c - 1;
				continue;
			}
			memset(&nodes[bt->n], 0, sizeof(bkn_t));
			nodes[bt->n].term = id;
			nodes[bt->n].slot = d;
			nodes[bt->n].sibling = c;
			if (prev) {
				nodes[prev - 1].sibling = bt->n + 1;
			} else {
				nodes[cur].child = bt->n + 1;
			}
			bt->n++;
			break;
		}
		if (bitpar) {
			for (unsigned j = 0; j < t->len; j++) {
				peq[t->val[j]] = 0;
			}
		}
	}
	return 0;
}

/* BFS numbering: children contiguous, ascending slot */
int
nxs_bk_flatten(const nxs_bktree_t *bt, const hterm_t *terms, const uint8_t *tmap, nxs_bkimage_t *out)
{
	const bkn_t *nodes = bt->nodes;
	const uint32_t n = bt->n;
	uint32_t *order = NULL, *level_of = NULL;
	uint64_t bytes_len = 0;

	memset(out, 0, sizeof(*out));
	if (n == 0) {
		return 0;
	}
	order = malloc((size_t)n * sizeof(uint32_t));	/* BFS rank -> build index */
	level_of = calloc(n, sizeof(uint32_t));
	out->nodes = calloc(n, sizeof(nxsgpu_bknode_t));
	if (!order || !level_of || !out->nodes) {
		free(order); free(level_of);
		free(out->nodes);
		out->nodes = NULL;
		return -1;
	}
	{
		uint32_t head = 0, tail = 0;
		order[tail++] = 0;
		while (head < tail) {
			const uint32_t rank = head, bi = order[head++];
			uint64_t bitmap = 0;

			out->nodes[rank].first_child = tail;
			for (uint32_t c = nodes[bi].child; c; c = nodes[c - 1].sibling) {
				bitmap |= UINT64_C(1) << nodes[c - 1].slot;
				level_of[tail] = level_of[rank] + 1;
				order[tail++] = c - 1;
			}
			out->nodes[rank].bitmap = bitmap;
			bytes_len += terms[nodes[bi].term].len;
		}
		out->depth = level_of[n - 1] + 1;
	}
	out->bytes = malloc(bytes_len + 16);
	if (!out->bytes) {
		free(order); free(level_of);
		free(out->nodes);
		out->nodes = NULL;
		return -1;
	}
	bytes_len = 0;
	for (uint32_t rank = 0; rank < n; rank++) {
		const uint32_t id = nodes[order[rank]].term;
		const hterm_t *t = &terms[id];
		nxsgpu_bknode_t *nd = &out->nodes[rank];

		nd->term_id = id;
		nd->str_off = (uint32_t)bytes_len;
		nd->str_len = t->len;
		/* idxterm_get_total(): idxterm.c:251-260 */
		nd->flags = rd64(tmap + t->tot_off) > 0 ? 1 : 0;
		memcpy(nd->inl, t->val, t->len < 8 ? t->len : 8);
		memcpy(out->bytes + bytes_len, t->val, t->len);
		bytes_len += t->len;
	}
	memset(out->bytes + bytes_len, 0, 16);
	out->bytes_len = bytes_len;
	out->n = n;
	free(order);
	free(level_of);
	return 0;
}

/* one-shot form (tests): tree over ids 1..last_id, flattened */
int
nxs_bk_build(const hterm_t *terms, uint32_t last_id, const uint8_t *tmap,
    nxs_bkimage_t *out)
{
	nxs_bktree_t *bt = nxs_bktree_create();
	int r = -1;

	memset(out, 0, sizeof(*out));
	if (bt && nxs_bktree_insert(bt, terms, 0, last_id) == 0) {
		r = nxs_bk_flatten(bt, terms, tmap, out);
	}
	nxs_bktree_destroy(bt);
	return r;
}

void
nxs_bk_free(nxs_bkimage_t *bk)
{
	free(bk->nodes);
	free(bk->bytes);
	memset(bk, 0, sizeof(*bk));
}

/* ---- doc blocks ---------------------------------------------------------- */

typedef struct {
	uint64_t	id;	/* 0 = dead */
	uint64_t	off;	/* block offset in the nxsdtmap image */
	uint32_t	n;	/* (term, count) pairs */
} ldoc_t;

typedef struct { uint64_t *k; uint32_t *v; size_t cap, n; } dmap_t;

static inline uint64_t
mix64(uint64_t x)
{
	x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
	x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
	return x ^ (x >> 33);
}

static void
dmap_put(dmap_t *m, uint64_t key, uint32_t val)
{
	size_t i;

	if ((m->n + 1) * 2 > m->cap) {
		dmap_t nm = { .cap = m->cap ? m->cap * 2 : 1024 };
		nm.k = calloc(nm.cap, sizeof(uint64_t));
		nm.v = calloc(nm.cap, sizeof(uint32_t));
		for (size_t j = 0; j < m->cap; j++) {
			if (m->k[j]) {
				dmap_put(&nm, m->k[j], m->v[j]);
			}
		}
		free(m->k);
		free(m->v);
		*m = nm;
	}
	i = mix64(key) & (m->cap - 1);
	while (m->k[i] && m->k[i] != key) {
		i = (i + 1) & (m->cap - 1);
	}
	if (!m->k[i]) {
		m->k[i] = key;
		m->n++;
	}
	m->v[i] = val;
}

/* returns index + 1, or 0 */
static uint32_t
dmap_get(const dmap_t *m, uint64_t key)
{
	size_t i;

	if (!m->cap) {
		return 0;
	}
	i = mix64(key) & (m->cap - 1);
	while (m->k[i]) {
		if (m->k[i] == key) {
			return m->v[i] + 1;
		}
		i = (i + 1) & (m->cap - 1);
	}
	return 0;
}

static int
cmp_ldoc(const void *a, const void *b)
{
	const ldoc_t *x = a, *y = b;
	return (x->id > y->id) - (x->id < y->id);
}

/*
 * Walk the doc blocks up to `stop_off` (exclusive) the way idx_dtmap_sync
 * consumes them; returns the live docs sorted by ascending doc id.
 */
static int
walk_dtmap(nxs_index_t *idx, uint64_t stop_off, ldoc_t **docs_out, uint64_t *n_out)
{
	const uint8_t *hdr = idx->dmap;
	const uint64_t data_len = rd64(hdr + 8);
	ldoc_t *docs = NULL;
	size_t n = 0, cap = 0;
	bool ascending = true;
	uint64_t last_id = 0;
	dmap_t map = { 0 };
	int ret = -1;

	for (uint64_t off = 0; off < data_len; ) {
		const uint8_t *p = hdr + DTMAP_HDR_LEN + off;
		const uint64_t remaining = data_len - off;
		uint64_t doc_id;
		uint32_t doc_len, np;

		if (DTMAP_HDR_LEN + off >= stop_off) {
			break;
		}
		if (remaining < 16) {
			nxs_decl_err(idx->nxs, NXS_ERR_FATAL, "corrupted dtmap index");
			goto out;
		}
		doc_id = rd64(p);
		doc_len = rd32(p + 8);
		np = rd32(p + 12);
		if ((uint64_t)np * 8 > remaining - 16) {
			nxs_decl_err(idx->nxs, NXS_ERR_FATAL, "corrupted dtmap index");
			goto out;
		}
		off += 16 + (uint64_t)np * 8;

		if (doc_id == 0) {
			continue;	/* deleted block: dtmap.c:364-367 */
		}
		if (doc_len == 0) {
			/* tombstone: drop the doc if it was loaded (dtmap.c:374-381) */
			uint32_t at = 0;
			if (ascending) {
				size_t lo = 0, hi = n;
				while (lo < hi) {
					const size_t mid = lo + (hi - lo) / 2;
					if (docs[mid].id < doc_id)
						lo = mid + 1;
					else
						hi = mid;
				}
				if (lo < n && docs[lo].id == doc_id &&
				    docs[lo].n != UINT32_MAX) {
					at = lo + 1;
				}
			} else {
				at = dmap_get(&map, doc_id);
				if (at && docs[at - 1].n == UINT32_MAX) {
					at = 0;
				}
			}
			if (at) {
				docs[at - 1].n = UINT32_MAX;	/* dead */
			}
			continue;
		}
		if (ascending && doc_id <= last_id && n) {
			/* out-of-order ids: switch to the hash-map bookkeeping */
			ascending = false;
			for (size_t i = 0; i < n; i++) {
				if (docs[i].n != UINT32_MAX) {
					dmap_put(&map, docs[i].id, (uint32_t)i);
				}
			}
		}
		if (!ascending) {
			const uint32_t at = dmap_get(&map, doc_id);
			if (at && docs[at - 1].n != UINT32_MAX) {
				/* idxdoc_create => EEXIST (idxdoc.c:41-45; dtmap.c:517-521) */
				nxs_decl_err(idx->nxs, NXS_ERR_SYSTEM, "idxdoc_create failed");
				goto out;
			}
		}
		if (n == cap) {
			cap = cap ? cap * 2 : 4096;
			docs = realloc(docs, cap * sizeof(ldoc_t));
			if (!docs) {
				nxs_decl_err(idx->nxs, NXS_ERR_SYSTEM, "out of memory");
				goto out;
			}
		}
		if (n >= UINT32_MAX - 1) {
			nxs_decl_err(idx->nxs, NXS_ERR_LIMIT, "too many documents");
			goto out;
		}
		docs[n].id = doc_id;
		docs[n].off = DTMAP_HDR_LEN + (off - 16 - (uint64_t)np * 8);
		docs[n].n = np;
		if (!ascending) {
			dmap_put(&map, doc_id, (uint32_t)n);
		}
		last_id = doc_id;
		n++;
	}

	/* compact the live docs; order by doc id (dense ordinals are ranks) */
	{
		size_t w = 0;
		for (size_t i = 0; i < n; i++) {
			if (docs[i].n != UINT32_MAX) {
				docs[w++] = docs[i];
			}
		}
		n = w;
		if (!ascending) {
			qsort(docs, n, sizeof(ldoc_t), cmp_ldoc);
		}
	}
	*docs_out = docs;
	*n_out = n;
	docs = NULL;
	ret = 0;
out:
	free(docs);
	free(map.k);
	free(map.v);
	return ret;
}

int
nxs_index_load(nxs_index_t *idx, const char *terms_path, const char *dtmap_path)
{
	nxs_bkimage_t bk = { 0 };
	nxsgpu_index_src_t src;
	uint64_t stop_off = UINT64_MAX;
	uint8_t *term_ok = NULL;
	int ret = -1;

	if (!idx->terms_path) {
		idx->terms_path = strdup(terms_path);
		idx->dtmap_path = strdup(dtmap_path);
	}
	if ((idx->tmap = map_file(terms_path, &idx->tmap_len)) == NULL) {
		nxs_decl_err(idx->nxs, NXS_ERR_SYSTEM,
		    "could not open terms index: %s", strerror(errno));
		return -1;
	}
	if ((idx->dmap = map_file(dtmap_path, &idx->dmap_len)) == NULL) {
		nxs_decl_err(idx->nxs, NXS_ERR_SYSTEM,
		    "could not open dtmap index: %s", strerror(errno));
		return -1;
	}
	idx->terms_consumed = 0;
	if (sync_terms(idx) == -1) {
		return -1;
	}
	if (idx->dmap_len < DTMAP_HDR_LEN || memcmp(idx->dmap, "NXS_D", 5) != 0) {
		nxs_decl_err(idx->nxs, NXS_ERR_FATAL, "corrupted dtmap index header");
		return -1;
	}
	if (idx->dmap[5] != 1) {
		nxs_decl_err(idx->nxs, NXS_ERR_FATAL,
		    "incompatible nxsearch index version");
		return -1;
	}
	if (DTMAP_HDR_LEN + rd64(idx->dmap + 8) > idx->dmap_len) {
		nxs_decl_err(idx->nxs, NXS_ERR_SYSTEM, "dtmap mapping failed");
		return -1;
	}
	if ((idx->bktree = nxs_bktree_create()) == NULL ||
	    nxs_bktree_insert(idx->bktree, idx->terms, 0, idx->last_id) == -1 ||
	    nxs_bk_flatten(idx->bktree, idx->terms, idx->tmap, &bk) == -1) {
		nxs_decl_err(idx->nxs, NXS_ERR_SYSTEM, "out of memory");
		return -1;
	}
	idx->bk_upto = idx->last_id;
	term_ok = calloc((size_t)idx->last_id + 1, 1);
	for (uint32_t id = 1; id <= idx->last_id; id++) {
		term_ok[id] = idx->terms[id].tot_off != 0;
	}

	if (nxsgpu_device_count() <= 0) {
		nxs_decl_err(idx->nxs, NXS_ERR_SYSTEM,
		    "no HIP device: the MI355X query path cannot run");
		goto out;
	}

	for (int attempt = 0; attempt < 3; attempt++) {
		const uint64_t data_len = be64toh(__atomic_load_n((const uint64_t *)(idx->dmap + 8), __ATOMIC_ACQUIRE));
		ldoc_t *docs = NULL;
		uint64_t n = 0, *pair_base, bad;
		const char *dev_env = getenv("NXS_GPU_DEVICE");

		if (walk_dtmap(idx, stop_off, &docs, &n) == -1) {
			goto out;
		}
		if (idx->n_shards > 1) {
			/* N4: this shard's slice of the docs (ascending doc id) */
			const uint64_t lo = n * idx->shard / idx->n_shards;
			const uint64_t hi = n * ((uint64_t)idx->shard + 1) / idx->n_shards;
			memmove(docs, docs + lo, (hi - lo) * sizeof(ldoc_t));
			n = hi - lo;
		}
		/* the doc table stays: ordinal = rank in ascending doc id */
		free(idx->h_doc_ids); free(idx->h_blk_off); free(idx->h_npairs); free(idx->h_alive);
		idx->cap_ord = n + n / 8 + 1024;
		idx->h_doc_ids = malloc(idx->cap_ord * sizeof(uint64_t));
		idx->h_blk_off = malloc(idx->cap_ord * sizeof(uint64_t));
		idx->h_npairs = malloc(idx->cap_ord * sizeof(uint32_t));
		idx->h_alive = malloc(idx->cap_ord);
		pair_base = malloc((n + 1) * sizeof(uint64_t));
		if (!idx->h_doc_ids || !idx->h_blk_off || !idx->h_npairs || !idx->h_alive || !pair_base) {
			free(docs);
			free(pair_base);
			nxs_decl_err(idx->nxs, NXS_ERR_SYSTEM, "out of memory");
			goto out;
		}
		pair_base[0] = 0;
		for (uint64_t i = 0; i < n; i++) {
			idx->h_blk_off[i] = docs[i].off;
			idx->h_doc_ids[i] = docs[i].id;
			idx->h_npairs[i] = docs[i].n;
			idx->h_alive[i] = 1;
			pair_base[i + 1] = pair_base[i] + docs[i].n;
		}
		idx->n_ord = n;
		free(docs);

		memset(&src, 0, sizeof(src));
		src.dtmap_img = idx->dmap;
		src.dtmap_len = DTMAP_HDR_LEN + data_len;
		src.blk_off = idx->h_blk_off;
		src.doc_ids = idx->h_doc_ids;
		src.pair_base = pair_base;
		src.n_docs = n;
		src.n_terms = idx->last_id;
		src.term_ok = term_ok;
		src.hdr_doc_count = rd32(idx->dmap + 24);	/* dtmap.c:671-677 */
		src.hdr_token_count = rd64(idx->dmap + 16);	/* dtmap.c:660-666 */
		src.bk_nodes = bk.nodes;
		src.n_bk = bk.n;
		src.bk_depth = bk.depth;
		src.bk_bytes = bk.bytes;
		src.bk_bytes_len = bk.bytes_len;
		src.default_algo = idx->algo;	/* the other ranking function's impacts: on first use */

		idx->device = idx->want_device ? idx->want_device - 1 : dev_env ? atoi(dev_env) : 0;
		idx->dev = nxsgpu_index_create(idx->device, &src);
		free(pair_base);
		if (!idx->dev) {
			nxs_decl_err(idx->nxs, NXS_ERR_SYSTEM, "device index build failed: %s",
			    nxsgpu_last_error());
			goto out;
		}
		bad = nxsgpu_index_first_bad_doc(idx->dev);
		if (bad == UINT64_MAX) {
			idx->n_docs = n;
			/* what has been consumed: everything, or up to the block that names
			 * a term nxsterms does not hold yet (retried by the next refresh) */
			idx->dt_consumed = stop_off == UINT64_MAX ? data_len : stop_off - DTMAP_HDR_LEN;
			idx->hdr_docs_seen = src.hdr_doc_count;
			idx->hdr_tokens_seen = src.hdr_token_count;
			idx->bk_flags_stale = false;
			ret = 0;
			break;
		}
		/*
		 * A block names a term that nxsterms does not hold: the
		 * reference stops consuming there (DTMAP_PARTIAL_SYNC at open,
		 * dtmap.c:143,527-535).  Re-walk up to that block.
		 */
		nxsgpu_index_destroy(idx->dev);
		idx->dev = NULL;
		stop_off = bad;
	}
	if (ret != 0 && idx->nxs->errcode == NXS_ERR_SUCCESS) {
		nxs_decl_err(idx->nxs, NXS_ERR_FATAL, "dtmap index is inconsistent");
	}
out:
	free(term_ok);
	nxs_bk_free(&bk);
	return ret;
}

static void
unload_snapshot(nxs_index_t *idx)
{
	if (idx->dev) {
		nxsgpu_index_destroy(idx->dev);
	}
	free(idx->terms);
	free(idx->thash);
	nxs_bktree_destroy(idx->bktree);
	free(idx->h_doc_ids);
	free(idx->h_blk_off);
	free(idx->h_npairs);
	free(idx->h_alive);
	if (idx->tmap) munmap(idx->tmap, idx->tmap_len);
	if (idx->dmap) munmap(idx->dmap, idx->dmap_len);
	idx->dev = NULL;
	idx->terms = NULL;
	idx->terms_cap = 0;
	idx->thash = NULL;
	idx->thash_cap = 0;
	idx->bktree = NULL;
	idx->bk_upto = 0;
	idx->h_doc_ids = idx->h_blk_off = NULL;
	idx->h_npairs = NULL;
	idx->h_alive = NULL;
	idx->n_ord = idx->cap_ord = 0;
	idx->tmap = idx->dmap = NULL;
	idx->last_id = idx->term_count = 0;
	idx->terms_consumed = idx->dt_consumed = 0;
	idx->n_docs = 0;
}

void
nxs_index_unload(nxs_index_t *idx)
{
	unload_snapshot(idx);
	free(idx->terms_path);
	free(idx->dtmap_path);
	idx->terms_path = idx->dtmap_path = NULL;
}

/* the file grew beyond the mapping: map it again; term pointers move along */
static int
remap_if_grown(nxs_index_t *idx)
{
	struct stat sb;

	/* mremap extends the MAP_SHARED file mapping without tearing the old pages
	 * down (the reference grows its files in 32 KiB steps, index.h:24: this
	 * happens often) */
	if (stat(idx->terms_path, &sb) == 0 && (size_t)sb.st_size > idx->tmap_len) {
		uint8_t *nm = mremap(idx->tmap, idx->tmap_len, (size_t)sb.st_size, MREMAP_MAYMOVE);
		if (nm == MAP_FAILED) {
			return -1;
		}
		if (nm != idx->tmap) {
			for (uint32_t id = 1; id <= idx->last_id; id++) {
				if (idx->terms[id].tot_off) {
					idx->terms[id].val = nm + (idx->terms[id].val - idx->tmap);
				}
			}
		}
		idx->tmap = nm;
		idx->tmap_len = (size_t)sb.st_size;
	}
	if (stat(idx->dtmap_path, &sb) == 0 && (size_t)sb.st_size > idx->dmap_len) {
		uint8_t *nm = mremap(idx->dmap, idx->dmap_len, (size_t)sb.st_size, MREMAP_MAYMOVE);
		if (nm == MAP_FAILED) {
			return -1;
		}
		idx->dmap = nm;
		idx->dmap_len = (size_t)sb.st_size;
	}
	return 0;
}

/* the whole snapshot again, built beside the old one and swapped in on success */
static int
refresh_rebuild(nxs_index_t *idx)
{
	nxs_index_t tmp;
	const nxs_err_t saved_code = idx->nxs->errcode;

	memset(&tmp, 0, sizeof(tmp));
	tmp.nxs = idx->nxs;
	tmp.algo = idx->algo;
	tmp.lowercase = idx->lowercase;
	tmp.terms_path = idx->terms_path;
	tmp.dtmap_path = idx->dtmap_path;
	if (nxs_index_load(&tmp, idx->terms_path, idx->dtmap_path) != 0) {
		/*
		 * The files cannot be consumed right now (a writer in the middle of
		 * an append): the old snapshot keeps serving, as the reference's
		 * partial sync does (dtmap.c:527-535); the next search tries again.
		 */
		unload_snapshot(&tmp);
		if (saved_code == NXS_ERR_SUCCESS) {
			nxs_clear_error(idx->nxs);
		}
		return 0;
	}
	if (idx->comm) {
		(void)nxsgpu_index_set_comm(idx->dev, NULL);
	}
	unload_snapshot(idx);
	idx->tmap = tmp.tmap;	idx->tmap_len = tmp.tmap_len;
	idx->dmap = tmp.dmap;	idx->dmap_len = tmp.dmap_len;
	idx->terms = tmp.terms;
	idx->terms_cap = tmp.terms_cap;
	idx->last_id = tmp.last_id;
	idx->term_count = tmp.term_count;
	idx->thash = tmp.thash;
	idx->thash_cap = tmp.thash_cap;
	idx->bktree = tmp.bktree;
	idx->bk_upto = tmp.bk_upto;
	idx->bk_flags_stale = false;
	idx->h_doc_ids = tmp.h_doc_ids;
	idx->h_blk_off = tmp.h_blk_off;
	idx->h_npairs = tmp.h_npairs;
	idx->h_alive = tmp.h_alive;
	idx->n_ord = tmp.n_ord;
	idx->cap_ord = tmp.cap_ord;
	idx->n_docs = tmp.n_docs;
	idx->dev = tmp.dev;
	idx->device = tmp.device;
	idx->terms_consumed = tmp.terms_consumed;
	idx->dt_consumed = tmp.dt_consumed;
	idx->hdr_docs_seen = tmp.hdr_docs_seen;
	idx->hdr_tokens_seen = tmp.hdr_tokens_seen;
	if (idx->comm) {
		(void)nxsgpu_index_set_comm(idx->dev, idx->comm);
	}
	idx->n_rebuilds++;
	return 0;
}

/* ordinal of a loaded doc id, or -1 */
static int64_t
ord_of(const nxs_index_t *idx, uint64_t doc_id)
{
	uint64_t lo = 0, hi = idx->n_ord;

	while (lo < hi) {
		const uint64_t mid = lo + (hi - lo) / 2;
		if (idx->h_doc_ids[mid] < doc_id) lo = mid + 1; else hi = mid;
	}
	return (lo < idx->n_ord && idx->h_doc_ids[lo] == doc_id) ? (int64_t)lo : -1;
}

/*
 * The reference re-syncs appended data before every search (search.c:309-312:
 * idx_terms_sync + idx_dtmap_sync(PARTIAL)).  Here: consume the term blocks and
 * doc blocks published since the last sync (terms.c:320-414, dtmap.c:440-544 --
 * new docs, tombstones of removed ones, partial sync when a block names a term
 * that is not visible yet) and merge the delta into the device index
 * (nxsgpu_index_apply): no re-read of the forward index, no re-sort, the BK-tree
 * re-flattened lazily (nxs_index_bk_sync).  Appended docs must carry ids above
 * every loaded one (what an indexer with growing ids produces); anything else --
 * a re-used or out-of-order id -- takes the full rebuild.  Returns 0 or -1.
 */
/*
 * Did another process publish anything since the last sync?  The four words
 * idx_terms_sync / idx_dtmap_sync look at (terms.c:320-330, dtmap.c:440-470):
 * both data_len fields (published last, with release semantics) and the header
 * counters.  Cheap enough for every search.
 */
bool
nxs_index_changed(const nxs_index_t *idx)
{
	if (!idx->tmap || !idx->dmap || idx->n_shards > 1) {
		return false;
	}
	const uint64_t t_now = be32toh(__atomic_load_n((const uint32_t *)(idx->tmap + 8), __ATOMIC_ACQUIRE));
	const uint64_t d_now = be64toh(__atomic_load_n((const uint64_t *)(idx->dmap + 8), __ATOMIC_ACQUIRE));

	return t_now != idx->terms_consumed || d_now != idx->dt_consumed ||
	    rd32(idx->dmap + 24) != idx->hdr_docs_seen || rd64(idx->dmap + 16) != idx->hdr_tokens_seen;
}

static double
dbg_ms(void)
{
	struct timespec ts;

	clock_gettime(CLOCK_MONOTONIC, &ts);
	return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec;
}

int
nxs_index_refresh(nxs_index_t *idx)
{
	double t_dbg[3] = { dbg_ms(), 0, 0 };
	uint64_t t_now, d_now, hd_docs, hd_tokens;
	uint64_t *nb_off = NULL, *nb_ids = NULL, *nb_base = NULL;
	uint32_t *dead_term = NULL, *dead_ord = NULL, *dead_list = NULL;
	uint8_t *term_ok = NULL;
	size_t n_new = 0, cap_new = 0, n_deadp = 0, cap_deadp = 0, n_dead = 0, cap_dead = 0;
	uint64_t off, consumed_to, max_id;
	nxsgpu_index_delta_t dl;
	const uint32_t old_last_id = idx->last_id;
	bool rebuild = false;
	int ret = -1;

	if (!idx->tmap || !idx->dmap) {
		return -1;
	}
	/* data_len is published last, with release semantics (terms.c:303-305,
	 * dtmap.c:333-337): an acquire load pairs with it */
	t_now = be32toh(__atomic_load_n((const uint32_t *)(idx->tmap + 8), __ATOMIC_ACQUIRE));
	d_now = be64toh(__atomic_load_n((const uint64_t *)(idx->dmap + 8), __ATOMIC_ACQUIRE));
	hd_docs = rd32(idx->dmap + 24);
	hd_tokens = rd64(idx->dmap + 16);
	if (t_now == idx->terms_consumed && d_now == idx->dt_consumed &&
	    hd_docs == idx->hdr_docs_seen && hd_tokens == idx->hdr_tokens_seen) {
		return 0;
	}
	if (idx->n_shards > 1) {
		return 0;	/* a doc shard is a static snapshot (include/nxs.h) */
	}
	if (idx->dev && nxsgpu_batches_in_flight(idx->dev) > 0) {
		return 0;	/* the device arrays are in use: between batches only */
	}
	if (remap_if_grown(idx) == -1) {
		return refresh_rebuild(idx);
	}
	if (t_now < idx->terms_consumed || d_now < idx->dt_consumed ||
	    TERMS_HDR_LEN + t_now > idx->tmap_len || DTMAP_HDR_LEN + d_now > idx->dmap_len) {
		return refresh_rebuild(idx);	/* the files were replaced */
	}
	if (sync_terms(idx) == -1) {
		nxs_clear_error(idx->nxs);
		return refresh_rebuild(idx);
	}

	/* idx_dtmap_sync over the new blocks */
	max_id = idx->n_ord ? idx->h_doc_ids[idx->n_ord - 1] : 0;
	consumed_to = idx->dt_consumed;
	for (off = idx->dt_consumed; off < d_now && !rebuild; ) {
		const uint8_t *p = idx->dmap + DTMAP_HDR_LEN + off;
		const uint64_t remaining = d_now - off;
		uint64_t doc_id;
		uint32_t doc_len, np;
		bool ok = true;

		if (remaining < 16) {
			rebuild = true;
			break;
		}
		doc_id = rd64(p);
		doc_len = rd32(p + 8);
		np = rd32(p + 12);
		if ((uint64_t)np * 8 > remaining - 16) {
			rebuild = true;
			break;
		}
		if (doc_id == 0) {
			/* deleted block: dtmap.c:364-367 */
		} else if (doc_len == 0) {
			/* tombstone (dtmap.c:374-381): drop the doc if it is loaded */
			const int64_t o = ord_of(idx, doc_id);
			bool pending = false;

			for (size_t i = 0; i < n_new && !pending; i++) {
				pending = nb_ids[i] == doc_id;
			}
			if (pending) {
				rebuild = true;		/* added and removed within one delta */
			} else if (o >= 0 && idx->h_alive[o]) {
				const uint8_t *blk = idx->dmap + idx->h_blk_off[o];
				const uint32_t n = idx->h_npairs[o];

				if (n_dead == cap_dead) {
					cap_dead = cap_dead ? cap_dead * 2 : 64;
					dead_list = realloc(dead_list, cap_dead * sizeof(uint32_t));
				}
				dead_list[n_dead++] = (uint32_t)o;
				idx->h_alive[o] = 2;	/* dying: committed below */
				if (n_deadp + n > cap_deadp) {
					cap_deadp = (n_deadp + n) * 2 + 64;
					dead_term = realloc(dead_term, cap_deadp * sizeof(uint32_t));
					dead_ord = realloc(dead_ord, cap_deadp * sizeof(uint32_t));
				}
				/* the removed doc's block still holds its pairs (only the doc
				 * id was zeroed: dtmap.c:603) */
				for (uint32_t j = 0; j < n; j++) {
					dead_term[n_deadp] = rd32(blk + 16 + 8 * (size_t)j);
					dead_ord[n_deadp] = (uint32_t)o;
					n_deadp++;
				}
			}
		} else {
			const int64_t o = ord_of(idx, doc_id);

			if (doc_id <= max_id || (o >= 0 && idx->h_alive[o])) {
				rebuild = true;		/* out-of-order or re-used id */
				break;
			}
			/* every pair must name a visible, live term: else stop here
			 * (partial sync, dtmap.c:406-413,527-535) */
			for (uint32_t j = 0; j < np && ok; j++) {
				const uint32_t tid = rd32(p + 16 + 8 * (size_t)j);
				ok = tid != 0 && tid <= idx->last_id && idx->terms[tid].tot_off != 0;
			}
			if (!ok) {
				break;
			}
			if (n_new == cap_new) {
				cap_new = cap_new ? cap_new * 2 : 64;
				nb_off = realloc(nb_off, cap_new * sizeof(uint64_t));
				nb_ids = realloc(nb_ids, cap_new * sizeof(uint64_t));
				nb_base = realloc(nb_base, (cap_new + 1) * sizeof(uint64_t));
			}
			if (n_new == 0) {
				if (!nb_base) {
					nb_base = malloc(2 * sizeof(uint64_t));
				}
				nb_base[0] = 0;
			}
			nb_off[n_new] = DTMAP_HDR_LEN + off;
			nb_ids[n_new] = doc_id;
			nb_base[n_new + 1] = nb_base[n_new] + np;
			n_new++;
			max_id = doc_id;
		}
		off += 16 + (uint64_t)np * 8;
		consumed_to = off;
	}
	if (rebuild || idx->n_ord + n_new >= UINT32_MAX - 1) {
		for (size_t i = 0; i < n_dead; i++) {
			idx->h_alive[dead_list[i]] = 1;
		}
		ret = refresh_rebuild(idx);
		goto out;
	}

	term_ok = calloc((size_t)idx->last_id + 1, 1);
	for (uint32_t id = 1; id <= idx->last_id; id++) {
		term_ok[id] = idx->terms[id].tot_off != 0;
	}
	memset(&dl, 0, sizeof(dl));
	dl.n_terms = idx->last_id;
	dl.term_ok = term_ok;
	dl.dtmap_img = idx->dmap;
	dl.dtmap_len = DTMAP_HDR_LEN + d_now;
	dl.blk_off = nb_off;
	dl.doc_ids = nb_ids;
	dl.pair_base = nb_base;
	dl.n_new = n_new;
	dl.dead_term = dead_term;
	dl.dead_ord = dead_ord;
	dl.n_dead_pairs = n_deadp;
	dl.hdr_doc_count = (uint32_t)hd_docs;
	dl.hdr_token_count = hd_tokens;
	t_dbg[1] = dbg_ms();
	if (nxsgpu_index_apply(idx->dev, &dl) != 0) {
		for (size_t i = 0; i < n_dead; i++) {
			idx->h_alive[dead_list[i]] = 1;
		}
		ret = refresh_rebuild(idx);
		goto out;
	}
	t_dbg[2] = dbg_ms();
	/* commit the host tables */
	for (size_t i = 0; i < n_dead; i++) {
		idx->h_alive[dead_list[i]] = 0;
	}
	if (idx->n_ord + n_new > idx->cap_ord) {
		idx->cap_ord = (idx->n_ord + n_new) * 2 + 1024;
		idx->h_doc_ids = realloc(idx->h_doc_ids, idx->cap_ord * sizeof(uint64_t));
		idx->h_blk_off = realloc(idx->h_blk_off, idx->cap_ord * sizeof(uint64_t));
		idx->h_npairs = realloc(idx->h_npairs, idx->cap_ord * sizeof(uint32_t));
		idx->h_alive = realloc(idx->h_alive, idx->cap_ord);
	}
	for (size_t i = 0; i < n_new; i++) {
		idx->h_doc_ids[idx->n_ord] = nb_ids[i];
		idx->h_blk_off[idx->n_ord] = nb_off[i];
		idx->h_npairs[idx->n_ord] = (uint32_t)(nb_base[i + 1] - nb_base[i]);
		idx->h_alive[idx->n_ord] = 1;
		idx->n_ord++;
	}
	idx->n_docs = idx->n_ord;
	idx->dt_consumed = consumed_to;
	idx->hdr_docs_seen = hd_docs;
	idx->hdr_tokens_seen = hd_tokens;
	/* term totals moved (dtmap.c:236,626): the "total > 0" flags of the BK image
	 * and, with new terms, the image itself are brought up to date before the
	 * next fuzzy search (nxs_index_bk_sync) */
	idx->bk_flags_stale = true;
	(void)old_last_id;
	idx->n_incremental++;
	ret = 0;
	if (getenv("NXS_GPU_DEBUG_TIMING")) {
		fprintf(stderr, "[nxs refresh] host walk %.1f ms, device apply %.1f ms, commit %.1f ms\n",
		    t_dbg[1] - t_dbg[0], t_dbg[2] - t_dbg[1], dbg_ms() - t_dbg[2]);
	}
out:
	free(nb_off);
	free(nb_ids);
	free(nb_base);
	free(dead_term);
	free(dead_ord);
	free(dead_list);
	free(term_ok);
	return ret;
}

/*
 * The BK-tree image the device searches, brought up to date before a fuzzy
 * pass: terms appended since the last image are inserted into the host tree
 * (bktree_insert order = term-id order, terms.c:404-405) and the image is
 * flattened and uploaded again; if only totals moved, the "total > 0" flags
 * (idxterm.c:238-242,251-260) are re-read and the image goes up only when one
 * of them changed.
 */
int
nxs_index_bk_sync(nxs_index_t *idx)
{
	nxs_bkimage_t bk = { 0 };
	int ret = -1;

	if (idx->bk_upto == idx->last_id && !idx->bk_flags_stale) {
		return 0;
	}
	if (!idx->bktree && (idx->bktree = nxs_bktree_create()) == NULL) {
		return -1;
	}
	if (nxs_bktree_insert(idx->bktree, idx->terms, idx->bk_upto, idx->last_id) == -1 ||
	    nxs_bk_flatten(idx->bktree, idx->terms, idx->tmap, &bk) == -1) {
		nxs_decl_err(idx->nxs, NXS_ERR_SYSTEM, "out of memory");
		goto out;
	}
	if (nxsgpu_index_set_bk(idx->dev, bk.nodes, bk.n, bk.depth, bk.bytes, bk.bytes_len) != 0) {
		nxs_decl_err(idx->nxs, NXS_ERR_SYSTEM, "%s", nxsgpu_last_error());
		goto out;
	}
	idx->bk_upto = idx->last_id;
	idx->bk_flags_stale = false;
	ret = 0;
out:
	nxs_bk_free(&bk);
	return ret;
}

/* refreshes that took the incremental path / the full rebuild (tests, bench) */
void
nxs_index_refresh_stats(const nxs_index_t *idx, uint64_t out[2])
{
	out[0] = idx->n_incremental;
	out[1] = idx->n_rebuilds;
}

/* exact Levenshtein of the host side (BK build), exported for the tests */
int
nxs_levdist_export(const uint8_t *a, size_t n, const uint8_t *b, size_t m)
{
	return nxs_levdist_host(a, n, b, m);
}

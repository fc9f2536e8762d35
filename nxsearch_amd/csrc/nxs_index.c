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

static int
load_terms(nxs_index_t *idx)
{
	const uint8_t *hdr = idx->tmap;
	size_t data_len, off, cap = 0, count = 0;

	if (idx->tmap_len < TERMS_HDR_LEN || memcmp(hdr, "NXS_T", 5) != 0) {
		nxs_decl_err(idx->nxs, NXS_ERR_FATAL, "corrupted terms index header");
		return -1;
	}
	if (hdr[5] != 1) {
		nxs_decl_err(idx->nxs, NXS_ERR_FATAL,
		    "incompatible nxsearch index version");
		return -1;
	}
	data_len = rd32(hdr + 8);
	if (TERMS_HDR_LEN + data_len > idx->tmap_len) {
		/* the reference maps in 32 KiB steps and fails likewise
		 * (idxmap.c:119-148) */
		nxs_decl_err(idx->nxs, NXS_ERR_SYSTEM, "terms mapping failed");
		return -1;
	}

	/* pass 1: count the blocks */
	for (off = 0; off < data_len; ) {
		const size_t remaining = data_len - off;
		size_t blk;
		uint16_t len;

		if (remaining < 2 || (len = rd16(hdr + TERMS_HDR_LEN + off)) == 0) {
			nxs_decl_err(idx->nxs, NXS_ERR_FATAL, "corrupted terms index");
			return -1;
		}
		blk = ((2 + (size_t)len + 1 + 7) & ~(size_t)7) + 8;	/* storage.h:60-65 */
		if (blk > remaining) {
			nxs_decl_err(idx->nxs, NXS_ERR_FATAL, "corrupted terms index");
			return -1;
		}
		off += blk;
		count++;
	}
	if (count >= UINT32_MAX) {
		nxs_decl_err(idx->nxs, NXS_ERR_LIMIT, "reached the term limit");
		return -1;
	}
	idx->terms = calloc(count + 2, sizeof(hterm_t));
	cap = 64;
	while (cap < count * 2 + 2) {
		cap <<= 1;
	}
	idx->thash = calloc(cap, sizeof(uint32_t));
	idx->thash_cap = cap;
	if (!idx->terms || !idx->thash) {
		nxs_decl_err(idx->nxs, NXS_ERR_SYSTEM, "out of memory");
		return -1;
	}

	/* pass 2: ids in file order; a duplicate string keeps its id unused */
	for (off = 0; off < data_len; ) {
		const uint8_t *p = hdr + TERMS_HDR_LEN + off;
		const uint16_t len = rd16(p);
		const size_t blk = ((2 + (size_t)len + 1 + 7) & ~(size_t)7) + 8;
		const uint32_t id = ++idx->last_id;
		size_t i = hash_bytes(p + 2, len) & (cap - 1);
		bool dup = false;

		while (idx->thash[i]) {
			const hterm_t *t = &idx->terms[idx->thash[i]];
			if (t->len == len && memcmp(t->val, p + 2, len) == 0) {
				dup = true;
				break;
			}
			i = (i + 1) & (cap - 1);
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
	return 0;
}

/* ---- BK-tree image ------------------------------------------------------ */

typedef struct {
	uint32_t	term;		/* term id */
	uint32_t	child;		/* head of the child list (node index + 1) */
	uint32_t	sibling;	/* next sibling (node index + 1) */
	uint32_t	slot;		/* distance slot under the parent */
} bkn_t;

static int
cmp_slot(const void *a, const void *b)
{
	const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
	return (x > y) - (x < y);
}

int
nxs_bk_build(const hterm_t *terms, uint32_t last_id, const uint8_t *tmap,
    nxs_bkimage_t *out)
{
	bkn_t *nodes = calloc((size_t)last_id + 1, sizeof(bkn_t));
	uint64_t peq[256];
	uint32_t n = 0, *order = NULL, *first = NULL, *level_of = NULL;
	uint64_t bytes_len = 0;

	memset(out, 0, sizeof(*out));
	memset(peq, 0, sizeof(peq));
	if (!nodes) {
		return -1;
	}

	/* bktree_insert in term-id order (terms.c:404-405; bktree.c:160-217) */
	for (uint32_t id = 1; id <= last_id; id++) {
		const hterm_t *t = &terms[id];
		const bool bitpar = t->len <= NXS_MYERS_MAXPAT;
		uint32_t cur;

		if (t->tot_off == 0) {
			continue;	/* duplicate string: never inserted */
		}
		if (n == 0) {
			nodes[n++].term = id;
			continue;
		}
		if (bitpar) {
			for (unsigned j = 0; j < t->len; j++) {
				peq[t->val[j]] |= UINT64_C(1) << j;
			}
		}
		cur = 0;
		for (;;) {
			const hterm_t *o = &terms[nodes[cur].term];
			uint32_t c;
			int d;

			if (bitpar) {
				nxs_myers_t s;
				nxs_myers_init(&s, t->len);
				for (unsigned j = 0; j < o->len; j++) {
					nxs_myers_step(&s, peq[o->val[j]]);
				}
				d = s.score;
			} else {
				d = nxs_levdist_host(t->val, t->len, o->val, o->len);
			}
			if (d <= 0) {
				break;		/* EEXIST: bktree.c:182-189 */
			}
			if (d > 63) {
				d = 63;		/* BKT_DIST_LIMIT: bktree.c:196 */
			}
			for (c = nodes[cur].child; c; c = nodes[c - 1].sibling) {
				if (nodes[c - 1].slot == (uint32_t)d) {
					break;
				}
			}
			if (c) {
				cur = c - 1;
				continue;
			}
			nodes[n].term = id;
			nodes[n].slot = d;
			nodes[n].sibling = nodes[cur].child;
			nodes[cur].child = n + 1;
			n++;
			break;
		}
		if (bitpar) {
			for (unsigned j = 0; j < t->len; j++) {
				peq[t->val[j]] = 0;
			}
		}
	}
	if (n == 0) {
		free(nodes);
		return 0;
	}

	/* BFS numbering: children contiguous, ascending slot */
	order = malloc((size_t)n * sizeof(uint32_t));	/* BFS rank -> build index */
	first = calloc(n, sizeof(uint32_t));
	level_of = calloc(n, sizeof(uint32_t));
	out->nodes = calloc(n, sizeof(nxsgpu_bknode_t));
	if (!order || !first || !level_of || !out->nodes) {
		free(nodes); free(order); free(first); free(level_of);
		free(out->nodes);
		return -1;
	}
	{
		uint32_t head = 0, tail = 0;
		order[tail++] = 0;
		while (head < tail) {
			const uint32_t rank = head, bi = order[head++];
			uint64_t kids[64];
			unsigned nk = 0;
			uint64_t bitmap = 0;

			for (uint32_t c = nodes[bi].child; c; c = nodes[c - 1].sibling) {
				kids[nk++] = ((uint64_t)nodes[c - 1].slot << 32) | (c - 1);
			}
			qsort(kids, nk, sizeof(uint64_t), cmp_slot);
			first[rank] = tail;
			for (unsigned k = 0; k < nk; k++) {
				bitmap |= UINT64_C(1) << (kids[k] >> 32);
				level_of[tail] = level_of[rank] + 1;
				order[tail++] = (uint32_t)kids[k];
			}
			out->nodes[rank].bitmap = bitmap;
			out->nodes[rank].first_child = first[rank];
			bytes_len += terms[nodes[bi].term].len;
		}
		out->depth = level_of[n - 1] + 1;
	}
	out->bytes = malloc(bytes_len + 16);
	if (!out->bytes) {
		free(nodes); free(order); free(first); free(level_of);
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
	free(nodes);
	free(order);
	free(first);
	free(level_of);
	return 0;
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
	if (load_terms(idx) == -1) {
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
	if (nxs_bk_build(idx->terms, idx->last_id, idx->tmap, &bk) == -1) {
		nxs_decl_err(idx->nxs, NXS_ERR_SYSTEM, "out of memory");
		return -1;
	}
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
		ldoc_t *docs = NULL;
		uint64_t n = 0, *blk_off, *doc_ids, *pair_base, bad;
		const char *dev_env = getenv("NXS_GPU_DEVICE");

		if (walk_dtmap(idx, stop_off, &docs, &n) == -1) {
			goto out;
		}
		blk_off = malloc((n + 1) * sizeof(uint64_t));
		doc_ids = malloc((n + 1) * sizeof(uint64_t));
		pair_base = malloc((n + 1) * sizeof(uint64_t));
		pair_base[0] = 0;
		for (uint64_t i = 0; i < n; i++) {
			blk_off[i] = docs[i].off;
			doc_ids[i] = docs[i].id;
			pair_base[i + 1] = pair_base[i] + docs[i].n;
		}
		free(docs);

		memset(&src, 0, sizeof(src));
		src.dtmap_img = idx->dmap;
		src.dtmap_len = DTMAP_HDR_LEN + rd64(idx->dmap + 8);
		src.blk_off = blk_off;
		src.doc_ids = doc_ids;
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

		idx->device = dev_env ? atoi(dev_env) : 0;
		idx->dev = nxsgpu_index_create(idx->device, &src);
		free(blk_off);
		free(doc_ids);
		free(pair_base);
		if (!idx->dev) {
			nxs_decl_err(idx->nxs, NXS_ERR_SYSTEM, "device index build failed: %s",
			    nxsgpu_last_error());
			goto out;
		}
		bad = nxsgpu_index_first_bad_doc(idx->dev);
		if (bad == UINT64_MAX) {
			idx->n_docs = n;
			idx->terms_seen = rd32(idx->tmap + 8);
			idx->dt_seen = rd64(idx->dmap + 8);
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
	if (idx->tmap) munmap(idx->tmap, idx->tmap_len);
	if (idx->dmap) munmap(idx->dmap, idx->dmap_len);
	idx->dev = NULL;
	idx->terms = NULL;
	idx->thash = NULL;
	idx->thash_cap = 0;
	idx->tmap = idx->dmap = NULL;
	idx->last_id = idx->term_count = 0;
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

/*
 * The reference re-syncs appended data before every search (search.c:309-312:
 * idx_terms_sync + idx_dtmap_sync).  The device index is an immutable
 * snapshot keyed by the two published data_len fields; when another process
 * has appended (or removed: that appends a tombstone) since, the snapshot is
 * rebuilt from the files.  Returns 0 (fresh or rebuilt) or -1.
 */
int
nxs_index_refresh(nxs_index_t *idx)
{
	uint64_t t_now, d_now;

	if (!idx->tmap || !idx->dmap) {
		return -1;
	}
	/* data_len is published last, with release semantics (terms.c:303-305,
	 * dtmap.c:333-337): an acquire load pairs with it */
	t_now = be32toh(__atomic_load_n((const uint32_t *)(idx->tmap + 8), __ATOMIC_ACQUIRE));
	d_now = be64toh(__atomic_load_n((const uint64_t *)(idx->dmap + 8), __ATOMIC_ACQUIRE));
	if (t_now == idx->terms_seen && d_now == idx->dt_seen) {
		return 0;
	}
	/*
	 * Build the new snapshot FIRST and swap on success: if the files cannot be
	 * consumed right now (a writer in the middle of an append) the old
	 * snapshot keeps serving, as the reference's partial sync does
	 * (DTMAP_PARTIAL_SYNC, dtmap.c:527-535); the next search tries again.
	 */
	{
		nxs_index_t tmp;
		nxs_err_t saved_code = idx->nxs->errcode;

		memset(&tmp, 0, sizeof(tmp));
		tmp.nxs = idx->nxs;
		tmp.algo = idx->algo;
		tmp.lowercase = idx->lowercase;
		tmp.terms_path = idx->terms_path;
		tmp.dtmap_path = idx->dtmap_path;
		if (nxs_index_load(&tmp, idx->terms_path, idx->dtmap_path) != 0) {
			unload_snapshot(&tmp);
			if (saved_code == NXS_ERR_SUCCESS) {
				nxs_clear_error(idx->nxs);
			}
			return 0;
		}
		if (idx->comm && nxsgpu_index_set_comm(idx->dev, NULL) != 0) {
			unload_snapshot(&tmp);
			return 0;	/* batches in flight: not now */
		}
		unload_snapshot(idx);
		idx->tmap = tmp.tmap;	idx->tmap_len = tmp.tmap_len;
		idx->dmap = tmp.dmap;	idx->dmap_len = tmp.dmap_len;
		idx->terms = tmp.terms;
		idx->last_id = tmp.last_id;
		idx->term_count = tmp.term_count;
		idx->thash = tmp.thash;
		idx->thash_cap = tmp.thash_cap;
		idx->n_docs = tmp.n_docs;
		idx->dev = tmp.dev;
		idx->device = tmp.device;
		idx->terms_seen = tmp.terms_seen;
		idx->dt_seen = tmp.dt_seen;
		if (idx->comm) {
			(void)nxsgpu_index_set_comm(idx->dev, idx->comm);
		}
		return 0;
	}
}

/* exact Levenshtein of the host side (BK build), exported for the tests */
int
nxs_levdist_export(const uint8_t *a, size_t n, const uint8_t *b, size_t m)
{
	return nxs_levdist_host(a, n, b, m);
}

/*
 * nxs_synth.c -- deterministic synthetic corpus writer (bench/test tooling).
 *
 * Emits VALID nxsterms / nxsdtmap files (reference src/index/storage.h:13-134;
 * interop rules: SURVEY.md appendix B) so that the real loader is exercised:
 *   terms  T unique strings over [a-z], length uniform 4..12, in term-id order
 *   docs   ids 1..D (or sparse increasing u64 ids); per doc 1+Poisson(mean)
 *          distinct terms drawn without replacement from Zipf(s=1) over the
 *          term rank (= term id), tf = 1+Geometric(1/2), doc_len = sum tf
 * Every doc has its own RNG stream (seeded by the doc number), so the output
 * does not depend on the number of writer threads.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <math.h>
#include <endian.h>
#include <fcntl.h>
#include <unistd.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>

#define	IDX_SIZE_STEP	(32UL * 1024)
#define	MAX_DISTINCT	1024

static inline uint64_t
splitmix64(uint64_t *s)
{
	uint64_t z = (*s += 0x9e3779b97f4a7c15ULL);
	z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
	z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
	return z ^ (z >> 31);
}

static inline double
u01(uint64_t *s)
{
	return (double)(splitmix64(s) >> 11) * (1.0 / 9007199254740992.0);
}

/*
 * Term strings: fills bytes (13 bytes reserved per term) and offs[n+1].
 * Returns the total byte length.
 */
uint64_t
nxs_synth_terms(uint32_t n_terms, uint64_t seed, uint8_t *bytes, uint32_t *offs)
{
	uint64_t st = seed * 0x2545f4914f6cdd1dULL + 1, cap = 64, o = 0;
	uint64_t *set;

	while (cap < (uint64_t)n_terms * 2 + 2) {
		cap <<= 1;
	}
	set = calloc(cap, sizeof(uint64_t));	/* packed strings: 5 bits/char + len */
	for (uint32_t t = 0; t < n_terms; t++) {
		for (;;) {
			const unsigned len = 4 + (unsigned)(splitmix64(&st) % 9);
			uint64_t key = len, h;
			uint8_t buf[12];
			size_t i;

			for (unsigned j = 0; j < len; j++) {
				buf[j] = 'a' + (uint8_t)(splitmix64(&st) % 26);
				key = (key << 5) | (uint64_t)(buf[j] - 'a' + 1);
			}
			h = key * 0x9e3779b97f4a7c15ULL;
			i = (h ^ (h >> 29)) & (cap - 1);
			while (set[i] && set[i] != key) {
				i = (i + 1) & (cap - 1);
			}
			if (set[i] == key) {
				continue;	/* duplicate: draw again */
			}
			set[i] = key;
			offs[t] = (uint32_t)o;
			memcpy(bytes + o, buf, len);
			o += len;
			break;
		}
	}
	offs[n_terms] = (uint32_t)o;
	free(set);
	return o;
}

typedef struct {
	uint64_t	n_docs;
	uint32_t	n_terms;
	uint64_t	seed;
	double		mean;
	int		sparse_ids;
	uint8_t *	dt_img;		/* mapped nxsdtmap */
	uint64_t *	blk_off;	/* [n_docs+1] */
	/* per thread */
	uint64_t	d0, d1;
	uint64_t *	totals;
	uint64_t	token_count;
	int		pass;
} job_t;

static inline uint64_t
doc_stream(uint64_t seed, uint64_t d)
{
	return seed ^ (d * 0xd1342543de82ef95ULL + 0x632be59bd9b4e019ULL);
}

static inline uint32_t
doc_ndistinct(const job_t *j, uint64_t *st)
{
	/* 1 + Poisson(mean), Knuth's product method */
	const double L = exp(-j->mean);
	double p = 1.0;
	uint32_t k = 0;

	do {
		k++;
		p *= u01(st);
	} while (p > L && k < 4 * MAX_DISTINCT);
	/* k - 1 ~ Poisson; + 1 */
	if (k > MAX_DISTINCT) k = MAX_DISTINCT;
	if (k > j->n_terms) k = j->n_terms;
	return k ? k : 1;
}

static uint64_t
doc_id_of(const job_t *j, uint64_t d)
{
	if (!j->sparse_ids) {
		return d;
	}
	{
		uint64_t s = j->seed ^ (d * 0x9e3779b97f4a7c15ULL);
		return d * 1000003ULL + splitmix64(&s) % 1000003ULL;
	}
}

static void *
worker(void *arg)
{
	job_t *j = arg;
	const double lnT = log((double)j->n_terms + 1.0);

	for (uint64_t d = j->d0; d < j->d1; d++) {
		uint64_t st = doc_stream(j->seed, d + 1);
		const uint32_t n = doc_ndistinct(j, &st);

		if (j->pass == 1) {
			j->blk_off[d + 1] = 16 + 8 * (uint64_t)n;	/* sizes; prefix later */
			continue;
		}
		{
			uint32_t ids[MAX_DISTINCT], tfs[MAX_DISTINCT], got = 0, doc_len = 0;
			uint32_t set[4 * MAX_DISTINCT];
			uint32_t cap = 16;
			uint8_t *p = j->dt_img + j->blk_off[d];
			uint64_t v64;
			uint32_t v32;

			while (cap < n * 2 + 2) {
				cap <<= 1;
			}
			memset(set, 0, cap * sizeof(uint32_t));
			while (got < n) {
				/* Zipf(s=1): P(r) ~ ln((r+1)/r) ~ 1/r */
				uint32_t r = (uint32_t)exp(u01(&st) * lnT);
				uint32_t i;
				if (r < 1) r = 1;
				if (r > j->n_terms) r = j->n_terms;
				i = (r * 2654435761u) & (cap - 1);
				while (set[i] && set[i] != r) {
					i = (i + 1) & (cap - 1);
				}
				if (set[i] == r) {
					continue;
				}
				set[i] = r;
				ids[got] = r;
				tfs[got] = 1 + (uint32_t)__builtin_ctzll(splitmix64(&st) | (1ULL << 40));
				doc_len += tfs[got];
				got++;
			}
			/* pairs sorted by term id (dtmap.c:239-241): insertion sort */
			for (uint32_t a = 1; a < n; a++) {
				const uint32_t ki = ids[a], kt = tfs[a];
				uint32_t b = a;
				while (b > 0 && ids[b - 1] > ki) {
					ids[b] = ids[b - 1];
					tfs[b] = tfs[b - 1];
					b--;
				}
				ids[b] = ki;
				tfs[b] = kt;
			}
			v64 = htobe64(doc_id_of(j, d + 1)); memcpy(p, &v64, 8);
			v32 = htobe32(doc_len); memcpy(p + 8, &v32, 4);
			v32 = htobe32(n); memcpy(p + 12, &v32, 4);
			for (uint32_t a = 0; a < n; a++) {
				v32 = htobe32(ids[a]); memcpy(p + 16 + 8 * a, &v32, 4);
				v32 = htobe32(tfs[a]); memcpy(p + 20 + 8 * a, &v32, 4);
				j->totals[ids[a]] += tfs[a];
			}
			j->token_count += doc_len;
		}
	}
	return NULL;
}

static uint8_t *
create_map(const char *path, uint64_t len)
{
	int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
	void *p;

	if (fd == -1) {
		return NULL;
	}
	if (ftruncate(fd, (off_t)len) == -1) {
		close(fd);
		return NULL;
	}
	p = mmap(NULL, len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
	close(fd);
	return p == MAP_FAILED ? NULL : p;
}

static void
run_pass(job_t *base, int threads, int pass, uint64_t *totals_all, uint64_t *token_count)
{
	pthread_t *th = calloc(threads, sizeof(pthread_t));
	job_t *jobs = calloc(threads, sizeof(job_t));
	const uint64_t per = (base->n_docs + threads - 1) / threads;

	for (int t = 0; t < threads; t++) {
		jobs[t] = *base;
		jobs[t].pass = pass;
		jobs[t].d0 = (uint64_t)t * per < base->n_docs ? (uint64_t)t * per : base->n_docs;
		jobs[t].d1 = jobs[t].d0 + per < base->n_docs ? jobs[t].d0 + per : base->n_docs;
		jobs[t].totals = (pass == 2) ? calloc((size_t)base->n_terms + 1, 8) : NULL;
		pthread_create(&th[t], NULL, worker, &jobs[t]);
	}
	for (int t = 0; t < threads; t++) {
		pthread_join(th[t], NULL);
		if (pass == 2) {
			for (uint32_t i = 1; i <= base->n_terms; i++) {
				totals_all[i] += jobs[t].totals[i];
			}
			*token_count += jobs[t].token_count;
			free(jobs[t].totals);
		}
	}
	free(th);
	free(jobs);
}

/*
 * Writes both files.  Returns 0 and the number of postings / tokens written.
 */
int
nxs_synth_write(const char *terms_path, const char *dtmap_path, uint64_t n_docs,
    uint32_t n_terms, uint64_t seed, double mean_distinct, int sparse_ids,
    int threads, uint64_t *out_postings, uint64_t *out_tokens)
{
	job_t base;
	uint8_t *tbytes, *timg, *dimg;
	uint32_t *toffs;
	uint64_t *totals, tdata = 0, tfile, ddata, dfile, token_count = 0, postings = 0;
	uint64_t v64;
	uint32_t v32;

	if (threads < 1) threads = 1;
	if (n_terms == 0 || n_docs >= (1ULL << 32)) {
		return -1;
	}
	memset(&base, 0, sizeof(base));
	base.n_docs = n_docs;
	base.n_terms = n_terms;
	base.seed = seed + 2;
	base.mean = mean_distinct;
	base.sparse_ids = sparse_ids;
	base.blk_off = calloc(n_docs + 2, sizeof(uint64_t));

	/* pass 1: block sizes -> offsets */
	run_pass(&base, threads, 1, NULL, NULL);
	base.blk_off[0] = 32;
	for (uint64_t d = 0; d < n_docs; d++) {
		postings += (base.blk_off[d + 1] - 16) / 8;
		base.blk_off[d + 1] += base.blk_off[d];
	}
	ddata = base.blk_off[n_docs] - 32;
	dfile = (32 + ddata + IDX_SIZE_STEP - 1) / IDX_SIZE_STEP * IDX_SIZE_STEP;
	if (dfile == 0) dfile = IDX_SIZE_STEP;
	if ((dimg = create_map(dtmap_path, dfile)) == NULL) {
		free(base.blk_off);
		return -1;
	}
	base.dt_img = dimg;

	/* pass 2: blocks + per-term totals */
	totals = calloc((size_t)n_terms + 1, sizeof(uint64_t));
	run_pass(&base, threads, 2, totals, &token_count);

	memcpy(dimg, "NXS_D", 5);
	dimg[5] = 1;
	v64 = htobe64(ddata); memcpy(dimg + 8, &v64, 8);
	v64 = htobe64(token_count); memcpy(dimg + 16, &v64, 8);
	v32 = htobe32((uint32_t)n_docs); memcpy(dimg + 24, &v32, 4);
	munmap(dimg, dfile);

	/* terms file */
	tbytes = malloc((size_t)n_terms * 13 + 16);
	toffs = malloc(((size_t)n_terms + 1) * sizeof(uint32_t));
	nxs_synth_terms(n_terms, seed + 1, tbytes, toffs);
	for (uint32_t t = 0; t < n_terms; t++) {
		const uint32_t len = toffs[t + 1] - toffs[t];
		tdata += ((2 + len + 1 + 7) & ~7ULL) + 8;
	}
	tfile = (16 + tdata + IDX_SIZE_STEP - 1) / IDX_SIZE_STEP * IDX_SIZE_STEP;
	if ((timg = create_map(terms_path, tfile)) == NULL) {
		free(base.blk_off); free(totals); free(tbytes); free(toffs);
		return -1;
	}
	memcpy(timg, "NXS_T", 5);
	timg[5] = 1;
	v32 = htobe32((uint32_t)tdata); memcpy(timg + 8, &v32, 4);
	{
		uint64_t o = 16;
		for (uint32_t t = 0; t < n_terms; t++) {
			const uint32_t len = toffs[t + 1] - toffs[t];
			const uint16_t l16 = htobe16((uint16_t)len);
			const uint64_t blk = ((2 + len + 1 + 7) & ~7ULL) + 8;
			memcpy(timg + o, &l16, 2);
			memcpy(timg + o + 2, tbytes + toffs[t], len);
			v64 = htobe64(totals[t + 1]);
			memcpy(timg + o + blk - 8, &v64, 8);
			o += blk;
		}
	}
	munmap(timg, tfile);

	if (out_postings) *out_postings = postings;
	if (out_tokens) *out_tokens = token_count;
	free(base.blk_off);
	free(totals);
	free(tbytes);
	free(toffs);
	return 0;
}

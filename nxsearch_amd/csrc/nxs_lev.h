/*
 * nxs_lev.h -- Levenshtein distance, shared by the C11 host code and the
 * HIP kernels (compiled by both gcc and hipcc).
 *
 * The reference computes the byte-wise Wagner-Fischer DP with one uint16_t
 * row (src/algo/levdist.c:67-150).  The distance is a pure function of the
 * two byte strings, so any exact algorithm returns identical results; here:
 * Myers' bit-vector algorithm in Hyyro's global-distance form (pattern up
 * to 64 bytes held in one 64-bit word, O(text length) word operations), and
 * the plain row DP for patterns longer than that.
 */
#ifndef NXS_LEV_H
#define NXS_LEV_H

#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#define	NXS_HD	__host__ __device__ static inline
#else
#define	NXS_HD	static inline
#endif

#define	NXS_MYERS_MAXPAT	64

typedef struct {
	uint64_t	pv, mv, top;
	int		score;
} nxs_myers_t;

/* pattern length m in 1..64 */
NXS_HD void
nxs_myers_init(nxs_myers_t *s, unsigned m)
{
	s->pv = ~UINT64_C(0);
	s->mv = 0;
	s->top = UINT64_C(1) << (m - 1);
	s->score = (int)m;
}

/* eq: bit j set iff pattern[j] == the text byte being consumed */
NXS_HD void
nxs_myers_step(nxs_myers_t *s, uint64_t eq)
{
	const uint64_t pv = s->pv, mv = s->mv;
	const uint64_t xv = eq | mv;
	const uint64_t xh = (((eq & pv) + pv) ^ pv) | eq;
	uint64_t ph = mv | ~(xh | pv);
	uint64_t mh = pv & xh;

	s->score += (int)((ph & s->top) != 0) - (int)((mh & s->top) != 0);
	ph = (ph << 1) | 1;	/* global distance: D[0][j] = j */
	mh <<= 1;
	s->pv = mh | ~(xv | ph);
	s->mv = ph & xv;
}

#if !defined(__HIPCC__) || !defined(__HIP_DEVICE_COMPILE__)
#include <stdlib.h>
#include <string.h>

/* per-pattern match table: peq[c] bit j <=> pat[j] == c */
static inline void
nxs_myers_peq(const uint8_t *pat, unsigned m, uint64_t peq[256])
{
	memset(peq, 0, 256 * sizeof(uint64_t));
	for (unsigned j = 0; j < m; j++) {
		peq[pat[j]] |= UINT64_C(1) << j;
	}
}

/* host-side exact distance for arbitrary lengths */
static inline int
nxs_levdist_host(const uint8_t *a, size_t n, const uint8_t *b, size_t m)
{
	if (n < m) {
		const uint8_t *t = a; a = b; b = t;
		size_t tl = n; n = m; m = tl;
	}
	/* b (length m) is the shorter string */
	if (m == 0) {
		return (int)n;
	}
	if (m <= NXS_MYERS_MAXPAT) {
		uint64_t peq[256];
		nxs_myers_t s;

		nxs_myers_peq(b, (unsigned)m, peq);
		nxs_myers_init(&s, (unsigned)m);
		for (size_t i = 0; i < n; i++) {
			nxs_myers_step(&s, peq[a[i]]);
		}
		return s.score;
	} else {
		uint32_t *row = (uint32_t *)malloc((m + 1) * sizeof(uint32_t));
		int d;

		for (size_t j = 0; j <= m; j++) {
			row[j] = (uint32_t)j;
		}
		for (size_t i = 0; i < n; i++) {
			uint32_t diag = (uint32_t)i, above;
			row[0] = (uint32_t)i + 1;
			for (size_t j = 1; j <= m; j++) {
				uint32_t v;
				above = row[j];
				v = diag + (a[i] != b[j - 1]);
				if (row[j - 1] + 1 < v) v = row[j - 1] + 1;
				if (above + 1 < v) v = above + 1;
				row[j] = v;
				diag = above;
			}
		}
		d = (int)row[m];
		free(row);
		return d;
	}
}
#endif

#endif

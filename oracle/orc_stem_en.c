/*
 * orc_stem_en.c -- TEST INFRASTRUCTURE (oracle): the English Snowball stemmer.
 *
 * The reference's `stemmer' filter calls libstemmer (src/core/filters_builtin.c:
 * 203-245: sb_stemmer_new(lang, NULL), sb_stemmer_stem); libstemmer is a
 * third-party dependency that is absent from /root/reference and from this
 * image.  This file restates the PUBLISHED algorithm
 * (snowballstem.org, "The English (Porter2) stemming algorithm", as shipped by
 * libstemmer 2.x) as a small cursor machine that executes the Snowball program
 * command by command -- a different construction from the product's
 * nxsearch_amd/csrc/nxs_stem_en.c, so that the two check each other.
 *
 * Parity unpinned beyond the words the reference's tests hold (t_scoring.c:16-163,
 * test.lua): no libstemmer output could be generated here.
 */
#include <stdbool.h>
#include <stdlib.h>
#include <string.h>

#include "nxs_oracle.h"

typedef struct {
	unsigned char *p;
	int	c, l, lb;	/* cursor, limit, backward limit */
	int	bra, ket;	/* slice */
	int	p1, p2;
	bool	Y_found;
} env_t;

/* ---- the machine's commands ---- */

static bool
in_set(const char *set, unsigned ch)
{
	return ch < 0x80 && ch != 0 && strchr(set, (int)ch) != NULL;
}

/* one character forward / backward (UTF-8); false at the limit */
static bool
next_f(env_t *z)
{
	if (z->c >= z->l) {
		return false;
	}
	z->c++;
	while (z->c < z->l && (z->p[z->c] & 0xc0) == 0x80) {
		z->c++;
	}
	return true;
}

static bool
next_b(env_t *z)
{
	if (z->c <= z->lb) {
		return false;
	}
	z->c--;
	while (z->c > z->lb && (z->p[z->c] & 0xc0) == 0x80) {
		z->c--;
	}
	return true;
}

/* grouping tests: consume one character if it is (not) in the set */
static bool
grp_f(env_t *z, const char *set, bool want)
{
	if (z->c >= z->l || in_set(set, z->p[z->c]) != want) {
		return false;
	}
	return next_f(z);
}

static bool
grp_b(env_t *z, const char *set, bool want)
{
	const int save = z->c;

	if (!next_b(z)) {
		return false;
	}
	if (in_set(set, z->p[z->c]) != want) {	/* (lead byte of a multi-byte character: never in a set) */
		z->c = save;
		return false;
	}
	return true;
}

static bool
gopast_f(env_t *z, const char *set, bool want)
{
	while (!grp_f(z, set, want)) {
		if (!next_f(z)) {
			return false;
		}
	}
	return true;
}

static bool
gopast_b(env_t *z, const char *set, bool want)
{
	while (!grp_b(z, set, want)) {
		if (!next_b(z)) {
			return false;
		}
	}
	return true;
}

static bool
lit_f(env_t *z, const char *s)
{
	const int n = (int)strlen(s);

	if (z->l - z->c < n || memcmp(z->p + z->c, s, (size_t)n) != 0) {
		return false;
	}
	z->c += n;
	return true;
}

static bool
lit_b(env_t *z, const char *s)
{
	const int n = (int)strlen(s);

	if (z->c - z->lb < n || memcmp(z->p + z->c - n, s, (size_t)n) != 0) {
		return false;
	}
	z->c -= n;
	return true;
}

/* substring among, backward: longest entry that ends at the cursor; -1 = none */
static int
among_b(env_t *z, const char *const *tab, int n)
{
	int best = -1, bl = -1;

	for (int i = 0; i < n; i++) {
		const int len = (int)strlen(tab[i]);
		if (len > bl && z->c - z->lb >= len && memcmp(z->p + z->c - len, tab[i], (size_t)len) == 0) {
			best = i;
			bl = len;
		}
	}
	if (best >= 0) {
		z->c -= bl;
	}
	return best;
}

static int
among_f(env_t *z, const char *const *tab, int n)
{
	int best = -1, bl = -1;

	for (int i = 0; i < n; i++) {
		const int len = (int)strlen(tab[i]);
		if (len > bl && z->l - z->c >= len && memcmp(z->p + z->c, tab[i], (size_t)len) == 0) {
			best = i;
			bl = len;
		}
	}
	if (best >= 0) {
		z->c += bl;
	}
	return best;
}

/* <- : replace the slice [bra, ket) */
static void
slice_from(env_t *z, const char *s)
{
	const int n = (int)strlen(s), old = z->ket - z->bra, adj = n - old;

	memmove(z->p + z->ket + adj, z->p + z->ket, (size_t)(z->l - z->ket));
	memcpy(z->p + z->bra, s, (size_t)n);
	z->l += adj;
	if (z->c >= z->ket) {
		z->c += adj;
	} else if (z->c > z->bra) {
		z->c = z->bra;
	}
	z->ket = z->bra + n;
}

/* <+ : insert at the cursor (here: always the end of the word, after the slice was used) */
static void
insert(env_t *z, const char *s)
{
	z->bra = z->ket = z->c;
	slice_from(z, s);
}

#define	V	"aeiouy"
#define	V_WXY	"aeiouywxY"
#define	VALID_LI "cdeghkmnrt"

static bool R1(const env_t *z) { return z->p1 <= z->c; }
static bool R2(const env_t *z) { return z->p2 <= z->c; }

/* define shortv as ( ( non-v_WXY v non-v ) or ( non-v v atlimit ) )   [backward] */
static bool
shortv(env_t *z)
{
	const int save = z->c;

	if (grp_b(z, V_WXY, false) && grp_b(z, V, true) && grp_b(z, V, false)) {
		return true;
	}
	z->c = save;
	if (grp_b(z, V, false) && grp_b(z, V, true) && z->c <= z->lb) {
		return true;
	}
	z->c = save;
	return false;
}

static void
prelude(env_t *z)
{
	int save;

	z->Y_found = false;
	save = z->c;					/* do ( ['''] delete ) */
	z->bra = z->c;
	if (lit_f(z, "'")) {
		z->ket = z->c;
		slice_from(z, "");
	}
	z->c = save;
	z->bra = z->c;					/* do ( ['y'] <-'Y' set Y_found ) */
	if (lit_f(z, "y")) {
		z->ket = z->c;
		slice_from(z, "Y");
		z->Y_found = true;
	}
	z->c = save;
	for (;;) {					/* do repeat ( goto (v ['y']) <-'Y' set Y_found ) */
		const int rsave = z->c;
		bool found = false;

		for (;;) {				/* goto */
			const int gsave = z->c;
			if (grp_f(z, V, true)) {
				z->bra = z->c;
				if (lit_f(z, "y")) {
					z->ket = z->c;
					z->c = gsave;
					found = true;
					break;
				}
			}
			z->c = gsave;
			if (!next_f(z)) {
				break;
			}
		}
		if (!found) {
			z->c = rsave;
			break;
		}
		slice_from(z, "Y");
		z->Y_found = true;
	}
	z->c = save;
}

static void
mark_regions(env_t *z)
{
	static const char *const pre[] = { "gener", "commun", "arsen" };
	const int save = z->c;

	z->p1 = z->p2 = z->l;
	if (among_f(z, pre, 3) < 0) {
		z->c = save;
		if (!gopast_f(z, V, true) || !gopast_f(z, V, false)) {
			z->c = save;
			return;
		}
	}
	z->p1 = z->c;
	if (gopast_f(z, V, true) && gopast_f(z, V, false)) {
		z->p2 = z->c;
	}
	z->c = save;
}

static void
Step_1a(env_t *z)
{
	static const char *const a0[] = { "'", "'s", "'s'" };
	static const char *const a1[] = { "sses", "ied", "ies", "s", "us", "ss" };
	int save = z->c, k;

	z->ket = z->c;					/* try ( [substring] among ( ''' ''s' ''s'' (delete) ) ) */
	if (among_b(z, a0, 3) >= 0) {
		z->bra = z->c;
		slice_from(z, "");
	} else {
		z->c = save;
	}
	z->ket = z->c;
	if ((k = among_b(z, a1, 6)) < 0) {
		return;
	}
	z->bra = z->c;
	switch (k) {
	case 0:
		slice_from(z, "ss");
		break;
	case 1: case 2: {				/* (hop 2 <-'i') or <-'ie' */
		const int s2 = z->c;
		if (next_b(z) && next_b(z)) {
			slice_from(z, "i");
		} else {
			z->c = s2;
			slice_from(z, "ie");
		}
		break;
	}
	case 3:						/* next gopast v delete */
		if (next_b(z) && gopast_b(z, V, true)) {
			slice_from(z, "");
		}
		break;
	default:
		break;
	}
}

static bool
exception2(env_t *z)
{
	static const char *const a[] = {
		"inning", "outing", "canning", "herring", "earring", "proceed", "exceed", "succeed",
	};
	const int save = z->c;

	z->ket = z->c;
	if (among_b(z, a, 8) >= 0 && z->c <= z->lb) {
		return true;
	}
	z->c = save;
	return false;
}

static void
Step_1b(env_t *z)
{
	static const char *const a[] = { "eed", "eedly", "ed", "edly", "ing", "ingly" };
	static const char *const b[] = { "at", "bl", "iz", "bb", "dd", "ff", "gg", "mm", "nn", "pp", "rr", "tt" };
	int k;

	z->ket = z->c;
	if ((k = among_b(z, a, 6)) < 0) {
		return;
	}
	z->bra = z->c;
	if (k < 2) {
		if (R1(z)) {
			slice_from(z, "ee");
		}
		return;
	}
	{						/* test gopast v */
		const int t = z->c;
		if (!gopast_b(z, V, true)) {
			return;
		}
		z->c = t;
	}
	slice_from(z, "");
	{						/* test substring among (...) */
		const int t = z->c;
		k = among_b(z, b, 12);
		if (k >= 0 && k < 3) {
			z->c = t;
			insert(z, "e");
		} else if (k >= 3) {
			z->c = t;			/* (test restores the cursor; the command list runs at it) */
			z->ket = z->c;
			(void)next_b(z);
			z->bra = z->c;
			slice_from(z, "");
		} else {				/* '' : atmark p1  test shortv  <+ 'e' */
			z->c = t;
			if (z->c == z->p1) {
				const bool sv = shortv(z);
				z->c = t;
				if (sv) {
					insert(z, "e");
				}
			}
		}
	}
}

static void
Step_1c(env_t *z)
{
	z->ket = z->c;
	if (!lit_b(z, "y") && !lit_b(z, "Y")) {
		return;
	}
	z->bra = z->c;
	if (!grp_b(z, V, false) || z->c <= z->lb) {
		return;
	}
	slice_from(z, "i");
}

static void
Step_2(env_t *z)
{
	static const char *const a[] = {
		"tional", "enci", "anci", "abli", "entli", "izer", "ization", "ational", "ation", "ator",
		"alism", "aliti", "alli", "fulness", "ousli", "ousness", "iveness", "iviti", "biliti", "bli",
		"ogi", "fulli", "lessli", "li",
	};
	static const char *const to[] = {
		"tion", "ence", "ance", "able", "ent", "ize", "ize", "ate", "ate", "ate",
		"al", "al", "al", "ful", "ous", "ous", "ive", "ive", "ble", "ble",
		"og", "ful", "less", "",
	};
	int k;

	z->ket = z->c;
	if ((k = among_b(z, a, 24)) < 0) {
		return;
	}
	z->bra = z->c;
	if (!R1(z)) {
		return;
	}
	if (k == 20 && !lit_b(z, "l")) {
		return;
	}
	if (k == 23 && !grp_b(z, VALID_LI, true)) {
		return;
	}
	slice_from(z, to[k]);
}

static void
Step_3(env_t *z)
{
	static const char *const a[] = { "tional", "ational", "alize", "icate", "iciti", "ical", "ful", "ness", "ative" };
	static const char *const to[] = { "tion", "ate", "al", "ic", "ic", "ic", "", "", "" };
	int k;

	z->ket = z->c;
	if ((k = among_b(z, a, 9)) < 0) {
		return;
	}
	z->bra = z->c;
	if (!R1(z) || (k == 8 && !R2(z))) {
		return;
	}
	slice_from(z, to[k]);
}

static void
Step_4(env_t *z)
{
	static const char *const a[] = {
		"al", "ance", "ence", "er", "ic", "able", "ible", "ant", "ement", "ment", "ent", "ism", "ate",
		"iti", "ous", "ive", "ize", "ion",
	};
	int k;

	z->ket = z->c;
	if ((k = among_b(z, a, 18)) < 0) {
		return;
	}
	z->bra = z->c;
	if (!R2(z)) {
		return;
	}
	if (k == 17 && !lit_b(z, "s") && !lit_b(z, "t")) {
		return;
	}
	slice_from(z, "");
}

static void
Step_5(env_t *z)
{
	static const char *const a[] = { "e", "l" };
	int k;

	z->ket = z->c;
	if ((k = among_b(z, a, 2)) < 0) {
		return;
	}
	z->bra = z->c;
	if (k == 0) {					/* R2 or (R1 not shortv) delete */
		bool del = R2(z);
		if (!del && R1(z)) {
			const int t = z->c;
			del = !shortv(z);
			z->c = t;
		}
		if (del) {
			slice_from(z, "");
		}
	} else if (R2(z) && lit_b(z, "l")) {
		slice_from(z, "");
	}
}

static bool
exception1(env_t *z)
{
	static const char *const a[] = {
		"skis", "skies", "dying", "lying", "tying", "idly", "gently", "ugly", "early", "only", "singly",
		"sky", "news", "howe", "atlas", "cosmos", "bias", "andes",
	};
	static const char *const to[] = {
		"ski", "sky", "die", "lie", "tie", "idl", "gentl", "ugli", "earli", "onli", "singl",
	};
	const int save = z->c;
	int k;

	z->bra = z->c;
	k = among_f(z, a, 18);
	if (k < 0 || z->c < z->l) {
		z->c = save;
		return false;
	}
	z->ket = z->c;
	if (k < 11) {
		slice_from(z, to[k]);
	}
	return true;
}

/* define stem as ( exception1 or not hop 3 or ( ... ) ); out must hold len + 2 bytes */
size_t
orc_stem_en(const char *in, size_t len, char *out, size_t cap)
{
	env_t z = { 0 };

	if (cap < len + 2) {
		return 0;
	}
	memcpy(out, in, len);
	z.p = (unsigned char *)out;
	z.l = (int)len;
	if (!exception1(&z)) {
		const int save = z.c;
		const bool hop3 = next_f(&z) && next_f(&z) && next_f(&z);
		z.c = save;
		if (hop3) {
			prelude(&z);
			mark_regions(&z);
			z.lb = z.c;			/* backwards ( */
			z.c = z.l;
			Step_1a(&z);
			z.c = z.l;
			if (!exception2(&z)) {
				z.c = z.l; Step_1b(&z);
				z.c = z.l; Step_1c(&z);
				z.c = z.l; Step_2(&z);
				z.c = z.l; Step_3(&z);
				z.c = z.l; Step_4(&z);
				z.c = z.l; Step_5(&z);
			}
			z.c = z.lb;			/* ) */
			if (z.Y_found) {		/* postlude */
				for (int i = 0; i < z.l; i++) {
					if (z.p[i] == 'Y') {
						z.p[i] = 'y';
					}
				}
			}
		}
	}
	out[z.l] = '\0';
	return (size_t)z.l;
}

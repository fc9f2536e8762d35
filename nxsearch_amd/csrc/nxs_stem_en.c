/*
 * nxs_stem_en.c -- the English Snowball stemmer ("Porter2") for QUERY tokens.
 *
 * The reference's `stemmer' filter (src/core/filters_builtin.c:203-245) hands
 * every token to libstemmer: sb_stemmer_new(lang, NULL) -- UTF-8 -- then
 * sb_stemmer_stem().  `lang' is the index's "lang" parameter, "en" by default
 * (src/core/nxs.c:271-276), and the filter is part of the DEFAULT filter list
 * (nxs.c:87-89,263-266): an index created without naming filters stems.
 *
 * libstemmer is not in this image, so this is a hand-written implementation
 * of the published algorithm (snowballstem.org, "The English (Porter2)
 * stemming algorithm", the form libstemmer 2.x ships), on UTF-8 bytes: vowels
 * and all suffixes are ASCII, every other code point is a non-vowel; where the
 * algorithm counts or steps over CHARACTERS (hop 3, hop 2, next, the short-
 * syllable test) whole code points are stepped over, as libstemmer's UTF-8
 * mode does.  Other languages stay refused (nxs_filters.c).
 *
 * In place: the result is never longer than the input.
 */
#include <string.h>
#include <stdbool.h>
#include <stddef.h>

#include "nxs_impl.h"

typedef struct {
	char *	w;
	int	n;	/* current length */
	int	p1, p2;	/* start of R1 / R2 (marks: not moved by later deletions) */
} stem_t;

static bool
vowel(char c)
{
	return c == 'a' || c == 'e' || c == 'i' || c == 'o' || c == 'u' || c == 'y';
}

/* start of the code point that ends at byte index `e' (exclusive); e > 0 */
static int
cp_back(const stem_t *z, int e)
{
	int j = e - 1;

	while (j > 0 && ((unsigned char)z->w[j] & 0xc0) == 0x80) {
		j--;
	}
	return j;
}

/* end of the code point that starts at byte index `b'; b < n */
static int
cp_fwd(const stem_t *z, int b)
{
	int j = b + 1;

	while (j < z->n && ((unsigned char)z->w[j] & 0xc0) == 0x80) {
		j++;
	}
	return j;
}

static bool
ends(const stem_t *z, const char *s)
{
	const int l = (int)strlen(s);

	return l <= z->n && memcmp(z->w + z->n - l, s, (size_t)l) == 0;
}

/* replace the last `l' bytes by `s' (never longer than what it replaces, but for +1 'e') */
static void
set_end(stem_t *z, int l, const char *s)
{
	const int sl = (int)strlen(s);

	memcpy(z->w + z->n - l, s, (size_t)sl);
	z->n += sl - l;
}

/* a vowel somewhere in w[0, e) */
static bool
has_vowel(const stem_t *z, int e)
{
	for (int i = 0; i < e; i++) {
		if (vowel(z->w[i])) {
			return true;
		}
	}
	return false;
}

/*
 * The short-syllable test, looking left from byte index `e':
 *   non-vowel other than w, x, Y  <-  vowel  <-  non-vowel,   or
 *   non-vowel  <-  vowel at the very beginning of the word.
 */
static bool
short_syllable(const stem_t *z, int e)
{
	int c, v;

	if (e <= 0) {
		return false;
	}
	c = cp_back(z, e);			/* the last character */
	if (vowel(z->w[c]) || c == 0) {
		return false;
	}
	v = cp_back(z, c);
	if (!vowel(z->w[v])) {
		return false;
	}
	if (z->w[c] != 'w' && z->w[c] != 'x' && z->w[c] != 'Y' && v > 0) {
		const int b = cp_back(z, v);
		if (!vowel(z->w[b])) {
			return true;
		}
	}
	return v == 0;
}

static const struct { const char *from, *to; } special[] = {
	{ "skis", "ski" }, { "skies", "sky" }, { "dying", "die" }, { "lying", "lie" },
	{ "tying", "tie" }, { "idly", "idl" }, { "gently", "gentl" }, { "ugly", "ugli" },
	{ "early", "earli" }, { "only", "onli" }, { "singly", "singl" },
	/* invariant */
	{ "sky", "sky" }, { "news", "news" }, { "howe", "howe" }, { "atlas", "atlas" },
	{ "cosmos", "cosmos" }, { "bias", "bias" }, { "andes", "andes" },
};

static const char *const after_1a[] = {
	"inning", "outing", "canning", "herring", "earring", "proceed", "exceed", "succeed",
};

static bool
whole_word(const stem_t *z, const char *s)
{
	return (int)strlen(s) == z->n && memcmp(z->w, s, (size_t)z->n) == 0;
}

static void
regions(stem_t *z)
{
	static const char *const pre[] = { "gener", "commun", "arsen" };
	int i = -1;

	z->p1 = z->p2 = z->n;
	for (unsigned k = 0; k < 3; k++) {
		const int l = (int)strlen(pre[k]);
		if (z->n >= l && memcmp(z->w, pre[k], (size_t)l) == 0) {
			i = l;
			break;
		}
	}
	if (i < 0) {
		/* past the first vowel, then past the first non-vowel after it */
		i = 0;
		while (i < z->n && !vowel(z->w[i])) {
			i++;
		}
		if (i >= z->n) {
			return;
		}
		while (i < z->n && vowel(z->w[i])) {
			i++;
		}
		if (i >= z->n) {
			return;
		}
		i = cp_fwd(z, i);
	}
	z->p1 = i;
	while (i < z->n && !vowel(z->w[i])) {
		i++;
	}
	if (i >= z->n) {
		return;
	}
	while (i < z->n && vowel(z->w[i])) {
		i++;
	}
	if (i >= z->n) {
		return;
	}
	z->p2 = cp_fwd(z, i);
}

static void
step_1a(stem_t *z)
{
	if (ends(z, "'s'")) {
		z->n -= 3;
	} else if (ends(z, "'s")) {
		z->n -= 2;
	} else if (ends(z, "'")) {
		z->n -= 1;
	}
	if (ends(z, "sses")) {
		z->n -= 2;
	} else if (ends(z, "ied") || ends(z, "ies")) {
		/* more than one character in front of the suffix: i, else ie */
		const int s = z->n - 3;
		if (s > 0 && cp_back(z, s) > 0) {
			z->n -= 2;
		} else {
			z->n -= 1;
			z->w[z->n - 1] = 'e';
		}
	} else if (ends(z, "us") || ends(z, "ss")) {
		/* nothing */
	} else if (ends(z, "s")) {
		/* delete if the part before contains a vowel that is not the letter right before the s */
		const int s = z->n - 1;
		if (s > 0 && has_vowel(z, cp_back(z, s))) {
			z->n -= 1;
		}
	}
}

static void
step_1b(stem_t *z)
{
	int l;

	if (ends(z, "eedly")) {
		l = 5;
	} else if (ends(z, "eed")) {
		l = 3;
	} else {
		l = 0;
	}
	if (l) {
		if (z->n - l >= z->p1) {
			set_end(z, l, "ee");
		}
		return;
	}
	if (ends(z, "ingly")) {
		l = 5;
	} else if (ends(z, "edly")) {
		l = 4;
	} else if (ends(z, "ing")) {
		l = 3;
	} else if (ends(z, "ed")) {
		l = 2;
	} else {
		return;
	}
	if (!has_vowel(z, z->n - l)) {
		return;
	}
	z->n -= l;
	if (ends(z, "at") || ends(z, "bl") || ends(z, "iz")) {
		z->w[z->n++] = 'e';
		return;
	}
	if (z->n >= 2 && z->w[z->n - 1] == z->w[z->n - 2] && strchr("bdfgmnprt", z->w[z->n - 1])) {
		z->n--;
		return;
	}
	if (z->n == z->p1 && short_syllable(z, z->n)) {
		z->w[z->n++] = 'e';
	}
}

static void
step_1c(stem_t *z)
{
	if (z->n >= 2 && (z->w[z->n - 1] == 'y' || z->w[z->n - 1] == 'Y')) {
		const int c = cp_back(z, z->n - 1);
		if (!vowel(z->w[c]) && c > 0) {
			z->w[z->n - 1] = 'i';
		}
	}
}

typedef struct { const char *suf, *rep; int cond; } rule_t;

/* the longest suffix of the table that the word ends with (NULL: none) */
static const rule_t *
longest(const stem_t *z, const rule_t *tab, size_t n)
{
	const rule_t *best = NULL;
	size_t bl = 0;

	for (size_t i = 0; i < n; i++) {
		const size_t l = strlen(tab[i].suf);
		if (l > bl && ends(z, tab[i].suf)) {
			best = &tab[i];
			bl = l;
		}
	}
	return best;
}

enum { C_NONE, C_OGI, C_LI, C_R2, C_ION };

static void
step_2(stem_t *z)
{
	static const rule_t tab[] = {
		{ "tional", "tion", 0 }, { "enci", "ence", 0 }, { "anci", "ance", 0 }, { "abli", "able", 0 },
		{ "entli", "ent", 0 }, { "izer", "ize", 0 }, { "ization", "ize", 0 }, { "ational", "ate", 0 },
		{ "ation", "ate", 0 }, { "ator", "ate", 0 }, { "alism", "al", 0 }, { "aliti", "al", 0 },
		{ "alli", "al", 0 }, { "fulness", "ful", 0 }, { "ousli", "ous", 0 }, { "ousness", "ous", 0 },
		{ "iveness", "ive", 0 }, { "iviti", "ive", 0 }, { "biliti", "ble", 0 }, { "bli", "ble", 0 },
		{ "ogi", "og", C_OGI }, { "fulli", "ful", 0 }, { "lessli", "less", 0 }, { "li", "", C_LI },
	};
	const rule_t *r = longest(z, tab, sizeof(tab) / sizeof(tab[0]));
	int l, s;

	if (!r) {
		return;
	}
	l = (int)strlen(r->suf);
	s = z->n - l;
	if (s < z->p1) {
		return;
	}
	if (r->cond == C_OGI && !(s > 0 && z->w[s - 1] == 'l')) {
		return;
	}
	if (r->cond == C_LI && !(s > 0 && strchr("cdeghkmnrt", z->w[s - 1]))) {
		return;
	}
	set_end(z, l, r->rep);
}

static void
step_3(stem_t *z)
{
	static const rule_t tab[] = {
		{ "tional", "tion", 0 }, { "ational", "ate", 0 }, { "alize", "al", 0 }, { "icate", "ic", 0 },
		{ "iciti", "ic", 0 }, { "ical", "ic", 0 }, { "ful", "", 0 }, { "ness", "", 0 },
		{ "ative", "", C_R2 },
	};
	const rule_t *r = longest(z, tab, sizeof(tab) / sizeof(tab[0]));
	int l, s;

	if (!r) {
		return;
	}
	l = (int)strlen(r->suf);
	s = z->n - l;
	if (s < z->p1 || (r->cond == C_R2 && s < z->p2)) {
		return;
	}
	set_end(z, l, r->rep);
}

static void
step_4(stem_t *z)
{
	static const rule_t tab[] = {
		{ "al", "", 0 }, { "ance", "", 0 }, { "ence", "", 0 }, { "er", "", 0 }, { "ic", "", 0 },
		{ "able", "", 0 }, { "ible", "", 0 }, { "ant", "", 0 }, { "ement", "", 0 }, { "ment", "", 0 },
		{ "ent", "", 0 }, { "ism", "", 0 }, { "ate", "", 0 }, { "iti", "", 0 }, { "ous", "", 0 },
		{ "ive", "", 0 }, { "ize", "", 0 }, { "ion", "", C_ION },
	};
	const rule_t *r = longest(z, tab, sizeof(tab) / sizeof(tab[0]));
	int l, s;

	if (!r) {
		return;
	}
	l = (int)strlen(r->suf);
	s = z->n - l;
	if (s < z->p2) {
		return;
	}
	if (r->cond == C_ION && !(s > 0 && (z->w[s - 1] == 's' || z->w[s - 1] == 't'))) {
		return;
	}
	z->n = s;
}

static void
step_5(stem_t *z)
{
	if (z->n < 1) {
		return;
	}
	if (z->w[z->n - 1] == 'e') {
		const int s = z->n - 1;
		if (s >= z->p2 || (s >= z->p1 && !short_syllable(z, s))) {
			z->n = s;
		}
	} else if (z->w[z->n - 1] == 'l') {
		const int s = z->n - 1;
		if (s >= z->p2 && s > 0 && z->w[s - 1] == 'l') {
			z->n = s;
		}
	}
}

/* sb_stemmer_stem() of the English stemmer on a UTF-8 token; returns the new length */
size_t
nxs_stem_en(char *w, size_t len)
{
	stem_t z = { .w = w, .n = (int)len };
	bool y_found = false;
	int cps = 0;

	if (len == 0 || len > 0x3fffffff) {
		return len;
	}
	for (unsigned k = 0; k < sizeof(special) / sizeof(special[0]); k++) {
		if (whole_word(&z, special[k].from)) {
			const size_t l = strlen(special[k].to);
			memcpy(w, special[k].to, l);
			return l;
		}
	}
	/* words of fewer than three characters are left alone */
	for (int i = 0; i < z.n && cps < 3; i = cp_fwd(&z, i)) {
		cps++;
	}
	if (cps < 3) {
		return len;
	}
	/* prelude: a leading apostrophe goes; y at the start or after a vowel is a consonant (Y) */
	if (z.w[0] == '\'') {
		memmove(z.w, z.w + 1, (size_t)--z.n);
	}
	if (z.n > 0 && z.w[0] == 'y') {
		z.w[0] = 'Y';
		y_found = true;
	}
	for (int i = 0; i + 1 < z.n; i++) {
		if (vowel(z.w[i]) && z.w[i + 1] == 'y') {
			z.w[i + 1] = 'Y';
			y_found = true;
		}
	}
	regions(&z);
	step_1a(&z);
	{
		bool stop = false;
		for (unsigned k = 0; k < sizeof(after_1a) / sizeof(after_1a[0]); k++) {
			stop = stop || whole_word(&z, after_1a[k]);
		}
		if (!stop) {
			step_1b(&z);
			step_1c(&z);
			step_2(&z);
			step_3(&z);
			step_4(&z);
			step_5(&z);
		}
	}
	if (y_found) {
		for (int i = 0; i < z.n; i++) {
			if (z.w[i] == 'Y') {
				z.w[i] = 'y';
			}
		}
	}
	return (size_t)z.n;
}

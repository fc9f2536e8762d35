/*
 * nxs_filters.c -- the filter pipeline applied to QUERY tokens (N2).
 *
 * The reference runs every leaf string of a query through the index's filter
 * pipeline before it looks the token up (tokenize_value, src/core/tokenizer.c:
 * 205-227, called from query_prepare, src/query/query.c:99-106).  The pipeline
 * is the "filters" list of the index's params.db (nxs.c:87-90,263-266), run in
 * list order (filters.c); its built-in stages are
 *
 *   normalizer  src/core/filters_builtin.c:37-81 = utf8_normalize (NFKC_Casefold,
 *               src/utils/utf8.c:263-328) then utf8_subs_diacritics (the ICU
 *               transform "NFKD; [:Nonspacing Mark:] Remove; Latin-ASCII; NFKC",
 *               utf8.c:30-31,212-261)
 *   stopwords   filters_builtin.c:88-199: tokens listed in
 *               {basedir}/filters/stopwords/{lang} are DISCARDED (only "en" is
 *               ever loaded, :89); a discarded token leaves its leaf without a
 *               token => the empty set (search.c:140)
 *   stemmer     filters_builtin.c:203-245: sb_stemmer_stem() of the index's
 *               "lang" (default "en", nxs.c:271-276).  libstemmer is not in this
 *               image: English is the hand-written Porter2 of nxs_stem_en.c; an
 *               index of another language cannot be opened here (loud failure
 *               instead of silently unstemmed lookups)
 *
 * ICU is the same library the reference links (src/Makefile:86).  Pure-ASCII
 * tokens -- every token of the synthetic corpora -- take a fast path that is
 * provably the same result (case folding of ASCII is A-Z -> a-z, and every
 * stage of the diacritics transform is the identity on ASCII); others go
 * through ICU under a lock (a UTransliterator is not safe for concurrent use).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#include <unicode/utypes.h>
#include <unicode/ustring.h>
#include <unicode/unorm2.h>
#include <unicode/utrans.h>

#include "nxs_impl.h"

#define	NORM_BUF_MULTI	3	/* utf8.c:33 */

enum { F_NORMALIZER = 1, F_STOPWORDS = 2, F_STEMMER = 3 };

struct nxs_filters {
	unsigned		stage[4];
	unsigned		n_stages;
	const UNormalizer2 *	norm;
	UTransliterator *	trans;
	pthread_mutex_t		mu;
	/* stop words: open-addressing set */
	char **			sw;
	size_t			sw_cap, sw_n;
};

static uint64_t
str_hash(const char *s, size_t len)
{
	uint64_t h = 0xcbf29ce484222325ULL;
	for (size_t i = 0; i < len; i++) {
		h = (h ^ (unsigned char)s[i]) * 0x100000001b3ULL;
	}
	return h ^ (h >> 29);
}

static bool
sw_has(const nxs_filters_t *f, const char *s, size_t len)
{
	size_t i;

	if (!f->sw_cap) {
		return false;
	}
	i = str_hash(s, len) & (f->sw_cap - 1);
	while (f->sw[i]) {
		if (strlen(f->sw[i]) == len && memcmp(f->sw[i], s, len) == 0) {
			return true;
		}
		i = (i + 1) & (f->sw_cap - 1);
	}
	return false;
}

static void
sw_put(nxs_filters_t *f, const char *s, size_t len)
{
	size_t i;

	if ((f->sw_n + 1) * 2 > f->sw_cap) {
		const size_t ncap = f->sw_cap ? f->sw_cap * 2 : 256;
		char **old = f->sw;
		const size_t ocap = f->sw_cap;

		f->sw = calloc(ncap, sizeof(char *));
		f->sw_cap = ncap;
		f->sw_n = 0;
		for (size_t j = 0; j < ocap; j++) {
			if (old[j]) {
				sw_put(f, old[j], strlen(old[j]));
				free(old[j]);
			}
		}
		free(old);
	}
	if (sw_has(f, s, len)) {
		return;
	}
	i = str_hash(s, len) & (f->sw_cap - 1);
	while (f->sw[i]) {
		i = (i + 1) & (f->sw_cap - 1);
	}
	f->sw[i] = strndup(s, len);
	f->sw_n++;
}

/* stopwords_load: filters_builtin.c:91-127 (one word per line) */
static void
sw_load(nxs_filters_t *f, const char *basedir, const char *lang)
{
	char *path = NULL, *line = NULL;
	size_t lcap = 0;
	ssize_t len;
	FILE *fp;

	if (!basedir || strcmp(lang, "en") != 0) {	/* stopword_langs: :89 */
		return;
	}
	if (asprintf(&path, "%s/filters/stopwords/%s", basedir, lang) == -1) {
		return;
	}
	fp = fopen(path, "r");
	free(path);
	if (!fp) {
		return;		/* no stop words */
	}
	while ((len = getline(&line, &lcap, fp)) > 0) {
		if (len <= 1) {
			continue;
		}
		line[--len] = '\0';
		sw_put(f, line, (size_t)len);
	}
	free(line);
	fclose(fp);
}

/*
 * `names`: the "filters" list of params.db, in order.  Returns NULL and sets
 * *err (static text) when a stage cannot be provided.
 */
nxs_filters_t *
nxs_filters_create(const char *basedir, const char *const *names, size_t n,
    const char *lang, const char **err)
{
	static const char rule[] = "NFKD; [:Nonspacing Mark:] Remove; Latin-ASCII; NFKC";
	nxs_filters_t *f = calloc(1, sizeof(*f));
	UErrorCode ec = U_ZERO_ERROR;
	UChar urule[sizeof(rule)];

	*err = NULL;
	if (!f) {
		*err = "out of memory";
		return NULL;
	}
	pthread_mutex_init(&f->mu, NULL);
	for (size_t i = 0; i < n; i++) {
		if (strcmp(names[i], "normalizer") == 0) {
			if (f->n_stages < 4) f->stage[f->n_stages++] = F_NORMALIZER;
		} else if (strcmp(names[i], "stopwords") == 0) {
			if (f->n_stages < 4) f->stage[f->n_stages++] = F_STOPWORDS;
			sw_load(f, basedir, lang ? lang : "en");
		} else if (strcmp(names[i], "stemmer") == 0) {
			/* sb_stemmer_new(lang, NULL): libstemmer's names of the English stemmer */
			const char *l = lang ? lang : "en";
			if (strcmp(l, "en") != 0 && strcmp(l, "eng") != 0 && strcmp(l, "english") != 0) {
				*err = "the index uses the `stemmer' filter for a language other than "
				    "English, which this build cannot apply to query tokens "
				    "(libstemmer is not available; only \"en\" is built in)";
				nxs_filters_destroy(f);
				return NULL;
			}
			if (f->n_stages < 4) f->stage[f->n_stages++] = F_STEMMER;
		} else {
			*err = "the index uses a filter this build does not provide";
			nxs_filters_destroy(f);
			return NULL;
		}
	}
	f->norm = unorm2_getNFKCCasefoldInstance(&ec);		/* utf8.c:69 */
	if (U_FAILURE(ec)) {
		*err = "ICU: no NFKC_Casefold normalizer";
		nxs_filters_destroy(f);
		return NULL;
	}
	ec = U_ZERO_ERROR;
	u_strFromUTF8(urule, (int32_t)(sizeof(urule) / sizeof(urule[0])), NULL, rule, -1, &ec);
	if (!U_FAILURE(ec)) {
		ec = U_ZERO_ERROR;
		f->trans = utrans_openU(urule, -1, UTRANS_FORWARD, NULL, 0, NULL, &ec);	/* utf8.c:84 */
	}
	if (U_FAILURE(ec) || !f->trans) {
		*err = "ICU: cannot build the diacritics transform";
		nxs_filters_destroy(f);
		return NULL;
	}
	return f;
}

void
nxs_filters_destroy(nxs_filters_t *f)
{
	if (!f) {
		return;
	}
	if (f->trans) {
		utrans_close(f->trans);
	}
	for (size_t i = 0; i < f->sw_cap; i++) {
		free(f->sw[i]);
	}
	free(f->sw);
	pthread_mutex_destroy(&f->mu);
	free(f);
}

/* normalizer_filter (filters_builtin.c:56-76) through ICU; 0 / -1 */
static int
normalize_icu(nxs_filters_t *f, char **val, size_t *len)
{
	UErrorCode ec = U_ZERO_ERROR;
	const int32_t cap0 = (int32_t)(*len + 1) * 2;
	UChar *src = malloc((size_t)cap0 * sizeof(UChar)), *dst = NULL;
	int32_t n = 0, c, cap1, limit, out_len = 0;
	char *out = NULL;
	int ret = -1;

	if (!src) {
		return -1;
	}
	/* utf8_normalize: utf8.c:269-328 */
	u_strFromUTF8(src, cap0, &n, *val, (int32_t)*len, &ec);
	if (U_FAILURE(ec) || n >= cap0) {
		goto out;
	}
	cap1 = ((n + 1) * NORM_BUF_MULTI + 63) & ~63;
	if ((dst = malloc((size_t)cap1 * sizeof(UChar))) == NULL) {
		goto out;
	}
	ec = U_ZERO_ERROR;
	c = unorm2_normalize(f->norm, src, n, dst, cap1, &ec);
	if (U_FAILURE(ec) || c >= cap1) {
		goto out;
	}
	/* utf8_subs_diacritics: utf8.c:218-261 (on the UTF-16 text directly; the
	 * reference's round trip through UTF-8 in between is lossless) */
	{
		const int32_t cap2 = ((c + 1) * NORM_BUF_MULTI * NORM_BUF_MULTI + 63) & ~63;
		UChar *t = realloc(dst, (size_t)cap2 * sizeof(UChar));
		if (!t) {
			goto out;
		}
		dst = t;
		limit = c;
		ec = U_ZERO_ERROR;
		pthread_mutex_lock(&f->mu);
		utrans_transUChars(f->trans, dst, &c, cap2, 0, &limit, &ec);
		pthread_mutex_unlock(&f->mu);
		if (U_FAILURE(ec)) {
			goto out;
		}
	}
	if ((out = malloc((size_t)c * 3 + 4)) == NULL) {
		goto out;
	}
	ec = U_ZERO_ERROR;
	u_strToUTF8(out, c * 3 + 4, &out_len, dst, c, &ec);
	if (U_FAILURE(ec)) {
		free(out);
		goto out;
	}
	out[out_len] = '\0';
	free(*val);
	*val = out;
	*len = (size_t)out_len;
	ret = 0;
out:
	free(src);
	free(dst);
	return ret;
}

/*
 * filter_pipeline_run on one token (tokenizer.c:215).  *val is a malloc'd,
 * NUL-terminated string and may be replaced.  1 = keep (FILT_MUTATION),
 * 0 = discarded (FILT_DISCARD), -1 = FILT_ERROR.
 */
int
nxs_filters_run(nxs_filters_t *f, char **val, size_t *len)
{
	for (unsigned s = 0; f && s < f->n_stages; s++) {
		if (f->stage[s] == F_NORMALIZER) {
			bool ascii = true;

			for (size_t i = 0; i < *len; i++) {
				if ((unsigned char)(*val)[i] >= 0x80) {
					ascii = false;
					break;
				}
			}
			if (ascii) {
				for (size_t i = 0; i < *len; i++) {
					if ((*val)[i] >= 'A' && (*val)[i] <= 'Z') {
						(*val)[i] += 32;
					}
				}
			} else if (normalize_icu(f, val, len) == -1) {
				return -1;
			}
		} else if (f->stage[s] == F_STOPWORDS) {
			if (sw_has(f, *val, *len)) {
				return 0;
			}
		} else if (f->stage[s] == F_STEMMER) {
			/* stemmer_filter: filters_builtin.c:219-238 (in place: never longer) */
			*len = nxs_stem_en(*val, *len);
			(*val)[*len] = '\0';
		}
	}
	return 1;
}

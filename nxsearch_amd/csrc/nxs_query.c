/*
 * nxs_query.c -- query front end of the device path: lexer, parser, token
 * set, and compilation into the device plan (nxsgpu_query_t).
 *
 * Behaviour follows the reference's front end (restated by hand: its lexer and
 * parser are re2c/lemon inputs, tools this image does not have):
 *   lexer      src/query/scan.re:43-121   longest match, earlier rule on ties
 *   grammar    src/query/grammar.y:66-120 %left OR < AND < NOT; "AND NOT" is
 *              one rule; juxtaposition = OR at the top level only
 *   errors     src/query/query.c:46-58    "syntax error near L:C: "..." ..."
 *   prepare    src/query/query.c:75-115   leaves visited right-to-left
 *   token set  src/core/tokenizer.c:94-199 dedupe by bytes, first-seen order
 *
 * Instead of an expression tree the parser emits a postfix program directly
 * (operator-precedence / shunting-yard), which is what the device evaluates
 * per document on the term-presence mask.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nxs_impl.h"

/* ---- lexer -------------------------------------------------------------- */

enum { CC_END, CC_SPACE, CC_OPEN, CC_CLOSE, CC_OTHER };

static inline int
char_class(unsigned char c)
{
	switch (c) {
	case 0:
		return CC_END;
	case ' ': case '\t': case '\v': case '\f': case '\r': case '\n':
		return CC_SPACE;		/* SP: scan.re:59 */
	case '(':
		return CC_OPEN;
	case ')':
		return CC_CLOSE;
	default:
		return CC_OTHER;		/* FF_STR alphabet: scan.re:76 */
	}
}

typedef struct {
	const char *	cur;		/* cursor */
	const char *	tok;		/* start of the current token */
	const char *	line_start;	/* "cur_line" of the reference */
	unsigned	line;
	size_t		len;		/* length of the current token */
} scanner_t;

static inline int
lower(int c)
{
	return (c >= 'A' && c <= 'Z') ? c + 32 : c;
}

/* length of a quoted string starting at s (0 if unterminated): scan.re:72-74 */
static size_t
quoted_len(const char *s)
{
	const char q = s[0];
	size_t i = 1;

	while (s[i]) {
		if (s[i] == '\\') {
			if (!s[i + 1]) {
				return 0;
			}
			i += 2;
		} else if (s[i] == q) {
			return i + 1;
		} else {
			i++;
		}
	}
	return 0;
}

static qtoken_t
scan_next(scanner_t *sc)
{
	for (;;) {
		const char *p = sc->cur;
		size_t run = 0, kw = 0, qs = 0;
		qtoken_t kw_tok = QTK_EOF;

		sc->tok = p;
		switch (char_class((unsigned char)*p)) {
		case CC_END:
			sc->len = 0;
			return QTK_EOF;
		case CC_SPACE:
			while (char_class((unsigned char)p[run]) == CC_SPACE) {
				run++;
			}
			/* a lone "\n" is the EOL rule (line accounting), any longer
			 * run is the WSP rule: scan.re:89-90 */
			if (run == 1 && *p == '\n') {
				sc->line_start = p;
				sc->line++;
			}
			sc->cur = p + run;
			continue;
		case CC_OPEN:
			sc->cur = p + 1;
			sc->len = 1;
			return QTK_BR_OPEN;
		case CC_CLOSE:
			sc->cur = p + 1;
			sc->len = 1;
			return QTK_BR_CLOSE;
		default:
			break;
		}
		while (char_class((unsigned char)p[run]) == CC_OTHER) {
			run++;
		}
		/* operators: '&' | 'AND', '|' | 'OR', 'NOT', case-insensitive */
		if (p[0] == '&') {
			kw = 1; kw_tok = QTK_AND;
		} else if (p[0] == '|') {
			kw = 1; kw_tok = QTK_OR;
		} else if (lower(p[0]) == 'a' && lower(p[1]) == 'n' && lower(p[2]) == 'd') {
			kw = 3; kw_tok = QTK_AND;
		} else if (lower(p[0]) == 'o' && lower(p[1]) == 'r') {
			kw = 2; kw_tok = QTK_OR;
		} else if (lower(p[0]) == 'n' && lower(p[1]) == 'o' && lower(p[2]) == 't') {
			kw = 3; kw_tok = QTK_NOT;
		}
		if (p[0] == '\'' || p[0] == '"') {
			qs = quoted_len(p);
		}
		/* longest match; ties go to the earlier rule: operators, then
		 * quoted strings, then free-form strings */
		if (kw && kw >= run && kw >= qs) {
			sc->cur = p + kw;
			sc->len = kw;
			return kw_tok;
		}
		if (qs && qs >= run) {
			sc->cur = p + qs;
			sc->len = qs;
			return QTK_QUOTED_STRING;
		}
		sc->cur = p + run;
		sc->len = run;
		return QTK_FF_STRING;
	}
}

int
nxs_query_lex(const char *query, int *kinds, size_t cap)
{
	scanner_t sc = { .cur = query, .line_start = query, .line = 1 };
	qtoken_t tk;
	int n = 0;

	while ((tk = scan_next(&sc)) != QTK_EOF) {
		if ((size_t)n < cap) {
			kinds[n] = tk;
		}
		n++;
	}
	return n;
}

/* ---- parser: operator precedence -> postfix ------------------------------ */

enum { OP_PAREN = 1, OP_JUXT, OP_OR, OP_AND, OP_ANDNOT };

static inline int
op_prec(int op)
{
	switch (op) {
	case OP_JUXT:	return 1;	/* expr_list: below every operator */
	case OP_OR:	return 2;
	case OP_AND:
	case OP_ANDNOT:	return 3;	/* rule precedence = its left-most terminal */
	default:	return 0;
	}
}

/*
 * Everything a query's front half allocates -- postfix items, operator stack,
 * leaf strings, the token list and its values -- comes from ONE block sized
 * from the query's length (every item, leaf and token costs at least one
 * input byte): one malloc and one free per query instead of ~25 of each, which
 * at a thousand queries per batch was half a millisecond of the host's time
 * on the critical path.
 */
static void *
arena_take(qparse_t *o, size_t n)
{
	void *p = o->arena + o->arena_used;

	n = (n + 7) & ~(size_t)7;
	if (o->arena_used + n > o->arena_cap) {
		return NULL;	/* cannot happen: the block is sized for the worst case */
	}
	o->arena_used += n;
	return p;
}

static char *
arena_strndup(qparse_t *o, const char *s, size_t n)
{
	char *p = arena_take(o, n + 1);

	if (p) {
		memcpy(p, s, n);
		p[n] = '\0';
	}
	return p;
}

typedef struct {
	qparse_t *	out;
	int *		ops;
	size_t		n_ops;
} pstate_t;

static void
emit(pstate_t *ps, uint8_t op, char *str)
{
	qparse_t *o = ps->out;

	o->items[o->n].op = op;
	o->items[o->n].str = str;
	o->items[o->n].token = -1;
	o->n++;
}

static void
emit_op(pstate_t *ps, int op)
{
	emit(ps, op == OP_AND ? NXSGPU_OP_AND :
	    op == OP_ANDNOT ? NXSGPU_OP_ANDNOT : NXSGPU_OP_OR, NULL);
}

static void
push_op(pstate_t *ps, int op)
{
	ps->ops[ps->n_ops++] = op;
}

/* all binary operators are left-associative: reduce while top >= incoming */
static void
reduce_for(pstate_t *ps, int op)
{
	while (ps->n_ops && ps->ops[ps->n_ops - 1] != OP_PAREN &&
	    op_prec(ps->ops[ps->n_ops - 1]) >= op_prec(op)) {
		emit_op(ps, ps->ops[--ps->n_ops]);
	}
}

static void
syntax_error(qparse_t *out, const scanner_t *sc)
{
	if (!out->error) {
		const unsigned col = (unsigned)(sc->tok - sc->line_start);
		if (asprintf(&out->errmsg, "syntax error near %u:%u: \"%.50s ...\"",
		    sc->line, col, sc->tok) == -1) {
			out->errmsg = NULL;
		}
		out->error = true;
	}
}

void
nxs_query_parse(const char *query, qparse_t *out)
{
	scanner_t sc = { .cur = query, .line_start = query, .line = 1 };
	pstate_t ps = { .out = out };
	enum { WANT_OPERAND, AFTER_AND, WANT_OPERATOR } st = WANT_OPERAND;
	unsigned depth = 0;
	const size_t len = strlen(query);

	memset(out, 0, sizeof(*out));
	/* items + operator stack (one per input token at most, plus the implied ORs
	 * of juxtaposition), leaf strings, then the token list and values of
	 * nxs_query_prepare (with room for what the normalizer may expand) */
	out->arena_cap = (2 * len + 4) * (sizeof(qitem_t) + sizeof(int)) + (len + 8) * 2
	    + (len + 2) * sizeof(qtok_t) + (len + 8) * 4 + 256;
	out->arena = malloc(out->arena_cap);
	if (!out->arena) {
		out->error = true;
		return;
	}
	out->items = arena_take(out, (2 * len + 4) * sizeof(qitem_t));
	ps.ops = arena_take(out, (2 * len + 4) * sizeof(int));
	for (;;) {
		const qtoken_t tk = scan_next(&sc);
		const bool operand = tk == QTK_FF_STRING || tk == QTK_QUOTED_STRING;

		if (st == AFTER_AND && tk == QTK_NOT) {
			ps.ops[ps.n_ops - 1] = OP_ANDNOT;	/* grammar.y:96-99 */
			st = WANT_OPERAND;
			continue;
		}
		if (st == WANT_OPERAND || st == AFTER_AND) {
			if (operand) {
				char *s = (tk == QTK_QUOTED_STRING) ?
				    strndup(sc.tok + 1, sc.len - 2) :	/* scan.re:108 */
				    strndup(sc.tok, sc.len);		/* scan.re:115 */
				emit(&ps, 0, s);
				st = WANT_OPERATOR;
				continue;
			}
			if (tk == QTK_BR_OPEN) {
				push_op(&ps, OP_PAREN);
				depth++;
				st = WANT_OPERAND;
				continue;
			}
			syntax_error(out, &sc);
			break;
		}
		/* WANT_OPERATOR */
		if (tk == QTK_AND) {
			reduce_for(&ps, OP_AND);
			push_op(&ps, OP_AND);
			st = AFTER_AND;
			continue;
		}
		if (tk == QTK_OR) {
			reduce_for(&ps, OP_OR);
			push_op(&ps, OP_OR);
			st = WANT_OPERAND;
			continue;
		}
		if (tk == QTK_BR_CLOSE && depth) {
			while (ps.ops[ps.n_ops - 1] != OP_PAREN) {
				emit_op(&ps, ps.ops[--ps.n_ops]);
			}
			ps.n_ops--;
			depth--;
			continue;
		}
		if ((operand || tk == QTK_BR_OPEN) && depth == 0) {
			/* expr_list ::= expr_list expr  => OR (grammar.y:81-84) */
			reduce_for(&ps, OP_JUXT);
			push_op(&ps, OP_JUXT);
			if (operand) {
				char *s = (tk == QTK_QUOTED_STRING) ?
				    strndup(sc.tok + 1, sc.len - 2) :
				    strndup(sc.tok, sc.len);
				emit(&ps, 0, s);
				st = WANT_OPERATOR;
			} else {
				push_op(&ps, OP_PAREN);
				depth++;
				st = WANT_OPERAND;
			}
			continue;
		}
		if (tk == QTK_EOF && depth == 0) {
			while (ps.n_ops) {
				emit_op(&ps, ps.ops[--ps.n_ops]);
			}
			break;
		}
		syntax_error(out, &sc);
		break;
	}
}

void
nxs_query_free(qparse_t *q)
{
	free(q->arena);		/* items, leaf strings, tokens: all of it */
	free(q->errmsg);
	memset(q, 0, sizeof(*q));
}

/* IR dump in the format of src/tests/t_queryparser.c:146-169 */
char *
nxs_query_repr(const qparse_t *q)
{
	char **st = calloc(q->n + 1, sizeof(char *));
	size_t sp = 0;
	char *res;

	if (q->error || q->n == 0) {
		free(st);
		return NULL;
	}
	for (size_t i = 0; i < q->n; i++) {
		const qitem_t *it = &q->items[i];
		char *s = NULL;

		if (it->op == 0) {
			if (asprintf(&s, "`%s`", it->str) == -1) s = NULL;
		} else {
			const char *name = it->op == NXSGPU_OP_AND ? "AND" :
			    it->op == NXSGPU_OP_OR ? "OR" : "NOT";
			char *b = st[--sp], *a = st[--sp];
			if (asprintf(&s, "(%s %s %s)", name, a, b) == -1) s = NULL;
			free(a);
			free(b);
		}
		st[sp++] = s;
	}
	res = st[0];
	free(st);
	return res;
}

/* ---- prepare: token set -------------------------------------------------- */

void
nxs_query_prepare(const nxs_index_t *idx, const char *query, qprep_t *out)
{
	qparse_t *pr = &out->parse;

	memset(out, 0, sizeof(*out));
	nxs_query_parse(query, pr);
	if (pr->error) {
		/* construct_query: search.c:190-194 */
		out->errcode = NXS_ERR_INVALID;
		if (asprintf(&out->errmsg, "query failed with %s",
		    pr->errmsg ? pr->errmsg : "out of memory") == -1) {
			out->errmsg = NULL;
		}
		return;
	}
	out->tokens = arena_take(pr, (pr->n + 1) * sizeof(qtok_t));
	if (!out->tokens) {
		out->errcode = NXS_ERR_SYSTEM;
		out->errmsg = strdup("out of memory");
		return;
	}
	memset(out->tokens, 0, (pr->n + 1) * sizeof(qtok_t));

	/*
	 * query_prepare pops its explicit stack from the back after pushing
	 * children left to right (query.c:89-95): leaves are met right to
	 * left, i.e. the postfix leaves in reverse.  Identical (filtered)
	 * strings share one token (tokenizer.c:100-107).
	 */
	for (size_t k = pr->n; k-- > 0; ) {
		qitem_t *it = &pr->items[k];
		size_t len, j;
		char *val;

		if (it->op != 0) {
			continue;
		}
		len = strlen(it->str);
		val = it->str;		/* (in the arena; a filter may hand back a malloc'd string) */
		/* tokenize_value: the index's filter pipeline on the leaf string
		 * (tokenizer.c:205-227; nxs_filters.c) */
		if (idx && idx->filters) {
			char *fv = val;
			const int act = nxs_filters_run(idx->filters, &fv, &len);

			if (act == 1 && fv != val) {
				/* the ICU path replaced the string: keep a copy in the arena
				 * (or, if it grew beyond the block's reserve, on the heap of the
				 * query: freed with the token list) */
				char *cp = arena_strndup(pr, fv, len);
				if (cp) {
					free(fv);
					fv = cp;
				} else {
					out->heap_vals = realloc(out->heap_vals, (out->n_heap_vals + 1) * sizeof(char *));
					out->heap_vals[out->n_heap_vals++] = fv;
				}
			} else if (act != 1 && fv != val) {
				free(fv);
			}
			val = fv;
			if (act == 0) {
				/* FILT_DISCARD (a stop word): no token; the leaf is the
				 * empty set (search.c:140) */
				continue;
			}
			if (act < 0) {
				/* FILT_ERROR => query_prepare fails (search.c:199-203) */
				out->errcode = NXS_ERR_FATAL;
				out->errmsg = strdup("query_prepare() failed");
				return;
			}
		} else if (idx && idx->lowercase) {
			/* (host-only tests without a pipeline object) */
			for (size_t c = 0; c < len; c++) {
				if (val[c] >= 'A' && val[c] <= 'Z') {
					val[c] += 32;
				}
			}
		}
		for (j = 0; j < out->n_tokens; j++) {
			if (out->tokens[j].len == len &&
			    memcmp(out->tokens[j].value, val, len) == 0) {
				break;
			}
		}
		if (j == out->n_tokens) {
			out->tokens[j].value = val;
			out->tokens[j].len = len;
			out->tokens[j].term_id = 0;
			out->n_tokens++;
		}
		it->token = (int)j;
	}
}

/*
 * Compile once the term ids are final (exact + fuzzy resolution done):
 * unresolved tokens are trimmed (TOKENSET_TRIM, tokenizer.c:186-192) and
 * their leaves evaluate to the empty set (search.c:140 for a NULL token; the
 * reference's behaviour for a trimmed one is a use-after-free, Q14).
 */
int
nxs_query_compile(qprep_t *q)
{
	const qparse_t *pr = &q->parse;
	nxsgpu_query_t *pl = &q->plan;
	int bit_small[64];
	unsigned hstack_small[160];
	int *bit = q->n_tokens + 1 <= 64 ? bit_small : calloc(q->n_tokens + 1, sizeof(int));
	unsigned *hstack = pr->n + 1 <= 160 ? hstack_small : calloc(pr->n + 1, sizeof(unsigned));
	size_t sp = 0, maxsp = 0;
	uint32_t live = 0;
	int ret = -1;

	memset(pl, 0, sizeof(*pl));
	for (size_t j = 0; j < q->n_tokens; j++) {
		bit[j] = q->tokens[j].term_id ? (int)live++ : -1;
	}
	if (live == 0) {
		q->empty = true;	/* run_query_logic: search.c:224-226 */
		ret = 0;
		goto out;
	}
	if (live > NXSGPU_WIDE_MAX_TOKENS) {
		q->errcode = NXS_ERR_LIMIT;
		if (asprintf(&q->errmsg, "too many query terms for the device path "
		    "(%u, limit %u)", live, NXSGPU_WIDE_MAX_TOKENS) == -1) q->errmsg = NULL;
		goto out;
	}
	/* height of the tree first (get_expr_bitmap recursion: search.c:126-131) */
	for (size_t i = 0; i < pr->n; i++) {
		if (pr->items[i].op == 0) {
			hstack[sp++] = 0;
		} else {
			const unsigned hb = hstack[--sp], ha = hstack[--sp];
			hstack[sp++] = 1 + (ha > hb ? ha : hb);
		}
		if (sp > maxsp) {
			maxsp = sp;
		}
	}
	if (hstack[0] > NXS_QUERY_RLIMIT) {
		q->errcode = NXS_ERR_LIMIT;
		if (asprintf(&q->errmsg, "query nesting limit reached (%u levels)",
		    NXS_QUERY_RLIMIT) == -1) q->errmsg = NULL;
		goto out;
	}
	/*
	 * More than the fixed-size plan holds (run_query_logic, search.c:210-278,
	 * has no bound on terms): a variable-size "wide" plan for the generic
	 * kernel (k_scanw) on the exact path.  Its evaluation stack is 128 deep; the
	 * nesting limit above keeps any program at 101 or less.
	 */
	if (live > NXSGPU_MAX_TOKENS || pr->n > NXSGPU_MAX_PROG || maxsp > 64) {
		uint32_t *wt = calloc(live, sizeof(uint32_t));
		uint16_t *wp = calloc(pr->n, sizeof(uint16_t));

		if (!wt || !wp || pr->n > 2 * NXSGPU_WIDE_MAX_TOKENS || maxsp > 128) {
			free(wt);
			free(wp);
			q->errcode = NXS_ERR_LIMIT;
			if (asprintf(&q->errmsg, "query too long for the device path "
			    "(%zu items)", pr->n) == -1) q->errmsg = NULL;
			goto out;
		}
		for (size_t j = 0; j < q->n_tokens; j++) {
			if (bit[j] >= 0) {
				wt[bit[j]] = q->tokens[j].term_id;
			}
		}
		for (size_t i = 0; i < pr->n; i++) {
			const qitem_t *it = &pr->items[i];

			if (it->op == 0) {
				const int b = it->token >= 0 ? bit[it->token] : -1;
				wp[i] = b >= 0 ? (uint16_t)b : NXSGPU_WOP_EMPTY;
			} else {
				wp[i] = it->op == NXSGPU_OP_AND ? NXSGPU_WOP_AND :
				    it->op == NXSGPU_OP_OR ? NXSGPU_WOP_OR : NXSGPU_WOP_ANDNOT;
			}
		}
		q->wide = true;
		q->wplan.n_tokens = live;
		q->wplan.term_id = wt;
		q->wplan.prog_len = (uint32_t)pr->n;
		q->wplan.prog = wp;
		ret = 0;
		goto out;
	}
	pl->n_tokens = live;
	for (size_t j = 0; j < q->n_tokens; j++) {
		if (bit[j] >= 0) {
			pl->term_id[bit[j]] = q->tokens[j].term_id;
		}
	}
	for (size_t i = 0; i < pr->n; i++) {
		const qitem_t *it = &pr->items[i];

		if (it->op == 0) {
			const int b = it->token >= 0 ? bit[it->token] : -1;
			pl->prog[pl->prog_len++] = b >= 0 ? (uint8_t)b : NXSGPU_OP_EMPTY;
		} else {
			pl->prog[pl->prog_len++] = it->op;
		}
	}
	/* truth table over the presence mask for the 8-token fast path */
	if (live <= 8) {
		for (unsigned m = 0; m < (1u << live); m++) {
			uint64_t st = 0;
			for (uint32_t i = 0; i < pl->prog_len; i++) {
				const uint8_t op = pl->prog[i];
				if (op < NXSGPU_MAX_TOKENS) {
					st = (st << 1) | ((m >> op) & 1);
				} else if (op == NXSGPU_OP_EMPTY) {
					st <<= 1;
				} else {
					const uint64_t b = st & 1, a = (st >> 1) & 1;
					const uint64_t r = op == NXSGPU_OP_AND ? (a & b) :
					    op == NXSGPU_OP_OR ? (a | b) : (a & ~b & 1);
					st = ((st >> 2) << 1) | r;
				}
			}
			if (st & 1) {
				pl->truth[m >> 5] |= 1u << (m & 31);
			}
		}
	}
	ret = 0;
out:
	if (bit != bit_small) {
		free(bit);
	}
	if (hstack != hstack_small) {
		free(hstack);
	}
	return ret;
}

/* the parse and the token list are needed until the plan is compiled; what
 * outlives them: errcode / errmsg, empty, the plans */
void
nxs_query_release_scratch(qprep_t *q)
{
	for (size_t j = 0; j < q->n_heap_vals; j++) {
		free(q->heap_vals[j]);
	}
	free(q->heap_vals);
	q->heap_vals = NULL;
	q->n_heap_vals = 0;
	q->tokens = NULL;
	q->n_tokens = 0;
	nxs_query_free(&q->parse);
}

void
nxs_query_release(qprep_t *q)
{
	nxs_query_release_scratch(q);
	free(q->errmsg);
	free((void *)q->wplan.term_id);
	free((void *)q->wplan.prog);
	nxs_query_free(&q->parse);
	memset(q, 0, sizeof(*q));
}

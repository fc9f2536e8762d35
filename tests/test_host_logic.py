"""CPU tier for the product's host side (no GPU): the hand-written lexer /
parser / plan compiler, the host Levenshtein + BK-tree image, the synthetic
corpus writer, and the C-ABI surface.  The oracle is used as the checker."""
import ctypes as C
import os
import random
import re

import pytest

import nxsearch_amd as N
import nxsfmt
import oracle_lib as O
from nxsearch_amd import corpus

TK = {1: "AND", 2: "OR", 3: "NOT", 4: "(", 5: ")", 6: "FF", 7: "QUOTED"}
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_exports_every_declared_symbol():
    L = C.CDLL(N.LIB_PATH)
    for sym in N.NXS_H_SYMBOLS + N.NXS_GPU_H_SYMBOLS:
        assert hasattr(L, sym), sym
    # and the headers declare nothing that the lists above miss
    for hdr, names in (("nxs.h", N.NXS_H_SYMBOLS), ("nxs_gpu.h", N.NXS_GPU_H_SYMBOLS)):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared = set(re.findall(r"\b(nxs(?:gpu)?_[a-z0-9_]+)\s*\(", text))
        assert declared == set(names), (hdr, declared ^ set(names))
    # test hooks and bench accessors are NOT in the public header (csrc/nxs_hooks.h) ...
    for sym in N.NXS_HOOK_SYMBOLS:
        assert hasattr(L, sym), sym


def test_production_build_has_no_test_hooks():
    """`make nohooks` = the library a consumer of include/nxs.h links: every declared symbol,
    none of nxs_hooks.h's (nxs_test_*, nxs_index_device, nxs_index_host_profile)."""
    import subprocess
    csrc = os.path.join(ROOT, "nxsearch_amd", "csrc")
    subprocess.run(["make", "-j8", "-C", csrc, "nohooks"], check=True, capture_output=True)
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(csrc, "libnxsearch_gpu_nohooks.so")],
                         check=True, capture_output=True, text=True).stdout
    syms = set(l.split()[-1] for l in out.splitlines() if l.strip())
    for sym in N.NXS_H_SYMBOLS + N.NXS_GPU_H_SYMBOLS:
        assert sym in syms, sym
    assert not [x for x in syms if x.startswith("nxs_test_")]
    assert not (set(N.NXS_HOOK_SYMBOLS) & syms)


def test_params_fromjson_is_what_the_lua_binding_needs(tmp_path):
    """nxs_params_fromjson (reference params.c:201-208; caller lua.c:99-110): the
    root object's string / unsigned / bool members, by type like the reference's
    getters (params.c:125-155); a syntax error is NXS_ERR_SYSTEM "params parsing
    failed: ..."."""
    L = N.lib()
    with N.Nxs(str(tmp_path)) as nxs:
        def parse(js):
            b = js.encode() if isinstance(js, str) else js
            return L.nxs_params_fromjson(nxs._h, b, len(b))

        def get(p, key):
            u, b = C.c_uint64(), C.c_bool()
            if L.nxs_params_get_uint(p, key, C.byref(u)) == 0:
                return int(u.value)
            if L.nxs_params_get_bool(p, key, C.byref(b)) == 0:
                return bool(b.value)
            v = L.nxs_params_get_str(p, key)
            return v.decode() if v is not None else None

        p = parse(' {"limit": 25, "algo" : "TF-IDF", "fuzzymatch": false, "lang":"en",\n'
                  '  "filters": ["normalizer", "stemmer"], "neg": -3, "real": 1.5, "nil": null,\n'
                  '  "nested": {"limit": 7}, "esc": "a\\"b\\u00e9\\ud83d\\ude00\\n", "big": 18446744073709551615,\n'
                  '  "huge": 18446744073709551616, "t": true} ')
        assert p
        assert get(p, b"limit") == 25 and get(p, b"algo") == "TF-IDF" and get(p, b"fuzzymatch") is False
        assert get(p, b"t") is True and get(p, b"lang") == "en"
        assert get(p, b"esc") == 'a"b\u00e9\U0001f600\n'
        assert get(p, b"big") == 2 ** 64 - 1
        for k in (b"neg", b"real", b"nil", b"nested", b"filters", b"huge", b"missing"):
            assert get(p, k) is None, k           # not a string / unsigned / bool member of the root
        L.nxs_params_release(p)
        # duplicate keys: the reference's getters use yyjson_mut_obj_get, i.e. the FIRST member of that
        # name whatever its kind -- a later duplicate is never seen, a first member of the wrong kind
        # hides a usable one behind it (search.c:96-101 then falls back to the default limit)
        for js, want in (('{"limit": 10, "limit": 20}', 10), ('{"limit": "x", "limit": 10}', "x"),
                         ('{"limit": null, "limit": 10}', None), ('{"limit": -1, "limit": 10}', None),
                         ('{"limit": [1], "limit": 10}', None), ('{"limit": 10, "limit": "x"}', 10)):
            p = parse(js)
            assert p and get(p, b"limit") == want, js
            L.nxs_params_release(p)
        for ok in ("{}", "[1, 2]", "3", '"x"', "  null  "):    # valid JSON, no parameters
            p = parse(ok)
            assert p and get(p, b"limit") is None
            L.nxs_params_release(p)
        for bad in ("", "{", '{"limit": 10,}', '{"limit" 10}', "{'limit': 10}", '{"a": 01}', '{"a": 1} x',
                    '{"a": "unclosed}', '{"a": tru}', '{"a": "\\x"}', '{"a": 1.}', '[1, 2', '{"a": "\\ud800"}'):
            assert not parse(bad), bad
            code, msg = nxs.error()
            assert code == 2 and msg.startswith("params parsing failed: "), (bad, code, msg)


@pytest.mark.timeout(120)
def test_worker_pool_hands_every_item_out_exactly_once():
    """The front half of a batch (parse, resolve, compile) runs on a worker pool
    whose runs wait for their ITEMS, not for their slowest sleeper: a thread that
    wakes up late must neither take items of the next run away nor work on stale
    ones.  Thousands of back-to-back runs with more threads than cores."""
    L = N.lib()
    L.nxs_test_pool.restype = C.c_size_t
    L.nxs_test_pool.argtypes = [C.c_uint, C.c_size_t, C.c_uint, C.c_size_t]
    assert L.nxs_test_pool(15, 1024, 3000, 16) == 0
    assert L.nxs_test_pool(31, 100, 3000, 1) == 0
    assert L.nxs_test_pool(3, 5000, 300, 32) == 0
    assert L.nxs_test_pool(0, 64, 10, 16) == 0


def test_no_gpu_means_loud_failure(tmp_path):
    if N.lib().nxsgpu_device_count() > 0:
        pytest.skip("a GPU is present")
    t, d, _ = nxsfmt.write_index(str(tmp_path), "idx", [(1, ["cat"])])
    with N.Nxs(str(tmp_path)) as nxs:
        with pytest.raises(N.NxsError) as e:
            nxs.open_index("idx")
        assert e.value.code == 2 and "no HIP device" in e.value.msg


def test_parser_golden_vectors(golden):
    for c in golden["queryparser"]["cases"]:
        assert [TK[k] for k in N.query_lex(c["query"])] == c["tokens"], c["query"]
        rep, err = N.query_repr(c["query"])
        if c["repr"] is None:
            assert rep is None and err.startswith("syntax error near")
        else:
            assert rep == c["repr"]


def test_parser_matches_oracle_on_random_queries():
    rng = random.Random(2024)
    atoms = ["a", "bb", "AND", "and", "OR", "or", "NOT", "not", "&", "|", "(", ")",
             "(", ")", "'q s'", '"d q"', "'un", "ANDx", "xOR", "\n", "  ", "\t",
             "ж", "a&b", "'a'b", "\\", "'e\\'s'", "c"]
    n_err = 0
    for _ in range(6000):
        q = " ".join(rng.choice(atoms) for _ in range(rng.randint(0, 9)))
        if rng.random() < 0.3:
            q = q.replace(" ", "", rng.randint(1, 3))
        assert N.query_lex(q) == O.query_lex(q), repr(q)
        got, want = N.query_repr(q), O.query_repr(q)
        assert got == want, repr(q)
        n_err += got[0] is None
    assert 500 < n_err < 5500          # both outcomes are well exercised


def _eval_prog(plan, mask):
    st = []
    for op in plan.prog[:plan.prog_len]:
        if op < 32:
            st.append((mask >> op) & 1)
        elif op == 0x40:
            st.append(0)
        else:
            b, a = st.pop(), st.pop()
            st.append(a & b if op == 0x80 else a | b if op == 0x81 else a & (1 - b))
    assert len(st) == 1
    return st[0]


def test_plan_token_order_truth_table_and_empty_leaves():
    words = ["a", "b", "c", "d", "e"]
    # right-to-left leaf order (query.c:89-95): a AND b AND c => c, b, a
    code, err, empty, p = N.compile_query("a AND b AND c", words)
    assert (code, empty, p.n_tokens) == (0, False, 3)
    assert list(p.term_id[:3]) == [3, 2, 1]
    # identical strings share a token, an unknown word is an empty-set leaf
    code, err, empty, p = N.compile_query("a OR zz OR a AND b", words)
    assert p.n_tokens == 2 and list(p.term_id[:2]) == [2, 1]
    assert 0x40 in list(p.prog[:p.prog_len])
    for m in range(4):
        a, b = (m >> 1) & 1, m & 1
        assert ((p.truth[0] >> m) & 1) == (a | (a & b)) == _eval_prog(p, m)
    # all tokens unknown => empty result, not an error (search.c:224-226)
    code, err, empty, p = N.compile_query("zz OR yy", words)
    assert (code, empty) == (0, True)
    # syntax error => NXS_ERR_INVALID with the reference's message shape
    code, err, empty, p = N.compile_query("a AND", words)
    assert code == 3 and err == 'query failed with syntax error near 1:5: " ..."'
    # nesting limit (search.c:70,126-131)
    code, err, _, _ = N.compile_query(" OR ".join(["a"] * 102), words)
    assert code == 6 and err == "query nesting limit reached (100 levels)"
    code, err, _, _ = N.compile_query(" OR ".join(["a"] * 101), words)
    assert code == 0
    # lower-casing stand-in for the normalizer filter
    code, err, empty, p = N.compile_query("A AND B", words, lowercase=True)
    assert (code, empty, p.n_tokens) == (0, False, 2)
    code, err, empty, p = N.compile_query("A AND B", words, lowercase=False)
    assert empty


def test_query_token_filters(golden, tmp_path):
    """N2: the normalizer stage (ICU NFKC_Casefold + the diacritics transform of
    src/utils/utf8.c:30-31) on the reference's own known answers
    (src/tests/t_utf8.c:85-150), stop words (filters_builtin.c:88-199), and the
    English stemmer, and the loud refusal of a stemmer for another language."""
    for src, want in golden["utf8"]["normalize"]:
        assert N.filter_token(src) == (1, want)
    # the filter runs utf8_normalize THEN utf8_subs_diacritics (filters_builtin.c:56-76):
    # the diacritics answers arrive case-folded
    for src, want in golden["utf8"]["diacritics"]:
        assert N.filter_token(src) == (1, want.lower())
    assert N.filter_token("UNIX") == (1, "unix")                 # ASCII fast path
    assert N.filter_token("ﬁnance ①") == (1, "finance 1")
    assert N.filter_token(b"\xff\xfe")[0] == -1                 # invalid UTF-8 => FILT_ERROR
    sw = tmp_path / "filters" / "stopwords"
    sw.mkdir(parents=True)
    (sw / "en").write_text("the\nand\n\nof\n")
    assert N.filter_token("The", basedir=str(tmp_path), stopwords=True) == (0, None)
    assert N.filter_token("Then", basedir=str(tmp_path), stopwords=True) == (1, "then")
    assert N.filter_token("the", basedir=str(tmp_path), stopwords=False) == (1, "the")
    # the stemmer stage runs after the normalizer (nxs.c:87-89: normalizer, stopwords, stemmer)
    assert N.filter_token("Foxes", stemmer=True) == (1, "fox")
    assert N.filter_token("JUMPED", stemmer=True) == (1, "jump")
    assert N.filter_token("Ⅷ", stemmer=True) == (1, "viii")
    # a stemmer for another language cannot be provided here (libstemmer absent): say so at open.
    # (lang "en" -- the default-created index -- passes this stage; without a GPU the open then
    # stops at the device, the GPU tier opens one: test_default_filters_raw_text_scoring)
    nxsfmt.write_index(str(tmp_path), "stemmed_de", [(1, ["cat"])], filters=["normalizer", "stopwords", "stemmer"],
                       lang="de")
    nxsfmt.write_index(str(tmp_path), "stemmed_en", [(1, ["cat"])], filters=["normalizer", "stopwords", "stemmer"])
    with N.Nxs(str(tmp_path)) as nxs:
        with pytest.raises(N.NxsError) as e:
            nxs.open_index("stemmed_de")
        assert e.value.code == 3 and "stemmer" in e.value.msg and "English" in e.value.msg
        if N.lib().nxsgpu_device_count() <= 0:
            with pytest.raises(N.NxsError) as e:
                nxs.open_index("stemmed_en")
            assert "stemmer" not in e.value.msg


def test_english_stemmer_product_vs_oracle(golden):
    """N2: the product's Porter2 (csrc/nxs_stem_en.c) and the oracle's (oracle/orc_stem_en.c) are two
    constructions of the published algorithm: both give the published sample vocabulary and the
    description's worked examples, the stems the reference's own tests imply (t_scoring.c: the
    pre-stemmed token streams of golden `scoring` are the raw texts of `scoring_raw` stemmed), and
    agree on 60k generated words (suffix chains, the special prefixes, apostrophes, non-ASCII)."""
    import json
    import random
    import oracle_lib as O

    def orc(w):
        b = w.encode()
        out = C.create_string_buffer(len(b) + 2)
        n = O.lib().orc_stem_en(b, len(b), out, len(b) + 2)
        return out.raw[:n].decode()

    def prod(w):
        act, r = N.filter_token(w, stemmer=True)
        assert act == 1
        return r

    pub = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "porter2_published.json")))
    for w, want in pub["sample"] + pub["rules"]:
        assert (orc(w), prod(w)) == (want, want), w
    for raw, st in zip(golden["scoring_raw"]["cases"], golden["scoring"]["cases"]):
        for k, text in raw["docs"].items():
            assert [orc(t.lower()) for t in text.split()] == st["docs"][k].split()
            assert [prod(t) for t in text.split()] == st["docs"][k].split()
    rng = random.Random(7)
    sufs = ["", "s", "es", "ed", "ing", "ly", "edly", "ingly", "eed", "eedly", "ies", "ied", "sses", "'s", "'s'", "'",
            "ational", "tional", "enci", "anci", "abli", "entli", "izer", "ization", "ation", "ator", "alism", "aliti",
            "alli", "fulness", "ousli", "ousness", "iveness", "iviti", "biliti", "bli", "ogi", "logi", "fulli", "lessli",
            "li", "alize", "icate", "iciti", "ical", "ful", "ness", "ative", "al", "ance", "ence", "er", "ic", "able",
            "ible", "ant", "ement", "ment", "ent", "ism", "ate", "iti", "ous", "ive", "ize", "ion", "sion", "tion", "e",
            "l", "ll", "y", "ey", "ay", "us", "ss", "at", "bl", "iz", "bb", "tt", "д", "яy", "y's", "дe", "яs", "дies"]
    letters = "aeiouybcdfghklmnprstvwxzдя"
    heads = ["gener", "commun", "arsen", "y", "'", "д", "sk", "succ", "proc", "exc", "inn", "out", "cann", "herr", "earr"]
    n = 0
    while n < 60000:
        w = "".join(rng.choice(letters) for _ in range(rng.randint(0, 6)))
        if rng.random() < 0.1:
            w = rng.choice(heads) + w
        w += rng.choice(sufs) + (rng.choice(sufs) if rng.random() < 0.3 else "")
        if not w:
            continue
        n += 1
        assert prod(w) == orc(N.filter_token(w)[1]), w


def _eval_wide(prog, present):
    st = []
    for op in prog:
        if op < 0x8000:
            st.append(op in present)
        elif op == 0xffff:
            st.append(False)
        else:
            b, a = st.pop(), st.pop()
            st.append((a and b) if op == 0xfff0 else (a or b) if op == 0xfff1 else (a and not b))
    assert len(st) == 1
    return st[0]


def test_wide_plans_beyond_32_tokens():
    """More than 32 live tokens / 256 program items / a 64-deep stack compile into
    the variable-size wide plan (the reference has no such bounds); its program
    is checked against Python set algebra on random presence sets."""
    words = ["w%d" % i for i in range(1, 1101)]
    rng = random.Random(77)
    # a OR b OR ... (40 terms): token order is right-to-left
    code, wide, tids, prog = N.compile_wide(" OR ".join(words[:40]), words)
    assert (code, wide) == (0, True) and tids == list(range(40, 0, -1))
    assert len(prog) == 79 and prog.count(0xfff1) == 39
    # 32 tokens still take the fixed-size plan
    code, wide, _, _ = N.compile_wide(" OR ".join(words[:32]), words)
    assert (code, wide) == (0, False)
    # > 256 program items with few tokens; deep right nesting
    code, wide, tids, prog = N.compile_wide(
        " OR ".join("(%s AND %s)" % (words[i % 20], words[(i * 7 + 3) % 20]) for i in range(70)), words)
    assert (code, wide, len(tids), len(prog)) == (0, True, 20, 279)
    q = "".join("%s AND (" % words[i] for i in range(69)) + words[69] + ")" * 69
    code, wide, tids, prog = N.compile_wide(q, words)
    assert (code, wide, len(tids)) == (0, True, 70)
    for _ in range(20):
        present = {i for i in range(70) if rng.random() < 0.9}
        assert _eval_wide(prog, present) == (len(present) == 70)
    # random mixed expressions over 50..200 tokens
    for _ in range(40):
        n = rng.randint(50, 200)
        ws = rng.sample(words, n)
        q, expr = ws[0], "P['%s']" % ws[0]
        # left-assoc chain with NOT>AND>OR precedence, mirrored in Python
        terms, ops = [ws[0]], []
        for w in ws[1:]:
            ops.append(rng.choice(["AND", "OR", "AND NOT"]))
            terms.append(w)
        q = terms[0] + "".join(" %s %s" % (o, t) for o, t in zip(ops, terms[1:]))
        code, wide, tids, prog = N.compile_wide(q, words)
        assert (code, wide, len(tids)) == (0, True, n), q[:60]
        bit = {("w%d" % t): i for i, t in enumerate(tids)}
        for _ in range(8):
            P = {w: rng.random() < 0.5 for w in ws}
            # AND / AND NOT bind tighter than OR; all left-associative
            groups, cur = [], P[terms[0]]
            for o, t in zip(ops, terms[1:]):
                if o == "OR":
                    groups.append(cur)
                    cur = P[t]
                elif o == "AND":
                    cur = cur and P[t]
                else:
                    cur = cur and not P[t]
            groups.append(cur)
            assert _eval_wide(prog, {bit[w] for w in ws if P[w]}) == any(groups), q[:60]
    # 1025 live tokens: beyond the wide plan too => NXS_ERR_LIMIT (documented)
    code, wide, _, _ = N.compile_wide("(" + " OR ".join(words[:600]) + ") AND (" + " OR ".join(words[600:1025]) + ")", words)
    assert code == 6
    code, wide, tids, _ = N.compile_wide("(" + " OR ".join(words[:600]) + ") AND (" + " OR ".join(words[600:1024]) + ")", words)
    assert code == 6          # nesting limit (100 levels) is hit first by a 600-term chain
    bal = lambda ws: ws[0] if len(ws) == 1 else "(%s OR %s)" % (bal(ws[:len(ws) // 2]), bal(ws[len(ws) // 2:]))
    code, wide, tids, _ = N.compile_wide(bal(words[:1024]), words)
    assert (code, wide, len(tids)) == (0, True, 1024)
    code, _, _, _ = N.compile_wide(bal(words[:1025]), words)
    assert code == 6


def test_plan_truth_table_matches_set_algebra_on_random_queries():
    rng = random.Random(5)
    words = list("abcdefgh")
    for _ in range(300):
        n = rng.randint(1, 8)
        toks = [rng.choice(words) for _ in range(n)]
        q = toks[0]
        for t in toks[1:]:
            q += rng.choice([" AND ", " OR ", " AND NOT ", " "]) + t
        if rng.random() < 0.3 and n >= 3:
            q = "(" + q.replace(" ", " ", 1) + ")"
            if "  " in q or re.search(r"\w \w", q):
                q = q[1:-1]                    # juxtaposition is top-level only
        code, err, empty, p = N.compile_query(q, words)
        assert code == 0, (q, err)
        nt = p.n_tokens
        assert 1 <= nt <= 8
        for m in range(1 << nt):
            assert ((p.truth[m >> 5] >> (m & 31)) & 1) == _eval_prog(p, m)
        assert (p.truth[0] & 1) == 0            # no term present => no match


def test_host_levdist_matches_oracle(golden):
    for a, b, exp in golden["levdist"]["pairs"]:
        assert N.levdist(a, b) == exp
    rng = random.Random(3)
    for _ in range(20000):
        a = bytes(rng.choice(b"abcz\xc4") for _ in range(rng.randint(0, 20)))
        b = bytes(rng.choice(b"abcz\xc4") for _ in range(rng.randint(0, 20)))
        if 0 in a or 0 in b:
            continue
        assert N.levdist(a, b) == O.levdist(a, b)
    for _ in range(60):          # around and beyond the 64-byte bit-vector width
        a = bytes(rng.choice(b"ab") for _ in range(rng.randint(55, 140)))
        b = bytes(rng.choice(b"ab") for _ in range(rng.randint(55, 140)))
        assert N.levdist(a, b) == O.levdist(a, b)


def _image_search(nodes, q, tol=2):
    """bktree_search over the flattened image, level-synchronous (what the
    device does); returns (matches in BFS order, visited)."""
    frontier, out, visited = [0], [], 0
    while frontier:
        nxt = []
        for i in frontier:
            nd = nodes[i]
            d = O.levdist(q, nd["term"])
            visited += 1
            if d <= tol:
                out.append(i)
            lo, hi = max(d - tol, 0), min(d + tol, 63)
            for slot in range(lo, hi):          # half-open (Q8)
                if (nd["bitmap"] >> slot) & 1:
                    below = bin(nd["bitmap"] & ((1 << slot) - 1)).count("1")
                    nxt.append(nd["first_child"] + below)
        frontier = nxt
    return out, visited


@pytest.mark.parametrize("seed,n,alphabet", [(1, 400, "abcd"), (2, 2500, "abcdefghijklmnopqrstuvwxyz")])
def test_bk_image_search_equals_oracle_bfs(seed, n, alphabet):
    rng = random.Random(seed)
    words = []
    while len(words) < n:
        w = "".join(rng.choice(alphabet) for _ in range(rng.randint(1, 10)))
        if rng.random() < 0.03 and words:
            w = rng.choice(words)
        words.append(w)
    nodes, depth = N.bk_image(words)
    orc = O.BKTree([w.encode() for w in words])
    assert len(nodes) == len(set(words))
    for _ in range(150):
        q = bytearray(rng.choice(words).encode())
        q[rng.randrange(len(q))] = ord(rng.choice(alphabet))
        res, nvis = orc.search(bytes(q), 2)
        got, gvis = _image_search(nodes, bytes(q), 2)
        # same visit count, same match SET and the same FIRST match (Q7);
        # BFS rank order == deque push order
        assert gvis == nvis
        assert [nodes[i]["term_id"] - 1 for i in got] == res


def test_synth_corpus_is_valid_and_deterministic(tmp_path):
    c1 = corpus.write_corpus(str(tmp_path / "a"), 3000, 500, seed=7, threads=1)
    c2 = corpus.write_corpus(str(tmp_path / "b"), 3000, 500, seed=7, threads=4)
    assert open(c1["dtmap"], "rb").read() == open(c2["dtmap"], "rb").read()
    assert open(c1["terms"], "rb").read() == open(c2["terms"], "rb").read()
    idx = O.Index(c1["terms"], c1["dtmap"])
    assert (idx.term_count, idx.dt_count, idx.doc_count) == (500, 3000, 3000)
    assert idx.token_count == c1["tokens"]
    assert sum(idx.df(t) for t in range(1, 501)) == c1["postings"]
    terms = corpus.term_strings(500, seed=7)
    assert len(set(terms)) == 500 and all(4 <= len(t) <= 12 for t in terms)
    for tid in (1, 2, 250, 500):
        assert idx.term(tid) == terms[tid - 1]
    assert idx.df(1) > idx.df(50) > idx.df(500) >= 0      # Zipf-ish
    # files are sized in 32 KiB steps (index.h:24)
    assert os.path.getsize(c1["dtmap"]) % 32768 == 0
    assert os.path.getsize(c1["terms"]) % 32768 == 0
    res = idx.search(terms[0].decode(), limit=10)
    assert len(res) == 10
    # sparse ids variant
    c3 = corpus.write_corpus(str(tmp_path / "c"), 500, 100, seed=1, sparse_ids=True)
    idx3 = O.Index(c3["terms"], c3["dtmap"])
    assert idx3.dt_count == 500
    ids = [d for d, _ in idx3.search(corpus.term_strings(100, 1)[0].decode(), limit=1000)]
    assert ids and max(ids) > 10 ** 6


def test_scan_kernels_own_exactly_their_prefetch_agprs(tmp_path):
    """The scan kernels keep posting windows in flight in accumulation registers
    that only their inline asm names (nxs_gpu_dev.h: bpair_*).  That is only
    sound while the compiler allocates no AGPR of its own in those kernels --
    under register pressure it would spill VGPRs into them.  The code object's
    metadata must show exactly the owned registers: 2 per window in flight."""
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    if not (os.path.exists(llvm + "/llvm-objdump") and os.path.exists(llvm + "/llvm-readelf")):
        pytest.skip("llvm binutils not present")
    src = open(os.path.join(ROOT, "nxsearch_amd", "csrc", "nxs_gpu_dev.h")).read()
    ring8 = int(re.search(r"#define\s+SCAN8_RING_MAX\s+(\d+)", src).group(1))
    ring8b = int(re.search(r"#define\s+SCAN8_RING_BIG\s+(\d+)", src).group(1))      # MODE_BIG (= 3) instantiations
    ringm = int(re.search(r"#define\s+SCANM_RING\s+(\d+)", src).group(1))
    ringr = int(re.search(r"#define\s+SCANR_RING\s+(\d+)", src).group(1))
    srcb = open(os.path.join(ROOT, "nxsearch_amd", "csrc", "nxs_gpu_scan_bit.hip")).read()
    ringb = int(re.search(r"#define\s+SB_RING\s+(\d+)", srcb).group(1))
    srcs = open(os.path.join(ROOT, "nxsearch_amd", "csrc", "nxs_gpu_scan_stripe.hip")).read()
    rings = int(re.search(r"#define\s+ST_RING\s+(\d+)", srcs).group(1))     # k_scans: one ring for all terms
    so = shutil.copy(N.LIB_PATH, str(tmp_path / "lib.so"))
    subprocess.run([llvm + "/llvm-objdump", "--offloading", so], check=True, capture_output=True)
    co = [f for f in os.listdir(str(tmp_path)) if "gfx950" in f]
    assert len(co) >= 1, co       # one code object per translation unit
    notes = "".join(subprocess.run([llvm + "/llvm-readelf", "--notes", str(tmp_path / f)],
                                   check=True, capture_output=True, text=True).stdout for f in sorted(co))
    seen = 0
    for m in re.finditer(r"\.agpr_count:\s+(\d+)(?:(?!\.agpr_count:).)*?\.name:\s+(\S+)", notes, re.S):
        agpr, name = int(m.group(1)), m.group(2)
        k8 = re.match(r"_Z7k_scan8ILi(\d+)ELi(\d+)ELi(\d+)EE", name)
        kr = re.match(r"_Z7k_scanrILi\d+ELi(\d+)ELb[01]EE", name)
        km = re.match(r"_Z7k_scanmILi(\d+)ELb[01]ELb[01]EE", name)
        kb = re.match(r"_Z7k_scanbILi(\d+)ELb[01]ELb[01]EE", name)
        ks = re.match(r"_Z7k_scansILi(\d+)ELb[01]ELb[01]EE", name)
        if ks:
            want = 2 * rings
        elif k8:
            mode, nt, mm = int(k8.group(1)), int(k8.group(2)), int(k8.group(3))
            want = 2 * (ring8b if mode == 3 else ring8) * nt if (nt >= 3 and mm != 2) else 0
        elif kr:
            want = 2 * ringr * int(kr.group(1))
        elif km:
            want = 2 * ringm * int(km.group(1))
        elif kb:
            want = 2 * ringb * int(kb.group(1))
        else:
            continue
        seen += 1
        assert agpr == want, (name, agpr, want)
    assert seen >= 20, seen

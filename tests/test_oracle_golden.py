"""Pins the CPU oracle (oracle/nxs_oracle.c) against every golden vector the
reference's own tests hold for the hot path."""
import struct

import pytest

import nxsfmt
import oracle_lib as O

TK = {1: "AND", 2: "OR", 3: "NOT", 4: "(", 5: ")", 6: "FF", 7: "QUOTED"}


def f32bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def make_index(tmp_path, docs, **kw):
    items = [(int(k), v.split()) for k, v in docs.items()]
    t, d, ids = nxsfmt.write_index(str(tmp_path), "idx", items)
    return O.Index(t, d, **kw), ids


def test_levdist_known_answers(golden):
    for a, b, exp in golden["levdist"]["pairs"]:
        assert O.levdist(a.encode(), b.encode()) == exp, (a, b)
        assert O.levdist(b.encode(), a.encode()) == exp, (b, a)


def test_bktree_known_answers(golden):
    g = golden["bktree"]
    t = O.BKTree([w.encode() for w in g["words"]])
    for i, q in enumerate(g["search"]):
        res, _ = t.search(q.encode(), 2)
        assert res and res[-1] == i, (q, res)   # deque_pop_back == the word


def test_bktree_half_open_child_range(golden):
    g = golden["bktree_halfopen"]
    words = [w.encode() for w in g["words"]]
    t = O.BKTree(words)
    res, _ = t.search(g["query"].encode(), 2)
    assert [words[i].decode() for i in res] == g["matches"]
    # the distance itself is within tolerance: it is the range that prunes it
    assert O.levdist(b"abxd", b"abxyz") == 2


def test_heap_tie_order(golden):
    for c in golden["heap_ties"]["cases"]:
        ids = sorted((int(k) for k in c["scores"]), reverse=True)
        sc = [float(c["scores"][str(i)]) for i in ids]
        got = [d for d, _ in O.topk(ids, sc, c["k"])]
        assert got == c["order"], c


def test_query_lexer_and_parser(golden):
    for c in golden["queryparser"]["cases"]:
        kinds = [TK[k] for k in O.query_lex(c["query"])]
        assert kinds == c["tokens"], c["query"]
        rep, err = O.query_repr(c["query"])
        if c["repr"] is None:
            assert rep is None and err and err.startswith("syntax error near")
        else:
            assert rep == c["repr"], c["query"]


def test_query_parser_extra_properties():
    # keywords are case-insensitive, & and | are aliases (scan.re:64-66)
    assert O.query_repr("a & b | c")[0] == "(OR (AND `a` `b`) `c`)"
    assert O.query_repr("a AnD nOt b")[0] == "(NOT `a` `b`)"
    # longest match: a keyword prefix of a longer word is a plain string
    assert O.query_repr("android ORacle NOTe")[0] == "(OR (OR `android` `ORacle`) `NOTe`)"
    # juxtaposition is OR at the top level only (grammar.y:81-84)
    assert O.query_repr("a b AND c")[0] == "(OR `a` (AND `b` `c`))"
    assert O.query_repr("(a b)")[0] is None
    # left associativity; AND NOT binds like AND
    assert O.query_repr("a AND b AND NOT c AND d")[0] == "(AND (NOT (AND `a` `b`) `c`) `d`)"
    assert O.query_repr("a OR b AND NOT c")[0] == "(OR `a` (NOT `b` `c`))"
    # a quoted string immediately followed by text is one free-form string
    assert O.query_repr("\"ab\"cd")[0] == "`\"ab\"cd`"
    # empty query and a dangling NOT are syntax errors (t_index.c:115-117)
    rep, err = O.query_repr("")
    assert rep is None and err == 'syntax error near 1:0: " ..."'
    assert O.query_repr("NOT a")[0] is None
    assert O.query_repr("a NOT b")[0] is None
    rep, err = O.query_repr("a AND")
    assert err == 'syntax error near 1:5: " ..."'


def test_scoring_known_answers(golden, tmp_path):
    g = golden["scoring"]
    tol = g["tolerance"]
    for n, c in enumerate(g["cases"]):
        idx, _ = make_index(tmp_path / str(n), c["docs"])
        for algo in (O.TF_IDF, O.BM25):
            res = dict(idx.search(c["query"], algo=algo))
            assert set(res) == {int(k) for k in c["scores"]}, (c["query"], res)
            for doc, exp in c["scores"].items():
                assert abs(res[int(doc)] - exp[algo]) < tol, (c["query"], doc, algo, res)
        idx.close()


def stem_docs(docs):
    """What the reference's indexer makes of a raw text with its default filters
    (normalizer, stopwords -- no stop-word file in the tests' basedir --, stemmer) on
    these ASCII, punctuation-free texts: split, lower-case, stem (the ORACLE's stemmer)."""
    import ctypes as C
    out = {}
    for k, text in docs.items():
        toks = []
        for w in text.lower().split():
            b = w.encode()
            buf = C.create_string_buffer(len(b) + 2)
            n = O.lib().orc_stem_en(b, len(b), buf, len(b) + 2)
            toks.append(buf.raw[:n].decode())
        out[k] = " ".join(toks)
    return out


def test_scoring_known_answers_from_raw_text(golden, tmp_path):
    """t_scoring.c:16-163 with the reference's RAW documents and queries (`foxes` must meet `fox`,
    the query `cats` the term `cat`): the oracle's stemmer on both sides."""
    g = golden["scoring_raw"]
    for n, c in enumerate(g["cases"]):
        idx, _ = make_index(tmp_path / str(n), stem_docs(c["docs"]), lowercase=True, stemmer=True)
        for algo in (O.TF_IDF, O.BM25):
            res = dict(idx.search(c["query"], algo=algo))
            assert set(res) == {int(k) for k in c["scores"]}, (c["query"], res)
            for doc, exp in c["scores"].items():
                assert abs(res[int(doc)] - exp[algo]) < g["tolerance"], (c["query"], doc, algo, res)
        idx.close()


def test_querylogic_known_answers(golden, tmp_path):
    g = golden["querylogic"]
    idx, _ = make_index(tmp_path, g["docs"], lowercase=True)
    for c in g["cases"]:
        for algo in (O.TF_IDF, O.BM25):
            got = sorted(d for d, _ in idx.search(c["query"], algo=algo))
            assert got == c["docs"], c["query"]


def test_toy_index_bit_patterns(golden, tmp_path):
    g = golden["toy"]
    items = [(int(k), v.split()) for k, v in g["docs"].items()]
    timg, dimg, ids = nxsfmt.build_images(items)
    # the writer reproduces the bytes the reference wrote for this corpus
    assert timg[:64].hex() == g["terms_hex"]
    assert dimg[:128].hex() == g["dtmap_hex"]
    idx, _ = make_index(tmp_path, g["docs"])
    assert (idx.doc_count, idx.token_count, idx.term_count) == (3, 8, 3)
    bm = idx.search(g["query"], algo=O.BM25)
    assert [(d, "0x%08x" % f32bits(s)) for d, s in bm] == [tuple(x) for x in g["bm25"]]
    tf = idx.search(g["query"], algo=O.TF_IDF)
    assert [(d, "0x%08x" % f32bits(s)) for d, s in tf] == [tuple(x) for x in g["tfidf"]]
    for tok, exp in g["fuzzy"].items():
        tid, _ = idx.fuzzy(tok.encode())
        assert (idx.term(tid).decode() if tid else None) == exp
    # query-level fuzzy fallback (tokenizer.c:177-180) and its switch
    assert [d for d, _ in idx.search("cot")] == [3, 1]
    assert idx.search("cot", fuzzymatch=False) == []
    assert idx.search("zzzzzzzz") == []


def test_resp_json(golden):
    g = golden["resp_json"]
    assert O.results_json([tuple(r) for r in g["results"]]) == g["json"]


def test_on_disk_format_golden_bytes(golden, tmp_path):
    t = golden["terms_db"]
    img = nxsfmt.terms_image([x.encode() for x in t["terms"]], t["totals"])
    assert img[:72].hex() == t["hex"]
    d = golden["dtmap_db"]
    blocks = [(b[0], b[1], [tuple(p) for p in b[2]]) for b in d["blocks"]]
    dimg = nxsfmt.dtmap_image(blocks, d["token_count"], d["doc_count"])
    assert dimg[:88].hex() == d["hex"]
    # ... and the oracle loader reads the reference's own bytes
    tp, dp = tmp_path / "nxsterms", tmp_path / "nxsdtmap"
    tp.write_bytes(bytes.fromhex(t["hex"]) + b"\0" * (32768 - 72))
    # term 3 ("term-3") is referenced by doc 1002 in the dtmap vector
    timg3 = nxsfmt.terms_image([b"some-term-1", b"another-term-2", b"term-3"], [1, 2, 1])
    dp.write_bytes(bytes.fromhex(d["hex"]) + b"\0" * (32768 - 88))
    idx = O.Index(str(tp), str(dp))
    # partial sync: doc 1002 references unknown term 3 => consumption stops
    assert (idx.term_count, idx.dt_count, idx.doc_count, idx.token_count) == (2, 1, 2, 4)
    assert idx.lookup(b"some-term-1") == 1 and idx.lookup(b"another-term-2") == 2
    idx.close()
    tp.write_bytes(timg3)
    idx = O.Index(str(tp), str(dp))
    assert (idx.term_count, idx.dt_count) == (3, 2)
    assert idx.df(1) == 1 and idx.df(2) == 1 and idx.df(3) == 1
    idx.close()


def test_search_errors_and_limits(golden, tmp_path):
    idx, _ = make_index(tmp_path, golden["toy"]["docs"])
    with pytest.raises(O.SearchError) as e:
        idx.search("cat", limit=0)
    assert e.value.code == 3 and e.value.msg == "invalid limit"
    with pytest.raises(O.SearchError) as e:
        idx.search("cat", limit=(1 << 32))
    assert e.value.code == 3
    with pytest.raises(O.SearchError) as e:
        idx.search("cat AND")
    assert e.value.code == 3 and e.value.msg.startswith("query failed with syntax error near")
    # nesting limit (search.c:70,126-131): a 102-deep left chain of ORs
    q = " OR ".join(["cat"] * 102)
    with pytest.raises(O.SearchError) as e:
        idx.search(q)
    assert e.value.code == 6
    assert [d for d, _ in idx.search(" OR ".join(["cat"] * 101))] == [3, 1]
    # count = min(limit, matched)  (Q13)
    assert len(idx.search("cat OR dog OR cow", limit=2)) == 2
    assert len(idx.search("cat OR dog OR cow", limit=10)) == 3


def test_deleted_docs_are_skipped(tmp_path):
    docs = [(1, "cat dog cow".split()), (2, "dog cow".split()), (3, "cat cat cat".split())]
    t, d, _ = nxsfmt.write_index(str(tmp_path), "idx", docs, removed=[1])
    idx = O.Index(t, d)
    assert (idx.dt_count, idx.doc_count, idx.token_count) == (2, 2, 5)
    assert [x for x, _ in idx.search("cat")] == [3]
    assert sorted(x for x, _ in idx.search("dog")) == [2]

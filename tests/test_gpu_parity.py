"""GPU parity tier (`-m gpu`): the HIP path, called through the C ABI
(include/nxs.h), against the CPU oracle on the same inputs.

Bar (BASELINE.json north_star): returned doc ids and their order bit-exact,
BM25 / TF-IDF float scores within 1e-5 relative -- in fact the tests demand
identical float bits, which the design guarantees (DESIGN.md "bit-exact
scores")."""
import os
import random
import shutil
import struct

import pytest

import nxsearch_amd as N
import nxsfmt
import oracle_lib as O
from nxsearch_amd import corpus

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5          # the stated tolerance; the observed error is 0 ulp


def bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


_ORACLE_MEMO = {}


def oracle_memo(test, oidx, q, **kw):
    """The oracle's answer, computed once per (test, query, arguments): the env variants of a
    parametrized test force different GPU paths over the SAME corpus and queries."""
    key = (test, q, tuple(sorted(kw.items())))
    if key not in _ORACLE_MEMO:
        _ORACLE_MEMO[key] = oidx.search(q, **kw)
    return _ORACLE_MEMO[key]


def oracle_many(c, jobs, workers=8):
    """The oracle's answers to `jobs` = [(query, algo, limit, fuzzymatch)] from tests/oracle_pool.py: a child
    program without a GPU context that loads the index once and forks `workers` processes."""
    import json
    import subprocess
    import sys
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump([list(j) for j in jobs], f)
    try:
        r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_pool.py"),
                            c["terms"], c["dtmap"], f.name, str(workers)], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        return [None if x is None else [(d, s) for d, s in x] for x in json.loads(r.stdout.strip().splitlines()[-1])]
    finally:
        os.unlink(f.name)


def assert_same(got, want, ctx=""):
    assert [d for d, _ in got] == [d for d, _ in want], ctx
    for (d, a), (_, b) in zip(got, want):
        assert abs(a - b) <= REL_TOL * abs(b), (ctx, d, a, b)
        assert bits(a) == bits(b), (ctx, d, a, b)


@pytest.fixture(scope="module")
def nxs(tmp_path_factory):
    h = N.Nxs(str(tmp_path_factory.mktemp("base")))
    yield h
    h.close()


def open_pair(nxs, tmp, docs, name="idx", lowercase=False, **kw):
    items = [(int(k), v.split()) for k, v in docs.items()] if isinstance(docs, dict) else docs
    t, d, ids = nxsfmt.write_index(str(tmp), name, items, **kw)
    return nxs.open_files(t, d, lowercase=lowercase), O.Index(t, d, lowercase=lowercase), ids


def test_toy_index_known_answers(golden, nxs, tmp_path):
    g = golden["toy"]
    gidx, oidx, _ = open_pair(nxs, tmp_path, g["docs"])
    bm = gidx.search("cat", algo="BM25")
    assert [(d, "0x%08x" % bits(s)) for d, s in bm] == [tuple(x) for x in g["bm25"]]
    tf = gidx.search("cat", algo="TF-IDF")
    assert [(d, "0x%08x" % bits(s)) for d, s in tf] == [tuple(x) for x in g["tfidf"]]
    # fuzzy fallback resolves `cot` -> `cat` on the device; `zzzzzzzz` -> nothing
    assert [d for d, _ in gidx.search("cot")] == [3, 1]
    assert gidx.search("cot", fuzzymatch=False) == []
    assert gidx.search("zzzzzzzz") == []
    # first BFS match with total > 0 wins (Q7): `dog` fuzzy-resolves to `cow`
    toks = ["cot", "zzzzzzzz", "dog", "co"]
    assert gidx.fuzzy(toks) == [1, 0, 3, 1] == [oidx.fuzzy(t.encode())[0] for t in toks]
    # response JSON through nxs_resp_tojson
    js = gidx.search("cat", json=True)
    assert js == O.results_json(oidx.search("cat"))
    gidx.close()


def test_reference_scoring_vectors(golden, nxs, tmp_path):
    g = golden["scoring"]
    for n, c in enumerate(g["cases"]):
        gidx, oidx, _ = open_pair(nxs, tmp_path / str(n), c["docs"])
        for algo, name in ((0, "TF-IDF"), (1, "BM25")):
            res = gidx.search(c["query"], algo=name)
            assert {d for d, _ in res} == {int(k) for k in c["scores"]}
            for d, s in res:
                assert abs(s - c["scores"][str(d)][algo]) < g["tolerance"]
            assert_same(res, oidx.search(c["query"], algo=algo), c["query"])
        gidx.close()


def test_default_filters_raw_text_scoring(golden, tmp_path):
    """N2 / boundary (b): an index as the reference creates it by DEFAULT -- params.db lists
    normalizer, stopwords, stemmer, lang "en" (nxs.c:87-89,263-276) -- opens by name, and
    t_scoring.c:16-163 runs with the reference's RAW texts: the documents are indexed as their stemmed
    token streams (the oracle's stemmer stands in for the indexer), the raw QUERY goes through the
    product's filter pipeline (`cats` -> `cat`, `Foxes` -> `fox`).  Scores to the reference's 1e-4,
    and bit-identical to the oracle with its own stemmer."""
    from test_oracle_golden import stem_docs
    g = golden["scoring_raw"]
    base = tmp_path / "b"
    base.mkdir()
    with N.Nxs(str(base)) as n2:
        for n, c in enumerate(g["cases"]):
            items = [(int(k), v.split()) for k, v in stem_docs(c["docs"]).items()]
            t, d, _ = nxsfmt.write_index(str(base), "raw%d" % n, items,
                                         filters=["normalizer", "stopwords", "stemmer"])
            gidx = n2.open_index("raw%d" % n)
            oidx = O.Index(t, d, lowercase=True, stemmer=True)
            for q in (c["query"], c["query"].upper(), c["query"].replace("fox", "Foxes").replace("dog", "dogs")):
                for algo, name in ((0, "TF-IDF"), (1, "BM25")):
                    res = gidx.search(q, algo=name)
                    assert {dd for dd, _ in res} == {int(k) for k in c["scores"]}, q
                    for dd, sc in res:
                        assert abs(sc - c["scores"][str(dd)][algo]) < g["tolerance"]
                    assert_same(res, oidx.search(q, algo=algo), q)
            gidx.close()


def test_reference_querylogic_vectors(golden, nxs, tmp_path):
    g = golden["querylogic"]
    gidx, oidx, _ = open_pair(nxs, tmp_path, g["docs"], lowercase=True)
    for c in g["cases"]:
        for algo in ("TF-IDF", "BM25"):
            assert sorted(d for d, _ in gidx.search(c["query"], algo=algo)) == c["docs"]
        assert_same(gidx.search(c["query"]), oidx.search(c["query"]), c["query"])
    gidx.close()


def test_resp_json_vector(golden, nxs, tmp_path):
    # two docs whose scores are exactly 3.0 / 1.5 cannot be forced through
    # BM25; the formatter itself is pinned on the host side instead
    gidx, oidx, _ = open_pair(nxs, tmp_path, {"1": "a", "2": "a a a"})
    assert gidx.search("a", json=True) == O.results_json(oidx.search("a"))
    gidx.close()


def test_errors_match_reference_conventions(golden, nxs, tmp_path):
    gidx, oidx, _ = open_pair(nxs, tmp_path, golden["toy"]["docs"])
    for kw, code, msg in (
            (dict(query="cat", limit=0), 3, "invalid limit"),
            (dict(query="cat", limit=1 << 32), 3, "invalid limit"),
            (dict(query="cat", algo="nope"), 3, "invalid algorithm"),
            (dict(query="cat AND"), 3, 'query failed with syntax error near 1:7: " ..."'),
            (dict(query=""), 3, 'query failed with syntax error near 1:0: " ..."'),
            (dict(query=" OR ".join(["cat"] * 102)), 6, "query nesting limit reached (100 levels)")):
        with pytest.raises(N.NxsError) as e:
            gidx.search(**kw)
        assert (e.value.code, e.value.msg) == (code, msg)
    # after a failure the next good call clears the slot (search.c:295)
    assert [d for d, _ in gidx.search("cat")] == [3, 1]
    assert gidx.nxs.error()[0] == 0
    assert len(gidx.search("cat OR dog OR cow", limit=2)) == 2       # Q13
    # the Lua binding's way in: nxs_params_fromjson (params.c:201-208, lua.c:99-110)
    assert gidx.search("cat OR dog OR cow", params_json='{"limit": 2, "algo": "TF-IDF", "fuzzymatch": false}') == \
        gidx.search("cat OR dog OR cow", limit=2, algo="TF-IDF", fuzzymatch=False)
    with pytest.raises(N.NxsError) as e:
        gidx.search("cat", params_json='{"limit": 0}')
    assert e.value.code == 3 and "invalid limit" in e.value.msg
    with pytest.raises(N.NxsError) as e:
        gidx.search("cat", params_json='{"limit": 2')
    assert e.value.code == 2 and e.value.msg.startswith("params parsing failed")
    gidx.close()


def random_corpus(rng, n_docs, vocab, max_len=12, sparse=False):
    docs, did = [], 0
    for _ in range(n_docs):
        did += rng.randint(1, 1000) if sparse else 1
        n = rng.randint(1, max_len)
        docs.append((did, [rng.choice(vocab) for _ in range(n)]))
    return docs


def random_query(rng, vocab, nmax=5):
    n = rng.randint(1, nmax)
    q = rng.choice(vocab)
    for _ in range(n - 1):
        q += rng.choice([" AND ", " OR ", " AND NOT ", " "]) + rng.choice(vocab)
    return q


@pytest.mark.parametrize("seed,n_docs,vocab_n,sparse", [
    (1, 60, 8, False),        # tiny, massive score ties
    (2, 3000, 40, False),     # several tiles, few distinct scores
    (3, 5000, 300, True),     # sparse u64 doc ids, ragged lists
    (4, 20000, 30, False),    # dense lists: many batches per tile
])
@pytest.mark.parametrize("bm", [False, True])
def test_random_corpora_all_limits(nxs, tmp_path, monkeypatch, seed, n_docs, vocab_n, sparse, bm):
    if bm:      # every conjunction through the block-presence bitmaps (k_scanq), every term with a bitmap
        monkeypatch.setenv("NXS_GPU_BM_GAIN", "0")
        monkeypatch.setenv("NXS_GPU_BM_SHARE", "1073741824")
    rng = random.Random(seed)
    vocab = ["w%d" % i for i in range(vocab_n)]
    weights = [1.0 / (i + 1) for i in range(vocab_n)]
    pool = rng.choices(vocab, weights, k=4096)
    docs = random_corpus(rng, n_docs, pool, sparse=sparse)
    gidx, oidx, _ = open_pair(nxs, tmp_path, docs)
    queries = [random_query(rng, vocab[:max(4, vocab_n // 3)]) for _ in range(60)]
    queries += [vocab[0], vocab[0] + " AND " + vocab[1], vocab[-1] + " OR " + vocab[0]]
    for limit in (1, 3, 10, 64, 65, 1000):
        for algo, name in ((1, "BM25"), (0, "TF-IDF")):
            want = [oidx.search(q, algo=algo, limit=limit, fuzzymatch=False) for q in queries]
            got = gidx.search_batch(queries, limit=limit, algo=name, fuzzymatch=False)
            for q, g, w in zip(queries, got, want):
                assert_same(g, w, (q, limit, name))
    # single-query entry point, default limit (1000)
    for q in queries[:10]:
        assert_same(gidx.search(q, fuzzymatch=False), oidx.search(q, fuzzymatch=False), q)
    gidx.close()


def test_many_tokens_use_the_wide_path(nxs, tmp_path):
    rng = random.Random(9)
    vocab = ["t%d" % i for i in range(30)]
    docs = random_corpus(rng, 2000, vocab, max_len=10)
    gidx, oidx, _ = open_pair(nxs, tmp_path, docs)
    qs = [" OR ".join(vocab[:12]),
          "(" + " AND ".join(vocab[:3]) + ") OR (" + " AND ".join(vocab[3:12]) + ")",
          " OR ".join(vocab[:20]) + " AND NOT " + vocab[25]]
    for q in qs:
        for limit in (10, 100):
            assert_same(gidx.search(q, limit=limit, fuzzymatch=False),
                        oidx.search(q, limit=limit, fuzzymatch=False), q)
    gidx.close()


def test_queries_beyond_the_fixed_size_plan(nxs, tmp_path):
    """run_query_logic (search.c:210-278) has no bound on the number of terms:
    33 / 40 / 100 / 300 distinct tokens, a program of more than 256 items and an
    evaluation stack deeper than 64 take the wide plan (k_scanw, exact path)."""
    rng = random.Random(19)
    vocab = ["t%d" % i for i in range(320)]
    weights = [1.0 / (i + 1) for i in range(len(vocab))]
    pool = rng.choices(vocab, weights, k=8192)
    docs = random_corpus(rng, 6000, pool, max_len=14, sparse=True)
    gidx, oidx, _ = open_pair(nxs, tmp_path, docs)
    qs = [" OR ".join(vocab[:33]), " OR ".join(vocab[:40]), " ".join(vocab[:100]),
          # 300 tokens, balanced so that the nesting limit (100) holds
          " OR ".join("(" + " OR ".join(vocab[g * 75:(g + 1) * 75]) + ")" for g in range(4)),
          " OR ".join(vocab[:150]),                    # chain of height 149: NXS_ERR_LIMIT as in the reference
          "(" + " OR ".join(vocab[:50]) + ") AND (" + " OR ".join(vocab[40:95]) + ") AND NOT " + vocab[3],
          # 140 leaves of 20 distinct tokens: > 256 program items, <= 32 tokens
          " OR ".join("(%s AND %s)" % (vocab[i % 20], vocab[(i * 7 + 3) % 20]) for i in range(70)),
          # right-nested: evaluation stack of 70
          "".join("%s AND (" % vocab[i] for i in range(69)) + vocab[69] + ")" * 69,
          " OR ".join(vocab[:35]) + " OR nosuchterm"]
    for limit in (10, 64, 1000):
        for algo, name in ((1, "BM25"), (0, "TF-IDF")):
            got = gidx.search_batch(qs + [vocab[0] + " AND " + vocab[1]], limit=limit, algo=name,
                                    fuzzymatch=False)
            for q, g in zip(qs + [vocab[0] + " AND " + vocab[1]], got):
                try:
                    want = oidx.search(q, algo=algo, limit=limit, fuzzymatch=False)
                except O.SearchError as e:
                    assert isinstance(g, N.NxsError) and g.code == e.code == 6, q[:40]
                    continue
                assert_same(g, want, (q[:40], limit, name))
    assert_same(gidx.search(qs[1], fuzzymatch=False), oidx.search(qs[1], fuzzymatch=False))
    gidx.close()


def test_two_strings_resolving_to_one_term_score_twice(nxs, golden, tmp_path):
    """Q6: different token strings that fuzzy-resolve to the SAME term stay two
    tokens (tokenizer.c:100-107 merges identical strings only): the term's score
    is added twice.  Exercises duplicate term ids in two slots of k_scanm /
    k_scanr / k_scan8."""
    rng = random.Random(23)
    vocab = ["linux", "unix", "erlang", "python", "kernel", "shell", "driver", "thread"]
    docs = random_corpus(rng, 5000, vocab + ["pad%d" % i for i in range(30)], max_len=9)
    gidx, oidx, _ = open_pair(nxs, tmp_path, docs)
    qs = ["linus OR linuz", "linus AND linuz", "linus OR linuz OR unix", "linus AND linuz AND kernel",
          "linus linuz linvx", "(linus OR erlang) AND linuz", "linux OR linus", "linux AND linus AND NOT shell",
          "pythan OR pythom OR pithon OR python OR kernel"]
    assert gidx.fuzzy(["linus", "linuz"]) == [oidx.fuzzy(b"linus")[0], oidx.fuzzy(b"linuz")[0]]
    assert len(set(gidx.fuzzy(["linus", "linuz", "linvx"]))) == 1
    for limit in (3, 10, 64, 1000):
        for algo, name in ((1, "BM25"), (0, "TF-IDF")):
            got = gidx.search_batch(qs, limit=limit, algo=name)
            for q, g in zip(qs, got):
                assert_same(g, oidx.search(q, algo=algo, limit=limit), (q, limit, name))
    gidx.close()


def test_candidate_overflow_falls_back_to_exact_path(nxs, tmp_path, monkeypatch):
    # ascending scores with doc id = the adversarial order for the candidate
    # filter: every doc beats the running threshold when fed descending...
    # feed order is DESCENDING doc id, so make scores grow as ids shrink.
    docs = [(i + 1, ["x"] * (1 + (3000 - i) // 40) + ["pad"] * 3) for i in range(3000)]
    gidx, oidx, _ = open_pair(nxs, tmp_path, docs)
    monkeypatch.setenv("NXS_GPU_SEGCAP", "16")
    monkeypatch.setenv("NXS_GPU_WAVES", "2")
    gidx.reconfigure()      # the switches are parsed once, at open
    for limit in (5, 10):
        assert_same(gidx.search("x", limit=limit), oidx.search("x", limit=limit), limit)
    gidx.close()


def test_deleted_and_tombstoned_docs(nxs, tmp_path):
    docs = [(1, "cat dog cow".split()), (2, "dog cow".split()), (3, "cat cat cat".split()),
            (4, "cow cat".split())]
    t, d, _ = nxsfmt.write_index(str(tmp_path), "idx", docs, removed=[1, 4])
    gidx, oidx = nxs.open_files(t, d), O.Index(t, d)
    for q in ("cat", "dog", "cow", "cat OR dog OR cow"):
        assert_same(gidx.search(q), oidx.search(q), q)
    gidx.close()


def test_partial_sync_when_a_term_is_missing(nxs, golden, tmp_path):
    t, dm = golden["terms_db"], golden["dtmap_db"]
    tp, dp = tmp_path / "nxsterms", tmp_path / "nxsdtmap"
    tp.write_bytes(bytes.fromhex(t["hex"]) + b"\0" * (32768 - 72))
    dp.write_bytes(bytes.fromhex(dm["hex"]) + b"\0" * (32768 - 88))
    gidx, oidx = nxs.open_files(str(tp), str(dp)), O.Index(str(tp), str(dp))
    assert oidx.dt_count == 1
    for q in ("some-term-1", "another-term-2", "term-3"):
        assert_same(gidx.search(q, fuzzymatch=False), oidx.search(q, fuzzymatch=False), q)
    gidx.close()


def test_index_open_by_name_reads_params_db(nxs, golden, tmp_path):
    base = tmp_path / "b2"
    items = [(int(k), v.split()) for k, v in golden["querylogic"]["docs"].items()]
    nxsfmt.write_index(str(base), "my-idx_1", items, algo="TF-IDF", filters=["normalizer"])
    t, d = base / "data" / "my-idx_1" / "nxsterms", base / "data" / "my-idx_1" / "nxsdtmap"
    with N.Nxs(str(base)) as n2:
        idx = n2.open_index("my-idx_1")
        oidx = O.Index(str(t), str(d), lowercase=True)
        # index default algo comes from params.db (nxs.c:404-410)
        assert_same(idx.search("Unix OR Linux"), oidx.search("Unix OR Linux", algo=O.TF_IDF))
        with pytest.raises(N.NxsError) as e:
            n2.open_index("my-idx_1")
        assert e.value.code == 4
        with pytest.raises(N.NxsError) as e:
            n2.open_index("nope")
        assert e.value.code == 5
        with pytest.raises(N.NxsError) as e:
            n2.open_index("bad/name")
        assert e.value.code == 3


def test_filter_pipeline_on_query_tokens(nxs, tmp_path):
    """N2: params.db filters { normalizer, stopwords }: non-ASCII query tokens go
    through ICU (NFKC_Casefold + diacritics), stop words are discarded (their
    leaf is the empty set, search.c:140)."""
    base = tmp_path / "b3"
    docs = [(1, ["azul", "henry", "viii"]), (2, ["azuolelis", "arbae", "azul"]), (3, ["fuglafjordur", "henry"]),
            (4, ["viii", "finance"])]
    nxsfmt.write_index(str(base), "n2", docs, filters=["normalizer", "stopwords"])
    sw = base / "filters" / "stopwords"
    sw.mkdir(parents=True)
    (sw / "en").write_text("the\nof\n")
    t, d = base / "data" / "n2" / "nxsterms", base / "data" / "n2" / "nxsdtmap"
    with N.Nxs(str(base)) as n2:
        idx = n2.open_index("n2")
        oidx = O.Index(str(t), str(d))
        for q, plain in (("AZÚL", "azul"), ("Henry AND Ⅷ", "henry AND viii"), ("ĄŽUOLĖLIS OR Árbæ", "azuolelis OR arbae"),
                         ("Fuglafjørður", "fuglafjordur"), ("THE OR azul", "nosuchterm OR azul"),
                         ("the AND azul", "nosuchterm AND azul"), ("ﬁnance AND NOT of", "finance AND NOT nosuchterm")):
            assert_same(idx.search(q, fuzzymatch=False), oidx.search(plain, fuzzymatch=False), q)
        with pytest.raises(N.NxsError) as e:
            idx.search(b"\xff\xfe AND azul")
        assert (e.value.code, e.value.msg) == (1, "query_prepare() failed")
        idx.close()


@pytest.mark.parametrize("seed,n_terms,alphabet", [(1, 500, "abcd"), (2, 20000, "abcdefghijklmnopqrstuvwxyz")])
def test_fuzzy_matches_oracle(nxs, tmp_path, seed, n_terms, alphabet):
    rng = random.Random(seed)
    words = set()
    while len(words) < n_terms:
        words.add("".join(rng.choice(alphabet) for _ in range(rng.randint(2, 11))))
    words = sorted(words)
    rng.shuffle(words)
    # one doc per 7 words; a few words get total 0 via deletion of their only doc
    docs = [(i + 1, words[i * 7:(i + 1) * 7]) for i in range((len(words) + 6) // 7)]
    t, d, _ = nxsfmt.write_index(str(tmp_path), "idx", docs, removed=[2, 5])
    gidx, oidx = nxs.open_files(t, d), O.Index(t, d)
    toks = []
    for _ in range(400):
        w = bytearray(rng.choice(words).encode())
        for _ in range(rng.randint(1, 3)):
            op = rng.random()
            pos = rng.randrange(len(w))
            if op < 0.5:
                w[pos] = ord(rng.choice(alphabet))
            elif op < 0.75 and len(w) > 1:
                del w[pos]
            else:
                w.insert(pos, ord(rng.choice(alphabet)))
        toks.append(bytes(w))
    toks += [b"q" * 70, b"ab" * 40, words[0].encode() + b"x" * 66]      # > 64 bytes: DP path
    got, vis = gidx.fuzzy(toks, want_visited=True)
    for tok, g, v in zip(toks, got, vis):
        want, wv = oidx.fuzzy(tok)
        assert (g, v) == (want, wv), tok
    # frontier queues far too small for the whole batch: the one-pass attempt
    # overflows and is repeated with fewer tokens at a time, same answers
    os.environ["NXS_GPU_FUZZY_ITEMS"] = str(3 * n_terms)
    try:
        gidx.reconfigure()
        got2, vis2 = gidx.fuzzy(toks, want_visited=True)
        assert gidx.fuzzy(toks) == got
    finally:
        del os.environ["NXS_GPU_FUZZY_ITEMS"]
        gidx.reconfigure()
    assert (got2, vis2) == (got, vis)
    # production form (no visit counts): the match-first search (screen all
    # (token, term) pairs, exact distance on the survivors, reachability of the
    # matches) for tokens of <= 64 bytes, the frontier search for the rest -- the
    # same winners
    def path_and_ids(tk):
        gidx.set_profiling(True); gidx.profile(reset=True)
        ids = gidx.fuzzy(tk)
        lv = gidx.profile(reset=True)["fuzzy_level"]; gidx.set_profiling(False)
        return ("match-first" if lv[0] > len(tk) else "frontier"), ids
    assert path_and_ids(toks[:400]) == ("match-first", got[:400])
    assert gidx.fuzzy(toks) == got                      # mixed batch: split
    assert gidx.fuzzy(toks[:1]) == got[:1] and gidx.fuzzy(toks[7:8]) == got[7:8]
    for env, val, path in (("NXS_GPU_FUZZY_BFS", "1", "frontier"),         # pruned frontier search
                           ("NXS_GPU_FUZZY_CAND", "1024", "frontier")):    # survivor queue overflows -> fallback
        os.environ[env] = val
        try:
            gidx.reconfigure()
            assert path_and_ids(toks[:400]) == (path, got[:400]), env
        finally:
            del os.environ[env]
            gidx.reconfigure()
    gidx.close()


def test_fuzzy_slot63_and_long_terms(nxs, tmp_path):
    """Terms at distance >= 63 from an ancestor hang in slot 63, which no search
    enters (the child range ends at min(d+2, 63), exclusive: bktree.c:150-156):
    they and everything below them are never returned.  Terms of 62..66 bytes can
    still be within 2 of a token of <= 64 bytes.  Both searches against the oracle."""
    rng = random.Random(11)
    short = set()
    while len(short) < 300:
        short.add("".join(rng.choice("abc") for _ in range(rng.randint(2, 11))))
    longw = set()
    while len(longw) < 60:
        longw.add("".join(rng.choice("xyz") for _ in range(rng.randint(60, 67))))
    words = sorted(short)[:5] + sorted(longw)[:3] + sorted((longw | short) - set(sorted(short)[:5]) - set(sorted(longw)[:3]),
                                                             key=lambda w: rng.random())
    docs = [(i + 1, words[i * 5:(i + 1) * 5]) for i in range((len(words) + 4) // 5)]
    t, d, _ = nxsfmt.write_index(str(tmp_path), "idx", docs)
    gidx, oidx = nxs.open_files(t, d), O.Index(t, d)
    toks = []
    for w in sorted(longw):
        for _ in range(3):
            b = bytearray(w.encode())
            for _ in range(rng.randint(1, 2)):
                pos = rng.randrange(len(b))
                if rng.random() < 0.5:
                    b[pos] = ord(rng.choice("xyzq"))
                else:
                    del b[pos]
            toks.append(bytes(b))
    toks += [w.encode()[:-1] for w in sorted(short)[:40]]
    want = [oidx.fuzzy(tk)[0] for tk in toks]
    assert 0 in want and any(want)                      # some unreachable, some found
    got, vis = gidx.fuzzy(toks, want_visited=True)      # frontier search, the reference's visit counts
    assert got == want and vis == [oidx.fuzzy(tk)[1] for tk in toks]
    assert gidx.fuzzy(toks) == want                     # match-first (<= 64 bytes) + frontier (the rest)
    shortq = [tk for tk in toks if len(tk) <= 64]
    gidx.set_profiling(True); gidx.profile(reset=True)
    assert gidx.fuzzy(shortq) == [w for tk, w in zip(toks, want) if len(tk) <= 64]
    assert gidx.profile(reset=True)["fuzzy_level"][0] > len(shortq)     # the match-first search ran
    gidx.set_profiling(False)
    gidx.close()


def test_batch_with_errors_and_fuzzy_tokens(nxs, golden, tmp_path):
    gidx, oidx, _ = open_pair(nxs, tmp_path, golden["querylogic"]["docs"], lowercase=True)
    qs = ["unix", "unix AND", "linus OR pithon", "zzzzzzzzzzz", "textbook AND NOT jave",
          "(erlang"]
    got = gidx.search_batch(qs)
    for q, g in zip(qs, got):
        try:
            want = oidx.search(q)
        except O.SearchError as e:
            assert isinstance(g, N.NxsError) and g.code == e.code
            continue
        assert_same(g, want, q)
    gidx.close()


def test_synthetic_corpus_medium(nxs, tmp_path):
    """100k docs / 5k terms written by the corpus tool; C2/C3-style queries."""
    c = corpus.write_corpus(str(tmp_path), 100_000, 5000, seed=11)
    terms = corpus.term_strings(5000, seed=11)
    gidx, oidx = nxs.open_files(c["terms"], c["dtmap"]), O.Index(c["terms"], c["dtmap"])
    qs = corpus.queries_single(terms, 24, seed=1, lo=1, hi=2000)
    qs += corpus.queries_bool5(terms, 40, seed=2, hi=300)
    for algo, name in ((1, "BM25"), (0, "TF-IDF")):
        got = gidx.search_batch(qs, limit=10, algo=name, fuzzymatch=False)
        for q, g in zip(qs, got):
            assert_same(g, oidx.search(q, algo=algo, limit=10, fuzzymatch=False), q)
    fq = corpus.queries_fuzzy(terms, 64, seed=3)
    ids = gidx.fuzzy(fq)
    for tok, g in zip(fq, ids):
        assert g == oidx.fuzzy(tok.encode())[0], tok
    gidx.close()


@pytest.mark.parametrize("env", [{}, {"NXS_GPU_OLDSCAN": "1"}, {"NXS_GPU_NOSCAN1": "1"},
                                 {"NXS_GPU_NOREQ": "1"}, {"NXS_GPU_WAVES": "64", "NXS_GPU_MINPOST": "1"},
                                 {"NXS_GPU_NOSCANR": "1"}, {"NXS_GPU_NOSCANR2": "1"},
                                 {"NXS_GPU_NOSCANR2": "1", "NXS_GPU_WAVES": "64", "NXS_GPU_MINPOST": "1"},
                                 {"NXS_GPU_NOSCANM": "1"}, {"NXS_GPU_SCANM_DENS": "1.0"},
                                 {"NXS_GPU_SCANM_DENS": "1.0", "NXS_GPU_WAVES": "64", "NXS_GPU_MINPOST": "1"},
                                 {"NXS_GPU_DROP_MINPOST": "1"}, {"NXS_GPU_NODROP": "1"},
                                 {"NXS_GPU_DROP_MINPOST": "1", "NXS_GPU_SCANM_DENS": "0.02"},
                                 {"NXS_GPU_DROP_MINPOST": "1", "NXS_GPU_WAVES": "64", "NXS_GPU_MINPOST": "1"},
                                 # the mask path's two kernels: presence bits (k_scanb: by default only the sparsest
                                 # queries) for everything up to 5 tokens / the byte map (k_scanm) for everything
                                 {"NXS_GPU_SCANB_DENS": "1.0"}, {"NXS_GPU_SCANB_DENS": "1.0", "NXS_GPU_SCANM_DENS": "1.0"},
                                 {"NXS_GPU_SCANB_DENS": "1.0", "NXS_GPU_SCANM_DENS": "1.0", "NXS_GPU_WAVES": "64", "NXS_GPU_MINPOST": "1"},
                                 {"NXS_GPU_NOSCANB": "1"},
                                 # the byte map on doc stripes (k_scans: the default where every term has a rank directory)
                                 # for every density / with many short ranges / off (k_scanm's register windows) / with
                                 # directories for every term
                                 {"NXS_GPU_SCANM_DENS": "1.0", "NXS_GPU_NOSCANB": "1"},
                                 {"NXS_GPU_SCANM_DENS": "1.0", "NXS_GPU_NOSCANB": "1", "NXS_GPU_WAVES": "64", "NXS_GPU_MINPOST": "1"},
                                 {"NXS_GPU_NOSCANS": "1"}, {"NXS_GPU_NOSCANS": "1", "NXS_GPU_SCANM_DENS": "1.0", "NXS_GPU_NOSCANB": "1"},
                                 {"NXS_GPU_BM_SHARE": "1073741824", "NXS_GPU_NOSCANB": "1"},
                                 # ... and as the sparse + dense class's second kernel (k_scans<.., DROP>, opt-in)
                                 {"NXS_GPU_SCANS_DROP": "1", "NXS_GPU_DROP_MINPOST": "1"},
                                 {"NXS_GPU_SCANS_DROP": "1", "NXS_GPU_DROP_MINPOST": "1", "NXS_GPU_WAVES": "64", "NXS_GPU_MINPOST": "1"},
                                 {"NXS_GPU_SCANS_DROP": "1", "NXS_GPU_DROP_MINPOST": "1", "NXS_GPU_SCANM_DENS": "0.02"},
                                 # single-token classes: every query's top range in a launch of its own
                                 {"NXS_GPU_SCAN1_SPLIT": "1", "NXS_GPU_WAVES": "4096", "NXS_GPU_MINPOST": "64"},
                                 # conjunctions through the block-presence bitmaps (k_scanq): whenever the required
                                 # terms have one / every term has one / many short ranges / never
                                 {"NXS_GPU_BM_GAIN": "0"}, {"NXS_GPU_BM_GAIN": "0", "NXS_GPU_BM_SHARE": "1073741824"},
                                 {"NXS_GPU_BM_GAIN": "0", "NXS_GPU_BM_SHARE": "1073741824", "NXS_GPU_WAVES": "64", "NXS_GPU_MINPOST": "1"},
                                 {"NXS_GPU_NOBLKMAP": "1"},
                                 # the sparse + dense class on k_cold + k_scanb<.., DROP>; its top ranges not sent ahead
                                 {"NXS_GPU_DROPB": "1", "NXS_GPU_DROP_MINPOST": "1"},
                                 {"NXS_GPU_DROPB": "1", "NXS_GPU_DROP_MINPOST": "1", "NXS_GPU_WAVES": "64", "NXS_GPU_MINPOST": "1"},
                                 {"NXS_GPU_DROP_MINPOST": "1", "NXS_GPU_DROP_SPLIT": "0"},
                                 {"NXS_GPU_DROP_MINPOST": "1", "NXS_GPU_DROP_NOEARLY": "1", "NXS_GPU_DROP_SPLIT": "2"}])
def test_every_scan_path_agrees_with_the_oracle(nxs, tmp_path, monkeypatch, env):
    """The tile path (k_scan8), the mask paths (k_scans, k_scanm, k_scanb), the generic
    kernel (k_scan), the single-token kernel and the skip logic are selected by
    query shape; force each of them over the same mixed workload."""
    for kk, v in env.items():
        monkeypatch.setenv(kk, v)
    c = corpus.write_corpus(str(tmp_path), 60_000, 3000, seed=21)
    terms = corpus.term_strings(3000, seed=21)
    gidx, oidx = nxs.open_files(c["terms"], c["dtmap"]), O.Index(c["terms"], c["dtmap"])
    rng = random.Random(4)
    T = lambda r: terms[r - 1].decode()
    qs = [T(rng.randint(1, 500)) for _ in range(8)]
    qs += corpus.queries_bool5(terms, 24, seed=6, hi=400)
    qs += ["%s AND %s" % (T(1), T(2)), "%s AND %s" % (T(3), T(700)), "%s AND %s" % (T(2800), T(1)),
           "%s AND (%s OR %s)" % (T(900), T(2), T(3)), "%s AND NOT %s" % (T(1), T(2)),
           "%s OR %s" % (T(2500), T(2900)), "(%s AND %s) OR %s" % (T(1), T(2), T(1500)),
           "%s AND %s AND %s" % (T(1), T(1200), T(3))]
    for limit in (10, 200):
        got = gidx.search_batch(qs, limit=limit, fuzzymatch=False)
        for q, g in zip(qs, got):
            assert_same(g, oracle_memo("paths", oidx, q, limit=limit, fuzzymatch=False), (env, q, limit))
    gidx.close()


@pytest.mark.parametrize("env", [{}, {"NXS_GPU_WAVES": "16"}, {"NXS_GPU_NOSCANR2": "1"}, {"NXS_GPU_NOSCANR": "1"},
                                 {"NXS_GPU_NOSCANM": "1"}, {"NXS_GPU_SCANM_DENS": "1.0"},
                                 {"NXS_GPU_SCANB_DENS": "1.0"}, {"NXS_GPU_SCANB_DENS": "1.0", "NXS_GPU_SCANM_DENS": "1.0"},
                                 {"NXS_GPU_NOSCANB": "1"}, {"NXS_GPU_NOSCANS": "1"},
                                 {"NXS_GPU_NOSCANS": "1", "NXS_GPU_NOSCANB": "1"},
                                 {"NXS_GPU_NOSCANB": "1", "NXS_GPU_BM_SHARE": "1073741824"},
                                 {"NXS_GPU_DROP_MINPOST": "1"}, {"NXS_GPU_DROP_MINPOST": "1", "NXS_GPU_SCANM_DENS": "0.005"},
                                 {"NXS_GPU_SCANS_DROP": "1", "NXS_GPU_DROP_MINPOST": "1"},
                                 {"NXS_GPU_SCANS_DROP": "1", "NXS_GPU_DROP_MINPOST": "1", "NXS_GPU_SCANM_DENS": "0.005"},
                                 {"NXS_GPU_BM_GAIN": "0"}, {"NXS_GPU_BM_GAIN": "0", "NXS_GPU_BM_SHARE": "1073741824"}])
def test_sparse_terms_on_a_larger_corpus(nxs, tmp_path, monkeypatch, env):
    """400k docs: queries whose terms are all sparse (few postings per tile, most
    tiles skipped or wiped), 3- and 7-token shapes, mixed operators."""
    for kk, v in env.items():
        monkeypatch.setenv(kk, v)
    c = corpus.write_corpus(str(tmp_path), 400_000, 20_000, seed=31)
    terms = corpus.term_strings(20_000, seed=31)
    gidx, oidx = nxs.open_files(c["terms"], c["dtmap"]), O.Index(c["terms"], c["dtmap"])
    qs = corpus.queries_bool5(terms, 24, seed=7, lo=200, hi=4000)
    qs += corpus.queries_bool5(terms, 12, seed=8, lo=2000, hi=20_000, k=3)
    qs += corpus.queries_bool5(terms, 12, seed=9, lo=30, hi=3000, k=7)
    T = lambda r: terms[r - 1].decode()
    qs += ["(%s OR %s) AND NOT %s" % (T(300), T(500), T(40)), "%s OR %s OR (%s AND %s)" % (T(900), T(901), T(50), T(60)),
           "(%s AND %s) OR (%s AND %s)" % (T(100), T(120), T(140), T(160))]
    for algo, name in ((1, "BM25"), (0, "TF-IDF")):
        got = gidx.search_batch(qs, limit=10, algo=name, fuzzymatch=False)
        for q, g in zip(qs, got):
            assert_same(g, oracle_memo("sparse", oidx, q, algo=algo, limit=10, fuzzymatch=False), (env, q))
    got = gidx.search_batch(qs[:12], limit=64, fuzzymatch=False)
    for q, g in zip(qs[:12], got):
        assert_same(g, oracle_memo("sparse", oidx, q, limit=64, fuzzymatch=False), (env, q, 64))
    gidx.close()


@pytest.mark.parametrize("seed,n_docs,vocab_n,max_len", [
    (11, 3000, 12, 6),       # tiny vocabulary: massive score ties, every term dense
    (12, 40000, 400, 10),    # Zipf vocabulary: sparse and dense terms in one query
    (13, 150000, 5000, 8),   # mostly sparse terms: wide tiles, few candidates
])
def test_mask_path_pure_or_queries(nxs, tmp_path, monkeypatch, seed, n_docs, vocab_n, max_len):
    """The mask path -- k_scanb (a presence bit per four docs, the windows' docs staged
    in LDS, candidates scored by lower-bound searches there; 2..5 tokens), k_scans (quantised
    score bounds in a byte per doc over doc stripes cut out of the lists by the rank directories)
    and k_scanm (the same bytes + exact scores from per-term register windows) -- forced for
    every pure-OR query of 2..8 tokens, whatever
    the density of its terms: identical ids, order and score bits.  (One corpus, one
    oracle; the index is opened once per kernel: the switches are read at open.)"""
    monkeypatch.setenv("NXS_GPU_SCANM_DENS", "1.0")
    monkeypatch.setenv("NXS_GPU_SCANB_DENS", "1.0")
    rng = random.Random(seed)
    vocab = ["w%d" % i for i in range(vocab_n)]
    weights = [1.0 / (i + 1) for i in range(vocab_n)]
    pool = rng.choices(vocab, weights, k=8192)
    docs = random_corpus(rng, n_docs, pool, max_len=max_len, sparse=(seed == 12))
    t, d, _ = nxsfmt.write_index(str(tmp_path), "idx", docs)
    oidx = O.Index(t, d)
    queries = []
    for _ in range(40):
        n = rng.randint(2, 8)          # k_scanm<3> (also for two tokens), <5> and <8>
        hi = rng.choice([min(vocab_n, 12), vocab_n // 2, vocab_n])
        queries.append(" OR ".join(rng.sample(vocab[:max(hi, n)], n)))
    queries += [" ".join(vocab[:4]), "%s OR %s OR %s" % (vocab[-1], vocab[-2], vocab[0])]
    want = {}
    monkeypatch.setenv("NXS_GPU_BM_SHARE", "1073741824")     # a rank directory for every term
    for kern in ("scanb", "scans", "scanm"):
        if kern != "scanb":
            monkeypatch.setenv("NXS_GPU_NOSCANB", "1")
        if kern == "scanm":
            monkeypatch.setenv("NXS_GPU_NOSCANS", "1")
        gidx = nxs.open_files(t, d)
        for limit in (1, 3, 10, 64):
            for algo, name in ((1, "BM25"), (0, "TF-IDF")):
                got = gidx.search_batch(queries, limit=limit, algo=name, fuzzymatch=False)
                for q, g in zip(queries, got):
                    key = (q, limit, algo)
                    if key not in want:
                        want[key] = oidx.search(q, algo=algo, limit=limit, fuzzymatch=False)
                    assert_same(g, want[key], (kern, q, limit, name))
        gidx.close()


@pytest.mark.parametrize("kern", ["scanb", "scans", "scanm"])
def test_mask_path_pending_overflow_falls_back(nxs, tmp_path, monkeypatch, kern):
    """The highest docs all hold every query term: the cold-start tile pushes
    more docs than the pending list takes, the query is flagged and re-run on
    the exact two-pass path."""
    monkeypatch.setenv("NXS_GPU_SCANM_DENS", "1.0")
    monkeypatch.setenv("NXS_GPU_SCANB_DENS", "1.0")
    monkeypatch.setenv("NXS_GPU_BM_SHARE", "1073741824")
    if kern != "scanb":
        monkeypatch.setenv("NXS_GPU_NOSCANB", "1")
    if kern == "scanm":
        monkeypatch.setenv("NXS_GPU_NOSCANS", "1")
    rng = random.Random(5)
    vocab = ["w%d" % i for i in range(50)]
    docs = random_corpus(rng, 4000, vocab, max_len=6)
    last = docs[-1][0]
    docs += [(last + 1 + i, ["w1", "w2", "w3", "w4", "w5"] * rng.randint(1, 3)) for i in range(200)]
    gidx, oidx, _ = open_pair(nxs, tmp_path, docs)
    qs = ["w1 OR w2 OR w3 OR w4 OR w5", "w1 OR w2 OR w3", "w5 OR w30 OR w2 OR w40"]
    for limit in (1, 10, 64):
        got = gidx.search_batch(qs, limit=limit, fuzzymatch=False)
        for q, g in zip(qs, got):
            assert_same(g, oidx.search(q, limit=limit, fuzzymatch=False), (q, limit))
    gidx.close()


def test_pipelined_device_batches(nxs, tmp_path):
    """nxsgpu_search_dev_begin/_end: batches in flight with their own outputs
    give what the blocking call gives; a fifth begin and a stray end are errors."""
    import torch
    c = corpus.write_corpus(str(tmp_path), 200_000, 8000, seed=41)
    terms = corpus.term_strings(8000, seed=41)
    gidx, oidx = nxs.open_files(c["terms"], c["dtmap"]), O.Index(c["terms"], c["dtmap"])
    dev = torch.device("cuda", 0)
    k = 10
    batches = [corpus.queries_bool5(terms, 48, seed=s_, hi=600) for s_ in (1, 2, 3)]
    plans = [gidx.plan_batch(b, limit=k, algo="BM25", fuzzymatch=False)[0] for b in batches]
    outs = [(torch.zeros((48, k), dtype=torch.int64, device=dev),
             torch.zeros((48, k), dtype=torch.float32, device=dev),
             torch.zeros((48,), dtype=torch.int32, device=dev)) for _ in batches]
    ptrs = lambda o: (o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
    with pytest.raises(N.NxsError):
        gidx.search_dev_end()                      # nothing in flight
    gidx.search_dev_begin(plans[0], 48, k, N.BM25, *ptrs(outs[0]))
    gidx.search_dev_begin(plans[1], 48, k, N.BM25, *ptrs(outs[1]))
    spare = [(torch.zeros((48, k), dtype=torch.int64, device=dev), torch.zeros((48, k), dtype=torch.float32, device=dev),
              torch.zeros((48,), dtype=torch.int32, device=dev)) for _ in range(3)]
    gidx.search_dev_begin(plans[0], 48, k, N.BM25, *ptrs(spare[0]))
    gidx.search_dev_begin(plans[1], 48, k, N.BM25, *ptrs(spare[1]))
    with pytest.raises(N.NxsError):
        gidx.search_dev_begin(plans[2], 48, k, N.BM25, *ptrs(spare[2]))  # NXSGPU_INFLIGHT = 4 is the limit
    with pytest.raises(N.NxsError):
        gidx.search_batch(batches[2], limit=k, fuzzymatch=False)         # blocking API refuses meanwhile
    assert gidx.search_dev_end() == 0 and gidx.search_dev_end() == 0     # batches 0 and 1 ...
    assert gidx.search_dev_end() == 0 and gidx.search_dev_end() == 0     # ... and their repeats
    assert torch.equal(spare[0][0], outs[0][0]) and torch.equal(spare[1][1], outs[1][1])
    gidx.search_dev_begin(plans[0], 48, k, N.BM25, *ptrs(outs[0]))
    gidx.search_dev_begin(plans[1], 48, k, N.BM25, *ptrs(outs[1]))
    assert gidx.search_dev_end() == 0              # batch 0
    gidx.search_dev_begin(plans[2], 48, k, N.BM25, *ptrs(outs[2]))
    assert gidx.search_dev_end() == 0              # batch 1
    assert gidx.search_dev_end() == 0              # batch 2
    for b, o in zip(batches, outs):
        ids, sc, cnt = (t.cpu() for t in o)
        for i, q in enumerate(b):
            want = oidx.search(q, limit=k, fuzzymatch=False)
            got = [(int(ids[i, j]), float(sc[i, j])) for j in range(int(cnt[i]))]
            assert_same(got, want, q)
    gidx.close()


def test_pipelined_string_batches(nxs, tmp_path):
    """nxs_index_search_batch_begin/_end: strings in, responses out, two batches
    in flight; per-step DISTINCT batches (a stale buffer would show)."""
    c = corpus.write_corpus(str(tmp_path), 200_000, 8000, seed=43)
    terms = corpus.term_strings(8000, seed=43)
    gidx, oidx = nxs.open_files(c["terms"], c["dtmap"]), O.Index(c["terms"], c["dtmap"])
    batches = [corpus.queries_bool5(terms, 40 + 7 * s_, seed=10 + s_, hi=600) for s_ in range(6)]
    batches[2] = batches[2][:5] + ["broken AND", "zzzzqqqq"] + batches[2][5:]
    with pytest.raises(N.NxsError):
        gidx.search_batch_end()                       # nothing in flight
    gidx.search_batch_begin(batches[0], limit=10, fuzzymatch=False)
    gidx.search_batch_begin(batches[1], limit=10, fuzzymatch=False)
    with pytest.raises(N.NxsError):
        gidx.search_batch(batches[2], limit=10, fuzzymatch=False)         # blocking call refuses meanwhile
    outs = []
    for i in range(2, len(batches)):
        outs.append(gidx.search_batch_end())
        gidx.search_batch_begin(batches[i], limit=10, fuzzymatch=False)
    outs.append(gidx.search_batch_end())
    outs.append(gidx.search_batch_end())
    for b, got in zip(batches, outs):
        assert len(got) == len(b)
        for q, g in zip(b, got):
            try:
                want = oidx.search(q, limit=10, fuzzymatch=False)
            except O.SearchError as e:
                assert isinstance(g, N.NxsError) and g.code == e.code
                continue
            assert_same(g, want, q)
    # limit > 64 through the same entry points (exact two-pass path at _end)
    gidx.search_batch_begin(batches[0][:9], limit=200, fuzzymatch=False)
    for q, g in zip(batches[0][:9], gidx.search_batch_end()):
        assert_same(g, oidx.search(q, limit=200, fuzzymatch=False), q)
    # NXS_BATCHES_INFLIGHT = 4 in flight, limits mixed (MODE_BIG batches replay on alternating streams);
    # a fifth is refused and leaves the four intact; results come back oldest first
    plan = [(batches[0], 10), (batches[1], 300), (batches[3], 1000), (batches[4], 10), (batches[5], 100), (batches[0], 1000)]
    for b, k in plan[:4]:
        gidx.search_batch_begin(b, limit=k, fuzzymatch=False)
    with pytest.raises(N.NxsError) as e:
        gidx.search_batch_begin(batches[5], limit=10, fuzzymatch=False)
    assert "already in flight" in e.value.msg
    gidx._pending = gidx._pending[:4]
    nxt = 4
    for b, k in plan:
        got = gidx.search_batch_end()
        if nxt < len(plan):
            gidx.search_batch_begin(plan[nxt][0], limit=plan[nxt][1], fuzzymatch=False)
            nxt += 1
        assert len(got) == len(b)
        for q, g in zip(b, got):
            assert_same(g, oracle_memo("pipe4", oidx, q, limit=k, fuzzymatch=False), (q, k))
    gidx.close()


def _mixed_workload(tmp_path, docs=200_000, n_terms=8000, n=192, seed=47):
    c = corpus.write_corpus(str(tmp_path), docs, n_terms, seed=seed)
    terms = corpus.term_strings(n_terms, seed=seed)
    qs = corpus.queries_mixed(terms, n, seed=5, hi=600)
    assert sum(1 for q in qs if any(w.encode() not in set(terms) for w in q.split() if w not in ("AND", "OR"))) > n // 8
    return c, terms, qs


def test_sharded_single_rank_through_rccl(nxs, tmp_path):
    """C5-style mixed BM25 + fuzzy batch through the sharded entry on ONE rank:
    a real RCCL communicator (ncclCommInitRank, world 1), the record path and
    the library's reassembly, against the oracle."""
    from nxsearch_amd import multi
    c, terms, qs = _mixed_workload(tmp_path)
    gidx, oidx = nxs.open_files(c["terms"], c["dtmap"]), O.Index(c["terms"], c["dtmap"])
    multi.attach(nxs, gidx, 0, 1)
    got = gidx.search_batch(qs, limit=10)
    for q, g in zip(qs, got):
        assert_same(g, oidx.search(q, limit=10), q)
    assert_same(gidx.search(qs[0], limit=10), oidx.search(qs[0], limit=10))    # never sharded
    # limits above 64 shard too (fixed-size records up to NXSGPU_BIG_K)
    for q, g in zip(qs[:24], gidx.search_batch(qs[:24], limit=300)):
        assert_same(g, oidx.search(q, limit=300), (q, 300))
    # a rank that cannot do its share still feeds the all-gather (STATUS_ABORT block):
    # the batch fails with this rank's own error, the next batch is in step again
    multi.inject_failure(gidx, "begin")
    with pytest.raises(N.NxsError) as e:
        gidx.search_batch(qs, limit=10)
    assert e.value.code == 2 and "injected failure" in e.value.msg
    for q, g in zip(qs, gidx.search_batch(qs, limit=10)):
        assert_same(g, oidx.search(q, limit=10), q)
    # the same with a healthy batch IN FLIGHT (the pipelined loop of a sharded server): the
    # failing _begin must not consume that batch's device slot -- its _end still returns its
    # own answers, and the batch after the failed one is in step
    gidx.search_batch_begin(qs[:100], limit=10)
    multi.inject_failure(gidx, "begin")
    with pytest.raises(N.NxsError) as e:
        gidx.search_batch_begin(qs[100:], limit=10)
    assert e.value.code == 2 and "injected failure" in e.value.msg
    for q, g in zip(qs[:100], gidx.search_batch_end()):
        assert_same(g, oidx.search(q, limit=10), q)
    for q, g in zip(qs[100:], gidx.search_batch(qs[100:], limit=10)):
        assert_same(g, oidx.search(q, limit=10), q)
    gidx.shard(0, 1, None)                            # detach
    got = gidx.search_batch(qs[:16], limit=10)
    for q, g in zip(qs, got):
        assert_same(g, oidx.search(q, limit=10), q)
    gidx.close()


def test_sharded_resync_in_a_pipelined_loop(nxs, tmp_path):
    """search.c:309-312 with a communicator attached (one rank, the same code path as N): a pipelined
    server that never drains still picks up what the indexer publishes.  The rank that notices the
    moved files says so in the flags word of its record block (NXSGPU_BLOCK_CHANGED); every rank reads
    all flags at the batch's _end and drains + refreshes at its NEXT _begin -- two batches after the
    change was noticed, at the same batch on every rank (tests/test_multi_gloo.py has the N = 2 form)."""
    from nxsearch_amd import multi
    ev = [("add", i + 1, ["cat", "dog", "w%d" % (i % 7)]) for i in range(300)]
    timg, dimg, _ = nxsfmt.build_images_log(ev)
    t, d = str(tmp_path / "nxsterms"), str(tmp_path / "nxsdtmap")
    open(t, "wb").write(timg + b"\0" * 262144)
    open(d, "wb").write(dimg + b"\0" * 262144)
    gidx = nxs.open_files(t, d)
    multi.attach(nxs, gidx, 0, 1)
    qs = ["cat", "emu", "dog OR emu", "w3 AND cat", "emu AND cat", "gnu"]
    import shutil
    oracles = []

    def snapshot():
        k = len(oracles)
        tt, dd = str(tmp_path / ("t%d" % k)), str(tmp_path / ("d%d" % k))
        shutil.copy(t, tt)
        shutil.copy(d, dd)
        oracles.append(O.Index(tt, dd))

    def publish(events):
        timg, dimg, _ = nxsfmt.build_images_log(events)
        nxsfmt.publish_in_place(t, d, timg, dimg)
        snapshot()

    snapshot()
    # published before _begin(p): noticed by _begin(p) (flag in batch p's block), read at _end(p)
    # -- which follows _begin(p + 1) --, so _begin(p + 2) drains, refreshes and sees it
    expect = {}
    gidx.search_batch_begin(qs, limit=10)
    expect[0] = 0
    for step in range(1, 12):
        if step == 2:
            ev.append(("add", 1000, ["cat", "emu"]))
            publish(ev)
        if step == 7:
            ev.append(("rm", 1000))
            ev.append(("add", 1001, ["gnu", "cat", "cat"]))
            publish(ev)
        gidx.search_batch_begin(qs, limit=10)
        expect[step] = 0 if step < 4 else 1 if step < 9 else 2
        got = gidx.search_batch_end()
        for q, g in zip(qs, got):
            assert_same(g, oracles[expect[step - 1]].search(q, limit=10), (step, q))
    got = gidx.search_batch_end()
    for q, g in zip(qs, got):
        assert_same(g, oracles[2].search(q, limit=10), ("last", q))
    # blocking calls (nothing in flight) re-sync at once, as before
    ev.append(("add", 1002, ["emu", "emu"]))
    publish(ev)
    for q, g in zip(qs, gidx.search_batch(qs, limit=10)):
        assert_same(g, oracles[3].search(q, limit=10), ("blocking", q))
    gidx.close()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_emulated_ranks_reassemble(nxs, tmp_path, monkeypatch, world):
    """Every rank of a W-rank run, one after the other on the one GPU, with the
    collective left out: each plans and scans ITS slice of the same mixed batch
    (errors, unresolvable and fuzzy tokens, a wide query, candidate overflow ->
    exact fix-up); the W record blocks go through the library's reassembly."""
    from nxsearch_amd import multi
    c, terms, qs = _mixed_workload(tmp_path, docs=120_000, n_terms=5000, n=61, seed=49)
    T = lambda r: terms[r - 1].decode()
    qs[3] = "cat AND"                                  # syntax error on its owner
    qs[11] = "zzzzzzzzzzzz OR qqqqqqqqqqqq"            # nothing resolves: empty
    qs[29] = " OR ".join(T(r) for r in range(1, 41))   # 40 tokens: wide plan
    qs[57] = T(1)                                      # dense single term
    monkeypatch.setenv("NXS_GPU_SEGCAP", "4")          # tiny segments: some queries overflow
    gidx, oidx = nxs.open_files(c["terms"], c["dtmap"]), O.Index(c["terms"], c["dtmap"])
    n, k = len(qs), 10
    cap = multi.shard_capacity(n, world)
    blocks = b""
    for r in range(world):
        multi.emulate(gidx, r, world)
        gidx.search_batch_begin(qs, limit=k)
        got = gidx.search_batch_end()
        assert all(isinstance(g, N.NxsError) for g in got)      # emulation hands out the block only
        blk = multi.emulated_block(gidx)
        assert len(blk) == multi.block_bytes(cap, k)
        blocks += blk
    multi.emulate(gidx, 0, 0)
    got = multi.assemble(blocks, world, cap, k, n)
    for q, g in zip(qs, got):
        try:
            want = oidx.search(q, limit=k)
        except O.SearchError as e:
            assert isinstance(g, N.NxsError) and g.code == e.code, q
            continue
        assert_same(g, want, (world, q[:50]))
    # nxs_index_shard_local: every rank materialises its own slice only -- together they give the batch
    for r in range(world):
        lo, hi = multi.shard_slice(n, r, world)
        part = multi.assemble(blocks, world, cap, k, n, only_rank=r)
        for i, g in enumerate(part):
            if lo <= i < hi:
                assert (isinstance(g, N.NxsError) and isinstance(got[i], N.NxsError) and g.code == got[i].code) or g == got[i]
            else:
                assert g is None
    assert gidx.shard_slice(n) == (0, n)                # nothing attached: the whole batch is this index's
    gidx.close()


def test_default_limit_batch_on_a_larger_corpus(nxs, tmp_path):
    """nxs_index_search(idx, NULL, ...) asks for 1000 results: the heap leaves the
    LDS for the two-pass global-memory path.  400k docs, batch + single calls."""
    c = corpus.write_corpus(str(tmp_path), 400_000, 20_000, seed=31)
    terms = corpus.term_strings(20_000, seed=31)
    gidx, oidx = nxs.open_files(c["terms"], c["dtmap"]), O.Index(c["terms"], c["dtmap"])
    qs = corpus.queries_bool5(terms, 20, seed=7, lo=1, hi=2000)
    qs += corpus.queries_single(terms, 6, seed=2, lo=1, hi=300)
    got = gidx.search_batch(qs, fuzzymatch=False)               # no limit given => 1000
    for q, g in zip(qs, got):
        want = oidx.search(q, fuzzymatch=False)
        assert len(want) <= 1000
        assert_same(g, want, q)
    assert any(len(g) == 1000 for g in got)
    for q in qs[:4]:
        assert_same(gidx.search(q), oidx.search(q), q)
    # the three homes of the replay's heap: across the lanes (limit <= 64), pairs in
    # LDS (65..8000), global memory beyond -- a limit of 20000 on queries with more
    # matches than that
    big = [q for q, g in zip(qs, got) if len(g) == 1000][:3]
    for limit in (64, 65, 8000, 8001, 20000):
        res = gidx.search_batch(big, limit=limit, fuzzymatch=False)
        for q, g in zip(big, res):
            assert_same(g, oidx.search(q, limit=limit, fuzzymatch=False), (q, limit))
    assert any(len(g) > 8001 for g in res)
    gidx.close()


@pytest.mark.parametrize("env", [{},
                                 # many short ranges: cold starts, thresholds handed down between ranges
                                 {"NXS_GPU_BIG_MINPOST": "0", "NXS_GPU_WAVES": "4096", "NXS_GPU_MINPOST": "1"},
                                 # candidate segments too small: the overflowed queries take the exact passes
                                 {"NXS_GPU_SEGCAP_BIG": "96", "NXS_GPU_BIG_MINPOST": "1"},
                                 {"NXS_GPU_NOSCANR": "1", "NXS_GPU_NOSCAN1": "1"}, {"NXS_GPU_OLDSCAN": "1"},
                                 {"NXS_GPU_ONEREPLAY": "1"},
                                 # every conjunctive query with bitmaps on k_scanq<.., BIG>, however many matches it expects
                                 # (more than a range's candidate list holds: the exact path), with many short ranges, and none
                                 {"NXS_GPU_BIGQ_EM": "1e18", "NXS_GPU_BM_GAIN": "0", "NXS_GPU_BM_SHARE": "1000000"},
                                 {"NXS_GPU_BIGQ_EM": "1e18", "NXS_GPU_BM_GAIN": "0", "NXS_GPU_BM_SHARE": "1000000",
                                  "NXS_GPU_BIG_MINPOST": "0", "NXS_GPU_WAVES": "4096", "NXS_GPU_MINPOST": "1"},
                                 {"NXS_GPU_BIGQ_EM": "0"}, {"NXS_GPU_AND_NOEARLY": "1"}])
def test_limits_above_64_ride_the_candidate_filter(nxs, tmp_path, monkeypatch, env):
    """64 < limit <= 8000 (the API's default is 1000, nxs_impl.h:39): the scan
    kernels filter on a histogram lower bound of the k-th best score (MODE_BIG) and
    the heap is replayed in LDS -- the same ids, order and float bits as the
    reference's heap (heap.c:58-221), on a tie-heavy corpus and a Zipf one, blocking
    and pipelined (two batches in flight, different limits)."""
    for kk, v in env.items():
        monkeypatch.setenv(kk, v)
    # (a) tiny vocabulary: massive score ties, thousands of matches per query
    vocab = ["t%d" % i for i in range(14)]
    rng = random.Random(77)
    docs = [(d + 1, [rng.choice(vocab) for _ in range(rng.randint(1, 6))]) for d in range(9000)]
    gidx, oidx, _ = open_pair(nxs, tmp_path, docs)
    qs = ["t0", "t1 OR t2", "t3 AND t4", "t5 OR t6 OR t7 OR t8 OR t9", "t1 AND NOT t2", "(t1 AND t2) OR t10",
          "t11 AND t12 AND t13", " OR ".join(vocab[:12])]
    for limit in (65, 100, 1000, 4000, 8000):
        for algo, name in ((1, "BM25"), (0, "TF-IDF")):
            got = gidx.search_batch(qs, limit=limit, algo=name, fuzzymatch=False)
            for q, g in zip(qs, got):
                assert_same(g, oracle_memo("big", oidx, q, algo=algo, limit=limit, fuzzymatch=False), (env, q, limit, name))
    gidx.search_batch_begin(qs, limit=300, fuzzymatch=False)
    gidx.search_batch_begin(qs[::-1], limit=10, fuzzymatch=False)
    for batch, limit in ((qs, 300), (qs[::-1], 10)):
        for q, g in zip(batch, gidx.search_batch_end()):
            assert_same(g, oracle_memo("big_a", oidx, q, limit=limit, fuzzymatch=False), (env, q, limit))
    gidx.close()
    # (b) Zipf corpus: five-term AND / OR shapes, single terms, sparse and dense lists
    sub = tmp_path / "z"
    sub.mkdir()
    c = corpus.write_corpus(str(sub), 200_000, 8000, seed=41)
    terms = corpus.term_strings(8000, seed=41)
    gidx, oidx = nxs.open_files(c["terms"], c["dtmap"]), O.Index(c["terms"], c["dtmap"])
    qs = corpus.queries_bool5(terms, 40, seed=5, lo=1, hi=800)
    qs += corpus.queries_single(terms, 8, seed=6, lo=1, hi=200)
    qs += corpus.queries_bool5(terms, 8, seed=7, lo=1, hi=60, k=3)
    for limit in (100, 1000):
        got = gidx.search_batch(qs, limit=limit, fuzzymatch=False)
        for q, g in zip(qs, got):
            assert_same(g, oracle_memo("big2", oidx, q, limit=limit, fuzzymatch=False), (env, q, limit))
    assert any(len(g) == 1000 for g in got)
    for q in qs[:3] + qs[40:42]:
        assert_same(gidx.search(q), oidx.search(q), q)             # params == NULL: limit 1000
    gidx.close()


def test_search_resyncs_appended_and_removed_docs(nxs, tmp_path):
    """search.c:309-312: every search first picks up what other processes
    appended.  The files are rewritten IN PLACE (same inode, MAP_SHARED), body
    first and the header's data_len last, like the reference's publish order."""
    docs = [(1, "cat dog cow".split()), (2, "dog cow".split())]
    t, d, _ = nxsfmt.write_index(str(tmp_path), "idx", docs)
    gidx = nxs.open_files(t, d)
    assert [x for x, _ in gidx.search("cat")] == [1]

    def publish(all_docs, removed=()):
        timg, dimg, _ = nxsfmt.build_images(all_docs, removed)
        for path, img, hdr in ((t, timg, 16), (d, dimg, 32)):
            with open(path, "r+b") as f:
                f.seek(hdr)
                f.write(img[hdr:])
                f.flush()
                f.seek(0)
                f.write(img[:hdr])
    docs2 = docs + [(3, "cat cat cat".split()), (7, "emu cat".split())]
    publish(docs2)
    oidx = O.Index(t, d)
    for q in ("cat", "emu", "dog OR emu"):
        assert_same(gidx.search(q), oidx.search(q), q)
    assert [x for x, _ in gidx.search("cat")][0] == 3
    publish(docs2, removed=[3])
    oidx = O.Index(t, d)
    for q in ("cat", "emu", "cow"):
        assert_same(gidx.search(q), oidx.search(q), q)
    gidx.close()


def test_resync_in_a_pipelined_loop_that_never_drains(nxs, tmp_path):
    """search.c:309-312 under pipelining: a server that keeps one batch in flight
    at all times (_begin(i+1) before _end(i), forever) still sees what an indexer
    appends and removes.  _begin notices the moved data_len words, finishes the
    batch in flight internally (its responses wait for the caller's _end),
    refreshes, and plans the new batch against the new snapshot."""
    ev = [("add", i + 1, ["cat", "dog", "w%d" % (i % 7)]) for i in range(300)]
    timg, dimg, _ = nxsfmt.build_images_log(ev)
    t, d = str(tmp_path / "nxsterms"), str(tmp_path / "nxsdtmap")
    open(t, "wb").write(timg + b"\0" * 262144)
    open(d, "wb").write(dimg + b"\0" * 262144)
    gidx = nxs.open_files(t, d)
    qs = ["cat", "emu", "dog OR emu", "w3 AND cat", "emu AND cat", "gnu"]
    import shutil

    def snapshot():                           # (a private copy: the oracle reads the header counters live)
        k = len(oracles)
        tt, dd = str(tmp_path / ("t%d" % k)), str(tmp_path / ("d%d" % k))
        shutil.copy(t, tt)
        shutil.copy(d, dd)
        return O.Index(tt, dd)
    oracles = []
    oracles.append(snapshot())                # snapshot 0
    seen_by = []                              # snapshot each batch was planned against

    def publish(events):
        timg, dimg, _ = nxsfmt.build_images_log(events)
        nxsfmt.publish_in_place(t, d, timg, dimg)
        oracles.append(snapshot())

    gidx.search_batch_begin(qs, limit=10)     # batch 0: snapshot 0
    seen_by.append(0)
    for step in range(1, 9):
        if step == 2:
            ev.append(("add", 1000, ["cat", "emu"]))
            publish(ev)                       # between _begin(1) and _begin(2), batch 1 in flight
        if step == 5:
            ev.append(("rm", 1000))
            ev.append(("add", 1001, ["gnu", "cat", "cat"]))
            publish(ev)
        gidx.search_batch_begin(qs, limit=10)             # never drains: batch step-1 is in flight
        seen_by.append(len(oracles) - 1)
        got = gidx.search_batch_end()                     # collects batch step-1
        want = oracles[seen_by[step - 1]]
        for q, g in zip(qs, got):
            assert_same(g, want.search(q, limit=10), (step, q))
    got = gidx.search_batch_end()
    for q, g in zip(qs, got):
        assert_same(g, oracles[seen_by[-1]].search(q, limit=10), ("last", q))
    # the change published at step 2 was seen by batch 2 (not later), the one at step 5 by batch 5
    assert seen_by[2] == 1 and seen_by[5] == 2
    assert [x for x, _ in gidx.search("emu")] == [] and [x for x, _ in gidx.search("gnu")] == [1001]
    gidx.close()


@pytest.mark.parametrize("late", [True, False])
def test_late_fuzzy_half_of_pipelined_batches(nxs, tmp_path, monkeypatch, late):
    """tokenizer.c:177-180 under pipelining: a batch's misses are ONE device pass that _begin leaves
    running (its second half -- winners, compile, queueing -- is done by the next _begin, after that
    batch's parse and the launch of ITS pass, or by the batch's own _end).  Mixed BM25 + misspelt
    tokens + syntax errors, 1-4 batches in flight, two limits; the caller's strings are gone when the
    second half runs (the binding frees them on return); NXS_LATE_FUZZY=0 = _begin waits itself."""
    from nxsearch_amd import multi
    if not late:
        monkeypatch.setenv("NXS_LATE_FUZZY", "0")
    c, terms, qs = _mixed_workload(tmp_path)
    qs = qs[:20] + ["broken AND", "zzzzqqqqzzzzqq"] + qs[20:]
    gidx, oidx = nxs.open_files(c["terms"], c["dtmap"]), O.Index(c["terms"], c["dtmap"])
    batches = [qs[i::6] for i in range(6)]

    def check(b, got, limit):
        assert len(got) == len(b)
        for q, g in zip(b, got):
            try:
                want = oracle_memo("latefz", oidx, q, limit=limit)
            except O.SearchError as e:
                assert isinstance(g, N.NxsError) and g.code == e.code, q
                continue
            assert_same(g, want, (q, limit))
    gidx.host_profile()
    for depth, limit in ((1, 10), (2, 10), (4, 10), (3, 100)):
        inflight = []
        for b in batches:
            gidx.search_batch_begin(b, limit=limit)
            inflight.append(b)
            if len(inflight) == depth:
                check(inflight[0], gidx.search_batch_end(), limit)
                inflight.pop(0)
        while inflight:
            check(inflight[0], gidx.search_batch_end(), limit)
            inflight.pop(0)
    hp = gidx.host_profile()
    assert (hp["fuzzy_launch_ms"] > 0.005) == late, hp       # (a pass queued per batch: tens of microseconds)
    # the blocking entry points between pipelined runs
    check(batches[0], gidx.search_batch(batches[0], limit=10), 10)
    assert_same(gidx.search(batches[1][0], limit=10), oracle_memo("latefz", oidx, batches[1][0], limit=10))
    # the plan cache took the late batches' plans too: the same strings again are not planned again
    if late:
        gidx.search_batch_begin(batches[2], limit=10)
        gidx.search_batch_begin(batches[3], limit=10)
        check(batches[2], gidx.search_batch_end(), 10)
        check(batches[3], gidx.search_batch_end(), 10)
        # a second half that fails: its _begin has returned success, so its _end reports it; the batches
        # around it are untouched
        fresh = [[q + " OR " + batches[(i + 1) % 6][j % len(batches[(i + 1) % 6])] for j, q in enumerate(batches[i])
                  if "broken" not in q] for i in range(3)]
        gidx.search_batch_begin(fresh[0], limit=10)
        multi.inject_failure(gidx, "late")
        gidx.search_batch_begin(fresh[1], limit=10)      # finishes fresh[0]'s second half: fails there
        gidx.search_batch_begin(fresh[2], limit=10)
        with pytest.raises(N.NxsError) as e:
            gidx.search_batch_end()
        assert e.value.code == 2 and "late half" in e.value.msg
        gidx._pending = gidx._pending[1:]
        check(fresh[1], gidx.search_batch_end(), 10)
        check(fresh[2], gidx.search_batch_end(), 10)
        # an index closed with a batch's pass still on the device
        gidx.search_batch_begin(fresh[0], limit=10)
    gidx.close()


def test_late_fuzzy_half_sees_a_resync(nxs, tmp_path):
    """A never-draining loop of batches WITH misses (each one's second half still open when the next
    _begin comes by) while an indexer appends a term that changes the fuzzy winners: the open half is
    finished against the OLD snapshot before the refresh, the new batch resolves against the new one
    (BK-tree image re-flattened with no pass in flight)."""
    import shutil
    ev = [("add", i + 1, ["carpet", "donkey", "w%d" % (i % 7)]) for i in range(300)]
    timg, dimg, _ = nxsfmt.build_images_log(ev)
    t, d = str(tmp_path / "nxsterms"), str(tmp_path / "nxsdtmap")
    open(t, "wb").write(timg + b"\0" * 262144)
    open(d, "wb").write(dimg + b"\0" * 262144)
    gidx = nxs.open_files(t, d)
    qs = ["carpit", "monkey", "donkeys OR monkey", "w3 AND carpets", "monkei AND carpet", "zebra"]
    oracles, seen_by = [], []

    def snapshot():
        k = len(oracles)
        tt, dd = str(tmp_path / ("t%d" % k)), str(tmp_path / ("d%d" % k))
        shutil.copy(t, tt)
        shutil.copy(d, dd)
        oracles.append(O.Index(tt, dd))
    snapshot()
    gidx.search_batch_begin(qs, limit=10)
    seen_by.append(0)
    for step in range(1, 8):
        if step == 2:
            ev.append(("add", 1000, ["carpet", "monkey"]))       # "monkey" now exists: "monkei" -> monkey, not donkey
            timg, dimg, _ = nxsfmt.build_images_log(ev)
            nxsfmt.publish_in_place(t, d, timg, dimg)
            snapshot()
        if step == 5:
            ev.append(("rm", 1000))
            ev.append(("add", 1001, ["zebra", "carpet"]))
            timg, dimg, _ = nxsfmt.build_images_log(ev)
            nxsfmt.publish_in_place(t, d, timg, dimg)
            snapshot()
        gidx.search_batch_begin(qs, limit=10)
        seen_by.append(len(oracles) - 1)
        got = gidx.search_batch_end()
        for q, g in zip(qs, got):
            assert_same(g, oracles[seen_by[step - 1]].search(q, limit=10), (step, q))
    for q, g in zip(qs, gidx.search_batch_end()):
        assert_same(g, oracles[seen_by[-1]].search(q, limit=10), ("last", q))
    assert seen_by[2] == 1 and seen_by[5] == 2
    assert [x for x, _ in gidx.search("monkei")] == [x for x, _ in oracles[-1].search("monkei")]
    gidx.close()


@pytest.mark.parametrize("bm", [False, True])
def test_incremental_refresh_interleaved_appends_and_removes(nxs, tmp_path, monkeypatch, bm):
    """N1: 100 interleaved appends (new docs with growing ids, some with new
    terms) and removals, each published in place and picked up by the NEXT
    search through the incremental path (delta merged into the device CSR, all
    impacts recomputed, BK-tree re-flattened for new terms) -- never a rebuild;
    results against a freshly loaded oracle after every step.  A re-used doc id
    then takes the rebuild path, also exactly."""
    import ctypes as C
    if bm:      # conjunctions through the block-presence bitmaps, rebuilt at every refresh
        monkeypatch.setenv("NXS_GPU_BM_GAIN", "0")
        monkeypatch.setenv("NXS_GPU_BM_SHARE", "1073741824")
    rng = random.Random(97)
    vocab = ["w%d" % i for i in range(60)]
    weights = [1.0 / (i + 1) for i in range(len(vocab))]
    mk = lambda: rng.choices(vocab, weights, k=rng.randint(2, 9))
    events = [("add", i + 1, mk()) for i in range(400)]
    timg, dimg, _ = nxsfmt.build_images_log(events)
    t, d = str(tmp_path / "nxsterms"), str(tmp_path / "nxsdtmap")
    # room for the appends: the files are sized once, like a preallocated index
    with open(t, "wb") as f:
        f.write(timg + b"\0" * (1 << 16))
    with open(d, "wb") as f:
        f.write(dimg + b"\0" * (1 << 18))
    gidx = nxs.open_files(t, d)
    stats = (C.c_uint64 * 2)()
    L = N.lib()
    L.nxs_index_refresh_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    alive, next_id, n_new_terms = set(range(1, 401)), 401, 0
    # (the plan cache keeps what a query string compiled to: a term that does not exist YET must be
    # found once it does -- every refresh clears the cache)
    assert gidx.search_batch(["zzzfuture", "w0 OR zzzfuture"], limit=10, fuzzymatch=False)[0] == []
    queries = ["w0", "w1 AND w2", "w3 OR w7 OR w20", "w0 AND NOT w1", "w5 OR w40 OR w55 OR w9 OR w2",
               "w2 AND w4 AND w1", "w3 AND (w0 OR w9) AND w1"]
    for step in range(100):
        r = rng.random()
        if r < 0.2 and len(alive) > 50:
            victim = rng.choice(sorted(alive))
            alive.discard(victim)
            events.append(("rm", victim))
        else:
            for _ in range(rng.randint(1, 3)):
                toks = mk()
                if rng.random() < 0.3:
                    n_new_terms += 1
                    toks.append("fresh%d" % n_new_terms)        # a term nxsterms did not hold
                events.append(("add", next_id, toks))
                alive.add(next_id)
                next_id += rng.randint(1, 5)
        timg, dimg, _ = nxsfmt.build_images_log(events)
        nxsfmt.publish_in_place(t, d, timg, dimg)
        oidx = O.Index(t, d)
        qs = queries + (["fresh%d" % n_new_terms, "fresh%d OR w1" % max(1, n_new_terms - 1)] if n_new_terms else [])
        if step % 10 == 0:
            qs = qs + ["w0 OR frush%d" % max(1, n_new_terms)]         # fuzzy -> the newest term (BK image)
        algo = ("BM25", 1) if step % 3 else ("TF-IDF", 0)
        got = gidx.search_batch(qs, limit=10, algo=algo[0])
        for q, g in zip(qs, got):
            assert_same(g, oidx.search(q, algo=algo[1], limit=10), (step, q))
        if step % 25 == 0:
            assert_same(gidx.search("w1 OR w2"), oidx.search("w1 OR w2"), step)     # default limit
    L.nxs_index_refresh_stats(gidx._h, stats)
    assert (stats[0], stats[1]) == (100, 0), list(stats)
    events.append(("add", next_id, ["zzzfuture", "w0"]))
    timg, dimg, _ = nxsfmt.build_images_log(events)
    nxsfmt.publish_in_place(t, d, timg, dimg)
    oidx = O.Index(t, d)
    for q, g in zip(["zzzfuture", "w0 OR zzzfuture"], gidx.search_batch(["zzzfuture", "w0 OR zzzfuture"], limit=10, fuzzymatch=False)):
        assert g and [d_ for d_, _ in g] == [d_ for d_, _ in oidx.search(q, limit=10, fuzzymatch=False)], q
    L.nxs_index_refresh_stats(gidx._h, stats)
    assert (stats[0], stats[1]) == (101, 0), list(stats)
    stats_base = 101
    # a removed id comes back: not an append of a higher id => full rebuild, same answers
    gone = sorted(set(range(1, 401)) - alive)[0]
    events.append(("add", gone, ["w0", "w1", "w1"]))
    timg, dimg, _ = nxsfmt.build_images_log(events)
    nxsfmt.publish_in_place(t, d, timg, dimg)
    oidx = O.Index(t, d)
    for q in queries:
        assert_same(gidx.search(q, limit=10), oidx.search(q, limit=10), q)
    L.nxs_index_refresh_stats(gidx._h, stats)
    assert (stats[0], stats[1]) == (stats_base, 1), list(stats)
    gidx.close()


def test_incremental_refresh_grows_the_files_and_keeps_serving(nxs, tmp_path):
    """The files outgrow their mapping (32 KiB steps, index.h:24): re-mapped;
    a block naming a term that nxsterms does not hold yet is NOT consumed
    (partial sync, dtmap.c:527-535) until the term arrives."""
    events = [("add", i + 1, ["a", "b", "c%d" % (i % 7)]) for i in range(50)]
    timg, dimg, _ = nxsfmt.build_images_log(events)
    t, d = str(tmp_path / "nxsterms"), str(tmp_path / "nxsdtmap")
    open(t, "wb").write(timg)
    open(d, "wb").write(dimg)
    gidx = nxs.open_files(t, d)
    assert len(gidx.search("a")) == 50
    # 3000 more docs: both files grow past 32 KiB
    events += [("add", 100 + i, ["a", "t%d" % i, "t%d" % (i // 2)]) for i in range(3000)]
    timg2, dimg2, _ = nxsfmt.build_images_log(events)
    assert len(dimg2) > len(dimg) and len(timg2) > len(timg)
    for path, img in ((t, timg2), (d, dimg2)):
        with open(path, "r+b") as f:
            f.seek(0, 2)
            f.write(b"\0" * (len(img) - f.tell()))
    # publish the dtmap FIRST: its new blocks name terms the term file lacks
    nxsfmt.publish_in_place(t, d, timg, dimg2)
    oidx_old = O.Index(t, d)
    assert_same(gidx.search("a", limit=5), oidx_old.search("a", limit=5))
    nxsfmt.publish_in_place(t, d, timg2, dimg2)
    oidx = O.Index(t, d)
    for q in ("a", "t2999", "t7 OR c3", "b AND NOT a"):
        assert_same(gidx.search(q, limit=20), oidx.search(q, limit=20), q)
    assert len(gidx.search("a", limit=5000)) == 3050
    gidx.close()


@pytest.mark.parametrize("n_shards", [2, 3])
def test_doc_sharded_collection_equals_the_whole_index(nxs, tmp_path, n_shards):
    """N4: the collection cut into doc ranges, every shard a device index of its
    own scoring with collection-wide N / token count / df; a batch runs on every
    shard and the shards' accepted candidates are replayed through the heap once
    more (highest docs first).  Same ids, order (massive ties) and score bits as
    the unsharded index = the oracle."""
    rng = random.Random(61 + n_shards)
    vocab = ["w%d" % i for i in range(50)]
    weights = [1.0 / (i + 1) for i in range(len(vocab))]
    pool = rng.choices(vocab, weights, k=4096)
    docs = random_corpus(rng, 9000, pool, max_len=7, sparse=True)
    t, d, _ = nxsfmt.write_index(str(tmp_path), "whole", docs)
    oidx = O.Index(t, d)
    shards = [nxs.open_shard(t, d, s_, n_shards) for s_ in range(n_shards)]
    queries = [random_query(rng, vocab[:16]) for _ in range(50)]
    queries += ["w0", "w0 AND w1", "w49 OR w0", "w3 OR w4 OR w5 OR w6 OR w7", "broken AND", "w1 AND NOT w0",
                "zzzz OR yyyy", "w2 OR ww3"]
    for limit in (1, 10, 64, 65, 300, 2000):        # (above 64: the merge's heap lives in LDS, k_replay_coop)
        for algo, name in ((1, "BM25"), (0, "TF-IDF")):
            got = nxs.docshard_search_batch(shards, queries, limit=limit, algo=name)
            for q, g in zip(queries, got):
                try:
                    want = oidx.search(q, algo=algo, limit=limit)
                except O.SearchError as e:
                    assert isinstance(g, N.NxsError) and g.code == e.code
                    continue
                assert_same(g, want, (n_shards, q, limit, name))
    # a synthetic corpus too (sparse and dense terms, C3-style queries)
    c = corpus.write_corpus(str(tmp_path / "syn"), 150_000, 6000, seed=71)
    terms = corpus.term_strings(6000, seed=71)
    o2 = O.Index(c["terms"], c["dtmap"])
    sh2 = [nxs.open_shard(c["terms"], c["dtmap"], s_, n_shards) for s_ in range(n_shards)]
    qs = corpus.queries_bool5(terms, 48, seed=5, hi=500) + corpus.queries_single(terms, 8, seed=6, lo=1, hi=200)
    for q, g in zip(qs, nxs.docshard_search_batch(sh2, qs, limit=10, fuzzymatch=False)):
        assert_same(g, o2.search(q, limit=10, fuzzymatch=False), q)
    for q, g in zip(qs[:16], nxs.docshard_search_batch(sh2, qs[:16], limit=1000, fuzzymatch=False)):
        assert_same(g, o2.search(q, limit=1000, fuzzymatch=False), (q, 1000))
    with pytest.raises(N.NxsError) as e:
        nxs.docshard_search_batch(sh2, qs[:2], limit=8001)      # beyond the fixed-size records
    assert e.value.code == 6
    # one process per shard, the ranks played one after the other on this GPU: every
    # rank's candidate block, the blocks laid out as the all-gather would, every rank's
    # merge -- all ranks hold the whole batch's responses
    for per_rank in nxs.docshard_emulated_ranks(sh2, qs, limit=10, fuzzymatch=False):
        for q, g in zip(qs, per_rank):
            assert_same(g, o2.search(q, limit=10, fuzzymatch=False), ("rank form", q))
    for per_rank in nxs.docshard_emulated_ranks(shards, queries[:40], cap=4096, limit=64):
        for q, g in zip(queries[:40], per_rank):
            try:
                want = oidx.search(q, limit=64)
            except O.SearchError as e:
                assert isinstance(g, N.NxsError) and g.code == e.code
                continue
            assert_same(g, want, ("rank form", n_shards, q))
    for s_ in shards + sh2:
        s_.close()


def test_doc_sharded_rank_form_through_rccl(nxs, tmp_path):
    """The one-process-per-shard entry points end to end on ONE rank: a real RCCL
    communicator (world 1), nxs_docshard_attach (all-gather of the df arrays) and
    nxs_docshard_search_batch_rank (all-gather of the candidate blocks, merge)."""
    from nxsearch_amd import multi
    c = corpus.write_corpus(str(tmp_path), 60_000, 3000, seed=73)
    terms = corpus.term_strings(3000, seed=73)
    oidx = O.Index(c["terms"], c["dtmap"])
    sh = nxs.open_shard(c["terms"], c["dtmap"], 0, 1)
    multi.attach(nxs, sh, 0, 1)
    nxs.docshard_attach(sh)
    qs = corpus.queries_bool5(terms, 40, seed=5, hi=400) + corpus.queries_single(terms, 8, seed=6, lo=1, hi=200)
    for q, g in zip(qs, nxs.docshard_search_batch_rank(sh, qs, limit=10, fuzzymatch=False)):
        assert_same(g, oidx.search(q, limit=10, fuzzymatch=False), q)
    sh.close()


@pytest.mark.parametrize("env", [{}, {"NXS_GPU_DROP_MINPOST": "1"}, {"NXS_GPU_DROP_MINPOST": "1", "NXS_GPU_WAVES": "32"},
                                 {"NXS_GPU_DROP_MINPOST": "1", "NXS_GPU_SCANM_DENS": "0.02"},
                                 # TF-IDF: the dense terms' caps and outlier lists -- few outliers, half of the list, none
                                 {"NXS_GPU_DROP_MINPOST": "1", "NXS_GPU_OUTL_SHARE": "64"},
                                 {"NXS_GPU_DROP_MINPOST": "1", "NXS_GPU_OUTL_SHARE": "2", "NXS_GPU_WAVES": "64"},
                                 {"NXS_GPU_TFIDF_NODROP": "1"},
                                 # the class's second kernel on presence bits (k_scanb<.., DROP>), outlier lists included
                                 {"NXS_GPU_DROPB": "1", "NXS_GPU_DROP_MINPOST": "1"},
                                 {"NXS_GPU_DROPB": "1", "NXS_GPU_DROP_MINPOST": "1", "NXS_GPU_WAVES": "32"},
                                 {"NXS_GPU_DROPB": "1", "NXS_GPU_DROP_MINPOST": "1", "NXS_GPU_OUTL_SHARE": "2", "NXS_GPU_WAVES": "64"},
                                 # ... on doc stripes (k_scans<.., DROP>: BM25; queries with outlier lists stay on k_scanm) / the plain
                                 # mask class on register windows only
                                 {"NXS_GPU_SCANS_DROP": "1", "NXS_GPU_DROP_MINPOST": "1"},
                                 {"NXS_GPU_SCANS_DROP": "1", "NXS_GPU_DROP_MINPOST": "1", "NXS_GPU_WAVES": "32"},
                                 {"NXS_GPU_NOSCANS": "1", "NXS_GPU_DROP_MINPOST": "1"}])
def test_dense_terms_leave_sparse_or_scans(nxs, tmp_path, monkeypatch, env):
    """k_scanm<.., DROP>: pure-OR queries that mix dense terms (8 % of the docs and
    more) with sparse ones.  The dense lists are scanned only until the threshold
    exceeds what they can contribute, then looked up per candidate; cold starts
    (tiny ranges, fewer sparse docs than k), massive ties and both rankings."""
    for kk, v in env.items():
        monkeypatch.setenv(kk, v)
    c = corpus.write_corpus(str(tmp_path), 300_000, 12_000, seed=83)
    terms = corpus.term_strings(12_000, seed=83)
    gidx, oidx = nxs.open_files(c["terms"], c["dtmap"]), O.Index(c["terms"], c["dtmap"])
    rng = random.Random(3)
    T = lambda r: terms[r - 1].decode()
    qs = []
    for _ in range(60):
        nd, ns = rng.randint(1, 2), rng.randint(1, 5)
        ranks = rng.sample(range(1, 25), nd) + rng.sample(range(40, 4000), ns)
        rng.shuffle(ranks)
        qs.append(" OR ".join(T(r) for r in ranks))
    qs += ["%s OR %s" % (T(1), T(11000)), "%s OR %s OR %s" % (T(2), T(3), T(9000)),    # almost no sparse docs
           "%s %s %s %s" % (T(5), T(300), T(301), T(302))]
    for limit in (1, 10, 64):
        for algo, name in ((1, "BM25"), (0, "TF-IDF")):
            got = gidx.search_batch(qs, limit=limit, algo=name, fuzzymatch=False)
            for q, g in zip(qs, got):
                assert_same(g, oracle_memo("dense", oidx, q, algo=algo, limit=limit, fuzzymatch=False), (env, q, limit, name))
    gidx.close()


def test_full_size_c3_c4_properties(nxs, tmp_path, monkeypatch):
    """BASELINE.json configs[1] at its FULL size (10M docs / 1M terms, a batch of 1024 five-term
    AND / OR queries, top-10) -- where the oracle answers only a sample in reasonable time -- through
    properties that do not depend on the size: every answer is sorted by descending score with valid
    doc ids; the same batch gives the same bits again, blocking and with two batches in flight; the
    top-10 scores are the first ten of the top-64 scores, and those the first 64 of the top-200 (the
    lane heap, the LDS heap and the histogram threshold agree); three independent scan algorithms --
    the byte-bound filter (k_scanm / k_scanr), the f32 accumulator tiles alone (NXS_GPU_NOSCANM,
    NXS_GPU_NOSCANR), the presence-bit kernel (k_scanb for every eligible query) and the block
    bitmaps on / off for the conjunctions -- return identical ids and score bits;
    single-term answers are the term's largest impacts; and a sample is checked against the oracle."""
    work = "/dev/shm/nxs_fullsize_%d" % os.getpid()
    try:
        c = corpus.write_corpus(work, 10_000_000, 1_000_000, seed=0)
        terms = corpus.term_strings(1_000_000, seed=0)
        gidx = nxs.open_files(c["terms"], c["dtmap"])
        qs = corpus.queries_bool5(terms, 1024, seed=3, hi=1000)
        bits = lambda res: [[(d, struct.pack("<f", s)) for d, s in r] for r in res]
        base = gidx.search_batch(qs, limit=10, fuzzymatch=False)
        assert not any(isinstance(r, N.NxsError) for r in base)
        n_docs = 10_000_000
        for r in base:
            assert len(r) <= 10 and len({d for d, _ in r}) == len(r)
            assert all(1 <= d <= n_docs for d, _ in r)
            assert all(a[1] >= b[1] for a, b in zip(r, r[1:]))
        assert sum(len(r) == 10 for r in base) >= 512         # every OR query fills its ten
        # idempotence: blocking again, then two batches in flight
        assert bits(gidx.search_batch(qs, limit=10, fuzzymatch=False)) == bits(base)
        gidx.search_batch_begin(qs, limit=10, fuzzymatch=False)
        gidx.search_batch_begin(qs[::-1], limit=10, fuzzymatch=False)
        assert bits(gidx.search_batch_end()) == bits(base)
        assert bits(gidx.search_batch_end()) == bits(base[::-1])
        # score prefixes across the heap homes
        sub = qs[:128]
        r64 = gidx.search_batch(sub, limit=64, fuzzymatch=False)
        r200 = gidx.search_batch(sub, limit=200, fuzzymatch=False)
        sc = lambda r: [struct.pack("<f", s) for _, s in r]
        for a, b, cc in zip(base, r64, r200):
            assert sc(b)[:len(a)] == sc(a) and sc(cc)[:len(b)] == sc(b)
            assert len(a) == min(10, len(b)) and len(b) == min(64, len(cc))
        # independent scan algorithms, same bits
        for env in ({"NXS_GPU_NOSCANM": "1", "NXS_GPU_NOSCANR": "1"},
                    {"NXS_GPU_NODROP": "1", "NXS_GPU_WAVES": "8192"},
                    {"NXS_GPU_NOBLKMAP": "1"},                  # conjunctions on k_scanr instead of the bitmaps
                    {"NXS_GPU_BM_GAIN": "0"},                   # ... every one of them on the bitmaps (k_scanq)
                    {"NXS_GPU_SCANB_DENS": "1.0"}):             # the sparse OR class on presence bits (k_scanb)
            for kk, v in env.items():
                monkeypatch.setenv(kk, v)
            gidx.reconfigure()
            assert bits(gidx.search_batch(qs, limit=10, fuzzymatch=False)) == bits(base), env
            for kk in env:
                monkeypatch.delenv(kk)
        gidx.reconfigure()
        # TF-IDF: dense terms leave the scan on a capped ceiling + outlier lists; without them (accumulator tiles) the same bits
        tf = gidx.search_batch(qs, limit=10, algo="TF-IDF", fuzzymatch=False)
        monkeypatch.setenv("NXS_GPU_TFIDF_NODROP", "1")
        gidx.reconfigure()
        assert bits(gidx.search_batch(qs, limit=10, algo="TF-IDF", fuzzymatch=False)) == bits(tf)
        monkeypatch.delenv("NXS_GPU_TFIDF_NODROP")
        gidx.reconfigure()
        # a single-term query's scores are its list's largest impacts: the top-3 of top-10 and of top-1000 agree
        for q in (terms[0].decode(), terms[99].decode(), terms[4999].decode()):
            a, b = gidx.search(q, limit=10, fuzzymatch=False), gidx.search(q, fuzzymatch=False)
            assert sc(b)[:10] == sc(a) and len(b) == 1000
        # the oracle on a sample of 40 (both operators; ~0.25 s per query): every 32nd query plus the
        # first OR queries that hold a dense term (rank <= 27: the k_cold + k_scanm<.., DROP> class)
        oidx = O.Index(c["terms"], c["dtmap"])
        dense = set(t.decode() for t in terms[:27])
        with_dense = [i for i, q in enumerate(qs) if " OR " in q and dense & set(q.split())][:8]
        assert len(with_dense) == 8
        sample = sorted(set(range(0, 1024, 32)) | set(with_dense))
        assert len(sample) >= 36
        for i in sample:
            assert_same(base[i], oidx.search(qs[i], limit=10, fuzzymatch=False), qs[i])
        for i in sample[::4] + with_dense[:2]:
            assert_same(tf[i], oidx.search(qs[i], algo=0, limit=10, fuzzymatch=False), (qs[i], "TF-IDF"))
        # ... and every fourth query of the batch (256 of 1024, both operators, every class of the work list) on eight
        # cores beside this process: tests/oracle_pool.py
        wide = list(range(1, 1024, 4))
        for i, want in zip(wide, oracle_many(c, [(qs[i], 1, 10, False) for i in wide])):
            assert want is not None
            assert_same(base[i], want, ("pool", qs[i]))
        # ---- configs[3]: Levenshtein d <= 2 over the 1M-term BK-tree, batch 1024 ----
        toks = corpus.queries_fuzzy(terms, 1024, seed=4)
        ids = gidx.fuzzy(toks)
        assert all(ids), "every token has a match at distance 1"

        def lev(a, b):
            row = list(range(len(b) + 1))
            for i, ca in enumerate(a, 1):
                prev, row[0] = row[0], i
                for j, cb in enumerate(b, 1):
                    prev, row[j] = row[j], min(row[j] + 1, row[j - 1] + 1, prev + (ca != cb))
            return row[-1]
        for tok, t in zip(toks, ids):
            assert lev(tok.encode(), terms[t - 1]) <= 2, (tok, t)
        assert gidx.fuzzy(toks) == ids                                   # idempotent
        assert gidx.fuzzy(toks[:1]) == ids[:1] and gidx.fuzzy(toks[500:600]) == ids[500:600]
        ids_v, vis = gidx.fuzzy(toks, want_visited=True)                 # the frontier search, visit counts
        assert ids_v == ids and min(vis) > 0
        monkeypatch.setenv("NXS_GPU_FUZZY_BFS", "1")                     # level-by-level frontier only
        gidx.reconfigure()
        assert gidx.fuzzy(toks) == ids
        monkeypatch.delenv("NXS_GPU_FUZZY_BFS")
        gidx.reconfigure()
        for i in range(0, 1024, 128):                                    # the oracle's BK-tree walk on a sample
            assert (ids[i], vis[i]) == oidx.fuzzy(toks[i].encode()), toks[i]
        # fuzzy tokens inside boolean queries resolve to the same terms (tokenset fallback)
        fq = ["%s OR %s" % (toks[i], terms[200 + i].decode()) for i in range(4)]
        pq = ["%s OR %s" % (terms[ids[i] - 1].decode(), terms[200 + i].decode()) for i in range(4)]
        assert bits(gidx.search_batch(fq, limit=10, fuzzymatch=True)) == bits(gidx.search_batch(pq, limit=10, fuzzymatch=False))
        oidx.close()
        gidx.close()
    finally:
        shutil.rmtree(work, ignore_errors=True)


def test_full_size_c5_properties(nxs, monkeypatch):
    """BASELINE.json configs[4] at its FULL size on one GPU: 50M docs / 2M terms (1.6 G postings), the
    8192-query mixed batch (75 % five-term AND / OR, 25 % with one misspelt token that the BK-tree
    search has to resolve), top-10.  The batch goes through the SHARDED entry as the eight ranks of
    the 8-GPU run, one after the other (each plans and scans ITS slice, the library reassembles
    the eight record blocks): identical ids and score bits to the unsharded answer; every answer
    sorted with valid, distinct doc ids; a query with a misspelt token == the same query with the
    resolved term; eight queries and eight fuzzy tokens against the oracle."""
    from nxsearch_amd import multi
    work = "/dev/shm/nxs_fullsize_c5_%d" % os.getpid()
    n_docs, n_terms, n, k, world = 50_000_000, 2_000_000, 8192, 10, 8
    try:
        c = corpus.write_corpus(work, n_docs, n_terms, seed=0)
        terms = corpus.term_strings(n_terms, seed=0)
        have = set(terms)
        gidx = nxs.open_files(c["terms"], c["dtmap"])
        qs = corpus.queries_mixed(terms, n, seed=6, hi=1000)
        bits = lambda res: [[(d, struct.pack("<f", s)) for d, s in r] for r in res]
        base = gidx.search_batch(qs, limit=k)
        assert not any(isinstance(r, N.NxsError) for r in base)
        for r in base:
            assert len(r) <= k and len({d for d, _ in r}) == len(r)
            assert all(1 <= d <= n_docs for d, _ in r)
            assert all(a[1] >= b[1] for a, b in zip(r, r[1:]))
        # the eight ranks of the sharded run, emulated on the one GPU
        cap = multi.shard_capacity(n, world)
        blocks = b""
        for r in range(world):
            multi.emulate(gidx, r, world)
            gidx.search_batch_begin(qs, limit=k)
            gidx.search_batch_end()
            blk = multi.emulated_block(gidx)
            assert len(blk) == multi.block_bytes(cap, k)
            blocks += blk
        multi.emulate(gidx, 0, 0)
        assert bits(multi.assemble(blocks, world, cap, k, n)) == bits(base)
        # fuzzy token inside a query == the resolved term inside it
        fz = [(i, q) for i, q in enumerate(qs) if any(w.encode() not in have for w in q.split() if w not in ("AND", "OR"))]
        assert len(fz) > n // 5
        toks = [[w for w in q.split() if w not in ("AND", "OR") and w.encode() not in have][0] for _, q in fz[:64]]
        ids = gidx.fuzzy(toks)
        assert all(ids)
        resolved = [q.replace(t, terms[tid - 1].decode()) for (_, q), t, tid in zip(fz[:64], toks, ids)]
        assert bits(gidx.search_batch(resolved, limit=k, fuzzymatch=False)) == bits([base[i] for i, _ in fz[:64]])
        # the oracle: eight queries (both operators, two with a misspelt token) and eight fuzzy tokens
        oidx = O.Index(c["terms"], c["dtmap"])
        sample = [0, 1, 4096, 4097, 8190, 8191, fz[0][0], fz[1][0]]
        for i in sample:
            assert_same(base[i], oidx.search(qs[i], limit=k), qs[i])
        for tok, t in zip(toks[:8], ids[:8]):
            assert t == oidx.fuzzy(tok.encode())[0], tok
        oidx.close()
        # ... and 64 more queries of the batch (every 128th from the 5th on: all shapes of the mix, fuzzy tokens resolved by
        # the oracle's own BK-tree walk) on eight cores beside this process: tests/oracle_pool.py
        wide = list(range(5, 8192, 128))
        for i, want in zip(wide, oracle_many(c, [(qs[i], 1, k, True) for i in wide])):
            if want is None:
                assert isinstance(base[i], N.NxsError), qs[i]
            else:
                assert_same(base[i], want, ("pool", qs[i]))
        gidx.close()
    finally:
        shutil.rmtree(work, ignore_errors=True)


def test_full_size_c2_properties(nxs):
    """BASELINE.json configs[1]: 1M docs / 100k terms, single-term BM25 top-10, at full size: sorted
    answers, blocking == pipelined, the single-token kernel == the accumulator tiles
    (NXS_GPU_NOSCAN1), top-10 a prefix of the default limit's 1000, TF-IDF and BM25 rank the same
    docs of a list differently but return as many, and the oracle on a sample."""
    work = "/dev/shm/nxs_fullsize_c2_%d" % os.getpid()
    try:
        c = corpus.write_corpus(work, 1_000_000, 100_000, seed=0)
        terms = corpus.term_strings(100_000, seed=0)
        gidx = nxs.open_files(c["terms"], c["dtmap"])
        qs = corpus.queries_single(terms, 1024, seed=2, lo=1, hi=100_000)
        bits = lambda res: [[(d, struct.pack("<f", s)) for d, s in r] for r in res]
        base = gidx.search_batch(qs, limit=10, fuzzymatch=False)
        for r in base:
            assert 1 <= len(r) <= 10 and all(a[1] >= b[1] for a, b in zip(r, r[1:]))
        gidx.search_batch_begin(qs, limit=10, fuzzymatch=False)
        gidx.search_batch_begin(qs, limit=10, fuzzymatch=False)
        assert bits(gidx.search_batch_end()) == bits(base) and bits(gidx.search_batch_end()) == bits(base)
        import os as _os
        _os.environ["NXS_GPU_NOSCAN1"] = "1"
        try:
            gidx.reconfigure()
            assert bits(gidx.search_batch(qs, limit=10, fuzzymatch=False)) == bits(base)
        finally:
            del _os.environ["NXS_GPU_NOSCAN1"]
            gidx.reconfigure()
        big = gidx.search_batch(qs[:64], fuzzymatch=False)               # params NULL => 1000
        tf = gidx.search_batch(qs[:64], limit=10, algo="TF-IDF", fuzzymatch=False)
        for a, b, t in zip(base, big, tf):
            assert [struct.pack("<f", s) for _, s in b][:len(a)] == [struct.pack("<f", s) for _, s in a]
            assert len(t) == len(a)
        oidx = O.Index(c["terms"], c["dtmap"])
        for i in range(0, 1024, 32):
            assert_same(base[i], oidx.search(qs[i], limit=10, fuzzymatch=False), qs[i])
        oidx.close()
        gidx.close()
    finally:
        shutil.rmtree(work, ignore_errors=True)

"""Validates the oracle restatement against the GENUINE reference
src/algo/{levdist,bktree,deque,heap}.c compiled in place into
oracle/_ref/libnxsref.so (oracle/Makefile).  Skipped when that library has
not been built (it cannot be built without /root/reference)."""
import random

import pytest

import oracle_lib as O

pytestmark = pytest.mark.skipif(O.ref() is None, reason="oracle/_ref not built")


def rand_word(rng, lo=1, hi=12, alphabet="abcdefgh"):
    return "".join(rng.choice(alphabet) for _ in range(rng.randint(lo, hi))).encode()


def test_levdist_matches_reference_on_random_strings():
    rng = random.Random(11)
    R = O.ref()
    for _ in range(20000):
        a = rand_word(rng, 0, 24, "abcxyz\xc4\x85 ")
        b = rand_word(rng, 0, 24, "abcxyz\xc4\x85 ")
        assert O.levdist(a, b) == R.ref_levdist(a, len(a), b, len(b))
    # long strings (row > 255, uint16 row)
    for _ in range(20):
        a = rand_word(rng, 200, 700, "ab")
        b = rand_word(rng, 200, 700, "ab")
        assert O.levdist(a, b) == R.ref_levdist(a, len(a), b, len(b))


@pytest.mark.parametrize("seed,n,alphabet", [(1, 300, "abcd"), (2, 3000, "abcdefghijklmnopqrstuvwxyz"), (3, 2000, "ab")])
def test_bktree_search_order_matches_reference(seed, n, alphabet):
    rng = random.Random(seed)
    words = []
    seen = set()
    while len(words) < n:
        w = rand_word(rng, 1, 10, alphabet)
        if rng.random() < 0.05 and words:
            w = rng.choice(words)          # duplicates are rejected (EEXIST)
        words.append(w)
        seen.add(w)
    ref = O.RefBKTree(words)
    orc = O.BKTree(words)
    for _ in range(300):
        q = bytearray(rng.choice(words))
        if q and rng.random() < 0.8:
            q[rng.randrange(len(q))] = ord(rng.choice(alphabet))
        if rng.random() < 0.3:
            q += rng.choice(alphabet).encode()
        for tol in (0, 1, 2, 3):
            r1, n1 = ref.search(bytes(q), tol)
            r2, n2 = orc.search(bytes(q), tol)
            assert r1 == r2, (q, tol)
            assert n1 == n2
    ref.close()
    orc.close()


def test_bktree_long_words_clamped_slot():
    # distances above 63 share one bucket (bktree.c:196)
    rng = random.Random(5)
    words = [rand_word(rng, 1, 4, "ab") for _ in range(50)]
    words += [rand_word(rng, 70, 120, "abc") for _ in range(30)]
    rng.shuffle(words)
    ref = O.RefBKTree(words)
    orc = O.BKTree(words)
    for q in [rand_word(rng, 1, 5, "ab") for _ in range(50)]:
        assert ref.search(q, 2) == orc.search(q, 2)


def test_topk_matches_reference_heap_with_massive_ties():
    rng = random.Random(7)
    for trial in range(400):
        n = rng.randint(0, 300)
        k = rng.choice([1, 2, 3, 5, 10, 17, 64, 1000])
        levels = rng.choice([1, 2, 3, 8, 1000])
        ids = list(range(n, 0, -1))                      # descending doc id
        sc = [float(rng.randrange(levels)) / 4 + 0.25 for _ in ids]
        assert O.topk(ids, sc, k) == O.ref_topk(ids, sc, k), (trial, n, k, levels)

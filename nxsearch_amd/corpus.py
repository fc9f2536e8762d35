"""Synthetic corpus tooling (bench + tests): writes VALID nxsterms/nxsdtmap
files through csrc/nxs_synth.c and builds the query sets of BASELINE.json's
configs."""
import ctypes as C
import os
import random

from . import SYNTH_PATH

_synth = None


def _lib():
    global _synth
    if _synth is None:
        if not os.path.exists(SYNTH_PATH):
            raise ImportError("%s is missing: run make -C nxsearch_amd/csrc" % SYNTH_PATH)
        S = C.CDLL(SYNTH_PATH)
        S.nxs_synth_write.restype = C.c_int
        S.nxs_synth_write.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64, C.c_uint32,
                                      C.c_uint64, C.c_double, C.c_int, C.c_int,
                                      C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        S.nxs_synth_terms.restype = C.c_uint64
        S.nxs_synth_terms.argtypes = [C.c_uint32, C.c_uint64, C.c_char_p,
                                      C.POINTER(C.c_uint32)]
        _synth = S
    return _synth


def write_corpus(dirpath, n_docs, n_terms, seed=0, mean_distinct=31.0,
                 sparse_ids=False, threads=None):
    """-> dict(terms=path, dtmap=path, postings=P, tokens=N)."""
    os.makedirs(dirpath, exist_ok=True)
    tp = os.path.join(dirpath, "nxsterms")
    dp = os.path.join(dirpath, "nxsdtmap")
    post, tok = C.c_uint64(), C.c_uint64()
    if threads is None:
        threads = min(os.cpu_count() or 1, 32)
    r = _lib().nxs_synth_write(os.fsencode(tp), os.fsencode(dp), n_docs, n_terms,
                               seed, mean_distinct, 1 if sparse_ids else 0,
                               threads, C.byref(post), C.byref(tok))
    if r != 0:
        raise OSError("nxs_synth_write failed")
    return dict(terms=tp, dtmap=dp, postings=post.value, tokens=tok.value,
                n_docs=n_docs, n_terms=n_terms, seed=seed)


def term_strings(n_terms, seed=0):
    """The term strings of write_corpus(seed) in term-id order (id = index+1)."""
    buf = C.create_string_buffer(n_terms * 13 + 16)
    offs = (C.c_uint32 * (n_terms + 1))()
    _lib().nxs_synth_terms(n_terms, seed + 1, buf, offs)
    raw = buf.raw
    return [raw[offs[i]:offs[i + 1]] for i in range(n_terms)]


def queries_single(terms, n, seed=3, lo=10, hi=10000):
    """C2: one term of rank uniform in [lo, hi]."""
    rng = random.Random(seed)
    hi = min(hi, len(terms))
    lo = min(lo, hi)
    return [terms[rng.randint(lo, hi) - 1].decode() for _ in range(n)]


def queries_bool5(terms, n, seed=3, hi=1000, k=5, lo=1):
    """C3: 5 distinct terms of rank uniform in [lo, hi]; half AND, half OR."""
    rng = random.Random(seed)
    hi = min(hi, len(terms))
    out = []
    for i in range(n):
        ranks = rng.sample(range(lo, hi + 1), min(k, hi - lo + 1))
        op = " AND " if i % 2 == 0 else " OR "
        out.append(op.join(terms[r - 1].decode() for r in ranks))
    return out


def queries_mixed(terms, n, seed=6, hi=1000, fuzzy_share=0.25):
    """C5: 75 % C3-style queries, 25 % with ONE of the five terms replaced by a
    misspelling (one byte substituted, exact lookup fails, a d<=1 match exists)
    that the fuzzy path has to resolve.  Seed-stable."""
    rng = random.Random(seed)
    have = set(terms)
    base = queries_bool5(terms, n, seed=seed + 1, hi=hi)
    out = []
    for i, q in enumerate(base):
        if rng.random() < fuzzy_share:
            parts = q.split(" ")
            slots = [j for j, p in enumerate(parts) if p not in ("AND", "OR")]
            j = rng.choice(slots)
            while True:
                t = bytearray(parts[j].encode())
                t[rng.randrange(len(t))] = ord("a") + rng.randrange(26)
                if bytes(t) not in have:
                    break
            parts[j] = bytes(t).decode()
            q = " ".join(parts)
        out.append(q)
    return out


def queries_fuzzy(terms, n, seed=4):
    """C4: an existing term with one random byte substituted such that the
    exact lookup fails (a d<=1 match exists)."""
    rng = random.Random(seed)
    have = set(terms)
    out = []
    while len(out) < n:
        t = bytearray(terms[rng.randrange(len(terms))])
        t[rng.randrange(len(t))] = ord("a") + rng.randrange(26)
        t = bytes(t)
        if t not in have:
            out.append(t.decode())
    return out

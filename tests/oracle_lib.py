"""ctypes bindings to the CPU oracle (oracle/nxs_oracle.c) and, when built,
the genuine reference algo/ sources (oracle/_ref/libnxsref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this.
"""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libnxsref.so")

TF_IDF, BM25 = 0, 1


class Result(C.Structure):
    _fields_ = [("doc_id", C.c_uint64), ("score", C.c_float)]


def build_oracle():
    """Compile the oracle (and oracle/_ref when /root/reference exists)."""
    subprocess.run(["make", "-C", ORACLE_DIR], check=True,
                   stdout=subprocess.DEVNULL)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(ORACLE_SO):
        build_oracle()
    L = C.CDLL(ORACLE_SO)
    L.orc_levdist.restype = C.c_int
    L.orc_levdist.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    L.orc_bm25.restype = C.c_float
    L.orc_bm25.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_uint64, C.c_uint64]
    L.orc_tf_idf.restype = C.c_float
    L.orc_tf_idf.argtypes = [C.c_int, C.c_uint32, C.c_uint64]
    L.orc_topk.restype = C.c_size_t
    L.orc_topk.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_float),
                           C.c_size_t, C.c_size_t,
                           C.POINTER(C.c_uint64), C.POINTER(C.c_float)]
    L.orc_query_repr.restype = C.c_void_p
    L.orc_query_repr.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    L.orc_query_lex.restype = C.c_int
    L.orc_query_lex.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.c_size_t]
    L.orc_bkt_create.restype = C.c_void_p
    L.orc_bkt_destroy.argtypes = [C.c_void_p]
    L.orc_bkt_insert.restype = C.c_int
    L.orc_bkt_insert.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.orc_bkt_search.restype = C.c_size_t
    L.orc_bkt_search.argtypes = [C.c_void_p, C.c_uint, C.c_char_p, C.c_size_t,
                                 C.POINTER(C.c_uint32), C.c_size_t,
                                 C.POINTER(C.c_uint64)]
    L.orc_index_load.restype = C.c_void_p
    L.orc_index_load.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
    L.orc_index_free.argtypes = [C.c_void_p]
    L.orc_index_set_lowercase.argtypes = [C.c_void_p, C.c_bool]
    L.orc_index_set_stemmer.argtypes = [C.c_void_p, C.c_bool]
    L.orc_stem_en.restype = C.c_size_t
    L.orc_stem_en.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    for name, rt in (("orc_index_term_count", C.c_uint32),
                     ("orc_index_dt_count", C.c_uint64),
                     ("orc_index_doc_count", C.c_uint32),
                     ("orc_index_token_count", C.c_uint64),
                     ("orc_last_pairs", C.c_uint64)):
        getattr(L, name).restype = rt
        getattr(L, name).argtypes = [C.c_void_p]
    L.orc_index_lookup.restype = C.c_uint32
    L.orc_index_lookup.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.orc_index_fuzzy.restype = C.c_uint32
    L.orc_index_fuzzy.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t,
                                  C.POINTER(C.c_uint64)]
    L.orc_index_df.restype = C.c_uint64
    L.orc_index_df.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_index_term.restype = C.c_void_p
    L.orc_index_term.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_size_t)]
    L.orc_index_score.restype = C.c_float
    L.orc_index_score.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint64]
    L.orc_search.restype = C.c_int
    L.orc_search.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_uint64,
                             C.c_bool, C.POINTER(Result), C.c_size_t,
                             C.POINTER(C.c_uint32), C.POINTER(C.c_int),
                             C.c_char_p, C.c_size_t]
    L.orc_results_json.restype = C.c_void_p
    L.orc_results_json.argtypes = [C.POINTER(Result), C.c_size_t]
    _lib = L
    return L


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def _take_str(ptr):
    if not ptr:
        return None
    s = C.string_at(ptr).decode("utf-8", "surrogateescape")
    _libc.free(ptr)
    return s


def levdist(a: bytes, b: bytes) -> int:
    return lib().orc_levdist(a, len(a), b, len(b))


def query_repr(q):
    """-> (repr | None, errmsg | None)"""
    if isinstance(q, str):
        q = q.encode()
    err = C.c_void_p()
    r = lib().orc_query_repr(q, C.byref(err))
    return _take_str(r), _take_str(err.value)


def query_lex(q):
    if isinstance(q, str):
        q = q.encode()
    kinds = (C.c_int * 256)()
    n = lib().orc_query_lex(q, kinds, 256)
    return list(kinds[:n])


def topk(ids, scores, cap):
    n = len(ids)
    a = (C.c_uint64 * max(n, 1))(*ids)
    s = (C.c_float * max(n, 1))(*scores)
    oi = (C.c_uint64 * max(n, 1))()
    os_ = (C.c_float * max(n, 1))()
    cnt = lib().orc_topk(a, s, n, cap, oi, os_)
    return [(oi[i], os_[i]) for i in range(cnt)]


class BKTree:
    def __init__(self, words=()):
        self.h = lib().orc_bkt_create()
        for w in words:
            self.insert(w)

    def insert(self, w: bytes):
        return lib().orc_bkt_insert(self.h, w, len(w))

    def search(self, w: bytes, tol=2, cap=1 << 20):
        out = (C.c_uint32 * cap)()
        nd = C.c_uint64()
        n = lib().orc_bkt_search(self.h, tol, w, len(w), out, cap, C.byref(nd))
        return list(out[:min(n, cap)]), nd.value

    def close(self):
        if self.h:
            lib().orc_bkt_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class SearchError(Exception):
    def __init__(self, code, msg):
        super().__init__(f"[{code}] {msg}")
        self.code = code
        self.msg = msg


class Index:
    """Oracle-side index over an nxsterms/nxsdtmap file pair."""

    def __init__(self, terms_path, dtmap_path, lowercase=False, stemmer=False):
        err = C.create_string_buffer(256)
        self.h = lib().orc_index_load(os.fsencode(terms_path),
                                      os.fsencode(dtmap_path), err, 256)
        if not self.h:
            raise RuntimeError(err.value.decode())
        lib().orc_index_set_lowercase(self.h, lowercase)
        lib().orc_index_set_stemmer(self.h, stemmer)

    @property
    def term_count(self):
        return lib().orc_index_term_count(self.h)

    @property
    def dt_count(self):
        return lib().orc_index_dt_count(self.h)

    @property
    def doc_count(self):
        return lib().orc_index_doc_count(self.h)

    @property
    def token_count(self):
        return lib().orc_index_token_count(self.h)

    def lookup(self, tok: bytes):
        return lib().orc_index_lookup(self.h, tok, len(tok))

    def fuzzy(self, tok: bytes):
        v = C.c_uint64()
        t = lib().orc_index_fuzzy(self.h, tok, len(tok), C.byref(v))
        return t, v.value

    def df(self, term_id):
        return lib().orc_index_df(self.h, term_id)

    def term(self, term_id):
        n = C.c_size_t()
        p = lib().orc_index_term(self.h, term_id, C.byref(n))
        return C.string_at(p, n.value) if p else None

    def score(self, algo, term_id, doc_id):
        return lib().orc_index_score(self.h, algo, term_id, doc_id)

    def search(self, query, algo=BM25, limit=1000, fuzzymatch=True):
        if isinstance(query, str):
            query = query.encode()
        cap = min(limit, 1 << 20) if limit > 0 else 1
        out = (Result * cap)()
        cnt = C.c_uint32()
        code = C.c_int()
        err = C.create_string_buffer(512)
        r = lib().orc_search(self.h, query, algo, limit, fuzzymatch, out, cap,
                             C.byref(cnt), C.byref(code), err, 512)
        if r != 0:
            raise SearchError(code.value, err.value.decode("utf-8", "replace"))
        return [(out[i].doc_id, out[i].score) for i in range(min(cnt.value, cap))]

    @property
    def last_pairs(self):
        return lib().orc_last_pairs(self.h)

    def close(self):
        if self.h:
            lib().orc_index_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def results_json(results):
    n = len(results)
    arr = (Result * max(n, 1))()
    for i, (d, s) in enumerate(results):
        arr[i].doc_id = d
        arr[i].score = s
    return _take_str(lib().orc_results_json(arr, n))


# ---- genuine reference algo/ (oracle/_ref), optional -------------------

def ref():
    """libnxsref.so handle or None when it has not been built."""
    global _ref
    if _ref is not None:
        return _ref
    if not os.path.exists(REF_SO):
        if os.path.isdir("/root/reference/src/algo"):
            build_oracle()
        if not os.path.exists(REF_SO):
            return None
    R = C.CDLL(REF_SO)
    R.ref_levdist.restype = C.c_int
    R.ref_levdist.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    R.ref_bkt_build.restype = C.c_void_p
    R.ref_bkt_build.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.c_size_t]
    R.ref_bkt_ndist.restype = C.c_uint64
    R.ref_bkt_ndist.argtypes = [C.c_void_p]
    R.ref_bkt_search.restype = C.c_size_t
    R.ref_bkt_search.argtypes = [C.c_void_p, C.c_uint, C.c_char_p, C.c_size_t,
                                 C.POINTER(C.c_uint32), C.c_size_t,
                                 C.POINTER(C.c_uint64)]
    R.ref_bkt_destroy.argtypes = [C.c_void_p]
    R.ref_topk.restype = C.c_size_t
    R.ref_topk.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_float),
                           C.c_size_t, C.c_size_t,
                           C.POINTER(C.c_uint64), C.POINTER(C.c_float)]
    _ref = R
    return R


class RefBKTree:
    """BK-tree built and searched by the reference's own bktree.c/levdist.c."""

    def __init__(self, words):
        self.blob = b"".join(words)
        offs = [0]
        for w in words:
            offs.append(offs[-1] + len(w))
        self.offs = (C.c_uint32 * len(offs))(*offs)
        self.h = ref().ref_bkt_build(self.blob, self.offs, len(words))
        self.build_ndist = ref().ref_bkt_ndist(self.h)

    def search(self, w: bytes, tol=2, cap=1 << 20):
        out = (C.c_uint32 * cap)()
        nd = C.c_uint64()
        n = ref().ref_bkt_search(self.h, tol, w, len(w), out, cap, C.byref(nd))
        return list(out[:min(n, cap)]), nd.value

    def close(self):
        if self.h:
            ref().ref_bkt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ref_topk(ids, scores, cap):
    n = len(ids)
    a = (C.c_uint64 * max(n, 1))(*ids)
    s = (C.c_float * max(n, 1))(*scores)
    oi = (C.c_uint64 * max(n, 1))()
    os_ = (C.c_float * max(n, 1))()
    cnt = ref().ref_topk(a, s, n, cap, oi, os_)
    return [(oi[i], os_[i]) for i in range(cnt)]

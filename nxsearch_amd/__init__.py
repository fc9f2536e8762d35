"""nxsearch_amd -- MI355X-native query/ranking path behind nxsearch's C API.

Thin ctypes binding over ``csrc/libnxsearch_gpu.so`` (C11 host code + HIP
kernels for gfx950).  The class layout mirrors the reference's Lua binding
(`nxs.open(basedir)`, `index:search(query, params)`, reference
src/core/lua.c:341-366) and, underneath, its C API (include/nxs.h).

There is no CPU fallback: importing works anywhere, but opening an index
without a HIP device raises, and a missing shared library raises on import of
the binding (`lib()`).
"""
import ctypes as C
import os

__all__ = ["Nxs", "Index", "NxsError", "lib", "build", "LIB_PATH"]

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("NXS_GPU_LIB") or os.path.join(CSRC, "libnxsearch_gpu.so")
SYNTH_PATH = os.path.join(CSRC, "libnxssynth.so")

MAX_TOKENS, MAX_PROG, FAST_K = 32, 256, 64
TF_IDF, BM25 = 0, 1
ERR_NAMES = ["SUCCESS", "FATAL", "SYSTEM", "INVALID", "EXISTS", "MISSING", "LIMIT"]


class NxsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("NXS_ERR_%s: %s" % (ERR_NAMES[code] if 0 <= code < 7 else code, msg))
        self.code = code
        self.msg = msg


class GpuQuery(C.Structure):
    """nxsgpu_query_t (include/nxs_gpu.h)"""
    _fields_ = [("n_tokens", C.c_uint32),
                ("term_id", C.c_uint32 * MAX_TOKENS),
                ("prog_len", C.c_uint32),
                ("prog", C.c_uint8 * MAX_PROG),
                ("truth", C.c_uint32 * 8)]


class GpuBkNode(C.Structure):
    """nxsgpu_bknode_t"""
    _fields_ = [("bitmap", C.c_uint64), ("first_child", C.c_uint32),
                ("term_id", C.c_uint32), ("str_off", C.c_uint32),
                ("str_len", C.c_uint16), ("flags", C.c_uint16),
                ("inl", C.c_uint8 * 8)]


class BkImage(C.Structure):
    """nxs_bkimage_t (csrc/nxs_impl.h)"""
    _fields_ = [("nodes", C.POINTER(GpuBkNode)), ("n", C.c_uint32),
                ("depth", C.c_uint32), ("bytes", C.POINTER(C.c_uint8)),
                ("bytes_len", C.c_uint64)]


class GpuResults(C.Structure):
    """nxsgpu_results_t"""
    _fields_ = [("n_queries", C.c_uint32), ("counts", C.POINTER(C.c_uint32)),
                ("offsets", C.POINTER(C.c_uint64)), ("doc_ids", C.POINTER(C.c_uint64)),
                ("scores", C.POINTER(C.c_float)), ("postings", C.c_uint64),
                ("candidates", C.c_uint64), ("exact_requeries", C.c_uint32)]


class GpuProfile(C.Structure):
    """nxsgpu_profile_t"""
    _fields_ = [("launches", C.c_uint64), ("scan_ms", C.c_double),
                ("replay_ms", C.c_double), ("fuzzy_ms", C.c_double),
                ("postings", C.c_uint64), ("fuzzy_visits", C.c_uint64),
                ("fuzzy_pairs", C.c_uint64), ("fuzzy_level", C.c_uint64 * 40),
                ("fuzzy_filter_ms", C.c_double), ("fuzzy_dist_ms", C.c_double),
                ("fuzzy_chain_ms", C.c_double), ("fuzzy_checked", C.c_uint64),
                ("n_cls", C.c_uint32), ("cls_key", C.c_uint32 * 16), ("cls_launches", C.c_uint64 * 16),
                ("cls_ms", C.c_double * 16), ("cls_postings", C.c_uint64 * 16), ("cls_queries", C.c_uint64 * 16)]


# every symbol include/nxs.h and include/nxs_gpu.h declare
NXS_H_SYMBOLS = [
    "nxs_open", "nxs_close", "nxs_get_error", "nxs_params_create", "nxs_params_fromjson",
    "nxs_params_set_str", "nxs_params_set_uint", "nxs_params_set_bool",
    "nxs_params_release", "nxs_index_open", "nxs_index_close",
    "nxs_index_search", "nxs_resp_iter_reset", "nxs_resp_iter_result",
    "nxs_resp_resultcount", "nxs_resp_tojson", "nxs_resp_release",
    "nxs_index_search_batch", "nxs_index_open_files",
    "nxs_index_plan_batch", "nxs_index_search_batch_begin",
    "nxs_index_search_batch_end", "nxs_shard_unique_id", "nxs_index_shard",
    "nxs_index_shard_local", "nxs_index_shard_slice",
    "nxs_index_open_shard", "nxs_docshard_search_batch",
    "nxs_docshard_attach", "nxs_docshard_search_batch_rank",
]
# csrc/nxs_hooks.h: test hooks + bench accessors, only in builds with -DNXS_TEST_HOOKS (the default)
NXS_HOOK_SYMBOLS = ["nxs_index_device", "nxs_index_host_profile", "nxs_index_shard_info", "nxs_test_pool", "nxs_test_assemble",
                    "nxs_test_fixup_scan", "nxs_test_inject_failure"]
NXS_GPU_H_SYMBOLS = [
    "nxsgpu_device_count", "nxsgpu_last_error", "nxsgpu_index_create",
    "nxsgpu_index_destroy", "nxsgpu_index_df", "nxsgpu_index_postings",
    "nxsgpu_index_docs", "nxsgpu_index_first_bad_doc", "nxsgpu_search",
    "nxsgpu_results_free", "nxsgpu_search_dev", "nxsgpu_search_dev_begin",
    "nxsgpu_search_dev_end", "nxsgpu_fuzzy", "nxsgpu_fuzzy_begin", "nxsgpu_fuzzy_end",
    "nxsgpu_set_profiling", "nxsgpu_get_profile", "nxsgpu_synchronize",
    "nxsgpu_search_wide", "nxsgpu_shard_slice", "nxsgpu_shard_capacity",
    "nxsgpu_comm_unique_id", "nxsgpu_comm_create", "nxsgpu_comm_destroy",
    "nxsgpu_comm_rank", "nxsgpu_comm_world", "nxsgpu_comm_rccl_count", "nxsgpu_comm_stats", "nxsgpu_comm_allgather",
    "nxsgpu_index_set_comm", "nxsgpu_batch_begin", "nxsgpu_batch_end",
    "nxsgpu_batches_in_flight", "nxsgpu_index_reconfigure", "nxsgpu_index_set_parallel", "nxsgpu_hbm_read_gbs",
    "nxsgpu_hbm_calibrate",
    "nxsgpu_index_apply", "nxsgpu_index_set_bk", "nxsgpu_index_set_global_df",
    "nxsgpu_search_candidates", "nxsgpu_merge_candidates",
]

_lib = None


def build():
    """(Re)build the shared libraries in-tree with make + hipcc."""
    import subprocess
    subprocess.run(["make", "-C", CSRC], check=True, stdout=subprocess.DEVNULL)


def lib():
    """The loaded libnxsearch_gpu.so; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `make -C %s` (hipcc, gfx950). "
            "There is no CPU fallback for the query path." % (LIB_PATH, CSRC))
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, cp = C.c_void_p, C.c_char_p
    L.nxs_open.restype = vp
    L.nxs_open.argtypes = [cp]
    L.nxs_close.argtypes = [vp]
    L.nxs_get_error.restype = C.c_int
    L.nxs_get_error.argtypes = [vp, C.POINTER(cp)]
    L.nxs_params_create.restype = vp
    L.nxs_params_fromjson.restype = vp
    L.nxs_params_fromjson.argtypes = [vp, cp, C.c_size_t]
    L.nxs_params_get_str.restype = cp
    L.nxs_params_get_str.argtypes = [vp, cp]
    L.nxs_params_get_uint.argtypes = [vp, cp, C.POINTER(C.c_uint64)]
    L.nxs_params_get_bool.argtypes = [vp, cp, C.POINTER(C.c_bool)]
    L.nxs_params_set_str.argtypes = [vp, cp, cp]
    L.nxs_params_set_uint.argtypes = [vp, cp, C.c_uint64]
    L.nxs_params_set_bool.argtypes = [vp, cp, C.c_bool]
    L.nxs_params_release.argtypes = [vp]
    L.nxs_index_open.restype = vp
    L.nxs_index_open.argtypes = [vp, cp]
    L.nxs_index_open_files.restype = vp
    L.nxs_index_open_files.argtypes = [vp, cp, cp, cp, C.c_bool]
    L.nxs_index_close.argtypes = [vp]
    L.nxs_index_device.restype = vp
    L.nxs_index_device.argtypes = [vp]
    L.nxs_index_search.restype = vp
    L.nxs_index_search.argtypes = [vp, vp, cp, C.c_size_t]
    L.nxs_index_search_batch.restype = C.c_int
    L.nxs_index_search_batch.argtypes = [vp, vp, C.POINTER(cp), C.c_size_t,
                                         C.POINTER(vp), C.POINTER(C.c_int)]
    L.nxs_index_search_batch_begin.restype = C.c_int
    L.nxs_index_search_batch_begin.argtypes = [vp, vp, C.POINTER(cp), C.c_size_t]
    L.nxs_index_search_batch_end.restype = C.c_int
    L.nxs_index_search_batch_end.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int)]
    L.nxs_shard_unique_id.restype = C.c_int
    L.nxs_shard_unique_id.argtypes = [vp, C.c_char_p]
    L.nxs_index_shard.restype = C.c_int
    L.nxs_index_shard.argtypes = [vp, C.c_int, C.c_int, C.c_char_p]
    L.nxs_index_plan_batch.restype = C.c_int
    L.nxs_index_plan_batch.argtypes = [vp, vp, C.POINTER(cp), C.c_size_t,
                                       C.POINTER(GpuQuery), C.POINTER(C.c_int)]
    L.nxs_resp_iter_reset.argtypes = [vp]
    L.nxs_resp_iter_result.restype = C.c_bool
    L.nxs_resp_iter_result.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_float)]
    L.nxs_resp_resultcount.restype = C.c_uint
    L.nxs_resp_resultcount.argtypes = [vp]
    L.nxs_resp_tojson.restype = vp
    L.nxs_resp_tojson.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.nxs_resp_release.argtypes = [vp]
    # device shim
    L.nxsgpu_device_count.restype = C.c_int
    L.nxsgpu_last_error.restype = cp
    L.nxsgpu_index_df.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.nxsgpu_index_postings.restype = C.c_uint64
    L.nxsgpu_index_postings.argtypes = [vp]
    L.nxsgpu_index_docs.restype = C.c_uint64
    L.nxsgpu_index_docs.argtypes = [vp]
    L.nxsgpu_search.restype = C.c_int
    L.nxsgpu_search.argtypes = [vp, C.c_int, C.c_uint64, C.POINTER(GpuQuery),
                                C.c_uint32, C.POINTER(GpuResults)]
    L.nxsgpu_results_free.argtypes = [C.POINTER(GpuResults)]
    L.nxsgpu_search_dev.restype = C.c_int
    L.nxsgpu_search_dev.argtypes = [vp, C.c_int, C.c_uint32, C.POINTER(GpuQuery),
                                    C.c_uint32, vp, vp, vp]
    L.nxsgpu_search_dev_begin.restype = C.c_int
    L.nxsgpu_search_dev_begin.argtypes = L.nxsgpu_search_dev.argtypes
    L.nxsgpu_search_dev_end.restype = C.c_int
    L.nxsgpu_search_dev_end.argtypes = [vp]
    L.nxsgpu_fuzzy.restype = C.c_int
    L.nxsgpu_fuzzy.argtypes = [vp, cp, C.POINTER(C.c_uint32), C.c_uint32,
                               C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    L.nxsgpu_shard_slice.argtypes = [C.c_uint64, C.c_int, C.c_int,
                                     C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.nxsgpu_shard_capacity.restype = C.c_uint64
    L.nxsgpu_shard_capacity.argtypes = [C.c_uint64, C.c_int]
    L.nxsgpu_batches_in_flight.restype = C.c_int
    L.nxsgpu_batches_in_flight.argtypes = [vp]
    L.nxsgpu_index_reconfigure.argtypes = [vp]
    L.nxsgpu_hbm_read_gbs.restype = C.c_double
    L.nxsgpu_hbm_read_gbs.argtypes = [vp, C.c_int]
    L.nxsgpu_set_profiling.argtypes = [vp, C.c_int]
    L.nxsgpu_get_profile.argtypes = [vp, C.POINTER(GpuProfile), C.c_int]
    L.nxsgpu_synchronize.argtypes = [vp]
    # host-only test hooks
    L.nxs_test_query_repr.restype = vp
    L.nxs_test_query_repr.argtypes = [cp, C.POINTER(vp)]
    L.nxs_query_lex.restype = C.c_int
    L.nxs_query_lex.argtypes = [cp, C.POINTER(C.c_int), C.c_size_t]
    L.nxs_test_compile.restype = C.c_int
    L.nxs_test_compile.argtypes = [cp, C.POINTER(cp), C.c_uint32, C.c_bool,
                                   C.POINTER(GpuQuery), C.POINTER(C.c_int),
                                   cp, C.c_size_t]
    L.nxs_test_compile_wide.restype = C.c_int
    L.nxs_test_compile_wide.argtypes = [cp, C.POINTER(cp), C.c_uint32, C.POINTER(C.c_int),
                                        C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32,
                                        C.POINTER(C.c_uint32), C.POINTER(C.c_uint16), C.c_uint32]
    L.nxs_test_bk_image.restype = C.c_int
    L.nxs_test_bk_image.argtypes = [C.POINTER(cp), C.c_uint32, C.POINTER(BkImage)]
    L.nxs_bk_free.argtypes = [C.POINTER(BkImage)]
    L.nxs_test_levdist.restype = C.c_int
    L.nxs_test_levdist.argtypes = [cp, C.c_size_t, cp, C.c_size_t]
    _lib = L
    return L


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def _take(ptr):
    if not ptr:
        return None
    s = C.string_at(ptr).decode("utf-8", "surrogateescape")
    _libc.free(ptr)
    return s


def _b(s):
    return s.encode("utf-8", "surrogateescape") if isinstance(s, str) else s


class Nxs:
    """`nxs_t`: a library instance bound to a base directory."""

    def __init__(self, basedir):
        self._h = lib().nxs_open(os.fsencode(basedir))
        if not self._h:
            raise NxsError(2, "nxs_open(%r) failed" % (basedir,))

    def error(self):
        msg = C.c_char_p()
        code = lib().nxs_get_error(self._h, C.byref(msg))
        return code, (msg.value or b"").decode("utf-8", "replace")

    def _raise(self):
        raise NxsError(*self.error())

    def open_index(self, name):
        h = lib().nxs_index_open(self._h, _b(name))
        if not h:
            self._raise()
        return Index(self, h)

    def open_files(self, terms_path, dtmap_path, algo="BM25", lowercase=False):
        h = lib().nxs_index_open_files(self._h, os.fsencode(terms_path),
                                       os.fsencode(dtmap_path), _b(algo), lowercase)
        if not h:
            self._raise()
        return Index(self, h)

    def open_shard(self, terms_path, dtmap_path, shard, n_shards, algo="BM25", lowercase=False,
                   device=-1):
        """nxs_index_open_shard(): shard `shard` of a doc-sharded collection (N4)."""
        L = lib()
        L.nxs_index_open_shard.restype = C.c_void_p
        L.nxs_index_open_shard.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_bool,
                                           C.c_uint, C.c_uint, C.c_int]
        h = L.nxs_index_open_shard(self._h, os.fsencode(terms_path), os.fsencode(dtmap_path),
                                   _b(algo), lowercase, shard, n_shards, device)
        if not h:
            self._raise()
        return Index(self, h)

    def docshard_search_batch(self, shards, queries, limit=None, algo=None, fuzzymatch=None):
        """nxs_docshard_search_batch(): one batch over all shards, merged exactly."""
        L = lib()
        L.nxs_docshard_search_batch.restype = C.c_int
        L.nxs_docshard_search_batch.argtypes = [C.POINTER(C.c_void_p), C.c_uint, C.c_void_p,
                                                C.POINTER(C.c_char_p), C.c_size_t,
                                                C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
        n = len(queries)
        hs = (C.c_void_p * len(shards))(*[s._h for s in shards])
        qs = (C.c_char_p * max(n, 1))(*[_b(q) for q in queries])
        resps = (C.c_void_p * max(n, 1))()
        errs = (C.c_int * max(n, 1))()
        p = _make_params(limit, algo, fuzzymatch)
        try:
            r = L.nxs_docshard_search_batch(hs, len(shards), p, qs, n, resps, errs)
        finally:
            if p:
                L.nxs_params_release(p)
        if r < 0:
            self._raise()
        out = []
        for i in range(n):
            if resps[i]:
                out.append(_drain(resps[i]))
                L.nxs_resp_release(resps[i])
            else:
                out.append(NxsError(errs[i], "query %d failed" % i))
        return out

    def _collect(self, r, n, resps, errs):
        L = lib()
        if r < 0:
            self._raise()
        out = []
        for i in range(n):
            if resps[i]:
                out.append(_drain(resps[i]))
                L.nxs_resp_release(resps[i])
            else:
                out.append(NxsError(errs[i], "query %d failed" % i))
        return out

    def docshard_search_batch_rank(self, shard, queries, limit=None, algo=None, fuzzymatch=None):
        """nxs_docshard_search_batch_rank(): this rank's shard + one all-gather + merge."""
        L = lib()
        L.nxs_docshard_search_batch_rank.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_char_p), C.c_size_t,
                                                     C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
        n = len(queries)
        qs = (C.c_char_p * max(n, 1))(*[_b(q) for q in queries])
        resps, errs = (C.c_void_p * max(n, 1))(), (C.c_int * max(n, 1))()
        p = _make_params(limit, algo, fuzzymatch)
        try:
            r = L.nxs_docshard_search_batch_rank(shard._h, p, qs, n, resps, errs)
        finally:
            if p:
                L.nxs_params_release(p)
        return self._collect(r, n, resps, errs)

    def docshard_attach(self, shard):
        """nxs_docshard_attach(): collective; collection-wide df for this rank's shard."""
        L = lib()
        L.nxs_docshard_attach.argtypes = [C.c_void_p]
        if L.nxs_docshard_attach(shard._h) != 0:
            self._raise()

    def docshard_emulated_ranks(self, shards, queries, cap=512, limit=None, algo=None, fuzzymatch=None):
        """tests: the one-process-per-shard form with the ranks played one after the
        other on this GPU -- every rank's candidate block (nxs_test_docshard_block),
        the blocks concatenated as the all-gather would, then every rank's merge
        (nxs_test_docshard_finish).  -> one result list per rank."""
        L = lib()
        vp = C.c_void_p
        L.nxs_test_docshard_set_df.argtypes = [C.POINTER(vp), C.c_uint]
        L.nxs_test_docshard_block.argtypes = [vp, vp, C.POINTER(C.c_char_p), C.c_size_t, C.c_uint32,
                                              C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        L.nxs_test_docshard_finish.argtypes = [vp, vp, C.POINTER(C.c_char_p), C.c_size_t, C.c_uint32,
                                               C.c_char_p, C.POINTER(vp), C.POINTER(C.c_int)]
        libc = C.CDLL(None)
        libc.free.argtypes = [C.c_void_p]
        hs = (vp * len(shards))(*[s._h for s in shards])
        if L.nxs_test_docshard_set_df(hs, len(shards)) != 0:
            self._raise()
        n = len(queries)
        qs = (C.c_char_p * max(n, 1))(*[_b(q) for q in queries])
        p = _make_params(limit, algo, fuzzymatch)
        try:
            gathered = b""
            for sh in shards:
                blk, ln = C.POINTER(C.c_uint8)(), C.c_size_t()
                if L.nxs_test_docshard_block(sh._h, p, qs, n, cap, C.byref(blk), C.byref(ln)) != 0:
                    self._raise()
                gathered += C.string_at(blk, ln.value)
                libc.free(blk)
            outs = []
            for sh in shards:
                resps, errs = (vp * max(n, 1))(), (C.c_int * max(n, 1))()
                r = L.nxs_test_docshard_finish(sh._h, p, qs, n, cap, gathered, resps, errs)
                outs.append(self._collect(r, n, resps, errs))
        finally:
            if p:
                L.nxs_params_release(p)
        return outs

    def shard_unique_id(self):
        """nxs_shard_unique_id(): the bytes rank 0 hands to the other ranks."""
        buf = C.create_string_buffer(128)
        if lib().nxs_shard_unique_id(self._h, buf) != 0:
            self._raise()
        return buf.raw

    def close(self):
        if self._h:
            lib().nxs_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def _make_params(limit=None, algo=None, fuzzymatch=None):
    if limit is None and algo is None and fuzzymatch is None:
        return None
    L = lib()
    p = L.nxs_params_create()
    if limit is not None:
        L.nxs_params_set_uint(p, b"limit", limit)
    if algo is not None:
        L.nxs_params_set_str(p, b"algo", _b(algo))
    if fuzzymatch is not None:
        L.nxs_params_set_bool(p, b"fuzzymatch", bool(fuzzymatch))
    return p


def _drain(resp):
    L = lib()
    out = []
    d, s = C.c_uint64(), C.c_float()
    L.nxs_resp_iter_reset(resp)
    while L.nxs_resp_iter_result(resp, C.byref(d), C.byref(s)):
        out.append((d.value, s.value))
    assert len(out) == L.nxs_resp_resultcount(resp)
    return out


class Index:
    """`nxs_index_t` opened for searching on the GPU."""

    def __init__(self, nxs, h):
        self.nxs = nxs
        self._h = h

    @property
    def device(self):
        return lib().nxs_index_device(self._h)

    def search(self, query, limit=None, algo=None, fuzzymatch=None, json=False, params_json=None):
        """nxs_index_search(): -> [(doc_id, score), ...] (or the JSON text).
        params_json: the parameters as the Lua binding passes them (nxs_params_fromjson)."""
        L = lib()
        if params_json is not None:
            pj = _b(params_json)
            p = L.nxs_params_fromjson(self.nxs._h, pj, len(pj))
            if not p:
                self.nxs._raise()
        else:
            p = _make_params(limit, algo, fuzzymatch)
        q = _b(query)
        try:
            resp = L.nxs_index_search(self._h, p, q, len(q))
        finally:
            if p:
                L.nxs_params_release(p)
        if not resp:
            self.nxs._raise()
        try:
            if json:
                n = C.c_size_t()
                return _take(L.nxs_resp_tojson(resp, C.byref(n)))
            return _drain(resp)
        finally:
            L.nxs_resp_release(resp)

    def search_batch(self, queries, limit=None, algo=None, fuzzymatch=None):
        """nxs_index_search_batch(): list of result lists; a failed query
        yields an NxsError instance in its slot."""
        L = lib()
        n = len(queries)
        qs = (C.c_char_p * n)(*[_b(q) for q in queries])
        resps = (C.c_void_p * n)()
        errs = (C.c_int * n)()
        p = _make_params(limit, algo, fuzzymatch)
        try:
            r = L.nxs_index_search_batch(self._h, p, qs, n, resps, errs)
        finally:
            if p:
                L.nxs_params_release(p)
        if r < 0:
            self.nxs._raise()
        out = []
        for i in range(n):
            if resps[i]:
                out.append(_drain(resps[i]))
                L.nxs_resp_release(resps[i])
            else:
                out.append(NxsError(errs[i], "query %d failed" % i))
        return out

    def search_batch_begin(self, queries, limit=None, algo=None, fuzzymatch=None):
        """nxs_index_search_batch_begin(): queue a batch (at most NXS_BATCHES_INFLIGHT = 4 in flight)."""
        L = lib()
        n = len(queries)
        qs = (C.c_char_p * max(n, 1))(*[_b(q) for q in queries])
        p = _make_params(limit, algo, fuzzymatch)
        try:
            r = L.nxs_index_search_batch_begin(self._h, p, qs, n)
        finally:
            if p:
                L.nxs_params_release(p)
        if r != 0:
            self.nxs._raise()
        self._pending = getattr(self, "_pending", []) + [n]

    def search_batch_end(self):
        """nxs_index_search_batch_end(): the oldest batch's result lists."""
        L = lib()
        pend = getattr(self, "_pending", [])
        n = pend[0] if pend else 1
        resps = (C.c_void_p * max(n, 1))()
        errs = (C.c_int * max(n, 1))()
        r = L.nxs_index_search_batch_end(self._h, resps, errs)
        if r < 0:
            self.nxs._raise()
        self._pending = pend[1:]
        out = []
        for i in range(n):
            if resps[i]:
                out.append(_drain(resps[i]))
                L.nxs_resp_release(resps[i])
            else:
                out.append(NxsError(errs[i], "query %d failed" % i))
        return out

    def shard(self, rank, world, uid):
        """nxs_index_shard(): collective; `uid` from shard_unique_id() of rank 0."""
        if lib().nxs_index_shard(self._h, rank, world, uid) != 0:
            self.nxs._raise()

    def shard_local(self, on=True):
        """nxs_index_shard_local(): responses of the own slice only (others None)."""
        L = lib()
        L.nxs_index_shard_local.argtypes = [C.c_void_p, C.c_bool]
        if L.nxs_index_shard_local(self._h, on) != 0:
            self.nxs._raise()

    def shard_slice(self, n):
        """nxs_index_shard_slice(): the part of an n-query batch whose responses this index delivers."""
        lo, hi = C.c_size_t(), C.c_size_t()
        L = lib()
        L.nxs_index_shard_slice.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.nxs_index_shard_slice(self._h, n, C.byref(lo), C.byref(hi))
        return lo.value, hi.value

    def shard_info(self):
        """What the attached communicator is and has carried (bench evidence): RCCL's own rank count
        (ncclCommCount), the library's world, all-gathers queued, bytes this rank contributed."""
        out = (C.c_uint64 * 4)()
        L = lib()
        L.nxs_index_shard_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.nxs_index_shard_info(self._h, out)
        return {"rccl_ranks": int(C.c_int64(out[0]).value), "world": int(out[1]), "allgathers": int(out[2]),
                "bytes_contributed": int(out[3])}

    def host_profile(self):
        """nxs_index_host_profile(): per-batch host phase times in ms."""
        out = (C.c_double * 12)()
        L = lib()
        L.nxs_index_host_profile.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.nxs_index_host_profile(self._h, out)
        n = max(out[4], 1.0)
        return {"plan_ms": round(1e3 * out[0] / n, 4), "queue_ms": round(1e3 * out[1] / n, 4),
                "wait_ms": round(1e3 * out[2] / n, 4), "resps_ms": round(1e3 * out[3] / n, 4),
                "begin_ms": round(1e3 * out[6] / n, 4), "end_ms": round(1e3 * out[7] / n, 4),
                "fuzzy_wait_ms": round(1e3 * out[8] / n, 4), "front_ms": round(1e3 * out[9] / n, 4),
                "fuzzy_launch_ms": round(1e3 * out[10] / n, 4), "back_ms": round(1e3 * out[11] / n, 4),
                "batches": int(out[4]), "exact_requeries": int(out[5])}

    def reconfigure(self):
        """Re-read the NXS_GPU_* switches (parsed once at open); tests/tools."""
        lib().nxsgpu_index_reconfigure(self.device)

    def plan_batch(self, queries, limit=None, algo=None, fuzzymatch=None):
        """nxs_index_plan_batch(): -> (ctypes array of GpuQuery, [err codes])."""
        L = lib()
        n = len(queries)
        qs = (C.c_char_p * n)(*[_b(q) for q in queries])
        plans = (GpuQuery * max(n, 1))()
        errs = (C.c_int * max(n, 1))()
        p = _make_params(limit, algo, fuzzymatch)
        try:
            r = L.nxs_index_plan_batch(self._h, p, qs, n, plans, errs)
        finally:
            if p:
                L.nxs_params_release(p)
        if r < 0:
            self.nxs._raise()
        return plans, list(errs[:n])

    def search_dev(self, plans, n, limit, algo, d_ids, d_scores, d_counts):
        """nxsgpu_search_dev(): results stay on the device at the given raw
        device pointers ([n][limit] u64 / f32, [n] u32).  Returns 0, or 1 if
        some query needs the exact two-pass path."""
        r = lib().nxsgpu_search_dev(self.device, algo, limit, plans, n,
                                    d_ids, d_scores, d_counts)
        if r < 0:
            raise NxsError(1, lib().nxsgpu_last_error().decode())
        return r

    def search_dev_begin(self, plans, n, limit, algo, d_ids, d_scores, d_counts):
        """nxsgpu_search_dev_begin(): queue a batch (at most NXS_BATCHES_INFLIGHT = 4 in flight)."""
        if lib().nxsgpu_search_dev_begin(self.device, algo, limit, plans, n,
                                         d_ids, d_scores, d_counts) != 0:
            raise NxsError(1, lib().nxsgpu_last_error().decode())

    def search_dev_end(self):
        """nxsgpu_search_dev_end(): wait for the oldest batch in flight -> 0 / 1."""
        r = lib().nxsgpu_search_dev_end(self.device)
        if r < 0:
            raise NxsError(1, lib().nxsgpu_last_error().decode())
        return r

    def fuzzy(self, tokens, want_visited=False):
        """Device BK-tree search for raw tokens -> term ids (0 = none)."""
        L = lib()
        toks = [_b(t) for t in tokens]
        blob = b"".join(toks)
        offs = [0]
        for t in toks:
            offs.append(offs[-1] + len(t))
        n = len(toks)
        ids = (C.c_uint32 * max(n, 1))()
        vis = (C.c_uint64 * max(n, 1))() if want_visited else None
        L.nxs_index_bk_sync.argtypes = [C.c_void_p]
        L.nxs_index_bk_sync(self._h)        # terms appended since the last image (N1)
        r = L.nxsgpu_fuzzy(self.device, blob, (C.c_uint32 * (n + 1))(*offs), n, ids, vis)
        if r != 0:
            raise NxsError(1, L.nxsgpu_last_error().decode())
        if want_visited:
            return list(ids[:n]), list(vis[:n])
        return list(ids[:n])

    def set_plan_cache(self, on=True):
        """bench: the index's plan cache (query string -> compiled plan) on / off."""
        L = lib()
        L.nxs_index_set_plan_cache.argtypes = [C.c_void_p, C.c_int]
        L.nxs_index_set_plan_cache(self._h, 1 if on else 0)

    def set_profiling(self, on=True):
        lib().nxsgpu_set_profiling(self.device, 1 if on else 0)

    def profile(self, reset=False):
        p = GpuProfile()
        lib().nxsgpu_get_profile(self.device, C.byref(p), 1 if reset else 0)
        d = {k: getattr(p, k) for k, _ in GpuProfile._fields_}
        d["fuzzy_level"] = list(d["fuzzy_level"])
        n = d.pop("n_cls")
        cls = [{"key": p.cls_key[i], "launches": p.cls_launches[i], "ms": p.cls_ms[i],
                "postings": p.cls_postings[i], "queries": p.cls_queries[i]} for i in range(n)]
        for k in ("cls_key", "cls_launches", "cls_ms", "cls_postings", "cls_queries"):
            d.pop(k)
        d["classes"] = cls
        return d

    def close(self):
        if self._h:
            lib().nxs_index_close(self._h)
            self._h = None


# ---- host-only helpers (run without a GPU; used by the CPU test tier) ------

def query_repr(q):
    err = C.c_void_p()
    r = lib().nxs_test_query_repr(_b(q), C.byref(err))
    return _take(r), _take(err.value)


def query_lex(q):
    kinds = (C.c_int * 512)()
    n = lib().nxs_query_lex(_b(q), kinds, 512)
    return list(kinds[:n])


def compile_query(q, words, lowercase=False):
    """-> (code, errmsg, empty, GpuQuery) against a word list (ids 1..n)."""
    arr = (C.c_char_p * max(len(words), 1))(*[_b(w) for w in words])
    plan = GpuQuery()
    empty = C.c_int()
    err = C.create_string_buffer(256)
    code = lib().nxs_test_compile(_b(q), arr, len(words), lowercase,
                                  C.byref(plan), C.byref(empty), err, 256)
    return code, err.value.decode(), bool(empty.value), plan


def filter_token(s, basedir=None, stopwords=False, stemmer=False):
    """The query-token filter pipeline on one string -> (action, result):
    action 1 keep, 0 discarded (stop word), -1 error."""
    L = lib()
    L.nxs_test_filter.restype = C.c_void_p
    L.nxs_test_filter.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.POINTER(C.c_int)]
    act = C.c_int()
    r = L.nxs_test_filter(os.fsencode(basedir) if basedir else None, (1 if stopwords else 0) | (2 if stemmer else 0),
                          _b(s), C.byref(act))
    return act.value, _take(r)


def compile_wide(q, words):
    """-> (code, wide, term_ids, prog) for a query over words "w1".."wN"."""
    arr = (C.c_char_p * max(len(words), 1))(*[_b(w) for w in words])
    wide, nt, pl = C.c_int(), C.c_uint32(), C.c_uint32()
    tids = (C.c_uint32 * 4096)()
    prog = (C.c_uint16 * 8192)()
    code = lib().nxs_test_compile_wide(_b(q), arr, len(words), C.byref(wide), C.byref(nt),
                                       tids, 4096, C.byref(pl), prog, 8192)
    return code, bool(wide.value), list(tids[:nt.value]), list(prog[:pl.value])


def bk_image(words):
    """Flattened BK-tree of a word list -> list of node dicts in BFS order."""
    arr = (C.c_char_p * max(len(words), 1))(*[_b(w) for w in words])
    img = BkImage()
    if lib().nxs_test_bk_image(arr, len(words), C.byref(img)) != 0:
        raise MemoryError
    nodes = []
    for i in range(img.n):
        nd = img.nodes[i]
        s = bytes(img.bytes[nd.str_off:nd.str_off + nd.str_len])
        nodes.append(dict(bitmap=nd.bitmap, first_child=nd.first_child,
                          term_id=nd.term_id, term=s, flags=nd.flags,
                          inl=bytes(nd.inl)))
    depth = img.depth
    lib().nxs_bk_free(C.byref(img))
    return nodes, depth


def levdist(a, b):
    a, b = _b(a), _b(b)
    return lib().nxs_test_levdist(a, len(a), b, len(b))

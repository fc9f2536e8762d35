"""Query-sharded multi-GPU execution -- a binding of the
library's own sharding (include/nxs.h: nxs_shard_unique_id, nxs_index_shard).

Queries are independent and the index is read-only during a batch, so a batch
shards BY QUERY: rank r of W takes the contiguous slice
[n*r//W, n*(r+1)//W), every rank holds a full replica of the device index (N,
token_count and df are identical on all replicas, so no statistics are
exchanged), and ONE RCCL all-gather of the ranks' record blocks -- fixed-size
per-query records `count u32 | flags u32 | k x u64 doc id | k x f32 score`,
128 B at k = 10 -- reassembles the batch on every rank over xGMI.  All of it
lives behind the C ABI (csrc/nxs_api.c, nxs_gpu.hip); this module only
distributes the communicator's unique id with whatever channel the
application already has (here: torch.distributed) and wraps the test hooks.
The reference has no counterpart: it scales by running independent worker
processes (compose/nginx.conf:2).
"""
import ctypes as C

from . import lib, NxsError


def shard_slice(n, rank, world):
    """Contiguous slice of an n-query batch owned by `rank` (nxsgpu_shard_slice)."""
    lo, hi = C.c_uint64(), C.c_uint64()
    lib().nxsgpu_shard_slice(n, rank, world, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def shard_capacity(n, world):
    """Largest slice = record slots per rank's block (nxsgpu_shard_capacity)."""
    return lib().nxsgpu_shard_capacity(n, world)


def rec_bytes(k):
    return (8 + 12 * k + 7) & ~7


def block_bytes(n_slots, k):
    return n_slots * rec_bytes(k) + (((n_slots + 1) * 4 + 7) & ~7)


def attach(nxs, index, rank, world, dist=None, device=None):
    """Collective: rank 0 creates the RCCL unique id, torch.distributed carries
    its 128 bytes to the other ranks, every rank builds the communicator."""
    import torch
    if world <= 1 and dist is None:
        uid = nxs.shard_unique_id()
    else:
        t = torch.zeros(128, dtype=torch.uint8, device=device or "cpu")
        if rank == 0:
            t.copy_(torch.frombuffer(bytearray(nxs.shard_unique_id()), dtype=torch.uint8))
        dist.broadcast(t, src=0)
        uid = bytes(t.cpu().numpy().tobytes())
    index.shard(rank, world, uid)


# ---- test hooks (CPU-side stand-ins; see nxs_api.c "Sharding without a second GPU")

def emulate(index, rank, world):
    """The index plays rank `rank` of `world` without a collective (0: off)."""
    L = lib()
    L.nxs_test_shard_emulate.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.nxs_test_shard_emulate(index._h, rank, world)


def emulated_block(index):
    """The record block the last emulated batch would have contributed."""
    L = lib()
    L.nxs_test_shard_block.restype = C.c_size_t
    L.nxs_test_shard_block.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    n = L.nxs_test_shard_block(index._h, None, 0)
    buf = C.create_string_buffer(max(n, 1))
    L.nxs_test_shard_block(index._h, buf, n)
    return buf.raw[:n]


def pack_block(results, n_slots, k):
    """Host stand-in for the device: [(status, [(doc, score), ...]), ...] for
    the slice's queries -> one record block."""
    L = lib()
    L.nxs_test_pack_record.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                       C.POINTER(C.c_uint64), C.POINTER(C.c_float), C.c_uint32]
    buf = C.create_string_buffer(max(block_bytes(n_slots, k), 1))
    for slot, (status, rs) in enumerate(results):
        rs = [] if status else rs[:k]
        ids = (C.c_uint64 * max(len(rs), 1))(*[d for d, _ in rs])
        sc = (C.c_float * max(len(rs), 1))(*[s for _, s in rs])
        L.nxs_test_pack_record(buf, n_slots, k, slot, len(rs), ids, sc, status)
    return buf.raw[:block_bytes(n_slots, k)]


def pack_abort(n_slots, k, code):
    """The block of a rank that cannot do its share of the batch: every status
    word says STATUS_ABORT | code.  The rank still takes part in the all-gather."""
    L = lib()
    L.nxs_test_pack_abort.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32]
    buf = C.create_string_buffer(max(block_bytes(n_slots, k), 1))
    L.nxs_test_pack_abort(buf, n_slots, k, code)
    return buf.raw[:block_bytes(n_slots, k)]


def mark_inexact(block, n_slots, k, slot):
    """tests: the record of `slot` says "inexact" (its owner re-runs the query in the fix-up round)."""
    L = lib()
    L.nxs_test_mark_inexact.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32]
    buf = C.create_string_buffer(block, len(block))
    L.nxs_test_mark_inexact(buf, n_slots, k, slot)
    return buf.raw[:len(block)]


def mark_changed(block, n_slots, k):
    """tests: the block's flags word says "this rank saw the index files move"."""
    L = lib()
    L.nxs_test_mark_changed.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32]
    buf = C.create_string_buffer(block, len(block))
    L.nxs_test_mark_changed(buf, n_slots, k)
    return buf.raw[:len(block)]


def blocks_changed(blocks, world, n_slots, k):
    """What every rank reads off the gathered blocks: all ranks re-sync at their next _begin."""
    L = lib()
    L.nxs_test_blocks_changed.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32]
    L.nxs_test_blocks_changed.restype = C.c_int
    return bool(L.nxs_test_blocks_changed(blocks, world, n_slots, k))


def fixup_scan(blocks, world, n_slots, k, n, rank):
    """What every rank reads off the gathered blocks (the library's fixup_scan): (a fix-up
    round is needed, the local indexes of `rank`'s queries to re-run)."""
    L = lib()
    L.nxs_test_fixup_scan.restype = C.c_int
    L.nxs_test_fixup_scan.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_size_t, C.c_int,
                                      C.POINTER(C.c_uint32), C.POINTER(C.c_size_t)]
    which = (C.c_uint32 * max(n_slots, 1))()
    nw = C.c_size_t()
    r = L.nxs_test_fixup_scan(blocks, world, n_slots, k, n, rank, which, C.byref(nw))
    return bool(r), [which[i] for i in range(nw.value)]


def fixup_verify(blocks, world, n_slots, k, n):
    """After the second all-gather (the library's fixup_verify): -1, or the rank that failed the batch."""
    L = lib()
    L.nxs_test_fixup_verify.restype = C.c_int
    L.nxs_test_fixup_verify.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_size_t]
    return L.nxs_test_fixup_verify(blocks, world, n_slots, k, n)


class ShardAborted(NxsError):
    """A rank aborted the sharded batch; every rank raises this for the batch."""
    def __init__(self, rank, code):
        NxsError.__init__(self, code, "rank %d aborted the sharded batch" % rank)
        self.rank = rank


def inject_failure(index, which, nth=1):
    """tests: the nth next _begin ("begin") / exact fix-up round ("fixup") / late second half ("late") fails."""
    L = lib()
    L.nxs_test_inject_failure.argtypes = [C.c_void_p, C.c_int, C.c_uint]
    L.nxs_test_inject_failure(index._h, {"begin": 0, "fixup": 1, "fixup_recv": 2, "late": 3}[which], nth)


def assemble(blocks, world, n_slots, k, n, only_rank=-1):
    """What every rank does with the gathered blocks (resps_from_blocks):
    -> list of result lists; a failed query is an NxsError in its slot.
    only_rank >= 0 (nxs_index_shard_local): that rank's slice alone, None elsewhere."""
    from . import _drain
    L = lib()
    L.nxs_test_assemble.restype = C.c_int
    L.nxs_test_assemble.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_size_t,
                                    C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int]
    resps = (C.c_void_p * max(n, 1))()
    errs = (C.c_int * max(n, 1))()
    r = L.nxs_test_assemble(blocks, world, n_slots, k, n, resps, errs, only_rank)
    own = shard_slice(n, only_rank, world) if only_rank >= 0 else (0, n)
    if r <= -2:
        raise ShardAborted(-2 - r, errs[0] if n else 1)
    if r < 0:
        raise NxsError(1, "assemble failed")
    out = []
    for i in range(n):
        if resps[i]:
            out.append(_drain(resps[i]))
            L.nxs_resp_release(resps[i])
        elif not own[0] <= i < own[1]:
            out.append(None)                     # another rank's query: not materialised here
        else:
            out.append(NxsError(errs[i], "query %d failed" % i))
    return out

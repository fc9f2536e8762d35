"""N > 1 path on CPU (no GPU here): two processes over the gloo backend run a
REAL sharded search of a tiny corpus -- every rank plans nothing by hand: it
takes its slice of the batch (the library's nxsgpu_shard_slice), answers its
queries (the oracle stands in for the device scan, which needs a GPU), packs
them into the library's record block, all-gathers the blocks (gloo stands in
for RCCL: the ONE collective of the path), and runs the library's reassembly
(resps_from_blocks, the code nxs_index_search_batch_end runs on every rank).
The `-m gpu` tier runs the same reassembly on blocks the GPU produced
(tests/test_gpu_parity.py::test_sharded_*)."""
import os
import socket
import struct
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import nxsfmt
import oracle_lib as O
from nxsearch_amd import multi

DOCS = {1: "cat dog cow", 2: "dog cow", 3: "cat cat cat", 4: "emu cat dog", 5: "cow emu",
        6: "gnu cat", 7: "dog dog gnu cow", 8: "emu", 9: "cat cow gnu", 10: "yak"}
QUERIES = ["cat", "dog AND cow", "emu OR gnu", "cat AND NOT dog", "zebra", "cat AND", "yak OR cot",
           "cow", "(cat OR dog) AND gnu", "gnu", "dog cow emu", "'cat'", "cow AND (", "emu AND cat",
           "yak", "dog OR dog", "cat OR zebra"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _answer(oidx, q, k):
    try:
        return 0, oidx.search(q, limit=k)
    except O.SearchError as e:
        return e.code, []


def _worker(rank, world, port, tdir, n, k, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oidx = O.Index(os.path.join(tdir, "nxsterms"), os.path.join(tdir, "nxsdtmap"))
    queries = QUERIES[:n]
    lo, hi = multi.shard_slice(n, rank, world)
    cap = multi.shard_capacity(n, world)
    mine = [_answer(oidx, q, k) for q in queries[lo:hi]]
    block = multi.pack_block(mine, cap, k)
    assert len(block) == multi.block_bytes(cap, k)
    send = torch.frombuffer(bytearray(block), dtype=torch.uint8)
    recv = torch.empty(world * len(block), dtype=torch.uint8)
    dist.all_gather_into_tensor(recv, send)           # the one collective
    got = multi.assemble(recv.numpy().tobytes(), world, cap, k, n)
    ok = len(got) == n
    for q, g in zip(queries, got):
        code, want = _answer(oidx, q, k)
        if code:
            ok &= isinstance(g, Exception) and g.code == code
        else:
            ok &= [d for d, _ in g] == [d for d, _ in want]
            ok &= [struct.pack("<f", s) for _, s in g] == [struct.pack("<f", s) for _, s in want]
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_slices_partition_the_batch():
    for n in (0, 1, 7, 1024, 8191):
        for w in (1, 2, 3, 8):
            sl = [multi.shard_slice(n, r, w) for r in range(w)]
            assert sl[0][0] == 0 and sl[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(sl, sl[1:]))
            assert max(h - l for l, h in sl) == multi.shard_capacity(n, w)
            assert max(h - l for l, h in sl) - min(h - l for l, h in sl) <= 1


def test_record_layout_is_the_documented_one():
    # u32 count | u32 flags | k x u64 | k x f32, padded to 8: 128 B at k = 10
    assert multi.rec_bytes(10) == 128 and multi.rec_bytes(1) == 24 and multi.rec_bytes(64) == 776
    blk = multi.pack_block([(0, [(7, 1.5), (3, 0.25)]), (3, [])], 3, 10)
    assert len(blk) == 3 * 128 + 16
    assert struct.unpack_from("<IIQQ", blk, 0) == (2, 0, 7, 3)
    assert struct.unpack_from("<ff", blk, 8 + 80) == (1.5, 0.25)
    assert struct.unpack_from("<IIII", blk, 3 * 128) == (0, 3, 0, 0)     # 3 status words + the flags word
    assert multi.block_bytes(2, 10) == 2 * 128 + 16 and multi.block_bytes(0, 10) == 8
    assert not multi.blocks_changed(blk, 1, 3, 10)
    assert struct.unpack_from("<I", multi.mark_changed(blk, 3, 10), 3 * 128 + 12) == (1,)
    got = multi.assemble(blk, 1, 3, 10, 3)
    assert got[0] == [(7, 1.5), (3, 0.25)] and got[1].code == 3 and got[2] == []


@pytest.mark.parametrize("n,k", [(5, 10), (17, 10), (16, 3)])
def test_two_rank_sharded_search_reassembles_the_batch(tmp_path, n, k):
    world = 2
    nxsfmt.write_index(str(tmp_path), "idx", [(d, t.split()) for d, t in DOCS.items()])
    tdir = str(tmp_path)
    if not os.path.exists(os.path.join(tdir, "nxsterms")):
        tdir = os.path.join(tdir, "data", "idx")
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, tdir, n, k, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def _worker_abort(rank, world, port, tdir, n, k, ret):
    """Rank 1 cannot do its share (planning failed, out of memory ...): it still
    contributes a block -- all status words STATUS_ABORT | code -- so the ONE
    collective completes on every rank, and every rank fails the batch together."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oidx = O.Index(os.path.join(tdir, "nxsterms"), os.path.join(tdir, "nxsdtmap"))
    lo, hi = multi.shard_slice(n, rank, world)
    cap = multi.shard_capacity(n, world)
    if rank == 1:
        block = multi.pack_abort(cap, k, 2)                  # NXS_ERR_SYSTEM
    else:
        block = multi.pack_block([_answer(oidx, q, k) for q in QUERIES[lo:hi]], cap, k)
    send = torch.frombuffer(bytearray(block), dtype=torch.uint8)
    recv = torch.empty(world * len(block), dtype=torch.uint8)
    dist.all_gather_into_tensor(recv, send)                  # nobody is left waiting here
    try:
        multi.assemble(recv.numpy().tobytes(), world, cap, k, n)
        ret[rank] = "no error"
    except multi.ShardAborted as e:
        ret[rank] = (e.rank, e.code)
    # the next batch is in step again
    block = multi.pack_block([_answer(oidx, q, k) for q in QUERIES[lo:hi]], cap, k)
    send = torch.frombuffer(bytearray(block), dtype=torch.uint8)
    dist.all_gather_into_tensor(recv, send)
    got = multi.assemble(recv.numpy().tobytes(), world, cap, k, n)
    assert len(got) == n
    dist.barrier()
    dist.destroy_process_group()


def test_a_failing_rank_cannot_strand_its_peers(tmp_path):
    world, n, k = 2, 9, 10
    nxsfmt.write_index(str(tmp_path), "idx", [(d, t.split()) for d, t in DOCS.items()])
    tdir = str(tmp_path)
    if not os.path.exists(os.path.join(tdir, "nxsterms")):
        tdir = os.path.join(tdir, "data", "idx")
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_abort, args=(world, _free_port(), tdir, n, k, ret), nprocs=world, join=True)
    assert dict(ret) == {0: (1, 2), 1: (1, 2)}


def _worker8(rank, world, port, tdir, n, k, ret):
    """Eight ranks, a C5-shaped batch (n = 8192: 1024 queries per rank), records marked inexact on
    several ranks (candidate overflow) -> every rank reads the same flags and takes the fix-up
    round; in the second batch rank 5's exact pass fails IN the fix-up round: it marks its block
    and still joins the second all-gather, every rank fails that batch, the third is in step."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oidx = O.Index(os.path.join(tdir, "nxsterms"), os.path.join(tdir, "nxsdtmap"))
    queries = [QUERIES[i % len(QUERIES)] for i in range(n)]
    lo, hi = multi.shard_slice(n, rank, world)
    cap = multi.shard_capacity(n, world)
    memo = {}

    def answer(q):
        if q not in memo:
            memo[q] = _answer(oidx, q, k)
        return memo[q]

    def gather(block):
        send = torch.frombuffer(bytearray(block), dtype=torch.uint8)
        recv = torch.empty(world * len(block), dtype=torch.uint8)
        dist.all_gather_into_tensor(recv, send)
        return recv.numpy().tobytes()

    def check(got):
        ok = len(got) == n
        for q, g in zip(queries, got):
            code, want = answer(q)
            if code:
                ok &= isinstance(g, Exception) and g.code == code
            else:
                ok &= [d for d, _ in g] == [d for d, _ in want]
                ok &= [struct.pack("<f", s) for _, s in g] == [struct.pack("<f", s) for _, s in want]
        return ok

    out = []
    own_times = []
    for batch, failing in ((0, None), (1, 5), (2, None)):
        mine = [answer(q) for q in queries[lo:hi]]
        # first pass: the overflowed queries carry a truncated answer and the "inexact" mark
        inexact = [i for i in range(hi - lo) if (lo + i) % 97 == 3 and mine[i][0] == 0 and rank in (1, 5, 6)]
        first = [(c, r[:1] if i in inexact else r) for i, (c, r) in enumerate(mine)]
        block = multi.pack_block(first, cap, k)
        for i in inexact:
            block = multi.mark_inexact(block, cap, k, i)
        blocks = gather(block)                                   # the batch's one collective ...
        need, which = multi.fixup_scan(blocks, world, cap, k, n, rank)
        assert need and which == inexact, (rank, which, inexact)
        # ... and the fix-up round every rank enters together
        blk_bytes = multi.block_bytes(cap, k)
        own = blocks[rank * blk_bytes:(rank + 1) * blk_bytes]
        if failing == rank:
            own = multi.pack_abort(cap, k, 2)
        else:
            own = multi.pack_block(mine, cap, k) if which else own
        blocks2 = gather(own)
        bad = multi.fixup_verify(blocks2, world, cap, k, n)
        if failing is None:
            assert bad == -1
            t0 = time.perf_counter()
            every = multi.assemble(blocks2, world, cap, k, n)
            t_all = time.perf_counter() - t0
            ok = check(every)
            # nxs_index_shard_local: the own slice only -- the same answers there, nothing materialised
            # for the queries the other ranks own (a rank's host work per batch is then O(n / world))
            t0 = time.perf_counter()
            own_only = multi.assemble(blocks2, world, cap, k, n, only_rank=rank)
            t_own = time.perf_counter() - t0
            for i, g in enumerate(own_only):
                if lo <= i < hi:
                    ok &= (isinstance(g, Exception) and isinstance(every[i], Exception) and g.code == every[i].code) or g == every[i]
                else:
                    ok &= g is None
            own_times.append((t_own, t_all))
            out.append(ok)
        else:
            out.append(bad == failing)
    # (the own slice is an eighth of the batch: well under half the time of all of it, whatever the host's noise)
    out.append(min(a for a, _ in own_times) < 0.5 * min(b for _, b in own_times))
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_fixup_round_and_an_abort_inside_it(tmp_path):
    world, n, k = 8, 8192, 10
    nxsfmt.write_index(str(tmp_path), "idx", [(d, t.split()) for d, t in DOCS.items()])
    tdir = str(tmp_path)
    if not os.path.exists(os.path.join(tdir, "nxsterms")):
        tdir = os.path.join(tdir, "data", "idx")
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker8, args=(world, _free_port(), tdir, n, k, ret), nprocs=world, join=True)
    assert dict(ret) == {r: [True, True, True, True] for r in range(world)}


def _worker_resync(rank, world, port, tdir, n, k, ret):
    """A pipelined sharded server (one batch always in flight): the indexer's append reaches rank 1's
    eyes first (it plans batch 2 a moment later than rank 0).  Its block for batch 2 carries the
    "changed" flag; every rank reads the flags of all blocks when it collects batch 2, so every rank
    decides to drain and re-read the files at the SAME later _begin -- the drain runs the fix-up round
    of the batch in flight, a collective, and must not interleave differently on different ranks."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oidx = O.Index(os.path.join(tdir, "nxsterms"), os.path.join(tdir, "nxsdtmap"))
    queries = QUERIES[:n]
    lo, hi = multi.shard_slice(n, rank, world)
    cap = multi.shard_capacity(n, world)
    mine = [_answer(oidx, q, k) for q in queries[lo:hi]]
    pending, inflight, log = False, [], []
    for batch in range(6):
        # _begin(batch): drain first if the flags of an EARLIER batch said so
        if pending and inflight:
            log.append(("drain", batch))
            pending = False
        block = multi.pack_block(mine, cap, k)
        if rank == 1 and batch == 2:
            block = multi.mark_changed(block, cap, k)
        send = torch.frombuffer(bytearray(block), dtype=torch.uint8)
        recv = torch.empty(world * len(block), dtype=torch.uint8)
        dist.all_gather_into_tensor(recv, send)           # queued by _begin
        inflight.append(recv.numpy().tobytes())
        if len(inflight) == 2:                            # _end(batch - 1)
            if multi.blocks_changed(inflight.pop(0), world, cap, k):
                pending = True
    ret[rank] = log
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_agree_on_the_batch_that_resyncs(tmp_path):
    world, n, k = 2, 9, 10
    nxsfmt.write_index(str(tmp_path), "idx", [(d, t.split()) for d, t in DOCS.items()])
    tdir = str(tmp_path)
    if not os.path.exists(os.path.join(tdir, "nxsterms")):
        tdir = os.path.join(tdir, "data", "idx")
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_resync, args=(world, _free_port(), tdir, n, k, ret), nprocs=world, join=True)
    # batch 2's flags are read at _end(2), which follows _begin(3): both ranks drain in _begin(4)
    assert dict(ret) == {0: [("drain", 4)], 1: [("drain", 4)]}

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
    assert struct.unpack_from("<III", blk, 3 * 128) == (0, 3, 0)
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

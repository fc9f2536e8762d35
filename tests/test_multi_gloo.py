"""N > 1 path on CPU: two processes over the gloo backend exercise the
query sharding and the all-gather of per-rank top-k records (the same code
bench.py runs over RCCL/xGMI)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nxsearch_amd import multi


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_topk(qi, k):
    """Deterministic stand-in for one query's device result."""
    cnt = qi % (k + 1)
    ids = [(qi * 1000003 + j * 7919) % (1 << 40) for j in range(cnt)]
    sc = [1.0 / (1 + qi + j) for j in range(cnt)]
    return cnt, ids, sc


def _worker(rank, world, port, n, k, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sb = multi.ShardedBatch(n, k, rank, world, torch.device("cpu"))
    for i, qi in enumerate(range(sb.lo, sb.hi)):
        cnt, ids, sc = _fake_topk(qi, k)
        sb.counts[i] = cnt
        sb.ids[i, :cnt] = torch.tensor(ids, dtype=torch.int64)
        sb.scores[i, :cnt] = torch.tensor(sc, dtype=torch.float32)
    sb.gather(dist)
    ids, scores, counts = sb.assemble()
    ok = ids.shape == (n, k) and counts.shape == (n,)
    for qi in range(n):
        cnt, eids, esc = _fake_topk(qi, k)
        ok &= int(counts[qi]) == cnt
        ok &= ids[qi, :cnt].tolist() == eids
        ok &= torch.allclose(scores[qi, :cnt], torch.tensor(esc, dtype=torch.float32))
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


@pytest.mark.parametrize("n", [5, 64])
def test_two_rank_allgather_reassembles_the_batch(n):
    world, k = 2, 10
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, k, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}

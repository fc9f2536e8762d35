"""Query-sharded multi-GPU execution.

Queries are independent and the index is read-only during a batch, so a batch
shards BY QUERY: rank r of W takes the contiguous slice
[n*r//W, n*(r+1)//W) of the batch, every rank holds a full replica of the
device index (N, token_count and df are identical on all replicas, so no
statistics are exchanged), and ONE all-gather of the fixed-size per-query
records -- count + k x (u64 doc id, f32 score), 124 B at k = 10 -- reassembles
the batch on every rank.  On ROCm the "nccl" backend is RCCL; the records
travel over xGMI.  The reference has no counterpart: it scales by running
independent worker processes (compose/nginx.conf:2).
"""
import torch


def shard_slice(n, rank, world):
    """Contiguous slice of an n-query batch owned by `rank`."""
    return n * rank // world, n * (rank + 1) // world


def shard_capacity(n, world):
    """Largest shard size (the all-gather needs equal-sized contributions)."""
    return max(shard_slice(n, r, world)[1] - shard_slice(n, r, world)[0]
               for r in range(world))


class ShardedBatch:
    """Buffers of one sharded batch: local [cap, k] + gathered [W*cap, k]."""

    def __init__(self, n, k, rank, world, device):
        self.n, self.k, self.rank, self.world = n, k, rank, world
        self.lo, self.hi = shard_slice(n, rank, world)
        self.cap = shard_capacity(n, world)
        z = dict(device=device)
        self.ids = torch.zeros((self.cap, k), dtype=torch.int64, **z)
        self.scores = torch.zeros((self.cap, k), dtype=torch.float32, **z)
        self.counts = torch.zeros((self.cap,), dtype=torch.int32, **z)
        if world > 1:
            self.g_ids = torch.empty((world * self.cap, k), dtype=torch.int64, **z)
            self.g_scores = torch.empty((world * self.cap, k), dtype=torch.float32, **z)
            self.g_counts = torch.empty((world * self.cap,), dtype=torch.int32, **z)
        else:
            self.g_ids, self.g_scores, self.g_counts = self.ids, self.scores, self.counts

    def gather(self, dist=None, group=None):
        """All-gather the per-rank top-k records (RCCL over xGMI on GPUs)."""
        if self.world > 1:
            dist.all_gather_into_tensor(self.g_ids, self.ids, group=group)
            dist.all_gather_into_tensor(self.g_scores, self.scores, group=group)
            dist.all_gather_into_tensor(self.g_counts, self.counts, group=group)

    def assemble(self):
        """-> (ids [n,k], scores [n,k], counts [n]) in original query order."""
        if self.world == 1:
            return self.ids[:self.n], self.scores[:self.n], self.counts[:self.n]
        parts = []
        for r in range(self.world):
            lo, hi = shard_slice(self.n, r, self.world)
            parts.append(slice(r * self.cap, r * self.cap + (hi - lo)))
        cat = lambda t: torch.cat([t[p] for p in parts], dim=0)
        return cat(self.g_ids), cat(self.g_scores), cat(self.g_counts)


def search_sharded(index, queries, limit=10, algo="BM25", fuzzymatch=False,
                   rank=0, world=1, device=None, dist=None, group=None):
    """Run `queries` (the same list on every rank) sharded by query over the
    ranks' GPUs; every rank returns the full (ids, scores, counts) tensors."""
    from . import BM25, TF_IDF
    sb = ShardedBatch(len(queries), limit, rank, world, device)
    mine = queries[sb.lo:sb.hi]
    if mine:
        plans, errs = index.plan_batch(mine, limit=limit, algo=algo, fuzzymatch=fuzzymatch)
        r = index.search_dev(plans, len(mine), limit, BM25 if algo.upper() == "BM25" else TF_IDF,
                             sb.ids.data_ptr(), sb.scores.data_ptr(), sb.counts.data_ptr())
        if r != 0:
            # a query overflowed its candidate segments: exact host-copy path
            res = index.search_batch(mine, limit=limit, algo=algo, fuzzymatch=fuzzymatch)
            for i, rs in enumerate(res):
                sb.counts[i] = len(rs)
                for j, (d, s) in enumerate(rs):
                    sb.ids[i, j] = d if d < (1 << 63) else d - (1 << 64)
                    sb.scores[i, j] = s
    sb.gather(dist, group)
    return sb.assemble()

"""Doc-ID-range sharding of the search path across ranks.

Shard g owns docs [g*n/G, (g+1)*n/G) with its own graph, entry node and ord->doc map (global doc ids);
there is no traversal-time communication.  Per batch every rank searches all queries on its shard, then ONE
all-gather of k x (doc, score) 8-byte pairs per query and a k-way merge by (score desc, doc asc) — the same shape as
Lucene's per-leaf search + TopDocs.merge that already wraps this path in the reference.

The collective is torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests);
the merge is a callable so the GPU run uses jv_merge_topk_device and CPU tests use the oracle's merge.
"""
from __future__ import annotations


def shard_range(n_total: int, world: int, rank: int):
    """contiguous doc-id range [lo, hi) of shard `rank`."""
    lo = (n_total * rank) // world
    hi = (n_total * (rank + 1)) // world
    return lo, hi


def gather_topk(dist, torch, docs, scores, world: int, buf=None):
    """ONE all-gather of the per-shard top-k lists as 8-byte (doc, score bits) pairs.
    docs/scores: [nq][k] -> ([nq][world*k] docs, [nq][world*k] scores), shard-major per query.
    `buf` (optional, [world][nq][k][2] int32 on the same device) avoids a per-call allocation."""
    nq, k = docs.shape
    pairs = torch.stack((docs.to(torch.int32), scores.contiguous().view(torch.int32)), dim=-1).contiguous()  # [nq][k][2]
    if buf is None or tuple(buf.shape) != (world, nq, k, 2):
        buf = torch.empty((world, nq, k, 2), dtype=torch.int32, device=docs.device)
    if docs.is_cuda and dist.get_backend() == "gloo":  # debug path (several ranks on one GPU): stage through the host
        torch.cuda.current_stream(docs.device).synchronize()
        hb = torch.empty(buf.shape, dtype=torch.int32)
        dist.all_gather_into_tensor(hb.view(-1), pairs.view(-1).cpu())
        buf.copy_(hb)
    else:
        dist.all_gather_into_tensor(buf.view(-1), pairs.view(-1))
    g = buf.permute(1, 0, 2, 3).reshape(nq, world * k, 2)
    gd = g[..., 0].contiguous()
    gs = g[..., 1].contiguous().view(torch.float32)
    return gd, gs


def sharded_search(dist, torch, local_search, merge, queries, k: int, world: int):
    """local_search(queries) -> (docs [nq][k] with GLOBAL doc ids, scores [nq][k]);
    merge(gathered_docs [nq][world*k], gathered_scores, k) -> (docs [nq][k], scores [nq][k])."""
    docs, scores = local_search(queries)
    if world == 1:
        return docs, scores
    gd, gs = gather_topk(dist, torch, docs, scores, world)
    return merge(gd, gs, k)

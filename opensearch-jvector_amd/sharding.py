"""Doc-ID-range sharding of the search path across ranks.

Shard g owns docs [g*n/G, (g+1)*n/G) with its own graph, entry node and ord->doc map (global doc ids);
there is no traversal-time communication.  Per batch every rank searches all queries on its shard, then ONE
all-gather of k x (doc, score) per query and a k-way merge by (score desc, doc asc) — the same shape as
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


def gather_topk(dist, torch, docs, scores, world: int, out_docs=None, out_scores=None):
    """all-gather per-shard top-k lists.  docs/scores: [nq][k] -> [nq][world*k] (shard-major per query)."""
    nq, k = docs.shape
    if out_docs is None:
        out_docs = torch.empty((world, nq, k), dtype=docs.dtype, device=docs.device)
        out_scores = torch.empty((world, nq, k), dtype=scores.dtype, device=scores.device)
    if docs.is_cuda and dist.get_backend() == "gloo":  # debug path (several ranks on one GPU): stage through the host
        torch.cuda.current_stream(docs.device).synchronize()
        hd, hs = torch.empty(out_docs.shape, dtype=docs.dtype), torch.empty(out_scores.shape, dtype=scores.dtype)
        dist.all_gather_into_tensor(hd.view(-1), docs.contiguous().view(-1).cpu())
        dist.all_gather_into_tensor(hs.view(-1), scores.contiguous().view(-1).cpu())
        out_docs.copy_(hd)
        out_scores.copy_(hs)
    else:
        dist.all_gather_into_tensor(out_docs.view(-1), docs.contiguous().view(-1))
        dist.all_gather_into_tensor(out_scores.view(-1), scores.contiguous().view(-1))
    gd = out_docs.permute(1, 0, 2).reshape(nq, world * k).contiguous()
    gs = out_scores.permute(1, 0, 2).reshape(nq, world * k).contiguous()
    return gd, gs


def sharded_search(dist, torch, local_search, merge, queries, k: int, world: int):
    """local_search(queries) -> (docs [nq][k] with GLOBAL doc ids, scores [nq][k]);
    merge(gathered_docs [nq][world*k], gathered_scores, k) -> (docs [nq][k], scores [nq][k])."""
    docs, scores = local_search(queries)
    if world == 1:
        return docs, scores
    gd, gs = gather_topk(dist, torch, docs, scores, world)
    return merge(gd, gs, k)

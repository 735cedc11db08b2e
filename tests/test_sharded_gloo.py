"""N>1 path on CPU: world_size-2 gloo, doc-range shards, all-gather of per-shard top-k, merge.
The per-shard engine here is the oracle (CPU); on the GPU box bench.py runs the same module with RCCL and
the HIP merge kernel.  Parity definition: the sharded result must equal the merge of the
shard-local results, and must equal a single-index brute force over the union when each shard is exact."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, d, k, rk, out_dir):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    graft.load_package()
    b = importlib.import_module("opensearch_jvector_amd.binding")
    bl = importlib.import_module("opensearch_jvector_amd.builder")
    dg = importlib.import_module("opensearch_jvector_amd.datagen")
    sh = importlib.import_module("opensearch_jvector_amd.sharding")
    po = graft.load_oracle()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sh.shard_range(n_total, world, rank)
    base = dg.splitmix_uniform(42, hi - lo, d, row_offset=lo)     # each rank generates only its doc range
    queries = dg.splitmix_uniform(43, 16, d)
    ord2doc = np.arange(lo, hi, dtype=np.int32)
    ix = bl.build_index_cpu(base, 0, R=16, L=60, ord2doc=ord2doc, max_doc=n_total, threads=2)
    orc = po.Oracle(b, ix)

    def local_search(q):
        r = orc.search_batch(q.numpy(), k, rk, threads=2)
        return torch.from_numpy(r.docs), torch.from_numpy(r.scores)

    def merge(gd, gs, kk):
        od, os_ = po.merge_topk(b, gd.numpy(), gs.numpy(), kk)
        return torch.from_numpy(od), torch.from_numpy(os_)

    docs, scores = sh.sharded_search(dist, torch, local_search, merge, torch.from_numpy(queries), k, world)
    ldocs, lscores = local_search(torch.from_numpy(queries))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), docs=docs.numpy(), scores=scores.numpy(),
             ldocs=ldocs.numpy(), lscores=lscores.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_search_equals_merge_of_shards(tmp_path, pkg, pyoracle):
    world, n_total, d, k, rk = 2, 3000, 24, 10, 40
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, d, k, rk, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    # every rank ends with the same merged answer
    assert np.array_equal(r0["docs"], r1["docs"]) and np.array_equal(r0["scores"], r1["scores"])
    # it is the (score desc, doc asc) merge of the two shard-local answers
    gd = np.concatenate([r0["ldocs"], r1["ldocs"]], axis=1)
    gs = np.concatenate([r0["lscores"], r1["lscores"]], axis=1)
    od, os_ = pyoracle.merge_topk(pkg.binding, gd, gs, k)
    assert np.array_equal(od, r0["docs"]) and np.array_equal(os_, r0["scores"])
    # docs of shard 0 are < n/2, shard 1 >= n/2; merged scores are non-increasing
    assert (r0["ldocs"] < n_total // 2).all() and (r1["ldocs"] >= n_total // 2).all()
    assert (np.diff(r0["scores"], axis=1) <= 0).all()
    # recall against brute force over the whole corpus
    base = pkg.datagen.splitmix_uniform(42, n_total, d)
    q = pkg.datagen.splitmix_uniform(43, 16, d)
    ix = pkg.binding.IndexData(vectors=base, adj=np.full((n_total, 1), -1, np.int32), entry_node=0)
    truth, _ = pyoracle.Oracle(pkg.binding, ix).brute_force(q, k)
    rec = np.mean([len(set(r0["docs"][i]) & set(truth[i])) / k for i in range(16)])
    assert rec >= 0.9


def test_shard_ranges_partition_the_corpus(pkg):
    sh = importlib.import_module("opensearch_jvector_amd.sharding")
    for n, w in [(10, 3), (100_000_000, 8), (7, 8), (0, 2)]:
        rs = [sh.shard_range(n, w, r) for r in range(w)]
        assert rs[0][0] == 0 and rs[-1][1] == n
        assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))

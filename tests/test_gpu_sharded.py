"""N>1 path with the HIP engine: two ranks (gloo rendezvous, both on GPU 0 of the single-GPU test box), doc-range
shards, each rank searches its shard with the HIP kernels through the C ABI, ONE fused all-gather of (doc, score)
pairs, jv_merge_topk_device on every rank.  Parity: the merged answer must equal the oracle's merge of
the oracle's shard-local answers — ids and score bits.  (The 8-GPU RCCL run is the driver's; the collective call is
the same torch.distributed all_gather_into_tensor, bench.py.)"""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, d, k, rk, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    graft.load_package()
    b = importlib.import_module("opensearch_jvector_amd.binding")
    bl = importlib.import_module("opensearch_jvector_amd.builder")
    dg = importlib.import_module("opensearch_jvector_amd.datagen")
    sh = importlib.import_module("opensearch_jvector_amd.sharding")
    po = graft.load_oracle()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sh.shard_range(n_total, world, rank)
    base = dg.splitmix_uniform(42, hi - lo, d, row_offset=lo)     # each rank generates only its doc range
    queries = dg.splitmix_uniform(43, 96, d)
    ord2doc = np.arange(lo, hi, dtype=np.int32)
    ix = bl.build_index_cpu(base, 0, R=16, L=60, pq_M=16, ord2doc=ord2doc, max_doc=n_total, threads=2)
    gpu = b.GpuIndex(ix, device=0, flags=b.DESC_FUSED_ADC)
    nq = queries.shape[0]
    tq = torch.from_numpy(queries).to(dev)
    o_nodes = torch.empty((nq, k), dtype=torch.int32, device=dev)
    o_docs = torch.empty((nq, k), dtype=torch.int32, device=dev)
    o_scores = torch.empty((nq, k), dtype=torch.float32, device=dev)
    o_count = torch.empty((nq,), dtype=torch.int32, device=dev)
    o_stats = torch.empty((nq, 4), dtype=torch.int32, device=dev)
    o_flags = torch.empty((nq,), dtype=torch.int32, device=dev)
    m_docs = torch.empty((nq, k), dtype=torch.int32, device=dev)
    m_scores = torch.empty((nq, k), dtype=torch.float32, device=dev)

    def local_search(q):
        gpu.search_batch_device(q.data_ptr(), nq, k, rk, o_nodes.data_ptr(), o_docs.data_ptr(), o_scores.data_ptr(),
                                o_count.data_ptr(), o_stats.data_ptr(), o_flags.data_ptr())
        return o_docs, o_scores

    def merge(gd, gs, kk):
        b.merge_topk_device(0, gd.data_ptr(), gs.data_ptr(), nq, world, kk, m_docs.data_ptr(), m_scores.data_ptr())
        torch.cuda.synchronize()
        return m_docs, m_scores

    docs, scores = sh.sharded_search(dist, torch, local_search, merge, tq, k, world)
    want = po.Oracle(b, ix).search_batch(queries, k, rk)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), docs=docs.cpu().numpy(), scores=scores.cpu().numpy(),
             ldocs=o_docs.cpu().numpy(), lscores=o_scores.cpu().numpy(), lstats=o_stats.cpu().numpy(),
             odocs=want.docs, oscores=want.scores, ostats=want.stats)
    dist.barrier()
    gpu.close()
    dist.destroy_process_group()


def test_two_rank_hip_engine_sharded_search_equals_oracle_merge(tmp_path, pkg, pyoracle):
    import torch.multiprocessing as mp
    world, n_total, d, k, rk = 2, 6000, 64, 10, 50
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, d, k, rk, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"rank{i}.npz") for i in range(world)]
    for i in range(world):   # shard-local HIP answers == shard-local oracle answers (ids, score bits, counters)
        assert np.array_equal(r[i]["ldocs"], r[i]["odocs"])
        assert np.array_equal(r[i]["lscores"].view(np.uint32), r[i]["oscores"].view(np.uint32))
        assert np.array_equal(r[i]["lstats"], r[i]["ostats"])
    # every rank ends with the same merged answer, and it is the oracle's merge of the oracle's shard answers
    assert np.array_equal(r[0]["docs"], r[1]["docs"]) and np.array_equal(r[0]["scores"].view(np.uint32), r[1]["scores"].view(np.uint32))
    gd = np.concatenate([r[0]["odocs"], r[1]["odocs"]], axis=1)
    gs = np.concatenate([r[0]["oscores"], r[1]["oscores"]], axis=1)
    od, os_ = pyoracle.merge_topk(pkg.binding, gd, gs, k)
    assert np.array_equal(od, r[0]["docs"]) and np.array_equal(os_.view(np.uint32), r[0]["scores"].view(np.uint32))
    assert (r[0]["ldocs"] < n_total // 2).all() and (r[1]["ldocs"] >= n_total // 2).all()


def _rccl_one_rank_worker(rank, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    graft.load_package()
    sh = importlib.import_module("opensearch_jvector_amd.sharding")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)   # exactly bench.py's call
    g = torch.Generator(device="cpu").manual_seed(5)
    docs = torch.randint(0, 1 << 30, (96, 10), generator=g, dtype=torch.int32).to(dev)
    scores = torch.randn((96, 10), generator=g).to(dev)
    gd, gs = sh.gather_topk(dist, torch, docs, scores, 1)                        # all_gather_into_tensor on RCCL
    tmax = torch.tensor([1.25], device=dev, dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)                                   # bench.py's max-over-ranks timing
    lst = [torch.empty_like(docs)]
    dist.all_gather(lst, docs)                                                    # bench.py's ground-truth gather
    dist.barrier()
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "rccl1.npz"), gd=gd.cpu().numpy(), gs=gs.cpu().numpy(), docs=docs.cpu().numpy(),
             scores=scores.cpu().numpy(), tmax=tmax.cpu().numpy(), lst=lst[0].cpu().numpy(), backend=np.array([dist.get_backend()]))
    dist.destroy_process_group()


def test_rccl_backend_executes_with_one_rank(tmp_path):
    """The boxes this repo is built on have ONE GPU, so no N > 1 RCCL run exists.  This runs the same torch.distributed calls
    bench.py makes (`init_process_group("nccl", device_id=…)`, `all_gather_into_tensor` of the pair buffer, `all_gather`, `all_reduce(MAX)`,
    `barrier`) on a one-rank RCCL communicator: librccl loads, the communicator comes up on this image and the calls execute."""
    import torch.multiprocessing as mp
    mp.spawn(_rccl_one_rank_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    r = np.load(tmp_path / "rccl1.npz")
    assert str(r["backend"][0]) == "nccl"
    assert np.array_equal(r["gd"], r["docs"]) and np.array_equal(r["gs"].view(np.uint32), r["scores"].view(np.uint32))
    assert np.array_equal(r["lst"], r["docs"]) and float(r["tmax"][0]) == 1.25

#!/usr/bin/env python3
"""bench.py — queries/sec of the jVector GraphSearcher hot path on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path (beam search + scoring [+ PQ ADC + exact rerank]) over one batch
of synthetic queries that is already resident in HBM.  One process per GPU; for --gpus N > 1 the
corpus is sharded by doc-ID range (each rank owns its own graph over `n` docs), every rank searches
the whole batch, per-shard top-k lists are all-gathered over RCCL and merged on the GPU.

Prints ONE JSON line (rank 0).  PyTorch is plumbing here: device memory, streams, torch.distributed.
The oracle (oracle/) is used only for the cpu_baseline leg and a parity spot-check, never for `value`.
"""
from __future__ import annotations

import argparse
import importlib
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)

WORKLOADS = {
    # BASELINE.json configs[1]
    "c2": dict(desc="1Mx768 dot-product fp32 resident, rerankK swept for recall@10>=0.95", n=1_000_000, d=768, sim=1,
               pq_M=0, normalize=True),
    # BASELINE.json configs[2] — the configuration the metric is quoted on
    # (262 144 queries per step: per-query work varies 3x around its mean, and the drain of the slowest queries at
    # the end of a launch costs ~7 % at 65 536 queries per step, ~2 % here)
    "c3": dict(desc="10Mx768 L2 PQ-32 ADC + full-precision rerank (DiskANN two-pass)", n=10_000_000, d=768, sim=0,
               pq_M=32, normalize=False, batch=262144),
    # BASELINE.json configs[4]: 256 concurrent queries on the C3 index.  Run on the fp32 parity path: at B = 256 the
    # only dense contraction (the LUT build, 2*B*256*d = 0.1 GFLOP) is ~0.02 % of a step and the rerank has no
    # candidates shared between queries, so a bf16 MFMA path would change scores without changing throughput
    # (DESIGN.md section 8).
    "c5": dict(desc="batch=256 concurrent queries on the C3 index (10Mx768 PQ-32 fused + rerank), fp32 parity path",
               n=10_000_000, d=768, sim=0, pq_M=32, normalize=False, batch=256),
    # BASELINE.json configs[3]: 100Mx1536 PQ-64 over 8 GPUs = 12.5M docs per GPU (always "weak": n per GPU fixed)
    "c4": dict(desc="100Mx1536 PQ-64 DiskANN, doc-ID-range shards of 12.5M per GPU", n=12_500_000, d=1536, sim=0,
               pq_M=64, normalize=False, per_gpu=True),
}


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


# ------------------------------------------------------------------------------------------------
# synthetic data, generated directly in HBM
# ------------------------------------------------------------------------------------------------
def make_generators(torch, d, device, centres, rank_r, seed=44):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    cen = torch.rand((centres, d), generator=g, device=device, dtype=torch.float32)
    basis = torch.randn((rank_r, d), generator=g, device=device, dtype=torch.float32) / math.sqrt(rank_r)
    return cen, basis


def gen_rows(torch, n, d, seed, row_offset, cen, basis, sigma_sub, sigma_iso, normalize, device):
    """Clustered synthetic embeddings: centre[i mod C] + low-rank Gaussian (intrinsic dim = rank of
    `basis`) + small isotropic noise.  Deterministic in (seed, row_offset, chunking)."""
    out = torch.empty((n, d), dtype=torch.float32, device=device)
    g = torch.Generator(device=device)
    chunk = 1 << 17
    C = cen.shape[0]
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        g.manual_seed(seed * 1_000_003 + row_offset + s)
        idx = (torch.arange(s, s + m, device=device) + row_offset) % C
        z = torch.randn((m, basis.shape[0]), generator=g, device=device, dtype=torch.float32)
        x = cen[idx] + sigma_sub * (z @ basis) + sigma_iso * torch.randn((m, d), generator=g, device=device, dtype=torch.float32)
        if normalize:
            x = x / x.norm(dim=1, keepdim=True)
        out[s:s + m] = x
    return out


def make_block_generators(torch, d, device, centres, M=32, per=2, grank=8, seed=44):
    """Product-structured latent model for the PQ workload: every PQ subspace (d/M dims) is driven by `per`
    latent factors of its own (cluster centre + Gaussian), plus a weak shared rank-`grank` component.
    Intrinsic dimension = M*per; 8-bit codebooks per subspace can resolve it, as they can for real
    embeddings whose PQ-32 recall is usable — i.i.d. 768-d noise is not."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    ds = d // M
    zc = torch.randn((centres, M * per), generator=g, device=device, dtype=torch.float32)
    Bl = torch.zeros((M * per, d), device=device, dtype=torch.float32)
    for m in range(M):
        Bl[per * m:per * m + per, m * ds:(m + 1) * ds] = torch.randn((per, ds), generator=g, device=device) / math.sqrt(per)
    Bg = torch.randn((grank, d), generator=g, device=device, dtype=torch.float32) / math.sqrt(grank)
    return zc, Bl, Bg


def gen_rows_block(torch, n, d, seed, row_offset, zc, Bl, Bg, sigma, gscale, iso, normalize, device):
    out = torch.empty((n, d), dtype=torch.float32, device=device)
    g = torch.Generator(device=device)
    chunk = 1 << 17
    C = zc.shape[0]
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        g.manual_seed(seed * 1_000_003 + row_offset + s)
        idx = (torch.arange(s, s + m, device=device) + row_offset) % C
        z = zc[idx] + sigma * torch.randn((m, zc.shape[1]), generator=g, device=device, dtype=torch.float32)
        x = z @ Bl + gscale * (torch.randn((m, Bg.shape[0]), generator=g, device=device, dtype=torch.float32) @ Bg) \
            + iso * torch.randn((m, d), generator=g, device=device, dtype=torch.float32)
        if normalize:
            x = x / x.norm(dim=1, keepdim=True)
        out[s:s + m] = x
    return out


def brute_force_topk(torch, base, queries, k, sim, row_offset=0):
    """exact top-k doc ids on the GPU (ground truth for recall): fp32 GEMM in chunks."""
    nq = queries.shape[0]
    best_s = torch.full((nq, k), -float("inf"), device=base.device)
    best_i = torch.full((nq, k), -1, dtype=torch.int64, device=base.device)
    chunk = 1 << 18
    qn = (queries * queries).sum(1, keepdim=True)
    for s in range(0, base.shape[0], chunk):
        b = base[s:s + chunk]
        dots = queries @ b.T
        if sim == 0:
            sc = -(qn - 2 * dots + (b * b).sum(1)[None, :])
        else:
            sc = dots
        cs, ci = sc.topk(min(k, b.shape[0]), dim=1)
        alls = torch.cat([best_s, cs], 1)
        alli = torch.cat([best_i, ci + s + row_offset], 1)
        ts, ti = alls.topk(k, dim=1)
        best_s, best_i = ts, torch.gather(alli, 1, ti)
    return best_i


def recall_of(found_docs, truth):
    f = found_docs.cpu().numpy()
    t = truth.cpu().numpy()
    tot = 0.0
    for a, b in zip(f, t):
        tot += len(set(int(x) for x in a if x >= 0) & set(int(x) for x in b)) / len(b)
    return tot / len(f)


# ------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=os.environ.get("JV_BENCH_WORKLOAD", "c3"), choices=sorted(WORKLOADS))
    ap.add_argument("--n", type=int, default=int(os.environ.get("JV_BENCH_N", "0")), help="docs per GPU (0 = workload default)")
    ap.add_argument("--batch", type=int, default=int(os.environ.get("JV_BENCH_BATCH", "65536")), help="queries per step")
    ap.add_argument("--rerankk", type=int, default=int(os.environ.get("JV_BENCH_RERANKK", "0")), help="0 = sweep for recall>=0.95")
    ap.add_argument("--builder", default=os.environ.get("JV_BENCH_BUILDER", "gpu"), choices=["gpu", "cpu"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--profile-mode", action="store_true",
                    help="only warm-up + timed launches of the query kernel (needs --rerankk): no recall sweep, no p50, no CPU leg")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    # one rank per GPU; (debug only: JV_BENCH_BACKEND=gloo lets several ranks share one GPU to exercise the
    # shard / all-gather / merge path on a single-GPU box — the collective then stages through the host)
    backend = os.environ.get("JV_BENCH_BACKEND", "nccl")
    local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)

    graft.load_package()
    binding = importlib.import_module("opensearch_jvector_amd.binding")
    builder = importlib.import_module("opensearch_jvector_amd.builder")
    sharding = importlib.import_module("opensearch_jvector_amd.sharding")

    # engine tunables for experiments: JV_OPT_<name>=<int>  ->  jv_set_option(name, value)
    for key, val in os.environ.items():
        if key.startswith("JV_OPT_"):
            binding.set_option(key[len("JV_OPT_"):].lower(), int(val))
            log(f"option {key[len('JV_OPT_'):].lower()} = {val}")

    wl = dict(WORKLOADS[args.workload])
    if "batch" in wl and "JV_BENCH_BATCH" not in os.environ and "--batch" not in sys.argv:
        args.batch = wl["batch"]
    n_cfg = args.n if args.n > 0 else wl["n"]
    d, sim, pq_M = wl["d"], wl["sim"], wl["pq_M"]
    R, L, k = 32, 100, 10
    # N > 1: doc-ID-range shards.  Default = the north star's curve: the SAME corpus (10M docs) split over the
    # ranks ("strong": total work fixed; every rank searches every query on its n/N docs, rerankK is re-swept
    # so that the MERGED recall@10 stays >= 0.95).  JV_BENCH_SCALING=weak keeps n docs per GPU instead.
    scaling = os.environ.get("JV_BENCH_SCALING", "strong" if (world > 1 and not wl.get("per_gpu")) else "weak")
    if world > 1 and scaling == "strong":
        lo_doc, hi_doc = sharding.shard_range(n_cfg, world, rank)
        n, row_offset, n_total = hi_doc - lo_doc, lo_doc, n_cfg
    else:
        n, row_offset, n_total = n_cfg, rank * n_cfg, n_cfg * world

    # ---- data in HBM ----
    t0 = time.time()
    centres = max(64, min(4096, n_total // 256))
    nq_pool = max(args.batch * 2, 4096)
    if pq_M:
        sigma = float(os.environ.get("JV_BENCH_SIGMA", "0.35"))
        # 64 latent factors in total (2 per subspace at M = 32, 1 at M = 64): the same intrinsic dimension for C3 and C4
        zc, Bl, Bg = make_block_generators(torch, d, device, centres, M=pq_M, per=max(1, 64 // pq_M))
        base = gen_rows_block(torch, n, d, 42, row_offset, zc, Bl, Bg, sigma, 0.1, 0.005, wl["normalize"], device)
        queries = gen_rows_block(torch, nq_pool, d, 43, 0, zc, Bl, Bg, sigma, 0.1, 0.005, wl["normalize"], device)
    else:
        cen, basis = make_generators(torch, d, device, centres, 32)
        base = gen_rows(torch, n, d, 42, row_offset, cen, basis, 0.15, 0.01, wl["normalize"], device)
        queries = gen_rows(torch, nq_pool, d, 43, 0, cen, basis, 0.15, 0.01, wl["normalize"], device)
    torch.cuda.synchronize()
    log(f"rank {rank}: generated {n}x{d} base + {nq_pool} queries in {time.time() - t0:.1f}s")

    # ---- index construction (write side; not timed, not the hot path) ----
    t0 = time.time()
    pq = None
    if args.builder == "gpu":
        gbuild = importlib.import_module("opensearch_jvector_amd.builder_gpu")
        adj_t, entry = gbuild.build_graph_gpu(torch, base, sim, R=R, L=L, alpha=1.2, device_index=local_rank)
        if pq_M:
            pq = gbuild.pq_train_encode_gpu(torch, base, pq_M, sim)
    else:
        host = base.cpu().numpy()
        adj_np, entry = builder.build_graph_cpu(host, sim, R=R, L=L)
        adj_t = torch.from_numpy(adj_np).to(device)
        if pq_M:
            cb, cenq, codes, K = builder.pq_train_encode_cpu(host, pq_M, sim)
            pq = dict(codebooks=cb, centroid=cenq, codes=torch.from_numpy(codes).to(device), K=K)
        del host
    ord2doc = (torch.arange(n, device=device, dtype=torch.int32) + row_offset) if world > 1 else None
    torch.cuda.synchronize()
    build_s = time.time() - t0
    log(f"rank {rank}: built index ({args.builder}) in {build_s:.1f}s, entry={entry}")

    desc, keep = binding.make_desc_device(
        n, d, R, base.data_ptr(), adj_t.data_ptr(), entry, sim, device=local_rank,
        pq_M=pq_M, pq_K=(pq["K"] if pq else 0), pq_codebooks=(pq["codebooks"] if pq else None),
        pq_centroid=(pq["centroid"] if pq else None), pq_codes_ptr=(pq["codes"].data_ptr() if pq else 0),
        ord2doc_ptr=(ord2doc.data_ptr() if ord2doc is not None else 0), max_doc=n_total, borrow=True,
        extra_flags=(binding.DESC_FUSED_ADC if (pq and os.environ.get("JV_BENCH_FUSED", "1") == "1") else 0))
    index = binding.GpuIndex(desc=desc, keepalive=keep, flags=binding.DESC_BORROW)

    # ---- search plumbing: everything device-resident, own stream ----
    stream = torch.cuda.Stream(device=device)
    B = args.batch
    n_gt = 1024  # queries with exact ground truth (recall is quoted on these)
    OB = max(B, n_gt)  # output rows: a timed step uses the first B, the recall sweep the first n_gt
    out_nodes = torch.empty((OB, k), dtype=torch.int32, device=device)
    out_docs = torch.empty((OB, k), dtype=torch.int32, device=device)
    out_scores = torch.empty((OB, k), dtype=torch.float32, device=device)
    out_count = torch.empty((OB,), dtype=torch.int32, device=device)
    out_stats = torch.empty((OB, 4), dtype=torch.int32, device=device)
    out_flags = torch.empty((OB,), dtype=torch.int32, device=device)
    if world > 1:
        gather_docs = torch.empty((world, B, k), dtype=torch.int32, device=device)
        gather_scores = torch.empty((world, B, k), dtype=torch.float32, device=device)
        merged_docs = torch.empty((OB, k), dtype=torch.int32, device=device)
        merged_scores = torch.empty((OB, k), dtype=torch.float32, device=device)

    def gpu_merge(gd, gs, kk):
        nqm = gd.shape[0]
        binding.merge_topk_device(local_rank, gd.data_ptr(), gs.data_ptr(), nqm, world, kk, merged_docs.data_ptr(),
                                  merged_scores.data_ptr(), stream=stream.cuda_stream)
        return merged_docs[:nqm], merged_scores[:nqm]

    def run_step(qbatch, rk, nq=B):
        """one pass of the hot path over one batch; returns the final (docs, scores) tensors"""
        def local_search(q):
            index.search_batch_device(q.data_ptr(), nq, k, rk, out_nodes.data_ptr(), out_docs.data_ptr(),
                                      out_scores.data_ptr(), out_count.data_ptr(), out_stats.data_ptr(),
                                      out_flags.data_ptr(), stream=stream.cuda_stream)
            return out_docs[:nq], out_scores[:nq]
        with torch.cuda.stream(stream):
            return sharding.sharded_search(dist, torch, local_search, gpu_merge, qbatch, k, world)

    # ---- ground truth + rerankK selection (recall@10 >= 0.95) ----
    gt_local = brute_force_topk(torch, base, queries[:n_gt], k, sim, row_offset)
    if world > 1:
        # global ground truth = merge of per-shard exact top-k
        qn = queries[:n_gt]
        if sim == 0:
            def sc_of(ids):
                v = base[(ids - row_offset).clamp(0, n - 1)]
                return -((qn[:, None, :] - v) ** 2).sum(-1)
        else:
            def sc_of(ids):
                v = base[(ids - row_offset).clamp(0, n - 1)]
                return (qn[:, None, :] * v).sum(-1)
        ls = sc_of(gt_local)
        if backend == "nccl":
            all_i = [torch.empty_like(gt_local) for _ in range(world)]
            all_s = [torch.empty_like(ls) for _ in range(world)]
            dist.all_gather(all_i, gt_local)
            dist.all_gather(all_s, ls)
        else:
            all_i = [torch.empty_like(gt_local, device="cpu") for _ in range(world)]
            all_s = [torch.empty_like(ls, device="cpu") for _ in range(world)]
            dist.all_gather(all_i, gt_local.cpu())
            dist.all_gather(all_s, ls.cpu())
            all_i, all_s = [t.to(device) for t in all_i], [t.to(device) for t in all_s]
        ci, cs = torch.cat(all_i, 1), torch.cat(all_s, 1)
        top = cs.topk(k, dim=1).indices
        gt = torch.gather(ci, 1, top)
    else:
        gt = gt_local
    sweep = [args.rerankk] if args.rerankk > 0 else (list(range(20, 200, 10)) + [200, 225, 250, 300, 350, 400])
    chosen, chosen_recall, sweep_log = None, 0.0, []
    if args.profile_mode:
        if args.rerankk <= 0:
            raise SystemExit("--profile-mode needs --rerankk")
        sweep, chosen, chosen_recall = [], args.rerankk, float("nan")
    for rk in sweep:
        docs, _ = run_step(queries[:n_gt], rk, nq=n_gt)
        stream.synchronize()
        rec = recall_of(docs[:n_gt], gt)
        sweep_log.append((rk, round(rec, 4)))
        chosen, chosen_recall = rk, rec
        if rec >= 0.95:
            break
    log(f"recall sweep (rerankK, recall@10): {sweep_log}")
    rk = chosen

    # ---- timed region ----
    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    batches = [queries[i * B:(i + 1) * B] for i in range(nq_pool // B)]
    for w in range(args.warmup):
        run_step(batches[w % len(batches)], rk)
    barrier()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    stat_sums = torch.zeros((4,), dtype=torch.int64, device=device)
    barrier()
    t_start = time.perf_counter()
    for s in range(args.steps):
        evs[s][0].record(stream)
        index.search_batch_device(batches[s % len(batches)].data_ptr(), B, k, rk, out_nodes.data_ptr(), out_docs.data_ptr(),
                                  out_scores.data_ptr(), out_count.data_ptr(), out_stats.data_ptr(),
                                  out_flags.data_ptr(), stream=stream.cuda_stream)
        evs[s][1].record(stream)
        if world > 1:
            with torch.cuda.stream(stream):
                gd, gs = sharding.gather_topk(dist, torch, out_docs[:B], out_scores[:B], world, gather_docs, gather_scores)
                gpu_merge(gd, gs, k)
        with torch.cuda.stream(stream):
            stat_sums += out_stats[:B].to(torch.int64).sum(0)  # per-query counters -> algorithmic bytes
    barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        tmax = torch.tensor([elapsed], device=(device if backend == "nccl" else "cpu"), dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    total_queries = args.steps * B
    qps = total_queries / elapsed
    kernel_ms = [a.elapsed_time(b) for a, b in evs]
    kernel_avg_ms = float(np.mean(kernel_ms))
    overflowed = int((out_flags[:B].cpu().numpy().astype(np.uint32) & np.uint32(1)).sum())

    # ---- algorithmic bytes; counters are the reference's own (J/JVectorReader.java:183-187) ----
    st = stat_sums.cpu().numpy().astype(np.float64)
    visited, reranked, expanded = st[0], st[1], st[2]
    if pq_M:
        # LUT_q = 1024*d (one codebook read per query: each query builds its own table)
        bytes_total = visited * pq_M + expanded * 4 * (R + 1) + reranked * 4 * d + total_queries * 1024.0 * d
    else:
        bytes_total = visited * 4 * d + expanded * 4 * (R + 1)
    bytes_per_launch = bytes_total / args.steps
    achieved_gbs = bytes_per_launch / (kernel_avg_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath):
        # HBM bytes per launch from the PMC passes of the same command (tools/profile_bench.sh ->
        # tools/summarize_profile.py); used only when workload, n, batch and rerankK all match this run
        try:
            tj = json.load(open(tpath)).get("entries", {}).get(args.workload)
            if world == 1 and tj and tj.get("n") == n and tj.get("batch") == B and tj.get("rerankK") == rk:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    # ---- p50 latency: one query in flight through the host-pointer API ----
    qh = queries[:(0 if args.profile_mode else 200)].cpu().numpy()
    lat = [0.0] * 21
    for i in range(len(qh)):
        t1 = time.perf_counter()
        index.search(qh[i], k, rk)
        lat.append((time.perf_counter() - t1) * 1e3)
    p50 = float(np.median(lat[20:]))
    # PCIe-inclusive batch rate through the host-pointer API (queries in pageable host memory, results copied back)
    pcie_qps = None
    if not args.profile_mode:
        hb = queries[:min(B, 16384)].cpu().numpy()
        index.search_batch(hb[:256], k, rk)
        t1 = time.perf_counter()
        index.search_batch(hb, k, rk)
        pcie_qps = round(len(hb) / (time.perf_counter() - t1), 1)
    # the reference's own calling pattern: searcher threads issuing ONE query per call on a shared handle
    # (JVectorConcurrentQueryTests.java:78-138); the library combines concurrent calls into batch launches
    caller_rows = None
    if not args.profile_mode and world == 1:
        try:
            hostmod = importlib.import_module("opensearch_jvector_amd.host")
            hq = queries[:4096].cpu().numpy()
            caller_rows = []
            for T in (1, 64, 256):
                r = hostmod.concurrent_search_bench(index, hq, k, rk, T, 1.5)
                caller_rows.append({"caller_threads": T, "qps": round(r["qps"], 1), "p50_ms": round(r["p50_ms"], 4),
                                    "p99_ms": round(r["p99_ms"], 4)})
        except Exception as e:  # pragma: no cover - reported, never fatal for the headline number
            caller_rows = f"unavailable: {e!r}"

    fused_on = bool(pq_M) and os.environ.get("JV_BENCH_FUSED", "1") == "1"
    main_kernel = "jv_search_pqf_kernel" if fused_on else "jv_search_lds_kernel"
    result = {
        "metric": "queries/sec at recall@10>=0.95",
        "value": round(qps, 1),
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"{args.workload}: {wl['desc']}" + ("" if n_total == wl["n"] * (world if scaling == "weak" else 1) else f" [n_total={n_total}]"),
            "docs_per_gpu": n, "total_docs": n_total, "dim": d, "similarity": ["l2", "dot", "cosine"][sim],
            "R": R, "ef_construction": L, "k": k, "rerankK": rk, "pq_M": pq_M, "queries_per_step": B,
            "sharding": "doc-id range, RCCL all-gather of per-shard top-k + GPU merge" if world > 1 else "single GPU",
            "graph_builder": args.builder, "pq_layout": ("fused" if (pq_M and os.environ.get("JV_BENCH_FUSED", "1") == "1") else ("plain" if pq_M else None)),
        },
        "recall_at_10": (None if chosen_recall != chosen_recall else round(chosen_recall, 4)),
        "recall_sweep": sweep_log,
        "p50_latency_ms": round(p50, 4),
        "host_api_qps_pcie_inclusive": pcie_qps,
        "single_query_api": caller_rows,
        "per_query": {"visited": round(visited / total_queries, 1), "expanded": round(expanded / total_queries, 1),
                      "reranked": round(reranked / total_queries, 1),
                      "algorithmic_bytes": round(bytes_total / total_queries, 1)},
        "big_path_queries_last_step": overflowed,
        "build_seconds": round(build_s, 1),
        "roofline": {"bound": "hbm", "achieved": round(achieved_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved_gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "kernel": main_kernel, "kernel_avg_ms": round(kernel_avg_ms, 4),
                     "algorithmic_bytes_per_launch": round(bytes_per_launch, 1)},
    }

    # ---- CPU baseline: the oracle (a port/restatement, NOT real jVector) on this box's host cores ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.profile_mode:
        try:
            result["cpu_baseline"] = cpu_baseline(torch, binding, base, adj_t, entry, sim, pq, queries, k, rk, out_nodes,
                                                  index, run_step, stream, args.cpu_seconds)
        except MemoryError as e:  # pragma: no cover
            result["cpu_baseline"] = {"value": None, "unit": "queries/s", "cores": 0, "kind": "port", "sample": f"skipped: {e}"}
    if rank == 0:
        print(json.dumps(result), flush=True)
    index.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(torch, binding, base, adj_t, entry, sim, pq, queries, k, rk, out_nodes, index, run_step, stream, budget_s):
    import psutil
    pyoracle = graft.load_oracle()
    need = base.numel() * 4 + adj_t.numel() * 4
    if psutil.virtual_memory().available < need * 1.3:
        raise MemoryError(f"host RAM too small for a {need / 1e9:.1f} GB index copy")
    t0 = time.time()
    ix = binding.IndexData(vectors=base.cpu().numpy(), adj=adj_t.cpu().numpy(), entry_node=entry, similarity=sim)
    if pq:
        ix.pq_codebooks, ix.pq_centroid, ix.pq_codes = pq["codebooks"], pq["centroid"], pq["codes"].cpu().numpy()
        ix.pq_M, ix.pq_K = ix.pq_codes.shape[1], pq["K"]
    orc = pyoracle.Oracle(binding, ix)
    log(f"cpu_baseline: index copied to host in {time.time() - t0:.1f}s")
    cores = os.cpu_count() or 1
    pool = queries.cpu().numpy()
    orc.search_batch(pool[:min(1024, len(pool))], k, rk, threads=cores)  # warm the thread pool / page in the index
    nsample = min(len(pool), 2048)
    while True:
        sample = pool[:nsample]
        t1 = time.perf_counter()
        r = orc.search_batch(sample, k, rk, threads=cores)
        dt = time.perf_counter() - t1
        if dt >= 0.6 * budget_s or nsample >= len(pool):
            break
        nsample = int(min(len(pool), max(nsample * 2, nsample * budget_s / max(dt, 1e-3))))
    # parity spot-check of the measured GPU path against the oracle on the same queries
    docs, _ = run_step(queries[:nsample] if nsample <= out_nodes.shape[0] else queries[:out_nodes.shape[0]], rk,
                       nq=min(nsample, out_nodes.shape[0]))
    stream.synchronize()
    m = min(nsample, out_nodes.shape[0])
    same = bool(np.array_equal(out_nodes[:m].cpu().numpy(), r.nodes[:m]))
    # single-thread figure (the reference's JMH style is one thread)
    t2 = time.perf_counter()
    orc.search_batch(sample[:max(8, min(64, nsample))], k, rk, threads=1)
    dt1 = time.perf_counter() - t2
    cpu_model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(nsample / dt, 1), "unit": "queries/s", "cores": r.threads, "kind": "port",
            "sample": f"{nsample} queries of the same workload, same index/rerankK, OpenMP one query per thread "
                      f"({dt:.1f}s); C restatement of jVector's search (real jVector needs a JVM: not available)",
            "single_thread_qps": round(max(8, min(64, nsample)) / dt1, 1), "cpu_model": cpu_model,
            "host_threads": cores, "gpu_ids_equal_oracle_on_sample": same}


if __name__ == "__main__":
    main()

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
BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md: ~2.5 PF dense; the 5 PF headline is 2:1 sparse)
# the reference's only published numbers (README.md:90-95, benchmark-jmh/.../FormatBenchmarkQueryWithRandomVectors.java): ms/op of
# ONE thread issuing the same K = 100 query (rerankK = K * 5 through the plain KnnFloatVectorQuery re-wrap), L2, d = 128
JMH_PUBLISHED_MS = {("fp32", 1000): 0.146, ("fp32", 10000): 0.332, ("fp32", 100000): 0.451,
                    ("pq", 1000): 0.147, ("pq", 10000): 0.181, ("pq", 100000): 0.194}

WORKLOADS = {
    # BASELINE.json configs[1]
    "c2": dict(desc="1Mx768 dot-product fp32 resident, rerankK swept for recall@10>=0.95", n=1_000_000, d=768, sim=1,
               pq_M=0, normalize=True),
    # SURVEY 8(d)'s own distribution B at the headline's size, on the EXACT provider (no PQ: 32-byte codes cannot rank it —
    # recall@10 0.10 at rerankK 900): one number on the survey's distribution next to the rotated headline
    "c2b": dict(desc="10Mx768 L2 fp32 exact provider (no PQ) on SURVEY 8(d) distribution B (4 096-centre Gaussian mixture)",
                n=10_000_000, d=768, sim=0, pq_M=0, normalize=False, dist="mixtureB"),
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
    # the reference's JMH shape (README.md:90-95): handled by jmh_workload() — latency of one query per call from one thread
    "jmh": dict(desc="1k/10k/100k x 128 java.util.Random(42) vectors, L2, K=100, rerankK=500, one query per jv_search call from one thread",
                n=100_000, d=128, sim=0, pq_M=0, normalize=False),
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


def make_rotation(torch, d, device, seed=45):
    """fixed random orthogonal matrix (QR of a Gaussian): multiplying the block model by it removes the alignment
    between the latent factors and the PQ subspaces while keeping every distance and the intrinsic dimension"""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    a = torch.randn((d, d), generator=g, dtype=torch.float64)
    q, r = torch.linalg.qr(a)
    q = q * torch.sign(torch.diagonal(r))[None, :]
    return q.to(torch.float32).to(device)


def gen_rows_mixture(torch, n, d, seed, row_offset, cen, sigma, normalize, device):
    """SURVEY 8(d) distribution B: Gaussian mixture, centre[i mod C] + sigma * N(0, I)"""
    out = torch.empty((n, d), dtype=torch.float32, device=device)
    g = torch.Generator(device=device)
    chunk = 1 << 17
    C = cen.shape[0]
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        g.manual_seed(seed * 1_000_003 + row_offset + s)
        idx = (torch.arange(s, s + m, device=device) + row_offset) % C
        x = cen[idx] + sigma * torch.randn((m, d), generator=g, device=device, dtype=torch.float32)
        if normalize:
            x = x / x.norm(dim=1, keepdim=True)
        out[s:s + m] = x
    return out


def gen_rows_block(torch, n, d, seed, row_offset, zc, Bl, Bg, sigma, gscale, iso, normalize, device, rot=None):
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
        if rot is not None:
            x = x @ rot
        if normalize:
            x = x / x.norm(dim=1, keepdim=True)
        out[s:s + m] = x
    return out


DISTS = ("rotated", "aligned", "mixtureB", "uniformA")


def _i64(v):
    """a 64-bit constant as torch's signed int64 sees it"""
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def splitmix_uniform_torch(torch, seed, n, d, row_offset, device, normalize=False):
    """SURVEY 8(d) distribution A, the reference's own data shape (i.i.d. uniform [0,1)^d: B/FormatBenchmarkQueryWithRandomVectors.java:77-86
    draws java.util.Random floats) from the counter-based generator of opensearch_jvector_amd.datagen.splitmix_uniform — element (i, j)
    depends only on (seed, (row_offset + i) d + j) — computed in HBM with int64 arithmetic (wrapping multiplies, logical shifts spelled
    out): bit-equal to the numpy generator (tests/test_abi_and_datagen.py)."""
    srl = lambda x, sh: (x >> sh) & ((1 << (64 - sh)) - 1)
    out = torch.empty((n, d), dtype=torch.float32, device=device)
    cols = torch.arange(d, device=device, dtype=torch.int64)[None, :]
    add = _i64(seed * 0xD1342543DE82EF95)
    chunk = 1 << 17
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        idx = (torch.arange(s, s + m, device=device, dtype=torch.int64)[:, None] + row_offset) * d + cols
        z = idx + add + _i64(0x9E3779B97F4A7C15)
        z = (z ^ srl(z, 30)) * _i64(0xBF58476D1CE4E5B9)
        z = (z ^ srl(z, 27)) * _i64(0x94D049BB133111EB)
        z = z ^ srl(z, 31)
        x = srl(z, 40).to(torch.float32) / float(1 << 24)
        if normalize:
            x = x / x.norm(dim=1, keepdim=True)
        out[s:s + m] = x
    return out


def make_pq_data(torch, dist_name, n, nq, d, pq_M, row_offset, n_total, normalize, device):
    """base + query rows of one of the three PQ-workload distributions (all generated in HBM):
       aligned  — product-structured latent model whose blocks coincide with the PQ subspaces (round-1 data; the
                  most favourable input a product quantiser can get);
       rotated  — the same model times a fixed random orthogonal matrix: same distances, same intrinsic dimension
                  (64 + 8), no alignment with the subspaces (DEFAULT);
       mixtureB — SURVEY 8(d) distribution B: 4 096-centre Gaussian mixture, centres ~ U[0,1)^d, sigma = 0.05
                  (full-rank isotropic noise inside a cluster: not rankable by 32-byte codes, reported for honesty);
       uniformA — SURVEY 8(d) distribution A, the reference benchmark's own data shape: i.i.d. uniform [0,1)^d, base seed 42,
                  queries seed 43 (no structure at all: in 768 dimensions every point is about equally far from every other,
                  so neither the graph's beam nor 32-byte codes can rank it — reported for honesty as well)."""
    centres = max(64, min(4096, n_total // 256))
    if dist_name == "uniformA":
        return (splitmix_uniform_torch(torch, 42, n, d, row_offset, device, normalize),
                splitmix_uniform_torch(torch, 43, nq, d, 0, device, normalize))
    if dist_name == "mixtureB":
        g = torch.Generator(device=device)
        g.manual_seed(44)
        cen = torch.rand((centres, d), generator=g, device=device, dtype=torch.float32)
        base = gen_rows_mixture(torch, n, d, 42, row_offset, cen, 0.05, normalize, device)
        queries = gen_rows_mixture(torch, nq, d, 43, 0, cen, 0.05, normalize, device)
        return base, queries
    sigma = float(os.environ.get("JV_BENCH_SIGMA", "0.35"))
    # 64 latent factors in total (2 per subspace at M = 32, 1 at M = 64): the same intrinsic dimension for C3 and C4
    zc, Bl, Bg = make_block_generators(torch, d, device, centres, M=pq_M, per=max(1, 64 // pq_M))
    rot = make_rotation(torch, d, device) if dist_name == "rotated" else None
    base = gen_rows_block(torch, n, d, 42, row_offset, zc, Bl, Bg, sigma, 0.1, 0.005, normalize, device, rot=rot)
    queries = gen_rows_block(torch, nq, d, 43, 0, zc, Bl, Bg, sigma, 0.1, 0.005, normalize, device, rot=rot)
    return base, queries


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
SWEEP = list(range(20, 200, 10)) + [200, 225, 250, 300, 350, 400, 500, 600, 700, 800, 900, 1000, 1100, 1200, 1300, 1400, 1600, 1800, 2000, 2400, 2800, 3200]
JV_FLAG_FAILED = 0x40000000
JV_FLAG_OVERFLOW = 0x80000000


def roofline_object(bytes_per_launch, kernel_avg_ms, call_avg_ms, traffic, traffic_source, main_kernel, pq_M, fused):
    """the `roofline` object of the bench line: algorithmic bytes of one step's call (SURVEY 8(d) per-query figure x the step's
    queries) / the duration of the WHOLE call on the device (first search launch + visited-count kernels + the launches that redo
    flagged rows), against the 8 TB/s HBM peak.  The bytes are counted from the counters of every row, whichever launch finished
    it, so the time must cover every launch too: round 5 divided them by the first launch's time alone, and a distribution whose
    rows mostly finish in later rungs printed a fraction above 1.  The first launch's own duration
    (HIP events on its stream, option time_search_kernel) is kept beside it as a share of the call, not as a fraction of peak."""
    achieved_gbs = bytes_per_launch / (call_avg_ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(achieved_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved_gbs / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
            "kernel": main_kernel, "call_avg_ms": round(call_avg_ms, 4),
            "time_source": "HIP events around the whole device call on the stream it is enqueued on (every launch of the step)",
            "frac_whole_call": round(achieved_gbs / HBM_PEAK_GBS, 4),
            "first_launch": {"kernel_avg_ms": round(kernel_avg_ms, 4),
                             "share_of_call": round(min(1.0, kernel_avg_ms / call_avg_ms), 4) if call_avg_ms > 0 else None,
                             "time_source": "HIP events around the call's first search launch, recorded by the library on the launch stream (option time_search_kernel)"},
            "kernel_avg_ms": round(kernel_avg_ms, 4),
            "algorithmic_bytes_per_launch": round(bytes_per_launch, 1),
            "formula": ("expanded*R*(M+4) + reranked*4d + 1024d/B" if fused else
                        ("visited*M + expanded*4(R+1) + reranked*4d + 1024d/B" if pq_M else "visited*4d + expanded*4(R+1)"))}


def algorithmic_bytes(visited, reranked, expanded, nq_total, launches, pq_M, d, R, fused):
    """SURVEY 8(d), summed over `nq_total` queries issued in `launches` launches.
       fp32 path:              visited*4d + expanded*4(R+1)
       PQ, plain layout:       visited*M + expanded*4(R+1) + reranked*4d + LUT/B
       PQ, fused layout:       expanded*R*(M+4)            + reranked*4d + LUT/B
    LUT = 1024*d bytes of codebook; a launch's queries share that read (every wave builds its own table from the
    L2-resident codebook), so it is counted ONCE per launch = LUT_q / B per query."""
    if not pq_M:
        return visited * 4.0 * d + expanded * 4.0 * (R + 1)
    lut = launches * 1024.0 * d
    if fused:
        return expanded * float(R) * (pq_M + 4) + reranked * 4.0 * d + lut
    return visited * float(pq_M) + expanded * 4.0 * (R + 1) + reranked * 4.0 * d + lut


class Engine:
    """one built index + its device-resident search plumbing"""

    def __init__(self, torch, dist, binding, sharding, device, local_rank, world, backend, base, queries, sim, R, L, k,
                 pq_M, row_offset, n_total, builder_kind, builder_mod, B, n_gt, fused, replicas=1, replica_rank=0):
        self.torch, self.dist, self.binding, self.sharding = torch, dist, binding, sharding
        self.device, self.local_rank, self.world, self.backend = device, local_rank, world, backend
        self.base, self.queries, self.sim, self.R, self.k, self.pq_M = base, queries, sim, R, k, pq_M
        self.row_offset, self.n_total, self.B, self.n_gt, self.fused = row_offset, n_total, B, n_gt, fused
        # "replicas" (the right choice whenever the index fits one GPU): `world` is 1 for everything the engine does —
        # whole index, no gather, no merge — and `replicas` ranks answer DIFFERENT batches of queries side by side
        self.replicas, self.replica_rank = replicas, replica_rank
        n, d = base.shape
        self.n, self.d = n, d
        t0 = time.time()
        pq = None
        if builder_kind == "gpu":
            gbuild = importlib.import_module("opensearch_jvector_amd.builder_gpu")
            adj_t, entry = gbuild.build_graph_gpu(torch, base, sim, R=R, L=L, alpha=1.2, device_index=local_rank,
                                                  refine_passes=int(os.environ.get("JV_BENCH_REFINE", "0")))
            if pq_M:
                pq = gbuild.pq_train_encode_gpu(torch, base, pq_M, sim)
        else:
            host = base.cpu().numpy()
            adj_np, entry = builder_mod.build_graph_cpu(host, sim, R=R, L=L)
            adj_t = torch.from_numpy(adj_np).to(device)
            if pq_M:
                cb, cenq, codes, K = builder_mod.pq_train_encode_cpu(host, pq_M, sim)
                pq = dict(codebooks=cb, centroid=cenq, codes=torch.from_numpy(codes).to(device), K=K)
            del host
        self.adj_t, self.entry, self.pq = adj_t, entry, pq
        self.ord2doc = (torch.arange(n, device=device, dtype=torch.int32) + row_offset) if world > 1 else None
        torch.cuda.synchronize()
        self.build_s = time.time() - t0
        desc, keep = binding.make_desc_device(
            n, d, R, base.data_ptr(), adj_t.data_ptr(), entry, sim, device=local_rank,
            pq_M=pq_M, pq_K=(pq["K"] if pq else 0), pq_codebooks=(pq["codebooks"] if pq else None),
            pq_centroid=(pq["centroid"] if pq else None), pq_codes_ptr=(pq["codes"].data_ptr() if pq else 0),
            ord2doc_ptr=(self.ord2doc.data_ptr() if self.ord2doc is not None else 0), max_doc=n_total, borrow=True,
            extra_flags=(binding.DESC_FUSED_ADC if (pq and fused) else 0))
        self.index = binding.GpuIndex(desc=desc, keepalive=keep, flags=binding.DESC_BORROW)
        # ---- search plumbing: everything device-resident, own stream ----
        self.stream = torch.cuda.Stream(device=device)
        OB = max(B, n_gt)  # output rows: a timed step uses the first B, the recall sweep the first n_gt
        self.out_nodes = torch.empty((OB, k), dtype=torch.int32, device=device)
        self.out_docs = torch.empty((OB, k), dtype=torch.int32, device=device)
        self.out_scores = torch.empty((OB, k), dtype=torch.float32, device=device)
        self.out_count = torch.empty((OB,), dtype=torch.int32, device=device)
        self.out_stats = torch.empty((OB, 4), dtype=torch.int32, device=device)
        self.out_flags = torch.empty((OB,), dtype=torch.int32, device=device)
        if world > 1:
            self.gather_buf = torch.empty((world, B, k, 2), dtype=torch.int32, device=device)
            self.merged_docs = torch.empty((OB, k), dtype=torch.int32, device=device)
            self.merged_scores = torch.empty((OB, k), dtype=torch.float32, device=device)

    def close(self):
        self.index.close()

    def launch(self, q, nq, rk):
        """enqueue one pass of the hot path on this engine's stream (no host sync)"""
        self.index.search_batch_device(q.data_ptr(), nq, self.k, rk, self.out_nodes.data_ptr(), self.out_docs.data_ptr(),
                                       self.out_scores.data_ptr(), self.out_count.data_ptr(), self.out_stats.data_ptr(),
                                       self.out_flags.data_ptr(), stream=self.stream.cuda_stream)

    def gpu_merge(self, gd, gs, kk):
        nqm = gd.shape[0]
        self.binding.merge_topk_device(self.local_rank, gd.data_ptr(), gs.data_ptr(), nqm, self.world, kk,
                                       self.merged_docs.data_ptr(), self.merged_scores.data_ptr(),
                                       stream=self.stream.cuda_stream)
        return self.merged_docs[:nqm], self.merged_scores[:nqm]

    def run_step(self, qbatch, rk, nq=None):
        """one pass of the hot path over one batch; returns the final (docs, scores) tensors"""
        nq = self.B if nq is None else nq

        def local_search(q):
            self.launch(q, nq, rk)
            return self.out_docs[:nq], self.out_scores[:nq]
        with self.torch.cuda.stream(self.stream):
            return self.sharding.sharded_search(self.dist, self.torch, local_search, self.gpu_merge, qbatch, self.k, self.world)

    def ground_truth(self):
        torch, dist, world, k, sim = self.torch, self.dist, self.world, self.k, self.sim
        base, queries, n_gt, row_offset, n = self.base, self.queries, self.n_gt, self.row_offset, self.n
        gt_local = brute_force_topk(torch, base, queries[:n_gt], k, sim, row_offset)
        if world == 1:
            return gt_local
        # global ground truth = merge of per-shard exact top-k
        qn = queries[:n_gt]
        v = base[(gt_local - row_offset).clamp(0, n - 1)]
        ls = -((qn[:, None, :] - v) ** 2).sum(-1) if sim == 0 else (qn[:, None, :] * v).sum(-1)
        if self.backend == "nccl":
            all_i = [torch.empty_like(gt_local) for _ in range(world)]
            all_s = [torch.empty_like(ls) for _ in range(world)]
            dist.all_gather(all_i, gt_local)
            dist.all_gather(all_s, ls)
        else:
            all_i = [torch.empty_like(gt_local, device="cpu") for _ in range(world)]
            all_s = [torch.empty_like(ls, device="cpu") for _ in range(world)]
            dist.all_gather(all_i, gt_local.cpu())
            dist.all_gather(all_s, ls.cpu())
            all_i, all_s = [t.to(self.device) for t in all_i], [t.to(self.device) for t in all_s]
        ci, cs = torch.cat(all_i, 1), torch.cat(all_s, 1)
        return torch.gather(ci, 1, cs.topk(k, dim=1).indices)

    def sweep(self, gt, candidates, target=0.95):
        """smallest rerankK of `candidates` whose recall@k reaches `target` (the last one tried otherwise)"""
        chosen, rec, slog = None, 0.0, []
        for rk in candidates:
            docs, _ = self.run_step(self.queries[:self.n_gt], rk, nq=self.n_gt)
            self.stream.synchronize()
            self.check_flags(self.n_gt)
            rec = recall_of(docs[:self.n_gt], gt)
            slog.append((rk, round(rec, 4)))
            chosen = rk
            if rec >= target:
                break
        return chosen, rec, slog

    def check_flags(self, nq):
        """a query that exhausted even the HBM-scratch rung has no valid row: never let it into a reported number"""
        fl = self.out_flags[:nq].cpu().numpy().astype(np.uint32)
        bad = int(((fl & np.uint32(JV_FLAG_FAILED | JV_FLAG_OVERFLOW)) != 0).sum())
        if bad:
            raise SystemExit(f"bench: {bad} of {nq} queries were flagged FAILED/OVERFLOW by the engine (results invalid)")
        return int((fl & np.uint32(1)).sum())

    def timed(self, rk, steps, warmup, barrier, batch=None):
        torch, dist, world, k = self.torch, self.dist, self.world, self.k
        B = self.B if batch is None else min(batch, self.B)
        nq_pool = self.queries.shape[0]
        batches = [self.queries[i * B:(i + 1) * B] for i in range(nq_pool // B)]
        if self.replicas > 1:   # every replica starts at another batch of the pool
            r0 = self.replica_rank % len(batches)
            batches = batches[r0:] + batches[:r0]
        for w in range(warmup):
            self.run_step(batches[w % len(batches)], rk)
        barrier()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        stat_sums = torch.zeros((4,), dtype=torch.int64, device=self.device)
        bad = torch.zeros((1,), dtype=torch.int64, device=self.device)
        # (the bookkeeping below runs its torch kernels once before the clock starts: the warm-up steps go through run_step and do
        #  not touch them, and their first use loads code objects — 70 ms that landed inside C4's ten timed steps)
        with torch.cuda.stream(self.stream):
            stat_sums += self.out_stats[:B].to(torch.int64).sum(0)
            bad += ((self.out_flags[:B] & JV_FLAG_FAILED) != 0).sum() + (self.out_flags[:B] < 0).sum()
            stat_sums.zero_()
            bad.zero_()
        # the dominant kernel's own duration: HIP events around the call's first search launch, recorded inside the library on the
        # stream it launches on (option time_search_kernel); the events around the whole call also contain the visited-count
        # kernels and the launch that redoes flagged rows
        self.index.set_option("time_search_kernel", 1)
        kt0 = (self.index.counter("search_kernel_ns"), self.index.counter("search_kernel_timed"))
        barrier()
        t_start = time.perf_counter()
        for s in range(steps):
            evs[s][0].record(self.stream)
            self.launch(batches[s % len(batches)], B, rk)
            evs[s][1].record(self.stream)
            with torch.cuda.stream(self.stream):
                if world > 1:
                    gd, gs = self.sharding.gather_topk(dist, torch, self.out_docs[:B], self.out_scores[:B], world, self.gather_buf)
                    self.gpu_merge(gd, gs, k)
                stat_sums += self.out_stats[:B].to(torch.int64).sum(0)  # per-query counters -> algorithmic bytes
                bad += ((self.out_flags[:B] & JV_FLAG_FAILED) != 0).sum() + (self.out_flags[:B] < 0).sum()
        barrier()
        elapsed = time.perf_counter() - t_start
        if world > 1 or self.replicas > 1:
            tmax = torch.tensor([elapsed], device=(self.device if self.backend == "nccl" else "cpu"), dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        if int(bad.item()):
            raise SystemExit(f"bench: {int(bad.item())} queries of the timed steps were flagged FAILED/OVERFLOW (results invalid)")
        call_avg_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
        kt1 = (self.index.counter("search_kernel_ns"), self.index.counter("search_kernel_timed"))
        self.index.set_option("time_search_kernel", 0)
        kernel_avg_ms = (kt1[0] - kt0[0]) / (kt1[1] - kt0[1]) * 1e-6 if kt1[1] - kt0[1] == steps else call_avg_ms
        big = self.check_flags(B)
        st = stat_sums.cpu().numpy().astype(np.float64)
        # (replicas: the job's queries = every replica's; the counters and kernel times below stay THIS rank's — the roofline is per GPU)
        return dict(elapsed=elapsed, qps=steps * B * self.replicas / elapsed, kernel_avg_ms=kernel_avg_ms, call_avg_ms=call_avg_ms, visited=st[0], reranked=st[1],
                    expanded=st[2], total_queries=steps * B, big_path_last_step=big)

    def timed_in_flight(self, rk, steps, streams):
        """small-batch workloads (c5): `streams` batches of B queries in flight at once — successive batches of a server arrive
        while the previous ones are still being answered; every stream is an independent caller with its own outputs.  A
        report next to `value` (which stays one batch at a time), never `value` itself."""
        torch, B, k, dev = self.torch, self.B, self.k, self.device
        nq_pool = self.queries.shape[0]
        batches = [self.queries[i * B:(i + 1) * B] for i in range(nq_pool // B)]
        ss = [torch.cuda.Stream(device=dev) for _ in range(streams)]
        outs = [[torch.empty((B, k), dtype=torch.int32, device=dev), torch.empty((B, k), dtype=torch.int32, device=dev),
                 torch.empty((B, k), dtype=torch.float32, device=dev), torch.empty((B,), dtype=torch.int32, device=dev),
                 torch.empty((B, 4), dtype=torch.int32, device=dev), torch.zeros((B,), dtype=torch.int32, device=dev)] for _ in range(streams)]
        bad = [torch.zeros((1,), dtype=torch.int64, device=dev) for _ in range(streams)]

        def go(n_steps):
            for s_ in range(n_steps):
                for i in range(streams):
                    q = batches[(s_ * streams + i) % len(batches)]
                    self.index.search_batch_device(q.data_ptr(), B, k, rk, *[t.data_ptr() for t in outs[i]], stream=ss[i].cuda_stream)
                    with torch.cuda.stream(ss[i]):
                        bad[i] += ((outs[i][5] & JV_FLAG_FAILED) != 0).sum() + (outs[i][5] < 0).sum()
        go(2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        go(steps)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if sum(int(b_.item()) for b_ in bad):
            raise SystemExit("bench: queries of the in-flight leg were flagged FAILED/OVERFLOW (results invalid)")
        return dict(batches_in_flight=streams, qps=round(steps * streams * B / el, 1), ms_per_batch=round(el * 1e3 / steps, 3))


def jmh_workload(args):
    """`--workload jmh`: the reference's published benchmark shape.  FormatBenchmarkQueryWithRandomVectors: numDocs x 128 floats from
    new Random(42).nextFloat(), the query = the next 128 draws of the same stream, EUCLIDEAN, K = 100 through a plain
    KnnFloatVectorQuery (the reader re-wraps the collector with over-query 5 -> rerankK 500, J/JVectorReader.java:133-144), one
    thread, the same query every op.  jvector_quantized = PQ with the default subspaces for 128-d (64,
    J/JVectorIndexQuantization.java:428-446) once numDocs >= 1 024 (DEFAULT_MINIMUM_BATCH_SIZE_FOR_QUANTIZATION) — the
    1 000-doc row of that codec is NOT quantized.  Here: the same vectors (datagen.java_random_*), graphs from the sequential CPU
    builder (R = 32, ef_construction = 100), one query per jv_search call from one thread; beside it the CPU port on one thread.
    Hardware differs (the README does not name its machine): vs_baseline is a ratio of ms/op, reported with that caveat."""
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    torch.cuda.set_device(0)
    graft.load_package()
    binding = importlib.import_module("opensearch_jvector_amd.binding")
    builder = importlib.import_module("opensearch_jvector_amd.builder")
    datagen = importlib.import_module("opensearch_jvector_amd.datagen")
    pyoracle = graft.load_oracle()
    K, oqf, d = 100, 5, 128
    sizes = [int(x) for x in os.environ.get("JV_BENCH_JMH_SIZES", "1000,10000,100000").split(",")]
    ops = int(os.environ.get("JV_BENCH_JMH_OPS", "2000"))
    rows = []
    for n in sizes:
        stream = datagen.java_random_floats(42, (n + 1) * d)
        base, query = stream[:n * d].reshape(n, d), stream[n * d:]
        d2 = ((base.astype(np.float64) - query.astype(np.float64)) ** 2).sum(1)
        kth = np.sort(1.0 / (1.0 + d2))[::-1][min(K, n) - 1]   # BenchmarkCommon.findExpectedKthMaxScore
        for codec in ("fp32", "pq"):
            pq_M = 64 if (codec == "pq" and n >= 1024) else 0
            t0 = time.time()
            ix = builder.build_index_cpu(base, 0, R=32, L=100, pq_M=pq_M)
            build_s = time.time() - t0
            gpu = binding.GpuIndex(ix, device=0, flags=(binding.DESC_FUSED_ADC if pq_M else 0))
            orc = pyoracle.Oracle(binding, ix)
            want = orc.search_batch(query[None, :], K, K * oqf, threads=1)
            for _ in range(200):
                got = gpu.search(query, K, K * oqf)
            lat = []
            for _ in range(ops):
                t1 = time.perf_counter()
                got = gpu.search(query, K, K * oqf)
                lat.append((time.perf_counter() - t1) * 1e3)
            same = bool(np.array_equal(got.nodes[0], want.nodes[0]) and np.array_equal(got.scores[0].view(np.uint32), want.scores[0].view(np.uint32))
                        and np.array_equal(got.stats[0], want.stats[0]))
            recall = float((got.scores[0][:got.count[0]] >= np.float32(kth) * (1 - 1e-6)).mean())   # BenchmarkCommon.calculateRecall
            cpu_ops = max(50, min(ops, 500))
            orc.search_batch(query[None, :], K, K * oqf, threads=1)
            t1 = time.perf_counter()
            for _ in range(cpu_ops):
                orc.search_batch(query[None, :], K, K * oqf, threads=1)
            cpu_ms = (time.perf_counter() - t1) * 1e3 / cpu_ops
            ref = JMH_PUBLISHED_MS[(codec, n)] if (codec, n) in JMH_PUBLISHED_MS else None
            gpu_ms = float(np.mean(lat))
            rows.append({"codec": "jvector_not_quantized" if codec == "fp32" else "jvector_quantized", "numDocs": n, "dimension": d,
                         "pq_M": pq_M, "gpu_ms_per_op": round(gpu_ms, 4), "gpu_p50_ms": round(float(np.median(lat)), 4),
                         "cpu_port_1_thread_ms_per_op": round(cpu_ms, 4), "reference_published_ms_per_op": ref,
                         "vs_baseline": (None if ref is None else round(gpu_ms / ref, 3)),
                         "cpu_port_vs_published": (None if ref is None else round(cpu_ms / ref, 3)),
                         "recall_jmh_definition": round(recall, 4), "gpu_equals_oracle": same,
                         "served_by_query_server": int(gpu.counter("served_queries")), "build_seconds": round(build_s, 1)})
            log(f"jmh {rows[-1]}")
            gpu.close()
    head = [r for r in rows if r["numDocs"] == max(sizes) and r["codec"] == "jvector_not_quantized"][0]
    cpu_model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    result = {"metric": "ms/op, FormatBenchmarkRandomVectors.benchmarkSearch shape (one thread, one query per call, K=100, rerankK=500)",
              "value": head["gpu_ms_per_op"], "unit": "ms/op", "n_gpus": 1, "steps": ops, "warmup": 200, "ms_per_step": head["gpu_ms_per_op"],
              "higher_is_better": False, "scaling": "strong", "vs_baseline": head["vs_baseline"], "dtype": "f32",
              "data": "synthetic (java.util.Random(42).nextFloat(), the reference benchmark's own generator)",
              "config": {"workload": f"jmh: {WORKLOADS['jmh']['desc']}", "k": K, "rerankK": K * oqf, "R": 32, "ef_construction": 100,
                         "similarity": "l2", "graph_builder": "cpu (sequential insertion)"},
              "rows": rows,
              "note": "vs_baseline = this engine's ms/op / the README's ms/op on unnamed hardware (one JVM thread, Lucene IndexSearcher on top); "
                      "a lone query is a chain of dependent expansions, the GPU's worst case — the CPU port's one-thread figure on "
                      f"this box's host ({cpu_model}) is printed beside it",
              "cpu_baseline": {"value": head["cpu_port_1_thread_ms_per_op"], "unit": "ms/op", "cores": 1, "kind": "port",
                               "sample": f"{max(50, min(ops, 500))} repeats of the same query on the same index"}}
    print(json.dumps(result), flush=True)


def exact_batch_report(torch, binding, eng, k, rk, B=256, sels=(0.5, 0.2, 0.1, 0.01, 0.001), graph_min_sel=0.1):
    """BASELINE.json configs[4] where the contraction is real: B concurrent queries under ONE doc filter.  Per selectivity: the
    batched exact scorer (jv_score_ordinals_batch_device: bf16 MFMA pre-filter over the bf16 mirror + canonical fp32 re-score;
    answers = the exact top k) timed with HIP events on the launch stream, beside the graph search of the same queries under the
    same filter (jv_search_batch_device, rerankK of the headline) and the recall of the graph's answer against the exact one.
    The roofline of the exact call is quoted both ways: bf16 flops of the tile passes vs the dense bf16 peak, and the bytes of
    the gathered mirror rows vs 8 TB/s (whole call: sample pass + bar + filter pass + re-score)."""
    dev, n, d = eng.device, eng.n, eng.d
    q = eng.queries[:B].contiguous()
    o = [torch.empty((B, k), dtype=torch.int32, device=dev), torch.empty((B, k), dtype=torch.int32, device=dev),
         torch.empty((B, k), dtype=torch.float32, device=dev), torch.empty((B,), dtype=torch.int32, device=dev)]
    go = [torch.empty((B, k), dtype=torch.int32, device=dev), torch.empty((B, k), dtype=torch.int32, device=dev),
          torch.empty((B, k), dtype=torch.float32, device=dev), torch.empty((B,), dtype=torch.int32, device=dev),
          torch.empty((B, 4), dtype=torch.int32, device=dev), torch.zeros((B,), dtype=torch.int32, device=dev)]
    st = torch.cuda.Stream(device=dev)
    kp = (d + 63) // 64 * 64
    rows, checks = [], []
    sample_q = torch.arange(0, B, max(1, B // 16), device=dev)[:16]   # the queries the oracle re-answers afterwards
    sample_q_np = sample_q.cpu().numpy()
    rng = np.random.default_rng(7)
    t0 = time.time()
    eng.index.score_ordinals_batch_device(q.data_ptr(), 1, k, *[t.data_ptr() for t in o], d_ordinals=0, count=0)  # builds the mirror
    torch.cuda.synchronize()
    mirror_s = time.time() - t0
    for sel in sels:
        bits = rng.random(n) < sel
        words = np.packbits(bits, bitorder="little")
        words = np.concatenate([words, np.zeros((-len(words)) % 8, np.uint8)]).view(np.int64)
        acc = torch.from_numpy(words).to(dev)
        with torch.cuda.stream(st):
            info = eng.index.score_ordinals_batch_device(q.data_ptr(), B, k, *[t.data_ptr() for t in o], d_accept=acc.data_ptr(),
                                                         accept_num_docs=n, stream=st.cuda_stream, want_info=True)
            reps = 5
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(reps):
                eng.index.score_ordinals_batch_device(q.data_ptr(), B, k, *[t.data_ptr() for t in o], d_accept=acc.data_ptr(),
                                                      accept_num_docs=n, stream=st.cuda_stream)
            e1.record(st)
        st.synchronize()
        ms = e0.elapsed_time(e1) / reps
        C, S = int(info[0]), int(info[1])
        row = {"selectivity": sel, "candidates": C, "queries_per_batch": B, "exact_ms_per_batch": round(ms, 4), "exact_qps": round(B / ms * 1e3, 1),
               "rows_rescored_per_query": round(info[2] / B, 1), "queries_overflowed": int(info[3]), "prefilter_sample_rows": S}
        if S > 0:
            flops = 2.0 * B * (C + S) * kp
            byts = float(C + S) * kp * 2 + info[2] * 4.0 * d
            row["roofline_flops"] = {"bound": "mfma", "achieved": round(flops / ms / 1e9, 2), "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                     "frac": round(flops / ms / 1e9 / BF16_PEAK_TFLOPS, 4)}
            row["roofline_bytes"] = {"bound": "hbm", "achieved": round(byts / ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": round(byts / ms / 1e6 / HBM_PEAK_GBS, 4),
                                     "formula": "(candidates + sample) * 2 * roundup(d, 64) + rows_rescored * 4d; whole call, HIP events"}
        if sel >= graph_min_sel:
            with torch.cuda.stream(st):
                eng.index.search_batch_device(q.data_ptr(), B, k, rk, *[t.data_ptr() for t in go], stream=st.cuda_stream,
                                              d_accept=acc.data_ptr(), accept_num_docs=n)
                g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                g0.record(st)
                eng.index.search_batch_device(q.data_ptr(), B, k, rk, *[t.data_ptr() for t in go], stream=st.cuda_stream,
                                              d_accept=acc.data_ptr(), accept_num_docs=n)
                g1.record(st)
            st.synchronize()
            gms = g0.elapsed_time(g1)
            gn, xn = go[0].cpu().numpy(), o[0].cpu().numpy()
            stats = go[4].cpu().numpy().astype(np.float64)
            row.update({"graph_ms_per_batch": round(gms, 3), "graph_qps": round(B / gms * 1e3, 1), "graph_rerankK": rk,
                        "graph_recall_at_10_vs_exact": round(float(np.mean([len(set(gn[i]) & set(xn[i])) / k for i in range(B)])), 4),
                        "graph_visited_plus_expanded_mean": round(float((stats[:, 0] + stats[:, 2]).mean()), 1),
                        "lucene_would_discard_graph_result": bool(((stats[:, 0] + stats[:, 2]) >= C).mean() > 0.5)})
        # kept for the oracle check (cpu_baseline leg, exact_batch_oracle_check): the filter and the answers of a sample of the queries
        row["exact_equals_oracle_on_sample"] = None   # (None = not checked: no CPU leg in this run)
        checks.append({"selectivity": sel, "words": words.view(np.uint64).copy(), "sample": sample_q_np,
                       "nodes": o[0][sample_q].cpu().numpy(), "scores": o[2][sample_q].cpu().numpy(), "count": o[3][sample_q].cpu().numpy()})
        rows.append(row)
        log(f"exact batch: {row}")
        del acc
    return {"what": "B queries under one doc filter: batched exact scorer (bf16 MFMA pre-filter + fp32 re-score, answers = exact top k) "
                    "beside the graph search under the same filter", "kernel": "jvx_qs_kernel / jvx_tile_kernel (v_mfma_f32_32x32x16_bf16)",
            "mirror_build_seconds": round(mirror_s, 2), "rows": rows, "_checks": checks}


def exact_batch_oracle_check(report, orc, queries_np, k, threads):
    """Every row of exact_batch_report against the oracle's exact scan (jvo_brute_force: JVectorVectorScorer.score over the accepted
    ordinals + (score desc, ordinal asc), J/JVectorVectorScorer.java:36-53) on a sample of the batch's queries: ids, order, score
    bits and count.  Part of the cpu_baseline leg (the only place bench.py may call the oracle); fills
    `exact_equals_oracle_on_sample` of every row."""
    checks = report.pop("_checks", [])
    all_ok = True
    t0 = time.time()
    for row, c in zip(report["rows"], checks):
        qs = queries_np[c["sample"]]
        wn, ws = orc.brute_force(qs, k, accept=c["words"], threads=threads)
        cnt = (wn >= 0).sum(1)
        ok = bool(np.array_equal(c["nodes"], wn) and np.array_equal(c["scores"].view(np.uint32), ws.view(np.uint32)) and
                  np.array_equal(c["count"], cnt))
        row["exact_equals_oracle_on_sample"] = ok
        row["oracle_sample_queries"] = int(len(c["sample"]))
        all_ok = all_ok and ok
    report["exact_equals_oracle_on_sample"] = all_ok if checks else None
    report["oracle_check"] = (f"{len(checks)} selectivities x {len(checks[0]['sample']) if checks else 0} queries of the batch against the oracle's exact "
                              f"scan over the accepted ordinals (ids, order, score bits, count), {time.time() - t0:.1f}s")
    log(f"exact batch vs oracle: {[(r['selectivity'], r['exact_equals_oracle_on_sample']) for r in report['rows']]}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=os.environ.get("JV_BENCH_WORKLOAD", "c3"), choices=sorted(WORKLOADS))
    ap.add_argument("--dist", default=os.environ.get("JV_BENCH_DIST", "rotated"), choices=DISTS,
                    help="synthetic distribution of the PQ workloads (c3/c4/c5); c2 has its own low-rank mixture")
    ap.add_argument("--no-dist-comparison", action="store_true",
                    help="skip the rerankK/recall/QPS lines of the two other distributions (c3, N=1 only)")
    ap.add_argument("--n", type=int, default=int(os.environ.get("JV_BENCH_N", "0")), help="docs per GPU (0 = workload default)")
    ap.add_argument("--batch", type=int, default=int(os.environ.get("JV_BENCH_BATCH", "65536")), help="queries per step")
    ap.add_argument("--rerankk", type=int, default=int(os.environ.get("JV_BENCH_RERANKK", "0")), help="0 = sweep for recall>=0.95")
    ap.add_argument("--builder", default=os.environ.get("JV_BENCH_BUILDER", "gpu"), choices=["gpu", "cpu"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--profile-mode", action="store_true",
                    help="only warm-up + timed launches of the query kernel (needs --rerankk): no recall sweep, no p50, no CPU leg")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU under
    # torch.distributed.run) as a CHILD process — before this process touches HIP — relay its output and exit with its code.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        code = 1
        for attempt in range(2):
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            log("launching " + " ".join(cmd))
            t_launch = time.time()
            code = subprocess.call(cmd, env=env)
            # a launcher that dies within seconds never got its ranks together (the probed port was taken in between, a
            # rendezvous hiccup): one more try on a fresh port; anything later is the benchmark's own failure
            if code == 0 or time.time() - t_launch > 20.0:
                break
            log(f"launcher exited with {code} after {time.time() - t_launch:.1f} s: retrying once on a new port")
        raise SystemExit(code)

    if args.workload == "jmh":
        if args.gpus != 1:
            raise SystemExit("bench: --workload jmh is a one-GPU, one-thread latency shape")
        return jmh_workload(args)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("JV_BENCH_LAUNCH_CHECK") == "1":  # (tests/test_bench_launch.py: the launch path alone, no GPU needed)
        # (one write() of the whole line: ranks share the launcher's stdout pipe, and print() may split text and newline)
        sys.stdout.flush()
        os.write(1, f"launch-check rank {rank} of {world} local_rank {local_rank} gpus {args.gpus}\n".encode())
        raise SystemExit(0 if world == args.gpus else 3)

    import torch
    import torch.distributed as dist

    if world != args.gpus:
        raise SystemExit(f"bench: WORLD_SIZE={world} but --gpus {args.gpus}: launch N ranks for --gpus N (or run `python bench.py --gpus N`, which does)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    # one rank per GPU; (debug only: JV_BENCH_BACKEND=gloo lets several ranks share one GPU to exercise the
    # shard / all-gather / merge path on a single-GPU box — the collective then stages through the host)
    backend = os.environ.get("JV_BENCH_BACKEND", "nccl")
    if world > 1 and backend == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit(f"bench: --gpus {world} needs {world} GPUs, this node shows {torch.cuda.device_count()} "
                         "(JV_BENCH_BACKEND=gloo lets ranks share a GPU for a functional check only)")
    local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)
        if dist.get_world_size() != args.gpus or dist.get_backend() != backend:
            raise SystemExit(f"bench: process group is {dist.get_backend()} x {dist.get_world_size()}, expected {backend} x {args.gpus}")

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
    data_M = pq_M  # the synthetic data's block structure follows the workload's subspace count ...
    if pq_M and os.environ.get("JV_BENCH_CODEC_PQ_M"):
        # ... while the CODEC may use another one (e.g. 192 = the plugin's default for 768-d fields; not the BASELINE configs)
        pq_M = int(os.environ["JV_BENCH_CODEC_PQ_M"])
    R, L, k = 32, 100, 10
    fused = bool(pq_M) and os.environ.get("JV_BENCH_FUSED", "1") == "1"
    # N > 1: doc-ID-range shards.  Default = the north star's curve: the SAME corpus (10M docs) split over the
    # ranks ("strong": total work fixed; every rank searches every query on its n/N docs, rerankK is re-swept
    # so that the MERGED recall@10 stays >= 0.95).  JV_BENCH_SCALING=weak keeps n docs per GPU instead.
    # (the label does not depend on N: c2/c3/c5 lines belong to the strong curve — the same corpus on 1, 2, 4, 8 GPUs —,
    #  c4 lines to the weak one)
    # JV_BENCH_SCALING=replicas: every rank holds the WHOLE corpus and answers its own batches of queries, no collective on the data
    # path — what SURVEY 8(e) asks to be reported beside the shards for a corpus that fits one GPU (c2 / c3 / c5).  Work per GPU is
    # fixed as N grows, so the line says "weak"; config.sharding names the mode.
    scaling_mode = os.environ.get("JV_BENCH_SCALING", "weak" if wl.get("per_gpu") else "strong")
    if scaling_mode not in ("strong", "weak", "replicas"):
        raise SystemExit(f"bench: JV_BENCH_SCALING={scaling_mode!r} (strong | weak | replicas)")
    replicas = world if (scaling_mode == "replicas" and world > 1) else 1
    scaling = "weak" if scaling_mode == "replicas" else scaling_mode
    shard_world = 1 if replicas > 1 else world   # ranks one query's answer is merged over
    if replicas > 1:
        n, row_offset, n_total = n_cfg, 0, n_cfg
    elif world > 1 and scaling == "strong":
        lo_doc, hi_doc = sharding.shard_range(n_cfg, world, rank)
        n, row_offset, n_total = hi_doc - lo_doc, lo_doc, n_cfg
    else:
        n, row_offset, n_total = n_cfg, rank * n_cfg, n_cfg * world
    B = args.batch
    n_gt = 1024  # queries with exact ground truth (recall is quoted on these)
    nq_pool = max(B * 2, 4096)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def make_engine(dist_name):
        t0 = time.time()
        if pq_M or wl.get("dist"):
            base, queries = make_pq_data(torch, dist_name, n, nq_pool, d, data_M or 32, row_offset, n_total, wl["normalize"], device)
        else:
            centres = max(64, min(4096, n_total // 256))
            cen, basis = make_generators(torch, d, device, centres, 32)
            base = gen_rows(torch, n, d, 42, row_offset, cen, basis, 0.15, 0.01, wl["normalize"], device)
            queries = gen_rows(torch, nq_pool, d, 43, 0, cen, basis, 0.15, 0.01, wl["normalize"], device)
        torch.cuda.synchronize()
        log(f"rank {rank}: generated {n}x{d} base ({dist_name if pq_M else 'low-rank mixture'}) + {nq_pool} queries in {time.time() - t0:.1f}s")
        eng = Engine(torch, dist, binding, sharding, device, local_rank, shard_world, backend, base, queries, sim, R, L, k, pq_M,
                     row_offset, n_total, args.builder, builder, B, n_gt, fused, replicas=replicas, replica_rank=rank)
        log(f"rank {rank}: built index ({args.builder}) in {eng.build_s:.1f}s, entry={eng.entry}")
        return eng

    dist_name = wl["dist"] if wl.get("dist") else (args.dist if pq_M else "lowrank-mixture")
    eng = make_engine(dist_name)

    # ---- ground truth + rerankK selection (recall@10 >= 0.95) ----
    if args.profile_mode:
        if args.rerankk <= 0:
            raise SystemExit("--profile-mode needs --rerankk")
        rk, chosen_recall, sweep_log = args.rerankk, float("nan"), []
    else:
        gt = eng.ground_truth()
        rk, chosen_recall, sweep_log = eng.sweep(gt, [args.rerankk] if args.rerankk > 0 else SWEEP)
        log(f"recall sweep (rerankK, recall@10): {sweep_log}")
    target_met = bool(chosen_recall >= 0.95) if chosen_recall == chosen_recall else None

    # ---- timed region ----
    t = eng.timed(rk, args.steps, args.warmup, barrier)
    elapsed, qps, kernel_avg_ms, total_queries = t["elapsed"], t["qps"], t["kernel_avg_ms"], t["total_queries"]
    bytes_total = algorithmic_bytes(t["visited"], t["reranked"], t["expanded"], total_queries, args.steps, pq_M, d, R, fused)
    bytes_per_launch = bytes_total / args.steps
    achieved_gbs = bytes_per_launch / (kernel_avg_ms * 1e-3) / 1e9
    traffic = None
    traffic_source = None  # (traffic is NOT measured in this run: it is the kept PMC result of the same command, labelled as such)
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath):
        # HBM bytes per launch from the PMC passes of the same command (tools/profile_bench.sh ->
        # tools/summarize_profile.py); used only when workload, distribution, n, batch and rerankK all match this run
        try:
            tj = json.load(open(tpath)).get("entries", {}).get(args.workload)
            if world == 1 and tj and tj.get("n") == n and tj.get("batch") == B and tj.get("rerankK") == rk \
                    and tj.get("dist", "aligned") == dist_name:
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_source = f"{tj.get('source', 'profiles/traffic_latest.json')} (PMC passes kept from {tj.get('date', 'an earlier run')}; not re-measured by this run)"
        except Exception:
            traffic = None

    index, queries = eng.index, eng.queries
    # ---- p50 latency: one query in flight through the host-pointer API (first 20 calls are warm-up) ----
    p50 = None
    if not args.profile_mode:
        qh = queries[:220].cpu().numpy()
        lat = []
        for i in range(len(qh)):
            t1 = time.perf_counter()
            index.search(qh[i], k, rk)
            lat.append((time.perf_counter() - t1) * 1e3)
        p50 = float(np.median(lat[20:])) if len(lat) > 20 else None
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
    # the same pattern WITH a doc filter (a filtered k-NN query: J/JVectorReader.java:157-163): one bitset of selectivity 0.5
    # over the doc ids on every call, every answer compared with the batch API's
    filtered_rows = None
    if not args.profile_mode and world == 1 and fused and isinstance(caller_rows, list):
        try:
            acc = binding.accept_words(np.nonzero(np.random.default_rng(5).random(eng.n) < 0.5)[0], eng.n)
            hqf = queries[:1024].cpu().numpy()
            want = index.search_batch(hqf, k, rk, accept=acc, accept_num_docs=eng.n).nodes
            filtered_rows = []
            for T in (1, 256):
                r = hostmod.concurrent_search_bench(index, hqf, k, rk, T, 1.5, want, accept=acc, accept_num_docs=eng.n)
                filtered_rows.append({"selectivity": 0.5, "caller_threads": T, "qps": round(r["qps"], 1), "p50_ms": round(r["p50_ms"], 4),
                                      "p99_ms": round(r["p99_ms"], 4), "answers_differing_from_batch_api": r["mismatches"]})
        except Exception as e:  # pragma: no cover
            filtered_rows = f"unavailable: {e!r}"

    in_flight = None
    if not args.profile_mode and world == 1:
        try:
            # (large batches: two in flight show what the call's tail costs — the counting kernels and the launch that redoes the few
            #  flagged rows of batch i run while batch i + 1 searches; `value` stays one batch at a time)
            small = eng.B <= 1024
            in_flight = [eng.timed_in_flight(rk, max(args.steps, 10) if small else 4, S)
                         for S in (int(x) for x in os.environ.get("JV_BENCH_IN_FLIGHT", "1,2,3,4,8" if small else "1,2").split(","))]
        except (Exception, SystemExit) as e:  # pragma: no cover - a side report never costs the line its headline number
            in_flight = f"unavailable: {e!r}"
    # BASELINE.json configs[4] (C5: batch = 256 concurrent queries on the C3 index) inside the default line: one batch of 256 at a
    # time, device-resident, the same timing protocol as `value` (a step = its slowest query: a latency figure in throughput clothes)
    c5_row = None
    if not args.profile_mode and world == 1 and args.workload == "c3" and B > 256:
        try:
            t5 = eng.timed(rk, 20, 3, barrier, batch=256)
            c5_row = {"workload": "c5: the same index, 256 queries per step, one step at a time", "value": round(t5["qps"], 1), "unit": "queries/s",
                      "ms_per_step": round(t5["elapsed"] / 20 * 1e3, 4), "queries_per_step": 256, "rerankK": rk, "steps": 20, "warmup": 3}
        except (Exception, SystemExit) as e:  # pragma: no cover
            c5_row = f"unavailable: {e!r}"
    exact_batch = None
    if not args.profile_mode and world == 1 and args.workload in ("c3", "c5") and os.environ.get("JV_BENCH_EXACT_BATCH", "1") == "1":
        try:
            exact_batch = exact_batch_report(torch, binding, eng, k, rk)
        except (Exception, SystemExit) as e:  # pragma: no cover - a side report never costs the line its headline number
            exact_batch = f"unavailable: {e!r}"
    # the kernel that carries the step (csrc/jv_abi.cpp enqueue_batch): the persistent jv_search_pqp_kernel for pools beyond
    # 256 entries and wherever its register-table variant applies (PQ-32, not cosine, more than 4 x CUs queries per launch),
    # else round 1's jv_search_pqf_kernel; exact indexes run on jv_search_lds_kernel
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    reg_table = pq_M == 32 and sim != 2 and eng.B > 4 * cus
    main_kernel = ("jv_search_pqp_kernel" if (rk + 64 + R > 256 or reg_table) else "jv_search_pqf_kernel") if fused else "jv_search_lds_kernel"
    if fused and eng.index.counter("launches_pqw") > 0:
        main_kernel = "jv_search_pqw_kernel"  # several waves per query (PQ-32 / PQ-64, L2 or dot product, no filter)
    recall_txt = "nan" if chosen_recall != chosen_recall else f"{chosen_recall:.4f}"
    metric = "queries/sec at recall@10>=0.95" if target_met in (True, None) else \
        f"queries/sec at recall@10={recall_txt} (target 0.95 NOT reached by any rerankK of the sweep)"
    if replicas > 1:
        shard_txt = (f"replicas: each of the {world} GPUs holds the whole corpus and answers its own batches of {B} queries per step; "
                     "no collective on the data path (the process group only carries the barrier and the max-over-ranks clock)")
    elif world > 1:
        shard_txt = ("doc-id range, RCCL all-gather of per-shard top-k + GPU merge" if backend == "nccl" else
                     f"doc-id range, {backend} all-gather (host-staged debug backend, NOT RCCL) + GPU merge")
    else:
        shard_txt = "single GPU"
    result = {
        "metric": metric,
        "value": round(qps, 1),
        "unit": "queries/s",
        "n_gpus": world,
        "rccl_ranks": (world if (world > 1 and backend == "nccl" and replicas == 1) else 0),
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"{args.workload}: {wl['desc']}" + ("" if n_total == wl["n"] * (world if scaling == "weak" else 1) else f" [n_total={n_total}]") +
                        ("" if pq_M == data_M else f" [codec overridden: PQ-{pq_M} instead of the workload's PQ-{data_M}]"),
            "distribution": dist_name,
            "docs_per_gpu": n, "total_docs": n_total, "dim": d, "similarity": ["l2", "dot", "cosine"][sim],
            "R": R, "ef_construction": L, "k": k, "rerankK": rk, "pq_M": pq_M, "queries_per_step": B * replicas,
            "sharding": shard_txt,
            "graph_builder": args.builder, "pq_layout": ("fused" if fused else ("plain" if pq_M else None)),
        },
        "recall_at_10": (None if chosen_recall != chosen_recall else round(chosen_recall, 4)),
        "recall_target_met": target_met,
        "recall_sweep": sweep_log,
        "p50_latency_ms": (None if p50 is None else round(p50, 4)),
        "host_api_qps_pcie_inclusive": pcie_qps,
        "single_query_api": caller_rows,
        "batches_in_flight": in_flight,
        "single_query_api_filtered": filtered_rows,
        "c5_batch_256": c5_row,
        "exact_batch_shared_filter": exact_batch,
        "per_query": {"visited": round(t["visited"] / total_queries, 1), "expanded": round(t["expanded"] / total_queries, 1),
                      "reranked": round(t["reranked"] / total_queries, 1),
                      "algorithmic_bytes": round(bytes_total / total_queries, 1)},
        "big_path_queries_last_step": t["big_path_last_step"],
        "build_seconds": round(eng.build_s, 1),
        "roofline": roofline_object(bytes_per_launch, kernel_avg_ms, t["call_avg_ms"], traffic, traffic_source, main_kernel, pq_M, fused),
    }
    if not pq_M:
        # 65 536 queries over 4 096 shared cluster centres re-read the same rows ~160x per launch: those re-reads are
        # served by L2 / Infinity Cache, so this is a fraction of the SPEC peak with cache hits included, not an HBM rate
        result["roofline"]["note"] = "algorithmic bytes / time vs the 8 TB/s spec peak; hot rows are re-read from L2/MALL, so cache hits are included"

    # ---- CPU baseline: the oracle (a port/restatement, NOT real jVector) on this box's host cores ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.profile_mode:
        try:
            result["cpu_baseline"] = cpu_baseline(torch, binding, eng, rk, args.cpu_seconds, exact_report=exact_batch)
        except MemoryError as e:  # pragma: no cover
            result["cpu_baseline"] = {"value": None, "unit": "queries/s", "cores": 0, "kind": "port", "sample": f"skipped: {e}"}

    # ---- the two other distributions (same workload, N = 1): rerankK / recall / QPS, no host-API or CPU legs ----
    if pq_M and world == 1 and args.workload == "c3" and not args.profile_mode and not args.no_dist_comparison:
        rows = [{"distribution": dist_name, "rerankK": rk, "recall_at_10": result["recall_at_10"],
                 "recall_target_met": target_met, "qps": round(qps, 1), "roofline_frac": result["roofline"]["frac"]}]
        eng.close()
        del eng, index, queries
        torch.cuda.empty_cache()
        for other in [x for x in DISTS if x != dist_name]:
            try:
                e2 = make_engine(other)
                gt2 = e2.ground_truth()
                # bounded: the sweep stops at rerankK = 900 and the rate is taken on 2 steps of <= 65 536 queries (a
                # distribution 32-byte codes cannot rank would otherwise spend minutes in its slowest configuration)
                rk2, rec2, slog2 = e2.sweep(gt2, [r for r in SWEEP if r <= 900])
                log(f"[{other}] recall sweep: {slog2}")
                t2 = e2.timed(rk2, 2, 1, barrier, batch=65536)
                by2 = algorithmic_bytes(t2["visited"], t2["reranked"], t2["expanded"], t2["total_queries"], 2, pq_M, d, R, fused)
                rows.append({"distribution": other, "rerankK": rk2, "recall_at_10": round(rec2, 4),
                             "recall_target_met": bool(rec2 >= 0.95), "qps": round(t2["qps"], 1),
                             "roofline_frac": round(by2 / 2 / (t2["call_avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "queries_per_step": min(65536, B), "recall_sweep_tail": slog2[-3:]})
                e2.close()
                del e2, gt2
                torch.cuda.empty_cache()
            except SystemExit as ex:  # a distribution the engine cannot serve validly is reported, not hidden
                rows.append({"distribution": other, "error": str(ex)})
        result["dist_comparison"] = rows
        eng = None
    if isinstance(result.get("exact_batch_shared_filter"), dict):
        result["exact_batch_shared_filter"].pop("_checks", None)   # (numpy arrays kept for the oracle check; not part of the line)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if eng is not None:
        eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def usable_cpus():
    """(threads to use, CFS quota in CPUs or None).  The GPU boxes run this in a container whose cgroup grants a CPU-time
    quota well below the hardware thread count (measured: cpu.max = 16 CPUs on a 256-thread host); more runnable threads
    than the quota get throttled and the oracle's throughput DROPS (tools/cpu_scaling.py: 51 k QPS at 16 threads, 22 k at
    256 on the same index), so the baseline uses as many threads as the quota allows."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, -(-int(q) // int(p)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = max(1, -(-q // p))
        except (OSError, ValueError):
            pass
    return (min(n, quota) if quota else n), quota


def cpu_baseline(torch, binding, eng, rk, budget_s, exact_report=None):
    import psutil
    pyoracle = graft.load_oracle()
    base, adj_t, entry, sim, pq, queries, k = eng.base, eng.adj_t, eng.entry, eng.sim, eng.pq, eng.queries, eng.k
    need = base.numel() * 4 + adj_t.numel() * 4
    if psutil.virtual_memory().available < need * 1.3:
        raise MemoryError(f"host RAM too small for a {need / 1e9:.1f} GB index copy")
    t0 = time.time()
    # host copy first-touched by all threads (pages on every NUMA node; see oracle/jv_oracle.h: jvo_parallel_copy)
    spread = lambda t: pyoracle.spread_to_host(binding.JvIndexDesc, t.contiguous())
    ix = binding.IndexData(vectors=spread(base), adj=spread(adj_t), entry_node=entry, similarity=sim)
    if pq:
        ix.pq_codebooks, ix.pq_centroid, ix.pq_codes = pq["codebooks"], pq["centroid"], spread(pq["codes"])
        ix.pq_M, ix.pq_K = ix.pq_codes.shape[1], pq["K"]
    orc = pyoracle.Oracle(binding, ix)
    log(f"cpu_baseline: index copied to host in {time.time() - t0:.1f}s")
    host_threads = os.cpu_count() or 1
    cores, quota = usable_cpus()
    pool = queries.cpu().numpy()
    orc.search_batch(pool[:min(4 * cores, len(pool))], k, rk, threads=cores)  # warm the thread pool + per-thread scratch
    nsample = min(len(pool), 4096)
    while True:
        sample = pool[:nsample]
        t1 = time.perf_counter()
        r = orc.search_batch(sample, k, rk, threads=cores)
        dt = time.perf_counter() - t1
        if dt >= 0.6 * budget_s or nsample >= len(pool):
            break
        nsample = int(min(len(pool), max(nsample * 2, nsample * budget_s / max(dt, 1e-3))))
    # parity spot-check of the measured GPU path against the oracle on the same queries: ids, score bits, counters
    m = min(nsample, eng.out_nodes.shape[0])
    eng.run_step(queries[:m], rk, nq=m)
    eng.stream.synchronize()
    same_ids = bool(np.array_equal(eng.out_nodes[:m].cpu().numpy(), r.nodes[:m]))
    same_bits = bool(np.array_equal(eng.out_scores[:m].cpu().numpy().view(np.uint32), r.scores[:m].view(np.uint32)))
    same_stats = bool(np.array_equal(eng.out_stats[:m].cpu().numpy(), r.stats[:m]))
    if isinstance(exact_report, dict) and exact_report.get("_checks"):
        exact_batch_oracle_check(exact_report, orc, pool, k, cores)
    # the same sample with the explicit AVX2 paths (look-up-table gathers, software prefetch of code rows and rerank rows):
    # same operation order, so the answers must be identical; the better of the two is `value`
    simd = None
    try:
        orc.lib.jvo_set_simd(1)
        orc.search_batch(sample[:min(4 * cores, nsample)], k, rk, threads=cores)
        t3 = time.perf_counter()
        r2 = orc.search_batch(sample, k, rk, threads=cores)
        dt_simd = time.perf_counter() - t3
        simd = {"qps": round(nsample / dt_simd, 1), "ids_equal_plain_loops": bool(np.array_equal(r2.nodes, r.nodes)),
                "score_bits_equal_plain_loops": bool(np.array_equal(r2.scores.view(np.uint32), r.scores.view(np.uint32)))}
    finally:
        orc.lib.jvo_set_simd(0)
    plain_qps = nsample / dt
    if simd and simd["qps"] > plain_qps and simd["ids_equal_plain_loops"]:
        dt = dt_simd
    # single-thread figure (the reference's JMH style is one thread)
    n1 = max(8, min(256, nsample))
    orc.search_batch(sample[:8], k, rk, threads=1)
    t2 = time.perf_counter()
    orc.search_batch(sample[:n1], k, rk, threads=1)
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
            "sample": f"{nsample} queries of the same workload, same index/rerankK, OpenMP one query per thread with "
                      f"per-thread reusable searcher scratch ({dt:.1f}s); C restatement of jVector's search (real jVector "
                      f"needs a JVM: not available)",
            "plain_loops_qps": round(plain_qps, 1), "explicit_avx2": simd,
            "single_thread_qps": round(n1 / dt1, 1), "cpu_model": cpu_model,
            "host_threads": host_threads, "cpu_quota_cpus": quota, "gpu_ids_equal_oracle_on_sample": same_ids,
            "gpu_score_bits_equal_oracle_on_sample": same_bits, "gpu_counters_equal_oracle_on_sample": same_stats}


if __name__ == "__main__":
    main()

"""GPU index construction (WRITE side — not the graded hot path).

The reference builds its Vamana graph and PQ with the jvector library at flush/merge
(J/JVectorWriter.java:1383-1422 ``getGraph``: parallel ``addGraphNode`` + ``cleanup``;
J/JVectorIndexQuantization.java:114-140 ``computePqVectors``).  To have 1M-10M-node inputs for the
search benchmark within minutes, this module builds them on the GPU:

* **batched Vamana**: insert nodes in batches; every batch (a) runs the PRODUCT search kernel
  (``jv_search_batch_device``, beam = ef_construction) over the graph built so far — the builder is a
  client of the hot path, exactly as jvector's builder is a client of ``GraphSearcher`` —, (b) prunes
  each node's candidates with jvector-style diversity selection (alpha sweep 1.0 -> alpha), (c) adds
  back-links with ``neighborOverflow`` slack and re-prunes the rows that overflow, and finally (d)
  ``cleanup`` prunes every row to R.  The diversity selection of (b)-(d) — scores to the centre, ordering, the
  candidate x candidate products and the alpha sweep — is ONE hand-written kernel per call
  (csrc/jv_build_kernels.hip jvb_prune_rows_kernel); what is left in torch is buffer bookkeeping.
* **PQ**: Lloyd k-means per subspace (256 clusters, global centring iff EUCLIDEAN) + encoding.

Defaults R=32, ef_construction=100, alpha=1.2, overflow=1.2 are the reference's
(J/JVectorFormat.java:34-35, K/common/KNNConstants.java:106-107).
"""
from __future__ import annotations

import math
import time

import numpy as np

import ctypes as C
import os
import sys

from . import binding

_HERE = os.path.dirname(os.path.abspath(__file__))
_PRUNE_LIB = None


def _prune_lib():
    """lib/libjvbuildgpu.so (csrc/jv_build_kernels.hip): the diversity-selection kernel."""
    global _PRUNE_LIB
    if _PRUNE_LIB is None:
        path = os.path.join(_HERE, "lib", "libjvbuildgpu.so")
        lib = C.CDLL(path)
        vp = C.c_void_p
        lib.jvb_prune_rows_device.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp, C.c_int,
                                              vp, vp]
        lib.jvb_prune_rows_device.restype = C.c_int
        lib.jvb_backlinks_device.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp]
        lib.jvb_backlinks_device.restype = C.c_int
        lib.jvb_pq_encode_device.argtypes = [vp, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_int,
                                             C.c_int, vp]
        lib.jvb_pq_encode_device.restype = C.c_int
        lib.jvb_pq_train_device.argtypes = [vp, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_longlong, vp, C.c_int,
                                            vp, vp, vp, vp, vp, C.c_int, vp]
        lib.jvb_pq_train_device.restype = C.c_int
        lib.jvb_pq_train_device2.argtypes = [vp, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_longlong, vp, C.c_int,
                                             vp, vp, vp, vp, vp, C.c_int, C.c_ulonglong, vp, vp, vp]
        lib.jvb_pq_train_device2.restype = C.c_int
        _PRUNE_LIB = lib
    return _PRUNE_LIB


def _scores_from_gram(torch, G, sq_a, sq_b, sim):
    """jVector similarity scores from dot products (App. A.4 mappings)."""
    if sim == 0:
        d2 = (sq_a + sq_b - 2.0 * G).clamp_min_(0.0)
        return 1.0 / (1.0 + d2)
    if sim == 1:
        return (1.0 + G) * 0.5
    return (1.0 + G / (sq_a * sq_b).clamp_min(1e-30).sqrt()) * 0.5


PRUNE_MAX_CANDIDATES = 160  # csrc/jv_build_kernels.hip JVB_PRUNE_MAX_LC (the candidate x candidate matrix lives in LDS)


_TRUNC_WARNED = [False]


def _truncate_candidates(torch, base, centers, cand, keep, sim):
    """Rows with more candidates than the selection kernel takes (ef_construction > 160, or ef_construction + row in a
    refine pass / merge): keep the `keep` best by score to the centre (ties: lower id first; empty slots last).  jvector's
    RobustPrune walks ALL candidates in that order and stops at R selected ones, so the tail beyond the best 160 only matters
    when fewer than R of the first 160 survive the diversity test — logged once, measured by the L = 200 test."""
    S, Lc = cand.shape
    out = torch.empty((S, keep), dtype=torch.int32, device=cand.device)
    d = int(base.shape[1])
    rows = max(1, (64 << 20) // (Lc * d * 4))
    for s0 in range(0, S, rows):
        c = cand[s0:s0 + rows].long()
        ok = c >= 0
        v = base[c.clamp_min(0)]                                   # [rows][Lc][d]
        q = base[centers[s0:s0 + rows].long()].unsqueeze(1)          # [rows][1][d]
        dots = (v * q).sum(2)
        sc = _scores_from_gram(torch, dots, (v * v).sum(2), (q * q).sum(2), sim)
        sc = torch.where(ok, sc, torch.full_like(sc, -float("inf")))
        # (score desc, id asc): stable sort by id first, then stable by score
        o1 = torch.argsort(torch.where(ok, c, torch.full_like(c, 1 << 40)), dim=1, stable=True)
        sc1 = torch.gather(sc, 1, o1)
        o2 = torch.argsort(sc1, dim=1, descending=True, stable=True)
        pick = torch.gather(o1, 1, o2)[:, :keep]
        out[s0:s0 + rows] = torch.gather(cand[s0:s0 + rows], 1, pick).to(torch.int32)
    return out


def robust_prune(torch, base, centers, cand, R, alpha, sim):
    """Diversity selection for every row: centers [S] (node ids), cand [S][Lc] (node ids, -1 = empty).
    Returns sel [S][R] (-1 padded), nsel [S].  jvector semantics: candidates in descending score to
    the centre (ties: lower id first, duplicates dropped); for a in (1.0, 1.2, .. alpha): keep c unless some
    already-selected s has sim(c, s) > sim(c, centre) * a.
    All of it runs in ONE hand-written kernel per call (csrc/jv_build_kernels.hip jvb_prune_rows_kernel: a workgroup per row
    scores, orders, multiplies and selects straight from the vectors in HBM); this function only allocates the outputs.
    More than PRUNE_MAX_CANDIDATES candidates per row (ef_construction is a user-facing mapping parameter of the reference,
    K/common/KNNConstants.java METHOD_PARAMETER_EF_CONSTRUCTION): the best 160 by score are handed to the kernel."""
    S, Lc = cand.shape
    dev = base.device
    if dev.type != "cuda":
        raise RuntimeError("builder_gpu.robust_prune needs the HIP builder library (no CPU fallback)")
    if R > PRUNE_MAX_CANDIDATES:
        raise ValueError(f"R = {R}: the selection kernel keeps at most {PRUNE_MAX_CANDIDATES} candidates per row")
    if Lc > PRUNE_MAX_CANDIDATES:
        if not _TRUNC_WARNED[0]:
            _TRUNC_WARNED[0] = True
            print(f"[builder_gpu] {Lc} candidates per row: the selection kernel takes {PRUNE_MAX_CANDIDATES}; "
                  "keeping the best by score to the centre", file=sys.stderr, flush=True)
        cand = _truncate_candidates(torch, base, centers, cand, PRUNE_MAX_CANDIDATES, sim)
        Lc = PRUNE_MAX_CANDIDATES
    sel = torch.empty((S, R), dtype=torch.int32, device=dev)
    nsel = torch.empty((S,), dtype=torch.int32, device=dev)
    if S == 0:
        return sel, nsel
    c64 = centers.to(torch.int64).contiguous()
    cd = cand.to(torch.int32).contiguous()
    rc = _prune_lib().jvb_prune_rows_device(base.data_ptr(), int(base.shape[1]), int(base.stride(0)), int(sim), c64.data_ptr(), cd.data_ptr(),
                                            int(cd.stride(0)), int(S), int(Lc), int(R), float(alpha), sel.data_ptr(), int(sel.stride(0)),
                                            nsel.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    if rc != 0:
        raise RuntimeError(f"jvb_prune_rows_device failed: {rc}")
    return sel, nsel


def approx_medoid(torch, base, sim):
    n, d = base.shape
    acc = torch.zeros((d,), dtype=torch.float64, device=base.device)
    ch = 1 << 18
    for s in range(0, n, ch):
        acc += base[s:s + ch].sum(0, dtype=torch.float64)
    mean = (acc / n).to(torch.float32)
    best, best_s = 0, -float("inf")
    for s in range(0, n, ch):
        b = base[s:s + ch]
        dots = b @ mean
        sc = _scores_from_gram(torch, dots, (b * b).sum(1), (mean * mean).sum(), sim)
        v, i = sc.max(0)
        if float(v) > best_s:
            best_s, best = float(v), int(i) + s
    return best


def build_graph_gpu(torch, base, sim, R=32, L=100, alpha=1.2, overflow=1.2, device_index=0, max_batch=16384,
                    search_fn=None, verbose=True, refine_passes=0, initial_adj=None, initial_entry=None):
    """Returns (adj [n][R] int32 tensor on base.device, entry_node).

    refine_passes > 0: after the batched insertion every node is searched for AGAIN on the finished graph and its row is
    re-selected from (search result + current row), back-links included — what jvector's sequential addGraphNode gets for
    free (every insert sees all earlier ones; a batch's members do not see each other) plus its cleanup()
    (J/JVectorWriter.java:1383-1422).  One pass restores the reference's KA15 recall floor (tests/test_gpu_builder.py).

    initial_adj [n0][R] (+ initial_entry): rows 0..n0-1 of `base` already have a graph (a leading segment's,
    merge_leading_segment_gpu); only rows n0.. are inserted, into it."""
    n, d = base.shape
    dev = base.device
    assert d % 4 == 0, "GPU builder needs 16-B aligned rows (d % 4 == 0)"
    if not (1 <= R <= 128):
        raise ValueError(f"R = {R}: the GPU builder takes 1 <= R <= 128 (row + overflow slack must fit the selection kernel)")
    if L < 1:
        raise ValueError(f"ef_construction = {L} < 1")
    bl_state = {}   # back-link scratch of THIS build (head / next / overflow rows): never shared between builds or threads
    Rcap = int(math.ceil(R * overflow))
    Rcap = (Rcap + 3) & ~3
    adj = torch.full((n, Rcap), -1, dtype=torch.int32, device=dev)
    deg = torch.zeros((n,), dtype=torch.int32, device=dev)
    if n == 0:
        return adj[:, :R].contiguous(), -1
    order = torch.arange(n, device=dev, dtype=torch.int64)
    n0 = 1
    if initial_adj is not None:
        n0 = int(initial_adj.shape[0])
        assert 0 < n0 <= n and initial_adj.shape[1] == R and initial_entry is not None and 0 <= initial_entry < n0
        adj[:n0, :R] = initial_adj.to(device=dev, dtype=torch.int32)
        deg[:n0] = (adj[:n0] >= 0).sum(1).to(torch.int32)
        entry = int(initial_entry)
    else:
        entry = approx_medoid(torch, base, sim)
        if entry != 0:
            order[0], order[entry] = entry, 0
    index = None
    if search_fn is None:
        desc, keep = binding.make_desc_device(n, d, Rcap, base.data_ptr(), adj.data_ptr(), entry, sim,
                                              device=device_index, borrow=True,
                                              extra_flags=binding.DESC_BUILD_CLIENT)
        index = binding.GpuIndex(desc=desc, keepalive=keep, flags=binding.DESC_BORROW)
        stream = torch.cuda.current_stream(dev)
        o_nodes = torch.empty((max_batch, L), dtype=torch.int32, device=dev)
        o_scores = torch.empty((max_batch, L), dtype=torch.float32, device=dev)
        o_count = torch.empty((max_batch,), dtype=torch.int32, device=dev)
        o_stats = torch.empty((max_batch, 4), dtype=torch.int32, device=dev)
        o_flags = torch.empty((max_batch,), dtype=torch.int32, device=dev)

        def search_fn(q, B):
            # torch's default stream has handle 0, and jv_search_batch_device reads a NULL stream as "the library's own stream,
            # synchronous": NOT ordered behind the gather that is still writing q on torch's stream.  (Round 4: at d = 1 536 the
            # 84 MB gather of a 13 655-row batch lost that race now and then — searches with half-written queries, builds that
            # differed from run to run, C4's recall at rerankK 1 200 anywhere between 0.940 and 0.952.)
            if stream.cuda_stream == 0:
                stream.synchronize()
            index.search_batch_device(q.data_ptr(), B, L, L, o_nodes.data_ptr(), 0, o_scores.data_ptr(),
                                      o_count.data_ptr(), o_stats.data_ptr(), o_flags.data_ptr(),
                                      stream=stream.cuda_stream)
            return o_nodes[:B]

    t0 = time.time()
    pos = n0
    it = 0
    while pos < n:
        B = min(max_batch, max(1, pos // 2), n - pos)
        u = order[pos:pos + B]
        q = base[u].contiguous()
        cand = search_fn(q, B)                                   # [B][L] best-first candidates
        sel, nsel = robust_prune(torch, base, u, cand, R, alpha, sim)
        adj[u, :R] = sel
        adj[u, R:] = -1
        deg[u] = nsel
        _add_backlinks(torch, base, adj, deg, u, sel, R, Rcap, alpha, sim, bl_state)
        pos += B
        it += 1
        if verbose and (it % 50 == 0 or pos >= n):
            torch.cuda.synchronize() if dev.type == "cuda" else None
            print(f"[builder_gpu] inserted {pos}/{n} ({time.time() - t0:.1f}s)", file=sys.stderr, flush=True)
    for rp in range(refine_passes):
        for s in range(0 if initial_adj is None else n0, n, max_batch):  # (a merge refines what it inserted)
            u = torch.arange(s, min(n, s + max_batch), device=dev, dtype=torch.int64)
            B = int(u.numel())
            cand = search_fn(base[u].contiguous(), B)
            both = torch.cat([cand, adj[u]], dim=1)
            sel, nsel = robust_prune(torch, base, u, both, R, alpha, sim)
            adj[u, :R] = sel
            adj[u, R:] = -1
            deg[u] = nsel
            _add_backlinks(torch, base, adj, deg, u, sel, R, Rcap, alpha, sim, bl_state)
        if verbose:
            torch.cuda.synchronize() if dev.type == "cuda" else None
            print(f"[builder_gpu] refine pass {rp + 1}/{refine_passes} done ({time.time() - t0:.1f}s)", file=sys.stderr, flush=True)
    # cleanup: every row down to R
    over = torch.nonzero(deg > R).squeeze(1)
    ch = 8192
    for s in range(0, over.numel(), ch):
        o = over[s:s + ch]
        sel, nsel = robust_prune(torch, base, o, adj[o], R, alpha, sim)
        adj[o, :R] = sel
        adj[o, R:] = -1
        deg[o] = nsel
    out = adj[:, :R].contiguous()
    if index is not None:
        torch.cuda.synchronize()
        index.close()
    return out, entry


def merge_leading_segment_gpu(torch, base, lead_adj, lead_entry, lead_live, sim, R=32, L=100, alpha=1.2, refine_passes=1,
                              max_repair_rows=4, **kw):
    """Incremental merge into the LEADING segment's graph (J/JVectorWriter.java:1166-1341, tryLeadingSegmentMerge): instead
    of rebuilding the merged field from scratch, the largest segment's graph is loaded, the other segments' live vectors are
    inserted into it (addGraphNode), the leading segment's deleted nodes are removed (markNodeDeleted + cleanup: a live
    node that pointed at a deleted one re-selects its row from its live neighbours and the deleted neighbours' live
    neighbours), and the ordinals are compacted in order.

    base [n][d]: rows 0..n0-1 = the leading segment's vectors in its ordinal order (deleted ones included: they stay
    traversable until the cleanup, as in jvector), rows n0.. = the other segments' live vectors ("mid" ordinals).
    lead_adj [n0][R], lead_entry, lead_live [n0] bool.
    Returns (adj [n_live][R] int32, entry, final_to_mid [n_live] int64): compact, order-preserving "final" ordinals."""
    dev = base.device
    n = int(base.shape[0])
    n0 = int(lead_adj.shape[0])
    live = torch.ones((n,), dtype=torch.bool, device=dev)
    live[:n0] = lead_live.to(device=dev, dtype=torch.bool)
    adj, entry = build_graph_gpu(torch, base, sim, R=R, L=L, alpha=alpha, refine_passes=refine_passes,
                                 initial_adj=lead_adj, initial_entry=lead_entry, **kw)
    adj = adj.clone()
    dead = ~live
    if bool(dead.any()):
        nb = adj.long().clamp_min(0)
        nb_dead = (adj >= 0) & dead[nb]                                   # [n][R] edges into deleted nodes
        touched = torch.nonzero(nb_dead.any(1) & live).squeeze(1)         # live nodes that lose a neighbour
        for s in range(0, touched.numel(), 8192):
            u = touched[s:s + 8192]
            rows = adj[u]
            rd = nb_dead[u]
            # the first `max_repair_rows` deleted neighbours of every row lend their own rows as candidates
            rank = torch.cumsum(rd.to(torch.int32), 1) - 1
            extra = torch.full((u.numel(), max_repair_rows, R), -1, dtype=torch.int32, device=dev)
            for k in range(max_repair_rows):
                pick = rd & (rank == k)
                has = pick.any(1)
                src = (rows.long().clamp_min(0) * pick).sum(1)            # the k-th deleted neighbour (0 where none)
                extra[:, k] = torch.where(has[:, None], adj[src], torch.full_like(adj[src], -1))
            cand = torch.cat([rows, extra.reshape(u.numel(), -1)], 1)
            cand = torch.where((cand >= 0) & live[cand.long().clamp_min(0)], cand, torch.full_like(cand, -1))
            sel, _ = robust_prune(torch, base, u, cand, R, alpha, sim)
            adj[u] = sel
    final_to_mid = torch.nonzero(live).squeeze(1)
    mid_to_final = torch.full((n,), -1, dtype=torch.int32, device=dev)
    mid_to_final[final_to_mid] = torch.arange(final_to_mid.numel(), dtype=torch.int32, device=dev)
    out = adj[final_to_mid]
    out = torch.where(out >= 0, mid_to_final[out.long().clamp_min(0)], out)
    # (rows keep jvector's "no holes" form: valid neighbours first)
    order = torch.argsort((out < 0).to(torch.int8), dim=1, stable=True)
    out = torch.gather(out, 1, order).contiguous()
    if not bool(live[entry]):
        entry_mid = int(final_to_mid[approx_medoid(torch, base[final_to_mid], sim)])
    else:
        entry_mid = entry
    return out, int(mid_to_final[entry_mid]), final_to_mid


def _add_backlinks(torch, base, adj, deg, u, sel, R, Rcap, alpha, sim, stt):
    """for every new edge u -> s add s -> u; rows that would exceed Rcap are re-pruned to R.
    Grouping, ordering and the append run in csrc/jv_build_kernels.hip (jvb_backlinks_device: per-target lists, sources sorted
    by id — reproducible); `stt` is the calling build's scratch dict (allocated once per build_graph_gpu call and dropped with
    it: two builds — another field's with a different m, or another thread's — never share head / next / overflow buffers)."""
    dev = base.device
    B = int(u.shape[0])
    if B == 0:
        return
    n = int(adj.shape[0])
    if "head" not in stt:
        stt["head"] = torch.full((n,), -1, dtype=torch.int32, device=dev)
        stt["counters"] = torch.zeros((2,), dtype=torch.int32, device=dev)
    E = B * int(sel.shape[1])
    if stt.get("E", 0) < E:
        stt["next"] = torch.empty((E,), dtype=torch.int32, device=dev)
        stt["touched"] = torch.empty((E,), dtype=torch.int32, device=dev)
        stt["E"] = E
    Lc = min(PRUNE_MAX_CANDIDATES, 2 * int(Rcap))   # the row + as many new sources again (the smallest ids when a target has more)
    ov_rows = min(E, n)
    if stt.get("ov_rows", 0) < ov_rows or stt.get("Lc") != Lc:
        ov_rows = max(ov_rows, stt.get("ov_rows", 0))
        stt["Lc"] = Lc
        stt["ov_nodes"] = torch.empty((ov_rows,), dtype=torch.int32, device=dev)
        stt["ov_cand"] = torch.empty((ov_rows, Lc), dtype=torch.int32, device=dev)
        stt["ov_rows"] = ov_rows
    u64 = u.to(torch.int64).contiguous()
    sel32 = sel.contiguous()
    lib = _prune_lib()
    rc = lib.jvb_backlinks_device(u64.data_ptr(), B, sel32.data_ptr(), int(sel32.stride(0)), int(sel32.shape[1]), adj.data_ptr(), int(Rcap),
                                  deg.data_ptr(), stt["head"].data_ptr(), stt["next"].data_ptr(), stt["touched"].data_ptr(),
                                  stt["counters"].data_ptr(), stt["ov_nodes"].data_ptr(), stt["ov_cand"].data_ptr(), int(stt["ov_rows"]), Lc,
                                  torch.cuda.current_stream(dev).cuda_stream)
    if rc != 0:
        raise RuntimeError(f"jvb_backlinks_device failed: {rc}")
    nov = int(stt["counters"][1].item())
    if nov == 0:
        return
    assert nov <= stt["ov_rows"]
    ch = 8192
    for s0 in range(0, nov, ch):
        o = stt["ov_nodes"][s0:min(nov, s0 + ch)].long()
        sel2, nsel2 = robust_prune(torch, base, o, stt["ov_cand"][s0:min(nov, s0 + ch)], R, alpha, sim)
        adj[o, :R] = sel2
        adj[o, R:] = -1
        deg[o] = nsel2


def pq_encode_gpu(torch, base, M, K, codebooks_t, centroid_t):
    """PQ codes of every row of `base` on the GPU (csrc/jv_build_kernels.hip jvb_pq_encode_kernel): nearest centroid per
    subspace under the canonical fmaf-chain distance, ties -> lowest index; bit-identical to jvb_pq_encode_cpu.
    codebooks_t: flat float32 tensor (concat over m of [K][ds_m]); centroid_t: [d] tensor or None.  Returns uint8 [n][M]."""
    n, d = base.shape
    dev = base.device
    sizes = [d // M + (1 if m < d % M else 0) for m in range(M)]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    cb_off = np.concatenate([[0], np.cumsum([K * s_ for s_ in sizes])[:-1]]).astype(np.int64)
    t_off = torch.from_numpy(offs).to(dev)
    t_cb = torch.from_numpy(cb_off).to(dev)
    codes = torch.empty((n, M), dtype=torch.uint8, device=dev)
    cbt = codebooks_t.to(dev, torch.float32).contiguous()
    cen = None if centroid_t is None else centroid_t.to(dev, torch.float32).contiguous()
    rc = _prune_lib().jvb_pq_encode_device(base.data_ptr(), n, d, base.stride(0), M, K, t_off.data_ptr(), cbt.data_ptr(),
                                           t_cb.data_ptr(), (cen.data_ptr() if cen is not None else None), codes.data_ptr(), M,
                                           int(max(sizes)), torch.cuda.current_stream(dev).cuda_stream)
    if rc != 0:
        raise RuntimeError(f"jvb_pq_encode_device failed: {rc}")
    return codes


def pq_train_encode_gpu(torch, base, M, sim, K=256, iters=8, max_train=128000, seed=1, seeding=None):
    """ProductQuantization.compute analogue: K = min(256, n) clusters per subspace, global centring iff
    EUCLIDEAN (J/JVectorIndexQuantization.java:122-131).  Training runs in csrc/jv_build_kernels.hip
    (jvb_pq_train_device: column mean, sample gather, Lloyd's algorithm with the ENCODER as the assign step and a
    reduction-tree update, no atomics -> reproducible bit for bit); this function only allocates the buffers and picks the
    seeded initial centroids.  Returns dict(codebooks (np), centroid (np|None), codes (uint8 tensor [n][M]), K)."""
    n, d = base.shape
    dev = base.device
    K = min(K, n)
    center = sim == 0
    sizes = [d // M + (1 if m < d % M else 0) for m in range(M)]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    cb_off = np.concatenate([[0], np.cumsum([K * s_ for s_ in sizes])[:-1]]).astype(np.int64)
    nt = min(n, max_train)
    rows = (torch.arange(nt, device=dev, dtype=torch.int64) * n) // nt
    # seeding: "kmeans++" (jvector's KMeansPlusPlusClusterer, J/JVectorIndexQuantization.java:122-131; on the device, deterministic,
    # the CPU builder's splitmix64 stream) or "random" (seeded sample rows: rounds 1-4).  Default from JV_PQ_SEEDING, else
    # "random": tools/pq_quality.py measured the two within +/- 0.003 recall@10 of each other on the benchmark's data, and
    # the benchmark's operating points (rerankK) were taken on "random".
    seeding = seeding or os.environ.get("JV_PQ_SEEDING", "random")
    if seeding not in ("random", "kmeans++"):
        raise ValueError(f"seeding = {seeding!r}")
    init = None
    if seeding == "random" or max(d // M + (1 if d % M else 0), 1) > 256:
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        init = torch.stack([torch.randperm(nt, generator=g, device=dev)[:K] for _ in range(M)]).contiguous()   # [M][K] sample positions
    t_off = torch.from_numpy(offs).to(dev)
    t_cb = torch.from_numpy(cb_off).to(dev)
    centroid = torch.empty((d,), dtype=torch.float32, device=dev) if center else None
    sample = torch.empty((nt, d), dtype=torch.float32, device=dev)
    sample_codes = torch.empty((nt, M), dtype=torch.uint8, device=dev)
    partial = torch.empty((64, d), dtype=torch.float64, device=dev)
    codebooks_t = torch.empty((int(K * d),), dtype=torch.float32, device=dev)
    lib = _prune_lib()
    init_scratch = torch.empty((M, K), dtype=torch.int64, device=dev) if init is None else None
    mind_scratch = torch.empty((M, nt), dtype=torch.float32, device=dev) if init is None else None
    rc = lib.jvb_pq_train_device2(base.data_ptr(), n, d, base.stride(0), M, K, t_off.data_ptr(), t_cb.data_ptr(), rows.data_ptr(), nt,
                                  (init.data_ptr() if init is not None else None), iters, (centroid.data_ptr() if center else None),
                                  sample.data_ptr(), sample_codes.data_ptr(), partial.data_ptr(), codebooks_t.data_ptr(), int(max(sizes)),
                                  int(seed), (init_scratch.data_ptr() if init is None else None),
                                  (mind_scratch.data_ptr() if init is None else None), torch.cuda.current_stream(dev).cuda_stream)
    if rc != 0:
        raise RuntimeError(f"jvb_pq_train_device failed: {rc}")
    del sample, sample_codes, partial, init_scratch, mind_scratch
    codes = pq_encode_gpu(torch, base, M, K, codebooks_t, centroid)
    return dict(codebooks=codebooks_t.cpu().numpy().astype(np.float32), centroid=(centroid.cpu().numpy() if center else None), codes=codes, K=K)

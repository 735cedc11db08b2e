"""ctypes view of the write-side helper (lib/libjvbuild.so, csrc/jv_build.h): graph + PQ construction.

NOT the hot path: the reference builds these with the jvector library at flush/merge
(J/JVectorWriter.java:1383-1422, J/JVectorIndexQuantization.java:114-140).  Used to make inputs for
tests, smoke() and bench.py.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .binding import IndexData, SIM_EUCLIDEAN

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libjvbuild.so")
_lib = None


def load_library(path: str = LIB_PATH) -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} is missing: run __graft_entry__.build()")
        lib = C.CDLL(path)
        vp, i32, f32, u64 = C.c_void_p, C.c_int32, C.c_float, C.c_uint64
        lib.jvb_build_graph_cpu.argtypes = [vp, i32, i32, i32, i32, i32, f32, f32, i32, i32, vp, vp]
        lib.jvb_build_graph_cpu.restype = C.c_int
        lib.jvb_build_upper_layers_cpu.argtypes = [vp, i32, i32, i32, i32, i32, f32, i32, u64, vp, vp, vp, vp]
        lib.jvb_build_upper_layers_cpu.restype = C.c_int
        lib.jvb_pq_train_cpu.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, u64, i32, vp, vp]
        lib.jvb_pq_train_cpu.restype = C.c_int
        lib.jvb_pq_encode_cpu.argtypes = [vp, i32, i32, i32, i32, vp, vp, i32, vp]
        lib.jvb_pq_encode_cpu.restype = C.c_int
        _lib = lib
    return _lib


def build_graph_cpu(vectors: np.ndarray, similarity: int = SIM_EUCLIDEAN, R: int = 32, L: int = 100,
                    alpha: float = 1.2, overflow: float = 1.2, max_batch: int = 256, threads: int = 0):
    """Batched Vamana (defaults = the reference's: J/JVectorFormat.java:34-35, KNNConstants.java:106-107)."""
    lib = load_library()
    v = np.ascontiguousarray(vectors, dtype=np.float32)
    n, d = v.shape
    adj = np.full((n, R), -1, dtype=np.int32)
    entry = C.c_int32(-1)
    rc = lib.jvb_build_graph_cpu(v.ctypes.data, n, d, similarity, R, L, alpha, overflow, max_batch, threads,
                                 adj.ctypes.data, C.addressof(entry))
    if rc != 0:
        raise RuntimeError(f"jvb_build_graph_cpu failed: {rc}")
    return adj, int(entry.value)


def build_upper_layers_cpu(vectors: np.ndarray, similarity: int, R: int, num_layers: int, L: int = 100,
                           alpha: float = 1.2, seed: int = 7):
    lib = load_library()
    v = np.ascontiguousarray(vectors, dtype=np.float32)
    n, d = v.shape
    counts = np.zeros(num_layers, dtype=np.int32)
    rc = lib.jvb_build_upper_layers_cpu(v.ctypes.data, n, d, similarity, R, L, alpha, num_layers, seed,
                                        counts.ctypes.data, None, None, None)
    if rc != 0:
        raise RuntimeError("jvb_build_upper_layers_cpu failed")
    nodes = [np.zeros(int(c), dtype=np.int32) for c in counts]
    adjs = [np.full((int(c), R), -1, dtype=np.int32) for c in counts]
    np_ptrs = (C.c_void_p * num_layers)(*[a.ctypes.data for a in nodes])
    ad_ptrs = (C.c_void_p * num_layers)(*[a.ctypes.data for a in adjs])
    entry = C.c_int32(-1)
    rc = lib.jvb_build_upper_layers_cpu(v.ctypes.data, n, d, similarity, R, L, alpha, num_layers, seed,
                                        counts.ctypes.data, np_ptrs, ad_ptrs, C.addressof(entry))
    if rc != 0:
        raise RuntimeError("jvb_build_upper_layers_cpu failed")
    return nodes, adjs, int(entry.value)


def pq_train_encode_cpu(vectors: np.ndarray, M: int, similarity: int, K: int = 256, iters: int = 6,
                        max_train: int = 128000, seed: int = 1, threads: int = 0):
    """256 clusters (min(256, n)), global centring iff EUCLIDEAN (J/JVectorIndexQuantization.java:122-131)."""
    lib = load_library()
    v = np.ascontiguousarray(vectors, dtype=np.float32)
    n, d = v.shape
    K = min(K, n)
    center = 1 if similarity == SIM_EUCLIDEAN else 0
    codebooks = np.zeros(K * d, dtype=np.float32)
    centroid = np.zeros(d, dtype=np.float32)
    rc = lib.jvb_pq_train_cpu(v.ctypes.data, n, d, M, K, center, iters, max_train, seed, threads,
                              codebooks.ctypes.data, centroid.ctypes.data)
    if rc != 0:
        raise RuntimeError("jvb_pq_train_cpu failed")
    codes = np.zeros((n, M), dtype=np.uint8)
    rc = lib.jvb_pq_encode_cpu(v.ctypes.data, n, d, M, K, codebooks.ctypes.data,
                               centroid.ctypes.data if center else None, threads, codes.ctypes.data)
    if rc != 0:
        raise RuntimeError("jvb_pq_encode_cpu failed")
    return codebooks, (centroid if center else None), codes, K


def build_index_cpu(vectors: np.ndarray, similarity: int = SIM_EUCLIDEAN, R: int = 32, L: int = 100,
                    alpha: float = 1.2, pq_M: int = 0, hierarchy_layers: int = 0, score_scale: float = 1.0,
                    ord2doc: np.ndarray | None = None, max_doc: int = 0, threads: int = 0) -> IndexData:
    adj, entry = build_graph_cpu(vectors, similarity, R, L, alpha, threads=threads)
    ix = IndexData(vectors=np.ascontiguousarray(vectors, dtype=np.float32), adj=adj, entry_node=entry,
                   similarity=similarity, score_scale=score_scale, ord2doc=ord2doc, max_doc=max_doc)
    if hierarchy_layers > 0 and ix.n > 0:
        nodes, adjs, top_entry = build_upper_layers_cpu(vectors, similarity, R, hierarchy_layers, L, alpha)
        ix.upper_nodes, ix.upper_adj = nodes, adjs
        ix.entry_node = top_entry
    if pq_M > 0 and ix.n > 0:
        cb, cen, codes, K = pq_train_encode_cpu(vectors, pq_M, similarity, threads=threads)
        ix.pq_codebooks, ix.pq_centroid, ix.pq_codes, ix.pq_M, ix.pq_K = cb, cen, codes, pq_M, K
    return ix


def nvq_encode(base: np.ndarray, M: int = 2, growth: float = 1.0, midpoint: float = 0.0):
    """8-bit NVQ records for test / benchmark inputs (write side; NOT jvector's trained encoder: the parameter fit of
    NVQuantization lives in the library).  Per vector: subtract the global mean, then per subvector keep (growthRate,
    midpoint, minValue, maxValue) and quantise every component through the same logistic map the reference's decoder
    inverts (J/JVectorIndexQuantization.java:319-361).  Returns (params [n][M][4] f32, bytes [n][d] u8, mean [d] f32)."""
    x = np.asarray(base, dtype=np.float32)
    n, d = x.shape
    mean = x.mean(axis=0).astype(np.float32)
    c = (x - mean).astype(np.float64)
    sizes = [d // M + (1 if m < d % M else 0) for m in range(M)]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
    params = np.zeros((n, M, 4), dtype=np.float32)
    codes = np.zeros((n, d), dtype=np.uint8)
    for m in range(M):
        seg = c[:, offs[m]:offs[m + 1]]
        mn = seg.min(axis=1)
        mx = seg.max(axis=1)
        mx = np.where(mx - mn < 1e-6, mn + 1e-6, mx)
        delta = mx - mn
        sg = growth / delta
        sm = midpoint * delta

        def logistic(v):
            # logisticNQT (J/JVectorIndexQuantization.java:344-350): 2^t with a piecewise-linear mantissa, so that the
            # reference's logitNQT decoder inverts it up to the 8-bit step
            t = (v * sg[:, None] - sg[:, None] * sm[:, None]).astype(np.float32)
            p = np.floor(t + np.float32(1.0)).astype(np.int32)
            f = ((t - p.astype(np.float32)) * np.float32(0.5) + np.float32(1.0)).astype(np.float32)
            z = (f.view(np.int32) + (p << 23)).view(np.float32).astype(np.float64)
            return z / (z + 1.0)
        lo = logistic(mn[:, None])
        hi = logistic(mx[:, None])
        q = np.rint((logistic(seg) - lo) / np.maximum(hi - lo, 1e-12) * 255.0)
        codes[:, offs[m]:offs[m + 1]] = np.clip(q, 0, 255).astype(np.uint8)
        params[:, m, 0] = growth
        params[:, m, 1] = midpoint
        params[:, m, 2] = mn.astype(np.float32)
        params[:, m, 3] = mx.astype(np.float32)
    return params, codes, mean

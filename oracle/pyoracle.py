"""ctypes view of the CPU oracle (oracle/libjvoracle.so).  TEST INFRASTRUCTURE: import this only from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg (see oracle/jv_oracle.h)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libjvoracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def load(desc_type):
    """desc_type: the ctypes JvIndexDesc structure class of the package binding (same C struct)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        lib = C.CDLL(LIB_PATH)
        vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
        P = C.POINTER(desc_type)
        lib.jvo_search.argtypes = [P, vp, i32, i32, f32, f32, vp, i64, vp, vp, vp, vp, vp]
        lib.jvo_search.restype = C.c_int
        lib.jvo_search_batch.argtypes = [P, vp, i32, i32, i32, f32, f32, vp, i64, vp, vp, vp, vp, vp, C.c_int]
        lib.jvo_search_batch.restype = C.c_int
        lib.jvo_score_ordinals.argtypes = [P, vp, vp, i32, vp]
        lib.jvo_score_ordinals.restype = None
        lib.jvo_brute_force.argtypes = [P, vp, i32, i32, vp, vp, vp, C.c_int]
        lib.jvo_brute_force.restype = None
        lib.jvo_merge_topk.argtypes = [vp, vp, i32, i32, i32, vp, vp]
        lib.jvo_merge_topk.restype = None
        lib.jvo_exact_score.argtypes = [i32, f32, vp, vp, i32]
        lib.jvo_exact_score.restype = f32
        lib.jvo_raw_dot.argtypes = [vp, vp, i32]
        lib.jvo_raw_dot.restype = f32
        lib.jvo_raw_l2.argtypes = [vp, vp, i32]
        lib.jvo_raw_l2.restype = f32
        lib.jvo_encode_key.argtypes = [i32, f32]
        lib.jvo_encode_key.restype = i64
        lib.jvo_float_to_sortable_int.argtypes = [f32]
        lib.jvo_float_to_sortable_int.restype = i32
        lib.jvo_pq_build_lut.argtypes = [P, vp, vp]
        lib.jvo_pq_build_lut.restype = None
        lib.jvo_nvq_dequantize.argtypes = [P, i32, vp]
        lib.jvo_nvq_dequantize.restype = None
        lib.jvo_parallel_copy.argtypes = [vp, vp, C.c_size_t]
        lib.jvo_parallel_copy.restype = None
        lib.jvo_set_simd.argtypes = [C.c_int]
        lib.jvo_set_simd.restype = None
        lib.jvo_get_simd.argtypes = []
        lib.jvo_get_simd.restype = C.c_int
        _lib = lib
    return _lib


def spread_to_host(desc_type, device_tensor, slab_bytes=1 << 30):
    """numpy copy of a (row-major, contiguous) torch device tensor whose pages are first touched by all OpenMP threads
    (jvo_parallel_copy), slab by slab, so that a multi-socket host serves the index from every NUMA node."""
    import numpy as np
    lib = load(desc_type)
    t = device_tensor
    out = np.empty(tuple(t.shape), dtype={4: {True: np.float32, False: np.int32}, 1: {False: np.uint8}}[t.element_size()][t.is_floating_point()])
    if t.numel() == 0:
        return out
    row_bytes = out.strides[0]
    rows = max(1, slab_bytes // max(1, row_bytes))
    for a in range(0, t.shape[0], rows):
        part = t[a:a + rows].cpu().numpy()
        lib.jvo_parallel_copy(out[a:].ctypes.data, part.ctypes.data, part.nbytes)
    return out


class Oracle:
    """The oracle over one index (a jv_index_desc on host arrays)."""

    def __init__(self, binding, ix):
        self.b = binding
        self.lib = load(binding.JvIndexDesc)
        self.desc, self._keep = binding.make_desc(ix)
        self.d = ix.d
        self.n = ix.n

    def search_batch(self, queries, topK, rerankK, threshold=0.0, rerank_floor=0.0, accept=None,
                     accept_num_docs=0, threads=0):
        q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, self.d)
        nq = q.shape[0]
        nodes = np.full((nq, topK), -1, dtype=np.int32)
        docs = np.full((nq, topK), -1, dtype=np.int32)
        scores = np.zeros((nq, topK), dtype=np.float32)
        count = np.zeros(nq, dtype=np.int32)
        stats = np.zeros((nq, 4), dtype=np.int32)
        acc = None if accept is None else np.ascontiguousarray(accept, dtype=np.uint64)
        used = self.lib.jvo_search_batch(C.byref(self.desc), q.ctypes.data, nq, topK, rerankK, threshold,
                                         rerank_floor, None if acc is None else acc.ctypes.data, accept_num_docs,
                                         nodes.ctypes.data, docs.ctypes.data, scores.ctypes.data,
                                         count.ctypes.data, stats.ctypes.data, threads)
        res = self.b.SearchResult(nodes, docs, scores, count, stats)
        res.threads = used
        return res

    def search(self, query, topK, rerankK, **kw):
        return self.search_batch(np.asarray(query, dtype=np.float32).reshape(1, -1), topK, rerankK, threads=1, **kw)

    def check_args(self, topK, rerankK):
        q = np.zeros(self.d, dtype=np.float32)
        return self.lib.jvo_search(C.byref(self.desc), q.ctypes.data, topK, rerankK, 0.0, 0.0, None, 0,
                                   None, None, None, None, None)

    def score_ordinals(self, query, ordinals):
        q = np.ascontiguousarray(query, dtype=np.float32)
        o = np.ascontiguousarray(ordinals, dtype=np.int32)
        out = np.zeros(o.shape[0], dtype=np.float32)
        self.lib.jvo_score_ordinals(C.byref(self.desc), q.ctypes.data, o.ctypes.data, o.shape[0], out.ctypes.data)
        return out

    def nvq_dequantize(self, node):
        out = np.zeros(self.d, dtype=np.float32)
        self.lib.jvo_nvq_dequantize(C.byref(self.desc), int(node), out.ctypes.data)
        return out

    def brute_force(self, queries, k, accept=None, threads=0):
        q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, self.d)
        nq = q.shape[0]
        nodes = np.full((nq, k), -1, dtype=np.int32)
        scores = np.zeros((nq, k), dtype=np.float32)
        acc = None if accept is None else np.ascontiguousarray(accept, dtype=np.uint64)
        self.lib.jvo_brute_force(C.byref(self.desc), q.ctypes.data, nq, k, None if acc is None else acc.ctypes.data,
                                 nodes.ctypes.data, scores.ctypes.data, threads)
        return nodes, scores


def merge_topk(binding, docs, scores, k):
    lib = load(binding.JvIndexDesc)
    docs = np.ascontiguousarray(docs, dtype=np.int32)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    nq, total = docs.shape
    lists = total // k
    od = np.zeros((nq, k), dtype=np.int32)
    os_ = np.zeros((nq, k), dtype=np.float32)
    lib.jvo_merge_topk(docs.ctypes.data, scores.ctypes.data, nq, lists, k, od.ctypes.data, os_.ctypes.data)
    return od, os_

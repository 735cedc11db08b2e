"""ctypes view of the C ABI declared in include/jvgpu.h (lib/libjvgpu.so).

Plumbing only: it marshals numpy arrays / raw device pointers into the C calls.  There is no CPU
fallback — if the HIP library is missing or no GPU is present the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libjvgpu.so")

JV_OK, JV_EINVAL, JV_ENOMEM, JV_EDEVICE, JV_EUNSUPPORTED, JV_EINTERNAL = 0, -1, -2, -3, -4, -5
SIM_EUCLIDEAN, SIM_DOT_PRODUCT, SIM_COSINE = 0, 1, 2
DESC_DEVICE_POINTERS, DESC_BORROW, DESC_FUSED_ADC, DESC_BUILD_CLIENT = 0x1, 0x2, 0x4, 0x8
NUM_STATS = 4

# every symbol include/jvgpu.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "jv_index_create", "jv_index_destroy", "jv_search", "jv_search_batch", "jv_search_batch_device",
    "jv_score_ordinals", "jv_merge_topk_device", "jv_index_get_info", "jv_set_option", "jv_last_error",
    "jv_abi_version", "jv_search_ex", "jv_search_batch_ex", "jv_index_set_option", "jv_index_get_counter", "jv_shard_group_create",
    "jv_shard_group_destroy", "jv_search_sharded_batch", "jv_search_sharded_batch_ex", "jv_shard_group_set_option",
    "jv_score_ordinals_batch", "jv_score_ordinals_batch_device", "jv_exact_search",
]
XB_NO_PREFILTER, XB_FORCE_PREFILTER, XB_TOPK_MAX, XB_INFO_WORDS = 0x1, 0x2, 1024, 4
QFLAG_RETRIED_BIG, QFLAG_EARLY_TERMINATED = 0x1, 0x2


class JvError(RuntimeError):
    """A failing C-ABI call; ``code`` is the jv_status, the text is jv_last_error()."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"jvgpu error {code}: {msg}")
        self.code = code


class JvLayerDesc(C.Structure):
    _fields_ = [("count", C.c_int32), ("degree", C.c_int32), ("nodes", C.c_void_p), ("adj", C.c_void_p)]


class JvIndexDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("flags", C.c_uint32), ("device", C.c_int32), ("n", C.c_int32),
        ("d", C.c_int32), ("R", C.c_int32), ("similarity", C.c_int32), ("score_scale", C.c_float),
        ("entry_node", C.c_int32), ("num_upper_layers", C.c_int32), ("upper_layers", C.c_void_p),
        ("vectors", C.c_void_p), ("adj", C.c_void_p),
        ("pq_M", C.c_int32), ("pq_K", C.c_int32), ("pq_sub_sizes", C.c_void_p), ("pq_codebooks", C.c_void_p),
        ("pq_centroid", C.c_void_p), ("pq_codes", C.c_void_p),
        ("ord2doc", C.c_void_p), ("max_doc", C.c_int32), ("reserved", C.c_int32),
        ("nvq_M", C.c_int32), ("reserved2", C.c_int32), ("nvq_sub_sizes", C.c_void_p), ("nvq_params", C.c_void_p),
        ("nvq_bytes", C.c_void_p), ("nvq_global_mean", C.c_void_p),
    ]


class JvExactBatchParams(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("topK", C.c_int32), ("accept_doc_words", C.c_void_p), ("accept_num_docs", C.c_int64),
                ("ordinals", C.c_void_p), ("count", C.c_int32), ("flags", C.c_uint32), ("accept_key", C.c_uint64)]


class JvIndexInfo(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("d", C.c_int32), ("R", C.c_int32), ("similarity", C.c_int32), ("pq_M", C.c_int32),
        ("pq_K", C.c_int32), ("num_upper_layers", C.c_int32), ("device", C.c_int32), ("hbm_bytes", C.c_int64),
        ("row_stride_floats", C.c_int32), ("fused_adc", C.c_int32), ("scratch_bytes", C.c_int64),
        ("filter_cache_hits", C.c_int64), ("filter_cache_misses", C.c_int64),
    ]


class JvSearchParams(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("topK", C.c_int32), ("rerankK", C.c_int32), ("threshold", C.c_float),
        ("rerankFloor", C.c_float), ("accept_doc_words", C.c_void_p), ("accept_num_docs", C.c_int64),
        ("visit_limit", C.c_int64), ("accept_key", C.c_uint64),
    ]


@dataclass
class IndexData:
    """Host-side flattened index: what FieldEntry holds after load (J/JVectorReader.java:284-337)."""
    vectors: np.ndarray                  # [n][d] float32
    adj: np.ndarray                      # [n][R] int32, -1 padded
    entry_node: int
    similarity: int = SIM_EUCLIDEAN
    score_scale: float = 1.0
    upper_nodes: List[np.ndarray] = field(default_factory=list)   # layer l+1: ascending ordinals
    upper_adj: List[np.ndarray] = field(default_factory=list)     # layer l+1: [count][degree]
    pq_codebooks: Optional[np.ndarray] = None   # flat float32
    pq_centroid: Optional[np.ndarray] = None    # [d]
    pq_codes: Optional[np.ndarray] = None       # [n][M] uint8
    pq_M: int = 0
    pq_K: int = 0
    ord2doc: Optional[np.ndarray] = None        # [n] int32
    max_doc: int = 0
    # NVQ-inline vectors (quantType 2): exact scores are taken against the dequantised record
    nvq_M: int = 0
    nvq_params: Optional[np.ndarray] = None     # [n][nvq_M][4] float32: growthRate, midpoint, minValue, maxValue
    nvq_bytes: Optional[np.ndarray] = None      # [n][d] uint8
    nvq_global_mean: Optional[np.ndarray] = None  # [d] float32

    @property
    def n(self) -> int:
        return int(self.vectors.shape[0])

    @property
    def d(self) -> int:
        return int(self.vectors.shape[1])

    @property
    def R(self) -> int:
        return int(self.adj.shape[1])


def _ptr(a: Optional[np.ndarray]) -> Optional[int]:
    return None if a is None else a.ctypes.data


def make_desc(ix: IndexData, device: int = 0, flags: int = 0):
    """Build a jv_index_desc over host numpy arrays. Returns (desc, keepalive)."""
    keep = []

    def own(a, dtype):
        if a is None:
            return None
        b = np.ascontiguousarray(a, dtype=dtype)
        keep.append(b)
        return b

    vec = own(ix.vectors, np.float32)
    adj = own(ix.adj, np.int32)
    desc = JvIndexDesc()
    desc.struct_size = C.sizeof(JvIndexDesc)
    desc.flags = flags
    desc.device = device
    desc.n, desc.d, desc.R = ix.n, ix.d, ix.R
    desc.similarity = ix.similarity
    desc.score_scale = ix.score_scale
    desc.entry_node = ix.entry_node
    desc.vectors = _ptr(vec)
    desc.adj = _ptr(adj)
    nl = len(ix.upper_nodes)
    desc.num_upper_layers = nl
    if nl:
        arr = (JvLayerDesc * nl)()
        for l in range(nl):
            nd = own(ix.upper_nodes[l], np.int32)
            ad = own(ix.upper_adj[l], np.int32)
            arr[l].count = nd.shape[0]
            arr[l].degree = ad.shape[1] if ad.ndim == 2 and ad.shape[0] else max(1, ix.R)
            arr[l].nodes = _ptr(nd)
            arr[l].adj = _ptr(ad)
        keep.append(arr)
        desc.upper_layers = C.cast(arr, C.c_void_p)
    desc.pq_M = ix.pq_M
    desc.pq_K = ix.pq_K
    desc.pq_sub_sizes = None
    desc.pq_codebooks = _ptr(own(ix.pq_codebooks, np.float32))
    desc.pq_centroid = _ptr(own(ix.pq_centroid, np.float32))
    desc.pq_codes = _ptr(own(ix.pq_codes, np.uint8))
    desc.ord2doc = _ptr(own(ix.ord2doc, np.int32))
    desc.max_doc = ix.max_doc if ix.max_doc else ix.n
    desc.nvq_M = ix.nvq_M
    if ix.nvq_M:
        desc.nvq_params = _ptr(own(ix.nvq_params, np.float32))
        desc.nvq_bytes = _ptr(own(ix.nvq_bytes, np.uint8))
        desc.nvq_global_mean = _ptr(own(ix.nvq_global_mean, np.float32))
    return desc, keep


def make_desc_device(n: int, d: int, R: int, vectors_ptr: int, adj_ptr: int, entry_node: int, similarity: int,
                     device: int = 0, score_scale: float = 1.0, pq_M: int = 0, pq_K: int = 0,
                     pq_codebooks: Optional[np.ndarray] = None, pq_centroid: Optional[np.ndarray] = None,
                     pq_codes_ptr: int = 0, ord2doc_ptr: int = 0, max_doc: int = 0, borrow: bool = True,
                     extra_flags: int = 0):
    """jv_index_desc over arrays that already live in HBM (raw device pointers, e.g. torch data_ptr()).
    Codebooks/centroid stay host arrays (tiny). Returns (desc, keepalive)."""
    keep = []
    desc = JvIndexDesc()
    desc.struct_size = C.sizeof(JvIndexDesc)
    desc.flags = DESC_DEVICE_POINTERS | (DESC_BORROW if borrow else 0) | extra_flags
    desc.device = device
    desc.n, desc.d, desc.R = n, d, R
    desc.similarity = similarity
    desc.score_scale = score_scale
    desc.entry_node = entry_node
    desc.vectors = vectors_ptr or None
    desc.adj = adj_ptr or None
    desc.num_upper_layers = 0
    desc.pq_M, desc.pq_K = pq_M, pq_K
    if pq_M:
        cb = np.ascontiguousarray(pq_codebooks, dtype=np.float32)
        keep.append(cb)
        desc.pq_codebooks = cb.ctypes.data
        if pq_centroid is not None:
            cen = np.ascontiguousarray(pq_centroid, dtype=np.float32)
            keep.append(cen)
            desc.pq_centroid = cen.ctypes.data
        desc.pq_codes = pq_codes_ptr or None
    desc.ord2doc = ord2doc_ptr or None
    desc.max_doc = max_doc if max_doc else n
    return desc, keep


def accept_words(doc_ids, num_docs: int) -> np.ndarray:
    """doc-space bitset (bit i = doc i accepted) as uint64 words, like Lucene's FixedBitSet.getBits()."""
    words = np.zeros((num_docs + 63) // 64, dtype=np.uint64)
    docs = np.asarray(doc_ids, dtype=np.int64).reshape(-1)
    if docs.size:
        np.bitwise_or.at(words, docs >> 6, np.uint64(1) << (docs & 63).astype(np.uint64))
    return words


_lib = None


def load_library(path: str = LIB_PATH) -> C.CDLL:
    """dlopen the engine. Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} is missing: build the HIP engine first (python -c 'import __graft_entry__ as g; g.build()')")
    lib = C.CDLL(path)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    lib.jv_index_create.argtypes = [C.POINTER(JvIndexDesc), C.POINTER(vp)]
    lib.jv_index_create.restype = C.c_int
    lib.jv_index_destroy.argtypes = [vp]
    lib.jv_index_destroy.restype = None
    lib.jv_search.argtypes = [vp, vp, i32, i32, f32, f32, vp, i64, vp, vp, vp, vp, vp]
    lib.jv_search.restype = C.c_int
    lib.jv_search_batch.argtypes = [vp, vp, i32, i32, i32, f32, f32, vp, i64, vp, vp, vp, vp, vp]
    lib.jv_search_batch.restype = C.c_int
    lib.jv_search_batch_device.argtypes = [vp, vp, i32, i32, i32, f32, f32, vp, i64, vp, vp, vp, vp, vp, vp, vp]
    lib.jv_search_batch_device.restype = C.c_int
    lib.jv_score_ordinals.argtypes = [vp, vp, vp, i32, vp]
    lib.jv_score_ordinals.restype = C.c_int
    lib.jv_score_ordinals_batch.argtypes = [vp, vp, i32, C.POINTER(JvExactBatchParams), vp, vp, vp, vp, vp]
    lib.jv_score_ordinals_batch.restype = C.c_int
    lib.jv_score_ordinals_batch_device.argtypes = [vp, vp, i32, C.POINTER(JvExactBatchParams), vp, vp, vp, vp, vp, vp]
    lib.jv_score_ordinals_batch_device.restype = C.c_int
    lib.jv_exact_search.argtypes = [vp, vp, C.POINTER(JvExactBatchParams), vp, vp, vp, vp]
    lib.jv_exact_search.restype = C.c_int
    lib.jv_merge_topk_device.argtypes = [i32, vp, vp, i32, i32, i32, vp, vp, vp]
    lib.jv_merge_topk_device.restype = C.c_int
    lib.jv_index_get_info.argtypes = [vp, C.POINTER(JvIndexInfo)]
    lib.jv_index_get_info.restype = C.c_int
    lib.jv_set_option.argtypes = [C.c_char_p, i64]
    lib.jv_set_option.restype = C.c_int
    lib.jv_last_error.argtypes = []
    lib.jv_last_error.restype = C.c_char_p
    lib.jv_abi_version.argtypes = []
    lib.jv_abi_version.restype = C.c_int
    lib.jv_search_ex.argtypes = [vp, vp, C.POINTER(JvSearchParams), vp, vp, vp, vp, vp, vp]
    lib.jv_search_ex.restype = C.c_int
    lib.jv_search_batch_ex.argtypes = [vp, vp, i32, C.POINTER(JvSearchParams), vp, vp, vp, vp, vp, vp, vp]
    lib.jv_search_batch_ex.restype = C.c_int
    lib.jv_index_set_option.argtypes = [vp, C.c_char_p, i64]
    lib.jv_index_set_option.restype = C.c_int
    lib.jv_index_get_counter.argtypes = [vp, C.c_char_p, C.POINTER(i64)]
    lib.jv_index_get_counter.restype = C.c_int
    lib.jv_shard_group_create.argtypes = [C.POINTER(vp), i32, C.POINTER(vp)]
    lib.jv_shard_group_create.restype = C.c_int
    lib.jv_shard_group_destroy.argtypes = [vp]
    lib.jv_shard_group_destroy.restype = None
    lib.jv_search_sharded_batch.argtypes = [vp, vp, i32, i32, i32, f32, f32, vp, vp, vp, vp]
    lib.jv_search_sharded_batch.restype = C.c_int
    lib.jv_search_sharded_batch_ex.argtypes = [vp, vp, i32, C.POINTER(JvSearchParams), vp, vp, vp, vp, vp, vp]
    lib.jv_search_sharded_batch_ex.restype = C.c_int
    lib.jv_shard_group_set_option.argtypes = [vp, C.c_char_p, i64]
    lib.jv_shard_group_set_option.restype = C.c_int
    _lib = lib
    return lib


def _check(lib, rc: int):
    if rc != JV_OK:
        raise JvError(rc, lib.jv_last_error().decode("utf-8", "replace"))


def set_option(name: str, value: int):
    """Default of a tunable for indexes created AFTER this call (existing handles: GpuIndex.set_option)."""
    lib = load_library()
    _check(lib, lib.jv_set_option(name.encode(), int(value)))


@dataclass
class SearchResult:
    nodes: np.ndarray    # [nq][topK] ordinals (-1 = empty)
    docs: np.ndarray     # [nq][topK] Lucene doc ids
    scores: np.ndarray   # [nq][topK]
    count: np.ndarray    # [nq]
    stats: np.ndarray    # [nq][4] visited, reranked, expanded, expandedBaseLayer


class GpuIndex:
    """An HBM-resident index handle (jv_index*)."""

    def __init__(self, ix: IndexData | None = None, device: int = 0, flags: int = 0, desc: JvIndexDesc | None = None,
                 keepalive=None):
        self.lib = load_library()
        if desc is None:
            desc, keepalive = make_desc(ix, device=device, flags=flags)
        self._keep = keepalive
        self.d = desc.d
        self.n = desc.n
        h = C.c_void_p()
        _check(self.lib, self.lib.jv_index_create(C.byref(desc), C.byref(h)))
        self.handle = h
        if not (flags & DESC_BORROW):
            self._keep = None  # the library copied everything

    def close(self):
        if getattr(self, "handle", None):
            self.lib.jv_index_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name: str, value: int):
        """Per-index tunable (jv_index_set_option)."""
        _check(self.lib, self.lib.jv_index_set_option(self.handle, name.encode(), int(value)))

    def counter(self, name: str) -> int:
        """Launch counters per kernel family (jv_index_get_counter)."""
        out = C.c_int64(0)
        _check(self.lib, self.lib.jv_index_get_counter(self.handle, name.encode(), C.byref(out)))
        return int(out.value)

    def _params(self, topK, rerankK, threshold, rerank_floor, acc, accept_num_docs, visit_limit, accept_key):
        p = JvSearchParams()
        p.struct_size = C.sizeof(JvSearchParams)
        p.topK, p.rerankK, p.threshold, p.rerankFloor = topK, rerankK, threshold, rerank_floor
        p.accept_doc_words = _ptr(acc)
        p.accept_num_docs = accept_num_docs
        p.visit_limit = visit_limit
        p.accept_key = accept_key
        return p

    def search_batch_ex(self, queries: np.ndarray, topK: int, rerankK: int, threshold: float = 0.0, rerank_floor: float = 0.0,
                        accept: Optional[np.ndarray] = None, accept_num_docs: int = 0, visit_limit: int = 0, accept_key: int = 0):
        """jv_search_batch_ex: returns (SearchResult, status [nq], flags [nq], rc) — never raises for per-query failures."""
        q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, self.d)
        nq = q.shape[0]
        nodes = np.full((nq, topK), -1, dtype=np.int32)
        docs = np.full((nq, topK), -1, dtype=np.int32)
        scores = np.zeros((nq, topK), dtype=np.float32)
        count = np.zeros(nq, dtype=np.int32)
        stats = np.zeros((nq, NUM_STATS), dtype=np.int32)
        status = np.zeros(nq, dtype=np.int32)
        flags = np.zeros(nq, dtype=np.int32)
        acc = None if accept is None else np.ascontiguousarray(accept, dtype=np.uint64)
        p = self._params(topK, rerankK, threshold, rerank_floor, acc, accept_num_docs, visit_limit, accept_key)
        rc = self.lib.jv_search_batch_ex(self.handle, q.ctypes.data, nq, C.byref(p), nodes.ctypes.data, docs.ctypes.data,
                                         scores.ctypes.data, count.ctypes.data, stats.ctypes.data, status.ctypes.data,
                                         flags.ctypes.data)
        if rc not in (JV_OK, JV_ENOMEM):
            _check(self.lib, rc)
        return SearchResult(nodes, docs, scores, count, stats), status, flags, rc

    def search_ex(self, query: np.ndarray, topK: int, rerankK: int, threshold: float = 0.0, rerank_floor: float = 0.0,
                  accept: Optional[np.ndarray] = None, accept_num_docs: int = 0, visit_limit: int = 0, accept_key: int = 0):
        """jv_search_ex: returns (SearchResult, flags)."""
        q = np.ascontiguousarray(query, dtype=np.float32).reshape(self.d)
        nodes = np.full((1, topK), -1, dtype=np.int32)
        docs = np.full((1, topK), -1, dtype=np.int32)
        scores = np.zeros((1, topK), dtype=np.float32)
        count = np.zeros(1, dtype=np.int32)
        stats = np.zeros((1, NUM_STATS), dtype=np.int32)
        flags = np.zeros(1, dtype=np.int32)
        acc = None if accept is None else np.ascontiguousarray(accept, dtype=np.uint64)
        p = self._params(topK, rerankK, threshold, rerank_floor, acc, accept_num_docs, visit_limit, accept_key)
        _check(self.lib, self.lib.jv_search_ex(self.handle, q.ctypes.data, C.byref(p), nodes.ctypes.data, docs.ctypes.data,
                                               scores.ctypes.data, count.ctypes.data, stats.ctypes.data, flags.ctypes.data))
        return SearchResult(nodes, docs, scores, count, stats), int(flags[0])

    def info(self) -> JvIndexInfo:
        out = JvIndexInfo()
        _check(self.lib, self.lib.jv_index_get_info(self.handle, C.byref(out)))
        return out

    def search_batch(self, queries: np.ndarray, topK: int, rerankK: int, threshold: float = 0.0,
                     rerank_floor: float = 0.0, accept: Optional[np.ndarray] = None,
                     accept_num_docs: int = 0) -> SearchResult:
        q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, self.d)
        nq = q.shape[0]
        nodes = np.full((nq, topK), -1, dtype=np.int32)
        docs = np.full((nq, topK), -1, dtype=np.int32)
        scores = np.zeros((nq, topK), dtype=np.float32)
        count = np.zeros(nq, dtype=np.int32)
        stats = np.zeros((nq, NUM_STATS), dtype=np.int32)
        acc = None if accept is None else np.ascontiguousarray(accept, dtype=np.uint64)
        _check(self.lib, self.lib.jv_search_batch(
            self.handle, q.ctypes.data, nq, topK, rerankK, threshold, rerank_floor, _ptr(acc), accept_num_docs,
            nodes.ctypes.data, docs.ctypes.data, scores.ctypes.data, count.ctypes.data, stats.ctypes.data))
        return SearchResult(nodes, docs, scores, count, stats)

    def search(self, query: np.ndarray, topK: int, rerankK: int, threshold: float = 0.0, rerank_floor: float = 0.0,
               accept: Optional[np.ndarray] = None, accept_num_docs: int = 0) -> SearchResult:
        q = np.ascontiguousarray(query, dtype=np.float32).reshape(self.d)
        nodes = np.full((1, topK), -1, dtype=np.int32)
        docs = np.full((1, topK), -1, dtype=np.int32)
        scores = np.zeros((1, topK), dtype=np.float32)
        count = np.zeros(1, dtype=np.int32)
        stats = np.zeros((1, NUM_STATS), dtype=np.int32)
        acc = None if accept is None else np.ascontiguousarray(accept, dtype=np.uint64)
        _check(self.lib, self.lib.jv_search(
            self.handle, q.ctypes.data, topK, rerankK, threshold, rerank_floor, _ptr(acc), accept_num_docs,
            nodes.ctypes.data, docs.ctypes.data, scores.ctypes.data, count.ctypes.data, stats.ctypes.data))
        return SearchResult(nodes, docs, scores, count, stats)

    def search_batch_device(self, d_queries: int, nq: int, topK: int, rerankK: int, d_nodes: int, d_docs: int,
                            d_scores: int, d_count: int, d_stats: int, d_flags: int = 0, stream: int = 0,
                            threshold: float = 0.0, rerank_floor: float = 0.0, d_accept: int = 0,
                            accept_num_docs: int = 0):
        """Raw device pointers (ints), e.g. torch tensors' data_ptr(); enqueued on `stream`."""
        _check(self.lib, self.lib.jv_search_batch_device(
            self.handle, d_queries, nq, topK, rerankK, threshold, rerank_floor, d_accept or None, accept_num_docs,
            d_nodes, d_docs or None, d_scores, d_count, d_stats, d_flags or None, stream or None))

    def score_ordinals(self, query: np.ndarray, ordinals: np.ndarray) -> np.ndarray:
        q = np.ascontiguousarray(query, dtype=np.float32).reshape(self.d)
        o = np.ascontiguousarray(ordinals, dtype=np.int32)
        out = np.zeros(o.shape[0], dtype=np.float32)
        _check(self.lib, self.lib.jv_score_ordinals(self.handle, q.ctypes.data, o.ctypes.data, o.shape[0], out.ctypes.data))
        return out


    def score_ordinals_batch(self, queries: np.ndarray, topK: int, accept: Optional[np.ndarray] = None, accept_num_docs: int = 0,
                             ordinals: Optional[np.ndarray] = None, flags: int = 0, accept_key: int = 0):
        """jv_score_ordinals_batch: the exact top-k of every query over ONE shared candidate set (doc filter, ordinal list, or
        every live ordinal).  Returns (nodes [nq][topK], docs, scores, count [nq], info [4])."""
        q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, self.d)
        nq = q.shape[0]
        nodes = np.full((nq, topK), -1, dtype=np.int32)
        docs = np.full((nq, topK), -1, dtype=np.int32)
        scores = np.zeros((nq, topK), dtype=np.float32)
        count = np.zeros(nq, dtype=np.int32)
        info = np.zeros(XB_INFO_WORDS, dtype=np.int64)
        acc = None if accept is None else np.ascontiguousarray(accept, dtype=np.uint64)
        o = None if ordinals is None else np.ascontiguousarray(ordinals, dtype=np.int32)
        p = JvExactBatchParams()
        p.struct_size = C.sizeof(JvExactBatchParams)
        p.topK = topK
        p.accept_doc_words = _ptr(acc)
        p.accept_num_docs = accept_num_docs
        p.ordinals = _ptr(o)
        p.count = 0 if o is None else int(o.shape[0])
        p.flags = flags
        p.accept_key = accept_key
        _check(self.lib, self.lib.jv_score_ordinals_batch(self.handle, q.ctypes.data, nq, C.byref(p), nodes.ctypes.data, docs.ctypes.data,
                                                          scores.ctypes.data, count.ctypes.data, info.ctypes.data))
        return nodes, docs, scores, count, info

    def exact_search(self, query: np.ndarray, topK: int, accept: np.ndarray, accept_num_docs: int, accept_key: int = 0):
        """jv_exact_search: ONE query under a doc filter (Lucene's exactSearch per leaf and query); concurrent calls with the same
        filter are combined into batch calls inside the library.  Returns (nodes [topK], docs, scores, count)."""
        q = np.ascontiguousarray(query, dtype=np.float32).reshape(self.d)
        nodes = np.full(topK, -1, dtype=np.int32)
        docs = np.full(topK, -1, dtype=np.int32)
        scores = np.zeros(topK, dtype=np.float32)
        count = np.zeros(1, dtype=np.int32)
        acc = np.ascontiguousarray(accept, dtype=np.uint64)
        p = JvExactBatchParams()
        p.struct_size = C.sizeof(JvExactBatchParams)
        p.topK = topK
        p.accept_doc_words = acc.ctypes.data
        p.accept_num_docs = accept_num_docs
        p.ordinals = None
        p.count = 0
        p.flags = 0
        p.accept_key = accept_key
        _check(self.lib, self.lib.jv_exact_search(self.handle, q.ctypes.data, C.byref(p), nodes.ctypes.data, docs.ctypes.data,
                                                  scores.ctypes.data, count.ctypes.data))
        return nodes, docs, scores, int(count[0])

    def score_ordinals_batch_device(self, d_queries: int, nq: int, topK: int, d_nodes: int, d_docs: int, d_scores: int, d_count: int,
                                    d_accept: int = 0, accept_num_docs: int = 0, d_ordinals: int = 0, count: int = 0, flags: int = 0,
                                    stream: int = 0, want_info: bool = False):
        """Raw device pointers (ints); enqueued behind `stream`.  Returns info [4] when want_info (that synchronises)."""
        p = JvExactBatchParams()
        p.struct_size = C.sizeof(JvExactBatchParams)
        p.topK = topK
        p.accept_doc_words = d_accept or None
        p.accept_num_docs = accept_num_docs
        p.ordinals = d_ordinals or None
        p.count = count
        p.flags = flags
        p.accept_key = 0
        info = np.zeros(XB_INFO_WORDS, dtype=np.int64) if want_info else None
        _check(self.lib, self.lib.jv_score_ordinals_batch_device(self.handle, d_queries, nq, C.byref(p), d_nodes or None, d_docs or None,
                                                                 d_scores or None, d_count or None, _ptr(info), stream or None))
        return info


class ShardGroup:
    """jv_shard_group: doc-range shards of one field searched and merged in one call (one process, any devices)."""

    def __init__(self, shards: Sequence[GpuIndex]):
        self.lib = load_library()
        self._shards = list(shards)
        arr = (C.c_void_p * len(shards))(*[s.handle for s in shards])
        h = C.c_void_p()
        _check(self.lib, self.lib.jv_shard_group_create(arr, len(shards), C.byref(h)))
        self.handle = h
        self.d = shards[0].d

    def close(self):
        if getattr(self, "handle", None):
            self.lib.jv_shard_group_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def search_batch(self, queries: np.ndarray, topK: int, rerankK: int, threshold: float = 0.0, rerank_floor: float = 0.0):
        q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, self.d)
        nq = q.shape[0]
        docs = np.full((nq, topK), -1, dtype=np.int32)
        scores = np.zeros((nq, topK), dtype=np.float32)
        count = np.zeros(nq, dtype=np.int32)
        stats = np.zeros((nq, NUM_STATS), dtype=np.int32)
        _check(self.lib, self.lib.jv_search_sharded_batch(self.handle, q.ctypes.data, nq, topK, rerankK, threshold, rerank_floor,
                                                          docs.ctypes.data, scores.ctypes.data, count.ctypes.data, stats.ctypes.data))
        return SearchResult(docs.copy(), docs, scores, count, stats)


    def set_option(self, name: str, value: int):
        """jv_shard_group_set_option ("gather": 0 peer copies, 1 RCCL all-gather)"""
        _check(self.lib, self.lib.jv_shard_group_set_option(self.handle, name.encode(), int(value)))

    def search_batch_ex(self, queries: np.ndarray, topK: int, rerankK: int, threshold: float = 0.0, rerank_floor: float = 0.0,
                        accept: Optional[np.ndarray] = None, accept_num_docs: int = 0, visit_limit: int = 0):
        """jv_search_sharded_batch_ex: doc filter over the GLOBAL doc-id space, visit limit per shard search; returns
        (SearchResult, status [nq], flags [nq], rc)."""
        q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, self.d)
        nq = q.shape[0]
        docs = np.full((nq, topK), -1, dtype=np.int32)
        scores = np.zeros((nq, topK), dtype=np.float32)
        count = np.zeros(nq, dtype=np.int32)
        stats = np.zeros((nq, NUM_STATS), dtype=np.int32)
        status = np.zeros(nq, dtype=np.int32)
        flags = np.zeros(nq, dtype=np.int32)
        acc = None if accept is None else np.ascontiguousarray(accept, dtype=np.uint64)
        p = JvSearchParams()
        p.struct_size = C.sizeof(JvSearchParams)
        p.topK, p.rerankK, p.threshold, p.rerankFloor = topK, rerankK, threshold, rerank_floor
        p.accept_doc_words = _ptr(acc)
        p.accept_num_docs = accept_num_docs
        p.visit_limit = visit_limit
        p.accept_key = 0
        rc = self.lib.jv_search_sharded_batch_ex(self.handle, q.ctypes.data, nq, C.byref(p), docs.ctypes.data, scores.ctypes.data,
                                                 count.ctypes.data, stats.ctypes.data, status.ctypes.data, flags.ctypes.data)
        if rc not in (JV_OK, JV_ENOMEM):
            _check(self.lib, rc)
        return SearchResult(docs.copy(), docs, scores, count, stats), status, flags, rc


def merge_topk_device(device: int, d_docs: int, d_scores: int, nq: int, lists: int, k: int, d_out_docs: int,
                      d_out_scores: int, stream: int = 0):
    lib = load_library()
    _check(lib, lib.jv_merge_topk_device(device, d_docs, d_scores, nq, lists, k, d_out_docs, d_out_scores, stream or None))

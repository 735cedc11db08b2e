"""ctypes view of the C++ host mirror (lib/libjvhost.so: JVectorReader, JVectorKnnFloatVectorQuery,
JVectorKnnCollector, GraphNodeIdToDocMap, VectorSimilarityMapper — see host/jvector_host.hpp)."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import binding

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libjvhost.so")
_lib = None

LUCENE_SIM = {"EUCLIDEAN": 0, "DOT_PRODUCT": 1, "COSINE": 2, "MAXIMUM_INNER_PRODUCT": 3}


class HostError(RuntimeError):
    """code: -1 IllegalArgumentException, -4 UnsupportedOperationException, -3 IOException"""

    def __init__(self, code, msg):
        super().__init__(f"host error {code}: {msg}")
        self.code = code


def load_library(path: str = LIB_PATH):
    global _lib
    if _lib is None:
        binding.load_library()  # libjvhost links against libjvgpu
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} is missing: run __graft_entry__.build()")
        lib = C.CDLL(path)
        vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
        lib.jvh_last_error.restype = C.c_char_p
        lib.jvh_reader_open.argtypes = [C.POINTER(binding.JvIndexDesc), i32, vp, i32, i32, C.POINTER(vp)]
        lib.jvh_reader_close.argtypes = [vp]
        lib.jvh_reader_close.restype = None
        lib.jvh_query_search_leaf.argtypes = [vp, vp, i32, i32, i32, f32, f32, vp, vp, i32, vp, vp, vp, vp, vp]
        lib.jvh_query_search_leaf_batch.argtypes = [vp, vp, i32, i32, i32, i32, f32, f32, vp, vp, i32, i32, C.c_double, vp, vp, vp, vp]
        lib.jvh_reader_search_plain_collector.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp, vp]
        lib.jvh_reader_search_bytes.argtypes = [vp]
        lib.jvh_counters.argtypes = [vp]
        lib.jvh_counters.restype = None
        lib.jvh_docmap_roundtrip.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp]
        lib.jvh_similarity_ord_to_dist_func.argtypes = [i32, vp]
        lib.jvh_similarity_dist_func_to_ord.argtypes = [i32]
        lib.jvh_concurrent_search_bench.argtypes = [vp, vp, i32, i32, i32, i32, i32, C.c_double, vp, vp]
        lib.jvh_concurrent_search_bench_filtered.argtypes = [vp, vp, i32, i32, i32, i32, i32, C.c_double, vp, vp, C.c_int64, C.c_uint64, vp]
        lib.jvh_meta_write.argtypes = [vp, C.c_char_p, i32, vp, i32, vp, C.c_int64, vp]
        lib.jvh_meta_read.argtypes = [vp, C.c_int64, vp, C.c_char_p, vp, i32, vp, vp, vp, C.c_int64]
        _lib = lib
    return _lib


def _check(lib, rc):
    if rc != 0:
        raise HostError(rc, lib.jvh_last_error().decode("utf-8", "replace"))


def counters():
    lib = load_library()
    out = (C.c_int64 * 5)()
    lib.jvh_counters(out)
    return dict(zip(["KNN_QUERY_VISITED_NODES", "KNN_QUERY_RERANKED_COUNT", "KNN_QUERY_EXPANDED_NODES",
                     "KNN_QUERY_EXPANDED_BASE_LAYER_NODES", "KNN_QUERY_GRAPH_SEARCH_TIME"], list(out)))


def docmap_roundtrip(ord2doc, max_doc_id, old_to_new=None):
    lib = load_library()
    o = np.ascontiguousarray(ord2doc, dtype=np.int32)
    otn = None if old_to_new is None else np.ascontiguousarray(old_to_new, dtype=np.int32)
    buf = np.zeros(16 + 5 * (len(o) + 2), dtype=np.uint8)
    nbytes = C.c_int32(buf.size)
    out_o2d = np.zeros(len(o), dtype=np.int32)
    out_d2o = np.zeros(max(max_doc_id + 1, (int(otn.max()) + 1) if otn is not None and otn.size else 0) + 1, dtype=np.int32)
    out_max = C.c_int32(0)
    _check(lib, lib.jvh_docmap_roundtrip(o.ctypes.data, len(o), max_doc_id, None if otn is None else otn.ctypes.data,
                                         buf.ctypes.data, C.addressof(nbytes), out_o2d.ctypes.data, out_d2o.ctypes.data,
                                         C.addressof(out_max)))
    return bytes(buf[:nbytes.value]), out_o2d, out_d2o[:out_max.value]


def ord_to_dist_func(ord_: int) -> int:
    lib = load_library()
    out = C.c_int(0)
    _check(lib, lib.jvh_similarity_ord_to_dist_func(ord_, C.addressof(out)))
    return out.value


def dist_func_to_ord(lucene_sim: int) -> int:
    return load_library().jvh_similarity_dist_func_to_ord(lucene_sim)


class JVectorReader:
    """One-field reader over a flattened index (the segment-open stand-in), searching on the GPU."""

    def __init__(self, ix: "binding.IndexData", lucene_similarity: str, device: int = 0):
        self.lib = load_library()
        self.ix = ix
        desc, self._keep = binding.make_desc(ix, device=device)
        o2d = np.ascontiguousarray(ix.ord2doc if ix.ord2doc is not None else np.arange(ix.n), dtype=np.int32)
        self.max_doc = ix.max_doc if ix.max_doc else ix.n
        h = C.c_void_p()
        _check(self.lib, self.lib.jvh_reader_open(C.byref(desc), LUCENE_SIM[lucene_similarity], o2d.ctypes.data, len(o2d),
                                                   self.max_doc - 1, C.byref(h)))
        self.handle = h
        self.d = ix.d

    def close(self):
        if self.handle:
            self.lib.jvh_reader_close(self.handle)
            self.handle = None

    def search_leaf(self, target, k, over_query_factor=5, threshold=0.0, rerank_floor=0.0,
                    filter_docs=None, deleted_docs=()):
        """new JVectorKnnFloatVectorQuery(field, target, k, filter, oqf, thr, floor) run against this leaf."""
        t = np.ascontiguousarray(target, dtype=np.float32)
        fw = None if filter_docs is None else binding.accept_words(filter_docs, self.max_doc)
        lw = None
        if len(deleted_docs):
            live = np.ones(self.max_doc, dtype=bool)
            live[list(deleted_docs)] = False
            lw = binding.accept_words(np.nonzero(live)[0], self.max_doc)
        docs = np.zeros(max(k, 1), dtype=np.int32)
        scores = np.zeros(max(k, 1), dtype=np.float32)
        count, exact = C.c_int32(0), C.c_int32(0)
        total = C.c_int64(0)
        _check(self.lib, self.lib.jvh_query_search_leaf(
            self.handle, t.ctypes.data, len(t), k, over_query_factor, threshold, rerank_floor,
            None if fw is None else fw.ctypes.data, None if lw is None else lw.ctypes.data, self.max_doc,
            docs.ctypes.data, scores.ctypes.data, C.addressof(count), C.addressof(total), C.addressof(exact)))
        return docs[:count.value].tolist(), scores[:count.value].tolist(), total.value, bool(exact.value)

    def search_leaf_batch(self, targets, k, over_query_factor=5, threshold=0.0, rerank_floor=0.0, filter_docs=None,
                          filter_words=None, deleted_docs=(), exact_when_cheaper=False, crossover_selectivity=0.25):
        """JVectorKnnFloatVectorQuery::searchLeafBatch: the per-leaf logic for MANY queries under one filter — every query gets
        what search_leaf gives it (the graph search with visitLimit = cost and the exact fallback are one engine call each).
        Returns (docs [nq][k], scores [nq][k], count [nq], used_exact [nq])."""
        t = np.ascontiguousarray(targets, dtype=np.float32).reshape(-1, self.d)
        nq = t.shape[0]
        fw = filter_words if filter_words is not None else (None if filter_docs is None else binding.accept_words(filter_docs, self.max_doc))
        lw = None
        if len(deleted_docs):
            live = np.ones(self.max_doc, dtype=bool)
            live[list(deleted_docs)] = False
            lw = binding.accept_words(np.nonzero(live)[0], self.max_doc)
        docs = np.full((nq, max(k, 1)), -1, dtype=np.int32)
        scores = np.zeros((nq, max(k, 1)), dtype=np.float32)
        count = np.zeros(nq, dtype=np.int32)
        exact = np.zeros(nq, dtype=np.int32)
        _check(self.lib, self.lib.jvh_query_search_leaf_batch(
            self.handle, t.ctypes.data, nq, self.d, k, over_query_factor, threshold, rerank_floor,
            None if fw is None else fw.ctypes.data, None if lw is None else lw.ctypes.data, self.max_doc,
            1 if exact_when_cheaper else 0, float(crossover_selectivity), docs.ctypes.data, scores.ctypes.data, count.ctypes.data,
            exact.ctypes.data))
        return docs, scores, count, exact.astype(bool)

    def search_plain_collector(self, target, k, accept_docs=None):
        t = np.ascontiguousarray(target, dtype=np.float32)
        aw = None if accept_docs is None else binding.accept_words(accept_docs, self.max_doc)
        docs = np.zeros(max(k, 1), dtype=np.int32)
        scores = np.zeros(max(k, 1), dtype=np.float32)
        count = C.c_int32(0)
        visited = C.c_int64(0)
        _check(self.lib, self.lib.jvh_reader_search_plain_collector(
            self.handle, t.ctypes.data, k, None if aw is None else aw.ctypes.data, self.max_doc,
            docs.ctypes.data, scores.ctypes.data, C.addressof(count), C.addressof(visited)))
        return docs[:count.value].tolist(), scores[:count.value].tolist(), visited.value

    def search_bytes(self):
        _check(self.lib, self.lib.jvh_reader_search_bytes(self.handle))


def concurrent_search_bench(index: "binding.GpuIndex", queries: np.ndarray, topK: int, rerankK: int, threads: int,
                            seconds: float, check_nodes: np.ndarray | None = None, accept: np.ndarray | None = None,
                            accept_num_docs: int = 0, accept_key: int = 0) -> dict:
    """`threads` native threads each issuing one jv_search at a time on the same handle — the reference's calling
    pattern (T/index/engine/JVectorConcurrentQueryTests.java:78-138)."""
    lib = load_library()
    q = np.ascontiguousarray(queries, dtype=np.float32)
    chk = None if check_nodes is None else np.ascontiguousarray(check_nodes, dtype=np.int32)
    out = np.zeros(5, dtype=np.float64)
    acc = None if accept is None else np.ascontiguousarray(accept, dtype=np.uint64)
    _check(lib, lib.jvh_concurrent_search_bench_filtered(
        index.handle, q.ctypes.data, q.shape[0], q.shape[1], topK, rerankK, threads, float(seconds),
        None if chk is None else chk.ctypes.data, None if acc is None else acc.ctypes.data, accept_num_docs, accept_key, out.ctypes.data))
    return {"threads": threads, "qps": float(out[0]), "p50_ms": float(out[1]), "p99_ms": float(out[2]),
            "completed": int(out[3]), "mismatches": int(out[4])}


class MetaField(C.Structure):
    """One VectorIndexFieldMetadata record of a .meta-jvector file (J/JVectorWriter.java:512-563)."""
    _fields_ = [("fieldNumber", C.c_int32), ("vectorEncoding", C.c_int32), ("similarityOrd", C.c_int32),
                ("vectorDimension", C.c_int32), ("vectorIndexOffset", C.c_int64), ("vectorIndexLength", C.c_int64),
                ("compressedVectorsOffset", C.c_int64), ("compressedVectorsLength", C.c_int64),
                ("quantizationType", C.c_int32), ("degreeOverflow", C.c_float), ("mapSize", C.c_int32),
                ("mapMaxDoc", C.c_int32), ("ord2doc", C.c_void_p)]


def meta_write(segment_id: bytes, suffix: str, version: int, fields) -> bytes:
    """fields: list of dicts with the MetaField members + 'ord2doc' (int32 array).  Returns the file's bytes."""
    lib = load_library()
    arr = (MetaField * len(fields))()
    keep = []
    for i, f in enumerate(fields):
        o2d = np.ascontiguousarray(f["ord2doc"], dtype=np.int32)
        keep.append(o2d)
        for k in ("fieldNumber", "vectorEncoding", "similarityOrd", "vectorDimension", "vectorIndexOffset", "vectorIndexLength",
                  "compressedVectorsOffset", "compressedVectorsLength", "quantizationType", "degreeOverflow"):
            setattr(arr[i], k, f[k])
        arr[i].mapSize = o2d.shape[0]
        arr[i].mapMaxDoc = f["mapMaxDoc"]
        arr[i].ord2doc = o2d.ctypes.data
    cap = 1 << 20
    out = (C.c_uint8 * cap)()
    n = C.c_int64()
    sid = (C.c_uint8 * 16)(*segment_id)
    _check(lib, lib.jvh_meta_write(sid, suffix.encode(), version, arr, len(fields), out, cap, C.byref(n)))
    return bytes(out[:n.value])


def meta_read(data: bytes, segment_id: bytes, suffix: str):
    """Returns (version, [dict per field]) or raises HostError (-3: corrupt / truncated / mismatching file)."""
    lib = load_library()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    arr = (MetaField * 64)()
    nf, ver = C.c_int(), C.c_int()
    ords = np.zeros(1 << 20, dtype=np.int32)
    sid = (C.c_uint8 * 16)(*segment_id)
    _check(lib, lib.jvh_meta_read(buf, len(data), sid, suffix.encode(), arr, 64, C.byref(nf), C.byref(ver), ords.ctypes.data, ords.shape[0]))
    fields, used = [], 0
    for i in range(nf.value):
        f = {k: getattr(arr[i], k) for k, _ in MetaField._fields_ if k != "ord2doc"}
        f["ord2doc"] = ords[used:used + arr[i].mapSize].copy()
        used += arr[i].mapSize
        fields.append(f)
    return ver.value, fields

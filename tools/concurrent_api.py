#!/usr/bin/env python3
"""The reference's real calling pattern at the boundary: T searcher threads, one jv_search (one query) per call on
a shared handle (C3-like index, n from $N).  Prints queries/s, p50/p99 per thread count, and checks every answer
against the batch API's."""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
g.load_package()
b = importlib.import_module("opensearch_jvector_amd.binding")
gb = importlib.import_module("opensearch_jvector_amd.builder_gpu")
host = importlib.import_module("opensearch_jvector_amd.host")
import bench
n = int(os.environ.get("N", 2_000_000)); d = 768; M = int(os.environ.get("M", 32)); rk = int(os.environ.get("RK", 160)); NQ = 8192
secs = float(os.environ.get("SECS", 3))
dev = torch.device("cuda", 0)
zc, Bl, Bg = bench.make_block_generators(torch, d, dev, max(64, min(4096, n // 256)), M=M, per=2)
if os.environ.get("DIST"):   # bench.py's distributions ("rotated": the headline data, rerankK 1200)
    base, q = bench.make_pq_data(torch, os.environ["DIST"], n, NQ, d, M, 0, n, False, dev)
else:
    base = bench.gen_rows_block(torch, n, d, 42, 0, zc, Bl, Bg, 0.35, 0.1, 0.005, False, dev)
    q = bench.gen_rows_block(torch, NQ, d, 43, 0, zc, Bl, Bg, 0.35, 0.1, 0.005, False, dev)
adj, entry = gb.build_graph_gpu(torch, base, 0, R=32, L=100, verbose=False)
pq = gb.pq_train_encode_gpu(torch, base, M, 0)
desc, keep = b.make_desc_device(n, d, 32, base.data_ptr(), adj.data_ptr(), entry, 0, pq_M=M, pq_K=pq["K"], pq_codebooks=pq["codebooks"],
                                pq_centroid=pq["centroid"], pq_codes_ptr=pq["codes"].data_ptr(), borrow=True, extra_flags=b.DESC_FUSED_ADC)
ix = b.GpuIndex(desc=desc, keepalive=keep, flags=b.DESC_BORROW)
qh = q.cpu().numpy()
acc, acc_key = None, 0
if os.environ.get("SEL"):   # one doc filter of this selectivity on every call (KEY=1: named for the filter cache)
    acc = b.accept_words(np.nonzero(np.random.default_rng(5).random(n) < float(os.environ["SEL"]))[0], n)
    acc_key = 77 if os.environ.get("KEY") == "1" else 0
want = ix.search_batch(qh, 10, rk, accept=acc, accept_num_docs=(n if acc is not None else 0)).nodes
for opt in os.environ.get("JV_OPTS", "").split(","):
    if "=" in opt:
        k_, v_ = opt.split("=")
        ix.set_option(k_, int(v_))
if os.environ.get("BIG_FIRST"):  # diagnostic: a large device-API launch first, like bench.py does
    Bb = int(os.environ["BIG_FIRST"])
    qb = bench.gen_rows_block(torch, Bb, d, 45, 0, zc, Bl, Bg, 0.35, 0.1, 0.005, False, dev)
    ob = [torch.empty((Bb, 10), dtype=torch.int32, device=dev), torch.empty((Bb, 10), dtype=torch.int32, device=dev),
          torch.empty((Bb, 10), dtype=torch.float32, device=dev), torch.empty((Bb,), dtype=torch.int32, device=dev),
          torch.empty((Bb, 4), dtype=torch.int32, device=dev), torch.empty((Bb,), dtype=torch.int32, device=dev)]
    st = torch.cuda.Stream(device=dev)
    for _ in range(3):
        ix.search_batch_device(qb.data_ptr(), Bb, 10, rk, *[t_.data_ptr() for t_ in ob], stream=st.cuda_stream)
    st.synchronize()
    ix.search_batch(qh[:4096], 10, rk)
rows = []
for T in [int(x) for x in os.environ.get("THREADS", "1,8,32,64,128,256").split(",")]:
    r = host.concurrent_search_bench(ix, qh, 10, rk, T, secs, want, accept=acc, accept_num_docs=(n if acc is not None else 0), accept_key=acc_key)
    rows.append(r)
    r["counters"] = {nm: ix.counter(nm) for nm in ("launches_pqw", "launches_pqp", "launches_big", "launches_serve", "served_queries")}
    print(json.dumps(r), flush=True)
    assert r["mismatches"] == 0, "single-query answers differ from the batch API's"
print("done")

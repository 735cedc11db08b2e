#!/usr/bin/env python3
"""Throughput of filtered searches (shared doc filter, batch device API) on a C3-like index, n from $N."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
g.load_package()
b = importlib.import_module("opensearch_jvector_amd.binding")
stamps = os.environ.get("STAMPS", "0") == "1"   # diagnostic build (make stamps): cycles per expansion by phase of the pool wave
if stamps:
    b.LIB_PATH = os.path.join(os.path.dirname(b.LIB_PATH), "libjvgpu_stamps.so")
    b.load_library(b.LIB_PATH)
gb = importlib.import_module("opensearch_jvector_amd.builder_gpu")
import bench
n = int(os.environ.get("N", 2_000_000)); d = 768; M = 32; rk = int(os.environ.get("RK", 160)); B = int(os.environ.get("B", 16384))
dev = torch.device("cuda", 0)
base, q = bench.make_pq_data(torch, os.environ.get("DIST", "aligned"), n, B, d, M, 0, n, False, dev)
adj, entry = gb.build_graph_gpu(torch, base, 0, R=32, L=100, verbose=False)
CM = int(os.environ.get("CODEC_M", M))   # the codec's subspaces (192 = the plugin's default for 768-d fields); the DATA keeps the PQ-32 block structure
pq = gb.pq_train_encode_gpu(torch, base, CM, 0)
desc, keep = b.make_desc_device(n, d, 32, base.data_ptr(), adj.data_ptr(), entry, 0, pq_M=CM, pq_K=pq["K"], pq_codebooks=pq["codebooks"],
                                pq_centroid=pq["centroid"], pq_codes_ptr=pq["codes"].data_ptr(), borrow=True, extra_flags=b.DESC_FUSED_ADC)
ix = b.GpuIndex(desc=desc, keepalive=keep, flags=b.DESC_BORROW)
for key, val in os.environ.items():   # JV_OPT_<name>=<int> -> per-index option
    if key.startswith("JV_OPT_"):
        ix.set_option(key[len("JV_OPT_"):].lower(), int(val))
o = [torch.empty((B, 10), dtype=torch.int32, device=dev), torch.empty((B, 10), dtype=torch.int32, device=dev),
     torch.empty((B, 10), dtype=torch.float32, device=dev), torch.empty((B,), dtype=torch.int32, device=dev),
     torch.zeros((B, 4), dtype=torch.int32, device=dev), torch.empty((B,), dtype=torch.int32, device=dev)]
rng = np.random.default_rng(5)
for sel in [None if x == "None" else float(x) for x in os.environ.get("SELS", "None,0.9,0.5,0.3,0.2").split(",")]:
    if sel is None:
        acc_ptr = 0
    else:
        bits = rng.random(n) < sel
        words = np.packbits(bits, bitorder="little")
        words = np.concatenate([words, np.zeros((-len(words)) % 8, np.uint8)]).view(np.uint64)
        acc = torch.from_numpy(words.view(np.int64)).to(dev)
        acc_ptr = acc.data_ptr()
    if stamps:
        dbg = torch.zeros(16, dtype=torch.int64, device=dev)
        ix.set_option("dbg_ptr", dbg.data_ptr())
    for it in range(3):
        if stamps:
            dbg.zero_()
        torch.cuda.synchronize(); t = time.time()
        ix.search_batch_device(q.data_ptr(), B, 10, rk, *[t_.data_ptr() for t_ in o], d_accept=acc_ptr, accept_num_docs=(n if acc_ptr else 0))
        torch.cuda.synchronize(); dt = time.time() - t
    st = o[4].cpu().numpy().astype(np.float64).mean(0)
    fl = o[5].cpu().numpy().astype(np.uint32)
    xb = ""
    if sel is not None and os.environ.get("XB", "1") == "1":   # the batched exact scorer on the same queries and filter
        xo = [torch.empty((B, 10), dtype=torch.int32, device=dev), torch.empty((B, 10), dtype=torch.int32, device=dev),
              torch.empty((B, 10), dtype=torch.float32, device=dev), torch.empty((B,), dtype=torch.int32, device=dev)]
        for it in range(3):
            torch.cuda.synchronize(); t = time.time()
            info = ix.score_ordinals_batch_device(q.data_ptr(), B, 10, *[t_.data_ptr() for t_ in xo], d_accept=acc_ptr, accept_num_docs=n,
                                                  want_info=(it == 2))
            torch.cuda.synchronize(); dtx = time.time() - t
        gn, xn = o[0].cpu().numpy(), xo[0].cpu().numpy()
        rec = float(np.mean([len(set(gn[i]) & set(xn[i])) / 10 for i in range(min(B, 2048))]))
        xb = f" | exact scan of the filter: {B / dtx:9.0f} QPS (candidates {int(info[0])}, re-scored/query {info[2] / B:.0f}); recall@10 of the graph answer {rec:.4f}"
    print(f"selectivity {sel}: {B / dt:9.0f} QPS{xb}  visited {st[0]:.0f} expanded {st[2]:.0f}  big-path {int((fl & 1).sum())}  count<10: {int((o[3].cpu().numpy() < 10).sum())}", flush=True)
    if stamps:
        v = dbg.cpu().numpy().astype(np.float64); ne = max(st[2], 1) * B
        names = {15: "find", 0: "candidates + barrier A", 9: "wait for blocks", 1: "mark + log", 2: "ADC + exchange", 3: "boundary + rank + dedupe", 14: "ranks among new keys",
                 4: "shift + insert (incl. 14)", 5: "boundary tracking + trim", 6: "visited-count pass", 13: "rerank + top-K"}
        print("   cycles per expansion (pool wave): " + ", ".join(f"{nm} {v[i] / ne:.0f}" for i, nm in names.items()) + f"; passes/exp {v[8] / ne:.2f} dual/exp {v[10] / ne:.2f}", flush=True)
        ix.set_option("dbg_ptr", 0)
print("done")

#!/usr/bin/env python3
"""Per-phase cycle shares of the PQ pool loop from the diagnostic build (lib/libjvgpu_stamps.so).
Shares only — the stamped build is slower than the product and its run time is never quoted."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as g
g.load_package()
b = importlib.import_module("opensearch_jvector_amd.binding")
b.LIB_PATH = os.path.join(os.path.dirname(b.LIB_PATH), "libjvgpu_stamps.so")
b.load_library(b.LIB_PATH)
gb = importlib.import_module("opensearch_jvector_amd.builder_gpu")
import bench

n = int(os.environ.get("N", 2_000_000)); d = 768; M = 32; rk = int(os.environ.get("RK", 200)); B = int(os.environ.get("B", 65536))
dev = torch.device("cuda", 0)
base, q = bench.make_pq_data(torch, os.environ.get("DIST", "rotated"), n, B, d, M, 0, n, False, dev)
adj, entry = gb.build_graph_gpu(torch, base, 0, R=32, L=100, verbose=False)
pq = gb.pq_train_encode_gpu(torch, base, M, 0)
for fused in (1,):
    desc, keep = b.make_desc_device(n, d, 32, base.data_ptr(), adj.data_ptr(), entry, 0, pq_M=M, pq_K=pq["K"],
                                    pq_codebooks=pq["codebooks"], pq_centroid=pq["centroid"], pq_codes_ptr=pq["codes"].data_ptr(),
                                    borrow=True, extra_flags=(b.DESC_FUSED_ADC if fused else 0))
    ix = b.GpuIndex(desc=desc, keepalive=keep, flags=b.DESC_BORROW)
    for key, val in os.environ.items():   # JV_OPT_<name>=<int> -> per-index option
        if key.startswith("JV_OPT_"):
            ix.set_option(key[len("JV_OPT_"):].lower(), int(val))
    dbg = torch.zeros(16, dtype=torch.int64, device=dev)
    o = [torch.empty((B, 10), dtype=torch.int32, device=dev), torch.empty((B, 10), dtype=torch.int32, device=dev),
         torch.empty((B, 10), dtype=torch.float32, device=dev), torch.empty((B,), dtype=torch.int32, device=dev),
         torch.empty((B, 4), dtype=torch.int32, device=dev), torch.empty((B,), dtype=torch.int32, device=dev)]
    acc_ptr = 0
    if os.environ.get("SEL"):   # shared doc filter of this selectivity
        bits = np.random.default_rng(5).random(n) < float(os.environ["SEL"])
        words = np.packbits(bits, bitorder="little")
        words = np.concatenate([words, np.zeros((-len(words)) % 8, np.uint8)]).view(np.uint64)
        acc = torch.from_numpy(words.view(np.int64)).to(dev)
        acc_ptr = acc.data_ptr()
    for it in range(2):
        dbg.zero_()
        ix.set_option("dbg_ptr", dbg.data_ptr())
        torch.cuda.synchronize(); t = time.time()
        ix.search_batch_device(q.data_ptr(), B, 10, rk, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(),
                               o[4].data_ptr(), o[5].data_ptr(), d_accept=acc_ptr, accept_num_docs=(n if acc_ptr else 0))
        torch.cuda.synchronize(); dt = time.time() - t
    ix.set_option("dbg_ptr", 0)
    v = dbg.cpu().numpy().astype(np.float64)
    st = o[4].cpu().numpy().astype(np.float64).mean(0)
    # stamp slots of the persistent pool kernel (jv_pqp_body.h); JV_OPT_no_pqp=1 shows the round-1 kernel's slots instead
    names = ["find best/runner-up + block select", "prefetch issue + mark expanded + log", "ADC", "boundary test + rank search + dedupe", "ranks among new + shift + insert",
             "boundary + trim", "visited-count pass", "LUT build + entry point", "rerank + top-K"]
    cyc = np.concatenate([v[:8], v[13:14]])
    ne = max(st[2], 1) * B
    print(f"fused={fused} rk={rk}: {B / dt:.0f} QPS (stamped build), expansions/query {st[2]:.1f}, visited/query {st[0]:.1f}")
    for i, nme in enumerate(names):
        print(f"   {nme:36s} {100 * cyc[i] / cyc.sum():5.1f} %   {cyc[i] / ne:8.0f} cycles/expansion")
    print(f"   nk == 0: {v[9] / ne:.3f}, nk == 1: {v[10] / ne:.3f} of the expansions; ranks-among-new-keys part of the insert phase: {v[14] / ne:.0f} cycles/expansion")
    print(f"   per expansion: candidates past the boundary {v[8] / ne:.2f}, already-in-pool (chunk-first) {v[9] / ne:.2f}, already-in-pool {v[10] / ne:.2f}, "
          f"inserts {v[11] / ne:.2f}, whole chunks shifted per insert {v[12] / max(v[11], 1):.2f}")
    ix.close()

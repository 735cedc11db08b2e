#!/usr/bin/env python3
"""Why queries leave the headline kernel (jv_search_pqp_kernel): the bail-out reason rides in bits 8..11 of the flag word
when the ladder is switched off (option pqf_only).  1 score below threshold, 2 expansion log full, 3 more than 63 boundary
ties, 4 visited-count classes exhausted, 5 strict-admission tie (DESIGN.md "Single-pool search"), 6 rerankFloor corner."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as g
g.load_package()
b = importlib.import_module("opensearch_jvector_amd.binding")
gb = importlib.import_module("opensearch_jvector_amd.builder_gpu")
import bench

n = int(os.environ.get("N", 2_000_000)); d = 768; M = 32; rk = int(os.environ.get("RK", 1200)); B = int(os.environ.get("B", 65536))
dev = torch.device("cuda", 0)
base, q = bench.make_pq_data(torch, os.environ.get("DIST", "rotated"), n, B, d, M, 0, n, False, dev)
adj, entry = gb.build_graph_gpu(torch, base, 0, R=32, L=100, verbose=False)
pq = gb.pq_train_encode_gpu(torch, base, M, 0)
desc, keep = b.make_desc_device(n, d, 32, base.data_ptr(), adj.data_ptr(), entry, 0, pq_M=M, pq_K=pq["K"], pq_codebooks=pq["codebooks"],
                                pq_centroid=pq["centroid"], pq_codes_ptr=pq["codes"].data_ptr(), borrow=True, extra_flags=b.DESC_FUSED_ADC)
ix = b.GpuIndex(desc=desc, keepalive=keep, flags=b.DESC_BORROW)
o = [torch.empty((B, 10), dtype=torch.int32, device=dev), torch.empty((B, 10), dtype=torch.int32, device=dev),
     torch.empty((B, 10), dtype=torch.float32, device=dev), torch.empty((B,), dtype=torch.int32, device=dev),
     torch.zeros((B, 4), dtype=torch.int32, device=dev), torch.empty((B,), dtype=torch.int32, device=dev)]
ix.set_option("pqf_only", 1)
ix.search_batch_device(q.data_ptr(), B, 10, rk, *[t.data_ptr() for t in o])
torch.cuda.synchronize()
fl = o[5].cpu().numpy().astype(np.uint32)
ov = (fl & 0x80000000) != 0
why = (fl >> 8) & 0xF
print("flagged", int(ov.sum()), "of", B, {int(w): int(((why == w) & ov).sum()) for w in range(1, 8)})

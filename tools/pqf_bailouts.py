#!/usr/bin/env python3
"""Diagnostics: how many queries does the PQF kernel hand to the ladder, and why (C3-like index, n from $N)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
g.load_package()
b = importlib.import_module("opensearch_jvector_amd.binding")
if os.environ.get("JV_LIB"):
    b.LIB_PATH = os.path.join(os.path.dirname(b.LIB_PATH), os.environ["JV_LIB"])
    b.load_library(b.LIB_PATH)
gb = importlib.import_module("opensearch_jvector_amd.builder_gpu")
import bench
n = int(os.environ.get("N", 10_000_000)); d = 768; M = 32; rk = int(os.environ.get("RK", 160)); B = 65536
dev = torch.device("cuda", 0)
zc, Bl, Bg = bench.make_block_generators(torch, d, dev, 4096, M=M, per=2)
base = bench.gen_rows_block(torch, n, d, 42, 0, zc, Bl, Bg, 0.35, 0.1, 0.005, False, dev)
q = bench.gen_rows_block(torch, B, d, 43, 0, zc, Bl, Bg, 0.35, 0.1, 0.005, False, dev)
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
ix.set_option("pqf_only", 0)
fl = o[5].cpu().numpy().astype(np.uint32)
ov = (fl & 0x80000000) != 0
why = (fl >> 8) & 0xF
print("flagged", int(ov.sum()), "of", B, {int(w): int(((why == w) & ov).sum()) for w in range(1, 6)}, "(1 thr, 2 log, 3 tie slack, 4 visited-count, 5 strict-admission tie)")
st = o[4].cpu().numpy()
print("expanded: mean %.1f p99 %.0f max %d; visited: mean %.1f p99 %.0f max %d" % (st[~ov, 2].mean(), np.percentile(st[~ov, 2], 99), st[~ov, 2].max(), st[~ov, 0].mean(), np.percentile(st[~ov, 0], 99), st[~ov, 0].max()))

# cost of the ladder launches when nothing is flagged: PQF only vs full ladder, HIP events
def timed(only):
    ix.set_option("pqf_only", only)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for it in range(6):
        e0.record()
        ix.search_batch_device(q.data_ptr(), B, 10, rk, *[t.data_ptr() for t in o])
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ix.set_option("pqf_only", 0)
    return float(np.median(ts[1:]))
print("PQF only: %.3f ms; PQF + ladder: %.3f ms" % (timed(1), timed(0)))
for slots in (4096, 16384):
    ix.set_option("no_escalation", 0)
print("done")

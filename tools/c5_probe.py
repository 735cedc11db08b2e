#!/usr/bin/env python3
"""Why does a launch of 256 queries take ~3x a lone query's time?  Launch time by batch size for (a) distinct queries and (b) one
query repeated (no tail: every workgroup does the same work), plus the batch's largest expansion count.
env: N (docs), RK, DIST, JV_OPT_<name>."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
g.load_package()
b = importlib.import_module("opensearch_jvector_amd.binding")
gb = importlib.import_module("opensearch_jvector_amd.builder_gpu")
import bench
n = int(os.environ.get("N", 2_000_000)); d = 768; M = 32; rk = int(os.environ.get("RK", 1200))
dev = torch.device("cuda", 0)
base, q = bench.make_pq_data(torch, os.environ.get("DIST", "rotated"), n, 1024, d, M, 0, n, False, dev)
adj, entry = gb.build_graph_gpu(torch, base, 0, R=32, L=100, verbose=False)
pq = gb.pq_train_encode_gpu(torch, base, M, 0)
desc, keep = b.make_desc_device(n, d, 32, base.data_ptr(), adj.data_ptr(), entry, 0, pq_M=M, pq_K=pq["K"], pq_codebooks=pq["codebooks"],
                                pq_centroid=pq["centroid"], pq_codes_ptr=pq["codes"].data_ptr(), borrow=True, extra_flags=b.DESC_FUSED_ADC)
ix = b.GpuIndex(desc=desc, keepalive=keep, flags=b.DESC_BORROW)
for key, val in os.environ.items():
    if key.startswith("JV_OPT_"):
        ix.set_option(key[len("JV_OPT_"):].lower(), int(val))
BM = 1024
o = [torch.empty((BM, 10), dtype=torch.int32, device=dev), torch.empty((BM, 10), dtype=torch.int32, device=dev),
     torch.empty((BM, 10), dtype=torch.float32, device=dev), torch.empty((BM,), dtype=torch.int32, device=dev),
     torch.zeros((BM, 4), dtype=torch.int32, device=dev), torch.empty((BM,), dtype=torch.int32, device=dev)]


def timed(qq, B):
    best = 1e9
    for it in range(6):
        torch.cuda.synchronize(); t = time.time()
        ix.search_batch_device(qq.data_ptr(), B, 10, rk, *[t_.data_ptr() for t_ in o])
        torch.cuda.synchronize(); best = min(best, time.time() - t)
    return best


for B in [int(x) for x in os.environ.get("BS", "1,8,32,64,128,256,512,768,1024").split(",")]:
    t_a = timed(q, B)
    ex = o[4][:B, 2].cpu().numpy()
    same = q[:1].repeat(B, 1).contiguous()
    t_b = timed(same, B)
    print(f"B={B:5d}: distinct {t_a * 1e3:7.3f} ms (expansions mean {ex.mean():.0f} max {ex.max()}), one query repeated {t_b * 1e3:7.3f} ms "
          f"({int(o[4][0, 2])} expansions)", flush=True)
# the slowest query of the first 256, alone: is its time proportional to its expansions?
timed(q, 256)
ex = o[4][:256, 2].cpu().numpy(); vis = o[4][:256, 0].cpu().numpy()
for i in (int(np.argmax(ex)), int(np.argsort(ex)[128]), int(np.argmin(ex))):
    one = q[i:i + 1].contiguous()
    t1 = timed(one, 1)
    print(f"query {i}: {ex[i]} expansions, visited {vis[i]}: alone {t1 * 1e3:.3f} ms = {t1 * 1e9 / ex[i]:.0f} ns per expansion", flush=True)
ix.close()

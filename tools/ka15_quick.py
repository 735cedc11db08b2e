#!/usr/bin/env python3
"""KA15-style recall of the GPU builder's graphs (java.util.Random vectors, d = 128, L2, k = 10) by size, over-query factor and
number of refine passes; exact search through the C ABI against brute force."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
b = importlib.import_module("opensearch_jvector_amd.binding")
gb = importlib.import_module("opensearch_jvector_amd.builder_gpu")
dg = importlib.import_module("opensearch_jvector_amd.datagen")
dev = torch.device("cuda", 0)
for n in [int(x) for x in os.environ.get("NS", "1500,20000").split(",")]:
    base = dg.java_random_vectors(42, n, 128)
    nq = int(os.environ.get("NQ", 100))
    q = dg.java_random_vectors(43, nq, 128)
    bt = torch.from_numpy(base).to(dev)
    qt = torch.from_numpy(q).to(dev)
    d2 = (qt * qt).sum(1)[:, None] + (bt * bt).sum(1)[None, :] - 2 * qt @ bt.T
    truth = torch.topk(-d2, 10, dim=1).indices.cpu().numpy()
    if os.environ.get("CPU", "1") == "1":   # the C builder (libjvbuild.so: sequential insertion like jvector's addGraphNode) for scale
        bl = importlib.import_module("opensearch_jvector_amd.builder")
        t = time.time()
        ixc = bl.build_index_cpu(base, 0, R=32, L=100)
        bt_s = time.time() - t
        gpu = b.GpuIndex(ixc)
        out = []
        for oqf in (5, 20):
            got = gpu.search_batch(q, 10, 10 * oqf)
            rec10 = np.mean([len(set(got.nodes[i]) & set(truth[i])) / 10 for i in range(10)])
            rec = np.mean([len(set(got.nodes[i]) & set(truth[i])) / 10 for i in range(nq)])
            out.append(f"oqf {oqf}: first-10-queries {rec10:.3f} all {rec:.4f}")
        print(f"n={n} CPU sequential builder build {bt_s:.1f}s  " + "  ".join(out), flush=True)
        gpu.close()
    for passes in (0, 1, 2):
        t = time.time()
        adj, entry = gb.build_graph_gpu(torch, bt, 0, R=32, L=100, verbose=False, refine_passes=passes)
        torch.cuda.synchronize(); bt_s = time.time() - t
        ix = b.IndexData(vectors=base, adj=adj.cpu().numpy(), entry_node=entry, similarity=0)
        gpu = b.GpuIndex(ix)
        out = []
        for oqf in (5, 20):
            got = gpu.search_batch(q, 10, 10 * oqf)
            rec10 = np.mean([len(set(got.nodes[i]) & set(truth[i])) / 10 for i in range(10)])
            rec = np.mean([len(set(got.nodes[i]) & set(truth[i])) / 10 for i in range(nq)])
            out.append(f"oqf {oqf}: first-10-queries {rec10:.3f} all {rec:.4f}")
        print(f"n={n} refine_passes={passes} build {bt_s:.1f}s  " + "  ".join(out), flush=True)
        gpu.close()

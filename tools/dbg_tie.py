import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
g.load_package()
b = importlib.import_module("opensearch_jvector_amd.binding")
bl = importlib.import_module("opensearch_jvector_amd.builder")
pyoracle = g.load_oracle()
seed = 11
rng = np.random.default_rng(seed)
n, d, R = 500, int(os.environ.get('D', '4')), 16
M = 2 if d == 4 else 32
base = rng.integers(0, 3, size=(n, d)).astype(np.float32)
base[:, 4:] = 0
adj = np.stack([rng.permutation(n)[:R] for _ in range(n)]).astype(np.int32)
q = rng.integers(0, 3, size=(64, d)).astype(np.float32) + np.float32(0.5) * (rng.random((64, d)) < 0.3)
q[:, 4:] = 0
for sim in (0, 1):
    ix = b.IndexData(vectors=base, adj=adj, entry_node=int(rng.integers(0, n)), similarity=sim)
    cb, cen, codes, K = bl.pq_train_encode_cpu(base, M, sim)
    ixq = b.IndexData(vectors=base, adj=adj, entry_node=ix.entry_node, similarity=sim, pq_codebooks=cb, pq_centroid=cen, pq_codes=codes, pq_M=M, pq_K=K)
    gpu = b.GpuIndex(ixq, flags=b.DESC_FUSED_ADC)
    gpu.set_option("lutr_min_queries", 0)
    gpu.set_option("no_escalation", int(os.environ.get("NOESC", "0")))
    orc = pyoracle.Oracle(b, ixq)
    for k, rk in ((1, 1), (2, 2), (3, 4), (5, 8), (10, 16), (10, 40), (20, 100)):
        g = gpu.search_batch(q, k, rk); w = orc.search_batch(q, k, rk)
        bad = [i for i in range(64) if not (np.array_equal(g.stats[i], w.stats[i]) and np.array_equal(g.nodes[i], w.nodes[i]))]
        print("sim", sim, "k", k, "rk", rk, "bad", len(bad), [(i, g.stats[i].tolist(), w.stats[i].tolist(), g.nodes[i].tolist(), w.nodes[i].tolist()) for i in bad[:4]], flush=True)

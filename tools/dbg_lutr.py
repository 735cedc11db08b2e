import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
g.load_package()
b = importlib.import_module("opensearch_jvector_amd.binding")
bl = importlib.import_module("opensearch_jvector_amd.builder")
dg = importlib.import_module("opensearch_jvector_amd.datagen")
po = g.load_oracle()
base = dg.splitmix_uniform(42, 3000, 64); q = dg.splitmix_uniform(43, 8, 64)
ix = bl.build_index_cpu(base, 0, R=32, L=50, pq_M=32)
gpu = b.GpuIndex(ix, flags=b.DESC_FUSED_ADC)
orc = po.Oracle(b, ix)
gpu.set_option("lutr_min_queries", 0); gpu.set_option("no_lutr", 1)
for rk in (10, 50, 96, 97, 160, 161, 200, 300, 600, 1200):
    want = orc.search_batch(q, 10, rk)
    r, st, fl, rc = gpu.search_batch_ex(q, 10, rk)
    print(rk, "stats", r.stats[0].tolist(), "want", want.stats[0].tolist(), "eq", np.array_equal(r.nodes, want.nodes))

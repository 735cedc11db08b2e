"""Ad-hoc GPU check + timing used during development (not a pytest file)."""
import importlib, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
import numpy as np
import __graft_entry__ as g
g.load_package()
b = importlib.import_module("opensearch_jvector_amd.binding")
bl = importlib.import_module("opensearch_jvector_amd.builder")
dg = importlib.import_module("opensearch_jvector_amd.datagen")
po = g.load_oracle()
g.smoke()
n, d = int(os.environ.get("N", 20000)), int(os.environ.get("D", 768))
base = dg.l2_normalize(dg.gaussian_mixture(42, n, d, centres=256))
q = dg.l2_normalize(dg.gaussian_mixture(43, 2048, d, centres=256))
t = time.time(); ix = bl.build_index_cpu(base, 1, R=32, L=100); print("cpu build s", time.time() - t)
gpu = b.GpuIndex(ix)
o = po.Oracle(b, ix)
t = time.time(); want = o.search_batch(q, 10, 100); tc = time.time() - t
print("cpu qps", len(q) / tc, "threads", want.threads)
for it in range(3):
    t = time.time(); got = gpu.search_batch(q, 10, 100); tg = time.time() - t
    print("gpu qps", len(q) / tg)
print("ids equal", np.array_equal(got.nodes, want.nodes), "stats equal", np.array_equal(got.stats, want.stats),
      "bits equal", np.array_equal(got.scores.view(np.uint32), want.scores.view(np.uint32)))
print("mean stats", got.stats.mean(0))
gt, _ = o.brute_force(q[:256], 10)
print("recall", np.mean([len(set(got.nodes[i]) & set(gt[i])) / 10 for i in range(256)]))

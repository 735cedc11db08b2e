#!/usr/bin/env python3
"""How the CPU restatement (oracle, OpenMP one query per thread) scales with threads on this host: QPS at 1, 8, 32, 64,
128, 256 threads on a 2M x 768 PQ-32 index (same shape as C3, smaller n), plus the container's CPU quota — context for
bench.py's cpu_baseline figure."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as g
g.load_package()
b = importlib.import_module("opensearch_jvector_amd.binding")
gb = importlib.import_module("opensearch_jvector_amd.builder_gpu")
po = g.load_oracle()
import bench

for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    if os.path.exists(f):
        print(f, open(f).read().strip())
print("sched_getaffinity:", len(os.sched_getaffinity(0)), "os.cpu_count:", os.cpu_count())
n, d, M, rk = int(os.environ.get("N", 2_000_000)), 768, 32, int(os.environ.get("RK", 160))
dev = torch.device("cuda", 0)
base, q = bench.make_pq_data(torch, os.environ.get("DIST", "aligned"), n, 16384, d, M, 0, n, False, dev)
adj, entry = gb.build_graph_gpu(torch, base, 0, R=32, L=100, verbose=False)
pq = gb.pq_train_encode_gpu(torch, base, M, 0)
sp = lambda t: po.spread_to_host(b.JvIndexDesc, t.contiguous())
ix = b.IndexData(vectors=sp(base), adj=sp(adj), entry_node=entry, similarity=0)
ix.pq_codebooks, ix.pq_centroid, ix.pq_codes = pq["codebooks"], pq["centroid"], sp(pq["codes"])
ix.pq_M, ix.pq_K = M, pq["K"]
orc = po.Oracle(b, ix)
qs = q.cpu().numpy()
orc.search_batch(qs[:1024], 10, rk, threads=os.cpu_count())
for th in (1, 8, 16, 24, 32, 64, 128, 256):
    m = min(len(qs), 128 * th)
    t0 = time.time(); done = 0
    while time.time() - t0 < 4.0:      # long enough for the CFS quota to bite (short bursts run unthrottled)
        orc.search_batch(qs[:m], 10, rk, threads=th); done += m
    dt = time.time() - t0
    print(f"threads {th:4d}: {done / dt:10.1f} QPS ({done} queries, {dt:.2f}s)", flush=True)

#!/usr/bin/env python3
"""Wall-clock time and reproducibility of the GPU graph build (builder_gpu.build_graph_gpu) on C3-shaped data, and the recall the
built graph gives with the exact provider.  env: N (1M), D (768), DIST (rotated), REPEAT (2: the second build must equal the first)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
g.load_package()
gb = importlib.import_module("opensearch_jvector_amd.builder_gpu")
import bench
n = int(os.environ.get("N", 1_000_000)); d = int(os.environ.get("D", 768))
dev = torch.device("cuda", 0)
base, q = bench.make_pq_data(torch, os.environ.get("DIST", "rotated"), n, 4096, d, 32, 0, n, False, dev)
prev = None
for rep in range(int(os.environ.get("REPEAT", 2))):
    torch.cuda.synchronize(); t = time.time()
    adj, entry = gb.build_graph_gpu(torch, base, 0, R=32, L=100, verbose=False)
    torch.cuda.synchronize(); dt = time.time() - t
    deg = (adj >= 0).sum(1).float()
    same = None if prev is None else bool(torch.equal(prev, adj))
    print(f"build {rep}: {n} x {d} in {dt:.1f} s ({n / dt:,.0f} nodes/s), entry {entry}, mean degree {deg.mean().item():.2f}, min {int(deg.min().item())}; equal to the previous build: {same}", flush=True)
    prev = adj

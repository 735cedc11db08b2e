#!/usr/bin/env python3
"""Is the GPU graph build reproducible?  Builds the same index twice and compares the adjacency arrays row by row.
env: N (docs), D (dimension), M (data model's subspaces), DIST."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.load_package()
gb = importlib.import_module("opensearch_jvector_amd.builder_gpu")
import bench
n = int(os.environ.get("N", 500_000)); d = int(os.environ.get("D", 1536)); M = int(os.environ.get("M", 64))
dev = torch.device("cuda", 0)
base, _ = bench.make_pq_data(torch, os.environ.get("DIST", "rotated"), n, 64, d, M, 0, n, False, dev)
outs = []
for i in range(int(os.environ.get("RUNS", 3))):
    t0 = time.time()
    adj, entry = gb.build_graph_gpu(torch, base, 0, R=32, L=100, verbose=False)
    torch.cuda.synchronize()
    outs.append((adj.clone(), entry))
    print(f"build {i}: {time.time() - t0:.1f} s, entry {entry}, degree mean {(adj >= 0).sum(1).float().mean().item():.3f}", flush=True)
for i in range(1, len(outs)):
    diff = (outs[i][0] != outs[0][0]).any(1)
    srt = (torch.sort(outs[i][0], 1).values != torch.sort(outs[0][0], 1).values).any(1)
    print(f"build {i} vs 0: rows differing {int(diff.sum())} of {n} (as sets: {int(srt.sum())}), entry equal {outs[i][1] == outs[0][1]}")

#!/usr/bin/env python3
"""Host-pointer batch calls next to one-query traffic that makes the resident query-server grid start again and again
(bursts separated by pauses longer than serve_idle_ms): the latency of every batch call, and how many grid starts it saw.
env: SPIN (serve_spin_waiters), TWO (0), BOPT_*, IDLE_MS (3), BURST (20), PAUSE_MS (10), SECS (6), NQ (64), RK (120), N (4000), D (64)"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
import importlib
b = importlib.import_module("opensearch_jvector_amd.binding"); bl = importlib.import_module("opensearch_jvector_amd.builder"); dg = importlib.import_module("opensearch_jvector_amd.datagen")
E = lambda k, v: int(os.environ.get(k, v))
n, d, rk, nq = E("N", 4000), E("D", 64), E("RK", 120), E("NQ", 64)
base = dg.splitmix_uniform(31, n, d); q = dg.splitmix_uniform(32, 512, d)
ix = bl.build_index_cpu(base, 0, R=32, L=60, pq_M=32)
if "SPIN" in os.environ: b.set_option("serve_spin_waiters", E("SPIN", 4))   # (default of indexes created after this: 0 = naps only, round 5's 6214c03 made it 4)
gpu = b.GpuIndex(ix, flags=b.DESC_FUSED_ADC)
gpu.set_option("serve_idle_ms", E("IDLE_MS", 3))
gb = gpu
if E("TWO", 0):   # the batch calls go through a second handle on the same device (BOPT_<option>=<int> apply to it)
    gb = b.GpuIndex(ix, flags=b.DESC_FUSED_ADC)
    for k, v in os.environ.items():
        if k.startswith("BOPT_"): gb.set_option(k[5:].lower(), int(v))
want = gpu.search_batch(q, 10, rk)
stop = threading.Event(); lat = []; starts = []; bad = []

def singles(tid):
    i = tid
    while not stop.is_set():
        for _ in range(E("BURST", 20)):
            j = i % len(q); i += 7
            r = gpu.search(q[j], 10, rk)
            if not np.array_equal(r.nodes[0], want.nodes[j]): bad.append(("single", j))
        time.sleep(E("PAUSE_MS", 10) / 1000.0)

def batches():
    while not stop.is_set():
        s0 = gpu.counter("launches_serve"); t = time.perf_counter()
        r = gb.search_batch(q[:nq], 10, rk)
        lat.append((time.perf_counter() - t) * 1e3); starts.append(gpu.counter("launches_serve") - s0)
        if not np.array_equal(r.nodes, want.nodes[:nq]): bad.append(("batch",))

ts = [threading.Thread(target=singles, args=(t,)) for t in range(E("CALLERS", 4))] + [threading.Thread(target=batches)]
[t.start() for t in ts]; time.sleep(E("SECS", 6)); stop.set(); [t.join(timeout=120) for t in ts]
lat = np.array(lat[3:]); starts = np.array(starts[3:])   # (the first calls allocate their launch context)
print(f"grid starts {gpu.counter('launches_serve')}, served {gpu.counter('served_queries')}, batch calls {len(lat)}: p50 {np.percentile(lat,50):.2f} ms p99 {np.percentile(lat,99):.2f} ms max {lat.max():.2f} ms; "
      f"calls that overlapped a start: {int((starts>0).sum())}, their p50 {np.percentile(lat[starts>0],50) if (starts>0).any() else 0:.2f} max {lat[starts>0].max() if (starts>0).any() else 0:.2f} ms; wrong answers {len(bad)}; stuck threads {sum(t.is_alive() for t in ts)}")
gpu.close()

#!/usr/bin/env python3
"""Diagnostic: one-query jv_search traffic (served by the resident grid) next to exact calls; prints who makes progress.
usage (GPU box): JV_SERVE_TRACE=1 python tools/exact_beside_server_probe.py [seconds]"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
graft.load_package()
import importlib
b = importlib.import_module("opensearch_jvector_amd.binding")
bl = importlib.import_module("opensearch_jvector_amd.builder")
dg = importlib.import_module("opensearch_jvector_amd.datagen")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
rng = np.random.default_rng(5)
n, d = 20000, 64
base = dg.splitmix_uniform(61, n, d)
q = dg.splitmix_uniform(62, 256, d)
ix = bl.build_index_cpu(base, 0, R=32, L=60, pq_M=32)
gpu = b.GpuIndex(ix, flags=b.DESC_FUSED_ADC)
docs = np.nonzero(rng.random(n) < 0.3)[0].astype(np.int32)
words = b.accept_words(docs, n)
gpu.search_batch(q, 10, 120)
stop = threading.Event()
prog = {"search": [0] * 4, "exact": [0] * 2, "where": ["-"] * 2}

def searcher(t):
    i = t
    while not stop.is_set():
        gpu.search(q[i % len(q)], 10, 120)
        i += 5
        prog["search"][t] += 1

def exact(t):
    it = 0
    while not stop.is_set():
        qi = (t * 40 + it) % len(q)
        if it % 2:
            prog["where"][t] = "exact_search"
            gpu.exact_search(q[qi], 10, words, n)
        else:
            prog["where"][t] = "score_ordinals_batch"
            gpu.score_ordinals_batch(q[qi:qi + 1], 10, accept=words, accept_num_docs=n, flags=b.XB_FORCE_PREFILTER)
        prog["where"][t] = "between"
        it += 1
        prog["exact"][t] += 1

ts = [threading.Thread(target=searcher, args=(t,), daemon=True) for t in range(4)]
[t.start() for t in ts]
time.sleep(0.3)
es = [threading.Thread(target=exact, args=(t,), daemon=True) for t in range(int(os.environ.get("EXACT_THREADS", "2")))]
[t.start() for t in es]
t0 = time.time()
while time.time() - t0 < secs:
    time.sleep(1.0)
    print(f"t={time.time() - t0:5.1f}s search {prog['search']} exact {prog['exact']} {prog['where']} alive {gpu.counter('serve_alive')} "
          f"grid starts {gpu.counter('launches_serve')} served {gpu.counter('served_queries')} exact_calls {gpu.counter('exact_calls')} batches {gpu.counter('exact_batches')}", flush=True)
stop.set()
time.sleep(1.0)
print("done (daemon threads are abandoned if stuck)", flush=True)
os._exit(0)

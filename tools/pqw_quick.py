#!/usr/bin/env python3
"""Development check of the several-waves-per-query kernel (jv_kernels_pqw.hip) against the one-wave kernel on a
C3-shaped index: identical ids / score bits / counters / flags, and the throughput of both.
env: SIM (0 L2 | 1 dot | 2 cosine), N (docs, default 2M), B (queries per launch), RKS (comma list), DIST, STAMPS=1 (diagnostic build + phase shares),
JV_OPT_<name>=<int> per-index options applied to both runs, JV_LIB=<path> another build of libjvgpu.so."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as g
g.load_package()
b = importlib.import_module("opensearch_jvector_amd.binding")
stamps = os.environ.get("STAMPS", "0") == "1"
if stamps:
    b.LIB_PATH = os.path.join(os.path.dirname(b.LIB_PATH), "libjvgpu_stamps.so")
    b.load_library(b.LIB_PATH)
elif os.environ.get("JV_LIB"):  # an experimental build of the library (A/B runs)
    b.LIB_PATH = os.path.abspath(os.environ["JV_LIB"])
    b.load_library(b.LIB_PATH)
gb = importlib.import_module("opensearch_jvector_amd.builder_gpu")
import bench

n = int(os.environ.get("N", 2_000_000)); d = int(os.environ.get("D", 768)); M = int(os.environ.get("M", 32)); B = int(os.environ.get("B", 65536))
rks = [int(x) for x in os.environ.get("RKS", "160,1200").split(",")]
dev = torch.device("cuda", 0)
t0 = time.time()
base, q = bench.make_pq_data(torch, os.environ.get("DIST", "rotated"), n, B, d, M, 0, n, False, dev)
SIM = int(os.environ.get("SIM", "0"))   # 0 L2, 1 dot product, 2 cosine (round 6: cosine runs on the several-waves kernels)
adj, entry = gb.build_graph_gpu(torch, base, SIM, R=32, L=100, verbose=False)
pq = gb.pq_train_encode_gpu(torch, base, M, SIM)
desc, keep = b.make_desc_device(n, d, 32, base.data_ptr(), adj.data_ptr(), entry, SIM, pq_M=M, pq_K=pq["K"],
                                pq_codebooks=pq["codebooks"], pq_centroid=pq["centroid"], pq_codes_ptr=pq["codes"].data_ptr(),
                                borrow=True, extra_flags=b.DESC_FUSED_ADC)
ix = b.GpuIndex(desc=desc, keepalive=keep, flags=b.DESC_BORROW)
print(f"index ready in {time.time() - t0:.1f} s", flush=True)
for key, val in os.environ.items():
    if key.startswith("JV_OPT_"):
        ix.set_option(key[len("JV_OPT_"):].lower(), int(val))
dbg = torch.zeros(32, dtype=torch.int64, device=dev)  # [0, 16): the search kernel's phases; [16, 28): jv_visited_fast_kernel's


def run(no_pqw, rk, iters=3):
    ix.set_option("no_pqw", no_pqw)
    o = [torch.full((B, 10), -7, dtype=torch.int32, device=dev), torch.full((B, 10), -7, dtype=torch.int32, device=dev),
         torch.zeros((B, 10), dtype=torch.float32, device=dev), torch.zeros((B,), dtype=torch.int32, device=dev),
         torch.zeros((B, 4), dtype=torch.int32, device=dev), torch.zeros((B,), dtype=torch.int32, device=dev)]
    best = 1e9
    for it in range(iters):
        if stamps:
            dbg.zero_()
            ix.set_option("dbg_ptr", dbg.data_ptr())
        torch.cuda.synchronize(); t = time.time()
        ix.search_batch_device(q.data_ptr(), B, 10, rk, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(),
                               o[4].data_ptr(), o[5].data_ptr())
        torch.cuda.synchronize(); best = min(best, time.time() - t)
    if stamps:
        ix.set_option("dbg_ptr", 0)
    return [x.cpu().numpy() for x in o], best, dbg.cpu().numpy().astype(np.float64)


names = ["find best/runner-up + block select", "prefetch issue + mark expanded + log", "ADC (+ exchange)", "boundary test + rank search + dedupe",
         "ranks among new + shift + insert", "boundary + trim", "visited-count pass", "LUT build + entry point", "rerank + top-K"]
for rk in rks:
    old, t_old, v_old = run(1, rk)
    new, t_new, v_new = run(0, rk)
    eq = [np.array_equal(old[i].view(np.uint32) if old[i].dtype == np.float32 else old[i], new[i].view(np.uint32) if new[i].dtype == np.float32 else new[i]) for i in range(6)]
    st = new[4].astype(np.float64).mean(0)
    bad = int((old[0] != new[0]).any(1).sum())
    print(f"rk={rk}: one-wave {B / t_old:,.0f} QPS, several-waves {B / t_new:,.0f} QPS ({t_old / t_new:.2f}x); equal nodes/docs/scores/count/stats/flags = {eq}; "
          f"queries with different ids {bad}; expansions/query {st[2]:.1f} visited {st[0]:.1f}; flagged(new) {(new[5] != 0).sum()}", flush=True)
    if (new[5] != 0).any():
        why = (new[5].view(np.uint32) >> 8) & 0xFF
        print("  flagged rows by reason code:", {int(k): int((why[new[5] != 0] == k).sum()) for k in np.unique(why[new[5] != 0])}, flush=True)
    if not all(eq):
        i = int(np.argmax((old[0] != new[0]).any(1) | (old[4] != new[4]).any(1)))
        print("  first differing query", i, "\n   old", old[0][i], old[4][i], old[5][i], "\n   new", new[0][i], new[4][i], new[5][i])
    if stamps:
        for label, v in (("one-wave", v_old), ("several-waves (wave 0)", v_new)):
            cyc = np.concatenate([v[:8], v[13:14]])
            ne = max(st[2], 1) * B
            print(f"  {label}: cycles per expansion by phase")
            for i, nme in enumerate(names):
                print(f"     {nme:40s} {100 * cyc[i] / max(cyc.sum(), 1):5.1f} %   {cyc[i] / ne:8.0f}")
            if label != "one-wave" and v[16:28].sum() > 0:
                vv = v[16:28]
                steps = max(vv[7], 1)
                vn = ["close / clear / barriers", "rows arrive", "full-width rounds", "packing", "packed walk", "next steps + loads issued", "tail"]
                print("  jv_visited_fast_kernel: cycles per step and wave (steps per query: %.2f)" % (vv[7] / B / 3))
                for i, nme in enumerate(vn):
                    print(f"     {nme:40s} {100 * vv[i] / max(vv[:7].sum(), 1):5.1f} %   {vv[i] / steps:8.0f}")
                print(f"     full-width rounds / step {vv[8] / steps:.2f}, packed ids / step {vv[9] / steps:.1f}, packed-walk instructions / step {vv[10] / steps:.1f}")
            print(f"     raw slots / expansion: " + " ".join(f"[{i}]={v[i] / ne:.2f}" if i in (8, 10, 11, 12) else f"[{i}]={v[i] / ne:.0f}" for i in range(16)))
if os.environ.get("SINGLE", "0") == "1":   # one-query calls from one thread through the host API (the query server): p50 / p99
    qh = q[:400].cpu().numpy()
    for rk in rks:
        for i in range(50):
            ix.search(qh[i], 10, rk)
        lat = []
        for i in range(50, len(qh)):
            t = time.perf_counter(); ix.search(qh[i], 10, rk); lat.append((time.perf_counter() - t) * 1e3)
        print(f"one query per call, rk={rk}: p50 {np.percentile(lat, 50):.3f} ms  p99 {np.percentile(lat, 99):.3f} ms  mean {np.mean(lat):.3f} ms", flush=True)
ix.close()

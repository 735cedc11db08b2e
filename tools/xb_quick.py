"""Quick timing of the batched exact scorer (jv_score_ordinals_batch_device): N docs x d, B queries under one filter of
selectivity SEL.  N=2000000 D=768 B=256 SEL=0.1,0.01,0.001 python tools/xb_quick.py
Prints per selectivity: ms per batch, QPS, candidates, rows re-scored per query, bf16 TFLOP/s and GB/s of the tile pass."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as graft  # noqa: E402

graft.load_package()
import importlib  # noqa: E402

b = importlib.import_module("opensearch_jvector_amd.binding")

N = int(os.environ.get("N", 2_000_000))
D = int(os.environ.get("D", 768))
B = int(os.environ.get("B", 256))
K = int(os.environ.get("K", 10))
SIM = int(os.environ.get("SIM", 0))
SELS = [float(x) for x in os.environ.get("SEL", "0.1,0.01,0.001").split(",")]
REPS = int(os.environ.get("REPS", 10))
FLAGS = int(os.environ.get("FLAGS", 0))

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(42)
cen = torch.randn((4096, D), generator=g, device=dev)
base = torch.empty((N, D), dtype=torch.float32, device=dev)
for s in range(0, N, 1 << 18):
    e = min(N, s + (1 << 18))
    base[s:e] = cen[torch.randint(0, 4096, (e - s,), generator=g, device=dev)] + 0.35 * torch.randn((e - s, D), generator=g, device=dev)
q = cen[torch.randint(0, 4096, (B,), generator=g, device=dev)] + 0.35 * torch.randn((B, D), generator=g, device=dev)
adj = torch.full((N, 4), -1, dtype=torch.int32, device=dev)
desc, keep = b.make_desc_device(N, D, 4, base.data_ptr(), adj.data_ptr(), 0, SIM, borrow=True)
gpu = b.GpuIndex(desc=desc, keepalive=keep, flags=b.DESC_BORROW)
o_nodes = torch.empty((B, K), dtype=torch.int32, device=dev)
o_docs = torch.empty((B, K), dtype=torch.int32, device=dev)
o_scores = torch.empty((B, K), dtype=torch.float32, device=dev)
o_count = torch.empty((B,), dtype=torch.int32, device=dev)
st = torch.cuda.current_stream(dev)
rng = np.random.default_rng(1)
for sel in SELS:
    acc = np.nonzero(rng.random(N) < sel)[0].astype(np.int32)
    tl = torch.from_numpy(acc).to(dev)
    words = torch.from_numpy(b.accept_words(acc, N).view(np.int64)).to(dev)

    def call(info=False, filt=False):
        if filt:
            return gpu.score_ordinals_batch_device(q.data_ptr(), B, K, o_nodes.data_ptr(), o_docs.data_ptr(), o_scores.data_ptr(), o_count.data_ptr(),
                                                   d_accept=words.data_ptr(), accept_num_docs=N, stream=st.cuda_stream, want_info=info, flags=FLAGS)
        return gpu.score_ordinals_batch_device(q.data_ptr(), B, K, o_nodes.data_ptr(), o_docs.data_ptr(), o_scores.data_ptr(), o_count.data_ptr(),
                                               d_ordinals=tl.data_ptr(), count=len(acc), stream=st.cuda_stream, want_info=info, flags=FLAGS)
    t0 = time.time()
    info = call(True)
    first = time.time() - t0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / REPS
    t0 = time.time()
    for _ in range(REPS):
        call(filt=True)
    torch.cuda.synchronize()
    ms_f = (time.time() - t0) * 1e3 / REPS
    C = int(info[0])
    kp = (D + 63) // 64 * 64
    flop = 2.0 * B * (C + int(info[1])) * kp
    print(f"sel={sel}: C={C} sample={int(info[1])} rescored/query={info[2] / B:.0f} overflowed={int(info[3])} | first call {first * 1e3:.1f} ms | "
          f"list form {ms:.3f} ms/batch = {B / ms * 1e3:.0f} QPS | filter form (list built per call) {ms_f:.3f} ms = {B / ms_f * 1e3:.0f} QPS | "
          f"whole call: {flop / ms / 1e9:.1f} TFLOP/s bf16, {C * kp * 2 / ms / 1e6:.0f} GB/s mirror bytes", flush=True)
gpu.close()

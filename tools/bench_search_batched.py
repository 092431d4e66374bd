#!/usr/bin/env python3
"""The batched search (Q >= 32 needles: fp16 MFMA candidates + exact re-score) at cfg5's table: N = 1M x d = 100, top-50, Q = 1024 (and 48, 256).
Per-kernel HIP-event times of one gr_cosine_topk_dev call and the whole call; first rows checked against the five-needle path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gan-reverser_amd")); sys.path.insert(0, ROOT)
import numpy as np
import ganrev._lib as L
ctx = L.default_context()
N, d, k = 1_000_000, 100, 50
dev = ctx.malloc(4 * N * d)
ctx.fill_normal(dev, N * d, 42)
for Q in ([int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else (48, 256, 1024)):      # python tools/bench_search_batched.py [Q,Q,...]
    q = (np.arange(Q, dtype=np.int64) * 977 + 100) % N
    q[:5] = np.arange(1, 6) * 100
    r0 = ctx.search_reruns()
    idx, sc = ctx.cosine_topk(None, q, k, emb_dev=dev, n=N, d=d)
    ctx.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.cosine_topk(None, q, k, emb_dev=dev, n=N, d=d)
    dt = (time.perf_counter() - t0) / reps
    ctx.set_timing(2)
    for _ in range(reps):
        ctx.cosine_topk(None, q, k, emb_dev=dev, n=N, d=d)
    kt = ctx.kernel_times(); ctx.set_timing(0)
    i5, s5 = ctx.cosine_topk(None, q[:5], k, emb_dev=dev, n=N, d=d)
    same = bool(np.array_equal(i5, idx[:5]) and np.array_equal(s5, sc[:5]))
    ksum = sum(x['total_ms'] for x in kt) / reps
    print(f"Q={Q}: {dt*1e3:.3f} ms per call incl. host round trip, kernels {ksum:.3f} ms = {2.0*N*d*Q/ksum/1e9:.1f} TFLOP/s; reruns {ctx.search_reruns()-r0}; first 5 equal the five-needle path: {same}")
    print("   " + "; ".join(f"{x['kernel']}: {x['total_ms']/reps*1e3:.1f} us" for x in kt))
ctx.free(dev)

#!/usr/bin/env python3
"""apply_r.lua search at BASELINE.json configs[4] size: N = 1M embeddings x d = 100, top-50 (and the reference's own
N = 10k x 32, top-100), Q = 5 needles (apply_r.lua:267).  Times gr_cosine_topk_dev with the embeddings resident in HBM
and checks the indices bit-exactly against the CPU oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gan-reverser_amd")); sys.path.insert(0, ROOT)
import numpy as np
import ganrev._lib as L
from oracle import oracle
ctx = L.default_context()
for (N, d, k, Q) in [(10000, 32, 100, 5), (1000000, 100, 50, 5), (1000000, 100, 50, 8)]:
    dev = ctx.malloc(4 * N * d)
    ctx.fill_normal(dev, N * d, 42)
    emb = ctx.download(dev, (N, d))
    q = np.arange(1, Q + 1, dtype=np.int64) * 100
    idx, sc = ctx.cosine_topk(None, q, k, emb_dev=dev, n=N, d=d)          # warm-up + result
    ctx.synchronize()
    t0 = time.perf_counter(); reps = 10
    for _ in range(reps):
        idx, sc = ctx.cosine_topk(None, q, k, emb_dev=dev, n=N, d=d)
    dt = (time.perf_counter() - t0) / reps
    ctx.set_timing(2)
    for _ in range(reps):
        ctx.cosine_topk(None, q, k, emb_dev=dev, n=N, d=d)
    kt = ctx.kernel_times(); ctx.set_timing(0)
    print("   " + "; ".join(f"{x['kernel']}: {x['total_ms']/reps*1e3:.1f} us ({x['launches']//reps} launches, {x['bytes']/max(x['total_ms'],1e-9)/1e6:.0f} GB/s)" for x in kt))
    t1 = time.perf_counter()
    ridx, rsc = oracle.cosine_topk(emb, q, k)
    tc = time.perf_counter() - t1
    ok = np.array_equal(idx, ridx) and np.array_equal(sc, rsc)
    print(f"N={N} d={d} Q={Q} k={k}: GPU {dt*1e3:.3f} ms/search (incl. host round trip) = {N*d*4/dt/1e9:.1f} GB/s of embedding bytes; "
          f"oracle CPU {tc*1e3:.0f} ms; bit-exact indices+scores: {ok}")
    ctx.free(dev)

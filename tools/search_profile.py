"""Per-kernel times of one cfg5 search (1M x 100, 5 needles, top-50): python tools/search_profile.py [N] [d] [Q]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gan-reverser_amd"))
import ganrev._lib as L
ctx = L.default_context()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 100
Q = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ACCF = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
dev = ctx.malloc(4 * N * d)
ctx.fill_normal(dev, N * d, 7)
q = (np.arange(Q, dtype=np.int64) * 100) % N
for rep in range(3):
    ctx.cosine_topk(None, q, 50, accumulate_in_float=ACCF, emb_dev=dev, n=N, d=d)
ctx.synchronize()
t0 = time.perf_counter()
for rep in range(20):
    ctx.cosine_topk(None, q, 50, accumulate_in_float=ACCF, emb_dev=dev, n=N, d=d)
ctx.synchronize()
print("wall per search %.4f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
ctx.set_timing(2)
for rep in range(5):
    ctx.cosine_topk(None, q, 50, accumulate_in_float=ACCF, emb_dev=dev, n=N, d=d)
for k in sorted(ctx.kernel_times(), key=lambda k: -k["total_ms"]):
    print("%-28s x%-3d %.4f ms each" % (k["kernel"], k["launches"] / 5, k["total_ms"] / k["launches"]))
ctx.set_timing(0)

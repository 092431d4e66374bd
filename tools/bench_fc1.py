#!/usr/bin/env python3
"""Per-kernel HIP-event times of R.fc1's three GEMMs (nn.Linear(128 h w -> 512), models.lua:447: forward, gradInput, accGradParameters) at the cfg2 / cfg3 shapes,
training mode, f16x3:   python tools/bench_fc1.py        (ablation build: GR_GEMM_WGS / GR_GEMM_MIN_KLEN / GR_GEMM_SMALL_TILES select the split-K plan)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gan-reverser_amd"), ROOT]
import numpy as np
import ganrev._lib as L
from ganrev import nn, synth
ctx = L.default_context(); ctx.set_conv_mode("f16x3")
for B, K, N in ((256, 8192, 512), (512, 32768, 512)):
    m = nn.Sequential(); m.add(nn.Linear(K, N)); synth.init_params(m, 3); m.training()
    x = synth.normal((B, K), 5); gy = synth.normal((B, N), 6)
    m.forward(x); m.backward(x, gy)
    ctx.set_timing(2)
    for _ in range(5):
        m.forward(x); m.backward(x, gy)
    kt = ctx.kernel_times(); ctx.set_timing(0)
    rows = [(k["kernel"], k["launches"], k["total_ms"] / k["launches"]) for k in kt if "gemm" in k["kernel"]]
    tot = sum(k["total_ms"] for k in kt if "gemm" in k["kernel"]) / 5
    print(f"B={B} K={K} N={N}: gemm kernels {tot*1e3:.1f} us per fwd+bwd  |  " + "; ".join(f"{n} x{l} {t*1e3:.1f} us" for n, l, t in rows), flush=True)

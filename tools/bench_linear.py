#!/usr/bin/env python3
"""Timing of one nn.Linear forward in evaluate() mode at G.fc's shape family (batch x K -> N), with and without the BatchNorm + ReLU epilogue:
   python tools/bench_linear.py [B] [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gan-reverser_amd"), ROOT]
import numpy as np
import ganrev._lib as L
from ganrev import nn, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
ctx = L.default_context(); ctx.set_conv_mode("f16x3")
for K in (32, 100, 128, 256):
    for ep in (False, True):
        m = nn.Sequential()
        m.add(nn.Linear(K, N))
        if ep:
            m.add(nn.BatchNormalization(N)); m.add(nn.ReLU())
        synth.init_params(m, 3)
        m.evaluate()
        x = synth.normal((B, K), 5)
        m.forward(x)
        ctx.set_timing(2)
        for _ in range(3): m.forward(x)
        kt = ctx.kernel_times(); ctx.set_timing(0)
        rows = [(k["kernel"], k["launches"], k["total_ms"] / k["launches"]) for k in kt if k["kernel"].startswith("gemm")]
        print(f"B={B} K={K} N={N} epilogue={ep}: " + "; ".join(f"{n} x{l} {t*1e3:.1f} us" for n, l, t in rows), f"-> {B*N*4/1e6:.0f} MB out", flush=True)

#!/usr/bin/env python3
# NEEDS THE ABLATION BUILD: make -C gan-reverser_amd/csrc ablate, then GANREV_LIB=$PWD/gan-reverser_amd/ganrev/libganrev_ablate.so python tools/stagger_p16.py ...
# (the shipping library does not answer the gr_set_tuning keys / GR_* switches this script flips: they make kernels compute wrong results by design)
"""Sweep of the start delay of the second-dispatched workgroups in conv3x3_p16_quad_kernel (interleaved rounds, one process)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gan-reverser_amd"))
import ganrev._lib as L
ctx = L.default_context(); ctx.set_conv_mode("f16x3")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for name, cin, cout, h, w in [("R.conv2/3", 64, 64, 32, 32), ("R.conv5/6", 128, 128, 16, 16), ("R.conv4", 64, 128, 16, 16)]:
    vals = [0, 2, 4, 6, 8, 10, 14, 20]
    res = {v: [] for v in vals}
    for rnd in range(5):
        for v in vals:
            ctx.set_tuning("p16_stagger", v)
            res[v].append(ctx.bench_conv3(5, B, cin, cout, h, w, 10))
    print(name, " ".join(f"{v}:{statistics.median(res[v]) * 1e3:.1f}" for v in vals), "us (stagger x512 clocks : median, with statistics epilogue)")

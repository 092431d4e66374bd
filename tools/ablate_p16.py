#!/usr/bin/env python3
# NEEDS THE ABLATION BUILD: make -C gan-reverser_amd/csrc ablate, then GANREV_LIB=$PWD/gan-reverser_amd/ganrev/libganrev_ablate.so python tools/ablate_p16.py ...
# (the shipping library does not answer the gr_set_tuning keys / GR_* switches this script flips: they make kernels compute wrong results by design)
"""Ablation of conv3x3_p16_wide_kernel (cdna_hip_programming.md section 7, diagnostic loop step 2): the same launch with parts
switched off (gr_set_tuning "p16_debug": 1 no output stores, 2 no statistics, 4 no DMA, 8 no MFMA), interleaved rounds in one
process, median of the rounds.  Outputs are wrong by design when anything is off: only the times matter."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gan-reverser_amd"))
import ganrev._lib as L
ctx = L.default_context(); ctx.set_conv_mode("f16x3")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
shapes = [("R.conv2/3", 64, 64, 32, 32), ("R.conv5/6", 128, 128, 16, 16)]
variants = [("full+stats", 5, 0), ("full", 4, 0), ("no stores", 4, 1), ("no DMA", 4, 4), ("no MFMA", 4, 8),
            ("only MFMA+LDS reads", 4, 5), ("only DMA", 4, 9), ("only stores", 4, 12), ("nothing", 4, 13), ("old split_wide kernel", 0, 0)]
variants = [(f"{n} [v{v}]", w, d, v) for v in (1, 0) for (n, w, d) in variants if not (v == 0 and w == 0)]
for name, cin, cout, h, w in shapes:
    res = {v[0]: [] for v in variants}
    for rnd in range(5):
        for vname, which, dbg, var in variants:
            ctx.set_tuning("p16_variant", var)
            ctx.set_tuning("p16_debug", dbg)
            res[vname].append(ctx.bench_conv3(which, B, cin, cout, h, w, 10))
    ctx.set_tuning("p16_debug", 0)
    print(f"{name} B={B} {cin}->{cout} @{h}x{w}")
    for vname, _, _, _ in variants:
        print(f"   {vname:34s} {statistics.median(res[vname]) * 1e3:8.1f} us   (min {min(res[vname]) * 1e3:.1f})")

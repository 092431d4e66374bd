#!/usr/bin/env python3
# NEEDS THE ABLATION BUILD: make -C gan-reverser_amd/csrc ablate, then GANREV_LIB=$PWD/gan-reverser_amd/ganrev/libganrev_ablate.so python tools/ablate_up2.py ...
# (the shipping library does not answer the gr_set_tuning keys / GR_* switches this script flips: they make kernels compute wrong results by design)
"""Ablation of the up-sampling convolution kernels (cdna_hip_programming.md section 7, diagnostic loop step 2): the same launch with
parts switched off (gr_set_tuning "up2_debug": 1 no output stores, 2 no MFMA, 4 no activation staging, 8 no weight DMA) and the
eight-wave kernel beside the four-wave one, interleaved rounds in one process, median of the rounds.  Outputs are wrong by design
when anything is off: only the times matter."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gan-reverser_amd"))
import ganrev._lib as L
ctx = L.default_context(); ctx.set_conv_mode("f16x3")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
shapes = [("G2.convB", 256, 128, 32, 32)] if B <= 256 else [("G.convA", 512, 256, 32, 32), ("G.convB", 256, 128, 64, 64)]
STAG = [int(v) for v in os.environ.get("STAG", "0").split(",")]
variants = [("eight-wave kernel", 0, 0), ("four-wave full", 2, 0), ("no stores", 2, 1), ("no MFMA", 2, 2), ("no act staging", 2, 4), ("no weight DMA", 2, 8),
            ("only MFMA + LDS reads", 2, 13), ("only staging (act + weights)", 2, 3), ("only stores", 2, 14), ("nothing", 2, 15)]
for name, cin, cout, h, w in shapes:
    vs = variants if STAG == [0] else [(f"four-wave, stagger {d}", 2, 0, d) for d in STAG]
    vs = [v if len(v) == 4 else v + (0,) for v in vs]
    res = {v[0]: [] for v in vs}
    for rnd in range(5):
        for vname, quad, dbg, stag in vs:
            ctx.set_tuning("up2_quad", quad); ctx.set_tuning("up2_debug", dbg); ctx.set_tuning("up2_stagger", stag)
            res[vname].append(ctx.bench_conv3(3, B, cin, cout, h, w, 10))
    ctx.set_tuning("up2_debug", 0); ctx.set_tuning("up2_quad", 1); ctx.set_tuning("up2_stagger", 0)
    variants_ = vs
    print(f"{name} B={B} {cin}->{cout} @{h}x{w} (output plane)")
    for vname, _, _, _ in variants_:
        print(f"   {vname:34s} {statistics.median(res[vname]) * 1e3:8.1f} us   (min {min(res[vname]) * 1e3:.1f})")

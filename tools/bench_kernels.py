#!/usr/bin/env python3
"""Micro-benchmark of the conv kernels at the cfg2 / cfg3 layer shapes (HIP events inside gr_bench_conv3)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gan-reverser_amd"))
import ganrev._lib as L
ctx = L.default_context()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
which_names = {0: "fwd", 1: "dgrad", 2: "wgrad", 3: "up2fwd", 4: "p16fwd", 5: "p16fwd+stats"}
shapes = [("R.conv2/3", 64, 64, 32, 32), ("R.conv4", 64, 128, 16, 16), ("R.conv5/6", 128, 128, 16, 16),
          ("G.convA", 512, 256, 16, 16), ("G.convB", 256, 128, 32, 32), ("R.conv1", 1, 64, 32, 32), ("G.convC", 128, 1, 32, 32),
          ("G2.convA", 128, 256, 16, 16), ("G2.convB", 256, 128, 32, 32),      # G2: G at cfg2 (gray 32x32)
          ("G3.convC", 128, 3, 64, 64)]                                        # G's last convolution at cfg3 (64x64 RGB; run with B = 512)
sel = sys.argv[2].split(",") if len(sys.argv) > 2 else None
for name, cin, cout, h, w in shapes:
    if sel and not any(s in name for s in sel):
        continue
    for which in (0, 1, 2, 3, 4, 5):
        if name.startswith("G") and which not in (0, 3):
            continue
        if which in (4, 5) and (ctx.conv_mode() != "f16x3" or cin % 16 or cout % 64):
            continue
        if which == 3 and (not name.split(".")[1].startswith("conv") or name.endswith("convC") or ctx.conv_mode() != "f16x3"):
            continue
        ms = ctx.bench_conv3(which, B, cin, cout, h, w, 10)
        fl = 2.0 * B * h * w * cin * cout * 9
        print(f"{name:10s} {which_names[which]:6s} B={B} {cin:4d}->{cout:4d} @{h}x{w}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s  ({fl/ms/1e9/157.3*100:5.1f}% of fp32 MFMA peak)")

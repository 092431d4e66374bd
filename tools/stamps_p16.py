#!/usr/bin/env python3
# NEEDS THE ABLATION BUILD: make -C gan-reverser_amd/csrc ablate, then GANREV_LIB=$PWD/gan-reverser_amd/ganrev/libganrev_ablate.so python tools/stamps_p16.py ...
# (the shipping library does not answer the gr_set_tuning keys / GR_* switches this script flips: they make kernels compute wrong results by design)
"""In-kernel time stamps of conv3x3_p16_quad_kernel (cdna_hip_programming.md section 7): per workgroup, wave 0 records
s_memtime at the start, after every 'chunk landed' barrier, after every 'chunk consumed' barrier and at the end, plus HW_ID.
Prints, for a few CUs, the timeline of the workgroups that ran there."""
import os, sys, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gan-reverser_amd"))
import ganrev._lib as L
ctx = L.default_context(); ctx.set_conv_mode("f16x3")
B, cin, cout, h, w = 256, 64, 64, 32, 32
if len(sys.argv) > 1 and sys.argv[1] == "conv5": cin, cout, h, w = 128, 128, 16, 16
ntiles = 4096
buf = ctx.malloc(ntiles * 32 * 8)
ctx.check(ctx.lib.gr_debug_stamps(ctx.h, L._ptr(buf)), "stamps")
ctx.bench_conv3(4, B, cin, cout, h, w, 3)
ctx.upload(np.zeros(ntiles * 32, np.uint64), buf)
ctx.set_tuning("p16_debug", 32)
ms = ctx.bench_conv3(4, B, cin, cout, h, w, 1)      # 3 warm-up launches + 1 timed: the buffer holds the last launch
ctx.set_tuning("p16_debug", 0)
st = ctx.download(buf, (ntiles, 32), np.uint64)
st = st[st[:, 31] > 0]
print(f"{len(st)} workgroups stamped; launch {ms * 1e3:.1f} us (with stamps)")
hw = st[:, 0].astype(np.int64)
wave_slot, simd, cu, sh, se, xcc = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, (hw >> 32) & 15
t0 = st[:, 2].min()
nn = st[:, 31].astype(np.int64)
rt0, rt1 = st[:, 1].astype(np.int64), st[np.arange(len(st)), nn - 1].astype(np.int64)
mt0, mt1 = st[:, 2].astype(np.int64), st[np.arange(len(st)), nn - 2].astype(np.int64)
clk = float(np.median((mt1 - mt0) / np.maximum(1, rt1 - rt0))) * 100.0       # s_memrealtime ticks at 100 MHz
print(f"shader clock ~ {clk:.0f} MHz (median over workgroups); stamps per workgroup {int(nn[0]) - 3}")
np.save(os.path.join(ROOT, "gpurun_out", "stamps.npy"), st)
groups = collections.defaultdict(list)
for i in range(len(st)):
    groups[(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]))].append(i)
print("distinct (xcc, se, sh, cu):", len(groups), " workgroups per CU:", collections.Counter(len(v) for v in groups.values()))
for key in list(groups)[:4]:
    print("CU", key)
    for i in groups[key]:
        n = int(st[i, 31]) - 1
        ts = (st[i, 2:n].astype(np.int64) - int(t0)) / clk       # us: start, then (landed, consumed) per chunk, end
        print(f"   slot {int(wave_slot[i])} simd {int(simd[i])}: " + " ".join(f"{t:.1f}" for t in ts))

# aggregate: mean duration of each phase over all workgroups
n = int(nn[0]) - 1
T = (st[:, 2:n].astype(np.int64) - st[:, 2:3].astype(np.int64)) / clk
d = np.diff(T, axis=1)
names = []
for c in range((n - 4) // 2): names += [f"dma{c}", f"mma{c}"]
names += ["epilogue"]
print("mean phase durations (us):", " ".join(f"{nm}={v:.2f}" for nm, v in zip(names, d.mean(0))))
print("start spread (us): min %.2f max %.2f ; end: min %.2f max %.2f" % (((st[:, 2].astype(np.int64) - int(t0)) / clk).min(), ((st[:, 2].astype(np.int64) - int(t0)) / clk).max(), T[:, -1].min() , (T[:, -1] + (st[:, 2].astype(np.int64) - int(t0)) / clk).max()))

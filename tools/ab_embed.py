#!/usr/bin/env python3
"""A/B of the evaluate()-mode hand-over (gr_set_tuning "eval_p16"): cfg5's G -> R pipeline over `rows` images, interleaved on one box.
   python tools/ab_embed.py [rows] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gan-reverser_amd"), ROOT]
import ganrev._lib as L
from ganrev import models, nn_utils, synth
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 51200
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dims, nd, batch = (3, 64, 64), 100, 512
ctx = L.default_context(); ctx.set_conv_mode("f16x3")
G = models.create_G(dims, nd); synth.init_params(G, 1)
R = models.create_R(dims, nd); synth.init_params(R, 2)
G._ctx = R._ctx = ctx
G.evaluate(); R.evaluate()
gnet, rnet = G.device_net((nd,)), R.device_net(dims)
noise = nn_utils.createNoiseInputsDev(ctx, rows, nd, "normal", seed=4242)
table = nn_utils.DeviceTensor(ctx, (rows, nd))
tabs = {}
for rep in range(reps):
    for mode in (0, 1):
        ctx.set_tuning("eval_p16", mode)
        L.embed_dev(gnet, [rnet], noise.ptr, 2 * batch, batch, [table.ptr]); ctx.synchronize()
        ctx.event_record(100); L.embed_dev(gnet, [rnet], noise.ptr, rows, batch, [table.ptr]); ctx.event_record(101)
        ms = ctx.event_elapsed_ms(100, 101)
        tabs[mode] = table.numpy().copy()
        print(f"rep {rep} eval_p16={mode}: {rows / ms * 1e3:9.0f} img/s  {ms / (rows / batch):.3f} ms/chunk", flush=True)
import numpy as np
d = np.abs(tabs[0] - tabs[1]).max(); print("max |table(p16) - table(fp32)| =", d, " max|table| =", np.abs(tabs[0]).max())
for mode in (0, 1):
    ctx.set_tuning("eval_p16", mode)
    ctx.set_timing(2); L.embed_dev(gnet, [rnet], noise.ptr, 4 * batch, batch, [table.ptr]); ctx.synchronize()
    kt = ctx.kernel_times(); ctx.set_timing(0)
    agg = {}
    for k in kt:
        if k["kernel"].startswith(("timer_", "range_")): continue
        a = agg.setdefault(k["kernel"], [0, 0.0]); a[0] += k["launches"]; a[1] += k["total_ms"]
    print(f"--- eval_p16={mode}: per chunk")
    for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"  {name:52s} x{n / 4:4.1f} {t / 4:.4f} ms")

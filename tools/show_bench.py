#!/usr/bin/env python3
"""Print the headline numbers and the per-kernel table of a bench.py JSON line (file argument)."""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(d["value"], "img/s", d["ms_per_step"], "ms | dominant", d["roofline"]["kernel"], "frac", d["roofline"]["frac"], "| r_convs", d.get("r_convs"), "| elementwise", d.get("elementwise"))
for m, r in d.get("modes", {}).items():
    print("  mode", m, r["images_per_sec"], "img/s", r["ms_per_step"], "ms", r["roofline"]["kernel"], r["roofline"]["frac"], "r_convs frac fp32 peak", r["r_convs"]["frac_of_fp32_mfma_peak"])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for k, v in list(d["kernels"].items())[:n]:
    print(f"{k:50s} {v['ms_per_step']:.4f} x{v['launches_per_step']} tf={v['tflops']} gbs={v['gbs']}")
for key in ("search_cfg5", "cpu_baseline"):
    if d.get(key): print(key, d[key])

"""Compact view of one or two bench.py lines (diagnostic): python tools/bench_diff.py new.json [old.json]"""
import json, sys
def load(p):
    return json.loads(open(p).read().strip().splitlines()[-1])
new = load(sys.argv[1]); old = load(sys.argv[2]) if len(sys.argv) > 2 else None
def head(d, tag):
    print(f"{tag}: {d['value']:.0f} img/s  {d['ms_per_step']:.4f} ms  roofline {d['roofline']['kernel']} {d['roofline']['frac']}  r_convs {d['r_convs']['ms_per_step']}  "
          f"elementwise {d['elementwise']['ms_per_step']}  kernel_ms {d['kernel_ms_per_step']}" + (f"  | cfg3 {d['cfg3']['images_per_sec']:.0f} img/s {d['cfg3']['ms_per_step']:.3f} ms ew {d['cfg3']['elementwise']['ms_per_step']} rconv {d['cfg3']['r_convs']['ms_per_step']}" if 'cfg3' in d else ""))
head(new, "new")
if old: head(old, "old")
for sect in ("", "cfg3"):
    kn = (new[sect] if sect else new)["kernels"] if (sect == "" or sect in new) else {}
    ko = ((old[sect] if sect else old)["kernels"] if old and (sect == "" or sect in old) else {})
    print(f"--- kernels {sect or 'cfg2'} (ms/step new | old)")
    for k in sorted(set(kn) | set(ko), key=lambda k: -(kn.get(k, {}).get("ms_per_step") or 0)):
        a, b = kn.get(k, {}).get("ms_per_step"), ko.get(k, {}).get("ms_per_step")
        if (a or 0) < 0.004 and (b or 0) < 0.004: continue
        print(f"  {k[:52]:52s} {a if a is not None else '-':>8} | {b if b is not None else '-':>8}   {kn.get(k, {}).get('gbs') or ''}")

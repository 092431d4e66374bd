"""Group a rocprofv3 kernel_trace csv by (kernel, grid) and print count / mean duration.  Usage: trace_by_grid.py DIR [substring]"""
import csv, glob, sys, collections
d, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            key = (r["Kernel_Name"][:70], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
            acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)[: max(1, len(v) * 3 // 4)]
    print(f"{k[0]:70s} grid=({k[1]},{k[2]},{k[3]}) n={len(v):4d} mean={sum(v)/len(v):8.1f} us  trimmed={sum(v2)/len(v2):8.1f} us")

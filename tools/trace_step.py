"""One steady-state step out of a rocprofv3 kernel_trace.csv of bench.py: kernels in order with durations and the idle gap in
front of each.  usage: trace_step.py <kernel_trace.csv> [step index from the end, default 2]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
# a step starts at each fill_normal_kernel (the trainer's new noise)
starts = [i for i, r in enumerate(rows) if "fill_normal_kernel" in r["Kernel_Name"]]
lo, hi = starts[-back - 1], starts[-back]
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"]); prev_end = t0
busy = idle = 0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = max(0, s - prev_end); idle += gap; busy += e - s
    name = r["Kernel_Name"].replace("void ", "").replace("gr::", "").split("(")[0]
    print("%9.1f us  +%6.1f gap  %7.1f us  %s" % ((s - t0) / 1e3, gap / 1e3, (e - s) / 1e3, name[:90]))
    prev_end = max(prev_end, e)
print("kernels %d  busy %.1f us  idle %.1f us  span %.1f us" % (len(step), busy / 1e3, idle / 1e3, (prev_end - t0) / 1e3))

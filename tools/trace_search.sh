#!/bin/bash
# rocprofv3 kernel trace of the cfg5 search (tools/search_profile.py): per-kernel durations and the gaps between the three launches of one call
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rm -rf $R/gpurun_out/trace_search
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_search -- python3 $R/tools/search_profile.py > $R/gpurun_out/trace_search.log 2>&1
python3 - <<PY
import csv, glob
f = sorted(glob.glob("$R/gpurun_out/trace_search/**/*_kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if any(k in r["Kernel_Name"] for k in ("cos_approx", "batched_select", "search_select"))]
last = rows[-9:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    n = r["Kernel_Name"].replace("void gr::", "").split("(")[0][:60]
    print(f"{n:62s} start {int(r['Start_Timestamp']) - t0:8d} ns  dur {int(r['End_Timestamp']) - int(r['Start_Timestamp']):7d} ns")
PY

#!/bin/bash
# on the GPU box: tools/probe/run_fetch_probe.sh  ->  gpurun_out/fetch_calibration.txt
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rm -rf $R/gpurun_out/fetchcal
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/fetchcal -- $R/tools/probe/fetch_probe > $R/gpurun_out/fetchcal.log 2>&1
python3 - <<PY > $R/gpurun_out/fetch_calibration.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for f in glob.glob("$R/gpurun_out/fetchcal/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "FETCH_SIZE": continue
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        a = agg[k]; a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
true = float(1 << 30)
print("# tools/probe/fetch_probe under rocprofv3 --pmc FETCH_SIZE: every kernel reads a 1 GiB buffer exactly once (true bytes 1073741824)")
print("# factor = true bytes / (FETCH_SIZE x 1024): what the counter must be multiplied by for that load shape")
print(f"{'kernel':24s} {'launches':>8s} {'FETCH_SIZE KiB':>16s} {'factor':>8s} {'GB/s':>8s}")
for k in sorted(agg):
    n, v, t = agg[k]
    if n == 0 or v == 0: continue
    print(f"{k:24s} {n:8d} {v / n:16.0f} {true / (v / n * 1024):8.3f} {true / (t / n):8.1f}")
PY
cat $R/gpurun_out/fetch_calibration.txt

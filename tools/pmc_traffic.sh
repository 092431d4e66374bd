#!/bin/bash
# HBM traffic per kernel of the real step, two ways (run on the GPU box): usage tools/pmc_traffic.sh [cfg2|cfg3] [kernel-substring]
#  (1) the guide's way (MI355X_MICROARCH.md, HBM): FETCH_SIZE / WRITE_SIZE in separate passes, KiB, FETCH doubled on gfx950;
#  (2) by request size: TCC_EA0_RDREQ_{32B,64B,128B}_sum and TCC_EA0_WRREQ_{sum,64B_sum}: bytes = 32 a + 64 b + 128 c, no correction factor.
# tools/probe/fetch_probe calibrates (1) against known byte counts: the factor 2 holds for every vector load / LDS-DMA shape whose contiguous
# piece is a whole 128-byte line; 64-byte pieces read 1.36-1.55, scalar-cache loads 1.0 (profiles/r04_fetch_calibration.txt).
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
WL=${1:-cfg2}; F=${2:-}
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1)); rm -rf $R/gpurun_out/pmct_$i
  timeout -k 10 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmct_$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-search --no-gan --no-sustained --workload $WL --modes f16x3 --traffic off > $R/gpurun_out/pmct_$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("$R/gpurun_out/pmct_*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("gr::", "").split("(")[0]
        if "$F" and "$F" not in k: continue
        a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
print("# tools/pmc_traffic.sh $WL: per-launch averages, MB.  guide = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes); by-size = 32/64/128-byte EA requests counted separately")
print(f"{'kernel':48s} {'launches':>8s} {'2xFETCH':>9s} {'rd by size':>10s} {'(128B share)':>12s} {'WRITE':>8s} {'wr by size':>10s}")
for k in sorted(agg, key=lambda k: -agg[k].get("FETCH_SIZE", [0, 0])[1]):
    g = lambda c: agg[k][c][1] / agg[k][c][0] if c in agg[k] and agg[k][c][0] else 0.0
    n = agg[k]["FETCH_SIZE"][0] if "FETCH_SIZE" in agg[k] else 0
    rd = 32 * g("TCC_EA0_RDREQ_32B_sum") + 64 * g("TCC_EA0_RDREQ_64B_sum") + 128 * g("TCC_EA0_RDREQ_128B_sum")
    wr = 64 * g("TCC_EA0_WRREQ_64B_sum") + 32 * (g("TCC_EA0_WRREQ_sum") - g("TCC_EA0_WRREQ_64B_sum"))
    if 2 * 1024 * g("FETCH_SIZE") + 1024 * g("WRITE_SIZE") < 1e6: continue
    print(f"{k:48s} {n:8d} {2 * 1024 * g('FETCH_SIZE') / 1e6:9.1f} {rd / 1e6:10.1f} {128 * g('TCC_EA0_RDREQ_128B_sum') / max(rd, 1):12.2f} {1024 * g('WRITE_SIZE') / 1e6:8.1f} {wr / 1e6:10.1f}")
PY

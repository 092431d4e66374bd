#!/usr/bin/env python3
"""Turn the rocprofv3 outputs a gpurun call left under gpurun_out/ into the committed summaries under profiles/.

  gpurun_out/prof_stats_<workload>/**/_kernel_stats.csv       (rocprofv3 --kernel-trace --stats)   -> profiles/<tag>_<workload>_kernel_stats.csv
  gpurun_out/prof_fetch_<workload>|prof_write_<workload>/**/_counter_collection.csv (--pmc FETCH_SIZE / WRITE_SIZE, separate passes)
                                                                                       -> profiles/<tag>_hbm_traffic.json
                                                                                       -> profiles/traffic.json (read by bench.py)
HBM bytes per launch follow MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half the bytes of a wide coalesced streaming read, so the corrected figure doubles it.  Round 4 calibrated the factor per load
shape (tools/probe/fetch_probe.hip: exactly 2.0 for every vector-load and LDS-DMA shape that reads whole 128-byte lines, 1.0 for
scalar-cache loads) and cross-checked it per kernel against the request-size counters (tools/pmc_traffic.sh): both raw and corrected are kept.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    n = name.replace("void ", "").replace("gr::", "")
    return n.split("(")[0]


def pmc_avg(d, counter):
    files = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]
    if not files:
        return {}
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != counter:
            continue
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return {k: (n, v / n) for k, (n, v) in agg.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    workload = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    stats = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_stats_{workload}", "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if stats:
        shutil.copy(stats[-1], os.path.join(out, f"{tag}_{workload}_kernel_stats.csv"))
    fetch, write = pmc_avg(f"prof_fetch_{workload}", "FETCH_SIZE"), pmc_avg(f"prof_write_{workload}", "WRITE_SIZE")
    summary = {}
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, (0, 0.0))[1] * 1024
        w = write.get(k, (0, 0.0))[1] * 1024
        summary[k] = dict(fetch_bytes_raw=round(f), write_bytes=round(w), hbm_bytes_raw=round(f + w),
                          hbm_bytes_corrected=round(2 * f + w), launches_sampled=fetch.get(k, write.get(k))[0])
    json.dump(dict(workload=workload, note=__doc__.strip().split("\n\n")[-1] if False else
                   "per-launch averages; corrected = 2*FETCH_SIZE + WRITE_SIZE (KiB -> bytes), see tools/collect_profiles.py",
                   kernels=summary), open(os.path.join(out, f"{tag}_{workload}_hbm_traffic.json"), "w"), indent=1)
    tpath = os.path.join(out, "traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    traffic[workload] = {k: v["hbm_bytes_corrected"] for k, v in summary.items()}
    json.dump(traffic, open(tpath, "w"), indent=1)
    p = os.path.join(ROOT, "gpurun_out", f"bench_prof_{workload}.json")      # the FULL result of the profiled run (bench.py --detail)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(out, f"{tag}_{workload}_bench_under_rocprof.json"))
    print("wrote", sorted(os.listdir(out)))


if __name__ == "__main__":
    main()

#!/bin/bash
# PMC passes over the conv micro-benchmark (run on the GPU box): usage tools/pmc_conv.sh <shape-filter> <batch>
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
F=${1:-conv2}; B=${2:-256}
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $set | cut -d' ' -f1)
  GR_CONV_MODE=${GR_CONV_MODE:-f32} timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/tools/bench_kernels.py $B $F > $R/gpurun_out/pmc_$tag.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0,0.0]))
dur = collections.defaultdict(lambda: [0,0.0])
for f in glob.glob("$R/gpurun_out/pmc_*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ","").replace("gr::","").split("(")[0]
        if "conv3x3" not in k: continue
        a = agg[k][r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
        d = dur[k]; d[0]+=1; d[1]+= (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
print("# derived: clock = GRBM_GUI_ACTIVE / 8 XCDs / time; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x time x clock);")
print("#          waiting = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES; VALU / MFMA = SQ_INSTS_VALU / SQ_INSTS_MFMA (wave instructions)")
for k in agg:
    g = lambda c: agg[k][c][1] / agg[k][c][0] if c in agg[k] else float("nan")
    t = dur[k][1] / dur[k][0] * 1e-9
    clock = g("GRBM_GUI_ACTIVE") / 8 / t
    print(f"{k:50s} avg {t*1e6:7.1f} us  clock {clock/1e9:4.2f} GHz  MFMA busy {100*g('SQ_VALU_MFMA_BUSY_CYCLES')/(1024*t*clock):5.1f}% of SIMD cycles  "
          f"waiting {100*g('SQ_WAIT_INST_ANY')/g('SQ_WAVE_CYCLES'):5.1f}% of wave cycles  VALU/MFMA {g('SQ_INSTS_VALU')/max(g('SQ_INSTS_MFMA'),1):5.2f}  "
          f"LDS bank-conflict cycles {100*g('SQ_LDS_BANK_CONFLICT')/max(g('SQ_LDS_IDX_ACTIVE'),1):4.1f}% of LDS cycles")
print()
print("# raw per-dispatch averages")
for k in agg:
    print(k, "avg_us", round(dur[k][1]/dur[k][0]/1e3,1))
    for c,(n,v) in sorted(agg[k].items()):
        print(f"   {c:28s} {v/n:16.1f}")
PY

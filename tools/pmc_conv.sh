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
for k in agg:
    print(k, "avg_us", round(dur[k][1]/dur[k][0]/1e3,1))
    for c,(n,v) in sorted(agg[k].items()):
        print(f"   {c:28s} {v/n:16.1f}")
PY

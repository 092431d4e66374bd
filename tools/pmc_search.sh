#!/bin/bash
# PMC passes over tools/bench_search_batched.py (run on the GPU box): per-kernel counters of the batched search (48, 256, 1024 needles mixed in the averages).  usage: tools/pmc_search.sh [kernel-substring] [Q,Q,...]  (the same counter passes over tools/bench_search_batched.py)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
F=${1:-conv3x3}; WL=${2:-cfg2}
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmcs_$tag
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmcs_$tag -- python3 $R/tools/bench_search_batched.py ${2:-48,256,1024} > $R/gpurun_out/pmcs_$tag.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0,0.0]))
dur = collections.defaultdict(lambda: [0,0.0])
for f in glob.glob("$R/gpurun_out/pmcs_*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ","").replace("gr::","").split("(")[0]
        if "$F" not in k: continue
        a = agg[k][r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
        d = dur[k]; d[0]+=1; d[1]+= (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
print("# rocprofv3 --pmc passes over bench.py (tools/pmc_step.sh): per-dispatch averages inside the real $WL step")
print("# clock = GRBM_GUI_ACTIVE / 8 XCDs / time is printed only for dispatches of 0.3 ms or more (MI355X_MICROARCH.md, DVFS give-back: the quotient reads high on")
print("# shorter ones - round 3's files showed 2.4-5.6 GHz there); for shorter dispatches MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x time x clock) is given as the")
print("# range between the clocks these kernels were measured to hold in-kernel (1.9 GHz) and the chip maximum (2.4 GHz); waiting = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES")
for k in sorted(agg):
    g = lambda c: agg[k][c][1] / agg[k][c][0] if c in agg[k] and agg[k][c][0] else float("nan")
    t = dur[k][1] / dur[k][0] * 1e-9
    clock = g("GRBM_GUI_ACTIVE") / 8 / t
    mf = g('SQ_VALU_MFMA_BUSY_CYCLES')
    if t >= 0.3e-3 and clock <= 2.45e9:
        cs, bs = f"clock {clock/1e9:4.2f} GHz", f"MFMA busy {100*mf/(1024*t*clock):5.1f}%"
    else:
        cs, bs = "clock  n/a     ", f"MFMA busy {100*mf/(1024*t*2.4e9):4.1f}-{100*mf/(1024*t*1.9e9):4.1f}%"
    print(f"{k:46s} avg {t*1e6:7.1f} us  {cs}  {bs}  waiting {100*g('SQ_WAIT_INST_ANY')/g('SQ_WAVE_CYCLES'):5.1f}%  "
          f"wait-LDS {100*g('SQ_WAIT_INST_LDS')/g('SQ_WAVE_CYCLES'):5.1f}%  VALU/MFMA {g('SQ_INSTS_VALU')/max(g('SQ_INSTS_MFMA'),1):5.2f}  LDS/MFMA {g('SQ_INSTS_LDS')/max(g('SQ_INSTS_MFMA'),1):5.2f}  "
          f"bank-conflict {100*g('SQ_LDS_BANK_CONFLICT')/max(g('SQ_LDS_IDX_ACTIVE'),1):5.1f}% of LDS cycles  LDS active {100*g('SQ_ACTIVE_INST_LDS')/max(g('SQ_BUSY_CYCLES'),1):5.1f}%  "
          f"wave instructions per dispatch: VALU {g('SQ_INSTS_VALU'):.3g} SALU {g('SQ_INSTS_SALU'):.3g} VMEM-read {g('SQ_INSTS_VMEM_RD'):.3g}")
PY

#!/bin/bash
# rocprofv3 kernel trace of bench.py (f16x3 only) and one steady-state step listed with its idle gaps (tools/trace_step.py): tools/trace_step.sh [cfg2|cfg3]
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; WL=${1:-cfg2}; cd /tmp
rm -rf $R/gpurun_out/trace_step_$WL
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_step_$WL -- python3 $R/bench.py --workload $WL --modes f16x3 --steps 10 --warmup 3 --no-cpu-baseline --no-search --no-gan --no-sustained --traffic off > $R/gpurun_out/trace_step_$WL.log 2>&1
F=$(ls $R/gpurun_out/trace_step_$WL/*/*_kernel_trace.csv | tail -1)
python3 $R/tools/trace_step.py $F 6 > $R/gpurun_out/trace_step_$WL.txt
tail -3 $R/gpurun_out/trace_step_$WL.txt

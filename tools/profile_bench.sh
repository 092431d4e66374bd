#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 passes over bench.py, outputs under gpurun_out/prof_*_<workload> for tools/collect_profiles.py
# (run HERE afterwards: gpurun merges gpurun_out/ back, not profiles/).
#   tools/profile_bench.sh [workload] [extra bench.py args]
# (the --pmc passes run a 100 k-row cfg5 corpus: with the full 1 M rows a pass issues > 65 k dispatches and rocprofv3's counter collection
# segfaulted inside a kernel launch - round 4, after the evaluate()-mode hand-over added four small launches per chunk)
# Pass 1: --kernel-trace --stats (per-kernel durations).  Passes 2, 3: --pmc FETCH_SIZE / WRITE_SIZE, each on its own
# (MI355X_MICROARCH.md, HBM section), with --kernel-trace only.  --traffic off: bench.py must not start its own nested rocprofv3
# run from inside a profiled process (under --pmc the preloaded profiler library has initialised the GPU, and the launcher hop
# env -> python3 of the nested run is then an exec from a GPU-initialised process, which the box refuses).
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; WL=${1:-cfg2}; shift
cd /tmp
rm -rf $R/gpurun_out/prof_stats_$WL $R/gpurun_out/prof_fetch_$WL $R/gpurun_out/prof_write_$WL
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats_$WL -- python3 $R/bench.py --workload $WL --steps 10 --warmup 3 --no-cpu-baseline --no-sustained --traffic off "$@" > $R/gpurun_out/bench_prof_$WL.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_fetch_$WL -- python3 $R/bench.py --workload $WL --steps 3 --warmup 2 --no-cpu-baseline --no-sustained --traffic off --embed-rows 100000 --embed-train-steps 50 "$@" > $R/gpurun_out/bench_fetch.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_write_$WL -- python3 $R/bench.py --workload $WL --steps 3 --warmup 2 --no-cpu-baseline --no-sustained --traffic off --embed-rows 100000 --embed-train-steps 50 "$@" > $R/gpurun_out/bench_write.log 2>&1 || exit 1
tail -1 $R/gpurun_out/bench_prof_$WL.log | cut -c1-300

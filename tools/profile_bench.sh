#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 passes over bench.py, outputs under gpurun_out/prof_*_<workload> for tools/collect_profiles.py
# (run HERE afterwards: gpurun merges gpurun_out/ back, not profiles/).
#   tools/profile_bench.sh [workload] [extra bench.py args]
# Pass 1: --kernel-trace --stats (per-kernel durations).  Passes 2, 3: --pmc FETCH_SIZE / WRITE_SIZE, each on its own
# (MI355X_MICROARCH.md, HBM section), with --kernel-trace only.  --traffic off: bench.py must not start its own nested rocprofv3
# run from inside a profiled process (under --pmc the preloaded profiler library has initialised the GPU, and the launcher hop
# env -> python3 of the nested run is then an exec from a GPU-initialised process, which the box refuses).
# ONE workload and ONE arithmetic per trace.  The stats pass runs `--modes f16x3 --no-search --no-gan` so that every AverageNs in
# profiles/<tag>_<workload>_kernel_stats.csv belongs to that workload's training step (no embed / GAN / f32 / bf16x6 launches mixed into a kernel's average);
# the cfg5 embed + search pipeline gets a pass of its own: tools/profile_bench.sh embed.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; WL=${1:-cfg2}; shift
cd /tmp
if [ "$WL" = embed ]; then
  rm -rf $R/gpurun_out/prof_stats_embed
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats_embed -- python3 $R/bench.py --workload cfg2 --modes f16x3 --steps 2 --warmup 1 --no-cpu-baseline --no-sustained --no-gan --traffic off --embed-rows 200000 --detail $R/gpurun_out/bench_prof_embed.json "$@" > $R/gpurun_out/bench_prof_embed.log 2>&1 || exit 1
  tail -n 1 $R/gpurun_out/bench_prof_embed.log | cut -c1-300
  exit 0
fi
ONE="--workload $WL --modes f16x3 --no-search --no-gan --no-cpu-baseline --no-sustained --traffic off"
rm -rf $R/gpurun_out/prof_stats_$WL $R/gpurun_out/prof_fetch_$WL $R/gpurun_out/prof_write_$WL
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats_$WL -- python3 $R/bench.py $ONE --steps 30 --warmup 5 --detail $R/gpurun_out/bench_prof_$WL.json "$@" > $R/gpurun_out/bench_prof_$WL.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_fetch_$WL -- python3 $R/bench.py $ONE --steps 3 --warmup 2 --detail /tmp/bench_fetch.json "$@" > $R/gpurun_out/bench_fetch.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_write_$WL -- python3 $R/bench.py $ONE --steps 3 --warmup 2 --detail /tmp/bench_write.json "$@" > $R/gpurun_out/bench_write.log 2>&1 || exit 1
tail -n 1 $R/gpurun_out/bench_prof_$WL.log | cut -c1-300

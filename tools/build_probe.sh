#!/bin/bash
# Timing-only probe library: libganrev_probe16.so = the production sources with -DGR_PROBE_SHAPE16 (conv.hip: every 32x32x16 f16 MFMA of the
# plain / operand-ready forward and data-gradient kernels issued as two 16x16x32 on the same registers; results wrong by design).
# A/B on one box:  GANREV_LIB=$PWD/gan-reverser_amd/ganrev/libganrev_probe16.so python tools/bench_kernels.py 512 R.conv
set -e
cd "$(dirname "$0")/../gan-reverser_amd/csrc"
mkdir -p build_probe
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -I../../include -DGR_PROBE_SHAPE16 -c conv.hip -o build_probe/conv.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build_probe/conv.o build/gemm.o build/elem.o build/search.o build/kmeans.o build/convk.o build/mfmaloop.o build/net.o -o ../ganrev/libganrev_probe16.so -lrccl -lroctx64
echo built ../ganrev/libganrev_probe16.so

#!/bin/bash
# same-box interleaved A/B of G's last convolution (conv3x3_fewout_kernel) between tools/probe/libganrev_base.so and libganrev_new.so: kernel micro-benchmark at the cfg2 and cfg3 shapes
for i in 1 2 3; do
  for arm in base new; do
    echo "== $arm"
    GANREV_LIB=$PWD/tools/probe/libganrev_$arm.so python tools/bench_kernels.py 256 G.convC | grep fwd
    GANREV_LIB=$PWD/tools/probe/libganrev_$arm.so python tools/bench_kernels.py 512 G3.convC | grep fwd
  done
done

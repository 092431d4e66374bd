// What a streaming kernel can reach on tensors the size of cfg2's activations (67 MB each: they sit in the 256 MB Infinity Cache between passes) and of cfg3's
// (537 MB: HBM).  The element-wise pipeline kernels of the step run at 3.3-3.8 TB/s at cfg2 and 4.5-5.4 TB/s at cfg3: is that the
// kernels or the chip?  read2: sum of two tensors (pass A's traffic); read2write1: c = a + b (pass B's); each with G grid-stride workgroups of 256 threads and
// U float4 loads in flight per thread and tensor.  Not product code.   build: hipcc --offload-arch=gfx950 -O3 stream_probe.hip -o stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int U>
__global__ __launch_bounds__(256) void read2(const float4* __restrict__ a, const float4* __restrict__ b, size_t n4, float* __restrict__ out) {
  float acc = 0.f;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += stride * U) {
    float4 x[U], y[U];
#pragma unroll
    for (int k = 0; k < U; ++k) { const size_t i = i0 + k * stride; const size_t j = i < n4 ? i : n4 - 1; x[k] = a[j]; y[k] = b[j]; }
#pragma unroll
    for (int k = 0; k < U; ++k) acc += x[k].x * y[k].x + x[k].y * y[k].y + x[k].z * y[k].z + x[k].w * y[k].w;
  }
  if (acc == 1.2345f) out[0] = acc;
}
template <int U>
__global__ __launch_bounds__(256) void read2write1(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ c, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += stride * U) {
    float4 x[U], y[U];
#pragma unroll
    for (int k = 0; k < U; ++k) { const size_t i = i0 + k * stride; const size_t j = i < n4 ? i : n4 - 1; x[k] = a[j]; y[k] = b[j]; }
#pragma unroll
    for (int k = 0; k < U; ++k) { const size_t i = i0 + k * stride; if (i < n4) c[i] = make_float4(x[k].x + y[k].x, x[k].y + y[k].y, x[k].z + y[k].z, x[k].w + y[k].w); }
  }
}
template <typename F> static float timeit(F f, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(e0); for (int i = 0; i < reps; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}
int main() {
  for (size_t mb : {67, 537}) {
    const size_t bytes = mb << 20, n4 = bytes / 16;
    float4 *a, *b, *c; float* out;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes); hipMalloc(&out, 64);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes); hipMemset(c, 0, bytes);
    for (int g : {1024, 2048, 4096, 8192, 16384}) {
      const float r4 = timeit([&] { hipLaunchKernelGGL(read2<4>, dim3(g), dim3(256), 0, 0, a, b, n4, out); }, 20);
      const float r8 = timeit([&] { hipLaunchKernelGGL(read2<8>, dim3(g), dim3(256), 0, 0, a, b, n4, out); }, 20);
      const float w4 = timeit([&] { hipLaunchKernelGGL(read2write1<4>, dim3(g), dim3(256), 0, 0, a, b, c, n4); }, 20);
      const float w8 = timeit([&] { hipLaunchKernelGGL(read2write1<8>, dim3(g), dim3(256), 0, 0, a, b, c, n4); }, 20);
      printf("%4zu MB tensors, grid %5d: read2 U4 %.1f us = %.2f TB/s, U8 %.1f us = %.2f TB/s | read2write1 U4 %.1f us = %.2f TB/s, U8 %.1f us = %.2f TB/s\n", mb, g,
             r4 * 1e3, 2.0 * bytes / r4 / 1e9, r8 * 1e3, 2.0 * bytes / r8 / 1e9, w4 * 1e3, 3.0 * bytes / w4 / 1e9, w8 * 1e3, 3.0 * bytes / w8 / 1e9);
    }
    hipFree(a); hipFree(b); hipFree(c); hipFree(out);
  }
  return 0;
}

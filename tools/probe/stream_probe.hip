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
// pass A's access pattern with none of its arithmetic: block (c, sp) reads plane c (HW floats) of images [sp * per, (sp + 1) * per) of two [B][C][HW] tensors,
// four float4 groups per thread in flight, one block reduction at the end (as post_backward_a_vec_kernel)
__global__ __launch_bounds__(256) void planes(const float4* __restrict__ a, const float4* __restrict__ b, int B, int C, int q4, int per, float* __restrict__ out) {
  __shared__ float sh[4];
  const int c = blockIdx.x, b0 = blockIdx.y * per, b1 = min(B, b0 + per);
  const unsigned tot = b1 > b0 ? (unsigned)(b1 - b0) * q4 : 0u;
  float acc = 0.f;
  for (unsigned j0 = threadIdx.x; j0 < tot; j0 += 1024) {
    float4 x[4], y[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const unsigned j = min(j0 + 256u * u, tot - 1), bb = j / q4, i = j - bb * q4; const size_t e = ((size_t)(b0 + bb) * C + c) * q4 + i; x[u] = a[e]; y[u] = b[e]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += x[u].x * y[u].x + x[u].y * y[u].y + x[u].z * y[u].z + x[u].w * y[u].w;
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.y * gridDim.x + c] = sh[0] + sh[1] + sh[2] + sh[3];
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
    if (mb == 67) {
      float* o2; hipMalloc(&o2, 4 * 65536);
      for (int splits : {16, 32, 64, 128, 256}) {
        const int B = 256, C = 64, q4 = 256, per = B / splits;
        const float t = timeit([&] { hipLaunchKernelGGL(planes, dim3(C, splits), dim3(256), 0, 0, a, b, B, C, q4, per, o2); }, 20);
        printf("planes pattern 256 x 64 x 32x32 (2 x 67 MB), %3d splits (%5d blocks): %.1f us = %.2f TB/s\n", splits, C * splits, t * 1e3, 2.0 * 67108864.0 / t / 1e9);
      }
      for (int splits : {8, 16, 32, 64}) {
        const int B = 256, C = 128, q4 = 64, per = B / splits;
        const float t = timeit([&] { hipLaunchKernelGGL(planes, dim3(C, splits), dim3(256), 0, 0, a, b, B, C, q4, per, o2); }, 20);
        printf("planes pattern 256 x 128 x 16x16 (2 x 33.5 MB), %3d splits (%5d blocks): %.1f us = %.2f TB/s\n", splits, C * splits, t * 1e3, 2.0 * 33554432.0 / t / 1e9);
      }
      hipFree(o2);
    }
    hipFree(a); hipFree(b); hipFree(c); hipFree(out);
  }
  return 0;
}

// mfmaloop.hip — the SUSTAINED ceiling of the f16x3 convolution inner loop on this device, measured next to the kernels it bounds.
//
// bench.py prices the dominant convolution kernel against the dense f16 MFMA peak of MI355X_MICROARCH.md (2500 TFLOP/s at 2.4 GHz
// / 3 products = 833 TFLOP/s of fp32-accurate FLOPs).  On random data the chip does not hold 2.4 GHz under MFMA load ('DVFS
// give-back' in that guide): a bare loop - operands re-read from LDS by ds_read_b128, three fp16 products per accumulate, 128
// accumulator registers per wave, two waves per SIMD, nothing else - sustains ~1500 TFLOP/s on v_mfma_f32_32x32x16_f16 (the shape
// conv.hip uses) and ~1760 on v_mfma_f32_16x16x32_f16 at the same output tile and LDS bytes per FLOP (round 3, tools/probe/
// shape_probe.hip).  gr_bench_mfma_loop runs that loop for `launches` launches so that the line can state, beside frac = achieved /
// 833, what fraction of the loop's own sustained rate the kernel reaches IN THE SAME RUN ON THE SAME DEVICE (devices differ by
// ~10 % in the clock they hold).  Diagnostic entry point like gr_bench_conv3: nothing on the product path calls it.
#include "kernels.h"
#include <vector>

namespace gr {
typedef _Float16 ml_f16x8 __attribute__((ext_vector_type(8)));
typedef float ml_f32x16 __attribute__((ext_vector_type(16)));
typedef float ml_f32x4 __attribute__((ext_vector_type(4)));
constexpr int ML_KSTEPS = 3;          // distinct operand sets in LDS, cycled (72 KB: two workgroups per CU)
constexpr int ML_A = ML_KSTEPS * 2 * 4 * 64, ML_B = ML_KSTEPS * 2 * 8 * 64;     // uint4 vectors: A [k][term][4 blocks][64 lanes], B [k][term][8 blocks][64 lanes]

template <int SHAPE>   // 0: 32x32x16 (2 A blocks x 4 B blocks per K = 16), 1: 16x16x32 (4 A blocks x 8 B blocks per K = 32): 64 channels x 128 pixels per wave
__global__ __launch_bounds__(256, 2) void mfma_loop_kernel(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
  extern __shared__ uint4 ml_lds[];
  for (int i = threadIdx.x; i < ML_A + ML_B; i += 256) ml_lds[i] = src[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const uint4* A = ml_lds; const uint4* B = ml_lds + ML_A;
  float sum = 0.f;
  if (SHAPE == 0) {
    ml_f32x16 acc[2][4];
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < ML_KSTEPS; ++ks) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          asm volatile("" ::: "memory");             // the operands are re-read from LDS every step (no hoisting out of the loop)
          uint4 a[2][2], b[4][2];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int m = 0; m < 2; ++m) a[m][t] = A[((ks * 2 + t) * 4 + half * 2 + m) * 64 + lane];
#pragma unroll
            for (int n = 0; n < 4; ++n) b[n][t] = B[((ks * 2 + t) * 8 + half * 4 + n) * 64 + lane];
          }
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ml_f16x8, a[m][1]), __builtin_bit_cast(ml_f16x8, b[n][0]), acc[m][n], 0, 0, 0);
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ml_f16x8, a[m][0]), __builtin_bit_cast(ml_f16x8, b[n][1]), acc[m][n], 0, 0, 0);
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ml_f16x8, a[m][0]), __builtin_bit_cast(ml_f16x8, b[n][0]), acc[m][n], 0, 0, 0);
            }
        }
      }
    }
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) sum += acc[m][n][r];
  } else {
    ml_f32x4 acc[4][8];
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 8; ++n) for (int r = 0; r < 4; ++r) acc[m][n][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < ML_KSTEPS; ++ks) {
        asm volatile("" ::: "memory");
        uint4 a[4][2], b[8][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
          for (int m = 0; m < 4; ++m) a[m][t] = A[((ks * 2 + t) * 4 + m) * 64 + lane];
#pragma unroll
          for (int n = 0; n < 8; ++n) b[n][t] = B[((ks * 2 + t) * 8 + n) * 64 + lane];
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 8; ++n) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ml_f16x8, a[m][1]), __builtin_bit_cast(ml_f16x8, b[n][0]), acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ml_f16x8, a[m][0]), __builtin_bit_cast(ml_f16x8, b[n][1]), acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ml_f16x8, a[m][0]), __builtin_bit_cast(ml_f16x8, b[n][0]), acc[m][n], 0, 0, 0);
          }
      }
    }
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 8; ++n) for (int r = 0; r < 4; ++r) sum += acc[m][n][r];
  }
  out[blockIdx.x * 256 + threadIdx.x] = sum;
}

size_t mfma_loop_workspace_bytes() { return (size_t)(ML_A + ML_B) * 16 + 512 * 256 * sizeof(float); }
// f16 FLOPs one launch issues (3 products per fp32-accurate multiply-add: divide by 3 for the figure comparable with "833")
double mfma_loop_flops(int iters) { return 512.0 * 4 * iters * ML_KSTEPS * 96.0 * 16384.0; }
// workspace: mfma_loop_workspace_bytes() of device memory; operands = random fp16 in [-1, 1) (zeros would let the chip clock higher)
void launch_mfma_loop_fill(void* workspace, hipStream_t s) {
  const size_t nv = ML_A + ML_B;
  std::vector<unsigned short> h(nv * 8);
  unsigned st = 12345u;
  for (auto& v : h) {
    st = st * 1664525u + 1013904223u;
    const _Float16 hf = (_Float16)(((st >> 8) * (1.0f / 8388608.0f)) - 1.0f);
    v = __builtin_bit_cast(unsigned short, hf);
  }
  (void)hipMemcpyAsync(workspace, h.data(), nv * 16, hipMemcpyHostToDevice, s);
  (void)hipStreamSynchronize(s);
}
void launch_mfma_loop(int shape, void* workspace, int iters, hipStream_t s) {
  const uint4* src = static_cast<const uint4*>(workspace);
  float* out = reinterpret_cast<float*>(static_cast<char*>(workspace) + (size_t)(ML_A + ML_B) * 16);
  const size_t lds = (size_t)(ML_A + ML_B) * 16;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_loop_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_loop_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  if (shape == 0) hipLaunchKernelGGL(mfma_loop_kernel<0>, dim3(512), dim3(256), lds, s, src, out, iters);
  else hipLaunchKernelGGL(mfma_loop_kernel<1>, dim3(512), dim3(256), lds, s, src, out, iters);
}
}  // namespace gr

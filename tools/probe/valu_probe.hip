// Hardware probe (not product code): issue cost on gfx950 of the VALU instructions the cosine-search inner loop is made of -
// v_mul_f32, v_cvt_f64_f32, v_add_f64, and the integer widening of an fp32 bit pattern to fp64 - as cycles per wave
// instruction with 1 and with 4 waves per SIMD (s_memtime around 8 independent chains x 256 iterations).
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double widen(float p) {      // exact (double)p for finite p, integer ops only (zero and denormals -> 0: probe only)
  const unsigned u = __float_as_uint(p);
  const unsigned e = (u >> 23) & 0xffu;
  unsigned long long d = ((unsigned long long)(u & 0x80000000u) << 32) | ((unsigned long long)(e + 896u) << 52) | ((unsigned long long)(u & 0x7fffffu) << 29);
  if (e == 0) d = (unsigned long long)(u & 0x80000000u) << 32;
  return __longlong_as_double((long long)d);
}
template <int WHAT>
__global__ void probe(float* out, long long* cyc, float x0) {
  float a[8]; double s[8];
  for (int i = 0; i < 8; ++i) { a[i] = x0 + i + threadIdx.x * 1e-3f; s[i] = i; }
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 256; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (WHAT == 0) a[i] = a[i] * 1.0001f;
      if (WHAT == 1) s[i] = s[i] + 1.5;
      if (WHAT == 2) { s[i] = (double)a[i]; a[i] = (float)__double2hiint(s[i]) * 1e-30f + a[i]; }   // cvt + a cheap dependency so it cannot hoist
      if (WHAT == 3) { s[i] = widen(a[i]); a[i] = (float)__double2hiint(s[i]) * 1e-30f + a[i]; }
      if (WHAT == 4) { a[i] = (float)__double2hiint(s[i]) * 1e-30f + a[i]; }
      if (WHAT == 5) { s[i] += (double)(a[i] * 1.0001f); }
      if (WHAT == 6) { s[i] += widen(a[i] * 1.0001f); }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float r = 0; for (int i = 0; i < 8; ++i) r += a[i] + (float)s[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int WHAT> static void run(const char* name, float* out, long long* cyc) {
  for (int waves = 1; waves <= 4; waves *= 4) {
    hipLaunchKernelGGL(probe<WHAT>, dim3(1), dim3(256 * waves), 0, 0, out, cyc, 1.0f);
    long long h = 0; (void)hipDeviceSynchronize(); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("VALU_PROBE %-34s %d wave(s)/SIMD: %.2f cycles per loop body instance (8 x 256 instances per wave)\n", name, waves, (double)h / (8 * 256));
  }
}
int main() {
  float* out; long long* cyc; (void)hipMalloc(&out, 4 * 4096); (void)hipMalloc(&cyc, 64);
  run<0>("v_mul_f32", out, cyc);
  run<1>("v_add_f64", out, cyc);
  run<4>("(dependency only: cvt_i32 mul add)", out, cyc);
  run<2>("v_cvt_f64_f32 + dependency", out, cyc);
  run<3>("integer widen + dependency", out, cyc);
  run<5>("mul_f32, cvt_f64_f32, add_f64", out, cyc);
  run<6>("mul_f32, integer widen, add_f64", out, cyc);
  return 0;
}

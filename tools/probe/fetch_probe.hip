// Calibration of rocprofv3's FETCH_SIZE per load shape on gfx950 (not product code).  MI355X_MICROARCH.md (HBM): FETCH_SIZE reports half
// the bytes of a wide coalesced 16-B-per-lane streaming read; "other access widths are uncalibrated: calibrate on a known byte count in
// your own access pattern".  Each kernel below reads every byte of a 1 GiB buffer exactly once (larger than the 256 MiB Infinity Cache) in
// one load shape this library's hot kernels use; tools/probe/run_fetch_probe.sh runs the binary under `rocprofv3 --pmc FETCH_SIZE` and
// divides the true byte count by what the counter reports: the per-shape factor that bench.py / tools/collect_profiles.py apply.
//   seg<BYTES, SEGL>: every lane loads BYTES (4 / 8 / 16); SEGL consecutive lanes read one contiguous segment (SEGL * BYTES bytes), the
//                     64 / SEGL segments of a wave-instruction lie 128 KiB apart  (SEGL = 64: the plain streaming read; 8 x 16 B = the
//                     128-byte plane rows of the few-output convolution and of the 32-wide tiles; 5 x 16 B ~ the search's 80-byte row pieces)
//   dma<SEGL>:        the same with buffer_load_dwordx4 ... lds (LDS-DMA, 16 B per lane) - the operand-ready convolution / weight-gradient kernels
//   sload:            scalar-cache loads (s_load_dwordx8: weights of the VALU kernels)
// build: hipcc --offload-arch=gfx950 -O3 fetch_probe.hip -o fetch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr size_t BUF = 1ull << 30;
constexpr int STREAMS = 8192;                 // segment s of the buffer lives in stream (s % STREAMS): consecutive segments are BUF / STREAMS = 128 KiB apart

template <int BYTES> struct Vec;
template <> struct Vec<4> { typedef unsigned T; };
template <> struct Vec<8> { typedef uint2 T; };
template <> struct Vec<16> { typedef uint4 T; };
__device__ __forceinline__ unsigned fold(unsigned v) { return v; }
__device__ __forceinline__ unsigned fold(uint2 v) { return v.x ^ v.y; }
__device__ __forceinline__ unsigned fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

// unit u (of BYTES) -> its place in the buffer: segment = u / SEGL, lane in segment = u % SEGL
template <int BYTES, int SEGL>
__device__ __forceinline__ size_t place(size_t u) {
  const size_t nseg = BUF / ((size_t)BYTES * SEGL), per = nseg / STREAMS;
  const size_t seg = u / SEGL, lis = u % SEGL;
  if (SEGL == 64) return u;                                        // plain streaming order
  return ((seg % STREAMS) * per + seg / STREAMS) * SEGL + lis;
}
template <int BYTES, int SEGL>
__global__ __launch_bounds__(256) void seg(const unsigned char* __restrict__ buf, unsigned* __restrict__ out) {
  typedef typename Vec<BYTES>::T V;
  const size_t n = BUF / BYTES, stride = (size_t)gridDim.x * 256;
  unsigned acc = 0;
  for (size_t u0 = (size_t)blockIdx.x * 256 + threadIdx.x; u0 < n; u0 += stride * 4) {
    V v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const size_t u = u0 + k * stride; v[k] = u < n ? reinterpret_cast<const V*>(buf)[place<BYTES, SEGL>(u)] : V{}; }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc ^= fold(v[k]);
  }
  if (acc == 0x12345678u) out[0] = acc;                            // keeps the loads alive
}
// the few-output convolution's shape at 64-wide planes: 8 lanes x 16 B = one 128-byte line = the LEFT half of a 256-byte plane row, the
// next 8 lanes the left half of the next row (256 bytes on); the right halves are read by other workgroups much later
__global__ __launch_bounds__(256) void halfrow(const unsigned char* __restrict__ buf, unsigned* __restrict__ out) {
  const size_t n = BUF / 16, stride = (size_t)gridDim.x * 256;
  unsigned acc = 0;
  for (size_t u0 = (size_t)blockIdx.x * 256 + threadIdx.x; u0 < n; u0 += stride * 4) {
    uint4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const size_t u = u0 + k * stride, hh = u / (n / 2), w = u % (n / 2), line = w / 8, l8 = w % 8;
      v[k] = u < n ? reinterpret_cast<const uint4*>(buf)[(line * 2 + hh) * 8 + l8] : uint4{};
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc ^= fold(v[k]);
  }
  if (acc == 0x12345678u) out[0] = acc;
}
// the same rows with both halves read by ONE wave-instruction's neighbours in time: lanes 0-7 left half of row r, the workgroup's next
// instruction the right half (what a 64-wide tile would do)
__global__ __launch_bounds__(256) void fullrow_two_steps(const unsigned char* __restrict__ buf, unsigned* __restrict__ out) {
  const size_t n = BUF / 16, stride = (size_t)gridDim.x * 256;
  unsigned acc = 0;
  for (size_t u0 = (size_t)blockIdx.x * 256 + threadIdx.x; u0 < n / 2; u0 += stride * 2) {
    uint4 v[4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const size_t w = u0 + k * stride, line = w / 8, l8 = w % 8;
      v[2 * k] = w < n / 2 ? reinterpret_cast<const uint4*>(buf)[(line * 2) * 8 + l8] : uint4{};
      v[2 * k + 1] = w < n / 2 ? reinterpret_cast<const uint4*>(buf)[(line * 2 + 1) * 8 + l8] : uint4{};
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc ^= fold(v[k]);
  }
  if (acc == 0x12345678u) out[0] = acc;
}
template <int SEGL>
__global__ __launch_bounds__(256) void dma(const unsigned char* __restrict__ buf, unsigned* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4* lds = reinterpret_cast<uint4*>(smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // one descriptor per 1 GiB does not fit a 32-bit range together with the offsets below: four windows of 256 MiB
  const size_t n = BUF / 16, stride = (size_t)gridDim.x * 256;
  unsigned acc = 0;
  for (size_t u0 = (size_t)blockIdx.x * 256 + tid; u0 < n; u0 += stride * 4) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const size_t u = u0 + k * stride;
      const size_t byte = (u < n ? place<16, SEGL>(u) : 0) * 16;
      const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(buf) + (byte & ~(size_t)0x0FFFFFFF), 0, 0x10000000, 0x00020000);
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + (wave * 4 + k) * 64), 16, (int)(byte & 0x0FFFFFFF), 0, 0, 0);
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 4; ++k) acc ^= lds[(wave * 4 + k) * 64 + lane].x;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void sload(const unsigned* __restrict__ buf, unsigned* __restrict__ out) {
  // every WAVE reads its own 32-byte blocks through the scalar cache (uniform address -> s_load_dwordx8)
  const size_t nblk = BUF / 32, waves = (size_t)gridDim.x * 4;
  const size_t w = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  unsigned acc = 0;
  for (size_t b = w; b < nblk; b += waves) {
    const unsigned* p = buf + b * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc ^= p[j];
  }
  if (acc == 0x12345678u) out[0] = acc;
}
#define RUN(name, ...) do { hipLaunchKernelGGL(__VA_ARGS__); if (hipDeviceSynchronize() != hipSuccess) { printf("FETCH_PROBE %s failed\n", name); return 2; } printf("FETCH_PROBE ran %s\n", name); } while (0)
int main() {
  unsigned char* buf; unsigned* out;
  if (hipMalloc(&buf, BUF) || hipMalloc(&out, 64)) { printf("alloc failed\n"); return 1; }
  hipMemset(buf, 1, BUF); hipMemset(out, 0, 64);
  const dim3 g(256 * 16), b(256);
  for (int rep = 0; rep < 2; ++rep) {
    RUN("seg<16,64>", (seg<16, 64>), g, b, 0, 0, buf, out);
    RUN("seg<16,32>", (seg<16, 32>), g, b, 0, 0, buf, out);
    RUN("seg<16,16>", (seg<16, 16>), g, b, 0, 0, buf, out);
    RUN("seg<16,8>", (seg<16, 8>), g, b, 0, 0, buf, out);
    RUN("seg<16,4>", (seg<16, 4>), g, b, 0, 0, buf, out);
    RUN("seg<8,64>", (seg<8, 64>), g, b, 0, 0, buf, out);
    RUN("seg<8,16>", (seg<8, 16>), g, b, 0, 0, buf, out);
    RUN("seg<4,64>", (seg<4, 64>), g, b, 0, 0, buf, out);
    RUN("seg<4,32>", (seg<4, 32>), g, b, 0, 0, buf, out);
    RUN("seg<4,16>", (seg<4, 16>), g, b, 0, 0, buf, out);
    RUN("dma<64>", (dma<64>), g, b, 16 * 64 * 16, 0, buf, out);
    RUN("dma<32>", (dma<32>), g, b, 16 * 64 * 16, 0, buf, out);
    RUN("dma<8>", (dma<8>), g, b, 16 * 64 * 16, 0, buf, out);
    RUN("halfrow", halfrow, g, b, 0, 0, buf, out);
    RUN("fullrow_two_steps", fullrow_two_steps, g, b, 0, 0, buf, out);
    RUN("sload", sload, g, b, 0, 0, reinterpret_cast<const unsigned*>(buf), out);
  }
  printf("FETCH_PROBE done: every kernel read %zu bytes\n", BUF);
  return 0;
}

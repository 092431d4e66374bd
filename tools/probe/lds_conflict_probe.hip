// ds_read_b128 bank-conflict probe (not product code): 64 lanes read one 16-byte vector each at vector index f(lane) for a set of lane -> address
// patterns taken from the convolution kernels; the loop is LDS-bound, so time per read shows the conflict factor of a pattern.
//   hipcc --offload-arch=gfx950 -O3 lds_conflict_probe.hip -o lds_conflict_probe && ./lds_conflict_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void probe(const int* __restrict__ idx, float* out, int iters) {
  extern __shared__ uint4 lds[];
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = make_uint4(i, i + 1, i + 2, i + 3);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const unsigned a = (unsigned)(idx[lane] & 1023) * 16u;      // byte address; the 16 reads of a round add 16 vectors each (same slots, other rows)
  u32x4 v0, v1, v2, v3, v4, v5, v6, v7;
  for (int it = 0; it < iters; ++it) {
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:256\n\tds_read_b128 %2, %8 offset:512\n\tds_read_b128 %3, %8 offset:768\n\t"
                 "ds_read_b128 %4, %8 offset:1024\n\tds_read_b128 %5, %8 offset:1280\n\tds_read_b128 %6, %8 offset:1536\n\tds_read_b128 %7, %8 offset:1792\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(a) : "memory");
  }
  out[blockIdx.x * 256 + threadIdx.x] = (float)(v0.x ^ v1.x ^ v2.x ^ v3.x ^ v4.x ^ v5.x ^ v6.x ^ v7.x);
}
int main() {
  struct Pat { const char* name; int (*f)(int); };
  static Pat pats[] = {
    {"64 contiguous vectors", [](int l) { return l; }},
    {"quarters at +0,+32,+64,+96 (A reads, 16x16x32)", [](int l) { return (l >> 4) * 32 + (l & 15); }},
    {"halves at +0,+32 of 32 contiguous (A reads, 32x32x16)", [](int l) { return (l >> 5) * 32 + (l & 31); }},
    {"B TW>=16, 16x16x32: q&1 -> +648, q>>1 -> +1", [](int l) { int q = l >> 4; return (q & 1) * 648 + (q >> 1) + (l & 15); }},
    {"B TW=8, 16x16x32: rows 4 apart (PC 10), q&1 -> +800, q>>1 -> +1", [](int l) { int q = l >> 4, j = l & 15; return (q & 1) * 800 + (q >> 1) + (j >> 3) * 40 + (j & 7); }},
    {"B 32x32x16 TW=16: rows 8 apart (PC 18), h -> +648", [](int l) { int j = l & 31; return (l >> 5) * 648 + (j >> 4) * 144 + (j & 15); }},
    {"B 16x16x32 with planes padded: q&1 -> +656, q>>1 -> +1", [](int l) { int q = l >> 4; return (q & 1) * 656 + (q >> 1) + (l & 15); }},
    {"B 16x16x32, q>>1 -> +1 only (both halves same plane)", [](int l) { int q = l >> 4; return (q >> 1) + (l & 15); }},
    {"B 16x16x32, q&1 -> +648 only", [](int l) { int q = l >> 4; return (q & 1) * 648 + (l & 15); }},
    {"B 16x16x32, q&1 -> +644 (== 4 mod 16), q>>1 -> +1", [](int l) { int q = l >> 4; return (q & 1) * 644 + (q >> 1) + (l & 15); }},
    {"all lanes same vector", [](int) { return 5; }},
    {"stride 2 vectors", [](int l) { return 2 * l; }},
  };
  int* d; float* o; hipMalloc(&d, 64 * 4); hipMalloc(&o, 1024 * 256 * 4);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 2048 * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (auto& p : pats) {
    std::vector<int> h(64); for (int l = 0; l < 64; ++l) h[l] = p.f(l);
    hipMemcpy(d, h.data(), 256, hipMemcpyHostToDevice);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(probe, dim3(1024), dim3(256), 2048 * 16, 0, d, o, 4000);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    // per CU: 4 workgroups x 4 waves x 4000 x 8 reads of 1 KB
    printf("%-70s %.3f ms  %.1f B/clk/CU at 2.4 GHz\n", p.name, best, 16.0 * 4000 * 8 * 1024 / (best * 1e-3) / 2.4e9);
  }
  return 0;
}

// Hardware probe (not product code): lane mapping of ds_read_b64_tr_b16 on gfx950 as the P16 weight-gradient kernel uses it.
// LDS holds a [64 rows][16 columns] image of 16-bit values v = 100 * row + col; 16-lane group g reads the block of rows
// 4g .. 4g+3: lane 4q+p of the group supplies the address of row q, columns 4p .. 4p+3; expected: lane i of the group
// receives column i of the four rows (element j = row 4g + j).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void tr_probe(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 16];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 16; i += 64) lds[i] = src[i];
  __syncthreads();
  const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
#if defined(__HIP_DEVICE_COMPILE__)
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + (4 * g + q) * 16 + 4 * p));
  for (int j = 0; j < 4; ++j) dst[lane * 4 + j] = (unsigned short)v[j];
#endif
}
int main() {
  unsigned short h[1024], o[256];
  for (int r = 0; r < 64; ++r) for (int c = 0; c < 16; ++c) h[r * 16 + c] = (unsigned short)(100 * r + c);
  unsigned short *ds, *dd; (void)hipMalloc(&ds, sizeof h); (void)hipMalloc(&dd, sizeof o);
  (void)hipMemcpy(ds, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, ds, dd);
  if (hipDeviceSynchronize() != hipSuccess) { printf("TR_PROBE launch failed\n"); return 2; }
  (void)hipMemcpy(o, dd, sizeof o, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) if (o[l * 4 + j] != 100 * (4 * (l >> 4) + j) + (l & 15)) ++bad;
  printf("TR_PROBE bad=%d ; lane 5 got %d %d %d %d (expected 5 105 205 305) ; lane 21 got %d %d %d %d (expected 405 505 605 705)\n", bad,
         o[20], o[21], o[22], o[23], o[84], o[85], o[86], o[87]);
  return bad ? 1 : 0;
}

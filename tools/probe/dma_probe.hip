// Hardware probe (not product code): `buffer_load_dwordx4 ... offen lds` (LDS-DMA) on gfx950 -
//  (1) lane-linear LDS destination with a per-lane source offset, (2) a lane whose offset is parked past the
//  descriptor's range must leave ZEROS in its LDS slot (the conv kernels rely on this for the zero padding).
// build: hipcc --offload-arch=gfx950 -O3 dma_probe.hip -o dma_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void dma_probe(const uint4* __restrict__ src, uint4* __restrict__ dst, int nbytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4* lds = reinterpret_cast<uint4*>(smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 512; i += blockDim.x) lds[i] = make_uint4(0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu);
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(src), 0, nbytes, 0x00020000);
  for (int i = 0; i < 2; ++i) {
    const int e = wave * 128 + i * 64 + (63 - lane);                 // reversed inside the instruction
    const int voff = (lane % 3 == 1) ? 0x7FFFF000 : e * 16;            // every third lane parked out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + wave * 128 + i * 64), 16, voff, 0, 0, 0);
  }
  __syncthreads();
  for (int i = tid; i < 512; i += blockDim.x) dst[i] = lds[i];
}
int main() {
  std::vector<uint4> h(512), o(512);
  for (int i = 0; i < 512; ++i) h[i] = make_uint4(i, i * 3 + 1, i * 7 + 2, i * 11 + 3);
  uint4 *ds, *dd; hipMalloc(&ds, 512 * 16); hipMalloc(&dd, 512 * 16);
  hipMemcpy(ds, h.data(), 512 * 16, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(dma_probe, dim3(1), dim3(256), 512 * 16, 0, ds, dd, 512 * 16);
  if (hipDeviceSynchronize() != hipSuccess) { printf("DMA_PROBE launch failed\n"); return 2; }
  hipMemcpy(o.data(), dd, 512 * 16, hipMemcpyDeviceToHost);
  int bad = 0, zeros = 0, stale = 0;
  for (int w = 0; w < 4; ++w) for (int i = 0; i < 2; ++i) for (int l = 0; l < 64; ++l) {
    const int slot = w * 128 + i * 64 + l, e = w * 128 + i * 64 + (63 - l);
    const uint4 v = o[slot];
    if (l % 3 == 1) { if (v.x == 0 && v.y == 0 && v.z == 0 && v.w == 0) ++zeros; else if (v.x == 0xdeadbeefu) ++stale; else ++bad; }
    else if (v.x != h[e].x || v.y != h[e].y || v.z != h[e].z || v.w != h[e].w) ++bad;
  }
  printf("DMA_PROBE bad=%d parked_lanes_zeroed=%d parked_lanes_left_stale=%d\n", bad, zeros, stale);
  return bad ? 1 : 0;
}

// Early hardware/toolchain probe (not product code): checks the fp32 MFMA fragment
// layout, that a hipcc-7.2 code object runs under whichever HIP runtime the process
// has loaded (torch's or /opt/rocm's), and that RCCL initialises with nranks=1.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void mfma_probe(const float* A, const float* B, float* C) {
  // A: 32x2 row-major [i][k], B: 2x32 row-major [k][j], C: 32x32
  int l = threadIdx.x;
  float a = A[(l & 31) * 2 + (l >> 5)];
  float b = B[(l >> 5) * 32 + (l & 31)];
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    int col = l & 31;
    C[row * 32 + col] = acc[r];
  }
}

extern "C" int probe_mfma(float* dA, float* dB, float* dC) {
  hipLaunchKernelGGL(mfma_probe, dim3(1), dim3(64), 0, 0, dA, dB, dC);
  hipError_t e = hipDeviceSynchronize();
  return (int)e;
}

extern "C" int probe_selftest() {
  float hA[64], hB[64], hC[1024], ref[1024];
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 2; ++k) hA[i * 2 + k] = (float)(i * 3 + k * 7 + 1);
  for (int k = 0; k < 2; ++k) for (int j = 0; j < 32; ++j) hB[k * 32 + j] = (float)(j * 5 - k * 11 + 2);
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) ref[i * 32 + j] = hA[i * 2] * hB[j] + hA[i * 2 + 1] * hB[32 + j];
  float *dA, *dB, *dC;
  if (hipMalloc(&dA, sizeof hA) || hipMalloc(&dB, sizeof hB) || hipMalloc(&dC, sizeof hC)) return -1;
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
  hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  int e = probe_mfma(dA, dB, dC);
  if (e) return -100 - e;
  hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 1024; ++i) if (hC[i] != ref[i]) ++bad;
  hipFree(dA); hipFree(dB); hipFree(dC);
  return bad;
}

extern "C" int probe_info(char* buf, int n) {
  hipDeviceProp_t p; int v = 0, rv = 0;
  if (hipGetDeviceProperties(&p, 0)) return -1;
  hipRuntimeGetVersion(&rv); hipDriverGetVersion(&v);
  Dl_info di; const char* where = "?";
  if (dladdr((void*)&hipDeviceSynchronize, &di) && di.dli_fname) where = di.dli_fname;
  snprintf(buf, n, "name=%s arch=%s CUs=%d clock=%dkHz memclk=%dkHz mem=%zuMB l2=%d lds=%zu runtime=%d driver=%d hiplib=%s",
           p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate, p.memoryClockRate,
           p.totalGlobalMem >> 20, p.l2CacheSize, p.sharedMemPerBlock, rv, v, where);
  return 0;
}

extern "C" int probe_rccl() {
  ncclUniqueId id; ncclComm_t comm;
  ncclResult_t r = ncclGetUniqueId(&id);
  if (r) return -10 - (int)r;
  r = ncclCommInitRank(&comm, 1, id, 0);
  if (r) return -20 - (int)r;
  float h[256]; for (int i = 0; i < 256; ++i) h[i] = (float)i;
  float* d; hipMalloc(&d, sizeof h); hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  hipStream_t s; hipStreamCreate(&s);
  r = ncclAllReduce(d, d, 256, ncclFloat, ncclSum, comm, s);
  if (r) return -30 - (int)r;
  hipStreamSynchronize(s);
  float o[256]; hipMemcpy(o, d, sizeof o, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 256; ++i) if (o[i] != h[i]) ++bad;
  ncclCommDestroy(comm); hipFree(d); hipStreamDestroy(s);
  Dl_info di; if (dladdr((void*)&ncclAllReduce, &di) && di.dli_fname) printf("rccl lib: %s\n", di.dli_fname);
  return bad;
}

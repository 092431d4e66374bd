// gemm.hip — fp32 MFMA GEMM for nn.Linear (reference models.lua:115 G.fc, models.lua:447,451 R.fc1/fc2):
//   C[m][n] (+)= sum_k A(m,k) * B(n,k) (+ bias[n]),   A(m,k) = A[m*rsA + k*ksA],  B(n,k) = Bm[n*rsB + k*ksB]
// forward  y = x W^T + b : A = x   (K-contiguous), B = W        (K-contiguous)
// bwd-data gx = gy W     : A = gy  (K-contiguous), B(n=i,k=o)=W (N-contiguous)
// bwd-wt   gW += gy^T x  : A(m=o,k=b)=gy (M-contiguous), B(n=i,k=b)=x (N-contiguous)
// Workgroup tile 64x64 (4 waves, one 32x32 MFMA accumulator each), K chunks of 32 through LDS; split-K over
// blockIdx.z into fp32 slabs that a reduce kernel sums in a fixed order (deterministic, no atomics).
#include "kernels.h"

namespace gr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
  const float* A; const float* Bm; float* C; float* slab; const float* bias;
  long rsA, ksA, rsB, ksB, ldc;
  int M, N, K, klen, nsplit, accumulate;
  ConvEpilogue ep;       // optional per-column epilogue (evaluate()-mode BatchNorm over the N features + activation); nsplit == 1 only
  int has_ep;
  unsigned* amax_out;    // nullable (nsplit == 1): max|C| folded into this f16x3 scale slot
  const unsigned *amax_a, *amax_b;   // f16x3 kernel: scale slots (max|A|, max|B|)
};

// out = act(((v - mean[n]) * invstd[n]) * gamma[n] + beta[n]): nn.BatchNormalization in evaluate() mode + activation on the
// Linear output (G: models.lua:115-117), the same operation order and roundings as the stand-alone pipeline kernel
__device__ __forceinline__ float gemm_epilogue(const ConvEpilogue& ep, float v, int n) {
  if (ep.mean) v = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(v, ep.mean[n]), ep.invstd[n]), ep.gamma[n]), ep.beta[n]);
  switch (ep.act) {
    case ACT_ELU: return v <= 0.f ? (expf(v) - 1.f) : v;
    case ACT_RELU: return v > 0.f ? v : 0.f;
    case ACT_LEAKYRELU: return v > 0.f ? v : __fmul_rn(v, ep.slope);
    case ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    case ACT_TANH: return tanhf(v);
    default: return v;
  }
}

// A 64(rows) x 32(k) operand tile is fetched into 8 registers per thread (one float4 pair along k when the operand is
// K-contiguous and 16-byte aligned, scalars otherwise) while the MFMAs of the previous tile run, then written to LDS as
// T[k][row] (row stride 65: conflict-free both for the transposing writes and for the MFMA operand reads).
template <bool KCONTIG>
__device__ __forceinline__ void tile_load(const float* __restrict__ P, long rs, long ks, int row0, int nrows,
                                          int k0, int kend, bool vec, float (&v)[8], int tid) {
  if (KCONTIG && vec) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = tid + 256 * i, r = f >> 3, k = (f & 7) * 4;     // 8 float4 per row
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row0 + r < nrows && k0 + k < kend) t = *reinterpret_cast<const float4*>(P + (long)(row0 + r) * rs + k0 + k);
      v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int r, k;
      if (KCONTIG) { k = tid & 31; r = (tid >> 5) + 8 * i; }
      else { r = tid & 63; k = (tid >> 6) + 4 * i; }
      v[i] = (row0 + r < nrows && k0 + k < kend) ? P[(long)(row0 + r) * rs + (long)(k0 + k) * ks] : 0.f;
    }
  }
}
template <bool KCONTIG>
__device__ __forceinline__ void tile_store(float* T, bool vec, const float (&v)[8], int tid) {
  if (KCONTIG && vec) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = tid + 256 * i, r = f >> 3, k = (f & 7) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) T[(k + j) * 65 + r] = v[4 * i + j];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int r, k;
      if (KCONTIG) { k = tid & 31; r = (tid >> 5) + 8 * i; }
      else { r = tid & 63; k = (tid >> 6) + 4 * i; }
      T[k * 65 + r] = v[i];
    }
  }
}

template <bool AK, bool BK>
__global__ __launch_bounds__(256) void gemm_mfma_kernel(GemmArgs a) {
  __shared__ float As[32 * 65];
  __shared__ float Bs[32 * 65];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int kbeg = blockIdx.z * a.klen;
  const int kend = min(a.K, kbeg + a.klen);
  const bool avec = AK && (a.rsA & 3) == 0 && (kbeg & 3) == 0 && (a.K & 3) == 0 && ((uintptr_t)a.A & 15) == 0;
  const bool bvec = BK && (a.rsB & 3) == 0 && (kbeg & 3) == 0 && (a.K & 3) == 0 && ((uintptr_t)a.Bm & 15) == 0;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float av[8], bv[8];
  tile_load<AK>(a.A, a.rsA, a.ksA, m0, a.M, kbeg, kend, avec, av, tid);
  tile_load<BK>(a.Bm, a.rsB, a.ksB, n0, a.N, kbeg, kend, bvec, bv, tid);
  for (int k0 = kbeg; k0 < kend; k0 += 32) {
    tile_store<AK>(As, avec, av, tid);
    tile_store<BK>(Bs, bvec, bv, tid);
    __syncthreads();
    if (k0 + 32 < kend) {
      tile_load<AK>(a.A, a.rsA, a.ksA, m0, a.M, k0 + 32, kend, avec, av, tid);
      tile_load<BK>(a.Bm, a.rsB, a.ksB, n0, a.N, k0 + 32, kend, bvec, bv, tid);
    }
#pragma unroll
    for (int kk = 0; kk < 32; kk += 2) {
      const float x = As[(kk + h) * 65 + wm * 32 + l31];
      const float y = Bs[(kk + h) * 65 + wn * 32 + l31];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int n = n0 + wn * 32 + l31;
  float omax = 0.f;
  if (n < a.N) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (m < a.M) {
        if (a.nsplit > 1) {
          a.slab[((size_t)blockIdx.z * a.M + m) * a.N + n] = acc[r];
        } else {
          float v = acc[r] + (a.bias ? a.bias[n] : 0.f);
          if (a.has_ep) v = gemm_epilogue(a.ep, v, n);
          float* c = a.C + (long)m * a.ldc + n;
          v = a.accumulate ? *c + v : v;
          *c = v;
          omax = fmaxf(omax, fabsf(v));
        }
      }
    }
  }
  if (a.amax_out && a.nsplit == 1) absmax_commit(omax, a.amax_out);
}

// ---------------------------------------------------------------- f16x3 GEMM (same arithmetic as the f16x3 convolutions)
// Both operands are scaled by the power of two their device-tracked maxima call for and split into two fp16 terms while they
// are staged; three products per 16-wide k step on v_mfma_f32_32x32x16_f16, result scaled back with ldexp.  A workgroup tile is
// 64 x 64 with K chunks of 32 like the fp32 kernel (6 MFMAs per wave and chunk instead of 16 twice as long).
//   LDS operand image [2 terms][4 k-groups of 8][64 rows] of 16-byte vectors, per operand
template <bool KCONTIG>
__device__ __forceinline__ void tile_load8(const float* __restrict__ P, long rs, long ks, int row0, int nrows, int k0, int kend, bool vec,
                                           float (&v)[8], int tid) {
  // K-contiguous: thread = (row tid >> 2, k-group tid & 3): 32 bytes of one row; otherwise thread = (row tid & 63, k-group tid >> 6)
  const int r = KCONTIG ? tid >> 2 : tid & 63, g = KCONTIG ? tid & 3 : tid >> 6;
  const bool rin = row0 + r < nrows;
  if (KCONTIG && vec) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int k = k0 + 8 * g + 4 * i;
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rin && k < kend) t = *reinterpret_cast<const float4*>(P + (long)(row0 + r) * rs + k);      // K % 4 == 0: whole vectors
      v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = k0 + 8 * g + i;
      v[i] = (rin && k < kend) ? P[(long)(row0 + r) * rs + (long)k * ks] : 0.f;
    }
  }
}
template <bool KCONTIG>
__device__ __forceinline__ void tile_store8(uint4* T, float sc, const float (&v)[8], int tid) {
  const int r = KCONTIG ? tid >> 2 : tid & 63, g = KCONTIG ? tid & 3 : tid >> 6;
  uint4 t0, t1;
  split8_f16(v, sc, t0, t1);
  T[(0 * 4 + g) * 64 + r] = t0; T[(1 * 4 + g) * 64 + r] = t1;
}

template <bool AK, bool BK>
__global__ __launch_bounds__(256) void gemm_f16x3_kernel(GemmArgs a) {
  __shared__ uint4 As[2 * 4 * 64];
  __shared__ uint4 Bs[2 * 4 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int kbeg = blockIdx.z * a.klen;
  const int kend = min(a.K, kbeg + a.klen);
  const bool avec = AK && (a.rsA & 3) == 0 && (kbeg & 3) == 0 && (a.K & 3) == 0 && ((uintptr_t)a.A & 15) == 0;
  const bool bvec = BK && (a.rsB & 3) == 0 && (kbeg & 3) == 0 && (a.K & 3) == 0 && ((uintptr_t)a.Bm & 15) == 0;
  const int ka = f16_scale_exp(absmax_read(a.amax_a)), kb = f16_scale_exp(absmax_read(a.amax_b));
  const float sa = pow2f(ka), sb = pow2f(kb);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float av[8], bv[8];
  tile_load8<AK>(a.A, a.rsA, a.ksA, m0, a.M, kbeg, kend, avec, av, tid);
  tile_load8<BK>(a.Bm, a.rsB, a.ksB, n0, a.N, kbeg, kend, bvec, bv, tid);
  for (int k0 = kbeg; k0 < kend; k0 += 32) {
    tile_store8<AK>(As, sa, av, tid);
    tile_store8<BK>(Bs, sb, bv, tid);
    __syncthreads();
    if (k0 + 32 < kend) {
      tile_load8<AK>(a.A, a.rsA, a.ksA, m0, a.M, k0 + 32, kend, avec, av, tid);
      tile_load8<BK>(a.Bm, a.rsB, a.ksB, n0, a.N, k0 + 32, kend, bvec, bv, tid);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {                    // two 16-wide k steps: lanes 0-31 take k-group 2kk, lanes 32-63 k-group 2kk+1
      uint4 x[2], y[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) { x[t] = As[(t * 4 + 2 * kk + h) * 64 + wm * 32 + l31]; y[t] = Bs[(t * 4 + 2 * kk + h) * 64 + wn * 32 + l31]; }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x[1]), __builtin_bit_cast(f16x8, y[0]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x[0]), __builtin_bit_cast(f16x8, y[1]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x[0]), __builtin_bit_cast(f16x8, y[0]), acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int n = n0 + wn * 32 + l31;
  float omax = 0.f;
  if (n < a.N) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (m < a.M) {
        const float res = ldexpf(acc[r], -(ka + kb));
        if (a.nsplit > 1) {
          a.slab[((size_t)blockIdx.z * a.M + m) * a.N + n] = res;
        } else {
          float v = res + (a.bias ? a.bias[n] : 0.f);
          if (a.has_ep) v = gemm_epilogue(a.ep, v, n);
          float* c = a.C + (long)m * a.ldc + n;
          v = a.accumulate ? *c + v : v;
          *c = v;
          omax = fmaxf(omax, fabsf(v));
        }
      }
    }
  }
  if (a.amax_out && a.nsplit == 1) absmax_commit(omax, a.amax_out);
}

__global__ void gemm_splitk_reduce_kernel(const float* __restrict__ slab, float* __restrict__ C, const float* __restrict__ bias,
                                          long ldc, int M, int N, int nsplit, int accumulate) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= (long)M * N) return;
  const int n = (int)(i % N); const int m = (int)(i / N);
  float s = 0.f;
  for (int z = 0; z < nsplit; ++z) s += slab[(size_t)z * M * N + i];
  s += bias ? bias[n] : 0.f;
  float* c = C + (long)m * ldc + n;
  *c = accumulate ? *c + s : s;
}

static void gemm_plan(int M, int N, int K, int& nsplit, int& klen) {
  const long tiles = (long)((M + 63) / 64) * ((N + 63) / 64);
  nsplit = 1;
  if (tiles < 256 && K >= 256) {
    long want = (512 + tiles - 1) / tiles;
    const long maxs = K / 128;
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    nsplit = (int)want;
  }
  klen = round_up((K + nsplit - 1) / nsplit, 32);
  nsplit = (K + klen - 1) / klen;
}

size_t gemm_workspace_bytes(int M, int N, int K) {
  int ns, kl; gemm_plan(M, N, K, ns, kl);
  return ns > 1 ? sizeof(float) * (size_t)ns * M * N : 0;
}

bool gemm_epilogue_possible(int M, int N, int K) { int ns, kl; gemm_plan(M, N, K, ns, kl); return ns == 1; }

void launch_gemm(const float* A, long rsA, long ksA, const float* Bm, long rsB, long ksB,
                 float* C, long ldc, const float* bias, bool accumulate, int M, int N, int K,
                 void* workspace, hipStream_t s, const ConvEpilogue* ep, unsigned* amax_out,
                 const unsigned* amax_a, const unsigned* amax_b) {
  GemmArgs a{};
  if (ep) { a.ep = *ep; a.has_ep = 1; }
  a.amax_out = amax_out; a.amax_a = amax_a; a.amax_b = amax_b;
  const bool f16 = amax_a != nullptr && amax_b != nullptr;
  a.A = A; a.Bm = Bm; a.C = C; a.slab = reinterpret_cast<float*>(workspace); a.bias = bias;
  a.rsA = rsA; a.ksA = ksA; a.rsB = rsB; a.ksB = ksB; a.ldc = ldc;
  a.M = M; a.N = N; a.K = K; a.accumulate = accumulate ? 1 : 0;
  gemm_plan(M, N, K, a.nsplit, a.klen);
  dim3 grid((N + 63) / 64, (M + 63) / 64, a.nsplit);
  const bool ak = ksA == 1, bk = ksB == 1;
  {
  KtScope kt(f16 ? "gemm_f16x3_kernel" : "gemm_mfma_kernel", 2.0 * M * N * (double)K, 4.0 * ((double)M * K + (double)N * K + (double)M * N), s);
  if (f16) {
    if (ak && bk) hipLaunchKernelGGL((gemm_f16x3_kernel<true, true>), grid, dim3(256), 0, s, a);
    else if (ak && !bk) hipLaunchKernelGGL((gemm_f16x3_kernel<true, false>), grid, dim3(256), 0, s, a);
    else if (!ak && bk) hipLaunchKernelGGL((gemm_f16x3_kernel<false, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((gemm_f16x3_kernel<false, false>), grid, dim3(256), 0, s, a);
  }
  else if (ak && bk) hipLaunchKernelGGL((gemm_mfma_kernel<true, true>), grid, dim3(256), 0, s, a);
  else if (ak && !bk) hipLaunchKernelGGL((gemm_mfma_kernel<true, false>), grid, dim3(256), 0, s, a);
  else if (!ak && bk) hipLaunchKernelGGL((gemm_mfma_kernel<false, true>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((gemm_mfma_kernel<false, false>), grid, dim3(256), 0, s, a);
  }
  if (a.nsplit > 1) {
    const long n = (long)M * N;
    KtScope kt("gemm_splitk_reduce_kernel", (double)n * a.nsplit, 4.0 * n * (a.nsplit + 1.0), s);
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                       a.slab, C, bias, ldc, M, N, a.nsplit, a.accumulate);
  }
}

}  // namespace gr

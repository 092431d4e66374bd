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
  int vec_store;         // big kernel: whole blocks leave through the LDS transpose (16-byte stores); 0 = the dword stores (GR_GEMM_DWORD_STORES: A/B control)
  unsigned* amax_out;    // nullable (nsplit == 1): max|C| folded into this f16x3 scale slot
  const unsigned *amax_a, *amax_b;   // f16x3 kernel: scale slots (max|A|, max|B|)
};

// out = act(((v - mean[n]) * invstd[n]) * gamma[n] + beta[n]): nn.BatchNormalization in evaluate() mode + activation on the
// Linear output (G: models.lua:115-117), the same operation order and roundings as the stand-alone pipeline kernel
__device__ __forceinline__ float gemm_activation(const ConvEpilogue& ep, float v) {
  switch (ep.act) {
    case ACT_ELU: return v <= 0.f ? (expf(v) - 1.f) : v;
    case ACT_RELU: return v > 0.f ? v : 0.f;
    case ACT_LEAKYRELU: return v > 0.f ? v : __fmul_rn(v, ep.slope);
    case ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    case ACT_TANH: return tanhf(v);
    default: return v;
  }
}

// Store one 32 x 32 accumulator block (MFMA layout: lane -> column n, register r -> row mb + (r & 3) + 8 (r >> 2)).
// The per-column parameters and, when accumulating, the 16 old values are fetched BEFORE the first store: loads and stores
// share one in-order counter on this hardware, so a load issued after a store waits for that store to reach memory, and the
// plain "load, add, store" per element loop serialises 16 memory round trips (measured: 244 us of a 330 us launch).
__device__ __forceinline__ void gemm_store_block(const GemmArgs& a, const f32x16& acc, int mb, int n, int shift, float& omax) {
  if (n >= a.N) return;
  const bool full = mb + 27 < a.M;          // every row of the block inside the matrix: no per-row tests
  if (a.nsplit > 1) {
    float* sl = a.slab + ((size_t)blockIdx.z * a.M + mb) * a.N + n;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dm = (r & 3) + 8 * (r >> 2);
      if (full || mb + dm < a.M) sl[(size_t)dm * a.N] = ldexpf(acc[r], shift);
    }
    return;
  }
  float v[16];
  const float bias = a.bias ? a.bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = ldexpf(acc[r], shift) + bias;
  if (a.has_ep) {
    if (a.ep.mean) {
      const float mean = a.ep.mean[n], invstd = a.ep.invstd[n], gamma = a.ep.gamma[n], beta = a.ep.beta[n];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(v[r], mean), invstd), gamma), beta);
    }
    switch (a.ep.act) {                     // one uniform branch per block, not per element
      case ACT_RELU:
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
        break;
      case ACT_LEAKYRELU:
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = v[r] > 0.f ? v[r] : __fmul_rn(v[r], a.ep.slope);
        break;
      case ACT_NONE: break;
      default:
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = gemm_activation(a.ep, v[r]);
    }
  }
  float* c0 = a.C + (long)mb * a.ldc + n;
  if (a.accumulate) {
    float old[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dm = (r & 3) + 8 * (r >> 2);
      old[r] = (full || mb + dm < a.M) ? c0[(long)dm * a.ldc] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = old[r] + v[r];
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int dm = (r & 3) + 8 * (r >> 2);
    if (full || mb + dm < a.M) { c0[(long)dm * a.ldc] = v[r]; omax = fmaxf(omax, fabsf(v[r])); }
  }
}

// The same block through a per-wave LDS transpose: 16-byte stores.  The accumulator layout gives one dword per lane and store - 16 store
// instructions per block, each two 128-byte row pieces: store-ISSUE-bound (the convolution kernels met the same wall: conv.hip, "Output stores
// through an LDS transpose"): G.fc at cfg3, a 268 MB output stream behind a K = 100 product, ran at 1.85 TB/s, and a third of an fc1 launch
// was its split-K slab going out dword by dword.  Here a wave writes the block's 32 x 32 values to its own staging rows (stride 36 floats)
// and reads them back four consecutive columns per lane: 4 store instructions of 8 rows x 128 bytes.  For blocks that lie whole inside the
// matrix, with N and ldc multiples of 4 and no accumulation into C; everything else takes gemm_store_block.  mb0 = the block's first row.
__device__ __forceinline__ bool gemm_block_vec_ok(const GemmArgs& a, int mb0, int nb0) {
  return a.vec_store && !a.accumulate && mb0 + 32 <= a.M && nb0 + 32 <= a.N && (a.N & 3) == 0 && (a.ldc & 3) == 0 && (((uintptr_t)a.C | (uintptr_t)a.slab) & 15) == 0;
}
__device__ __forceinline__ void gemm_store_block_vec(const GemmArgs& a, const f32x16& acc, int mb0, int nb0, int shift, float& omax, float* stg, int lane) {
  const int l31 = lane & 31, h = lane >> 5, n = nb0 + l31;
  float v[16];
  if (a.nsplit > 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = ldexpf(acc[r], shift);
  } else {
    const float bias = a.bias ? a.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = ldexpf(acc[r], shift) + bias;
    if (a.has_ep) {
      if (a.ep.mean) {
        const float mean = a.ep.mean[n], invstd = a.ep.invstd[n], gamma = a.ep.gamma[n], beta = a.ep.beta[n];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(v[r], mean), invstd), gamma), beta);
      }
      switch (a.ep.act) {
        case ACT_RELU:
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
          break;
        case ACT_LEAKYRELU:
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = v[r] > 0.f ? v[r] : __fmul_rn(v[r], a.ep.slope);
          break;
        case ACT_NONE: break;
        default:
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = gemm_activation(a.ep, v[r]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * h) * 36 + l31] = v[r];
  float* base = a.nsplit > 1 ? a.slab + ((size_t)blockIdx.z * a.M + mb0) * a.N + nb0 : a.C + (long)mb0 * a.ldc + nb0;
  const long ld = a.nsplit > 1 ? (long)a.N : a.ldc;
  const int c4 = (lane & 7) * 4, r0 = lane >> 3;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = r0 + 8 * i;
    const float4 t = *reinterpret_cast<const float4*>(stg + row * 36 + c4);       // (a wave's LDS operations execute in order: no barrier)
    *reinterpret_cast<float4*>(base + (long)row * ld + c4) = t;
    if (a.nsplit == 1) omax = fmaxf(omax, fmaxf(fmaxf(fabsf(t.x), fabsf(t.y)), fmaxf(fabsf(t.z), fabsf(t.w))));
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
  float omax = 0.f;
  gemm_store_block(a, acc, m0 + wm * 32 + 4 * h, n0 + wn * 32 + l31, 0, omax);
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
  T[(0 * 4 + g) * 68 + r] = t0; T[(1 * 4 + g) * 68 + r] = t1;       // 68: the four k-groups of a row land in distinct banks
}

template <bool AK, bool BK>
__global__ __launch_bounds__(256) void gemm_f16x3_kernel(GemmArgs a) {
  __shared__ uint4 As[2 * 4 * 68];
  __shared__ uint4 Bs[2 * 4 * 68];
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
      for (int t = 0; t < 2; ++t) { x[t] = As[(t * 4 + 2 * kk + h) * 68 + wm * 32 + l31]; y[t] = Bs[(t * 4 + 2 * kk + h) * 68 + wn * 32 + l31]; }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x[1]), __builtin_bit_cast(f16x8, y[0]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x[0]), __builtin_bit_cast(f16x8, y[1]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x[0]), __builtin_bit_cast(f16x8, y[0]), acc, 0, 0, 0);
    }
    __syncthreads();
  }
  float omax = 0.f;
  gemm_store_block(a, acc, m0 + wm * 32 + 4 * h, n0 + wn * 32 + l31, -(ka + kb), omax);
  if (a.amax_out && a.nsplit == 1) absmax_commit(omax, a.amax_out);
}

// 128 x 128 tile: every wave owns a 64 x 64 quadrant (2 x 2 accumulators), so an operand vector read from LDS feeds two MFMAs and
// the staging work per MFMA halves.  Used when the output has at least two such tiles per dimension pair to spare (see gemm_plan).
// Operand tile loads through a buffer descriptor based at the tile's first element: rows outside the matrix and k beyond
// the split's end are parked past the descriptor's range (the hardware returns 0), so the loop holds no exec-masked
// branches and no 64-bit address arithmetic; the chunk advance rides in the scalar offset.
template <bool KCONTIG>
struct BigTileLoader {
  __amdgpu_buffer_rsrc_t rsrc;
  int voff[2], kg[2], kstep;
  bool vec;
  static constexpr int PARK = (int)0x80000000;
  __device__ __forceinline__ void init(const float* P, long rs, long ks, int row0, int nrows, int K, int kbeg, bool vec_, int tid) {
    const size_t total = ((size_t)(nrows - 1) * rs + (size_t)(K - 1) * ks + 1) * sizeof(float);
    const size_t used = ((size_t)row0 * rs + (size_t)kbeg * ks) * sizeof(float);
    const size_t left = total - used;
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P + (size_t)row0 * rs + (size_t)kbeg * ks), 0,
                                             (int)(left < 0x7FFFF000ul ? left : 0x7FFFF000ul), 0x00020000);
    kstep = (int)(ks * 4); vec = vec_;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + 256 * u;
      const int r = KCONTIG ? e >> 2 : e & 127, g = KCONTIG ? e & 3 : e >> 7;
      kg[u] = 8 * g;
      voff[u] = row0 + r < nrows ? (int)((long)r * rs * 4 + (long)(8 * g) * ks * 4) : PARK;
    }
  }
  // kc = k0 - kbeg (first k of the chunk relative to the descriptor base), kleft = kend - k0
  __device__ __forceinline__ void load(int kc, int kleft, float (&v)[2][8]) const {
    const int soff = kc * kstep;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (KCONTIG && vec) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int off = kg[u] + 4 * i < kleft ? voff[u] + 16 * i : PARK;
          const uint4 t = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, soff, 0));
          v[u][4 * i] = __uint_as_float(t.x); v[u][4 * i + 1] = __uint_as_float(t.y);
          v[u][4 * i + 2] = __uint_as_float(t.z); v[u][4 * i + 3] = __uint_as_float(t.w);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int off = kg[u] + i < kleft ? voff[u] + i * kstep : PARK;
          v[u][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, soff, 0));
        }
      }
    }
  }
};
// Row stride (16-byte vectors) of a k-group plane of the 128-tile operand images.  K-contiguous staging stores, per 8-lane group of a ds_write_b128, the four
// k-groups of two consecutive rows: vectors g * BIG_RS + r, g = 0..3, r = 0, 1.  They fall on distinct 16-byte slots of the 128-byte bank row iff BIG_RS = 2
// (mod 8): 130.  (132 until round 6 - "the four k-groups of a row in distinct banks" held for g = 0, 1 only: slots 0, 4, 0, 4 - and the counters showed
// 20-31 % of these kernels' LDS cycles as bank conflicts, profiles/r06_pmc_gemm_cfg3.txt.)  The MFMA operand reads are 32 consecutive vectors: any stride.
constexpr int BIG_RS = 130;
template <bool KCONTIG>
__device__ __forceinline__ void tile_store8_big(uint4* T, float sc, const float (&v)[2][8], int tid) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = tid + 256 * u;
    const int r = KCONTIG ? e >> 2 : e & 127, g = KCONTIG ? e & 3 : e >> 7;
    uint4 t0, t1;
    split8_f16(v[u], sc, t0, t1);
    T[(0 * 4 + g) * BIG_RS + r] = t0; T[(1 * 4 + g) * BIG_RS + r] = t1;
  }
}

template <bool AK, bool BK>
__global__ __launch_bounds__(256, 2) void gemm_f16x3_big_kernel(GemmArgs a) {
  __shared__ uint4 As[2 * 2 * 4 * BIG_RS];
  __shared__ uint4 Bs[2 * 2 * 4 * BIG_RS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
  const int kbeg = blockIdx.z * a.klen;
  const int kend = min(a.K, kbeg + a.klen);
  const bool avec = AK && (a.rsA & 3) == 0 && (kbeg & 3) == 0 && (a.K & 3) == 0 && ((uintptr_t)a.A & 15) == 0;
  const bool bvec = BK && (a.rsB & 3) == 0 && (kbeg & 3) == 0 && (a.K & 3) == 0 && ((uintptr_t)a.Bm & 15) == 0;
  const int ka = f16_scale_exp(absmax_read(a.amax_a)), kb = f16_scale_exp(absmax_read(a.amax_b));
  const float sa = pow2f(ka), sb = pow2f(kb);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // Two LDS images: chunk k + 1 is split and stored while chunk k is multiplied, one barrier per chunk; the global loads of
  // chunk k + 2 are in flight across that whole period.
  float av[2][8], bv[2][8];
  BigTileLoader<AK> la; la.init(a.A, a.rsA, a.ksA, m0, a.M, a.K, kbeg, avec, tid);
  BigTileLoader<BK> lb; lb.init(a.Bm, a.rsB, a.ksB, n0, a.N, a.K, kbeg, bvec, tid);
  la.load(0, kend - kbeg, av); lb.load(0, kend - kbeg, bv);
  tile_store8_big<AK>(As, sa, av, tid);
  tile_store8_big<BK>(Bs, sb, bv, tid);
  if (kbeg + 32 < kend) { la.load(32, kend - kbeg - 32, av); lb.load(32, kend - kbeg - 32, bv); }
  __syncthreads();
  int cur = 0;
  for (int k0 = kbeg; k0 < kend; k0 += 32, cur ^= 1) {
    const uint4* Ac = As + cur * (2 * 4 * BIG_RS);
    const uint4* Bc = Bs + cur * (2 * 4 * BIG_RS);
    if (k0 + 32 < kend) {
      tile_store8_big<AK>(As + (cur ^ 1) * (2 * 4 * BIG_RS), sa, av, tid);
      tile_store8_big<BK>(Bs + (cur ^ 1) * (2 * 4 * BIG_RS), sb, bv, tid);
      if (k0 + 64 < kend) { la.load(k0 + 64 - kbeg, kend - k0 - 64, av); lb.load(k0 + 64 - kbeg, kend - k0 - 64, bv); }
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 x[2][2], y[2][2];                          // [block][term]
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          x[i][t] = Ac[(t * 4 + 2 * kk + h) * BIG_RS + wm * 64 + i * 32 + l31];
          y[i][t] = Bc[(t * 4 + 2 * kk + h) * BIG_RS + wn * 64 + i * 32 + l31];
        }
      // term-major: consecutive MFMAs go to different accumulators (the per-accumulator order lo*hi, hi*lo, hi*hi is the 64-tile kernel's)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x[i][1]), __builtin_bit_cast(f16x8, y[j][0]), acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x[i][0]), __builtin_bit_cast(f16x8, y[j][1]), acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x[i][0]), __builtin_bit_cast(f16x8, y[j][0]), acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  float omax = 0.f;
  float* stg = reinterpret_cast<float*>(As) + wave * (32 * 36);      // the operand images are dead (the loop ends on a barrier): 4.6 KB of staging per wave
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int mb0 = m0 + wm * 64 + i * 32, nb0 = n0 + wn * 64 + j * 32;
      if (gemm_block_vec_ok(a, mb0, nb0)) gemm_store_block_vec(a, acc[i][j], mb0, nb0, -(ka + kb), omax, stg, lane);
      else gemm_store_block(a, acc[i][j], mb0 + 4 * h, nb0 + l31, -(ka + kb), omax);
    }
  if (a.amax_out && a.nsplit == 1) absmax_commit(omax, a.amax_out);
}

__global__ void gemm_splitk_reduce_kernel(const float* __restrict__ slab, float* __restrict__ C, const float* __restrict__ bias,
                                          long ldc, int M, int N, int nsplit, int accumulate) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= (long)M * N) return;
  const int n = (int)(i % N); const int m = (int)(i / N);
  float s = 0.f;
  int z = 0;
  for (; z + 8 <= nsplit; z += 8) {          // eight loads in flight per thread, added in split order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = slab[(size_t)(z + u) * M * N + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; z < nsplit; ++z) s += slab[(size_t)z * M * N + i];
  s += bias ? bias[n] : 0.f;
  float* c = C + (long)m * ldc + n;
  *c = accumulate ? *c + s : s;
}

static void gemm_plan(int M, int N, int K, int& nsplit, int& klen, int tile = 64) {
  const long tiles = (long)((M + tile - 1) / tile) * ((N + tile - 1) / tile);
  nsplit = 1;
  if (tiles < 256 && K >= 256) {
    // (Round 6, profiles/r06_gemm_plan_sweep.txt: R.fc1's three GEMMs take 26-28 us at cfg2 and 86 us at cfg3 under EVERY plan - 256 ... 2048 workgroups, K runs of
    //  64 ... 512, 64- or 128-wide tiles - and an XCD-aware tile order that keeps a split's tiles on one L2 changed nothing either (r06_ab_gemm_xcd_order.txt):
    //  the kernels are bound by splitting their operands while staging, 9-12 VALU instructions per MFMA on the counters (r06_pmc_gemm_cfg3.txt), not by the plan.)
    static const int wgs_want = GR_KNOB("GR_GEMM_WGS", 512), min_klen = GR_KNOB("GR_GEMM_MIN_KLEN", 128);      // (ablation build: workgroups aimed for, shortest K run per split)
    long want = (wgs_want + tiles - 1) / tiles;
    const long maxs = K / min_klen;
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    nsplit = (int)want;
  }
  klen = round_up((K + nsplit - 1) / nsplit, 32);
  nsplit = (K + klen - 1) / klen;
}

size_t gemm_workspace_bytes(int M, int N, int K) {
  // the larger of the two tilings' split counts (launch_gemm picks the tiling from the operand mode)
  int ns, kl, nb, kb; gemm_plan(M, N, K, ns, kl); gemm_plan(M, N, K, nb, kb, 128);
  if (nb > ns) ns = nb;
  return ns > 1 ? sizeof(float) * (size_t)ns * M * N : 0;
}

bool gemm_epilogue_possible(int M, int N, int K) { int ns, kl; gemm_plan(M, N, K, ns, kl); return ns == 1; }

void launch_gemm(const float* A, long rsA, long ksA, const float* Bm, long rsB, long ksB,
                 float* C, long ldc, const float* bias, bool accumulate, int M, int N, int K,
                 void* workspace, hipStream_t s, const ConvEpilogue* ep, unsigned* amax_out,
                 const unsigned* amax_a, const unsigned* amax_b) {
  GemmArgs a{};
  if (ep) { a.ep = *ep; a.has_ep = 1; }
  static const int vec_store = GR_KNOB_SET("GR_GEMM_DWORD_STORES") ? 0 : 1;
  a.vec_store = vec_store;
  a.amax_out = amax_out; a.amax_a = amax_a; a.amax_b = amax_b;
  const bool f16 = amax_a != nullptr && amax_b != nullptr;
  static const bool big_on = !GR_KNOB_SET("GR_GEMM_SMALL_TILES");
  bool big = f16 && big_on && M >= 128 && N >= 128;
  a.A = A; a.Bm = Bm; a.C = C; a.slab = reinterpret_cast<float*>(workspace); a.bias = bias;
  a.rsA = rsA; a.ksA = ksA; a.rsB = rsB; a.ksB = ksB; a.ldc = ldc;
  a.M = M; a.N = N; a.K = K; a.accumulate = accumulate ? 1 : 0;
  gemm_plan(M, N, K, a.nsplit, a.klen, big ? 128 : 64);
  // the 128-tile kernel addresses a tile with 32-bit offsets from its first element
  if (big && ((127.0 * rsA + (a.klen + 32.0) * ksA) * 4 >= 0x7FFFF000 || (127.0 * rsB + (a.klen + 32.0) * ksB) * 4 >= 0x7FFFF000)) {
    big = false;
    gemm_plan(M, N, K, a.nsplit, a.klen, 64);
  }
  if (big && a.nsplit > 1 && gemm_epilogue_possible(M, N, K)) {
    // callers fuse epilogues whenever the 64-tile plan keeps K whole: stay on that plan then
    big = false;
    gemm_plan(M, N, K, a.nsplit, a.klen, 64);
  }
  const int T = big ? 128 : 64;
  dim3 grid((N + T - 1) / T, (M + T - 1) / T, a.nsplit);
  const bool ak = ksA == 1, bk = ksB == 1;
  {
  KtScope kt(big ? "gemm_f16x3_big_kernel" : (f16 ? "gemm_f16x3_kernel" : "gemm_mfma_kernel"), 2.0 * M * N * (double)K, 4.0 * ((double)M * K + (double)N * K + (double)M * N), s);
  if (big) {
    if (ak && bk) hipLaunchKernelGGL((gemm_f16x3_big_kernel<true, true>), grid, dim3(256), 0, s, a);
    else if (ak && !bk) hipLaunchKernelGGL((gemm_f16x3_big_kernel<true, false>), grid, dim3(256), 0, s, a);
    else if (!ak && bk) hipLaunchKernelGGL((gemm_f16x3_big_kernel<false, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((gemm_f16x3_big_kernel<false, false>), grid, dim3(256), 0, s, a);
  }
  else if (f16) {
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

// search.hip — recovered-noise cosine-similarity search (reference apply_r.lua:265-282 createImages loop,
// apply_r.lua:396-400 cosineSimilarity -> nn.CosineDistance) as three HBM-bound kernels:
//   1. needle_prep:   gather the Q needle rows, their squared norms (w22)
//   2. cos_keys:      one pass over emb[N][d]; per row the Q scores in the reference's exact op order
//                     (fp32 products, fp64 -- or fp32 -- sequential row sums, fp32 reciprocal/sqrt/mul),
//                     emitted as 64-bit sort keys  (orderable(score) << 32) | ~index
//   3. topk_pass:     per 2048-key chunk a bitonic sort in LDS keeps the k largest keys; repeated until one chunk is
//                     left.  Largest key first == (score desc, index asc): the tie order the oracle defines
//                     (the reference's table.sort is unstable, apply_r.lua:275).
// Large tables (N >= FILTER_MIN_ROWS) never write the N x Q keys: a strided SAMPLE of SAMPLE_ROWS rows is scored and
// sorted first; its k-th largest key is a lower bound of the true k-th largest key (the sample is a subset), so the one
// pass over emb keeps only keys >= that bound - about N * k / SAMPLE_ROWS of them per needle, each workgroup writing into
// its own SLOT entries - and one selection kernel (radix-select of the k-th score, then a sort of the few keys at or above it) finishes.
// Same scores, same keys, same result, bit for bit; a list that overflows (adversarial order) raises a status word and the
// caller reruns the unfiltered path.
#include "kernels.h"
#include <type_traits>

namespace gr {

// one LDS-DMA wave-instruction (as in conv.hip): 64 lanes x 16 bytes land at lds_dst + 16 * lane, lane l fetching rsrc[voff_l + soff]; lanes
// whose offset lies past the descriptor's range write zeros
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, uint4* lds_dst, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
#else
  (void)rsrc; (void)lds_dst; (void)voff; (void)soff;
#endif
}

constexpr int QG = 8;        // needles scored per pass of cos_keys (register accumulators)
constexpr int ROWS = 256;    // rows per workgroup
constexpr int CHUNK = 2048;  // keys per top-k workgroup
constexpr long FILTER_MIN_ROWS = 1 << 17;   // below this the unfiltered path is a handful of microseconds anyway
constexpr int SAMPLE_ROWS = 16384;          // rows scored ahead for the filter bound
constexpr int SLOT = 32;                    // candidate keys a workgroup (ROWS rows) may keep per needle: expected ROWS * k / SAMPLE_ROWS = 0.8

template <bool ACCF>
__global__ void needle_prep_kernel(const float* __restrict__ emb, int d, const long* __restrict__ rows, int Q,
                                   float* __restrict__ needles, float* __restrict__ w22, unsigned* __restrict__ counts,
                                   unsigned* __restrict__ status) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q == 0 && status) *status = 0u;
  if (q >= Q) return;
  if (counts) counts[q] = 0u;
  const float* a = emb + rows[q] * (long)d;
  typename std::conditional<ACCF, float, double>::type s = 0;
  for (int i = 0; i < d; ++i) { const float v = a[i]; needles[(long)q * d + i] = v; s += v * v; }
  float w = (float)s;
  w = w + 1e-12f;
  w22[q] = 1.f / w;
}

__device__ __forceinline__ uint32_t orderable(float f) {
  f = f + 0.f;  // -0 -> +0 so that equal scores compare equal
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float unorderable(uint32_t u) {
  const uint32_t b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  return __uint_as_float(b);
}

// MODE 0: keys[q][j] for every row j.  MODE 1 (sample): block rows are i = 0 .. N - 1 of the strided sample, row j = i * stride;
// keys[q][i] carry the true index j.  MODE 2 (filter): keys >= bound[q] go to the workgroup's own entries keys[q][workgroup][SLOT]
// (counts[q][workgroup] = how many wanted in; the selection kernel checks it against SLOT).  NQ needles per pass, compile-time: their values for the
// staged columns sit in LDS next to the row tile and are read as broadcast float4s (a scalar load per needle and column
// inside the loop serialised on the scalar cache: 259 us at cfg5 against 5x less now).  Columns past d are staged as zeros on
// both sides: they add exact zeros to the sums, so every chunk runs the full unrolled DC columns.
// DC columns per chunk (32, or 20 when that divides d and 32 does not - d = 100 of cfg5: five full chunks instead of three and
// one of 4 real columns); tile row stride TS = DC + 4 or + 8 floats: rows stay 16-byte aligned and TS / 4 is odd, so the float4
// reads of 16 lanes hit 64 distinct banks
template <bool ACCF, int MODE, int NQ, int DC>
__global__ __launch_bounds__(ROWS) void cos_keys_kernel(const float* __restrict__ emb, long N, int d,
                                                         const float* __restrict__ needles, const float* __restrict__ w22,
                                                         int q0, unsigned long long* __restrict__ keys, long stride,
                                                         const unsigned long long* __restrict__ bound, unsigned* __restrict__ counts, int dbg_) {
  const int dbg = GR_DBG(dbg_);
  typedef typename std::conditional<ACCF, float, double>::type acc_t;
  constexpr int TS = ((DC / 4) & 1) ? DC + 8 : DC + 4;
  __shared__ __attribute__((aligned(16))) float tile[ROWS * TS];
  __shared__ __attribute__((aligned(16))) float ndt[QG * DC];
  __shared__ unsigned lds_cnt[QG];
  const int tid = threadIdx.x;
  if (MODE == 2 && tid < QG) lds_cnt[tid] = 0u;          // (published by the first barrier of the column loop)
  const long r0 = (long)blockIdx.x * ROWS;
  const bool vec = (d & 3) == 0;                 // rows are 16-byte aligned: stage with float4 loads
  acc_t s1[NQ], s3 = 0;
#pragma unroll
  for (int q = 0; q < NQ; ++q) s1[q] = 0;
  // 8 lanes cover one 128-byte row segment.  All DC / 4 loads of a thread are issued back to back (out-of-range slots read a
  // valid address and are zeroed afterwards: a branch around each load made the compiler wait for every load before issuing
  // the next), and the NEXT chunk's loads are issued before the arithmetic on the current one, so HBM latency hides behind it.
  float4 v[DC / 4]; float nreg = 0.f;
  auto fetch = [&](int c0) {
    const int dc = min(DC, d - c0);
#pragma unroll
    for (int i = 0; i < DC / 4; ++i) {
      const int e = tid + i * ROWS, r = e / (DC / 4), c = (e - r * (DC / 4)) * 4;
      const bool ok = r0 + r < N && c < dc && !(dbg & 1);
      const long row = ok ? (MODE == 1 ? (r0 + r) * stride : r0 + r) : 0;
      v[i] = *reinterpret_cast<const float4*>(emb + row * (long)d + (ok ? c0 + c : 0));
      if (!ok) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid < NQ * DC) { const int q = tid / DC, c = tid - q * DC; nreg = c < dc ? needles[(long)(q0 + q) * d + c0 + c] : 0.f; }
  };
  if (vec) fetch(0);
  for (int c0 = 0; c0 < d; c0 += DC) {
    const int dc = min(DC, d - c0);
    if (vec) {
#pragma unroll
      for (int i = 0; i < DC / 4; ++i) {
        const int e = tid + i * ROWS, r = e / (DC / 4), c = (e - r * (DC / 4)) * 4;
        *reinterpret_cast<float4*>(tile + r * TS + c) = v[i];
      }
      if (tid < NQ * DC) ndt[tid] = nreg;
    } else {
      for (int e = tid; e < ROWS * DC; e += ROWS) {
        const int r = e / DC, c = e - r * DC;
        float x = 0.f;
        if (r0 + r < N && c < dc) x = emb[(MODE == 1 ? (r0 + r) * stride : r0 + r) * (long)d + c0 + c];
        tile[r * TS + c] = x;
      }
      if (tid < NQ * DC) { const int q = tid / DC, c = tid - q * DC; ndt[tid] = c < dc ? needles[(long)(q0 + q) * d + c0 + c] : 0.f; }
    }
    __syncthreads();
    if (vec && c0 + DC < d) fetch(c0 + DC);
    const float4* row4 = reinterpret_cast<const float4*>(tile + tid * TS);
    const float4* nd4 = reinterpret_cast<const float4*>(ndt);
    if (!(dbg & 2))
#pragma unroll
    for (int c4 = 0; c4 < DC / 4; ++c4) {
      const float4 bv = row4[c4];
      const float b[4] = {bv.x, bv.y, bv.z, bv.w};
      float nv[NQ][4];
#pragma unroll
      for (int q = 0; q < NQ; ++q) { const float4 t = nd4[q * (DC / 4) + c4]; nv[q][0] = t.x; nv[q][1] = t.y; nv[q][2] = t.z; nv[q][3] = t.w; }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {            // column order inside the chunk = the reference's summation order
        s3 += (acc_t)(b[jj] * b[jj]);
#pragma unroll
        for (int q = 0; q < NQ; ++q) s1[q] += (acc_t)(nv[q][jj] * b[jj]);
      }
    }
    __syncthreads();
  }
  const long j = r0 + tid;
  if (j < N && !(dbg & 4)) {
    float w32 = (float)s3;
    w32 = w32 + 1e-12f;
    w32 = 1.f / w32;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      float w = w22[q0 + q] * w32;
      w = sqrtf(w);
      const float sc = (float)s1[q] * w;
      const long row = MODE == 1 ? j * stride : j;
      const unsigned long long key = ((unsigned long long)orderable(sc) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)row);
      if (MODE == 2) {
        // candidates go to this workgroup's own SLOT entries of needle q (position from an LDS counter): a global counter per
        // needle serialised 15 000 returning atomics on five addresses - 330 of the kernel's 460 us at cfg5
        if (key >= bound[q0 + q]) {
          const unsigned pos = atomicAdd(&lds_cnt[q], 1u);
          if (pos < (unsigned)SLOT) keys[((long)(q0 + q) * gridDim.x + blockIdx.x) * SLOT + pos] = key;
        }
      } else keys[(long)(q0 + q) * N + j] = key;
    }
  }
  if (MODE == 2) {
    __syncthreads();
    if (tid < NQ) counts[(long)(q0 + tid) * gridDim.x + blockIdx.x] = (dbg & 4) ? 0u : lds_cnt[tid];     // every workgroup writes its count: no fill needed
  }
}

// One workgroup (1024 threads) per needle, over either the dense sample keys[q][n] (BOUND_ONLY) or the filter's per-workgroup
// entries keys[q][n][SLOT] with counts[q][n].  Every thread takes the LARGEST of the keys it walks; the k-th largest of those
// 1024 maxima is a lower bound of the k-th largest key overall - the k maxima above it are k distinct keys.  BOUND_ONLY: (that key's score << 32) is the filter
// bound.  Otherwise the keys at or above that maximum - k of them plus the few the bound lets through - are gathered, sorted in
// LDS and the first k decoded into (index, score).  A workgroup that wanted more than SLOT entries, or more than CHUNK gathered
// keys, raises *status: the caller reruns the unfiltered path.
template <bool BOUND_ONLY, typename F>
__device__ __forceinline__ void for_each_key(const unsigned long long* __restrict__ src, long n, const unsigned* __restrict__ cnt, F f) {
  if (BOUND_ONLY) {
    for (long i0 = threadIdx.x; i0 < n; i0 += 8 * 1024) {       // 8 independent loads in flight per thread
      unsigned long long kk[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) kk[u] = i0 + u * 1024 < n ? src[i0 + u * 1024] : 0ull;
#pragma unroll
      for (int u = 0; u < 8; ++u) if (i0 + u * 1024 < n) f(kk[u]);
    }
  } else {
    for (long g0 = threadIdx.x; g0 < n; g0 += 4 * 1024) {       // n workgroups' entries; counts of 4 workgroups first, then their keys
      unsigned c[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) c[u] = g0 + u * 1024 < n ? min(cnt[g0 + u * 1024], (unsigned)SLOT) : 0u;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        for (unsigned e = 0; e < c[u]; ++e) f(src[(g0 + u * 1024) * SLOT + e]);
    }
  }
}
template <bool BOUND_ONLY>
__global__ __launch_bounds__(1024) void topk_select_kernel(const unsigned long long* __restrict__ keys, long n,
                                                           const unsigned* __restrict__ counts, int k,
                                                           unsigned long long* __restrict__ bound_out,
                                                           long* __restrict__ idx, float* __restrict__ score, unsigned* __restrict__ status) {
  __shared__ __attribute__((aligned(16))) unsigned long long list[CHUNK];       // first the 1024 thread maxima, then the gathered keys
  __shared__ unsigned long long sh_bound;
  __shared__ unsigned list_n, over;
  const int q = blockIdx.x, tid = threadIdx.x;
  const unsigned long long* src = keys + (long)q * n * (BOUND_ONLY ? 1 : SLOT);
  const unsigned* cnt = BOUND_ONLY ? nullptr : counts + (long)q * n;
  if (tid == 0) { sh_bound = 0ull; list_n = 0u; over = 0u; }
  __syncthreads();
  unsigned long long mine = 0ull;
  if (!BOUND_ONLY) {
    unsigned o = 0u;
    for (long g = tid; g < n; g += 1024) o |= cnt[g] > (unsigned)SLOT ? 1u : 0u;
    if (o) over = 1u;
  }
  for_each_key<BOUND_ONLY>(src, n, cnt, [&](unsigned long long key) { mine = key > mine ? key : mine; });
  list[tid] = mine;
  __syncthreads();
  if (!BOUND_ONLY && over) { if (tid == 0 && status) *status = 1u; return; }
  // k-th largest of the 1024 maxima: bitonic sort in LDS, descending (ranking each maximum against the other 1023 costs 8 MB
  // of LDS reads per needle - 30 us on one CU; the sort moves 0.9 MB).  Empty threads hold 0: with fewer than k non-empty
  // threads entry k - 1 is 0 and everything passes.
  for (int size = 2; size <= 1024; size <<= 1)
    for (int st = size >> 1; st > 0; st >>= 1) {
      if (tid < 512) {
        const int lo = ((tid / st) * st * 2) + (tid % st), hi = lo + st;
        const bool desc = ((lo & size) == 0);
        const unsigned long long a = list[lo], b = list[hi];
        if ((a < b) == desc) { list[lo] = b; list[hi] = a; }
      }
      __syncthreads();
    }
  if (tid == 0) sh_bound = list[k - 1];
  __syncthreads();
  const unsigned long long bnd = sh_bound;
  if (BOUND_ONLY) { if (tid == 0) bound_out[q] = bnd & 0xFFFFFFFF00000000ull; return; }
  __syncthreads();                                  // every thread has read the maxima: the list is reused for the gathered keys
  for_each_key<BOUND_ONLY>(src, n, cnt, [&](unsigned long long key) {
    if (key >= bnd) { const unsigned pos = atomicAdd(&list_n, 1u); if (pos < (unsigned)CHUNK) list[pos] = key; }
  });
  __syncthreads();
  const unsigned m = list_n;
  if (m > (unsigned)CHUNK) { if (tid == 0 && status) *status = 1u; return; }
  int P = 64; while (P < (int)m) P <<= 1;
  for (int i = tid; i < P; i += 1024) if (i >= (int)m) list[i] = 0ull;
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1)
    for (int st = size >> 1; st > 0; st >>= 1) {
      if (tid < P / 2) {
        const int lo = ((tid / st) * st * 2) + (tid % st), hi = lo + st;
        const bool desc = ((lo & size) == 0);
        const unsigned long long a = list[lo], b = list[hi];
        if ((a < b) == desc) { list[lo] = b; list[hi] = a; }
      }
      __syncthreads();
    }
  for (int r = tid; r < k; r += 1024) {
    const unsigned long long key = list[r];
    idx[(long)q * k + r] = (long)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull));
    if (score) score[(long)q * k + r] = unorderable((uint32_t)(key >> 32));
  }
}

// unfiltered path.  keys_in: [Q][n_in] ; keys_out: [Q][nchunks*k]
__global__ __launch_bounds__(1024) void topk_pass_kernel(const unsigned long long* __restrict__ kin, long n_in, int k,
                                                         unsigned long long* __restrict__ kout, long n_out) {
  __shared__ unsigned long long sk[CHUNK];
  const int q = blockIdx.y; const long c0 = (long)blockIdx.x * CHUNK;
  const unsigned long long* src = kin + (long)q * n_in;
  for (int i = threadIdx.x; i < CHUNK; i += blockDim.x) sk[i] = (c0 + i < n_in) ? src[c0 + i] : 0ull;
  __syncthreads();
  for (int size = 2; size <= CHUNK; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const int t = threadIdx.x;                  // 1024 threads, one compare-exchange each
      const int lo = ((t / stride) * stride * 2) + (t % stride), hi = lo + stride;
      const bool desc = ((lo & size) == 0);       // descending blocks first -> whole array descending at the end
      const unsigned long long a = sk[lo], b = sk[hi];
      if ((a < b) == desc) { sk[lo] = b; sk[hi] = a; }
      __syncthreads();
    }
  unsigned long long* dst = kout + (long)q * n_out + (long)blockIdx.x * k;
  for (int i = threadIdx.x; i < k; i += blockDim.x) dst[i] = sk[i];
}

__global__ void topk_decode_kernel(const unsigned long long* __restrict__ keys, long stride, int Q, int k,
                                   long* __restrict__ idx, float* __restrict__ score) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Q * k) return;
  const int q = i / k, r = i - q * k;
  const unsigned long long key = keys[(long)q * stride + r];
  idx[i] = (long)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull));
  if (score) score[i] = unorderable((uint32_t)(key >> 32));
}

// ------------------------------------------------------------------ many needles at once: candidates from one MFMA GEMM, exact re-score
// For Q >= BATCH_MIN_Q needles the exact pass above is compute-bound (N * d * Q fp32 products with fp64 sums on the VALU: 16 ms
// for 1024 needles over 10^6 x 100).  The batched path (north_star: "one MFMA GEMM + top-k") keeps the RESULT exact and uses the
// matrix pipe only to decide which rows can matter:
//   1. approximate cosines  emb x needles^T  on v_mfma_f32_32x32x16_f16 (round 6; bf16 until then): both operands rounded to fp16 while staged, row
//      norms from the fp16 values.  The approximate score is the cosine of the ROUNDED vectors a^ = a + da.  fp16 keeps 11 significant bits (round to
//      nearest: relative error 2^-11 per NORMAL element) and has a narrow exponent range, so a vector is only taken when its squared norm (of the rounded
//      values) lies in [2^-10, 65504^2]: no element overflows (|a_i| <= |a|), and elements in the subnormal range are off by at most 2^-25 each, together
//      sqrt(128) 2^-25 = 2^-21.5 <= 2^-16.5 |a|.  Then |da| <= u |a| with u = 2^-11 + 2^-16.5, each vector turns by at most asin(u) and
//      |approximate - exact| <= 2 u + O(u^2) < 2^-10 + 2^-15; the products of two fp16 values are exact in fp32, their accumulation over d <= 128 columns
//      adds <= 128 x 2^-24 = 2^-17, the fp32 steps behind the sums and the 1e-12 in the denominators ~1e-7.  BERR = 2^-10 + 2^-13 bounds all of it with
//      a margin of 8e-5.  A row outside the norm range (or a zero row) raises the overflow status like an overflowing list - the call reruns
//      unbatched, exact as ever - and a needle outside it gets the threshold -inf, which overflows every list: same rerun.  bf16 (u = 2^-8) needed
//      BERR = 2^-7 + 2^-10: the cuts are eight times tighter now, an eighth of the candidates reach the queues, the lists and the exact re-score
//      (same box, 1024 needles over 1 M x 100: kernels 0.603 -> 0.562 ms, profiles/r06_ab_search_fp16_candidates.txt);
//   2. a strided sample of SAMPLE_ROWS rows first: tau_q = (k-th largest approximate sample score) - 2 BERR is a lower bound
//      of every approximate score whose exact score can reach the true k-th largest one;
//   3. the pass over the table keeps (row, approximate score) pairs >= tau_q, each workgroup in its own BSLOT entries per needle;
//   4. per needle: the k-th largest approximate candidate score minus 2 BERR cuts the ~3000 candidates down to ~k + a few,
//      those are re-scored EXACTLY (the op order of cos_keys_kernel: fp32 products, sequential fp64 sums, same w22 / w32
//      arithmetic), turned into the same 64-bit keys, sorted, decoded.
// Every row whose exact score is among the k best passes both cuts, so indices and scores are bit-identical to the unbatched
// search; an overflowing entry list raises the status word and the caller reruns the unbatched path.
constexpr int BATCH_MIN_Q = 32;
constexpr int AQ_MAX = 8;             // needles of the small path (cos_approx_kernel below): their rows travel by value
struct SmallQ { long rows[AQ_MAX]; };
constexpr int BQ_MAX = 2048;          // needles per call of the batched path (LDS counters)
constexpr long BSAMPLE_ROWS = 65536;  // the batched path's sample: 256 workgroups of 256 rows, ONE value per (workgroup, needle) - their maximum - leaves the kernel
constexpr int BSLOT = 16;             // (row, score) entries per workgroup (256 rows) and needle: expected 1.2 at cfg5, P(> 16) ~ 1e-14
constexpr int BD_MAX = 128;           // widest row the batched kernel stages whole
#define GR_BERR 0.0010986328125f   /* 2^-10 + 2^-13: see the bound above */
#define GR_BNRM_MIN 9.765625e-4f   /* 2^-10: smallest squared norm a vector may have on the fp16 candidate pass */
#define GR_BNRM_MAX 4.2907e9f      /* < 65504^2: largest */
typedef _Float16 f16x8s __attribute__((ext_vector_type(8)));
typedef float f32x16s __attribute__((ext_vector_type(16)));
// (the names keep their round-2 spelling: the staged 16-bit image was bf16 until round 6; it is IEEE fp16 now, round to nearest even)
__device__ __forceinline__ unsigned short to_bf16(float x) { const _Float16 h = (_Float16)x; return __builtin_bit_cast(unsigned short, h); }
__device__ __forceinline__ float from_bf16(unsigned short u) { return (float)__builtin_bit_cast(_Float16, u); }

// lane i of a 16-lane DPP row receives lane i - n's value (row_shr:n = 0x110 + n); lanes without a source keep their own
template <int CTRL>
__device__ __forceinline__ float dpp_row_shr(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
// needles as the MFMA kernel stages them: bf16 rows [Qpad64][KS] (zero columns past d, zero rows past Q), sqrt(w22), tau = +inf past Q
__global__ void needles_bf16_kernel(const float* __restrict__ needles, const float* __restrict__ w22, int Q, int Qpad, int d, int KS,
                                    unsigned short* __restrict__ nb16, float* __restrict__ sw22s, float* __restrict__ tau) {
  const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (e < (long)Qpad * KS) {
    const int q = (int)(e / KS), c = (int)(e - (long)q * KS);
    nb16[e] = to_bf16((q < Q && c < d) ? needles[(long)q * d + c] : 0.f);
  }
  if (e < Qpad) { sw22s[e] = e < Q ? sqrtf(w22[e]) : 0.f; if (e >= Q) tau[e] = INFINITY; }
}

// MODE 0: rows are i * stride (the sample), scores out[q][i].  MODE 1: every row, candidates >= tau[q].
// Workgroup = 256 rows (4 waves x 64) against ALL needles, 64 at a time.  The row tile goes through LDS once (fp32 -> bf16, the
// MFMA B-operand layout, row norms) and then lives in registers (2 row blocks x NK k-steps of 16-byte vectors per lane); the LDS
// it used holds the needle tiles from then on, double-buffered: the next tile's vectors are requested before the current tile's
// MFMAs (its bf16 image is prepared once by needles_bf16_kernel, so staging is plain 16-byte copies).
template <int MODE, int NK>
__global__ __launch_bounds__(256, 2) void cos_mfma_kernel(const float* __restrict__ emb, long N, int d, long stride,
                                                          const unsigned short* __restrict__ nb16, const float* __restrict__ sw22s, int Q,
                                                          const float* __restrict__ tau, float* __restrict__ out,
                                                          unsigned* __restrict__ cand_idx, float* __restrict__ cand_sc, unsigned* __restrict__ counts, int qcap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int KP = NK * 16, KS = KP + 8;                      // bf16 elements per staged row; KS / 8 is odd: conflict-free 16-byte reads
  constexpr int TV = 64 * KS / 8, NTV = (TV + 255) / 256;         // 16-byte vectors of one needle tile, per thread
  unsigned short* rowsB = reinterpret_cast<unsigned short*>(smem);              // [256][KS], later two needle tiles [2][64][KS]
  float* sw32s = reinterpret_cast<float*>(rowsB + 256 * KS);                     // [256] sqrt(1 / (|row|^2 + 1e-12))
  float* sw22t = sw32s + 256;                                                    // [2][64]
  float* taut = sw22t + 128;                                                     // [2][64]
  unsigned* lds_cnt = reinterpret_cast<unsigned*>(taut + 128);                   // [Q] (MODE 1)
  // MODE 1: a queue of passing (needle, row, score) entries per WAVE, [4][qcap] x 8 bytes behind the counters (launcher: whatever two workgroups per CU leave, 0 = none)
  const int dbg = GR_DBG(qcap >> 16);                                            // ablation build: GR_BATCHED_DEBUG bits 1 no epilogue, 2 no MFMA, 4 no row loads (results wrong by design)
  qcap &= 0xffff;
  unsigned* wg_ovf = lds_cnt + (((Q > 128 ? Q : 128) + 1) & ~1);                  // [2]: some wave's queue overflowed
  unsigned long long* wqueue = reinterpret_cast<unsigned long long*>(wg_ovf + 2) + (size_t)(threadIdx.x >> 6) * qcap;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const long r0 = (long)blockIdx.x * 256;
  if (MODE == 1) { for (int q = tid; q < Q; q += 256) lds_cnt[q] = 0u; if (tid == 0) *wg_ovf = 0u; }
  if (MODE == 0 && tid < 128) lds_cnt[tid] = 0u;               // two slots of 64 per-needle maxima (orderable bits; 0 = below everything)
  // stage the row tile as bf16 (zero columns past d, zero rows past N): float4 loads when the rows are 16-byte aligned
  if ((d & 3) == 0) {
    constexpr int C4 = KP / 4, TOT = 256 * C4, PER = (TOT + 255) / 256, HALF = (PER + 1) / 2;
#pragma unroll
    for (int part = 0; part < 2; ++part) {
      float4 v[HALF];
#pragma unroll
      for (int u = 0; u < HALF; ++u) {
        const int e = tid + 256 * (part * HALF + u), r = e / C4, c = (e - r * C4) * 4;
        const bool ok = e < TOT && c < d && r0 + r < N && !(dbg & 4);
        const long row = ok ? (MODE == 0 ? (r0 + r) * stride : r0 + r) : 0;
        v[u] = *reinterpret_cast<const float4*>(emb + row * (long)d + (ok ? c : 0));
        if (!ok) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < HALF; ++u) {
        const int e = tid + 256 * (part * HALF + u), r = e / C4, c = (e - r * C4) * 4;
        if (e < TOT) {
          uint2 pk;
          pk.x = to_bf16(v[u].x) | (unsigned)to_bf16(v[u].y) << 16; pk.y = to_bf16(v[u].z) | (unsigned)to_bf16(v[u].w) << 16;
          *reinterpret_cast<uint2*>(rowsB + r * KS + c) = pk;
        }
      }
    }
  } else {
    for (int e0 = tid; e0 < 256 * KP; e0 += 256 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + 256 * u, r = e / KP, c = e - r * KP;
        const bool ok = e < 256 * KP && c < d && r0 + r < N;
        const long row = ok ? (MODE == 0 ? (r0 + r) * stride : r0 + r) : 0;
        v[u] = emb[row * (long)d + (ok ? c : 0)];
        if (!ok) v[u] = 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + 256 * u, r = e / KP, c = e - r * KP;
        if (e < 256 * KP) rowsB[r * KS + c] = to_bf16(v[u]);
      }
    }
  }
  __syncthreads();
  {
    float nrm = 0.f;
    const uint4* rv = reinterpret_cast<const uint4*>(rowsB + tid * KS);
#pragma unroll
    for (int c8 = 0; c8 < KP / 8; ++c8) {
      const uint4 t = rv[c8];
      const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float a = from_bf16((unsigned short)(w[j] & 0xffffu)), b = from_bf16((unsigned short)(w[j] >> 16)); nrm += a * a; nrm += b * b; }
    }
    // a row the fp16 image cannot carry within the proven bound (see the head of this section): its scale becomes NaN - it passes no threshold and enters
    // no maximum - and in the pass over the table it marks the workgroup overflowed, so that the call reruns unbatched
    const bool in_table = r0 + tid < N;
    const bool good = nrm >= GR_BNRM_MIN && nrm <= GR_BNRM_MAX;
    sw32s[tid] = (in_table && good) ? sqrtf(1.f / (nrm + 1e-12f)) : NAN;
    if (MODE == 1 && in_table && !good) *wg_ovf = 1u;             // (its zeroing at the top of the kernel is behind the staging barrier)
  }
  uint4 bop[2][NK];                                             // this lane's B operands: rows 64 wave + 32 rb + l31, k octet h of every k-step
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) bop[rb][kk] = *reinterpret_cast<const uint4*>(rowsB + (64 * wave + 32 * rb + l31) * KS + kk * 16 + 8 * h);
  __syncthreads();                                              // rows are in registers: the tile region now holds needle tiles
  float s32[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) s32[rb] = (MODE == 1 && r0 + 64 * wave + 32 * rb + l31 >= N) ? NAN : sw32s[64 * wave + 32 * rb + l31];   // (a NaN score passes no threshold)
  uint4* ndA = reinterpret_cast<uint4*>(rowsB);                 // [2][TV]
  const uint4* nsrc = reinterpret_cast<const uint4*>(nb16);
  const int ntiles_all = (Q + 63) / 64;
  // MODE 0 (the sample): gridDim.y workgroups share a row tile and walk disjoint runs of needle tiles.  One workgroup per CU walking all 16 tiles of 1024
  // needles is a chain of 16 barrier-to-barrier steps (68 us for 13 GFLOP, round 4); four per row tile re-read 26 MB of sample rows from L2 and take a quarter
  // of the steps each.  MODE 1 always walks every tile (gridDim.y = 1).
  const int nt0 = MODE == 0 ? (int)((long)blockIdx.y * ntiles_all / gridDim.y) : 0, ntiles = MODE == 0 ? (int)((long)(blockIdx.y + 1) * ntiles_all / gridDim.y) : ntiles_all;
  uint4 pre[NTV]; float pre_w = 0.f, pre_t = 0.f;
  auto fetch = [&](int nt) {
#pragma unroll
    for (int u = 0; u < NTV; ++u) { const int e = tid + 256 * u; pre[u] = e < TV ? nsrc[(long)nt * TV + e] : make_uint4(0, 0, 0, 0); }
    if (tid < 64) { pre_w = sw22s[nt * 64 + tid]; if (MODE == 1) pre_t = tau[nt * 64 + tid]; }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int u = 0; u < NTV; ++u) { const int e = tid + 256 * u; if (e < TV) ndA[buf * TV + e] = pre[u]; }
    if (tid < 64) { sw22t[buf * 64 + tid] = pre_w; if (MODE == 1) taut[buf * 64 + tid] = pre_t; }
  };
  if (nt0 < ntiles) { fetch(nt0); commit(nt0 & 1); }
  for (int nt = nt0; nt < ntiles; ++nt) {
    const int q0 = nt * 64, cur = nt & 1;
    __syncthreads();                                            // tile nt is published; tile nt - 1's buffer is free
    if (nt + 1 < ntiles) fetch(nt + 1);
    const unsigned short* at = reinterpret_cast<const unsigned short*>(ndA + cur * TV);
    f32x16s acc[2][2];                                          // [needle block][row block]
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][rb][r] = 0.f;
    if (!(dbg & 2))
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
      uint4 a[2];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) a[nb] = *reinterpret_cast<const uint4*>(at + (32 * nb + l31) * KS + kk * 16 + 8 * h);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
          acc[nb][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8s, a[nb]), __builtin_bit_cast(f16x8s, bop[rb][kk]), acc[nb][rb], 0, 0, 0);
    }
    // this lane's 32 needles of the tile are 8 runs of 4 (accumulator register r <-> needle 32 nb + 8 (r >> 2) + 4 h + (r & 3)):
    // their thresholds (MODE 1: tau / sqrt(w22), so that one multiply per value decides) or scales (MODE 0) come in as 8
    // float4s up front - one LDS read and wait per value made the epilogue 20x longer than the tile's MFMAs
    float4 pv[2][4];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) pv[nb][rq] = *reinterpret_cast<const float4*>((MODE == 1 ? taut : sw22t) + cur * 64 + 32 * nb + 8 * rq + 4 * h);
    float vmax0[2][16];                                         // MODE 0: this lane's maximum per needle over its two row blocks
    if (MODE == 0 && tid < 64 && nt > nt0) {                    // the previous tile's maxima are complete (the barrier above): out they go, slot cleared
      const int qp = (nt - 1) * 64 + tid;
      if (qp < Q) out[(long)qp * gridDim.x + blockIdx.x] = unorderable(lds_cnt[(cur ^ 1) * 64 + tid]) * sw22t[(cur ^ 1) * 64 + tid];
      lds_cnt[(cur ^ 1) * 64 + tid] = 0u;
    }
    if (MODE == 1 && (dbg & 1)) {
    } else if (MODE == 1 && qcap > 0) {
      // (Round 6, measured and removed - git history, profiles/r06_ab_search_block_prefilter.txt: testing a 16-register block as a whole - d = value - threshold
      //  by one fma per register, their maximum by v_max3, ONE vote per block, and only lanes with a hit walking their registers into LDS atomics - was
      //  bit-identical and SLOWER: main pass 415 -> 495 us for 1024 needles.  The ablation of this epilogue (r06_ablate_search_batched.txt: 167 of the pass's
      //  422 us with the fp16 bound) is therefore not its multiplies and votes as such: the queue version's work overlaps the CU partner's MFMAs, the
      //  16-deep predicated hit walk with its dependent atomics did not.)
      // Round 5.  0.5 % of the 64 x 64 values of a wave's tile pass their threshold, so ~17 of the 64 (needle block, row block, register) positions have a
      // passing lane somewhere in the wave.  Round 4 entered the hit path at each of them - an LDS atomic WITH return, a wait, an LDS read, two scattered
      // stores: ~300 cycles each, one after the other, 5000 cycles per tile against 900 for its 28 MFMAs (243 TFLOP/s = 0.098 of the bf16 peak).  Now a
      // position with a hit costs a wave vote, a prefix count and ONE fire-and-forget 8-byte LDS write into the wave's own queue (no atomic, no wait: the
      // position is scalar base + mbcnt); the queue is drained once per tile, one entry per lane, so the atomics and stores of all ~20 entries overlap.
      // An entry that does not fit the wave's queue IS dropped - and raises wg_ovf, which forces this workgroup's counts past BSLOT: the whole call then
      // reruns unbatched (gr_search_stats counts it), so the result stays exact; a queue sized for ~3x the expected hits makes that a rare path.
      unsigned qn = 0u;                                          // wave-uniform: passing values so far (may exceed qcap: see below)
      const unsigned lo_base = ((unsigned)(64 * wave + l31) << 8) | (unsigned)(4 * h);
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            // four positions (one float4 of thresholds) per scalar branch: their compares are independent VALU work, the four wave votes are OR-ed on the
            // scalar unit.  One position at a time, a vote waited for its compare and the branch for the vote: ~30 cycles per position, 64 positions per tile
            // (ablation, 1 M x 100 against 1024 needles: this epilogue was 200 us of the pass's 443; skeleton 131, row loads 34, MFMA 77).
            const float4 p4 = pv[nb][rq];
            const float pqs[4] = {p4.x, p4.y, p4.z, p4.w};
            float v[4]; unsigned long long bal[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = acc[nb][rb][4 * rq + j] * s32[rb]; bal[j] = __ballot(v[j] >= pqs[j]); }   // (the threshold is +inf past Q; rows past N carry a NaN scale: never true)
            if (__builtin_expect((bal[0] | bal[1] | bal[2] | bal[3]) != 0ull, 0)) {      // out of line: a group without a hit falls through
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (bal[j]) {
                  const unsigned pos = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(bal[j] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal[j], 0u));
                  if (v[j] >= pqs[j] && pos < (unsigned)qcap) {
                    unsigned lo = lo_base;
                    asm volatile("" : "+v"(lo));                 // (keeps the 64 per-position constants from being hoisted out of the tile loop into 64 live registers: spills)
                    wqueue[pos] = ((unsigned long long)__float_as_uint(v[j]) << 32) | (lo + (unsigned)(((32 * rb) << 8) | (32 * nb + j + 8 * rq)));
                  }
                  qn += (unsigned)__popcll(bal[j]);
                }
            }
          }
      }
      // a wave whose tile passes more values than its queue holds (never on tables the sample describes: ~20 expected, 256 slots) marks the workgroup: its
      // counts are then reported as overflowed for every needle and the call reruns on the unbatched path, exactly as for an overflowing candidate list
      if (qn > (unsigned)qcap && lane == 0) *wg_ovf = 1u;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // (LDS operations of one wave execute in order; this keeps the compiler from moving the reads up)
      const unsigned nq = qn < (unsigned)qcap ? qn : (unsigned)qcap;
      for (unsigned e = lane; e < nq; e += 64) {
        const unsigned long long ent = wqueue[e];
        const unsigned lo = (unsigned)ent, ql = lo & 63u, rowl = lo >> 8;
        const float v = __uint_as_float((unsigned)(ent >> 32));
        const int q = q0 + (int)ql;
        const unsigned p2 = atomicAdd(&lds_cnt[q], 1u);
        if (p2 < (unsigned)BSLOT) { const long at2 = ((long)q * gridDim.x + blockIdx.x) * BSLOT + p2; cand_idx[at2] = (unsigned)(r0 + rowl); cand_sc[at2] = v * sw22t[cur * 64 + ql]; }
      }
    } else {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const long i = r0 + 64 * wave + 32 * rb + l31;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ql = 32 * nb + (r & 3) + 8 * (r >> 2) + 4 * h, q = q0 + ql;
          const float4 p4 = pv[nb][r >> 2];
          const float pq = (r & 3) == 0 ? p4.x : ((r & 3) == 1 ? p4.y : ((r & 3) == 2 ? p4.z : p4.w));
          const float v = acc[nb][rb][r] * s32[rb];
          if (MODE == 0) { if (i < N && s32[rb] == s32[rb]) vmax0[nb][r] = rb == 0 ? v : fmaxf(vmax0[nb][r], v); else if (rb == 0) vmax0[nb][r] = -INFINITY; }      // (a row past the table or outside the fp16 norm range - NaN scale - enters no maximum: leaving rows out only lowers the threshold)
          else if (v >= pq) {                                   // (the threshold is +inf past Q; rows past N carry a NaN scale: never true)
            const unsigned pos = atomicAdd(&lds_cnt[q], 1u);
            if (pos < (unsigned)BSLOT) { const long at2 = ((long)q * gridDim.x + blockIdx.x) * BSLOT + pos; cand_idx[at2] = (unsigned)i; cand_sc[at2] = v * sw22t[cur * 64 + ql]; }
          }
        }
    }
    }
    if (MODE == 0) {
      // the sample leaves ONE value per (workgroup, needle): the maximum over the workgroup's 256 rows (their k-th largest over the
      // workgroups bounds the k-th largest sample score from below - k distinct rows at or above it - without writing S x Q scores).
      // Lanes with the same h hold the same needles for 32 different rows: shuffle tree over them, then one LDS max per (wave, needle).
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float m = vmax0[nb][r];
          // maximum over each 16-lane DPP row (row_shr 1, 2, 4, 8: lane 15 of the row ends with it - VALU moves, no LDS crossbar), then the
          // four row leaders add their value to the needle's LDS word (lanes 15, 31: h = 0; 47, 63: h = 1)
          m = fmaxf(m, dpp_row_shr<0x111>(m)); m = fmaxf(m, dpp_row_shr<0x112>(m)); m = fmaxf(m, dpp_row_shr<0x114>(m)); m = fmaxf(m, dpp_row_shr<0x118>(m));
          if ((lane & 15) == 15) atomicMax(&lds_cnt[cur * 64 + 32 * nb + (r & 3) + 8 * (r >> 2) + 4 * h], orderable(m));
        }
    }
    if (nt + 1 < ntiles) commit(cur ^ 1);
  }
  if (MODE == 0 && nt0 < ntiles) {
    __syncthreads();
    const int lastb = (ntiles - 1) & 1, qp = (ntiles - 1) * 64 + tid;
    if (tid < 64 && qp < Q) out[(long)qp * gridDim.x + blockIdx.x] = unorderable(lds_cnt[lastb * 64 + tid]) * sw22t[lastb * 64 + tid];
  }
  if (MODE == 1) {
    __syncthreads();
    const unsigned ovf = *wg_ovf;
    for (int q = tid; q < Q; q += 256) counts[(long)q * gridDim.x + blockIdx.x] = ovf ? (unsigned)BSLOT + 1u : lds_cnt[q];
  }
}

// k-th largest of up to 1024 per-thread maxima (orderable 32-bit scores), as in topk_select_kernel: sorted descending in LDS
__device__ __forceinline__ unsigned kth_of_maxima(unsigned mine, unsigned* list, int k) {
  list[threadIdx.x] = mine;
  __syncthreads();
  for (int size = 2; size <= 1024; size <<= 1)
    for (int st = size >> 1; st > 0; st >>= 1) {
      if (threadIdx.x < 512) {
        const int lo = ((threadIdx.x / st) * st * 2) + (threadIdx.x % st), hi = lo + st;
        const bool desc = ((lo & size) == 0);
        const unsigned a = list[lo], b = list[hi];
        if ((a < b) == desc) { list[lo] = b; list[hi] = a; }
      }
      __syncthreads();
    }
  const unsigned r = list[k - 1];
  __syncthreads();
  return r;
}
// tau[q] from the approximate sample scores [Q][S]
// (a needle whose squared norm 1 / w22 - 1e-12 lies outside the fp16 pass's range gets -inf: every row passes, every list overflows, the call reruns unbatched)
__device__ __forceinline__ bool batched_needle_ok(float w22q) { const float n2 = 1.f / w22q; return n2 >= GR_BNRM_MIN && n2 <= GR_BNRM_MAX; }
__global__ __launch_bounds__(1024) void batched_tau_kernel(const float* __restrict__ samp, long S, int k, const float* __restrict__ sw22s, float* __restrict__ tau, const float* __restrict__ w22) {
  __shared__ unsigned list[1024];
  const int q = blockIdx.x;
  unsigned mine = 0u;
  for (long i = threadIdx.x; i < S; i += 1024) { const unsigned o = orderable(samp[(long)q * S + i]); mine = o > mine ? o : mine; }
  const unsigned kth = kth_of_maxima(mine, list, k);
  // stored divided by sqrt(w22): cos_mfma_kernel compares (dot * sqrt(w32)) with it; the rounding of the division is far inside the 2 BERR slack
  if (threadIdx.x == 0) tau[q] = (kth && batched_needle_ok(w22[q])) ? (unorderable(kth) - 2.f * GR_BERR) / sw22s[q] : -INFINITY;
}
// the same threshold by ONE wave per needle (four needles per workgroup): the <= 256 sample maxima as orderable bit patterns, four per lane, and the exact
// k-th largest built bit by bit from the top (res |= bit while at least k patterns are >= the trial value) - no LDS, no block barriers (the 1024-thread
// bitonic sort above took 23 us for 1024 needles)
__global__ __launch_bounds__(256) void batched_tau_wave_kernel(const float* __restrict__ samp, int S, int Q, int k, const float* __restrict__ sw22s, float* __restrict__ tau, const float* __restrict__ w22) {
  const int lane = threadIdx.x & 63, q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= Q) return;
  unsigned ov[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) { const int e = lane + 64 * u; ov[u] = e < S ? orderable(samp[(long)q * S + e]) : 0u; }
  unsigned res = 0u;
#pragma unroll 1
  for (int b = 31; b >= 0; --b) {
    const unsigned t = res | (1u << b);
    const int c = __popcll(__ballot(ov[0] >= t)) + __popcll(__ballot(ov[1] >= t)) + __popcll(__ballot(ov[2] >= t)) + __popcll(__ballot(ov[3] >= t));
    if (c >= k) res = t;
  }
  if (lane == 0) tau[q] = (res && batched_needle_ok(w22[q])) ? (unorderable(res) - 2.f * GR_BERR) / sw22s[q] : -INFINITY;
}
// per needle: second cut on the approximate scores, exact re-score of what is left, sort, decode
template <bool ACCF>
__global__ __launch_bounds__(1024) void batched_select_kernel(const float* __restrict__ emb, int d, const float* __restrict__ needles,
                                                              const float* __restrict__ w22, const unsigned* __restrict__ cand_idx,
                                                              const float* __restrict__ cand_sc, const unsigned* __restrict__ counts, long nwg, int k,
                                                              long* __restrict__ idx, float* __restrict__ score, unsigned* __restrict__ status,
                                                              int slot, float margin2, SmallQ qr, int from_rows) {
  typedef typename std::conditional<ACCF, float, double>::type acc_t;
  __shared__ __attribute__((aligned(16))) unsigned long long keys[CHUNK];
  __shared__ unsigned rows[CHUNK];
  __shared__ unsigned list_n, over;
  __shared__ float sh_w22;
  unsigned* list = reinterpret_cast<unsigned*>(keys);            // the 1024 maxima live in the key array before it is needed
  const int q = blockIdx.x, tid = threadIdx.x;
  const unsigned* cnt = counts + (long)q * nwg;
  const unsigned* ci = cand_idx + (long)q * nwg * slot;
  const float* cs = cand_sc + (long)q * nwg * slot;
  if (tid == 0) { list_n = 0u; over = 0u; }
  __syncthreads();
  // The needle itself: from the workspace (batched path: needle_prep_kernel's copy and w22), or - a handful of needles, rows by value - straight
  // from the table, its 1 / (|a|^2 + 1e-12) formed here in needle_prep_kernel's arithmetic by one thread while the others scan the lists
  const float* nd = from_rows ? emb + qr.rows[q] * (long)d : needles + (long)q * d;
  if (from_rows && tid == 1023) {
    acc_t t = 0;
    for (int i0 = 0; i0 < d; i0 += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = nd[i0 + u < d ? i0 + u : d - 1];
#pragma unroll
      for (int u = 0; u < 16; ++u) if (i0 + u < d) t += v[u] * v[u];
    }
    float w = (float)t;
    w = w + 1e-12f;
    sh_w22 = 1.f / w;
  }
  // a list's entries are contiguous: eight scores per round as two 16-byte loads (one dependent load per entry, ~8 per list, was most of this kernel)
  unsigned mine = 0u, o = 0u;
  for (long g = tid; g < nwg; g += 1024) {
    const unsigned c0 = cnt[g];
    if (c0 > (unsigned)slot) o = 1u;
    const unsigned c = min(c0, (unsigned)slot);
    for (unsigned e0 = 0; e0 < c; e0 += 8) {
      const float4 va = *reinterpret_cast<const float4*>(cs + g * slot + e0), vb = *reinterpret_cast<const float4*>(cs + g * slot + e0 + 4);
      const float v[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
#pragma unroll
      for (int u = 0; u < 8; ++u) if (e0 + u < c) { const unsigned ov = orderable(v[u]); mine = ov > mine ? ov : mine; }
    }
  }
  if (o) over = 1u;
  const unsigned kth = kth_of_maxima(mine, list, k);
  if (over) { if (tid == 0 && status) *status = 1u; return; }
  const float tau2 = kth ? unorderable(kth) - margin2 : -INFINITY;
  for (long g = tid; g < nwg; g += 1024) {
    const unsigned c = min(cnt[g], (unsigned)slot);
    for (unsigned e0 = 0; e0 < c; e0 += 8) {
      const float4 va = *reinterpret_cast<const float4*>(cs + g * slot + e0), vb = *reinterpret_cast<const float4*>(cs + g * slot + e0 + 4);
      const float v[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (e0 + u < c && v[u] >= tau2) { const unsigned pos = atomicAdd(&list_n, 1u); if (pos < (unsigned)CHUNK) rows[pos] = ci[g * slot + e0 + u]; }
    }
  }
  __syncthreads();
  const unsigned m = list_n;
  if (m > (unsigned)CHUNK) { if (tid == 0 && status) *status = 1u; return; }
  // exact scores, cos_keys_kernel's arithmetic: fp32 products, sequential sums over the columns, the same w22 / w32 steps
  const float w22q = from_rows ? sh_w22 : w22[q];
  for (unsigned i = tid; i < m; i += 1024) {
    const long row = rows[i];
    const float* b = emb + row * (long)d;
    acc_t s1 = 0, s3 = 0;
    if ((d & 3) == 0) {
      // eight float4s of the row (and of the needle) are requested together, THEN summed in column order: a loop of d dependent scalar loads was
      // most of this kernel (and the whole row at once - 64 vectors - did not fit the 128 registers of a 1024-thread workgroup)
      const int n4 = d >> 2;
#pragma unroll 1
      for (int c0 = 0; c0 < n4; c0 += 8) {
        float4 bv4[8], nv4[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int c4 = c0 + u < n4 ? c0 + u : n4 - 1;
          bv4[u] = reinterpret_cast<const float4*>(b)[c4]; nv4[u] = reinterpret_cast<const float4*>(nd)[c4];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (c0 + u < n4) {
          const float bb[4] = {bv4[u].x, bv4[u].y, bv4[u].z, bv4[u].w}, nn[4] = {nv4[u].x, nv4[u].y, nv4[u].z, nv4[u].w};
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) { s3 += (acc_t)(bb[jj] * bb[jj]); s1 += (acc_t)(nn[jj] * bb[jj]); }
        }
      }
    } else
    for (int c = 0; c < d; ++c) { const float bv = b[c]; s3 += (acc_t)(bv * bv); s1 += (acc_t)(nd[c] * bv); }
    float w32 = (float)s3;
    w32 = w32 + 1e-12f;
    w32 = 1.f / w32;
    float w = w22q * w32;
    w = sqrtf(w);
    const float sc = (float)s1 * w;
    keys[i] = ((unsigned long long)orderable(sc) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)row);
  }
  int P = 64; while (P < (int)m) P <<= 1;
  for (int i = tid; i < P; i += 1024) if (i >= (int)m) keys[i] = 0ull;
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1)
    for (int st = size >> 1; st > 0; st >>= 1) {
      if (tid < P / 2) {
        const int lo = ((tid / st) * st * 2) + (tid % st), hi = lo + st;
        const bool desc = ((lo & size) == 0);
        const unsigned long long a = keys[lo], b = keys[hi];
        if ((a < b) == desc) { keys[lo] = b; keys[hi] = a; }
      }
      __syncthreads();
    }
  for (int r = tid; r < k; r += 1024) {
    const unsigned long long key = keys[r];
    idx[(long)q * k + r] = (long)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull));
    if (score) score[(long)q * k + r] = unorderable((uint32_t)(key >> 32));
  }
}

// ------------------------------------------------------------------ a handful of needles (the reference's five, apply_r.lua:267): fp32 filter + exact re-score
// The exact pass (cos_keys_kernel) streams emb in column chunks of 80-128 bytes per row - every 128-byte line is touched by several chunks, two
// barriers per chunk - and reached 3.1-3.5 TB/s whatever its arithmetic (round 4 ablation: 114 us with the multiply-adds switched off, 127 with
// them).  The exact op order (fp32 products summed sequentially in fp64) is only needed for rows that can make the top k, so this path, like the
// batched one above, lets a cheap APPROXIMATE score decide which rows those are and scores only them exactly (batched_select_kernel):
//   cos_approx_kernel: one wave per workgroup, thread = row, tiles of 64 whole rows - 64 * d * 4 contiguous bytes - brought HBM -> LDS by LDS-DMA
//   (one tile per workgroup, five or six workgroups per CU: APPROX_NB below), the needles' values next to them; fp32 FMA sums in
//   any order.  |approximate - exact as computed| <= (2 d + 16) 2^-24 =: eps  (both are within (d + 8) 2^-24 of the real cosine: each product and
//   each of the few fp32 steps behind the sums rounds by 2^-24 relative, and sum |a_i b_i| <= |a| |b|), so with margins of 2 eps at the two cuts
//   no row of the exact top k is lost: indices and scores stay bit-identical to the exact search.
//   MODE 0: the strided sample; every workgroup leaves the maximum of its 64 rows per needle, and the LAST workgroup to finish (arrival counter,
//   agent-scope release / acquire around it) takes the k-th largest of those S / 64 maxima (k of them are k distinct rows at or above it) minus
//   2 eps as the needle's threshold tau: one launch instead of needle_prep + sample + bound.  MODE 1: every row, (row, score) pairs >= tau into
//   the workgroup's own entries.
constexpr int APPROX_NB = 1;          // LDS tiles per workgroup of the main pass.  Round 4 measured all three (d = 100, same box): 2 tiles (the next one streams in behind the
                                      // current one's arithmetic), 3 workgroups per CU: 101 us; 3 tiles, 2 per CU: 120 us; ONE tile, 5 per CU: 83 us (d = 32: 46 -> 31 us).  The
                                      // pass is bound by what one wave gets through (25 DMA instructions + 150 LDS reads + 600 FMAs per tile, each step waiting for the one
                                      // before), so the LDS buys more as resident waves than as tiles in flight behind one wave.
constexpr int SBINS = 1024;           // bins of the sample's histogram of workgroup maxima over [-1, 1] (cos_approx_kernel, MODE 0)
constexpr bool APPROX_HALF = true;    // main pass: 32-row tiles, two lanes per row (cos_approx_kernel): eight workgroups per CU instead of five; same box at d = 100 / 128 / 32:
                                      // 83 -> 75, 108 -> 96, 31.5 -> 28.7 us (75 us for 400 MB = 5.7 TB/s with the event timer's overhead in it)
constexpr int ASLOT = 96;             // entries per (persistent workgroup, needle): expected ~4 at cfg5 (5200 candidates over 1280 workgroups)
static int approx_wgs(int d, int Q) {        // ONE resident round (up to 8 one-wave workgroups per CU): a grid larger than what fits ran its last part alone (d = 128: 335 us)
  const int d4 = d / 4, v = (d4 & 1) ? d4 : d4 + 1;
  const int nq = Q <= 2 ? 2 : (Q <= 5 ? 5 : 8);      // the instantiation launch_approx_nq picks
  const int tr = APPROX_HALF ? 32 : 64, nj = (tr * v + 63) / 64;
  const size_t lds = (size_t)APPROX_NB * 64 * nj * 16 + (size_t)nq * d * 4 + 256 + 64;
  int per_cu = (int)((size_t)160 * 1024 / lds); if (per_cu > 8) per_cu = 8; if (per_cu < 1) per_cu = 1;      // (8 x 256 = 2048 lists per needle: small_select_kernel's LPT)
  return 256 * per_cu;
}
struct ApproxArgs {
  float* needles; float* w22;                    // [Q][d], [Q]: written by workgroup 0 of the sample launch for the selection kernel (exact: needle_prep's arithmetic)
  unsigned* hist; unsigned* counter; float* tau;  // sample: [AQ_MAX][SBINS] histogram of the workgroups' maxima (zero between searches), arrival counter, [Q] thresholds
  unsigned* cand_idx; float* cand_sc; unsigned* counts;   // main pass: [Q][nwg][ASLOT], [Q][nwg]
  unsigned* status;
  int Q, k, accf; float eps2;
  int dbg;                                       // diagnostic ablations of the sample launch (GR_SEARCH_DEBUG bits 8, 16, 32: results then rely on an earlier call's thresholds)
};
template <int D4, int NQ, int MODE>
__global__ __launch_bounds__(64) void cos_approx_kernel(const float* __restrict__ emb, long N, long stride, SmallQ qr, ApproxArgs a) {
  constexpr int V = (D4 & 1) ? D4 : D4 + 1, d = D4 * 4;      // vectors per LDS row: odd, so that the 16 lanes of a ds_read_b128 group hit 64 distinct banks
  // NB tiles per workgroup in a ring: while one is multiplied, NB - 1 are in flight (APPROX_NB: two; three lost)
  constexpr int NB = MODE == 0 ? 1 : APPROX_NB;
  // HALF (the main pass): tiles of 32 rows, TWO lanes per row - lane l and l + 32 take the two halves of the row's columns and add their partial
  // sums through v_permlane32_swap.  A tile is then 12.8 KB at d = 100 and eight one-wave workgroups fit a CU instead of five: the pass is bound by
  // what one wave gets through (APPROX_NB), so the same bytes split over twice the waves move faster.
  constexpr bool HALF = MODE == 1 && APPROX_HALF;
  constexpr int TR = HALF ? 32 : 64, NJ = (TR * V + 63) / 64;            // rows per tile; DMA instructions per tile (the last one partly parked)
  __shared__ __attribute__((aligned(16))) uint4 tile[NB][64 * NJ];
  __shared__ __attribute__((aligned(16))) float4 nd[NQ * D4];
  const int lane = threadIdx.x, wg = blockIdx.x, nwg = gridDim.x, Q = a.Q;
  const size_t bytes = (size_t)N * (MODE == 0 ? stride : 1) * d * 4;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(emb), 0, (int)(bytes < 0x7FFFF000ul ? bytes : 0x7FFFF000ul), 0x00020000);
  // DMA instruction j of a tile covers LDS vectors 64 j + lane = (row, column vector) of the padded row-major tile; the pad column is parked out of range
  int voff[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int e = 64 * j + lane, r = e / V, c4 = e - r * V;
    voff[j] = (c4 < D4 && r < TR) ? (int)(((long)r * (MODE == 0 ? stride : 1) * d + 4 * c4) * 4) : (int)0x7FFFF000;
  }
  const long ntiles = (N + TR - 1) / TR;
  auto request = [&](long t, int buf) {
    const int soff = (int)(t * TR * (MODE == 0 ? stride : 1) * d * 4);      // (rows past N lie past the descriptor's range: zeros)
#pragma unroll
    for (int j = 0; j < NJ; ++j) lds_dma16(rs, &tile[buf][64 * j], voff[j], soff);
  };
#pragma unroll
  for (int j = 0; j < NB - 1; ++j)
    if (wg + (long)j * nwg < ntiles) request(wg + (long)j * nwg, j);
  // the needles: every workgroup gathers them itself (Q rows of d floats; rows come in the kernel arguments - no upload, no launch of their own)
  for (int e = lane; e < NQ * D4; e += 64) {
    const int q = e / D4, c4 = e - q * D4;
    nd[e] = *reinterpret_cast<const float4*>(emb + qr.rows[q < Q ? q : 0] * (long)d + 4 * c4);
  }
  __syncthreads();
  float w22a[NQ];                      // approximate 1 / (|needle|^2 + 1e-12): same arithmetic in every workgroup
#pragma unroll
  for (int q = 0; q < NQ; ++q) {       // lanes across the row's vectors, then a shuffle tree (a serial walk - 25 dependent LDS reads per needle - was 5 us per workgroup)
    float t = 0.f;
    for (int c4 = lane; c4 < D4; c4 += 64) { const float4 n = nd[q * D4 + c4]; t = fmaf(n.x, n.x, t); t = fmaf(n.y, n.y, t); t = fmaf(n.z, n.z, t); t = fmaf(n.w, n.w, t); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
    w22a[q] = 1.f / (t + 1e-12f);
  }
  if (MODE == 0 && wg == 0 && lane == 0 && a.status) *a.status = 0u;
  float tauq[NQ]; unsigned cnt[NQ]; float wmax[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) { tauq[q] = (MODE == 1 && q < Q) ? a.tau[q] : INFINITY; cnt[q] = 0u; wmax[q] = -INFINITY; }
  int buf = 0;
  for (long t = wg; t < ntiles; t += nwg, buf = buf + 1 == NB ? 0 : buf + 1) {
    if (MODE == 0 && (GR_DBG(a.dbg) & 32)) break;                         // ablation: no sample tile
    if (NB == 1) request(t, 0);                                   // (the sample: one tile per workgroup)
    // tile t has landed once at most the requests issued AFTER it are outstanding: V per tile already requested behind it (vmcnt counts in issue order)
    if (NB >= 3 && t + (long)(NB - 2) * nwg < ntiles) {
      if (NB == 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NJ <= 63 ? NJ : 0) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NJ <= 63 ? 2 * NJ : 0) : "memory");
    } else if (NB >= 4 && t + nwg < ntiles) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NJ <= 63 ? NJ : 0) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                 // (one wave: orders the LDS-DMA writes before the reads below for the compiler too)
    if (NB > 1 && t + (long)(NB - 1) * nwg < ntiles) request(t + (long)(NB - 1) * nwg, buf == 0 ? NB - 1 : buf - 1);   // into the buffer the previous round multiplied
    const uint4* row = &tile[buf][(HALF ? (lane & 31) : lane) * V];
    float s3 = 0.f, s1[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) s1[q] = 0.f;
    if constexpr (HALF) {
      // this lane's half of the columns: [cbeg, cbeg + ncol), four vectors per round; slots past ncol (d = 100: 13 + 12 columns) read a valid
      // column and are multiplied away
      constexpr int CH = (D4 + 1) / 2;
      const int hf = lane >> 5, cbeg = hf * CH, ncol = hf ? D4 - CH : CH;
#pragma unroll 1
      for (int c0 = 0; c0 < CH; c0 += 4) {
        float4 b[4], n[NQ][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int cc = c0 + u < ncol ? cbeg + c0 + u : cbeg;
          b[u] = __builtin_bit_cast(float4, row[cc]);
          if (c0 + u >= ncol) b[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int q = 0; q < NQ; ++q) n[q][u] = nd[q * D4 + cc];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          s3 = fmaf(b[u].x, b[u].x, s3); s3 = fmaf(b[u].y, b[u].y, s3); s3 = fmaf(b[u].z, b[u].z, s3); s3 = fmaf(b[u].w, b[u].w, s3);
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            s1[q] = fmaf(n[q][u].x, b[u].x, s1[q]); s1[q] = fmaf(n[q][u].y, b[u].y, s1[q]); s1[q] = fmaf(n[q][u].z, b[u].z, s1[q]); s1[q] = fmaf(n[q][u].w, b[u].w, s1[q]);
          }
        }
      }
      // the two halves of a row: every lane ends with the row's full sums (a + b = b + a: both lanes hold the same bits)
      auto pair_sum = [&](float v) {
#if defined(__HIP_DEVICE_COMPILE__)
        const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
        return v + __builtin_bit_cast(float, lane < 32 ? sw[1] : sw[0]);
#else
        return v;
#endif
      };
      s3 = pair_sum(s3);
#pragma unroll
      for (int q = 0; q < NQ; ++q) s1[q] = pair_sum(s1[q]);
    } else {
    // UB column vectors per round: their (NQ + 1) x UB LDS reads are in flight together, then the FMAs (a fully unrolled row - 150 reads at
    // d = 100, five needles - took all 512 registers and spilled)
    constexpr int UB = (D4 % 5 == 0) ? 5 : 4;
    static_assert(D4 % UB == 0, "row width");
#pragma unroll 1
    for (int c0 = 0; c0 < D4; c0 += UB) {
      float4 b[UB], n[NQ][UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        b[u] = __builtin_bit_cast(float4, row[c0 + u]);
#pragma unroll
        for (int q = 0; q < NQ; ++q) n[q][u] = nd[q * D4 + c0 + u];
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        s3 = fmaf(b[u].x, b[u].x, s3); s3 = fmaf(b[u].y, b[u].y, s3); s3 = fmaf(b[u].z, b[u].z, s3); s3 = fmaf(b[u].w, b[u].w, s3);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          s1[q] = fmaf(n[q][u].x, b[u].x, s1[q]); s1[q] = fmaf(n[q][u].y, b[u].y, s1[q]); s1[q] = fmaf(n[q][u].z, b[u].z, s1[q]); s1[q] = fmaf(n[q][u].w, b[u].w, s1[q]);
        }
      }
    }
    }
    const long i = t * TR + (HALF ? (lane & 31) : lane);
    const float w32 = 1.f / (s3 + 1e-12f);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const float v = s1[q] * sqrtf(w22a[q] * w32);
      if (MODE == 0) { if (i < N) wmax[q] = fmaxf(wmax[q], v); }
      else {
        const bool hit = (!HALF || lane < 32) && i < N && q < Q && v >= tauq[q];
        const unsigned long long m = __ballot(hit);
        if (m) {
          const unsigned pos = cnt[q] + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
          if (hit && pos < (unsigned)ASLOT) { const long at = ((long)q * nwg + wg) * ASLOT + pos; a.cand_idx[at] = (unsigned)i; a.cand_sc[at] = v; }
          cnt[q] += (unsigned)__popcll(m);
        }
      }
    }
  }
  if (MODE == 1) {
    if (lane < Q) {
      unsigned c = 0u;
#pragma unroll
      for (int q = 0; q < NQ; ++q) if (q == lane) c = cnt[q];
      a.counts[(long)lane * nwg + wg] = c;
    }
    return;
  }
  // MODE 0: this workgroup's maximum per needle goes into a histogram over [-1, 1] (one agent-scope atomic add per needle: performed at the memory side,
  // nothing to write back - no release fence), then the arrival; the LAST workgroup reads the SBINS bins per needle (16 per lane, all needles' loads in
  // flight together), takes a suffix count over the lanes and the lower edge of the bin that holds the k-th largest maximum - k distinct rows at or above
  // it - minus 2 eps as the threshold tau, and leaves the bins zero for the next search.  A bin is 2 / SBINS = 0.002 wide: tau sits that much low at
  // most (a few per cent more candidates).  (Earlier forms of the round kept the maxima themselves: plain stores + a release fence per workgroup, then an
  // exact k-th by a 64-bucket estimate or a bitwise radix select on the last wave - 12 of the launch's 22 us went to that tail.)
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    float m = wmax[q];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0 && q < Q) {
      int bin = (int)((m + 1.f) * (0.5f * (float)SBINS));               // (-inf: a workgroup without rows lands in bin 0)
      bin = bin < 0 ? 0 : (bin > SBINS - 1 ? SBINS - 1 : bin);
      __hip_atomic_fetch_add(a.hist + q * SBINS + bin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (GR_DBG(a.dbg) & 16) return;                                          // ablation: no arrival, no threshold
  unsigned arrived = 0u;
  if (lane == 0) arrived = __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  arrived = (unsigned)__shfl((int)arrived, 0, 64);
  if (arrived != (unsigned)nwg - 1u) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) __hip_atomic_store(a.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next search
  if (GR_DBG(a.dbg) & 8) return;                                           // ablation: the last workgroup's threshold computation (leaves the bins dirty)
  constexpr int BPL = SBINS / 64;                                  // bins per lane: lane L owns bins [BPL L, BPL L + BPL)
  unsigned hb[NQ][BPL];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int u = 0; u < BPL; ++u)
      hb[q][u] = q < Q ? __hip_atomic_load(a.hist + q * SBINS + BPL * lane + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    if (q >= Q) break;
    unsigned own = 0u;
#pragma unroll
    for (int u = 0; u < BPL; ++u) own += hb[q][u];
    unsigned suf = own;                                            // maxima in bins >= BPL * lane
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const unsigned t = (unsigned)__shfl_down((int)suf, off, 64); if (lane + off < 64) suf += t; }
    const unsigned above = suf - own;
    const unsigned total = (unsigned)__shfl((int)suf, 0, 64);
    if (suf >= (unsigned)a.k && above < (unsigned)a.k) {           // exactly one lane when there are k maxima at all
      unsigned run = above; int bsel = 0;
#pragma unroll
      for (int u = BPL - 1; u >= 0; --u) { run += hb[q][u]; if (run >= (unsigned)a.k) { bsel = BPL * lane + u; break; } }
      a.tau[q] = -1.f + (float)bsel * (2.f / (float)SBINS) - a.eps2 - 2.4e-7f;      // (2.4e-7: the rounding of m + 1 in the bin index can lift a maximum just below an edge into the bin above it)
    }
    if (total < (unsigned)a.k && lane == 0) a.tau[q] = -INFINITY;
#pragma unroll
    for (int u = 0; u < BPL; ++u) a.hist[q * SBINS + BPL * lane + u] = 0u;
  }
}
// Selection of the small path (one workgroup of 256 per needle): second cut, exact re-score of what is left, sort, results and a completion word
// straight into the caller's (pinned host) block.  Round 4, second form - batched_select_kernel (1024 threads) spent most of its 26 us in block
// barriers: a bitonic sort of 1024 per-thread maxima (55 steps) to find the cut, 8 dependent loads per row and needle round, a second sort.  Here:
//   cut     a histogram of the candidates' approximate scores over [tau, 1] in 2048 bins (LDS atomics), a suffix count over the bins, the lowest bin
//           edge with k candidates at or above it (minus one bin: the float rounding of a bin index) minus the 2 eps margin - every row of the
//           exact top k is at or above it (as for batched_select_kernel's cut: k rows with approximate score >= E have exact scores >= E - eps);
//   scores  the needle staged in LDS once, the row 16 vectors per round; cos_keys_kernel's arithmetic (fp32 products, sequential sums);
//   done    every thread's stores fenced at system scope, then ONE word per needle = the call's sequence number: the host polls it (no stream
//           synchronisation: 1-3 us per search, measured).
constexpr int SSEL_BINS = 2048, SSEL_MAX = 256;
// NT threads (256: the small path's <= 2048 lists per needle; 512: the batched path's N / 256 lists), SLOT entries per list, FROM_ROWS: the needle is row
// qr.rows[q] of the table and its norm is formed here (small path) / the needle and its 1 / (|a|^2 + 1e-12) come from needle_prep_kernel's arrays (batched);
// lo_scale (nullable): tau[q] * lo_scale[q] is the lower edge of the candidates' scores (the batched path stores tau divided by sqrt(w22)).
template <bool ACCF, int NT, int SLOT_, bool FROM_ROWS>
__global__ __launch_bounds__(NT) void small_select_kernel(const float* __restrict__ emb, int d, const unsigned* __restrict__ cand_idx,
                                                         const float* __restrict__ cand_sc, const unsigned* __restrict__ counts, int nwg, int k,
                                                         long* __restrict__ idx, float* __restrict__ score, unsigned* __restrict__ status,
                                                         float margin2, SmallQ qr, const float* __restrict__ tau, unsigned* __restrict__ done, unsigned seq,
                                                         const float* __restrict__ needles, const float* __restrict__ w22, const float* __restrict__ lo_scale) {
  typedef typename std::conditional<ACCF, float, double>::type acc_t;
  __shared__ unsigned hist[SSEL_BINS];
  __shared__ __attribute__((aligned(16))) unsigned long long keys[SSEL_MAX];
  __shared__ unsigned rows[SSEL_MAX];
  __shared__ __attribute__((aligned(16))) float ndl[BD_MAX];
  __shared__ unsigned wtot[NT / 64];
  __shared__ unsigned list_n, over, cutbin;
  __shared__ float sh_w22;
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned* cnt = counts + (long)q * nwg;
  constexpr int ASLOT = SLOT_;                                     // (shadows the small path's constant: this instantiation's list length)
  const unsigned* ci = cand_idx + (long)q * nwg * ASLOT;
  const float* cs = cand_sc + (long)q * nwg * ASLOT;
  const float* nd = FROM_ROWS ? emb + qr.rows[q] * (long)d : needles + (long)q * d;
  const float lo = lo_scale ? tau[q] * lo_scale[q] - 1e-6f : tau[q];
  bool fail = !(lo > -INFINITY);                                    // no threshold (fewer than k sample maxima): the unfiltered search decides
  for (int i = tid; i < SSEL_BINS; i += NT) hist[i] = 0u;
  if (tid == 0) { list_n = 0u; over = 0u; cutbin = 0u; }
  for (int c = tid; c < d; c += NT) ndl[c] = nd[c];
  __syncthreads();
  if (!FROM_ROWS) { if (tid == NT - 1) sh_w22 = w22[q]; }
  else if (tid == NT - 1) {                                                 // 1 / (|needle|^2 + 1e-12) in needle_prep_kernel's arithmetic, from the staged copy
    acc_t t = 0;                                                    // (read from global memory by this one thread it was 7 dependent rounds: ~10 us)
    for (int i = 0; i < d; ++i) { const float v = ndl[i]; t += v * v; }
    float w = (float)t;
    w = w + 1e-12f;
    sh_w22 = 1.f / w;
  }
  const float span = 1.0001f - lo, inv = span > 0.f ? (float)SSEL_BINS / span : 0.f, width = span / (float)SSEL_BINS;
  // pass 1: histogram of the approximate scores.  The lists were written by the pass before, on other XCDs: every load here is a trip to the
  // fabric (~1.5 us) - a thread's counts are requested together, then the first eight entries of all its lists together (an average list holds
  // four); a loop of count -> entries -> next list was 10 dependent trips per thread, most of this kernel.
  constexpr int LPT = 8;                                            // lists per thread (nwg <= 2048)
  unsigned lc[LPT]; float4 la[LPT], lb[LPT];
  unsigned o = 0u;
#pragma unroll
  for (int j = 0; j < LPT; ++j) { const int g = tid + NT * j; lc[j] = (!fail && g < nwg) ? cnt[g] : 0u; }
#pragma unroll
  for (int j = 0; j < LPT; ++j) {
    const int g = tid + NT * j;
    if (lc[j] > (unsigned)ASLOT) { o = 1u; lc[j] = (unsigned)ASLOT; }
    la[j] = lc[j] > 0u ? *reinterpret_cast<const float4*>(cs + (long)g * ASLOT) : make_float4(0.f, 0.f, 0.f, 0.f);
    lb[j] = lc[j] > 4u ? *reinterpret_cast<const float4*>(cs + (long)g * ASLOT + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  auto bin_of = [&](float v) { int b_ = (int)((v - lo) * inv); return b_ < 0 ? 0 : (b_ > SSEL_BINS - 1 ? SSEL_BINS - 1 : b_); };
#pragma unroll
  for (int j = 0; j < LPT; ++j) {
    const int g = tid + NT * j;
    const float v[8] = {la[j].x, la[j].y, la[j].z, la[j].w, lb[j].x, lb[j].y, lb[j].z, lb[j].w};
#pragma unroll
    for (int u = 0; u < 8; ++u) if ((unsigned)u < lc[j]) atomicAdd(&hist[bin_of(v[u])], 1u);
    for (unsigned e = 8; e < lc[j]; ++e) atomicAdd(&hist[bin_of(cs[(long)g * ASLOT + e])], 1u);      // (2 % of the lists)
  }
  if (o) over = 1u;
  __syncthreads();
  fail = fail || over != 0u;
  // suffix count over the bins: thread t owns bins BPT t .. BPT t + BPT - 1; S(t) = candidates in bins >= BPT t
  constexpr int BPT = SSEL_BINS / NT;
  unsigned own = 0u;
#pragma unroll
  for (int j = 0; j < BPT; ++j) own += hist[BPT * tid + j];
  unsigned suf = own;                                               // suffix sum over the lanes of the wave, then the waves above
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const unsigned t = (unsigned)__shfl_down((int)suf, off, 64); if (lane + off < 64) suf += t; }
  if (lane == 0) wtot[wave] = suf;
  __syncthreads();
  for (int w = wave + 1; w < NT / 64; ++w) suf += wtot[w];
  const unsigned above = suf - own;                                 // candidates in bins >= 8 (t + 1)
  if (suf >= (unsigned)k && above < (unsigned)k) {                  // exactly one thread when there are k candidates at all (else the cut stays at bin 0)
    unsigned run = above;
    for (int j = BPT - 1; j >= 0; --j) { run += hist[BPT * tid + j]; if (run >= (unsigned)k) { cutbin = (unsigned)(BPT * tid + j); break; } }
  }
  __syncthreads();
  const float tau2 = lo + ((float)cutbin - 1.f) * width - margin2;
  // pass 2: the candidates at or above the cut (their scores are still in registers)
#pragma unroll
  for (int j = 0; j < LPT; ++j) {
    const int g = tid + NT * j;
    const float v[8] = {la[j].x, la[j].y, la[j].z, la[j].w, lb[j].x, lb[j].y, lb[j].z, lb[j].w};
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if ((unsigned)u < lc[j] && v[u] >= tau2) { const unsigned pos = atomicAdd(&list_n, 1u); if (pos < (unsigned)SSEL_MAX) rows[pos] = ci[(long)g * ASLOT + u]; }
    for (unsigned e = 8; e < lc[j]; ++e)
      if (cs[(long)g * ASLOT + e] >= tau2) { const unsigned pos = atomicAdd(&list_n, 1u); if (pos < (unsigned)SSEL_MAX) rows[pos] = ci[(long)g * ASLOT + e]; }
  }
  __syncthreads();
  const unsigned m = list_n;
  fail = fail || m > (unsigned)SSEL_MAX || m < (unsigned)k;
  if (!fail) {
    // exact scores, cos_keys_kernel's arithmetic: fp32 products, sequential sums over the columns, the same w22 / w32 steps
    const float w22q = sh_w22;
    if ((unsigned)tid < m) {
      const long row = rows[tid];
      const float* b = emb + row * (long)d;
      acc_t s1 = 0, s3 = 0;
      if ((d & 3) == 0) {
        const int n4 = d >> 2;
#pragma unroll 1
        for (int c0 = 0; c0 < n4; c0 += 32) {                      // the whole row in one round of loads (d <= 128)
          float4 bv4[32];
#pragma unroll
          for (int u = 0; u < 32; ++u) bv4[u] = reinterpret_cast<const float4*>(b)[c0 + u < n4 ? c0 + u : n4 - 1];
#pragma unroll
          for (int u = 0; u < 32; ++u) if (c0 + u < n4) {
            const float4 nv = reinterpret_cast<const float4*>(ndl)[c0 + u];
            const float bb[4] = {bv4[u].x, bv4[u].y, bv4[u].z, bv4[u].w}, nn[4] = {nv.x, nv.y, nv.z, nv.w};
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) { s3 += (acc_t)(bb[jj] * bb[jj]); s1 += (acc_t)(nn[jj] * bb[jj]); }
          }
        }
      } else
      for (int c = 0; c < d; ++c) { const float bv = b[c]; s3 += (acc_t)(bv * bv); s1 += (acc_t)(ndl[c] * bv); }
      float w32 = (float)s3;
      w32 = w32 + 1e-12f;
      w32 = 1.f / w32;
      float w = w22q * w32;
      w = sqrtf(w);
      const float sc = (float)s1 * w;
      keys[tid] = ((unsigned long long)orderable(sc) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)row);
    }
    int P = 64; while (P < (int)m) P <<= 1;
    if (tid >= (int)m && tid < P) keys[tid] = 0ull;
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1)
      for (int st = size >> 1; st > 0; st >>= 1) {
        if (tid < P / 2) {
          const int l0 = ((tid / st) * st * 2) + (tid % st), h0 = l0 + st;
          const bool desc = ((l0 & size) == 0);
          const unsigned long long x = keys[l0], y = keys[h0];
          if ((x < y) == desc) { keys[l0] = y; keys[h0] = x; }
        }
        __syncthreads();
      }
    for (int r = tid; r < k; r += NT) {
      const unsigned long long key = keys[r];
      idx[(long)q * k + r] = (long)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull));
      if (score) score[(long)q * k + r] = unorderable((uint32_t)(key >> 32));
    }
  } else if (tid == 0 && status) *status = 1u;
  // completion: the results (and the status word) are visible to the host before the needle's word carries this call's sequence number
  if (done) {                                                      // (uniform: only the small path hands its results to a polling host)
    __threadfence_system();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(done + q, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

bool cosine_topk_small_path(long N, int d, int Q, int k) {
  static const bool on = !GR_KNOB_SET("GR_SEARCH_NO_APPROX");
  const int d4 = d / 4;
  return on && Q >= 1 && Q <= AQ_MAX && (d & 3) == 0 && (d4 == 8 || d4 == 16 || d4 == 25 || d4 == 32) && N >= FILTER_MIN_ROWS && k <= 128 &&
         (size_t)N * d * 4 < 0x7FFFF000ul;
}
template <int D4, int MODE>
static void launch_approx_nq(int Q, unsigned grid, hipStream_t s, const float* emb, long N, long stride, const SmallQ& qr, const ApproxArgs& a) {
  if (Q <= 2) hipLaunchKernelGGL((cos_approx_kernel<D4, 2, MODE>), dim3(grid), dim3(64), 0, s, emb, N, stride, qr, a);
  else if (Q <= 5) hipLaunchKernelGGL((cos_approx_kernel<D4, 5, MODE>), dim3(grid), dim3(64), 0, s, emb, N, stride, qr, a);
  else hipLaunchKernelGGL((cos_approx_kernel<D4, 8, MODE>), dim3(grid), dim3(64), 0, s, emb, N, stride, qr, a);
}
template <int MODE>
static void launch_approx(int d4, int Q, unsigned grid, hipStream_t s, const float* emb, long N, long stride, const SmallQ& qr, const ApproxArgs& a) {
  switch (d4) {
    case 8: launch_approx_nq<8, MODE>(Q, grid, s, emb, N, stride, qr, a); break;
    case 16: launch_approx_nq<16, MODE>(Q, grid, s, emb, N, stride, qr, a); break;
    case 25: launch_approx_nq<25, MODE>(Q, grid, s, emb, N, stride, qr, a); break;
    default: launch_approx_nq<32, MODE>(Q, grid, s, emb, N, stride, qr, a); break;
  }
}

static long chunks_of(long n) { return (n + CHUNK - 1) / CHUNK; }

size_t cosine_topk_workspace_bytes(long N, int d, int Q, int k) {
  const long n1 = chunks_of(N) * k, n2 = chunks_of(n1) * k;
  size_t keys = sizeof(unsigned long long) * (size_t)Q * (N + n1 + n2);
  if (cosine_topk_small_path(N, d, Q, k)) {
    // the small-needle path carves its candidate lists out of the key area (launch_cosine_topk): maxima [AQ_MAX][256] | tau [AQ_MAX] | pad 8 |
    // rows [Q][wgs][ASLOT] | scores [Q][wgs][ASLOT] | counts [Q][wgs].  Below ~197 K rows that is MORE than the keys (out-of-bounds
    // device writes on a fresh context at 131072 <= N < 197 K) - the workspace is the larger of the two layouts.
    const size_t awgs = (size_t)approx_wgs(d, Q);
    const size_t small = sizeof(float) * ((size_t)AQ_MAX * 256 + AQ_MAX) + sizeof(unsigned) * 8 + (size_t)Q * awgs * ((size_t)ASLOT * 8 + 4) + 256;
    if (small > keys) keys = small;
  }
  return sizeof(float) * ((size_t)Q * d + Q + 8) + sizeof(unsigned) * (size_t)(Q + 8) + 1024 + keys;
}

int g_search_debug = 0;      // diagnostic ablations (GR_SEARCH_DEBUG: 1 no global loads, 2 no arithmetic, 4 no epilogue; results are then wrong by design)
template <bool ACCF, int MODE>
static void launch_keys_nq(int nq, unsigned nb, hipStream_t s, const float* emb, long N, int d, const float* needles, const float* w22, int q0,
                           unsigned long long* keys, long stride, const unsigned long long* bound, unsigned* counts) {
  const bool dc20 = d % 20 == 0 && d % 32 != 0;
#define GR_KEYS(NQ_) do { if (dc20) hipLaunchKernelGGL((cos_keys_kernel<ACCF, MODE, NQ_, 20>), dim3(nb), dim3(ROWS), 0, s, emb, N, d, needles, w22, q0, keys, stride, bound, counts, g_search_debug); \
                          else hipLaunchKernelGGL((cos_keys_kernel<ACCF, MODE, NQ_, 32>), dim3(nb), dim3(ROWS), 0, s, emb, N, d, needles, w22, q0, keys, stride, bound, counts, g_search_debug); } while (0)
  switch (nq) {
    case 1: GR_KEYS(1); break; case 2: GR_KEYS(2); break; case 3: GR_KEYS(3); break; case 4: GR_KEYS(4); break;
    case 5: GR_KEYS(5); break; case 6: GR_KEYS(6); break; case 7: GR_KEYS(7); break; default: GR_KEYS(8); break;
  }
#undef GR_KEYS
}
static void launch_keys(bool accf, int mode, int nq, unsigned nb, hipStream_t s, const float* emb, long N, int d, const float* needles, const float* w22,
                        int q0, unsigned long long* keys, long stride, const unsigned long long* bound, unsigned* counts) {
  if (accf) {
    if (mode == 0) launch_keys_nq<true, 0>(nq, nb, s, emb, N, d, needles, w22, q0, keys, stride, bound, counts);
    else if (mode == 1) launch_keys_nq<true, 1>(nq, nb, s, emb, N, d, needles, w22, q0, keys, stride, bound, counts);
    else launch_keys_nq<true, 2>(nq, nb, s, emb, N, d, needles, w22, q0, keys, stride, bound, counts);
  } else {
    if (mode == 0) launch_keys_nq<false, 0>(nq, nb, s, emb, N, d, needles, w22, q0, keys, stride, bound, counts);
    else if (mode == 1) launch_keys_nq<false, 1>(nq, nb, s, emb, N, d, needles, w22, q0, keys, stride, bound, counts);
    else launch_keys_nq<false, 2>(nq, nb, s, emb, N, d, needles, w22, q0, keys, stride, bound, counts);
  }
}

// status_dev (nullable): receives 0, or 1 when the filtered path dropped candidates (rerun with unfiltered = 1)
int launch_cosine_topk(const float* emb, long N, int d, const long* query_rows_dev, int Q, int k,
                       long* idx_out, float* score_out, int accf, void* workspace, hipStream_t s, unsigned* status_dev, int unfiltered,
                       const long* query_rows_host, unsigned* arrival_counter, unsigned* done_words, unsigned seq) {
  if (k > 1024 || k < 1 || k > N || N >= 0xFFFFFFFFl || d < 1 || d > 4096 * 4) return -1;
  g_search_debug = GR_KNOB("GR_SEARCH_DEBUG", 0);
  // workspace carve: needles [Q][d] | w22 [Q] | counts [Q] | keys A | keys B | keys C
  char* w = reinterpret_cast<char*>(workspace);
  float* needles = reinterpret_cast<float*>(w); w += sizeof(float) * (size_t)Q * d;
  float* w22 = reinterpret_cast<float*>(w); w += sizeof(float) * (size_t)((Q + 3) / 4 * 4);
  unsigned* counts = reinterpret_cast<unsigned*>(w); w += sizeof(unsigned) * (size_t)((Q + 3) / 4 * 4);
  w = reinterpret_cast<char*>(((uintptr_t)w + 255) & ~(uintptr_t)255);
  unsigned long long* keysA = reinterpret_cast<unsigned long long*>(w);
  const long n1 = chunks_of(N) * k;
  unsigned long long* keysB = keysA + (size_t)Q * N;
  unsigned long long* keysC = keysB + (size_t)Q * n1;
  const bool filter = !unfiltered && status_dev && N >= FILTER_MIN_ROWS && k * 8 <= SAMPLE_ROWS && k <= CHUNK / 2;      // (entries + sample + bounds + counts fit the N keys of region A: SLOT * 8 / ROWS + ... < 8 bytes per row)
  if (filter && query_rows_host && arrival_counter && cosine_topk_small_path(N, d, Q, k)) {
    // keys A = maxima [Q][256] | tau [8] | arrival counter | candidate rows [Q][wgs][ASLOT] | scores | counts [Q][wgs]
    const long S = SAMPLE_ROWS, stride = N / S;
    const unsigned swg = (unsigned)(S / 64);
    float* wgmax = reinterpret_cast<float*>(keysA); float* tau = wgmax + (size_t)AQ_MAX * 256; unsigned* counter = arrival_counter; unsigned* hist = arrival_counter + 16;     // (the context's: zero between searches)
    unsigned* pad_ = reinterpret_cast<unsigned*>(tau + AQ_MAX);
    const int awgs = approx_wgs(d, Q);
    unsigned* cidx = pad_ + 8; float* csc = reinterpret_cast<float*>(cidx + (size_t)Q * awgs * ASLOT);
    unsigned* wcnt = reinterpret_cast<unsigned*>(csc + (size_t)Q * awgs * ASLOT);
    SmallQ qr{}; for (int q = 0; q < AQ_MAX; ++q) qr.rows[q] = query_rows_host[q < Q ? q : 0];
    ApproxArgs a{};
    a.needles = needles; a.w22 = w22; a.hist = hist; a.counter = counter; a.tau = tau; a.cand_idx = cidx; a.cand_sc = csc; a.counts = wcnt;
    a.status = status_dev; a.Q = Q; a.k = k; a.accf = accf; a.eps2 = 2.f * (float)(2 * d + 16) * 5.9604645e-8f; a.dbg = g_search_debug;
    {
      KtScope kt("cos_approx_kernel (sample + bound)", 0.0, 4.0 * S * d, s);
      launch_approx<0>(d / 4, Q, swg, s, emb, S, stride, qr, a);
    }
    {
      KtScope kt("cos_approx_kernel", 2.0 * N * d * Q, 4.0 * N * d, s);
      launch_approx<1>(d / 4, Q, (unsigned)awgs, s, emb, N, 1L, qr, a);
    }
    static const bool old_select = GR_KNOB_SET("GR_SEARCH_OLD_SELECT");       // A/B: round 4's first selection kernel (no completion words: the caller synchronises)
    if (old_select || k > SSEL_MAX / 2) {
      KtScope kt("batched_select_kernel", 0.0, 0.0, s);
      if (accf) hipLaunchKernelGGL(batched_select_kernel<true>, dim3(Q), dim3(1024), 0, s, emb, d, needles, w22, cidx, csc, wcnt, (long)awgs, k, idx_out, score_out, status_dev, ASLOT, a.eps2, qr, 1);
      else hipLaunchKernelGGL(batched_select_kernel<false>, dim3(Q), dim3(1024), 0, s, emb, d, needles, w22, cidx, csc, wcnt, (long)awgs, k, idx_out, score_out, status_dev, ASLOT, a.eps2, qr, 1);
      return 0;
    }
    KtScope kt("small_select_kernel", 0.0, 0.0, s);
    if (accf) hipLaunchKernelGGL((small_select_kernel<true, 256, ASLOT, true>), dim3(Q), dim3(256), 0, s, emb, d, cidx, csc, wcnt, awgs, k, idx_out, score_out, status_dev, a.eps2, qr, tau, done_words, seq, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr);
    else hipLaunchKernelGGL((small_select_kernel<false, 256, ASLOT, true>), dim3(Q), dim3(256), 0, s, emb, d, cidx, csc, wcnt, awgs, k, idx_out, score_out, status_dev, a.eps2, qr, tau, done_words, seq, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr);
    return done_words ? 2 : 0;        // 2: the needles' completion words will carry `seq`
  }
  if (accf) hipLaunchKernelGGL(needle_prep_kernel<true>, dim3((Q + 63) / 64), dim3(64), 0, s, emb, d, query_rows_dev, Q, needles, w22, counts, status_dev);
  else hipLaunchKernelGGL(needle_prep_kernel<false>, dim3((Q + 63) / 64), dim3(64), 0, s, emb, d, query_rows_dev, Q, needles, w22, counts, status_dev);
  static const bool batched_on = !GR_KNOB_SET("GR_SEARCH_NO_BATCHED");
  static const int batch_min_q = GR_KNOB("GR_BATCH_MIN_Q", BATCH_MIN_Q);
  if (filter && batched_on && Q >= batch_min_q && Q <= BQ_MAX && d <= BD_MAX && k <= 128 && N >= 2 * BSAMPLE_ROWS) {      // (k distinct workgroup maxima must exist: 256 workgroups)
    // keys A = sample scores [Q][S] | tau [Qpad] | sqrt(w22) [Qpad] | bf16 needles [Qpad][KS] | candidate rows [Q][nwg][BSLOT] | scores | counts [Q][nwg]
    // sample: BSAMPLE_ROWS strided rows in workgroups of 256; each leaves its maximum per needle, tau from the k-th largest of those
    const long S = BSAMPLE_ROWS, nwg = (N + 255) / 256, stride = N / S, swg = S / 256;
    const int NK = d <= 32 ? 2 : (d <= 64 ? 4 : (d <= 112 ? 7 : 8)), KS = NK * 16 + 8, Qpad = (Q + 63) / 64 * 64;
    float* samp = reinterpret_cast<float*>(keysA); float* tau = samp + (size_t)Q * swg; float* sw22s = tau + Qpad;
    unsigned short* nb16 = reinterpret_cast<unsigned short*>(sw22s + Qpad);
    unsigned* cidx = reinterpret_cast<unsigned*>(nb16 + (size_t)Qpad * KS); float* csc = reinterpret_cast<float*>(cidx + (size_t)Q * nwg * BSLOT);
    unsigned* wcnt = reinterpret_cast<unsigned*>(csc + (size_t)Q * nwg * BSLOT);
    const size_t lds0 = (size_t)256 * KS * 2 + sizeof(float) * (256 + 128 + 128) + sizeof(unsigned) * (size_t)((((Q > 128 ? Q : 128) + 1) & ~1) + 2);
    // the main pass's per-wave hit queues: what two workgroups per CU leave of the LDS, at most 256 entries per wave (~20 expected per tile); none below 32
    int qcap = (int)(((size_t)80 * 1024 - lds0) / (4 * 8)); qcap = qcap > 256 ? 256 : (qcap < 32 ? 0 : qcap & ~31);
    qcap = GR_KNOB("GR_BATCHED_QCAP", qcap);                   // ablation build: 0 = round 4's direct hit path
    const size_t lds = lds0 + (size_t)4 * 8 * qcap;
    qcap |= GR_KNOB("GR_BATCHED_DEBUG", 0) << 16;
    hipLaunchKernelGGL(needles_bf16_kernel, dim3((unsigned)(((long)Qpad * KS + 255) / 256)), dim3(256), 0, s, needles, w22, Q, Qpad, d, KS, nb16, sw22s, tau);
#define GR_MFMA(MODE_, grid_, gy_, N_, stride_, tau_, out_, ci_, cs_, wc_)                                                                  \
    do {                                                                                                                              \
      switch (NK) {                                                                                                                   \
        case 2: (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cos_mfma_kernel<MODE_, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); hipLaunchKernelGGL((cos_mfma_kernel<MODE_, 2>), dim3(grid_, gy_), dim3(256), lds, s, emb, N_, d, stride_, nb16, sw22s, Q, tau_, out_, ci_, cs_, wc_, qcap); break; \
        case 4: (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cos_mfma_kernel<MODE_, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); hipLaunchKernelGGL((cos_mfma_kernel<MODE_, 4>), dim3(grid_, gy_), dim3(256), lds, s, emb, N_, d, stride_, nb16, sw22s, Q, tau_, out_, ci_, cs_, wc_, qcap); break; \
        case 7: (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cos_mfma_kernel<MODE_, 7>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); hipLaunchKernelGGL((cos_mfma_kernel<MODE_, 7>), dim3(grid_, gy_), dim3(256), lds, s, emb, N_, d, stride_, nb16, sw22s, Q, tau_, out_, ci_, cs_, wc_, qcap); break; \
        default: (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cos_mfma_kernel<MODE_, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); hipLaunchKernelGGL((cos_mfma_kernel<MODE_, 8>), dim3(grid_, gy_), dim3(256), lds, s, emb, N_, d, stride_, nb16, sw22s, Q, tau_, out_, ci_, cs_, wc_, qcap); break; \
      }                                                                                                                               \
    } while (0)
    {
      KtScope kt("cos_mfma_kernel (sample)", 2.0 * S * d * Q, 4.0 * S * d, s);
      const int tiles_q = (Q + 63) / 64, gy = GR_KNOB("GR_BATCHED_SAMPLE_Y", tiles_q >= 16 ? 4 : (tiles_q >= 4 ? 2 : 1));
      GR_MFMA(0, (unsigned)((S + 255) / 256), (unsigned)gy, S, stride, (const float*)nullptr, samp, (unsigned*)nullptr, (float*)nullptr, (unsigned*)nullptr);
    }
    static const bool old_tail = GR_KNOB_SET("GR_BATCHED_OLD_TAIL");      // A/B: round 4's first threshold and selection kernels (bitonic sorts of 1024 per-thread maxima)
    if (old_tail || swg > 256) {
      KtScope kt("batched_tau_kernel", 0.0, 4.0 * swg * Q, s);
      hipLaunchKernelGGL(batched_tau_kernel, dim3(Q), dim3(1024), 0, s, samp, swg, k, sw22s, tau, w22);
    } else {
      KtScope kt("batched_tau_wave_kernel", 0.0, 4.0 * swg * Q, s);
      hipLaunchKernelGGL(batched_tau_wave_kernel, dim3((unsigned)((Q + 3) / 4)), dim3(256), 0, s, samp, (int)swg, Q, k, sw22s, tau, w22);
    }
    {
      KtScope kt("cos_mfma_kernel", 2.0 * N * d * Q, 4.0 * N * d, s);
      GR_MFMA(1, (unsigned)nwg, 1u, N, 1L, (const float*)tau, (float*)nullptr, cidx, csc, wcnt);
    }
#undef GR_MFMA
    // (the histogram-cut selection kernel of the small path, instantiated for these lists - small_select_kernel<ACCF, 512, BSLOT, false>, tau scaled by sqrt(w22) -
    // was measured here: 109 us against this kernel's 101 for 1024 needles.  Both read 3907 sixty-four-byte lists per needle, 70 % of them non-empty, one DRAM
    // line each: 256 MB of scattered reads set the time, not the barriers of the sort; not used.  Round 6, again with the fp16 bound's eight times fewer
    // candidates (profiles/r06_ab_search_hist_select.txt): 87 us against 84-88 for 1024 needles, 26 against 36 for 256 - still the lists' scattered lines.
    // Also round 6, built, bit-identical, removed (git history; profiles/r06_ab_search_compact_lists.txt): ONE compact list per needle, filled at the end of the
    // main pass from wave queues that live as long as the workgroup (one global atomic add with return per entry) - the selection drops to 49-52 us, but the
    // 0.6 M returning atomics cost the main pass more than the per-tile drain and the 4 M count stores they replace: 411 -> 437 us at 1024 needles (the call
    // -2 %), 98 -> 179 us at 48 needles (3907 workgroups x 4 waves on 48 counters).  Device-scope atomics are performed at the memory side.)
    KtScope kt("batched_select_kernel", 0.0, 0.0, s);
    if (accf) hipLaunchKernelGGL(batched_select_kernel<true>, dim3(Q), dim3(1024), 0, s, emb, d, needles, w22, cidx, csc, wcnt, nwg, k, idx_out, score_out, status_dev, BSLOT, 2.f * GR_BERR, SmallQ{}, 0);
    else hipLaunchKernelGGL(batched_select_kernel<false>, dim3(Q), dim3(1024), 0, s, emb, d, needles, w22, cidx, csc, wcnt, nwg, k, idx_out, score_out, status_dev, BSLOT, 2.f * GR_BERR, SmallQ{}, 0);
    return 0;
  }
  if (filter) {
    // keys A = candidate entries [Q][nb][SLOT] | sample keys [Q][SAMPLE_ROWS] | bounds [Q] | counts [Q][nb]
    const unsigned nbs = (unsigned)((SAMPLE_ROWS + ROWS - 1) / ROWS), nb = (unsigned)((N + ROWS - 1) / ROWS);
    unsigned long long* cand = keysA; unsigned long long* samp = keysA + (size_t)Q * nb * SLOT; unsigned long long* bnd = samp + (size_t)Q * SAMPLE_ROWS;
    unsigned* wg_counts = reinterpret_cast<unsigned*>(bnd + ((Q + 3) / 4 * 4));
    const long stride = N / SAMPLE_ROWS;
    for (int q0 = 0; q0 < Q; q0 += QG) {
      KtScope kt("cos_keys_kernel (sample)", 0.0, 4.0 * SAMPLE_ROWS * d, s);
      launch_keys(accf != 0, 1, min(QG, Q - q0), nbs, s, emb, SAMPLE_ROWS, d, needles, w22, q0, samp, stride, nullptr, nullptr);
    }
    {
      KtScope kt("topk_select_kernel (bound)", 0.0, 8.0 * Q * SAMPLE_ROWS, s);
      hipLaunchKernelGGL(topk_select_kernel<true>, dim3(Q), dim3(1024), 0, s, samp, (long)SAMPLE_ROWS, nullptr, k, bnd, nullptr, nullptr, nullptr);
    }
    for (int q0 = 0; q0 < Q; q0 += QG) {
      KtScope kt("cos_keys_kernel", 2.0 * N * d * (Q - q0 < QG ? Q - q0 : QG), 4.0 * N * d, s);
      launch_keys(accf != 0, 2, min(QG, Q - q0), nb, s, emb, N, d, needles, w22, q0, cand, 1, bnd, wg_counts);
    }
    KtScope kt("topk_select_kernel", 0.0, 8.0 * Q * 4096, s);
    hipLaunchKernelGGL(topk_select_kernel<false>, dim3(Q), dim3(1024), 0, s, cand, (long)nb, wg_counts, k, nullptr, idx_out, score_out, status_dev);
    return 0;
  }
  const unsigned nb = (unsigned)((N + ROWS - 1) / ROWS);
  for (int q0 = 0; q0 < Q; q0 += QG) {
    KtScope kt("cos_keys_kernel", 2.0 * N * d * (Q - q0 < QG ? Q - q0 : QG), 4.0 * N * d + 8.0 * N * (Q - q0 < QG ? Q - q0 : QG), s);
    launch_keys(accf != 0, 0, min(QG, Q - q0), nb, s, emb, N, d, needles, w22, q0, keysA, 1, nullptr, nullptr);
  }
  const unsigned long long* cur = keysA; long n_cur = N;
  unsigned long long* bufs[2] = {keysB, keysC};
  int which = 0;
  while (true) {
    const long nch = chunks_of(n_cur), n_out = nch * k;
    unsigned long long* out = bufs[which];
    KtScope kt("topk_pass_kernel", 0.0, 8.0 * Q * (n_cur + n_out), s);
    hipLaunchKernelGGL(topk_pass_kernel, dim3((unsigned)nch, Q), dim3(1024), 0, s, cur, n_cur, k, out, n_out);
    cur = out; n_cur = n_out; which ^= 1;
    if (nch == 1) break;
  }
  hipLaunchKernelGGL(topk_decode_kernel, dim3((Q * k + 255) / 256), dim3(256), 0, s, cur, n_cur, Q, k, idx_out, score_out);
  return 0;
}

}  // namespace gr

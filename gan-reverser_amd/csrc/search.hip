// search.hip — recovered-noise cosine-similarity search (reference apply_r.lua:265-282 createImages loop,
// apply_r.lua:396-400 cosineSimilarity -> nn.CosineDistance) as three HBM-bound kernels:
//   1. needle_prep:   gather the Q needle rows, their squared norms (w22)
//   2. cos_keys:      one pass over emb[N][d]; per row the Q scores in the reference's exact op order
//                     (fp32 products, fp64 -- or fp32 -- sequential row sums, fp32 reciprocal/sqrt/mul),
//                     emitted as 64-bit sort keys  (orderable(score) << 32) | ~index
//   3. topk_pass:     per 2048-key chunk a bitonic sort in LDS keeps the k largest keys; repeated until one chunk is
//                     left.  Largest key first == (score desc, index asc): the tie order the oracle defines
//                     (the reference's table.sort is unstable, apply_r.lua:275).
#include "kernels.h"
#include <type_traits>

namespace gr {

constexpr int QG = 8;        // needles scored per pass of cos_keys (register accumulators)
constexpr int ROWS = 256;    // rows per workgroup
constexpr int DC = 32;       // columns staged per step
constexpr int CHUNK = 2048;  // keys per top-k workgroup

template <bool ACCF>
__global__ void needle_prep_kernel(const float* __restrict__ emb, int d, const long* __restrict__ rows, int Q,
                                   float* __restrict__ needles, float* __restrict__ w22) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= Q) return;
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

template <bool ACCF>
__global__ __launch_bounds__(ROWS) void cos_keys_kernel(const float* __restrict__ emb, long N, int d,
                                                         const float* __restrict__ needles, const float* __restrict__ w22,
                                                         int q0, int Q, unsigned long long* __restrict__ keys) {
  typedef typename std::conditional<ACCF, float, double>::type acc_t;
  __shared__ __attribute__((aligned(16))) float tile[ROWS * (DC + 1)];
  const int tid = threadIdx.x;
  const long r0 = (long)blockIdx.x * ROWS;
  const int nq = min(QG, Q - q0);
  const bool vec = (d & 3) == 0;                 // rows are 16-byte aligned: stage with float4 loads
  acc_t s1[QG], s3 = 0;
#pragma unroll
  for (int q = 0; q < QG; ++q) s1[q] = 0;
  for (int c0 = 0; c0 < d; c0 += DC) {
    const int dc = min(DC, d - c0);
    if (vec) {
      // 8 lanes cover one 128-byte row segment
      for (int e = tid; e < ROWS * (DC / 4); e += ROWS) {
        const int r = e / (DC / 4), c = (e - r * (DC / 4)) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r0 + r < N && c < dc) v = *reinterpret_cast<const float4*>(emb + (r0 + r) * (long)d + c0 + c);
        float* t = tile + r * (DC + 1) + c;
        t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
      }
    } else {
      for (int e = tid; e < ROWS * DC; e += ROWS) {
        const int r = e / DC, c = e - r * DC;
        float v = 0.f;
        if (r0 + r < N && c < dc) v = emb[(r0 + r) * (long)d + c0 + c];
        tile[r * (DC + 1) + c] = v;
      }
    }
    __syncthreads();
    const float* row = tile + tid * (DC + 1);
    const float* nd = needles + (long)q0 * d + c0;      // wave-uniform addresses: the needle values travel in SGPRs
    for (int c = 0; c < dc; ++c) {
      const float b = row[c];
      s3 += (acc_t)(b * b);
#pragma unroll
      for (int q = 0; q < QG; ++q)
        if (q < nq) s1[q] += (acc_t)(nd[(long)q * d + c] * b);
    }
    __syncthreads();
  }
  const long j = r0 + tid;
  if (j < N) {
    float w32 = (float)s3;
    w32 = w32 + 1e-12f;
    w32 = 1.f / w32;
#pragma unroll
    for (int q = 0; q < QG; ++q)
      if (q < nq) {
        float w = w22[q0 + q] * w32;
        w = sqrtf(w);
        const float sc = (float)s1[q] * w;
        keys[(long)(q0 + q) * N + j] = ((unsigned long long)orderable(sc) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)j);
      }
  }
}

// keys_in: [Q][n_in] ; keys_out: [Q][nchunks*k]
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

static long chunks_of(long n) { return (n + CHUNK - 1) / CHUNK; }

size_t cosine_topk_workspace_bytes(long N, int d, int Q, int k) {
  const long n1 = chunks_of(N) * k, n2 = chunks_of(n1) * k;
  return sizeof(float) * ((size_t)Q * d + Q + 8) + 512 + sizeof(unsigned long long) * (size_t)Q * (N + n1 + n2);
}

int launch_cosine_topk(const float* emb, long N, int d, const long* query_rows_dev, int Q, int k,
                       long* idx_out, float* score_out, int accf, void* workspace, hipStream_t s) {
  if (k > 1024 || k < 1 || k > N || N >= 0xFFFFFFFFl || d < 1 || d > 4096 * 4) return -1;
  // workspace carve: needles [Q][d] | w22 [Q] | keys A | keys B | keys C
  char* w = reinterpret_cast<char*>(workspace);
  float* needles = reinterpret_cast<float*>(w); w += sizeof(float) * (size_t)Q * d;
  float* w22 = reinterpret_cast<float*>(w); w += sizeof(float) * (size_t)((Q + 3) / 4 * 4);
  w = reinterpret_cast<char*>(((uintptr_t)w + 255) & ~(uintptr_t)255);
  unsigned long long* keysA = reinterpret_cast<unsigned long long*>(w);
  const long n1 = chunks_of(N) * k;
  unsigned long long* keysB = keysA + (size_t)Q * N;
  unsigned long long* keysC = keysB + (size_t)Q * n1;
  if (accf) hipLaunchKernelGGL(needle_prep_kernel<true>, dim3((Q + 63) / 64), dim3(64), 0, s, emb, d, query_rows_dev, Q, needles, w22);
  else hipLaunchKernelGGL(needle_prep_kernel<false>, dim3((Q + 63) / 64), dim3(64), 0, s, emb, d, query_rows_dev, Q, needles, w22);
  const unsigned nb = (unsigned)((N + ROWS - 1) / ROWS);
  for (int q0 = 0; q0 < Q; q0 += QG) {
    KtScope kt("cos_keys_kernel", 2.0 * N * d * (Q - q0 < QG ? Q - q0 : QG), 4.0 * N * d + 8.0 * N * (Q - q0 < QG ? Q - q0 : QG), s);
    if (accf) hipLaunchKernelGGL(cos_keys_kernel<true>, dim3(nb), dim3(ROWS), 0, s, emb, N, d, needles, w22, q0, Q, keysA);
    else hipLaunchKernelGGL(cos_keys_kernel<false>, dim3(nb), dim3(ROWS), 0, s, emb, N, d, needles, w22, q0, Q, keysA);
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

// kmeans.hip — clustering of the recovered noise vectors: apply_r.lua:197-217.
//
//   createClusterImages (apply_r.lua:197-231) = unsup.kmeans(attributes, nbClusters, nbIterations) followed by a
//   nearest-centroid pass in which the reference scores every (row, centroid) pair with cosineSimilarity (apply_r.lua:396-400)
//   and KEEPS THE MINIMUM (apply_r.lua:207-214: `dist < minDist`), i.e. the least similar centroid - preserved, flagged, and
//   selectable (take_min = 0 gives the most similar one).
//
// unsup.kmeans is an un-vendored luarock (koraykv/unsup, kmeans.lua; no version pinned by the reference).  Restated
// [upstream, from memory]: per iteration  c2 = 0.5 * sum(centroids^2, 2);  label(x) = argmax_j (centroid_j . x - c2_j)
// (first maximum wins);  centroid_j = sum of its rows / count (clusters that received no row keep their centroid);
// totalcounts += counts.  Upstream evaluates the products with sgemm and adds the member rows in fp32 (BLAS-defined order);
// here the dot products are sequential fp32 (j ascending, no FMA contraction: the oracle does the same, so labels are
// bit-reproducible) and the member sums are fp64, rounded to fp32 once before the fp32 division by the count.
// The initial centroids (upstream: k rows of N(0,1) from Torch's Mersenne twister, each divided by its norm) are an INPUT.
//
// Kernels (all HBM-bound: one pass over x[N][d] each):
//   kmeans_assign_kernel      thread = row, 256 rows x 32 columns staged through LDS per step, centroids via the scalar cache
//   kmeans_accumulate_kernel  workgroup = 512 consecutive rows, thread = column: fp64 sums per (cluster, column) in LDS, in row
//                             order (deterministic), one partial block per workgroup
//   kmeans_update_kernel      partial blocks summed in workgroup order; new centroids, c2, counts
//   cosine_assign_kernel      thread = row: the k cosine similarities in nn.CosineDistance's exact op order (fp32 products,
//                             fp64 sequential row sums, fp32 reciprocal / sqrt / multiply) and their arg-min or arg-max
#include "kernels.h"

namespace gr {

constexpr int KM_ROWS = 256, KM_DC = 32, KM_KMAX = 32, KM_RPB = 512;

__global__ __launch_bounds__(KM_ROWS) void kmeans_assign_kernel(const float* __restrict__ x, long N, int d, const float* __restrict__ cent,
                                                                 const float* __restrict__ c2, int k, int* __restrict__ labels) {
  __shared__ __attribute__((aligned(16))) float tile[KM_ROWS * (KM_DC + 1)];
  const int tid = threadIdx.x;
  const long r0 = (long)blockIdx.x * KM_ROWS;
  float s[KM_KMAX];
#pragma unroll
  for (int j = 0; j < KM_KMAX; ++j) s[j] = 0.f;
  for (int c0 = 0; c0 < d; c0 += KM_DC) {
    const int dc = min(KM_DC, d - c0);
    for (int e = tid; e < KM_ROWS * KM_DC; e += KM_ROWS) {
      const int r = e / KM_DC, c = e - r * KM_DC;
      tile[r * (KM_DC + 1) + c] = (r0 + r < N && c < dc) ? x[(r0 + r) * (long)d + c0 + c] : 0.f;
    }
    __syncthreads();
    const float* row = tile + tid * (KM_DC + 1);
    for (int c = 0; c < dc; ++c) {
      const float b = row[c];
#pragma unroll
      for (int j = 0; j < KM_KMAX; ++j)
        if (j < k) s[j] = s[j] + cent[(long)j * d + c0 + c] * b;      // wave-uniform centroid address; no contraction (Makefile)
    }
    __syncthreads();
  }
  if (r0 + tid < N) {
    float best = 0.f; int bi = 0;
#pragma unroll
    for (int j = 0; j < KM_KMAX; ++j)
      if (j < k) { const float v = s[j] - c2[j]; if (j == 0 || v > best) { best = v; bi = j; } }
    labels[r0 + tid] = bi;
  }
}

__global__ __launch_bounds__(256) void kmeans_accumulate_kernel(const float* __restrict__ x, long N, int d, const int* __restrict__ labels, int k,
                                                                 double* __restrict__ part_sum /*[nblk][k][d]*/, int* __restrict__ part_cnt /*[nblk][k]*/) {
  extern __shared__ double acc[];                      // [k][d]
  __shared__ int lab[KM_RPB];
  const int tid = threadIdx.x;
  const long r0 = (long)blockIdx.x * KM_RPB;
  const int nr = (int)min((long)KM_RPB, N - r0);
  for (int e = tid; e < k * d; e += 256) acc[e] = 0.0;
  for (int e = tid; e < nr; e += 256) lab[e] = labels[r0 + e];
  __syncthreads();
  for (int t = tid; t < d; t += 256)                   // thread owns column t: adds the rows in order
    for (int r = 0; r < nr; ++r) acc[lab[r] * d + t] += (double)x[(r0 + r) * (long)d + t];
  if (tid < k) { int cnt = 0; for (int r = 0; r < nr; ++r) cnt += lab[r] == tid; part_cnt[(long)blockIdx.x * k + tid] = cnt; }
  __syncthreads();
  for (int e = tid; e < k * d; e += 256) part_sum[(long)blockIdx.x * k * d + e] = acc[e];
}

// one workgroup per cluster: sums the partial blocks in order, writes the new centroid row, its c2 and the counts
__global__ __launch_bounds__(256) void kmeans_update_kernel(const double* __restrict__ part_sum, const int* __restrict__ part_cnt, int nblk, int k, int d,
                                                             float* __restrict__ cent, float* __restrict__ c2, float* __restrict__ counts, float* __restrict__ totalcounts) {
  __shared__ double red[256];
  __shared__ int s_cnt;
  const int j = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) { int c = 0; for (int b = 0; b < nblk; ++b) c += part_cnt[(long)b * k + j]; s_cnt = c; }
  __syncthreads();
  const int cnt = s_cnt;
  double sq = 0.0;                                     // this thread's share of sum(centroid^2) (accreal = double, fixed order below)
  for (int t = tid; t < d; t += 256) {
    float v = cent[(long)j * d + t];
    if (cnt != 0) {
      double sm = 0.0;
      for (int b = 0; b < nblk; ++b) sm += part_sum[((long)b * k + j) * d + t];
      v = (float)sm / (float)cnt;                      // summation[i]:div(counts[i]) on float tensors
      cent[(long)j * d + t] = v;
    }
    const float p = v * v;                             // pow(centroids, 2) is a float tensor
    sq += (double)p;
  }
  red[tid] = sq;
  __syncthreads();
  if (tid == 0) {
    // torch.sum over the row: sequential in column order.  Thread t holds columns t, t+256, ...: for d <= 256 (every case of
    // the reference: d = noise dimension) summing red[] in thread order IS column order.
    double tot = 0.0;
    for (int t = 0; t < 256; ++t) tot += red[t];
    c2[j] = (float)tot * 0.5f;
    counts[j] = (float)cnt;
    totalcounts[j] += (float)cnt;
  }
}

__global__ void kmeans_c2_kernel(const float* __restrict__ cent, int k, int d, float* __restrict__ c2) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= k) return;
  double s = 0.0;
  for (int t = 0; t < d; ++t) { const float p = cent[(long)j * d + t] * cent[(long)j * d + t]; s += (double)p; }
  c2[j] = (float)s * 0.5f;
}

// w32[j] = 1 / (sum(c_j^2) + 1e-12) in nn.CosineDistance's op order
__global__ void cosine_centroid_prep_kernel(const float* __restrict__ cent, int k, int d, float* __restrict__ w32) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= k) return;
  double s = 0.0;
  for (int t = 0; t < d; ++t) { const float p = cent[(long)j * d + t] * cent[(long)j * d + t]; s += (double)p; }
  float w = (float)s;
  w = w + 1e-12f;
  w32[j] = 1.f / w;
}

__global__ __launch_bounds__(KM_ROWS) void cosine_assign_kernel(const float* __restrict__ x, long N, int d, const float* __restrict__ cent,
                                                                 const float* __restrict__ w32, int k, int take_min,
                                                                 int* __restrict__ labels, float* __restrict__ sims) {
  __shared__ __attribute__((aligned(16))) float tile[KM_ROWS * (KM_DC + 1)];
  const int tid = threadIdx.x;
  const long r0 = (long)blockIdx.x * KM_ROWS;
  float best = 0.f; int bi = 0;
  for (int j0 = 0; j0 < k; j0 += KM_KMAX) {            // KM_KMAX centroids per pass over the row (one pass for k <= 32)
    const int kk = min(KM_KMAX, k - j0);
    double s1[KM_KMAX], s2 = 0.0;
#pragma unroll
    for (int j = 0; j < KM_KMAX; ++j) s1[j] = 0.0;
    for (int c0 = 0; c0 < d; c0 += KM_DC) {
      const int dc = min(KM_DC, d - c0);
      for (int e = tid; e < KM_ROWS * KM_DC; e += KM_ROWS) {
        const int r = e / KM_DC, c = e - r * KM_DC;
        tile[r * (KM_DC + 1) + c] = (r0 + r < N && c < dc) ? x[(r0 + r) * (long)d + c0 + c] : 0.f;
      }
      __syncthreads();
      const float* row = tile + tid * (KM_DC + 1);
      for (int c = 0; c < dc; ++c) {
        const float a = row[c];
        const float aa = a * a;
        s2 += (double)aa;
#pragma unroll
        for (int j = 0; j < KM_KMAX; ++j)
          if (j < kk) { const float p = a * cent[(long)(j0 + j) * d + c0 + c]; s1[j] += (double)p; }
      }
      __syncthreads();
    }
    float w22 = (float)s2;
    w22 = w22 + 1e-12f; w22 = 1.f / w22;
#pragma unroll
    for (int j = 0; j < KM_KMAX; ++j)
      if (j < kk) {
        float w = w22 * w32[j0 + j];
        w = sqrtf(w);
        const float sc = (float)s1[j] * w;
        const bool first = j0 + j == 0;
        if (first || (take_min ? sc < best : sc > best)) { best = sc; bi = j0 + j; }
      }
  }
  if (r0 + tid < N) { labels[r0 + tid] = bi; sims[r0 + tid] = best; }
}

size_t kmeans_workspace_bytes(long N, int d, int k) {
  const long nblk = (N + KM_RPB - 1) / KM_RPB;
  return sizeof(double) * (size_t)nblk * k * d + sizeof(int) * (size_t)nblk * k + sizeof(int) * (size_t)N + 1024;
}

// x, cent (in: initial, out: final), c2 / counts / totalcounts [k] are device buffers; workspace from kmeans_workspace_bytes
int launch_kmeans(const float* x, long N, int d, int k, int niter, float* cent, float* c2, float* counts, float* totalcounts,
                  int* labels_out /*nullable: labels of the last iteration*/, void* workspace, hipStream_t s) {
  if (k > KM_KMAX || (size_t)k * d * sizeof(double) > 60 * 1024 || d > 256) return 1;
  const long nblk = (N + KM_RPB - 1) / KM_RPB;
  double* part_sum = reinterpret_cast<double*>(workspace);
  int* part_cnt = reinterpret_cast<int*>(part_sum + (size_t)nblk * k * d);
  int* labels = labels_out ? labels_out : part_cnt + (size_t)nblk * k;
  (void)hipMemsetAsync(totalcounts, 0, sizeof(float) * k, s);
  hipLaunchKernelGGL(kmeans_c2_kernel, dim3(1), dim3(64), 0, s, cent, k, d, c2);
  const size_t lds = sizeof(double) * (size_t)k * d;
  for (int it = 0; it < niter; ++it) {
    { KtScope kt("kmeans_assign_kernel", 2.0 * N * d * k, 4.0 * N * d, s);
      hipLaunchKernelGGL(kmeans_assign_kernel, dim3((unsigned)((N + KM_ROWS - 1) / KM_ROWS)), dim3(KM_ROWS), 0, s, x, N, d, cent, c2, k, labels); }
    { KtScope kt("kmeans_accumulate_kernel", (double)N * d, 4.0 * N * d, s);
      hipLaunchKernelGGL(kmeans_accumulate_kernel, dim3((unsigned)nblk), dim3(256), lds, s, x, N, d, labels, k, part_sum, part_cnt); }
    hipLaunchKernelGGL(kmeans_update_kernel, dim3(k), dim3(256), 0, s, part_sum, part_cnt, (int)nblk, k, d, cent, c2, counts, totalcounts);
  }
  return 0;
}

int launch_cosine_assign(const float* x, long N, int d, const float* cent, int k, int take_min, float* w32 /*[k] scratch*/,
                         int* labels, float* sims, hipStream_t s) {
  if (k <= 0) return 1;
  hipLaunchKernelGGL(cosine_centroid_prep_kernel, dim3((k + 63) / 64), dim3(64), 0, s, cent, k, d, w32);
  KtScope kt("cosine_assign_kernel", 2.0 * N * d * k, 4.0 * N * d, s);
  hipLaunchKernelGGL(cosine_assign_kernel, dim3((unsigned)((N + KM_ROWS - 1) / KM_ROWS)), dim3(KM_ROWS), 0, s, x, N, d, cent, w32, k, take_min, labels, sims);
  return 0;
}

}  // namespace gr

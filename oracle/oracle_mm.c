/*
 * oracle_mm.c — the 3x3 convolutions of the CPU oracle as im2col + blocked sgemm per sample (TEST INFRASTRUCTURE /
 * CPU BASELINE, see ganrev_oracle.h).
 *
 * This is the STRUCTURE Torch7's CPU path has for nn.SpatialConvolution (THNN SpatialConvolutionMM [upstream, from memory]:
 * per sample `finput = unfolded_copy(input)` of shape [Cin*9][H*W], then output = weight[Cout][Cin*9] x finput + bias through BLAS
 * sgemm; updateGradInput = weight^T x gradOutput folded back by unfolded_acc; accGradParameters = gradOutput x finput^T) and what
 * SURVEY.md section 8d names for the CPU baseline column.  The reference's `utils/nn_utils.lua:402-404` touches the same
 * `finput` buffers.  oracle_blas.c's direct loops stay the parity oracle (pinned by tests/golden); this path is selected with
 * go_set_conv_impl(1), pinned against the direct loops (tests/test_oracle_golden.py) and timed by bench.py as cpu_baseline.
 * The sgemm is a plain cache-blocked, register-tiled C loop nest (OpenMP over samples) - a baseline, not a tuned BLAS.
 */
#include "ganrev_oracle.h"
#include <omp.h>
#include <stdlib.h>
#include <string.h>

/* finput[(i*9 + ky*3 + kx)][y*W + x] = in[i][y+ky-1][x+kx-1] (zero outside): THNN unfolded_copy for 3x3 s1 p1 */
static void unfold3(const float* in, float* col, int Cin, int H, int W) {
  const long HW = (long)H * W;
  for (int i = 0; i < Cin; ++i)
    for (int ky = 0; ky < 3; ++ky)
      for (int kx = 0; kx < 3; ++kx) {
        float* c = col + ((long)i * 9 + ky * 3 + kx) * HW;
        const float* ip = in + (long)i * HW;
        for (int y = 0; y < H; ++y) {
          const int sy = y + ky - 1;
          float* crow = c + (long)y * W;
          if (sy < 0 || sy >= H) { memset(crow, 0, sizeof(float) * W); continue; }
          const float* irow = ip + (long)sy * W;
          if (kx == 0) { crow[0] = 0.f; memcpy(crow + 1, irow, sizeof(float) * (W - 1)); }
          else if (kx == 1) memcpy(crow, irow, sizeof(float) * W);
          else { memcpy(crow, irow + 1, sizeof(float) * (W - 1)); crow[W - 1] = 0.f; }
        }
      }
}

/* gin[i][y+ky-1][x+kx-1] += col[(i*9+ky*3+kx)][y*W+x]: THNN unfolded_acc */
static void fold3_acc(const float* col, float* gin, int Cin, int H, int W) {
  const long HW = (long)H * W;
  memset(gin, 0, sizeof(float) * Cin * HW);
  for (int i = 0; i < Cin; ++i)
    for (int ky = 0; ky < 3; ++ky)
      for (int kx = 0; kx < 3; ++kx) {
        const float* c = col + ((long)i * 9 + ky * 3 + kx) * HW;
        float* gp = gin + (long)i * HW;
        const int x0 = kx == 0 ? 1 : 0, x1 = kx == 2 ? W - 1 : W;
        for (int y = 0; y < H; ++y) {
          const int sy = y + ky - 1;
          if (sy < 0 || sy >= H) continue;
          float* grow = gp + (long)sy * W + (kx - 1);
          const float* crow = c + (long)y * W;
#pragma omp simd
          for (int x = x0; x < x1; ++x) grow[x] += crow[x];
        }
      }
}

/* C[M][N] (+)= A[M][K] * B[K][N], row-major; blocks of 128 k x 512 n (B block in L2), 4-row register tile */
static void sgemm_nn(const float* A, const float* Bm, float* Cm, int M, int N, int K, int accumulate) {
  enum { KB = 128, NB = 512 };
  if (!accumulate) memset(Cm, 0, sizeof(float) * (size_t)M * N);
  for (int n0 = 0; n0 < N; n0 += NB) {
    const int nb = N - n0 < NB ? N - n0 : NB;
    for (int k0 = 0; k0 < K; k0 += KB) {
      const int kb = K - k0 < KB ? K - k0 : KB;
      int m = 0;
      for (; m + 4 <= M; m += 4) {
        float* c0 = Cm + (long)m * N + n0; float* c1 = c0 + N; float* c2 = c1 + N; float* c3 = c2 + N;
        for (int k = 0; k < kb; ++k) {
          const float a0 = A[(long)m * K + k0 + k], a1 = A[(long)(m + 1) * K + k0 + k], a2 = A[(long)(m + 2) * K + k0 + k], a3 = A[(long)(m + 3) * K + k0 + k];
          const float* b = Bm + (long)(k0 + k) * N + n0;
#pragma omp simd
          for (int n = 0; n < nb; ++n) { const float bv = b[n]; c0[n] += a0 * bv; c1[n] += a1 * bv; c2[n] += a2 * bv; c3[n] += a3 * bv; }
        }
      }
      for (; m < M; ++m) {
        float* c0 = Cm + (long)m * N + n0;
        for (int k = 0; k < kb; ++k) {
          const float a0 = A[(long)m * K + k0 + k];
          const float* b = Bm + (long)(k0 + k) * N + n0;
#pragma omp simd
          for (int n = 0; n < nb; ++n) c0[n] += a0 * b[n];
        }
      }
    }
  }
}

/* C[M][J] += A[M][N] * B[J][N]^T (dot products over N): 2 x 4 register tile */
static void sgemm_nt_acc(const float* A, const float* Bm, float* Cm, int M, int J, int N) {
  for (int m = 0; m < M; m += 2) {
    const int m2 = m + 1 < M;
    const float* a0 = A + (long)m * N; const float* a1 = A + (long)(m + m2) * N;
    int j = 0;
    for (; j + 4 <= J; j += 4) {
      const float* b0 = Bm + (long)j * N; const float* b1 = b0 + N; const float* b2 = b1 + N; const float* b3 = b2 + N;
      float s00 = 0, s01 = 0, s02 = 0, s03 = 0, s10 = 0, s11 = 0, s12 = 0, s13 = 0;
#pragma omp simd reduction(+ : s00, s01, s02, s03, s10, s11, s12, s13)
      for (int n = 0; n < N; ++n) {
        const float x0 = a0[n], x1 = a1[n];
        s00 += x0 * b0[n]; s01 += x0 * b1[n]; s02 += x0 * b2[n]; s03 += x0 * b3[n];
        s10 += x1 * b0[n]; s11 += x1 * b1[n]; s12 += x1 * b2[n]; s13 += x1 * b3[n];
      }
      float* c0 = Cm + (long)m * J + j;
      c0[0] += s00; c0[1] += s01; c0[2] += s02; c0[3] += s03;
      if (m2) { float* c1 = c0 + J; c1[0] += s10; c1[1] += s11; c1[2] += s12; c1[3] += s13; }
    }
    for (; j < J; ++j) {
      const float* b0 = Bm + (long)j * N;
      float s0 = 0, s1 = 0;
#pragma omp simd reduction(+ : s0, s1)
      for (int n = 0; n < N; ++n) { s0 += a0[n] * b0[n]; s1 += a1[n] * b0[n]; }
      Cm[(long)m * J + j] += s0;
      if (m2) Cm[(long)(m + 1) * J + j] += s1;
    }
  }
}

void go_conv3_forward_mm(const float* in, const float* w, const float* bias, float* out, int B, int Cin, int Cout, int H, int W) {
  const long HW = (long)H * W; const int K = Cin * 9;
#pragma omp parallel
  {
    float* col = (float*)malloc(sizeof(float) * (size_t)K * HW);
#pragma omp for schedule(static)
    for (int b = 0; b < B; ++b) {
      float* o = out + (long)b * Cout * HW;
      unfold3(in + (long)b * Cin * HW, col, Cin, H, W);
      for (int c = 0; c < Cout; ++c) { const float bv = bias ? bias[c] : 0.f; float* op = o + (long)c * HW; for (long p = 0; p < HW; ++p) op[p] = bv; }
      sgemm_nn(w, col, o, Cout, (int)HW, K, 1);
    }
    free(col);
  }
}

void go_conv3_backward_data_mm(const float* gout, const float* w, float* gin, int B, int Cin, int Cout, int H, int W) {
  const long HW = (long)H * W; const int K = Cin * 9;
  float* wt = (float*)malloc(sizeof(float) * (size_t)K * Cout);          /* weight^T [Cin*9][Cout] */
  for (int o = 0; o < Cout; ++o) for (int k = 0; k < K; ++k) wt[(long)k * Cout + o] = w[(long)o * K + k];
#pragma omp parallel
  {
    float* col = (float*)malloc(sizeof(float) * (size_t)K * HW);
#pragma omp for schedule(static)
    for (int b = 0; b < B; ++b) {
      sgemm_nn(wt, gout + (long)b * Cout * HW, col, K, (int)HW, Cout, 0);
      fold3_acc(col, gin + (long)b * Cin * HW, Cin, H, W);
    }
    free(col);
  }
  free(wt);
}

void go_conv3_backward_weight_mm(const float* in, const float* gout, float* gw, float* gb, int B, int Cin, int Cout, int H, int W) {
  const long HW = (long)H * W; const int K = Cin * 9;
  const int T = omp_get_max_threads();
  float* part = (float*)calloc((size_t)T * Cout * K, sizeof(float));     /* one accumulator per thread, added in thread order */
#pragma omp parallel
  {
    float* col = (float*)malloc(sizeof(float) * (size_t)K * HW);
    float* acc = part + (size_t)omp_get_thread_num() * Cout * K;
#pragma omp for schedule(static)
    for (int b = 0; b < B; ++b) {
      unfold3(in + (long)b * Cin * HW, col, Cin, H, W);
      sgemm_nt_acc(gout + (long)b * Cout * HW, col, acc, Cout, K, (int)HW);
    }
    free(col);
  }
  for (int t = 0; t < T; ++t) { const float* a = part + (size_t)t * Cout * K; for (long e = 0; e < (long)Cout * K; ++e) gw[e] += a[e]; }
  free(part);
  if (gb) {
#pragma omp parallel for schedule(static)
    for (int o = 0; o < Cout; ++o) {
      float s = 0.f;
      for (int b = 0; b < B; ++b) {
        const float* go = gout + ((long)b * Cout + o) * HW;
        float sb = 0.f;
#pragma omp simd reduction(+ : sb)
        for (long p = 0; p < HW; ++p) sb += go[p];
        s += sb;
      }
      gb[o] += s;
    }
  }
}

/*
 * oracle_blas.c — contraction ops of the CPU oracle (TEST INFRASTRUCTURE, see ganrev_oracle.h).
 *
 * Restates nn.SpatialConvolution(…,3,3,1,1,1,1) and nn.Linear as the reference instantiates them
 * (models.lua:409-436 for R, models.lua:115,122,128,132 for G; models.lua:447,451 Linear).
 * Upstream THNN evaluates these as im2col + BLAS sgemm in fp32; summation order there is
 * BLAS-defined, so parity for these ops is tolerance-based (1e-4), never bit-exact.  Here the
 * sums run in fp32 too (compiled with FMA contraction allowed, like a BLAS would be).
 */
#include "ganrev_oracle.h"
#include <string.h>

/* 0 (default): the direct 9-tap loop nests below - the parity oracle, pinned by tests/golden.  1: im2col + blocked sgemm per
 * sample (oracle_mm.c: the structure of THNN's SpatialConvolutionMM, SURVEY.md 8d's CPU-baseline structure). */
static int g_conv_impl = 0;
void go_set_conv_impl(int impl) { g_conv_impl = impl ? 1 : 0; }
int go_get_conv_impl(void) { return g_conv_impl; }

/* out[b,o,y,x] = bias[o] + sum_{i,ky,kx} w[o,i,ky,kx] * in[b,i,y+ky-1,x+kx-1]   (cross-correlation, zero pad) */
void go_conv3_forward(const float* in, const float* w, const float* bias, float* out,
                      int B, int Cin, int Cout, int H, int W) {
  if (g_conv_impl) { go_conv3_forward_mm(in, w, bias, out, B, Cin, Cout, H, W); return; }
  const long HW = (long)H * W;
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int o = 0; o < Cout; ++o) {
      float* op = out + ((long)b * Cout + o) * HW;
      const float bv = bias ? bias[o] : 0.f;
      for (long p = 0; p < HW; ++p) op[p] = bv;
      for (int i = 0; i < Cin; ++i) {
        const float* ip = in + ((long)b * Cin + i) * HW;
        const float* wp = w + ((long)o * Cin + i) * 9;
        for (int ky = 0; ky < 3; ++ky) {
          const int y0 = ky == 0 ? 1 : 0, y1 = ky == 2 ? H - 1 : H;
          for (int kx = 0; kx < 3; ++kx) {
            const int x0 = kx == 0 ? 1 : 0, x1 = kx == 2 ? W - 1 : W;
            const float wv = wp[ky * 3 + kx];
            for (int y = y0; y < y1; ++y) {
              float* orow = op + (long)y * W;
              const float* irow = ip + (long)(y + ky - 1) * W + (kx - 1);
#pragma omp simd
              for (int x = x0; x < x1; ++x) orow[x] += wv * irow[x];
            }
          }
        }
      }
    }
}

/* gin[b,i,y+ky-1,x+kx-1] += w[o,i,ky,kx] * gout[b,o,y,x]   (updateGradInput; gin is overwritten) */
void go_conv3_backward_data(const float* gout, const float* w, float* gin,
                            int B, int Cin, int Cout, int H, int W) {
  if (g_conv_impl) { go_conv3_backward_data_mm(gout, w, gin, B, Cin, Cout, H, W); return; }
  const long HW = (long)H * W;
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < Cin; ++i) {
      float* gi = gin + ((long)b * Cin + i) * HW;
      memset(gi, 0, sizeof(float) * HW);
      for (int o = 0; o < Cout; ++o) {
        const float* go = gout + ((long)b * Cout + o) * HW;
        const float* wp = w + ((long)o * Cin + i) * 9;
        for (int ky = 0; ky < 3; ++ky) {
          const int y0 = ky == 0 ? 1 : 0, y1 = ky == 2 ? H - 1 : H;
          for (int kx = 0; kx < 3; ++kx) {
            const int x0 = kx == 0 ? 1 : 0, x1 = kx == 2 ? W - 1 : W;
            const float wv = wp[ky * 3 + kx];
            for (int y = y0; y < y1; ++y) {
              const float* grow = go + (long)y * W;
              float* irow = gi + (long)(y + ky - 1) * W + (kx - 1);
#pragma omp simd
              for (int x = x0; x < x1; ++x) irow[x] += wv * grow[x];
            }
          }
        }
      }
    }
}

/* gw[o,i,ky,kx] += sum_{b,y,x} gout[b,o,y,x]*in[b,i,y+ky-1,x+kx-1] ; gb[o] += sum gout   (accGradParameters, scale 1) */
void go_conv3_backward_weight(const float* in, const float* gout, float* gw, float* gb,
                              int B, int Cin, int Cout, int H, int W) {
  if (g_conv_impl) { go_conv3_backward_weight_mm(in, gout, gw, gb, B, Cin, Cout, H, W); return; }
  const long HW = (long)H * W;
#pragma omp parallel for collapse(2) schedule(static)
  for (int o = 0; o < Cout; ++o)
    for (int i = 0; i < Cin; ++i) {
      float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (int b = 0; b < B; ++b) {
        const float* go = gout + ((long)b * Cout + o) * HW;
        const float* ip = in + ((long)b * Cin + i) * HW;
        for (int ky = 0; ky < 3; ++ky) {
          const int y0 = ky == 0 ? 1 : 0, y1 = ky == 2 ? H - 1 : H;
          for (int kx = 0; kx < 3; ++kx) {
            const int x0 = kx == 0 ? 1 : 0, x1 = kx == 2 ? W - 1 : W;
            float s = 0.f;
            for (int y = y0; y < y1; ++y) {
              const float* grow = go + (long)y * W;
              const float* irow = ip + (long)(y + ky - 1) * W + (kx - 1);
#pragma omp simd reduction(+ : s)
              for (int x = x0; x < x1; ++x) s += grow[x] * irow[x];
            }
            acc[ky * 3 + kx] += s;
          }
        }
      }
      float* g = gw + ((long)o * Cin + i) * 9;
      for (int t = 0; t < 9; ++t) g[t] += acc[t];
    }
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

/* ---- nn.SpatialConvolution(Cin, Cout, K, K, 1, 1, (K-1)/2, (K-1)/2), K odd: the D network's 5x5 layer (models.lua:275,297 createNxN;
 * SURVEY.md 8f rank 4).  Same loops as the 3x3 functions above with the window size a parameter; THNN evaluates it as im2col +
 * sgemm per sample (SpatialConvolutionMM), so parity is tolerance-based here too. ---- */
void go_convk_forward(const float* in, const float* w, const float* bias, float* out,
                      int B, int Cin, int Cout, int H, int W, int K) {
  const long HW = (long)H * W; const int P = (K - 1) / 2;
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int o = 0; o < Cout; ++o) {
      float* op = out + ((long)b * Cout + o) * HW;
      const float bv = bias ? bias[o] : 0.f;
      for (long p = 0; p < HW; ++p) op[p] = bv;
      for (int i = 0; i < Cin; ++i) {
        const float* ip = in + ((long)b * Cin + i) * HW;
        const float* wp = w + ((long)o * Cin + i) * K * K;
        for (int ky = 0; ky < K; ++ky) {
          const int y0 = P - ky > 0 ? P - ky : 0, y1 = H + P - ky < H ? H + P - ky : H;
          for (int kx = 0; kx < K; ++kx) {
            const int x0 = P - kx > 0 ? P - kx : 0, x1 = W + P - kx < W ? W + P - kx : W;
            const float wv = wp[ky * K + kx];
            for (int y = y0; y < y1; ++y) {
              float* orow = op + (long)y * W;
              const float* irow = ip + (long)(y + ky - P) * W + (kx - P);
#pragma omp simd
              for (int x = x0; x < x1; ++x) orow[x] += wv * irow[x];
            }
          }
        }
      }
    }
}

void go_convk_backward_data(const float* gout, const float* w, float* gin,
                            int B, int Cin, int Cout, int H, int W, int K) {
  const long HW = (long)H * W; const int P = (K - 1) / 2;
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < Cin; ++i) {
      float* gi = gin + ((long)b * Cin + i) * HW;
      memset(gi, 0, sizeof(float) * HW);
      for (int o = 0; o < Cout; ++o) {
        const float* go = gout + ((long)b * Cout + o) * HW;
        const float* wp = w + ((long)o * Cin + i) * K * K;
        for (int ky = 0; ky < K; ++ky) {
          const int y0 = P - ky > 0 ? P - ky : 0, y1 = H + P - ky < H ? H + P - ky : H;
          for (int kx = 0; kx < K; ++kx) {
            const int x0 = P - kx > 0 ? P - kx : 0, x1 = W + P - kx < W ? W + P - kx : W;
            const float wv = wp[ky * K + kx];
            for (int y = y0; y < y1; ++y) {
              const float* grow = go + (long)y * W;
              float* irow = gi + (long)(y + ky - P) * W + (kx - P);
#pragma omp simd
              for (int x = x0; x < x1; ++x) irow[x] += wv * grow[x];
            }
          }
        }
      }
    }
}

void go_convk_backward_weight(const float* in, const float* gout, float* gw, float* gb,
                              int B, int Cin, int Cout, int H, int W, int K) {
  const long HW = (long)H * W; const int P = (K - 1) / 2;
#pragma omp parallel for collapse(2) schedule(static)
  for (int o = 0; o < Cout; ++o)
    for (int i = 0; i < Cin; ++i) {
      float* g = gw + ((long)o * Cin + i) * K * K;
      for (int ky = 0; ky < K; ++ky) {
        const int y0 = P - ky > 0 ? P - ky : 0, y1 = H + P - ky < H ? H + P - ky : H;
        for (int kx = 0; kx < K; ++kx) {
          const int x0 = P - kx > 0 ? P - kx : 0, x1 = W + P - kx < W ? W + P - kx : W;
          float acc = 0.f;
          for (int b = 0; b < B; ++b) {
            const float* go = gout + ((long)b * Cout + o) * HW;
            const float* ip = in + ((long)b * Cin + i) * HW;
            float s = 0.f;
            for (int y = y0; y < y1; ++y) {
              const float* grow = go + (long)y * W;
              const float* irow = ip + (long)(y + ky - P) * W + (kx - P);
#pragma omp simd reduction(+ : s)
              for (int x = x0; x < x1; ++x) s += grow[x] * irow[x];
            }
            acc += s;
          }
          g[ky * K + kx] += acc;
        }
      }
    }
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

/* y = x W^T + b   with W[out][in]  (nn.Linear:updateOutput) */
void go_linear_forward(const float* in, const float* w, const float* bias, float* out, int B, int I, int O) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int o = 0; o < O; ++o) {
      const float* x = in + (long)b * I;
      const float* wr = w + (long)o * I;
      float s = 0.f;
#pragma omp simd reduction(+ : s)
      for (int k = 0; k < I; ++k) s += x[k] * wr[k];
      out[(long)b * O + o] = s + (bias ? bias[o] : 0.f);
    }
}

/* gin = gout W */
void go_linear_backward_data(const float* gout, const float* w, float* gin, int B, int I, int O) {
#pragma omp parallel for schedule(static)
  for (int b = 0; b < B; ++b) {
    float* gi = gin + (long)b * I;
    memset(gi, 0, sizeof(float) * I);
    for (int o = 0; o < O; ++o) {
      const float g = gout[(long)b * O + o];
      const float* wr = w + (long)o * I;
#pragma omp simd
      for (int k = 0; k < I; ++k) gi[k] += g * wr[k];
    }
  }
}

/* gw += gout^T x ; gb += sum_b gout */
void go_linear_backward_weight(const float* in, const float* gout, float* gw, float* gb, int B, int I, int O) {
#pragma omp parallel for schedule(static)
  for (int o = 0; o < O; ++o) {
    float* g = gw + (long)o * I;
    float sb = 0.f;
    for (int b = 0; b < B; ++b) {
      const float gv = gout[(long)b * O + o];
      const float* x = in + (long)b * I;
      sb += gv;
#pragma omp simd
      for (int k = 0; k < I; ++k) g[k] += gv * x[k];
    }
    if (gb) gb[o] += sb;
  }
}

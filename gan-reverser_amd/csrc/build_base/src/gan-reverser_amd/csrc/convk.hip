// convk.hip — nn.SpatialConvolution(Cin, Cout, K, K, 1, 1, (K-1)/2, (K-1)/2) for odd K other than 3, and nn.PReLU's slope
// gradient: the module types the D network adds to the path (reference models.lua:272-337 create_D2: createNxN(128, 64, 5, ..)
// at :297 and nn.PReLU at :276; trained by adversarial.lua:37-205).  SURVEY.md 8f rank 4.
//
// fp32 VALU kernels (no arithmetic modes): one K x K layer sits in D next to six 3x3 layers that run on the MFMA kernels of
// conv.hip, so these only have to be far from the critical path, not at a roofline:
//   * forward / data gradient: one direct kernel.  A workgroup owns a 16x16 pixel tile of one image and COT output channels;
//     input channels are staged through LDS eight at a time (with the halo), the weights of one (channel, tap) for the COT
//     output channels are ONE uniform (scalar-cache) load of a tap-major image `wt[ci][ky][kx][co]` built per call, so the
//     inner loop is one LDS read + COT v_fmac with an SGPR operand.  The data gradient is the same kernel on the
//     transposed + flipped image.
//   * weight gradient: thread = (4 output channels) x (1 input channel), all K*K taps in registers (100 accumulators at K = 5);
//     dy tiles pixel-major in LDS (one 16-byte read gives the four channels), x tiles with the halo; the batch is split over
//     blockIdx.z and the per-split results are summed in split order by a second kernel (deterministic, += into the gradient).
#include "kernels.h"

namespace gr {

constexpr int KC_CI = 8;          // input channels staged per round (direct kernel)
constexpr int KC_TILE = 16;       // 16 x 16 output pixels per workgroup


constexpr int KC_KS = 4;         // channel slices of the sliced direct kernel
size_t convk_image_floats(int cin_eff, int cout_eff, int K) { return (size_t)round_up(cin_eff, KC_CI * KC_KS) * K * K * round_up(cout_eff, 32); }

// wt[ci][ky][kx][co], zero rows / columns up to the padded extents.  bwd = 0: ci = input plane i, co = output plane o,
// wt = w[o][i][ky][kx].  bwd = 1 (data gradient; the kernel's input is dy): ci = o, co = i, wt = w[o][i][K-1-ky][K-1-kx].
__global__ void convk_weight_image_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cin, int Cout, int K, int bwd,
                                          int CiP, int CoP) {
  const long total = (long)CiP * K * K * CoP;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int co = (int)(e % CoP); long r = e / CoP;
    const int kx = (int)(r % K); r /= K;
    const int ky = (int)(r % K); const int ci = (int)(r / K);
    float v = 0.f;
    if (!bwd) { if (ci < Cin && co < Cout) v = w[(((long)co * Cin + ci) * K + ky) * K + kx]; }
    else if (ci < Cout && co < Cin) v = w[(((long)ci * Cin + co) * K + (K - 1 - ky)) * K + (K - 1 - kx)];
    wt[e] = v;
  }
}

template <int K, int COT, int NI>
__global__ __launch_bounds__(256) void convk_direct_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           int B, int Cin, int Cout, int CoP, int H, int W, int tiles_x) {
  // NI images per workgroup share every weight load (NI = 2: see the launcher - measured slower)
  constexpr int P = (K - 1) / 2, TW = KC_TILE + K - 1, TWS = TW + 1;
  __shared__ float xt[NI][KC_CI][TW][TWS];
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
  const int tile = blockIdx.x, y0 = (tile / tiles_x) * KC_TILE, x0 = (tile % tiles_x) * KC_TILE;
  const int co0 = blockIdx.y * COT, b0 = blockIdx.z * NI;
  const long HW = (long)H * W;
  float acc[NI][COT];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int o = 0; o < COT; ++o) acc[i][o] = 0.f;
  for (int c0 = 0; c0 < Cin; c0 += KC_CI) {
    __syncthreads();
    for (int e = t; e < NI * KC_CI * TW * TW; e += 256) {
      const int i = e / (KC_CI * TW * TW), r0 = e % (KC_CI * TW * TW);
      const int ci = r0 / (TW * TW), r = (r0 / TW) % TW, c = r0 % TW;
      const int gy = y0 + r - P, gx = x0 + c - P;
      float v = 0.f;
      if (b0 + i < B && c0 + ci < Cin && gy >= 0 && gy < H && gx >= 0 && gx < W) v = x[((long)(b0 + i) * Cin + c0 + ci) * HW + (long)gy * W + gx];
      xt[i][ci][r][c] = v;
    }
    __syncthreads();
    for (int ci = 0; ci < KC_CI; ++ci) {
      const float* wrow = wt + (long)(c0 + ci) * K * K * CoP + co0;       // uniform: scalar loads
#pragma unroll
      for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const float* wp = wrow + (ky * K + kx) * CoP;
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            const float v = xt[i][ci][ty + ky][tx + kx];
#pragma unroll
            for (int o = 0; o < COT; ++o) acc[i][o] = fmaf(v, wp[o], acc[i][o]);
          }
        }
    }
  }
  const int py = y0 + ty, px = x0 + tx;
  if (py < H && px < W) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
      if (b0 + i < B) {
#pragma unroll
        for (int o = 0; o < COT; ++o)
          if (co0 + o < Cout) y[((long)(b0 + i) * Cout + co0 + o) * HW + (long)py * W + px] = acc[i][o] + (bias ? bias[co0 + o] : 0.f);
      }
  }
}

// The same convolution with the input channels of a round split over KS groups of four waves (1024 threads at KS = 4): every use
// of a scalar-loaded weight waits for ALL outstanding scalar loads (SMEM returns out of order: lgkmcnt 0), ~250 cycles per
// three taps = 24 v_pk_fma, so a SIMD needs several waves to keep its VALU busy - and a small batch gives the one-image kernel
// one wave per SIMD (290 us per launch at batch 32, 364 at 256).  The groups' partial sums meet in LDS, added in group order.
template <int K, int COT, int KS>
__global__ __launch_bounds__(256 * KS) void convk_direct_sliced_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                                       const float* __restrict__ bias, float* __restrict__ y,
                                                                       int Cin, int Cout, int CoP, int H, int W, int tiles_x) {
  constexpr int P = (K - 1) / 2, TW = KC_TILE + K - 1, TWS = TW + 1, XT = KC_CI * TW * TWS;
  static_assert(KS * XT >= (KS - 1) * COT * 256, "the reduction reuses the staging buffer");
  __shared__ float lds[KS * XT];
  const int t = threadIdx.x, tx = t & 15, ty = (t >> 4) & 15;
  const int ks = __builtin_amdgcn_readfirstlane(t >> 8);         // uniform in a wave (four waves per group): keeps the weight loads scalar
  const int tile = blockIdx.x, y0 = (tile / tiles_x) * KC_TILE, x0 = (tile % tiles_x) * KC_TILE;
  const int co0 = blockIdx.y * COT, b = blockIdx.z;
  const long HW = (long)H * W;
  float acc[COT];
#pragma unroll
  for (int o = 0; o < COT; ++o) acc[o] = 0.f;
  const float* xb = x + (long)b * Cin * HW;
  float (*xt)[TW][TWS] = reinterpret_cast<float (*)[TW][TWS]>(lds + ks * XT);
  for (int c0 = 0; c0 < Cin; c0 += KC_CI * KS) {
    __syncthreads();
    for (int e = t; e < KS * KC_CI * TW * TW; e += 256 * KS) {
      const int cc = e / (TW * TW), r = (e / TW) % TW, c = e % TW;       // cc = slice * KC_CI + channel in the slice
      const int gy = y0 + r - P, gx = x0 + c - P;
      float v = 0.f;
      if (c0 + cc < Cin && gy >= 0 && gy < H && gx >= 0 && gx < W) v = xb[(long)(c0 + cc) * HW + (long)gy * W + gx];
      lds[(cc * TW + r) * TWS + c] = v;
    }
    __syncthreads();
    for (int ci = 0; ci < KC_CI; ++ci) {
      const float* wrow = wt + (long)(c0 + ks * KC_CI + ci) * K * K * CoP + co0;       // uniform per wave: scalar loads (rows past Cin: the image is padded to a multiple of KC_CI * 4)
#pragma unroll
      for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const float v = xt[ci][ty + ky][tx + kx];
          const float* wp = wrow + (ky * K + kx) * CoP;
#pragma unroll
          for (int o = 0; o < COT; ++o) acc[o] = fmaf(v, wp[o], acc[o]);
        }
    }
  }
  __syncthreads();
  const int pix = t & 255;
  if (ks > 0) {
#pragma unroll
    for (int o = 0; o < COT; ++o) lds[((ks - 1) * COT + o) * 256 + pix] = acc[o];
  }
  __syncthreads();
  const int py = y0 + ty, px = x0 + tx;
  if (ks == 0 && py < H && px < W) {
#pragma unroll
    for (int o = 0; o < COT; ++o) {
      float v = acc[o];
#pragma unroll
      for (int g = 1; g < KS; ++g) v += lds[((g - 1) * COT + o) * 256 + pix];
      if (co0 + o < Cout) y[((long)b * Cout + co0 + o) * HW + (long)py * W + px] = v + (bias ? bias[co0 + o] : 0.f);
    }
  }
}

constexpr int KW_CI = 16, KW_CO = 64, KW_ROWS = 8;     // weight-gradient workgroup: 16 input x 64 output channels, 8 x 16 pixel tiles

template <int K>
__global__ __launch_bounds__(256) void convk_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                          int B, int Cin, int Cout, int H, int W) {
  constexpr int P = (K - 1) / 2, XR = KW_ROWS + K - 1, XC = KC_TILE + K - 1, XCS = XC + 1, DS = KW_CO + 4;
  __shared__ float xt[KW_CI][XR][XCS];
  __shared__ __attribute__((aligned(16))) float dyt[KW_ROWS * KC_TILE][DS];
  const int t = threadIdx.x, og = t & 15, cl = t >> 4;
  const int c0 = blockIdx.x * KW_CI, o0 = blockIdx.y * KW_CO;
  const long HW = (long)H * W;
  float acc[K * K][4];
#pragma unroll
  for (int k = 0; k < K * K; ++k) { acc[k][0] = acc[k][1] = acc[k][2] = acc[k][3] = 0.f; }
  const int tiles_x = (W + KC_TILE - 1) / KC_TILE, tiles_y = (H + KW_ROWS - 1) / KW_ROWS;
  for (int b = blockIdx.z; b < B; b += gridDim.z)
    for (int tile = 0; tile < tiles_x * tiles_y; ++tile) {
      const int y0 = (tile / tiles_x) * KW_ROWS, x0 = (tile % tiles_x) * KC_TILE;
      __syncthreads();
      for (int e = t; e < KW_CI * XR * XC; e += 256) {
        const int ci = e / (XR * XC), r = (e / XC) % XR, c = e % XC;
        const int gy = y0 + r - P, gx = x0 + c - P;
        float v = 0.f;
        if (c0 + ci < Cin && gy >= 0 && gy < H && gx >= 0 && gx < W) v = x[((long)b * Cin + c0 + ci) * HW + (long)gy * W + gx];
        xt[ci][r][c] = v;
      }
      for (int e = t; e < KW_CO * KW_ROWS * KC_TILE; e += 256) {
        const int o = e / (KW_ROWS * KC_TILE), p = e % (KW_ROWS * KC_TILE);
        const int gy = y0 + p / KC_TILE, gx = x0 + p % KC_TILE;
        float v = 0.f;
        if (o0 + o < Cout && gy < H && gx < W) v = dy[((long)b * Cout + o0 + o) * HW + (long)gy * W + gx];
        dyt[p][o] = v;
      }
      __syncthreads();
      for (int r = 0; r < KW_ROWS; ++r)
        for (int cx = 0; cx < KC_TILE; cx += 4) {
          float xs[K][4 + K - 1];
#pragma unroll
          for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int j = 0; j < 4 + K - 1; ++j) xs[ky][j] = xt[cl][r + ky][cx + j];
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const float4 d = *reinterpret_cast<const float4*>(&dyt[r * KC_TILE + cx + p][og * 4]);
#pragma unroll
            for (int ky = 0; ky < K; ++ky)
#pragma unroll
              for (int kx = 0; kx < K; ++kx) {
                const float v = xs[ky][p + kx];
                acc[ky * K + kx][0] = fmaf(v, d.x, acc[ky * K + kx][0]);
                acc[ky * K + kx][1] = fmaf(v, d.y, acc[ky * K + kx][1]);
                acc[ky * K + kx][2] = fmaf(v, d.z, acc[ky * K + kx][2]);
                acc[ky * K + kx][3] = fmaf(v, d.w, acc[ky * K + kx][3]);
              }
          }
        }
    }
  const int ci = c0 + cl;
  if (ci < Cin) {
    float* pb = part + (long)blockIdx.z * Cout * Cin * K * K;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int o = o0 + og * 4 + j;
      if (o < Cout) {
        float* dst = pb + ((long)o * Cin + ci) * K * K;
#pragma unroll
        for (int k = 0; k < K * K; ++k) dst[k] = acc[k][j];
      }
    }
  }
}

__global__ void convk_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ gw, long n, int splits) {
  const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (e >= n) return;
  float s = 0.f;
  for (int k = 0; k < splits; ++k) s += part[(long)k * n + e];      // split order: deterministic
  gw[e] += s;
}

bool convk_supported(int K) { return K == 5; }       // the one window size models.lua instantiates besides 3
static int convk_splits(int B) { return B < 64 ? B : 64; }       // x 8 input-channel groups at Cin = 128: two workgroups per CU
size_t convk_workspace_bytes(int B, int Cin, int Cout, int K) {
  size_t img = convk_image_floats(Cin, Cout, K), img_b = convk_image_floats(Cout, Cin, K);
  size_t parts = (size_t)convk_splits(B) * Cin * Cout * K * K;
  size_t m = img > img_b ? img : img_b;
  return sizeof(float) * (m > parts ? m : parts) + 256;
}

template <int K>
static void convk_direct(const float* in, const float* w, const float* bias, float* out, void* ws, int B, int cin_eff, int cout_eff, int Cin, int Cout,
                         int H, int W, bool bwd, hipStream_t s) {
  float* wt = static_cast<float*>(ws);
  const int CiP = round_up(cin_eff, KC_CI * KC_KS), CoP = round_up(cout_eff, 32);
  {
    KtScope kt("convk_weight_image_kernel", 0, 8.0 * Cin * Cout * K * K, s);
    const long total = (long)CiP * K * K * CoP;
    convk_weight_image_kernel<<<(unsigned)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024), 256, 0, s>>>(w, wt, Cin, Cout, K, bwd ? 1 : 0, CiP, CoP);
  }
  const int tiles_x = (W + KC_TILE - 1) / KC_TILE, tiles_y = (H + KC_TILE - 1) / KC_TILE;
  KtScope kt(bwd ? "convk_direct_kernel(dgrad)" : "convk_direct_kernel", 2.0 * B * H * W * (double)Cin * Cout * K * K,
             4.0 * B * H * W * (Cin + Cout), s);
  constexpr int COT = 16;
  // two images per workgroup (each weight load used twice) measured SLOWER at the D network's shape, 430 vs 364 us (84 instead
  // of 50 VGPRs, twice the LDS): the scalar cache is not what bounds the one-image kernel.  Kept behind a switch.
  // measured (GAN batch, D network's shape): batch 32 - 290 -> 83 us per launch sliced; batch 256 (1024+ workgroups) - 364 us either way
  static const int sliced_env = GR_KNOB("GR_CONVK_SLICED", -1);
  const bool sliced = sliced_env >= 0 ? sliced_env != 0 : (long)tiles_x * tiles_y * (CoP / COT) * B < 1024;
  if (sliced) {
    dim3 grid(tiles_x * tiles_y, CoP / COT, B);
    convk_direct_sliced_kernel<K, COT, KC_KS><<<grid, 256 * KC_KS, 0, s>>>(in, wt, bias, out, cin_eff, cout_eff, CoP, H, W, tiles_x);
    return;
  }
  static const bool two_images = GR_KNOB_SET("GR_CONVK_NI2");
  if (two_images && (long)tiles_x * tiles_y * (CoP / COT) * ((B + 1) / 2) >= 512) {
    dim3 grid(tiles_x * tiles_y, CoP / COT, (B + 1) / 2);
    convk_direct_kernel<K, COT, 2><<<grid, 256, 0, s>>>(in, wt, bias, out, B, cin_eff, cout_eff, CoP, H, W, tiles_x);
  } else {
    dim3 grid(tiles_x * tiles_y, CoP / COT, B);
    convk_direct_kernel<K, COT, 1><<<grid, 256, 0, s>>>(in, wt, bias, out, B, cin_eff, cout_eff, CoP, H, W, tiles_x);
  }
}

// out[b,o,y,x] = bias[o] + sum w[o,i,ky,kx] in[b,i,y+ky-P,x+kx-P]        (ws: convk_workspace_bytes)
void launch_convk_forward(const float* in, const float* w, const float* bias, float* out, void* ws, int B, int Cin, int Cout, int H, int W, int K, hipStream_t s) {
  if (K == 5) convk_direct<5>(in, w, bias, out, ws, B, Cin, Cout, Cin, Cout, H, W, false, s);
}
// gin[b,i,y,x] = sum w[o,i,ky,kx] gout[b,o,y-ky+P,x-kx+P]
void launch_convk_backward_data(const float* gout, const float* w, float* gin, void* ws, int B, int Cin, int Cout, int H, int W, int K, hipStream_t s) {
  if (K == 5) convk_direct<5>(gout, w, nullptr, gin, ws, B, Cout, Cin, Cin, Cout, H, W, true, s);
}
// gw[o,i,ky,kx] += sum_{b,y,x} gout[b,o,y,x] in[b,i,y+ky-P,x+kx-P]
void launch_convk_backward_weight(const float* in, const float* gout, float* gw, void* ws, int B, int Cin, int Cout, int H, int W, int K, hipStream_t s) {
  float* part = static_cast<float*>(ws);
  const int splits = convk_splits(B);
  dim3 grid((Cin + KW_CI - 1) / KW_CI, (Cout + KW_CO - 1) / KW_CO, splits);
  {
    KtScope kt("convk_wgrad_kernel", 2.0 * B * H * W * (double)Cin * Cout * K * K, 4.0 * B * H * W * (Cin + Cout), s);
    if (K == 5) convk_wgrad_kernel<5><<<grid, 256, 0, s>>>(in, gout, part, B, Cin, Cout, H, W);
  }
  const long n = (long)Cin * Cout * K * K;
  KtScope kt("convk_wgrad_reduce_kernel", (double)n * splits, 4.0 * n * (splits + 2), s);
  convk_wgrad_reduce_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(part, gw, n, splits);
}

// ---------------------------------------------------------------- nn.PReLU() (one shared slope): gradWeight[0] += sum_{z <= 0} g * z
// (THNN PReLU.c accGradParameters, nOutputPlane == 0).  Products in fp32, sums in fp64: per-thread, per-workgroup, then the
// workgroups' partials in index order by the last kernel (deterministic).
constexpr int PRELU_BLOCKS = 2048;         // 8 workgroups per CU: the two streams are read at HBM speed (256 workgroups of scalar loads reached 2.3 TB/s)
__global__ __launch_bounds__(256) void prelu_grad_partial_kernel(const float* __restrict__ g, const float* __restrict__ z, long n, double* __restrict__ part) {
  __shared__ double sh[256];
  double s = 0;
  const long n4 = (((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(z)) & 15) == 0) ? n >> 2 : 0;    // 16-byte loads when both streams allow them
  const float4* g4 = reinterpret_cast<const float4*>(g);
  const float4* z4 = reinterpret_cast<const float4*>(z);
  const long stride = 256L * gridDim.x;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < n4; e += stride) {
    const float4 zv = z4[e], gv = g4[e];
    if (!(zv.x > 0.f)) s += (double)(gv.x * zv.x);
    if (!(zv.y > 0.f)) s += (double)(gv.y * zv.y);
    if (!(zv.z > 0.f)) s += (double)(gv.z * zv.z);
    if (!(zv.w > 0.f)) s += (double)(gv.w * zv.w);
  }
  for (long e = (n4 << 2) + blockIdx.x * 256L + threadIdx.x; e < n; e += stride) {      // tail / unaligned streams
    const float zv = z[e];
    if (!(zv > 0.f)) s += (double)(g[e] * zv);
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w]; __syncthreads(); }
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}
// A second launch rather than "the last workgroup to finish adds the partials": that needs one atomic per workgroup on ONE
// counter, and same-address atomics serialise at ~22 ns each - 45 us for 2048 workgroups (measured: 0.41 -> 0.98 ms per GAN batch).
__global__ __launch_bounds__(256) void prelu_grad_final_kernel(const double* __restrict__ part, int nparts, float* __restrict__ gslope) {
  __shared__ double sh[256];
  double s = 0;
  for (int k = threadIdx.x; k < nparts; k += 256) s += part[k];      // fixed assignment and tree: deterministic
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w]; __syncthreads(); }
  if (threadIdx.x == 0) gslope[0] += (float)sh[0];
}
size_t prelu_grad_workspace_bytes() { return sizeof(double) * PRELU_BLOCKS; }
void launch_prelu_grad(const float* g, const float* z, long n, double* part, float* gslope, hipStream_t s) {
  KtScope kt("prelu_grad_kernel", 2.0 * n, 8.0 * n, s);
  long blocks = (n / 4 + 255) / 256;            // one float4 per thread and round at most: small tensors get small grids
  if (blocks < 1) blocks = 1;
  if (blocks > PRELU_BLOCKS) blocks = PRELU_BLOCKS;
  prelu_grad_partial_kernel<<<(unsigned)blocks, 256, 0, s>>>(g, z, n, part);
  prelu_grad_final_kernel<<<1, 256, 0, s>>>(part, (int)blocks, gslope);
}

}  // namespace gr

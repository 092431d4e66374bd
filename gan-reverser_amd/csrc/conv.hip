// conv.hip — 3x3 stride-1 pad-1 cross-correlation on gfx950 as implicit GEMM on the fp32 MFMA
// (v_mfma_f32_32x32x2_f32: exact fp32 products/accumulation, 256 FLOP/clk/CU).
//
// Replaces nn.SpatialConvolution / cudnn.SpatialConvolution(…,3,3,1,1,1,1) as instantiated at
// reference models.lua:122,128,132 (G, with the preceding nn.SpatialUpSamplingNearest(2) folded
// into the input addressing) and models.lua:409,414,419,426,431,436 (R): forward, backward-data
// (same kernel on transposed+flipped weights) and backward-weight.
//
// GEMM view (forward):  M = Cout, N = B*H*W output pixels, K = 9*Cin.
//   * one workgroup (4 waves) owns a tile of 256 output pixels of one image x CT = 32*MT output channels;
//     wave w owns pixels [64w, 64w+64) as two 32-lane pixel groups x MT 32-channel row blocks.
//   * K is walked in chunks of 8 input channels (72 k-values); per chunk the zero-padded input patch
//     [8][rows+2][cols+2] and the k-major weight slice [72][CT] are staged global -> registers -> LDS,
//     double buffered, one barrier per chunk.
//   * k order inside a chunk is (tap, channel pair): the two halves of a wave (lanes 0-31 / 32-63) take
//     adjacent input channels at the same tap, so the B operand read is patch[ci+h][pix + tap offset]
//     (consecutive lanes -> consecutive LDS words) and the A operand read is wts[k+h][o] (same).
#include "kernels.h"
#include <string>
#include <cstdlib>

namespace gr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// blocks b and b+8 share an XCD (round-robin dispatch, observed): give each XCD a contiguous run of logical tiles
// so that the o-tiles of one pixel tile and neighbouring pixel tiles hit the same L2.  Bijective for any n.
__device__ __forceinline__ int xcd_remap(int bid, int n) {
  const int q = n >> 3, r = n & 7, xcd = bid & 7, slot = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

struct ConvArgs {
  const float* in; const float* wt; const float* bias; float* out;
  int B, Cin, Cout, H, W;
  int up, nchunks, cout_pad, tiles_x, tiles_y, n_otiles;
  int n_tiles = 0;      // wide kernel: logical tiles (a workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ...)
  ConvEpilogue ep;      // ep.mean != nullptr: evaluate()-mode BatchNorm + activation applied before the store
  const unsigned *amax_in = nullptr, *amax_w = nullptr;   // f16x3 mode: bit patterns of max|in| and max|weights| (device)
  unsigned* amax_out = nullptr;                           // nullable: slot that receives max|out|
  // nullable: per-channel (sum, sum of squares) of the stored output over this workgroup's pixels, [Cout][stat_tiles][2]
  // (training-mode BatchNorm statistics without a second pass over y; summed in a fixed order by bn_stats_finalize_tiles)
  double* stat_part = nullptr; int stat_tiles = 0;
  int nt_out = 0;        // non-temporal output stores (kernels.h store4; launchers set it from g_nt_stores)
  // evaluate() mode, f16x3 (kernels.h P16Out): the epilogue's result as the consuming convolution's operand-ready image, scaled by the
  // bound in p16_scale; `out` may then be null (no fp32 copy)
  uint4* p16_out = nullptr; const unsigned* p16_scale = nullptr;
};

// out = act(((conv + bias - mean) * invstd) * gamma + beta): the per-channel pipeline of an evaluate()-mode stage
// (G on this path: models.lua:122-124,128-130), same operation order and roundings as the stand-alone pipeline kernel.
// the same epilogue for a whole register block: the (workgroup-uniform) activation switch is taken once, not per element
template <int N>
__device__ __forceinline__ void conv_act_block(const ConvEpilogue& ep, float (&v)[N]) {
  switch (ep.act) {
    case ACT_ELU:        // the hardware exponential, as the pipeline kernels take it (elem.hip act_fwd: |error| < 3e-7 for z <= 0)
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] = v[i] <= 0.f ? (__expf(v[i]) - 1.f) : v[i];
      break;
    case ACT_RELU:
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] = v[i] > 0.f ? v[i] : 0.f;
      break;
    case ACT_LEAKYRELU:
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] = v[i] > 0.f ? v[i] : __fmul_rn(v[i], ep.slope);
      break;
    case ACT_SIGMOID:
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] = 1.f / (1.f + expf(-v[i]));
      break;
    case ACT_TANH:
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] = tanhf(v[i]);
      break;
    default: break;
  }
}
__device__ __forceinline__ float conv_epilogue(const ConvEpilogue& ep, float v, int o) {
  if (ep.mean) v = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(v, ep.mean[o]), ep.invstd[o]), ep.gamma[o]), ep.beta[o]);
  switch (ep.act) {
    case ACT_ELU: return v <= 0.f ? (__expf(v) - 1.f) : v;
    case ACT_RELU: return v > 0.f ? v : 0.f;
    case ACT_LEAKYRELU: return v > 0.f ? v : __fmul_rn(v, ep.slope);
    case ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    case ACT_TANH: return tanhf(v);
    default: return v;
  }
}

// NI > 1: the tile is NI whole images of IH = PT/(NI*TW) rows each (planes too small to fill a tile on their own, e.g. 16x16);
// their zero-padded patches are stacked in LDS, so tap offsets never cross from one image into the next.
template <int MT, int TW, int NG, int NI = 1>
__global__ __launch_bounds__(256, 2) void conv3x3_mfma_kernel(ConvArgs a) {
  constexpr int PT = 128 * NG, TR = PT / TW, IH = PT / (NI * TW), PR = NI * (IH + 2), PC = TW + 2, PS = PR * PC, CK = CONV_CK, CT = MT * 32;
  static_assert(NI == 1 || IH * NI * TW == PT, "tile must hold whole images");
  constexpr int NSLOT = (PS + 255) / 256;
  constexpr int WROWS = 9 * CK;
  constexpr int WV4 = WROWS * CT / 4;            // float4s of one weight chunk
  constexpr int NWF = WV4 / 256, WREM = WV4 % 256;   // full rounds of 256 float4 loads + a partial one
  static_assert((CK * PS * 4) % 16 == 0, "weight LDS region must stay 16-byte aligned");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* patch = smem;                 // [2][CK*PS]
  float* wts = smem + 2 * CK * PS;     // [2][WROWS*CT]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int ot = bid % a.n_otiles; bid /= a.n_otiles;
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y; const int b = (bid / a.tiles_y) * NI;
  const int y0 = ty * TR, x0 = tx * TW, o0 = ot * CT;
  const int H = a.H, W = a.W;
  const int Hs = a.up ? H >> 1 : H, Ws = a.up ? W >> 1 : W;
  const size_t HWs = (size_t)Hs * Ws;

  int src_off[NSLOT]; bool inb[NSLOT];
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) {
    const int e = tid + 256 * s, rr = e / PC, c = e - rr * PC;
    const int img = NI > 1 ? rr / (IH + 2) : 0, r = NI > 1 ? rr - img * (IH + 2) : rr;
    const int yy = y0 + r - 1, xx = x0 + c - 1;
    inb[s] = e < PS && yy >= 0 && yy < H && xx >= 0 && xx < W && b + img < a.B;
    src_off[s] = (a.up ? (yy >> 1) * Ws + (xx >> 1) : yy * Ws + xx) + img * a.Cin * (int)HWs;
  }
  const float* in_base = a.in + (size_t)b * a.Cin * HWs;

  // staging registers of the next chunk (kept as plain scalars/arrays indexed by constants so they stay in VGPRs)
  float pv[CK][NSLOT];
  float4 wv0 = make_float4(0.f, 0.f, 0.f, 0.f), wv1 = wv0, wv2 = wv0, wv3 = wv0, wv4 = wv0, wv5 = wv0, wv6 = wv0, wv7 = wv0, wv8 = wv0;
  static_assert(NWF + (WREM > 0) <= 9, "weight staging registers");
#define GR_WLOAD(i, reg)                                                                         \
  if ((i) < NWF || ((i) == NWF && WREM > 0 && tid < WREM)) {                                      \
    const int f_ = tid + 256 * (i), row_ = f_ / (CT / 4), c4_ = f_ - row_ * (CT / 4);            \
    reg = *reinterpret_cast<const float4*>(wp_ + (size_t)row_ * a.cout_pad + c4_ * 4);           \
  }
#define GR_WSTORE(i, reg)                                                                        \
  if ((i) < NWF || ((i) == NWF && WREM > 0 && tid < WREM)) *reinterpret_cast<float4*>(wd_ + (tid + 256 * (i)) * 4) = reg;
#define GR_LOAD_CHUNK(ch_)                                                                       \
  {                                                                                              \
    _Pragma("unroll") for (int cil = 0; cil < CK; ++cil) {                                       \
      const int ci = (ch_) * CK + cil;                                                           \
      const float* p_ = in_base + (size_t)ci * HWs;                                              \
      _Pragma("unroll") for (int s = 0; s < NSLOT; ++s) pv[cil][s] = (inb[s] && ci < a.Cin) ? p_[src_off[s]] : 0.f; \
    }                                                                                            \
    const float* wp_ = a.wt + (size_t)(ch_) * WROWS * a.cout_pad + o0;                           \
    GR_WLOAD(0, wv0) GR_WLOAD(1, wv1) GR_WLOAD(2, wv2) GR_WLOAD(3, wv3) GR_WLOAD(4, wv4)         \
    GR_WLOAD(5, wv5) GR_WLOAD(6, wv6) GR_WLOAD(7, wv7) GR_WLOAD(8, wv8)                          \
  }
#define GR_STORE_CHUNK(buf_)                                                                     \
  {                                                                                              \
    float* pd_ = patch + (buf_) * CK * PS;                                                       \
    _Pragma("unroll") for (int cil = 0; cil < CK; ++cil)                                         \
      _Pragma("unroll") for (int s = 0; s < NSLOT; ++s) {                                        \
        const int e_ = tid + 256 * s;                                                            \
        if (e_ < PS) pd_[cil * PS + e_] = pv[cil][s];                                            \
      }                                                                                          \
    float* wd_ = wts + (buf_) * WROWS * CT;                                                      \
    GR_WSTORE(0, wv0) GR_WSTORE(1, wv1) GR_WSTORE(2, wv2) GR_WSTORE(3, wv3) GR_WSTORE(4, wv4)    \
    GR_WSTORE(5, wv5) GR_WSTORE(6, wv6) GR_WSTORE(7, wv7) GR_WSTORE(8, wv8)                      \
  }

  f32x16 acc[MT][NG];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int ng = 0; ng < NG; ++ng)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][ng][r] = 0.f;

  int pixoff[NG];
#pragma unroll
  for (int ng = 0; ng < NG; ++ng) {
    const int p = (wave * NG + ng) * 32 + l31, prr = p / TW, pc = p - prr * TW;
    const int pr = NI > 1 ? prr + 2 * (prr / IH) : prr;      // skip the two padding rows between stacked images
    pixoff[ng] = pr * PC + pc + h * PS;
  }
  const int aoff = h * CT + l31;

  GR_LOAD_CHUNK(0)
  GR_STORE_CHUNK(0)
  __syncthreads();
  for (int ch = 0; ch < a.nchunks; ++ch) {
    if (ch + 1 < a.nchunks) GR_LOAD_CHUNK(ch + 1)
    const float* pb = patch + (ch & 1) * CK * PS;
    const float* wb = wts + (ch & 1) * WROWS * CT;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
      for (int cp = 0; cp < CK / 2; ++cp) {
        float av[MT], bv[NG];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) av[mt] = wb[(tap * CK + 2 * cp) * CT + aoff + mt * 32];
#pragma unroll
        for (int ng = 0; ng < NG; ++ng) bv[ng] = pb[(2 * cp) * PS + ky * PC + kx + pixoff[ng]];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int ng = 0; ng < NG; ++ng)
            acc[mt][ng] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt], bv[ng], acc[mt][ng], 0, 0, 0);
      }
    }
    if (ch + 1 < a.nchunks) GR_STORE_CHUNK((ch + 1) & 1)
    __syncthreads();
  }

  // epilogue: C/D map of the 32x32 MFMA: column (pixel) = lane&31, row (channel) = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int ng = 0; ng < NG; ++ng) {
    const int p = (wave * NG + ng) * 32 + l31, prr = p / TW, pc = p - prr * TW;
    const int img = NI > 1 ? prr / IH : 0, pr = NI > 1 ? prr - img * IH : prr;
    const int y = y0 + pr, x = x0 + pc;
    if (y < H && x < W && b + img < a.B) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = o0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (o < a.Cout) {
            const float bv = a.bias ? a.bias[o] : 0.f;
            a.out[(((size_t)(b + img) * a.Cout + o) * H + y) * W + x] = conv_epilogue(a.ep, acc[mt][ng][r] + bv, o);
          }
        }
    }
  }
}

// ---------------------------------------------------------------- few output channels (G's last conv, models.lua:132: 128 -> 1/3)
// M = Cout <= 4 would waste 7/8 of a 32-row MFMA block and the layer is HBM-bound anyway (reads 128 planes to emit 1-3), so
// this is a VALU kernel: one thread = 4 consecutive output pixels x all CO channels; input channels stream through LDS in
// chunks of 8 (register-prefetched float4 rows + scalar halo columns), weights come in through the scalar cache.
// KS = 2 / 4: the tile is 16 / 8 rows and the KS groups of the workgroup each take 1/KS of every chunk's channels for the
// same pixels, added through LDS (in group order) at the end - KS times the workgroups (a batch-256 layer on 32x32 planes
// has one 32-row tile per CU: 4 waves per CU, every chunk's load latency exposed; cfg2: 57 -> 47.5 us with two groups, 38.8
// with four, 3.5 TB/s).
template <int CO, int KS = 1, int TW = 32>
__global__ __launch_bounds__(256, (TW == 64 && CO <= 3) ? 4 : 3) void conv3x3_fewout_kernel(ConvArgs a, const float* __restrict__ w_native) {
  // TW = 64 (later in round 4): chunks of FOUR channels (a 21 KB patch) and four waves per SIMD instead of eight channels and three - the kernel is
  // bound by what its resident waves get through between two barriers, not by the barriers: 0.341 -> 0.31 ms at cfg3 (same box); five waves per
  // SIMD spill (0.70 ms), two-channel chunks at six even more (1.38 ms).
  // TW = 64 (round 4): a 64-wide plane as ONE tile column.  With 32-wide tiles every workgroup's two halo columns (4-byte loads at x0 - 1
  // and x0 + 32) pull the NEIGHBOUR tile's whole 128-byte line of every patch row out of the fabric - horizontally adjacent tiles run on
  // different XCDs (round-robin dispatch), so the neighbour's L2 copy does not help: G's last convolution at cfg3 fetched 1981 MB per launch
  // for 1074 MB of input (request-size counters, profiles/r04_traffic_two_ways_cfg3.txt) at 5.1 TB/s - it was traffic-bound, not LDS-bound.
  constexpr int SPR = TW / 4;                              // 4-pixel strips per tile row
  // Round 6 (counters: 39.5 % of this kernel's LDS cycles were bank conflicts at LDS-active 94.6 %, profiles/r05_pmc_step_conv_cfg2.txt).
  // ds_read_b128 serves a wave in the fixed lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+ 32), one LDS cycle each when their 16 lanes hit the 16
  // distinct 16-byte slots of the 256-byte bank row (MI355X_MICROARCH.md, LDS).  A patch row is SPR consecutive lanes; what decides is the slot offset
  // between the rows a group straddles:
  //   TW = 32 (8 strips per row: a group takes half-rows of FOUR rows): conflict-free iff those rows sit 0, 8, 0, 8 slots apart (mod 16) - row stride 12
  //            slots (48 floats; 40 before: 2-way conflicts) with a wave's eight lane octets on rows 0, 2, 4, 6, 1, 3, 5, 7 of its eight;
  //   TW = 64 (16 strips per row: a group takes parts of TWO rows): those must be 0 slots apart (mod 16) - the 18-slot stride stays, a wave's four lane
  //            sixteens take rows r, r + 8, r + 1, r + 9 (8 x 18 = 144 = 0 mod 16; adjacent rows are 2 apart: 2-way conflicts before).
  // The patch WRITES follow the same row order (row_perm below); weights are broadcast reads.  (Enumerated: no stride under 12 slots and no other row
  // order of eight rows at strides 10 ... 25 serves both lane groups conflict-free.)
  constexpr int TRr = 1024 / TW / KS, CK = (TW == 64 ? 4 : 8), PR = TRr + 2, PCS = (TW == 32 ? 48 : TW + 8), PS = PR * PCS;   // interior columns at [4, TW + 4), halo at 3 and TW + 4
  constexpr int NROWS = CK * PR;                           // patch rows of a chunk, contiguous at stride PCS
  auto row_perm = [](int t) {                              // which patch row the t-th run of SPR consecutive lanes takes
    if (TW == 32) return (t & ~7) + 2 * (t & 3) + ((t >> 2) & 1);
    if (TW == 64) return t < (NROWS & ~15) ? (t & ~15) + ((t & 15) >> 1) + 8 * (t & 1) : t;
    return t;
  };
  static_assert(TW != 32 || NROWS % 8 == 0, "whole groups of eight patch rows");
  constexpr int NV = (CK * PR * SPR + 255) / 256;          // float4 loads per thread per chunk (interior)
  constexpr int NHL = (CK * PR * 2 + 255) / 256;           // scalar loads per thread per chunk (halo columns)
  // the chunk's weights sit in LDS next to the patch ([ci][o][tap], rows padded to float4s) and are read back as broadcast
  // vectors: scalar-cache loads share the LDS counter and return out of order, so every use of one drained the whole
  // LDS queue (128 -> 1 at cfg2: 77 -> 57 us; 128 -> 3 at cfg3, VALU-bound: 531 -> 507 us)
  constexpr int WS = (CO * 9 + 3) & ~3;
  __shared__ __attribute__((aligned(16))) float patch[CK * PS];
  __shared__ __attribute__((aligned(16))) float wsh[CK * WS];
  // the two halo columns of every patch row in an array of their own, [row][left, right]: read as single dwords by the edge strips of a row (two lanes in
  // eight / sixteen); inside the patch rows (stride 48 / 72 floats) the rows a 32-lane group covers put them on ONE bank - a 4-way (TW = 32) / 2-way (TW = 64)
  // conflict per halo read, which is where the conflicts went when the float4 reads became conflict-free (counters of the first attempt:
  // profiles/r06_pmc_step_conv_cfg2_first.txt, 39.5 -> 36.1 %); here consecutive rows sit two banks apart
  __shared__ float halo[NROWS * 2];
  const int tid = threadIdx.x;
  int bid = blockIdx.x;
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y; const int b = bid / a.tiles_y;
  const int y0 = ty * TRr, x0 = tx * TW, H = a.H, W = a.W;
  const size_t HW = (size_t)H * W;
  const float* in_base = a.in + (size_t)b * a.Cin * HW;
  constexpr int TPG = 256 / KS;                            // threads per channel group: TRr rows x SPR strips
  const int kg = tid / TPG;                                // channel group of this thread (KS groups share the chunk's 8 channels)
  const int row = (TW == 32 || TW == 64) ? row_perm((tid % TPG) / SPR) : (tid % TPG) / SPR, strip = tid % SPR;    // this thread's 4 output pixels: (y0+row, x0+4*strip ..+3); rows in the conflict-free order (TRr is a multiple of 8 / 16: row_perm stays inside the tile)
  static_assert(CK * WS <= 512, "two weight words per thread");
  const int wl_c = tid / WS, wl_e = tid % WS, wl_c2 = (tid + 256) / WS, wl_e2 = (tid + 256) % WS;
  float wreg[2];
  // accumulators as two float2 per channel: the 9 x CO x 4 multiply-adds of an input channel run as chained PACKED FMAs
  // (v_pk_fma_f32: two per instruction) - the scalar spelling `acc += w0 v0 + w1 v1 + w2 v2` came out as mul + 2 fma + add, 144
  // VALU instructions per input channel at CO = 3 where 54 packed ones do (the layer is VALU-bound at CO = 3: round 3)
  typedef float fo_f2 __attribute__((ext_vector_type(2)));
  fo_f2 acc2[CO][2];
#pragma unroll
  for (int o = 0; o < CO; ++o) { acc2[o][0] = fo_f2{0.f, 0.f}; acc2[o][1] = fo_f2{0.f, 0.f}; }
  float4 xv[NV]; float hv[NHL];
#define GR_FO_LOAD(ch_)                                                                          \
  {                                                                                              \
    _Pragma("unroll") for (int i = 0; i < NV; ++i) {                                             \
      const int f = tid + 256 * i, q = f % SPR, t_ = row_perm(f / SPR), rr = t_ % PR, cil = t_ / PR; \
      const int ci = (ch_) * CK + cil, yy = y0 + rr - 1, xx = x0 + 4 * q;                        \
      xv[i] = (cil < CK && ci < a.Cin && yy >= 0 && yy < H && xx < W)                            \
                  ? *reinterpret_cast<const float4*>(in_base + (size_t)ci * HW + (size_t)yy * W + xx) \
                  : make_float4(0.f, 0.f, 0.f, 0.f);                                             \
    }                                                                                            \
    _Pragma("unroll") for (int i = 0; i < NHL; ++i) {                                            \
      const int e = tid + 256 * i, side = e & 1, rr = (e >> 1) % PR, cil = (e >> 1) / PR;        \
      const int ci = (ch_) * CK + cil, yy = y0 + rr - 1, xx = side ? x0 + TW : x0 - 1;           \
      hv[i] = (cil < CK && ci < a.Cin && yy >= 0 && yy < H && xx >= 0 && xx < W)                 \
                  ? in_base[(size_t)ci * HW + (size_t)yy * W + xx] : 0.f;                        \
    }                                                                                            \
    {                                                                                            \
      const int ci = (ch_) * CK + wl_c, ci2 = (ch_) * CK + wl_c2;                                \
      wreg[0] = (wl_c < CK && wl_e < CO * 9 && ci < a.Cin) ? w_native[((size_t)(wl_e / 9) * a.Cin + ci) * 9 + wl_e % 9] : 0.f;      \
      wreg[1] = (wl_c2 < CK && wl_e2 < CO * 9 && ci2 < a.Cin) ? w_native[((size_t)(wl_e2 / 9) * a.Cin + ci2) * 9 + wl_e2 % 9] : 0.f; \
    }                                                                                            \
  }
  const int nch = (a.Cin + CK - 1) / CK;
  GR_FO_LOAD(0)
  for (int ch = 0; ch < nch; ++ch) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + 256 * i, q = f % SPR, t_ = row_perm(f / SPR), rr = t_ % PR, cil = t_ / PR;
      if (cil < CK) *reinterpret_cast<float4*>(patch + cil * PS + rr * PCS + 4 + 4 * q) = xv[i];
    }
#pragma unroll
    for (int i = 0; i < NHL; ++i) {
      const int e = tid + 256 * i, side = e & 1, rr = (e >> 1) % PR, cil = (e >> 1) / PR;
      if (cil < CK) halo[(cil * PR + rr) * 2 + side] = hv[i];
    }
    if (wl_c < CK) wsh[wl_c * WS + wl_e] = wreg[0];
    if (wl_c2 < CK) wsh[wl_c2 * WS + wl_e2] = wreg[1];
    __syncthreads();
    if (ch + 1 < nch) GR_FO_LOAD(ch + 1)
#pragma unroll 2
    for (int cil = kg * (CK / KS); cil < (kg + 1) * (CK / KS); ++cil) {
      const int ci = ch * CK + cil;
      if (ci < a.Cin) {
        float wr[WS];
#pragma unroll
        for (int q = 0; q < WS / 4; ++q) {
          const float4 t = *reinterpret_cast<const float4*>(wsh + cil * WS + 4 * q);
          wr[4 * q] = t.x; wr[4 * q + 1] = t.y; wr[4 * q + 2] = t.z; wr[4 * q + 3] = t.w;
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const float* pr = patch + cil * PS + (row + ky) * PCS + 4 * strip;
          const float4 m = *reinterpret_cast<const float4*>(pr + 4);
          // the pixels left and right of this thread's four come from the neighbouring lanes' vectors (DPP row shifts: the 8 or 16
          // strips of a patch row are consecutive lanes of ONE 16-lane DPP row); only the two edge strips read the halo columns
          // from LDS, in one instruction.  (Both neighbours as single-dword LDS reads: 64 lanes on 16 banks, 59 % of this kernel's
          // LDS cycles were bank conflicts on the counters.)
          float lf = dpp_take<0x111, 0xF>(m.w), rt = dpp_take<0x101, 0xF>(m.x);      // row_shr:1 / row_shl:1
          if (strip == 0 || strip == SPR - 1) { const float hv = halo[(cil * PR + row + ky) * 2 + (strip == 0 ? 0 : 1)]; if (strip == 0) lf = hv; else rt = hv; }
          const fo_f2 p01 = {lf, m.x}, p12 = {m.x, m.y}, p23 = {m.y, m.z}, p34 = {m.z, m.w}, p45 = {m.w, rt};
#pragma unroll
          for (int o = 0; o < CO; ++o) {
            const float w0 = wr[o * 9 + ky * 3], w1 = wr[o * 9 + ky * 3 + 1], w2 = wr[o * 9 + ky * 3 + 2];
            const fo_f2 s0 = {w0, w0}, s1 = {w1, w1}, s2 = {w2, w2};
            acc2[o][0] = __builtin_elementwise_fma(s0, p01, acc2[o][0]);      // pixels 0, 1: taps kx = 0, 1, 2
            acc2[o][0] = __builtin_elementwise_fma(s1, p12, acc2[o][0]);
            acc2[o][0] = __builtin_elementwise_fma(s2, p23, acc2[o][0]);
            acc2[o][1] = __builtin_elementwise_fma(s0, p23, acc2[o][1]);      // pixels 2, 3
            acc2[o][1] = __builtin_elementwise_fma(s1, p34, acc2[o][1]);
            acc2[o][1] = __builtin_elementwise_fma(s2, p45, acc2[o][1]);
          }
        }
      }
    }
    __syncthreads();
  }
#undef GR_FO_LOAD
  float acc[CO][4];
#pragma unroll
  for (int o = 0; o < CO; ++o) { acc[o][0] = acc2[o][0].x; acc[o][1] = acc2[o][0].y; acc[o][2] = acc2[o][1].x; acc[o][3] = acc2[o][1].y; }
  if (KS > 1) {            // groups 1 .. KS-1 hand their partial sums over (the patch is dead: every thread is past the last chunk's barrier)
    static_assert((KS - 1) * CO * 4 * TPG <= CK * PS, "the hand-over reuses the patch");
    float* red = patch;
    if (kg > 0) {
#pragma unroll
      for (int o = 0; o < CO; ++o)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[((kg - 1) * CO * 4 + o * 4 + j) * TPG + (tid % TPG)] = acc[o][j];
    }
    __syncthreads();
    if (kg > 0) return;
#pragma unroll
    for (int g = 1; g < KS; ++g)          // added in group order: the same bits on every run
#pragma unroll
      for (int o = 0; o < CO; ++o)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[o][j] += red[((g - 1) * CO * 4 + o * 4 + j) * TPG + tid];
  }
  const int y = y0 + row, x = x0 + 4 * strip;
  if (y < H && x < W) {
#pragma unroll
    for (int o = 0; o < CO; ++o) {
      const float bv = a.bias ? a.bias[o] : 0.f;
      *reinterpret_cast<float4*>(a.out + (((size_t)b * a.Cout + o) * H + y) * W + x) =
          make_float4(conv_epilogue(a.ep, acc[o][0] + bv, o), conv_epilogue(a.ep, acc[o][1] + bv, o),
                      conv_epilogue(a.ep, acc[o][2] + bv, o), conv_epilogue(a.ep, acc[o][3] + bv, o));
    }
  }
}

// Few input channels (Cin <= 3: R's first convolution on gray / RGB images, models.lua:409): 9*Cin multiply-adds per output
// cannot feed a matrix pipe (a 16-channel MFMA chunk would be 13/16 or 15/16 zeros); the layer is bound by writing its 64
// output planes.  VALU kernel: a workgroup owns a 32x32 output tile of one image, a thread 4 consecutive pixels of a row;
// its 3 x 6 x Cin input window lives in registers, weights arrive through the scalar cache (uniform per workgroup), every
// output channel costs 36*Cin FMAs and one float4 store.  Accumulation order: (ci, ky, kx) ascending, fp32 FMA.
template <int CI, bool PO>
__device__ __forceinline__ void conv_fewin_body(const ConvArgs& a, const float* __restrict__ w_native) {
  const int tid = threadIdx.x;
  int bid = blockIdx.x;
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y; const int b = bid / a.tiles_y;
  const int H = a.H, W = a.W;
  const size_t HW = (size_t)H * W;
  const int y = ty * 32 + (tid >> 3), x = tx * 32 + 4 * (tid & 7);
  const bool pin = y < H && x < W;                            // W % 4 == 0: the four pixels are in or out together
  float v[CI][3][6];
#pragma unroll
  for (int ci = 0; ci < CI; ++ci)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = y + ky - 1;
      const bool rin = pin && yy >= 0 && yy < H;
      const float* rp = a.in + ((size_t)b * CI + ci) * HW + (size_t)(rin ? yy : 0) * W + (pin ? x : 0);
      const float4 m = rin ? *reinterpret_cast<const float4*>(rp) : make_float4(0.f, 0.f, 0.f, 0.f);
      v[ci][ky][0] = (rin && x > 0) ? rp[-1] : 0.f;
      v[ci][ky][1] = m.x; v[ci][ky][2] = m.y; v[ci][ky][3] = m.z; v[ci][ky][4] = m.w;
      v[ci][ky][5] = (rin && x + 4 < W) ? rp[4] : 0.f;
    }
  __shared__ float red[4][256][2];                            // per-wave channel sums for the BatchNorm statistics (Cout <= 256)
  const bool stats = a.stat_part != nullptr;
  float omax = 0.f;
  // blockIdx.y: slice of the output channels (small grids only: B * tiles workgroups of four waves leave the CUs at one wave
  // per SIMD, nothing to hide the scalar weight loads behind - cfg2: 256 workgroups, 50 us for a 20 us store stream)
  const int o_per = (a.Cout + (int)gridDim.y - 1) / (int)gridDim.y, o_beg = blockIdx.y * o_per, o_end = min(a.Cout, o_beg + o_per);
  if constexpr (PO) {
    // operand-ready output (evaluate() mode): eight channels at a time, a pixel's 8 results split into the two fp16 term vectors -
    // 4 consecutive pixels per thread = 64 contiguous bytes per term, the 8 threads of a row 512 (o_per is a multiple of 8: launcher)
    const float sc = pow2f(f16_scale_exp(absmax_read(a.p16_scale)));
    const int G = a.Cout >> 3;
    for (int o8 = o_beg; o8 < o_end; o8 += 8) {
      float r8[8][4];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int o = o8 + j;
        const float* wp = w_native + (size_t)o * CI * 9;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              const float wv = wp[(ci * 3 + ky) * 3 + kx];
#pragma unroll
              for (int q = 0; q < 4; ++q) acc[q] = fmaf(wv, v[ci][ky][q + kx], acc[q]);
            }
        const float bv = a.bias ? a.bias[o] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) r8[j][q] = conv_epilogue(a.ep, acc[q] + bv, o);
        if (pin) {
          const float4 r = make_float4(r8[j][0], r8[j][1], r8[j][2], r8[j][3]);
          if (a.out) *reinterpret_cast<float4*>(a.out + (((size_t)b * a.Cout + o) * H + y) * W + x) = r;
          omax = absmax4(omax, r);
        }
      }
      if (pin) {
        uint4* dst = a.p16_out + ((size_t)b * G + (o8 >> 3)) * 2 * HW + (size_t)y * W + x;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float x8[8] = {r8[0][q], r8[1][q], r8[2][q], r8[3][q], r8[4][q], r8[5][q], r8[6][q], r8[7][q]};
          uint4 t0, t1;
          split8_f16(x8, sc, t0, t1);
          dst[q] = t0; dst[HW + q] = t1;
        }
      }
    }
    if (a.amax_out) absmax_commit(omax, a.amax_out);
    return;
  }
  for (int o = o_beg; o < o_end; ++o) {
    const float* wp = w_native + (size_t)o * CI * 9;          // uniform address: scalar loads
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float wv = wp[(ci * 3 + ky) * 3 + kx];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = fmaf(wv, v[ci][ky][j + kx], acc[j]);
        }
    const float bv = a.bias ? a.bias[o] : 0.f;
    float4 r = make_float4(conv_epilogue(a.ep, acc[0] + bv, o), conv_epilogue(a.ep, acc[1] + bv, o),
                           conv_epilogue(a.ep, acc[2] + bv, o), conv_epilogue(a.ep, acc[3] + bv, o));
    if (pin) {
      *reinterpret_cast<float4*>(a.out + (((size_t)b * a.Cout + o) * H + y) * W + x) = r;
      omax = absmax4(omax, r);
    }
    if (stats) {                                              // training mode: r is the raw output (no epilogue)
      float sv = pin ? (r.x + r.y) + (r.z + r.w) : 0.f, qv = pin ? (r.x * r.x + r.y * r.y) + (r.z * r.z + r.w * r.w) : 0.f;
      sv = wave_sum(sv); qv = wave_sum(qv);
      if ((tid & 63) == 63) { red[tid >> 6][o][0] = sv; red[tid >> 6][o][1] = qv; }
    }
  }
  if (stats) {
    __syncthreads();
    for (int e = 2 * o_beg + tid; e < 2 * o_end; e += 256) {
      const int o = e >> 1, wh = e & 1;
      const double t = ((double)red[0][o][wh] + (double)red[1][o][wh]) + ((double)red[2][o][wh] + (double)red[3][o][wh]);
      a.stat_part[((size_t)o * a.stat_tiles + blockIdx.x) * 2 + wh] = t;
    }
  }
  if (a.amax_out) absmax_commit(omax, a.amax_out);
}
template <int CI>
__global__ __launch_bounds__(256) void conv3x3_fewin_kernel(ConvArgs a, const float* __restrict__ w_native) { conv_fewin_body<CI, false>(a, w_native); }
template <int CI>
__global__ __launch_bounds__(256, 4) void conv3x3_fewin_p16o_kernel(ConvArgs a, const float* __restrict__ w_native) { conv_fewin_body<CI, true>(a, w_native); }
bool conv_fewin_applies(int Cin, int W, bool up) { return Cin <= 3 && !up && W % 4 == 0 && W >= 4; }
bool conv_fewin_p16_out_supported(int Cout, int H, int W) { return Cout % 8 == 0 && Cout <= 256 && W % 4 == 0; }
void launch_conv3x3_fewin(const float* in, const float* w_native, const float* bias, float* out, int B, int Cin, int Cout, int H, int W,
                          hipStream_t s, const ConvEpilogue* ep, unsigned* amax_out, double* stat_part, int* stat_tiles, const P16Out* p16o) {
  ConvArgs a{};
  if (ep) a.ep = *ep;
  if (p16o && p16o->p16) { a.p16_out = reinterpret_cast<uint4*>(p16o->p16); a.p16_scale = p16o->scale; }
  a.in = in; a.bias = bias; a.out = out; a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.amax_out = amax_out;
  a.tiles_x = (W + 31) / 32; a.tiles_y = (H + 31) / 32;
  const int grid = B * a.tiles_x * a.tiles_y;
  if (stat_tiles) { const bool ok = stat_part && Cout <= 256; a.stat_part = ok ? stat_part : nullptr; a.stat_tiles = grid; *stat_tiles = ok ? grid : 0; }
  const double px = (double)B * H * W;
  const std::string name = std::string(a.p16_out ? "conv3x3_fewin_p16o_kernel<" : "conv3x3_fewin_kernel<") + std::to_string(Cin) + ">";
  KtScope kt(name.c_str(), 2.0 * px * Cout * Cin * 9.0, 4.0 * (px * Cin + px * Cout + 9.0 * Cin * Cout), s);
  int og = 1;                                                  // output-channel slices: aim at >= 1024 workgroups, >= 8 channels each
  while (grid * og < 1024 && Cout / (2 * og) >= 8 && (!a.p16_out || Cout % (16 * og) == 0)) og *= 2;      // operand-ready output: whole 8-channel groups per slice
  if (a.p16_out) {
    switch (Cin) {
      case 1: hipLaunchKernelGGL(conv3x3_fewin_p16o_kernel<1>, dim3(grid, og), dim3(256), 0, s, a, w_native); break;
      case 2: hipLaunchKernelGGL(conv3x3_fewin_p16o_kernel<2>, dim3(grid, og), dim3(256), 0, s, a, w_native); break;
      default: hipLaunchKernelGGL(conv3x3_fewin_p16o_kernel<3>, dim3(grid, og), dim3(256), 0, s, a, w_native); break;
    }
    return;
  }
  switch (Cin) {
    case 1: hipLaunchKernelGGL(conv3x3_fewin_kernel<1>, dim3(grid, og), dim3(256), 0, s, a, w_native); break;
    case 2: hipLaunchKernelGGL(conv3x3_fewin_kernel<2>, dim3(grid, og), dim3(256), 0, s, a, w_native); break;
    default: hipLaunchKernelGGL(conv3x3_fewin_kernel<3>, dim3(grid, og), dim3(256), 0, s, a, w_native); break;
  }
}

template <int MT, int TW, int NG, int NI = 1>
static void launch_conv_t(const ConvArgs& a0, hipStream_t s) {
  ConvArgs a = a0;
  constexpr int TR = 128 * NG / TW, IH = 128 * NG / (NI * TW), PS = NI * (IH + 2) * (TW + 2), CT = MT * 32;
  a.tiles_x = (a.W + TW - 1) / TW;
  a.tiles_y = NI > 1 ? 1 : (a.H + TR - 1) / TR;
  a.n_otiles = a.cout_pad / CT;
  const size_t lds = sizeof(float) * (2 * CONV_CK * PS + 2 * 9 * CONV_CK * CT);
  const int grid = ((a.B + NI - 1) / NI) * a.tiles_x * a.tiles_y * a.n_otiles;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_mfma_kernel<MT, TW, NG, NI>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  static const std::string name = "conv3x3_mfma_kernel<" + std::to_string(MT) + ", " + std::to_string(TW) + ", " + std::to_string(NG) + (NI > 1 ? ", " + std::to_string(NI) : std::string()) + ">";
  const double px = (double)a.B * a.H * a.W;
  KtScope kt(name.c_str(), 2.0 * px * a.Cout * a.Cin * 9.0,
             4.0 * (px * a.Cin / (a.up ? 4 : 1) + px * a.Cout + 9.0 * a.Cin * a.Cout), s);
  hipLaunchKernelGGL((conv3x3_mfma_kernel<MT, TW, NG, NI>), dim3(grid), dim3(256), lds, s, a);
}

int g_conv_variant = 0;   // tuning hook (GR_CONV_VARIANT env): 0 = default heuristics
template <int MT, int NG>
static void launch_conv_mt(const ConvArgs& a, hipStream_t s) {
  if (a.W <= 8) launch_conv_t<MT, 8, 2>(a, s);
  else if (a.W <= 16) launch_conv_t<MT, 16, NG>(a, s);
  else launch_conv_t<MT, 32, NG>(a, s);
}

void launch_conv3x3(const float* in, const float* wt, const float* bias, float* out,
                    int B, int Cin, int Cout, int H, int W, bool up, hipStream_t s, const float* w_native, const ConvEpilogue* ep) {
  const ConvWeightLayout L = conv_weight_layout(Cin, Cout);
  ConvArgs a{};
  if (ep) a.ep = *ep;
  a.in = in; a.wt = wt; a.bias = bias; a.out = out;
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.up = up ? 1 : 0;
  a.nchunks = L.cin_pad / CONV_CK; a.cout_pad = L.cout_pad;
  if (w_native && Cout <= 4 && !up && W % 4 == 0 && W >= 16) {
    static const int fo_split = GR_KNOB("GR_FEWOUT_SPLIT", 4);     // 0: 32-row tiles, 1: two channel groups on 16-row tiles, 4: four on 8-row tiles
    // (decided by the plane alone, not by the batch: a row must get the same bits whatever batch it travels in)
    const bool ks2 = fo_split && H % 16 == 0 && ((W + 31) / 32) * ((H + 31) / 32) < 4;      // small planes: too few 32-row tiles per image to fill the chip
    const bool ks4 = fo_split == 4 && ks2 && H % 8 == 0;
    // planes whose width is a multiple of 64: one 64-wide x 16-row tile per workgroup (no halo columns fetched from a neighbour's lines)
    static const bool fo_wide = !GR_KNOB_SET("GR_FEWOUT_NO_WIDE");
    const bool wide64 = fo_wide && !ks2 && W % 64 == 0;
    a.tiles_x = wide64 ? W / 64 : (W + 31) / 32; a.tiles_y = wide64 ? (H + 15) / 16 : (ks4 ? H / 8 : (ks2 ? H / 16 : (H + 31) / 32)); a.n_otiles = 1;
    const int grid = B * a.tiles_x * a.tiles_y;
    const double px = (double)B * H * W;
    const std::string fo_name = "conv3x3_fewout_kernel<" + std::to_string(Cout <= 3 ? Cout : 4) + ">";
    KtScope kt(fo_name.c_str(), 2.0 * px * Cout * Cin * 9.0, 4.0 * (px * Cin + px * Cout + 9.0 * Cin * Cout), s);
    if (ks4) switch (Cout) {
      case 1: hipLaunchKernelGGL((conv3x3_fewout_kernel<1, 4>), dim3(grid), dim3(256), 0, s, a, w_native); break;
      case 2: hipLaunchKernelGGL((conv3x3_fewout_kernel<2, 4>), dim3(grid), dim3(256), 0, s, a, w_native); break;
      case 3: hipLaunchKernelGGL((conv3x3_fewout_kernel<3, 4>), dim3(grid), dim3(256), 0, s, a, w_native); break;
      default: hipLaunchKernelGGL((conv3x3_fewout_kernel<4, 4>), dim3(grid), dim3(256), 0, s, a, w_native); break;
    } else if (ks2) switch (Cout) {
      case 1: hipLaunchKernelGGL((conv3x3_fewout_kernel<1, 2>), dim3(grid), dim3(256), 0, s, a, w_native); break;
      case 2: hipLaunchKernelGGL((conv3x3_fewout_kernel<2, 2>), dim3(grid), dim3(256), 0, s, a, w_native); break;
      case 3: hipLaunchKernelGGL((conv3x3_fewout_kernel<3, 2>), dim3(grid), dim3(256), 0, s, a, w_native); break;
      default: hipLaunchKernelGGL((conv3x3_fewout_kernel<4, 2>), dim3(grid), dim3(256), 0, s, a, w_native); break;
    } else if (wide64) switch (Cout) {
      case 1: hipLaunchKernelGGL((conv3x3_fewout_kernel<1, 1, 64>), dim3(grid), dim3(256), 0, s, a, w_native); break;
      case 2: hipLaunchKernelGGL((conv3x3_fewout_kernel<2, 1, 64>), dim3(grid), dim3(256), 0, s, a, w_native); break;
      case 3: hipLaunchKernelGGL((conv3x3_fewout_kernel<3, 1, 64>), dim3(grid), dim3(256), 0, s, a, w_native); break;
      default: hipLaunchKernelGGL((conv3x3_fewout_kernel<4, 1, 64>), dim3(grid), dim3(256), 0, s, a, w_native); break;
    } else
    switch (Cout) {
      case 1: hipLaunchKernelGGL(conv3x3_fewout_kernel<1>, dim3(grid), dim3(256), 0, s, a, w_native); break;
      case 2: hipLaunchKernelGGL(conv3x3_fewout_kernel<2>, dim3(grid), dim3(256), 0, s, a, w_native); break;
      case 3: hipLaunchKernelGGL(conv3x3_fewout_kernel<3>, dim3(grid), dim3(256), 0, s, a, w_native); break;
      default: hipLaunchKernelGGL(conv3x3_fewout_kernel<4>, dim3(grid), dim3(256), 0, s, a, w_native); break;
    }
    return;
  }
  static int variant = -1;
  if (variant < 0) { variant = GR_KNOB("GR_CONV_VARIANT", 0); }
  const bool big_img = (long)H * W >= 512;            // a 512-pixel tile needs at least that many pixels per image
  if (L.cout_pad % 64 != 0) { launch_conv_mt<1, 2>(a, s); return; }
  // measured on MI355X (B=256): 512-pixel tiles (NG=4) beat 256-pixel ones on 32x32 planes (G.convB 1333 -> 1198 us);
  // 128-channel row blocks with one workgroup per CU (MT=4) do not (G.convA 1212 -> 1269 us).
  if (variant == 1 && L.cout_pad % 128 == 0) launch_conv_mt<4, 2>(a, s);
  else if (variant == 2) launch_conv_mt<2, 2>(a, s);
  else if (big_img && W >= 32) launch_conv_mt<2, 4>(a, s);
  // two stacked 16x16 images per 512-pixel tile: correct, but register-staged prefetch spills at 128 accumulators
  // (R.conv5 172 -> 318 us); kept behind the tuning hook until the staging moves to LDS-DMA
  else if (variant == 4 && H == 16 && W == 16 && B >= 2) launch_conv_t<2, 16, 4, 2>(a, s);
  else launch_conv_mt<2, 2>(a, s);
}

// ---------------------------------------------------------------- fp32-accurate convolution on the bf16 MFMA ("bf16x6")
// Every fp32 operand is split into three bf16 terms x = x0 + x1 + x2 (|x - x0 - x1 - x2| <= 2^-27 |x|); the product keeps the
// six terms of order < 3 (x0w0, x0w1, x1w0, x0w2, x1w1, x2w0), accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The dropped
// terms are below 2^-24 relative, i.e. the result carries fp32-level error (measured against float64: the same 2e-6 as the
// fp32 MFMA path), while the matrix pipe runs 16x faster per instruction: 16/6 = 2.7x the fp32-MFMA ceiling.
// Tile: 256 pixels x 32 output channels per workgroup, K walked in chunks of 16 input channels: one MFMA = one tap x 16
// channels (lanes 0-31 take channels 0-7, lanes 32-63 channels 8-15).
//   LDS patch image  [3 terms][2 halves][pixel][8 ch] bf16  (16 B per lane, consecutive pixels -> conflict-free b128 reads)
//   LDS weight image [3 terms][9 taps][2 halves][32 o][8 ch] bf16, copied verbatim from the prepped global image.
typedef short bf16x8 __attribute__((ext_vector_type(8)));
constexpr int BF_CK = 16;

__device__ __forceinline__ unsigned short f32_to_bf16(float x) { const __bf16 h = (__bf16)x; return __builtin_bit_cast(unsigned short, h); }
__device__ __forceinline__ float bf16_to_f32(unsigned short u) { return __uint_as_float((unsigned)u << 16); }

// ---------------------------------------------------------------- fp32-accurate convolution on the f16 MFMA ("f16x3")
// Two fp16 terms carry 22 significand bits, so x = x0 + x1 with |x - x0 - x1| <= 2^-22 |x| and the three products
// x0w0, x0w1, x1w0 (the dropped x1w1 is below 2^-22) give fp32-level error with HALF the MFMAs of bf16x6 (measured against
// float64 on R fwd+bwd: outputs 1.8e-6, gradients 1.6e-6 relative; plain fp32: 1.7e-6 / 2.3e-6 -
// tools/split_precision_experiment.py).  fp16 has 5 exponent bits, so each tensor is first scaled by a power of two that
// puts its largest magnitude into [2^14, 2^15) (exact; elements more than 2^17 below the maximum lose relative but not
// absolute precision: their error stays below 2^-39 of the maximum); the result is scaled back with one v_ldexp_f32.
// The maxima are tracked on the device (producer kernels or absmax_kernel), never read by the host.
__device__ __forceinline__ void split8_bf16(const float* x, uint4& r0, uint4& r1, uint4& r2) {
  unsigned short t0[8], t1[8], t2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = x[j];
    t0[j] = f32_to_bf16(v); const float e1 = v - bf16_to_f32(t0[j]);
    t1[j] = f32_to_bf16(e1); const float e2 = e1 - bf16_to_f32(t1[j]);
    t2[j] = f32_to_bf16(e2);
  }
  r0 = make_uint4(t0[0] | (unsigned)t0[1] << 16, t0[2] | (unsigned)t0[3] << 16, t0[4] | (unsigned)t0[5] << 16, t0[6] | (unsigned)t0[7] << 16);
  r1 = make_uint4(t1[0] | (unsigned)t1[1] << 16, t1[2] | (unsigned)t1[3] << 16, t1[4] | (unsigned)t1[5] << 16, t1[6] | (unsigned)t1[7] << 16);
  r2 = make_uint4(t2[0] | (unsigned)t2[1] << 16, t2[2] | (unsigned)t2[3] << 16, t2[4] | (unsigned)t2[5] << 16, t2[6] | (unsigned)t2[7] << 16);
}
// one tap x 16 channels of the split product, smallest terms first
template <int NTERM>
__device__ __forceinline__ f32x16 split_mma(const uint4* av, const uint4* bv, f32x16 acc) {
  if (NTERM == 3) {
#define GR_M_(i, j) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[i]), __builtin_bit_cast(bf16x8, bv[j]), acc, 0, 0, 0);
    GR_M_(2, 0) GR_M_(1, 1) GR_M_(0, 2) GR_M_(1, 0) GR_M_(0, 1) GR_M_(0, 0)
#undef GR_M_
  } else {
#ifdef GR_PROBE_SHAPE16
    // TIMING-ONLY probe build (tools/build_probe.sh; results are wrong by design): every v_mfma_f32_32x32x16_f16 replaced by two
    // v_mfma_f32_16x16x32_f16 on the same operand registers and quarters of the same accumulator - same LDS reads, same multiply-adds,
    // same register footprint.  What the instruction SHAPE alone would buy these kernels, before any re-tiling is written.
    f32x4 q0 = {acc[0], acc[1], acc[2], acc[3]}, q1 = {acc[4], acc[5], acc[6], acc[7]}, q2 = {acc[8], acc[9], acc[10], acc[11]}, q3 = {acc[12], acc[13], acc[14], acc[15]};
#define GR_Q_(i, j, qa, qb) qa = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av[i]), __builtin_bit_cast(f16x8, bv[j]), qa, 0, 0, 0); \
                            qb = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av[i]), __builtin_bit_cast(f16x8, bv[j]), qb, 0, 0, 0);
    GR_Q_(1, 0, q0, q1) GR_Q_(0, 1, q2, q3) GR_Q_(0, 0, q0, q1)
#undef GR_Q_
    acc = f32x16{q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3], q2[0], q2[1], q2[2], q2[3], q3[0], q3[1], q3[2], q3[3]};
#else
#define GR_M_(i, j) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[i]), __builtin_bit_cast(f16x8, bv[j]), acc, 0, 0, 0);
    GR_M_(1, 0) GR_M_(0, 1) GR_M_(0, 0)
#undef GR_M_
#endif
  }
  return acc;
}

// NI (round 4) = whole images stacked in one tile (planes exactly TW wide and TR / NI high: the D network's 8x8 planes, NI = 4): a 256-pixel
// tile of ONE 8x8 image was three quarters padding (conv3x3_split_kernel<8, 1, 2> ran the deep tower of create_D2 at 83 TFLOP/s).  Each image
// keeps its own zero halo rows in the patch, as in the wide kernels.
// KS (round 4) = taps per side: 3, or 5 for the D network's nn.SpatialConvolution(128, 64, 5, 5, 1, 1, 2, 2) (models.lua:297) - the same
// kernel with a 2-pixel halo and 25 taps (conv5x5_split_kernel below; f16x3 only), in place of the fp32 VALU kernel of convk.hip
// (74 TFLOP/s at batch 256: 15 % of the GAN batch).
template <int TW, int MT, int NTERM, int NI, int KS>
__device__ __forceinline__ void conv_split_body(const ConvArgs& a, const uint4* __restrict__ wsplit) {
  constexpr int NT = 256 * MT;                                    // MT = 2: 8 waves = 2 channel blocks x 4 pixel quarters
  constexpr int HALO = KS / 2, KK = KS * KS;
  constexpr int NG = 2, PT = 256, TR = PT / TW, IH = TR / NI, PR = NI * (IH + 2 * HALO), PC = TW + 2 * HALO, PS = PR * PC, CT = 32 * MT;
  constexpr int NEH = 2 * PS, NSL = (NEH + NT - 1) / NT;         // (pixel, half) pairs staged per thread
  constexpr int WROWS = NTERM * KK * 2, WV = WROWS * CT, NWV = (WV + NT - 1) / NT;   // 16-byte weight vectors per chunk
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint4* patch = reinterpret_cast<uint4*>(smem_raw);              // [NTERM][2][PS]   (one uint4 = 8 bf16 / f16)
  uint4* wts = patch + NTERM * 2 * PS;                            // [NTERM][9][2][CT]
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, wmt = tid >> 8, l31 = lane & 31, h = lane >> 5;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int ot = bid % a.n_otiles; bid /= a.n_otiles;
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y; const int b = (bid / a.tiles_y) * NI;      // (first image of the tile)
  const int y0 = ty * TR, x0 = tx * TW, o0 = ot * CT;
  const int H = a.H, W = a.W;
  const int Hs = a.up ? H >> 1 : H, Ws = a.up ? W >> 1 : W;
  const size_t HWs = (size_t)Hs * Ws;
  int kin = 0, ktot = 0;
  if (NTERM == 2) { kin = f16_scale_exp(absmax_read(a.amax_in)); ktot = kin + f16_scale_exp(absmax_read(a.amax_w)); }
  const float sc_in = pow2f(kin);
  int src_off[NSL]; bool inb[NSL]; int sh[NSL];
#pragma unroll
  for (int s = 0; s < NSL; ++s) {
    const int eh = tid + NT * s, hh = eh >= PS ? 1 : 0, e = eh - hh * PS, r0_ = e / PC, c = e - r0_ * PC;
    const int img = NI > 1 ? r0_ / (IH + 2 * HALO) : 0, r = NI > 1 ? r0_ - img * (IH + 2 * HALO) : r0_;
    const int yy = y0 + r - HALO, xx = x0 + c - HALO;
    inb[s] = eh < NEH && yy >= 0 && yy < H && xx >= 0 && xx < W && b + img < a.B;
    src_off[s] = (a.up ? (yy >> 1) * Ws + (xx >> 1) : yy * Ws + xx) + img * a.Cin * (int)HWs;
    sh[s] = hh;
  }
  const float* in_base = a.in + (size_t)b * a.Cin * HWs;
  float pv[NSL][8];
  uint4 wv[NWV];
#define GR_BF_LOAD(ch_)                                                                                   \
  {                                                                                                       \
    _Pragma("unroll") for (int s = 0; s < NSL; ++s)                                                       \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                     \
        const int ci = (ch_) * BF_CK + 8 * sh[s] + j;                                                     \
        pv[s][j] = (inb[s] && ci < a.Cin) ? in_base[(size_t)ci * HWs + src_off[s]] : 0.f;                 \
      }                                                                                                   \
    const uint4* wp_ = wsplit + (size_t)(ch_) * WROWS * a.cout_pad + o0;                                  \
    _Pragma("unroll") for (int i = 0; i < NWV; ++i) {                                                     \
      const int f = tid + NT * i, row = f / CT, col = f - row * CT;                                       \
      wv[i] = f < WV ? wp_[(size_t)row * a.cout_pad + col] : make_uint4(0, 0, 0, 0);                      \
    }                                                                                                     \
  }
#define GR_BF_STORE()                                                                                     \
  {                                                                                                       \
    _Pragma("unroll") for (int s = 0; s < NSL; ++s) {                                                     \
      const int eh = tid + NT * s;                                                                        \
      if (eh < NEH) {                                                                                     \
        const int hh = eh >= PS ? 1 : 0, e = eh - hh * PS;                                                \
        uint4 t0, t1, t2;                                                                                 \
        if (NTERM == 3) { split8_bf16(pv[s], t0, t1, t2); patch[(2 * 2 + hh) * PS + e] = t2; }            \
        else split8_f16(pv[s], sc_in, t0, t1);                                                            \
        patch[(0 * 2 + hh) * PS + e] = t0; patch[(1 * 2 + hh) * PS + e] = t1;                             \
      }                                                                                                   \
    }                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < NWV; ++i) {                                                     \
      const int f = tid + NT * i;                                                                         \
      if (f < WV) wts[f] = wv[i];                                                                         \
    }                                                                                                     \
  }

  f32x16 acc[NG];
#pragma unroll
  for (int ng = 0; ng < NG; ++ng)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ng][r] = 0.f;
  int pix[NG];
#pragma unroll
  for (int ng = 0; ng < NG; ++ng) {
    const int p = (wave * NG + ng) * 32 + l31, pr = p / TW, pc = p - pr * TW;
    pix[ng] = h * PS + (NI > 1 ? pr + 2 * HALO * (pr / IH) : pr) * PC + pc;
  }
  const int nchunks = (a.Cin + BF_CK - 1) / BF_CK;
  GR_BF_LOAD(0)
  for (int ch = 0; ch < nchunks; ++ch) {
    GR_BF_STORE()
    __syncthreads();
    if (ch + 1 < nchunks) GR_BF_LOAD(ch + 1)
    // operand fetch for tap t+1 is issued before the MFMAs of tap t (two register sets, statically indexed)
    uint4 avA[NTERM], bvA[NG][NTERM], avB[NTERM], bvB[NG][NTERM];
#define GR_BF_OPS(tap_, av_, bv_)                                                                        \
    {                                                                                                     \
      const int toff_ = ((tap_) / KS) * PC + ((tap_) % KS);                                               \
      _Pragma("unroll") for (int s = 0; s < NTERM; ++s) {                                                 \
        av_[s] = wts[((s * KK + (tap_)) * 2 + h) * CT + wmt * 32 + l31];                                  \
        _Pragma("unroll") for (int ng = 0; ng < NG; ++ng) bv_[ng][s] = patch[s * 2 * PS + pix[ng] + toff_]; \
      }                                                                                                   \
    }
#define GR_BF_MMA(av_, bv_)                                                                               \
    _Pragma("unroll") for (int ng = 0; ng < NG; ++ng) acc[ng] = split_mma<NTERM>(av_, bv_[ng], acc[ng]);
#define GR_BF_PIN() __builtin_amdgcn_sched_group_barrier(0x100, 3 * NTERM, 0); __builtin_amdgcn_sched_group_barrier(0x008, NTERM == 3 ? 12 : 6, 0);
    if (KS == 3) {
    GR_BF_OPS(0, avA, bvA)
    GR_BF_OPS(1, avB, bvB) GR_BF_MMA(avA, bvA) GR_BF_PIN()
    GR_BF_OPS(2, avA, bvA) GR_BF_MMA(avB, bvB) GR_BF_PIN()
    GR_BF_OPS(3, avB, bvB) GR_BF_MMA(avA, bvA) GR_BF_PIN()
    GR_BF_OPS(4, avA, bvA) GR_BF_MMA(avB, bvB) GR_BF_PIN()
    GR_BF_OPS(5, avB, bvB) GR_BF_MMA(avA, bvA) GR_BF_PIN()
    GR_BF_OPS(6, avA, bvA) GR_BF_MMA(avB, bvB) GR_BF_PIN()
    GR_BF_OPS(7, avB, bvB) GR_BF_MMA(avA, bvA) GR_BF_PIN()
    GR_BF_OPS(8, avA, bvA) GR_BF_MMA(avB, bvB) GR_BF_PIN()
    GR_BF_MMA(avA, bvA)
    } else {
      // the same two-register-set pipeline, tap pairs walked by an unrolled loop (KK is odd: the last tap is in set A)
      GR_BF_OPS(0, avA, bvA)
#pragma unroll
      for (int t2 = 0; t2 < KK / 2; ++t2) {
        GR_BF_OPS(2 * t2 + 1, avB, bvB) GR_BF_MMA(avA, bvA) GR_BF_PIN()
        GR_BF_OPS(2 * t2 + 2, avA, bvA) GR_BF_MMA(avB, bvB) GR_BF_PIN()
      }
      GR_BF_MMA(avA, bvA)
    }
#undef GR_BF_OPS
#undef GR_BF_MMA
#undef GR_BF_PIN
    __syncthreads();
  }
#undef GR_BF_LOAD
#undef GR_BF_STORE
  float omax = 0.f;
#pragma unroll
  for (int ng = 0; ng < NG; ++ng) {
    const int p = (wave * NG + ng) * 32 + l31, pr0 = p / TW, pc = p - pr0 * TW;
    const int img = NI > 1 ? pr0 / IH : 0, pr = NI > 1 ? pr0 - img * IH : pr0;
    const int y = y0 + pr, x = x0 + pc;
    if (y < H && x < W && b + img < a.B) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = o0 + wmt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (o < a.Cout) {
          const float bvv = a.bias ? a.bias[o] : 0.f;
          const float v = NTERM == 2 ? ldexpf(acc[ng][r], -ktot) : acc[ng][r];
          const float res = conv_epilogue(a.ep, v + bvv, o);
          a.out[(((size_t)(b + img) * a.Cout + o) * H + y) * W + x] = res;
          omax = fmaxf(omax, fabsf(res));
        }
      }
    }
  }
  if (a.amax_out) absmax_commit(omax, a.amax_out);
}
template <int TW, int MT, int NTERM, int NI = 1>
__global__ __launch_bounds__(256 * MT, 2) void conv3x3_split_kernel(ConvArgs a, const uint4* __restrict__ wsplit) { conv_split_body<TW, MT, NTERM, NI, 3>(a, wsplit); }
template <int TW, int MT, int NTERM>
__global__ __launch_bounds__(256 * MT, 2) void conv5x5_split_kernel(ConvArgs a, const uint4* __restrict__ wsplit) { conv_split_body<TW, MT, NTERM, 1, 5>(a, wsplit); }

// Which pixel of the 512-pixel tile lane l of 32-lane group g = p / 32 owns (p = 32 g + l).  ds_read_b128 serves a wave in four
// fixed 16-lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...): it is conflict-free when lane l reads 16-byte slot
// (base + l) mod 16 of the 256-byte bank row.  A 32-wide tile row does that by itself.  A 16-wide tile puts lanes 16-31 on
// another patch row: rows 8 apart are 8 * 18 slots = 0 (mod 16) apart (adjacent rows are 18 = 2 apart: 2-way conflicts, a
// third of all LDS cycles on the counters).  An 8-wide tile (patch rows of 10 slots, 8 stacked images) needs row offsets
// 0, 8, 0, 8 (mod 16): rows q, q+4 of image i and of image i+4.
template <int TW>
__device__ __forceinline__ void tile_pixel(int p, int& prr, int& pc) {
  if (TW == 16) { const int g = p >> 5, l = p & 31; prr = (g >> 3) * 16 + (g & 7) + 8 * (l >> 4); pc = l & 15; }
  else if (TW == 8) { const int g = p >> 5, l = p & 31, j = l >> 3; prr = ((g >> 2) + 4 * (j >> 1)) * 8 + (g & 3) + 4 * (j & 1); pc = l & 7; }
  else { prr = p / TW; pc = p - prr * TW; }
}

// 512-pixel x 64-channel tile: all 8 waves keep BOTH 32-channel blocks (4 accumulators each) for their own 64 pixels, so one
// weight fetch and one patch conversion feed twice the MFMAs of the 256-pixel tile and every operand read feeds two MFMAs.
// NI > 1: the tile is NI whole images (16x16 planes: two of them), their zero-padded patches stacked in LDS.
template <int TW, int NI, int NTERM, bool DB>
__global__ __launch_bounds__(512, 2) void conv3x3_split_wide_kernel(ConvArgs a, const uint4* __restrict__ wsplit) {
  constexpr int NT = 512, MT = 2;
  constexpr int NG = 2, PT = 512, TR = PT / TW, IH = PT / (NI * TW), PR = NI * (IH + 2), PC = TW + 2, PS = PR * PC, CT = 32 * MT;
  // NI > 1 (whole images per tile): every halo slot is zero padding for every chunk and tile - zeroed once, and the staging
  // walks only the 512 real pixels x 2 channel halves (2 slots per thread instead of 3)
  constexpr bool COMPACT = NI > 1;
  constexpr int NEH = 2 * PS, NSL = COMPACT ? 2 : (NEH + NT - 1) / NT;         // (pixel, half) pairs staged per thread
  constexpr int WROWS = NTERM * 9 * 2, WV = WROWS * CT, NWV = (WV + NT - 1) / NT;   // 16-byte weight vectors per chunk
  static_assert((TW == 32 && NI == 1) || (TW == 16 && NI == 2), "tile_pixel assumes these tilings");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // DB: two LDS images; chunk c+1 is converted and stored in the middle of chunk c's MFMA phase (one barrier per chunk)
  constexpr int LBUF = NTERM * 2 * PS + WROWS * CT;               // uint4s of one (patch, weights) image
  uint4* patch = reinterpret_cast<uint4*>(smem_raw);              // [NTERM][2][PS]   (one uint4 = 8 bf16 / f16)
  uint4* wts = patch + NTERM * 2 * PS;                            // [NTERM][9][2][CT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int H = a.H, W = a.W;
  const int Hs = a.up ? H >> 1 : H, Ws = a.up ? W >> 1 : W;
  const size_t HWs = (size_t)Hs * Ws;
  // Persistent workgroups: with two operand images in LDS only one workgroup fits a CU, so the chip would run in lock-step
  // rounds - every CU loading, then multiplying, then storing at the same time, HBM idle two thirds of a round (PMC: waves
  // alive 34 us of which 12 us MFMA on R.conv2 at cfg2).  A workgroup walks tiles L, L + gridDim.x, ...: the first chunk of
  // the next tile is requested BEFORE the epilogue of the current one, so its stores drain behind the next tile's MFMAs.
  struct Geo { int y0, x0, o0, b, tile; };
  auto tile_geo = [&](int L) {
    int bid = xcd_remap(L, a.n_tiles);
    Geo g; g.tile = bid / a.n_otiles;
    const int ot = bid % a.n_otiles; bid /= a.n_otiles;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y; g.b = (bid / a.tiles_y) * NI;
    g.y0 = ty * TR; g.x0 = tx * TW; g.o0 = ot * CT;
    return g;
  };
  // Staging loads go through buffer descriptors: one 32-bit byte offset per staged (pixel, half) pair, the channel /
  // chunk part of the address in the scalar offset, and padding / out-of-image positions parked past the descriptor's
  // range (the hardware returns 0 for them) - no 64-bit pointers, no exec-masked branches around the loads.
  auto in_rsrc = [&](const Geo& g) {
    const float* in_base = a.in + (size_t)g.b * a.Cin * HWs;
    const size_t in_left = (size_t)(a.B - g.b) * a.Cin * HWs * sizeof(float);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in_base), 0, (int)(in_left < 0x7FFFF000ul ? in_left : 0x7FFFF000ul), 0x00020000);
  };
  const int nchunks = (a.Cin + BF_CK - 1) / BF_CK;
  const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wsplit), 0,
      (int)((size_t)nchunks * WROWS * a.cout_pad * 16), 0x00020000);
  int kin = 0, ktot = 0;
  if (NTERM == 2) { kin = f16_scale_exp(absmax_read(a.amax_in)); ktot = kin + f16_scale_exp(absmax_read(a.amax_w)); }
  const float sc_in = pow2f(kin);
  int clim[NSL], eoff[NSL];                                       // eoff: slot in the term-0 patch image (half * PS + position), -1 = none
#pragma unroll
  for (int s = 0; s < NSL; ++s) {
    if (COMPACT) {
      const int q = tid + NT * s, hh = q >> 9, p = q & 511, prr = p / TW, pc = p - prr * TW, img = prr / IH, r = prr - img * IH;
      clim[s] = a.Cin - 8 * hh;                                    // channel j of chunk ch is real iff ch*16 + j < clim
      eoff[s] = hh * PS + (img * (IH + 2) + r + 1) * PC + pc + 1;
    } else {
      const int eh = tid + NT * s;
      clim[s] = a.Cin - 8 * (eh >= PS ? 1 : 0);
      eoff[s] = eh < NEH ? eh : -1;
    }
  }
  auto stage_offsets = [&](const Geo& g, int (&voff_)[NSL]) {
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
      if (COMPACT) {
        const int q = tid + NT * s, hh = q >> 9, p = q & 511, prr = p / TW, pc = p - prr * TW, img = prr / IH, r = prr - img * IH;
        const int yy = g.y0 + r, xx = g.x0 + pc;
        const bool inb = yy < H && xx < W && g.b + img < a.B;
        const int so = (a.up ? (yy >> 1) * Ws + (xx >> 1) : yy * Ws + xx) + (img * a.Cin + 8 * hh) * (int)HWs;
        voff_[s] = inb ? so * 4 : (int)0x7FFFF000;
        continue;
      }
      const int eh = tid + NT * s, hh = eh >= PS ? 1 : 0, e = eh - hh * PS, rr = e / PC, c = e - rr * PC;
      const int img = NI > 1 ? rr / (IH + 2) : 0, r = NI > 1 ? rr - img * (IH + 2) : rr;
      const int yy = g.y0 + r - 1, xx = g.x0 + c - 1;
      const bool inb = eh < NEH && yy >= 0 && yy < H && xx >= 0 && xx < W && g.b + img < a.B;
      const int so = (a.up ? (yy >> 1) * Ws + (xx >> 1) : yy * Ws + xx) + (img * a.Cin + 8 * hh) * (int)HWs;
      voff_[s] = inb ? so * 4 : (int)0x7FFFF000;
    }
  };
  int L = blockIdx.x;
  Geo g = tile_geo(L);
  __amdgpu_buffer_rsrc_t rin = in_rsrc(g);
  int voff[NSL];
  stage_offsets(g, voff);
  int wvoff = ((tid >> 6) * a.cout_pad + g.o0 + (tid & 63)) * 16;   // weight vector f = tid + NT*i: row (tid>>6) + 8i
  float pv[NSL][8];
  uint4 wv[NWV];
  static_assert(CT == 64 && NT % CT == 0, "weight rows advance by NT / CT per staging slot");
#define GR_BF_LOAD(ch_, rin_, voff_, wvoff_)                                                              \
  {                                                                                                       \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                       \
      const int soff_ = (int)(((ch_) * BF_CK + j) * HWs * 4);                                             \
      _Pragma("unroll") for (int s = 0; s < NSL; ++s)                                                     \
        pv[s][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rin_, voff_[s], soff_, 0)); \
    }                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < NWV; ++i) {                                                     \
      const int soff_ = (((ch_) * WROWS + (NT / CT) * i) * a.cout_pad) * 16;                              \
      wv[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rwt, wvoff_, soff_, 0));    \
    }                                                                                                     \
  }
#define GR_BF_STORE(patch, wts, ch_)                                                                      \
  {                                                                                                       \
    if (((ch_) + 1) * BF_CK > a.Cin) {   /* last, partial chunk: channels past Cin read the next image - zero them */ \
      _Pragma("unroll") for (int s = 0; s < NSL; ++s)                                                     \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) if ((ch_) * BF_CK + j >= clim[s]) pv[s][j] = 0.f;   \
    }                                                                                                     \
    _Pragma("unroll") for (int s = 0; s < NSL; ++s) {                                                     \
      if (COMPACT || eoff[s] >= 0) {                                                                      \
        uint4 t0, t1, t2;                                                                                 \
        if (NTERM == 3) { split8_bf16(pv[s], t0, t1, t2); patch[4 * PS + eoff[s]] = t2; }                 \
        else split8_f16(pv[s], sc_in, t0, t1);                                                            \
        patch[eoff[s]] = t0; patch[2 * PS + eoff[s]] = t1;                                                \
      }                                                                                                   \
    }                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < NWV; ++i) {                                                     \
      const int f = tid + NT * i;                                                                         \
      if (f < WV) wts[f] = wv[i];                                                                         \
    }                                                                                                     \
  }

  f32x16 acc[MT][NG];
  int pix[NG];
#pragma unroll
  for (int ng = 0; ng < NG; ++ng) {
    const int p = (wave * NG + ng) * 32 + l31; int prr, pc; tile_pixel<TW>(p, prr, pc);
    const int pr = NI > 1 ? prr + 2 * (prr / IH) : prr;                   // skip the padding rows between stacked images
    pix[ng] = h * PS + pr * PC + pc;
  }
#define GR_BF_OPS(patch, wts, tap_, av_, bv_)                                                            \
    {                                                                                                     \
      const int toff_ = ((tap_) / 3) * PC + ((tap_) % 3);                                                 \
      _Pragma("unroll") for (int s = 0; s < NTERM; ++s) {                                                 \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) av_[mt][s] = wts[((s * 9 + (tap_)) * 2 + h) * CT + mt * 32 + l31]; \
        _Pragma("unroll") for (int ng = 0; ng < NG; ++ng) bv_[ng][s] = patch[s * 2 * PS + pix[ng] + toff_]; \
      }                                                                                                   \
    }
#define GR_BF_MMA(av_, bv_)                                                                               \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                     \
    _Pragma("unroll") for (int ng = 0; ng < NG; ++ng) acc[mt][ng] = split_mma<NTERM>(av_[mt], bv_[ng], acc[mt][ng]);
  // one tap's worth: (MT + NG) * NTERM LDS reads (for the next tap) first, then MT * NG * (NTERM == 3 ? 6 : 3) MFMAs
#define GR_BF_PIN() __builtin_amdgcn_sched_group_barrier(0x100, (MT + NG) * NTERM, 0); __builtin_amdgcn_sched_group_barrier(0x008, MT * NG * (NTERM == 3 ? 6 : 3), 0);
  float omax = 0.f;
  GR_BF_LOAD(0, rin, voff, wvoff)
  if (COMPACT) {                                                   // the padding slots, once (both images when double-buffered)
    for (int i = tid; i < (DB ? 2 : 1) * LBUF; i += NT) if (i % LBUF < NTERM * 2 * PS) patch[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
  }
  for (;;) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int ng = 0; ng < NG; ++ng)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][ng][r] = 0.f;
  if (DB) {
    GR_BF_STORE(patch, wts, 0)
    if (nchunks > 1) GR_BF_LOAD(1, rin, voff, wvoff)
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
      uint4* pc_ = patch + (ch & 1) * LBUF; uint4* wc_ = wts + (ch & 1) * LBUF;
      uint4* pn_ = patch + ((ch + 1) & 1) * LBUF; uint4* wn_ = wts + ((ch + 1) & 1) * LBUF;
      uint4 avA[MT][NTERM], bvA[NG][NTERM], avB[MT][NTERM], bvB[NG][NTERM];
      // operands of tap t+1 are fetched before the MFMAs of tap t (two register sets, pinned order)
      GR_BF_OPS(pc_, wc_, 0, avA, bvA)
      GR_BF_OPS(pc_, wc_, 1, avB, bvB) GR_BF_MMA(avA, bvA) GR_BF_PIN()
      GR_BF_OPS(pc_, wc_, 2, avA, bvA) GR_BF_MMA(avB, bvB) GR_BF_PIN()
      GR_BF_OPS(pc_, wc_, 3, avB, bvB) GR_BF_MMA(avA, bvA) GR_BF_PIN()
      GR_BF_OPS(pc_, wc_, 4, avA, bvA) GR_BF_MMA(avB, bvB) GR_BF_PIN()
      // the other image is free since the last barrier: convert + store chunk ch+1 while the matrix pipe drains the taps
      // above (the partner wave on this SIMD keeps it busy meanwhile), then fetch chunk ch+2 behind the remaining taps
      if (ch + 1 < nchunks) GR_BF_STORE(pn_, wn_, ch + 1)
      if (ch + 2 < nchunks) GR_BF_LOAD(ch + 2, rin, voff, wvoff)
      GR_BF_OPS(pc_, wc_, 5, avB, bvB) GR_BF_MMA(avA, bvA) GR_BF_PIN()
      GR_BF_OPS(pc_, wc_, 6, avA, bvA) GR_BF_MMA(avB, bvB) GR_BF_PIN()
      GR_BF_OPS(pc_, wc_, 7, avB, bvB) GR_BF_MMA(avA, bvA) GR_BF_PIN()
      GR_BF_OPS(pc_, wc_, 8, avA, bvA) GR_BF_MMA(avB, bvB) GR_BF_PIN()
      GR_BF_MMA(avA, bvA)
      __syncthreads();
    }
  } else {
    for (int ch = 0; ch < nchunks; ++ch) {
      GR_BF_STORE(patch, wts, ch)
      __syncthreads();
      if (ch + 1 < nchunks) GR_BF_LOAD(ch + 1, rin, voff, wvoff)
      if (NTERM == 3) {
        // one operand set: a second set (tap t+1 fetched behind tap t's MFMAs) measured no faster on bf16x6 - on random data
        // those kernels run at the clock the chip holds under MFMA load, not at an issue or latency limit
        uint4 avA[MT][NTERM], bvA[NG][NTERM];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) { GR_BF_OPS(patch, wts, tap, avA, bvA) GR_BF_MMA(avA, bvA) }
      } else {
        // f16x3 has twice the LDS reads per MFMA: with one set the compiler waits on lgkmcnt(0) three times per tap
        uint4 avA[MT][NTERM], bvA[NG][NTERM], avB[MT][NTERM], bvB[NG][NTERM];
        GR_BF_OPS(patch, wts, 0, avA, bvA)
        GR_BF_OPS(patch, wts, 1, avB, bvB) GR_BF_MMA(avA, bvA) GR_BF_PIN()
        GR_BF_OPS(patch, wts, 2, avA, bvA) GR_BF_MMA(avB, bvB) GR_BF_PIN()
        GR_BF_OPS(patch, wts, 3, avB, bvB) GR_BF_MMA(avA, bvA) GR_BF_PIN()
        GR_BF_OPS(patch, wts, 4, avA, bvA) GR_BF_MMA(avB, bvB) GR_BF_PIN()
        GR_BF_OPS(patch, wts, 5, avB, bvB) GR_BF_MMA(avA, bvA) GR_BF_PIN()
        GR_BF_OPS(patch, wts, 6, avA, bvA) GR_BF_MMA(avB, bvB) GR_BF_PIN()
        GR_BF_OPS(patch, wts, 7, avB, bvB) GR_BF_MMA(avA, bvA) GR_BF_PIN()
        GR_BF_OPS(patch, wts, 8, avA, bvA) GR_BF_MMA(avB, bvB) GR_BF_PIN()
        GR_BF_MMA(avA, bvA)
      }
      __syncthreads();
    }
  }
  // first chunk of the next tile: requested now, so that the stores below drain behind its arrival and the next tile's MFMAs
  const int Ln = L + (int)gridDim.x;
  const bool more = Ln < a.n_tiles;
  Geo gn = g; __amdgpu_buffer_rsrc_t rinn = rin; int voffn[NSL]; int wvoffn = wvoff;
  if (more) {
    gn = tile_geo(Ln); rinn = in_rsrc(gn); stage_offsets(gn, voffn);
    wvoffn = ((tid >> 6) * a.cout_pad + gn.o0 + (tid & 63)) * 16;
    GR_BF_LOAD(0, rinn, voffn, wvoffn)
  }
  const int y0 = g.y0, x0 = g.x0, o0 = g.o0, b = g.b;
  bool pin[NG]; size_t obase[NG];
#pragma unroll
  for (int ng = 0; ng < NG; ++ng) {
    const int p = (wave * NG + ng) * 32 + l31; int prr, pc; tile_pixel<TW>(p, prr, pc);
    const int img = NI > 1 ? prr / IH : 0, pr = NI > 1 ? prr - img * IH : prr;
    const int y = y0 + pr, x = x0 + pc;
    pin[ng] = y < H && x < W && b + img < a.B;
    obase[ng] = ((size_t)(b + img) * a.Cout * H + y) * W + x;
  }
  // scale back + bias in place
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = o0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const float bvv = (a.bias && o < a.Cout) ? a.bias[o] : 0.f;
#pragma unroll
      for (int ng = 0; ng < NG; ++ng) acc[mt][ng][r] = (NTERM == 2 ? ldexpf(acc[mt][ng][r], -ktot) : acc[mt][ng][r]) + bvv;
    }
  if (a.stat_part) {
    // BatchNorm batch statistics of what is about to be stored.  Per (wave, channel) the 64 pixels are two per lane over 32
    // lanes: the per-lane sums go through LDS transposed ([wave][quantity][channel][lane], row stride 33), one thread adds a
    // row in lane order, then the 8 waves are added in fp64 - everything in a fixed order.  (640 DPP adds per wave did the
    // same at four times the cost; the operand images in LDS are dead by now.)
    float* red = reinterpret_cast<float*>(smem_raw);             // [8 waves][2][32 channels][33]  = 67.6 KB per channel block
    float* rowsum = red + 8 * 2 * 32 * 33;                       // [512]
    const int tile = g.tile;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float sv = 0.f, qv = 0.f;
#pragma unroll
        for (int ng = 0; ng < NG; ++ng) { const float v = pin[ng] ? acc[mt][ng][r] : 0.f; sv += v; qv += v * v; }
        const int chl = (r & 3) + 8 * (r >> 2) + 4 * h;            // channel within this 32-channel block
        red[((wave * 2 + 0) * 32 + chl) * 33 + l31] = sv;
        red[((wave * 2 + 1) * 32 + chl) * 33 + l31] = qv;
      }
      __syncthreads();
      {
        const float* row = red + tid * 33;                         // row tid = (wave, quantity, channel)
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) t += row[i];
        rowsum[tid] = t;
      }
      __syncthreads();
      if (tid < 64) {                                              // (quantity, channel): the 8 waves in order
        const int wh = tid >> 5, chl = tid & 31;
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += (double)rowsum[(w * 2 + wh) * 32 + chl];
        const int o = o0 + mt * 32 + chl;
        if (o < a.Cout) a.stat_part[((size_t)o * a.stat_tiles + tile) * 2 + wh] = t;
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int ng = 0; ng < NG; ++ng) {
    if (pin[ng]) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = o0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (o < a.Cout) {
          const float res = conv_epilogue(a.ep, acc[mt][ng][r], o);
          a.out[obase[ng] + (size_t)o * H * W] = res;
          omax = fmaxf(omax, fabsf(res));
        }
      }
    }
  }
  if (!more) break;
  if (COMPACT && a.stat_part) {                                    // the statistics block used the first image as scratch: padding slots again
    for (int i = tid; i < NTERM * 2 * PS; i += NT) patch[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
  }
  L = Ln; g = gn; rin = rinn; wvoff = wvoffn;
#pragma unroll
  for (int s = 0; s < NSL; ++s) voff[s] = voffn[s];
  // (every wave is past the operand images: the chunk loop and the statistics block both end with a barrier)
  }
#undef GR_BF_OPS
#undef GR_BF_MMA
#undef GR_BF_PIN
#undef GR_BF_LOAD
#undef GR_BF_STORE
  if (a.amax_out) absmax_commit(omax, a.amax_out);
}

// ---------------------------------------------------------------- f16x3 convolution on OPERAND-READY activations ("P16")
// The kernel above re-loads fp32 activations one dword per lane and re-splits them to fp16 hi/lo on the VALU for every
// 16-channel chunk in every workgroup (PMC, R.conv2 at cfg2: 4.5 VALU instructions per MFMA, matrix pipe busy 30 %).  Here
// the PRODUCER of the activation (post_forward_g8_kernel / post_backward_b_g8_kernel, elem.hip) has already written the
// operand image:   p16[b][g = channel / 8][t = term][pixel] = 16 bytes = the 8 fp16 halves of term t (hi, lo) of channels
// 8g .. 8g+7 at that pixel, scaled by the power of two that the tensor's scale slot defines - the same 4 bytes per value as
// fp32.  A chunk's patch image [term][half][position] is then a pure gather of 16-byte vectors: it goes HBM -> LDS by
// LDS-DMA (buffer_load_dwordx4 ... lds: 64 consecutive LDS slots per wave-instruction, per-lane source address; positions
// outside the image are parked past the descriptor's range and arrive as zeros = the padding), the weight slab likewise.
// No staging registers, no VALU work, no ds_write: per chunk a wave issues ~10 DMA instructions next to its 108 MFMAs.
// Two LDS images; the DMA of the next chunk IN THE STREAM (tiles are walked persistently, so that is the next tile's first
// chunk at a tile's end) is issued at the top of a chunk and waited for (vmcnt(0) + barrier) at its bottom.
constexpr int P16_PAD = 64;          // LDS regions are multiples of one wave-instruction's 64 vectors
// Barrier that PUBLISHES LDS-DMA data: every wave first waits for its own DMA (s_waitcnt vmcnt(0), written out: hipcc's own
// wait in front of __syncthreads() came out as vmcnt(8) in conv3x3_p16_quad_kernel - a wave could pass the barrier with DMA
// still in flight and another wave read the stale image: rare, run-to-run different results), then the workgroup barrier.
__device__ __forceinline__ void dma_publish_barrier() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}
// one LDS-DMA wave-instruction: 64 lanes x 16 bytes land at lds_dst + 16 * lane, lane l fetching rsrc[voff_l + soff]
// (out-of-range lanes write zeros - verified on gfx950 by tools/probe/dma_probe.hip).  The builtin only exists in the device
// pass; hipcc drops the kernel's host stub if the host pass meets it, hence the guard.
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, uint4* lds_dst, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
#else
  (void)rsrc; (void)lds_dst; (void)voff; (void)soff;
#endif
}
// the same with the non-temporal cache policy (aux = 2): for streams that one workgroup reads once
__device__ __forceinline__ void lds_dma16_nt(__amdgpu_buffer_rsrc_t rsrc, uint4* lds_dst, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 2);
#else
  (void)rsrc; (void)lds_dst; (void)voff; (void)soff;
#endif
}
template <int TW, int NI>
__global__ __launch_bounds__(512, 2) void conv3x3_p16_wide_kernel(ConvArgs a, const uint4* __restrict__ wsplit, const uint4* __restrict__ xin) {
  constexpr int MT = 2, NTERM = 2;
  constexpr int NG = 2, PT = 512, TR = PT / TW, IH = PT / (NI * TW), PR = NI * (IH + 2), PC = TW + 2, PS = PR * PC, CT = 32 * MT;
  constexpr int PV = NTERM * 2 * PS, PVP = (PV + P16_PAD - 1) / P16_PAD * P16_PAD;      // patch vectors (padded to whole instructions)
  constexpr int WROWS = NTERM * 9 * 2, WV = WROWS * CT;                                 // weight vectors: one instruction per row
  constexpr int LBUF = PVP + WV;
  constexpr int NPI = PVP / 64, NPS = (NPI + 7) / 8, NWS = (WROWS + 7) / 8;            // DMA instructions per chunk / per wave
  static_assert((TW == 32 && NI == 1) || (TW == 16 && NI == 2), "tile_pixel assumes these tilings");
  static_assert(CT == 64, "one weight row = one wave-instruction");
  static_assert(2 * LBUF * 16 <= 160 * 1024, "LDS");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint4* lds = reinterpret_cast<uint4*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int H = a.H, W = a.W, HW = H * W;
  const int G = a.Cin >> 3;                                        // 8-channel groups (Cin % 16 == 0 on this path)
  struct Geo { int y0, x0, o0, b, tile; };
  auto tile_geo = [&](int L) {
    int bid = xcd_remap(L, a.n_tiles);
    Geo g; g.tile = bid / a.n_otiles;
    const int ot = bid % a.n_otiles; bid /= a.n_otiles;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y; g.b = (bid / a.tiles_y) * NI;
    g.y0 = ty * TR; g.x0 = tx * TW; g.o0 = ot * CT;
    return g;
  };
  const int nchunks = a.Cin / BF_CK;
  // one descriptor for the whole activation tensor (< 2 GB), one for the weight image
  const size_t xbytes = (size_t)a.B * G * 2 * HW * 16;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(xin), 0, (int)(xbytes < 0x7FFFF000ul ? xbytes : 0x7FFFF000ul), 0x00020000);
  const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wsplit), 0,
      (int)((size_t)nchunks * WROWS * a.cout_pad * 16), 0x00020000);
  const int ktot = f16_scale_exp(absmax_read(a.amax_in)) + f16_scale_exp(absmax_read(a.amax_w));
  // byte offset (chunk 0) of the source vector of patch slot e = 64 * (wave + 8 j) + lane, for this tile; parked when the
  // slot is padding, outside the image, or past the batch
  auto stage_offsets = [&](const Geo& g, int (&voff_)[NPS]) {
#pragma unroll
    for (int j = 0; j < NPS; ++j) {
      const int e = 64 * (wave + 8 * j) + lane;
      const int q = e / PS, pos = e - q * PS, t = q >> 1, hh = q & 1;     // plane q = term * 2 + half
      const int rr = pos / PC, c = pos - rr * PC;
      const int img = NI > 1 ? rr / (IH + 2) : 0, r = NI > 1 ? rr - img * (IH + 2) : rr;
      const int yy = g.y0 + r - 1, xx = g.x0 + c - 1;
      const bool inb = e < PV && yy >= 0 && yy < H && xx >= 0 && xx < W && g.b + img < a.B;
      voff_[j] = inb ? ((((g.b + img) * G + hh) * 2 + t) * HW + (int)p16_pos((unsigned)(yy * W + xx))) * 16 : (int)0x7FFFF000;
    }
  };
  int woff[NWS];                                                     // weight row r = wave + 8 j of the chunk's slab
  auto weight_offsets = [&](const Geo& g) {
#pragma unroll
    for (int j = 0; j < NWS; ++j) { const int r = wave + 8 * j; woff[j] = r < WROWS ? (r * a.cout_pad + g.o0 + lane) * 16 : (int)0x7FFFF000; }
  };
  // LDS-DMA of chunk ch of the tile described by (voff_, woff) into image buf_
#define GR_P16_DMA(buf_, ch_, voff_)                                                                      \
  {                                                                                                       \
    uint4* img_ = lds + (buf_) * LBUF;                                                                    \
    const int psoff_ = (ch_) * HW * 64;                               /* 2 groups x 2 terms x HW x 16 B per chunk */ \
    _Pragma("unroll") for (int j = 0; j < NPS; ++j) {                                                     \
      const int i_ = wave + 8 * j;                                                                        \
      if (i_ < NPI) lds_dma16(rin, img_ + 64 * i_, voff_[j], psoff_);                                     \
    }                                                                                                     \
    const int wsoff_ = (ch_) * WROWS * a.cout_pad * 16;                                                   \
    _Pragma("unroll") for (int j = 0; j < NWS; ++j) {                                                     \
      const int r_ = wave + 8 * j;                                                                        \
      if (r_ < WROWS) lds_dma16(rwt, img_ + PVP + 64 * r_, woff[j], wsoff_);                              \
    }                                                                                                     \
  }
  f32x16 acc[MT][NG];
  int pix[NG];
#pragma unroll
  for (int ng = 0; ng < NG; ++ng) {
    const int p = (wave * NG + ng) * 32 + l31; int prr, pc; tile_pixel<TW>(p, prr, pc);
    const int pr = NI > 1 ? prr + 2 * (prr / IH) : prr;                   // skip the padding rows between stacked images
    pix[ng] = h * PS + pr * PC + pc;
  }
#define GR_BF_OPS(patch, wts, tap_, av_, bv_)                                                            \
    {                                                                                                     \
      const int toff_ = ((tap_) / 3) * PC + ((tap_) % 3);                                                 \
      _Pragma("unroll") for (int s = 0; s < NTERM; ++s) {                                                 \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) av_[mt][s] = wts[((s * 9 + (tap_)) * 2 + h) * CT + mt * 32 + l31]; \
        _Pragma("unroll") for (int ng = 0; ng < NG; ++ng) bv_[ng][s] = patch[s * 2 * PS + pix[ng] + toff_]; \
      }                                                                                                   \
    }
#define GR_BF_MMA(av_, bv_)                                                                               \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                     \
    _Pragma("unroll") for (int ng = 0; ng < NG; ++ng) acc[mt][ng] = split_mma<NTERM>(av_[mt], bv_[ng], acc[mt][ng]);
#define GR_BF_PIN() __builtin_amdgcn_sched_group_barrier(0x100, (MT + NG) * NTERM, 0); __builtin_amdgcn_sched_group_barrier(0x008, MT * NG * 3, 0);
  float omax = 0.f;
  const int dbg = GR_DBG(a.up);                                              // diagnostic bit mask (0 in production): see g_p16_debug
  int L = blockIdx.x;
  Geo g = tile_geo(L);
  int voff[NPS];
  stage_offsets(g, voff);
  weight_offsets(g);
  int cc = 0;                                                        // chunks consumed so far: image cc & 1 holds the current one
  GR_P16_DMA(0, 0, voff)
  dma_publish_barrier();
  for (;;) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int ng = 0; ng < NG; ++ng)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][ng][r] = 0.f;
    const int Ln = L + (int)gridDim.x;
    const bool more = Ln < a.n_tiles;
    Geo gn = g; int voffn[NPS];
#pragma unroll
    for (int j = 0; j < NPS; ++j) voffn[j] = voff[j];
    for (int ch = 0; ch < nchunks; ++ch, ++cc) {
      // the other image is free since the barrier that ended the previous chunk: fetch the next chunk of the stream into it
      if (ch + 1 < nchunks) { if (!(dbg & 4)) GR_P16_DMA((cc + 1) & 1, ch + 1, voff) }
      else if (more) {                                               // last chunk of this tile: the next tile's first chunk
        gn = tile_geo(Ln); stage_offsets(gn, voffn); weight_offsets(gn);
        if (!(dbg & 4)) GR_P16_DMA((cc + 1) & 1, 0, voffn)
      }
      const uint4* pc_ = lds + (cc & 1) * LBUF; const uint4* wc_ = pc_ + PVP;
      if (!(dbg & 8)) {
      uint4 avA[MT][NTERM], bvA[NG][NTERM], avB[MT][NTERM], bvB[NG][NTERM];
      GR_BF_OPS(pc_, wc_, 0, avA, bvA)
      GR_BF_OPS(pc_, wc_, 1, avB, bvB) GR_BF_MMA(avA, bvA) GR_BF_PIN()
      GR_BF_OPS(pc_, wc_, 2, avA, bvA) GR_BF_MMA(avB, bvB) GR_BF_PIN()
      GR_BF_OPS(pc_, wc_, 3, avB, bvB) GR_BF_MMA(avA, bvA) GR_BF_PIN()
      GR_BF_OPS(pc_, wc_, 4, avA, bvA) GR_BF_MMA(avB, bvB) GR_BF_PIN()
      GR_BF_OPS(pc_, wc_, 5, avB, bvB) GR_BF_MMA(avA, bvA) GR_BF_PIN()
      GR_BF_OPS(pc_, wc_, 6, avA, bvA) GR_BF_MMA(avB, bvB) GR_BF_PIN()
      GR_BF_OPS(pc_, wc_, 7, avB, bvB) GR_BF_MMA(avA, bvA) GR_BF_PIN()
      GR_BF_OPS(pc_, wc_, 8, avA, bvA) GR_BF_MMA(avB, bvB) GR_BF_PIN()
      GR_BF_MMA(avA, bvA)
      }
      dma_publish_barrier();                                         // the DMA issued above has landed; every wave is past image cc & 1
    }
    const int y0 = g.y0, x0 = g.x0, o0 = g.o0, b = g.b;
    bool pin[NG]; size_t obase[NG];
#pragma unroll
    for (int ng = 0; ng < NG; ++ng) {
      const int p = (wave * NG + ng) * 32 + l31; int prr, pc; tile_pixel<TW>(p, prr, pc);
      const int img = NI > 1 ? prr / IH : 0, pr = NI > 1 ? prr - img * IH : prr;
      const int y = y0 + pr, x = x0 + pc;
      pin[ng] = y < H && x < W && b + img < a.B;
      obase[ng] = ((size_t)(b + img) * a.Cout * H + y) * W + x;
    }
    // scale back + bias in place
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = o0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float bvv = (a.bias && o < a.Cout) ? a.bias[o] : 0.f;
#pragma unroll
        for (int ng = 0; ng < NG; ++ng) acc[mt][ng][r] = ldexpf(acc[mt][ng][r], -ktot) + bvv;
      }
    if (a.stat_part && !(dbg & 2)) {
      // BatchNorm batch statistics of what is about to be stored (as in conv3x3_split_wide_kernel).  Scratch = the image the
      // last chunk was read from (cc - 1): the other one already holds the next tile's first chunk.
      float* red = reinterpret_cast<float*>(lds + ((cc - 1) & 1) * LBUF);   // [8 waves][2][32 channels][33]
      float* rowsum = red + 8 * 2 * 32 * 33;                       // [512]
      static_assert((8 * 2 * 32 * 33 + 512) * 4 <= LBUF * 16, "statistics scratch fits one image");
      const int tile = g.tile;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float sv = 0.f, qv = 0.f;
#pragma unroll
          for (int ng = 0; ng < NG; ++ng) { const float v = pin[ng] ? acc[mt][ng][r] : 0.f; sv += v; qv += v * v; }
          const int chl = (r & 3) + 8 * (r >> 2) + 4 * h;            // channel within this 32-channel block
          red[((wave * 2 + 0) * 32 + chl) * 33 + l31] = sv;
          red[((wave * 2 + 1) * 32 + chl) * 33 + l31] = qv;
        }
        __syncthreads();
        {
          const float* row = red + tid * 33;                         // row tid = (wave, quantity, channel)
          float t = 0.f;
#pragma unroll
          for (int i = 0; i < 32; ++i) t += row[i];
          rowsum[tid] = t;
        }
        __syncthreads();
        if (tid < 64) {                                              // (quantity, channel): the 8 waves in order
          const int wh = tid >> 5, chl = tid & 31;
          double t = 0.0;
#pragma unroll
          for (int w = 0; w < 8; ++w) t += (double)rowsum[(w * 2 + wh) * 32 + chl];
          const int o = o0 + mt * 32 + chl;
          if (o < a.Cout) a.stat_part[((size_t)o * a.stat_tiles + tile) * 2 + wh] = t;
        }
        __syncthreads();
      }
    }
#pragma unroll
    for (int ng = 0; ng < NG; ++ng) {
      if (pin[ng] && !(dbg & 1)) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = o0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (o < a.Cout) {
            const float res = conv_epilogue(a.ep, acc[mt][ng][r], o);
            a.out[obase[ng] + (size_t)o * H * W] = res;
            omax = fmaxf(omax, fabsf(res));
          }
        }
      }
    }
    if (!more) break;
    L = Ln; g = gn;
#pragma unroll
    for (int j = 0; j < NPS; ++j) voff[j] = voffn[j];
  }
#undef GR_BF_OPS
#undef GR_BF_MMA
#undef GR_BF_PIN
#undef GR_P16_DMA
  if (a.amax_out) absmax_commit(omax, a.amax_out);
}

// ---------------------------------------------------------------- the same, as TWO independent workgroups per CU
// Ablation of conv3x3_p16_wide_kernel on R.conv2 at cfg2 (tools/ablate_p16.py): skeleton 10.7 us + MFMA and LDS reads 35.5 +
// output stores 22 + DMA 9 = the 76 us it takes - its eight waves move through DMA issue, multiply and epilogue in lock-step and
// nothing overlaps.  Here a workgroup is FOUR waves (one per SIMD) that own the same 512-pixel x 64-channel tile, 128 pixels x
// 64 channels = 8 accumulator blocks per wave, with ONE operand image (76.8 KB): two workgroups are resident per CU and run
// out of phase by themselves, so while one waits for its DMA or stores its tile the other one has the matrix pipe.  An operand
// vector feeds more MFMAs than before (12 reads per 24 MFMAs per tap); staging and compute of ONE workgroup are serial
// (DMA -> barrier -> 216 MFMAs per wave -> barrier).
// NG = 32-pixel groups per wave: 4 (512-pixel tiles) or 2 (256-pixel tiles = ONE 16x16 image: a batch-256 layer on 16x16 planes
// has only 128 two-image tiles per 64 output channels - one workgroup or none per CU, so nothing hides a workgroup's DMA
// waits and epilogue; single-image tiles double the grid).
// MT = 32-channel output blocks per workgroup: 2, or 1 (with NG = 2: 256 pixels x 32 channels, a 40 KB image and ~110 registers -
// four workgroups per CU where the batch-256 16x16 layers would otherwise run two).
// PO: the result leaves as the NEXT convolution's operand-ready image (evaluate() mode; ConvArgs::p16_out) - its own kernel symbol
// (conv3x3_p16_quad_po_kernel) so that the training-path instantiations stay exactly what they were
template <int TW, int NI, int NG, int MT, bool PO>
__device__ __forceinline__ void conv_p16_quad_body(const ConvArgs& a, const uint4* __restrict__ wsplit, const uint4* __restrict__ xin) {
  constexpr int NTERM = 2, NW = 4;
  constexpr int PT = 128 * NG, TR = PT / TW, IH = PT / (NI * TW), PR = NI * (IH + 2), PC = TW + 2, PS = PR * PC, CT = 32 * MT;
  constexpr int PV = NTERM * 2 * PS, PVP = (PV + P16_PAD - 1) / P16_PAD * P16_PAD;
  constexpr int WROWS = NTERM * 9 * 2, WV = WROWS * CT;
  constexpr int LBUF = PVP + WV;
  constexpr int NPI = PVP / 64, NPS = (NPI + NW - 1) / NW, NWI = WV / 64, NWS = (NWI + NW - 1) / NW;   // DMA instructions: 64 vectors each
  static_assert((TW == 32 && NI == 1 && NG == 4) || (TW == 16 && NI == 2 && NG == 4) || (TW == 16 && NI == 1 && NG == 2) || (TW == 32 && NI == 1 && NG == 2), "tile_pixel assumes these tilings");
  static_assert(MT == 2 || NG == 2, "32-channel tiles only with 256-pixel tiles");
  static_assert((MT == 1 ? 4 : 2) * LBUF * 16 <= 160 * 1024, "two (MT = 2) or four (MT = 1) workgroups per CU");
  static_assert(WV % 64 == 0, "whole DMA instructions of weights");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint4* lds = reinterpret_cast<uint4*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int H = a.H, W = a.W, HW = H * W;
  const int G = a.Cin >> 3;
  const int dbg = GR_DBG(a.up);
  int bid = xcd_remap(blockIdx.x, a.n_tiles);
  const int tile = bid / a.n_otiles;
  const int ot = bid % a.n_otiles; bid /= a.n_otiles;
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y; const int b = (bid / a.tiles_y) * NI;
  const int y0 = ty * TR, x0 = tx * TW, o0 = ot * CT;
  const int nchunks = a.Cin / BF_CK;
  const size_t xbytes = (size_t)a.B * G * 2 * HW * 16;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(xin), 0, (int)(xbytes < 0x7FFFF000ul ? xbytes : 0x7FFFF000ul), 0x00020000);
  const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wsplit), 0,
      (int)((size_t)nchunks * WROWS * a.cout_pad * 16), 0x00020000);
  const int ktot = f16_scale_exp(absmax_read(a.amax_in)) + f16_scale_exp(absmax_read(a.amax_w));
  int voff[NPS], woff[NWS];
#pragma unroll
  for (int j = 0; j < NPS; ++j) {
    const int e = 64 * (wave + NW * j) + lane;
    const int q = e / PS, pos = e - q * PS, t = q >> 1, hh = q & 1;     // plane q = term * 2 + half
    const int rr = pos / PC, c = pos - rr * PC;
    const int img = NI > 1 ? rr / (IH + 2) : 0, r = NI > 1 ? rr - img * (IH + 2) : rr;
    const int yy = y0 + r - 1, xx = x0 + c - 1;
    const bool inb = e < PV && yy >= 0 && yy < H && xx >= 0 && xx < W && b + img < a.B;
    voff[j] = inb ? ((((b + img) * G + hh) * 2 + t) * HW + (int)p16_pos((unsigned)(yy * W + xx))) * 16 : (int)0x7FFFF000;
  }
#pragma unroll
  for (int j = 0; j < NWS; ++j) {          // weight vector f = 64 * instruction + lane of the [WROWS][CT] image (lane-linear in LDS)
    const int f = 64 * (wave + NW * j) + lane, r = f / CT, col = f - r * CT;
    woff[j] = f < WV ? (r * a.cout_pad + o0 + col) * 16 : (int)0x7FFFF000;
  }
#define GR_P16_DMA(ch_)                                                                                   \
  {                                                                                                       \
    const int psoff_ = (ch_) * HW * 64;                                                                   \
    _Pragma("unroll") for (int j = 0; j < NPS; ++j) {                                                     \
      const int i_ = wave + NW * j;                                                                       \
      if (i_ < NPI) { if (dbg & 16) lds_dma16_nt(rin, lds + 64 * i_, voff[j], psoff_); else lds_dma16(rin, lds + 64 * i_, voff[j], psoff_); } \
    }                                                                                                     \
    const int wsoff_ = (ch_) * WROWS * a.cout_pad * 16;                                                   \
    _Pragma("unroll") for (int j = 0; j < NWS; ++j) {                                                     \
      const int r_ = wave + NW * j;                                                                       \
      if (r_ < NWI) lds_dma16(rwt, lds + PVP + 64 * r_, woff[j], wsoff_);                                 \
    }                                                                                                     \
  }
  f32x16 acc[MT][NG];
  int pix[NG];
#pragma unroll
  for (int ng = 0; ng < NG; ++ng) {
    const int p = (wave * NG + ng) * 32 + l31; int prr, pc; tile_pixel<TW>(p, prr, pc);
    const int pr = NI > 1 ? prr + 2 * (prr / IH) : prr;
    pix[ng] = h * PS + pr * PC + pc;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int ng = 0; ng < NG; ++ng)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][ng][r] = 0.f;
  const uint4* patch = lds; const uint4* wts = lds + PVP;
  // diagnostic build path (dbg & 32, never in production): wave 0 stamps its phases into a.wt (reused as a debug buffer)
  unsigned long long* stamps = (dbg & 32) ? reinterpret_cast<unsigned long long*>(const_cast<float*>(a.wt)) + (size_t)blockIdx.x * 32 : nullptr;
  int nstamp = 0;
#define GR_STAMP() if (stamps && tid == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); stamps[nstamp++] = t_; }
  if (stamps && tid == 0) {
    stamps[nstamp++] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | ((32 - 1) << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | ((32 - 1) << 11)) << 32);
    unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); stamps[nstamp++] = t_;
  }
  GR_STAMP()
  // The two workgroups of a CU should alternate (one multiplies while the other stages / stores).  A grid that fits the chip
  // in one round starts them all together, in phase; the workgroups dispatched second (ids >= 256: observed placement, used
  // for speed only) therefore start half a compute phase late.
  // (which two workgroups share a CU is the dispatcher's business: the second one to arrive finds its waves in the odd wave
  // slots of the SIMDs - HW_ID.wave_id - whatever its block index is)
  if (a.nchunks > 0) {
    const unsigned hwid = __builtin_amdgcn_s_getreg(4 | (0 << 6) | ((4 - 1) << 11));     // HW_REG_HW_ID bits [3:0] = wave slot on its SIMD
    if (hwid & 1) {
#pragma unroll 1
      for (int i = 0; i < a.nchunks; ++i) __builtin_amdgcn_s_sleep(8);  // a.nchunks (unused otherwise on this path) = the delay in units of 8 x 64 clocks
    }
  }
  // PO: the evaluate()-mode BatchNorm coefficients of this workgroup's CT channels wait in LDS behind the operand image (the epilogue's 4 values per
  // channel as lane-dependent global loads - 64 per lane and 32-channel block, a round trip each - cost more than the multiplies of a 64-channel layer)
  float* bnp = reinterpret_cast<float*>(smem_raw + (size_t)LBUF * 16);
  if constexpr (PO) {
    if (a.ep.mean && tid < CT) {
      const int o = min(o0 + tid, a.Cout - 1);
      bnp[tid] = a.ep.mean[o]; bnp[CT + tid] = a.ep.invstd[o]; bnp[2 * CT + tid] = a.ep.gamma[o]; bnp[3 * CT + tid] = a.ep.beta[o];
    }
  }
  if (!(dbg & 4)) GR_P16_DMA(0)
  for (int ch = 0; ch < nchunks; ++ch) {
    dma_publish_barrier();                                           // the image holds chunk ch
    GR_STAMP()
    if (dbg & 256) __builtin_amdgcn_s_setprio(1);                     // experiment (GR_P16_DEBUG bit 256): the multiplying workgroup outranks its CU partner's epilogue / DMA issue
    if (!(dbg & 8)) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int toff = (tap / 3) * PC + (tap % 3);
        uint4 av[MT][NTERM], bv[NG][NTERM];
#pragma unroll
        for (int s = 0; s < NTERM; ++s) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) av[mt][s] = wts[((s * 9 + tap) * 2 + h) * CT + mt * 32 + l31];
#pragma unroll
          for (int ng = 0; ng < NG; ++ng) bv[ng][s] = patch[s * 2 * PS + pix[ng] + toff];
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int ng = 0; ng < NG; ++ng) acc[mt][ng] = split_mma<NTERM>(av[mt], bv[ng], acc[mt][ng]);
      }
    }
    if (dbg & 256) __builtin_amdgcn_s_setprio(0);
    __syncthreads();                                                 // every wave is past the image
    GR_STAMP()
    if (ch + 1 < nchunks && !(dbg & 4)) GR_P16_DMA(ch + 1)
  }
#undef GR_P16_DMA
  bool pin[NG]; size_t obase[NG];
#pragma unroll
  for (int ng = 0; ng < NG; ++ng) {
    const int p = (wave * NG + ng) * 32 + l31; int prr, pc; tile_pixel<TW>(p, prr, pc);
    const int img = NI > 1 ? prr / IH : 0, pr = NI > 1 ? prr - img * IH : prr;
    const int y = y0 + pr, x = x0 + pc;
    pin[ng] = y < H && x < W && b + img < a.B;
    obase[ng] = ((size_t)(b + img) * a.Cout * H + y) * W + x;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = o0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const float bvv = (a.bias && o < a.Cout) ? a.bias[o] : 0.f;
#pragma unroll
      for (int ng = 0; ng < NG; ++ng) acc[mt][ng][r] = ldexpf(acc[mt][ng][r], -ktot) + bvv;
    }
  if (a.stat_part && !(dbg & 2)) {
    // BatchNorm batch statistics of what is about to be stored: per (wave, channel) 128 pixels = four per lane over 32 lanes;
    // the per-lane sums go through LDS transposed ([wave][quantity][channel][lane], row stride 33), one thread adds a row in
    // lane order, then the 4 waves are added in fp64 - a fixed order.  The operand image is dead by now.
    float* red = reinterpret_cast<float*>(smem_raw);             // [4 waves][2][32 channels][33]
    float* rowsum = red + NW * 2 * 32 * 33;                      // [256]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float sv = 0.f, qv = 0.f;
#pragma unroll
        for (int ng = 0; ng < NG; ++ng) { const float v = pin[ng] ? acc[mt][ng][r] : 0.f; sv += v; qv += v * v; }
        const int chl = (r & 3) + 8 * (r >> 2) + 4 * h;
        red[((wave * 2 + 0) * 32 + chl) * 33 + l31] = sv;
        red[((wave * 2 + 1) * 32 + chl) * 33 + l31] = qv;
      }
      __syncthreads();
      {
        const float* row = red + tid * 33;                         // row tid = (wave, quantity, channel)
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) t += row[i];
        rowsum[tid] = t;
      }
      __syncthreads();
      if (tid < 64) {
        const int wh = tid >> 5, chl = tid & 31;
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += (double)rowsum[(w * 2 + wh) * 32 + chl];
        const int o = o0 + mt * 32 + chl;
        if (o < a.Cout) a.stat_part[((size_t)o * a.stat_tiles + tile) * 2 + wh] = t;
      }
      __syncthreads();
    }
  }
  // Output stores through an LDS transpose.  The accumulator layout (lane = pixel, register = channel) gives one dword per
  // lane per store: 256 store instructions per lane and tile, 2.2 TB/s (store-issue-bound: 30 of the kernel's 69 us on R.conv2
  // at cfg2).  Per 32-channel block each wave writes its 128 pixels x 32 channels to its own LDS region [channel][pixel]
  // (row stride 132 floats) and reads them back as 4 consecutive pixels of one channel per lane: 16-byte stores, a
  // wave-instruction covering all 128 pixels (512 contiguous bytes on a 32-wide plane) of two channels - 32 store
  // instructions per lane and tile.  The operand image is dead by now; the regions are per wave, so no barrier is needed
  // (a wave's LDS operations execute in order).
  float omax = 0.f;
  {
    constexpr int RS = 32 * NG + 4, NQ = 8 * NG, CPI = 64 / NQ;      // row stride, pixel quads per wave, channels per store instruction
    static_assert(NW * 32 * RS * 4 <= LBUF * 16, "staging fits the operand image");
    float* stg = reinterpret_cast<float*>(smem_raw) + wave * 32 * RS;
    // this lane's quad of pixels: k = lane % NQ -> pixels 4k .. 4k+3 of the wave's 32 * NG (consecutive in x)
    const int kq = lane % NQ, hq = lane / NQ;
    const int pq = (wave * NG + (kq >> 3)) * 32 + 4 * (kq & 7); int prrq, pcq; tile_pixel<TW>(pq, prrq, pcq);
    const int imgq = NI > 1 ? prrq / IH : 0, prq = NI > 1 ? prrq - imgq * IH : prrq;
    const int yq = y0 + prq, xq = x0 + pcq;
    const bool inq = yq < H && xq < W && b + imgq < a.B && !(dbg & 1);
    float* outq = a.out + ((size_t)(b + imgq) * a.Cout * H + yq) * W + xq;
    // (training-mode stages store the raw output: the per-element epilogue switch is taken once per workgroup, not 128 times)
    const bool plain = a.ep.mean == nullptr && a.ep.act == ACT_NONE;
    const bool want_max = a.amax_out != nullptr;
    float sc16 = 1.f;
    if constexpr (PO) sc16 = pow2f(f16_scale_exp(absmax_read(a.p16_scale)));
    if constexpr (PO) {      // (PO kernels write the operand-ready image only: the launcher passes no fp32 destination)
      // Operand-ready output ONLY (the evaluate()-mode chain: nobody reads the fp32 tensor) - straight from the accumulators, no LDS.  A lane
      // holds, of every 8-channel group g of a 32-channel block, the four channels 8 g + 4 h .. + 3 of ITS pixel; lane l + 32 holds the other
      // four of the same pixel.  Per group the epilogue's values are split and packed into two dwords per term; v_permlane32_swap then
      // trades halves between the lane pair - groups (p, p + 2): the low lane ends with all of group p, the high lane with all of group
      // p + 2 - and every lane stores whole 16-byte vectors: 32 consecutive pixels x 16 bytes per half-wave and (group, term) plane.
      // (First form of the round: the LDS staging image read back column by column - 128 ds_write + 128 ds_read per lane and block, 0.30 ms
      // of cfg5's 1.25 ms in these kernels went to the epilogue.)
      const int Gout = a.Cout >> 3;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {                            // channel groups pp and pp + 2 of the block: the pair that trades halves
          unsigned X[NG][2][4];                                     // [pixel group][group pp / pp + 2][hi.x, hi.y, lo.x, lo.y]
#pragma unroll
          for (int gi = 0; gi < 2; ++gi)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int r = 4 * (pp + 2 * gi) + j;
              const int chl = (r & 3) + 8 * (r >> 2) + 4 * h, cl = mt * 32 + chl;
              float v4[NG];
              if (a.ep.mean) {
                const float bm = bnp[cl], bi = bnp[CT + cl], bg = bnp[2 * CT + cl], bb = bnp[3 * CT + cl];
#pragma unroll
                for (int ng = 0; ng < NG; ++ng) v4[ng] = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(acc[mt][ng][r], bm), bi), bg), bb);
              } else {
#pragma unroll
                for (int ng = 0; ng < NG; ++ng) v4[ng] = acc[mt][ng][r];
              }
              conv_act_block<NG>(a.ep, v4);
#pragma unroll
              for (int ng = 0; ng < NG; ++ng) {
                if (want_max && pin[ng] && o0 + mt * 32 + chl < a.Cout) omax = fmaxf(omax, fabsf(v4[ng]));
                const float sv = v4[ng] * sc16;                     // split8_f16's roundings
                const _Float16 h0 = (_Float16)sv; const float rr = sv - (float)h0; const _Float16 h1 = (_Float16)rr;
                const unsigned u0 = __builtin_bit_cast(unsigned short, h0), u1 = __builtin_bit_cast(unsigned short, h1);
                if (j & 1) { X[ng][gi][j >> 1] |= u0 << 16; X[ng][gi][2 + (j >> 1)] |= u1 << 16; }
                else { X[ng][gi][j >> 1] = u0; X[ng][gi][2 + (j >> 1)] = u1; }
              }
            }
#pragma unroll
          for (int ng = 0; ng < NG; ++ng) {
#pragma unroll
            for (int d4 = 0; d4 < 4; ++d4) {
#if defined(__HIP_DEVICE_COMPILE__)
              const auto sw = __builtin_amdgcn_permlane32_swap(X[ng][0][d4], X[ng][1][d4], false, false);      // vdst lanes 32-63 <-> src lanes 0-31
              X[ng][0][d4] = sw[0]; X[ng][1][d4] = sw[1];
#endif
            }
            const int grp = (o0 + mt * 32) / 8 + (h ? pp + 2 : pp);
            if (pin[ng] && grp < Gout) {
              const int p = (wave * NG + ng) * 32 + l31; int prr, pc; tile_pixel<TW>(p, prr, pc);
              const int img = NI > 1 ? prr / IH : 0, pr = NI > 1 ? prr - img * IH : prr;
              uint4* dst = a.p16_out + ((size_t)(b + img) * Gout + grp) * 2 * HW + (size_t)(y0 + pr) * W + (x0 + pc);
              store4(dst, make_uint4(X[ng][0][0], X[ng][0][1], X[ng][1][0], X[ng][1][1]), false);
              store4(dst + HW, make_uint4(X[ng][0][2], X[ng][0][3], X[ng][1][2], X[ng][1][3]), false);
            }
          }
        }
      }
    }
    if constexpr (!PO) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (plain) {
#pragma unroll
        for (int ng = 0; ng < NG; ++ng)
#pragma unroll
          for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * h) * RS + ng * 32 + l31] = acc[mt][ng][r];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int chl = (r & 3) + 8 * (r >> 2) + 4 * h, o = min(o0 + mt * 32 + chl, a.Cout - 1);
          float v4[NG];
#pragma unroll
          for (int ng = 0; ng < NG; ++ng) v4[ng] = a.ep.mean ? __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(acc[mt][ng][r], a.ep.mean[o]), a.ep.invstd[o]), a.ep.gamma[o]), a.ep.beta[o]) : acc[mt][ng][r];
          conv_act_block<NG>(a.ep, v4);
#pragma unroll
          for (int ng = 0; ng < NG; ++ng) stg[chl * RS + ng * 32 + l31] = v4[ng];
        }
      }
      {
#pragma unroll
      for (int i = 0; i < 32 / CPI; ++i) {
        const int chl = CPI * i + hq, o = o0 + mt * 32 + chl;
        const float4 v = *reinterpret_cast<const float4*>(stg + chl * RS + 4 * kq);
        if (inq && o < a.Cout) {
          store4(outq + (size_t)o * H * W, v, a.nt_out != 0);
          if (want_max) omax = absmax4(omax, v);
        }
      }
      }
    }
    }
  }
  if (stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); GR_STAMP() if (tid == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); stamps[nstamp++] = t_; stamps[31] = nstamp; } }
#undef GR_STAMP
  if (a.amax_out) absmax_commit(omax, a.amax_out);
}
template <int TW, int NI, int NG = 4, int MT = 2>
__global__ __launch_bounds__(256, MT == 1 ? 4 : 2) void conv3x3_p16_quad_kernel(ConvArgs a, const uint4* __restrict__ wsplit, const uint4* __restrict__ xin) {
  conv_p16_quad_body<TW, NI, NG, MT, false>(a, wsplit, xin);
}
template <int TW, int NI, int NG = 4, int MT = 2>
__global__ __launch_bounds__(256, MT == 1 ? 4 : 2) void conv3x3_p16_quad_po_kernel(ConvArgs a, const uint4* __restrict__ wsplit, const uint4* __restrict__ xin) {
  conv_p16_quad_body<TW, NI, NG, MT, true>(a, wsplit, xin);
}

// ---------------------------------------------------------------- the same layer on v_mfma_f32_16x16x32_f16: 16x16 planes, 32-channel chunks
// VERDICT round 3, item 2.  The shape probe (tools/build_probe.sh) priced the instruction shape at -7...-8 % per launch for the kernel above; K = 32 needs
// 32-channel chunks, whose image (patch 43 KB + weights 37 KB for 256 pixels x 32 output channels) is 78 KB: for the ONE instantiation that already runs
// 256-pixel x 32-channel tiles - conv3x3_p16_quad_kernel<16, 1, 2, 1>, the dominant kernel at batch 256 - that still leaves two workgroups per CU (four
// before), the LDS reads per FLOP unchanged and half as many chunk hand-overs.  (64-channel tiles or 512-pixel tiles do not fit twice: §4.00.)
//   K index of the instruction = 8 q + e, q = lane / 16: q & 1 = which 8-channel group of a 16-channel sub-chunk, q >> 1 = which sub-chunk - so the operand
//   images are the kernel above's, two sub-chunks side by side: patch [term][4 groups][PS] (plane stride a multiple of 16 vectors: the four lane quarters of a
//   ds_read_b128 then start at the same 16-byte slot), weights [2 sub-chunks][term][9 taps][2 halves][32 o] = two consecutive chunk images of the prep kernel.
//   A operand (weights): lane -> output channel 16 mb + lane % 16; B operand (patch): lane -> pixel x = lane % 16 of row nb's tile row; accumulator block
//   [mb][nb] (16 o x 16 px): lane holds o = 16 mb + 4 q + i, i = 0..3, at pixel x.  A wave owns four tile rows (the kernel above's pixel order: tile_pixel)
//   and both 16-channel blocks: 8 blocks, 32 accumulator registers; per tap 4 + 8 operand reads for 24 instructions of 16 cycles (12 for 6 x 32 before).
// Epilogue: scale back + bias, BatchNorm statistics of the stored values (sums over a wave's 64 pixels by DPP inside the 16-lane rows, the four waves added in
// fp64 in wave order: a fixed order), then the kernel above's LDS transpose to 16-byte stores.  Training-mode output only (raw y; no fused epilogue, no PO).
template <int TW> constexpr int k32_ps() { return ((256 / TW + 2) * (TW + 2) + 15) / 16 * 16; }      // patch plane stride (vectors): 336 (16 x 16 tile) / 352 (8 rows x 32)
// TW = 32: tiles of 8 rows x 32 on planes whose width is a multiple of 32 - the image is then 80 KiB exactly (patch 44 KB + weights 36.9 KB): still two per CU
// Round 5: the kernel WALKS its units (pixel tile x 32-channel block) blockIdx.x, blockIdx.x + gridDim.x, ... - with one workgroup
// per unit (gridDim.x = n_tiles, what rounds 3-4 launched) every unit paid its own prologue: the first chunk's 80 KB requested, and waited for, with the
// workgroup's waves idle, while its CU partner may be in its own epilogue.  A workgroup that goes on to a next unit requests that unit's first PATCH (43 KB,
// the larger half) as soon as its last chunk is multiplied - before the epilogue, which stages through the WEIGHT half of the image (36 KB: it fits) - and the
// first weights right after the epilogue's last LDS read.  The epilogue's barriers wait for LDS operations only (s_waitcnt lgkmcnt(0); s_barrier), not for the
// request in flight.  Launcher: gridDim.x = min(n_tiles, 2 per CU), a multiple of 8, so that a workgroup's units stay on its XCD's run of logical tiles.
__device__ __forceinline__ void lds_only_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
template <int TW>
__global__ __launch_bounds__(256, 2) void conv3x3_p16_k32_kernel(ConvArgs a, const uint4* __restrict__ wsplit, const uint4* __restrict__ xin) {
  static_assert(TW == 16 || TW == 32, "one 16x16 image, or 8 rows of a 32-wide tile column");
  constexpr int NW = 4, TR = 256 / TW, PR = TR + 2, PC = TW + 2, PS0 = PR * PC, PS = k32_ps<TW>(), CT = 32, MB = 2, NB = 4;
  constexpr int PV = 2 * 4 * PS, PVP = (PV + 63) / 64 * 64, WR16 = 2 * 9 * 2, WV = 2 * WR16 * CT, LBUF = PVP + WV;
  constexpr int NPI = PVP / 64, NPS = (NPI + NW - 1) / NW, NWI = WV / 64, NWS = (NWI + NW - 1) / NW;
  static_assert(2 * LBUF * 16 <= 160 * 1024, "two workgroups per CU");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint4* lds = reinterpret_cast<uint4*>(smem_raw);
  unsigned char* wregion = smem_raw + (size_t)PVP * 16;           // the weight half of the image: the epilogue's staging area
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, q = lane >> 4;
  const int H = a.H, W = a.W, HW = H * W;
  const int G = a.Cin >> 3;
  const int nchunks = a.Cin / 32;
  const size_t xbytes = (size_t)a.B * G * 2 * HW * 16;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(xin), 0, (int)(xbytes < 0x7FFFF000ul ? xbytes : 0x7FFFF000ul), 0x00020000);
  const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wsplit), 0, (int)((size_t)(a.Cin / 16) * WR16 * a.cout_pad * 16), 0x00020000);
  const int ktot = f16_scale_exp(absmax_read(a.amax_in)) + f16_scale_exp(absmax_read(a.amax_w));
  int voff[NPS], woff[NWS];
  int tile, y0, x0, o0, b;
  // WHOLE (TW = 16: the launcher takes this kernel on 16 x 16 planes only): a unit's pixel tile is one whole image, so the per-lane DMA offsets are the same for
  // every unit - computed ONCE per workgroup for image 0 / output block 0; the unit's image and output block travel in the SCALAR offset of the DMA
  // instructions (round 6: the per-unit recomputation was ~600 VALU instructions - six patch slots and five weight slots with two integer divisions each -
  // between a unit's last multiply and the next unit's first request; the PMC pass counted 2.1 VALU per MFMA over the kernel, the chunk loop itself has 7 per
  // 216.  Same box, interleaved: 0.2681 -> 0.2650 ms for the six launches of a cfg2 step, profiles/r06_ab_k32_geometry_cfg2.txt)
  constexpr bool WHOLE = TW == 16;
  int psbase = 0, wsbase = 0;                                     // scalar byte offsets of the unit's image / output-channel block (WHOLE)
  auto lane_offsets = [&](int bb, int oo0) {
#pragma unroll
    for (int j = 0; j < NPS; ++j) {
      const int e = 64 * (wave + NW * j) + lane;
      const int p8 = e / PS, pos = e - p8 * PS, t = p8 >> 2, grp = p8 & 3;      // plane p8 = term * 4 + group
      const int rr = pos / PC, c = pos - rr * PC;
      const int yy = y0 + rr - 1, xx = x0 + c - 1;
      const bool inb = e < PV && pos < PS0 && yy >= 0 && yy < H && xx >= 0 && xx < W && bb < a.B;
      voff[j] = inb ? (((bb * G + grp) * 2 + t) * HW + yy * W + xx) * 16 : (int)0x7FFFF000;
    }
#pragma unroll
    for (int j = 0; j < NWS; ++j) {          // weight vector f of the [2 sub-chunks][36 rows][32 o] image
      const int f = 64 * (wave + NW * j) + lane, cc = f / (WR16 * CT), rem = f - cc * (WR16 * CT), r = rem / CT, col = rem - r * CT;
      woff[j] = f < WV ? ((cc * WR16 + r) * a.cout_pad + oo0 + col) * 16 : (int)0x7FFFF000;
    }
  };
  auto unit_geometry = [&](int u) {                               // unit u -> tile / channel block, and the per-lane DMA offsets of its operand image
    int bid = xcd_remap(u, a.n_tiles);
    tile = bid / a.n_otiles;
    const int ot = bid % a.n_otiles; bid /= a.n_otiles;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y; b = bid / a.tiles_y;
    y0 = ty * TR; x0 = tx * TW; o0 = ot * CT;
    if (WHOLE) { psbase = b * G * 2 * HW * 16; wsbase = o0 * 16; }     // (u < n_tiles = B x n_otiles: b < B)
    else lane_offsets(b, o0);
  };
  if (WHOLE) { y0 = 0; x0 = 0; lane_offsets(0, 0); }
#define GR_K32_DMA_PATCH(ch_)                                                                             \
  {                                                                                                       \
    const int psoff_ = psbase + (ch_) * HW * 128;                      /* 4 groups x 2 terms x HW vectors */ \
    _Pragma("unroll") for (int j = 0; j < NPS; ++j) {                                                     \
      const int i_ = wave + NW * j;                                                                       \
      if (i_ < NPI) lds_dma16(rin, lds + 64 * i_, voff[j], psoff_);                                       \
    }                                                                                                     \
  }
#define GR_K32_DMA_WEIGHTS(ch_)                                                                           \
  {                                                                                                       \
    const int wsoff_ = wsbase + (ch_) * 2 * WR16 * a.cout_pad * 16;                                       \
    _Pragma("unroll") for (int j = 0; j < NWS; ++j) {                                                     \
      const int r_ = wave + NW * j;                                                                       \
      if (r_ < NWI) lds_dma16(rwt, lds + PVP + 64 * r_, woff[j], wsoff_);                                 \
    }                                                                                                     \
  }
  int pix[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    int prr, pc; tile_pixel<TW>((wave * 2 + (nb >> 1)) * 32 + 16 * (nb & 1), prr, pc);      // the 16 pixels of block nb: one tile row from column pc (0 or 16)
    pix[nb] = q * PS + prr * PC + pc + l15;
  }
  const uint4* patch = lds; const uint4* wts = lds + PVP;
  const int wq = ((q >> 1) * WR16 + (q & 1)) * CT + l15;           // this lane's weight column: sub-chunk q >> 1, channel half q & 1, output channel l15 (+ 16 mb)
  int u = blockIdx.x;
  if (u >= a.n_tiles) return;
  unit_geometry(u);
  GR_K32_DMA_PATCH(0)
  GR_K32_DMA_WEIGHTS(0)
  for (;;) {
    f32x4 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ch = 0; ch < nchunks; ++ch) {
      dma_publish_barrier();
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int toff = (tap / 3) * PC + (tap % 3);
        uint4 av[MB][2], bv[NB][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) av[mb][t] = wts[wq + ((t * 9 + tap) * 2) * CT + mb * 16];
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) bv[nb][t] = patch[t * 4 * PS + pix[nb] + toff];
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            f32x4 c_ = acc[mb][nb];
            c_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av[mb][1]), __builtin_bit_cast(f16x8, bv[nb][0]), c_, 0, 0, 0);
            c_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av[mb][0]), __builtin_bit_cast(f16x8, bv[nb][1]), c_, 0, 0, 0);
            c_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av[mb][0]), __builtin_bit_cast(f16x8, bv[nb][0]), c_, 0, 0, 0);
            acc[mb][nb] = c_;
          }
      }
      __syncthreads();
      if (ch + 1 < nchunks) { GR_K32_DMA_PATCH(ch + 1) GR_K32_DMA_WEIGHTS(ch + 1) }
    }
    // this unit's geometry is needed by the epilogue; the next unit's patch request needs the next unit's offsets: keep the former in scalars
    const int e_tile = tile, e_y0 = y0, e_x0 = x0, e_o0 = o0, e_b = b;
    const int un = u + (int)gridDim.x;
    const bool more = un < a.n_tiles;
    if (more) { unit_geometry(un); GR_K32_DMA_PATCH(0) }          // lands in the patch half while the epilogue works in the weight half
    // (Round 6, measured and removed - git history, profiles/r06_ab_k32_register_transpose_cfg2.txt: an epilogue that transposes in registers - 4 x 4 quad_perm
    //  exchanges, 16 VALU per four values - needs no LDS staging, so the next unit's WEIGHTS can be requested here too and nothing is waited for at a unit start.
    //  A timing-only probe of that request order promised -11 % (r06_ab_k32_early_weights_bound_cfg2.txt) - on an operand image the staging had overwritten, i.e.
    //  on garbage: the real kernel measured +0.5 ... +0.8 % (0.2598 -> 0.2620 ms for the six launches).  The unit-start wait is hidden by the CU's other workgroup;
    //  what the probe showed was the clock the chip holds on trivial operands - MI355X_MICROARCH.md, DVFS give-back - not a schedule.)
    // scale back + bias: lane holds channel chl = 16 mb + 4 q + i of pixel l15 of block nb
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int o = e_o0 + mb * 16 + 4 * q + i;
        const float bvv = (a.bias && o < a.Cout) ? a.bias[o] : 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mb][nb][i] = ldexpf(acc[mb][nb][i], -ktot) + bvv;
      }
    const bool inimg = e_b < a.B;
    if (a.stat_part) {
      // per (wave, channel): the 64 pixels = four per lane over the 16 lanes of the channel's DPP row; then the four waves in fp64, in wave order
      float* wsum = reinterpret_cast<float*>(wregion);             // [4 waves][2][32 channels]  (the weight image is dead: every wave is past the last chunk's barrier)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float sv = 0.f, qv = 0.f;
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) { const float v = inimg ? acc[mb][nb][i] : 0.f; sv += v; qv += v * v; }
          // sum over the 16 lanes of the row (row_shr 1, 2, 4, 8 with zero fill: lane 15 of the row ends with the total, in a fixed order)
          sv += dpp_take<0x111, 0xF>(sv); sv += dpp_take<0x112, 0xF>(sv); sv += dpp_take<0x114, 0xF>(sv); sv += dpp_take<0x118, 0xF>(sv);
          qv += dpp_take<0x111, 0xF>(qv); qv += dpp_take<0x112, 0xF>(qv); qv += dpp_take<0x114, 0xF>(qv); qv += dpp_take<0x118, 0xF>(qv);
          if (l15 == 15) { const int chl = mb * 16 + 4 * q + i; wsum[(wave * 2 + 0) * 32 + chl] = sv; wsum[(wave * 2 + 1) * 32 + chl] = qv; }
        }
      lds_only_barrier();
      if (tid < 64) {
        const int wh = tid >> 5, chl = tid & 31;
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += (double)wsum[(w * 2 + wh) * 32 + chl];
        const int o = e_o0 + chl;
        if (o < a.Cout) a.stat_part[((size_t)o * a.stat_tiles + e_tile) * 2 + wh] = t;
      }
      lds_only_barrier();
      if (GR_DBG(a.wt != nullptr)) {      // ablation build, GR_K32_FENCE_PROBE=1: what a release + arrival per unit would cost (a BatchNorm finalisation by the last arriver needs it)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (tid == 0) atomicAdd(reinterpret_cast<unsigned*>(const_cast<float*>(a.wt)) + (e_o0 / CT) * 32, 1u);
      }
    }
    // output stores through the per-wave LDS transpose of the kernel above: [channel][pixel of the wave's 64] -> four consecutive pixels of a channel per lane
    float omax = 0.f;
    {
      constexpr int RS = 64 + 4, NQ = 16, CPI = 64 / NQ;
      static_assert(NW * 32 * RS * 4 <= WV * 16, "staging fits the weight half of the operand image");
      float* stg = reinterpret_cast<float*>(wregion) + wave * 32 * RS;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int i = 0; i < 4; ++i) stg[(mb * 16 + 4 * q + i) * RS + nb * 16 + l15] = acc[mb][nb][i];
      const int kq = lane % NQ, hq = lane / NQ;                     // pixel quad kq of the wave's 64 pixels (block kq / 4, x = 4 (kq % 4)), channel row hq + CPI i
      int prrq, pcq; tile_pixel<TW>(wave * 64 + 4 * kq, prrq, pcq);
      const int yq = e_y0 + prrq, xq = e_x0 + pcq;
      float* outq = a.out + ((size_t)e_b * a.Cout * H + yq) * W + xq;
      const bool want_max = a.amax_out != nullptr;
#pragma unroll
      for (int i2 = 0; i2 < 32 / CPI; ++i2) {
        const int chl = CPI * i2 + hq, o = e_o0 + chl;
        const float4 v = *reinterpret_cast<const float4*>(stg + chl * RS + 4 * kq);
        if (inimg && o < a.Cout) {
          store4(outq + (size_t)o * H * W, v, a.nt_out != 0);
          if (want_max) omax = absmax4(omax, v);
        }
      }
    }
    if (a.amax_out) absmax_commit(omax, a.amax_out);
    if (!more) break;
    lds_only_barrier();                                            // every wave has read its staged values: the weight half is free
    GR_K32_DMA_WEIGHTS(0)
    u = un;
  }
#undef GR_K32_DMA_PATCH
#undef GR_K32_DMA_WEIGHTS
}

// Epilogue stores of the up-sampling kernels.  A lane ends with the 2x2 outputs of its source pixel (x, y): two float2 per channel,
// 8 bytes per lane - store-issue-bound (MI355X_MICROARCH.md: a dwordx2-per-lane store tail runs at ~7 B/clk/CU, dwordx4 halves it;
// ablation round 3: the stores are 34-39 us of G.convB's 246 at cfg2), and rocprofv3's WRITE_SIZE reads 2.4x the bytes for them.
// Lanes l and l ^ 1 hold horizontally adjacent source pixels (tile_pixel: pc = lane & (TW - 1)), so they swap half of their values
// through DPP (quad_perm [1,0,3,2]: a VALU move, no LDS): the even lane ends with output row 2y, columns 2x .. 2x+3, the odd
// lane with row 2y+1 of the same four columns - ONE 16-byte store per lane and channel.
__device__ __forceinline__ float dpp_swap1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
}
// v = {out(2y, 2x), out(2y, 2x+1), out(2y+1, 2x), out(2y+1, 2x+1)} of this lane's source pixel; returns the lane's float4 and the output
// row phase (0 / 1) and source column (x or x - 1) it belongs to
__device__ __forceinline__ float4 up2_pair_rows(const float* v, bool odd) {
  const float s0 = odd ? v[0] : v[2], s1 = odd ? v[1] : v[3];          // what the partner needs: my other row
  const float r0 = dpp_swap1(s0), r1 = dpp_swap1(s1);
  return odd ? make_float4(r0, r1, v[2], v[3]) : make_float4(v[0], v[1], r0, r1);
}
// ---------------------------------------------------------------- nearest x2 up-sampling + 3x3 convolution as four 2x2 convolutions
// nn.SpatialUpSamplingNearest(2) followed by the 3x3 convolution (G: models.lua:121-122,127-128) reads every source pixel
// through several taps: for output pixel (2y+a, 2x+b) the three tap rows 2y+a-1 .. 2y+a+1 of the up-sampled plane are only
// TWO source rows (y-1+a and y+a), and likewise the columns.  Summing the taps that land on the same source pixel turns the
// layer into four 2x2 convolutions of the SOURCE plane, one per output phase (a, b): 4 multiply-adds per output instead
// of 9 (2.25x fewer MFMAs), identical zero-padding behaviour (a tap group never straddles the border), and the only
// numerical difference is the order in which up to four weights are added (done in double by the prep kernel).
// Workgroup: 512 source pixels x 32 output channels x all four phases (a 128-row "virtual channel" block: accumulator
// block [a][b], so a lane ends up with the 2x2 outputs of its source pixel and stores two float2).  K is walked in chunks
// of 16 input channels like the plain kernel; per chunk 3 rows x 3 source columns of B operands feed 16 weight slots
// (a, dy, b, dx) - one patch conversion serves 96 MFMAs per wave.  f16x3 arithmetic only.
//   LDS patch   [2 terms][2 halves][PS] (zero-padded source patch, 1-pixel halo; NI whole images stacked)
//   LDS weights [2 terms][2 a][8 slots = (dy, b, dx)][2 halves][32 o]
// MFMA shape: v_mfma_f32_16x16x32_f16.  A bare LDS + MFMA loop holds 12-15 % more on it than on 32x32x16 (the chip keeps a higher clock
// under it: roofline.sustained in the bench line), and this layer's K fits it exactly - two column slots x 16 channels per (row phase, dy,
// column phase).  Round 3, same box, against the 32x32x16 version of this kernel (same LDS and HBM images; git history): 202 -> 177 and
// 184 -> 168 us at cfg2, 1687 -> 1498 and 1385 -> 1310 us at cfg3.  Accumulator block: lane = pixel of a 16-pixel block, register = channel.
// one source row r3_ of a chunk: per column phase b the B operands of the wave's NB pixel blocks (both terms), per valid row phase a the
// A operands of its MB channel blocks, three products per accumulator block (small terms first, as split_mma orders them).
// Uses the enclosing kernel's acc[2][2][MB][NB], pix[NB], wq, PS, PC.
#define GR_UP16_ROW(pc_, wc_, r3_)                                                                              \
  _Pragma("unroll") for (int pb = 0; pb < 2; ++pb) {                                                           \
    uint4 bv[NB][2];                                                                                           \
    _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                                          \
      _Pragma("unroll") for (int t = 0; t < 2; ++t) bv[nb][t] = (pc_)[t * 2 * PSL + pix[nb] + (r3_) * PC + pb]; \
    _Pragma("unroll") for (int pa = 0; pa < 2; ++pa) {                                                         \
      const int dy = (r3_) - pa;                    /* row phase a reads source rows y-1+a (dy 0) and y+a (dy 1) */ \
      if (dy < 0 || dy > 1) continue;                                                                          \
      uint4 av[MB][2];                                                                                         \
      _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                                        \
        _Pragma("unroll") for (int t = 0; t < 2; ++t) av[mb][t] = (wc_)[((t * 2 + pa) * 8 + (dy * 2 + pb) * 2) * 64 + wq + mb * 16]; \
      _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                                        \
        _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) {                                                    \
          f32x4 c_ = acc[pa][pb][mb][nb];                                                                      \
          c_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av[mb][1]), __builtin_bit_cast(f16x8, bv[nb][0]), c_, 0, 0, 0); \
          c_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av[mb][0]), __builtin_bit_cast(f16x8, bv[nb][1]), c_, 0, 0, 0); \
          c_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av[mb][0]), __builtin_bit_cast(f16x8, bv[nb][0]), c_, 0, 0, 0); \
          acc[pa][pb][mb][nb] = c_;                                                                            \
        }                                                                                                      \
    }                                                                                                          \
  }
template <int TW, int NI, bool DB>
__global__ __launch_bounds__(512, 2) void conv3x3_up2_f16x3_kernel(ConvArgs a, const uint4* __restrict__ wup) {
  constexpr int NT = 512, NG = 2, PT = 512, TR = PT / TW, IH = PT / (NI * TW), PR = NI * (IH + 2), PC = TW + 2, PS = PR * PC;
  // plane stride in LDS: a multiple of 16 vectors.  A ds_read_b128 whose four 16-lane groups start at different 16-byte slots of the 256-byte
  // bank row runs at half rate (tools/probe/lds_conflict_probe: +648 vectors = 8 slots: 127 B/clk/CU, +656: 239) - on the counters a third
  // of this kernel's LDS cycles were bank conflicts with the unpadded planes (round 3, 16x16x32 lane mapping: lane quarter = channel half)
  constexpr int PSL = (PS + 15) / 16 * 16;
  // NI > 1: the tile is NI whole images, so every halo slot of the patch is zero padding for every chunk.  Those slots are
  // zeroed once and the per-chunk staging walks only the 512 real pixels x 2 channel halves (2 per thread instead of
  // 3 (16-wide) or 4 (8-wide) slots: a third / half of the loads, conversions and LDS stores).
  constexpr bool COMPACT = NI > 1;
  constexpr int NEH = 2 * PS, NSL = COMPACT ? 2 : (NEH + NT - 1) / NT;
  constexpr int WV = 2 * 2 * 8 * 2 * 32, NWV = WV / NT;           // 2048 weight vectors per chunk: 4 per thread
  static_assert(NI == 1 || IH * NI * TW == PT, "tile must hold whole images");
  static_assert((TW == 32 && NI == 1) || (TW == 16 && NI == 2) || (TW == 8 && NI == 8), "tile_pixel assumes these tilings");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint4* patch = reinterpret_cast<uint4*>(smem_raw);              // [2][2][PSL]
  uint4* wts = patch + 2 * 2 * PSL;                                // [2 terms][2 a][8 slots][2 halves][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, q = lane >> 4, hh = q & 1, dxq = q >> 1;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int ot = bid % a.n_otiles; bid /= a.n_otiles;
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y; const int b = (bid / a.tiles_y) * NI;
  const int y0 = ty * TR, x0 = tx * TW, o0 = ot * 32;
  const int Hs = a.H >> 1, Ws = a.W >> 1;                        // source plane
  const size_t HWs = (size_t)Hs * Ws;
  const float* in_base = a.in + (size_t)b * a.Cin * HWs;
  const size_t in_left = (size_t)(a.B - b) * a.Cin * HWs * sizeof(float);
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in_base), 0,
      (int)(in_left < 0x7FFFF000ul ? in_left : 0x7FFFF000ul), 0x00020000);
  const int nchunks = (a.Cin + BF_CK - 1) / BF_CK;
  // weight image [chunk][term][a][slot][half][cout_pad]: 64 rows per chunk, this workgroup takes its 32 channels of each
  const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wup), 0,
      (int)((size_t)nchunks * 64 * a.cout_pad * 16), 0x00020000);
  const int kin = f16_scale_exp(absmax_read(a.amax_in));
  const int ktot = kin + f16_scale_exp(absmax_read(a.amax_w)) - 2;   // summed weights: up to 4 max|w|
  const float sc_in = pow2f(kin);
  int voff[NSL], clim[NSL], eoff[NSL];                             // eoff: slot index in the term-0 patch image ((half) * PS + position)
#pragma unroll
  for (int s = 0; s < NSL; ++s) {
    if (COMPACT) {
      const int q = tid + NT * s, hh = q >> 9, p = q & 511, prr = p / TW, pc = p - prr * TW, img = prr / IH, r = prr - img * IH;
      const int yy = y0 + r, xx = x0 + pc;
      const bool inb = yy < Hs && xx < Ws && b + img < a.B;
      const int so = yy * Ws + xx + (img * a.Cin + 8 * hh) * (int)HWs;
      voff[s] = inb ? so * 4 : (int)0x7FFFF000;
      clim[s] = a.Cin - 8 * hh;
      eoff[s] = hh * PSL + (img * (IH + 2) + r + 1) * PC + pc + 1;
    } else {
      const int eh = tid + NT * s, hh = eh >= PS ? 1 : 0, e = eh - hh * PS, rr = e / PC, c = e - rr * PC;
      const int yy = y0 + rr - 1, xx = x0 + c - 1;
      const bool inb = eh < NEH && yy >= 0 && yy < Hs && xx >= 0 && xx < Ws && b < a.B;
      const int so = yy * Ws + xx + (8 * hh) * (int)HWs;
      voff[s] = inb ? so * 4 : (int)0x7FFFF000;
      clim[s] = a.Cin - 8 * hh;
      eoff[s] = eh < NEH ? hh * PSL + e : -1;
    }
  }
  const int wvoff = ((tid >> 5) * a.cout_pad + o0 + (tid & 31)) * 16;   // weight vector f = tid + NT*i: row (tid>>5) + 16i
  float pv[NSL][8];
  uint4 wv[NWV];
#define GR_UP_LOAD(ch_)                                                                                   \
  {                                                                                                       \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                       \
      const int soff_ = (int)(((ch_) * BF_CK + j) * HWs * 4);                                             \
      _Pragma("unroll") for (int s = 0; s < NSL; ++s)                                                     \
        pv[s][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rin, voff[s], soff_, 0)); \
    }                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < NWV; ++i)                                                       \
      wv[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rwt, wvoff, ((ch_) * 64 + 16 * i) * a.cout_pad * 16, 0)); \
  }
#define GR_UP_STORE(patch, wts, ch_)                                                                      \
  {                                                                                                       \
    if (((ch_) + 1) * BF_CK > a.Cin) {                                                                    \
      _Pragma("unroll") for (int s = 0; s < NSL; ++s)                                                     \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) if ((ch_) * BF_CK + j >= clim[s]) pv[s][j] = 0.f;   \
    }                                                                                                     \
    _Pragma("unroll") for (int s = 0; s < NSL; ++s) {                                                     \
      if (COMPACT || eoff[s] >= 0) {                                                                      \
        uint4 t0, t1;                                                                                     \
        split8_f16(pv[s], sc_in, t0, t1);                                                                 \
        patch[eoff[s]] = t0; patch[2 * PSL + eoff[s]] = t1;                                                \
      }                                                                                                   \
    }                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < NWV; ++i) wts[tid + NT * i] = wv[i];                            \
  }
  // v_mfma_f32_16x16x32_f16: K = 32 = the two column slots dx = 0, 1 of a (row phase, dy, column phase) x 16 input channels.  Lane
  // quarter q = lane / 16 feeds K rows 8q .. 8q+7: channel half q & 1 of slot dx = q >> 1 - the weight image and the patch are read
  // as they are, at per-lane bases (weights: + dx * 64 vectors; patch: one source column to the right for dx = 1).
  constexpr int MB = 2, NB = 2 * NG;                               // 16-channel / 16-pixel blocks per wave
  f32x4 acc[2][2][MB][NB];                                         // [row phase a][column phase b][channel block][pixel block]
#pragma unroll
  for (int pa = 0; pa < 2; ++pa)
#pragma unroll
    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[pa][pb][mb][nb][r] = 0.f;
  int pix[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int p = (wave * NB + nb) * 16 + l15; int prr, pc; tile_pixel<TW>(p, prr, pc);
    const int pr = NI > 1 ? prr + 2 * (prr / IH) : prr;
    pix[nb] = hh * PSL + pr * PC + pc + dxq;                        // patch row pr = source row y - 1; column slot dx reads x - 1 + b + dx
  }
  const int wq = dxq * 64 + hh * 32 + l15;
  constexpr int LBUF = 2 * 2 * PSL + WV;                           // uint4s of one (patch, weights) image
  // the epilogue's per-channel values (bias, evaluate()-mode BatchNorm coefficients of the workgroup's 32 channels) wait in LDS behind the
  // images: as lane-dependent global loads inside the epilogue they were 40 round trips per pixel block, repeated for each of the blocks
  float* epar = reinterpret_cast<float*>(smem_raw + (size_t)(DB ? 2 : 1) * LBUF * 16);
  if (tid < 32) {
    const int o = min(o0 + tid, a.Cout - 1);
    epar[tid] = a.bias ? a.bias[o] : 0.f;
    if (a.ep.mean) { epar[32 + tid] = a.ep.mean[o]; epar[64 + tid] = a.ep.invstd[o]; epar[96 + tid] = a.ep.gamma[o]; epar[128 + tid] = a.ep.beta[o]; }
  }
  GR_UP_LOAD(0)
  if (COMPACT) {                                                   // the padding slots, once (both images when double-buffered)
    for (int i = tid; i < (DB ? 2 : 1) * LBUF; i += NT) if (i % LBUF < 2 * 2 * PSL) patch[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
  }
  if (DB) {
    // two LDS images (TW >= 16: 144-148 KB): chunk ch+1 is converted and stored after the first source row of chunk ch, one
    // barrier per chunk - as in conv3x3_split_wide_kernel
    GR_UP_STORE(patch, wts, 0)
    if (nchunks > 1) GR_UP_LOAD(1)
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
      const uint4* pc_ = patch + (ch & 1) * LBUF; const uint4* wc_ = wts + (ch & 1) * LBUF;
      uint4* pn_ = patch + ((ch + 1) & 1) * LBUF; uint4* wn_ = wts + ((ch + 1) & 1) * LBUF;
      GR_UP16_ROW(pc_, wc_, 0)
      if (ch + 1 < nchunks) GR_UP_STORE(pn_, wn_, ch + 1)
      if (ch + 2 < nchunks) GR_UP_LOAD(ch + 2)
      GR_UP16_ROW(pc_, wc_, 1)
      GR_UP16_ROW(pc_, wc_, 2)
      __syncthreads();
    }
  } else {
  for (int ch = 0; ch < nchunks; ++ch) {
    GR_UP_STORE(patch, wts, ch)
    __syncthreads();
    if (ch + 1 < nchunks) GR_UP_LOAD(ch + 1)
    GR_UP16_ROW(patch, wts, 0)
    GR_UP16_ROW(patch, wts, 1)
    GR_UP16_ROW(patch, wts, 2)
    __syncthreads();
  }
  }
#undef GR_UP_LOAD
#undef GR_UP_STORE
  // epilogue: scale back + bias (+ evaluate()-mode BatchNorm) per channel, one activation switch per block, then two output
  // rows x float2 per lane
  float omax = 0.f;
  const bool has_bn = a.ep.mean != nullptr;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int p = (wave * NB + nb) * 16 + l15; int prr, pc; tile_pixel<TW>(p, prr, pc);
    const int img = NI > 1 ? prr / IH : 0, pr = NI > 1 ? prr - img * IH : prr;
    const int y = y0 + pr, x = x0 + pc;
    const bool pin = y < Hs && x < Ws && b + img < a.B;
    float v[32];                                                   // [mb][r][pa][pb]: accumulator register r of block mb = channel mb * 16 + q * 4 + r
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cl = mb * 16 + q * 4 + r;                          // bias and BatchNorm coefficients from the LDS block (epar)
        const float bvv = epar[cl];
        float mean = 0.f, invstd = 1.f, gam = 1.f, bet = 0.f;
        if (has_bn) { mean = epar[32 + cl]; invstd = epar[64 + cl]; gam = epar[96 + cl]; bet = epar[128 + cl]; }
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
          float t = ldexpf(acc[ph >> 1][ph & 1][mb][nb][r], -ktot) + bvv;
          if (has_bn) t = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(t, mean), invstd), gam), bet);   // same order as conv_epilogue
          v[(mb * 4 + r) * 4 + ph] = t;
        }
      }
    conv_act_block<32>(a.ep, v);
    {
      // (both lanes of a pair share the row and the image: `pin` is the same for them, Ws is even)
      const bool odd = (pc & 1) != 0;
      float* orow = a.out + ((size_t)(b + img) * a.Cout * a.H + 2 * y + (odd ? 1 : 0)) * a.W + 2 * (x - (odd ? 1 : 0));
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = o0 + mb * 16 + q * 4 + r;
          const float4 res = up2_pair_rows(v + (mb * 4 + r) * 4, odd);   // every lane takes part in the exchange
          if (pin && o < a.Cout) {
            store4(orow + (size_t)o * a.H * a.W, res, a.nt_out != 0);
            omax = absmax4(omax, res);
          }
        }
    }
  }
  if (a.amax_out) absmax_commit(omax, a.amax_out);
}

// ---------------------------------------------------------------- the same layer as TWO independent four-wave workgroups per CU
// conv3x3_up2_f16x3_kernel keeps two operand images (144-148 KB): one eight-wave workgroup per CU whose waves move through
// staging, multiply and epilogue in lock-step - the structure conv3x3_p16_quad_kernel left behind for R's layers (+35 % there).
// PMC in the real step: matrix pipe busy 37 % (P16 kernels: 44-49 %), and making the workgroups persistent so that a tile's
// epilogue overlaps the next tile's loads changed nothing (round 3, measured): the loss is in the steady state, not at the tile
// boundaries.  Here a workgroup is FOUR waves (one per SIMD) on 256 source pixels x 32 output channels x the four output
// phases, ONE operand image of ~54 KB, so two workgroups share a CU and run out of phase by themselves: while one converts
// its next chunk, waits for its weights or stores its tile, the other one has the matrix pipe.  The activations still arrive as
// fp32 (G's stages run in evaluate() mode, their producers cannot know the tensor's maximum before it exists, so there is no
// operand-ready image to read) and are split on the VALU: they are prefetched into registers behind the MFMAs of the chunk
// before.  The weights are pre-split by the prep kernel, so they go HBM/L2 -> LDS by DMA (no registers, no ds_write).
// Per chunk and wave: 96 MFMAs, 62 ds_read_b128, ~60 VALU, 6 ds_write_b128, 8 DMA instructions.
// ---------------------------------------------------------------- the four-wave kernel on v_mfma_f32_16x16x32_f16 (see conv3x3_up2_f16x3_kernel)
template <int TW, int NI>
__global__ __launch_bounds__(256, 2) void conv3x3_up2q_f16x3_kernel(ConvArgs a, const uint4* __restrict__ wup) {
  constexpr int NW = 4, NT = 64 * NW, NG = 2, PT = 64 * NW, TR = PT / TW, IH = PT / (NI * TW), PR = NI * (IH + 2), PC = TW + 2, PS = PR * PC;
  static_assert(NI == 1 && (TW == 16 || TW == 32), "one image (16x16) or 8 rows of a 32-wide plane per tile");
  constexpr int NEH = 2 * PS, NSL = (NEH + NT - 1) / NT;            // (position, half) pairs staged per thread
  constexpr int PSL = (PS + 15) / 16 * 16;                          // plane stride in LDS (conv3x3_up2_f16x3_kernel: the four lane quarters of a read start at the same 16-byte slot)
  constexpr int PV = 2 * 2 * PSL, PVP = (PV + P16_PAD - 1) / P16_PAD * P16_PAD;
  constexpr int WV = 2 * 2 * 8 * 2 * 32, NWI = WV / 64, NWS = NWI / NW;   // weight vectors per chunk; DMA instructions; per wave
  static_assert(2 * (PVP + WV) * 16 <= 160 * 1024, "two workgroups per CU");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint4* patch = reinterpret_cast<uint4*>(smem_raw);              // [2 terms][2 halves][PS]
  uint4* wts = patch + PVP;                                       // [2 terms][2 a][8 slots][2 halves][32 o]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, q = lane >> 4, hh = q & 1, dxq = q >> 1;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int ot = bid % a.n_otiles; bid /= a.n_otiles;
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y; const int b = bid / a.tiles_y;
  const int y0 = ty * TR, x0 = tx * TW, o0 = ot * 32;
  // the epilogue's per-channel values in LDS behind the image (conv3x3_up2_f16x3_kernel: epar)
  float* epar = reinterpret_cast<float*>(smem_raw + (size_t)(PVP + WV) * 16);
  if (tid < 32) {
    const int o = min(o0 + tid, a.Cout - 1);
    epar[tid] = a.bias ? a.bias[o] : 0.f;
    if (a.ep.mean) { epar[32 + tid] = a.ep.mean[o]; epar[64 + tid] = a.ep.invstd[o]; epar[96 + tid] = a.ep.gamma[o]; epar[128 + tid] = a.ep.beta[o]; }
  }
  const int Hs = a.H >> 1, Ws = a.W >> 1;                        // source plane
  const size_t HWs = (size_t)Hs * Ws;
  const float* in_base = a.in + (size_t)b * a.Cin * HWs;
  const size_t in_left = (size_t)(a.B - b) * a.Cin * HWs * sizeof(float);
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in_base), 0,
      (int)(in_left < 0x7FFFF000ul ? in_left : 0x7FFFF000ul), 0x00020000);
  const int nchunks = (a.Cin + BF_CK - 1) / BF_CK;
  const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wup), 0,
      (int)((size_t)nchunks * 64 * a.cout_pad * 16), 0x00020000);
  const int kin = f16_scale_exp(absmax_read(a.amax_in));
  const int ktot = kin + f16_scale_exp(absmax_read(a.amax_w)) - 2;   // summed weights: up to 4 max|w|
  const float sc_in = pow2f(kin);
  const int dbg = GR_DBG(a.up >> 1);            // diagnostic ablations (gr_set_tuning "up2_debug"; outputs are then wrong by design): 1 no stores, 2 no MFMA, 4 no activation staging, 8 no weight DMA
  int voff[NSL], eoff[NSL];
#pragma unroll
  for (int s = 0; s < NSL; ++s) {
    const int eh = tid + NT * s, eh_h = eh >= PS ? 1 : 0, e = eh - eh_h * PS, rr = e / PC, c = e - rr * PC;
    const int yy = y0 + rr - 1, xx = x0 + c - 1;
    const bool inb = eh < NEH && yy >= 0 && yy < Hs && xx >= 0 && xx < Ws && b < a.B;
    const int so = yy * Ws + xx + (8 * eh_h) * (int)HWs;
    voff[s] = inb ? so * 4 : (int)0x7FFFF000;
    eoff[s] = eh < NEH ? eh_h * PSL + e : -1;
  }
  // weight DMA: instruction i = wave + NW * j covers LDS vectors 64 i .. 64 i + 63 = rows 2 i, 2 i + 1 of the chunk's 64 rows
  const int woff0 = ((2 * wave + (lane >> 5)) * a.cout_pad + o0 + (lane & 31)) * 16, wstep = 2 * NW * a.cout_pad * 16;
  float pv[NSL][8];
#define GR_UQ_LOAD(ch_)                                                                                   \
  {                                                                                                       \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                       \
      const int soff_ = (int)(((ch_) * BF_CK + j) * HWs * 4);                                             \
      _Pragma("unroll") for (int s = 0; s < NSL; ++s)                                                     \
        pv[s][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rin, voff[s], soff_, 0)); \
    }                                                                                                     \
  }
#define GR_UQ_DMAW(ch_)                                                                                   \
  {                                                                                                       \
    const int wsoff_ = (ch_) * 64 * a.cout_pad * 16;                                                      \
    _Pragma("unroll") for (int j = 0; j < NWS; ++j) lds_dma16(rwt, wts + 64 * (wave + NW * j), woff0 + j * wstep, wsoff_); \
  }
#define GR_UQ_STORE(ch_)                                                                                  \
  {                                                                                                       \
    if (((ch_) + 1) * BF_CK > a.Cin) {                                                                    \
      _Pragma("unroll") for (int s = 0; s < NSL; ++s) {                                                   \
        const int clim_ = a.Cin - ((tid + NT * s) >= PS ? 8 : 0);                                         \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) if ((ch_) * BF_CK + j >= clim_) pv[s][j] = 0.f;     \
      }                                                                                                   \
    }                                                                                                     \
    _Pragma("unroll") for (int s = 0; s < NSL; ++s) {                                                     \
      if (eoff[s] >= 0) {                                                                                 \
        uint4 t0, t1;                                                                                     \
        split8_f16(pv[s], sc_in, t0, t1);                                                                 \
        patch[eoff[s]] = t0; patch[2 * PSL + eoff[s]] = t1;                                                \
      }                                                                                                   \
    }                                                                                                     \
  }
  constexpr int MB = 2, NB = 2 * NG;                               // 16-channel / 16-pixel blocks per wave (conv3x3_up2_f16x3_kernel)
  f32x4 acc[2][2][MB][NB];
#pragma unroll
  for (int pa = 0; pa < 2; ++pa)
#pragma unroll
    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[pa][pb][mb][nb][r] = 0.f;
  int pix[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int p = (wave * NB + nb) * 16 + l15; int prr, pc; tile_pixel<TW>(p, prr, pc);
    pix[nb] = hh * PSL + prr * PC + pc + dxq;                       // patch row prr = source row y - 1; column slot dx reads x - 1 + b + dx
  }
  const int wq = dxq * 64 + hh * 32 + l15;
  // The two workgroups of a CU run the same program from (almost) the same start: left alone they stay IN phase - both convert,
  // both multiply (sharing the matrix pipe), both store - and the phases add up instead of overlapping (ablation, round 3:
  // skeleton 38 + MFMA 122 + staging 62 + stores 34 = 256 us against 246 measured on G.convB at cfg2).  The workgroup whose waves
  // sit in the odd wave slots of their SIMDs (HW_ID.wave_id: the one that arrived second, whatever its block index) starts late
  // by a.nchunks x 512 clocks (gr_set_tuning "up2_stagger"), about half a chunk iteration.
  if (a.nchunks > 0) {
    const unsigned hwid = __builtin_amdgcn_s_getreg(4 | (0 << 6) | ((4 - 1) << 11));     // HW_REG_HW_ID bits [3:0] = wave slot on its SIMD
    if (hwid & 1) {
#pragma unroll 1
      for (int i = 0; i < a.nchunks; ++i) __builtin_amdgcn_s_sleep(8);
    }
  }
  if (!(dbg & 4)) GR_UQ_LOAD(0)
  if (!(dbg & 8)) GR_UQ_DMAW(0)
  for (int ch = 0; ch < nchunks; ++ch) {
    if (!(dbg & 4)) GR_UQ_STORE(ch)                                // the image is free: every wave passed the barrier below
    dma_publish_barrier();                                         // this chunk's weights have landed, every wave's patch stores are visible
    if (ch + 1 < nchunks && !(dbg & 4)) GR_UQ_LOAD(ch + 1)         // lands behind the MFMAs
    if (!(dbg & 2))
    { GR_UP16_ROW(patch, wts, 0) GR_UP16_ROW(patch, wts, 1) GR_UP16_ROW(patch, wts, 2) }
    __syncthreads();                                               // every wave is past the image
    if (ch + 1 < nchunks && !(dbg & 8)) GR_UQ_DMAW(ch + 1)
  }
#undef GR_UQ_LOAD
#undef GR_UQ_DMAW
#undef GR_UQ_STORE
  // epilogue: scale back + bias (+ evaluate()-mode BatchNorm) per channel, one activation switch per block, then two output
  // rows x float2 per lane
  float omax = 0.f;
  const bool has_bn = a.ep.mean != nullptr;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int p = (wave * NB + nb) * 16 + l15; int prr, pc; tile_pixel<TW>(p, prr, pc);
    const int y = y0 + prr, x = x0 + pc;
    const bool pin = y < Hs && x < Ws && b < a.B && !(dbg & 1);
    float v[32];                                                   // [mb][r][pa][pb]
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cl = mb * 16 + q * 4 + r;                          // bias and BatchNorm coefficients from the LDS block (epar)
        const float bvv = epar[cl];
        float mean = 0.f, invstd = 1.f, gam = 1.f, bet = 0.f;
        if (has_bn) { mean = epar[32 + cl]; invstd = epar[64 + cl]; gam = epar[96 + cl]; bet = epar[128 + cl]; }
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
          float t = ldexpf(acc[ph >> 1][ph & 1][mb][nb][r], -ktot) + bvv;
          if (has_bn) t = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(t, mean), invstd), gam), bet);   // same order as conv_epilogue
          v[(mb * 4 + r) * 4 + ph] = t;
        }
      }
    conv_act_block<32>(a.ep, v);
    {
      const bool odd = (pc & 1) != 0;
      float* orow = a.out + ((size_t)b * a.Cout * a.H + 2 * y + (odd ? 1 : 0)) * a.W + 2 * (x - (odd ? 1 : 0));
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = o0 + mb * 16 + q * 4 + r;
          const float4 res = up2_pair_rows(v + (mb * 4 + r) * 4, odd);
          if (pin && o < a.Cout) {
            store4(orow + (size_t)o * a.H * a.W, res, a.nt_out != 0);
            omax = absmax4(omax, res);
          }
        }
    }
  }
  if (a.amax_out) absmax_commit(omax, a.amax_out);
}

#undef GR_UP16_ROW

// weight image of the up-sampling kernel: [cin_pad16/16][2 terms][2 a][8 slots (dy, b, dx)][2 halves][cout_pad32][8 ch] f16,
// slot weight = sum of the taps (ky, kx) that read source pixel (y - 1 + a + dy, x - 1 + b + dx): ky in {0} / {1,2} (a = 0) or
// {0,1} / {2} (a = 1) for dy = 0 / 1, the same for kx with (b, dx); summed in double, scaled by 2^(k_w - 2), split in two terms.
__device__ __forceinline__ void up2_taps(int phase, int d, int& k0, int& k1) {
  if (phase == 0) { if (d == 0) { k0 = 0; k1 = 0; } else { k0 = 1; k1 = 2; } }
  else { if (d == 0) { k0 = 0; k1 = 1; } else { k0 = 2; k1 = 2; } }
}
__device__ void weight_up2_split(const float* __restrict__ w, unsigned short* __restrict__ dst, int cin, int cout, int cin_pad, int cout_pad,
                                 const unsigned* amax, long i0, long stride) {
  const float sc = pow2f(f16_scale_exp(absmax_read(amax)) - 2);
  const long n = (long)(cin_pad / BF_CK) * 2 * 8 * 2 * cout_pad * 8;     // one thread per (chunk, a, slot, half, o, j): writes both terms
  for (long i = i0; i < n; i += stride) {
    const int j = (int)(i & 7); long r = i >> 3;
    const int oo = (int)(r % cout_pad); r /= cout_pad;
    const int hh = (int)(r & 1); r >>= 1;
    const int slot = (int)(r & 7); r >>= 3;
    const int pa = (int)(r & 1); const int ch = (int)(r >> 1);
    const int dy = slot >> 2, pb = (slot >> 1) & 1, dx = slot & 1;
    const int ci = ch * BF_CK + 8 * hh + j;
    double v = 0.0;
    if (ci < cin && oo < cout) {
      int ky0, ky1, kx0, kx1;
      up2_taps(pa, dy, ky0, ky1); up2_taps(pb, dx, kx0, kx1);
      const float* wp = w + ((long)oo * cin + ci) * 9;
      for (int ky = ky0; ky <= ky1; ++ky) for (int kx = kx0; kx <= kx1; ++kx) v += (double)wp[ky * 3 + kx];
    }
    const double x = v * (double)sc;
    const _Float16 h0 = (_Float16)x; const double rr = x - (double)h0; const _Float16 h1 = (_Float16)rr;
    const long term = (long)2 * 8 * 2 * cout_pad * 8;                  // elements of one term plane of a chunk
    const long within = ((((long)pa * 8 + slot) * 2 + hh) * cout_pad + oo) * 8 + j;
    dst[(long)ch * 2 * term + within] = __builtin_bit_cast(unsigned short, h0);
    dst[(long)ch * 2 * term + term + within] = __builtin_bit_cast(unsigned short, h1);
  }
}
__global__ void conv_weight_up2_split_kernel(const float* __restrict__ w, unsigned short* __restrict__ dst, int cin, int cout,
                                             int cin_pad, int cout_pad, const unsigned* __restrict__ amax) {
  weight_up2_split(w, dst, cin, cout, cin_pad, cout_pad, amax, blockIdx.x * (long)blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}
size_t conv_weight_up2_bytes(int cin, int cout) { return (size_t)(round_up(cin, BF_CK) / BF_CK) * 64 * round_up(cout, 32) * 16; }
bool conv_up2_supported(int Cin, int Cout, int H, int W) {
  const int Hs = H / 2, Ws = W / 2;
  if (H % 2 || W % 4 || Cout <= 4) return false;          // W % 4: the epilogue pairs horizontally adjacent source pixels into 16-byte stores
  return (Hs == 8 && Ws == 8) || (Hs == 16 && Ws == 16) || Ws >= 17;
}
void launch_conv_weight_up2_split(const float* w_native, void* dst, int cin, int cout, hipStream_t s, unsigned* amax_w, bool take_absmax) {
  if (take_absmax) launch_absmax(w_native, (long)cin * cout * 9, amax_w, s);
  KtScope kt("conv_weight_up2_split_kernel", 0.0, 0.0, s);
  hipLaunchKernelGGL(conv_weight_up2_split_kernel, dim3(512), dim3(256), 0, s, w_native, reinterpret_cast<unsigned short*>(dst), cin, cout,
                     round_up(cin, BF_CK), round_up(cout, 32), amax_w);
}
template <int TW, int NI, bool DB>
static void launch_conv_up2_db(ConvArgs a, const void* wup, hipStream_t s) {
  constexpr int TR = 512 / TW, IH = 512 / (NI * TW), PS = NI * (IH + 2) * (TW + 2), PSL = (PS + 15) / 16 * 16;
  const int Hs = a.H / 2, Ws = a.W / 2;
  a.tiles_x = (Ws + TW - 1) / TW; a.tiles_y = NI > 1 ? 1 : (Hs + TR - 1) / TR;
  a.cout_pad = round_up(a.Cout, 32); a.n_otiles = a.cout_pad / 32;
  const size_t lds = (DB ? 2 : 1) * 16 * (size_t)(2 * 2 * PSL + 2 * 2 * 8 * 2 * 32) + 5 * 32 * 4;     // + the epilogue's per-channel block
  const int grid = ((a.B + NI - 1) / NI) * a.tiles_x * a.tiles_y * a.n_otiles;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_up2_f16x3_kernel<TW, NI, DB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
  static const std::string name = "conv3x3_up2_f16x3_kernel<" + std::to_string(TW) + ", " + std::to_string(NI) + (DB ? ", true>" : ", false>");   // as rocprofv3 prints it
  const double px = (double)a.B * a.H * a.W;
  // FLOPs reported = those of the layer as the reference defines it (9 taps per output); the kernel issues 4/9 of them
  KtScope kt(name.c_str(), 2.0 * px * a.Cout * a.Cin * 9.0, 4.0 * (px * a.Cin / 4 + px * a.Cout + 9.0 * a.Cin * a.Cout), s);
  hipLaunchKernelGGL((conv3x3_up2_f16x3_kernel<TW, NI, DB>), dim3(grid), dim3(512), lds, s, a, reinterpret_cast<const uint4*>(wup));
}
template <int TW, int NI>
static void launch_conv_up2_t(const ConvArgs& a, const void* wup, hipStream_t s) {
  constexpr int IH = 512 / (NI * TW), PS = NI * (IH + 2) * (TW + 2), PSL = (PS + 15) / 16 * 16;
  constexpr bool FITS = 2 * 16 * (2 * 2 * PSL + 2 * 2 * 8 * 2 * 32) + 5 * 32 * 4 <= 160 * 1024;    // two LDS images where they fit (not the 8-wide tile)
#ifdef GR_ABLATE      // GR_UP2_DB=0: one LDS image where two fit (the A/B control; the shipping library instantiates the kernel it launches)
  static int db = -1;
  if (db < 0) { db = GR_KNOB("GR_UP2_DB", 1); }
  if (FITS && db) launch_conv_up2_db<TW, NI, FITS>(a, wup, s);
  else launch_conv_up2_db<TW, NI, false>(a, wup, s);
#else
  launch_conv_up2_db<TW, NI, FITS>(a, wup, s);
#endif
}
template <int TW, int NI>
static void launch_conv_up2q(ConvArgs a, const void* wup, hipStream_t s) {
  constexpr int PT = 256, TR = PT / TW, IH = PT / (NI * TW), PS = NI * (IH + 2) * (TW + 2), PSL = (PS + 15) / 16 * 16, PV = 2 * 2 * PSL, PVP = (PV + 63) / 64 * 64;
  const int Hs = a.H / 2, Ws = a.W / 2;
  a.tiles_x = (Ws + TW - 1) / TW; a.tiles_y = (Hs + TR - 1) / TR;
  a.cout_pad = round_up(a.Cout, 32); a.n_otiles = a.cout_pad / 32;
  a.nchunks = g_up2_stagger;
  const size_t lds = 16 * (size_t)(PVP + 2 * 2 * 8 * 2 * 32) + 5 * 32 * 4;       // + the epilogue's per-channel block
  const int grid = a.B * a.tiles_x * a.tiles_y * a.n_otiles;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_up2q_f16x3_kernel<TW, NI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
  static const std::string name = "conv3x3_up2q_f16x3_kernel<" + std::to_string(TW) + ", " + std::to_string(NI) + ">";   // as rocprofv3 prints it
  const double px = (double)a.B * a.H * a.W;
  KtScope kt(name.c_str(), 2.0 * px * a.Cout * a.Cin * 9.0, 4.0 * (px * a.Cin / 4 + px * a.Cout + 9.0 * a.Cin * a.Cout), s);
  hipLaunchKernelGGL((conv3x3_up2q_f16x3_kernel<TW, NI>), dim3(grid), dim3(256), lds, s, a, reinterpret_cast<const uint4*>(wup));
}
// in: [B, Cin, H/2, W/2]; out: [B, Cout, H, W] = conv3x3(nearest-upsample x2 (in)); wup from launch_conv_weight_up2_split
void launch_conv3x3_up2_f16x3(const float* in, const void* wup, const float* bias, float* out, int B, int Cin, int Cout, int H, int W,
                              hipStream_t s, const ConvEpilogue* ep, const unsigned* amax_in, const unsigned* amax_w, unsigned* amax_out) {
  ConvArgs a{};
  if (ep) a.ep = *ep;
  a.in = in; a.bias = bias; a.out = out; a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.up = 1 | (g_up2_debug << 1); a.nt_out = (g_nt_stores >> 1) & 1;
  a.amax_in = amax_in; a.amax_w = amax_w; a.amax_out = amax_out;
  const int Ws = W / 2, Hs = H / 2;
  const int quad = g_up2_quad;                                       // 0: the eight-wave kernels everywhere (A/B runs: gr_set_tuning "up2_quad", GR_UP2_QUAD)
  if (Ws == 8) launch_conv_up2_t<8, 8>(a, wup, s);
  // four-wave kernel: measured (round 3, same process, interleaved): 32-wide source tiles 1591 us against 1689 (G.convB at cfg3), 16x16
  // source planes 1499 against 1464 (G.convA at cfg3) and 210 against 203 (G.convB at cfg2) - so only the 32-wide tiles take it
  // (quad = 2 forces it onto the 16x16 planes too: tools/ablate_up2.py)
  else if (Ws == 16) { if (quad >= 2 && Hs == 16) launch_conv_up2q<16, 1>(a, wup, s); else launch_conv_up2_t<16, 2>(a, wup, s); }
  else { if (quad) launch_conv_up2q<32, 1>(a, wup, s); else launch_conv_up2_t<32, 1>(a, wup, s); }
}

// ---------------------------------------------------------------- max|x| of a tensor (f16x3 scale tracking)
// slot holds the bit pattern of a non-negative float (orders like an unsigned); the caller zeroes it before the first writer
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ slot) {
  float m = 0.f;
  const long n4 = n >> 2, stride = (long)gridDim.x * blockDim.x;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 v = x4[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  for (long i = (n4 << 2) + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += stride) m = fmaxf(m, fabsf(x[i]));
  absmax_commit(m, slot);
}
void launch_absmax(const float* x, long n, unsigned* slot, hipStream_t s, bool slot_is_zero) {
  if (!slot_is_zero) (void)hipMemsetAsync(slot, 0, sizeof(unsigned) * AMAX_WORDS, s);
  long blocks = (n / 4 + 255) / 256; if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
  KtScope kt("absmax_kernel", 0.0, 4.0 * (double)n, s);
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, n, slot);
}

// native fp32 [cout][cin][3][3] -> split image [cin_pad16/16][NTERM][9 taps][2 halves][cout_pad32][8 ch] of bf16 (3 terms)
// or scaled f16 (2 terms; amax = bit pattern of max|w|)  (backward-data: the transposed + flipped weights)
__device__ __forceinline__ void weight_split_store(unsigned short* dst, long base, long within, long term, float v, int nterm, float sc) {
  if (nterm == 3) {
    const unsigned short t0 = f32_to_bf16(v); const float r1 = v - bf16_to_f32(t0);
    const unsigned short t1 = f32_to_bf16(r1); const float r2 = r1 - bf16_to_f32(t1);
    const unsigned short t2 = f32_to_bf16(r2);
    dst[base + within] = t0; dst[base + term + within] = t1; dst[base + 2 * term + within] = t2;
  } else {
    const float x = v * sc;
    const _Float16 h0 = (_Float16)x; const float r = x - (float)h0; const _Float16 h1 = (_Float16)r;
    dst[base + within] = __builtin_bit_cast(unsigned short, h0); dst[base + term + within] = __builtin_bit_cast(unsigned short, h1);
  }
}
__global__ void conv_weight_split_kernel(const float* __restrict__ w, unsigned short* __restrict__ dst,
                                         int cin, int cout, int CI, int CO, int cin_pad, int cout_pad, int bwd,
                                         int nterm, const unsigned* __restrict__ amax, int kk) {      // kk = taps: 9 (3x3) or 25 (5x5)
  const long n = (long)(cin_pad / BF_CK) * kk * 2 * cout_pad * 8;     // one thread per (chunk, tap, half, o, j): writes all terms
  const float sc = nterm == 2 ? pow2f(f16_scale_exp(absmax_read(amax))) : 1.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i & 7); long r = i >> 3;
    const int oo = (int)(r % cout_pad); r /= cout_pad;
    const int hh = (int)(r & 1); r >>= 1;
    const int tap = (int)(r % kk); const int ch = (int)(r / kk);
    const int ci = ch * BF_CK + 8 * hh + j;
    float v = 0.f;
    if (ci < CI && oo < CO) v = bwd ? w[((long)ci * cin + oo) * kk + (kk - 1 - tap)] : w[((long)oo * cin + ci) * kk + tap];
    const long term = (long)kk * 2 * cout_pad * 8;
    weight_split_store(dst, (long)ch * nterm * term, (((long)tap * 2 + hh) * cout_pad + oo) * 8 + j, term, v, nterm, sc);
  }
}

size_t conv_weight_split_bytes(int cin, int cout, bool bwd, int ksz) {
  const int CI = bwd ? cout : cin, CO = bwd ? cin : cout;
  return (size_t)round_up(CI, BF_CK) * ksz * ksz * round_up(CO, 32) * 3 * sizeof(unsigned short);   // sized for 3 terms; f16x3 uses 2/3 of it
}
void launch_conv_weight_split(const float* w_native, void* dst, int cin, int cout, bool bwd, hipStream_t s, int nterm, unsigned* amax, int ksz, bool take_absmax) {
  const int CI = bwd ? cout : cin, CO = bwd ? cin : cout, kk = ksz * ksz;
  const int cin_pad = round_up(CI, BF_CK), cout_pad = round_up(CO, 32);
  const long n = (long)(cin_pad / BF_CK) * kk * 2 * cout_pad * 8;
  const int grid = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
  if (nterm == 2 && take_absmax) launch_absmax(w_native, (long)cin * cout * kk, amax, s);
  KtScope kt("conv_weight_split_kernel", 0.0, 4.0 * kk * cin * cout + 2.0 * nterm * (double)n, s);
  hipLaunchKernelGGL(conv_weight_split_kernel, dim3(grid), dim3(256), 0, s, w_native, reinterpret_cast<unsigned short*>(dst),
                     cin, cout, CI, CO, cin_pad, cout_pad, bwd ? 1 : 0, nterm, amax, kk);
}

template <int TW, int MT, int NTERM, int NI = 1>
static void launch_conv_split_t(ConvArgs a, const void* wsplit, hipStream_t s) {
  constexpr int TR = 256 / TW, IH = TR / NI, PS = NI * (IH + 2) * (TW + 2), CT = 32 * MT;
  a.tiles_x = (a.W + TW - 1) / TW; a.tiles_y = NI > 1 ? 1 : (a.H + TR - 1) / TR;        // (NI > 1: planes are exactly TW x IH, the caller checked)
  a.cout_pad = round_up(a.Cout, 32); a.n_otiles = a.cout_pad / CT;
  const size_t lds = 16 * (size_t)(NTERM * 2 * PS + NTERM * 9 * 2 * CT);
  const int grid = ((a.B + NI - 1) / NI) * a.tiles_x * a.tiles_y * a.n_otiles;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_split_kernel<TW, MT, NTERM, NI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
  static const std::string name = "conv3x3_split_kernel<" + std::to_string(TW) + ", " + std::to_string(MT) + ", " + std::to_string(NTERM) + (NI > 1 ? ", " + std::to_string(NI) : std::string()) + ">";   // as rocprofv3 prints it (NTERM 3 = bf16x6, 2 = f16x3)
  const double px = (double)a.B * a.H * a.W;
  KtScope kt(name.c_str(), 2.0 * px * a.Cout * a.Cin * 9.0, 4.0 * (px * a.Cin / (a.up ? 4 : 1) + px * a.Cout + 9.0 * a.Cin * a.Cout), s);
  hipLaunchKernelGGL((conv3x3_split_kernel<TW, MT, NTERM, NI>), dim3(grid), dim3(256 * MT), lds, s, a, reinterpret_cast<const uint4*>(wsplit));
}

template <int TW, int NI, int NTERM, bool DB>
static void launch_conv_split_wide_db(ConvArgs a, const void* wsplit, hipStream_t s) {
  constexpr int TR = 512 / TW, IH = 512 / (NI * TW), PS = NI * (IH + 2) * (TW + 2), CT = 64;
  a.tiles_x = (a.W + TW - 1) / TW; a.tiles_y = NI > 1 ? 1 : (a.H + TR - 1) / TR;
  a.cout_pad = round_up(a.Cout, 32); a.n_otiles = a.cout_pad / CT;
  const size_t lds = (DB ? 2 : 1) * 16 * (size_t)(NTERM * 2 * PS + NTERM * 9 * 2 * CT);
  static_assert((DB ? 2 : 1) * 16 * (NTERM * 2 * PS + NTERM * 9 * 2 * CT) <= 160 * 1024, "LDS");
  a.n_tiles = ((a.B + NI - 1) / NI) * a.tiles_x * a.tiles_y * a.n_otiles;
  a.stat_tiles = a.n_tiles / a.n_otiles;
  // persistent workgroups: one per CU with two LDS images, two with one (a multiple of 8 so that a workgroup's tiles stay
  // on its XCD); GR_CONV_PERSIST=0 launches one workgroup per tile
  static int persist = -1;
  if (persist < 0) { persist = GR_KNOB("GR_CONV_PERSIST", 1); }
  const int resident = 256 * (DB ? 1 : 2);
  const int grid = (persist && a.n_tiles > resident) ? resident : a.n_tiles;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_split_wide_kernel<TW, NI, NTERM, DB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
  // as rocprofv3 prints it: <TW, NI, NTERM (3 = bf16x6, 2 = f16x3), double-buffered>
  static const std::string name = "conv3x3_split_wide_kernel<" + std::to_string(TW) + ", " + std::to_string(NI) + ", " + std::to_string(NTERM) + (DB ? ", true>" : ", false>");
  const double px = (double)a.B * a.H * a.W;
  KtScope kt(name.c_str(), 2.0 * px * a.Cout * a.Cin * 9.0, 4.0 * (px * a.Cin / (a.up ? 4 : 1) + px * a.Cout + 9.0 * a.Cin * a.Cout), s);
  hipLaunchKernelGGL((conv3x3_split_wide_kernel<TW, NI, NTERM, DB>), dim3(grid), dim3(512), lds, s, a, reinterpret_cast<const uint4*>(wsplit));
}
template <int TW, int NI, int NTERM>
static void launch_conv_split_wide(const ConvArgs& a, const void* wsplit, hipStream_t s) {
  // two f16x3 images fit the 160 KB LDS (152-157 KB): double-buffered; three-term bf16 images do not
#ifdef GR_ABLATE      // GR_CONV_DB=0: single-buffered f16x3 images (the A/B control)
  static int db = -1;
  if (db < 0) { db = GR_KNOB("GR_CONV_DB", 1); }
  if (NTERM == 2 && db) launch_conv_split_wide_db<TW, NI, 2, true>(a, wsplit, s);
  else launch_conv_split_wide_db<TW, NI, NTERM, false>(a, wsplit, s);
#else
  if constexpr (NTERM == 2) launch_conv_split_wide_db<TW, NI, 2, true>(a, wsplit, s);
  else launch_conv_split_wide_db<TW, NI, NTERM, false>(a, wsplit, s);
#endif
}

// returns the number of statistics tiles written per channel (0: the chosen kernel does not produce them)
template <int NTERM>
static int launch_conv3x3_split_n(ConvArgs a, const void* wsplit, hipStream_t s) {
  const int B = a.B, Cout = a.Cout, H = a.H, W = a.W;
  const double* want_stats = a.stat_part;
  a.stat_part = nullptr;
  ConvArgs aw = a; aw.stat_part = const_cast<double*>(want_stats);      // only the 512-pixel kernels fill it
  static int variant = -1;
  if (variant < 0) { variant = GR_KNOB("GR_BF16X6_VARIANT", 0); }
  const bool wide = round_up(Cout, 32) % 64 == 0 && variant != 1;      // 64 output channels per workgroup (8 waves share one patch)
  static const bool stack8 = !GR_KNOB_SET("GR_NO_STACK8");
  // four 8x8 images per tile - while that still leaves half a workgroup per CU (batch 32 of the GAN game: 32 workgroups stacked, 0.38 -> 0.45 ms;
  // batch 256: 1.37 -> 0.5 ms)
  const long st_tiles = (B + 3) / 4, cp32 = round_up(Cout, 32) / 32;
  if (stack8 && W == 8 && H == 8 && !a.up && wide && st_tiles * (cp32 / 2) >= g_stack8_min_wgs) launch_conv_split_t<8, 2, NTERM, 4>(a, wsplit, s);
  else if (stack8 && W == 8 && H == 8 && !a.up && st_tiles * cp32 >= g_stack8_min_wgs) launch_conv_split_t<8, 1, NTERM, 4>(a, wsplit, s);
  else if (W <= 8) launch_conv_split_t<8, 1, NTERM>(a, wsplit, s);
  else if (W <= 16) {
    // two stacked 16x16 images per 512-pixel tile (G.convA 799 -> 716 us) when that still leaves a workgroup for every CU
    if (wide && variant != 4 && H == 16 && W == 16 && (long)((B + 1) / 2) * (round_up(Cout, 32) / 64) >= 256) { launch_conv_split_wide<16, 2, NTERM>(aw, wsplit, s); return want_stats ? ((B + 1) / 2) : 0; }
    else if (wide) launch_conv_split_t<16, 2, NTERM>(a, wsplit, s); else launch_conv_split_t<16, 1, NTERM>(a, wsplit, s);
  }
  else if (wide && variant != 4 && (long)H * W >= 512 && (long)B * ((H * W + 511) / 512) * (round_up(Cout, 32) / 64) >= 256) { launch_conv_split_wide<32, 1, NTERM>(aw, wsplit, s); return want_stats ? B * ((W + 31) / 32) * ((H + 15) / 16) : 0; }   // measured: -6 % vs the 256-pixel tile
  else { if (wide) launch_conv_split_t<32, 2, NTERM>(a, wsplit, s); else launch_conv_split_t<32, 1, NTERM>(a, wsplit, s); }
  return 0;
}

// nterm 3: bf16x6 (amax_* unused); nterm 2: f16x3, amax_in / amax_w = device slots holding the bit patterns of max|in|, max|w|
// 5x5 stride 1 pad 2 on the f16x3 split kernel: 256-pixel tiles (16 rows x 16 columns, or 8 rows x 32), 32 output channels per workgroup (the
// weight image of a 16-channel chunk is 25 taps x 2 terms x 2 halves x 32 x 16 B = 51 KB: two workgroups per CU)
bool conv5x5_split_supported(int Cin, int Cout, int H, int W) { return Cin >= 16 && Cin % 8 == 0 && W >= 8 && H >= 4; }
void launch_conv5x5_split(const float* in, const void* wsplit, const float* bias, float* out, int B, int Cin, int Cout, int H, int W,
                          hipStream_t s, const unsigned* amax_in, const unsigned* amax_w) {
  ConvArgs a{};
  a.in = in; a.bias = bias; a.out = out; a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.up = 0;
  a.amax_in = amax_in; a.amax_w = amax_w;
  a.cout_pad = round_up(Cout, 32); a.n_otiles = a.cout_pad / 32;
  const double px = (double)B * H * W;
  auto go = [&](auto tw) {
    constexpr int TW = decltype(tw)::value, TR = 256 / TW, PS = (TR + 4) * (TW + 4);
    a.tiles_x = (W + TW - 1) / TW; a.tiles_y = (H + TR - 1) / TR;
    const size_t lds = 16 * (size_t)(2 * 2 * PS + 2 * 25 * 2 * 32);
    const int grid = B * a.tiles_x * a.tiles_y * a.n_otiles;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv5x5_split_kernel<TW, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
    static const std::string name = "conv5x5_split_kernel<" + std::to_string(TW) + ", 1, 2>";
    KtScope kt(name.c_str(), 2.0 * px * Cout * Cin * 25.0, 4.0 * (px * Cin + px * Cout + 25.0 * Cin * Cout), s);
    hipLaunchKernelGGL((conv5x5_split_kernel<TW, 1, 2>), dim3(grid), dim3(256), lds, s, a, reinterpret_cast<const uint4*>(wsplit));
  };
  if (W <= 8) go(std::integral_constant<int, 8>{});
  else if (W <= 16) go(std::integral_constant<int, 16>{});
  else go(std::integral_constant<int, 32>{});
}
void launch_conv3x3_split(const float* in, const void* wsplit, const float* bias, float* out,
                          int B, int Cin, int Cout, int H, int W, bool up, hipStream_t s, const ConvEpilogue* ep,
                          int nterm, const unsigned* amax_in, const unsigned* amax_w, unsigned* amax_out,
                          double* stat_part, int* stat_tiles) {
  ConvArgs a{};
  if (ep) a.ep = *ep;
  a.in = in; a.wt = nullptr; a.bias = bias; a.out = out;
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.up = up ? 1 : 0;
  a.amax_in = amax_in; a.amax_w = amax_w; a.amax_out = amax_out;
  a.stat_part = stat_tiles ? stat_part : nullptr;
  const int nt = nterm == 2 ? launch_conv3x3_split_n<2>(a, wsplit, s) : launch_conv3x3_split_n<3>(a, wsplit, s);
  if (stat_tiles) *stat_tiles = nt;
}

// f16x3 convolution on an operand-ready (P16) activation: see conv3x3_p16_wide_kernel
int g_p16_debug = 0;
int g_up2_quad = GR_KNOB("GR_UP2_QUAD", 1);
int g_up2_stagger = GR_KNOB("GR_UP2_STAGGER", 0);
int g_nt_stores = GR_KNOB("GR_NT_STORES", 0);      // kernels.h store4: which kernels store non-temporally (measured: no effect; off)
int g_up2_debug = 0;       // diagnostic ablations of conv3x3_up2q_f16x3_kernel (gr_set_tuning "up2_debug")
void* g_p16_stamps = nullptr;     // diagnostic: device buffer of 32 x 8 bytes per workgroup (gr_debug_stamps)
int g_stack8_min_wgs = 128;     // smallest grid at which 8x8 planes are stacked four to a tile (gr_set_tuning("stack8_min_wgs"): tests force the path)
int g_p16_min_tiles = 128;      // below half a workgroup per CU the 256-pixel-tile kernels fill the chip better (gr_set_tuning("p16_min_tiles"): tests force the path)
bool conv_p16_supported(int B, int Cin, int Cout, int H, int W) {
  static int on = -1;
  if (on < 0) { on = GR_KNOB_SET("GR_NO_P16") ? 0 : 1; }
  if (!on || Cin % 16 != 0 || (H * W) % 256 != 0 || round_up(Cout, 32) % 64 != 0 || (size_t)B * Cin * H * W * 4 >= 0x7FFFF000ul) return false;
  const long otiles = round_up(Cout, 32) / 64;
  if (H == 16 && W == 16) return (long)((B + 1) / 2) * otiles >= g_p16_min_tiles;
  return W >= 32 && W % 32 == 0 && H % 16 == 0 && (long)B * (H / 16) * (W / 32) * otiles >= g_p16_min_tiles;
}
int g_p16_stagger = 0;           // start delay (x 512 clocks) of the second-dispatched workgroups: measured useless (tools/stagger_p16.py), kept as a knob
int g_p16_variant = 1;          // 1: four-wave workgroups, two per CU (conv3x3_p16_quad_kernel); 0: eight-wave persistent (conv3x3_p16_wide_kernel)
template <int TW, int NI, int NG = 4, int MT = 2>
static int launch_conv_p16_quad(ConvArgs a, const void* wsplit, const void* xin, hipStream_t s) {
  constexpr int PT = 128 * NG, TR = PT / TW, IH = PT / (NI * TW), PS = NI * (IH + 2) * (TW + 2), CT = 32 * MT;
  constexpr int PVP = (4 * PS + P16_PAD - 1) / P16_PAD * P16_PAD, LBUF = PVP + 36 * CT;
  a.tiles_x = (a.W + TW - 1) / TW; a.tiles_y = NI > 1 ? 1 : (a.H + TR - 1) / TR;
  a.cout_pad = round_up(a.Cout, 32); a.n_otiles = a.cout_pad / CT;
  const size_t lds = 16 * (size_t)LBUF;
  a.n_tiles = ((a.B + NI - 1) / NI) * a.tiles_x * a.tiles_y * a.n_otiles;
  a.stat_tiles = a.n_tiles / a.n_otiles;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_p16_quad_kernel<TW, NI, NG, MT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true;
    if (GR_KNOB_SET("GR_DEBUG_OCC")) {
      int nb = -1; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&conv3x3_p16_quad_kernel<TW, NI, NG, MT>), 256, lds);
      fprintf(stderr, "conv3x3_p16_quad_kernel<%d, %d> (NG %d): %zu B LDS per workgroup, occupancy query says %d workgroups per CU\n", TW, NI, NG, lds, nb);
    }
  }
  static const std::string name = "conv3x3_p16_quad_kernel<" + std::to_string(TW) + ", " + std::to_string(NI) + ", " + std::to_string(NG) + ", " + std::to_string(MT) + ">";   // as rocprofv3 prints it (default template arguments included)
  const double px = (double)a.B * a.H * a.W;
  if (a.p16_out) {      // evaluate() mode: the result leaves operand-ready (same kernel body, its own symbol)
    const size_t lds_po = lds + 16 * CT;                            // + the BatchNorm coefficients of the CT channels
    static_assert((MT == 1 ? 4 : 2) * (16 * (size_t)LBUF + 16 * CT) <= 160 * 1024, "PO: the coefficient block fits beside the images of a CU's workgroups");
    static bool attr_po = false;
    if (!attr_po) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_p16_quad_po_kernel<TW, NI, NG, MT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_po); attr_po = true; }
    static const std::string name_po = "conv3x3_p16_quad_po_kernel<" + std::to_string(TW) + ", " + std::to_string(NI) + ", " + std::to_string(NG) + ", " + std::to_string(MT) + ">";
    KtScope kt(name_po.c_str(), 2.0 * px * a.Cout * a.Cin * 9.0, 4.0 * (px * a.Cin + px * a.Cout + 9.0 * a.Cin * a.Cout), s);
    hipLaunchKernelGGL((conv3x3_p16_quad_po_kernel<TW, NI, NG, MT>), dim3(a.n_tiles), dim3(256), lds_po, s, a, reinterpret_cast<const uint4*>(wsplit), reinterpret_cast<const uint4*>(xin));
    return a.stat_tiles;
  }
  KtScope kt(name.c_str(), 2.0 * px * a.Cout * a.Cin * 9.0, 4.0 * (px * a.Cin + px * a.Cout + 9.0 * a.Cin * a.Cout), s);
  hipLaunchKernelGGL((conv3x3_p16_quad_kernel<TW, NI, NG, MT>), dim3(a.n_tiles), dim3(256), lds, s, a, reinterpret_cast<const uint4*>(wsplit), reinterpret_cast<const uint4*>(xin));
  return a.stat_tiles;
}
template <int TW, int NI>
static int launch_conv_p16_t(ConvArgs a, const void* wsplit, const void* xin, hipStream_t s) {
#ifndef GR_ABLATE      // the eight-wave persistent kernel (conv3x3_p16_wide_kernel, round 2) lost to the four-wave one: ablation build only ("p16_variant" 0)
  return launch_conv_p16_quad<TW, NI>(a, wsplit, xin, s);
#else
  if (g_p16_variant == 1) return launch_conv_p16_quad<TW, NI>(a, wsplit, xin, s);
  constexpr int TR = 512 / TW, IH = 512 / (NI * TW), PS = NI * (IH + 2) * (TW + 2), CT = 64;
  constexpr int PVP = (4 * PS + P16_PAD - 1) / P16_PAD * P16_PAD, LBUF = PVP + 36 * CT;
  a.tiles_x = (a.W + TW - 1) / TW; a.tiles_y = NI > 1 ? 1 : (a.H + TR - 1) / TR;
  a.cout_pad = round_up(a.Cout, 32); a.n_otiles = a.cout_pad / CT;
  const size_t lds = 2 * 16 * (size_t)LBUF;
  a.n_tiles = ((a.B + NI - 1) / NI) * a.tiles_x * a.tiles_y * a.n_otiles;
  a.stat_tiles = a.n_tiles / a.n_otiles;
  const int grid = a.n_tiles > 256 ? 256 : a.n_tiles;                 // persistent: one workgroup per CU walks the tiles
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_p16_wide_kernel<TW, NI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
  static const std::string name = "conv3x3_p16_wide_kernel<" + std::to_string(TW) + ", " + std::to_string(NI) + ">";   // as rocprofv3 prints it
  const double px = (double)a.B * a.H * a.W;
  KtScope kt(name.c_str(), 2.0 * px * a.Cout * a.Cin * 9.0, 4.0 * (px * a.Cin + px * a.Cout + 9.0 * a.Cin * a.Cout), s);
  hipLaunchKernelGGL((conv3x3_p16_wide_kernel<TW, NI>), dim3(grid), dim3(512), lds, s, a, reinterpret_cast<const uint4*>(wsplit), reinterpret_cast<const uint4*>(xin));
  return a.stat_tiles;
#endif
}
bool conv_p16_out_supported(int Cout) { return g_p16_variant == 1 && Cout % 8 == 0; }
// 256-pixel x 32-channel tiles, 32-channel chunks on v_mfma_f32_16x16x32_f16 (conv3x3_p16_k32_kernel): training-mode output only
template <int TW>
static int launch_conv_p16_k32(ConvArgs a, const void* wsplit, const void* xin, hipStream_t s) {
  constexpr int TR = 256 / TW, PVP = (2 * 4 * k32_ps<TW>() + 63) / 64 * 64, LBUF = PVP + 2 * 36 * 32;
  a.tiles_x = (a.W + TW - 1) / TW; a.tiles_y = (a.H + TR - 1) / TR;
  a.cout_pad = round_up(a.Cout, 32); a.n_otiles = a.cout_pad / 32;
  const size_t lds = 16 * (size_t)LBUF;
  a.n_tiles = a.B * a.tiles_x * a.tiles_y * a.n_otiles;
  a.stat_tiles = a.n_tiles / a.n_otiles;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_p16_k32_kernel<TW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
  static const std::string name = "conv3x3_p16_k32_kernel<" + std::to_string(TW) + ">";
  const double px = (double)a.B * a.H * a.W;
  KtScope kt(name.c_str(), 2.0 * px * a.Cout * a.Cin * 9.0, 4.0 * (px * a.Cin + px * a.Cout + 9.0 * a.Cin * a.Cout), s);
  // two resident workgroups per CU walk the units (a multiple of 8 workgroups: blockIdx.x & 7 = the XCD, and xcd_remap keys the unit's place on that);
  // GR_K32_PERSIST=0 (ablation build): one workgroup per unit, as in rounds 3-4
  static const int persist = GR_KNOB("GR_K32_PERSIST", 1);
#ifdef GR_ABLATE
  { static unsigned* probe = nullptr; if (GR_KNOB("GR_K32_FENCE_PROBE", 0)) { if (!probe) { (void)hipMalloc((void**)&probe, 4096); (void)hipMemset(probe, 0, 4096); } a.wt = reinterpret_cast<const float*>(probe); } }
#endif
  const int grid = (persist && a.n_tiles > 512) ? 512 : a.n_tiles;
  hipLaunchKernelGGL(conv3x3_p16_k32_kernel<TW>, dim3(grid), dim3(256), lds, s, a, reinterpret_cast<const uint4*>(wsplit), reinterpret_cast<const uint4*>(xin));
  return a.stat_tiles;
}
void launch_conv3x3_p16(const void* x_p16, const void* wsplit, const float* bias, float* out, int B, int Cin, int Cout, int H, int W,
                        hipStream_t s, const ConvEpilogue* ep, const unsigned* amax_in, const unsigned* amax_w, unsigned* amax_out,
                        double* stat_part, int* stat_tiles, const P16Out* p16o) {
  ConvArgs a{};
  if (ep) a.ep = *ep;
  if (p16o && p16o->p16 && g_p16_variant == 1) { a.p16_out = reinterpret_cast<uint4*>(p16o->p16); a.p16_scale = p16o->scale; }
  a.in = nullptr; a.wt = (g_p16_debug & 32) ? reinterpret_cast<const float*>(g_p16_stamps) : nullptr; a.bias = bias; a.out = out;
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.up = g_p16_debug; a.nchunks = g_p16_stagger; a.nt_out = g_nt_stores & 1;
  a.amax_in = amax_in; a.amax_w = amax_w; a.amax_out = amax_out;
  a.stat_part = stat_tiles ? stat_part : nullptr;
  // 16x16 planes: two-image tiles when they still give two workgroups per CU, single-image tiles otherwise
  static const int single = GR_KNOB("GR_P16_SINGLE", 1);
  const long two_img_tiles = (long)((B + 1) / 2) * (round_up(Cout, 32) / 64);
  int nt;
  static const int narrow = GR_KNOB("GR_P16_NARROW", 1);     // 32-channel output tiles on single-image tiles: four workgroups per CU (six launches 0.303 -> 0.294 ms at cfg2: small, the L2 -> LDS traffic doubles)
  static const int k32 = GR_KNOB("GR_P16_K32", 1);      // 32-channel chunks on the 16x16x32 MFMA where the single-image narrow tiles run (0: the 32x32x16 kernel, the A/B control)
  const bool plain_out = a.ep.mean == nullptr && a.ep.act == ACT_NONE && !a.p16_out && out != nullptr;
  if (H == 16 && W == 16 && k32 && single && narrow && g_p16_variant == 1 && two_img_tiles < 512 && Cin % 32 == 0 && plain_out && !g_p16_debug) nt = launch_conv_p16_k32<16>(a, wsplit, x_p16, s);
  else if (k32 >= 2 && !(H == 16 && W == 16) && g_p16_variant == 1 && W % 32 == 0 && H % 8 == 0 && Cin % 32 == 0 && plain_out && !g_p16_debug) nt = launch_conv_p16_k32<32>(a, wsplit, x_p16, s);      // experiment: 8-row tiles of 32-wide planes
  else if (H == 16 && W == 16) nt = (single && g_p16_variant == 1 && two_img_tiles < 512) ? (narrow ? launch_conv_p16_quad<16, 1, 2, 1>(a, wsplit, x_p16, s) : launch_conv_p16_quad<16, 1, 2>(a, wsplit, x_p16, s))
                                                                                  : launch_conv_p16_t<16, 2>(a, wsplit, x_p16, s);
  else {
    static const int half32 = GR_KNOB("GR_P16_HALF32", 0);     // 256-pixel tiles (8 rows x 32) on wider planes
    const long tiles512 = (long)B * ((H + 15) / 16) * ((W + 31) / 32) * (round_up(Cout, 32) / 64);
    nt = (half32 && g_p16_variant == 1 && H % 8 == 0 && tiles512 < half32) ? launch_conv_p16_quad<32, 1, 2>(a, wsplit, x_p16, s) : launch_conv_p16_t<32, 1>(a, wsplit, x_p16, s);
  }
  if (stat_tiles) *stat_tiles = stat_part ? nt : 0;
}

// ---------------------------------------------------------------- a-priori bound of an evaluate()-mode stage's output (kernels.h)
__global__ __launch_bounds__(64) void conv_weight_l1_kernel(const float* __restrict__ w, int fan_in, float* __restrict__ wl1) {
  const float* row = w + (size_t)blockIdx.x * fan_in;
  double t = 0.0;
  for (int i = threadIdx.x; i < fan_in; i += 64) t += (double)fabsf(row[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
  if (threadIdx.x == 0) wl1[blockIdx.x] = (float)t * 1.000001f;        // rounded up: an upper bound of the exact sum
}
void launch_conv_weight_l1(const float* w_native, int cout, int fan_in, float* wl1, hipStream_t s) {
  hipLaunchKernelGGL(conv_weight_l1_kernel, dim3(cout), dim3(64), 0, s, w_native, fan_in, wl1);
}
__global__ __launch_bounds__(256) void eval_bound_kernel(const float* __restrict__ wl1, const float* __restrict__ bias, ConvEpilogue ep, int Cout, float post_scale,
                                                         const unsigned* __restrict__ in_max, unsigned* __restrict__ bound_out) {
  __shared__ float red[4];
  const float M = __uint_as_float(absmax_read(in_max));
  float best = 0.f;
  for (int o = threadIdx.x; o < Cout; o += 256) {
    float v = wl1 ? wl1[o] * M + (bias ? fabsf(bias[o]) : 0.f) : M;
    if (ep.mean) v = (v + fabsf(ep.mean[o])) * fabsf(ep.invstd[o]) * fabsf(ep.gamma[o]) + fabsf(ep.beta[o]);
    if (ep.act == ACT_SIGMOID || ep.act == ACT_TANH) v = fminf(v, 1.f);
    else if (ep.act == ACT_LEAKYRELU) v *= fmaxf(1.f, fabsf(ep.slope));
    best = fmaxf(best, v);                                   // ELU, ReLU: |act(z)| <= |z|
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) best = fmaxf(best, __shfl_xor(best, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float b = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * post_scale * 1.001f;   // the margin covers the rounding of the sums it bounds
    bound_out[0] = __float_as_uint(fminf(b, 3.0e38f));       // entry 0 of a zeroed slot: consumers take the maximum over the entries
  }
}
void launch_eval_bound(const float* wl1, const float* bias, const ConvEpilogue* ep, int Cout, float post_scale, const unsigned* in_max, unsigned* bound_out, hipStream_t s) {
  ConvEpilogue e; if (ep) e = *ep;
  hipLaunchKernelGGL(eval_bound_kernel, dim3(1), dim3(256), 0, s, wl1, bias, e, Cout, post_scale, in_max, bound_out);
}

// ---------------------------------------------------------------- weight layout preparation
// dst[((ch*9 + tap)*8 + cil)*cout_pad + oo]
//   forward:        = W[oo][ci][tap]                (ci = ch*8+cil over Cin, oo over Cout)
//   backward-data:  = W[ci][oo][8-tap]              (ci over Cout, oo over Cin; spatial flip = 8-tap)
__global__ void conv_weight_prep_kernel(const float* __restrict__ w, float* __restrict__ dst,
                                        int cin, int cout, int CI, int CO, int cin_pad, int cout_pad, int bwd) {
  const long n = (long)cin_pad * 9 * cout_pad;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int oo = (int)(i % cout_pad); long r = i / cout_pad;
    const int cil = (int)(r % CONV_CK); r /= CONV_CK;
    const int tap = (int)(r % 9); const int ch = (int)(r / 9);
    const int ci = ch * CONV_CK + cil;
    float v = 0.f;
    if (ci < CI && oo < CO)
      v = bwd ? w[((long)ci * cin + oo) * 9 + (8 - tap)] : w[((long)oo * cin + ci) * 9 + tap];
    dst[i] = v;
  }
}

void launch_conv_weight_prep(const float* w_native, float* wt, int cin, int cout, bool bwd, hipStream_t s) {
  // forward: reduce over CI=cin, produce CO=cout ; backward-data: reduce over CI=cout, produce CO=cin
  const int CI = bwd ? cout : cin, CO = bwd ? cin : cout;
  const ConvWeightLayout L = conv_weight_layout(CI, CO);
  const long n = (long)L.elems();
  const int grid = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
  KtScope kt("conv_weight_prep_kernel", 0.0, 4.0 * (9.0 * cin * cout + (double)n), s);
  hipLaunchKernelGGL(conv_weight_prep_kernel, dim3(grid), dim3(256), 0, s, w_native, wt, cin, cout, CI, CO,
                     L.cin_pad, L.cout_pad, bwd ? 1 : 0);
}

// One launch prepares every convolution of a net (forward + backward-data images, either flavour): blockIdx.y = job.
__global__ void conv_weight_prep_batch_kernel(const PrepJob* __restrict__ jobs, const float* __restrict__ params) {
  const PrepJob j = jobs[blockIdx.y];
  if (j.split == 5) return;      // nn.Linear weights of the f16x3 GEMM: only their maximum is taken (the GEMM splits while staging)
  const float* w = params + j.w_off;
  if (j.split) {
    unsigned short* dst = reinterpret_cast<unsigned short*>(j.dst);
    const int nterm = j.split == 2 ? 2 : 3;
    const float sc = nterm == 2 ? pow2f(f16_scale_exp(absmax_read(j.amax))) : 1.f;
    const long n = (long)(j.cin_pad / BF_CK) * 9 * 2 * j.cout_pad * 8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
      const int jj = (int)(i & 7); long r = i >> 3;
      const int oo = (int)(r % j.cout_pad); r /= j.cout_pad;
      const int hh = (int)(r & 1); r >>= 1;
      const int tap = (int)(r % 9); const int ch = (int)(r / 9);
      const int ci = ch * BF_CK + 8 * hh + jj;
      float v = 0.f;
      if (ci < j.CI && oo < j.CO) v = j.bwd ? w[((long)ci * j.cin + oo) * 9 + (8 - tap)] : w[((long)oo * j.cin + ci) * 9 + tap];
      const long term = (long)9 * 2 * j.cout_pad * 8;
      weight_split_store(dst, (long)ch * nterm * term, (((long)tap * 2 + hh) * j.cout_pad + oo) * 8 + jj, term, v, nterm, sc);
    }
  } else {
    float* dst = reinterpret_cast<float*>(j.dst);
    const long n = (long)j.cin_pad * 9 * j.cout_pad;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
      const int oo = (int)(i % j.cout_pad); long r = i / j.cout_pad;
      const int cil = (int)(r % CONV_CK); r /= CONV_CK;
      const int tap = (int)(r % 9); const int ch = (int)(r / 9);
      const int ci = ch * CONV_CK + cil;
      float v = 0.f;
      if (ci < j.CI && oo < j.CO) v = j.bwd ? w[((long)ci * j.cin + oo) * 9 + (8 - tap)] : w[((long)oo * j.cin + ci) * 9 + tap];
      dst[i] = v;
    }
  }
}

// max|w| of every f16x3 job's weight tensor (blockIdx.y = job), slots zeroed by the launcher
__global__ __launch_bounds__(256) void conv_weight_absmax_batch_kernel(const PrepJob* __restrict__ jobs, const float* __restrict__ params) {
  const PrepJob j = jobs[blockIdx.y];
  if ((j.split != 2 && j.split != 5) || j.bwd) return;   // the backward-data image shares the forward image's slot; 5 = maximum only
  const float* w = params + j.w_off;
  const long n = (long)j.cin * j.cout * (j.split == 5 ? 1 : 9);     // nn.Linear weight [cout][cin] / conv weight [cout][cin][3][3]
  // 16-byte vectors over the aligned body, scalars for the ragged ends (an nn.Linear weight is 17 M floats at cfg3: with the
  // 16 scalar-loading workgroups this kernel started with it took 1 ms of a 17 ms step)
  const long head = min(n, (long)((16 - (reinterpret_cast<uintptr_t>(w) & 15)) & 15) / 4);
  const long nv = (n - head) >> 2, tail = head + 4 * nv;
  const float4* wv = reinterpret_cast<const float4*>(w + head);
  float m = 0.f;
  {     // eight 16-byte loads in flight per thread (fc1's 4.2 M weights are 16 vectors per thread: one at a time they were 12 us of launch per step)
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i0 = blockIdx.x * (long)blockDim.x + threadIdx.x; i0 < nv; i0 += 8 * stride) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const long i = i0 + u * stride; v[u] = wv[i < nv ? i : nv - 1]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) m = absmax4(m, v[u]);       // (a clamped duplicate of the last vector changes no maximum)
    }
  }
  if (blockIdx.x == 0) {
    if ((long)threadIdx.x < head) m = fmaxf(m, fabsf(w[threadIdx.x]));
    if (tail + threadIdx.x < n) m = fmaxf(m, fabsf(w[tail + threadIdx.x]));
  }
  if (blockIdx.x * (long)blockDim.x < nv || blockIdx.x == 0) absmax_commit(m, j.amax);
}

PrepJob make_prep_job(long w_off, void* dst, int cin, int cout, bool bwd, int split, unsigned* amax) {
  PrepJob j{};
  j.w_off = w_off; j.dst = dst; j.cin = cin; j.cout = cout; j.bwd = bwd ? 1 : 0; j.split = split; j.amax = amax;
  j.CI = bwd ? cout : cin; j.CO = bwd ? cin : cout;
  j.cin_pad = round_up(j.CI, split ? BF_CK : CONV_CK); j.cout_pad = round_up(j.CO, 32);
  return j;
}
void launch_conv_weight_prep_batch(const PrepJob* jobs_dev, int njobs, const float* params, hipStream_t s,
                                   unsigned* amax_slots, int n_slots) {
  if (njobs <= 0) return;
  if (amax_slots) {                                   // f16x3 images: weight maxima first
    (void)hipMemsetAsync(amax_slots, 0, sizeof(unsigned) * AMAX_WORDS * n_slots, s);
    KtScope kt("conv_weight_absmax_batch_kernel", 0.0, 0.0, s);
    hipLaunchKernelGGL(conv_weight_absmax_batch_kernel, dim3(256, njobs), dim3(256), 0, s, jobs_dev, params);
  }
  KtScope kt("conv_weight_prep_batch_kernel", 0.0, 0.0, s);
  hipLaunchKernelGGL(conv_weight_prep_batch_kernel, dim3(96, njobs), dim3(256), 0, s, jobs_dev, params);
}

// ---------------------------------------------------------------- backward-weight
// gw[o][ci][tap] += sum_{b,y,x} dy[b,o,y,x] * x[b,ci,y+ky-1,x+kx-1]
// GEMM view: M = Cout, N = (tap, ci), K = B*H*W.  A workgroup owns a 64(o) x 64(ci) x 9(tap) block and a
// strided subset of 64-pixel tiles; wave (mt, cg) keeps 9 accumulators (one per tap) of 32(o) x 32(ci).
// Partial blocks go to a slab [split][tap][o][ci]; a second kernel sums the slabs into the native layout.
struct WgradArgs {
  const float* x; const float* dy; float* slab;
  int B, Cin, Cout, H, W;
  int tiles_x, tiles_y, n_ob, n_cb, nsplit, cinp, coutp;
  long tiles_total;
  const unsigned *amax_x = nullptr, *amax_dy = nullptr;    // f16x3 mode: bit patterns of max|x|, max|dy| (device)
};

// one scalar -> its split terms (halo columns of the weight-gradient kernels); f16 terms of the scaled value when NTERM == 2
template <int NTERM>
__device__ __forceinline__ void split1(float v, float sc, unsigned short* t) {
  if (NTERM == 3) {
    t[0] = f32_to_bf16(v); const float r1 = v - bf16_to_f32(t[0]);
    t[1] = f32_to_bf16(r1); const float r2 = r1 - bf16_to_f32(t[1]);
    t[2] = f32_to_bf16(r2);
  } else {
    const float x = v * sc;
    const _Float16 h0 = (_Float16)x; const float r = x - (float)h0; const _Float16 h1 = (_Float16)r;
    t[0] = __builtin_bit_cast(unsigned short, h0); t[1] = __builtin_bit_cast(unsigned short, h1); t[2] = 0;
  }
}
template <int NTERM>
__device__ __forceinline__ void split8(const float* x, float sc, uint4* t) {
  if (NTERM == 3) split8_bf16(x, t[0], t[1], t[2]); else split8_f16(x, sc, t[0], t[1]);
}
template <int NTERM>
__device__ __forceinline__ f32x16 mma16(const uint4& a, const uint4& b, f32x16 acc) {
  if (NTERM == 3) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}

template <int TW>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_kernel(WgradArgs a) {
  constexpr int PT = 64, TR = PT / TW, PR = TR + 2, PC = TW + 2, PSR = PR * PC, PSW = PSR | 1, DYS = PT + 1;
  constexpr int NX = (64 * PSR + 255) / 256;
  __shared__ float dyT[64 * DYS];
  __shared__ float xp[64 * PSW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int mt = wave >> 1, cg = wave & 1;
  int bid = blockIdx.x;
  const int cb = bid % a.n_cb; bid /= a.n_cb;
  const int ob = bid % a.n_ob; const int split = bid / a.n_ob;
  const int o0 = ob * 64, c0 = cb * 64;
  const int H = a.H, W = a.W;
  const size_t HW = (size_t)H * W;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int dpix = tid & 63, dpr = dpix / TW, dpc = dpix - dpr * TW;
  for (long tile = split; tile < a.tiles_total; tile += a.nsplit) {
    long t = tile;
    const int tx = (int)(t % a.tiles_x); t /= a.tiles_x;
    const int ty = (int)(t % a.tiles_y); const int b = (int)(t / a.tiles_y);
    const int y0 = ty * TR, x0 = tx * TW;
    // dy tile: [64 o][64 pixels]
    {
      const int y = y0 + dpr, x = x0 + dpc;
      const bool pin = y < H && x < W;
      const float* dbase = a.dy + ((size_t)b * a.Cout) * HW + (size_t)y * W + x;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int o = (tid >> 6) + 4 * i;
        float v = 0.f;
        if (pin && o0 + o < a.Cout) v = dbase[(size_t)(o0 + o) * HW];
        dyT[o * DYS + dpix] = v;
      }
    }
    // x patch: [64 ci][PR][PC] zero padded
    {
      const float* xbase = a.x + ((size_t)b * a.Cin) * HW;
#pragma unroll 4
      for (int i = 0; i < NX; ++i) {
        const int e = tid + 256 * i;
        if (e < 64 * PSR) {
          const int ci = e / PSR, rem = e - ci * PSR, r = rem / PC, c = rem - r * PC;
          const int yy = y0 + r - 1, xx = x0 + c - 1;
          float v = 0.f;
          if (c0 + ci < a.Cin && yy >= 0 && yy < H && xx >= 0 && xx < W) v = xbase[(size_t)(c0 + ci) * HW + (size_t)yy * W + xx];
          xp[ci * PSW + rem] = v;
        }
      }
    }
    __syncthreads();
    const float* ap = dyT + (mt * 32 + l31) * DYS + h;
    const float* bp = xp + (cg * 32 + l31) * PSW + h;
#pragma unroll
    for (int s = 0; s < PT / 2; ++s) {
      const int p0 = 2 * s, pr = p0 / TW, pc = p0 - pr * TW;
      const float av = ap[p0];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const float bv = bp[(pr + ky) * PC + pc + kx];
        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tap], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  float* sl = a.slab + (size_t)split * 9 * a.coutp * a.cinp;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = o0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const int ci = c0 + cg * 32 + l31;
      sl[((size_t)tap * a.coutp + o) * a.cinp + ci] = acc[tap][r];
    }
}

// Vectorised, software-pipelined variant (W % 4 == 0, tile = 64/TW rows x TW cols): every thread fetches its share of
// the NEXT tile (float4 interior columns + scalar halo columns + float4 dy) into registers before the MFMA phase of the
// current tile, so HBM/L2 latency hides behind 288 MFMAs and one workgroup per CU keeps the matrix pipe busy.
template <int TW>
__global__ __launch_bounds__(256, 1) void conv3x3_wgrad_vec_kernel(WgradArgs a) {
  constexpr int PT = 64, TR = PT / TW, PR = TR + 2, PC = TW + 2, PSR = PR * PC, PSW = PSR | 1, DYS = PT + 1;
  constexpr int Q4 = TW / 4;
  constexpr int NXV = 64 * PR * Q4 / 256;            // float4 loads per thread: x interior
  constexpr int NH = (64 * PR * 2 + 255) / 256;      // scalar loads per thread: x halo columns
  constexpr int NDV = 64 * PT / 4 / 256;             // float4 loads per thread: dy
  static_assert((64 * PR * Q4) % 256 == 0, "interior float4 count must divide evenly");
  __shared__ float dyT[64 * DYS];
  __shared__ float xp[64 * PSW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int mt = wave >> 1, cg = wave & 1;
  int bid = blockIdx.x;
  const int cb = bid % a.n_cb; bid /= a.n_cb;
  const int ob = bid % a.n_ob; const int split = bid / a.n_ob;
  const int o0 = ob * 64, c0 = cb * 64;
  const int H = a.H, W = a.W;
  const size_t HW = (size_t)H * W;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  float4 xv[NXV]; float hv[NH]; float4 dv[NDV];
#define GR_WG_LOAD(tile_)                                                                                     \
  {                                                                                                           \
    long t_ = (tile_);                                                                                        \
    const int tx_ = (int)(t_ % a.tiles_x); t_ /= a.tiles_x;                                                   \
    const int ty_ = (int)(t_ % a.tiles_y); const int b_ = (int)(t_ / a.tiles_y);                              \
    const int y0_ = ty_ * TR, x0_ = tx_ * TW;                                                                 \
    const float* xb_ = a.x + ((size_t)b_ * a.Cin) * HW;                                                       \
    _Pragma("unroll") for (int i = 0; i < NXV; ++i) {                                                         \
      const int f = tid + 256 * i, q = f % Q4, r = (f / Q4) % PR, ci = f / (Q4 * PR);                         \
      const int yy = y0_ + r - 1, xx = x0_ + 4 * q;                                                           \
      xv[i] = (c0 + ci < a.Cin && yy >= 0 && yy < H && xx < W)                                                \
                  ? *reinterpret_cast<const float4*>(xb_ + (size_t)(c0 + ci) * HW + (size_t)yy * W + xx)      \
                  : make_float4(0.f, 0.f, 0.f, 0.f);                                                          \
    }                                                                                                         \
    _Pragma("unroll") for (int i = 0; i < NH; ++i) {                                                          \
      const int e = tid + 256 * i, side = e & 1, r = (e >> 1) % PR, ci = (e >> 1) / PR;                       \
      const int yy = y0_ + r - 1, xx = side ? x0_ + TW : x0_ - 1;                                             \
      hv[i] = (e < 64 * PR * 2 && c0 + ci < a.Cin && yy >= 0 && yy < H && xx >= 0 && xx < W)                  \
                  ? xb_[(size_t)(c0 + ci) * HW + (size_t)yy * W + xx] : 0.f;                                  \
    }                                                                                                         \
    const float* db_ = a.dy + ((size_t)b_ * a.Cout) * HW;                                                     \
    _Pragma("unroll") for (int i = 0; i < NDV; ++i) {                                                         \
      const int f = tid + 256 * i, q = f % (PT / 4), o = f / (PT / 4);                                        \
      const int px = 4 * q, pr = px / TW, pc = px - pr * TW, y = y0_ + pr, x = x0_ + pc;                      \
      dv[i] = (o0 + o < a.Cout && y < H && x < W)                                                             \
                  ? *reinterpret_cast<const float4*>(db_ + (size_t)(o0 + o) * HW + (size_t)y * W + x)         \
                  : make_float4(0.f, 0.f, 0.f, 0.f);                                                          \
    }                                                                                                         \
  }

  long tile = split;
  bool have = tile < a.tiles_total;
  if (have) GR_WG_LOAD(tile)
  while (have) {
    // registers -> LDS (the previous tile's MFMA phase ended at the barrier below)
#pragma unroll
    for (int i = 0; i < NXV; ++i) {
      const int f = tid + 256 * i, q = f % Q4, r = (f / Q4) % PR, ci = f / (Q4 * PR);
      float* d = xp + ci * PSW + r * PC + 1 + 4 * q;
      d[0] = xv[i].x; d[1] = xv[i].y; d[2] = xv[i].z; d[3] = xv[i].w;
    }
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const int e = tid + 256 * i, side = e & 1, r = (e >> 1) % PR, ci = (e >> 1) / PR;
      if (e < 64 * PR * 2) xp[ci * PSW + r * PC + (side ? TW + 1 : 0)] = hv[i];
    }
#pragma unroll
    for (int i = 0; i < NDV; ++i) {
      const int f = tid + 256 * i, q = f % (PT / 4), o = f / (PT / 4);
      float* d = dyT + o * DYS + 4 * q;
      d[0] = dv[i].x; d[1] = dv[i].y; d[2] = dv[i].z; d[3] = dv[i].w;
    }
    __syncthreads();
    const long next = tile + a.nsplit;
    const bool have_next = next < a.tiles_total;
    if (have_next) GR_WG_LOAD(next)
    const float* ap = dyT + (mt * 32 + l31) * DYS + h;
    const float* bp = xp + (cg * 32 + l31) * PSW + h;
#pragma unroll
    for (int s = 0; s < PT / 2; ++s) {
      const int p0 = 2 * s, pr = p0 / TW, pc = p0 - pr * TW;
      const float av = ap[p0];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const float bv = bp[(pr + ky) * PC + pc + kx];
        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tap], 0, 0, 0);
      }
    }
    __syncthreads();
    tile = next; have = have_next;
  }
#undef GR_WG_LOAD
  float* sl = a.slab + (size_t)split * 9 * a.coutp * a.cinp;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = o0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const int ci = c0 + cg * 32 + l31;
      sl[((size_t)tap * a.coutp + o) * a.cinp + ci] = acc[tap][r];
    }
}

// Weight gradient on the bf16 MFMA with the same fp32-accurate 3-term split ("bf16x6").  K = pixels: one MFMA consumes 16
// consecutive pixels of one image row (lanes 0-31 the first 8, lanes 32-63 the next 8), so both operands are 16-byte vectors
// of 8 bf16 along x.  dy vectors are aligned; the x vector of tap kx is the aligned vector shifted by kx-1 columns, built in
// registers from the aligned vector and its left/right neighbour with v_alignbit (no shifted copies in LDS).
// Workgroup = 64 o x 64 ci x 9 taps over a strided set of 32-pixel tiles (1 row x 32 or 2 rows x 16), waves (mt, cg) keep
// nine 32x32 accumulators; two workgroups per CU overlap one's split/convert/store phase with the other's MFMAs.
template <int TW, int WPS, int NTERM>
__global__ __launch_bounds__(256, WPS) void conv3x3_wgrad_split_kernel(WgradArgs a) {
  constexpr int PT = 32, TR = PT / TW, PR = TR + 2, GI = TW / 8, GR = GI + 2;     // interior / total 8-pixel groups per row
  constexpr int XS = PR * GR + ((PR * GR) % 2 == 0 ? 1 : 0);                      // 16-byte units per channel, odd
  constexpr int DU = PT / 8, DS = DU + ((DU % 2 == 0) ? 1 : 0);                   // dy units per channel, odd
  constexpr int NXV = 64 * PR * GI / 256;        // interior vectors (8 floats) per thread
  constexpr int NHV = (64 * PR * 2 + 255) / 256; // halo scalars per thread
  static_assert((64 * PR * GI) % 256 == 0 && 64 * DU == 256, "staging shape");
  __shared__ uint4 xs[NTERM * 64 * XS];
  __shared__ uint4 ds[NTERM * 64 * DS];
  int kx_ = 0, kdy_ = 0;
  if (NTERM == 2) { kx_ = f16_scale_exp(absmax_read(a.amax_x)); kdy_ = f16_scale_exp(absmax_read(a.amax_dy)); }
  const float sc_x = pow2f(kx_), sc_dy = pow2f(kdy_);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int mt = wave >> 1, cg = wave & 1;
  int bid = blockIdx.x;
  const int cb = bid % a.n_cb; bid /= a.n_cb;
  const int ob = bid % a.n_ob; const int split = bid / a.n_ob;
  const int o0 = ob * 64, c0 = cb * 64;
  const int H = a.H, W = a.W;
  const size_t HW = (size_t)H * W;
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  float xv[NXV][8], hv[NHV], dv[8];
#define GR_WB_LOAD(tile_)                                                                                     \
  {                                                                                                           \
    long t_ = (tile_);                                                                                        \
    const int tx_ = (int)(t_ % a.tiles_x); t_ /= a.tiles_x;                                                   \
    const int ty_ = (int)(t_ % a.tiles_y); const int b_ = (int)(t_ / a.tiles_y);                              \
    const int y0_ = ty_ * TR, x0_ = tx_ * TW;                                                                 \
    const float* xb_ = a.x + ((size_t)b_ * a.Cin) * HW;                                                       \
    _Pragma("unroll") for (int i = 0; i < NXV; ++i) {                                                         \
      const int f = tid + 256 * i, q = f % GI, r = (f / GI) % PR, ci = f / (GI * PR);                         \
      const int yy = y0_ + r - 1, xx = x0_ + 8 * q;                                                           \
      const bool ok = c0 + ci < a.Cin && yy >= 0 && yy < H && xx < W;                                         \
      const float4* p_ = reinterpret_cast<const float4*>(xb_ + (size_t)(c0 + ci) * HW + (size_t)yy * W + xx); \
      const float4 u0 = ok ? p_[0] : make_float4(0.f, 0.f, 0.f, 0.f), u1 = ok ? p_[1] : make_float4(0.f, 0.f, 0.f, 0.f); \
      xv[i][0] = u0.x; xv[i][1] = u0.y; xv[i][2] = u0.z; xv[i][3] = u0.w;                                     \
      xv[i][4] = u1.x; xv[i][5] = u1.y; xv[i][6] = u1.z; xv[i][7] = u1.w;                                     \
    }                                                                                                         \
    _Pragma("unroll") for (int i = 0; i < NHV; ++i) {                                                         \
      const int e = tid + 256 * i, side = e & 1, r = (e >> 1) % PR, ci = (e >> 1) / PR;                       \
      const int yy = y0_ + r - 1, xx = side ? x0_ + TW : x0_ - 1;                                             \
      hv[i] = (e < 64 * PR * 2 && c0 + ci < a.Cin && yy >= 0 && yy < H && xx >= 0 && xx < W)                  \
                  ? xb_[(size_t)(c0 + ci) * HW + (size_t)yy * W + xx] : 0.f;                                  \
    }                                                                                                         \
    {                                                                                                         \
      const int q = tid % DU, o = tid / DU, px = 8 * q, pr = px / TW, pc = px - pr * TW;                      \
      const int y = y0_ + pr, x = x0_ + pc;                                                                   \
      const bool ok = o0 + o < a.Cout && y < H && x < W;                                                      \
      const float4* p_ = reinterpret_cast<const float4*>(a.dy + ((size_t)b_ * a.Cout + o0 + o) * HW + (size_t)y * W + x); \
      const float4 u0 = ok ? p_[0] : make_float4(0.f, 0.f, 0.f, 0.f), u1 = ok ? p_[1] : make_float4(0.f, 0.f, 0.f, 0.f); \
      dv[0] = u0.x; dv[1] = u0.y; dv[2] = u0.z; dv[3] = u0.w; dv[4] = u1.x; dv[5] = u1.y; dv[6] = u1.z; dv[7] = u1.w; \
    }                                                                                                         \
  }
  long tile = split;
  bool have = tile < a.tiles_total;
  if (have) GR_WB_LOAD(tile)
  while (have) {
    // registers -> split -> LDS
#pragma unroll
    for (int i = 0; i < NXV; ++i) {
      const int f = tid + 256 * i, q = f % GI, r = (f / GI) % PR, ci = f / (GI * PR);
      uint4 tt[3];
      split8<NTERM>(xv[i], sc_x, tt);
      const int u = ci * XS + r * GR + q + 1;
#pragma unroll
      for (int t = 0; t < NTERM; ++t) xs[t * 64 * XS + u] = tt[t];
    }
#pragma unroll
    for (int i = 0; i < NHV; ++i) {
      const int e = tid + 256 * i, side = e & 1, r = (e >> 1) % PR, ci = (e >> 1) / PR;
      if (e < 64 * PR * 2) {
        // left halo group: only its last element (column x0-1) is ever read; right halo group: only its first (column x0+TW)
        unsigned short st[3];
        split1<NTERM>(hv[i], sc_x, st);
        const int u = ci * XS + r * GR + (side ? GR - 1 : 0);
#pragma unroll
        for (int t = 0; t < NTERM; ++t) xs[t * 64 * XS + u] = side ? make_uint4(st[t], 0, 0, 0) : make_uint4(0, 0, 0, (unsigned)st[t] << 16);
      }
    }
    {
      const int q = tid % DU, o = tid / DU;
      uint4 tt[3];
      split8<NTERM>(dv, sc_dy, tt);
      const int u = o * DS + q;
#pragma unroll
      for (int t = 0; t < NTERM; ++t) ds[t * 64 * DS + u] = tt[t];
    }
    __syncthreads();
    const long next = tile + a.nsplit;
    const bool have_next = next < a.tiles_total;
    if (have_next) GR_WB_LOAD(next)
    // compute: k16 steps = (tile row, 16-pixel half of the row)
#pragma unroll
    for (int ks = 0; ks < PT / 16; ++ks) {
      const int pr = (16 * ks) / TW, g = ((16 * ks) % TW) / 8 + h;          // this lane's 8-pixel group in row pr
      uint4 av[NTERM];
#pragma unroll
      for (int s = 0; s < NTERM; ++s) av[s] = ds[(s * 64 + mt * 32 + l31) * DS + pr * GI + g];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int ub = (cg * 32 + l31) * XS + (pr + ky) * GR + g;            // left neighbour unit; +1 current, +2 right neighbour
#pragma unroll
        for (int t = 0; t < NTERM; ++t) {                                    // term of x; pairs with dy terms 0 .. NTERM-1-t
          const uint4 vl = xs[t * 64 * XS + ub], vc = xs[t * 64 * XS + ub + 1], vr = xs[t * 64 * XS + ub + 2];
          uint4 k0, k2;
          k0.x = __builtin_amdgcn_alignbit(vc.x, vl.w, 16); k0.y = __builtin_amdgcn_alignbit(vc.y, vc.x, 16);
          k0.z = __builtin_amdgcn_alignbit(vc.z, vc.y, 16); k0.w = __builtin_amdgcn_alignbit(vc.w, vc.z, 16);
          k2.x = __builtin_amdgcn_alignbit(vc.y, vc.x, 16); k2.y = __builtin_amdgcn_alignbit(vc.z, vc.y, 16);
          k2.z = __builtin_amdgcn_alignbit(vc.w, vc.z, 16); k2.w = __builtin_amdgcn_alignbit(vr.x, vc.w, 16);
#pragma unroll
          for (int sA = NTERM - 1 - t; sA >= 0; --sA) {
            acc[ky * 3 + 0] = mma16<NTERM>(av[sA], k0, acc[ky * 3 + 0]);
            acc[ky * 3 + 1] = mma16<NTERM>(av[sA], vc, acc[ky * 3 + 1]);
            acc[ky * 3 + 2] = mma16<NTERM>(av[sA], k2, acc[ky * 3 + 2]);
          }
        }
      }
    }
    __syncthreads();
    tile = next; have = have_next;
  }
#undef GR_WB_LOAD
  float* sl = a.slab + (size_t)split * 9 * a.coutp * a.cinp;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = o0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const int ci = c0 + cg * 32 + l31;
      sl[((size_t)tap * a.coutp + o) * a.cinp + ci] = NTERM == 2 ? ldexpf(acc[tap][r], -(kx_ + kdy_)) : acc[tap][r];
    }
}

// Rolling-window variant of the bf16x6 weight gradient: a workgroup walks a run of consecutive image rows, so each x row is
// split/converted ONCE (into a ring of row groups in LDS) instead of once per tile that touches it (3x for one-row tiles).
// Per step: TR rows of dy and TR new rows of x are fetched behind the previous step's 108 MFMAs per wave.
template <int TW, int NTERM>
__global__ __launch_bounds__(256, TW == 32 ? 2 : 1) void conv3x3_wgrad_split_roll_kernel(WgradArgs a, int rows_per_seg) {
  constexpr int PT = 32, TR = PT / TW, NGP = (TR + 2) / TR, RR = NGP * TR, GI = TW / 8, GR = GI + 2;  // ring: NGP groups of TR rows
  constexpr int XS = RR * GR + ((RR * GR) % 2 == 0 ? 1 : 0);
  constexpr int DU = PT / 8, DS = DU + ((DU % 2 == 0) ? 1 : 0);
  static_assert(64 * TR * GI == 256 && 64 * DU == 256 && (TR + 2) % TR == 0, "one interior vector and one dy vector per thread");
  __shared__ uint4 xs[NTERM * 64 * XS];
  __shared__ uint4 ds[NTERM * 64 * DS];
  int kx_ = 0, kdy_ = 0;
  if (NTERM == 2) { kx_ = f16_scale_exp(absmax_read(a.amax_x)); kdy_ = f16_scale_exp(absmax_read(a.amax_dy)); }
  const float sc_x = pow2f(kx_), sc_dy = pow2f(kdy_);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int mt = wave >> 1, cg = wave & 1;
  int bid = blockIdx.x;
  const int cb = bid % a.n_cb; bid /= a.n_cb;
  const int ob = bid % a.n_ob; const int split = bid / a.n_ob;
  const int o0 = ob * 64, c0 = cb * 64;
  const int H = a.H, W = a.W;
  const size_t HW = (size_t)H * W;
  const int nspi = H / rows_per_seg, tiles_x = a.tiles_x;
  const long nsegs = (long)a.B * nspi * tiles_x;
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // this thread's share of a row group: interior vector (ci, row in group, 8-pixel group q), halo scalar, dy vector
  const int xq = tid % GI, xr = (tid / GI) % TR, xci = tid / (GI * TR);
  const int hside = tid & 1, hr = (tid >> 1) % TR, hci = (tid >> 1) / TR;     // valid for tid < 64*TR*2
  const int dq = tid % DU, dO = tid / DU;
  float xv[8], hv, dv[8];
  // fetch x rows [ybase_, ybase_+TR) of image b_ (columns x0_..x0_+TW-1 plus the two halo columns) into registers
#define GR_ROLL_LOAD_X(b_, ybase_, x0_)                                                                       \
  {                                                                                                           \
    const float* xb_ = a.x + ((size_t)(b_) * a.Cin) * HW;                                                     \
    const int yy_ = (ybase_) + xr, xx_ = (x0_) + 8 * xq;                                                      \
    const bool ok_ = c0 + xci < a.Cin && yy_ >= 0 && yy_ < H && xx_ < W;                                      \
    const float4* p_ = reinterpret_cast<const float4*>(xb_ + (size_t)(c0 + xci) * HW + (size_t)yy_ * W + xx_); \
    const float4 u0 = ok_ ? p_[0] : make_float4(0.f, 0.f, 0.f, 0.f), u1 = ok_ ? p_[1] : make_float4(0.f, 0.f, 0.f, 0.f); \
    xv[0] = u0.x; xv[1] = u0.y; xv[2] = u0.z; xv[3] = u0.w; xv[4] = u1.x; xv[5] = u1.y; xv[6] = u1.z; xv[7] = u1.w; \
    const int hy_ = (ybase_) + hr, hx_ = hside ? (x0_) + TW : (x0_) - 1;                                      \
    hv = (tid < 64 * TR * 2 && c0 + hci < a.Cin && hy_ >= 0 && hy_ < H && hx_ >= 0 && hx_ < W)                \
             ? xb_[(size_t)(c0 + hci) * HW + (size_t)hy_ * W + hx_] : 0.f;                                    \
  }
#define GR_ROLL_LOAD_DY(b_, y_, x0_)                                                                          \
  {                                                                                                           \
    const int px_ = 8 * dq, pr_ = px_ / TW, pc_ = px_ - pr_ * TW, yy_ = (y_) + pr_, xx_ = (x0_) + pc_;        \
    const bool ok_ = o0 + dO < a.Cout && yy_ < H && xx_ < W;                                                  \
    const float4* p_ = reinterpret_cast<const float4*>(a.dy + ((size_t)(b_) * a.Cout + o0 + dO) * HW + (size_t)yy_ * W + xx_); \
    const float4 u0 = ok_ ? p_[0] : make_float4(0.f, 0.f, 0.f, 0.f), u1 = ok_ ? p_[1] : make_float4(0.f, 0.f, 0.f, 0.f); \
    dv[0] = u0.x; dv[1] = u0.y; dv[2] = u0.z; dv[3] = u0.w; dv[4] = u1.x; dv[5] = u1.y; dv[6] = u1.z; dv[7] = u1.w; \
  }
  // registers -> split -> ring slot `slot_` (rows slot_*TR .. slot_*TR+TR-1 of the ring)
#define GR_ROLL_STORE_X(slot_)                                                                                \
  {                                                                                                           \
    uint4 tt_[3];                                                                                             \
    split8<NTERM>(xv, sc_x, tt_);                                                                             \
    const int u = xci * XS + ((slot_) * TR + xr) * GR + xq + 1;                                               \
    _Pragma("unroll") for (int t = 0; t < NTERM; ++t) xs[t * 64 * XS + u] = tt_[t];                           \
    if (tid < 64 * TR * 2) {                                                                                  \
      unsigned short st_[3];                                                                                  \
      split1<NTERM>(hv, sc_x, st_);                                                                           \
      const int uh = hci * XS + ((slot_) * TR + hr) * GR + (hside ? GR - 1 : 0);                              \
      _Pragma("unroll") for (int t = 0; t < NTERM; ++t)                                                       \
        xs[t * 64 * XS + uh] = hside ? make_uint4(st_[t], 0, 0, 0) : make_uint4(0, 0, 0, (unsigned)st_[t] << 16); \
    }                                                                                                         \
  }
#define GR_ROLL_STORE_DY()                                                                                    \
  {                                                                                                           \
    uint4 tt_[3];                                                                                             \
    split8<NTERM>(dv, sc_dy, tt_);                                                                            \
    const int u = dO * DS + dq;                                                                               \
    _Pragma("unroll") for (int t = 0; t < NTERM; ++t) ds[t * 64 * DS + u] = tt_[t];                           \
  }

  for (long seg = split; seg < nsegs; seg += a.nsplit) {
    long t_ = seg;
    const int tx = (int)(t_ % tiles_x); t_ /= tiles_x;
    const int sp = (int)(t_ % nspi); const int b = (int)(t_ / nspi);
    const int ys = sp * rows_per_seg, x0 = tx * TW, nsteps = rows_per_seg / TR;
    __syncthreads();                                            // previous segment's last MFMA phase is done with LDS
    // prime the ring with row groups 0 .. NGP-1 (rows ys-1 ...) and dy of step 0
#pragma unroll
    for (int g = 0; g < NGP; ++g) {
      GR_ROLL_LOAD_X(b, ys - 1 + g * TR, x0)
      GR_ROLL_STORE_X(g)
    }
    GR_ROLL_LOAD_DY(b, ys, x0)
    GR_ROLL_STORE_DY()
    __syncthreads();
    if (nsteps > 1) { GR_ROLL_LOAD_X(b, ys - 1 + NGP * TR, x0) GR_ROLL_LOAD_DY(b, ys + TR, x0) }
    for (int st = 0; st < nsteps; ++st) {
      // ring row of tile-relative row i = pr + ky (0 .. TR+1): group (st + i / TR) % NGP, row i % TR inside it
      int rowoff[TR + 2];
#pragma unroll
      for (int i = 0; i < TR + 2; ++i) rowoff[i] = (((st + i / TR) % NGP) * TR + i % TR) * GR;
#pragma unroll
      for (int ks = 0; ks < PT / 16; ++ks) {
        const int pr = (16 * ks) / TW, g = ((16 * ks) % TW) / 8 + h;
        uint4 av[NTERM];
#pragma unroll
        for (int s = 0; s < NTERM; ++s) av[s] = ds[(s * 64 + mt * 32 + l31) * DS + pr * GI + g];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int ub = (cg * 32 + l31) * XS + rowoff[pr + ky] + g;
#pragma unroll
          for (int t = 0; t < NTERM; ++t) {
            const uint4 vl = xs[t * 64 * XS + ub], vc = xs[t * 64 * XS + ub + 1], vr = xs[t * 64 * XS + ub + 2];
            uint4 k0, k2;
            k0.x = __builtin_amdgcn_alignbit(vc.x, vl.w, 16); k0.y = __builtin_amdgcn_alignbit(vc.y, vc.x, 16);
            k0.z = __builtin_amdgcn_alignbit(vc.z, vc.y, 16); k0.w = __builtin_amdgcn_alignbit(vc.w, vc.z, 16);
            k2.x = __builtin_amdgcn_alignbit(vc.y, vc.x, 16); k2.y = __builtin_amdgcn_alignbit(vc.z, vc.y, 16);
            k2.z = __builtin_amdgcn_alignbit(vc.w, vc.z, 16); k2.w = __builtin_amdgcn_alignbit(vr.x, vc.w, 16);
#pragma unroll
            for (int sA = NTERM - 1 - t; sA >= 0; --sA) {
              acc[ky * 3 + 0] = mma16<NTERM>(av[sA], k0, acc[ky * 3 + 0]);
              acc[ky * 3 + 1] = mma16<NTERM>(av[sA], vc, acc[ky * 3 + 1]);
              acc[ky * 3 + 2] = mma16<NTERM>(av[sA], k2, acc[ky * 3 + 2]);
            }
          }
        }
      }
      if (st + 1 < nsteps) {
        __syncthreads();                                        // all waves done reading the oldest group and dy
        GR_ROLL_STORE_X(st % NGP)                               // group st is dead: its slot takes group st + NGP
        GR_ROLL_STORE_DY()
        __syncthreads();
        if (st + 2 < nsteps) { GR_ROLL_LOAD_X(b, ys - 1 + (st + 1 + NGP) * TR, x0) GR_ROLL_LOAD_DY(b, ys + (st + 2) * TR, x0) }
      }
    }
  }
#undef GR_ROLL_LOAD_X
#undef GR_ROLL_LOAD_DY
#undef GR_ROLL_STORE_X
#undef GR_ROLL_STORE_DY
  float* sl = a.slab + (size_t)split * 9 * a.coutp * a.cinp;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = o0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const int ci = c0 + cg * 32 + l31;
      sl[((size_t)tap * a.coutp + o) * a.cinp + ci] = NTERM == 2 ? ldexpf(acc[tap][r], -(kx_ + kdy_)) : acc[tap][r];
    }
}

// Few input channels (9*Cin <= 32: R.conv1 on gray / RGB images, models.lua:409): the whole (ci, tap) axis fits ONE
// 32-wide MFMA column block, so the GEMM is M = Cout, N = 32 (9*Cin used), K = pixels and the kernel is HBM-bound on dy.
// Wave (mt, kh): output-channel block mt, pixel half kh of each 64-pixel tile; the two halves meet in LDS at the end.
template <int TW>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_small_kernel(WgradArgs a) {
  constexpr int PT = 64, TR = PT / TW, PR = TR + 2, PC = TW + 2, PSR = PR * PC, DYS = PT + 1, MAXC = 3;
  // Round 6 (37-39 % of this kernel's LDS cycles were bank conflicts).  Lane l31 of the B operand reads the patch at the offset
  // of ITS (input channel, tap): ci * plane + ky * row + kx.  With the geometric strides (row = TW + 2 = 34 = 2 mod 32, plane = 136 = 8 mod 32) taps
  // (0, 2) / (1, 0) and (1, 2) / (2, 0) share a bank.  LDS strides of their own - row = 3 (mod 32), plane = 9 (mod 32) - put the 27 (ci, ky, kx) on banks
  // 9 ci + 3 ky + kx: all distinct (ds_read_b32 serves 32 lanes per cycle on 32 banks).
  constexpr int PCS = PC + ((3 - PC % 32) + 32) % 32;              // 35 (TW = 32), 35 (TW = 16: 18 -> 35)
  constexpr int PSS = PR * PCS + ((9 - (PR * PCS) % 32) + 32) % 32;
  static_assert(PCS % 32 == 3 && PSS % 32 == 9, "conflict-free tap offsets");
  constexpr int NDV = 64 * PT / 4 / 256;
  constexpr int NXS = (MAXC * PSR + 255) / 256;
  __shared__ float dyT[64 * DYS];
  __shared__ float xp[MAXC * PSS + 64];
  __shared__ float red[2][32 * 33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int mt = wave & 1, kh = wave >> 1;
  const int ob = blockIdx.x % a.n_ob, split = blockIdx.x / a.n_ob;
  const int o0 = ob * 64;
  const int H = a.H, W = a.W, Cin = a.Cin;
  const size_t HW = (size_t)H * W;
  // column j of the MFMA tile = (ci, tap); unused columns read offset 0 (their results are dropped by the reduce)
  int boff = 0;
  if (l31 < 9 * Cin) { const int ci = l31 / 9, tap = l31 - ci * 9, ky = tap / 3, kx = tap - ky * 3; boff = ci * PSS + ky * PCS + kx; }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float4 dv[NDV]; float xs[NXS];
#define GR_WS_LOAD(tile_)                                                                                     \
  {                                                                                                           \
    long t_ = (tile_);                                                                                        \
    const int tx_ = (int)(t_ % a.tiles_x); t_ /= a.tiles_x;                                                   \
    const int ty_ = (int)(t_ % a.tiles_y); const int b_ = (int)(t_ / a.tiles_y);                              \
    const int y0_ = ty_ * TR, x0_ = tx_ * TW;                                                                 \
    const float* xb_ = a.x + ((size_t)b_ * Cin) * HW;                                                         \
    _Pragma("unroll") for (int i = 0; i < NXS; ++i) {                                                         \
      const int e = tid + 256 * i, ci = e / PSR, rem = e - ci * PSR, r = rem / PC, c = rem - r * PC;          \
      const int yy = y0_ + r - 1, xx = x0_ + c - 1;                                                           \
      xs[i] = (ci < Cin && yy >= 0 && yy < H && xx >= 0 && xx < W) ? xb_[(size_t)ci * HW + (size_t)yy * W + xx] : 0.f; \
    }                                                                                                         \
    const float* db_ = a.dy + ((size_t)b_ * a.Cout) * HW;                                                     \
    _Pragma("unroll") for (int i = 0; i < NDV; ++i) {                                                         \
      const int f = tid + 256 * i, q = f % (PT / 4), o = f / (PT / 4);                                        \
      const int px = 4 * q, pr = px / TW, pc = px - pr * TW, y = y0_ + pr, x = x0_ + pc;                      \
      dv[i] = (o0 + o < a.Cout && y < H && x < W)                                                             \
                  ? *reinterpret_cast<const float4*>(db_ + (size_t)(o0 + o) * HW + (size_t)y * W + x)         \
                  : make_float4(0.f, 0.f, 0.f, 0.f);                                                          \
    }                                                                                                         \
  }
  long tile = split;
  bool have = tile < a.tiles_total;
  if (have) GR_WS_LOAD(tile)
  while (have) {
#pragma unroll
    for (int i = 0; i < NXS; ++i) { const int e = tid + 256 * i, ci = e / PSR, rem = e - ci * PSR, r = rem / PC, c = rem - r * PC; if (e < MAXC * PSR) xp[ci * PSS + r * PCS + c] = xs[i]; }
#pragma unroll
    for (int i = 0; i < NDV; ++i) {
      const int f = tid + 256 * i, q = f % (PT / 4), o = f / (PT / 4);
      float* d = dyT + o * DYS + 4 * q;
      d[0] = dv[i].x; d[1] = dv[i].y; d[2] = dv[i].z; d[3] = dv[i].w;
    }
    __syncthreads();
    const long next = tile + a.nsplit;
    const bool have_next = next < a.tiles_total;
    if (have_next) GR_WS_LOAD(next)
    const float* ap = dyT + (mt * 32 + l31) * DYS + h + kh * (PT / 2);
    const float* bp = xp + boff + h;
#pragma unroll
    for (int s = 0; s < PT / 4; ++s) {
      const int p0 = kh * (PT / 2) + 2 * s;     // kh is wave-uniform; pr/pc below are computed at run time from it
      const int pr = p0 / TW, pc = p0 - pr * TW;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s], bp[pr * PCS + pc], acc, 0, 0, 0);
    }
    __syncthreads();
    tile = next; have = have_next;
  }
#undef GR_WS_LOAD
  // combine the two pixel halves, then one slab [split][32 cols][coutp]
  if (kh == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[mt][((r & 3) + 8 * (r >> 2) + 4 * h) * 33 + l31] = acc[r];
  }
  __syncthreads();
  if (kh == 0) {
    float* sl = a.slab + (size_t)split * 32 * a.coutp;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
      sl[(size_t)l31 * a.coutp + o0 + mt * 32 + row] = acc[r] + red[mt][row * 33 + l31];
    }
  }
}

// slab [split][32][coutp] -> gw[o][ci][tap] += sum over splits (fixed order): 16 outputs x 16 split groups per block
__global__ __launch_bounds__(256) void conv3x3_wgrad_small_reduce_kernel(const float* __restrict__ slab, float* __restrict__ gw,
                                                                         int Cin, int Cout, int coutp, int nsplit) {
  __shared__ float part[16][17];
  const int lo = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + lo, n = 9 * Cin * Cout;
  float s = 0.f;
  int o = 0, j = 0;
  if (i < n) {
    o = i % Cout; j = i / Cout;      // j = ci*9 + tap
    const int per = (nsplit + 15) / 16, k0 = grp * per, k1 = min(nsplit, k0 + per);
    const float* p = slab + (size_t)j * coutp + o;
#pragma unroll 8
    for (int k = k0; k < k1; ++k) s += p[(size_t)k * 32 * coutp];
  }
  part[grp][lo] = s;
  __syncthreads();
  if (grp == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += part[g][lo];
    gw[(size_t)o * Cin * 9 + j] += t;
  }
}

// slab [split][tap][o][ci] -> gw[o][ci][tap] += sum over splits, in a fixed order (deterministic).
// 256 threads = 32 consecutive ci x 8 split groups; the 8 partial sums meet in LDS.
__global__ __launch_bounds__(256) void conv3x3_wgrad_reduce8_kernel(const float* __restrict__ slab, float* __restrict__ gw,
                                                                    int Cin, int Cout, int cinp, int coutp, int nsplit) {
  __shared__ float part[8][33];
  const int lane = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const long n = (long)9 * Cout * cinp;
  const long i = (long)blockIdx.x * 32 + lane;
  float s = 0.f;
  int ci = 0, o = 0, tap = 0;
  if (i < n) {
    ci = (int)(i % cinp); long r = i / cinp;
    o = (int)(r % Cout); tap = (int)(r / Cout);
    const size_t stride = (size_t)9 * coutp * cinp;
    const float* p = slab + ((size_t)tap * coutp + o) * cinp + ci;
    const int per = (nsplit + 7) / 8, k0 = grp * per, k1 = min(nsplit, k0 + per);
    int k = k0;
    for (; k + 8 <= k1; k += 8) {            // eight loads in flight per lane, added in split order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(k + u) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < k1; ++k) s += p[(size_t)k * stride];
  }
  part[grp][lane] = s;
  __syncthreads();
  if (grp == 0 && i < n && ci < Cin) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) t += part[g][lane];
    gw[((size_t)o * Cin + ci) * 9 + tap] += t;
  }
}

// slab [split][tap][ob][cb][wo][wc][q][lane] float4 (the P16 weight-gradient kernels' accumulator order) -> gw[o][ci][tap] += sum
// over splits, in a fixed order (deterministic).  256 threads = 32 consecutive float4s x 8 split groups; the partial sums meet in LDS.
__global__ __launch_bounds__(256) void conv3x3_wgrad_reduce_tiled_kernel(const float4* __restrict__ slab, float* __restrict__ gw,
                                                                         int Cin, int Cout, int n_ob, int n_cb, int nsplit) {
  __shared__ float4 part[8][33];
  const int l = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const long n4 = (long)9 * Cout * Cin / 4;
  const long f = (long)blockIdx.x * 32 + l;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (f < n4) {
    const float4* p = slab + f;
    const int per = (nsplit + 7) / 8, k0 = grp * per, k1 = min(nsplit, k0 + per);
    int k = k0;
    for (; k + 8 <= k1; k += 8) {            // eight loads in flight per lane, added in split order
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(k + u) * n4];
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; k < k1; ++k) { const float4 v = p[(size_t)k * n4]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
  }
  part[grp][l] = s;
  __syncthreads();
  if (grp == 0 && f < n4) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int g = 0; g < 8; ++g) { const float4 v = part[g][l]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
    const int lane = (int)(f & 63), q = (int)(f >> 6) & 3, wc = (int)(f >> 8) & 1, wo = (int)(f >> 9) & 1;
    long rest = f >> 10;
    const int cb = (int)(rest % n_cb); rest /= n_cb;
    const int ob = (int)(rest % n_ob), tap = (int)(rest / n_ob);
    const int o = ob * 64 + wo * 32 + 8 * q + 4 * (lane >> 5), ci = cb * 64 + wc * 32 + (lane & 31);
    float* g0 = gw + ((size_t)o * Cin + ci) * 9 + tap;
    const size_t so = (size_t)Cin * 9;
    g0[0] += t.x; g0[so] += t.y; g0[2 * so] += t.z; g0[3 * so] += t.w;
  }
}

#ifdef GR_ABLATE
// Probe (ablation build, GR_WGRAD_REDUCE_PROBE=1): what "the LAST ARRIVER of an (output block, channel block) tile sums its tile's
// splits" costs - ONE workgroup per tile walks the tile's 9 x 64 x 64 / 4 float4 positions and adds the nsplit slabs in split order, exactly the reads the
// folded epilogue would issue from the last workgroup of the producer (a lower bound: the real thing also waits for the slowest split).  Result discarded
// (written to the slab's first split, which nobody reads afterwards).  Timed by the kernel timer as "wgrad_reduce_one_wg_probe".
__global__ __launch_bounds__(512) void conv3x3_wgrad_reduce_one_wg_probe_kernel(float4* __restrict__ slab, int n_ob, int n_cb, int nsplit, long n4) {
  const int tile = blockIdx.x;                                    // (tap-major slab: tile t owns positions [tap][t][1024 float4] for the 9 taps)
  for (int i = threadIdx.x; i < 9 * 1024; i += 512) {
    const int tap = i >> 10, w = i & 1023;
    const long f = ((long)tap * n_ob * n_cb + tile) * 1024 + w;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = 0;
    for (; k + 8 <= nsplit; k += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = slab[(size_t)(k + u) * n4 + f];
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; k < nsplit; ++k) { const float4 v = slab[(size_t)k * n4 + f]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    slab[f] = s;
  }
}
#endif
__global__ void conv3x3_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ gw,
                                            int Cin, int Cout, int cinp, int coutp, int nsplit) {
  const long n = (long)9 * Cout * cinp;
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int ci = (int)(i % cinp); long r = i / cinp;
  const int o = (int)(r % Cout); const int tap = (int)(r / Cout);
  if (ci >= Cin) return;
  const size_t stride = (size_t)9 * coutp * cinp;
  const float* p = slab + ((size_t)tap * coutp + o) * cinp + ci;
  float s = 0.f;
  for (int k = 0; k < nsplit; ++k) s += p[(size_t)k * stride];
  gw[((size_t)o * Cin + ci) * 9 + tap] += s;
}

// ---------------------------------------------------------------- weight gradient on operand-ready (P16) x and dy
// gw[o][ci][ky][kx] += sum over (b, y, x) of dy[b, o, y, x] * x[b, ci, y + ky - 1, x + kx - 1]  as nine GEMMs with
// M = o, N = ci, K = pixels, three f16 MFMA products per 16-pixel step (dy0 x0, dy0 x1, dy1 x0).  Both operands sit in HBM as
// [pixel][8 channels] fp16 vectors (P16) whose K index - the pixel - is the ROW of the LDS image, so the MFMA fragments (8
// consecutive k of one channel per lane) come out of LDS by the transposing read ds_read_b64_tr_b16 (4 pixels x 16 channels
// per 16 lanes), and a tap is nothing but a row offset into the zero-padded x patch: no shifted copies, no v_alignbit, no
// conversion - the image of 64 pixels (dy: 16 KB) and its padded x patch (35 KB) arrive by LDS-DMA.  Workgroup = 4 waves =
// 64 o x 64 ci (wave: 32 x 32, nine accumulators), single LDS image, two workgroups per CU; K is split over workgroups
// (contiguous runs of 64-pixel chunks), partial sums go to the slab the existing reduction adds up.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 lds_tr16(const uint4* vec, int byte_off) {     // vec: a 16-byte LDS vector; byte_off: 0 or 8
#if defined(__HIP_DEVICE_COMPILE__)
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(reinterpret_cast<const unsigned char*>(vec) + byte_off));
  return __builtin_bit_cast(uint2, v);
#else
  (void)vec; (void)byte_off; return make_uint2(0, 0);
#endif
}
struct WgradP16Args {
  const uint4* x; const uint4* dy; float* slab;
  int B, Cin, Cout, H, W, n_ob, n_cb, nsplit, cinp, coutp, units;     // units = B * H * W / 64 chunks of 64 pixels
  const unsigned *amax_x, *amax_dy;
  int dy_nt;               // non-temporal LDS-DMA of the dy stream (four-wave kernel)
  int plain_order;         // GR_WGRAD_PLAIN_ORDER=1: consecutive block ids = the combinations of one split (A/B of the XCD-aware numbering)
};
// MFMA shape: v_mfma_f32_16x16x32_f16, K = 32 pixels per step, the wave's 32 x 32 block per tap as 2 x 2 accumulator blocks of 16 x 16 (a
// 16-lane group of a transposing read covers 8 pixels of ONE 16-channel block).  Round 3, same box, against the 32x32x16 version (16 pixels
// per step, one f32x16 accumulator per tap; git history): 1018 -> 878 us (32-wide) and 794 -> 736 (64-wide) at cfg3, 137 -> 124 at cfg2.
// Workgroup -> (input-channel block, output-channel block, batch split).  The n_cb x n_ob workgroups of one split stream the SAME chunks:
// each x block is read by n_ob of them, each dy block by n_cb.  Numbered consecutively they land on n_cb x n_ob DIFFERENT XCDs (blocks are
// dealt round-robin over the 8 XCDs: b and b + 8 share one, MI355X_MICROARCH.md) and nobody finds the other's lines in its L2: the
// 128-channel layers fetched 2.07x their operands (1111 MB per launch for 537 MB at cfg3: profiles/r04_traffic_two_ways_cfg3.txt).  Here the
// combinations of a split are 8 block ids apart - the same XCD, dispatched within the same fraction of a microsecond - while consecutive ids
// walk the splits.  Speed only: any placement computes the same slabs.
__device__ __forceinline__ void wgrad_p16_block(const WgradP16Args& a, int bid, int& cb, int& ob, int& split) {
  const int C = a.n_cb * a.n_ob;
  if (C > 1 && a.nsplit % 8 == 0 && !a.plain_order) {
    const int grp = bid / (8 * C), rem = bid - grp * 8 * C, combo = rem >> 3;
    split = grp * 8 + (rem & 7);
    cb = combo % a.n_cb; ob = combo / a.n_cb;
  } else {
    cb = bid % a.n_cb; bid /= a.n_cb;
    ob = bid % a.n_ob; split = bid / a.n_ob;
  }
}
template <int W_>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_p16_kernel(WgradP16Args a) {
  constexpr int R = 64 / W_ > 0 ? 64 / W_ : 1;                 // image rows per 64-pixel chunk (W_ = 16, 32 or 64)
  constexpr int PR = R + 2, PC = W_ + 2, PS = PR * PC;          // x patch positions
  // plane strides (vectors) padded to 2 (mod 8): the four 8-channel groups a transposing read touches then start 64 bytes apart
  // in the 256-byte bank row (unpadded: 32-wide planes put all four on the same banks, and the dy planes of every width)
  constexpr int PSP = PS + (10 - PS % 8) % 8, DSP = 66;
  constexpr int XV = 16 * PSP, XVP = (XV + 63) / 64 * 64, DV = (16 * DSP + 63) / 64 * 64;      // vectors: 8 groups x 2 terms x positions
  constexpr int NXI = XVP / 64, NXS = (NXI + 3) / 4, NDS = 4;   // DMA instructions per wave: x patch, dy (16 planes / 4 waves)
  static_assert(W_ == 16 || W_ == 32 || W_ == 64, "plane widths of this path");
  static_assert(2 * (XVP + DV) * 16 <= 160 * 1024, "two workgroups per CU");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint4* xs = reinterpret_cast<uint4*>(smem_raw);               // [ci group 8][term 2][PS]
  uint4* ds = xs + XVP;                                         // [o group 8][term 2][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  // wave tile: ALL 64 output channels x the wave's 16 input channels x 9 taps.  A tap's x fragment cannot be shared between taps, so what a B read
  // feeds is the number of output-channel blocks it meets: four here (52 transposing-read pairs per 108 MFMAs) against two with 32 x 32 tiles
  // (80 per 108) - these kernels sat at 87-92 % LDS-active on the counters (round 3, profiles/r03_pmc_step_conv_cfg3.txt)
  int cb, ob, split;
  wgrad_p16_block(a, blockIdx.x, cb, ob, split);
  const int H = a.H, HW = H * W_, Gin = a.Cin >> 3, Gout = a.Cout >> 3;
  const int cpi = HW / 64;                                      // chunks per image
  const int u0 = (int)((long)split * a.units / a.nsplit), u1 = (int)((long)(split + 1) * a.units / a.nsplit);
  const size_t xbytes = (size_t)a.B * Gin * 2 * HW * 16, dbytes = (size_t)a.B * Gout * 2 * HW * 16;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(a.x), 0, (int)(xbytes < 0x7FFFF000ul ? xbytes : 0x7FFFF000ul), 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(a.dy), 0, (int)(dbytes < 0x7FFFF000ul ? dbytes : 0x7FFFF000ul), 0x00020000);
  const int ktot = f16_scale_exp(absmax_read(a.amax_x)) + f16_scale_exp(absmax_read(a.amax_dy));
  f32x4 acc[9][4];                                              // [tap][16-o block]: lane = ci (16 wave + li), register = o 16 mo + 4 G + r
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][m][r] = 0.f;
  // transposing-read addresses for v_mfma_f32_16x16x32_f16 (K = 32 pixels per step): 16-lane group G = lane >> 4 covers pixels 8 G .. +3
  // (+4 for the second read) of ONE 16-channel block; lane 4q + p of the group points at pixel row q, channels 4p .. 4p+3 of that block
  const int li = lane & 15, q = li >> 2, pp = li & 3, G = lane >> 4;
  const int chg = pp >> 1, boff = 8 * (pp & 1);                  // 8-channel group within the 16-channel block, byte offset in the vector
  const int pxl = 8 * G + q;                                     // pixel within a 32-pixel step (first read; second: + 4)
  // per-lane vector addresses of step 0, block 0; a step or a block adds a uniform offset.  x patch: 32 pixels are two rows of a 16-wide plane
  const uint4* abase = ds + chg * 2 * DSP + pxl;
  const uint4* bbase = xs + (wave * 2 + chg) * 2 * PSP + (W_ == 16 ? (pxl >> 4) * PC + (pxl & 15) : pxl);
  // DMA addresses: the flat index -> (plane, row, column) decomposition of a patch vector does not depend on the chunk, only the
  // image (scalar offset of the instruction) and the chunk's first row do.  Per lane and instruction: static byte offset with the
  // patch row in its low 4 bits (15 = never valid).  (Recomputing it per chunk cost ~30 VALU instructions per DMA instruction:
  // 7 VALU per MFMA on the counters.)
  int xst[NXS], dst_[NDS];
#pragma unroll
  for (int j = 0; j < NXS; ++j) {
    const int i = wave + 4 * j, e = 64 * i + lane;
    const int pl = e / PSP, pos = e - pl * PSP, g = pl >> 1, t = pl & 1, rr = pos / PC, c = pos - rr * PC, xx = c - 1;
    const bool valid = i < NXI && e < XV && pos < PS && xx >= 0 && xx < W_;
    xst[j] = valid ? (((((cb * 8 + g) * 2 + t) * HW + (rr - 1) * W_ + xx) * 16 + W_ * 16) | rr) : 15;      // (+ one row: rr - 1 may be -1; taken off again below)
  }
#pragma unroll
  for (int j = 0; j < NDS; ++j) {
    const int pl = wave + 4 * j, g = pl >> 1, t = pl & 1;
    dst_[j] = (((ob * 8 + g) * 2 + t) * HW + lane) * 16;
  }
  for (int u = u0; u < u1; ++u) {
    const int b = u / cpi, cidx = u - b * cpi, p0 = cidx * 64, y0 = p0 / W_;      // the chunk's 64 pixels: rows y0 .. y0 + R - 1 (W_ = 64: one row)
    __syncthreads();                                             // every wave is past the previous chunk's image
    // x patch: flat vector index e = 64 i + lane over [group][term][position]
    const int xsoff = b * Gin * 2 * HW * 16, dsoff = b * Gout * 2 * HW * 16;      // the image: scalar offsets (uniform)
    const int yrow = (y0 - 1) * W_ * 16, p0b = p0 * 16;
#pragma unroll
    for (int j = 0; j < NXS; ++j) {
      const int i = wave + 4 * j, rr = xst[j] & 15;
      const bool inb = rr != 15 && (unsigned)(y0 + rr - 1) < (unsigned)H;
      const int voff = inb ? (xst[j] & ~15) + yrow : (int)0x7FFFF000;
      if (i < NXI) lds_dma16(rx, xs + 64 * i, voff, xsoff);
    }
    // dy is streamed once per (input-channel block): with the non-temporal policy its lines do not push the x rows - re-read by the next
    // chunk - out of the XCD's L2 (a.dy_nt: GR_WGRAD_DY_NT)
    if (a.dy_nt) {
#pragma unroll
      for (int j = 0; j < NDS; ++j) lds_dma16_nt(rd, ds + DSP * (wave + 4 * j), dst_[j] + p0b, dsoff);
    } else {
#pragma unroll
      for (int j = 0; j < NDS; ++j) lds_dma16(rd, ds + DSP * (wave + 4 * j), dst_[j] + p0b, dsoff);      // plane = (o group, term): 64 pixels = one instruction
    }
    dma_publish_barrier();                                       // the image has landed
#pragma unroll 1
    for (int ks = 0; ks < 2; ++ks) {
      const int srow = (32 * ks) / W_, scol = 32 * ks - srow * W_;     // the step's 32 pixels start at row srow of the chunk, column scol
      // A = dy: [o][k]; two transposing reads (4 pixels each) per term and 16-channel block
      uint4 av[4][2];
#pragma unroll
      for (int mo = 0; mo < 4; ++mo)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const uint4* base = abase + (mo * 4 + t) * DSP + 32 * ks;
          const uint2 lo = lds_tr16(base, boff), hi = lds_tr16(base + 4, boff);
          av[mo][t] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
      const uint4* bstep = bbase + srow * PC + scol;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - 3 * ky;
        uint4 bv[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          // (the second read is 4 pixels further along x: same row - 8-pixel runs never straddle a row, W_ % 16 == 0)
          const uint4* base = bstep + t * PSP + ky * PC + kx;
          const uint2 lo = lds_tr16(base, boff), hi = lds_tr16(base + 4, boff);
          bv[t] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
#pragma unroll
        for (int mo = 0; mo < 4; ++mo) {
          f32x4 c_ = acc[tap][mo];
          c_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av[mo][1]), __builtin_bit_cast(f16x8, bv[0]), c_, 0, 0, 0);
          c_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av[mo][0]), __builtin_bit_cast(f16x8, bv[1]), c_, 0, 0, 0);
          c_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av[mo][0]), __builtin_bit_cast(f16x8, bv[0]), c_, 0, 0, 0);
          acc[tap][mo] = c_;
        }
      }
    }
  }
  // Slab in the ACCUMULATORS' own order, [split][tap][ob][cb][wo][wc][q = r / 4][lane] float4 = registers 4q .. 4q+3 of a lane (four
  // output channels of one input channel): 36 16-byte stores per lane, every wave-instruction 1 KB contiguous.  The [tap][o][ci]
  // order it replaces (round 3) took 144 dword stores per lane - a store-ISSUE-bound tail (MI355X_MICROARCH.md: dword / dwordx2
  // store tails run at a few bytes per clock per CU) that was most of the kernel's fixed ~23 us.  conv3x3_wgrad_reduce_tiled_kernel
  // reads the same order back with 16-byte loads and scatters only the final 9 * Cout * Cin values.
  float4* slp = reinterpret_cast<float4*>(a.slab) + (size_t)split * 9 * a.coutp * a.cinp / 4;
  // (accumulator block mo of wave w, lane group G: output channels 16 mo + 4 G .. + 3 of input channel 16 w + li - entered at the slab position
  // [wo][wc][q][lane] that a 32 x 32 accumulator layout gives that (o, ci) quad: the format conv3x3_wgrad_reduce_tiled_kernel reads)
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int mo = 0; mo < 4; ++mo) {
      const int wo = mo >> 1, wc = wave >> 1, qs = 2 * (mo & 1) + (G >> 1), ls = 32 * (G & 1) + 16 * (wave & 1) + li;
      const size_t f = ((((((size_t)tap * a.n_ob + ob) * a.n_cb + cb) * 2 + wo) * 2 + wc) * 4 + qs) * 64 + ls;
      const f32x4 c_ = acc[tap][mo];
      slp[f] = make_float4(ldexpf(c_[0], -ktot), ldexpf(c_[1], -ktot), ldexpf(c_[2], -ktot), ldexpf(c_[3], -ktot));
    }
}
// The same with the two workgroups of a CU fused into ONE of eight waves whose halves PING-PONG: while half A multiplies its
// chunk, half B requests its next chunk by DMA and waits for it; a workgroup barrier swaps the roles.  Four-wave workgroups
// left this to chance (both resident workgroups often loaded, or multiplied, at the same time); here a multiplying half always
// has the matrix pipe to itself and a loading half always has a full multiply phase to hide its DMA behind.  Each half
// accumulates its own part of the workgroup's pixel range; at the end half B's accumulators go through LDS into half A's, so
// the kernel leaves HALF as many slabs (one per CU instead of two): half the slab write and half the reduction.
// FREE (round 5): the halves do NOT alternate - each runs load -> multiply over its own chunks behind a barrier of its own four waves (an LDS arrival counter:
// gfx950 has no named barriers), exactly as two four-wave workgroups of conv3x3_wgrad_p16_kernel would, and they meet only for the hand-over at the end.  On
// 32-wide planes strict alternation lost (70 -> 98 us: the load phase is longer than the multiply phase); free-running halves keep the four-wave kernel's
// timing and still leave ONE slab per CU instead of two - half the slab write and half the reduction.
__device__ __forceinline__ void half_barrier(unsigned* cnt, unsigned& target, int lane) {
  target += 4u;                                                   // four waves arrive per crossing; the counter never wraps within a launch
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (lane == 0) atomicAdd(cnt, 1u);
  unsigned seen;
  do {
    seen = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    seen = (unsigned)__builtin_amdgcn_readfirstlane((int)seen);
    if ((int)(seen - target) < 0) __builtin_amdgcn_s_sleep(1);
  } while ((int)(seen - target) < 0);
  asm volatile("" ::: "memory");
}
template <int W_, bool FREE = false>
__global__ __launch_bounds__(512, 1) void conv3x3_wgrad_p16_pp_kernel(WgradP16Args a) {
  constexpr int R = 64 / W_ > 0 ? 64 / W_ : 1;                 // image rows per 64-pixel chunk (W_ = 16, 32 or 64)
  constexpr int PR = R + 2, PC = W_ + 2, PS = PR * PC;          // x patch positions
  // plane strides (vectors) padded to 2 (mod 8): the four 8-channel groups a transposing read touches then start 64 bytes apart
  // in the 256-byte bank row (unpadded: 32-wide planes put all four on the same banks, and the dy planes of every width)
  constexpr int PSP = PS + (10 - PS % 8) % 8, DSP = 66;
  constexpr int XV = 16 * PSP, XVP = (XV + 63) / 64 * 64, DV = (16 * DSP + 63) / 64 * 64, IMG = XVP + DV;      // vectors: 8 groups x 2 terms x positions
  constexpr int NXI = XVP / 64, NXS = (NXI + 3) / 4, NDS = 4;   // DMA instructions per wave: x patch, dy (16 planes / 4 waves)
  static_assert(W_ == 16 || W_ == 32 || W_ == 64, "plane widths of this path");
  static_assert(2 * IMG * 16 + 64 <= 160 * 1024, "two operand images (+ the halves' arrival counters)");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int half = threadIdx.x >> 8, tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  uint4* xs = reinterpret_cast<uint4*>(smem_raw) + half * IMG;  // this half's image: [ci group 8][term 2][PS]
  uint4* ds = xs + XVP;                                         // [o group 8][term 2][64]
  // wave tile: ALL 64 output channels x the wave's 16 input channels x 9 taps.  A tap's x fragment cannot be shared between taps, so what a B read
  // feeds is the number of output-channel blocks it meets: four here (52 transposing-read pairs per 108 MFMAs) against two with 32 x 32 tiles
  // (80 per 108) - these kernels sat at 87-92 % LDS-active on the counters (round 3, profiles/r03_pmc_step_conv_cfg3.txt)
  int cb, ob, split;
  wgrad_p16_block(a, blockIdx.x, cb, ob, split);
  const int H = a.H, HW = H * W_, Gin = a.Cin >> 3, Gout = a.Cout >> 3;
  const int cpi = HW / 64;                                      // chunks per image
  const int w0 = (int)((long)split * a.units / a.nsplit), w1 = (int)((long)(split + 1) * a.units / a.nsplit), wm = w0 + (w1 - w0 + 1) / 2;
  const int u0 = half ? wm : w0, u1 = half ? w1 : wm;          // half A: the first (not smaller) part of the workgroup's chunks
  const int nA = wm - w0, nmine = u1 - u0;
  const size_t xbytes = (size_t)a.B * Gin * 2 * HW * 16, dbytes = (size_t)a.B * Gout * 2 * HW * 16;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(a.x), 0, (int)(xbytes < 0x7FFFF000ul ? xbytes : 0x7FFFF000ul), 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(a.dy), 0, (int)(dbytes < 0x7FFFF000ul ? dbytes : 0x7FFFF000ul), 0x00020000);
  const int ktot = f16_scale_exp(absmax_read(a.amax_x)) + f16_scale_exp(absmax_read(a.amax_dy));
  f32x4 acc[9][4];                                              // [tap][16-o block]: lane = ci (16 wave + li), register = o 16 mo + 4 G + r
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][m][r] = 0.f;
  // transposing-read addresses for v_mfma_f32_16x16x32_f16 (K = 32 pixels per step): 16-lane group G = lane >> 4 covers pixels 8 G .. +3
  // (+4 for the second read) of ONE 16-channel block; lane 4q + p of the group points at pixel row q, channels 4p .. 4p+3 of that block
  const int li = lane & 15, q = li >> 2, pp = li & 3, G = lane >> 4;
  const int chg = pp >> 1, boff = 8 * (pp & 1);                  // 8-channel group within the 16-channel block, byte offset in the vector
  const int pxl = 8 * G + q;                                     // pixel within a 32-pixel step (first read; second: + 4)
  // per-lane vector addresses of step 0, block 0; a step or a block adds a uniform offset.  x patch: 32 pixels are two rows of a 16-wide plane
  const uint4* abase = ds + chg * 2 * DSP + pxl;
  const uint4* bbase = xs + (wave * 2 + chg) * 2 * PSP + (W_ == 16 ? (pxl >> 4) * PC + (pxl & 15) : pxl);
  // DMA addresses: the flat index -> (plane, row, column) decomposition of a patch vector does not depend on the chunk, only the
  // image (scalar offset of the instruction) and the chunk's first row do.  Per lane and instruction: static byte offset with the
  // patch row in its low 4 bits (15 = never valid).  (Recomputing it per chunk cost ~30 VALU instructions per DMA instruction:
  // 7 VALU per MFMA on the counters.)
  int xst[NXS], dst_[NDS];
#pragma unroll
  for (int j = 0; j < NXS; ++j) {
    const int i = wave + 4 * j, e = 64 * i + lane;
    const int pl = e / PSP, pos = e - pl * PSP, g = pl >> 1, t = pl & 1, rr = pos / PC, c = pos - rr * PC, xx = c - 1;
    const bool valid = i < NXI && e < XV && pos < PS && xx >= 0 && xx < W_;
    xst[j] = valid ? (((((cb * 8 + g) * 2 + t) * HW + (rr - 1) * W_ + xx) * 16 + W_ * 16) | rr) : 15;      // (+ one row: rr - 1 may be -1; taken off again below)
  }
#pragma unroll
  for (int j = 0; j < NDS; ++j) {
    const int pl = wave + 4 * j, g = pl >> 1, t = pl & 1;
    dst_[j] = (((ob * 8 + g) * 2 + t) * HW + lane) * 16;
  }
  auto load = [&](int u) {                                       // DMA of chunk u into this half's image, complete on return
    const int b = u / cpi, cidx = u - b * cpi, p0 = cidx * 64, y0 = p0 / W_;      // the chunk's 64 pixels: rows y0 .. y0 + R - 1 (W_ = 64: one row)
    // x patch: flat vector index e = 64 i + lane over [group][term][position]
    const int xsoff = b * Gin * 2 * HW * 16, dsoff = b * Gout * 2 * HW * 16;      // the image: scalar offsets (uniform)
    const int yrow = (y0 - 1) * W_ * 16, p0b = p0 * 16;
#pragma unroll
    for (int j = 0; j < NXS; ++j) {
      const int i = wave + 4 * j, rr = xst[j] & 15;
      const bool inb = rr != 15 && (unsigned)(y0 + rr - 1) < (unsigned)H;
      const int voff = inb ? (xst[j] & ~15) + yrow : (int)0x7FFFF000;
      if (i < NXI) lds_dma16(rx, xs + 64 * i, voff, xsoff);
    }
#pragma unroll
    for (int j = 0; j < NDS; ++j) lds_dma16(rd, ds + DSP * (wave + 4 * j), dst_[j] + p0b, dsoff);      // plane = (o group, term): 64 pixels = one instruction
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto multiply = [&]() {
#pragma unroll 1
    for (int ks = 0; ks < 2; ++ks) {
      const int srow = (32 * ks) / W_, scol = 32 * ks - srow * W_;     // the step's 32 pixels start at row srow of the chunk, column scol
      // A = dy: [o][k]; two transposing reads (4 pixels each) per term and 16-channel block
      uint4 av[4][2];
#pragma unroll
      for (int mo = 0; mo < 4; ++mo)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const uint4* base = abase + (mo * 4 + t) * DSP + 32 * ks;
          const uint2 lo = lds_tr16(base, boff), hi = lds_tr16(base + 4, boff);
          av[mo][t] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
      const uint4* bstep = bbase + srow * PC + scol;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - 3 * ky;
        uint4 bv[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          // (the second read is 4 pixels further along x: same row - 8-pixel runs never straddle a row, W_ % 16 == 0)
          const uint4* base = bstep + t * PSP + ky * PC + kx;
          const uint2 lo = lds_tr16(base, boff), hi = lds_tr16(base + 4, boff);
          bv[t] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
#pragma unroll
        for (int mo = 0; mo < 4; ++mo) {
          f32x4 c_ = acc[tap][mo];
          c_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av[mo][1]), __builtin_bit_cast(f16x8, bv[0]), c_, 0, 0, 0);
          c_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av[mo][0]), __builtin_bit_cast(f16x8, bv[1]), c_, 0, 0, 0);
          c_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av[mo][0]), __builtin_bit_cast(f16x8, bv[0]), c_, 0, 0, 0);
          acc[tap][mo] = c_;
        }
      }
    }
  };
  // step st: half A multiplies its chunk st while half B loads its chunk st; then half B multiplies while half A loads chunk
  // st + 1 (nA >= nB: the loop runs over half A's chunks; a half past its range idles through the barriers)
  if constexpr (FREE) {
    unsigned* hcnt = reinterpret_cast<unsigned*>(smem_raw + (size_t)2 * IMG * 16) + half * 8;      // one counter per half, 32 bytes apart
    if (tid == 0) *hcnt = 0u;
    __syncthreads();
    unsigned target = 0u;
    for (int st = 0; st < nmine; ++st) {
      load(u0 + st);                                              // (returns with this wave's pieces landed)
      half_barrier(hcnt, target, lane);                           // the half's image is complete
      multiply();
      half_barrier(hcnt, target, lane);                           // every wave of the half is past the image
    }
    __syncthreads();
  } else {
  if (half == 0 && nmine > 0) load(u0);
  __syncthreads();
  for (int st = 0; st < nA; ++st) {
    if (half == 0) multiply(); else if (st < nmine) load(u0 + st);
    __syncthreads();
    if (half == 1) { if (st < nmine) multiply(); } else if (st + 1 < nmine) load(u0 + st + 1);
    __syncthreads();
  }
  }
  // half B's accumulators into half A's, through LDS (both images are dead): 9 taps x 1024 floats per wave = 147 KB per half,
  // handed over in three rounds of three taps
  float* red = reinterpret_cast<float*>(smem_raw);               // [3 taps][4 waves][16 r][64 lanes]
  static_assert(3 * 4 * 16 * 64 * 4 <= 2 * IMG * 16, "three taps of partial sums fit the two images");
#pragma unroll
  for (int t0 = 0; t0 < 9; t0 += 3) {
    if (half == 1) {
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((t * 4 + wave) * 16 + r) * 64 + lane] = acc[t0 + t][r >> 2][r & 3];
    }
    __syncthreads();
    if (half == 0) {
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t0 + t][r >> 2][r & 3] += red[((t * 4 + wave) * 16 + r) * 64 + lane];
    }
    __syncthreads();
  }
  if (half == 1) return;
  // Slab in the ACCUMULATORS' own order, [split][tap][ob][cb][wo][wc][q = r / 4][lane] float4 = registers 4q .. 4q+3 of a lane (four
  // output channels of one input channel): 36 16-byte stores per lane, every wave-instruction 1 KB contiguous.  The [tap][o][ci]
  // order it replaces (round 3) took 144 dword stores per lane - a store-ISSUE-bound tail (MI355X_MICROARCH.md: dword / dwordx2
  // store tails run at a few bytes per clock per CU) that was most of the kernel's fixed ~23 us.  conv3x3_wgrad_reduce_tiled_kernel
  // reads the same order back with 16-byte loads and scatters only the final 9 * Cout * Cin values.
  float4* slp = reinterpret_cast<float4*>(a.slab) + (size_t)split * 9 * a.coutp * a.cinp / 4;
  // (accumulator block mo of wave w, lane group G: output channels 16 mo + 4 G .. + 3 of input channel 16 w + li - entered at the slab position
  // [wo][wc][q][lane] that a 32 x 32 accumulator layout gives that (o, ci) quad: the format conv3x3_wgrad_reduce_tiled_kernel reads)
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int mo = 0; mo < 4; ++mo) {
      const int wo = mo >> 1, wc = wave >> 1, qs = 2 * (mo & 1) + (G >> 1), ls = 32 * (G & 1) + 16 * (wave & 1) + li;
      const size_t f = ((((((size_t)tap * a.n_ob + ob) * a.n_cb + cb) * 2 + wo) * 2 + wc) * 4 + qs) * 64 + ls;
      const f32x4 c_ = acc[tap][mo];
      slp[f] = make_float4(ldexpf(c_[0], -ktot), ldexpf(c_[1], -ktot), ldexpf(c_[2], -ktot), ldexpf(c_[3], -ktot));
    }
}
// GR_WGRAD_PP: 1 (default) = the ping-pong kernel on 16-wide planes only, 2 = everywhere, 0 = never.  Measured at cfg2 per launch:
// 16-wide 52 -> 50 us, and the slab reduction 16.5 -> 10.8 us; 32-wide 70 -> 98 us (its 34 KB x patch per 64 pixels makes the
// load phase longer than the multiply phase, and strict alternation then idles the matrix pipe more than chance did).
static int wgrad_pp_mode() { static int v = -1; if (v < 0) { v = GR_KNOB("GR_WGRAD_PP", 1); } return v; }
// GR_WGRAD_FREE (ablation build): planes at least this wide take the free-running halves (default 32: the 32- and 64-wide layers); 0 = never (rounds 3-4)
static int wgrad_free_from() { static const int v = GR_KNOB("GR_WGRAD_FREE", 32); return v; }
static bool wgrad_free(int W) { return wgrad_free_from() > 0 && W >= wgrad_free_from() && wgrad_pp_mode() != 2; }
static bool wgrad_pp(int W) { const int m = wgrad_pp_mode(); return m == 2 || (m == 1 && W == 16) || wgrad_free(W); }
template <int W_, bool FREE_>
static void launch_wgrad_p16_pp_tf(const WgradP16Args& a, int grid, size_t lds, hipStream_t s) {
  static bool st = false;
  if (!st) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wgrad_p16_pp_kernel<W_, FREE_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds + 64); st = true; }
  hipLaunchKernelGGL((conv3x3_wgrad_p16_pp_kernel<W_, FREE_>), dim3(grid), dim3(512), lds + (FREE_ ? 64 : 0), s, a);
}
template <int W_>
static void launch_wgrad_p16_pp_t(const WgradP16Args& a, int grid, size_t lds, hipStream_t s) {
#ifdef GR_ABLATE      // every (width, variant) pair exists in the ablation build only; the shipping library instantiates what it launches: <16, false>, <32, true>, <64, true>
  if (wgrad_free(W_)) launch_wgrad_p16_pp_tf<W_, true>(a, grid, lds, s); else launch_wgrad_p16_pp_tf<W_, false>(a, grid, lds, s);
#else
  launch_wgrad_p16_pp_tf<W_, (W_ >= 32)>(a, grid, lds, s);
#endif
}

static bool wgrad_use_vec(int W) { return W >= 16 && W % 4 == 0; }
static bool wgrad_use_small(int Cin, int W) { return Cin <= 3 && W >= 16 && W % 4 == 0; }
static bool wgrad_use_bf16x6(int mode, int Cin, int W) { return mode >= 1 && Cin > 3 && W >= 16 && W % 8 == 0; }   // either split flavour
static void wgrad_geometry(int B, int Cin, int Cout, int H, int W, WgradArgs& a, int& TW, int mode = 0) {
  TW = W <= 8 ? 8 : (W <= 16 ? 16 : 32);
  const bool split = wgrad_use_bf16x6(mode, Cin, W);
  const int TR = (split ? 32 : 64) / TW;
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  a.tiles_x = (W + TW - 1) / TW; a.tiles_y = (H + TR - 1) / TR;
  a.tiles_total = (long)B * a.tiles_x * a.tiles_y;
  a.cinp = round_up(Cin, 64); a.coutp = round_up(Cout, 64);
  a.n_ob = a.coutp / 64; a.n_cb = a.cinp / 64;
  long want = (wgrad_use_vec(W) ? 256 : 512) / (a.n_ob * a.n_cb);   // vec kernel: one software-pipelined workgroup per CU
  if (wgrad_use_small(Cin, W)) want = 1024 / a.n_ob;
  // measured (B=256): 32-wide planes 512 splits (two workgroups per CU overlap convert/store with MFMAs), 16-wide planes 256
  // measured (f16x3): 512 partial blocks pay off on 32-wide planes once there are many row steps per block (cfg3: 1.75 vs 1.93 ms),
  // 256 otherwise (cfg2: the slab round trip of the extra blocks costs more than the overlap gains, -0.02 ms)
  else if (split) { want = GR_KNOB("GR_WGRAD_SPLITS", ((TW == 32 && (long)B * H * a.tiles_x >= 16384) ? 512 : 256)) / (a.n_ob * a.n_cb); }
  if (want < 1) want = 1;
  if (want > a.tiles_total) want = a.tiles_total;
  a.nsplit = (int)want;
}

size_t conv_wgrad_workspace_bytes(int B, int Cin, int Cout, int H, int W, int mode) {
  WgradArgs a{}; int TW;
  wgrad_geometry(B, Cin, Cout, H, W, a, TW, mode);
  if (wgrad_use_small(Cin, W)) return sizeof(float) * (size_t)a.nsplit * 32 * a.coutp;
  return sizeof(float) * (size_t)a.nsplit * 9 * a.coutp * a.cinp;
}

template <int NTERM>
static void launch_wgrad_split(const WgradArgs& a, int TW, int rps, int wv, int grid_, hipStream_t s) {
  if (rps > 0) {
    if (TW == 16) hipLaunchKernelGGL((conv3x3_wgrad_split_roll_kernel<16, NTERM>), dim3(grid_), dim3(256), 0, s, a, rps);
    else hipLaunchKernelGGL((conv3x3_wgrad_split_roll_kernel<32, NTERM>), dim3(grid_), dim3(256), 0, s, a, rps);
  } else if (TW == 16) {
    if (wv == 1) hipLaunchKernelGGL((conv3x3_wgrad_split_kernel<16, 1, NTERM>), dim3(grid_), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv3x3_wgrad_split_kernel<16, 2, NTERM>), dim3(grid_), dim3(256), 0, s, a);
  } else {
    if (wv == 1) hipLaunchKernelGGL((conv3x3_wgrad_split_kernel<32, 1, NTERM>), dim3(grid_), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv3x3_wgrad_split_kernel<32, 2, NTERM>), dim3(grid_), dim3(256), 0, s, a);
  }
}

bool conv_wgrad_p16_supported(int B, int Cin, int Cout, int H, int W) {
  static int on = -1;
  if (on < 0) { on = GR_KNOB_SET("GR_NO_P16_WGRAD") ? 0 : 1; }
  return on && Cin % 64 == 0 && Cout % 64 == 0 && (W == 16 || W == 32 || W == 64) && (H * W) % 256 == 0 &&
         (size_t)B * (Cin > Cout ? Cin : Cout) * H * W * 4 < 0x7FFFF000ul;
}
static int wgrad_p16_splits(int B, int Cin, int Cout, int H, int W) {
  const int blocks = (Cin / 64) * (Cout / 64);
  static int wgs_env = -1;
  if (wgs_env < 0) { wgs_env = GR_KNOB("GR_WGRAD_P16_WGS", 0); }
  const int wgs = wgs_env ? wgs_env : (wgrad_pp(W) ? 256 : 512);      // ping-pong: ONE eight-wave workgroup per CU; else two four-wave ones
  int want = wgs / blocks; if (want < 1) want = 1;
  const long units = (long)B * H * W / 64;
  if (want > units) want = (int)units;
  return want;
}
size_t conv_wgrad_p16_workspace_bytes(int B, int Cin, int Cout, int H, int W) {
  return sizeof(float) * (size_t)wgrad_p16_splits(B, Cin, Cout, H, W) * 9 * Cin * Cout;
}
template <int W_>
static void launch_wgrad_p16_t(const WgradP16Args& a, int grid, size_t lds, hipStream_t s) {
  static bool st = false;
  if (!st) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wgrad_p16_kernel<W_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); st = true; }
  hipLaunchKernelGGL(conv3x3_wgrad_p16_kernel<W_>, dim3(grid), dim3(256), lds, s, a);
}
void launch_conv3x3_wgrad_p16(const void* x_p16, const void* dy_p16, float* gw, void* workspace, int B, int Cin, int Cout, int H, int W,
                              hipStream_t s, const unsigned* amax_x, const unsigned* amax_dy) {
  WgradP16Args a{};
  a.x = reinterpret_cast<const uint4*>(x_p16); a.dy = reinterpret_cast<const uint4*>(dy_p16); a.slab = reinterpret_cast<float*>(workspace);
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.n_ob = Cout / 64; a.n_cb = Cin / 64; a.cinp = Cin; a.coutp = Cout;
  a.nsplit = wgrad_p16_splits(B, Cin, Cout, H, W); a.units = (int)((long)B * H * W / 64);
  a.amax_x = amax_x; a.amax_dy = amax_dy;
  { static const int plain = GR_KNOB("GR_WGRAD_PLAIN_ORDER", 0); a.plain_order = plain; }
  { static const int dynt = GR_KNOB("GR_WGRAD_DY_NT", 0); a.dy_nt = dynt; }
  const int grid = a.nsplit * a.n_ob * a.n_cb;
  const double px = (double)B * H * W;
  {
    const int R = 64 / W > 0 ? 64 / W : 1, PS = (R + 2) * (W + 2), PSP = PS + (10 - PS % 8) % 8;
    const size_t lds = 16 * (size_t)((16 * PSP + 63) / 64 * 64 + (16 * 66 + 63) / 64 * 64);
    const std::string nm = "conv3x3_wgrad_p16_kernel<" + std::to_string(W) + ">";
    const std::string nm2 = "conv3x3_wgrad_p16_pp_kernel<" + std::to_string(W) + (wgrad_free(W) ? ", true>" : ", false>");      // as rocprofv3 prints them (FREE = true: free-running halves)
    KtScope kt(wgrad_pp(W) ? nm2.c_str() : nm.c_str(), 2.0 * px * Cout * Cin * 9.0, 4.0 * (px * Cin + px * Cout + 9.0 * Cin * Cout), s);
    if (wgrad_pp(W)) {
      if (W == 16) launch_wgrad_p16_pp_t<16>(a, grid, 2 * lds, s); else if (W == 32) launch_wgrad_p16_pp_t<32>(a, grid, 2 * lds, s); else launch_wgrad_p16_pp_t<64>(a, grid, 2 * lds, s);
    }
#ifdef GR_ABLATE      // two four-wave workgroups per CU (rounds 2-4): an A/B control now
    else if (W == 16) launch_wgrad_p16_t<16>(a, grid, lds, s); else if (W == 32) launch_wgrad_p16_t<32>(a, grid, lds, s); else launch_wgrad_p16_t<64>(a, grid, lds, s);
#endif
  }
  const long n_ = (long)9 * Cout * a.cinp;
  KtScope kt("conv3x3_wgrad_reduce_tiled_kernel", (double)n_ * a.nsplit, 4.0 * n_ * (a.nsplit + 2.0), s);
  hipLaunchKernelGGL(conv3x3_wgrad_reduce_tiled_kernel, dim3((unsigned)((n_ / 4 + 31) / 32)), dim3(256), 0, s, reinterpret_cast<const float4*>(a.slab), gw,
                     Cin, Cout, a.n_ob, a.n_cb, a.nsplit);
#ifdef GR_ABLATE
  { static const int probe = GR_KNOB("GR_WGRAD_REDUCE_PROBE", 0);
    if (probe) {
      KtScope kt2("wgrad_reduce_one_wg_probe", (double)n_ * a.nsplit, 4.0 * n_ * a.nsplit, s);
      hipLaunchKernelGGL(conv3x3_wgrad_reduce_one_wg_probe_kernel, dim3(a.n_ob * a.n_cb), dim3(512), 0, s, reinterpret_cast<float4*>(a.slab), a.n_ob, a.n_cb, a.nsplit, n_ / 4);
    } }
#endif
}

void launch_conv3x3_wgrad(const float* x, const float* dy, float* gw, void* workspace,
                          int B, int Cin, int Cout, int H, int W, hipStream_t s, int mode,
                          const unsigned* amax_x, const unsigned* amax_dy) {
  WgradArgs a{}; int TW;
  wgrad_geometry(B, Cin, Cout, H, W, a, TW, mode);
  a.x = x; a.dy = dy; a.slab = reinterpret_cast<float*>(workspace);
  a.amax_x = amax_x; a.amax_dy = amax_dy;
  if (wgrad_use_bf16x6(mode, Cin, W) && !wgrad_use_small(Cin, W)) {
    const double px_ = (double)B * H * W;
    {
      const int grid_ = a.nsplit * a.n_ob * a.n_cb;
      static int wv = -1;
      if (wv < 0) { wv = GR_KNOB("GR_WGRAD_VARIANT", 2); }   // 2: rolling-window kernel; 1: per-tile kernel (AGPR accumulators)
      const int TRr = 32 / TW;
      int rps = 0;                                                 // rows per segment of the rolling-window kernel
      if (wv == 2 && H % TRr == 0) {
        // longest run of rows (a divisor of H, multiple of TR) that still yields >= grid_ segments
        for (int cand = H; cand >= TRr; --cand)
          if (H % cand == 0 && cand % TRr == 0) {
            const long nsegs = (long)B * (H / cand) * a.tiles_x;         // balanced: a multiple of the splits, or many per split
            if (nsegs >= a.nsplit && (nsegs % a.nsplit == 0 || nsegs >= 4L * a.nsplit || cand == TRr)) { rps = cand; break; }
          }
      }
      const std::string nt_ = mode == 2 ? "2>" : "3>";      // as rocprofv3 prints them: last template argument = number of split terms
      const std::string nm_ = rps > 0 ? std::string("conv3x3_wgrad_split_roll_kernel<") + (TW == 16 ? "16, " : "32, ") + nt_
                                      : std::string("conv3x3_wgrad_split_kernel<") + (TW == 16 ? "16, " : "32, ") + (wv == 1 ? "1, " : "2, ") + nt_;
      KtScope kt(nm_.c_str(), 2.0 * px_ * Cout * Cin * 9.0, 4.0 * (px_ * Cin + px_ * Cout + 9.0 * Cin * Cout), s);
      if (mode == 2) launch_wgrad_split<2>(a, TW, rps, wv, grid_, s); else launch_wgrad_split<3>(a, TW, rps, wv, grid_, s);
    }
    const long n_ = (long)9 * Cout * a.cinp;
    KtScope kt("conv3x3_wgrad_reduce8_kernel", (double)n_ * a.nsplit, 4.0 * n_ * (a.nsplit + 2.0), s);
    hipLaunchKernelGGL(conv3x3_wgrad_reduce8_kernel, dim3((unsigned)((n_ + 31) / 32)), dim3(256), 0, s,
                       a.slab, gw, Cin, Cout, a.cinp, a.coutp, a.nsplit);
    return;
  }
  if (wgrad_use_small(Cin, W)) {
    const double px_ = (double)B * H * W;
    {
      KtScope kt(TW == 16 ? "conv3x3_wgrad_small_kernel<16>" : "conv3x3_wgrad_small_kernel<32>", 2.0 * px_ * Cout * Cin * 9.0,
                 4.0 * (px_ * Cin + px_ * Cout + 9.0 * Cin * Cout), s);
      const int grid_ = a.nsplit * a.n_ob;
      if (TW == 16) hipLaunchKernelGGL(conv3x3_wgrad_small_kernel<16>, dim3(grid_), dim3(256), 0, s, a);
      else hipLaunchKernelGGL(conv3x3_wgrad_small_kernel<32>, dim3(grid_), dim3(256), 0, s, a);
    }
    const int n_ = 9 * Cin * Cout;
    KtScope kt("conv3x3_wgrad_small_reduce_kernel", (double)n_ * a.nsplit, 4.0 * n_ * (a.nsplit + 2.0), s);
    hipLaunchKernelGGL(conv3x3_wgrad_small_reduce_kernel, dim3((n_ + 15) / 16), dim3(256), 0, s, a.slab, gw, Cin, Cout, a.coutp, a.nsplit);
    return;
  }
  const int grid = a.nsplit * a.n_ob * a.n_cb;
  const double px = (double)B * H * W;
  {
    const char* name = wgrad_use_vec(W) ? (TW == 16 ? "conv3x3_wgrad_vec_kernel<16>" : "conv3x3_wgrad_vec_kernel<32>")
                       : (TW == 8 ? "conv3x3_wgrad_kernel<8>" : (TW == 16 ? "conv3x3_wgrad_kernel<16>" : "conv3x3_wgrad_kernel<32>"));
    KtScope kt(name, 2.0 * px * Cout * Cin * 9.0, 4.0 * (px * Cin + px * Cout + 9.0 * Cin * Cout), s);
    if (wgrad_use_vec(W)) {
      if (TW == 16) hipLaunchKernelGGL(conv3x3_wgrad_vec_kernel<16>, dim3(grid), dim3(256), 0, s, a);
      else hipLaunchKernelGGL(conv3x3_wgrad_vec_kernel<32>, dim3(grid), dim3(256), 0, s, a);
    } else if (TW == 8) hipLaunchKernelGGL(conv3x3_wgrad_kernel<8>, dim3(grid), dim3(256), 0, s, a);
    else if (TW == 16) hipLaunchKernelGGL(conv3x3_wgrad_kernel<16>, dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(conv3x3_wgrad_kernel<32>, dim3(grid), dim3(256), 0, s, a);
  }
  const long n = (long)9 * Cout * a.cinp;
  KtScope kt("conv3x3_wgrad_reduce8_kernel", (double)n * a.nsplit, 4.0 * n * (a.nsplit + 2.0), s);
  hipLaunchKernelGGL(conv3x3_wgrad_reduce8_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, s,
                     a.slab, gw, Cin, Cout, a.cinp, a.coutp, a.nsplit);
}

}  // namespace gr

// elem.hip — HBM-bound per-channel pipelines around the convolutions, the criterion and the optimiser.
// Compiled with -ffp-contract=off: every fp32 operation rounds where the Torch7 tensor op it replaces rounds,
// so given identical inputs these kernels match the oracle bit for bit except through expf/tanhf.
//
// Replaces (reference file:line):
//   nn.SpatialBatchNormalization / nn.BatchNormalization   models.lua:116,123,129,410..437,448
//   nn.ELU / cudnn.ReLU / nn.Sigmoid / nn.Tanh / nn.LeakyReLU  models.lua:411,117,133,453,18
//   nn.Dropout / nn.SpatialDropout / nn.SpatialMaxPooling  models.lua:402-405,412,439,422,440
//   nn.MSECriterion                                        train_r.lua:119,147,150
//   fevalR penalty+clamp and optim.adam                    train_r.lua:153-165,170
#include "kernels.h"
#include <type_traits>

namespace gr {

// ------------------------------------------------------------------ helpers
__device__ __forceinline__ double block_reduce_sum(double v, double* sh) {
  // deterministic tree: wave shuffle then fixed-order sum over waves
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double r = 0;
  if (threadIdx.x == 0) for (int w = 0; w < nw; ++w) r += sh[w];
  return r;  // valid on thread 0
}

// x / d for the plane-size divisors of the pipeline kernels (d > 0, uniform): a shift when d is a power of two - every shape of
// the path - instead of the ~20 VALU instructions of a 32-bit division per float4 (these kernels are VALU-bound, ~37 per element)
__device__ __forceinline__ unsigned udivp(unsigned x, unsigned d) { return (d & (d - 1)) == 0 ? x >> (__ffs(d) - 1) : x / d; }
__device__ __forceinline__ float post_slope(const PostArgs& a) { return a.act == ACT_PRELU ? *a.slope_dev : a.slope; }
// The pipeline kernels read their stage description (activation, mask kinds, pooling, BatchNorm) from the argument block and
// branch on it per element - uniform branches, but ~25 of them per float4 and every taken one a fetch bubble.  The three
// descriptions R's training step consists of (models.lua:409-440: conv-SBN-ELU-Dropout; ...-ELU-MaxPool-Dropout;
// ...-ELU-SpatialDropout-MaxPool) are instantiated with those fields as compile-time constants: same code, same arithmetic,
// the switches folded.  CB = 0 is the generic kernel.
template <int CB>
__device__ __forceinline__ void post_specialize(PostArgs& f) {
  if (CB == 1) { f.act = ACT_ELU; f.has_bn = 1; f.m1.kind = MASK_ELEM; f.pool = 0; f.m2.kind = MASK_NONE; }
  if (CB == 2) { f.act = ACT_ELU; f.has_bn = 1; f.m1.kind = MASK_NONE; f.pool = 1; f.m2.kind = MASK_ELEM; }
  if (CB == 3) { f.act = ACT_ELU; f.has_bn = 1; f.m1.kind = MASK_SPATIAL; f.pool = 1; f.m2.kind = MASK_NONE; }
  // the D network's stages (models.lua:272-337: no BatchNorm; a PReLU closes its stage, dropout / pooling follow element-wise)
  if (CB == 4) { f.act = ACT_PRELU; f.has_bn = 0; f.m1.kind = MASK_NONE; f.pool = 0; f.m2.kind = MASK_NONE; }
  if (CB == 5) { f.act = ACT_NONE; f.has_bn = 0; f.m1.kind = MASK_SPATIAL; f.pool = 1; f.m2.kind = MASK_NONE; }
  if (CB == 6) { f.act = ACT_NONE; f.has_bn = 0; f.m1.kind = MASK_NONE; f.pool = 1; f.m2.kind = MASK_NONE; }
  if (CB == 7) { f.act = ACT_NONE; f.has_bn = 0; f.m1.kind = MASK_SPATIAL; f.pool = 0; f.m2.kind = MASK_NONE; }
  // G in training mode (the GAN game trains it: models.lua:115-133): Linear/conv - BatchNorm - ReLU, and the Sigmoid of its last layer
  if (CB == 8) { f.act = ACT_RELU; f.has_bn = 1; f.m1.kind = MASK_NONE; f.pool = 0; f.m2.kind = MASK_NONE; }
  if (CB == 9) { f.act = ACT_SIGMOID; f.has_bn = 0; f.m1.kind = MASK_NONE; f.pool = 0; f.m2.kind = MASK_NONE; }
  // R in evaluate() mode (apply_r.lua's embedding): the two pooling stages (the other four ride in their convolutions' epilogues);
  // Dropout is the identity there, SpatialDropout a multiplication by 1 - p
  if (CB == 10) { f.act = ACT_ELU; f.has_bn = 1; f.m1.kind = MASK_NONE; f.pool = 1; f.m2.kind = MASK_NONE; }
  if (CB == 11) { f.act = ACT_ELU; f.has_bn = 1; f.m1.kind = MASK_SCALE; f.pool = 1; f.m2.kind = MASK_NONE; }
}
inline int post_combo(const PostArgs& f) {
  static const bool on = !GR_KNOB_SET("GR_POST_GENERIC");
  if (!on) return 0;
  if (!f.has_bn && f.m2.kind == MASK_NONE) {
    if (f.act == ACT_PRELU && f.m1.kind == MASK_NONE && !f.pool) return 4;
    if (f.act == ACT_NONE && f.m1.kind == MASK_SPATIAL && f.pool) return 5;
    if (f.act == ACT_NONE && f.m1.kind == MASK_NONE && f.pool) return 6;
    if (f.act == ACT_NONE && f.m1.kind == MASK_SPATIAL && !f.pool) return 7;
    if (f.act == ACT_SIGMOID && f.m1.kind == MASK_NONE && !f.pool) return 9;
  }
  if (f.act == ACT_RELU && f.has_bn && f.m1.kind == MASK_NONE && !f.pool && f.m2.kind == MASK_NONE) return 8;
  if (f.act != ACT_ELU || !f.has_bn) return 0;
  if (f.m1.kind == MASK_ELEM && !f.pool && f.m2.kind == MASK_NONE) return 1;
  if (f.m1.kind == MASK_NONE && f.pool && f.m2.kind == MASK_ELEM) return 2;
  if (f.m1.kind == MASK_SPATIAL && f.pool && f.m2.kind == MASK_NONE) return 3;
  if (f.m1.kind == MASK_NONE && f.pool && f.m2.kind == MASK_NONE) return 10;
  if (f.m1.kind == MASK_SCALE && f.pool && f.m2.kind == MASK_NONE) return 11;
  return 0;
}
template <typename F>
static void with_combo(int cb, F&& f) {          // f(std::integral_constant<int, CB>)
  switch (cb) {
    case 1: f(std::integral_constant<int, 1>{}); break;
    case 2: f(std::integral_constant<int, 2>{}); break;
    case 3: f(std::integral_constant<int, 3>{}); break;
    case 4: f(std::integral_constant<int, 4>{}); break;
    case 5: f(std::integral_constant<int, 5>{}); break;
    case 6: f(std::integral_constant<int, 6>{}); break;
    case 7: f(std::integral_constant<int, 7>{}); break;
    case 8: f(std::integral_constant<int, 8>{}); break;
    case 9: f(std::integral_constant<int, 9>{}); break;
    case 10: f(std::integral_constant<int, 10>{}); break;
    case 11: f(std::integral_constant<int, 11>{}); break;
    default: f(std::integral_constant<int, 0>{});
  }
}
__device__ __forceinline__ float act_fwd(float z, int act, float slope) {
  switch (act) {
    // ELU on the hardware exponential (v_exp_f32 of z * log2 e: 2 instructions instead of expf's ~15; these pipelines are
    // VALU-bound, not HBM-bound: ~35 instructions per element before).  For z <= 0 the result's absolute error stays below
    // 3e-7 (measured against expf over [-30, 0]): two orders below the 1e-4 bar, and expf was never bit-identical to the
    // host libm anyway.
    case ACT_ELU: return z <= 0.f ? (__expf(z) - 1.f) * 1.f : z;
    case ACT_RELU: return z > 0.f ? z : 0.f;
    case ACT_LEAKYRELU: case ACT_PRELU: return z > 0.f ? z : z * slope;
    case ACT_SIGMOID: return 1.f / (1.f + expf(-z));
    case ACT_TANH: return tanhf(z);
    default: return z;
  }
}
__device__ __forceinline__ float act_bwd(float g, float z, float a, int act, float slope) {
  switch (act) {
    case ACT_ELU: return a <= 0.f ? g * (a + 1.f) : g;
    case ACT_RELU: return a > 0.f ? g : 0.f;
    case ACT_LEAKYRELU: case ACT_PRELU: return z > 0.f ? g : g * slope;
    case ACT_SIGMOID: return g * (1.f - a) * a;
    case ACT_TANH: return g * (1.f - a * a);
    default: return g;
  }
}
// dz of one element from gradOutput g and the activation's INPUT z, the forward recomputed: act_bwd(g, z, act_fwd(z)).  ELU is
// spelled out: for z <= 0 the forward value a = e^z - 1 is <= 0 as well, so the two selects of the composition (z <= 0 in the
// forward, a <= 0 in the backward) are one - same operations on the same operands, bit-identical result, one compare and one
// select less per element in kernels that are VALU-bound.
__device__ __forceinline__ float act_bwd_z(float g, float z, int act, float slope) {
  if (act == ACT_ELU) return z <= 0.f ? g * (((__expf(z) - 1.f) * 1.f) + 1.f) : g;
  return act_bwd(g, z, act_fwd(z, act, slope), act, slope);
}
__device__ __forceinline__ float mask_mul(const MaskRef& m, long e, long bc) {
  switch (m.kind) {
    case MASK_ELEM: return ((m.bits[e >> 5] >> (e & 31)) & 1u) ? m.scale : 0.f;
    case MASK_SPATIAL: return ((m.bits[bc >> 5] >> (bc & 31)) & 1u) ? m.scale : 0.f;
    case MASK_SCALE: return m.scale;
    default: return 1.f;
  }
}
__device__ __forceinline__ float bn_apply(const PostArgs& a, float y, int c) {
  return a.has_bn ? ((y - a.mean[c]) * a.invstd[c]) * a.gamma[c] + a.beta[c] : y;
}

// ------------------------------------------------------------------ forward pipeline: BN -> act -> mask1 -> [pool] -> mask2
__global__ __launch_bounds__(256) void post_forward_kernel(PostArgs a) {
  const int H = a.H, W = a.W, Ho = a.pool ? H >> 1 : H, Wo = a.pool ? W >> 1 : W;
  const long HW = (long)H * W, HWo = (long)Ho * Wo;
  const long n = (long)a.B * a.C * HWo;
  float omax = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long bc = i / HWo; const int po = (int)(i - bc * HWo);
    const int c = (int)(bc % a.C);
    float r;
    if (a.pool) {
      const int yo = po / Wo, xo = po - yo * Wo;
      float best = -INFINITY; int bi = 0;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const long e = bc * HW + (long)(2 * yo + (t >> 1)) * W + 2 * xo + (t & 1);
        const float v = act_fwd(bn_apply(a, a.y[e], c), a.act, post_slope(a)) * mask_mul(a.m1, e, bc);
        if (v > best) { best = v; bi = t; }
      }
      a.pool_idx[i] = (uint8_t)bi;
      r = best;
    } else {
      r = act_fwd(bn_apply(a, a.y[i], c), a.act, post_slope(a)) * mask_mul(a.m1, i, bc);
    }
    const float res = r * mask_mul(a.m2, i, bc);
    a.out[i] = res;
    omax = fmaxf(omax, fabsf(res));
  }
  if (a.amax_out) absmax_commit(omax, a.amax_out);
}

// float4 variant: W % 4 == 0 (W % 8 == 0 with the pool).  One thread per 4 consecutive OUTPUT elements of a row; plane /
// channel come from 32-bit divisions once per thread (the scalar kernel above pays 64-bit divisions per element).
__device__ __forceinline__ float4 mask4(const MaskRef& m, unsigned e, unsigned bc) {
  float4 r = make_float4(1.f, 1.f, 1.f, 1.f);
  if (m.kind == MASK_ELEM) {
    const uint32_t w = m.bits[e >> 5] >> (e & 31);     // e % 4 == 0: the 4 bits never straddle a word
    r.x = (w & 1u) ? m.scale : 0.f; r.y = (w & 2u) ? m.scale : 0.f; r.z = (w & 4u) ? m.scale : 0.f; r.w = (w & 8u) ? m.scale : 0.f;
  } else if (m.kind == MASK_SPATIAL) {
    const float v = ((m.bits[bc >> 5] >> (bc & 31)) & 1u) ? m.scale : 0.f;
    r = make_float4(v, v, v, v);
  } else if (m.kind == MASK_SCALE) {
    r = make_float4(m.scale, m.scale, m.scale, m.scale);
  }
  return r;
}
__device__ __forceinline__ uint32_t mask_word(const MaskRef& m, unsigned e, unsigned bc) {      // the mask bits of elements e.. (e % 4 == 0), bit 0 first
  if (m.kind == MASK_ELEM) return m.bits[e >> 5] >> (e & 31);
  if (m.kind == MASK_SPATIAL) return ((m.bits[bc >> 5] >> (bc & 31)) & 1u) ? 0xFu : 0u;
  return 0xFu;
}
__device__ __forceinline__ float4 mask4_of(const MaskRef& m, uint32_t w) {
  if (m.kind == MASK_NONE) return make_float4(1.f, 1.f, 1.f, 1.f);
  if (m.kind == MASK_SCALE) return make_float4(m.scale, m.scale, m.scale, m.scale);
  return make_float4((w & 1u) ? m.scale : 0.f, (w & 2u) ? m.scale : 0.f, (w & 4u) ? m.scale : 0.f, (w & 8u) ? m.scale : 0.f);
}
__device__ __forceinline__ float4 bn_act4(const PostArgs& a, float4 v, float mean, float invstd, float g, float bt) {
  if (a.has_bn) {
    v.x = ((v.x - mean) * invstd) * g + bt; v.y = ((v.y - mean) * invstd) * g + bt;
    v.z = ((v.z - mean) * invstd) * g + bt; v.w = ((v.w - mean) * invstd) * g + bt;
  }
  v.x = act_fwd(v.x, a.act, post_slope(a)); v.y = act_fwd(v.y, a.act, post_slope(a));
  v.z = act_fwd(v.z, a.act, post_slope(a)); v.w = act_fwd(v.w, a.act, post_slope(a));
  return v;
}
__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
// Streaming loads of the pipeline kernels.  y and gradOutput are read once per pass; when the tensor is far larger than the
// Infinity Cache (256 MB) nothing of it survives until the next pass anyway, and a non-temporal load keeps it from evicting
// what does get reused.  Measured in the real step (round 3, same box, interleaved): cfg3 (268 / 537 MB tensors) pass A 1.02-1.04 ->
// 0.955-0.964 ms, pass B 1.105 -> 0.991 ms; cfg2 (34 / 67 MB tensors, which pass B finds in the cache) pass B 0.168 -> 0.182 ms
// (slower, and mostly because a non-temporal pass A no longer leaves the tensors in the cache for it).  So: passes A and B above 128 MB
// only (launchers; GR_POST_NT overrides: bit 0 pass A, 1 pass B, 2 forward).
typedef float nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_maybe_nt(const float* p, bool nt) {
  if (nt) { const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
  return *reinterpret_cast<const float4*>(p);
}
inline int post_nt_mode() { static const int m = GR_KNOB("GR_POST_NT", -1); return m; }   // -1: the size rule; else bit 0 pass A, 1 pass B, 2 forward
inline bool post_big(const PostArgs& f) { return 4.0 * f.B * f.C * f.H * f.W >= 128.0 * 1024 * 1024; }

template <int CB>
__global__ __launch_bounds__(256) void post_forward_vec_kernel(PostArgs a) {
  post_specialize<CB>(a);
  const unsigned H = a.H, W = a.W, Ho = a.pool ? H >> 1 : H, Wo = a.pool ? W >> 1 : W;
  const unsigned HW = H * W, HWo = Ho * Wo, q_per_plane = HWo >> 2, wq = Wo >> 2;
  const unsigned n4 = (unsigned)a.B * a.C * q_per_plane;
  float omax = 0.f;
  for (unsigned i4 = blockIdx.x * blockDim.x + threadIdx.x; i4 < n4; i4 += gridDim.x * blockDim.x) {
    const unsigned bc = udivp(i4, q_per_plane), within = i4 - bc * q_per_plane, c = bc - udivp(bc, (unsigned)a.C) * (unsigned)a.C;
    float mean = 0.f, invstd = 1.f, g = 1.f, bt = 0.f;
    if (a.has_bn) { mean = a.mean[c]; invstd = a.invstd[c]; g = a.gamma[c]; bt = a.beta[c]; }
    const unsigned eo = bc * HWo + within * 4;
    float4 r;
    if (a.pool) {
      const unsigned yo = udivp(within, wq), xo = (within - yo * wq) * 4;
      const unsigned e0 = bc * HW + (2 * yo) * W + 2 * xo, e1 = e0 + W;
      const float4 t0 = mul4(bn_act4(a, ld4_maybe_nt(a.y + e0, a.nt != 0), mean, invstd, g, bt), mask4(a.m1, e0, bc));
      const float4 t1 = mul4(bn_act4(a, ld4_maybe_nt(a.y + e0 + 4, a.nt != 0), mean, invstd, g, bt), mask4(a.m1, e0 + 4, bc));
      const float4 b0 = mul4(bn_act4(a, ld4_maybe_nt(a.y + e1, a.nt != 0), mean, invstd, g, bt), mask4(a.m1, e1, bc));
      const float4 b1 = mul4(bn_act4(a, ld4_maybe_nt(a.y + e1 + 4, a.nt != 0), mean, invstd, g, bt), mask4(a.m1, e1 + 4, bc));
      const float top[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
      const float bot[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      float o[4]; uint32_t idx = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float best = -INFINITY; uint32_t bi = 0;      // scan order (0,0) (0,1) (1,0) (1,1); first strictly greater wins
        if (top[2 * k] > best) { best = top[2 * k]; bi = 0; }
        if (top[2 * k + 1] > best) { best = top[2 * k + 1]; bi = 1; }
        if (bot[2 * k] > best) { best = bot[2 * k]; bi = 2; }
        if (bot[2 * k + 1] > best) { best = bot[2 * k + 1]; bi = 3; }
        o[k] = best; idx |= bi << (8 * k);
      }
      *reinterpret_cast<uint32_t*>(a.pool_idx + eo) = idx;
      r = make_float4(o[0], o[1], o[2], o[3]);
    } else {
      r = mul4(bn_act4(a, ld4_maybe_nt(a.y + eo, a.nt != 0), mean, invstd, g, bt), mask4(a.m1, eo, bc));
    }
    const float4 res = mul4(r, mask4(a.m2, eo, bc));
    *reinterpret_cast<float4*>(a.out + eo) = res;
    omax = absmax4(omax, res);
  }
  if (a.amax_out) absmax_commit(omax, a.amax_out);
}

// Operand-ready variant: one thread = 4 consecutive OUTPUT pixels of a row x the 8 channels of group g = c / 8.  Besides the
// fp32 tensor it writes the consumer's image p16[b][g][term][pixel] (16 bytes = the 8 channels' fp16 halves of one term),
// scaled by the power of two that the consumer's slot - an upper bound of max|out| fixed before this launch - defines: a wave's
// stores per term are 4 KB contiguous.  Same arithmetic per element as post_forward_vec_kernel.
bool post_g8_supported(int C, int H, int W, bool pool, bool backward) {   // (T8_PXT is defined further down)
  constexpr int T8_PXT = 1024;
  const int Ho = pool ? H >> 1 : H, Wo = pool ? W >> 1 : W;
  if (C % 8 != 0 || !(pool ? (W % 8 == 0 && H % 2 == 0) : (W % 4 == 0)) || H * W < 64) return false;
  const int n = backward ? H * W : Ho * Wo;                   // pixels of the tensor that is written operand-ready
  return n % 256 == 0 && (n <= T8_PXT || n % T8_PXT == 0);     // whole tiles of at most T8_PXT pixels; a wave's tasks of one step share a channel
}
// ---- operand-ready image through an LDS transpose
// A pixel's vector (8 channels) is assembled from eight channel planes.  Threads stay as light as in the float4 kernels (one
// channel x 4 consecutive pixels: full-line loads, ~60 registers, 8 waves per SIMD): each packs the fp16 hi / lo halves of
// its 4 pixels (8 bytes per term) into an LDS image [term][channel][pixel]; after a barrier the image is read back with the
// TRANSPOSING read ds_read_b64_tr_b16 (16 lanes: 4 channel rows x 16 pixels -> lane i holds pixel i's 4 channels; two reads =
// the 8 channels) and every lane stores one 16-byte vector: 64 lanes write 1 KB contiguous.  Row stride = 64 (mod 256) bytes:
// the four rows of a read fall on different bank groups.  (The first version gave a thread 8 channels x 4 pixels: 209
// registers, 2 waves per SIMD, 2.0-2.6 TB/s against 4.4 for the float4 kernels.)
constexpr int T8_PXT = 1024;                                   // pixels per tile (a 32x32 plane; 16 rows of a 64-wide one)
constexpr int T8_RSB = T8_PXT * 2 + 64;                        // bytes per (term, channel) row
typedef short s16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 t8_tr(const unsigned char* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p));
#else
  (void)p; return make_uint2(0, 0);
#endif
}
// 4 scaled values of one channel -> packed hi halves, packed lo halves (same roundings as split8_f16)
__device__ __forceinline__ void t8_pack(float4 v, float sc, uint2& hi, uint2& lo) {
  const float x[4] = {v.x * sc, v.y * sc, v.z * sc, v.w * sc};
  unsigned short a[4], b[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const _Float16 h0 = (_Float16)x[j]; const float r = x[j] - (float)h0; const _Float16 h1 = (_Float16)r;
    a[j] = __builtin_bit_cast(unsigned short, h0); b[j] = __builtin_bit_cast(unsigned short, h1);
  }
  hi = make_uint2(a[0] | (unsigned)a[1] << 16, a[2] | (unsigned)a[3] << 16);
  lo = make_uint2(b[0] | (unsigned)b[1] << 16, b[2] | (unsigned)b[3] << 16);
}
// phase 2: the tile's npx pixels x 2 terms, one vector per lane and step; dst = the image's term-0 plane at the tile's first pixel
template <int RSB>
__device__ __forceinline__ void t8_emit(const unsigned char* img, int npx, uint4* dst, size_t term_stride, bool nt = false) {
  const int lane16 = threadIdx.x & 15, q = lane16 >> 2, pq = lane16 & 3;
  for (int v = threadIdx.x; v < 2 * npx; v += 256) {
    const int t = v / npx, px = v - t * npx, c0 = px & ~15;              // the 16-lane group's block of pixels c0 .. c0 + 15 (t is wave-uniform)
    const unsigned char* row = img + (size_t)(t * 8 + q) * RSB + (c0 + 4 * pq) * 2;
    const uint2 lo4 = t8_tr(row), hi4 = t8_tr(row + 4 * RSB);         // channels 0-3 and 4-7 of pixel px
    store4(dst + (size_t)t * term_stride + px, make_uint4(lo4.x, lo4.y, hi4.x, hi4.y), nt);
  }
}
// PXT = pixels per tile: 1024, or 256 for planes of 256 output pixels (a 34 KB image for 1024 pixels limits a CU to four
// workgroups; 16x16 planes need 9 KB and fit eight)
template <bool POOL, int PXT, int CB>
__global__ __launch_bounds__(256) void post_forward_g8_kernel(PostArgs a) {
  post_specialize<CB>(a);
  constexpr int RSB = PXT * 2 + 64;                          // bytes per (term, channel) row: 64 (mod 256)
  __shared__ __attribute__((aligned(16))) unsigned char img[16 * RSB];
  const unsigned H = a.H, W = a.W, Ho = POOL ? H >> 1 : H, Wo = POOL ? W >> 1 : W;
  const unsigned HW = H * W, HWo = Ho * Wo, wq = Wo >> 2, G = (unsigned)a.C >> 3;
  const unsigned npx = HWo < (unsigned)PXT ? HWo : (unsigned)PXT, tiles = HWo / npx, qpt = npx >> 2;   // quads per tile
  const unsigned units = (unsigned)a.B * G * tiles;
  const float sc = pow2f(f16_scale_exp(absmax_read(a.p16_scale)));
  uint4* p16 = reinterpret_cast<uint4*>(a.p16);
  for (unsigned u = blockIdx.x; u < units; u += gridDim.x) {
    const unsigned bg = u / tiles, tile = u - bg * tiles, b = bg / G, g = bg - b * G;
    for (unsigned task = threadIdx.x; task < 8 * qpt; task += 256) {
      const unsigned j = udivp(task, qpt), within = tile * qpt + (task - j * qpt);       // channel 8g + j, quad `within` of the plane
      const unsigned c = 8 * g + j, bc = b * (unsigned)a.C + c;
      float mean = 0.f, invstd = 1.f, gm = 1.f, bt = 0.f;
      if (a.has_bn) { mean = a.mean[c]; invstd = a.invstd[c]; gm = a.gamma[c]; bt = a.beta[c]; }
      const unsigned eo = bc * HWo + within * 4;
      float4 r;
      if constexpr (POOL) {
        const unsigned yo = udivp(within, wq), xo = (within - yo * wq) * 4;
        const unsigned e0 = bc * HW + (2 * yo) * W + 2 * xo, e1 = e0 + W;
        const float4 t0 = mul4(bn_act4(a, ld4_maybe_nt(a.y + e0, a.nt != 0), mean, invstd, gm, bt), mask4(a.m1, e0, bc));
        const float4 t1 = mul4(bn_act4(a, ld4_maybe_nt(a.y + e0 + 4, a.nt != 0), mean, invstd, gm, bt), mask4(a.m1, e0 + 4, bc));
        const float4 b0 = mul4(bn_act4(a, ld4_maybe_nt(a.y + e1, a.nt != 0), mean, invstd, gm, bt), mask4(a.m1, e1, bc));
        const float4 b1 = mul4(bn_act4(a, ld4_maybe_nt(a.y + e1 + 4, a.nt != 0), mean, invstd, gm, bt), mask4(a.m1, e1 + 4, bc));
        const float top[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
        const float bot[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        float o[4]; uint32_t idx = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float best = -INFINITY; uint32_t bi = 0;      // scan order (0,0) (0,1) (1,0) (1,1); first strictly greater wins
          if (top[2 * k] > best) { best = top[2 * k]; bi = 0; }
          if (top[2 * k + 1] > best) { best = top[2 * k + 1]; bi = 1; }
          if (bot[2 * k] > best) { best = bot[2 * k]; bi = 2; }
          if (bot[2 * k + 1] > best) { best = bot[2 * k + 1]; bi = 3; }
          o[k] = best; idx |= bi << (8 * k);
        }
        *reinterpret_cast<uint32_t*>(a.pool_idx + eo) = idx;
        r = make_float4(o[0], o[1], o[2], o[3]);
      } else {
        r = mul4(bn_act4(a, ld4_maybe_nt(a.y + eo, a.nt != 0), mean, invstd, gm, bt), mask4(a.m1, eo, bc));
      }
      const float4 res = mul4(r, mask4(a.m2, eo, bc));
      if (a.out) store4(a.out + eo, res, a.nt_st != 0);                  // (null: every consumer takes the operand-ready image)
      uint2 hi, lo;
      t8_pack(res, sc, hi, lo);
      const unsigned qi = task - j * qpt;
      *reinterpret_cast<uint2*>(img + (size_t)j * RSB + qi * 8) = hi;
      *reinterpret_cast<uint2*>(img + (size_t)(8 + j) * RSB + qi * 8) = lo;
    }
    __syncthreads();
    t8_emit<RSB>(img, (int)npx, p16 + (size_t)bg * 2 * HWo + (size_t)tile * npx, HWo, a.nt_st != 0);
    __syncthreads();
  }
}

// fp32 NCHW -> operand-ready image, scaled by the slot's power of two (stand-alone entry points, micro-benchmarks; inside a
// net the pipeline kernels write the image directly)
__global__ __launch_bounds__(256) void to_p16_kernel(const float* __restrict__ x, uint4* __restrict__ p16, int B, int C, int HW, const unsigned* __restrict__ slot) {
  const float sc = pow2f(f16_scale_exp(absmax_read(slot)));
  const unsigned qpp = (unsigned)HW >> 2, G = (unsigned)C >> 3, n4 = (unsigned)B * G * qpp;
  for (unsigned i4 = blockIdx.x * blockDim.x + threadIdx.x; i4 < n4; i4 += gridDim.x * blockDim.x) {
    const unsigned bg = i4 / qpp, within = i4 - bg * qpp;
    float vals[8][4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)bg * 8 + j) * HW + within * 4);
      vals[j][0] = v.x; vals[j][1] = v.y; vals[j][2] = v.z; vals[j][3] = v.w;
    }
    uint4* dst = p16 + (size_t)bg * 2 * HW + within * 4;
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      const float x8[8] = {vals[0][px], vals[1][px], vals[2][px], vals[3][px], vals[4][px], vals[5][px], vals[6][px], vals[7][px]};
      uint4 t0, t1;
      split8_f16(x8, sc, t0, t1);
      dst[px] = t0; dst[HW + px] = t1;
    }
  }
}
// LDS tile of the 8-channel-group pipeline kernels: 2 = 256 pixels (9 KB per workgroup, eight workgroups per CU; default), 1 = 512,
// 0 = 1024 (34 KB, four per CU).  Measured per step, pass B / forward: cfg2 0.228 / 0.148 -> 0.207 / 0.135 -> 0.199 / 0.139 ms,
// cfg3 1.74 / 1.09 -> 1.23 / 0.80 -> 1.21 / 0.81 ms.
static int g8_half_tiles() { static int v = -1; if (v < 0) { v = GR_KNOB("GR_G8_HALF_TILES", 2); } return v; }
void launch_to_p16(const float* x, void* p16, int B, int C, int HW, const unsigned* slot, hipStream_t s) {
  long blocks = ((long)B * (C / 8) * (HW / 4) + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(to_p16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, reinterpret_cast<uint4*>(p16), B, C, HW, slot);
}

void launch_post_forward(const PostArgs& a0, hipStream_t s) {
  PostArgs a = a0;
  a.nt_st = (g_nt_stores >> 2) & 1;
  a.nt = post_nt_mode() >= 0 ? (post_nt_mode() >> 2) & 1 : 0;        // forward: measured no gain at either size (0.7356 -> 0.7385 ms at cfg3)
  const long n = (long)a.B * a.C * (a.pool ? (a.H >> 1) * (a.W >> 1) : a.H * a.W);
  if (a.p16) {      // caller checked post_g8_supported
    const long hwo = a.pool ? (long)(a.H >> 1) * (a.W >> 1) : (long)a.H * a.W;
    long blocks = (long)a.B * (a.C / 8) * (hwo > T8_PXT ? hwo / T8_PXT : 1);      // one (image, 8-channel group, pixel tile) per block step
    if (blocks > 4096 && hwo > 256) blocks = 4096;
    KtScope kt("post_forward_g8_kernel", 0.0, 4.0 * ((double)a.B * a.C * a.H * a.W + (a.out ? 2.0 : 1.0) * (double)n), s);
    if (hwo <= 256) {
      if (blocks > 8192) blocks = 8192;
      if (a.pool) with_combo(post_combo(a), [&](auto cb) { hipLaunchKernelGGL((post_forward_g8_kernel<true, 256, decltype(cb)::value>), dim3((unsigned)blocks), dim3(256), 0, s, a); });
      else with_combo(post_combo(a), [&](auto cb) { hipLaunchKernelGGL((post_forward_g8_kernel<false, 256, decltype(cb)::value>), dim3((unsigned)blocks), dim3(256), 0, s, a); });
    } else if (g8_half_tiles() == 2) {
      blocks *= 4; if (blocks > 8192) blocks = 8192;
      if (a.pool) with_combo(post_combo(a), [&](auto cb) { hipLaunchKernelGGL((post_forward_g8_kernel<true, 256, decltype(cb)::value>), dim3((unsigned)blocks), dim3(256), 0, s, a); });
      else with_combo(post_combo(a), [&](auto cb) { hipLaunchKernelGGL((post_forward_g8_kernel<false, 256, decltype(cb)::value>), dim3((unsigned)blocks), dim3(256), 0, s, a); });
    }
#ifdef GR_ABLATE      // GR_G8_HALF_TILES=1 / 0: 512- and 1024-pixel LDS tiles (round 2's measurements; the shipping library runs 256-pixel tiles everywhere)
    else if (g8_half_tiles() && hwo % 512 == 0) {
      blocks *= 2; if (blocks > 8192) blocks = 8192;
      if (a.pool) with_combo(post_combo(a), [&](auto cb) { hipLaunchKernelGGL((post_forward_g8_kernel<true, 512, decltype(cb)::value>), dim3((unsigned)blocks), dim3(256), 0, s, a); });
      else with_combo(post_combo(a), [&](auto cb) { hipLaunchKernelGGL((post_forward_g8_kernel<false, 512, decltype(cb)::value>), dim3((unsigned)blocks), dim3(256), 0, s, a); });
    } else {
      if (a.pool) with_combo(post_combo(a), [&](auto cb) { hipLaunchKernelGGL((post_forward_g8_kernel<true, 1024, decltype(cb)::value>), dim3((unsigned)blocks), dim3(256), 0, s, a); });
      else with_combo(post_combo(a), [&](auto cb) { hipLaunchKernelGGL((post_forward_g8_kernel<false, 1024, decltype(cb)::value>), dim3((unsigned)blocks), dim3(256), 0, s, a); });
    }
#endif
    return;
  }
  const bool vec = (a.pool ? (a.W % 8 == 0 && a.H % 2 == 0) : (a.W % 4 == 0)) && (long)a.B * a.C * a.H * a.W < (1l << 32);
  long blocks = ((vec ? n / 4 : n) + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  if (blocks < 1) blocks = 1;
  if (vec) {
    KtScope kt("post_forward_vec_kernel", 0.0, 4.0 * ((double)a.B * a.C * a.H * a.W + (double)n), s);
    with_combo(post_combo(a), [&](auto cb) { hipLaunchKernelGGL(post_forward_vec_kernel<decltype(cb)::value>, dim3((unsigned)blocks), dim3(256), 0, s, a); });
    return;
  }
  KtScope kt("post_forward_kernel", 0.0, 4.0 * ((double)a.B * a.C * a.H * a.W + (double)n), s);
  hipLaunchKernelGGL(post_forward_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------ BN statistics
static inline int stat_splits(long n) {
  long s = n / 4096;
  if (s < 1) s = 1;
  if (s > STAT_SPLITS) s = STAT_SPLITS;
  return (int)s;
}
// The float4 kernels split the BATCH: block sp owns images [sp*per, min(B, (sp+1)*per)), per = ceil(B / splits).  With
// splits = min(stat_splits, B) alone the last blocks can start past B (B = 29, per = 2: blocks 15.. own nothing), so the
// split count is re-derived from `per`: every block owns at least one image.  (The kernels also guard b0 >= b1.)
static inline int batch_splits(long n, int B) {
  int s = stat_splits(n);
  if (s > B) s = B;
  const int per = (B + s - 1) / s;
  return (B + per - 1) / per;
}

__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ y, int B, int C, int HW, int splits,
                                                               double* __restrict__ partials) {
  __shared__ double sh[8];
  const int c = blockIdx.x, sp = blockIdx.y;
  const long n = (long)B * HW, chunk = (n + splits - 1) / splits;
  const long j0 = sp * chunk, j1 = min(n, j0 + chunk);
  double s = 0, q = 0;
  for (long j = j0 + threadIdx.x; j < j1; j += blockDim.x) {
    const long b = j / HW; const int p = (int)(j - b * HW);
    const double v = y[(b * C + c) * HW + p];
    s += v; q += v * v;
  }
  s = block_reduce_sum(s, sh);
  q = block_reduce_sum(q, sh);
  if (threadIdx.x == 0) { partials[((long)c * STAT_SPLITS + sp) * 2] = s; partials[((long)c * STAT_SPLITS + sp) * 2 + 1] = q; }
}

__global__ void bn_stats_finalize_kernel(const double* __restrict__ partials, int C, int splits, double n,
                                         float* mean, float* invstd, float* run_mean, float* run_var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0, q = 0;
  for (int k = 0; k < splits; ++k) { s += partials[((long)c * STAT_SPLITS + k) * 2]; q += partials[((long)c * STAT_SPLITS + k) * 2 + 1]; }
  const double m = s / n;
  double vs = q - s * m;               // sum (x-mean)^2
  if (vs < 0) vs = 0;
  mean[c] = (float)m;
  invstd[c] = (float)(1.0 / sqrt(vs / n + 1e-5));
  if (run_mean) {
    run_mean[c] = (float)(0.1 * m + 0.9 * (double)run_mean[c]);
    run_var[c] = (float)(0.1 * (vs / (n - 1)) + 0.9 * (double)run_var[c]);
  }
}

__global__ void bn_eval_prepare_kernel(const float* rm, const float* rv, float* mean, float* invstd, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  mean[c] = rm[c];
  invstd[c] = (float)(1.0 / sqrt((double)rv[c] + 1e-5));
}

// float4 variant: block (c, split) walks its images; inside a plane 256 threads take consecutive float4s (no divisions)
__global__ __launch_bounds__(256) void bn_stats_partial_vec_kernel(const float* __restrict__ y, int B, int C, int HW, int splits,
                                                                   double* __restrict__ partials) {
  __shared__ double sh[8];
  const int c = blockIdx.x, sp = blockIdx.y;
  const int per = (B + splits - 1) / splits, b0 = sp * per, b1 = min(B, b0 + per), q4 = HW >> 2;
  double s = 0, q = 0;
  const unsigned tot = b1 > b0 ? (unsigned)(b1 - b0) * q4 : 0u;     // an empty slice contributes zeros
  for (unsigned j = threadIdx.x; j < tot; j += 256) {
    const unsigned bb = udivp(j, (unsigned)q4), i = j - bb * q4;
    const float4 v = reinterpret_cast<const float4*>(y + ((size_t)(b0 + bb) * C + c) * HW)[i];
    s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
    q += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  s = block_reduce_sum(s, sh);
  q = block_reduce_sum(q, sh);
  if (threadIdx.x == 0) { partials[((long)c * STAT_SPLITS + sp) * 2] = s; partials[((long)c * STAT_SPLITS + sp) * 2 + 1] = q; }
}

// one wave per channel: lane l adds pairs l, l + 64, ... of its row, the lanes meet in a fixed shuffle tree
__global__ __launch_bounds__(256) void pair_sums_kernel(const double* __restrict__ part, int stride, int count, int C, double* __restrict__ out) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  const double2* row = reinterpret_cast<const double2*>(part) + (size_t)c * stride;
  double s = 0, q = 0;
  for (int t = lane; t < count; t += 64) { const double2 v = row[t]; s += v.x; q += v.y; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s += __shfl_down(s, off, 64); q += __shfl_down(q, off, 64); }
  if (lane == 0) { out[2 * c] = s; out[2 * c + 1] = q; }
}
__global__ void pair_scatter_kernel(const double* __restrict__ in, int C, int stride, double* __restrict__ part) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  part[(size_t)c * stride * 2] = in[2 * c]; part[(size_t)c * stride * 2 + 1] = in[2 * c + 1];
}
void launch_pair_sums(const double* part, int stride, int count, int C, double* out, hipStream_t s) {
  KtScope kt("pair_sums_kernel", 0.0, 16.0 * count * C, s);
  hipLaunchKernelGGL(pair_sums_kernel, dim3((C + 3) / 4), dim3(256), 0, s, part, stride, count, C, out);
}
void launch_pair_scatter(const double* in, int C, int stride, double* part, hipStream_t s) {
  hipLaunchKernelGGL(pair_scatter_kernel, dim3((C + 255) / 256), dim3(256), 0, s, in, C, stride, part);
}

void launch_bn_stats(const float* y, int B, int C, int HW, double* partials, float* mean, float* invstd,
                     float* run_mean, float* run_var, int training, hipStream_t s, const StatSync* sync) {
  (void)training;
  const long n = (long)B * HW;
  int splits = stat_splits(n);
  if (HW % 4 == 0 && HW >= 64) {
    splits = batch_splits(n, B);
    KtScope kt("bn_stats_partial_vec_kernel", 0.0, 4.0 * (double)n * C, s);
    hipLaunchKernelGGL(bn_stats_partial_vec_kernel, dim3(C, splits), dim3(256), 0, s, y, B, C, HW, splits, partials);
  } else {
    KtScope kt("bn_stats_partial_kernel", 0.0, 4.0 * (double)n * C, s);
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(C, splits), dim3(256), 0, s, y, B, C, HW, splits, partials);
  }
  if (sync) {      // synchronised BatchNorm: the ranks' (sum, sum of squares) are added before the statistics are formed from them
    launch_pair_sums(partials, STAT_SPLITS, splits, C, sync->buf, s);
    if (sync->sum(sync->user, sync->buf, 2L * C)) return;
    launch_bn_stats_from_tiles(sync->buf, 1, C, sync->n_global, mean, invstd, run_mean, run_var, s, nullptr);
    return;
  }
  KtScope kt("bn_stats_finalize_kernel", 0.0, 0.0, s);
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, partials, C, splits, (double)n,
                     mean, invstd, run_mean, run_var);
}
// one workgroup per channel: the conv epilogue's per-tile (sum, sum of squares) added in a fixed order
__device__ __forceinline__ void amax_fold(unsigned* slot, int entry, float v) {
  unsigned* e = slot + (entry % AMAX_ENTRIES) * AMAX_STRIDE;
  atomicMax(e, __float_as_uint(v));
}
// One WAVE per channel (four channels per workgroup): every lane adds its share of the channel's per-tile (sum, sum of squares)
// pairs - all its loads issued before the first add - and the 64 lane sums meet in a fixed shuffle tree.  No LDS, no barrier:
// the kernel sits on the forward critical path between a convolution and its pipeline kernel six times per step, and the
// workgroup-per-channel version spent most of its 7.8 us in two barrier-separated block reductions (round 3: VERDICT item 4a).
__global__ __launch_bounds__(256) void bn_stats_finalize_tiles_kernel(const double* __restrict__ part, int tiles, double n, int C,
                                                                       float* mean, float* invstd, float* run_mean, float* run_var, BnBounds bd) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;                                            // (whole waves leave together)
  const unsigned ymax_bits = bd.amax_y ? absmax_read(bd.amax_y) : 0u;
  const double2* row = reinterpret_cast<const double2*>(part) + (size_t)c * tiles;
  double s = 0, q = 0;
  for (int t0 = lane; t0 < tiles; t0 += 64 * 8) {
    double2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int t = t0 + 64 * u; v[u] = t < tiles ? row[t] : make_double2(0.0, 0.0); }
#pragma unroll
    for (int u = 0; u < 8; ++u) { s += v[u].x; q += v[u].y; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s += __shfl_down(s, off, 64); q += __shfl_down(q, off, 64); }
  if (lane == 0) {
    const double m = s / n;
    double vs = q - s * m;               // sum (x-mean)^2
    if (vs < 0) vs = 0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(vs / n + 1e-5));
    if (run_mean) {
      run_mean[c] = (float)(0.1 * m + 0.9 * (double)run_mean[c]);
      run_var[c] = (float)(0.1 * (vs / (n - 1)) + 0.9 * (double)run_var[c]);
    }
    if (bd.amax_y) {
      // |y - mean| <= max|y| + |mean| for every element of the channel; every activation on this path has |act(z)| <= |z|
      // (ELU: |e^z - 1| <= |z| for z <= 0), Sigmoid / Tanh additionally <= 1; masks multiply by at most mask_scale
      const float ymax = __uint_as_float(ymax_bits), is = invstd[c], dev = (ymax + fabsf(mean[c])) * is;
      float zb = dev * fabsf(bd.gamma[c]) + fabsf(bd.beta[c]);
      if (bd.act == ACT_SIGMOID || bd.act == ACT_TANH) zb = fminf(zb, 1.f);
      if (bd.bound_out) amax_fold(bd.bound_out, c, zb * bd.mask_scale * 1.0001f);
      // backward: dy = ((dz - mean(dz)) - yhat * mean(yhat dz)) * invstd * gamma, |mean(dz)| <= max|dz|,
      // |mean(yhat dz)| <= sqrt(mean yhat^2) * max|dz| <= max|dz|  =>  |dy| <= (2 + max|yhat|) * invstd * |gamma| * max|dz|
      if (bd.kb_out) amax_fold(bd.kb_out, c, (2.f + dev) * is * fabsf(bd.gamma[c]) * 1.0001f);
    }
  }
}
void launch_bn_stats_from_tiles(const double* stat_part, int tiles, int C, double n, float* mean, float* invstd,
                                float* run_mean, float* run_var, hipStream_t s, const BnBounds* bounds) {
  KtScope kt("bn_stats_finalize_tiles_kernel", 0.0, 16.0 * tiles * C, s);
  BnBounds bd{}; if (bounds) bd = *bounds;
  hipLaunchKernelGGL(bn_stats_finalize_tiles_kernel, dim3((C + 3) / 4), dim3(256), 0, s, stat_part, tiles, n, C, mean, invstd, run_mean, run_var, bd);
}
void launch_bn_eval_prepare(const float* rm, const float* rv, float* mean, float* invstd, int C, hipStream_t s) {
  hipLaunchKernelGGL(bn_eval_prepare_kernel, dim3((C + 255) / 256), dim3(256), 0, s, rm, rv, mean, invstd, C);
}

// ------------------------------------------------------------------ backward pipeline
// pass A: dz = grad wrt the BN output (or wrt y when there is no BN); per-channel partial sums of dz and (y-mean)*dz
__global__ __launch_bounds__(256) void post_backward_a_kernel(PostBwdArgs a, int splits) {
  __shared__ double sh[8];
  const PostArgs& f = a.f;
  const int c = blockIdx.x, sp = blockIdx.y;
  const int H = f.H, W = f.W, Ho = f.pool ? H >> 1 : H, Wo = f.pool ? W >> 1 : W;
  const long HW = (long)H * W, HWo = (long)Ho * Wo;
  const long n = (long)f.B * HW, chunk = (n + splits - 1) / splits;
  const long j0 = sp * chunk, j1 = min(n, j0 + chunk);
  const float mean = f.has_bn ? f.mean[c] : 0.f;
  double s = 0, q = 0;
  float dmax = 0.f;
  for (long j = j0 + threadIdx.x; j < j1; j += blockDim.x) {
    const long b = j / HW; const int p = (int)(j - b * HW);
    const long bc = b * f.C + c, e = bc * HW + p;
    float g;
    if (f.pool) {
      const int yy = p / W, xx = p - yy * W, yo = yy >> 1, xo = xx >> 1;
      g = 0.f;
      if (yo < Ho && xo < Wo) {
        const long eo = bc * HWo + (long)yo * Wo + xo;
        const int t = ((yy & 1) << 1) | (xx & 1);
        if (f.pool_idx[eo] == t) g = a.gout[eo] * mask_mul(f.m2, eo, bc);
      }
    } else {
      g = a.gout[e] * mask_mul(f.m2, e, bc);
    }
    g = g * mask_mul(f.m1, e, bc);
    const float yv = f.y[e];
    const float z = bn_apply(f, yv, c);
    const float av = act_fwd(z, f.act, post_slope(f));
    const float dz = act_bwd(g, z, av, f.act, post_slope(f));
    a.dy[e] = dz;
    dmax = fmaxf(dmax, fabsf(dz));
    s += (double)dz;
    q += (double)(yv - mean) * (double)dz;
  }
  if (a.amax_dy && !f.has_bn) absmax_commit(dmax, a.amax_dy);      // without BN pass A's dz is the final dy
  s = block_reduce_sum(s, sh);
  q = block_reduce_sum(q, sh);
  if (threadIdx.x == 0) { a.partials[((long)c * STAT_SPLITS + sp) * 2] = s; a.partials[((long)c * STAT_SPLITS + sp) * 2 + 1] = q; }
}

// dz (gradient wrt the BatchNorm output, or wrt y without BN) of four consecutive pre-pool elements e .. e+3 of plane bc:
// gradOutput routed back through mask 2, the pool argmax, mask 1 and the activation.  Pass A sums it, pass B needs it again:
// with BatchNorm both passes call this (bit-identical results) and dz is never written to memory - one tensor write and one
// tensor read less than storing it.
// The same in two phases for kernels that work on several channels per thread: all loads of a channel first (so that the loads
// of eight channels are in flight together), the arithmetic afterwards.  Same operations in the same order as post_bwd_dz4.
struct BwdRaw { float4 g, y; uint32_t id2, m2w, m1w, t0; };

__device__ __forceinline__ BwdRaw post_bwd_load4(const PostBwdArgs& a, unsigned bc, unsigned e, unsigned i, unsigned obase, unsigned wq, unsigned Wo) {
  const PostArgs& f = a.f;
  BwdRaw r;
  if (f.pool) {
    const unsigned yy = udivp(i, wq), xx = (i - yy * wq) * 4, eo = obase + (yy >> 1) * Wo + (xx >> 1);
    const float2 go = *reinterpret_cast<const float2*>(a.gout + eo);
    r.g = make_float4(go.x, go.y, 0.f, 0.f);
    r.id2 = *reinterpret_cast<const uint16_t*>(f.pool_idx + eo);
    r.m2w = mask_word(f.m2, eo, bc);                  // eo is even: the bits of eo and eo + 1 sit in one word
    r.t0 = (yy & 1) << 1;
  } else {
    r.g = ld4_maybe_nt(a.gout + e, a.nt != 0);
    r.id2 = 0; r.t0 = 0;
    r.m2w = mask_word(f.m2, e, bc);
  }
  r.m1w = mask_word(f.m1, e, bc);
  r.y = ld4_maybe_nt(f.y + e, a.nt != 0);
  return r;
}
__device__ __forceinline__ float4 post_bwd_dz_of(const PostBwdArgs& a, const BwdRaw& r, float mean, float invstd, float gm, float bt) {
  const PostArgs& f = a.f;
  float4 g;
  if (f.pool) {
    const float4 m2 = mask4_of(f.m2, r.m2w);
    const float m20 = m2.x, m21 = m2.y;
    g.x = ((r.id2 & 0xff) == r.t0) ? r.g.x * m20 : 0.f;
    g.y = ((r.id2 & 0xff) == (r.t0 | 1)) ? r.g.x * m20 : 0.f;
    g.z = ((r.id2 >> 8) == r.t0) ? r.g.y * m21 : 0.f;
    g.w = ((r.id2 >> 8) == (r.t0 | 1)) ? r.g.y * m21 : 0.f;
  } else {
    g = mul4(r.g, mask4_of(f.m2, r.m2w));
  }
  g = mul4(g, mask4_of(f.m1, r.m1w));
  const float4 yv = r.y;
  float4 z = yv;
  if (f.has_bn) {
    z.x = ((yv.x - mean) * invstd) * gm + bt; z.y = ((yv.y - mean) * invstd) * gm + bt;
    z.z = ((yv.z - mean) * invstd) * gm + bt; z.w = ((yv.w - mean) * invstd) * gm + bt;
  }
  float4 dz;
  dz.x = act_bwd_z(g.x, z.x, f.act, post_slope(f)); dz.y = act_bwd_z(g.y, z.y, f.act, post_slope(f));
  dz.z = act_bwd_z(g.z, z.z, f.act, post_slope(f)); dz.w = act_bwd_z(g.w, z.w, f.act, post_slope(f));
  return dz;
}

__device__ __forceinline__ float4 post_bwd_dz4(const PostBwdArgs& a, unsigned bc, unsigned e, unsigned i, unsigned obase, unsigned wq, unsigned Wo,
                                               float mean, float invstd, float gm, float bt, float4& yv) {
  const PostArgs& f = a.f;
  float4 g;
  if (f.pool) {
    const unsigned yy = udivp(i, wq), xx = (i - yy * wq) * 4, eo = obase + (yy >> 1) * Wo + (xx >> 1);
    const float2 go = *reinterpret_cast<const float2*>(a.gout + eo);
    const uint32_t id2 = *reinterpret_cast<const uint16_t*>(f.pool_idx + eo);
    const float m20 = mask_mul(f.m2, eo, bc), m21 = mask_mul(f.m2, eo + 1, bc);
    const uint32_t t0 = (yy & 1) << 1;
    g.x = ((id2 & 0xff) == t0) ? go.x * m20 : 0.f;
    g.y = ((id2 & 0xff) == (t0 | 1)) ? go.x * m20 : 0.f;
    g.z = ((id2 >> 8) == t0) ? go.y * m21 : 0.f;
    g.w = ((id2 >> 8) == (t0 | 1)) ? go.y * m21 : 0.f;
  } else {
    g = mul4(*reinterpret_cast<const float4*>(a.gout + e), mask4(f.m2, e, bc));
  }
  g = mul4(g, mask4(f.m1, e, bc));
  yv = *reinterpret_cast<const float4*>(f.y + e);
  float4 z = yv;
  if (f.has_bn) {
    z.x = ((yv.x - mean) * invstd) * gm + bt; z.y = ((yv.y - mean) * invstd) * gm + bt;
    z.z = ((yv.z - mean) * invstd) * gm + bt; z.w = ((yv.w - mean) * invstd) * gm + bt;
  }
  float4 dz;
  dz.x = act_bwd_z(g.x, z.x, f.act, post_slope(f)); dz.y = act_bwd_z(g.y, z.y, f.act, post_slope(f));
  dz.z = act_bwd_z(g.z, z.z, f.act, post_slope(f)); dz.w = act_bwd_z(g.w, z.w, f.act, post_slope(f));
  return dz;
}

// float4 variants of pass A / pass B: block (c, split) walks its images, threads take consecutive pre-pool float4s.
template <int CB>
__global__ __launch_bounds__(256) void post_backward_a_vec_kernel(PostBwdArgs a, int splits) {
  __shared__ double sh[8];
  post_specialize<CB>(a.f);
  const PostArgs& f = a.f;
  const int c = blockIdx.x, sp = blockIdx.y;
  const unsigned H = f.H, W = f.W, Wo = f.pool ? W >> 1 : W, HW = H * W, HWo = f.pool ? (H >> 1) * Wo : HW;
  const int per = (f.B + splits - 1) / splits, b0 = sp * per, b1 = min(f.B, b0 + per);
  const unsigned q4 = HW >> 2, wq = W >> 2;
  float mean = 0.f, invstd = 1.f, gm = 1.f, bt = 0.f;
  if (f.has_bn) { mean = f.mean[c]; invstd = f.invstd[c]; gm = f.gamma[c]; bt = f.beta[c]; }
  double s = 0, q = 0;
  float dmax = 0.f;
  const unsigned tot = b1 > b0 ? (unsigned)(b1 - b0) * q4 : 0u;     // an empty slice contributes zeros
  {
    // four float4 groups per thread and round: all their loads are issued before the first is used (a block owns only ~1024
    // groups - four per thread - so without this every thread waits out one memory round trip per group); post_bwd_load4 /
    // post_bwd_dz_of are the two halves of post_bwd_dz4, same operations in the same order
    // A block owns several rounds of 1024 groups whenever its batch slice holds more than 4096 elements per channel (cfg3: 8 images of 64 x 64 = 8 rounds;
    // cfg2: one round); round k + 1 is requested before round k is worked on.  (The ablation launcher GR_PASSA_SPLIT_DIV, which MAKES blocks longer at
    // cfg2, measured no gain: profiles/r05_ab_passa_prefetch_cfg2.txt - the second buffer set is for the naturally long blocks.)
    BwdRaw r[4], rn[4]; unsigned ee[4], een[4];
    auto request = [&](unsigned j0, BwdRaw* rr, unsigned* e_) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned j = j0 + 256u * u;
        if (j < tot) {
          const unsigned bb = udivp(j, q4), i = j - bb * q4;
          const unsigned bc = (unsigned)(b0 + bb) * f.C + c, pbase = bc * HW, obase = bc * HWo;
          e_[u] = pbase + i * 4;
          rr[u] = post_bwd_load4(a, bc, e_[u], i, obase, wq, Wo);
        }
      }
    };
    if (threadIdx.x < tot) request(threadIdx.x, r, ee);
    for (unsigned j0 = threadIdx.x; j0 < tot; j0 += 1024) {
      const bool more = j0 + 1024 < tot;
      if (more) request(j0 + 1024, rn, een);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (j0 + 256u * u < tot) {
          const float4 dz = post_bwd_dz_of(a, r[u], mean, invstd, gm, bt);
          const float4 yv = r[u].y;
          if (!f.has_bn) *reinterpret_cast<float4*>(a.dy + ee[u]) = dz;     // with BatchNorm pass B recomputes dz: nothing stored here
          dmax = absmax4(dmax, dz);
          // the four elements of a group are added in fp32 (pairwise), the groups in fp64: 2 conversions + 2 fp64 adds per
          // group instead of 20 fp64-class instructions (half rate on gfx950, and these kernels are VALU-bound); the partial's
          // rounding is 2 ulp of a 4-term sum, far inside the 1e-4 bar
          s += (double)((dz.x + dz.y) + (dz.z + dz.w));
          q += (double)(((yv.x - mean) * dz.x + (yv.y - mean) * dz.y) + ((yv.z - mean) * dz.z + (yv.w - mean) * dz.w));
        }
      }
      if (more) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { r[u] = rn[u]; ee[u] = een[u]; }
      }
    }
  }
  if (a.amax_dy && !f.has_bn) absmax_commit(dmax, a.amax_dy);      // without BN pass A's dz is the final dy
  if (a.amax_dz) absmax_commit(dmax, a.amax_dz);                    // max|dz|: pass B's a-priori bound of max|dy| (operand-ready dy)
  s = block_reduce_sum(s, sh);
  q = block_reduce_sum(q, sh);
  if (threadIdx.x == 0) { a.partials[((long)c * STAT_SPLITS + sp) * 2] = s; a.partials[((long)c * STAT_SPLITS + sp) * 2 + 1] = q; }
}

// BatchNorm-backward coefficients of channel c from pass A's partial sums (what post_backward_finalize_kernel computes, same
// summation order): every pass-B workgroup derives them itself - 2 x `splits` doubles - instead of a launch in between; the
// workgroup of split 0 also accumulates the gamma / beta gradients.
__device__ __forceinline__ void post_bwd_coef(const PostBwdArgs& a, int c, int sp, int splits, double n, float* sh_coef) {
  __shared__ double sh_part[2 * STAT_SPLITS];      // fetched by 2 * splits threads at once, added by one in split order
  if ((int)threadIdx.x < 2 * splits) sh_part[threadIdx.x] = a.partials[(long)c * STAT_SPLITS * 2 + threadIdx.x];
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0, q = 0;
    for (int k = 0; k < splits; ++k) { s += sh_part[2 * k]; q += sh_part[2 * k + 1]; }
    const double invstd = a.f.invstd[c];
    sh_coef[0] = (float)(s / n);
    sh_coef[1] = (float)(q * invstd * invstd / n);
    if (sp == 0) { const double gs = a.gscale != 0.0 ? a.gscale : 1.0; a.ggamma[c] += (float)(q * invstd * gs); a.gbeta[c] += (float)(s * gs); }
  }
  __syncthreads();
}

template <int CB>
__global__ __launch_bounds__(256) void post_backward_b_vec_kernel(PostBwdArgs a, int splits, double n, int psplits) {
  __shared__ double sh[8];
  __shared__ float sh_coef[2];
  post_specialize<CB>(a.f);
  const PostArgs& f = a.f;
  const int c = blockIdx.x, sp = blockIdx.y;
  post_bwd_coef(a, c, sp, psplits, n, sh_coef);      // psplits: pass A's pairs per channel (one under synchronised BatchNorm); splits: this grid's batch slices
  const unsigned H = f.H, W = f.W, Wo = f.pool ? W >> 1 : W, HW = H * W, HWo = f.pool ? (H >> 1) * Wo : HW, q4 = HW >> 2, wq = W >> 2;
  const int per = (f.B + splits - 1) / splits, b0 = sp * per, b1 = min(f.B, b0 + per);
  const float mean = f.mean[c], invstd = f.invstd[c], w = f.gamma[c], bt = f.beta[c], gm = sh_coef[0], k = sh_coef[1];
  double s = 0;
  float dmax = 0.f;
  const unsigned tot = b1 > b0 ? (unsigned)(b1 - b0) * q4 : 0u;     // an empty slice contributes zeros
  {
    for (unsigned j = threadIdx.x; j < tot; j += 256) {
      const unsigned bb = udivp(j, q4), i = j - bb * q4;
      const unsigned bc = (unsigned)(b0 + bb) * f.C + c;
      const size_t base = (size_t)bc * HW;
      float4* dyp = reinterpret_cast<float4*>(a.dy + base);
      float4 yv;
      const float4 dz = post_bwd_dz4(a, bc, bc * HW + i * 4, i, bc * HWo, wq, Wo, mean, invstd, w, bt, yv);   // as pass A computed it
      float4 d;
      d.x = ((dz.x - gm) - (yv.x - mean) * k) * invstd * w; d.y = ((dz.y - gm) - (yv.y - mean) * k) * invstd * w;
      d.z = ((dz.z - gm) - (yv.z - mean) * k) * invstd * w; d.w = ((dz.w - gm) - (yv.w - mean) * k) * invstd * w;
      dyp[i] = d;
      dmax = absmax4(dmax, d);
      s += (double)((d.x + d.y) + (d.z + d.w));
    }
  }
  if (a.amax_dy) absmax_commit(dmax, a.amax_dy);
  s = block_reduce_sum(s, sh);
  if (threadIdx.x == 0) a.partials_b[(long)c * PB_SPLITS + sp] = s;
}

// Operand-ready pass B: block (8-channel group, batch slice), light threads (one channel x 4 consecutive pre-pool pixels) and
// the LDS transpose of the forward kernel (t8_pack / t8_emit).  Writes dy as the data- / weight-gradient convolutions' image
// dy_p16[b][g][term][pixel], scaled by the power of two of the bound K * max|dz| (K from the forward's statistics, max|dz| from
// pass A) that it also leaves in amax_dy, and as fp32 only when a consumer still needs that (a.dy != null).
template <int PXT, int CB>
__global__ __launch_bounds__(256) void post_backward_b_g8_kernel(PostBwdArgs a, int splits, int slices, double n, int dbg_) {
  const int dbg = GR_DBG(dbg_);
  post_specialize<CB>(a.f);
  constexpr int RSB = PXT * 2 + 64;
  __shared__ __attribute__((aligned(16))) unsigned char img[16 * RSB];
  __shared__ double sh_part[16];
  __shared__ float par[8][6];          // mean, invstd, gamma, beta, gm, k
  __shared__ double chsum[8];          // per-channel sums of dy
  const PostArgs& f = a.f;
  const int g = blockIdx.x, sp = blockIdx.y;
  const float bound = __uint_as_float(absmax_read(a.amax_dz)) * __uint_as_float(absmax_read(a.kb));
  // pass A's partial sums of the 8 channels, (sum, dot) x splits each: 16 sums of up to 64 terms.  Thread t adds four
  // consecutive splits of sum (t >> 4) itself, the 16 threads of a sum are then added in a fixed shuffle tree: 5 dependent
  // steps instead of a 64-step serial loop per block (this kernel runs one block per image: the prologue is not amortised).
  {
    const int si = threadIdx.x >> 4, part = threadIdx.x & 15, jc = si >> 1, which = si & 1;
    const double* row = a.partials + (long)(8 * g + jc) * STAT_SPLITS * 2 + which;
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int kk = part * 4 + k; if (kk < splits) v += row[2 * kk]; }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) v += __shfl_down(v, off, 16);
    if (part == 0) sh_part[si] = v;
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    const int c = 8 * g + threadIdx.x;
    const double s = sh_part[2 * threadIdx.x], q = sh_part[2 * threadIdx.x + 1];
    const double invstd = f.invstd[c];
    par[threadIdx.x][0] = f.mean[c]; par[threadIdx.x][1] = f.invstd[c]; par[threadIdx.x][2] = f.gamma[c]; par[threadIdx.x][3] = f.beta[c];
    par[threadIdx.x][4] = (float)(s / n);
    par[threadIdx.x][5] = (float)(q * invstd * invstd / n);
    if (sp == 0) { const double gs = a.gscale != 0.0 ? a.gscale : 1.0; a.ggamma[c] += (float)(q * invstd * gs); a.gbeta[c] += (float)(s * gs); }
  }
  if (threadIdx.x == 0) a.amax_dy[((blockIdx.x + blockIdx.y * gridDim.x) % AMAX_ENTRIES) * AMAX_STRIDE] = __float_as_uint(bound);   // the same value from every block
  __syncthreads();
  const float sc = pow2f(f16_scale_exp(__float_as_uint(bound)));
  const unsigned H = f.H, W = f.W, Wo = f.pool ? W >> 1 : W, HW = H * W, HWo = f.pool ? (H >> 1) * Wo : HW, wq = W >> 2;
  const unsigned G = (unsigned)f.C >> 3;
  const unsigned npx = HW < (unsigned)PXT ? HW : (unsigned)PXT, tiles = HW / npx, qpt = npx >> 2;
  const int per = (f.B + slices - 1) / slices, b0 = sp * per, b1 = min(f.B, b0 + per);      // blockIdx.y = one of `slices` batch slices (finer than pass A's splits)
  uint4* p16 = reinterpret_cast<uint4*>(a.dy_p16);
  const unsigned jc = threadIdx.x >> 5, q0 = threadIdx.x & 31;      // this thread's channel of the group and its first quad
  const float mean = par[jc][0], invstd = par[jc][1], w = par[jc][2], bt = par[jc][3], gm = par[jc][4], kk = par[jc][5];
  double csum = 0.0;
  for (int b = b0; b < b1; ++b)
    for (unsigned tile = 0; tile < tiles; ++tile) {
      const unsigned bcj = (unsigned)b * f.C + 8 * g + jc;
      // thread -> (channel jc = tid / 32, quads (tid & 31) + 32 k): a thread stays on ONE channel, so its bias-gradient sum is a
      // single register (an array indexed by a task slot went to scratch memory); four quads at a time: their loads (gradOutput,
      // y, masks, argmax) are in flight together, then the arithmetic.  qpt is a multiple of 64.
      for (unsigned k0 = 0; k0 < qpt / 32; k0 += 4) {
        BwdRaw raw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned qi = q0 + 32 * (k0 + u), i = tile * qpt + qi;
          if (k0 + u < qpt / 32) {
            if (!(dbg & 128)) raw[u] = post_bwd_load4(a, bcj, bcj * HW + i * 4, i, bcj * HWo, wq, Wo);
            else { raw[u].g = make_float4(1, 2, 3, 4); raw[u].y = make_float4(1, 1, 1, 1); raw[u].id2 = 0; raw[u].m1w = raw[u].m2w = 15; raw[u].t0 = 0; }
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned qi = q0 + 32 * (k0 + u), i = tile * qpt + qi;
          if (k0 + u < qpt / 32) {
            const float4 yv = raw[u].y;
            const float4 dz = post_bwd_dz_of(a, raw[u], mean, invstd, w, bt);                 // as pass A computed it
            float4 d;
            d.x = ((dz.x - gm) - (yv.x - mean) * kk) * invstd * w; d.y = ((dz.y - gm) - (yv.y - mean) * kk) * invstd * w;
            d.z = ((dz.z - gm) - (yv.z - mean) * kk) * invstd * w; d.w = ((dz.w - gm) - (yv.w - mean) * kk) * invstd * w;
            if (a.dy) store4(a.dy + (size_t)bcj * HW + 4 * (size_t)i, d, f.nt_st != 0);     // (null: both gradient kernels take the operand-ready image)
            csum += (double)((d.x + d.y) + (d.z + d.w));
            uint2 hi, lo;
            t8_pack(d, sc, hi, lo);
            *reinterpret_cast<uint2*>(img + (size_t)jc * RSB + qi * 8) = hi;
            *reinterpret_cast<uint2*>(img + (size_t)(8 + jc) * RSB + qi * 8) = lo;
          }
        }
      }
      __syncthreads();
      if (!(dbg & 64)) t8_emit<RSB>(img, (int)npx, p16 + ((size_t)b * G + g) * 2 * HW + (size_t)tile * npx, HW, f.nt_st != 0);
      __syncthreads();
    }
  // bias gradient: per-channel sums of dy = the 32 threads of a channel (one half-wave), added in a fixed shuffle tree
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) csum += __shfl_down(csum, off, 32);
  if (q0 == 0) chsum[jc] = csum;
  __syncthreads();
  if (threadIdx.x < 8) a.partials_b[(long)(8 * g + threadIdx.x) * PB_SPLITS + sp] = chsum[threadIdx.x];
}

__global__ void post_backward_finalize_kernel(PostBwdArgs a, int splits, double n) {
  const PostArgs& f = a.f;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= f.C) return;
  double s = 0, q = 0;
  for (int k = 0; k < splits; ++k) { s += a.partials[((long)c * STAT_SPLITS + k) * 2]; q += a.partials[((long)c * STAT_SPLITS + k) * 2 + 1]; }
  if (f.has_bn) {
    const double invstd = f.invstd[c];
    a.ggamma[c] += (float)(q * invstd);
    a.gbeta[c] += (float)s;
    a.coef[2 * c] = (float)(s / n);
    a.coef[2 * c + 1] = (float)(q * invstd * invstd / n);
  } else if (a.gbias) {
    a.gbias[c] += (float)s;
  }
}

// pass B (BN only): dy = ((dz - gm) - (y-mean)*k) * invstd * gamma ; per-channel sum of dy for the bias gradient
__global__ __launch_bounds__(256) void post_backward_b_kernel(PostBwdArgs a, int splits, double nn, int psplits) {
  __shared__ double sh[8];
  __shared__ float sh_coef[2];
  const PostArgs& f = a.f;
  const int c = blockIdx.x, sp = blockIdx.y;
  post_bwd_coef(a, c, sp, psplits, nn, sh_coef);
  const long HW = (long)f.H * f.W;
  const long n = (long)f.B * HW, chunk = (n + splits - 1) / splits;
  const long j0 = sp * chunk, j1 = min(n, j0 + chunk);
  const float mean = f.mean[c], invstd = f.invstd[c], w = f.gamma[c], gm = sh_coef[0], k = sh_coef[1];
  double s = 0;
  float dmax = 0.f;
  for (long j = j0 + threadIdx.x; j < j1; j += blockDim.x) {
    const long b = j / HW; const int p = (int)(j - b * HW);
    const long e = (b * f.C + c) * HW + p;
    const float d = ((a.dy[e] - gm) - (f.y[e] - mean) * k) * invstd * w;
    a.dy[e] = d;
    dmax = fmaxf(dmax, fabsf(d));
    s += (double)d;
  }
  if (a.amax_dy) absmax_commit(dmax, a.amax_dy);
  s = block_reduce_sum(s, sh);
  if (threadIdx.x == 0) a.partials_b[(long)c * PB_SPLITS + sp] = s;
}

// conv / linear bias gradients of several stages in one launch (blockIdx.y = stage): sums of pass B's per-split sums of dy
// One wave per channel: lane l adds the splits l, l + 64, ... (coalesced rows of partials), then a fixed shuffle tree: the same
// order on every run.  (One THREAD per channel walking its 256 partials one by one took 16 us for 0.4 MB.)
__global__ __launch_bounds__(256) void bias_grad_batch_kernel(BiasJobs jobs) {
  const BiasJob j = jobs.job[blockIdx.y];
  const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= j.C) return;
  double s = 0;
  for (int k = lane; k < j.splits; k += 64) s += j.partials[(long)c * PB_SPLITS + k];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if (lane == 0) j.gbias[c] += (float)s;
}
void launch_bias_grad_batch(BiasJobs& jobs, hipStream_t s) {
  if (jobs.n <= 0) return;
  int maxc = 1;
  for (int i = 0; i < jobs.n; ++i) if (jobs.job[i].C > maxc) maxc = jobs.job[i].C;
  KtScope kt("bias_grad_batch_kernel", 0.0, 0.0, s);
  hipLaunchKernelGGL(bias_grad_batch_kernel, dim3((maxc + 3) / 4, jobs.n), dim3(256), 0, s, jobs);
  jobs.n = 0;
}

void launch_post_backward(const PostBwdArgs& a0, hipStream_t s, BiasJobs* defer, const StatSync* sync) {
  PostBwdArgs a = a0, aB = a0;
  if (!a0.f.has_bn) sync = nullptr;          // nothing is reduced over the batch without a BatchNorm
  if (sync) aB.gscale = a.gscale = sync->grad_scale;
  // pass A and the operand-ready pass B: tensors the Infinity Cache cannot hold (at cfg2's 34 / 67 MB a non-temporal pass A takes from
  // pass B what it would have found in the cache: A 0.169 -> 0.161 ms but B 0.163 -> 0.181); the float4 pass B measured no gain
  a.nt = post_nt_mode() >= 0 ? post_nt_mode() & 1 : (post_big(a0.f) ? 1 : 0);
  aB.f.nt_st = (g_nt_stores >> 3) & 1;
  aB.nt = post_nt_mode() >= 0 ? (post_nt_mode() >> 1) & 1 : ((post_big(a0.f) && a0.dy_p16) ? 1 : 0);
  const PostArgs& f = a.f;
  const long n = (long)f.B * f.H * f.W;
  int splits = stat_splits(n);
  const double pre = (double)n * f.C, post = f.pool ? pre / 4 : pre;
  const bool vec = (f.pool ? (f.W % 8 == 0 && f.H % 2 == 0) : (f.W % 4 == 0)) && f.H * f.W >= 64 && pre < 4.0e9;
  if (vec) {
    splits = batch_splits(n, f.B);
    {
      static const int div = GR_KNOB("GR_PASSA_SPLIT_DIV", 1);
      if (div > 1 && !post_big(a0.f)) { int s2 = splits / div; if (s2 < 1) s2 = 1; const int per = (f.B + s2 - 1) / s2; splits = (f.B + per - 1) / per; }
    }
    KtScope kt("post_backward_a_vec_kernel", 0.0, 4.0 * ((f.has_bn ? 1.0 : 2.0) * pre + post), s);   // with BN: dz is not stored
    with_combo(post_combo(f), [&](auto cb) { hipLaunchKernelGGL(post_backward_a_vec_kernel<decltype(cb)::value>, dim3(f.C, splits), dim3(256), 0, s, a, splits); });
  } else {
    KtScope kt("post_backward_a_kernel", 0.0, 4.0 * (2.0 * pre + post), s);
    hipLaunchKernelGGL(post_backward_a_kernel, dim3(f.C, splits), dim3(256), 0, s, a, splits);
  }
  if (!f.has_bn) {      // no BatchNorm: pass A's dz is dy; only the bias gradient is left to sum
    if (!a.gbias) return;         // ... and an element-wise stage (dropout / pooling behind a PReLU: the D network) has no bias either
    hipLaunchKernelGGL(post_backward_finalize_kernel, dim3((f.C + 255) / 256), dim3(256), 0, s, a, splits, (double)n);
    return;
  }
  double nb = (double)n;        // elements per channel the BatchNorm-backward means are taken over
  int psplits = splits;         // pass A's (sum, dot) pairs per channel that pass B adds
  if (sync) {
    // synchronised BatchNorm: pass A's (sum dz, sum dz (y - mean)) added over the ranks; pass B then finds ONE pair per channel.  The
    // operand-ready dy is scaled by a bound built from max|dz|: the means pass B subtracts are global, so the maximum must be too.
    launch_pair_sums(a.partials, STAT_SPLITS, splits, f.C, sync->buf, s);
    if (sync->sum(sync->user, sync->buf, 2L * f.C)) return;
    launch_pair_scatter(sync->buf, f.C, STAT_SPLITS, a.partials, s);
    if (a.amax_dz && sync->max_u32(sync->user, a.amax_dz, AMAX_WORDS)) return;
    psplits = 1; nb = sync->n_global;
  }
  if (vec && a.dy_p16) {      // caller checked post_g8_supported
    KtScope kt("post_backward_b_g8_kernel", 0.0, 4.0 * ((a.dy ? 3.0 : 2.0) * pre + post), s);        // reads g and y, writes dy operand-ready (and as fp32 when a consumer needs that)
    // 8 channels per block: a (C / 8, splits) grid would leave 2 blocks per CU on a 64-channel layer (measured 72 us against
    // 35 for the per-channel kernel): the batch is sliced down to single images instead, up to PB_SPLITS slices
    int slices = f.B < PB_SPLITS ? f.B : PB_SPLITS;
    { const int per = (f.B + slices - 1) / slices; slices = (f.B + per - 1) / per; }
    if (f.H * f.W <= 256) with_combo(post_combo(f), [&](auto cb) { hipLaunchKernelGGL((post_backward_b_g8_kernel<256, decltype(cb)::value>), dim3(f.C / 8, slices), dim3(256), 0, s, aB, psplits, slices, nb, g_p16_debug); });
    else if (g8_half_tiles() == 2) with_combo(post_combo(f), [&](auto cb) { hipLaunchKernelGGL((post_backward_b_g8_kernel<256, decltype(cb)::value>), dim3(f.C / 8, slices), dim3(256), 0, s, aB, psplits, slices, nb, g_p16_debug); });
#ifdef GR_ABLATE
    else if (g8_half_tiles() && (f.H * f.W) % 512 == 0) with_combo(post_combo(f), [&](auto cb) { hipLaunchKernelGGL((post_backward_b_g8_kernel<512, decltype(cb)::value>), dim3(f.C / 8, slices), dim3(256), 0, s, aB, psplits, slices, nb, g_p16_debug); });
    else with_combo(post_combo(f), [&](auto cb) { hipLaunchKernelGGL((post_backward_b_g8_kernel<1024, decltype(cb)::value>), dim3(f.C / 8, slices), dim3(256), 0, s, aB, psplits, slices, nb, g_p16_debug); });
#endif
    if (a.gbias) {
      BiasJobs one{}; one.n = 0;
      BiasJobs* q = defer ? defer : &one;
      if (q->n == 16) launch_bias_grad_batch(*q, s);
      q->job[q->n++] = BiasJob{a.partials_b, a.gbias, f.C, slices};
      if (!defer) launch_bias_grad_batch(one, s);
    }
    return;
  } else if (vec) {
    KtScope kt("post_backward_b_vec_kernel", 0.0, 4.0 * (2.0 * pre + post), s);                      // reads g and y, writes dy
    with_combo(post_combo(f), [&](auto cb) { hipLaunchKernelGGL(post_backward_b_vec_kernel<decltype(cb)::value>, dim3(f.C, splits), dim3(256), 0, s, aB, splits, nb, psplits); });
  } else {
    KtScope kt("post_backward_b_kernel", 0.0, 4.0 * 3.0 * pre, s);
    hipLaunchKernelGGL(post_backward_b_kernel, dim3(f.C, splits), dim3(256), 0, s, a, splits, nb, psplits);
  }
  if (a.gbias) {
    BiasJobs one{}; one.n = 0;
    BiasJobs* q = defer ? defer : &one;
    if (q->n == 16) launch_bias_grad_batch(*q, s);
    q->job[q->n++] = BiasJob{a.partials_b, a.gbias, f.C, splits};
    if (!defer) launch_bias_grad_batch(one, s);
  }
}

// ------------------------------------------------------------------ f16x3 range guard (kernels.h)
__global__ __launch_bounds__(256) void channel_absmax_kernel(const float* __restrict__ t, int B, long HW, long sB, long sC, unsigned* __restrict__ chmax) {
  const int c = blockIdx.x;
  const long n = (long)B * HW;
  float m = 0.f;
  for (long j = (long)blockIdx.y * 256 + threadIdx.x; j < n; j += (long)gridDim.y * 256) {
    const long b = j / HW, i = j - b * HW;
    m = fmaxf(m, fabsf(t[b * sB + (long)c * sC + i]));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(chmax + c, __float_as_uint(m));
}
// one block: fold every thread's (max, smallest non-zero) pair, then thread 0 enters the spread (in bits: exponent of the
// maximum minus exponent of the smallest non-zero channel maximum, + 1 for the mantissas) into the largest seen so far OF ITS
// SIDE - *word = activation side (side 0: net input, gradOutput, BatchNorm pairs) | weight side (side 1) << 16.  Kernels of one
// stream run one after the other, so the read-modify-write needs no atomics.
__device__ __forceinline__ void spread_enter(float mx, float mn, unsigned* word, int side) {
  __shared__ float tmx[256], tmn[256];
  tmx[threadIdx.x] = mx; tmn[threadIdx.x] = mn;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 256; ++i) { mx = fmaxf(mx, tmx[i]); mn = fminf(mn, tmn[i]); }
    if (mx > 0.f && mn < mx) {
      const unsigned bits = (unsigned)min(ilogbf(mx) - ilogbf(mn) + 1, 0xffff);
      const unsigned w = *word;
      unsigned a = w & 0xffffu, b = w >> 16;
      if (side == 0) a = max(a, bits); else b = max(b, bits);
      *word = a | b << 16;
    }
  }
}
__global__ __launch_bounds__(256) void spread_verdict_kernel(unsigned* chmax, int C, unsigned* word, int side) {
  float mx = 0.f, mn = INFINITY;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float v = __uint_as_float(chmax[c]); chmax[c] = 0u;
    mx = fmaxf(mx, v); if (v > 0.f) mn = fminf(mn, v);
  }
  spread_enter(mx, mn, word, side);
}
__global__ __launch_bounds__(256) void pair_spread_kernel(const float* __restrict__ a, const float* __restrict__ b, int C, unsigned* word) {
  float mx = 0.f, mn = INFINITY;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float v = fmaxf(fabsf(a[c]), b ? fabsf(b[c]) : 0.f);
    mx = fmaxf(mx, v); if (v > 0.f) mn = fminf(mn, v);
  }
  spread_enter(mx, mn, word, 0);
}
void launch_channel_absmax(const float* t, int B, int C, long HW, long sB, long sC, unsigned* chmax, hipStream_t s) {
  const long n = (long)B * HW;
  int splits = (int)((n + 4095) / 4096);
  if (splits < 1) splits = 1;
  if ((long)splits * C > 8192) splits = (int)(8192 / C > 0 ? 8192 / C : 1);
  KtScope kt("range_guard_scan", 0.0, 4.0 * (double)n * C, s);
  hipLaunchKernelGGL(channel_absmax_kernel, dim3(C, splits), dim3(256), 0, s, t, B, HW, sB, sC, chmax);
}
void launch_spread_verdict(unsigned* chmax, int C, unsigned* word, int side, hipStream_t s) {
  hipLaunchKernelGGL(spread_verdict_kernel, dim3(1), dim3(256), 0, s, chmax, C, word, side);
}
void launch_pair_spread(const float* a, const float* b, int C, unsigned* sides, hipStream_t s) {
  hipLaunchKernelGGL(pair_spread_kernel, dim3(1), dim3(256), 0, s, a, b, C, sides);
}

// ------------------------------------------------------------------ nn.MSECriterion
__global__ __launch_bounds__(1024) void mse_kernel(const float* __restrict__ x, const float* __restrict__ t, long n, double inv_n,
                                                   float norm, double* loss, float* grad) {
  __shared__ double sh[16];
  double s = 0;
  for (long i = threadIdx.x; i < n; i += blockDim.x) {
    const float z = x[i] - t[i];
    s += (double)(z * z);
    if (grad) grad[i] = norm * z;
  }
  s = block_reduce_sum(s, sh);
  if (threadIdx.x == 0 && loss) *loss = s * inv_n;
}
void launch_mse(const float* x, const float* t, long n, long n_global, double* loss_dev, float* grad, hipStream_t s) {
  hipLaunchKernelGGL(mse_kernel, dim3(1), dim3(1024), 0, s, x, t, n, 1.0 / (double)n_global, (float)(2.0 / (double)n_global), loss_dev, grad);
}

// ------------------------------------------------------------------ R's head in ONE launch (gr_train_r_step; round 5)
// models.lua:446-451 + train_r.lua:147-151:  Linear(fc1) -> BatchNormalization -> act -> Dropout -> Linear(fc2) [-> Tanh] -> MSECriterion, forward AND backward
// down to the gradient wrt fc1's output.  Between fc1's GEMM and fc1's two backward GEMMs the step used to issue 14 kernels on tensors of B x 512 and B x nd
// floats - statistics partial / finalize / apply, a split-K GEMM and its reduction, the criterion, pass A / finalize of the last stage, two more small GEMMs,
// pass A / pass B of the BatchNorm stage: 79 us at cfg2, every one of them the 5 us floor of a dependent launch.  Here C1 / 8 workgroups (64) walk three
// phases separated by two grid barriers (all 64 are resident: the launch is alone on its stream and far below one workgroup per CU):
//   1  per 8-feature slice, all rows:  batch statistics (double sums, fixed order), running statistics, z = ((y - mean) invstd) gamma + beta, act, Dropout -> out1
//   2  per row slice, all features:    fc2 (+ bias, Tanh), loss partial, gradOutput, act' of fc2, gx = gy2 W2, times Dropout mask and act'(z) -> dz
//   3  per 8-feature slice, all rows:  BatchNorm backward (sums of dz and (y - mean) dz in double; grad gamma / beta; dy), bias gradient of fc1, weight and
//                                       bias gradient of fc2, max|dy| for the f16x3 GEMMs that follow
// Every value is formed by the operations of the kernels it replaces (bn_apply / act_fwd / act_bwd_z / mask_mul, the finalize kernels' double arithmetic);
// only the ORDER of the sums differs (rows in 32 interleaved groups instead of 256 threads x splits; fp32 dot products by fmaf in a fixed order), which is
// inside the 1e-4 bar every parity test holds the step to.  Deterministic: no floating-point atomics, every reduction in a fixed order.
// LLVM sinks a load into the (conditional) block of its only use and schedules for occupancy: a batch of independent loads written before a loop comes out as
// load - wait - use, one at a time - fatal in a kernel of 256 waves where every wait is a full memory latency.  An empty asm that names the batch as inputs
// pins every load of it above that point: one wait for all.
#define HEAD_KEEP8(a_, i_) asm volatile("" :: "v"((a_)[(i_)]), "v"((a_)[(i_) + 1]), "v"((a_)[(i_) + 2]), "v"((a_)[(i_) + 3]), "v"((a_)[(i_) + 4]), "v"((a_)[(i_) + 5]), "v"((a_)[(i_) + 6]), "v"((a_)[(i_) + 7]))
#define HEAD_KEEP16(a_, i_) asm volatile("" :: "v"((a_)[(i_)]), "v"((a_)[(i_) + 1]), "v"((a_)[(i_) + 2]), "v"((a_)[(i_) + 3]), "v"((a_)[(i_) + 4]), "v"((a_)[(i_) + 5]), "v"((a_)[(i_) + 6]), "v"((a_)[(i_) + 7]), \
                                              "v"((a_)[(i_) + 8]), "v"((a_)[(i_) + 9]), "v"((a_)[(i_) + 10]), "v"((a_)[(i_) + 11]), "v"((a_)[(i_) + 12]), "v"((a_)[(i_) + 13]), "v"((a_)[(i_) + 14]), "v"((a_)[(i_) + 15]))
// Dropout keep flag of element e as a multiplier: the launcher passes MASK_ELEM with its bits, or MASK_NONE with bits pointing at ANY readable words (the load
// stays unconditional: no branch, no sinking) - what mask_mul computes for these two kinds
__device__ __forceinline__ unsigned head_mask_word(const MaskRef& m, long e) { return m.bits[e >> 5]; }
__device__ __forceinline__ float head_mask_of(const MaskRef& m, unsigned w, long e) { const float k = ((w >> (e & 31)) & 1u) ? m.scale : 0.f; return m.kind == MASK_ELEM ? k : 1.f; }
struct HeadArgs {
  int B, C1, nd, rows_per_wg;
  const float* y1; float* out1;
  float* mean; float* invstd; float* run_mean; float* run_var; const float* gamma; const float* beta;
  MaskRef m1; int act1; float slope1; int act2;
  const float* W2; const float* b2; float* y2; float* out2;
  const float* target; double inv_n; float norm; double* loss; double* loss_part;
  float* gout; float* gy2; float* dy1;
  float* gW2; float* gb2; float* ggamma; float* gbeta; float* gb1;
  unsigned* amax_dy;
  unsigned* bar; unsigned bar_base;
  unsigned* fault; int spin_limit; // sticky device word a timed-out barrier sets (penalty_clamp_adam_kernel skips its update while it is set; the host turns it into GR_ERR_STATE)
  unsigned long long* stamps;      // ablation build: [workgroup][8] wall-clock stamps of the phases (tools/debug/debug_head.py)
};
constexpr int HEAD_FW = 8, HEAD_RG = 32, HEAD_RMAX = 16;
// Bounded: a launch that cannot become resident as a whole (a partitioned device, CUs held by other work for seconds) does not hang the GPU.  ANY workgroup
// that gives up sets the sticky fault word: the phases after it then run on incomplete data, so the step's update must not happen - penalty_clamp_adam_kernel
// reads the word and leaves theta / g / m / v untouched, and the host reports GR_ERR_STATE from the next call that synchronises (net.hip head_fault_check).
// Every workgroup still adds its arrival to the counter, so the counter stays in step with the host's running base whatever happened.
__device__ __forceinline__ bool head_grid_barrier(unsigned* ctr, unsigned target, unsigned* fault, int spin_limit) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  __shared__ int ok_;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int ok = 0;
    for (int spin = 0; spin < spin_limit; ++spin) {
      if ((int)(__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) { ok = 1; break; }
      __builtin_amdgcn_s_sleep(4);
    }
    if (!ok) __hip_atomic_fetch_or(fault, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    ok_ = ok;
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return ok_ != 0;
}
template <int RT>      // rows per workgroup in phase 2 (4: cfg2's 256 rows over 64 workgroups, 8: cfg3's 512, 16: the largest covered)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void head_fwd_bwd_kernel(HeadArgs a) {      // (one wave per SIMD: the scheduler may keep 64 loads in flight instead of trading them for an occupancy nobody uses)
  extern __shared__ __attribute__((aligned(16))) unsigned char head_smem[];
  __shared__ double sh_s[HEAD_RG][HEAD_FW], sh_q[HEAD_RG][HEAD_FW];
  __shared__ float sh_mean[HEAD_FW], sh_inv[HEAD_FW], sh_c0[HEAD_FW], sh_c1[HEAD_FW];
  __shared__ double sh_loss8[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x, NW = gridDim.x;
  const int B = a.B, C1 = a.C1, nd = a.nd;
  const double n = (double)B;
  bool alive = true;
#define HEAD_STAMP(i_) if (GR_DBG(a.stamps != nullptr) && tid == 0) a.stamps[blockIdx.x * 8 + (i_)] = wall_clock64();
  HEAD_STAMP(0)
  // Loops below run a FIXED number of steps with clamped indices and zero weights past the ends (rows past B, outputs past nd, columns past C1) and guard
  // only their stores: the first version predicated every step of 16-fold unrollings and came to 35 000 lines of ISA, slower than the 14 launches it replaced.
  // ---------------------------------------------------------------- phase 1: features f0 .. f0 + 7, all rows
  const int f0 = wg * HEAD_FW, ff = tid & (HEAD_FW - 1), rg = tid >> 3, f = f0 + ff;
  {
    double s = 0, q = 0;
    for (int b0 = rg; b0 < B; b0 += 8 * HEAD_RG) {               // eight loads in flight per thread, added in row order
      float v8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v8[u] = a.y1[(long)min(b0 + u * HEAD_RG, B - 1) * C1 + f];
      HEAD_KEEP8(v8, 0);
#pragma unroll
      for (int u = 0; u < 8; ++u) { const double v = b0 + u * HEAD_RG < B ? v8[u] : 0.f; s += v; q += v * v; }
    }
    sh_s[rg][ff] = s; sh_q[rg][ff] = q;
    __syncthreads();
    if (tid < HEAD_FW) {
      double ss = 0, qq = 0;
#pragma unroll 8
      for (int k = 0; k < HEAD_RG; ++k) { ss += sh_s[k][tid]; qq += sh_q[k][tid]; }
      const double m = ss / n;
      double vs = qq - ss * m;
      if (vs < 0) vs = 0;
      const float mf = (float)m, is = (float)(1.0 / sqrt(vs / n + 1e-5));
      const int c = f0 + tid;
      a.mean[c] = mf; a.invstd[c] = is;
      if (a.run_mean) {
        a.run_mean[c] = (float)(0.1 * m + 0.9 * (double)a.run_mean[c]);
        a.run_var[c] = (float)(0.1 * (vs / (n - 1)) + 0.9 * (double)a.run_var[c]);
      }
      sh_mean[tid] = mf; sh_inv[tid] = is;
    }
    __syncthreads();
    const float mean = sh_mean[ff], invstd = sh_inv[ff], gm = a.gamma[f], bt = a.beta[f];
    for (int b0 = rg; b0 < B; b0 += 8 * HEAD_RG) {
      float v8[8]; unsigned m8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const long e = (long)min(b0 + u * HEAD_RG, B - 1) * C1 + f; v8[u] = a.y1[e]; m8[u] = head_mask_word(a.m1, e); }
      HEAD_KEEP8(v8, 0); HEAD_KEEP8(m8, 0);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int b = b0 + u * HEAD_RG;
        const long e = (long)min(b, B - 1) * C1 + f;
        const float z = ((v8[u] - mean) * invstd) * gm + bt;
        const float o = act_fwd(z, a.act1, a.slope1) * head_mask_of(a.m1, m8[u], e);
        if (b < B) a.out1[(long)b * C1 + f] = o;
      }
    }
  }
  HEAD_STAMP(1)
  alive = head_grid_barrier(a.bar, a.bar_base + (unsigned)NW, a.fault, a.spin_limit) && alive;
  HEAD_STAMP(2)
  // ---------------------------------------------------------------- phase 2: rows r0 .. r0 + nrows - 1, all features
  // Every global load of the phase that does not depend on its own arithmetic goes out at its top - the slice's out1 rows, the first eight W2 rows of each
  // wave, fc2's bias, the criterion's targets, y1 / Dropout words of the slice - and is waited for ONCE; later batches (more than 32 outputs) are requested one
  // batch ahead of their use.  (256 waves on the chip: a load that waits alone costs a whole memory latency, ~2 us behind the barrier's cache invalidate.)
  const int r0 = wg * RT, nrows = max(0, min(RT, B - r0));
  float* xs = reinterpret_cast<float*>(head_smem);                 // [RT][C1]: out1 rows (phase 2), then [B][8] out1 columns (phase 3)
  float* gy2_s = xs + (size_t)(RT * C1 > B * HEAD_FW ? RT * C1 : B * HEAD_FW);      // [RT][nd]
  float* red = gy2_s + (size_t)RT * nd;                            // phase 3: [slices][OG][9]; phase 2: fc2's bias [nd]
  float* b2_s = red;
  double lacc = 0;
  {
    const int NJ = C1 >> 6;                                        // C1 % 64 == 0, C1 <= 512 (launcher): at most 8 columns per lane
    const long last1 = (long)B * C1 - 1, last2 = (long)B * nd - 1;
    float xr[2 * RT];                                              // RT * C1 / 256 <= 2 RT elements of the slice's rows per thread
#pragma unroll
    for (int k = 0; k < 2 * RT; ++k) { const long i = (long)r0 * C1 + tid + 256 * k; xr[k] = a.out1[i < last1 ? i : last1]; }
    float w8[8][8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) w8[u][j] = a.W2[(long)min(wave + 4 * u, nd - 1) * C1 + lane + 64 * (j < NJ ? j : 0)];
    const float b2r = a.b2[min(tid, nd - 1)];
    float tg[(RT * 128 + 255) / 256];                              // the criterion's targets of this thread's elements (nd <= 128)
#pragma unroll
    for (int k = 0; k < (RT * 128 + 255) / 256; ++k) { const long e2 = (long)r0 * nd + tid + 256 * k; tg[k] = a.target[e2 < last2 ? e2 : last2]; }
    float yv[2][RT]; unsigned mk[2][RT];                           // y1 and Dropout words of the thread's two feature columns (fx = tid, tid + 256)
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
      for (int r = 0; r < RT; ++r) { const long e = (long)min(r0 + r, B - 1) * C1 + min(tid + 256 * h2, C1 - 1); yv[h2][r] = a.y1[e]; mk[h2][r] = head_mask_word(a.m1, e); }
    float bnp[2][4];
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) { const int fx = min(tid + 256 * h2, C1 - 1); bnp[h2][0] = a.mean[fx]; bnp[h2][1] = a.invstd[fx]; bnp[h2][2] = a.gamma[fx]; bnp[h2][3] = a.beta[fx]; }
    // ---- one wait
#pragma unroll
    for (int k = 0; k < 2 * RT; k += 4) asm volatile("" :: "v"(xr[k]), "v"(xr[k + 1]), "v"(xr[k + 2]), "v"(xr[k + 3]));
#pragma unroll
    for (int u = 0; u < 8; ++u) HEAD_KEEP8(w8[u], 0);
    asm volatile("" :: "v"(b2r), "v"(tg[0]), "v"(bnp[0][0]), "v"(bnp[0][1]), "v"(bnp[0][2]), "v"(bnp[0][3]), "v"(bnp[1][0]), "v"(bnp[1][1]), "v"(bnp[1][2]), "v"(bnp[1][3]));
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
      for (int r = 0; r < RT; r += 4) asm volatile("" :: "v"(yv[h2][r]), "v"(yv[h2][r + 1]), "v"(yv[h2][r + 2]), "v"(yv[h2][r + 3]), "v"(mk[h2][r]), "v"(mk[h2][r + 1]), "v"(mk[h2][r + 2]), "v"(mk[h2][r + 3]));
#pragma unroll
    for (int k = 0; k < 2 * RT; ++k) { const int i = tid + 256 * k; if (i < RT * C1) xs[i] = (long)r0 * C1 + i <= last1 ? xr[k] : 0.f; }
    if (tid < nd) b2_s[tid] = b2r;
    __syncthreads();
    for (int ob = wave; ob < nd; ob += 32) {                       // this wave's outputs ob, ob + 4, ..., eight at a time
      float wn[8][8];                                              // the next batch, requested before this one is multiplied
      const bool more = ob + 32 < nd;
      if (more) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int j = 0; j < 8; ++j) wn[u][j] = a.W2[(long)min(ob + 32 + 4 * u, nd - 1) * C1 + lane + 64 * (j < NJ ? j : 0)];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int o = ob + 4 * u;
        const float bias2 = b2_s[min(o, nd - 1)];
#pragma unroll
        for (int r = 0; r < RT; ++r) {
          float pr = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) pr = fmaf((o < nd && j < NJ) ? w8[u][j] : 0.f, xs[r * C1 + lane + 64 * (j < NJ ? j : 0)], pr);
          const float v = wave_sum(pr);                            // DPP tree, fixed order; the total sits in lanes 48-63
          if (lane == 63 && o < nd) gy2_s[r * nd + o] = v + bias2;   // fc2's raw output; the criterion follows below, one element per thread
        }
      }
      if (more) {
#pragma unroll
        for (int u = 0; u < 8; ++u) HEAD_KEEP8(wn[u], 0);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int j = 0; j < 8; ++j) w8[u][j] = wn[u][j];
      }
    }
    // W2 columns of the thread's two features for the data gradient: the first 32 outputs now, behind the criterion
    float wc[2][32];
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
      for (int u = 0; u < 32; ++u) wc[h2][u] = a.W2[(long)min(u, nd - 1) * C1 + min(tid + 256 * h2, C1 - 1)];
    __syncthreads();
    // criterion and its gradient, element-wise over the slice's nrows x nd outputs (a wave's lane 63 doing this after every dot product was a chain of
    // dependent global loads and stores: 80 us of the kernel's first version)
#pragma unroll
    for (int k = 0; k < (RT * 128 + 255) / 256; ++k) {
      const int i = tid + 256 * k;
      if (i < nrows * nd) {
        const long e2 = (long)r0 * nd + i;
        const float y = gy2_s[i];
        const float out = act_fwd(y, a.act2, 0.f);
        const float zd = out - tg[k];
        lacc += (double)(zd * zd);
        const float g = a.norm * zd;
        const float g2 = act_bwd(g, y, out, a.act2, 0.f);
        a.y2[e2] = y;
        if (a.out2 != a.y2) a.out2[e2] = out;
        a.gout[e2] = g;
        a.gy2[e2] = g2;
        gy2_s[i] = g2;
      } else if (i < RT * nd) gy2_s[i] = 0.f;                     // rows past B: zero gradient
    }
    lacc = block_reduce_sum(lacc, sh_loss8);
    HEAD_STAMP(3)
    __syncthreads();
    if (tid == 0) a.loss_part[wg] = lacc;
    // gx = gy2 W2 for the rows of this slice, then through the Dropout mask and the activation's derivative: dz
    float acc[2][RT];
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
      for (int r = 0; r < RT; ++r) acc[h2][r] = 0.f;
    for (int o0 = 0; o0 < nd; o0 += 32) {
      float wcn[2][32];
      const bool more = o0 + 32 < nd;
      if (more) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
          for (int u = 0; u < 32; ++u) wcn[h2][u] = a.W2[(long)min(o0 + 32 + u, nd - 1) * C1 + min(tid + 256 * h2, C1 - 1)];
      } else {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) { HEAD_KEEP16(wc[h2], 0); HEAD_KEEP16(wc[h2], 16); }
      }
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        const int oo = min(o0 + u, nd - 1);
        const bool on = o0 + u < nd;
#pragma unroll
        for (int r = 0; r < RT; ++r) {
          const float gv = gy2_s[r * nd + oo];
          acc[0][r] = fmaf(gv, on ? wc[0][u] : 0.f, acc[0][r]);
          acc[1][r] = fmaf(gv, on ? wc[1][u] : 0.f, acc[1][r]);
        }
      }
      if (more) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) { HEAD_KEEP16(wcn[h2], 0); HEAD_KEEP16(wcn[h2], 16); }
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
          for (int u = 0; u < 32; ++u) wc[h2][u] = wcn[h2][u];
      }
    }
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int fx = tid + 256 * h2;
      const float mean = bnp[h2][0], invstd = bnp[h2][1], gm = bnp[h2][2], bt = bnp[h2][3];
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        const float g = acc[h2][r] * head_mask_of(a.m1, mk[h2][r], (long)min(r0 + r, B - 1) * C1 + min(fx, C1 - 1));
        const float z = ((yv[h2][r] - mean) * invstd) * gm + bt;
        const float dz = act_bwd_z(g, z, a.act1, a.slope1);
        if (r < nrows && fx < C1) a.dy1[(long)(r0 + r) * C1 + fx] = dz;
      }
    }
  }
  HEAD_STAMP(4)
  alive = head_grid_barrier(a.bar, a.bar_base + 2u * (unsigned)NW, a.fault, a.spin_limit) && alive;
  HEAD_STAMP(5)
  // ---------------------------------------------------------------- phase 3: features f0 .. f0 + 7, all rows
  {
    const float mean = sh_mean[ff], invstd = sh_inv[ff], gm = a.gamma[f];
    double s = 0, q = 0;
    for (int b0 = rg; b0 < B; b0 += 8 * HEAD_RG) {
      float d8[8], y8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const long e = (long)min(b0 + u * HEAD_RG, B - 1) * C1 + f; d8[u] = a.dy1[e]; y8[u] = a.y1[e]; }
      HEAD_KEEP8(d8, 0); HEAD_KEEP8(y8, 0);
#pragma unroll
      for (int u = 0; u < 8; ++u) { const float dz = b0 + u * HEAD_RG < B ? d8[u] : 0.f; s += (double)dz; q += (double)(y8[u] - mean) * (double)dz; }
    }
    __syncthreads();                                               // (sh_s / sh_q of phase 1 are long consumed; xs / gy2_s of phase 2 too)
    sh_s[rg][ff] = s; sh_q[rg][ff] = q;
    __syncthreads();
    if (tid < HEAD_FW) {
      double ss = 0, qq = 0;
#pragma unroll 8
      for (int k = 0; k < HEAD_RG; ++k) { ss += sh_s[k][tid]; qq += sh_q[k][tid]; }
      const double isd = (double)sh_inv[tid];
      const int c = f0 + tid;
      a.ggamma[c] += (float)(qq * isd);
      a.gbeta[c] += (float)ss;
      sh_c0[tid] = (float)(ss / n);
      sh_c1[tid] = (float)(qq * isd * isd / n);
    }
    __syncthreads();
    const float c0 = sh_c0[ff], c1 = sh_c1[ff];
    double sb = 0; float dmax = 0.f;
    for (int b0 = rg; b0 < B; b0 += 8 * HEAD_RG) {
      float d8[8], y8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const long e = (long)min(b0 + u * HEAD_RG, B - 1) * C1 + f; d8[u] = a.dy1[e]; y8[u] = a.y1[e]; }
      HEAD_KEEP8(d8, 0); HEAD_KEEP8(y8, 0);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int b = b0 + u * HEAD_RG;
        const float d = ((d8[u] - c0) - (y8[u] - mean) * c1) * invstd * gm;
        if (b < B) { a.dy1[(long)b * C1 + f] = d; dmax = fmaxf(dmax, fabsf(d)); sb += (double)d; }
      }
    }
    if (a.amax_dy) absmax_commit(dmax, a.amax_dy);
    sh_s[rg][ff] = sb;
    // out1 columns of this slice -> LDS for fc2's weight gradient
    for (int i = tid; i < B * HEAD_FW; i += 256) { const int b = i >> 3, c = i & 7; xs[i] = a.out1[(long)b * C1 + f0 + c]; }
    __syncthreads();
    if (tid < HEAD_FW) {
      double t = 0;
#pragma unroll 8
      for (int k = 0; k < HEAD_RG; ++k) t += sh_s[k][tid];
      a.gb1[f0 + tid] += (float)t;
    }
    // gW2[o][f0 + c] += sum_b gy2[b][o] out1[b][f0 + c]; gb2[o] += sum_b gy2[b][o] (workgroup 0)
    const int OG = nd <= 32 ? 32 : (nd <= 64 ? 64 : 128), slices = 256 / OG, o = tid & (OG - 1), sl = tid / OG, oc = min(o, nd - 1);
    float acc[HEAD_FW + 1];
#pragma unroll
    for (int c = 0; c <= HEAD_FW; ++c) acc[c] = 0.f;
    for (int b0 = sl; b0 < B; b0 += 16 * slices) {
      float g16[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) g16[u] = a.gy2[(long)min(b0 + u * slices, B - 1) * nd + oc];
      HEAD_KEEP16(g16, 0);
#pragma unroll
      for (int u = 0; u < 16; ++u) g16[u] = b0 + u * slices < B ? g16[u] : 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int b = min(b0 + u * slices, B - 1);
        const float g = g16[u];
        const float4 x0 = *reinterpret_cast<const float4*>(xs + b * HEAD_FW), x1 = *reinterpret_cast<const float4*>(xs + b * HEAD_FW + 4);
        acc[0] = fmaf(g, x0.x, acc[0]); acc[1] = fmaf(g, x0.y, acc[1]); acc[2] = fmaf(g, x0.z, acc[2]); acc[3] = fmaf(g, x0.w, acc[3]);
        acc[4] = fmaf(g, x1.x, acc[4]); acc[5] = fmaf(g, x1.y, acc[5]); acc[6] = fmaf(g, x1.z, acc[6]); acc[7] = fmaf(g, x1.w, acc[7]);
        acc[8] += g;
      }
    }
#pragma unroll
    for (int c = 0; c <= HEAD_FW; ++c) red[(sl * OG + o) * (HEAD_FW + 1) + c] = acc[c];
    __syncthreads();
    for (int i = tid; i < nd * (HEAD_FW + 1); i += 256) {
      const int oo = i / (HEAD_FW + 1), c = i - oo * (HEAD_FW + 1);
      float t = 0.f;
      for (int k = 0; k < slices; ++k) t += red[(k * OG + oo) * (HEAD_FW + 1) + c];
      if (c < HEAD_FW) a.gW2[(long)oo * C1 + f0 + c] += t;
      else if (wg == 0) a.gb2[oo] += t;
    }
    HEAD_STAMP(6)
    if (wg == 0 && tid == 0) {
      double t = 0;
      for (int k = 0; k < NW; ++k) t += a.loss_part[k];
      // (another workgroup may still time out after this one has passed: the sticky word, not this NaN, is what the host and the optimiser go by)
      const bool faulted = !alive || __hip_atomic_load(a.fault, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0u;
      *a.loss = faulted ? (double)NAN : t * a.inv_n;
    }
  }
}
size_t head_lds_bytes(int B, int C1, int nd, int rows_per_wg) {
  const size_t xs = (size_t)(rows_per_wg * C1 > B * HEAD_FW ? rows_per_wg * C1 : B * HEAD_FW);
  return sizeof(float) * (xs + (size_t)rows_per_wg * nd + (size_t)256 * (HEAD_FW + 1));
}
bool head_supported(int B, int C1, int nd) {
  if (B < 2 || C1 % 64 != 0 || C1 < 64 || C1 > 512 || nd < 1 || nd > 128) return false;
  const int NW = C1 / HEAD_FW, R = (B + NW - 1) / NW;
  // Measured (in-kernel stamps, tools/debug/head_stamps.py): cfg2's head (256 rows, nd 32: 4 rows per workgroup) 48.6 us against 79 us for the 14 launches it
  // replaces; cfg3's (512 rows, nd 100: 8 rows per workgroup, four batches of W2 per wave) 141 us against ~95 - its fc2 forward is a chain of 256 dependent
  // (LDS read, 8 FMAs, DPP wave sum) steps per wave.  The kernel is used where it wins: at most 4 rows per workgroup and nd <= 32.
  return R <= 4 && nd <= 32 && head_lds_bytes(B, C1, nd, 4) <= 60 * 1024;
}
void launch_head_fwd_bwd(const HeadLaunch& h, hipStream_t s) {
  HeadArgs a{};
  a.B = h.B; a.C1 = h.C1; a.nd = h.nd;
  const int NW = h.C1 / HEAD_FW;
  a.rows_per_wg = (h.B + NW - 1) / NW;
  a.y1 = h.y1; a.out1 = h.out1; a.mean = h.mean; a.invstd = h.invstd; a.run_mean = h.run_mean; a.run_var = h.run_var; a.gamma = h.gamma; a.beta = h.beta;
  a.m1 = h.m1; if (a.m1.kind != MASK_ELEM) { a.m1.kind = MASK_NONE; a.m1.bits = reinterpret_cast<const uint32_t*>(h.y1); a.m1.scale = 1.f; }      // (any readable words: head_mask_word)
  a.act1 = h.act1; a.slope1 = h.slope1; a.act2 = h.act2;
  a.W2 = h.W2; a.b2 = h.b2; a.y2 = h.y2; a.out2 = h.out2;
  a.target = h.target; a.inv_n = 1.0 / (double)h.n_global; a.norm = (float)(2.0 / (double)h.n_global); a.loss = h.loss; a.loss_part = h.loss_part;
  a.gout = h.gout; a.gy2 = h.gy2; a.dy1 = h.dy1;
  a.gW2 = h.gW2; a.gb2 = h.gb2; a.ggamma = h.ggamma; a.gbeta = h.gbeta; a.gb1 = h.gb1;
  a.amax_dy = h.amax_dy; a.bar = h.bar; a.bar_base = h.bar_base; a.fault = h.fault; a.spin_limit = h.spin_limit > 0 ? h.spin_limit : (1 << 22); a.stamps = reinterpret_cast<unsigned long long*>(g_p16_stamps);
  const size_t lds = head_lds_bytes(h.B, h.C1, h.nd, 4);      // (head_supported: at most 4 rows per workgroup)
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&head_fwd_bwd_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); attr = true; }
  KtScope kt("head_fwd_bwd_kernel", 2.0 * 3.0 * h.B * (double)h.C1 * h.nd, 4.0 * (6.0 * h.B * h.C1 + 3.0 * (double)h.nd * h.C1), s);
  hipLaunchKernelGGL(head_fwd_bwd_kernel<4>, dim3(NW), dim3(256), lds, s, a);
}

// ------------------------------------------------------------------ nn.BCECriterion (sizeAverage; train.lua:173's CRITERION, used by adversarial.lua)
// THNN BCECriterion.c with EPS = 1e-12: every term in double (the C source mixes float tensors with double literals), the sum in
// double; gradInput = -1/n (t - x) / ((1 - x + EPS)(x + EPS)) evaluated in double and rounded once - the same IEEE operations as
// the oracle, so the gradient is bit-identical; the loss differs by the device's log() (last bits).
__global__ __launch_bounds__(1024) void bce_kernel(const float* __restrict__ x, const float* __restrict__ t, long n, double* loss, float* grad) {
  __shared__ double sh[16];
  const double EPS = 1e-12, norm = 1.0 / (double)n;
  double s = 0;
  for (long i = threadIdx.x; i < n; i += blockDim.x) {
    const double xv = (double)x[i], tv = (double)t[i];
    s -= log(xv + EPS) * tv + log(1. - xv + EPS) * (1. - tv);
    if (grad) grad[i] = (float)(-norm * (tv - xv) / ((1. - xv + EPS) * (xv + EPS)));
  }
  s = block_reduce_sum(s, sh);
  if (threadIdx.x == 0 && loss) *loss = s * norm;
}
void launch_bce(const float* x, const float* t, long n, double* loss_dev, float* grad, hipStream_t s) {
  hipLaunchKernelGGL(bce_kernel, dim3(1), dim3(1024), 0, s, x, t, n, loss_dev, grad);
}

// y += x : nn.Concat:updateGradInput sums the gradInputs of its branches (models.lua:293-321, device-resident GAN step)
__global__ void add_inplace_kernel(float* __restrict__ y, const float* __restrict__ x, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = y[i] + x[i];
}
void launch_add_inplace(float* y, const float* x, long n, hipStream_t s) {
  if (n <= 0) return;
  const long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, s, y, x, n);
}

// ------------------------------------------------------------------ penalty + clamp + Adam, one pass over (theta, g, m, v)
__device__ __forceinline__ void adam_one(float& th, float& gv, float& mv, float& vv, const AdamConsts& c) {
  if (c.use_penalty) {
    const float sg = th > 0.f ? 1.f : (th < 0.f ? -1.f : 0.f);
    const float pen = sg * c.l1 + th * c.l2;
    gv = gv + pen;
  }
  if (c.use_clamp) gv = gv < -c.clamp ? -c.clamp : (gv > c.clamp ? c.clamp : gv);
  mv = mv * c.b1 + c.c1 * gv;
  vv = vv * c.b2 + (c.c2 * gv) * gv;
  const float denom = sqrtf(vv) + c.eps;
  th = th + (c.step * mv) / denom;
}
// four entries per thread as one 16-byte access per array (the flat vectors are hipMalloc'ed: 256-byte aligned); the last
// n % 4 entries go through the scalar path of the thread that would own the next vector
__global__ __launch_bounds__(256) void penalty_clamp_adam_kernel(float* __restrict__ theta, float* __restrict__ g, float* __restrict__ m,
                                                                 float* __restrict__ v, long n, AdamConsts c, const unsigned* __restrict__ skip) {
  // skip: the context's sticky fault word (a grid barrier of head_fwd_bwd_kernel timed out: the gradients of that step are incomplete).  A uniform scalar
  // load; while it is set no step changes theta, g, m or v - the host reports GR_ERR_STATE at its next synchronising call and clears it.
  if (skip != nullptr && *skip != 0u) return;
  const long n4 = n >> 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i <= n4; i += (long)gridDim.x * blockDim.x) {
    if (i < n4) {
      float4 th = reinterpret_cast<float4*>(theta)[i], gv = reinterpret_cast<float4*>(g)[i];
      float4 mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
      adam_one(th.x, gv.x, mv.x, vv.x, c); adam_one(th.y, gv.y, mv.y, vv.y, c);
      adam_one(th.z, gv.z, mv.z, vv.z, c); adam_one(th.w, gv.w, mv.w, vv.w, c);
      reinterpret_cast<float4*>(theta)[i] = th; reinterpret_cast<float4*>(g)[i] = gv;
      reinterpret_cast<float4*>(m)[i] = mv; reinterpret_cast<float4*>(v)[i] = vv;
    } else {
      for (long j = n4 << 2; j < n; ++j) {
        float th = theta[j], gv = g[j], mv = m[j], vv = v[j];
        adam_one(th, gv, mv, vv, c);
        theta[j] = th; g[j] = gv; m[j] = mv; v[j] = vv;
      }
    }
  }
}
void launch_penalty_clamp_adam(float* theta, float* g, float* m, float* v, long n, const AdamConsts& c, hipStream_t s, const unsigned* skip) {
  long blocks = ((n >> 2) + 1 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  KtScope kt("penalty_clamp_adam_kernel", 0.0, 32.0 * (double)n, s);   // read theta,g,m,v + write theta,g,m,v
  hipLaunchKernelGGL(penalty_clamp_adam_kernel, dim3((unsigned)blocks), dim3(256), 0, s, theta, g, m, v, n, c, skip);
}

// ------------------------------------------------------------------ counter-based RNG (Philox4x32-10)
struct u4 { uint32_t x, y, z, w; };
__device__ __forceinline__ u4 philox4x32(u4 ctr, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * ctr.x, p1 = (uint64_t)0xCD9E8D57u * ctr.z;
    u4 n;
    n.x = (uint32_t)(p1 >> 32) ^ ctr.y ^ k0; n.y = (uint32_t)p1;
    n.z = (uint32_t)(p0 >> 32) ^ ctr.w ^ k1; n.w = (uint32_t)p0;
    ctr = n; k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return ctr;
}

// keep-bit = Bernoulli(1-p).  p == 0.5: one Philox call yields 128 keep bits; otherwise one 32-bit uniform per element.
__global__ void gen_mask_kernel(uint32_t* words, long nwords, uint32_t thresh, int half, uint32_t s0, uint32_t s1,
                                uint32_t c0, uint32_t c1, uint32_t layer) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (half) {
    if (i * 4 >= nwords) return;
    const u4 r = philox4x32(u4{(uint32_t)i, layer, c0, c1}, s0, s1);
    const uint32_t o[4] = {r.x, r.y, r.z, r.w};
    for (int k = 0; k < 4; ++k) if (i * 4 + k < nwords) words[i * 4 + k] = o[k];
  } else {
    if (i >= nwords) return;
    uint32_t w = 0;
    for (int j = 0; j < 8; ++j) {
      const u4 r = philox4x32(u4{(uint32_t)i, (uint32_t)j | (layer << 8), c0, c1}, s0, s1);
      const uint32_t o[4] = {r.x, r.y, r.z, r.w};
      for (int k = 0; k < 4; ++k) w |= (o[k] >= thresh ? 1u : 0u) << (j * 4 + k);
    }
    words[i] = w;
  }
}
// every Dropout / SpatialDropout mask of one forward in ONE launch (blockIdx.y = job): the masks depend on (seed, forward
// counter, layer, element) only, so they can all be drawn before the first layer runs.  Same Philox indexing as gen_mask_kernel.
__global__ void gen_mask_batch_kernel(MaskJobs jobs, uint32_t s0, uint32_t s1, uint32_t c0, uint32_t c1) {
  const MaskJob j = jobs.job[blockIdx.y];
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (j.half) {
    if (i * 4 >= j.nwords) return;
    const u4 r = philox4x32(u4{(uint32_t)i, j.layer, c0, c1}, s0, s1);
    const uint32_t o[4] = {r.x, r.y, r.z, r.w};
    for (int k = 0; k < 4; ++k) if (i * 4 + k < j.nwords) j.words[i * 4 + k] = o[k];
  } else {
    if (i >= j.nwords) return;
    uint32_t w = 0;
    for (int jj = 0; jj < 8; ++jj) {
      const u4 r = philox4x32(u4{(uint32_t)i, (uint32_t)jj | (j.layer << 8), c0, c1}, s0, s1);
      const uint32_t o[4] = {r.x, r.y, r.z, r.w};
      for (int k = 0; k < 4; ++k) w |= (o[k] >= j.thresh ? 1u : 0u) << (jj * 4 + k);
    }
    j.words[i] = w;
  }
}
MaskJob make_mask_job(uint32_t* words, long n_elems, float p_drop, uint32_t layer) {
  MaskJob j{};
  j.words = words; j.nwords = (n_elems + 31) / 32; j.half = p_drop == 0.5f ? 1 : 0; j.layer = layer;
  j.thresh = (uint32_t)fmin(4294967295.0, (double)p_drop * 4294967296.0);
  return j;
}
void launch_gen_mask_batch(const MaskJobs& jobs, uint64_t seed, uint64_t counter, hipStream_t s) {
  if (jobs.n <= 0) return;
  long threads = 1;
  for (int i = 0; i < jobs.n; ++i) { const long t = jobs.job[i].half ? (jobs.job[i].nwords + 3) / 4 : jobs.job[i].nwords; if (t > threads) threads = t; }
  KtScope kt("gen_mask_batch_kernel", 0.0, 0.0, s);
  hipLaunchKernelGGL(gen_mask_batch_kernel, dim3((unsigned)((threads + 255) / 256), jobs.n), dim3(256), 0, s, jobs,
                     (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)counter, (uint32_t)(counter >> 32));
}
void launch_gen_mask(uint32_t* words, long n_elems, float p_drop, uint64_t seed, uint64_t counter, uint32_t layer, hipStream_t s) {
  const long nwords = (n_elems + 31) / 32;
  const int half = p_drop == 0.5f;
  const long threads = half ? (nwords + 3) / 4 : nwords;
  const uint32_t thresh = (uint32_t)fmin(4294967295.0, (double)p_drop * 4294967296.0);
  hipLaunchKernelGGL(gen_mask_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, words, nwords, thresh, half,
                     (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)counter, (uint32_t)(counter >> 32), layer);
}

__global__ void pack_mask_kernel(const uint8_t* keep, uint32_t* words, long n) {
  const long wi = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (wi * 32 >= n) return;
  uint32_t w = 0;
  for (int k = 0; k < 32; ++k) { const long e = wi * 32 + k; if (e < n && keep[e]) w |= 1u << k; }
  words[wi] = w;
}
__global__ void unpack_mask_kernel(const uint32_t* words, uint8_t* keep, long n) {
  const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (e < n) keep[e] = (words[e >> 5] >> (e & 31)) & 1u;
}
void launch_pack_mask(const uint8_t* keep, uint32_t* words, long n, hipStream_t s) {
  const long nw = (n + 31) / 32;
  hipLaunchKernelGGL(pack_mask_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, s, keep, words, n);
}
void launch_unpack_mask(const uint32_t* words, uint8_t* keep, long n, hipStream_t s) {
  hipLaunchKernelGGL(unpack_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, words, keep, n);
}

__global__ void fill_normal_kernel(float* dst, long n, uint32_t s0, uint32_t s1) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i * 4 >= n) return;
  const u4 r = philox4x32(u4{(uint32_t)i, (uint32_t)(i >> 32), 0x6e6f6973u, 0u}, s0, s1);
  const float u1 = ((float)(r.x >> 8) + 1.f) * (1.f / 16777216.f), u2 = (float)(r.y >> 8) * (1.f / 16777216.f);
  const float u3 = ((float)(r.z >> 8) + 1.f) * (1.f / 16777216.f), u4_ = (float)(r.w >> 8) * (1.f / 16777216.f);
  const float ra = sqrtf(-2.f * logf(u1)), rb = sqrtf(-2.f * logf(u3));
  const float o[4] = {ra * cosf(6.2831853071795864f * u2), ra * sinf(6.2831853071795864f * u2),
                      rb * cosf(6.2831853071795864f * u4_), rb * sinf(6.2831853071795864f * u4_)};
  for (int k = 0; k < 4; ++k) if (i * 4 + k < n) dst[i * 4 + k] = o[k];
}
void launch_fill_normal(float* dst, long n, uint64_t seed, hipStream_t s) {
  const long t = (n + 3) / 4;
  hipLaunchKernelGGL(fill_normal_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, dst, n, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// uniform(lo, hi): the other createNoiseInputs method (utils/nn_utils.lua:44-45 uniform(-1.0, 1.0)); 24-bit mantissa draws
__global__ void fill_uniform_kernel(float* dst, long n, float lo, float hi, uint32_t s0, uint32_t s1) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i * 4 >= n) return;
  const u4 r = philox4x32(u4{(uint32_t)i, (uint32_t)(i >> 32), 0x756e6966u, 0u}, s0, s1);
  const uint32_t o[4] = {r.x, r.y, r.z, r.w};
  for (int k = 0; k < 4; ++k) if (i * 4 + k < n) dst[i * 4 + k] = lo + (hi - lo) * ((float)(o[k] >> 8) * (1.f / 16777216.f));
}
void launch_fill_uniform(float* dst, long n, float lo, float hi, uint64_t seed, hipStream_t s) {
  const long t = (n + 3) / 4;
  hipLaunchKernelGGL(fill_uniform_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, dst, n, lo, hi, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// torch.dist(a_i, b_i) per row (apply_r.lua:369: 1 - torch.dist(images[i], fixedImage)): sqrt(sum (a-b)^2), fp32 difference and
// square, fp64 sum (TH accreal), one workgroup per row.
__global__ __launch_bounds__(256) void l2_distance_rows_kernel(const float* __restrict__ a, const float* __restrict__ b, long d, double* __restrict__ out) {
  __shared__ double sh[8];
  const float* pa = a + blockIdx.x * d; const float* pb = b + blockIdx.x * d;
  double s = 0;
  for (long i = threadIdx.x; i < d; i += blockDim.x) { const float t = fabsf(pa[i] - pb[i]); s += (double)(t * t); }
  s = block_reduce_sum(s, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = sqrt(s);
}
void launch_l2_distance_rows(const float* a, const float* b, long n, long d, double* out, hipStream_t s) {
  hipLaunchKernelGGL(l2_distance_rows_kernel, dim3((unsigned)n), dim3(256), 0, s, a, b, d, out);
}

// nn.SpatialUpSamplingNearest(2) (models.lua:121,127) as stand-alone passes: only the BACKWARD of the fused up-sampling +
// convolution stage materialises the up-sampled input (weight gradient) and folds the data gradient back (sum of each 2x2
// block) - G's forward never does (conv3x3_up2_f16x3_kernel), and G is forward-only on the train_r path.
__global__ __launch_bounds__(256) void upsample2_kernel(const float* __restrict__ x, float* __restrict__ up, long n_out, int Ho, int Wo) {
  const int Ws = Wo >> 1, Hs = Ho >> 1;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n_out; i += (long)gridDim.x * blockDim.x) {
    const long bc = i / ((long)Ho * Wo); const int p = (int)(i - bc * Ho * Wo), yy = p / Wo, xx = p - yy * Wo;
    up[i] = x[bc * Hs * Ws + (long)(yy >> 1) * Ws + (xx >> 1)];
  }
}
__global__ __launch_bounds__(256) void downsum2_kernel(const float* __restrict__ gup, float* __restrict__ gin, long n_in, int Hs, int Ws) {
  const int Wo = Ws * 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n_in; i += (long)gridDim.x * blockDim.x) {
    const long bc = i / ((long)Hs * Ws); const int p = (int)(i - bc * Hs * Ws), y = p / Ws, xx = p - y * Ws;
    const float* g = gup + bc * 4 * Hs * Ws + (long)(2 * y) * Wo + 2 * xx;
    gin[i] = ((g[0] + g[1]) + g[Wo]) + g[Wo + 1];       // THNN SpatialUpSamplingNearest.updateGradInput adds the four in scan order
  }
}
void launch_upsample2(const float* x, float* up, int B, int C, int Ho, int Wo, hipStream_t s) {
  const long n = (long)B * C * Ho * Wo; long blocks = (n + 255) / 256; if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(upsample2_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, up, n, Ho, Wo);
}
void launch_downsum2(const float* gup, float* gin, int B, int C, int Hs, int Ws, hipStream_t s) {
  const long n = (long)B * C * Hs * Ws; long blocks = (n + 255) / 256; if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(downsum2_kernel, dim3((unsigned)blocks), dim3(256), 0, s, gup, gin, n, Hs, Ws);
}

__global__ void scale_copy_kernel(const float* src, float* dst, long n, float scale) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = src[i] * scale;
}
__global__ __launch_bounds__(256) void zero_regions_kernel(ZeroJobs jobs) {
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  for (int j = 0; j < jobs.n; ++j) {
    uint4* p = reinterpret_cast<uint4*>(jobs.ptr[j]);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < jobs.n16[j]; i += (long)gridDim.x * blockDim.x) p[i] = z;
  }
}
void launch_zero_regions(const ZeroJobs& jobs, hipStream_t s) {
  long tot = 0; for (int j = 0; j < jobs.n; ++j) tot += jobs.n16[j];
  if (tot <= 0) return;
  long blocks = (tot + 255) / 256; if (blocks > 4096) blocks = 4096;
  KtScope kt("zero_regions_kernel", 0.0, 16.0 * (double)tot, s);
  hipLaunchKernelGGL(zero_regions_kernel, dim3((unsigned)blocks), dim3(256), 0, s, jobs);
}
void launch_scale_copy(const float* src, float* dst, long n, float scale, hipStream_t s) {
  long blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(scale_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, n, scale);
}

}  // namespace gr

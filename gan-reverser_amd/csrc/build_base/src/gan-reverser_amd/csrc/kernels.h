// kernels.h — internal launch interface between the net runtime (net.hip) and the gfx950 kernels.
// Not part of the ABI (that is include/ganrev.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdlib.h>

// ---------------------------------------------------------------- knobs
// The SHIPPING library (make: libganrev.so) reads four environment variables (GR_CONV_MODE, GR_RANGE_GUARD, GR_SIDE_WGRAD, GR_FUSED_HEAD: net.hip, gr_init) and
// answers the gr_set_tuning keys include/ganrev.h documents; nothing else selects kernels at run time.  Every other switch - A/B controls of
// variants that lost their measurement, and ablation bits that make kernels compute WRONG results by design (no stores, no MFMA, no DMA ...) - exists
// only in the ablation build (make ablate: libganrev_ablate.so, -DGR_ABLATE), where GR_KNOB reads the environment; in the shipping build it is its
// default, a compile-time constant the optimiser folds away, and GR_DBG(x) is 0 inside the kernels.  
#ifdef GR_ABLATE
#define GR_KNOB(name, def) (getenv(name) ? atoi(getenv(name)) : (def))
#define GR_KNOB_SET(name) (getenv(name) != nullptr)
#define GR_DBG(x) (x)
#else
#define GR_KNOB(name, def) (def)
#define GR_KNOB_SET(name) (false)
#define GR_DBG(x) 0
#endif

namespace gr {

// ---------------------------------------------------------------- conv3x3 (implicit GEMM on fp32 MFMA)
constexpr int CONV_CK = 8;  // input channels per LDS chunk (k = tap*8 + ci_local)

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// ---------------------------------------------------------------- optional per-kernel HIP-event timing (bench / roofline leg)
// When a timer is installed every launch_* brackets its kernel with two events ON THE LAUNCH STREAM and reports the
// kernel's name plus its ALGORITHMIC flops and bytes; with no timer installed these are two null checks.
struct KernelTimer {
  virtual void begin(const char* name, double flops, double bytes, hipStream_t s) = 0;
  virtual void end(hipStream_t s) = 0;
  virtual ~KernelTimer() {}
};
extern KernelTimer* g_ktimer;
struct KtScope {
  hipStream_t s;
  KtScope(const char* name, double flops, double bytes, hipStream_t st) : s(st) { if (g_ktimer) g_ktimer->begin(name, flops, bytes, s); }
  ~KtScope() { if (g_ktimer) g_ktimer->end(s); }
};

// Weights in "k-major" layout consumed by conv3x3_mfma: [Cin_pad/8][9 taps][8 ci][cout_pad]
struct ConvWeightLayout {
  int cin_pad, cout_pad;
  size_t elems() const { return (size_t)cin_pad * 9 * cout_pad; }
};
inline ConvWeightLayout conv_weight_layout(int cin, int cout) {
  return ConvWeightLayout{round_up(cin, CONV_CK), round_up(cout, 32)};
}
// optional fused epilogue of the conv kernels (evaluate()-mode BatchNorm + activation); mean == nullptr: BN skipped
struct ConvEpilogue { const float *mean = nullptr, *invstd = nullptr, *gamma = nullptr, *beta = nullptr; int act = 0; float slope = 0.f; };

// f16x3 scale tracking: a slot holds bit patterns of max|tensor| (a non-negative float orders like an unsigned).  A slot is
// AMAX_ENTRIES words, each on its own 128-byte line; a producer workgroup folds its maximum into entry (linear block id %
// AMAX_ENTRIES).  Workgroups are dispatched round-robin over the 8 XCDs, so one entry is only ever touched from one XCD
// and its atomics stay in that XCD's L2 (one shared word ping-pongs between the eight L2s: measured 180 us per launch on a
// 67 MB tensor against 20 us for the kernel itself).  The slot is zeroed before the producer runs; consumers take the
// maximum over the entries.  The pre-check may read a stale (only ever smaller) value: at worst a redundant atomic.
constexpr int AMAX_ENTRIES = 32, AMAX_STRIDE = 32, AMAX_WORDS = AMAX_ENTRIES * AMAX_STRIDE;   // 4 KB per tensor slot
#if defined(__HIPCC__)
__device__ __forceinline__ void absmax_commit(float m, unsigned* slot) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) {
    const unsigned b = __float_as_uint(m);
    const unsigned lb = blockIdx.x + blockIdx.y * gridDim.x;
    unsigned* e = slot + (lb % AMAX_ENTRIES) * AMAX_STRIDE;
    if (b > __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(e, b);
  }
}
// maximum over the entries of a slot; every lane of a fully active wave receives it
__device__ __forceinline__ unsigned absmax_read(const unsigned* slot) {
  unsigned v = slot[(threadIdx.x & (AMAX_ENTRIES - 1)) * AMAX_STRIDE];
#pragma unroll
  for (int o = AMAX_ENTRIES / 2; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o));
  return v;
}
// Cross-lane sums on the DPP path (full-rate VALU; __shfl_xor goes through ds_bpermute and costs an LDS instruction per step:
// 13 us per conv launch for the 64 channel sums of the BatchNorm statistics against 1 us this way).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_take(float v) {      // lanes outside ROW_MASK receive 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
// sum over the 32 lanes of each half-wave: the result is valid in lanes 16-31 and 48-63
__device__ __forceinline__ float half_wave_sum(float v) {
  v += dpp_take<0xB1, 0xF>(v);      // quad_perm [1,0,3,2]
  v += dpp_take<0x4E, 0xF>(v);      // quad_perm [2,3,0,1]
  v += dpp_take<0x141, 0xF>(v);     // row_half_mirror: the other quad of the 8-lane half holds the same sums
  v += dpp_take<0x140, 0xF>(v);     // row_mirror: the other half of the 16-lane row
  v += dpp_take<0x142, 0xA>(v);     // row_bcast15 into rows 1 and 3: lane 15 of the row before
  return v;
}
// sum over all 64 lanes: valid in lanes 48-63
__device__ __forceinline__ float wave_sum(float v) {
  v = half_wave_sum(v);
  v += dpp_take<0x143, 0xC>(v);     // row_bcast31 into rows 2 and 3: lane 31 holds the sum of the first half-wave
  return v;
}
// f16x3 operand handling shared by the convolution and GEMM kernels (see conv.hip for the arithmetic)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int f16_scale_exp(unsigned amax_bits) {      // k such that max|x| * 2^k lies in [2^14, 2^15)
  const int e = (int)((amax_bits >> 23) & 0xffu);
  return e == 0 ? 0 : min(141 - e, 126);
}
__device__ __forceinline__ float pow2f(int k) { return __uint_as_float((unsigned)(127 + k) << 23); }   // k in [-126, 127]
// split 8 scaled floats into the two fp16 term vectors (round-to-nearest both times; x - x0 is exact in fp32)
__device__ __forceinline__ void split8_f16(const float* x, float sc, uint4& t0, uint4& t1) {
  unsigned short a[8], b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = x[j] * sc;
    const _Float16 h0 = (_Float16)v; const float r = v - (float)h0; const _Float16 h1 = (_Float16)r;
    a[j] = __builtin_bit_cast(unsigned short, h0); b[j] = __builtin_bit_cast(unsigned short, h1);
  }
  t0 = make_uint4(a[0] | (unsigned)a[1] << 16, a[2] | (unsigned)a[3] << 16, a[4] | (unsigned)a[5] << 16, a[6] | (unsigned)a[7] << 16);
  t1 = make_uint4(b[0] | (unsigned)b[1] << 16, b[2] | (unsigned)b[3] << 16, b[4] | (unsigned)b[5] << 16, b[6] | (unsigned)b[7] << 16);
}
// Streaming stores (an experiment kept as a knob, OFF by default).  A plain store allocates its line in the XCD's L2 and the Infinity
// Cache; a non-temporal one does not.  Round 3 measured the outputs of the convolutions, the up-sampling convolutions, the forward
// pipeline and pass B stored non-temporally (g_nt_stores bits 0..3; gr_set_tuning "nt_stores", GR_NT_STORES) in the real step, same
// box, interleaved: cfg3 12.62-12.86 ms over all masks, cfg2 2.059-2.066 ms - no effect beyond run-to-run noise.  (A first reading
// of -7 % was an artefact: the diagnostic bit used to switch it on also switched off pass B's operand-ready image.)  Non-temporal
// LOADS in the backward pipeline passes do pay at cfg3: elem.hip, ld4_maybe_nt.
typedef float st_f4 __attribute__((ext_vector_type(4)));
typedef unsigned st_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store4(float* p, const float4& v, bool nt) {
  if (nt) { const st_f4 t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<st_f4*>(p)); }
  else *reinterpret_cast<float4*>(p) = v;
}
__device__ __forceinline__ void store4(uint4* p, const uint4& v, bool nt) {
  if (nt) { const st_u4 t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<st_u4*>(p)); }
  else *p = v;
}
__device__ __forceinline__ float absmax4(float m, const float4& v) {
  return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}
#endif

// native [cout][cin][3][3] -> k-major (forward) ; or the transposed+flipped k-major the backward-data pass needs
void launch_conv_weight_prep(const float* w_native, float* wt, int cin, int cout, bool for_backward_data, hipStream_t s);

// out[B,Cout,H,W] = conv3x3(in) (+bias).  `up`: in is [B,Cin,H/2,W/2] and is nearest-upsampled x2 while staged.
// wt is the k-major layout for (Cin -> Cout).
// w_native (nullable): the same weights in the module's own [Cout][Cin][3][3] layout; lets few-output-channel layers
// (Cout <= 4) take the HBM-bound VALU kernel instead of a 32-row MFMA block.
void launch_conv3x3(const float* in, const float* wt, const float* bias, float* out,
                    int B, int Cin, int Cout, int H, int W, bool up, hipStream_t s, const float* w_native = nullptr,
                    const ConvEpilogue* ep = nullptr);

// Cin <= 3 forward (R's first layer): HBM-bound VALU kernel on the native weights, any arithmetic mode
bool conv_fewin_applies(int Cin, int W, bool up);
// Operand-ready OUTPUT of a convolution epilogue (evaluate() mode, f16x3; round 4).  The epilogue's result act(BN(conv)) is written
// as the consuming convolution's P16 image instead of (out == nullptr) or beside the fp32 tensor.  The image's power-of-two scale has
// to be fixed before the tensor exists: `scale` is the consumer's scale slot and already holds an UPPER BOUND of max|result|
// (launch_eval_bound: a weight-norm bound, see there); the TRUE maximum still goes to amax_out, for the bound of the stage after.
struct P16Out { void* p16 = nullptr; const unsigned* scale = nullptr; };
void launch_conv3x3_fewin(const float* in, const float* w_native, const float* bias, float* out, int B, int Cin, int Cout, int H, int W,
                          hipStream_t s, const ConvEpilogue* ep = nullptr, unsigned* amax_out = nullptr,
                          double* stat_part = nullptr, int* stat_tiles = nullptr, const P16Out* p16o = nullptr);
bool conv_fewin_p16_out_supported(int Cout, int H, int W);

// fp32-accurate convolution on the bf16 MFMA: operands split into 3 bf16 terms, 6 products, fp32 accumulation ("bf16x6").
// wsplit = image made by launch_conv_weight_split (forward or backward-data flavour, like launch_conv_weight_prep).
size_t conv_weight_split_bytes(int cin, int cout, bool bwd, int ksz = 3);
// nterm 3 = bf16x6; nterm 2 = "f16x3": two fp16 terms of the power-of-two-scaled operands, 3 products.  The scales come from
// device slots holding the bit pattern of max|tensor| (amax_*), filled by launch_absmax or by the producing kernel.
void launch_absmax(const float* x, long n, unsigned* slot, hipStream_t s, bool slot_is_zero = false);   // slot_is_zero: the caller has just filled the slot with 0       // zeroes the slot, then max|x| -> slot
void launch_conv_weight_split(const float* w_native, void* dst, int cin, int cout, bool bwd, hipStream_t s, int nterm = 3, unsigned* amax = nullptr,
                              int ksz = 3, bool take_absmax = true);   // ksz 5: the 25-tap image of conv5x5_split_kernel; take_absmax false: amax already holds max|w|
// nn.SpatialConvolution(Cin, Cout, 5, 5, 1, 1, 2, 2) (models.lua:297) on the f16x3 split kernel: forward, and the data gradient with the bwd image (Cout -> Cin)
bool conv5x5_split_supported(int Cin, int Cout, int H, int W);
void launch_conv5x5_split(const float* in, const void* wsplit, const float* bias, float* out, int B, int Cin, int Cout, int H, int W,
                          hipStream_t s, const unsigned* amax_in, const unsigned* amax_w);        // nterm 2: also computes amax_w
void launch_conv3x3_split(const float* in, const void* wsplit, const float* bias, float* out,
                          int B, int Cin, int Cout, int H, int W, bool up, hipStream_t s, const ConvEpilogue* ep = nullptr,
                          int nterm = 3, const unsigned* amax_in = nullptr, const unsigned* amax_w = nullptr,
                          unsigned* amax_out = nullptr /* nullable: max|out| is folded into this slot by the epilogue */,
                          // BatchNorm batch statistics from the epilogue: stat_part [Cout][tiles][2] doubles (room for
                          // conv_stat_tiles_max tiles); *stat_tiles = tiles written per channel, 0 when the kernel chosen
                          // for this shape does not produce them (then run launch_bn_stats on the output)
                          double* stat_part = nullptr, int* stat_tiles = nullptr);
// Operand-ready activations ("P16"): p16[b][c / 8][term][p16_pos(pixel)] = 16 bytes = the 8 fp16 halves of term `term` (hi, lo) of
// channels 8(c/8) .. +7 at that pixel, scaled by the power of two the tensor's scale slot defines (f16_scale_exp of its
// bits).  Same bytes as the fp32 tensor.  Written by the pipeline kernels (elem.hip), read by LDS-DMA in the conv3x3_p16_* kernels.
// Pixels are linear inside a plane: the pipeline kernels assemble the vectors through an LDS transpose (elem.hip, t8_emit), so
// consecutive lanes hold consecutive pixels and a wave's store is 1 KB contiguous.  (A "quad-major" order - pixel 4q + k at
// 64k + q, for producers whose threads own 4 pixels x 8 channels - was measured first: those threads were too heavy.)
extern int g_p16_min_tiles, g_p16_variant, g_p16_stagger, g_up2_debug, g_up2_quad, g_up2_stagger, g_nt_stores;
extern int g_stack8_min_wgs;
extern void* g_p16_stamps;
extern int g_p16_debug;       // diagnostic builds only (GR_P16_DEBUG bit mask: 1 no output stores, 2 no statistics, 4 no DMA, 8 no MFMA)
void launch_to_p16(const float* x, void* p16, int B, int C, int HW, const unsigned* slot, hipStream_t s);   // C % 8 == 0, HW % 4 == 0
#if defined(__HIPCC__)
__host__ __device__ __forceinline__ unsigned p16_pos(unsigned p) { return p; }   // position of pixel p inside a (group, term) plane: linear
#endif
bool conv_p16_supported(int B, int Cin, int Cout, int H, int W);
void launch_conv3x3_p16(const void* x_p16, const void* wsplit, const float* bias, float* out, int B, int Cin, int Cout, int H, int W,
                        hipStream_t s, const ConvEpilogue* ep, const unsigned* amax_in, const unsigned* amax_w, unsigned* amax_out,
                        double* stat_part, int* stat_tiles, const P16Out* p16o = nullptr);
bool conv_p16_out_supported(int Cout);       // the P16 kernels can write their result operand-ready (whole 8-channel groups; the default four-wave kernels)
// Upper bound of max|act(BN(conv(x) + bias))| over a whole tensor BEFORE it is computed, from max|x| and the weights alone:
//   |conv_o| <= (sum_k |w_ok|) * max|x| + |bias_o|,  BatchNorm with running statistics is a per-channel affine map, and every activation
// here satisfies |act(z)| <= |z| (ELU, ReLU, LeakyReLU with |slope| <= 1) or <= 1 (Sigmoid, Tanh).  wl1 = per-output-channel L1 norms
// (launch_conv_weight_l1).  wl1 == nullptr: in_max is max|y| of the RAW main-op output (bias included) and only the pipeline is bounded.
// The bound overshoots the true maximum by a few bits (sqrt(9 Cin) x the crest factor of x); an f16x3 image scaled by a bound 2^m too large
// keeps 22 bits for elements down to 2^(m-17) of the true maximum and an absolute error of 2^(m-40) of it below - the true maximum of the
// tensor is tracked beside the bound (P16Out) so the overshoot does not compound from layer to layer.
void launch_conv_weight_l1(const float* w_native, int cout, int fan_in, float* wl1, hipStream_t s);
void launch_eval_bound(const float* wl1, const float* bias, const ConvEpilogue* ep, int Cout, float post_scale, const unsigned* in_max, unsigned* bound_out, hipStream_t s);
inline size_t conv_stat_tiles_max(int B, int H, int W) { return (size_t)B * ((H + 7) / 8) * ((W + 31) / 32) + 1; }     // smallest tile: 8 rows x 32 (or one 16x16 image)
// mean / invstd (+ running statistics) from the per-tile (sum, sum of squares) the conv epilogue wrote
// bounds (nullable): from max|y| (slot amax_y) and the fresh statistics, an upper bound of max|pipeline output| is folded into
// bound_out (the consuming convolution's scale slot) and the factor K with max|dy| <= K * max|dz| of the stage's backward
// into kb_out - both known BEFORE the kernels that write those tensors run, so they can write them operand-ready
struct BnBounds { const unsigned* amax_y; const float *gamma, *beta; int act; float mask_scale; unsigned* bound_out; unsigned* kb_out; };
void launch_bn_stats_from_tiles(const double* stat_part, int tiles, int C, double n, float* mean, float* invstd,
                                float* run_mean, float* run_var, hipStream_t s, const BnBounds* bounds = nullptr);

// nearest x2 up-sampling + conv3x3 as four 2x2 convolutions of the source plane (f16x3 arithmetic; forward only):
// 4 instead of 9 multiply-adds per output.  Shapes: source plane 8x8, 16x16 or at least 17 wide; Cout > 4.
bool conv_up2_supported(int Cin, int Cout, int H, int W);
size_t conv_weight_up2_bytes(int cin, int cout);
void launch_conv_weight_up2_split(const float* w_native, void* wup, int cin, int cout, hipStream_t s, unsigned* amax_w,
                                  bool take_absmax = true);   // take_absmax: recompute the slot max|w| first
void launch_conv3x3_up2_f16x3(const float* in, const void* wup, const float* bias, float* out, int B, int Cin, int Cout, int H, int W,
                              hipStream_t s, const ConvEpilogue* ep, const unsigned* amax_in, const unsigned* amax_w, unsigned* amax_out);

// all weight images of a net in one launch
// split: 0 = fp32 k-major image, 1 = bf16 3-term image, 2 = f16 2-term image scaled by the slot `amax` (max|w|)
struct PrepJob { long w_off; void* dst; int cin, cout, CI, CO, cin_pad, cout_pad, bwd, split; unsigned* amax; };
PrepJob make_prep_job(long w_off, void* dst, int cin, int cout, bool for_backward_data, int split, unsigned* amax = nullptr);
// amax_slots != null (f16 images): the n_slots weight maxima are recomputed first
void launch_conv_weight_prep_batch(const PrepJob* jobs_dev, int njobs, const float* params, hipStream_t s,
                                   unsigned* amax_slots = nullptr, int n_slots = 0);

// weight gradient: slab workspace sized by conv_wgrad_workspace(); result accumulated (+=) into gw native layout
// mode 1: bf16x6 split on the bf16 MFMA where the shape allows (W % 8 == 0, Cin > 3); 2: f16x3 split (needs the maxima of
// x and dy in device slots); 0: fp32 MFMA
size_t conv_wgrad_workspace_bytes(int B, int Cin, int Cout, int H, int W, int mode = 0);
void launch_conv3x3_wgrad(const float* x, const float* dy, float* gw, void* workspace,
                          int B, int Cin, int Cout, int H, int W, hipStream_t s, int mode = 0,
                          const unsigned* amax_x = nullptr, const unsigned* amax_dy = nullptr);
// weight gradient with BOTH operands operand-ready (P16): x = the stage's input image, dy = pass B's image
bool conv_wgrad_p16_supported(int B, int Cin, int Cout, int H, int W);
size_t conv_wgrad_p16_workspace_bytes(int B, int Cin, int Cout, int H, int W);
void launch_conv3x3_wgrad_p16(const void* x_p16, const void* dy_p16, float* gw, void* workspace, int B, int Cin, int Cout, int H, int W,
                              hipStream_t s, const unsigned* amax_x, const unsigned* amax_dy);
inline bool conv_wgrad_is_split(int mode, int Cin, int W) { return mode >= 1 && Cin > 3 && W >= 16 && W % 8 == 0; }

// ---------------------------------------------------------------- GEMM (Linear) on fp32 MFMA
// C[m][n] (+)= sum_k A(m,k) * B(n,k) (+ bias[n]);  A(m,k) = A[m*rsA + k*ksA], B(n,k) = Bm[n*rsB + k*ksB]
size_t gemm_workspace_bytes(int M, int N, int K);
// ep (nullable; only when gemm_epilogue_possible, i.e. no split-K): per-column evaluate()-mode BatchNorm + activation
bool gemm_epilogue_possible(int M, int N, int K);
void launch_gemm(const float* A, long rsA, long ksA, const float* Bm, long rsB, long ksB,
                 float* C, long ldc, const float* bias, bool accumulate, int M, int N, int K,
                 void* workspace, hipStream_t s, const ConvEpilogue* ep = nullptr, unsigned* amax_out = nullptr,
                 // both non-null: the f16x3 kernel (operands scaled by their tracked maxima and split in two fp16 terms while staged)
                 const unsigned* amax_a = nullptr, const unsigned* amax_b = nullptr);

// ---------------------------------------------------------------- per-channel pipelines (BN / act / dropout / pool)
enum Act { ACT_NONE = 0, ACT_ELU = 3, ACT_RELU = 4, ACT_LEAKYRELU = 5, ACT_SIGMOID = 6, ACT_TANH = 7,
           ACT_PRELU = 16 /* nn.PReLU() with one shared slope: LeakyReLU whose slope is read from device memory (PostArgs::slope_dev) */ };
enum MaskKind { MASK_NONE = 0, MASK_ELEM = 1, MASK_SPATIAL = 2, MASK_SCALE = 3 /* evaluate(): x*(1-p) */ };
struct MaskRef { int kind; const uint32_t* bits; float scale; };

struct PostArgs {
  const float* y;          // raw main-op output [B,C,H,W]
  float* out;              // pipeline output [B,C,Ho,Wo]
  int B, C, H, W;          // pre-pool dims
  int has_bn;
  const float *mean, *invstd, *gamma, *beta;  // per channel
  int act; float slope;
  const float* slope_dev;  // ACT_PRELU: the learnable slope (one float in the net's flat parameter vector)
  MaskRef m1;              // applied before the pool, indexed at [B,C,H,W] (ELEM) or [B,C] (SPATIAL)
  int pool;                // 2x2 max pool, stride 2
  uint8_t* pool_idx;       // [B,C,Ho,Wo] argmax 0..3
  MaskRef m2;              // applied after the pool, indexed at [B,C,Ho,Wo] / [B,C]
  unsigned* amax_out;      // nullable: max|out| is folded into this slot (f16x3 scale of the consuming convolution)
  // operand-ready copy of `out` for the convolution that consumes it (see conv_p16_supported): p16 != null selects the
  // 8-channel-group kernel; p16_scale = the consumer's scale slot, which already holds an UPPER BOUND of max|out| (written by
  // launch_bn_stats_from_tiles from the batch statistics and max|y|) - amax_out must then be null
  void* p16; const unsigned* p16_scale;
  int nt;                  // non-temporal loads of y (set by the launcher: tensors far larger than the Infinity Cache)
  int nt_st;               // non-temporal stores of the outputs (launcher: g_nt_stores)
};
void launch_post_forward(const PostArgs& a, hipStream_t s);

constexpr int PB_SPLITS = 256;   // row length of partials_b: pass B of the operand-ready pipeline slices the batch finer than pass A (blocks of 8 channels)
constexpr int STAT_SPLITS = 64;  // partial sums per channel (16 -> 64: 4 -> 16 waves per SIMD in flight on the 64-channel layers; pass A 1.41 -> 1.12 ms at cfg3)
// per-channel (sum, sumsq) partials in double -> mean / invstd (+ running stats update when run_mean != null)
// Synchronised BatchNorm under data parallelism (optional): the per-channel sums a BatchNorm layer reduces over its batch -
// (sum y, sum y^2) in the forward, (sum dz, sum dz (y - mean)) in the backward - are added over the ranks before they are used, so
// that P ranks of B images compute exactly what one device computes on P x B images (models.lua:410-448 on the global batch).
// `sum` is called in stream order with a compact device buffer [count] of doubles and must leave the SUM over the ranks in it;
// `max_u32` likewise with the element-wise maximum (the f16x3 scale bound of dy needs the GLOBAL max|dz|).  n_global = elements per
// channel over all ranks; grad_scale = 1 / ranks for the gamma / beta gradients, which come out of the GLOBAL sums on every rank and
// meet the gradient all-reduce (a SUM) afterwards.
struct StatSync {
  int (*sum)(void* user, double* buf, long count);
  int (*max_u32)(void* user, unsigned* buf, long count);
  void* user;
  double* buf;             // device scratch, >= 2 * C doubles
  double n_global;
  float grad_scale;
};
// per-channel sums of `count` (a, b) pairs at row stride `stride` pairs -> compact out[C][2]  (fixed order: deterministic)
void launch_pair_sums(const double* part, int stride, int count, int C, double* out, hipStream_t s);
// out[C][2] -> pair 0 of every row of a [C][stride][2] array
void launch_pair_scatter(const double* in, int C, int stride, double* part, hipStream_t s);
void launch_bn_stats(const float* y, int B, int C, int HW, double* partials /*[C][STAT_SPLITS][2]*/,
                     float* mean, float* invstd, float* run_mean, float* run_var, int training, hipStream_t s, const StatSync* sync = nullptr);
void launch_bn_eval_prepare(const float* run_mean, const float* run_var, float* mean, float* invstd, int C, hipStream_t s);

struct PostBwdArgs {
  PostArgs f;              // the forward description (y, masks, pool_idx, bn params)
  const float* gout;       // grad wrt pipeline output [B,C,Ho,Wo]
  float* dy;               // grad wrt raw y [B,C,H,W] (written)
  double* partials;        // [C][STAT_SPLITS][2]
  float* coef;             // [C][2] : gm, k   (BN backward coefficients)
  float* ggamma; float* gbeta;   // += (BN)
  float* gbias;            // += sum dy per channel (conv / linear bias), nullable
  unsigned* amax_dy;       // nullable: max|dy| is folded into this slot (f16x3 scale of the weight / data gradients)
  double* partials_b;      // [C][PB_SPLITS] pass B's per-channel sums of dy (bias gradient); separate from `partials`,
                           // which every pass-B workgroup of the channel still reads (BN coefficients are derived in pass B)
  // operand-ready copy of dy for the data-gradient convolution: dy_p16 != null selects the 8-channel-group pass B.  amax_dz
  // receives max|dz| from pass A; kb holds the forward's factor K (BnBounds); pass B writes the bound K * max|dz| into amax_dy
  // (which then must not be accumulated into) and scales by it.
  void* dy_p16; unsigned* amax_dz; const unsigned* kb;
  int nt;                  // non-temporal loads of gradOutput and y (set by the launcher, per pass)
  double gscale;           // factor on the gamma / beta gradients: 1, or 1 / ranks under synchronised BatchNorm (StatSync); 0 reads as 1
};
bool post_g8_supported(int C, int H, int W, bool pool, bool backward = false);
// Bias gradients are summed from partials_b by one batched launch for several stages (launch_bias_grad_batch) when
// `defer` is given; otherwise inside the call.
struct BiasJob { const double* partials; float* gbias; int C, splits; };     // partials: rows of PB_SPLITS
struct BiasJobs { BiasJob job[16]; int n; };
void launch_post_backward(const PostBwdArgs& a, hipStream_t s, BiasJobs* defer = nullptr, const StatSync* sync = nullptr);
void launch_bias_grad_batch(BiasJobs& jobs, hipStream_t s);    // runs and empties the list

// ---------------------------------------------------------------- K x K convolution (odd K other than 3) and nn.PReLU: convk.hip
bool convk_supported(int K);
size_t convk_workspace_bytes(int B, int Cin, int Cout, int K);
void launch_convk_forward(const float* in, const float* w /*[Cout][Cin][K][K]*/, const float* bias, float* out, void* ws, int B, int Cin, int Cout, int H, int W, int K, hipStream_t s);
void launch_convk_backward_data(const float* gout, const float* w, float* gin, void* ws, int B, int Cin, int Cout, int H, int W, int K, hipStream_t s);
void launch_convk_backward_weight(const float* in, const float* gout, float* gw /*+=*/, void* ws, int B, int Cin, int Cout, int H, int W, int K, hipStream_t s);
size_t prelu_grad_workspace_bytes();
void launch_prelu_grad(const float* g, const float* z, long n, double* part, float* gslope /*+=*/, hipStream_t s);

// ---------------------------------------------------------------- criterion / optimiser / misc
void launch_mse(const float* x, const float* t, long n, long n_global, double* loss_dev, float* grad, hipStream_t s);
// R's head in one launch (elem.hip, head_fwd_bwd_kernel): BatchNorm + act + Dropout of the fc1 stage, fc2 [+ Tanh], MSE, and their backward down to fc1's dy
struct HeadLaunch {
  int B, C1, nd; long n_global;
  const float* y1; float* out1;                                  // fc1: raw output (bias added), stage output
  float *mean, *invstd, *run_mean, *run_var; const float *gamma, *beta;
  MaskRef m1; int act1; float slope1; int act2;                  // fc1 stage: Dropout mask, activation; fc2 stage: ACT_NONE or ACT_TANH
  const float *W2, *b2; float *y2, *out2;                        // fc2: weights [nd][C1], bias, raw output, stage output (== y2 without an activation)
  const float* target; double* loss; double* loss_part;          // the criterion's target, device loss, [C1 / 8] partial sums
  float *gout, *gy2, *dy1;                                       // gradOutput of the net [B][nd], gradient wrt fc2's raw output, wrt fc1's raw output [B][C1]
  float *gW2, *gb2, *ggamma, *gbeta, *gb1;                       // accumulated into (accGradParameters)
  unsigned* amax_dy;                                             // f16x3: max|dy1| slot of fc1's backward GEMMs (nullable)
  unsigned* bar; unsigned bar_base;                              // grid-barrier arrival counter (monotonic; the launch adds 2 x C1 / 8) and its value before this launch
  unsigned* fault; int spin_limit;                               // sticky fault word a timed-out barrier sets; polls per barrier before giving up (0 = the default 2^22, ~5 s)
};
bool head_supported(int B, int C1, int nd);
void launch_head_fwd_bwd(const HeadLaunch& h, hipStream_t s);
void launch_add_inplace(float* y, const float* x, long n, hipStream_t s);        // y += x
void launch_bce(const float* x, const float* t, long n, double* loss_dev, float* grad, hipStream_t s);      // nn.BCECriterion (sizeAverage)
struct AdamConsts { float b1, b2, c1, c2, eps, step, l1, l2, clamp; int use_penalty, use_clamp; };
void launch_penalty_clamp_adam(float* theta, float* g, float* m, float* v, long n, const AdamConsts& c, hipStream_t s, const unsigned* skip = nullptr);
void launch_gen_mask(uint32_t* words, long n_elems, float p_drop, uint64_t seed, uint64_t counter, uint32_t layer, hipStream_t s);
// all masks of one forward in one launch (jobs travel in the kernel argument block)
struct MaskJob { uint32_t* words; long nwords; uint32_t thresh; int half; uint32_t layer; };
struct MaskJobs { MaskJob job[24]; int n; };
MaskJob make_mask_job(uint32_t* words, long n_elems, float p_drop, uint32_t layer);
void launch_gen_mask_batch(const MaskJobs& jobs, uint64_t seed, uint64_t counter, hipStream_t s);
void launch_pack_mask(const uint8_t* keep, uint32_t* words, long n, hipStream_t s);
void launch_unpack_mask(const uint32_t* words, uint8_t* keep, long n, hipStream_t s);
void launch_fill_normal(float* dst, long n, uint64_t seed, hipStream_t s);
void launch_fill_uniform(float* dst, long n, float lo, float hi, uint64_t seed, hipStream_t s);

// ---- f16x3 range guard.  The f16x3 arithmetic scales a whole tensor by one power of two: an entry 2^k below the tensor's
// maximum keeps about 40 - k bits (fp16's exponent range ends there).  Where a kernel's reduction runs over a channel index (the
// forward and the data gradient: activation x weight), the worst-case relative error of an output channel grows with the PRODUCT
// of the two tensors' per-channel spreads; where it runs over pixels (the weight gradient: activation x gradient, one channel
// pair per sum) with the larger one.  These kernels measure spreads: per-channel max|.| of a strided view t[b * sB + c * sC + i]
// (b < B, i < HW) into chmax[c] (atomicMax of bit patterns; chmax must be zero), then one verdict block enters
// log2(max_c / smallest non-zero chmax) (+1) into the largest spread seen so far on its SIDE - *word = activation side (0) |
// weight side (1) << 16 - and zeroes chmax again.  launch_pair_spread does the same, activation side, for the per-channel vector
// max(|a_c|, |b_c|) (BatchNorm gamma / beta: they set the channel ranges of every tensor behind a BatchNorm).  The caller compares
// the sum of the two sides with its budget (net.hip: 20 bits).
void launch_channel_absmax(const float* t, int B, int C, long HW, long sB, long sC, unsigned* chmax, hipStream_t s);
void launch_spread_verdict(unsigned* chmax, int C, unsigned* word, int side, hipStream_t s);
void launch_pair_spread(const float* a, const float* b, int C, unsigned* word, hipStream_t s);
void launch_l2_distance_rows(const float* a, const float* b, long n, long d, double* out, hipStream_t s);
void launch_scale_copy(const float* src, float* dst, long n, float scale, hipStream_t s);
// several regions zeroed by ONE launch (each a multiple of 16 bytes, 16-byte aligned): the fills a training step needs - the scale slots of both
// nets, the gradient vector - were three hipMemsetAsync kernels of ~6 us each at batch 256
struct ZeroJobs { void* ptr[4]; long n16[4]; int n; };
void launch_zero_regions(const ZeroJobs& jobs, hipStream_t s);
void launch_upsample2(const float* x, float* up, int B, int C, int Ho, int Wo, hipStream_t s);       // nearest x2, [B,C,Ho/2,Wo/2] -> [B,C,Ho,Wo]
void launch_downsum2(const float* gup, float* gin, int B, int C, int Hs, int Ws, hipStream_t s);   // its backward: sum of each 2x2 block

// ---------------------------------------------------------------- cosine top-k search
size_t cosine_topk_workspace_bytes(long N, int d, int Q, int k);
// idx_out/score_out are DEVICE buffers [Q][k].  status_dev (device word, nullable): large tables take the sample-bound filter
// (search.hip) when it is given; it receives 1 when a candidate list overflowed - call again with unfiltered = 1 - else 0.
// query_rows_host (nullable): the same rows in host memory - a handful of needles (cosine_topk_small_path) travel in the kernel arguments
int launch_cosine_topk(const float* emb, long N, int d, const long* query_rows_dev, int Q, int k,
                       long* idx_out, float* score_out, int accf, void* workspace, hipStream_t s, unsigned* status_dev = nullptr, int unfiltered = 0,
                       const long* query_rows_host = nullptr, unsigned* arrival_counter = nullptr,    // arrival_counter: SEARCH_STATE_WORDS zeroed device words the caller owns (word 0: the sample launch's arrival counter, words 16..: its histogram bins; both are left at 0 again)
                       unsigned* done_words = nullptr, unsigned seq = 0);   // small path: done_words[q] (host-visible) receives seq once needle q's results are written; the call then returns 2
constexpr int SEARCH_STATE_WORDS = 16 + 8 * 1024;
bool cosine_topk_small_path(long N, int d, int Q, int k);     // the filtered search takes the fp32-filter path (query rows by value, no device copy of them needed)

// ---------------------------------------------------------------- k-means + nearest-centroid pass (apply_r.lua:197-217)
size_t kmeans_workspace_bytes(long N, int d, int k);
// all pointers device; cent [k][d] holds the initial centroids on entry and the final ones on return; returns 1 when the
// shape is unsupported (k > 32, d > 256)
int launch_kmeans(const float* x, long N, int d, int k, int niter, float* cent, float* c2, float* counts, float* totalcounts,
                  int* labels_out, void* workspace, hipStream_t s);
int launch_cosine_assign(const float* x, long N, int d, const float* cent, int k, int take_min, float* w32_scratch,
                         int* labels, float* sims, hipStream_t s);

// mfmaloop.hip: the bare LDS-read + f16x3 MFMA loop (sustained ceiling of the convolution inner loop on this device; diagnostic)
size_t mfma_loop_workspace_bytes();
double mfma_loop_flops(int iters);
void launch_mfma_loop_fill(void* workspace, hipStream_t s);
void launch_mfma_loop(int shape, void* workspace, int iters, hipStream_t s);
}  // namespace gr

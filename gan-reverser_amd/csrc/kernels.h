// kernels.h — internal launch interface between the net runtime (net.hip) and the gfx950 kernels.
// Not part of the ABI (that is include/ganrev.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

// native [cout][cin][3][3] -> k-major (forward) ; or the transposed+flipped k-major the backward-data pass needs
void launch_conv_weight_prep(const float* w_native, float* wt, int cin, int cout, bool for_backward_data, hipStream_t s);

// out[B,Cout,H,W] = conv3x3(in) (+bias).  `up`: in is [B,Cin,H/2,W/2] and is nearest-upsampled x2 while staged.
// wt is the k-major layout for (Cin -> Cout).
// w_native (nullable): the same weights in the module's own [Cout][Cin][3][3] layout; lets few-output-channel layers
// (Cout <= 4) take the HBM-bound VALU kernel instead of a 32-row MFMA block.
void launch_conv3x3(const float* in, const float* wt, const float* bias, float* out,
                    int B, int Cin, int Cout, int H, int W, bool up, hipStream_t s, const float* w_native = nullptr,
                    const ConvEpilogue* ep = nullptr);

// fp32-accurate convolution on the bf16 MFMA: operands split into 3 bf16 terms, 6 products, fp32 accumulation ("bf16x6").
// wsplit = image made by launch_conv_weight_split (forward or backward-data flavour, like launch_conv_weight_prep).
size_t conv_weight_split_bytes(int cin, int cout, bool for_backward_data);
void launch_conv_weight_split(const float* w_native, void* wsplit, int cin, int cout, bool for_backward_data, hipStream_t s);
void launch_conv3x3_bf16x6(const float* in, const void* wsplit, const float* bias, float* out,
                           int B, int Cin, int Cout, int H, int W, bool up, hipStream_t s, const ConvEpilogue* ep = nullptr);

// all weight images of a net in one launch
struct PrepJob { long w_off; void* dst; int cin, cout, CI, CO, cin_pad, cout_pad, bwd, split; };
PrepJob make_prep_job(long w_off, void* dst, int cin, int cout, bool for_backward_data, bool split);
void launch_conv_weight_prep_batch(const PrepJob* jobs_dev, int njobs, const float* params, hipStream_t s);

// weight gradient: slab workspace sized by conv_wgrad_workspace(); result accumulated (+=) into gw native layout
// mode 1: bf16x6 split on the bf16 MFMA where the shape allows (W % 8 == 0, Cin > 3); 0: fp32 MFMA
size_t conv_wgrad_workspace_bytes(int B, int Cin, int Cout, int H, int W, int mode = 0);
void launch_conv3x3_wgrad(const float* x, const float* dy, float* gw, void* workspace,
                          int B, int Cin, int Cout, int H, int W, hipStream_t s, int mode = 0);

// ---------------------------------------------------------------- GEMM (Linear) on fp32 MFMA
// C[m][n] (+)= sum_k A(m,k) * B(n,k) (+ bias[n]);  A(m,k) = A[m*rsA + k*ksA], B(n,k) = Bm[n*rsB + k*ksB]
size_t gemm_workspace_bytes(int M, int N, int K);
void launch_gemm(const float* A, long rsA, long ksA, const float* Bm, long rsB, long ksB,
                 float* C, long ldc, const float* bias, bool accumulate, int M, int N, int K,
                 void* workspace, hipStream_t s);

// ---------------------------------------------------------------- per-channel pipelines (BN / act / dropout / pool)
enum Act { ACT_NONE = 0, ACT_ELU = 3, ACT_RELU = 4, ACT_LEAKYRELU = 5, ACT_SIGMOID = 6, ACT_TANH = 7 };
enum MaskKind { MASK_NONE = 0, MASK_ELEM = 1, MASK_SPATIAL = 2, MASK_SCALE = 3 /* evaluate(): x*(1-p) */ };
struct MaskRef { int kind; const uint32_t* bits; float scale; };

struct PostArgs {
  const float* y;          // raw main-op output [B,C,H,W]
  float* out;              // pipeline output [B,C,Ho,Wo]
  int B, C, H, W;          // pre-pool dims
  int has_bn;
  const float *mean, *invstd, *gamma, *beta;  // per channel
  int act; float slope;
  MaskRef m1;              // applied before the pool, indexed at [B,C,H,W] (ELEM) or [B,C] (SPATIAL)
  int pool;                // 2x2 max pool, stride 2
  uint8_t* pool_idx;       // [B,C,Ho,Wo] argmax 0..3
  MaskRef m2;              // applied after the pool, indexed at [B,C,Ho,Wo] / [B,C]
};
void launch_post_forward(const PostArgs& a, hipStream_t s);

constexpr int STAT_SPLITS = 16;  // partial sums per channel
// per-channel (sum, sumsq) partials in double -> mean / invstd (+ running stats update when run_mean != null)
void launch_bn_stats(const float* y, int B, int C, int HW, double* partials /*[C][STAT_SPLITS][2]*/,
                     float* mean, float* invstd, float* run_mean, float* run_var, int training, hipStream_t s);
void launch_bn_eval_prepare(const float* run_mean, const float* run_var, float* mean, float* invstd, int C, hipStream_t s);

struct PostBwdArgs {
  PostArgs f;              // the forward description (y, masks, pool_idx, bn params)
  const float* gout;       // grad wrt pipeline output [B,C,Ho,Wo]
  float* dy;               // grad wrt raw y [B,C,H,W] (written)
  double* partials;        // [C][STAT_SPLITS][2]
  float* coef;             // [C][2] : gm, k   (BN backward coefficients)
  float* ggamma; float* gbeta;   // += (BN)
  float* gbias;            // += sum dy per channel (conv / linear bias), nullable
};
void launch_post_backward(const PostBwdArgs& a, hipStream_t s);

// ---------------------------------------------------------------- criterion / optimiser / misc
void launch_mse(const float* x, const float* t, long n, long n_global, double* loss_dev, float* grad, hipStream_t s);
struct AdamConsts { float b1, b2, c1, c2, eps, step, l1, l2, clamp; int use_penalty, use_clamp; };
void launch_penalty_clamp_adam(float* theta, float* g, float* m, float* v, long n, const AdamConsts& c, hipStream_t s);
void launch_gen_mask(uint32_t* words, long n_elems, float p_drop, uint64_t seed, uint64_t counter, uint32_t layer, hipStream_t s);
void launch_pack_mask(const uint8_t* keep, uint32_t* words, long n, hipStream_t s);
void launch_unpack_mask(const uint32_t* words, uint8_t* keep, long n, hipStream_t s);
void launch_fill_normal(float* dst, long n, uint64_t seed, hipStream_t s);
void launch_l2_distance_rows(const float* a, const float* b, long n, long d, double* out, hipStream_t s);
void launch_scale_copy(const float* src, float* dst, long n, float scale, hipStream_t s);

// ---------------------------------------------------------------- cosine top-k search
size_t cosine_topk_workspace_bytes(long N, int d, int Q, int k);
// idx_out/score_out are DEVICE buffers [Q][k]
int launch_cosine_topk(const float* emb, long N, int d, const long* query_rows_dev, int Q, int k,
                       long* idx_out, float* score_out, int accf, void* workspace, hipStream_t s);

}  // namespace gr

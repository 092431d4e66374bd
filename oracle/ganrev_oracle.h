/*
 * ganrev_oracle.h — CPU ORACLE for the gan-reverser hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of what the reference (aleju/gan-reverser, Lua/Torch7)
 * computes on the path  G forward -> R forward/backward -> L2/clamp/Adam  and the
 * recovered-noise cosine-similarity search.  It exists to CHECK the HIP library; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it.  The product
 * path (gan-reverser_amd/) never links, imports or falls back to anything in oracle/.
 *
 * PARITY UNPINNED vs Torch7: the reference holds no tests / golden vectors for this path and
 * its arithmetic lives in un-vendored, un-pinned luarocks (torch/TH, nn/THNN, optim — circa
 * late 2015 / early 2016, "cudnn3", reference README.md:97) that cannot be built or run in this
 * image (no luajit/th).  The operator definitions below restate the published THNN / nn /
 * optim algorithms of that era; they are cross-checked op-by-op against PyTorch-CPU (the
 * lineal descendant of TH/THNN) by tests/test_oracle_vs_torch.py, and frozen as fixtures in
 * tests/golden/ (generator: tests/golden/make_golden.py).
 *
 * Reference call sites each function follows are cited at its definition.
 */
#ifndef GANREV_ORACLE_H
#define GANREV_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* layer kinds — the module types models.lua:104-143 (G3) and models.lua:389-464 (R) instantiate */
enum {
  GO_CONV3 = 1,        /* nn/cudnn.SpatialConvolution(a=Cin, b=Cout, 3,3,1,1,1,1) */
  GO_BN = 2,           /* nn.(Spatial)BatchNormalization(a=features), eps 1e-5, momentum 0.1, affine */
  GO_ELU = 3,          /* nn.ELU() alpha=1 */
  GO_RELU = 4,         /* cudnn.ReLU(true) */
  GO_LEAKYRELU = 5,    /* nn.LeakyReLU(p=negative slope) */
  GO_SIGMOID = 6,      /* nn.Sigmoid() */
  GO_TANH = 7,         /* nn.Tanh() */
  GO_DROPOUT = 8,      /* nn.Dropout(p, v1) ; flags bit0 = v2 (train-time 1/(1-p) scale), bit1 = always on */
  GO_SPATIAL_DROPOUT = 9, /* nn.SpatialDropout(p) */
  GO_MAXPOOL2 = 10,    /* nn.SpatialMaxPooling(2,2) */
  GO_UPSAMPLE2 = 11,   /* nn.SpatialUpSamplingNearest(2) */
  GO_VIEW = 12,        /* nn.View(a[,b,c]) */
  GO_LINEAR = 13,      /* nn.Linear(a=in, b=out) */
  GO_FULLCONV3 = 14,   /* nn.SpatialFullConvolution(a=Cin, b=Cout, 3,3,1,1,1,1) (north_star extra) */
  /* the D network's extra module types (models.lua:272-337 create_D2) */
  GO_CONVK = 15,       /* nn.SpatialConvolution(a=Cin, b=Cout, c=K, K, 1, 1, (K-1)/2, (K-1)/2), K odd (models.lua:275) */
  GO_PRELU = 16        /* nn.PReLU(): ONE learnable slope shared by every element (nOutputPlane = 0), initial value 0.25 (models.lua:276) */
};
#define GO_DROPOUT_V2 1
#define GO_DROPOUT_ALWAYS_ON 2

typedef struct { int32_t kind, a, b, c; float p; int32_t flags; } go_layer;
typedef struct go_net go_net;

/* ---- single operators (exported so tests can check each against PyTorch-CPU) ---- */
/* convolution implementation of the three go_conv3_* calls: 0 = direct loops (the parity oracle), 1 = im2col + blocked sgemm
 * per sample (oracle_mm.c; THNN SpatialConvolutionMM's structure; the CPU baseline bench.py reports) */
void go_set_conv_impl(int impl);
int go_get_conv_impl(void);
void go_conv3_forward_mm(const float* in, const float* w, const float* bias, float* out, int B, int Cin, int Cout, int H, int W);
void go_conv3_backward_data_mm(const float* gout, const float* w, float* gin, int B, int Cin, int Cout, int H, int W);
void go_conv3_backward_weight_mm(const float* in, const float* gout, float* gw, float* gb, int B, int Cin, int Cout, int H, int W);
void go_conv3_forward(const float* in, const float* w, const float* bias, float* out,
                      int B, int Cin, int Cout, int H, int W);
void go_conv3_backward_data(const float* gout, const float* w, float* gin,
                            int B, int Cin, int Cout, int H, int W);
void go_conv3_backward_weight(const float* in, const float* gout, float* gw, float* gb,
                              int B, int Cin, int Cout, int H, int W); /* accumulates (+=) */
void go_convk_forward(const float* in, const float* w, const float* bias, float* out, int B, int Cin, int Cout, int H, int W, int K);
void go_convk_backward_data(const float* gout, const float* w, float* gin, int B, int Cin, int Cout, int H, int W, int K);
void go_convk_backward_weight(const float* in, const float* gout, float* gw, float* gb, int B, int Cin, int Cout, int H, int W, int K); /* += */
void go_linear_forward(const float* in, const float* w, const float* bias, float* out, int B, int I, int O);
void go_linear_backward_data(const float* gout, const float* w, float* gin, int B, int I, int O);
void go_linear_backward_weight(const float* in, const float* gout, float* gw, float* gb, int B, int I, int O);
void go_bn_forward_train(const float* in, const float* gamma, const float* beta, float* out,
                         float* save_mean, float* save_invstd, float* run_mean, float* run_var,
                         int B, int C, int HW, int groups);
void go_bn_forward_eval(const float* in, const float* gamma, const float* beta, float* out,
                        const float* run_mean, const float* run_var, int B, int C, int HW);
void go_bn_backward_train(const float* in, const float* gout, const float* gamma, float* gin,
                          float* ggamma, float* gbeta, const float* save_mean, const float* save_invstd,
                          int B, int C, int HW, int groups);

/* ---- nn.Sequential restatement ---- */
go_net* go_net_create(const go_layer* layers, int n_layers, int C, int H, int W);
void go_net_destroy(go_net*);
int64_t go_net_param_count(const go_net*);
float* go_net_params(go_net*);         /* flat, getParameters() order (train_r.lua:122) */
float* go_net_grads(go_net*);
int go_net_out_dim(const go_net*, int* C, int* H, int* W);
int go_net_n_bn(const go_net*);
float* go_net_bn_running_mean(go_net*, int bn_index, int* n);
float* go_net_bn_running_var(go_net*, int bn_index, int* n);
void go_net_set_training(go_net*, int training);   /* :training() / :evaluate() */
void go_net_set_bn_groups(go_net*, int groups);    /* data-parallel emulation: per-rank batch statistics */
/* keep-flags (0/1 bytes) of one dropout layer for the next forward; n = B*C*H*W (Dropout) or B*C (SpatialDropout) */
int go_net_set_mask(go_net*, int layer_index, const uint8_t* keep, int64_t n);
int64_t go_net_mask_size(const go_net*, int layer_index, int B);
void go_net_zero_grads(go_net*);
void go_net_set_lean(go_net*, int lean);   /* backward frees each buffer once consumed (memory of full-size runs); layer outputs are gone afterwards */
int go_net_forward(go_net*, const float* in, int B, float* out);
int go_net_backward(go_net*, const float* in, const float* gout, int B, float* gin /*nullable*/);
/* nn.SpatialMaxPooling argmax (0..3 = (dy,dx) scan position) of the last forward: read it / force the next forwards to use a
 * given one (NULL: compute it again).  Parity-test hooks: see oracle_net.c. */
int64_t go_net_get_pool_index(const go_net*, int layer_index, uint8_t* out /*nullable: returns the count*/, int64_t cap);
int go_net_force_pool_index(go_net*, int layer_index, const uint8_t* idx /*nullable*/, int64_t n);
int go_net_force_act_side(go_net* n, int li, const uint8_t* side, int64_t cnt);   /* ReLU / LeakyReLU / PReLU: side of the kink to USE in backward (test hook) */
/* intermediate module outputs, for layer-by-layer debugging of the HIP path */
const float* go_net_layer_output(const go_net*, int layer_index, int64_t* n);

/* nn.MSECriterion (train_r.lua:119,147,150): returns loss; grad nullable */
double go_mse(const float* x, const float* t, int64_t n, float* grad);
double go_mse_scaled(const float* x, const float* t, int64_t n, int64_t n_global, float* grad);
double go_bce(const float* x, const float* t, int64_t n, float* grad /*nullable*/);   /* nn.BCECriterion (sizeAverage) */

typedef struct {
  double lr, beta1, beta2, eps;  /* Lua numbers; optim.adam defaults 1e-3, .9, .999, 1e-8 (train_r.lua:125,170) */
  double l1, l2, clamp;          /* OPT.R_L1=0, OPT.R_L2=1e-4, OPT.R_clamp=1 (train_r.lua:22-24) */
} go_hyper;
/* fevalR's penalty + clamp (train_r.lua:153-165) then optim.adam's update; t is the 1-based step */
void go_penalty_clamp_adam(float* theta, float* g, float* m, float* v, int64_t n, const go_hyper* h, int t,
                           double* penalty_out);
/* one whole iteration of train_r.lua:138-170 (noise given); masks must have been set on rnet */
int go_train_r_step(go_net* gnet, go_net* rnet, const float* noise, int B, const go_hyper* h,
                    float* m, float* v, int t, double* mse_out, float* images_out /*nullable*/);

/* apply_r.lua:396-400 (nn.CosineDistance) */
float go_cosine_similarity(const float* a, const float* b, int d, int accumulate_in_float);
/* apply_r.lua:266-282 search: for each query row, score all N rows, order (score desc, index asc), keep k */
void go_cosine_topk(const float* emb, int64_t N, int d, const int64_t* query_rows, int Q, int k,
                    int64_t* idx_out, float* score_out, int accumulate_in_float);

void go_set_threads(int n);       /* OpenMP threads used by the oracle (torch.setnumthreads, train_r.lua:41-43) */
int go_get_max_threads(void);

/* apply_r.lua:369  torch.dist(images[i], fixedImage): sqrt(sum (a-b)^2), fp32 difference/square, fp64 sum, per row */
void go_l2_distance_rows(const float* a, const float* b, int64_t n, int64_t d, double* out);
/* unsup.kmeans (apply_r.lua:198; restated from memory, see oracle_net.c) and the nearest-centroid loop apply_r.lua:205-217 */
void go_kmeans(const float* x, int64_t N, int d, int k, int niter, float* centroids_inout, float* totalcounts, int32_t* labels);
void go_cosine_assign(const float* x, int64_t N, int d, const float* centroids, int k, int take_min, int32_t* labels, float* sims);

#ifdef __cplusplus
}
#endif
#endif

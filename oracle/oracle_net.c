/*
 * oracle_net.c — element-wise ops, nn.Sequential, the train_r step and the apply_r search of the
 * CPU oracle (TEST INFRASTRUCTURE, see ganrev_oracle.h).  Compiled with -ffp-contract=off so every
 * fp32 operation rounds exactly where the Torch7 tensor op it restates rounds.
 *
 * Follows: models.lua:104-143 (G3), models.lua:389-464 (R), train_r.lua:138-170 (step),
 * apply_r.lua:265-282,396-400 (search), utils/nn_utils.lua:5-33 (batched forward is the caller's loop).
 * Operator definitions: Torch7 nn/THNN/optim of early 2016 (un-vendored; restated from the published
 * algorithms — see the header for the "parity unpinned" statement).
 */
#include "ganrev_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

/* ------------------------------------------------------------------ BatchNormalization
 * THNN BatchNormalization.c: per feature f over n = B*HW elements (accreal = double sums, two-pass variance):
 *   mean = sum/n ; var_sum = sum (x-mean)^2 ; invstd = 1/sqrt(var_sum/n + eps)
 *   running_mean = mom*mean + (1-mom)*running_mean ; running_var = mom*var_sum/(n-1) + (1-mom)*running_var
 *   out = ((x-mean)*invstd)*gamma + beta      (fp32)
 * `groups` > 1 evaluates the statistics over `groups` consecutive batch slices (what P data-parallel
 * ranks with per-rank statistics compute); running stats are then updated group after group.
 */
#define BN_EPS 1e-5
#define BN_MOM 0.1
void go_bn_forward_train(const float* in, const float* gamma, const float* beta, float* out,
                         float* save_mean, float* save_invstd, float* run_mean, float* run_var,
                         int B, int C, int HW, int groups) {
  if (groups < 1) groups = 1;
  const int Bg = B / groups;
#pragma omp parallel for schedule(static)
  for (int c = 0; c < C; ++c)
    for (int g = 0; g < groups; ++g) {
      const long n = (long)Bg * HW;
      double sum = 0;
      for (int b = g * Bg; b < (g + 1) * Bg; ++b) {
        const float* x = in + ((long)b * C + c) * HW;
        for (int p = 0; p < HW; ++p) sum += x[p];
      }
      const double mean = sum / n;
      double vs = 0;
      for (int b = g * Bg; b < (g + 1) * Bg; ++b) {
        const float* x = in + ((long)b * C + c) * HW;
        for (int p = 0; p < HW; ++p) { const double d = x[p] - mean; vs += d * d; }
      }
      const double invstd = 1.0 / sqrt(vs / n + BN_EPS);
      save_mean[g * C + c] = (float)mean;
      save_invstd[g * C + c] = (float)invstd;
      if (run_mean) {
        run_mean[c] = (float)(BN_MOM * mean + (1 - BN_MOM) * run_mean[c]);
        const double unb = vs / (n - 1);
        run_var[c] = (float)(BN_MOM * unb + (1 - BN_MOM) * run_var[c]);
      }
      const float mf = (float)mean, isf = (float)invstd, w = gamma[c], bb = beta[c];
      for (int b = g * Bg; b < (g + 1) * Bg; ++b) {
        const float* x = in + ((long)b * C + c) * HW;
        float* y = out + ((long)b * C + c) * HW;
        for (int p = 0; p < HW; ++p) y[p] = ((x[p] - mf) * isf) * w + bb;
      }
    }
}

void go_bn_forward_eval(const float* in, const float* gamma, const float* beta, float* out,
                        const float* run_mean, const float* run_var, int B, int C, int HW) {
#pragma omp parallel for schedule(static)
  for (int c = 0; c < C; ++c) {
    const float mf = run_mean[c];
    const float isf = (float)(1.0 / sqrt((double)run_var[c] + BN_EPS));
    const float w = gamma[c], bb = beta[c];
    for (int b = 0; b < B; ++b) {
      const float* x = in + ((long)b * C + c) * HW;
      float* y = out + ((long)b * C + c) * HW;
      for (int p = 0; p < HW; ++p) y[p] = ((x[p] - mf) * isf) * w + bb;
    }
  }
}

/* THNN BatchNormalization backward (train): sum = S gout ; dotp = S (x-mean)*gout
 *   gin = (gout - sum/n - (x-mean)*invstd^2*dotp/n) * invstd * gamma ; ggamma += dotp*invstd ; gbeta += sum */
void go_bn_backward_train(const float* in, const float* gout, const float* gamma, float* gin,
                          float* ggamma, float* gbeta, const float* save_mean, const float* save_invstd,
                          int B, int C, int HW, int groups) {
  if (groups < 1) groups = 1;
  const int Bg = B / groups;
#pragma omp parallel for schedule(static)
  for (int c = 0; c < C; ++c)
    for (int g = 0; g < groups; ++g) {
      const long n = (long)Bg * HW;
      const float mean = save_mean[g * C + c], invstd = save_invstd[g * C + c];
      double sum = 0, dotp = 0;
      for (int b = g * Bg; b < (g + 1) * Bg; ++b) {
        const float* x = in + ((long)b * C + c) * HW;
        const float* go = gout + ((long)b * C + c) * HW;
        for (int p = 0; p < HW; ++p) { sum += go[p]; dotp += (double)(x[p] - mean) * go[p]; }
      }
      if (gin) {
        const float k = (float)(dotp * invstd * invstd / n);
        const float gm = (float)(sum / n);
        const float w = gamma[c];
        for (int b = g * Bg; b < (g + 1) * Bg; ++b) {
          const float* x = in + ((long)b * C + c) * HW;
          const float* go = gout + ((long)b * C + c) * HW;
          float* gi = gin + ((long)b * C + c) * HW;
          for (int p = 0; p < HW; ++p) gi[p] = ((go[p] - gm) - (x[p] - mean) * k) * invstd * w;
        }
      }
      if (ggamma) ggamma[c] += (float)(dotp * invstd);
      if (gbeta) gbeta[c] += (float)sum;
    }
}

/* ------------------------------------------------------------------ nn.Sequential */
typedef struct {
  go_layer d;
  int inC, inH, inW, outC, outH, outW;
  int64_t w_off, b_off, w_n, b_n;  /* into flat params (weight then bias; BN gamma then beta) */
  int bn_index;
  float *run_mean, *run_var;       /* BN */
  float *save_mean, *save_invstd;  /* BN train cache, [groups][C] */
  uint8_t* keep; int64_t keep_n;   /* dropout keep flags set by the test */
  uint8_t* pool_idx;               /* maxpool argmax 0..3 */
  uint8_t* forced_idx; int64_t forced_n;  /* test hook: argmax to USE instead of computing it (go_net_force_pool_index) */
  float* out; int64_t out_cap;     /* module.output */
  int alias;                       /* nn.View: output shares the input's storage */
  float* gin; int64_t gin_cap;     /* module.gradInput */
} olayer;

struct go_net {
  int n; olayer* L;
  int C, H, W;
  int64_t n_params;
  float *params, *grads;
  int training, groups, n_bn, lastB;
  int lean;                        /* backward releases every buffer it has consumed (full-size parity runs: 10 instead of 21 GB at cfg3) */
};

static int64_t vol(int c, int h, int w) { return (int64_t)c * h * w; }

go_net* go_net_create(const go_layer* layers, int n_layers, int C, int H, int W) {
  go_net* net = (go_net*)calloc(1, sizeof(go_net));
  net->n = n_layers; net->L = (olayer*)calloc(n_layers, sizeof(olayer));
  net->C = C; net->H = H; net->W = W; net->training = 1; net->groups = 1;
  int c = C, h = H, w = W; int64_t off = 0;
  for (int i = 0; i < n_layers; ++i) {
    olayer* l = &net->L[i]; l->d = layers[i];
    l->inC = c; l->inH = h; l->inW = w; l->bn_index = -1;
    switch (l->d.kind) {
      case GO_CONV3: case GO_FULLCONV3:
        if (l->d.a != c) goto fail;
        l->w_off = off; l->w_n = (int64_t)l->d.a * l->d.b * 9; off += l->w_n;
        l->b_off = off; l->b_n = l->d.b; off += l->b_n; c = l->d.b; break;
      case GO_CONVK:
        if (l->d.a != c || l->d.c < 1 || l->d.c % 2 == 0) goto fail;
        l->w_off = off; l->w_n = (int64_t)l->d.a * l->d.b * l->d.c * l->d.c; off += l->w_n;
        l->b_off = off; l->b_n = l->d.b; off += l->b_n; c = l->d.b; break;
      case GO_PRELU:          /* nn.PReLU(): weight = Tensor(1) (nOutputPlane 0) */
        l->w_off = off; l->w_n = 1; off += 1; l->b_off = off; l->b_n = 0; break;
      case GO_LINEAR:
        if (l->d.a != vol(c, h, w)) goto fail;
        l->w_off = off; l->w_n = (int64_t)l->d.a * l->d.b; off += l->w_n;
        l->b_off = off; l->b_n = l->d.b; off += l->b_n; c = l->d.b; h = 1; w = 1; break;
      case GO_BN:
        if (l->d.a != c) goto fail;
        l->w_off = off; l->w_n = c; off += c; l->b_off = off; l->b_n = c; off += c;
        l->bn_index = net->n_bn++;
        l->run_mean = (float*)calloc(c, sizeof(float));
        l->run_var = (float*)malloc(sizeof(float) * c);
        for (int k = 0; k < c; ++k) l->run_var[k] = 1.f;
        break;
      case GO_MAXPOOL2: h /= 2; w /= 2; break;
      case GO_UPSAMPLE2: h *= 2; w *= 2; break;
      case GO_VIEW:
        if (vol(l->d.a, l->d.b > 0 ? l->d.b : 1, l->d.c > 0 ? l->d.c : 1) != vol(c, h, w)) goto fail;
        c = l->d.a; h = l->d.b > 0 ? l->d.b : 1; w = l->d.c > 0 ? l->d.c : 1; break;
      default: break;
    }
    l->outC = c; l->outH = h; l->outW = w;
  }
  net->n_params = off;
  net->params = (float*)calloc(off > 0 ? off : 1, sizeof(float));
  net->grads = (float*)calloc(off > 0 ? off : 1, sizeof(float));
  return net;
fail:
  go_net_destroy(net);
  return NULL;
}

void go_net_destroy(go_net* net) {
  if (!net) return;
  for (int i = 0; i < net->n; ++i) {
    olayer* l = &net->L[i];
    free(l->run_mean); free(l->run_var); free(l->save_mean); free(l->save_invstd);
    free(l->keep); free(l->pool_idx); free(l->forced_idx); if (!l->alias) free(l->out); free(l->gin);
  }
  free(net->L); free(net->params); free(net->grads); free(net);
}

int64_t go_net_param_count(const go_net* n) { return n->n_params; }
float* go_net_params(go_net* n) { return n->params; }
float* go_net_grads(go_net* n) { return n->grads; }
int go_net_out_dim(const go_net* n, int* C, int* H, int* W) {
  const olayer* l = &n->L[n->n - 1]; *C = l->outC; *H = l->outH; *W = l->outW; return 0;
}
int go_net_n_bn(const go_net* n) { return n->n_bn; }
static olayer* find_bn(go_net* n, int idx) {
  for (int i = 0; i < n->n; ++i) if (n->L[i].bn_index == idx) return &n->L[i];
  return NULL;
}
float* go_net_bn_running_mean(go_net* n, int idx, int* cnt) { olayer* l = find_bn(n, idx); if (!l) return NULL; if (cnt) *cnt = l->inC; return l->run_mean; }
float* go_net_bn_running_var(go_net* n, int idx, int* cnt) { olayer* l = find_bn(n, idx); if (!l) return NULL; if (cnt) *cnt = l->inC; return l->run_var; }
void go_net_set_training(go_net* n, int t) { n->training = t; }
void go_net_set_bn_groups(go_net* n, int g) { n->groups = g < 1 ? 1 : g; }
void go_net_set_lean(go_net* n, int lean) { n->lean = lean; }
void go_net_zero_grads(go_net* n) { memset(n->grads, 0, sizeof(float) * n->n_params); }

int64_t go_net_mask_size(const go_net* n, int li, int B) {
  if (li < 0 || li >= n->n) return -1;
  const olayer* l = &n->L[li];
  if (l->d.kind == GO_DROPOUT) return (int64_t)B * vol(l->inC, l->inH, l->inW);
  if (l->d.kind == GO_SPATIAL_DROPOUT) return (int64_t)B * l->inC;
  return -1;
}
int go_net_set_mask(go_net* n, int li, const uint8_t* keep, int64_t cnt) {
  if (li < 0 || li >= n->n) return -1;
  olayer* l = &n->L[li];
  if (l->d.kind != GO_DROPOUT && l->d.kind != GO_SPATIAL_DROPOUT) return -2;
  free(l->keep); l->keep = (uint8_t*)malloc(cnt); memcpy(l->keep, keep, cnt); l->keep_n = cnt;
  return 0;
}
const float* go_net_layer_output(const go_net* n, int li, int64_t* cnt) {
  if (li < 0 || li >= n->n) return NULL;
  if (cnt) *cnt = (int64_t)n->lastB * vol(n->L[li].outC, n->L[li].outH, n->L[li].outW);
  return n->L[li].out;
}

/* Parity-test hooks for nn.SpatialMaxPooling's argmax.  Two correct fp32 implementations may pick different elements of a
 * window whose two largest inputs differ by rounding noise; a test reads the argmax the oracle took, compares it with the
 * device's, and re-runs the oracle with the device's argmax forced so that every gradient can be held to the strict bar. */
int64_t go_net_get_pool_index(const go_net* n, int li, uint8_t* out, int64_t cap) {
  if (li < 0 || li >= n->n || n->L[li].d.kind != GO_MAXPOOL2 || !n->L[li].pool_idx) return -1;
  const int64_t cnt = (int64_t)n->lastB * vol(n->L[li].outC, n->L[li].outH, n->L[li].outW);
  if (out) { if (cap < cnt) return -2; memcpy(out, n->L[li].pool_idx, cnt); }
  return cnt;
}
int go_net_force_pool_index(go_net* n, int li, const uint8_t* idx, int64_t cnt) {   /* idx == NULL: back to computing it */
  if (li < 0 || li >= n->n || n->L[li].d.kind != GO_MAXPOOL2) return -1;
  olayer* l = &n->L[li];
  free(l->forced_idx); l->forced_idx = NULL; l->forced_n = 0;
  if (idx) { l->forced_idx = (uint8_t*)malloc(cnt); memcpy(l->forced_idx, idx, cnt); l->forced_n = cnt; }
  return 0;
}

/* The same for the kink of nn.ReLU / nn.LeakyReLU: an input within rounding noise of zero may fall on either side in two
 * correct fp32 implementations, and the derivative jumps there (by gout * (1 - slope)).  side[k] = 1: treat input k as
 * positive in the backward pass; NULL: back to the input's own sign. */
int go_net_force_act_side(go_net* n, int li, const uint8_t* side, int64_t cnt) {
  if (li < 0 || li >= n->n || (n->L[li].d.kind != GO_RELU && n->L[li].d.kind != GO_LEAKYRELU && n->L[li].d.kind != GO_PRELU)) return -1;
  olayer* l = &n->L[li];
  free(l->forced_idx); l->forced_idx = NULL; l->forced_n = 0;
  if (side) { l->forced_idx = (uint8_t*)malloc(cnt); memcpy(l->forced_idx, side, cnt); l->forced_n = cnt; }
  return 0;
}

static float* ensure(float** p, int64_t* cap, int64_t n) {
  if (*cap < n) { free(*p); *p = (float*)malloc(sizeof(float) * n); *cap = n; }
  return *p;
}

static int dropout_active(const go_net* net, const olayer* l) {
  return net->training || (l->d.flags & GO_DROPOUT_ALWAYS_ON);
}

int go_net_forward(go_net* net, const float* in, int B, float* out_host) {
  const float* x = in;
  net->lastB = B;
  for (int i = 0; i < net->n; ++i) {
    olayer* l = &net->L[i];
    const int64_t nin = B * vol(l->inC, l->inH, l->inW), nout = B * vol(l->outC, l->outH, l->outW);
    if (l->d.kind == GO_VIEW) { l->alias = 1; l->out = (float*)x; continue; }  /* shares storage */
    float* y = ensure(&l->out, &l->out_cap, nout);
    const int HWi = l->inH * l->inW;
    switch (l->d.kind) {
      case GO_CONV3:
        go_conv3_forward(x, net->params + l->w_off, net->params + l->b_off, y, B, l->inC, l->outC, l->inH, l->inW);
        break;
      case GO_FULLCONV3: {
        /* SpatialFullConvolution(3,3,1,1,1,1).forward == conv backward-data with weight [Cin][Cout][3][3], + bias */
        go_conv3_backward_data(x, net->params + l->w_off, y, B, l->outC, l->inC, l->inH, l->inW);
        const float* bias = net->params + l->b_off; const int HW = l->outH * l->outW;
        for (int b = 0; b < B; ++b) for (int o = 0; o < l->outC; ++o) {
          float* yp = y + ((int64_t)b * l->outC + o) * HW; for (int p = 0; p < HW; ++p) yp[p] += bias[o]; }
        break; }
      case GO_LINEAR:
        go_linear_forward(x, net->params + l->w_off, net->params + l->b_off, y, B, l->d.a, l->d.b);
        break;
      case GO_BN:
        if (net->training) {
          free(l->save_mean); free(l->save_invstd);
          l->save_mean = (float*)malloc(sizeof(float) * net->groups * l->inC);
          l->save_invstd = (float*)malloc(sizeof(float) * net->groups * l->inC);
          go_bn_forward_train(x, net->params + l->w_off, net->params + l->b_off, y, l->save_mean, l->save_invstd,
                              l->run_mean, l->run_var, B, l->inC, HWi, net->groups);
        } else {
          go_bn_forward_eval(x, net->params + l->w_off, net->params + l->b_off, y, l->run_mean, l->run_var, B, l->inC, HWi);
        }
        break;
      case GO_ELU:      /* THNN ELU.c: x <= 0 ? (exp(x)-1)*alpha : x */
        /* (real = float, exp() is the double function and alpha a double literal-typed accreal there: the expression is evaluated in double and rounded once) */
        for (int64_t k = 0; k < nin; ++k) y[k] = x[k] <= 0 ? (float)((exp((double)x[k]) - 1.0) * 1.0) : x[k];
        break;
      case GO_RELU:
        for (int64_t k = 0; k < nin; ++k) y[k] = x[k] > 0 ? x[k] : 0.f;
        break;
      case GO_LEAKYRELU:
        for (int64_t k = 0; k < nin; ++k) y[k] = x[k] > 0 ? x[k] : x[k] * l->d.p;
        break;
      case GO_CONVK:
        go_convk_forward(x, net->params + l->w_off, net->params + l->b_off, y, B, l->inC, l->outC, l->inH, l->inW, l->d.c);
        break;
      case GO_PRELU: {   /* THNN PReLU.c updateOutput, nOutputPlane == 0: x > 0 ? x : w[0] * x */
        const float w0 = net->params[l->w_off];
        for (int64_t k = 0; k < nin; ++k) y[k] = x[k] > 0 ? x[k] : w0 * x[k];
        break; }
      case GO_SIGMOID:
        for (int64_t k = 0; k < nin; ++k) y[k] = (float)(1.0 / (1.0 + exp(-(double)x[k])));      /* THNN Sigmoid.c: 1./(1.+ exp(-x)) - double literals, one rounding */
        break;
      case GO_TANH:
        for (int64_t k = 0; k < nin; ++k) y[k] = (float)tanh((double)x[k]);      /* THNN Tanh.c: tanh(*input) - the double function, one rounding */
        break;
      case GO_DROPOUT: {
        /* nn.Dropout: train: noise~Bernoulli(1-p) [v2: /(1-p)], out = in*noise ; eval: v1 out=in*(1-p), v2 identity */
        const int v2 = l->d.flags & GO_DROPOUT_V2;
        if (dropout_active(net, l)) {
          if (!l->keep || l->keep_n != nin) return -10 - i;
          const float s = v2 ? 1.f / (1.f - l->d.p) : 1.f;
          for (int64_t k = 0; k < nin; ++k) y[k] = x[k] * (l->keep[k] ? s : 0.f);
        } else if (!v2) {
          for (int64_t k = 0; k < nin; ++k) y[k] = x[k] * (1.f - l->d.p);
        } else memcpy(y, x, sizeof(float) * nin);
        break; }
      case GO_SPATIAL_DROPOUT: {
        /* nn.SpatialDropout: train: one Bernoulli(1-p) per (b,c), no rescale ; eval: out = in*(1-p) */
        if (net->training) {
          if (!l->keep || l->keep_n != (int64_t)B * l->inC) return -10 - i;
          for (int64_t bc = 0; bc < (int64_t)B * l->inC; ++bc) {
            const float s = l->keep[bc] ? 1.f : 0.f;
            for (int p = 0; p < HWi; ++p) y[bc * HWi + p] = x[bc * HWi + p] * s;
          }
        } else for (int64_t k = 0; k < nin; ++k) y[k] = x[k] * (1.f - l->d.p);
        break; }
      case GO_MAXPOOL2: {
        /* THNN SpatialMaxPooling 2x2 s2 floor: first strictly-greater wins, scan order (dy,dx) */
        free(l->pool_idx); l->pool_idx = (uint8_t*)malloc(nout);
        const int oh = l->outH, ow = l->outW, ih = l->inH, iw = l->inW;
        for (int64_t bc = 0; bc < (int64_t)B * l->inC; ++bc)
          for (int yy = 0; yy < oh; ++yy) for (int xx = 0; xx < ow; ++xx) {
            const float* s = x + bc * ih * iw + (2 * yy) * iw + 2 * xx;
            float m = -INFINITY; int mi = 0;
            if (l->forced_idx && l->forced_n == nout) { mi = l->forced_idx[bc * oh * ow + yy * ow + xx] & 3; m = s[(mi >> 1) * iw + (mi & 1)]; }
            else for (int t = 0; t < 4; ++t) { const float v = s[(t >> 1) * iw + (t & 1)]; if (v > m) { m = v; mi = t; } }
            y[bc * oh * ow + yy * ow + xx] = m; l->pool_idx[bc * oh * ow + yy * ow + xx] = (uint8_t)mi;
          }
        break; }
      case GO_UPSAMPLE2: {
        const int oh = l->outH, ow = l->outW, ih = l->inH, iw = l->inW;
        for (int64_t bc = 0; bc < (int64_t)B * l->inC; ++bc)
          for (int yy = 0; yy < oh; ++yy) for (int xx = 0; xx < ow; ++xx)
            y[bc * oh * ow + yy * ow + xx] = x[bc * ih * iw + (yy >> 1) * iw + (xx >> 1)];
        break; }
      default: return -1;
    }
    x = y;
  }
  if (out_host) memcpy(out_host, x, sizeof(float) * B * vol(net->L[net->n - 1].outC, net->L[net->n - 1].outH, net->L[net->n - 1].outW));
  return 0;
}

/* nn.Sequential:backward — reverse walk, gradInput then accGradParameters(scale=1) per module */
int go_net_backward(go_net* net, const float* in, const float* gout, int B, float* gin_host) {
  const float* g = gout;
  int g_owner = -1;                                             /* layer whose gin buffer `g` points into */
  for (int i = net->n - 1; i >= 0; --i) {
    olayer* l = &net->L[i];
    const float* x = (i == 0) ? in : net->L[i - 1].out;          /* this module's input */
    const float* yout = l->out;
    const int64_t nin = B * vol(l->inC, l->inH, l->inW);
    const int HWi = l->inH * l->inW;
    if (l->d.kind == GO_VIEW) continue;
    float* gi = ensure(&l->gin, &l->gin_cap, nin);
    switch (l->d.kind) {
      case GO_CONV3:
        if (i > 0 || gin_host) go_conv3_backward_data(g, net->params + l->w_off, gi, B, l->inC, l->outC, l->inH, l->inW);
        go_conv3_backward_weight(x, g, net->grads + l->w_off, net->grads + l->b_off, B, l->inC, l->outC, l->inH, l->inW);
        break;
      case GO_FULLCONV3:
        /* gradInput = conv-forward of gout with the same weight (roles swapped); gradWeight[i][o] += x (x) gout */
        go_conv3_forward(g, net->params + l->w_off, NULL, gi, B, l->outC, l->inC, l->inH, l->inW);
        go_conv3_backward_weight(g, x, net->grads + l->w_off, NULL, B, l->outC, l->inC, l->inH, l->inW);
        { const int HW = l->outH * l->outW;
          for (int o = 0; o < l->outC; ++o) { float s = 0; for (int b = 0; b < B; ++b) { const float* gp = g + ((int64_t)b * l->outC + o) * HW; for (int p = 0; p < HW; ++p) s += gp[p]; } net->grads[l->b_off + o] += s; } }
        break;
      case GO_LINEAR:
        go_linear_backward_data(g, net->params + l->w_off, gi, B, l->d.a, l->d.b);
        go_linear_backward_weight(x, g, net->grads + l->w_off, net->grads + l->b_off, B, l->d.a, l->d.b);
        break;
      case GO_BN:
        if (!net->training) return -2;
        go_bn_backward_train(x, g, net->params + l->w_off, gi, net->grads + l->w_off, net->grads + l->b_off,
                             l->save_mean, l->save_invstd, B, l->inC, HWi, net->groups);
        break;
      case GO_ELU:      /* THNN ELU.c: output <= 0 ? gout*(output+alpha) : gout */
        for (int64_t k = 0; k < nin; ++k) gi[k] = yout[k] <= 0 ? g[k] * (yout[k] + 1.f) : g[k];
        break;
      case GO_RELU:
        if (l->forced_idx && l->forced_n == nin) { for (int64_t k = 0; k < nin; ++k) gi[k] = l->forced_idx[k] ? g[k] : 0.f; break; }
        for (int64_t k = 0; k < nin; ++k) gi[k] = yout[k] > 0 ? g[k] : 0.f;
        break;
      case GO_LEAKYRELU:
        if (l->forced_idx && l->forced_n == nin) { for (int64_t k = 0; k < nin; ++k) gi[k] = l->forced_idx[k] ? g[k] : g[k] * l->d.p; break; }
        for (int64_t k = 0; k < nin; ++k) gi[k] = x[k] > 0 ? g[k] : g[k] * l->d.p;
        break;
      case GO_CONVK:
        if (i > 0 || gin_host) go_convk_backward_data(g, net->params + l->w_off, gi, B, l->inC, l->outC, l->inH, l->inW, l->d.c);
        go_convk_backward_weight(x, g, net->grads + l->w_off, net->grads + l->b_off, B, l->inC, l->outC, l->inH, l->inW, l->d.c);
        break;
      case GO_PRELU: {
        /* THNN PReLU.c, nOutputPlane == 0.  updateGradInput: x > 0 ? gout : w[0] * gout.  accGradParameters: gradWeight[0] +=
         * scale * sum over the elements with x <= 0 of gout * x (upstream sums in `real`; here in double: the parity bar is a
         * tolerance either way).  forced side (test hook): which elements count as positive. */
        const float w0 = net->params[l->w_off];
        const int forced = l->forced_idx && l->forced_n == nin;
        double sum = 0;
        for (int64_t k = 0; k < nin; ++k) {
          const int pos = forced ? l->forced_idx[k] != 0 : x[k] > 0;
          gi[k] = pos ? g[k] : w0 * g[k];
          if (!pos) sum += (double)(g[k] * x[k]);
        }
        net->grads[l->w_off] += (float)sum;
        break; }
      case GO_SIGMOID:
        for (int64_t k = 0; k < nin; ++k) gi[k] = g[k] * (1.f - yout[k]) * yout[k];
        break;
      case GO_TANH:
        for (int64_t k = 0; k < nin; ++k) gi[k] = g[k] * (1.f - yout[k] * yout[k]);
        break;
      case GO_DROPOUT: {
        const int v2 = l->d.flags & GO_DROPOUT_V2;
        if (dropout_active(net, l)) {
          const float s = v2 ? 1.f / (1.f - l->d.p) : 1.f;
          for (int64_t k = 0; k < nin; ++k) gi[k] = g[k] * (l->keep[k] ? s : 0.f);
        } else return -3;
        break; }
      case GO_SPATIAL_DROPOUT:
        if (!net->training) return -3;
        for (int64_t bc = 0; bc < (int64_t)B * l->inC; ++bc) {
          const float s = l->keep[bc] ? 1.f : 0.f;
          for (int p = 0; p < HWi; ++p) gi[bc * HWi + p] = g[bc * HWi + p] * s;
        }
        break;
      case GO_MAXPOOL2: {
        const int oh = l->outH, ow = l->outW, ih = l->inH, iw = l->inW;
        memset(gi, 0, sizeof(float) * nin);
        for (int64_t bc = 0; bc < (int64_t)B * l->inC; ++bc)
          for (int yy = 0; yy < oh; ++yy) for (int xx = 0; xx < ow; ++xx) {
            const int t = l->pool_idx[bc * oh * ow + yy * ow + xx];
            gi[bc * ih * iw + (2 * yy + (t >> 1)) * iw + 2 * xx + (t & 1)] += g[bc * oh * ow + yy * ow + xx];
          }
        break; }
      case GO_UPSAMPLE2: {
        const int oh = l->outH, ow = l->outW, ih = l->inH, iw = l->inW;
        memset(gi, 0, sizeof(float) * nin);
        for (int64_t bc = 0; bc < (int64_t)B * l->inC; ++bc)
          for (int yy = 0; yy < oh; ++yy) for (int xx = 0; xx < ow; ++xx)
            gi[bc * ih * iw + (yy >> 1) * iw + (xx >> 1)] += g[bc * oh * ow + yy * ow + xx];
        break; }
      default: return -1;
    }
    g = gi;
    if (net->lean) {   /* the consumed gradOutput and this module's own output are dead from here on */
      if (g_owner >= 0) { free(net->L[g_owner].gin); net->L[g_owner].gin = NULL; net->L[g_owner].gin_cap = 0; }
      if (!l->alias && i + 1 < net->n && net->L[i + 1].d.kind != GO_VIEW) { free(l->out); l->out = NULL; l->out_cap = 0; }
    }
    g_owner = i;
  }
  if (gin_host) memcpy(gin_host, g, sizeof(float) * B * vol(net->C, net->H, net->W));
  return 0;
}

/* ------------------------------------------------------------------ nn.MSECriterion (sizeAverage)
 * THNN MSECriterion.c: sum (accreal=double) of (x-t)^2, / n ; grad = 2/n * (x-t) */
double go_mse_scaled(const float* x, const float* t, int64_t n, int64_t n_global, float* grad) {
  double s = 0;
  for (int64_t k = 0; k < n; ++k) { const float z = x[k] - t[k]; s += (double)(z * z); }
  if (grad) { const float norm = (float)(2.0 / (double)n_global); for (int64_t k = 0; k < n; ++k) grad[k] = norm * (x[k] - t[k]); }
  return s / (double)n_global;
}
double go_mse(const float* x, const float* t, int64_t n, float* grad) { return go_mse_scaled(x, t, n, n, grad); }

/* nn.BCECriterion, sizeAverage (train.lua:173: CRITERION = nn.BCECriterion(), used by adversarial.lua; THNN BCECriterion.c, EPS 1e-12):
 *   output    = -1/n sum( log(x + EPS) * t + log(1 - x + EPS) * (1 - t) )
 *   gradInput = -1/n * (t - x) / ((1 - x + EPS) * (x + EPS))
 * The C expressions mix float tensors with double literals, so each term is evaluated in double; the sum is kept in double here
 * (THNN's accumulator type changed between revisions; same choice as go_mse). */
double go_bce(const float* x, const float* t, int64_t n, float* grad) {
  const double EPS = 1e-12, norm = 1.0 / (double)n;
  double s = 0;
  for (int64_t k = 0; k < n; ++k) s -= log((double)x[k] + EPS) * (double)t[k] + log(1. - (double)x[k] + EPS) * (1. - (double)t[k]);
  if (grad) for (int64_t k = 0; k < n; ++k) grad[k] = (float)(-norm * ((double)t[k] - (double)x[k]) / ((1. - (double)x[k] + EPS) * ((double)x[k] + EPS)));
  return s * norm;
}

/* ------------------------------------------------------------------ fevalR penalty+clamp, optim.adam
 * train_r.lua:153-165:  g += sign(theta)*L1 + theta*L2 ; g = clamp(g, -c, c)      (fp32 tensor ops)
 * optim/adam.lua (2016): m = m*b1 + (1-b1)*g ; v = v*b2 + (1-b2)*g*g ; denom = sqrt(v)+eps ;
 *                        theta += -(lr*sqrt(1-b2^t)/(1-b1^t)) * m / denom       (step size in Lua doubles)
 * TH op order: cadd r = t + value*src ; addcmul r += value*src1*src2 ; addcdiv r += value*src1/src2. */
void go_penalty_clamp_adam(float* theta, float* g, float* m, float* v, int64_t n, const go_hyper* h, int t,
                           double* penalty_out) {
  const float l1 = (float)h->l1, l2 = (float)h->l2;
  if (h->l1 != 0 || h->l2 != 0) {
    double n1 = 0, n2 = 0;
    for (int64_t k = 0; k < n; ++k) { n1 += fabs((double)theta[k]); n2 += (double)theta[k] * theta[k]; }
    if (penalty_out) *penalty_out = (double)h->l1 * n1 + (double)h->l2 * n2 / 2;   /* norm(.,2)^2/2 */
    for (int64_t k = 0; k < n; ++k) {
      const float sg = theta[k] > 0 ? 1.f : (theta[k] < 0 ? -1.f : 0.f);
      const float pen = sg * l1 + theta[k] * l2;
      g[k] = g[k] + pen;
    }
  } else if (penalty_out) *penalty_out = 0;
  if (h->clamp != 0) {
    const float c = (float)h->clamp;
    for (int64_t k = 0; k < n; ++k) g[k] = g[k] < -c ? -c : (g[k] > c ? c : g[k]);
  }
  const float b1 = (float)h->beta1, b2 = (float)h->beta2, eps = (float)h->eps;
  const float c1 = (float)(1.0 - h->beta1), c2 = (float)(1.0 - h->beta2);
  const double bc1 = 1.0 - pow(h->beta1, t), bc2 = 1.0 - pow(h->beta2, t);
  const float step = (float)(-(h->lr * sqrt(bc2) / bc1));
  for (int64_t k = 0; k < n; ++k) {
    m[k] = m[k] * b1 + c1 * g[k];
    v[k] = v[k] * b2 + (c2 * g[k]) * g[k];
    const float denom = sqrtf(v[k]) + eps;
    theta[k] = theta[k] + (step * m[k]) / denom;
  }
}

/* one iteration of train_r.lua:138-170 with the noise given (createNoiseInputs is an input generator) */
int go_train_r_step(go_net* gnet, go_net* rnet, const float* noise, int B, const go_hyper* h,
                    float* m, float* v, int t, double* mse_out, float* images_out) {
  int gc, gh, gw; go_net_out_dim(gnet, &gc, &gh, &gw);
  int rc_, rh, rw; go_net_out_dim(rnet, &rc_, &rh, &rw);
  const int64_t nimg = (int64_t)B * gc * gh * gw, nd = (int64_t)rc_ * rh * rw;
  float* images = (float*)malloc(sizeof(float) * nimg);
  float* preds = (float*)malloc(sizeof(float) * B * nd);
  float* dfdo = (float*)malloc(sizeof(float) * B * nd);
  go_net_set_training(gnet, 0);                               /* train_r.lua:70  MODEL_G:evaluate() */
  int rc = go_net_forward(gnet, noise, B, images);            /* train_r.lua:139 */
  if (!rc) {
    go_net_set_training(rnet, 1);
    go_net_zero_grads(rnet);                                  /* :143 */
    rc = go_net_forward(rnet, images, B, preds);              /* :146 */
  }
  if (!rc) {
    const double f = go_mse(preds, noise, B * nd, dfdo);      /* :147,150 */
    if (mse_out) *mse_out = f;
    rc = go_net_backward(rnet, images, dfdo, B, NULL);        /* :151 */
  }
  if (!rc) go_penalty_clamp_adam(rnet->params, rnet->grads, m, v, rnet->n_params, h, t, NULL);  /* :153-170 */
  if (images_out && !rc) memcpy(images_out, images, sizeof(float) * nimg);
  free(images); free(preds); free(dfdo);
  return rc;
}

/* ------------------------------------------------------------------ nn.CosineDistance, apply_r.lua:396-400
 * nn/CosineDistance.lua:updateOutput on two 1-D vectors (fp32 tensor ops; :sum accumulates in accreal=double):
 *   buffer=a.*b ; w1=sum(buffer) ; buffer=a.*a ; w22=sum(buffer)+1e-12 ; w22=1/w22 ; (w32 alike from b)
 *   w=w22*w32 ; w=sqrt(w) ; out=w1*w                                                                  */
float go_cosine_similarity(const float* a, const float* b, int d, int accf) {
  float w1, w22, w32;
  if (accf) {
    float s1 = 0, s2 = 0, s3 = 0;
    for (int i = 0; i < d; ++i) { s1 += a[i] * b[i]; s2 += a[i] * a[i]; s3 += b[i] * b[i]; }
    w1 = s1; w22 = s2; w32 = s3;
  } else {
    double s1 = 0, s2 = 0, s3 = 0;
    for (int i = 0; i < d; ++i) { s1 += (double)(a[i] * b[i]); s2 += (double)(a[i] * a[i]); s3 += (double)(b[i] * b[i]); }
    w1 = (float)s1; w22 = (float)s2; w32 = (float)s3;
  }
  w22 = w22 + 1e-12f; w22 = 1.f / w22;
  w32 = w32 + 1e-12f; w32 = 1.f / w32;
  float w = w22 * w32;
  w = sqrtf(w);
  return w1 * w;
}

typedef struct { float s; int64_t j; } cand;
static int cand_cmp(const void* pa, const void* pb) {
  const cand* a = (const cand*)pa; const cand* b = (const cand*)pb;
  if (a->s > b->s) return -1;
  if (a->s < b->s) return 1;
  return a->j < b->j ? -1 : (a->j > b->j ? 1 : 0);
}
/* apply_r.lua:266-282: score every row j against needle row (self included), sort descending, keep first k.
 * The reference's table.sort is unstable; ties are DEFINED here as index-ascending. */
void go_cosine_topk(const float* emb, int64_t N, int d, const int64_t* query_rows, int Q, int k,
                    int64_t* idx_out, float* score_out, int accf) {
  if (k > N) k = (int)N;
  for (int q = 0; q < Q; ++q) {
    cand* c = (cand*)malloc(sizeof(cand) * N);
    const float* a = emb + query_rows[q] * d;
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < N; ++j) { c[j].s = go_cosine_similarity(a, emb + j * d, d, accf); c[j].j = j; }
    qsort(c, N, sizeof(cand), cand_cmp);
    for (int r = 0; r < k; ++r) { idx_out[(int64_t)q * k + r] = c[r].j; if (score_out) score_out[(int64_t)q * k + r] = c[r].s; }
    free(c);
  }
}

/* TH THTensor_(dist)(a, b, 2): sum += pow(fabs(a-b), 2) in accreal (double), result pow(sum, 1/2)   — apply_r.lua:369 */
void go_l2_distance_rows(const float* a, const float* b, int64_t n, int64_t d, double* out) {
  for (int64_t i = 0; i < n; ++i) {
    double s = 0;
    for (int64_t j = 0; j < d; ++j) { const float t = fabsf(a[i * d + j] - b[i * d + j]); s += (double)(t * t); }
    out[i] = sqrt(s);
  }
}

/* ------------------------------------------------------------------ unsup.kmeans, called at apply_r.lua:198
 * unsup is an un-vendored, un-pinned luarock (koraykv/unsup, kmeans.lua).  [upstream, from memory]: per iteration
 *   c2 = sum(pow(centroids,2),2)*0.5 ; tmp = centroids * batch^T - c2 ; val,labels = max(tmp,1) (first maximum) ;
 *   summation += S^T * batch ; counts += sum(S,1) ; centroids[i] = summation[i]:div(counts[i]) where counts[i] ~= 0 ;
 *   totalcounts += counts.
 * Upstream's sgemm / fp32 accumulation order is BLAS-defined; restated with sequential fp32 dot products (no contraction)
 * and fp64 member sums rounded to fp32 once.  The initial centroids (upstream: normal() rows divided by their norm, from
 * Torch's RNG) are an input.  labels (nullable) receives the assignment of the last iteration. */
void go_kmeans(const float* x, int64_t N, int d, int k, int niter, float* cent, float* totalcounts, int32_t* labels) {
  float* c2 = (float*)malloc(sizeof(float) * k);
  double* sum = (double*)malloc(sizeof(double) * (size_t)k * d);
  int64_t* cnt = (int64_t*)malloc(sizeof(int64_t) * k);
  int32_t* lab = labels ? labels : (int32_t*)malloc(sizeof(int32_t) * N);
  for (int j = 0; j < k; ++j) totalcounts[j] = 0.f;
  for (int it = 0; it < niter; ++it) {
    for (int j = 0; j < k; ++j) {
      double s = 0;
      for (int t = 0; t < d; ++t) { const float p = cent[(size_t)j * d + t] * cent[(size_t)j * d + t]; s += (double)p; }
      c2[j] = (float)s * 0.5f;
    }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
      float best = 0.f; int bi = 0;
      for (int j = 0; j < k; ++j) {
        float s = 0.f;
        for (int t = 0; t < d; ++t) s = s + cent[(size_t)j * d + t] * x[i * d + t];
        const float v = s - c2[j];
        if (j == 0 || v > best) { best = v; bi = j; }
      }
      lab[i] = bi;
    }
    memset(sum, 0, sizeof(double) * (size_t)k * d); memset(cnt, 0, sizeof(int64_t) * k);
    for (int64_t i = 0; i < N; ++i) {
      double* sj = sum + (size_t)lab[i] * d;
      for (int t = 0; t < d; ++t) sj[t] += (double)x[i * d + t];
      cnt[lab[i]]++;
    }
    for (int j = 0; j < k; ++j) {
      if (cnt[j] != 0)
        for (int t = 0; t < d; ++t) cent[(size_t)j * d + t] = (float)sum[(size_t)j * d + t] / (float)cnt[j];
      totalcounts[j] += (float)cnt[j];
    }
  }
  free(c2); free(sum); free(cnt); if (!labels) free(lab);
}

/* apply_r.lua:205-217: for every row the centroid with the MINIMUM cosineSimilarity (take_min, what the reference does) or
 * the maximum; the first one on ties */
void go_cosine_assign(const float* x, int64_t N, int d, const float* cent, int k, int take_min, int32_t* labels, float* sims) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < N; ++i) {
    float best = 0.f; int bi = 0;
    for (int j = 0; j < k; ++j) {
      const float s = go_cosine_similarity(x + i * d, cent + (size_t)j * d, d, 0);
      if (j == 0 || (take_min ? s < best : s > best)) { best = s; bi = j; }
    }
    labels[i] = bi; sims[i] = best;
  }
}

/* thread count of the OpenMP loops (the process may have initialised libgomp before OMP_NUM_THREADS could be set) */
void go_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
int go_get_max_threads(void) { return omp_get_max_threads(); }

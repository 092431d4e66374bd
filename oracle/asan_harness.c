/*
 * asan_harness.c — drives the CPU oracle under AddressSanitizer + UBSan (TEST INFRASTRUCTURE; `make -C oracle asan`, run by
 * tests/test_sanitizers.py).  SURVEY.md section 5 (race / memory-error detection of the reference: none; the build plan asks
 * for a sanitizer build of the CPU oracle and host shim).  It walks the paths the parity tests use at ragged sizes: R and G
 * nets (models.lua:389-464, 104-143 layer kinds incl. the fixer's always-on Dropout, both pools, uneven planes), training and
 * evaluate() forwards, backward with and without gradInput, lean mode, both convolution implementations, the D network's 5x5
 * convolution + PReLU, one train_r iteration, BatchNorm in two groups, the search with ties and k > N, k-means, distances.
 * Exit code 0 and no sanitizer report = pass.  The GPU library is never built with sanitizers (gpurun refuses them).
 */
#include "ganrev_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static unsigned long long rng = 88172645463325252ULL;
static float rnd(void) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (float)((rng >> 11) * (1.0 / 9007199254740992.0)) * 2.f - 1.f; }
static float* randv(long n, float scale) { float* p = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1)); for (long i = 0; i < n; ++i) p[i] = rnd() * scale; return p; }
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "asan_harness: check failed: %s (line %d)\n", #c, __LINE__); exit(2); } } while (0)

static void fill_params(go_net* n) { long np = (long)go_net_param_count(n); float* p = go_net_params(n); for (long i = 0; i < np; ++i) p[i] = rnd() * 0.2f; }
static void fill_masks(go_net* n, const go_layer* L, int nl, int B) {
  for (int i = 0; i < nl; ++i) if (L[i].kind == GO_DROPOUT || L[i].kind == GO_SPATIAL_DROPOUT) {
    long m = (long)go_net_mask_size(n, i, B); CHECK(m > 0);
    uint8_t* k = (uint8_t*)malloc((size_t)m); for (long j = 0; j < m; ++j) k[j] = rnd() > (2 * L[i].p - 1);
    CHECK(go_net_set_mask(n, i, k, m) == 0); free(k);
  }
}

static void run_R(int C, int H, int W, int nd, int B, int fixer, int impl, int groups, int lean) {
  go_layer L[40]; int nl = 0;
#define ADD(k, a_, b_, c_, p_, f_) do { L[nl].kind = k; L[nl].a = a_; L[nl].b = b_; L[nl].c = c_; L[nl].p = p_; L[nl].flags = f_; ++nl; } while (0)
  if (fixer) ADD(GO_DROPOUT, 0, 0, 0, 0.5f, GO_DROPOUT_ALWAYS_ON);
  const int ch[6][2] = {{C, 8}, {8, 8}, {8, 8}, {8, 12}, {12, 12}, {12, 12}};
  for (int i = 0; i < 6; ++i) {
    ADD(GO_CONV3, ch[i][0], ch[i][1], 0, 0, 0); ADD(GO_BN, ch[i][1], 0, 0, 0, 0); ADD(GO_ELU, 0, 0, 0, 0, 0);
    if (i == 2) { ADD(GO_MAXPOOL2, 0, 0, 0, 0, 0); ADD(GO_DROPOUT, 0, 0, 0, 0.5f, GO_DROPOUT_V2); }
    else if (i == 5) { ADD(GO_SPATIAL_DROPOUT, 0, 0, 0, 0.25f, 0); ADD(GO_MAXPOOL2, 0, 0, 0, 0, 0); }
    else ADD(GO_DROPOUT, 0, 0, 0, 0.5f, GO_DROPOUT_V2);
  }
  const int feat = 12 * (H / 4) * (W / 4);
  ADD(GO_VIEW, feat, 0, 0, 0, 0); ADD(GO_LINEAR, feat, 16, 0, 0, 0); ADD(GO_BN, 16, 0, 0, 0, 0); ADD(GO_ELU, 0, 0, 0, 0, 0);
  ADD(GO_DROPOUT, 0, 0, 0, 0.5f, GO_DROPOUT_V2); ADD(GO_LINEAR, 16, nd, 0, 0, 0); ADD(GO_TANH, 0, 0, 0, 0, 0);
  go_net* n = go_net_create(L, nl, C, H, W); CHECK(n);
  fill_params(n);
  go_set_conv_impl(impl); go_net_set_bn_groups(n, groups); go_net_set_lean(n, lean);
  float* x = randv((long)B * C * H * W, 1.f); float* out = randv((long)B * nd, 0.f); float* gout = randv((long)B * nd, 0.1f);
  float* gin = randv((long)B * C * H * W, 0.f);
  go_net_set_training(n, 1); fill_masks(n, L, nl, B); go_net_zero_grads(n);
  CHECK(go_net_forward(n, x, B, out) == 0);
  for (int i = 0; i < nl; ++i) if (L[i].kind == GO_MAXPOOL2) { long cnt = (long)go_net_get_pool_index(n, i, NULL, 0); CHECK(cnt > 0);
    uint8_t* pi = (uint8_t*)malloc((size_t)cnt); CHECK(go_net_get_pool_index(n, i, pi, cnt) == cnt); CHECK(go_net_force_pool_index(n, i, pi, cnt) == 0);
    CHECK(go_net_force_pool_index(n, i, NULL, 0) == 0); free(pi); }
  CHECK(go_net_backward(n, x, gout, B, lean ? NULL : gin) == 0);
  if (!lean) { go_net_set_training(n, 0); if (fixer) fill_masks(n, L, 1, B); CHECK(go_net_forward(n, x, B, out) == 0); }
  for (long i = 0; i < (long)B * nd; ++i) CHECK(isfinite(out[i]));
  go_set_conv_impl(0);
  go_net_destroy(n); free(x); free(out); free(gout); free(gin);
}

static void run_G_and_step(int C, int H, int W, int nd, int B) {
  go_layer G[20]; int ng = 0; go_layer* L = G; int nl = 0;
  const int h4 = H / 4, w4 = W / 4;
  ADD(GO_LINEAR, nd, 16 * h4 * w4, 0, 0, 0); ADD(GO_BN, 16 * h4 * w4, 0, 0, 0, 0); ADD(GO_RELU, 0, 0, 0, 0, 0); ADD(GO_VIEW, 16, h4, w4, 0, 0);
  ADD(GO_UPSAMPLE2, 0, 0, 0, 0, 0); ADD(GO_CONV3, 16, 8, 0, 0, 0); ADD(GO_BN, 8, 0, 0, 0, 0); ADD(GO_RELU, 0, 0, 0, 0, 0);
  ADD(GO_UPSAMPLE2, 0, 0, 0, 0, 0); ADD(GO_CONV3, 8, 8, 0, 0, 0); ADD(GO_BN, 8, 0, 0, 0, 0); ADD(GO_RELU, 0, 0, 0, 0, 0);
  ADD(GO_CONV3, 8, C, 0, 0, 0); ADD(GO_SIGMOID, 0, 0, 0, 0, 0);
  ng = nl;
  go_net* g = go_net_create(G, ng, nd, 1, 1); CHECK(g); fill_params(g);
  go_layer R[12]; L = R; nl = 0;
  ADD(GO_CONV3, C, 8, 0, 0, 0); ADD(GO_BN, 8, 0, 0, 0, 0); ADD(GO_ELU, 0, 0, 0, 0, 0); ADD(GO_DROPOUT, 0, 0, 0, 0.5f, GO_DROPOUT_V2); ADD(GO_MAXPOOL2, 0, 0, 0, 0, 0);
  ADD(GO_VIEW, 8 * (H / 2) * (W / 2), 0, 0, 0, 0); ADD(GO_LINEAR, 8 * (H / 2) * (W / 2), nd, 0, 0, 0);
  go_net* r = go_net_create(R, nl, C, H, W); CHECK(r); fill_params(r);
  long np = (long)go_net_param_count(r);
  float* m = (float*)calloc((size_t)np, sizeof(float)); float* v = (float*)calloc((size_t)np, sizeof(float));
  float* noise = randv((long)B * nd, 1.f); float* img = randv((long)B * C * H * W, 0.f);
  go_hyper h = {1e-3, 0.9, 0.999, 1e-8, 1e-5, 1e-4, 1.0};
  for (int t = 1; t <= 2; ++t) { double mse = 0; go_net_set_training(r, 1); fill_masks(r, R, nl, B); CHECK(go_train_r_step(g, r, noise, B, &h, m, v, t, &mse, img) == 0); CHECK(isfinite(mse)); }
  /* G in training mode with backward (the GAN step's G leg) */
  go_net_set_training(g, 1); go_net_zero_grads(g);
  float* gi = randv((long)B * nd, 0.f); float* go = randv((long)B * C * H * W, 0.1f);
  CHECK(go_net_forward(g, noise, B, img) == 0); CHECK(go_net_backward(g, noise, go, B, gi) == 0);
  go_net_destroy(g); go_net_destroy(r); free(m); free(v); free(noise); free(img); free(gi); free(go);
}

static void run_D_pieces(int B) {
  go_layer D[12]; go_layer* L = D; int nl = 0;
  ADD(GO_CONVK, 3, 6, 5, 0, 0); ADD(GO_PRELU, 0, 0, 0, 0, 0); ADD(GO_MAXPOOL2, 0, 0, 0, 0, 0); ADD(GO_FULLCONV3, 6, 4, 0, 0, 0); ADD(GO_BN, 4, 0, 0, 0, 0);
  ADD(GO_LEAKYRELU, 0, 0, 0, 0.333f, 0); ADD(GO_VIEW, 4 * 5 * 7, 0, 0, 0, 0); ADD(GO_LINEAR, 4 * 5 * 7, 1, 0, 0, 0); ADD(GO_SIGMOID, 0, 0, 0, 0, 0);
  go_net* d = go_net_create(D, nl, 3, 10, 14); CHECK(d); fill_params(d);
  float* x = randv((long)B * 3 * 10 * 14, 1.f); float* out = randv(B, 0.f); float* t = randv(B, 0.f); float* g = randv(B, 0.f); float* gin = randv((long)B * 3 * 10 * 14, 0.f);
  for (int i = 0; i < B; ++i) t[i] = i & 1;
  go_net_set_training(d, 1); go_net_zero_grads(d);
  CHECK(go_net_forward(d, x, B, out) == 0);
  CHECK(isfinite(go_bce(out, t, B, g)));
  CHECK(go_net_backward(d, x, g, B, gin) == 0);
  go_net_destroy(d); free(x); free(out); free(t); free(g); free(gin);
}

static void run_search(void) {
  const long N = 777; const int d = 13;
  float* emb = randv(N * d, 1.f);
  memcpy(emb + 500 * d, emb + 7 * d, sizeof(float) * d);                 /* a duplicate row: an exact score tie */
  int64_t q[3] = {7, 0, N - 1}; int64_t idx[3 * 50]; float sc[3 * 50];
  for (int accf = 0; accf < 2; ++accf) { go_cosine_topk(emb, N, d, q, 3, 50, idx, sc, accf); CHECK(idx[0] == 7 && idx[1] == 500); }
  int64_t idx2[3 * 5]; float sc2[3 * 5]; go_cosine_topk(emb, 5, d, q, 2, 5, idx2, sc2, 0);      /* k == N */
  CHECK(fabsf(go_cosine_similarity(emb, emb, d, 0) - 1.f) < 1e-5f);
  double dist[10]; go_l2_distance_rows(emb, emb + 10 * d, 10, d, dist); CHECK(dist[0] >= 0);
  float cent[4 * 13]; memcpy(cent, emb, sizeof cent); float tot[4]; int32_t lab[777];
  go_kmeans(emb, N, d, 4, 3, cent, tot, lab);
  float sims[777]; go_cosine_assign(emb, N, d, cent, 4, 1, lab, sims); go_cosine_assign(emb, N, d, cent, 4, 0, lab, sims);
  free(emb);
}

int main(void) {
  go_set_threads(4);
  for (int impl = 0; impl < 2; ++impl) {
    run_R(1, 8, 8, 6, 5, 0, impl, 1, 0);
    run_R(2, 12, 20, 5, 3, 1, impl, 1, 0);        /* ragged plane, odd batch, fixer */
    run_R(3, 16, 16, 7, 4, 0, impl, 2, 1);        /* BatchNorm in two groups, lean backward */
  }
  run_G_and_step(3, 8, 12, 5, 4);
  run_D_pieces(6);
  run_search();
  printf("asan_harness: ok\n");
  return 0;
}

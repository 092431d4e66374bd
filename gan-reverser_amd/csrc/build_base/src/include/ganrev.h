/*
 * ganrev.h — C ABI of libganrev.so: the MI355X (gfx950) implementation of gan-reverser's hot path
 * (G forward, R forward/backward, L2+clamp+Adam, data-parallel gradient all-reduce, cosine top-k).
 *
 * The reference has no C header: its boundary is the Torch7 nn.Module / nn.Criterion / optim
 * protocol as *used* by the scripts.  Each entry point below cites the reference call it replaces
 * (paths relative to the reference checkout).  Conventions:
 *   - every call returns GR_OK (0) or a negative gr_status; no C++ exception or abort() crosses the ABI;
 *     gr_last_error(ctx) returns the message of the last failure on that context;
 *   - handles are opaque; one gr_ctx per process per GPU; calls on one ctx are serialised by the caller;
 *   - tensors are fp32, contiguous, NCHW (what the reference's host FloatTensors are, train_r.lua:63);
 *   - `*_host` pointers are host memory and the call is synchronous on return (Lua semantics);
 *     `*_dev` pointers are device memory of the ctx's GPU, work is enqueued on the ctx's stream
 *     (gr_stream) and the call returns without waiting.
 *   - no torch types, no HIP types in signatures (streams/pointers travel as void*).
 */
#ifndef GANREV_H
#define GANREV_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  GR_OK = 0,
  GR_ERR_INVALID = -1,      /* bad argument / shape mismatch */
  GR_ERR_UNSUPPORTED = -2,  /* layer sequence or size this build has no kernel for */
  GR_ERR_HIP = -3,          /* HIP runtime error (message has hipGetErrorString) */
  GR_ERR_NO_DEVICE = -4,    /* no gfx950 device: the library has no CPU fallback */
  GR_ERR_COMM = -5,         /* RCCL error */
  GR_ERR_STATE = -6         /* call order (e.g. backward before forward, backward in evaluate mode) */
} gr_status;

/* Module kinds = the nn.* / cudnn.* constructors on the path (models.lua:104-143, 389-464). Numeric values
 * are shared with the test oracle's go_layer so one descriptor list can drive both. */
enum {
  GR_CONV3 = 1,            /* nn/cudnn.SpatialConvolution(a=nInputPlane, b=nOutputPlane, 3,3,1,1,1,1)  models.lua:122,409 */
  GR_BN = 2,               /* nn.SpatialBatchNormalization(a) / nn.BatchNormalization(a)          models.lua:116,410,448 */
  GR_ELU = 3,              /* nn.ELU()                                                           models.lua:411 */
  GR_RELU = 4,             /* cudnn.ReLU(true)                                                   models.lua:117 */
  GR_LEAKYRELU = 5,        /* nn.LeakyReLU(p)            (north_star names it; models.lua:18 dead code) */
  GR_SIGMOID = 6,          /* nn.Sigmoid()                                                       models.lua:133 */
  GR_TANH = 7,             /* nn.Tanh()                                                          models.lua:453 */
  GR_DROPOUT = 8,          /* nn.Dropout(p[,v1])   flags: GR_DROPOUT_V2, GR_DROPOUT_ALWAYS_ON    models.lua:402-405,412,450 */
  GR_SPATIAL_DROPOUT = 9,  /* nn.SpatialDropout(p)                                               models.lua:439 */
  GR_MAXPOOL2 = 10,        /* nn.SpatialMaxPooling(2,2)                                          models.lua:422,440 */
  GR_UPSAMPLE2 = 11,       /* nn.SpatialUpSamplingNearest(2)                                     models.lua:121,127 */
  GR_VIEW = 12,            /* nn.View(a[,b,c])                                                   models.lua:118,446 */
  GR_LINEAR = 13,          /* nn.Linear(a=in, b=out)                                             models.lua:115,447,451 */
  GR_FULLCONV3 = 14,       /* nn.SpatialFullConvolution(a,b,3,3,1,1,1,1) (north_star names it; absent from the reference) */
  /* the module types the D network adds (models.lua:272-337 create_D2, trained by adversarial.lua:37-205) */
  GR_CONVK = 15,           /* nn.SpatialConvolution(a, b, c=K, K, 1, 1, (K-1)/2, (K-1)/2), K = 5           models.lua:275,297 */
  GR_PRELU = 16            /* nn.PReLU(): one learnable slope (nOutputPlane 0) in the flat vector; the host mirror starts it at 0.25  models.lua:276 */
};
#define GR_DROPOUT_V2 1         /* nn.Dropout default: train-time scale 1/(1-p), identity in evaluate() */
#define GR_DROPOUT_ALWAYS_ON 2  /* the fixer's `drop.evaluate = function() end` (models.lua:402-405) */

typedef struct { int32_t kind, a, b, c; float p; int32_t flags; } gr_layer_desc;

typedef struct gr_ctx gr_ctx;
typedef struct gr_net gr_net;

/* ---- context: replaces cutorch.setDevice / cutorch.manualSeed (train_r.lua:58-62) ---- */
int gr_init(int device, gr_ctx** out);
int gr_shutdown(gr_ctx* ctx);
const char* gr_last_error(gr_ctx* ctx);       /* replaces Lua error()/assert messages */
const char* gr_version(void);
void* gr_stream(gr_ctx* ctx);                 /* hipStream_t all work of this ctx is enqueued on */
int gr_synchronize(gr_ctx* ctx);
int gr_device_info(gr_ctx* ctx, char* buf, int buflen);

/* ---- nn.Sequential: models.create_G3 / create_R_default build one of these (models.lua:104,389) ---- */
int gr_net_create(gr_ctx* ctx, const gr_layer_desc* layers, int n_layers, int in_c, int in_h, int in_w, gr_net** out);
int gr_net_destroy(gr_net* net);
int gr_net_out_dim(gr_net* net, int* c, int* h, int* w);
/* m:getParameters() (train_r.lua:122): flat order = modules in sequence order, weight then bias, BN gamma then beta */
int64_t gr_net_param_count(gr_net* net);
int gr_net_get_params(gr_net* net, float* host);
int gr_net_set_params(gr_net* net, const float* host);
int gr_net_get_grads(gr_net* net, float* host);
int gr_net_set_grads(gr_net* net, const float* host);
int gr_net_zero_grads(gr_net* net);                        /* GRAD_PARAMETERS_R:zero()  train_r.lua:143 */
float* gr_net_params_dev(gr_net* net);                     /* device views of the two flat vectors */
float* gr_net_grads_dev(gr_net* net);
/* BN running statistics (module.running_mean / running_var; not part of getParameters) */
int gr_net_n_bn(gr_net* net);
int gr_net_bn_features(gr_net* net, int bn_index);
int gr_net_get_bn_running(gr_net* net, int bn_index, float* mean_host, float* var_host);
int gr_net_set_bn_running(gr_net* net, int bn_index, const float* mean_host, const float* var_host);
/* m:training() / m:evaluate()  (train_r.lua:70,189,222; apply_r.lua:64,94,103) */
int gr_net_set_training(gr_net* net, int training);
/* Dropout noise: production = counter-based Philox keyed (seed, forward-call counter, layer, element)
 * [replaces torch.manualSeed-driven MT19937, train_r.lua:38-39]; tests inject explicit keep flags. */
int gr_net_set_seed(gr_net* net, uint64_t seed);
int64_t gr_net_mask_size(gr_net* net, int layer_index, int batch);       /* elements of that layer's noise tensor */
int gr_net_set_mask(gr_net* net, int layer_index, const uint8_t* keep_host, int64_t n);  /* used by the NEXT forward only */
int gr_net_get_mask(gr_net* net, int layer_index, uint8_t* keep_host, int64_t n);        /* noise of the LAST forward */
/* m:forward(input) -> m.output   (train_r.lua:139,146; utils/nn_utils.lua:18) */
int gr_net_forward_host(gr_net* net, const float* in_host, int batch, float* out_host);
int gr_net_forward_dev(gr_net* net, const float* in_dev, int batch, float* out_dev /*nullable: result stays in m.output*/);
float* gr_net_output_dev(gr_net* net);                     /* m.output (device), valid until the next forward.  After an evaluate()-mode gr_net_forward_dev with a
                                                            * non-null out_dev the last stage wrote out_dev ITSELF (no copy): m.output then IS the caller's buffer and is
                                                            * valid only while the caller keeps out_dev alive (Torch7: the caller's tensor).  Same for gr_net_layer_output
                                                            * of the last layer. */
/* NN_UTILS.forwardBatched(model, input, batchSize) (utils/nn_utils.lua:5-33; apply_r.lua:146,152,153) on device-resident rows:
 * in_dev [rows x in] -> out_dev [rows x out] in chunks of `batch` rows (the last one ragged).  In evaluate() mode each chunk's last
 * kernel writes its rows of out_dev itself: the reference's per-row copy loop (utils/nn_utils.lua:25-28) has no counterpart.
 * m.output afterwards = the last chunk's rows of out_dev. */
int gr_net_forward_batched_dev(gr_net* net, const float* in_dev, int64_t rows, int batch, float* out_dev);
/* apply_r.lua:145-153 as one device-resident pipeline: per chunk of `batch` rows  images = G:forward(noise) (:146), then for each of
 * the n_rnets reverser nets (MODEL_R :152, MODEL_R_FIXER :153)  attributes_k = R_k:forward(images)  written to attr_out_dev[k]
 * [rows x nd_k] - the tables gr_cosine_topk_dev searches (apply_r.lua:265-282).  images_out_dev [rows x C x H x W] is nullable: the images
 * are kept only when the caller needs them (pixel-wise search, fix-faces); otherwise a chunk's images live in G's output buffer
 * until R has read them.  Every net keeps the mode its m:evaluate() / m:training() call set (apply_r.lua:64,94,103: evaluate). */
int gr_embed_dev(gr_net* gnet, gr_net* const* rnets, int n_rnets, const float* noise_dev, int64_t rows, int batch,
                 float* images_out_dev /*nullable*/, float* const* attr_out_dev);
/* m:backward(input, gradOutput) -> m.gradInput ; accumulates into the flat gradient  (train_r.lua:151) */
int gr_net_backward_host(gr_net* net, const float* in_host, const float* grad_out_host, int batch, float* grad_in_host /*nullable*/);
int gr_net_backward_dev(gr_net* net, const float* in_dev, const float* grad_out_dev, int batch, float* grad_in_dev /*nullable*/);
/* nn.SpatialMaxPooling.indices of the last forward (models.lua:422,440): one byte per output element, 0..3 = position in the
 * 2x2 window in scan order (dy, dx); n must be batch * C * Ho * Wo of that layer */
int gr_net_get_pool_index(gr_net* net, int layer_index, uint8_t* host, int64_t n);
/* debugging / layer-by-layer parity: copy out the output of module `layer_index` of the last forward */
int gr_net_layer_output(gr_net* net, int layer_index, float* host, int64_t n);

/* ---- nn.MSECriterion (train_r.lua:119,147,150). n_global = element count the mean is taken over
 * (= n on one GPU; = global batch * nd under data parallelism so a SUM all-reduce reproduces the reference). ---- */
int gr_mse_host(gr_ctx* ctx, const float* x_host, const float* t_host, int64_t n, int64_t n_global, double* loss_out, float* grad_host /*nullable*/);
int gr_mse_dev(gr_ctx* ctx, const float* x_dev, const float* t_dev, int64_t n, int64_t n_global, double* loss_dev /*1 double*/, float* grad_dev /*nullable*/);
/* nn.BCECriterion, sizeAverage = true (train.lua:173: CRITERION = nn.BCECriterion(), used by adversarial.lua; THNN BCECriterion.c, EPS = 1e-12):
 * loss = -1/n sum(log(x + EPS) t + log(1 - x + EPS) (1 - t)), gradInput = -1/n (t - x) / ((1 - x + EPS)(x + EPS)); terms in double. */
int gr_bce_host(gr_ctx* ctx, const float* x_host, const float* t_host, int64_t n, double* loss_out, float* grad_host /*nullable*/);
int gr_bce_dev(gr_ctx* ctx, const float* x_dev, const float* t_dev, int64_t n, double* loss_dev /*1 double*/, float* grad_dev /*nullable*/);

/* ---- fevalR penalty + clamp (train_r.lua:153-165) fused with optim.adam (train_r.lua:125,170) ---- */
typedef struct {
  double lr, beta1, beta2, eps;   /* optim.adam defaults: 1e-3, 0.9, 0.999, 1e-8 */
  double l1, l2, clamp;           /* OPT.R_L1 0, OPT.R_L2 1e-4, OPT.R_clamp 1   (train_r.lua:22-24) */
} gr_hyper;
int gr_adam_step(gr_net* net, const gr_hyper* h, int t /*1-based step; state.t after increment*/);
int gr_adam_reset(gr_net* net);                             /* OPTSTATE = {adam={R={}}}  train_r.lua:125 */
int gr_adam_get_state(gr_net* net, float* m_host, float* v_host);
int gr_adam_set_state(gr_net* net, const float* m_host, const float* v_host);

/* ---- data parallelism (NEW capability required by north_star; the reference is single-GPU, train_r.lua:34,58-62) ---- */
#define GR_COMM_ID_BYTES 128
int gr_comm_unique_id(gr_ctx* ctx, void* id_out /*GR_COMM_ID_BYTES, generated on rank 0, shipped by the host to all ranks*/);
int gr_comm_init(gr_ctx* ctx, const void* id, int nranks, int rank);    /* RCCL communicator over xGMI */
int gr_comm_destroy(gr_ctx* ctx);
int gr_comm_ranks(gr_ctx* ctx, int* nranks, int* rank);
/* Host-exchange hook (tests / bring-up on boxes whose ranks cannot each own a GPU; SURVEY.md section 4: "a fake comm that sums host
 * buffers in-process"): stands in for EVERY collective this context would issue through RCCL - the gradient and loss all-reduce of
 * gr_train_r_step, gr_allreduce_dev, the BatchNorm-statistics exchange of "sync_bn".  fn is called in stream order from inside the
 * library call with a device buffer of `count` elements and must return 0 with the reduction over all ranks in it:
 * kind 0 = fp32 SUM, 1 = fp64 SUM, 2 = uint32 MAX.  It may synchronise (gr_memcpy_d2h / gr_memcpy_h2d on this context are allowed
 * from inside it).  fn = NULL removes the hook.  Mutually exclusive with gr_comm_init. */
typedef int (*gr_exchange_fn)(void* user, void* buf_dev, int64_t count, int kind);
int gr_comm_set_host_exchange(gr_ctx* ctx, int nranks, int rank, gr_exchange_fn fn, void* user);
int gr_allreduce_grads(gr_net* net);                        /* SUM over ranks of the flat gradient; no-op when nranks == 1 */
int gr_allreduce_dev(gr_ctx* ctx, float* buf_dev, int64_t n);
/* all-gather of bytes_per_rank bytes from every rank into recv_dev [nranks x bytes_per_rank], rank order (the sharded search's
 * candidate exchange: apply_r.lua:266-282 over a corpus split across GPUs; one ncclAllGather).  One rank: a copy. */
int gr_allgather_dev(gr_ctx* ctx, const void* send_dev, void* recv_dev, int64_t bytes_per_rank);
int gr_broadcast_params(gr_net* net, int root);             /* make replicas identical before the first step */

/* ---- one whole iteration of train_r.lua:138-170:  images = G:forward(noise) ; R fwd ; MSE ; R bwd ;
 *      [all-reduce] ; L1/L2 + clamp ; Adam.   noise_dev is this rank's shard [batch x nd]; global_batch is the
 *      MSE normaliser's batch (== batch on one GPU).  loss_out receives the (global) un-penalised MSE. ---- */
int gr_train_r_step(gr_net* gnet, gr_net* rnet, const float* noise_dev, int batch, int global_batch,
                    const gr_hyper* h, int t, double* loss_out /*nullable: skipping it avoids a host sync*/);
/* per-phase device times (ms) of the last gr_train_r_step when timing is enabled: [G fwd, R fwd, loss, R bwd, allreduce, adam] */
/* convolution arithmetic (default 2): 1 = "bf16x6": every fp32 operand split into three bf16 terms, six products on
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation — fp32-level error (same parity bars), 2.7x the matrix rate;
 * 2 = "f16x3": every operand tensor scaled by a power of two (its device-tracked max|.| -> [2^14, 2^15)) and split into two
 * fp16 terms (22 significand bits), three products on v_mfma_f32_32x32x16_f16, result scaled back exactly — fp32-level
 * error (same parity bars) with half the MFMAs of bf16x6;
 * 0 = exact fp32 on v_mfma_f32_32x32x2_f32.  Also settable with the environment variable GR_CONV_MODE=f32|bf16x6|f16x3 before gr_init. */
int gr_set_conv_mode(gr_ctx* ctx, int mode);
int gr_get_conv_mode(gr_ctx* ctx);
/* Runtime knobs.  These are ALL the keys the shipping library answers (anything else: GR_ERR_INVALID); environment variables read once in gr_init:
 * GR_CONV_MODE (f32 | bf16x6 | f16x3, default f16x3), GR_RANGE_GUARD (0 | 1), GR_SIDE_WGRAD (-1 | 0 | 1), GR_FUSED_HEAD (0 | 1).  Every other GR_* switch
 * of earlier rounds - A/B controls of variants that lost their measurement, ablation bits that make kernels compute wrong results by design - exists only in
 * the ablation build (make -C gan-reverser_amd/csrc ablate -> libganrev_ablate.so, never loaded by the tests or bench.py).
 *   "p16_min_tiles"  (default 128, process-wide) smallest tile count at which a 3x3 convolution takes the operand-ready (P16) kernels; tests set 1 to
 *                    exercise that path on small shapes
 *   "stack8_min_wgs" (default 128, process-wide) smallest grid at which 8x8 planes are stacked four to a convolution tile; tests force the path
 *   "eval_p16"       (default 1) evaluate()-mode stages hand their output to the next convolution operand-ready - see below
 *   "side_wgrad"     (default -1 = by stage size: on from 2^26 activations) R's convolution weight gradients on a second stream beside the rest of backward;
 *                    bit-identical either way (measured: cfg3 -1.1 %, cfg2 +1.1 %: profiles/r05_ab_side_wgrad_*.txt)
 *   "fused_head"     (default 1) gr_train_r_step runs R's last two stages (Linear -> BatchNormalization -> ELU -> Dropout -> Linear [-> Tanh], models.lua:446-454),
 *                    the criterion (train_r.lua:147-151) and their backward in ONE launch where that wins (at most 4 rows per workgroup of 8 features:
 *                    batch <= 256 at 512 features, and nd <= 32); same operations per value as the stage-by-stage path, sums in another order (1e-6 on the
 *                    loss, 1e-4 of a module's largest entry on the gradients: tests/test_gpu_parity.py::test_head_kernel_equals_the_stage_by_stage_step).
 *                    0 = stage by stage: what gr_net_forward_* / gr_net_backward_* compute, bit for bit.  The launch synchronises its C1 / 8 workgroups with
 *                    two grid barriers, so it is only taken on a device with at least that many CUs; a barrier that still times out (~5 s: the CUs were held
 *                    by other work) sets a sticky device word - the optimiser update of that step and of every later one is skipped (parameters and Adam
 *                    state stay those of the last good step) and the next call that synchronises (gr_train_r_step with loss_out, gr_synchronize,
 *                    gr_net_get_params / gr_net_get_grads) returns GR_ERR_STATE once and re-arms the barrier
 *   "head_fault_inject" (default 0; test hook) 1 = the NEXT head launch waits at its barriers for an arrival count that never comes and gives up after 2^10
 *                    polls: the failure path above, on demand (tests/test_gpu_abi_behaviour.py)
 *   "sync_bn"        (default 0) synchronised BatchNorm under data parallelism - see below
 *   "range_guard"    (default 1) the f16x3 range guard - see below */
int gr_set_tuning(gr_ctx* ctx, const char* key, int value);
/* "eval_p16" (default 1; f16x3 arithmetic): in evaluate() mode (apply_r.lua:120-153: MODEL_R:forward on generated images) a stage hands its output to the
 * next 3x3 convolution as that convolution's operand-ready image (fp16 hi / lo vectors written by the convolution epilogue or the pooling stage's pipeline
 * kernel, scaled by an a-priori bound from the weights' per-channel L1 norms and the measured maximum of the stage's input) instead of as an fp32 tensor.
 * Same 1e-4 parity bar; a pure function of the stage's input and parameters (chunks, batch sizes and the host-memory calls agree bit for bit).  0 keeps
 * the fp32 tensors between the stages (gr_net_layer_output can then read them; a gr_net_backward_* after an evaluate()-mode forward needs them). */
/* "sync_bn" (default 0): synchronised BatchNorm under data parallelism (optional).  With a communicator (or the host-exchange
 * hook) on the context, every training-mode BatchNorm adds its per-channel batch sums over the ranks - (sum y, sum y^2) in the forward,
 * (sum dz, sum dz (y - mean)) in the backward, 2 x C doubles each - before it uses them, so that P ranks of B images compute what ONE
 * device computes on P x B images (models.lua:410-448 on the global batch; running statistics identical on every rank).  Default = per-rank
 * statistics (the oracle's "BatchNorm in P groups").  Needs equal shards: gr_train_r_step checks batch x ranks == global_batch. */
/* f16x3 range guard ("range_guard" 1/0 in gr_set_tuning, default on; GR_RANGE_GUARD=0 before gr_init turns it off).  f16x3 scales a
 * tensor by one power of two: an entry 2^k below the tensor maximum keeps about 40 - k bits, and an output channel's relative
 * error grows with the product of the per-channel spreads of the activation and the weight tensor multiplied.  The host-memory calls
 * (gr_net_forward_host / gr_net_backward_host) measure, before computing, the per-channel spread (log2 largest / smallest
 * non-zero channel maximum) of the input / gradOutput, of every weight tensor an f16x3 kernel reads (per input and per output
 * channel) and of the BatchNorm (gamma, beta) pairs, and run the pass on bf16x6 (fp32 exponent range) when the largest
 * activation-side spread (input, gradOutput, BatchNorm pairs) plus the largest weight-side spread exceed 20 bits.  gr_train_r_step scans the parameters every 64th step without synchronising and
 * switches the context to bf16x6 when a scan trips.  Counters: scan launches and passes sent to bf16x6 since gr_init (the
 * latter also in gr_kernel_times as "range_guard_fallback"). */
int gr_range_guard_stats(gr_ctx* ctx, int64_t* scans, int64_t* fallbacks);
/* The device-pointer calls (gr_net_forward_dev / gr_net_backward_dev) are NOT guarded: host loops built from them (the GAN game,
 * adversarial.lua:139-201 mirrored by ganrev.adversarial.DeviceGame) call this every few dozen batches per net: synchronous scan of
 * the net's weights and BatchNorm scales; a hostile spread keeps the context on bf16x6 (as gr_train_r_step's sampled scan does).
 * tripped_out (nullable): 1 when the context's guard has tripped. */
int gr_range_guard_scan_params(gr_net* net, int* tripped_out);
int gr_debug_stamps(gr_ctx* ctx, void* dev_buf);   /* diagnostic builds: device buffer for in-kernel time stamps (tools/stamps_p16.py) */
int gr_set_timing(gr_ctx* ctx, int mode /*0 off, 1 per-phase events in gr_train_r_step, 2 per-kernel events*/);
/* mode 2: JSON array of {kernel, phase, launches, total_ms, flops, bytes} (algorithmic flops/bytes) accumulated since it was
 * enabled; phase = the part of gr_train_r_step that launched it ("G forward", "R forward", "loss", "R backward", "adam") or "" */
int gr_kernel_times(gr_ctx* ctx, char* buf, int buflen);
int gr_last_step_times(gr_ctx* ctx, float* ms6);
/* HIP events on the ctx's own stream (a host timer or a torch.cuda.Event on another stream does not see this work):
 * gr_event_record marks slot (0..65535) at the current point of the stream; gr_event_elapsed_ms waits for slot b. */
int gr_event_record(gr_ctx* ctx, int slot);
int gr_event_elapsed_ms(gr_ctx* ctx, int slot_a, int slot_b, float* ms);

/* ---- apply_r.lua:265-282 search loop + apply_r.lua:396-400 cosineSimilarity (nn.CosineDistance) ----
 * For each query row q: score every row j of emb[N x d] (self included) with
 *   w1*sqrt(1/(sum a^2+1e-12) * 1/(sum b^2+1e-12)), order by (score desc, index asc), return the first k.
 * accumulate_in_float selects fp32 instead of the default fp64 row-sum accumulation (TH accreal). */
int gr_cosine_topk_host(gr_ctx* ctx, const float* emb_host, int64_t n, int d, const int64_t* query_rows_host, int q, int k,
                        int64_t* idx_out_host, float* score_out_host, int accumulate_in_float);
int gr_cosine_topk_dev(gr_ctx* ctx, const float* emb_dev, int64_t n, int d, const int64_t* query_rows_host, int q, int k,
                       int64_t* idx_out_host, float* score_out_host, int accumulate_in_float);
int gr_cosine_similarity_host(gr_ctx* ctx, const float* a_host, const float* b_host, int d, float* out);
/* Tables of 2^17 rows or more are searched through a bound taken from a strided 16384-row sample (only keys at or above the
 * sample's k-th largest key are kept: same result, bit for bit, without writing n x q keys).  With 32 or more needles (d <= 128)
 * the candidates come from ONE bf16 MFMA GEMM of the table against all needles (approximate cosines, error bound 2^-7 + 2^-10, two
 * cuts with twice that margin) and only they are scored in the exact op order: the result is still bit-identical.  When a table's
 * order defeats the sample (a candidate list overflows) the search runs again on every key.  reruns = how often that
 * happened since gr_init. */
int gr_search_stats(gr_ctx* ctx, int64_t* reruns);

/* ---- apply_r.lua:355-372 (detectAnomalies): out[i] = torch.dist(a[i], b[i]) = sqrt(sum_j (a_ij - b_ij)^2), rows of length d ---- */
int gr_l2_distance_rows_host(gr_ctx* ctx, const float* a_host, const float* b_host, int64_t n, int64_t d, double* out_host);

/* ---- apply_r.lua:197-217 (createClusterImages): clustering of the recovered noise vectors ----
 * gr_kmeans_host replaces `unsup.kmeans(attributes, nbClusters, nbIterations)` (apply_r.lua:198; un-vendored luarock, restated
 * from memory - see csrc/kmeans.hip): centroids_inout [k x d] carries the INITIAL centroids in (upstream draws them from
 * Torch's RNG and normalises each row) and the final ones out; total_counts_out [k] = members summed over the iterations
 * (upstream's second return value); labels_out [n] (nullable) = assignment of the last iteration.  k <= 32, d <= 256.
 * gr_cosine_assign_host replaces the loop apply_r.lua:205-217: for every row the cosine similarity (nn.CosineDistance op order)
 * to each centroid, keeping the MINIMUM when take_min != 0 - what the reference does (`dist < minDist`), although it names the
 * variable a distance - or the maximum otherwise; ties keep the first centroid. */
int gr_kmeans_host(gr_ctx* ctx, const float* x_host, int64_t n, int d, int k, int niter, float* centroids_inout_host,
                   float* total_counts_out_host, int32_t* labels_out_host);
int gr_cosine_assign_host(gr_ctx* ctx, const float* x_host, int64_t n, int d, const float* centroids_host, int k, int take_min,
                          int32_t* labels_out_host, float* sims_out_host);

/* ---- device memory helpers for hosts without a tensor library (LuaJIT FFI, ctypes) ---- */
int gr_malloc(gr_ctx* ctx, int64_t bytes, void** out_dev);
int gr_free(gr_ctx* ctx, void* dev);
int gr_memcpy_h2d(gr_ctx* ctx, void* dst_dev, const void* src_host, int64_t bytes);
int gr_memcpy_d2h(gr_ctx* ctx, void* dst_host, const void* src_dev, int64_t bytes);
/* fill a device buffer with N(0,1) (Philox + Box-Muller): synthetic createNoiseInputs (utils/nn_utils.lua:39-51) for benches */
int gr_fill_normal_dev(gr_ctx* ctx, float* dst_dev, int64_t n, uint64_t seed);
/* the other noise method of createNoiseInputs (utils/nn_utils.lua:44-45): uniform(lo, hi), lo = -1, hi = 1 in the reference */
int gr_fill_uniform_dev(gr_ctx* ctx, float* dst_dev, int64_t n, float lo, float hi, uint64_t seed);

/* ---- nn.Concat (models.lua:293-321, the D network) on device-resident tensors.  The container itself stays host code: it calls
 * its branches' gr_net_forward_dev / gr_net_backward_dev; these move its data without leaving the GPU (ctx stream, asynchronous).
 * gr_copy2d_dev: `rows` rows of `cols` floats between two row-major matrices (pitches in floats): a branch output [B x k] into
 * columns of the joined [B x sum k] output, or a column range of gradOutput into a contiguous [B x k] slice.
 * gr_add_dev: y += x, the sum of the branches' gradInputs (nn.Concat:updateGradInput). */
int gr_copy2d_dev(gr_ctx* ctx, float* dst_dev, int64_t dst_pitch, const float* src_dev, int64_t src_pitch, int64_t rows, int64_t cols);
int gr_add_dev(gr_ctx* ctx, float* y_dev, const float* x_dev, int64_t n);

/* ---- single-kernel entry points used by bench.py's roofline leg and by kernel-level parity tests ---- */
int gr_conv3_forward_dev(gr_ctx* ctx, const float* in_dev, const float* w_dev, const float* bias_dev, float* out_dev,
                         int batch, int cin, int cout, int h, int w, int upsample2);
int gr_conv3_backward_data_dev(gr_ctx* ctx, const float* gout_dev, const float* w_dev, float* gin_dev,
                               int batch, int cin, int cout, int h, int w);
int gr_conv3_backward_weight_dev(gr_ctx* ctx, const float* in_dev, const float* gout_dev, float* gw_dev /*+=*/,
                                 int batch, int cin, int cout, int h, int w);
/* times `iters` launches of the dominant conv kernel (R.conv2 shape by default) with HIP events on the ctx stream */
int gr_bench_conv3(gr_ctx* ctx, int which /*0 fwd,1 bwd-data,2 bwd-weight*/, int batch, int cin, int cout, int h, int w,
                   int iters, float* avg_ms_out);

/* sustained fp32-accurate TFLOP/s of the bare f16x3 inner loop (operands re-read from LDS, three fp16 MFMA products per accumulate, two
 * waves per SIMD, random data) on this device, after a warm-up under load: what the chip's clock under MFMA load leaves of the 833
 * TFLOP/s spec ceiling.  shape 0 = v_mfma_f32_32x32x16_f16 (the convolution kernels' instruction), 1 = v_mfma_f32_16x16x32_f16. */
int gr_bench_mfma_loop(gr_ctx* ctx, int shape, int launches, float* tflops_out);

#ifdef __cplusplus
}
#endif
#endif

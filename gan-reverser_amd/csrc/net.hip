// net.hip — the C ABI (include/ganrev.h) and the nn.Sequential runtime behind it.
//
// A gr_net is the reference's nn.Sequential (models.lua:104-143 G3, models.lua:389-464 R) compiled into
// STAGES:  [UpSample2] (Conv3x3 | Linear) [BN] [act] [Dropout|SpatialDropout] [MaxPool2] [Dropout]
// Each stage runs as: main MFMA kernel -> (training) BN statistics -> one fused per-channel pipeline kernel.
// Backward mirrors it (train_r.lua:151): pipeline backward (two passes around the BN reduction) ->
// weight-gradient kernel -> data-gradient kernel.  All device memory is owned by the net / ctx; there is
// no CPU fallback anywhere in this file.
#include "../../include/ganrev.h"
#include "kernels.h"
#include <rccl/rccl.h>
#include <roctracer/roctx.h>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace gr;

// ------------------------------------------------------------------ context
struct gr_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  ncclComm_t comm = nullptr;
  ncclComm_t stat_comm = nullptr;             // synchronised BatchNorm's own communicator (ncclCommSplit of comm): its collectives run on the COMPUTE stream while the gradient
                                              // buckets run on comm_stream - two streams never share one communicator
  int nranks = 1, rank = 0;
  hipStream_t comm_stream = nullptr;          // gradient buckets are reduced here, behind the rest of backward
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;
  void* ws = nullptr; size_t ws_bytes = 0;
  // weight gradients run beside the rest of backward (backward_impl): their own stream, workspace and events
  hipStream_t side_stream = nullptr; void* ws2 = nullptr; size_t ws2_bytes = 0;
  hipEvent_t ev_dy_ready = nullptr, ev_wgrad_done[2] = {nullptr, nullptr};
  // R's head in one launch (gr_train_r_step, elem.hip head_fwd_bwd_kernel): grid-barrier counter (monotonic) and per-workgroup loss partials
  unsigned* head_bar = nullptr; unsigned head_bar_count = 0; double* head_loss_part = nullptr;
  bool head_unchecked = false;         // a head kernel has been launched since the sticky fault word (d_loss + 48) was last read: head_fault_check
  int head_fault_inject = 0;           // gr_set_tuning "head_fault_inject" (test hook): the NEXT head launch waits at its barriers for an arrival count that never comes
  int cu_count = 0;                    // the grid barrier needs every workgroup of the head kernel resident at once: head_plan refuses devices with fewer CUs than workgroups
  int fused_head = 1;                  // gr_set_tuning "fused_head" (1 default; 0 = the stage-by-stage path: the A/B control and what every other entry point runs)
  hipEvent_t ev_prep_go = nullptr, ev_prep_done = nullptr;    // ablation build: R's per-step preparation on the side stream beside G's forward (GR_PREP_OVERLAP=1; it lost its A/B)
  int side_wgrad = -1;                 // gr_set_tuning "side_wgrad" / GR_SIDE_WGRAD: 1 on, 0 off, -1 (default) by size.  The MFMA kernels take the whole register file of a
                                       // CU (2 waves x 256 VGPRs per SIMD), so nothing becomes resident beside a weight gradient and only kernel tails overlap.  Round 5,
                                       // same box, interleaved (profiles/r05_ab_side_wgrad_*.txt): cfg2 1.976 -> 1.998 ms (slower: the tails are a few us and two streams
                                       // cost an event hand-over per stage), cfg3 12.005 -> 11.875 ms (faster: the slab write + reduction tail of a 0.3 ms launch hides
                                       // behind the data gradient).  Auto = on for stages of >= 2^26 activations (cfg3's layers; cfg2's have 2^24).  Bit-identical
                                       // either way.  Per-kernel timing (gr_set_timing 2) forces it off, so that kernel durations are not inflated by overlap.
  double* d_loss = nullptr;     // device scalar (64-byte block: +0 the loss, +16 the range guard's alarm word, +32 the search's arrival counter, +48 the head kernel's sticky fault word)
  double* h_loss = nullptr;     // pinned host scalar
  bool timing = false;
  int conv_mode = 2;            // 2 = f16x3 split (fp32-accurate, f16 MFMA; default), 1 = bf16x6 split (fp32-accurate, bf16 MFMA), 0 = exact fp32 MFMA
  unsigned* amax = nullptr;     // 4 scratch slots for the single-kernel entry points (f16x3 scales)
  hipEvent_t ev[7] = {};
  std::vector<hipEvent_t> marks;  // gr_event_record slots (bench: per-step times on THIS stream)
  float times[6] = {0, 0, 0, 0, 0, 0};
  // f16x3 range guard (kernels.h "range guard"): the alarm word lives behind the loss scalar (d_loss + 16 bytes,
  // mirrored at h_loss + 16), chmax is the per-channel scratch of the scans
  int range_guard = 1;                 // 1: on (gr_set_tuning "range_guard")
  unsigned* guard_chmax = nullptr; size_t guard_chmax_cap = 0;
  long guard_scans = 0, guard_fallbacks = 0;
  long search_reruns = 0;              // searches whose sample-bound filter overflowed and ran again unfiltered
  void* pin = nullptr; size_t pin_bytes = 0;   // pinned staging for small results (search)
  unsigned* search_state = nullptr;    // device words of the small search path (kernels.h SEARCH_STATE_WORDS)
  unsigned* pin_done = nullptr; unsigned search_seq = 0;   // per-needle completion words of the small search path (pinned, 64 bytes) and the sequence number they carry
  hipEvent_t ev_guard = nullptr; bool guard_pending = false;   // device-resident trainer: sampled scans, verdict read one call later
  bool guard_tripped = false;          // ... which found a hostile range: the context stays on bf16x6
  // synchronised BatchNorm (gr_set_tuning "sync_bn", SURVEY.md 8e optional) and the host-exchange hook that can stand in for RCCL
  int sync_bn = 0;
  double* sync_buf = nullptr; size_t sync_cap = 0;     // compact per-channel pairs that travel through the collective
  gr_exchange_fn xchg = nullptr; void* xchg_user = nullptr;
  int coll_rc = 0;                     // first failure of a collective issued from inside a kernel launcher (StatSync callbacks)
};

// ---- per-kernel event timer (gr_set_timing(ctx, 2)) -------------------------------------------------------------
namespace gr { KernelTimer* g_ktimer = nullptr; }
static int g_eval_p16 = GR_KNOB_SET("GR_NO_EVAL_P16") ? 0 : 1;      // gr_set_tuning "eval_p16"
static int g_kphase = 0;      // which part of gr_train_r_step is launching: 0 outside, 1 G forward, 2 R forward, 3 loss, 4 R backward, 5 Adam
struct EventTimer : gr::KernelTimer {
  struct Rec { std::string name; int phase; double flops, bytes; hipEvent_t e0, e1; bool ok; };
  std::vector<hipEvent_t> pool; size_t next = 0;
  std::vector<Rec> open_, recs;
  struct Agg { long launches = 0; double ms = 0, flops = 0, bytes = 0; };
  std::vector<std::pair<std::pair<std::string, int>, Agg>> agg;      // keyed by (kernel, phase)
  long failed = 0;               // samples whose events could not be created / recorded / read: reported, never counted as 0 ms
  std::string first_error;
  void note(hipError_t e, const char* what) {
    if (e == hipSuccess) return;
    if (first_error.empty()) first_error = std::string(what) + ": " + hipGetErrorString(e);
  }
  hipEvent_t get(bool& ok) {
    if (next == pool.size()) { hipEvent_t e = nullptr; hipError_t r = hipEventCreate(&e); note(r, "hipEventCreate"); if (r != hipSuccess) { ok = false; return nullptr; } pool.push_back(e); }
    return pool[next++];
  }
  void begin(const char* name, double flops, double bytes, hipStream_t s) override {
    Rec r{name, g_kphase, flops, bytes, nullptr, nullptr, true};
    r.e0 = get(r.ok); r.e1 = get(r.ok);
    if (r.ok) { hipError_t e = hipEventRecord(r.e0, s); note(e, "hipEventRecord"); r.ok = e == hipSuccess; }
    open_.push_back(r);
  }
  void end(hipStream_t s) override {
    Rec r = open_.back(); open_.pop_back();
    if (r.ok) { hipError_t e = hipEventRecord(r.e1, s); note(e, "hipEventRecord"); r.ok = e == hipSuccess; }
    recs.push_back(r);
  }
  // Waits for each sample's closing event itself (the kernels may have been launched on ANY context's stream: a caller that
  // synchronises only its own stream used to read unfinished events as 0 ms - VERDICT round 2, the GAN leg's table).
  void collect() {
    for (auto& r : recs) {
      float ms = 0;
      if (r.ok) { hipError_t e = hipEventSynchronize(r.e1); note(e, "hipEventSynchronize"); r.ok = e == hipSuccess; }
      if (r.ok) { hipError_t e = hipEventElapsedTime(&ms, r.e0, r.e1); note(e, "hipEventElapsedTime"); r.ok = e == hipSuccess; }
      if (!r.ok) { failed++; continue; }
      Agg* a = nullptr;
      for (auto& kv : agg) if (kv.first.first == r.name && kv.first.second == r.phase) a = &kv.second;
      if (!a) { agg.push_back({{r.name, r.phase}, Agg()}); a = &agg.back().second; }
      a->launches++; a->ms += ms; a->flops += r.flops; a->bytes += r.bytes;
    }
    recs.clear(); next = 0;
  }
  void reset() { recs.clear(); open_.clear(); agg.clear(); next = 0; failed = 0; first_error.clear(); }
  ~EventTimer() override { for (auto e : pool) (void)hipEventDestroy(e); }
};
static EventTimer* g_evtimer = nullptr;

static int fail(gr_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  if (c) c->err = buf;
  return code;
}
#define HIPCHK(ctx, call)                                                                             \
  do { hipError_t e_ = (call); if (e_ != hipSuccess)                                                  \
      return fail(ctx, GR_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#define NCCLCHK(ctx, call)                                                                            \
  do { ncclResult_t r_ = (call); if (r_ != ncclSuccess)                                               \
      return fail(ctx, GR_ERR_COMM, "%s failed: %s (%s:%d)", #call, ncclGetErrorString(r_), __FILE__, __LINE__); } while (0)
#define LAUNCHCHK(ctx)                                                                                \
  do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess)                                       \
      return fail(ctx, GR_ERR_HIP, "kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

// RCCL reports failures of already-enqueued collectives (a peer that died, a transport error) asynchronously: asked for after
// every group of collectives this library issues, so that a broken communicator surfaces as GR_ERR_COMM in the call that
// issued them (or the next one) instead of a hang in a later stream wait.
static int comm_check(gr_ctx* c) {
  if (!c->comm) return GR_OK;
  ncclResult_t st = ncclSuccess;
  NCCLCHK(c, ncclCommGetAsyncError(c->comm, &st));
  if (st != ncclSuccess && st != ncclInProgress) return fail(c, GR_ERR_COMM, "RCCL asynchronous error: %s", ncclGetErrorString(st));
  return GR_OK;
}
// A small collective in stream order on the COMPUTE stream (BatchNorm statistics sit on the critical path): kind 0 = fp32 SUM,
// 1 = fp64 SUM, 2 = uint32 MAX.  RCCL, or the host-exchange hook (gr_comm_set_host_exchange) when one is installed.
static bool have_peers(gr_ctx* c) { return c->xchg != nullptr || c->comm != nullptr; }
static int small_allreduce(gr_ctx* c, void* buf, long count, int kind) {
  if (c->xchg) {
    const int rc = c->xchg(c->xchg_user, buf, (int64_t)count, kind);
    return rc ? fail(c, GR_ERR_COMM, "host exchange hook failed (%d)", rc) : GR_OK;
  }
  if (!c->comm) return GR_OK;
  NCCLCHK(c, ncclAllReduce(buf, buf, (size_t)count, kind == 0 ? ncclFloat : (kind == 1 ? ncclDouble : ncclUint32), kind == 2 ? ncclMax : ncclSum, c->stat_comm ? c->stat_comm : c->comm, c->stream));
  return GR_OK;
}
// Collective (every rank reaches it at the same point: gr_comm_init with sync_bn already on, or gr_set_tuning "sync_bn" with a communicator): the
// BatchNorm statistics get a communicator of their own.  Without it the statistics all-reduces (compute stream) and the gradient buckets (comm_stream)
// would interleave on ONE communicator from two streams; RCCL serialises that correctly only as long as every rank issues in the same host order -
// true here by construction, but never run with a peer on this pool, so it is not relied on.
static int ensure_stat_comm(gr_ctx* c) {
  if (!c->comm || c->stat_comm || !c->sync_bn) return GR_OK;
  // a split with operations still outstanding on the parent is not supported: gradient buckets / the loss all-reduce of the previous step may be in flight
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->comm_stream) HIPCHK(c, hipStreamSynchronize(c->comm_stream));
  NCCLCHK(c, ncclCommSplit(c->comm, 0, c->rank, &c->stat_comm, nullptr));
  return GR_OK;
}
static int statsync_sum(void* user, double* buf, long count) { gr_ctx* c = static_cast<gr_ctx*>(user); const int r = small_allreduce(c, buf, count, 1); if (r && !c->coll_rc) c->coll_rc = r; return r; }
static int statsync_max(void* user, unsigned* buf, long count) { gr_ctx* c = static_cast<gr_ctx*>(user); const int r = small_allreduce(c, buf, count, 2); if (r && !c->coll_rc) c->coll_rc = r; return r; }
// the StatSync of a stage with C channels and n_local elements per channel on this rank, or null when BatchNorm is per rank
static const StatSync* stat_sync(gr_ctx* c, StatSync& ss, int C, double n_local) {
  if (!c->sync_bn || !have_peers(c)) return nullptr;
  if ((size_t)C * 2 > c->sync_cap) {
    // a failure here must NOT fall back to per-rank statistics: the peers would enter an all-reduce this rank never issues.  The error
    // is parked in coll_rc, which every caller of stat_sync returns before it issues anything else.
    if (c->sync_buf) {
      if (hipStreamSynchronize(c->stream) != hipSuccess) { if (!c->coll_rc) c->coll_rc = fail(c, GR_ERR_HIP, "sync-BN: stream synchronise failed before regrowing the statistics buffer"); return nullptr; }
      (void)hipFree(c->sync_buf); c->sync_buf = nullptr; c->sync_cap = 0;
    }
    const size_t cap = (size_t)(C > 2048 ? C : 2048) * 2;
    if (hipMalloc((void**)&c->sync_buf, sizeof(double) * cap) != hipSuccess) { c->sync_buf = nullptr; if (!c->coll_rc) c->coll_rc = fail(c, GR_ERR_HIP, "sync-BN: statistics buffer allocation failed (%zu bytes)", sizeof(double) * cap); return nullptr; }
    c->sync_cap = cap;
  }
  ss.sum = statsync_sum; ss.max_u32 = statsync_max; ss.user = c; ss.buf = c->sync_buf;
  ss.n_global = n_local * c->nranks; ss.grad_scale = 1.f / (float)c->nranks;     // equal shards (gr_train_r_step checks global_batch)
  return &ss;
}

// roctx range of one phase of gr_train_r_step (shows up in rocprofv3 --marker-trace / the rocprof timeline; a no-op without a tool)
struct PhaseRange { explicit PhaseRange(const char* name) { roctxRangePushA(name); } ~PhaseRange() { roctxRangePop(); } };

static int ensure_ws(gr_ctx* c, size_t bytes) {
  if (bytes <= c->ws_bytes) return GR_OK;
  if (c->ws) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(c->ws)); c->ws = nullptr; c->ws_bytes = 0; }
  bytes = (bytes + (1u << 20)) & ~(size_t)((1u << 20) - 1);
  HIPCHK(c, hipMalloc(&c->ws, bytes));
  c->ws_bytes = bytes;
  return GR_OK;
}
// workspace of the side stream (weight gradients running beside the rest of backward)
static int ensure_ws2(gr_ctx* c, size_t bytes) {
  if (bytes <= c->ws2_bytes) return GR_OK;
  if (c->ws2) { HIPCHK(c, hipStreamSynchronize(c->side_stream)); HIPCHK(c, hipFree(c->ws2)); c->ws2 = nullptr; c->ws2_bytes = 0; }
  bytes = (bytes + (1u << 20)) & ~(size_t)((1u << 20) - 1);
  HIPCHK(c, hipMalloc(&c->ws2, bytes));
  c->ws2_bytes = bytes;
  return GR_OK;
}

// The head kernel's sticky fault word (elem.hip head_grid_barrier).  Read - one 4-byte copy, only when a head kernel has run since the last look - by every
// call that synchronises the stream anyway: gr_train_r_step with a loss_out, gr_synchronize, gr_net_get_params / gr_net_get_grads.  Set means: a grid
// barrier timed out in some step since then, that step and every later one skipped its optimiser update (penalty_clamp_adam_kernel), so the parameters and
// Adam's moments are those of the last good step (BatchNorm running statistics and the gradient vector are not rolled back).  Reported ONCE as GR_ERR_STATE;
// the barrier state is reset so that training can go on.
static unsigned* head_fault_dev(gr_ctx* c) { return reinterpret_cast<unsigned*>(reinterpret_cast<char*>(c->d_loss) + 48); }
static int head_fault_check(gr_ctx* c) {
  if (!c->head_unchecked) return GR_OK;
  unsigned w = 0;
  HIPCHK(c, hipMemcpyAsync(&w, head_fault_dev(c), sizeof w, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->head_unchecked = false;
  if (!w) return GR_OK;
  HIPCHK(c, hipMemsetAsync(head_fault_dev(c), 0, sizeof(unsigned), c->stream));
  if (c->head_bar) HIPCHK(c, hipMemsetAsync(c->head_bar, 0, 256, c->stream));
  c->head_bar_count = 0;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return fail(c, GR_ERR_STATE, "head kernel: a grid barrier timed out (its workgroups were not resident together); every optimiser update since then was skipped - "
              "parameters and Adam state are those of the last good step.  gr_set_tuning \"fused_head\" 0 selects the stage-by-stage path");
}

extern "C" const char* gr_version(void) { return "ganrev-gfx950 0.6 (round 6)"; }

extern "C" int gr_init(int device, gr_ctx** out) {
  if (!out) return GR_ERR_INVALID;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return GR_ERR_NO_DEVICE;
  if (device < 0 || device >= n) return GR_ERR_INVALID;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, device) != hipSuccess) return GR_ERR_HIP;
  if (strncmp(p.gcnArchName, "gfx950", 6) != 0) return GR_ERR_NO_DEVICE;  // kernels are built for gfx950 only
  gr_ctx* c = new gr_ctx();
  c->device = device;
  c->cu_count = p.multiProcessorCount;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc((void**)&c->d_loss, 64) != hipSuccess || hipMalloc((void**)&c->amax, sizeof(unsigned) * 4 * AMAX_WORDS) != hipSuccess || hipHostMalloc((void**)&c->h_loss, 64) != hipSuccess) {
    delete c; return GR_ERR_HIP;
  }
  for (auto& e : c->ev) (void)hipEventCreate(&e);
  (void)hipEventCreateWithFlags(&c->ev_guard, hipEventDisableTiming);
  (void)hipMemsetAsync(c->d_loss, 0, 64, c->stream);
  memset(c->h_loss, 0, 64);
  { const char* d = getenv("GR_RANGE_GUARD"); if (d) c->range_guard = atoi(d); }
  gr::g_p16_debug = GR_KNOB("GR_P16_DEBUG", 0);      // diagnostic ablations (probe build only: tools/ablate_p16.py)
  { const char* m = getenv("GR_CONV_MODE"); if (m) c->conv_mode = (!strcmp(m, "f32") || !strcmp(m, "0")) ? 0 : ((!strcmp(m, "bf16x6") || !strcmp(m, "1")) ? 1 : 2); }
  (void)hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking);
  (void)hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking);
  (void)hipEventCreateWithFlags(&c->ev_dy_ready, hipEventDisableTiming);
  (void)hipEventCreateWithFlags(&c->ev_prep_go, hipEventDisableTiming); (void)hipEventCreateWithFlags(&c->ev_prep_done, hipEventDisableTiming);
  for (auto& e : c->ev_wgrad_done) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
  { const char* e = getenv("GR_SIDE_WGRAD"); if (e) c->side_wgrad = atoi(e); }
  { const char* e = getenv("GR_FUSED_HEAD"); if (e) c->fused_head = atoi(e) != 0; }
  (void)hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming);
  (void)hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming);
  *out = c;
  return GR_OK;
}

extern "C" int gr_shutdown(gr_ctx* c) {
  if (!c) return GR_ERR_INVALID;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);      // gradient buckets in flight: before the communicators go
  if (c->stat_comm) { ncclCommDestroy(c->stat_comm); c->stat_comm = nullptr; }
  if (c->comm) { ncclCommDestroy(c->comm); c->comm = nullptr; }
  if (c->ws) (void)hipFree(c->ws);
  if (c->side_stream) { (void)hipStreamSynchronize(c->side_stream); (void)hipStreamDestroy(c->side_stream); }
  if (c->ws2) (void)hipFree(c->ws2);
  if (c->ev_dy_ready) (void)hipEventDestroy(c->ev_dy_ready);
  if (c->head_bar) (void)hipFree(c->head_bar);
  if (c->head_loss_part) (void)hipFree(c->head_loss_part);
  if (c->ev_prep_go) (void)hipEventDestroy(c->ev_prep_go);
  if (c->ev_prep_done) (void)hipEventDestroy(c->ev_prep_done);
  for (auto& e : c->ev_wgrad_done) if (e) (void)hipEventDestroy(e);
  if (c->guard_chmax) (void)hipFree(c->guard_chmax);
  if (c->sync_buf) (void)hipFree(c->sync_buf);
  if (c->pin) (void)hipHostFree(c->pin);
  if (c->pin_done) (void)hipHostFree(c->pin_done);
  if (c->search_state) (void)hipFree(c->search_state);
  if (c->ev_guard) (void)hipEventDestroy(c->ev_guard);
  (void)hipFree(c->d_loss); (void)hipFree(c->amax); (void)hipHostFree(c->h_loss);
  for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : c->marks) if (e) (void)hipEventDestroy(e);
  if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
  if (c->ev_done) (void)hipEventDestroy(c->ev_done);
  if (c->comm_stream) { (void)hipStreamSynchronize(c->comm_stream); (void)hipStreamDestroy(c->comm_stream); }
  (void)hipStreamDestroy(c->stream);
  delete c;
  return GR_OK;
}
extern "C" const char* gr_last_error(gr_ctx* c) { return c ? c->err.c_str() : "null ctx"; }
extern "C" void* gr_stream(gr_ctx* c) { return c ? (void*)c->stream : nullptr; }
extern "C" int gr_synchronize(gr_ctx* c) { if (!c) return GR_ERR_INVALID; HIPCHK(c, hipStreamSynchronize(c->stream)); return head_fault_check(c); }
extern "C" int gr_device_info(gr_ctx* c, char* buf, int n) {
  if (!c || !buf) return GR_ERR_INVALID;
  hipDeviceProp_t p; HIPCHK(c, hipGetDeviceProperties(&p, c->device));
  int rv = 0; (void)hipRuntimeGetVersion(&rv);
  snprintf(buf, n, "arch=%s CUs=%d clock_khz=%d mem_mb=%zu hip_runtime=%d", p.gcnArchName, p.multiProcessorCount, p.clockRate,
           p.totalGlobalMem >> 20, rv);
  return GR_OK;
}
extern "C" int gr_set_conv_mode(gr_ctx* c, int mode) {
  if (!c || mode < 0 || mode > 2) return GR_ERR_INVALID;
  c->conv_mode = mode;
  return GR_OK;
}
extern "C" int gr_get_conv_mode(gr_ctx* c) { return c ? c->conv_mode : GR_ERR_INVALID; }
// diagnostic: give the P16 kernels a device buffer (32 x 8 bytes per workgroup) for in-kernel time stamps ("p16_debug" bit 32)
extern "C" int gr_debug_stamps(gr_ctx* c, void* dev_buf) { if (!c) return GR_ERR_INVALID; gr::g_p16_stamps = dev_buf; return GR_OK; }
extern "C" int gr_set_tuning(gr_ctx* c, const char* key, int value) {
  if (!c || !key) return GR_ERR_INVALID;
  if (!strcmp(key, "p16_min_tiles")) { gr::g_p16_min_tiles = value; return GR_OK; }
  if (!strcmp(key, "stack8_min_wgs")) { gr::g_stack8_min_wgs = value; return GR_OK; }      // four 8x8 images per convolution tile from this many workgroups on (default 128)
#ifdef GR_ABLATE   // ablation build only (make ablate): variants that lost their A/B, and ablation bits that make kernels compute wrong results by design
  if (!strcmp(key, "p16_stagger")) { gr::g_p16_stagger = value; return GR_OK; }
  if (!strcmp(key, "p16_variant")) { gr::g_p16_variant = value; return GR_OK; }
  if (!strcmp(key, "p16_debug")) { gr::g_p16_debug = value; return GR_OK; }
  if (!strcmp(key, "up2_debug")) { gr::g_up2_debug = value; return GR_OK; }     // diagnostic ablations of the four-wave up-sampling kernel
  if (!strcmp(key, "nt_stores")) { gr::g_nt_stores = value; return GR_OK; }        // bit mask: which kernels store their outputs non-temporally (kernels.h)
  if (!strcmp(key, "up2_stagger")) { gr::g_up2_stagger = value; return GR_OK; }   // start delay of a CU's second workgroup, x 512 clocks
  if (!strcmp(key, "up2_quad")) { gr::g_up2_quad = value; return GR_OK; }       // 1 (default): four-wave up-sampling kernel where it applies; 0: eight-wave     // diagnostic ablations (outputs are then wrong by design)
#endif
  if (!strcmp(key, "eval_p16")) { g_eval_p16 = value; return GR_OK; }           // evaluate()-mode stages hand their output over operand-ready (1, default) or as fp32 (0: the A/B control)
  if (!strcmp(key, "side_wgrad")) { c->side_wgrad = value; return GR_OK; }
  if (!strcmp(key, "fused_head")) { c->fused_head = value != 0; return GR_OK; }   // gr_train_r_step: R's last two stages + the criterion, forward and backward, in one launch
  if (!strcmp(key, "head_fault_inject")) { c->head_fault_inject = value != 0; return GR_OK; }   // test hook: the next head launch's grid barriers time out (after 2^10 polls), as on a device that cannot hold its workgroups together
  // synchronised BatchNorm under data parallelism: per-channel batch sums are all-reduced, fwd and bwd (include/ganrev.h)
  if (!strcmp(key, "sync_bn")) { c->sync_bn = value != 0; return ensure_stat_comm(c); }   // (collective when a communicator exists: every rank sets it at the same point)
  // f16x3 range guard on / off.  Off also clears a tripped trainer guard AND drops a verdict still in flight: that verdict is about the
  // nets scanned by an earlier gr_train_r_step and must not trip the context under whoever trains next on it.
  if (!strcmp(key, "range_guard")) { c->range_guard = value; if (!value) { c->guard_tripped = false; c->guard_pending = false; } return GR_OK; }
  return fail(c, GR_ERR_INVALID, "gr_set_tuning: unknown key %s", key);
}
extern "C" int gr_range_guard_stats(gr_ctx* c, int64_t* scans, int64_t* fallbacks) {
  if (!c) return GR_ERR_INVALID;
  if (scans) *scans = c->guard_scans;
  if (fallbacks) *fallbacks = c->guard_fallbacks;
  return GR_OK;
}
extern "C" int gr_search_stats(gr_ctx* c, int64_t* reruns) { if (!c) return GR_ERR_INVALID; if (reruns) *reruns = c->search_reruns; return GR_OK; }
extern "C" int gr_set_timing(gr_ctx* c, int en) {
  if (!c) return GR_ERR_INVALID;
  c->timing = en == 1;
  if (en == 2) {
    if (!g_evtimer) g_evtimer = new EventTimer();
    g_evtimer->reset();
    gr::g_ktimer = g_evtimer;
  } else {
    gr::g_ktimer = nullptr;
  }
  return GR_OK;
}
extern "C" int gr_kernel_times(gr_ctx* c, char* buf, int buflen) {
  if (!c || !buf || buflen < 64) return GR_ERR_INVALID;
  if (!g_evtimer) { snprintf(buf, buflen, "[]"); return GR_OK; }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  g_evtimer->collect();
  std::string out = "[";
  for (size_t i = 0; i < g_evtimer->agg.size(); ++i) {
    const auto& kv = g_evtimer->agg[i];
    char line[512];
    static const char* phase_names[] = {"", "G forward", "R forward", "loss", "R backward", "adam"};
    snprintf(line, sizeof line, "%s{\"kernel\": \"%s\", \"phase\": \"%s\", \"launches\": %ld, \"total_ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e}",
             i ? ", " : "", kv.first.first.c_str(), phase_names[kv.first.second], kv.second.launches, kv.second.ms, kv.second.flops, kv.second.bytes);
    out += line;
  }
  if (g_evtimer->failed > 0) {       // samples the timer could not read: a row of their own, so a table with holes says so
    char line[512];
    std::string esc;                 // the message comes from hipGetErrorString: keep the table parseable whatever it holds
    for (unsigned char ch : g_evtimer->first_error) {
      if (ch == '"' || ch == '\\') { esc += '\\'; esc += (char)ch; }
      else if (ch < 0x20) esc += ' ';
      else esc += (char)ch;
    }
    if (esc.size() > 300) esc.resize(300);
    snprintf(line, sizeof line, "%s{\"kernel\": \"timer_failed_samples\", \"phase\": \"%s\", \"launches\": %ld, \"total_ms\": 0.0, \"flops\": 0.0, \"bytes\": 0.0}",
             g_evtimer->agg.empty() ? "" : ", ", esc.c_str(), g_evtimer->failed);
    out += line;
  }
  if (c->guard_fallbacks > 0) {      // passes the f16x3 range guard sent to bf16x6 since gr_init (not a kernel: a count)
    char line[256];
    snprintf(line, sizeof line, "%s{\"kernel\": \"range_guard_fallback\", \"phase\": \"\", \"launches\": %ld, \"total_ms\": 0.0, \"flops\": 0.0, \"bytes\": 0.0}",
             (g_evtimer->agg.empty() && g_evtimer->failed == 0) ? "" : ", ", c->guard_fallbacks);
    out += line;
  }
  out += "]";
  if ((int)out.size() + 1 > buflen) return fail(c, GR_ERR_INVALID, "gr_kernel_times: buffer too small (%zu needed)", out.size() + 1);
  memcpy(buf, out.c_str(), out.size() + 1);
  return GR_OK;
}
extern "C" int gr_event_record(gr_ctx* c, int slot) {
  if (!c || slot < 0 || slot >= (1 << 16)) return GR_ERR_INVALID;
  if ((size_t)slot >= c->marks.size()) c->marks.resize((size_t)slot + 1, nullptr);
  if (!c->marks[slot]) HIPCHK(c, hipEventCreate(&c->marks[slot]));
  HIPCHK(c, hipEventRecord(c->marks[slot], c->stream));
  return GR_OK;
}
extern "C" int gr_event_elapsed_ms(gr_ctx* c, int a, int b, float* ms) {
  if (!c || !ms || a < 0 || b < 0 || (size_t)a >= c->marks.size() || (size_t)b >= c->marks.size() || !c->marks[a] || !c->marks[b]) return GR_ERR_INVALID;
  HIPCHK(c, hipEventSynchronize(c->marks[b]));
  HIPCHK(c, hipEventElapsedTime(ms, c->marks[a], c->marks[b]));
  return GR_OK;
}
extern "C" int gr_last_step_times(gr_ctx* c, float* ms6) { if (!c || !ms6) return GR_ERR_INVALID; memcpy(ms6, c->times, sizeof c->times); return GR_OK; }

extern "C" int gr_malloc(gr_ctx* c, int64_t bytes, void** out) { if (!c || !out) return GR_ERR_INVALID; HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, hipMalloc(out, bytes > 0 ? bytes : 1)); return GR_OK; }
extern "C" int gr_free(gr_ctx* c, void* p) { if (!c) return GR_ERR_INVALID; HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(p)); return GR_OK; }
extern "C" int gr_memcpy_h2d(gr_ctx* c, void* d, const void* h, int64_t b) { if (!c) return GR_ERR_INVALID; HIPCHK(c, hipMemcpyAsync(d, h, b, hipMemcpyHostToDevice, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream)); return GR_OK; }
extern "C" int gr_memcpy_d2h(gr_ctx* c, void* h, const void* d, int64_t b) { if (!c) return GR_ERR_INVALID; HIPCHK(c, hipMemcpyAsync(h, d, b, hipMemcpyDeviceToHost, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream)); return GR_OK; }
extern "C" int gr_fill_normal_dev(gr_ctx* c, float* d, int64_t n, uint64_t seed) { if (!c || !d) return GR_ERR_INVALID; launch_fill_normal(d, n, seed, c->stream); LAUNCHCHK(c); return GR_OK; }

extern "C" int gr_fill_uniform_dev(gr_ctx* c, float* d, int64_t n, float lo, float hi, uint64_t seed) { if (!c || !d) return GR_ERR_INVALID; launch_fill_uniform(d, n, lo, hi, seed, c->stream); LAUNCHCHK(c); return GR_OK; }

// ------------------------------------------------------------------ net
enum { ST_CONV = 1, ST_LINEAR = 2, ST_ELEM = 3 };

struct MaskSlot {
  int layer = -1, kind = MASK_NONE; float p = 0; int flags = 0;
  int C = 0, H = 0, W = 0;           // tensor the noise is drawn for (per sample)
  uint32_t* bits = nullptr; size_t words_cap = 0;
  bool injected = false; int64_t n_last = 0;
};

struct Stage {
  int kind = 0, first = 0, last = 0, main_layer = -1;
  int inC = 0, inH = 0, inW = 0;
  bool up = false, fullconv = false;
  int Cin = 0, Cout = 0, H = 0, W = 0;       // main-op output channels / spatial dims (ELEM: the input dims)
  int64_t w_off = -1, b_off = -1;
  bool has_bn = false; int64_t g_off = -1, be_off = -1; int bn_idx = -1;
  int act = ACT_NONE; float slope = 0;
  int ksz = 3;                              // window of the main convolution: 3 (conv.hip kernels) or an odd K convk.hip covers (GR_CONVK)
  int64_t slope_off = -1;                   // nn.PReLU: offset of its one learnable slope in the flat vectors
  int m1 = -1, m2 = -1; bool pool = false;
  bool has_post = false;
  int outC = 0, outH = 0, outW = 0;
  float *y = nullptr, *out = nullptr; uint8_t* pool_idx = nullptr;
  float *wt_fwd = nullptr, *wt_bwd = nullptr; uint64_t wt_version = 0;
  void *ws_fwd = nullptr, *ws_bwd = nullptr; uint64_t ws_version = 0;     // bf16x6 / f16x3 split images
  void* ws_up = nullptr; uint64_t ws_up_version = 0;   // f16x3 image of the fused up-sampling kernel (four 2x2 convolutions)
  uint64_t amax_x_fwd = 0;                  // gr_net::amax_gen at which amax_x was last taken
  unsigned *amax_x = nullptr, *amax_dy = nullptr, *amax_w = nullptr;   // f16x3: slots (in gr_net::amax) for max|x_in|, max|dy|, max|w|
  unsigned *amax_y = nullptr, *amax_kb = nullptr, *amax_dz = nullptr;  // max|y| (raw main-op output), the backward bound factor K (BnBounds), max|dz|
  void* x_p16 = nullptr; size_t x_p16_cap = 0;   // operand-ready copy of this stage's INPUT, written by the previous stage's pipeline kernel
  uint64_t x_p16_gen = 0;                         // gr_net::amax_gen at which x_p16 (and the bound in amax_x) was written
  // evaluate() mode (round 4): a convolution EPILOGUE writes the next stage's x_p16, scaled by an a-priori weight-norm bound in amax_x
  // (launch_eval_bound); the true max|x| it measures while storing goes to amax_xt and feeds the bound of the stage after
  unsigned* amax_xt = nullptr; const unsigned* x_true = nullptr;   // x_true: the slot that holds (a tight bound of) the TRUE maximum of what x_p16 holds
  float* wl1 = nullptr; uint64_t wl1_version = 0;                  // per-output-channel L1 norms of the weights (conv3x3 stages)
  uint64_t kb_gen = 0;                            // ... at which amax_kb / amax_y were written (operand-ready dy possible in the backward)
  float *mean = nullptr, *invstd = nullptr, *coef = nullptr; double* partials = nullptr; double* partials_b = nullptr;
  int stat_tiles_last = 0;                  // tiles the last forward's conv epilogue wrote (0: none - run the statistics pass)
  double* stat_part = nullptr;              // per-tile (sum, sum of squares) written by the conv epilogue in training mode (sized per batch)
  float *run_mean = nullptr, *run_var = nullptr;
  bool eval_ready = false;                  // mean / invstd hold the evaluate()-mode values of the current running statistics
  const float* x_in = nullptr;              // input of the last forward
  bool fused_epilogue = false;              // last forward wrote `out` straight from the conv epilogue (y not materialised)
  bool out_skipped = false;                 // last forward left `out` operand-ready only (the next stage's x_p16): no fp32 copy exists
};

struct gr_net {
  gr_ctx* ctx = nullptr;
  std::vector<gr_layer_desc> layers;
  std::vector<Stage> st;
  std::vector<MaskSlot> masks;
  std::vector<int> bn_stage;
  int inC = 0, inH = 0, inW = 0, outC = 0, outH = 0, outW = 0;
  int64_t n_params = 0;
  float *params = nullptr, *grads = nullptr, *adam_m = nullptr, *adam_v = nullptr;
  uint64_t params_version = 1;
  bool training = true;
  uint64_t seed = 1, fwd_counter = 0;   // fwd_counter: Philox counter of the dropout noise (gr_net_set_seed restarts it)
  uint64_t amax_gen = 0;                // generation of the f16x3 scale slots: one per forward, never restarted, so a reseed cannot make a stale slot look fresh
  int capB = 0, lastB = 0;
  float* last_out = nullptr;         // m.output of the last forward: the last stage's buffer, or the caller's destination it was written to directly
  float *in_buf = nullptr, *gout_buf = nullptr, *dy_buf = nullptr, *g_buf[2] = {nullptr, nullptr};
  void* dy_p16 = nullptr;            // operand-ready copy of dy_buf for the data-gradient convolution
  float* dy_buf_b = nullptr; void* dy_p16_b = nullptr;   // second pair: stages alternate, so stage s - 1 can write its dy while stage s's weight gradient (side stream) still reads
  bool wg_pending[2] = {false, false};                   // a side-stream weight gradient may still be reading dy pair k
  float* up_tmp[2] = {nullptr, nullptr}; size_t up_cap = 0;    // backward of a fused up-sampling stage: up-sampled input / data gradient at the up-sampled size
  size_t max_y = 0, max_in = 0;      // per-sample element counts
  uint8_t* mask_stage = nullptr; size_t mask_stage_cap = 0;
  PrepJob* jobs_dev[3] = {nullptr, nullptr, nullptr}; int njobs[3] = {0, 0, 0};   // [0] fp32 k-major images, [1] bf16x6, [2] f16x3 split images
  uint64_t prepped_version[3] = {0, 0, 0};
  unsigned* amax = nullptr;          // f16x3 scale tracking, groups of [nst] slots: x xt y | kb dy dz | w  (AG_*, AMAX_GROUPS)
  bool dy_slots_zeroed = false, w_slots_zeroed = false;   // set by forward_impl's single fill, consumed by backward / weight prep
  bool last_fwd_training = true;     // mode of the last forward (a backward after an evaluate()-mode forward is checked against THAT, not against the current mode)
  bool head_fused = false;           // gr_train_r_step: this forward stops after fc1's GEMM and this backward starts at fc1's GEMMs - the head kernel does what lies between
  int begun_B = 0;                   // > 0: forward_begin has already run for the next forward of this batch size (gr_train_r_step ran it on the side stream)
  int amax_prezeroed_groups = 0;     // > 0: the caller (gr_train_r_step's one fill per step) has just zeroed that many slot groups: the next forward skips its own fill
  bool keep_fp32 = false;            // range-guarded host calls: no lean (operand-ready only) tensors, so a backward can still fall back to bf16x6
  bool last_fwd_fell_back = false;   // the last guarded forward ran on bf16x6: its backward does too
  unsigned guard_sides = 0;           // the largest spreads (bits: activation side | weight side << 16) the last guarded forward measured
};

enum { AG_X = 0, AG_XT = 1, AG_Y = 2, AG_KB = 3, AG_DY = 4, AG_DZ = 5, AG_W = 6, AMAX_GROUPS = 7 };
static int64_t vol3(int c, int h, int w) { return (int64_t)c * h * w; }
static bool is_act(int k) { return k == GR_ELU || k == GR_RELU || k == GR_LEAKYRELU || k == GR_SIGMOID || k == GR_TANH || k == GR_PRELU; }

extern "C" int gr_net_destroy(gr_net* n) {
  if (!n) return GR_ERR_INVALID;
  gr_ctx* c = n->ctx;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (auto& s : n->st) {
    if (s.kind != ST_ELEM) (void)hipFree(s.y);
    if (s.has_post) (void)hipFree(s.out);
    (void)hipFree(s.pool_idx); (void)hipFree(s.wt_fwd); (void)hipFree(s.wt_bwd); (void)hipFree(s.ws_fwd); (void)hipFree(s.ws_up); (void)hipFree(s.ws_bwd);
    (void)hipFree(s.mean); (void)hipFree(s.invstd); (void)hipFree(s.coef); (void)hipFree(s.partials); (void)hipFree(s.partials_b); (void)hipFree(s.stat_part);
    (void)hipFree(s.run_mean); (void)hipFree(s.run_var); (void)hipFree(s.x_p16); (void)hipFree(s.wl1);
  }
  for (auto& m : n->masks) (void)hipFree(m.bits);
  (void)hipFree(n->params); (void)hipFree(n->grads); (void)hipFree(n->adam_m); (void)hipFree(n->adam_v);
  (void)hipFree(n->in_buf); (void)hipFree(n->gout_buf); (void)hipFree(n->dy_buf); (void)hipFree(n->g_buf[0]); (void)hipFree(n->g_buf[1]);
  (void)hipFree(n->dy_p16); (void)hipFree(n->dy_p16_b); (void)hipFree(n->dy_buf_b); (void)hipFree(n->up_tmp[0]); (void)hipFree(n->up_tmp[1]); (void)hipFree(n->mask_stage); (void)hipFree(n->jobs_dev[0]); (void)hipFree(n->jobs_dev[1]); (void)hipFree(n->jobs_dev[2]); (void)hipFree(n->amax);
  delete n;
  return GR_OK;
}

extern "C" int gr_net_create(gr_ctx* c, const gr_layer_desc* L, int nl, int in_c, int in_h, int in_w, gr_net** out) {
  if (!c || !L || nl <= 0 || !out || in_c <= 0 || in_h <= 0 || in_w <= 0) return fail(c, GR_ERR_INVALID, "gr_net_create: bad arguments");
  *out = nullptr;
  HIPCHK(c, hipSetDevice(c->device));
  gr_net* n = new gr_net();
  n->ctx = c; n->layers.assign(L, L + nl); n->inC = in_c; n->inH = in_h; n->inW = in_w;
  // pass 1: per-layer parameter offsets in getParameters() order + shape check
  std::vector<int64_t> woff(nl, -1), boff(nl, -1);
  {
    int cc = in_c, h = in_h, w = in_w; int64_t off = 0;
    for (int i = 0; i < nl; ++i) {
      const gr_layer_desc& d = L[i];
      switch (d.kind) {
        case GR_CONV3: case GR_FULLCONV3:
          if (d.a != cc) { delete n; return fail(c, GR_ERR_INVALID, "layer %d: conv expects %d input planes, got %d", i, d.a, cc); }
          woff[i] = off; off += (int64_t)d.a * d.b * 9; boff[i] = off; off += d.b; cc = d.b; break;
        case GR_CONVK:
          if (d.a != cc) { delete n; return fail(c, GR_ERR_INVALID, "layer %d: conv expects %d input planes, got %d", i, d.a, cc); }
          if (!convk_supported(d.c)) { delete n; return fail(c, GR_ERR_UNSUPPORTED, "layer %d: no kernel for a %dx%d convolution (3x3: GR_CONV3; 5x5: GR_CONVK)", i, d.c, d.c); }
          woff[i] = off; off += (int64_t)d.a * d.b * d.c * d.c; boff[i] = off; off += d.b; cc = d.b; break;
        case GR_PRELU: woff[i] = off; off += 1; break;          // nn.PReLU(): weight = Tensor(1)
        case GR_LINEAR:
          if (d.a != vol3(cc, h, w)) { delete n; return fail(c, GR_ERR_INVALID, "layer %d: linear expects %d inputs, got %lld", i, d.a, (long long)vol3(cc, h, w)); }
          woff[i] = off; off += (int64_t)d.a * d.b; boff[i] = off; off += d.b; cc = d.b; h = 1; w = 1; break;
        case GR_BN:
          if (d.a != cc) { delete n; return fail(c, GR_ERR_INVALID, "layer %d: BN expects %d features, got %d", i, d.a, cc); }
          woff[i] = off; off += cc; boff[i] = off; off += cc; break;
        case GR_MAXPOOL2: h /= 2; w /= 2; break;
        case GR_UPSAMPLE2: h *= 2; w *= 2; break;
        case GR_VIEW: {
          const int vb = d.b > 0 ? d.b : 1, vc = d.c > 0 ? d.c : 1;
          if (vol3(d.a, vb, vc) != vol3(cc, h, w)) { delete n; return fail(c, GR_ERR_INVALID, "layer %d: view size mismatch", i); }
          cc = d.a; h = vb; w = vc; break; }
        case GR_ELU: case GR_RELU: case GR_LEAKYRELU: case GR_SIGMOID: case GR_TANH: case GR_DROPOUT: case GR_SPATIAL_DROPOUT: break;
        default: delete n; return fail(c, GR_ERR_INVALID, "layer %d: unknown kind %d", i, d.kind);
      }
      if (h <= 0 || w <= 0) { delete n; return fail(c, GR_ERR_INVALID, "layer %d: empty spatial extent", i); }
    }
    n->n_params = off; n->outC = cc; n->outH = h; n->outW = w;
  }
  // pass 2: stages
  {
    int cc = in_c, h = in_h, w = in_w, i = 0;
    while (i < nl) {
      Stage s; s.first = i;
      while (i < nl && L[i].kind == GR_VIEW) { cc = L[i].a; h = L[i].b > 0 ? L[i].b : 1; w = L[i].c > 0 ? L[i].c : 1; ++i; }
      if (i >= nl) { if (!n->st.empty()) n->st.back().last = nl - 1; break; }
      s.inC = cc; s.inH = h; s.inW = w;
      if (L[i].kind == GR_UPSAMPLE2) {
        if (i + 1 >= nl || L[i + 1].kind != GR_CONV3) { gr_net_destroy(n); return fail(c, GR_ERR_UNSUPPORTED, "layer %d: UpSamplingNearest(2) is only fused in front of a 3x3 convolution", i); }
        s.up = true; h *= 2; w *= 2; ++i;
      }
      const int k = L[i].kind;
      if (k == GR_CONVK && s.up) { gr_net_destroy(n); return fail(c, GR_ERR_UNSUPPORTED, "layer %d: UpSamplingNearest(2) is only fused in front of a 3x3 convolution", i); }
      if (k == GR_CONV3 || k == GR_FULLCONV3 || k == GR_CONVK) {
        s.kind = ST_CONV; s.fullconv = k == GR_FULLCONV3; s.ksz = k == GR_CONVK ? L[i].c : 3; s.main_layer = i; s.Cin = L[i].a; s.Cout = L[i].b; s.H = h; s.W = w;
        s.w_off = woff[i]; s.b_off = boff[i]; cc = s.Cout; ++i;
      } else if (k == GR_LINEAR) {
        s.kind = ST_LINEAR; s.main_layer = i; s.Cin = L[i].a; s.Cout = L[i].b; s.H = 1; s.W = 1;
        s.w_off = woff[i]; s.b_off = boff[i]; cc = s.Cout; h = 1; w = 1; ++i;
      } else {
        s.kind = ST_ELEM; s.Cin = s.Cout = cc; s.H = h; s.W = w;
      }
      int phase = -1;
      while (i < nl) {
        const gr_layer_desc& d = L[i];
        int ph;
        if (d.kind == GR_BN) ph = 0;
        else if (is_act(d.kind)) ph = 1;
        else if (d.kind == GR_DROPOUT || d.kind == GR_SPATIAL_DROPOUT) ph = s.pool ? 4 : 2;
        else if (d.kind == GR_MAXPOOL2) ph = 3;
        else break;
        if (ph <= phase) break;
        // nn.PReLU's slope gradient needs the activation's own input and gradOutput as tensors: the PReLU closes its stage (what
        // follows it - dropout, pooling - is the next, element-wise stage), and behind a BatchNorm it opens a stage of its own
        if (d.kind == GR_PRELU && s.has_bn) break;
        phase = ph; s.has_post = true;
        if (ph == 0) { s.has_bn = true; s.g_off = woff[i]; s.be_off = boff[i]; s.bn_idx = (int)n->bn_stage.size(); n->bn_stage.push_back((int)n->st.size()); }
        else if (ph == 1) { s.act = d.kind; s.slope = d.p; if (d.kind == GR_PRELU) { s.slope_off = woff[i]; ++i; break; } }
        else if (ph == 3) { s.pool = true; h /= 2; w /= 2; }
        else {
          MaskSlot m; m.layer = i; m.kind = d.kind == GR_DROPOUT ? MASK_ELEM : MASK_SPATIAL; m.p = d.p; m.flags = d.flags;
          m.C = cc; m.H = h; m.W = w;
          if (ph == 2) s.m1 = (int)n->masks.size(); else s.m2 = (int)n->masks.size();
          n->masks.push_back(m);
        }
        ++i;
      }
      s.last = i - 1;
      s.outC = cc; s.outH = h; s.outW = w;
      if (s.kind == ST_ELEM && !s.has_post) { gr_net_destroy(n); return fail(c, GR_ERR_UNSUPPORTED, "layer %d: kind %d cannot start a stage", i, L[i].kind); }
      n->st.push_back(s);
    }
  }
  // allocate parameters, optimiser state, per-stage constant-size buffers
  const size_t pb = sizeof(float) * (size_t)(n->n_params > 0 ? n->n_params : 1);
  if (hipMalloc((void**)&n->params, pb) || hipMalloc((void**)&n->grads, pb) || hipMalloc((void**)&n->adam_m, pb) || hipMalloc((void**)&n->adam_v, pb)) {
    gr_net_destroy(n); return fail(c, GR_ERR_HIP, "parameter allocation failed");
  }
  (void)hipMemsetAsync(n->params, 0, pb, c->stream); (void)hipMemsetAsync(n->grads, 0, pb, c->stream);
  (void)hipMemsetAsync(n->adam_m, 0, pb, c->stream); (void)hipMemsetAsync(n->adam_v, 0, pb, c->stream);
  for (auto& s : n->st) {
    const int C = s.Cout;
    if (s.has_bn) {
      if (hipMalloc((void**)&s.run_mean, sizeof(float) * C) || hipMalloc((void**)&s.run_var, sizeof(float) * C)) { gr_net_destroy(n); return fail(c, GR_ERR_HIP, "alloc failed"); }
      std::vector<float> ones(C, 1.f);
      (void)hipMemsetAsync(s.run_mean, 0, sizeof(float) * C, c->stream);
      (void)hipMemcpy(s.run_var, ones.data(), sizeof(float) * C, hipMemcpyHostToDevice);
    }
    if (hipMalloc((void**)&s.mean, sizeof(float) * C) || hipMalloc((void**)&s.invstd, sizeof(float) * C) ||
        hipMalloc((void**)&s.coef, sizeof(float) * 2 * C) || hipMalloc((void**)&s.partials, sizeof(double) * 2 * STAT_SPLITS * C) ||
        hipMalloc((void**)&s.partials_b, sizeof(double) * PB_SPLITS * C)) {
      gr_net_destroy(n); return fail(c, GR_ERR_HIP, "alloc failed");
    }
    if (s.kind == ST_CONV && s.ksz == 3) {
      // forward reduces over Cin; backward-data reduces over Cout.  (FullConvolution swaps the two roles.)
      const ConvWeightLayout lf = s.fullconv ? conv_weight_layout(s.Cin, s.Cout) : conv_weight_layout(s.Cin, s.Cout);
      const ConvWeightLayout lb = conv_weight_layout(s.Cout, s.Cin);
      if (hipMalloc((void**)&s.wt_fwd, sizeof(float) * lf.elems()) || hipMalloc((void**)&s.wt_bwd, sizeof(float) * lb.elems())) { gr_net_destroy(n); return fail(c, GR_ERR_HIP, "alloc failed"); }
      if (!s.fullconv && (hipMalloc(&s.ws_fwd, conv_weight_split_bytes(s.Cin, s.Cout, false)) || hipMalloc(&s.ws_bwd, conv_weight_split_bytes(s.Cin, s.Cout, true)))) { gr_net_destroy(n); return fail(c, GR_ERR_HIP, "alloc failed"); }
      // SpatialFullConvolution(Cin -> Cout), weight [Cin][Cout][3][3] = the native weight of a convolution Cout -> Cin: its forward is
      // that convolution's backward-data, its backward-data that convolution's forward
      if (s.fullconv && (hipMalloc(&s.ws_fwd, conv_weight_split_bytes(s.Cout, s.Cin, true)) || hipMalloc(&s.ws_bwd, conv_weight_split_bytes(s.Cout, s.Cin, false)))) { gr_net_destroy(n); return fail(c, GR_ERR_HIP, "alloc failed"); }
      if (!s.fullconv && s.up && conv_up2_supported(s.Cin, s.Cout, s.H, s.W) && hipMalloc(&s.ws_up, conv_weight_up2_bytes(s.Cin, s.Cout))) { gr_net_destroy(n); return fail(c, GR_ERR_HIP, "alloc failed"); }
    }
    if (s.kind == ST_CONV && s.ksz == 3 && !s.up && !s.fullconv && hipMalloc((void**)&s.wl1, sizeof(float) * s.Cout)) { gr_net_destroy(n); return fail(c, GR_ERR_HIP, "alloc failed"); }
    if (s.kind == ST_CONV && s.ksz == 5 && !s.up && conv5x5_split_supported(s.Cin, s.Cout, s.H, s.W) &&
        (hipMalloc(&s.ws_fwd, conv_weight_split_bytes(s.Cin, s.Cout, false, 5)) || hipMalloc(&s.ws_bwd, conv_weight_split_bytes(s.Cin, s.Cout, true, 5)))) { gr_net_destroy(n); return fail(c, GR_ERR_HIP, "alloc failed"); }
    const size_t ye = (size_t)vol3(s.Cout, s.H, s.W), ie = (size_t)vol3(s.inC, s.inH, s.inW);
    if (ye > n->max_y) n->max_y = ye;
    if (ie > n->max_in) n->max_in = ie;
  }
  {
    std::vector<PrepJob> jf, js, jh;
    HIPCHK(c, hipMalloc((void**)&n->amax, sizeof(unsigned) * AMAX_WORDS * AMAX_GROUPS * n->st.size()));
    HIPCHK(c, hipMemset(n->amax, 0, sizeof(unsigned) * AMAX_WORDS * AMAX_GROUPS * n->st.size()));
    for (size_t si = 0, ns = n->st.size(); si < ns; ++si) {
      Stage& s = n->st[si];
      auto slot = [&](int grp) { return n->amax + AMAX_WORDS * (grp * ns + si); };
      s.amax_x = slot(AG_X); s.amax_xt = slot(AG_XT); s.amax_y = slot(AG_Y); s.amax_kb = slot(AG_KB); s.amax_dy = slot(AG_DY); s.amax_dz = slot(AG_DZ); s.amax_w = slot(AG_W);
    }
    for (auto& s : n->st) {
      if (s.kind != ST_CONV || s.ksz != 3) continue;
      if (!s.fullconv) {
        jf.push_back(make_prep_job(s.w_off, s.wt_fwd, s.Cin, s.Cout, false, 0));
        jf.push_back(make_prep_job(s.w_off, s.wt_bwd, s.Cin, s.Cout, true, 0));
        js.push_back(make_prep_job(s.w_off, s.ws_fwd, s.Cin, s.Cout, false, 1));
        js.push_back(make_prep_job(s.w_off, s.ws_bwd, s.Cin, s.Cout, true, 1));
        jh.push_back(make_prep_job(s.w_off, s.ws_fwd, s.Cin, s.Cout, false, 2, s.amax_w));
        jh.push_back(make_prep_job(s.w_off, s.ws_bwd, s.Cin, s.Cout, true, 2, s.amax_w));
      } else {
        // SpatialFullConvolution weight is [Cin][Cout][3][3]: its forward is the backward-data of a (Cout -> Cin) conv
        jf.push_back(make_prep_job(s.w_off, s.wt_fwd, s.Cout, s.Cin, true, 0));
        jf.push_back(make_prep_job(s.w_off, s.wt_bwd, s.Cout, s.Cin, false, 0));
        js.push_back(make_prep_job(s.w_off, s.ws_fwd, s.Cout, s.Cin, true, 1));
        js.push_back(make_prep_job(s.w_off, s.ws_bwd, s.Cout, s.Cin, false, 1));
        jh.push_back(make_prep_job(s.w_off, s.ws_fwd, s.Cout, s.Cin, true, 2, s.amax_w));
        jh.push_back(make_prep_job(s.w_off, s.ws_bwd, s.Cout, s.Cin, false, 2, s.amax_w));
      }
    }
    for (auto& s : n->st)          // nn.Linear weights: only their maximum (f16x3 GEMM scales)
      if (s.kind == ST_LINEAR && (int64_t)s.Cin * s.Cout >= (1 << 20)) jh.push_back(make_prep_job(s.w_off, nullptr, s.Cin, s.Cout, false, 5, s.amax_w));
    for (int m = 0; m < 3; ++m) {
      std::vector<PrepJob>& v = m == 0 ? jf : (m == 1 ? js : jh);
      n->njobs[m] = (int)v.size();
      if (!v.empty()) {
        HIPCHK(c, hipMalloc((void**)&n->jobs_dev[m], sizeof(PrepJob) * v.size()));
        HIPCHK(c, hipMemcpy(n->jobs_dev[m], v.data(), sizeof(PrepJob) * v.size(), hipMemcpyHostToDevice));
      }
    }
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *out = n;
  return GR_OK;
}

extern "C" int gr_net_out_dim(gr_net* n, int* c, int* h, int* w) { if (!n) return GR_ERR_INVALID; if (c) *c = n->outC; if (h) *h = n->outH; if (w) *w = n->outW; return GR_OK; }
extern "C" int64_t gr_net_param_count(gr_net* n) { return n ? n->n_params : -1; }
extern "C" float* gr_net_params_dev(gr_net* n) { return n ? n->params : nullptr; }
extern "C" float* gr_net_grads_dev(gr_net* n) { return n ? n->grads : nullptr; }

static int copy_flat(gr_net* n, float* dev, float* host_out, const float* host_in) {
  gr_ctx* c = n->ctx;
  const size_t b = sizeof(float) * (size_t)n->n_params;
  if (host_out) HIPCHK(c, hipMemcpyAsync(host_out, dev, b, hipMemcpyDeviceToHost, c->stream));
  if (host_in) HIPCHK(c, hipMemcpyAsync(dev, host_in, b, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GR_OK;
}
extern "C" int gr_net_get_params(gr_net* n, float* h) { if (!n || !h) return GR_ERR_INVALID; const int r = copy_flat(n, n->params, h, nullptr); return r ? r : head_fault_check(n->ctx); }
extern "C" int gr_net_set_params(gr_net* n, const float* h) { if (!n || !h) return GR_ERR_INVALID; n->params_version++; return copy_flat(n, n->params, nullptr, h); }
extern "C" int gr_net_get_grads(gr_net* n, float* h) { if (!n || !h) return GR_ERR_INVALID; const int r = copy_flat(n, n->grads, h, nullptr); return r ? r : head_fault_check(n->ctx); }
extern "C" int gr_net_set_grads(gr_net* n, const float* h) { if (!n || !h) return GR_ERR_INVALID; return copy_flat(n, n->grads, nullptr, h); }
extern "C" int gr_net_zero_grads(gr_net* n) { if (!n) return GR_ERR_INVALID; HIPCHK(n->ctx, hipMemsetAsync(n->grads, 0, sizeof(float) * (size_t)n->n_params, n->ctx->stream)); return GR_OK; }
extern "C" int gr_adam_reset(gr_net* n) {
  if (!n) return GR_ERR_INVALID;
  HIPCHK(n->ctx, hipMemsetAsync(n->adam_m, 0, sizeof(float) * (size_t)n->n_params, n->ctx->stream));
  HIPCHK(n->ctx, hipMemsetAsync(n->adam_v, 0, sizeof(float) * (size_t)n->n_params, n->ctx->stream));
  return GR_OK;
}
extern "C" int gr_adam_get_state(gr_net* n, float* m, float* v) { if (!n) return GR_ERR_INVALID; int r = m ? copy_flat(n, n->adam_m, m, nullptr) : 0; if (r) return r; return v ? copy_flat(n, n->adam_v, v, nullptr) : GR_OK; }
extern "C" int gr_adam_set_state(gr_net* n, const float* m, const float* v) { if (!n) return GR_ERR_INVALID; int r = m ? copy_flat(n, n->adam_m, nullptr, m) : 0; if (r) return r; return v ? copy_flat(n, n->adam_v, nullptr, v) : GR_OK; }

extern "C" int gr_net_n_bn(gr_net* n) { return n ? (int)n->bn_stage.size() : -1; }
extern "C" int gr_net_bn_features(gr_net* n, int i) { if (!n || i < 0 || i >= (int)n->bn_stage.size()) return -1; return n->st[n->bn_stage[i]].Cout; }
extern "C" int gr_net_get_bn_running(gr_net* n, int i, float* m, float* v) {
  if (!n || i < 0 || i >= (int)n->bn_stage.size()) return GR_ERR_INVALID;
  Stage& s = n->st[n->bn_stage[i]]; gr_ctx* c = n->ctx;
  if (m) HIPCHK(c, hipMemcpyAsync(m, s.run_mean, sizeof(float) * s.Cout, hipMemcpyDeviceToHost, c->stream));
  if (v) HIPCHK(c, hipMemcpyAsync(v, s.run_var, sizeof(float) * s.Cout, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GR_OK;
}
extern "C" int gr_net_set_bn_running(gr_net* n, int i, const float* m, const float* v) {
  if (!n || i < 0 || i >= (int)n->bn_stage.size()) return GR_ERR_INVALID;
  Stage& s = n->st[n->bn_stage[i]]; gr_ctx* c = n->ctx;
  if (m) HIPCHK(c, hipMemcpyAsync(s.run_mean, m, sizeof(float) * s.Cout, hipMemcpyHostToDevice, c->stream));
  if (v) HIPCHK(c, hipMemcpyAsync(s.run_var, v, sizeof(float) * s.Cout, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  s.eval_ready = false;
  return GR_OK;
}
extern "C" int gr_net_set_training(gr_net* n, int t) { if (!n) return GR_ERR_INVALID; n->training = t != 0; return GR_OK; }
extern "C" int gr_net_set_seed(gr_net* n, uint64_t seed) { if (!n) return GR_ERR_INVALID; n->seed = seed; n->fwd_counter = 0; return GR_OK; }

static MaskSlot* find_mask(gr_net* n, int layer) { for (auto& m : n->masks) if (m.layer == layer) return &m; return nullptr; }
static int64_t mask_elems(const MaskSlot& m, int B) { return m.kind == MASK_ELEM ? (int64_t)B * vol3(m.C, m.H, m.W) : (int64_t)B * m.C; }
extern "C" int64_t gr_net_mask_size(gr_net* n, int layer, int B) { if (!n) return -1; MaskSlot* m = find_mask(n, layer); return m ? mask_elems(*m, B) : -1; }

static int ensure_mask_bits(gr_net* n, MaskSlot& m, int64_t elems) {
  const size_t words = (size_t)((elems + 31) / 32) + 4;
  if (words > m.words_cap) {
    gr_ctx* c = n->ctx;
    if (m.bits) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(m.bits)); m.bits = nullptr; }
    HIPCHK(c, hipMalloc((void**)&m.bits, sizeof(uint32_t) * words));
    m.words_cap = words;
  }
  return GR_OK;
}
extern "C" int gr_net_set_mask(gr_net* n, int layer, const uint8_t* keep, int64_t cnt) {
  if (!n || !keep || cnt <= 0) return GR_ERR_INVALID;
  gr_ctx* c = n->ctx; MaskSlot* m = find_mask(n, layer);
  if (!m) return fail(c, GR_ERR_INVALID, "layer %d is not a Dropout / SpatialDropout", layer);
  if ((size_t)cnt > n->mask_stage_cap) {
    if (n->mask_stage) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(n->mask_stage)); n->mask_stage = nullptr; }
    HIPCHK(c, hipMalloc((void**)&n->mask_stage, (size_t)cnt)); n->mask_stage_cap = (size_t)cnt;
  }
  int r = ensure_mask_bits(n, *m, cnt); if (r) return r;
  HIPCHK(c, hipMemcpyAsync(n->mask_stage, keep, (size_t)cnt, hipMemcpyHostToDevice, c->stream));
  launch_pack_mask(n->mask_stage, m->bits, cnt, c->stream); LAUNCHCHK(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  m->injected = true; m->n_last = cnt;
  return GR_OK;
}
extern "C" int gr_net_get_mask(gr_net* n, int layer, uint8_t* keep, int64_t cnt) {
  if (!n || !keep || cnt <= 0) return GR_ERR_INVALID;
  gr_ctx* c = n->ctx; MaskSlot* m = find_mask(n, layer);
  if (!m || !m->bits || cnt > m->n_last) return fail(c, GR_ERR_STATE, "no noise recorded for layer %d", layer);
  if ((size_t)cnt > n->mask_stage_cap) {
    if (n->mask_stage) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(n->mask_stage)); n->mask_stage = nullptr; }
    HIPCHK(c, hipMalloc((void**)&n->mask_stage, (size_t)cnt)); n->mask_stage_cap = (size_t)cnt;
  }
  launch_unpack_mask(m->bits, n->mask_stage, cnt, c->stream); LAUNCHCHK(c);
  HIPCHK(c, hipMemcpyAsync(keep, n->mask_stage, (size_t)cnt, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GR_OK;
}

static int ensure_batch(gr_net* n, int B) {
  if (B <= n->capB) return GR_OK;
  gr_ctx* c = n->ctx;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (auto& s : n->st) {
    if (s.kind != ST_ELEM) { (void)hipFree(s.y); s.y = nullptr; HIPCHK(c, hipMalloc((void**)&s.y, sizeof(float) * (size_t)B * vol3(s.Cout, s.H, s.W))); }
    if (s.has_post) { (void)hipFree(s.out); s.out = nullptr; HIPCHK(c, hipMalloc((void**)&s.out, sizeof(float) * (size_t)B * vol3(s.outC, s.outH, s.outW))); }
    if (s.kind == ST_CONV && s.has_bn && !s.up) {
      (void)hipFree(s.stat_part); s.stat_part = nullptr;
      HIPCHK(c, hipMalloc((void**)&s.stat_part, sizeof(double) * 2 * (size_t)s.Cout * conv_stat_tiles_max(B, s.H, s.W)));
    }
    if (s.pool) { (void)hipFree(s.pool_idx); s.pool_idx = nullptr; HIPCHK(c, hipMalloc((void**)&s.pool_idx, (size_t)B * vol3(s.outC, s.outH, s.outW))); }
    if (s.kind == ST_CONV && s.ksz == 3 && !s.up && !s.fullconv && s.Cin % 16 == 0) {       // operand-ready input image (same bytes as the fp32 input)
      (void)hipFree(s.x_p16); s.x_p16 = nullptr; s.x_p16_gen = 0;
      HIPCHK(c, hipMalloc(&s.x_p16, sizeof(float) * (size_t)B * vol3(s.inC, s.inH, s.inW)));
    }
  }
  (void)hipFree(n->dy_p16); n->dy_p16 = nullptr; (void)hipFree(n->dy_p16_b); n->dy_p16_b = nullptr; (void)hipFree(n->dy_buf_b); n->dy_buf_b = nullptr;
  HIPCHK(c, hipMalloc(&n->dy_p16, sizeof(float) * (size_t)B * n->max_y));
  HIPCHK(c, hipMalloc(&n->dy_p16_b, sizeof(float) * (size_t)B * n->max_y));
  HIPCHK(c, hipMalloc((void**)&n->dy_buf_b, sizeof(float) * (size_t)B * n->max_y));
  (void)hipFree(n->in_buf); (void)hipFree(n->gout_buf); (void)hipFree(n->dy_buf); (void)hipFree(n->g_buf[0]); (void)hipFree(n->g_buf[1]);
  n->in_buf = n->gout_buf = n->dy_buf = n->g_buf[0] = n->g_buf[1] = nullptr;
  HIPCHK(c, hipMalloc((void**)&n->in_buf, sizeof(float) * (size_t)B * vol3(n->inC, n->inH, n->inW)));
  HIPCHK(c, hipMalloc((void**)&n->gout_buf, sizeof(float) * (size_t)B * vol3(n->outC, n->outH, n->outW)));
  HIPCHK(c, hipMalloc((void**)&n->dy_buf, sizeof(float) * (size_t)B * n->max_y));
  HIPCHK(c, hipMalloc((void**)&n->g_buf[0], sizeof(float) * (size_t)B * n->max_in));
  HIPCHK(c, hipMalloc((void**)&n->g_buf[1], sizeof(float) * (size_t)B * n->max_in));
  n->capB = B;
  return GR_OK;
}

// bf16x6 mode: every plain convolution runs on the split kernel except few-output-channel layers the HBM-bound VALU
// kernel covers (same predicate as launch_conv3x3); only the split images are kept current in that mode.
static bool fewout_applies(const Stage& s) {
  // GR_FEWOUT_MAX=0 sends the few-output layers to the MFMA split kernels too (A/B, round 3: G's last convolution 0.388 -> 0.53 ms at cfg3,
  // 39 -> 73 us at cfg2 - a 32-channel output block for 1-3 real channels)
  static const int maxc = GR_KNOB("GR_FEWOUT_MAX", 4);
  return s.ksz == 3 && s.Cout <= maxc && !s.up && s.W % 4 == 0 && s.W >= 16;
}
// f16x3 GEMM for the large nn.Linear layers (R.fc1: 90-97 % of R's parameters); small ones stay on the fp32 MFMA kernel
static bool use_f16_gemm(gr_net* n, const Stage& s) {
  static const bool on = !GR_KNOB_SET("GR_NO_F16_GEMM");
  return on && n->ctx->conv_mode == 2 && s.kind == ST_LINEAR && (int64_t)s.Cin * s.Cout >= (1 << 20);
}
// the 5x5 layer of the D network (models.lua:297) on the f16x3 split kernel (round 4; bf16x6 / f32 modes keep convk.hip's fp32 VALU kernels)
static bool convk_split(gr_net* n, const Stage& s) {
  static const bool on = !GR_KNOB_SET("GR_NO_CONV5_SPLIT");
  return on && n->ctx->conv_mode == 2 && s.kind == ST_CONV && s.ksz == 5 && !s.up && !s.fullconv && s.ws_fwd && conv5x5_split_supported(s.Cin, s.Cout, s.H, s.W);
}
static bool use_bf16x6(gr_net* n, const Stage& s) { return (n->ctx->conv_mode >= 1 && s.kind == ST_CONV && s.ksz == 3 && !fewout_applies(s)) || convk_split(n, s); }   // either split flavour
// Re-lay every convolution's weights (one launch) when the parameters changed since the last time.  bf16x6 mode needs the
// split images; the fp32 k-major images are still needed there by SpatialFullConvolution stages (no split kernel).
static int prep_weights(gr_net* n) {
  gr_ctx* c = n->ctx;
  const int mode = c->conv_mode;
  bool any_full = false;
  for (auto& s : n->st) any_full |= s.kind == ST_CONV && s.fullconv;
  for (int m = 0; m < 3; ++m) {
    if (!(m == mode || (m == 0 && any_full))) continue;
    if (n->prepped_version[m] == n->params_version) continue;
    // the bf16 and f16 images share their buffers: switching the mode invalidates the other flavour
    if (m == 2) launch_conv_weight_prep_batch(n->jobs_dev[m], n->njobs[m], n->params, c->stream, n->amax + AMAX_WORDS * AG_W * n->st.size(),
                                              n->w_slots_zeroed ? 0 : (int)n->st.size());     // 0: the caller has just zeroed the slots
    else launch_conv_weight_prep_batch(n->jobs_dev[m], n->njobs[m], n->params, c->stream);
    LAUNCHCHK(c);
    n->prepped_version[m] = n->params_version;
    if (m >= 1) n->prepped_version[3 - m] = 0;
  }
  if (mode == 2)            // 5x5 stages: their own 25-tap images (one max pass, two re-layouts)
    for (auto& s : n->st)
      if (convk_split(n, s) && s.ws_version != n->params_version) {
        launch_conv_weight_split(n->params + s.w_off, s.ws_fwd, s.Cin, s.Cout, false, c->stream, 2, s.amax_w, 5, true);
        launch_conv_weight_split(n->params + s.w_off, s.ws_bwd, s.Cin, s.Cout, true, c->stream, 2, s.amax_w, 5, false);
        LAUNCHCHK(c);
        s.ws_version = n->params_version;
      }
  if (mode == 2)            // up-sampling stages: the pre-summed four-phase image (its max|w| slot was just refreshed by the batch)
    for (auto& s : n->st)
      if (s.ws_up && s.ws_up_version != n->params_version) {
        launch_conv_weight_up2_split(n->params + s.w_off, s.ws_up, s.Cin, s.Cout, c->stream, s.amax_w, false);
        LAUNCHCHK(c);
        s.ws_up_version = n->params_version;
      }
  return GR_OK;
}

static MaskRef mask_ref(gr_net* n, int slot, bool& need_bits) {
  need_bits = false;
  MaskRef r{MASK_NONE, nullptr, 1.f};
  if (slot < 0) return r;
  MaskSlot& m = n->masks[slot];
  const bool v2 = (m.flags & GR_DROPOUT_V2) != 0;
  const bool active = m.kind == MASK_ELEM ? (n->training || (m.flags & GR_DROPOUT_ALWAYS_ON)) : n->training;
  if (active) {
    need_bits = true;
    r.kind = m.kind; r.bits = m.bits;
    r.scale = (m.kind == MASK_ELEM && v2) ? 1.f / (1.f - m.p) : 1.f;
  } else if (m.kind == MASK_SPATIAL || !v2) {
    r.kind = MASK_SCALE; r.scale = 1.f - m.p;
  }
  return r;
}

static PostArgs post_args(gr_net* n, Stage& s, int B) {
  PostArgs a{};
  a.y = s.kind == ST_ELEM ? s.x_in : s.y; a.out = s.out;
  a.B = B; a.C = s.Cout; a.H = s.H; a.W = s.W;
  a.has_bn = s.has_bn ? 1 : 0;
  a.mean = s.mean; a.invstd = s.invstd;
  a.gamma = s.has_bn ? n->params + s.g_off : nullptr; a.beta = s.has_bn ? n->params + s.be_off : nullptr;
  a.act = s.act; a.slope = s.slope; a.slope_dev = s.act == ACT_PRELU ? n->params + s.slope_off : nullptr;
  bool nb;
  a.m1 = mask_ref(n, s.m1, nb); a.m2 = mask_ref(n, s.m2, nb);
  a.pool = s.pool ? 1 : 0; a.pool_idx = s.pool_idx;
  a.amax_out = nullptr;
  return a;
}

// ------------------------------------------------------------------ f16x3 range guard
// f16x3 scales each tensor by ONE power of two: an entry 2^k below the tensor's maximum keeps about 40 - k bits (fp16's exponent
// range ends 2^-40 below the scaled maximum).  bf16x6 has fp32's exponent range and no such limit.  What a lost bit costs depends
// on what the kernel sums over.  The forward (x, weights) and the data gradient (dy, weights) sum over a CHANNEL index: an output
// channel whose weights sit on the small channels of x can be as small as the product of both spreads while the rounding of the
// large channels' (tiny) weights is not - worst case, the relative error grows with the PRODUCT of the activation-side and the
// weight-side spread.  The weight gradient (x, dy) sums over pixels, one channel pair per sum: each term carries the relative
// error of its own two channels, the LARGER spread counts.  The guard measures, before a pass computes anything, the per-channel
// spread (log2 of largest / smallest non-zero channel maximum) of what enters it, kept per side: ACTIVATION side - the net input
// (forward) or gradOutput (backward) and the BatchNorm (gamma, beta) pairs that set the channel ranges of every tensor behind a
// BatchNorm; WEIGHT side - every weight tensor an f16x3 kernel will read, per input channel and per output channel.  When the
// largest activation-side spread plus the largest weight-side spread exceed GUARD_BUDGET_BITS the whole pass runs on bf16x6
// (counted: gr_kernel_times "range_guard_fallback", gr_range_guard_stats).  Budget: 40 bits of range - 20 bits of spread leave
// 20 bits per entry of the smallest channel, ~1e-6 of that channel's maximum.
// (Until round 3 the two largest spreads of EITHER side were added.  Two BatchNorm layers never multiply each other, and Torch's
// default gamma ~ U(0, 1) spreads 8-14 bits over 64-512 channels: two of five default-initialised R nets tripped that rule at
// their first step and trained on bf16x6 for nothing - tools/debug/debug_guard_trip.py.)
enum { GUARD_BUDGET_BITS = 20 };
static bool guard_over_budget(unsigned word) { return (word & 0xffffu) + (word >> 16) > (unsigned)GUARD_BUDGET_BITS; }
static unsigned guard_merge(unsigned t, unsigned u) {        // per side, the larger of both words' entries
  const unsigned a = (t & 0xffffu) > (u & 0xffffu) ? (t & 0xffffu) : (u & 0xffffu), b = (t >> 16) > (u >> 16) ? (t >> 16) : (u >> 16);
  return a | b << 16;
}
static unsigned* guard_alarm_dev(gr_ctx* c) { return reinterpret_cast<unsigned*>(reinterpret_cast<char*>(c->d_loss) + 16); }
static volatile unsigned* guard_alarm_host(gr_ctx* c) { return reinterpret_cast<volatile unsigned*>(reinterpret_cast<char*>(c->h_loss) + 16); }
static bool f16_consumer(gr_net* n, const Stage& s) { return use_bf16x6(n, s) || use_f16_gemm(n, s); }     // (context in f16x3 mode)
static int guard_scan(gr_ctx* c, const float* t, int B, int C, long HW, long sB, long sC, int side) {
  if (C < 2) return GR_OK;
  if ((size_t)C > c->guard_chmax_cap) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->guard_chmax) (void)hipFree(c->guard_chmax);
    c->guard_chmax = nullptr; c->guard_chmax_cap = 0;
    HIPCHK(c, hipMalloc((void**)&c->guard_chmax, sizeof(unsigned) * (size_t)C));
    HIPCHK(c, hipMemsetAsync(c->guard_chmax, 0, sizeof(unsigned) * (size_t)C, c->stream));
    c->guard_chmax_cap = (size_t)C;
  }
  launch_channel_absmax(t, B, C, HW, sB, sC, c->guard_chmax, c->stream);
  launch_spread_verdict(c->guard_chmax, C, guard_alarm_dev(c), side, c->stream);
  c->guard_scans++;
  LAUNCHCHK(c);
  return GR_OK;
}
// weights and BatchNorm scales of every stage an f16x3 kernel serves
static int guard_scan_params(gr_net* n) {
  gr_ctx* c = n->ctx;
  for (auto& s : n->st) {
    if (!f16_consumer(n, s)) continue;
    const float* w = n->params + s.w_off;
    int r = 0;
    if (s.kind == ST_LINEAR) {             // W[out][in]
      r = guard_scan(c, w, s.Cout, s.Cin, 1, s.Cin, 1, 1); if (r) return r;
      r = guard_scan(c, w, 1, s.Cout, s.Cin, 0, s.Cin, 1); if (r) return r;
    } else if (s.fullconv) {               // W[in][out][3][3]
      r = guard_scan(c, w, 1, s.Cin, (long)s.Cout * 9, 0, (long)s.Cout * 9, 1); if (r) return r;
      r = guard_scan(c, w, s.Cin, s.Cout, 9, (long)s.Cout * 9, 9, 1); if (r) return r;
    } else {                               // W[out][in][K][K]
      const long kk = (long)s.ksz * s.ksz;
      r = guard_scan(c, w, s.Cout, s.Cin, kk, (long)s.Cin * kk, kk, 1); if (r) return r;
      r = guard_scan(c, w, 1, s.Cout, (long)s.Cin * kk, 0, (long)s.Cin * kk, 1); if (r) return r;
    }
  }
  for (size_t si = 0; si < n->st.size(); ++si) {
    Stage& s = n->st[si];
    const bool feeds = si + 1 < n->st.size() && f16_consumer(n, n->st[si + 1]);
    if (!s.has_bn || !(feeds || f16_consumer(n, s))) continue;
    launch_pair_spread(n->params + s.g_off, n->params + s.be_off, s.Cout, guard_alarm_dev(c), c->stream);
    c->guard_scans++;
  }
  LAUNCHCHK(c);
  return GR_OK;
}
// read the spread word (one stream synchronisation) and clear it
static int guard_verdict(gr_ctx* c, unsigned* alarm) {
  HIPCHK(c, hipMemcpyAsync((void*)guard_alarm_host(c), guard_alarm_dev(c), sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemsetAsync(guard_alarm_dev(c), 0, sizeof(unsigned), c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *alarm = *guard_alarm_host(c);
  return GR_OK;
}
static bool guard_applies(gr_net* n) {
  gr_ctx* c = n->ctx;
  if (c->conv_mode != 2 || !c->range_guard) return false;
  for (auto& s : n->st) if (f16_consumer(n, s)) return true;
  return false;
}
// Synchronous parameter scan for host loops built from the *_dev calls (ganrev.adversarial.DeviceGame: gr_net_forward_dev /
// gr_net_backward_dev are unguarded - their activations never pass through host memory): scans this net's weights and BatchNorm
// scales, waits for the verdict, and on a hostile spread keeps the CONTEXT on bf16x6 exactly as gr_train_r_step's sampled
// guard does.  *tripped (nullable) reports the state of the context's guard.
extern "C" int gr_range_guard_scan_params(gr_net* n, int* tripped) {
  if (!n) return GR_ERR_INVALID;
  gr_ctx* c = n->ctx;
  // a sampled scan of gr_train_r_step may still be in flight on this context: its verdict lands in the same host word the
  // synchronous scan below overwrites, so it is consumed first (otherwise a hostile range of the OTHER nets would go unnoticed)
  if (c->guard_pending) {
    HIPCHK(c, hipEventSynchronize(c->ev_guard));
    c->guard_pending = false;
    if (guard_over_budget(*guard_alarm_host(c)) && !c->guard_tripped) { c->guard_tripped = true; c->guard_fallbacks++; }
  }
  if (c->guard_tripped && c->conv_mode == 2) c->conv_mode = 1;
  if (guard_applies(n)) {
    int r = guard_scan_params(n); if (r) return r;
    unsigned sides = 0;
    r = guard_verdict(c, &sides); if (r) return r;
    if (guard_over_budget(sides) && !c->guard_tripped) { c->guard_tripped = true; c->guard_fallbacks++; c->conv_mode = 1; }
  }
  if (tripped) *tripped = c->guard_tripped ? 1 : 0;
  return GR_OK;
}
// view of a per-sample [C][H][W] tensor as channels: a flat feature vector (H = W = 1 behind a Linear) has C "channels" of one element
static int guard_scan_activation(gr_ctx* c, const float* t, int B, int C, int H, int W) {
  const long hw = (long)H * W;
  return guard_scan(c, t, B, C, hw, (long)C * hw, hw, 0);
}

// The per-forward preparation that depends on nothing but the parameters and the batch size: counters, the one fill of the f16x3 scale slots, the weight
// images / maxima of the current parameters, the Dropout noise.  Launches on c->stream - gr_train_r_step points that at the side stream for R, so that these
// three to four small launches (25 us at cfg2: each is a launch-latency floor, not work) run beside G's forward instead of between G and R.
static int forward_begin(gr_net* n, int B) {
  gr_ctx* c = n->ctx;
  const int prezeroed_groups = n->amax_prezeroed_groups;      // consumed on EVERY path out of this call (an early error return must not leave it set for a later forward)
  n->amax_prezeroed_groups = 0;
  HIPCHK(c, hipSetDevice(c->device));
  int r = ensure_batch(n, B); if (r) return r;
  n->fwd_counter++;
  n->amax_gen++;
  const bool f16 = c->conv_mode == 2;
  const size_t nst = n->st.size();
  // f16x3 scale slots [x | dy | w] x stages: ONE fill per forward (each hipMemsetAsync is a 5 us kernel of its own) - the x slots
  // (producers fold maxima in), in training mode the dy slots of the backward that follows, and the w slots when the weight
  // prep below is about to recompute the weight maxima
  n->w_slots_zeroed = false;
  if (f16) {
    const bool w_too = n->training && n->prepped_version[2] != n->params_version;
    const size_t groups = w_too ? AMAX_GROUPS : (n->training ? AG_W : AG_KB);      // x xt y | kb dy dz | w
    if ((size_t)prezeroed_groups < groups) HIPCHK(c, hipMemsetAsync(n->amax, 0, sizeof(unsigned) * AMAX_WORDS * nst * groups, c->stream));
    n->dy_slots_zeroed = groups >= (size_t)AG_W;
    n->w_slots_zeroed = groups == (size_t)AMAX_GROUPS;
  }
  r = prep_weights(n); if (r) return r;
  {
    // Dropout noise of every stage, drawn in one launch (injected masks - tests - are consumed instead)
    MaskJobs jobs{}; jobs.n = 0;
    for (auto& s : n->st)
      for (int slot : {s.m1, s.m2}) {
        if (slot < 0 || !s.has_post) continue;
        bool need; (void)mask_ref(n, slot, need);
        if (!need) continue;
        MaskSlot& m = n->masks[slot];
        const int64_t elems = mask_elems(m, B);
        if (m.injected) {
          if (m.n_last != elems) return fail(c, GR_ERR_INVALID, "injected noise for layer %d has %lld elements, forward needs %lld", m.layer, (long long)m.n_last, (long long)elems);
          m.injected = false;
        } else {
          r = ensure_mask_bits(n, m, elems); if (r) return r;
          if (jobs.n == 24) { launch_gen_mask_batch(jobs, n->seed, n->fwd_counter, c->stream); jobs.n = 0; }
          jobs.job[jobs.n++] = make_mask_job(m.bits, elems, m.p, (uint32_t)m.layer);
          m.n_last = elems;
        }
      }
    launch_gen_mask_batch(jobs, n->seed, n->fwd_counter, c->stream);
    LAUNCHCHK(c);
  }
  n->begun_B = B;
  return GR_OK;
}

static int forward_stages(gr_net* n, const float* in_dev, int B) {
  gr_ctx* c = n->ctx;
  int r = GR_OK;
  if (n->begun_B != B) { r = forward_begin(n, B); if (r) { n->begun_B = 0; return r; } }
  n->begun_B = 0;
  n->last_fwd_training = n->training;
  const float* x = in_dev;
  const bool f16 = c->conv_mode == 2;
  const size_t nst = n->st.size();
  for (size_t si = 0; si < nst; ++si) {
    Stage& s = n->st[si];
    // f16x3: the kernel that writes this stage's output also tracks its max|.| for the convolution that consumes it
    Stage* nx = (f16 && si + 1 < nst && (use_bf16x6(n, n->st[si + 1]) || use_f16_gemm(n, n->st[si + 1]))) ? &n->st[si + 1] : nullptr;
    unsigned* amax_next = nx ? nx->amax_x : nullptr;
    s.x_in = x;
    s.fused_epilogue = false; s.out_skipped = false;
    bool post_p16 = false;
    if (s.kind == ST_CONV && s.ksz != 3) {
      // K x K convolution (the D network's 5x5 layer): fp32 direct kernel, raw output always written, statistics by the pipeline
      if (convk_split(n, s)) {
        if (s.amax_x_fwd != n->amax_gen) { launch_absmax(x, (long)B * vol3(s.inC, s.inH, s.inW), s.amax_x, c->stream, true); s.amax_x_fwd = n->amax_gen; }
        launch_conv5x5_split(x, s.ws_fwd, n->params + s.b_off, s.y, B, s.Cin, s.Cout, s.H, s.W, c->stream, s.amax_x, s.amax_w);
      } else {
      r = ensure_ws(c, convk_workspace_bytes(B, s.Cin, s.Cout, s.ksz)); if (r) return r;
      launch_convk_forward(x, n->params + s.w_off, n->params + s.b_off, s.y, c->ws, B, s.Cin, s.Cout, s.H, s.W, s.ksz, c->stream);
      }
      s.stat_tiles_last = 0;
    } else if (s.kind == ST_CONV) {
      // evaluate() mode: BatchNorm is a per-channel affine map of running statistics, so BN + activation ride in the conv
      // epilogue and the raw conv output is never written (G on this path).  Needs: no pool, no active dropout noise.
      ConvEpilogue ep; const ConvEpilogue* epp = nullptr; float* dst = s.y;
      bool nb1 = false, nb2 = false;
      const MaskRef r1 = mask_ref(n, s.m1, nb1), r2 = mask_ref(n, s.m2, nb2);
      if (!n->training && s.has_post && !s.pool && r1.kind == MASK_NONE && r2.kind == MASK_NONE && s.act != ACT_PRELU) {
        if (s.has_bn) {
          if (!s.eval_ready) { launch_bn_eval_prepare(s.run_mean, s.run_var, s.mean, s.invstd, s.Cout, c->stream); s.eval_ready = true; }
          ep.mean = s.mean; ep.invstd = s.invstd; ep.gamma = n->params + s.g_off; ep.beta = n->params + s.be_off;
        }
        ep.act = s.act; ep.slope = s.slope; epp = &ep; dst = s.out; s.fused_epilogue = true;
      }
      // training-mode BatchNorm: the conv epilogue also leaves the per-channel (sum, sum of squares) of what it stores
      static const bool epi_stats_on = !GR_KNOB_SET("GR_NO_EPI_STATS");
      const bool want_stats = epi_stats_on && n->training && s.has_bn && s.stat_part && !s.fused_epilogue;
      int stat_tiles = 0;
      static const bool fewin_on = !GR_KNOB_SET("GR_NO_FEWIN");
      const bool is_fewin = fewin_on && !s.fullconv && conv_fewin_applies(s.Cin, s.W, s.up);
      const bool in_p16 = f16 && !s.up && s.x_p16 && s.x_p16_gen == n->amax_gen && use_bf16x6(n, s);      // this stage's input arrived operand-ready
      // evaluate() mode, f16x3 (round 4: apply_r.lua:145-153's corpus pipeline): the next convolution's input leaves THIS stage operand-ready
      // too - straight from the conv epilogue (`po`: BatchNorm + activation fused, scale = the weight-norm bound of launch_eval_bound) or, for
      // a stage with a pipeline kernel (pooling), from that kernel (`post_p16`: scale bounded from max|y|, which the conv epilogue measures).
      // Pure functions of the stage's input and parameters: the host-tensor mirror (gr_net_forward_host) computes the same bits.
      const bool nx_p16 = g_eval_p16 && f16 && !n->training && nx && nx->kind == ST_CONV && nx->ksz == 3 && !nx->up && !nx->fullconv && nx->x_p16 &&
                          use_bf16x6(n, *nx) && conv_p16_supported(B, nx->Cin, nx->Cout, nx->H, nx->W) && !s.up && !s.fullconv && s.wl1;
      const bool po = nx_p16 && s.fused_epilogue &&
                      (is_fewin ? conv_fewin_p16_out_supported(s.Cout, s.H, s.W) : (in_p16 && conv_p16_out_supported(s.Cout)));
      post_p16 = nx_p16 && !s.fused_epilogue && s.has_post && s.act != ACT_PRELU && post_g8_supported(s.Cout, s.H, s.W, s.pool) && (is_fewin || use_bf16x6(n, s));
      // f16x3 training: max|y| of the raw output rides along (slot amax_y): with the batch statistics it bounds max|pipeline
      // output| and max|dy| BEFORE the kernels that write those tensors run, so they can write them operand-ready (P16)
      const bool track_y = (f16 && want_stats) || post_p16;
      const bool last_writer = s.fused_epilogue || !s.has_post;
      unsigned* conv_amax_out = last_writer ? amax_next : (track_y ? s.amax_y : nullptr);
      P16Out p16o;
      if (po) {
        if (s.wl1_version != n->params_version) { launch_conv_weight_l1(n->params + s.w_off, s.Cout, s.Cin * 9, s.wl1, c->stream); s.wl1_version = n->params_version; }
        const unsigned* mslot = s.amax_x;                       // the true max|x| of this stage's input ...
        if (in_p16 && !is_fewin) mslot = s.x_true ? s.x_true : s.amax_x;                                  // ... tracked beside the bound its image is scaled by
        else if (s.amax_x_fwd != n->amax_gen) { launch_absmax(x, (long)B * vol3(s.inC, s.inH, s.inW), s.amax_x, c->stream, true); s.amax_x_fwd = n->amax_gen; }
        launch_eval_bound(s.wl1, n->params + s.b_off, &ep, s.Cout, 1.f, mslot, nx->amax_x, c->stream);
        p16o.p16 = nx->x_p16; p16o.scale = nx->amax_x;
        conv_amax_out = nx->amax_xt; dst = nullptr;
      }
      if (is_fewin) {
        launch_conv3x3_fewin(x, n->params + s.w_off, n->params + s.b_off, dst, B, s.Cin, s.Cout, s.H, s.W, c->stream, epp, conv_amax_out,
                             want_stats ? s.stat_part : nullptr, want_stats ? &stat_tiles : nullptr, po ? &p16o : nullptr);
        if (last_writer && nx) nx->amax_x_fwd = n->amax_gen;
      } else if (use_bf16x6(n, s)) {
        const int nterm = c->conv_mode == 2 ? 2 : 3;
        static const bool up2_on = !GR_KNOB_SET("GR_NO_UP2");
        if (nterm == 2 && in_p16) {
          // the previous stage's pipeline kernel left this stage's input operand-ready, scaled by the bound in amax_x
          launch_conv3x3_p16(s.x_p16, s.ws_fwd, n->params + s.b_off, dst, B, s.Cin, s.Cout, s.H, s.W, c->stream, epp, s.amax_x, s.amax_w, conv_amax_out,
                             want_stats ? s.stat_part : nullptr, want_stats ? &stat_tiles : nullptr, po ? &p16o : nullptr);
        } else {
          // input not produced by a tracking kernel (the net's own input, a GEMM, a VALU conv): take its maximum now
          if (nterm == 2 && s.amax_x_fwd != n->amax_gen) { launch_absmax(x, (long)B * vol3(s.inC, s.inH, s.inW), s.amax_x, c->stream, true); s.amax_x_fwd = n->amax_gen; }
          if (nterm == 2 && s.up && s.ws_up && up2_on)
            launch_conv3x3_up2_f16x3(x, s.ws_up, n->params + s.b_off, dst, B, s.Cin, s.Cout, s.H, s.W, c->stream, epp, s.amax_x, s.amax_w,
                                     last_writer ? amax_next : nullptr);
          else
            launch_conv3x3_split(x, s.ws_fwd, n->params + s.b_off, dst, B, s.Cin, s.Cout, s.H, s.W, s.up, c->stream, epp, nterm, s.amax_x, s.amax_w,
                                 (nterm == 2 && !s.up) ? conv_amax_out : (last_writer ? amax_next : nullptr),
                                 want_stats ? s.stat_part : nullptr, want_stats ? &stat_tiles : nullptr);
        }
        if (last_writer && nx) nx->amax_x_fwd = n->amax_gen;
      }
      else launch_conv3x3(x, s.wt_fwd, n->params + s.b_off, dst, B, s.Cin, s.Cout, s.H, s.W, s.up, c->stream, s.fullconv ? nullptr : n->params + s.w_off, epp);
      if (po) { nx->x_p16_gen = n->amax_gen; nx->x_true = nx->amax_xt; s.out_skipped = true; }      // (no fp32 copy: gr_net_layer_output says so)
      if (s.fused_epilogue) { LAUNCHCHK(c); x = s.out; continue; }
      s.stat_tiles_last = stat_tiles;
    } else if (s.kind == ST_LINEAR) {
      const size_t wsb = gemm_workspace_bytes(B, s.Cout, s.Cin);
      r = ensure_ws(c, wsb); if (r) return r;
      // evaluate() mode: per-feature BatchNorm + activation ride in the GEMM epilogue (G's first stage, models.lua:115-117:
      // the raw Linear output - 268 MB at cfg3 - is never written)
      bool nb1 = false, nb2 = false;
      const MaskRef r1 = mask_ref(n, s.m1, nb1), r2 = mask_ref(n, s.m2, nb2);
      if (!n->training && s.has_post && !s.pool && r1.kind == MASK_NONE && r2.kind == MASK_NONE && s.H == 1 && s.W == 1 && s.act != ACT_PRELU &&
          gemm_epilogue_possible(B, s.Cout, s.Cin)) {
        ConvEpilogue ep;
        if (s.has_bn) {
          if (!s.eval_ready) { launch_bn_eval_prepare(s.run_mean, s.run_var, s.mean, s.invstd, s.Cout, c->stream); s.eval_ready = true; }
          ep.mean = s.mean; ep.invstd = s.invstd; ep.gamma = n->params + s.g_off; ep.beta = n->params + s.be_off;
        }
        ep.act = s.act; ep.slope = s.slope;
        const bool f16g = use_f16_gemm(n, s);
        if (f16g && s.amax_x_fwd != n->amax_gen) { launch_absmax(x, (long)B * s.Cin, s.amax_x, c->stream, true); s.amax_x_fwd = n->amax_gen; }
        launch_gemm(x, s.Cin, 1, n->params + s.w_off, s.Cin, 1, s.out, s.Cout, n->params + s.b_off, false, B, s.Cout, s.Cin, c->ws, c->stream, &ep, amax_next,
                    f16g ? s.amax_x : nullptr, f16g ? s.amax_w : nullptr);
        if (nx) nx->amax_x_fwd = n->amax_gen;
        s.fused_epilogue = true;
        LAUNCHCHK(c);
        x = s.out; continue;
      }
      if (use_f16_gemm(n, s)) {
        if (s.amax_x_fwd != n->amax_gen) { launch_absmax(x, (long)B * s.Cin, s.amax_x, c->stream, true); s.amax_x_fwd = n->amax_gen; }
        launch_gemm(x, s.Cin, 1, n->params + s.w_off, s.Cin, 1, s.y, s.Cout, n->params + s.b_off, false, B, s.Cout, s.Cin, c->ws, c->stream,
                    nullptr, nullptr, s.amax_x, s.amax_w);
      } else
      launch_gemm(x, s.Cin, 1, n->params + s.w_off, s.Cin, 1, s.y, s.Cout, n->params + s.b_off, false, B, s.Cout, s.Cin, c->ws, c->stream);
    }
    LAUNCHCHK(c);
    if (n->head_fused && si + 2 == nst) {      // fc1's raw output is complete: the head kernel (gr_train_r_step) takes it from here, through fc2, the criterion and back
      s.eval_ready = false;
      Stage& s2 = n->st[si + 1];
      s2.x_in = s.out; s2.fused_epilogue = false; s2.out_skipped = false; s2.out = s2.has_post ? s2.out : s2.y;
      break;
    }
    if (!s.has_post) { s.out = s.y; x = s.out; continue; }
    const float* yv = s.kind == ST_ELEM ? x : s.y;
    PostArgs pa = post_args(n, s, B);
    bool p16_out = false;
    if (s.has_bn) {
      if (n->training) s.eval_ready = false;          // mean / invstd become batch statistics, the running statistics move
      if (n->training && s.kind == ST_CONV && s.stat_tiles_last > 0) {
        // f16x3: the statistics kernel also folds the a-priori bounds (BnBounds) - into the NEXT convolution's scale slot when
        // this stage's pipeline kernel can write that convolution's input operand-ready, and into this stage's backward factor
        BnBounds bd{}; const BnBounds* bdp = nullptr;
        if (f16 && !s.up && !s.fullconv) {
          p16_out = nx && nx->kind == ST_CONV && nx->x_p16 && !nx->up && !nx->fullconv && use_bf16x6(n, *nx) &&
                    post_g8_supported(s.Cout, s.H, s.W, s.pool) && conv_p16_supported(B, nx->Cin, nx->Cout, nx->H, nx->W);
          bd.amax_y = s.amax_y; bd.gamma = pa.gamma; bd.beta = pa.beta; bd.act = s.act;
          bd.mask_scale = fmaxf(1.f, pa.m1.scale) * fmaxf(1.f, pa.m2.scale);
          bd.bound_out = p16_out ? nx->amax_x : nullptr; bd.kb_out = s.amax_kb;
          bdp = &bd; s.kb_gen = n->amax_gen;
        }
        StatSync ss; const StatSync* sync = stat_sync(c, ss, s.Cout, (double)B * s.H * s.W);
        if (c->coll_rc) { r = c->coll_rc; c->coll_rc = 0; return r; }
        if (sync) {     // synchronised BatchNorm: the ranks' per-channel (sum, sum of squares) are added before the statistics are formed
          launch_pair_sums(s.stat_part, s.stat_tiles_last, s.stat_tiles_last, s.Cout, sync->buf, c->stream);
          r = small_allreduce(c, sync->buf, 2L * s.Cout, 1); if (r) return r;
          launch_bn_stats_from_tiles(sync->buf, 1, s.Cout, sync->n_global, s.mean, s.invstd, s.run_mean, s.run_var, c->stream, bdp);
        } else
        launch_bn_stats_from_tiles(s.stat_part, s.stat_tiles_last, s.Cout, (double)B * s.H * s.W, s.mean, s.invstd, s.run_mean, s.run_var, c->stream, bdp);
      }
      else if (n->training) {
        StatSync ss;
        launch_bn_stats(yv, B, s.Cout, s.H * s.W, s.partials, s.mean, s.invstd, s.run_mean, s.run_var, 1, c->stream, stat_sync(c, ss, s.Cout, (double)B * s.H * s.W));
        if (c->coll_rc) { r = c->coll_rc; c->coll_rc = 0; return r; }
      }
      else if (!s.eval_ready) { launch_bn_eval_prepare(s.run_mean, s.run_var, s.mean, s.invstd, s.Cout, c->stream); s.eval_ready = true; }
    }
    if (post_p16) {
      // evaluate() mode: bound of max|pipeline output| from max|y| (the conv epilogue's) and the running statistics, as the statistics
      // kernel folds it in training mode
      ConvEpilogue e2;
      if (s.has_bn) { e2.mean = s.mean; e2.invstd = s.invstd; e2.gamma = pa.gamma; e2.beta = pa.beta; }
      e2.act = s.act; e2.slope = s.slope;
      launch_eval_bound(nullptr, nullptr, &e2, s.Cout, fmaxf(1.f, pa.m1.scale) * fmaxf(1.f, pa.m2.scale), s.amax_y, nx->amax_x, c->stream);
      p16_out = true; nx->x_true = nx->amax_x;
    }
    pa.amax_out = p16_out ? nullptr : amax_next;      // operand-ready: the slot already holds the bound and must not move
    pa.p16 = p16_out ? nx->x_p16 : nullptr; pa.p16_scale = p16_out ? nx->amax_x : nullptr;
    // The fp32 copy of the stage output has one more reader than the next convolution's forward: that convolution's weight
    // gradient.  When it will take the operand-ready image too (every condition is fixed by the shapes and this forward), the
    // fp32 tensor is not written at all: the pipeline kernel writes 4 bytes per element, as it did before it wrote two formats.
    static const bool lean_on = !GR_KNOB_SET("GR_P16_KEEP_FP32");
    s.out_skipped = lean_on && !n->keep_fp32 && p16_out && nx->has_bn && n->dy_p16 && post_g8_supported(nx->Cout, nx->H, nx->W, nx->pool, true) &&
                    conv_wgrad_p16_supported(B, nx->Cin, nx->Cout, nx->H, nx->W) && nx->stat_part;
    if (post_p16) s.out_skipped = true;               // evaluate(): nothing else reads the fp32 tensor
    if (s.out_skipped) pa.out = nullptr;
    launch_post_forward(pa, c->stream);
    if (p16_out) nx->x_p16_gen = n->amax_gen;
    if (nx) nx->amax_x_fwd = n->amax_gen;
    LAUNCHCHK(c);
    x = s.out;
  }
  n->lastB = B;
  return GR_OK;
}

// One forward.  final_dst (nullable): where the caller wants the result.  In evaluate() mode the LAST stage writes it there itself (its
// kernel's destination pointer is swapped for the call: no staging copy - utils/nn_utils.lua:25-28's row loop becomes nothing at all);
// in training mode the stage buffers are state the backward reads, so the result is copied.  m.output (gr_net_output_dev) is
// wherever the last forward left its result.
static int forward_impl(gr_net* n, const float* in_dev, int B, float* final_dst = nullptr) {
  gr_ctx* c = n->ctx;
  HIPCHK(c, hipSetDevice(c->device));
  int r = ensure_batch(n, B); if (r) return r;          // (buffers may move: before the swap)
  Stage& ls = n->st.back();
  const bool redirect = final_dst && !n->training && ((uintptr_t)final_dst & 15) == 0;
  float** slot = ls.has_post ? &ls.out : &ls.y;
  float* const saved = *slot;
  if (redirect) *slot = final_dst;
  r = forward_stages(n, in_dev, B);
  if (redirect) { *slot = saved; if (!ls.has_post) ls.out = saved; }
  if (r) return r;
  n->last_out = ls.out;
  if (redirect) n->last_out = final_dst;
  else if (final_dst) {
    HIPCHK(c, hipMemcpyAsync(final_dst, ls.out, sizeof(float) * (size_t)B * vol3(n->outC, n->outH, n->outW), hipMemcpyDeviceToDevice, c->stream));
  }
  return GR_OK;
}

extern "C" float* gr_net_output_dev(gr_net* n) { return (n && !n->st.empty()) ? (n->last_out ? n->last_out : n->st.back().out) : nullptr; }

extern "C" int gr_net_forward_dev(gr_net* n, const float* in_dev, int B, float* out_dev) {
  if (!n || !in_dev || B <= 0) return GR_ERR_INVALID;
  n->head_fused = false;
  n->keep_fp32 = false; n->last_fwd_fell_back = false;      // device-resident callers: no stream synchronisation, so no range guard here (gr_train_r_step samples one)
  return forward_impl(n, in_dev, B, out_dev);
}

// NN_UTILS.forwardBatched(model, input, batchSize) (utils/nn_utils.lua:5-33) on device-resident rows: chunk c of `batch` rows goes
// through the net and lands at out_dev + c * batch * outdim - the chunk's last kernel writes there itself (forward_impl).
extern "C" int gr_net_forward_batched_dev(gr_net* n, const float* in_dev, int64_t rows, int batch, float* out_dev) {
  if (!n || !in_dev || !out_dev || rows <= 0 || batch <= 0) return GR_ERR_INVALID;
  n->keep_fp32 = false; n->last_fwd_fell_back = false;
  const int64_t iv = vol3(n->inC, n->inH, n->inW), ov = vol3(n->outC, n->outH, n->outW);
  for (int64_t off = 0; off < rows; off += batch) {
    const int b = (int)(rows - off < batch ? rows - off : batch);
    int r = forward_impl(n, in_dev + off * iv, b, out_dev + off * ov); if (r) return r;
  }
  return GR_OK;
}

// apply_r.lua:145-153 as ONE device-resident pipeline: per chunk  images = G:forward(noise)  (:146)  ->  attributes_k = R_k:forward(images)
// (:152 MODEL_R, :153 MODEL_R_FIXER), the recovered noise written straight into the [rows x nd] tables the search (apply_r.lua:265-282)
// reads.  The images never leave the GPU; they are kept (images_out_dev) only when the caller wants them (pixel-wise search, fix-faces).
extern "C" int gr_embed_dev(gr_net* g, gr_net* const* rnets, int n_rnets, const float* noise_dev, int64_t rows, int batch,
                            float* images_out_dev, float* const* attr_out_dev) {
  if (!g || !noise_dev || rows <= 0 || batch <= 0 || n_rnets < 0 || (n_rnets > 0 && (!rnets || !attr_out_dev))) return GR_ERR_INVALID;
  gr_ctx* c = g->ctx;
  const int64_t nd = vol3(g->inC, g->inH, g->inW), iv = vol3(g->outC, g->outH, g->outW);
  for (int k = 0; k < n_rnets; ++k) {
    if (!rnets[k] || !attr_out_dev[k]) return GR_ERR_INVALID;
    if (rnets[k]->ctx != c) return fail(c, GR_ERR_INVALID, "gr_embed_dev: nets live on different contexts");
    if (vol3(rnets[k]->inC, rnets[k]->inH, rnets[k]->inW) != iv) return fail(c, GR_ERR_INVALID, "gr_embed_dev: image dim mismatch between G and R net %d", k);
    rnets[k]->keep_fp32 = false; rnets[k]->last_fwd_fell_back = false;
  }
  g->keep_fp32 = false; g->last_fwd_fell_back = false;
  int r = GR_OK;
  for (int64_t off = 0; off < rows && !r; off += batch) {
    const int b = (int)(rows - off < batch ? rows - off : batch);
    g_kphase = 1;
    { PhaseRange pr("G forward"); r = forward_impl(g, noise_dev + off * nd, b, images_out_dev ? images_out_dev + off * iv : nullptr); }
    const float* images = g->last_out;
    g_kphase = 2;
    for (int k = 0; k < n_rnets && !r; ++k) {
      PhaseRange pr("R forward");
      const int64_t ov = vol3(rnets[k]->outC, rnets[k]->outH, rnets[k]->outW);
      r = forward_impl(rnets[k], images, b, attr_out_dev[k] + off * ov);
    }
  }
  g_kphase = 0;
  return r;
}

extern "C" int gr_net_forward_host(gr_net* n, const float* in_host, int B, float* out_host) {
  if (!n || !in_host || B <= 0) return GR_ERR_INVALID;
  gr_ctx* c = n->ctx;
  HIPCHK(c, hipSetDevice(c->device));
  int r = ensure_batch(n, B); if (r) return r;
  HIPCHK(c, hipMemcpyAsync(n->in_buf, in_host, sizeof(float) * (size_t)B * vol3(n->inC, n->inH, n->inW), hipMemcpyHostToDevice, c->stream));
  n->last_fwd_fell_back = false;
  n->keep_fp32 = guard_applies(n);
  if (n->keep_fp32) {
    // the first stage's view of the input: a Linear reads it as a flat feature vector
    const Stage& s0 = n->st.front();
    if (s0.kind == ST_LINEAR) r = guard_scan_activation(c, n->in_buf, B, s0.Cin, 1, 1);
    else r = guard_scan_activation(c, n->in_buf, B, n->inC, n->inH, n->inW);
    if (r) return r;
    r = guard_scan_params(n); if (r) return r;
    unsigned sides = 0;
    r = guard_verdict(c, &sides); if (r) return r;
    n->guard_sides = sides;
    n->last_fwd_fell_back = guard_over_budget(sides);
  }
  if (n->last_fwd_fell_back) { c->guard_fallbacks++; c->conv_mode = 1; }
  r = forward_impl(n, n->in_buf, B);
  if (n->last_fwd_fell_back) c->conv_mode = 2;
  if (r) return r;
  if (out_host) HIPCHK(c, hipMemcpyAsync(out_host, n->st.back().out, sizeof(float) * (size_t)B * vol3(n->outC, n->outH, n->outW), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GR_OK;
}

extern "C" int gr_net_layer_output(gr_net* n, int layer, float* host, int64_t cnt) {
  if (!n || !host || n->lastB <= 0) return GR_ERR_INVALID;
  gr_ctx* c = n->ctx;
  for (auto& s : n->st) {
    const float* p = nullptr; int64_t e = 0;
    if (layer == s.main_layer && !s.fused_epilogue) { p = s.y; e = (int64_t)n->lastB * vol3(s.Cout, s.H, s.W); }
    else if (layer == s.last && s.has_post) {
      if (s.out_skipped) return fail(c, GR_ERR_UNSUPPORTED, "layer %d: the last forward left this output operand-ready (fp16 hi/lo image of the next convolution) only", layer);
      p = s.out; e = (int64_t)n->lastB * vol3(s.outC, s.outH, s.outW);
    }
    if (p && &s == &n->st.back() && n->last_out && p == (s.has_post ? s.out : s.y)) p = n->last_out;   // the last forward wrote its result at the caller's destination
    if (p) {
      if (cnt != e) return fail(c, GR_ERR_INVALID, "layer %d output has %lld elements", layer, (long long)e);
      HIPCHK(c, hipMemcpyAsync(host, p, sizeof(float) * (size_t)e, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      return GR_OK;
    }
  }
  return fail(c, GR_ERR_UNSUPPORTED, "layer %d is fused into its stage; its output is never materialised", layer);
}

// nn.SpatialMaxPooling's `indices` of the last forward (models.lua:422,440), one byte per output element: 0..3 = position in the
// 2x2 window in scan order (dy, dx).  Lets a parity test tell an argmax that differs between two correct fp32 implementations
// (a window whose two largest inputs differ by rounding noise) from a wrong gradient.
extern "C" int gr_net_get_pool_index(gr_net* n, int layer, uint8_t* host, int64_t cnt) {
  if (!n || !host || n->lastB <= 0) return GR_ERR_INVALID;
  gr_ctx* c = n->ctx;
  if (layer < 0 || layer >= (int)n->layers.size() || n->layers[layer].kind != GR_MAXPOOL2) return fail(c, GR_ERR_INVALID, "layer %d is not a SpatialMaxPooling", layer);
  for (auto& s : n->st) {
    if (!s.pool || layer < s.first || layer > s.last) continue;
    const int64_t e = (int64_t)n->lastB * vol3(s.outC, s.outH, s.outW);
    if (cnt != e) return fail(c, GR_ERR_INVALID, "layer %d has %lld pooling windows", layer, (long long)e);
    if (!s.pool_idx) return fail(c, GR_ERR_STATE, "no forward recorded");
    HIPCHK(c, hipMemcpyAsync(host, s.pool_idx, (size_t)e, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GR_OK;
  }
  return fail(c, GR_ERR_STATE, "layer %d: pooling stage not found", layer);
}

// Gradient bucket [lo, hi) of the flat vector is final: reduce it over RCCL on the comm stream while the compute stream
// keeps running the backward of the earlier layers (the flat order is layer order, backward walks it from the end, so a
// finished bucket is always a suffix range; fc1's 90-97 % of the bytes are ready first).
static int reduce_bucket(gr_net* n, int64_t lo, int64_t hi) {
  gr_ctx* c = n->ctx;
  if (hi <= lo) return GR_OK;
  if (c->xchg) return small_allreduce(c, n->grads + lo, (long)(hi - lo), 0);        // host-exchange hook: in stream order on the compute stream
  HIPCHK(c, hipEventRecord(c->ev_ready, c->stream));
  HIPCHK(c, hipStreamWaitEvent(c->comm_stream, c->ev_ready, 0));
  NCCLCHK(c, ncclAllReduce(n->grads + lo, n->grads + lo, (size_t)(hi - lo), ncclFloat, ncclSum, c->comm, c->comm_stream));
  return GR_OK;
}
constexpr int64_t BUCKET_MIN_ELEMS = 1 << 20;

static int backward_impl(gr_net* n, const float* in_dev, const float* gout_dev, int B, float* gin_dev, bool reduce = false) {
  gr_ctx* c = n->ctx;
  HIPCHK(c, hipSetDevice(c->device));
  if (B != n->lastB) return fail(c, GR_ERR_STATE, "backward batch %d does not match the last forward (%d)", B, n->lastB);
  if (!n->last_fwd_training)       // an evaluate()-mode forward may have handed stage outputs over operand-ready only (forward_stages: po / post_p16); m:training() in between does not bring the tensors back
    for (auto& s : n->st)
      if (s.out_skipped) return fail(c, GR_ERR_STATE, "backward after an evaluate()-mode forward that kept stage outputs operand-ready only (gr_set_tuning \"eval_p16\" 0 keeps the fp32 tensors)");
  reduce = reduce && have_peers(c);
  int64_t bucket_hi = n->n_params;            // everything in [stage first offset, bucket_hi) is final but not yet reduced
  BiasJobs bias_jobs{}; bias_jobs.n = 0;
  { int r = prep_weights(n); if (r) return r; }   // no-op unless the arithmetic mode changed since the forward
  const bool f16 = c->conv_mode == 2;
  if (f16 && !n->dy_slots_zeroed)   // the dy slots (already zero when this is the first backward after a training-mode forward)
    HIPCHK(c, hipMemsetAsync(n->amax + AMAX_WORDS * AG_DY * n->st.size(), 0, sizeof(unsigned) * AMAX_WORDS * 2 * n->st.size(), c->stream));   // dy and dz groups
  n->dy_slots_zeroed = false;
  const float* g = gout_dev;
  for (int si = (int)n->st.size() - 1; si >= 0; --si) {
    Stage& s = n->st[si];
    if (s.has_bn && !n->training) return fail(c, GR_ERR_STATE, "backward through BatchNormalization requires training mode");
    // gr_train_r_step's head kernel has already run fc2's whole backward and fc1's pipeline backward (dy of fc1 sits in its dy buffer, max|dy| in its slot,
    // the gradients of fc2, of the BatchNorm and of fc1's bias are accumulated): fc1's two GEMMs are what is left of these two stages
    if (n->head_fused && si + 1 == (int)n->st.size()) continue;
    const bool head_here = n->head_fused && si + 2 == (int)n->st.size();
    const float* x = si == 0 ? in_dev : s.x_in;
    const bool need_gin = si > 0 || gin_dev != nullptr;
    float* gin = (si == 0 && gin_dev) ? gin_dev : n->g_buf[si & 1];
    // pipeline backward: g (wrt stage output) -> dy (wrt raw main-op output / ELEM input)
    // dy pair of this stage; the side-stream weight gradient that last read it (two stages ago) must be done before it is rewritten
    const int dk = si & 1;
    float* const dyb = dk ? n->dy_buf_b : n->dy_buf; void* const dyp = dk ? n->dy_p16_b : n->dy_p16;
    if (n->wg_pending[dk]) { HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_wgrad_done[dk], 0)); n->wg_pending[dk] = false; }
    PostBwdArgs pb{};
    pb.f = post_args(n, s, B);
    if (!s.has_post) { pb.f.out = nullptr; }
    pb.gout = g;
    pb.dy = s.kind == ST_ELEM ? gin : dyb;
    pb.partials = s.partials; pb.partials_b = s.partials_b; pb.coef = s.coef;
    pb.ggamma = s.has_bn ? n->grads + s.g_off : nullptr; pb.gbeta = s.has_bn ? n->grads + s.be_off : nullptr;
    pb.gbias = s.kind == ST_ELEM ? nullptr : n->grads + s.b_off;
    pb.amax_dy = (f16 && (s.kind == ST_CONV || use_f16_gemm(n, s))) ? s.amax_dy : nullptr;
    // operand-ready dy for the data-gradient convolution: needs the forward's bound factor of THIS forward (kb_gen)
    const bool dy_ok = f16 && s.kind == ST_CONV && s.ksz == 3 && !s.up && !s.fullconv && s.has_bn && n->dy_p16 && s.kb_gen == n->amax_gen &&
                       post_g8_supported(s.Cout, s.H, s.W, s.pool, true);
    const bool dgrad_p16 = dy_ok && need_gin && conv_p16_supported(B, s.Cout, s.Cin, s.H, s.W);
    // weight gradient with both operands operand-ready: this stage's input image (written by the previous stage's forward
    // pipeline kernel in THIS forward) and pass B's dy image
    const bool wgrad_p16 = dy_ok && s.x_p16 && s.x_p16_gen == n->amax_gen && conv_wgrad_p16_supported(B, s.Cin, s.Cout, s.H, s.W);
    const bool dy_p16 = dgrad_p16 || wgrad_p16;
    pb.dy_p16 = dy_p16 ? dyp : nullptr; pb.amax_dz = dy_p16 ? s.amax_dz : nullptr; pb.kb = s.amax_kb;
    if (s.kind == ST_CONV && si > 0 && n->st[si - 1].out_skipped && !wgrad_p16)
      return fail(c, GR_ERR_STATE, "stage %d: the forward left this stage's input operand-ready only (f16x3); backward in another arithmetic mode needs a new forward", si);
    static const bool lean_on = !GR_KNOB_SET("GR_P16_KEEP_FP32");
    if (lean_on && !n->keep_fp32 && wgrad_p16 && (dgrad_p16 || !need_gin)) pb.dy = nullptr;      // no fp32 reader of dy is left
    if (s.act == ACT_PRELU && !head_here) {
      // nn.PReLU accGradParameters: the stage ends at the PReLU, so g is its gradOutput and the raw main-op output (the stage
      // input for an element-wise stage) its input
      int r = ensure_ws(c, prelu_grad_workspace_bytes()); if (r) return r;
      launch_prelu_grad(g, s.kind == ST_ELEM ? x : s.y, (long)B * vol3(s.Cout, s.H, s.W), static_cast<double*>(c->ws), n->grads + s.slope_off, c->stream);
    }
    if (!head_here) {
      StatSync ss;
      launch_post_backward(pb, c->stream, &bias_jobs, s.has_bn ? stat_sync(c, ss, s.Cout, (double)B * s.H * s.W) : nullptr);       // bias gradients of several stages are summed by one launch
      if (c->coll_rc) { const int rc = c->coll_rc; c->coll_rc = 0; return rc; }
    }
    LAUNCHCHK(c);
    if (s.kind == ST_CONV && s.ksz != 3) {
      int r = ensure_ws(c, convk_workspace_bytes(B, s.Cin, s.Cout, s.ksz)); if (r) return r;
      launch_convk_backward_weight(x, dyb, n->grads + s.w_off, c->ws, B, s.Cin, s.Cout, s.H, s.W, s.ksz, c->stream);
      if (need_gin) {
        // the data gradient = the same convolution on the transposed + flipped weights (Cout -> Cin); max|dy| was folded into amax_dy by the pipeline backward
        if (convk_split(n, s) && conv5x5_split_supported(s.Cout, s.Cin, s.H, s.W)) launch_conv5x5_split(dyb, s.ws_bwd, nullptr, gin, B, s.Cout, s.Cin, s.H, s.W, c->stream, s.amax_dy, s.amax_w);
        else launch_convk_backward_data(dyb, n->params + s.w_off, gin, c->ws, B, s.Cin, s.Cout, s.H, s.W, s.ksz, c->stream);
      }
      LAUNCHCHK(c);
    } else if (s.kind == ST_CONV) {
      if (s.up) {
        // SpatialUpSamplingNearest(2) + SpatialConvolution backward (adversarial.lua:37-205 trains G through it): the weight
        // gradient needs the up-sampled input, the data gradient is folded back by summing each 2x2 block
        const size_t need = (size_t)B * vol3(s.Cin, s.H, s.W);
        if (need > n->up_cap) {
          HIPCHK(c, hipStreamSynchronize(c->stream));
          (void)hipFree(n->up_tmp[0]); (void)hipFree(n->up_tmp[1]); n->up_tmp[0] = n->up_tmp[1] = nullptr; n->up_cap = 0;
          HIPCHK(c, hipMalloc((void**)&n->up_tmp[0], sizeof(float) * need)); HIPCHK(c, hipMalloc((void**)&n->up_tmp[1], sizeof(float) * need));
          n->up_cap = need;
        }
        launch_upsample2(x, n->up_tmp[0], B, s.Cin, s.H, s.W, c->stream);
        int r = ensure_ws(c, conv_wgrad_workspace_bytes(B, s.Cin, s.Cout, s.H, s.W, c->conv_mode)); if (r) return r;
        if (c->conv_mode == 2 && conv_wgrad_is_split(2, s.Cin, s.W) && s.amax_x_fwd != n->amax_gen) launch_absmax(x, (long)B * vol3(s.inC, s.inH, s.inW), s.amax_x, c->stream);
        launch_conv3x3_wgrad(n->up_tmp[0], dyb, n->grads + s.w_off, c->ws, B, s.Cin, s.Cout, s.H, s.W, c->stream, c->conv_mode, s.amax_x, s.amax_dy);
        if (need_gin) {
          if (c->conv_mode >= 1 && s.Cin > 4) launch_conv3x3_split(dyb, s.ws_bwd, nullptr, n->up_tmp[1], B, s.Cout, s.Cin, s.H, s.W, false, c->stream, nullptr, c->conv_mode == 2 ? 2 : 3, s.amax_dy, s.amax_w);
          else launch_conv3x3(dyb, s.wt_bwd, nullptr, n->up_tmp[1], B, s.Cout, s.Cin, s.H, s.W, false, c->stream);
          launch_downsum2(n->up_tmp[1], gin, B, s.Cin, s.H / 2, s.W / 2, c->stream);
        }
        LAUNCHCHK(c);
      } else if (s.fullconv) {
        // nn.SpatialFullConvolution:accGradParameters: gradWeight[i][o] += x[i] (x) gradOutput[o] = the weight gradient of the convolution
        // Cout -> Cin with the roles of input and gradOutput swapped; gradInput = that convolution's forward of gradOutput
        int r = ensure_ws(c, conv_wgrad_workspace_bytes(B, s.Cout, s.Cin, s.H, s.W, c->conv_mode)); if (r) return r;
        if (c->conv_mode == 2 && conv_wgrad_is_split(2, s.Cout, s.W) && s.amax_x_fwd != n->amax_gen) launch_absmax(x, (long)B * vol3(s.inC, s.inH, s.inW), s.amax_x, c->stream);
        launch_conv3x3_wgrad(dyb, x, n->grads + s.w_off, c->ws, B, s.Cout, s.Cin, s.H, s.W, c->stream, c->conv_mode, s.amax_dy, s.amax_x);
        if (need_gin) {
          if (c->conv_mode >= 1 && s.Cin > 4) launch_conv3x3_split(dyb, s.ws_bwd, nullptr, gin, B, s.Cout, s.Cin, s.H, s.W, false, c->stream, nullptr, c->conv_mode == 2 ? 2 : 3, s.amax_dy, s.amax_w);
          else launch_conv3x3(dyb, s.wt_bwd, nullptr, gin, B, s.Cout, s.Cin, s.H, s.W, false, c->stream);
        }
        LAUNCHCHK(c);
      } else {
      int r = ensure_ws(c, conv_wgrad_workspace_bytes(B, s.Cin, s.Cout, s.H, s.W, c->conv_mode)); if (r) return r;
      if (c->conv_mode == 2) {
        // max|dy| was folded into s.amax_dy by the pipeline-backward kernel that wrote dy_buf
        // x's maximum is current when this stage's forward ran on the f16x3 kernel; otherwise (few-channel input, mode switched) take it now
        if (conv_wgrad_is_split(2, s.Cin, s.W) && s.amax_x_fwd != n->amax_gen) launch_absmax(x, (long)B * vol3(s.inC, s.inH, s.inW), s.amax_x, c->stream);
      }
      // The weight gradient has no consumer before Adam / the gradient all-reduce: it runs on the side stream, beside this
      // stage's data gradient and the memory-bound pipeline kernels of the stages after it (they leave the matrix pipe idle).
      // Its own workspace; the dy pair it reads is not rewritten before ev_wgrad_done[dk] (waited for two stages on).
      const bool side_want = c->side_wgrad < 0 ? (int64_t)B * vol3(s.Cout, s.H, s.W) >= ((int64_t)1 << 26) : c->side_wgrad != 0;
      const bool side = side_want && c->side_stream != nullptr && gr::g_ktimer == nullptr;
      hipStream_t ws_ = side ? c->side_stream : c->stream;
      void* wsp_ = c->ws;
      if (side) {
        r = ensure_ws2(c, wgrad_p16 ? conv_wgrad_p16_workspace_bytes(B, s.Cin, s.Cout, s.H, s.W) : conv_wgrad_workspace_bytes(B, s.Cin, s.Cout, s.H, s.W, c->conv_mode)); if (r) return r;
        wsp_ = c->ws2;
        HIPCHK(c, hipEventRecord(c->ev_dy_ready, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->side_stream, c->ev_dy_ready, 0));
      }
      if (wgrad_p16) {
        if (!side) { r = ensure_ws(c, conv_wgrad_p16_workspace_bytes(B, s.Cin, s.Cout, s.H, s.W)); if (r) return r; wsp_ = c->ws; }
        launch_conv3x3_wgrad_p16(s.x_p16, dyp, n->grads + s.w_off, wsp_, B, s.Cin, s.Cout, s.H, s.W, ws_, s.amax_x, s.amax_dy);
      } else
      launch_conv3x3_wgrad(x, dyb, n->grads + s.w_off, wsp_, B, s.Cin, s.Cout, s.H, s.W, ws_, c->conv_mode, s.amax_x, s.amax_dy);
      if (side) { HIPCHK(c, hipEventRecord(c->ev_wgrad_done[dk], c->side_stream)); n->wg_pending[dk] = true; }
      if (need_gin) {
        // backward-data = the same convolution on the transposed + flipped weights (Cout -> Cin)
        if (dgrad_p16) launch_conv3x3_p16(dyp, s.ws_bwd, nullptr, gin, B, s.Cout, s.Cin, s.H, s.W, c->stream, nullptr, s.amax_dy, s.amax_w, nullptr, nullptr, nullptr);
        else if (c->conv_mode >= 1) launch_conv3x3_split(dyb, s.ws_bwd, nullptr, gin, B, s.Cout, s.Cin, s.H, s.W, false, c->stream, nullptr, c->conv_mode == 2 ? 2 : 3, s.amax_dy, s.amax_w);
        else launch_conv3x3(dyb, s.wt_bwd, nullptr, gin, B, s.Cout, s.Cin, s.H, s.W, false, c->stream);
      }
      LAUNCHCHK(c);
      }
    } else if (s.kind == ST_LINEAR) {
      size_t wsb = gemm_workspace_bytes(s.Cout, s.Cin, B);
      const size_t wsb2 = gemm_workspace_bytes(B, s.Cin, s.Cout);
      if (wsb2 > wsb) wsb = wsb2;
      int r = ensure_ws(c, wsb); if (r) return r;
      const bool f16g = use_f16_gemm(n, s);
      if (f16g && s.amax_x_fwd != n->amax_gen) launch_absmax(x, (long)B * s.Cin, s.amax_x, c->stream);   // (mode switched since the forward)
      // gW[o][i] += sum_b dy[b][o] x[b][i]
      launch_gemm(dyb, 1, s.Cout, x, 1, s.Cin, n->grads + s.w_off, s.Cin, nullptr, true, s.Cout, s.Cin, B, c->ws, c->stream,
                  nullptr, nullptr, f16g ? s.amax_dy : nullptr, f16g ? s.amax_x : nullptr);
      // gx[b][i] = sum_o dy[b][o] W[o][i]
      if (need_gin) launch_gemm(dyb, s.Cout, 1, n->params + s.w_off, 1, s.Cin, gin, s.Cin, nullptr, false, B, s.Cin, s.Cout, c->ws, c->stream,
                                nullptr, nullptr, f16g ? s.amax_dy : nullptr, f16g ? s.amax_w : nullptr);
      LAUNCHCHK(c);
    }
    g = gin;
    if (reduce) {
      // lowest parameter offset this stage owns (its BN parameters follow its main op in flat order)
      int64_t lo = -1;
      if (s.w_off >= 0) lo = s.w_off; else if (s.g_off >= 0) lo = s.g_off;
      if (lo >= 0 && (bucket_hi - lo >= BUCKET_MIN_ELEMS || si == 0)) {
        for (int k2 = 0; k2 < 2; ++k2) if (n->wg_pending[k2]) { HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_wgrad_done[k2], 0)); n->wg_pending[k2] = false; }   // ... and final weight gradients
        launch_bias_grad_batch(bias_jobs, c->stream);             // the bucket must hold final bias gradients
        int r = reduce_bucket(n, lo, bucket_hi); if (r) return r;
        bucket_hi = lo;
      }
    }
  }
  for (int k2 = 0; k2 < 2; ++k2) if (n->wg_pending[k2]) { HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_wgrad_done[k2], 0)); n->wg_pending[k2] = false; }   // join the side stream
  launch_bias_grad_batch(bias_jobs, c->stream);
  LAUNCHCHK(c);
  if (reduce) {
    int r = reduce_bucket(n, 0, bucket_hi); if (r) return r;
    if (!c->xchg) {
      // Adam (compute stream) must see every reduced bucket
      HIPCHK(c, hipEventRecord(c->ev_done, c->comm_stream));
      HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_done, 0));
      r = comm_check(c); if (r) return r;
    }
  }
  return GR_OK;
}

extern "C" int gr_net_backward_dev(gr_net* n, const float* in_dev, const float* gout_dev, int B, float* gin_dev) {
  if (!n || !in_dev || !gout_dev || B <= 0) return GR_ERR_INVALID;
  n->head_fused = false;
  return backward_impl(n, in_dev, gout_dev, B, gin_dev);
}

extern "C" int gr_net_backward_host(gr_net* n, const float* in_host, const float* gout_host, int B, float* gin_host) {
  if (!n || !in_host || !gout_host || B <= 0) return GR_ERR_INVALID;
  gr_ctx* c = n->ctx;
  HIPCHK(c, hipSetDevice(c->device));
  if (B > n->capB) return fail(c, GR_ERR_STATE, "backward before forward");
  // the module caches state from forward; `input` must hold the same values (Torch7 contract), re-upload it
  HIPCHK(c, hipMemcpyAsync(n->in_buf, in_host, sizeof(float) * (size_t)B * vol3(n->inC, n->inH, n->inW), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(n->gout_buf, gout_host, sizeof(float) * (size_t)B * vol3(n->outC, n->outH, n->outW), hipMemcpyHostToDevice, c->stream));
  float* gin_dev = nullptr;
  if (gin_host) gin_dev = n->g_buf[0];   // stage 0 writes g_buf[0] anyway
  bool fall_back = false;
  if (guard_applies(n) && n->keep_fp32) {
    fall_back = n->last_fwd_fell_back;       // hostile input / parameters: they enter the gradients too
    if (!fall_back) {
      const Stage& sl = n->st.back();
      int r = sl.kind == ST_LINEAR ? guard_scan_activation(c, n->gout_buf, B, sl.Cout, 1, 1)
                                   : guard_scan_activation(c, n->gout_buf, B, n->outC, n->outH, n->outW);
      if (r) return r;
      unsigned sides = 0;
      r = guard_verdict(c, &sides); if (r) return r;
      fall_back = guard_over_budget(guard_merge(sides, n->guard_sides));      // gradOutput's spread joins the forward's
    }
  }
  if (fall_back) { c->guard_fallbacks++; c->conv_mode = 1; }
  int r = backward_impl(n, n->in_buf, n->gout_buf, B, gin_dev);
  if (fall_back) c->conv_mode = 2;
  if (r) return r;
  if (gin_host) HIPCHK(c, hipMemcpyAsync(gin_host, gin_dev, sizeof(float) * (size_t)B * vol3(n->inC, n->inH, n->inW), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GR_OK;
}

// ------------------------------------------------------------------ criterion
extern "C" int gr_mse_dev(gr_ctx* c, const float* x, const float* t, int64_t n, int64_t ng, double* loss_dev, float* grad) {
  if (!c || !x || !t || n <= 0 || ng <= 0) return GR_ERR_INVALID;
  launch_mse(x, t, n, ng, loss_dev, grad, c->stream); LAUNCHCHK(c);
  return GR_OK;
}
extern "C" int gr_mse_host(gr_ctx* c, const float* x, const float* t, int64_t n, int64_t ng, double* loss, float* grad) {
  if (!c || !x || !t || n <= 0 || ng <= 0) return GR_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  int r = ensure_ws(c, sizeof(float) * 3 * (size_t)n); if (r) return r;
  float* dx = (float*)c->ws; float* dt = dx + n; float* dg = dt + n;
  HIPCHK(c, hipMemcpyAsync(dx, x, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dt, t, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
  launch_mse(dx, dt, n, ng, c->d_loss, grad ? dg : nullptr, c->stream); LAUNCHCHK(c);
  if (grad) HIPCHK(c, hipMemcpyAsync(grad, dg, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->h_loss, c->d_loss, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (loss) *loss = *c->h_loss;
  return GR_OK;
}
// nn.BCECriterion (sizeAverage): train.lua:173's CRITERION, used by adversarial.lua (the GAN step's loss; first pieces of SURVEY.md 8f rank 4)
extern "C" int gr_bce_dev(gr_ctx* c, const float* x, const float* t, int64_t n, double* loss_dev, float* grad) {
  if (!c || !x || !t || n <= 0) return GR_ERR_INVALID;
  launch_bce(x, t, n, loss_dev, grad, c->stream); LAUNCHCHK(c);
  return GR_OK;
}
extern "C" int gr_bce_host(gr_ctx* c, const float* x, const float* t, int64_t n, double* loss, float* grad) {
  if (!c || !x || !t || n <= 0) return GR_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  int r = ensure_ws(c, sizeof(float) * 3 * (size_t)n); if (r) return r;
  float* dx = (float*)c->ws; float* dt = dx + n; float* dg = dt + n;
  HIPCHK(c, hipMemcpyAsync(dx, x, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dt, t, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
  launch_bce(dx, dt, n, c->d_loss, grad ? dg : nullptr, c->stream); LAUNCHCHK(c);
  if (grad) HIPCHK(c, hipMemcpyAsync(grad, dg, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->h_loss, c->d_loss, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (loss) *loss = *c->h_loss;
  return GR_OK;
}

// ------------------------------------------------------------------ nn.Concat on device-resident tensors (models.lua:293-321)
// The container stays on the host (it is a module that calls its children, not an operator); these two move its data
// without leaving the GPU: rows of one matrix into a column range of another (join the branch outputs / slice gradOutput),
// and the sum of the branches' gradInputs.
extern "C" int gr_copy2d_dev(gr_ctx* c, float* dst, int64_t dst_pitch, const float* src, int64_t src_pitch, int64_t rows, int64_t cols) {
  if (!c || !dst || !src || rows <= 0 || cols <= 0 || dst_pitch < cols || src_pitch < cols) return GR_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpy2DAsync(dst, sizeof(float) * (size_t)dst_pitch, src, sizeof(float) * (size_t)src_pitch, sizeof(float) * (size_t)cols, (size_t)rows,
                             hipMemcpyDeviceToDevice, c->stream));
  return GR_OK;
}
extern "C" int gr_add_dev(gr_ctx* c, float* y, const float* x, int64_t n) {
  if (!c || !y || !x || n <= 0) return GR_ERR_INVALID;
  launch_add_inplace(y, x, (long)n, c->stream); LAUNCHCHK(c);
  return GR_OK;
}

// ------------------------------------------------------------------ optimiser
static AdamConsts adam_consts(const gr_hyper* h, int t) {
  AdamConsts k{};
  k.b1 = (float)h->beta1; k.b2 = (float)h->beta2; k.eps = (float)h->eps;
  k.c1 = (float)(1.0 - h->beta1); k.c2 = (float)(1.0 - h->beta2);
  const double bc1 = 1.0 - std::pow(h->beta1, t), bc2 = 1.0 - std::pow(h->beta2, t);
  k.step = (float)(-(h->lr * std::sqrt(bc2) / bc1));
  k.l1 = (float)h->l1; k.l2 = (float)h->l2; k.clamp = (float)h->clamp;
  k.use_penalty = (h->l1 != 0 || h->l2 != 0) ? 1 : 0;
  k.use_clamp = h->clamp != 0 ? 1 : 0;
  return k;
}
extern "C" int gr_adam_step(gr_net* n, const gr_hyper* h, int t) {
  if (!n || !h || t < 1) return GR_ERR_INVALID;
  gr_ctx* c = n->ctx;
  launch_penalty_clamp_adam(n->params, n->grads, n->adam_m, n->adam_v, n->n_params, adam_consts(h, t), c->stream, head_fault_dev(c));
  LAUNCHCHK(c);
  n->params_version++;
  return GR_OK;
}

// ------------------------------------------------------------------ data parallelism (RCCL over xGMI)
static_assert(sizeof(ncclUniqueId) <= GR_COMM_ID_BYTES, "unique id does not fit");
extern "C" int gr_comm_unique_id(gr_ctx* c, void* id_out) {
  if (!c || !id_out) return GR_ERR_INVALID;
  ncclUniqueId id; NCCLCHK(c, ncclGetUniqueId(&id));
  memset(id_out, 0, GR_COMM_ID_BYTES); memcpy(id_out, &id, sizeof id);
  return GR_OK;
}
extern "C" int gr_comm_init(gr_ctx* c, const void* idb, int nranks, int rank) {
  if (!c || !idb || nranks < 1 || rank < 0 || rank >= nranks) return GR_ERR_INVALID;
  if (c->comm) return fail(c, GR_ERR_STATE, "communicator already initialised");
  if (c->xchg) return fail(c, GR_ERR_STATE, "a host-exchange hook is installed on this context");
  HIPCHK(c, hipSetDevice(c->device));
  ncclUniqueId id; memcpy(&id, idb, sizeof id);
  NCCLCHK(c, ncclCommInitRank(&c->comm, nranks, id, rank));
  c->nranks = nranks; c->rank = rank;
  return ensure_stat_comm(c);
}
extern "C" int gr_comm_destroy(gr_ctx* c) {
  if (!c) return GR_ERR_INVALID;
  if (c->comm) {
    HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipStreamSynchronize(c->comm_stream));
    if (c->stat_comm) { NCCLCHK(c, ncclCommDestroy(c->stat_comm)); c->stat_comm = nullptr; }
    NCCLCHK(c, ncclCommDestroy(c->comm)); c->comm = nullptr;
  }
  c->nranks = 1; c->rank = 0;
  return GR_OK;
}
// Host-exchange hook: stands in for the RCCL collectives of this context (gradient / loss / BatchNorm-statistics all-reduce).  What
// SURVEY.md section 4 asks for on a box whose ranks cannot each own a GPU: "a fake comm that sums host buffers in-process so the
// sharding / reduction logic is testable" - here with the HIP path as the compute (tests/test_gpu_syncbn.py: two contexts on one GPU).
extern "C" int gr_comm_set_host_exchange(gr_ctx* c, int nranks, int rank, gr_exchange_fn fn, void* user) {
  if (!c || nranks < 1 || rank < 0 || rank >= nranks) return GR_ERR_INVALID;
  if (c->comm) return fail(c, GR_ERR_STATE, "an RCCL communicator is active on this context");
  c->xchg = fn; c->xchg_user = user;
  c->nranks = fn ? nranks : 1; c->rank = fn ? rank : 0;
  return GR_OK;
}
extern "C" int gr_comm_ranks(gr_ctx* c, int* nr, int* r) { if (!c) return GR_ERR_INVALID; if (nr) *nr = c->nranks; if (r) *r = c->rank; return GR_OK; }
extern "C" int gr_allreduce_dev(gr_ctx* c, float* buf, int64_t n) {
  if (!c || !buf || n <= 0) return GR_ERR_INVALID;
  if (c->xchg) return small_allreduce(c, buf, (long)n, 0);
  if (!c->comm) return GR_OK;          // (a one-rank communicator still goes through RCCL: the path a 1-GPU box can exercise)
  NCCLCHK(c, ncclAllReduce(buf, buf, (size_t)n, ncclFloat, ncclSum, c->comm, c->stream));
  return comm_check(c);
}
// ncclAllGather of `bytes` bytes per rank (the sharded search's candidate exchange, SURVEY.md 8e: Q * k * 12 bytes per rank);
// with one rank (or no communicator) the rank's own block is copied to slot 0.
extern "C" int gr_allgather_dev(gr_ctx* c, const void* send, void* recv, int64_t bytes) {
  if (!c || !send || !recv || bytes <= 0) return GR_ERR_INVALID;
  if (!c->comm) {
    if (send != recv) HIPCHK(c, hipMemcpyAsync(recv, send, (size_t)bytes, hipMemcpyDeviceToDevice, c->stream));
    return GR_OK;
  }
  NCCLCHK(c, ncclAllGather(send, recv, (size_t)bytes, ncclChar, c->comm, c->stream));
  return comm_check(c);
}
extern "C" int gr_allreduce_grads(gr_net* n) { if (!n) return GR_ERR_INVALID; return gr_allreduce_dev(n->ctx, n->grads, n->n_params); }
extern "C" int gr_broadcast_params(gr_net* n, int root) {
  if (!n) return GR_ERR_INVALID;
  gr_ctx* c = n->ctx;
  if (c->nranks <= 1 || !c->comm) return GR_OK;
  NCCLCHK(c, ncclBroadcast(n->params, n->params, (size_t)n->n_params, ncclFloat, root, c->comm, c->stream));
  for (auto& s : n->st) if (s.has_bn) {
    NCCLCHK(c, ncclBroadcast(s.run_mean, s.run_mean, (size_t)s.Cout, ncclFloat, root, c->comm, c->stream));
    NCCLCHK(c, ncclBroadcast(s.run_var, s.run_var, (size_t)s.Cout, ncclFloat, root, c->comm, c->stream));
    s.eval_ready = false;        // the evaluate()-mode constants cached from the old running statistics are stale
  }
  n->params_version++;
  return comm_check(c);
}

// ------------------------------------------------------------------ the whole train_r.lua:138-170 iteration
// Can R's last two stages run in the head kernel?  Linear -> BatchNorm -> activation -> [Dropout] followed by Linear [-> Tanh], training mode, per-rank
// BatchNorm statistics (synchronised BatchNorm adds a collective between the phases: stage-by-stage path), shapes the kernel covers.  Fills every field of
// the launch that depends on the net only.
static bool head_plan(gr_net* n, int B, HeadLaunch& h) {
  gr_ctx* c = n->ctx;
  const size_t nst = n->st.size();
  if (nst < 3 || !n->training || (c->sync_bn && have_peers(c)) || n->keep_fp32) return false;
  Stage& s1 = n->st[nst - 2]; Stage& s2 = n->st[nst - 1];
  if (s1.kind != ST_LINEAR || s2.kind != ST_LINEAR || !s1.has_post || !s1.has_bn || s1.pool || s1.m2 >= 0 || s1.H != 1 || s1.W != 1) return false;
  if (s1.act == ACT_PRELU || s2.has_bn || s2.pool || s2.m1 >= 0 || s2.m2 >= 0 || !(s2.act == ACT_NONE || s2.act == ACT_TANH)) return false;
  if (s2.Cin != s1.Cout || !head_supported(B, s1.Cout, s2.Cout) || B > n->capB) return false;      // (gr_train_r_step has sized the buffers: ensure_batch)
  if (c->cu_count < s1.Cout / 8) return false;                 // the grid barrier needs its C1 / 8 workgroups (one per CU: 256 threads at one wave per SIMD) resident together - not on a partition with fewer CUs
  bool need = false;
  const MaskRef m1 = mask_ref(n, s1.m1, need);
  if (!(m1.kind == MASK_NONE || m1.kind == MASK_ELEM)) return false;
  const int i1 = (int)nst - 2, i2 = (int)nst - 1;
  h.B = B; h.C1 = s1.Cout; h.nd = s2.Cout;
  h.y1 = s1.y; h.out1 = s1.out;
  h.mean = s1.mean; h.invstd = s1.invstd; h.run_mean = s1.run_mean; h.run_var = s1.run_var; h.gamma = n->params + s1.g_off; h.beta = n->params + s1.be_off;
  h.m1 = m1; h.act1 = s1.act; h.slope1 = s1.slope; h.act2 = s2.act;
  h.W2 = n->params + s2.w_off; h.b2 = n->params + s2.b_off; h.y2 = s2.y; h.out2 = s2.has_post ? s2.out : s2.y;
  h.gout = n->gout_buf; h.gy2 = (i2 & 1) ? n->dy_buf_b : n->dy_buf; h.dy1 = (i1 & 1) ? n->dy_buf_b : n->dy_buf;
  h.gW2 = n->grads + s2.w_off; h.gb2 = n->grads + s2.b_off; h.ggamma = n->grads + s1.g_off; h.gbeta = n->grads + s1.be_off; h.gb1 = n->grads + s1.b_off;
  h.amax_dy = (c->conv_mode == 2 && use_f16_gemm(n, s1)) ? s1.amax_dy : nullptr;
  return h.y1 && h.out1 && h.y2 && h.out2 && h.gout && h.gy2 && h.dy1;
}

extern "C" int gr_train_r_step(gr_net* g, gr_net* rn, const float* noise_dev, int B, int GB, const gr_hyper* h, int t, double* loss_out) {
  if (!g || !rn || !noise_dev || !h || B <= 0 || GB < B || t < 1) return GR_ERR_INVALID;
  gr_ctx* c = rn->ctx;
  if (g->ctx != c) return fail(c, GR_ERR_INVALID, "G and R live on different contexts");
  const int64_t nd = vol3(rn->outC, rn->outH, rn->outW);
  if (vol3(g->inC, g->inH, g->inW) != nd) return fail(c, GR_ERR_INVALID, "noise dim mismatch: G takes %lld, R emits %lld", (long long)vol3(g->inC, g->inH, g->inW), (long long)nd);
  if (vol3(g->outC, g->outH, g->outW) != vol3(rn->inC, rn->inH, rn->inW)) return fail(c, GR_ERR_INVALID, "image dim mismatch between G and R");
  if (c->sync_bn && have_peers(c) && (int64_t)B * c->nranks != GB)
    return fail(c, GR_ERR_INVALID, "synchronised BatchNorm needs equal shards: batch %d x %d ranks != global batch %d", B, c->nranks, GB);
  const bool tm = c->timing;
  int r;
  // Per-step state of the two nets that must not outlive this call on ANY exit (an early error return used to leave head_fused set: a later gr_net_forward_* /
  // gr_net_backward_* on the same net then skipped its last two stages silently; likewise the 'slots already zeroed' and 'forward already begun' notes)
  struct StepState { gr_net* g; gr_net* rn; ~StepState() { rn->head_fused = false; g->amax_prezeroed_groups = rn->amax_prezeroed_groups = 0; rn->begun_B = 0; } } step_state{g, rn};
  // Range guard of the device-resident loop: no synchronisation is allowed here, so the parameter scans (weights, BatchNorm
  // scales of G and R) run every GUARD_PERIOD-th step and their verdict is read, without waiting, by a later call.  Once a
  // hostile spread shows, the context stays on bf16x6 (gr_set_tuning "range_guard" 0 clears it).  Latency: under 2 periods.
  enum { GUARD_PERIOD = 64 };
  g->keep_fp32 = rn->keep_fp32 = false; g->last_fwd_fell_back = rn->last_fwd_fell_back = false;
  if (c->guard_pending && hipEventQuery(c->ev_guard) == hipSuccess) {
    c->guard_pending = false;
    if (guard_over_budget(*guard_alarm_host(c)) && !c->guard_tripped) { c->guard_tripped = true; c->guard_fallbacks++; }
  }
  if (c->guard_tripped && c->conv_mode == 2) c->conv_mode = 1;
  if (c->conv_mode == 2 && c->range_guard && !c->guard_pending && t % GUARD_PERIOD == 1) {
    r = guard_scan_params(g); if (r) return r;
    r = guard_scan_params(rn); if (r) return r;
    HIPCHK(c, hipMemcpyAsync((void*)guard_alarm_host(c), guard_alarm_dev(c), sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemsetAsync(guard_alarm_dev(c), 0, sizeof(unsigned), c->stream));
    HIPCHK(c, hipEventRecord(c->ev_guard, c->stream));
    c->guard_pending = true;
  }
  if (tm) (void)hipEventRecord(c->ev[0], c->stream);
  g->training = false;                                         // train_r.lua:70  MODEL_G:evaluate()
  rn->training = true;
  {
    // ONE fill per step: the f16x3 scale slots of both nets (what their forwards would each zero themselves) and R's gradient vector
    // (train_r.lua:143 gradParameters:zero()) - three hipMemsetAsync kernels of ~6 us each at batch 256 otherwise
    ZeroJobs z{}; z.n = 0;
    static const bool one_fill = !GR_KNOB_SET("GR_NO_STEP_FILL");       // A/B control: every forward zeroes its own slots, the gradients get their own fill
    if (c->conv_mode == 2 && one_fill) {
      const int gg = AG_KB;                                                                      // G: evaluate() mode
      const int gr_ = rn->prepped_version[2] != rn->params_version ? AMAX_GROUPS : AG_W;         // R: training; the w group when the weight images are stale
      z.ptr[z.n] = g->amax; z.n16[z.n++] = (long)(sizeof(unsigned) * AMAX_WORDS * g->st.size() * gg / 16);
      z.ptr[z.n] = rn->amax; z.n16[z.n++] = (long)(sizeof(unsigned) * AMAX_WORDS * rn->st.size() * gr_ / 16);
      g->amax_prezeroed_groups = gg; rn->amax_prezeroed_groups = gr_;
    }
    if (one_fill && rn->n_params % 4 == 0 && ((uintptr_t)rn->grads & 15) == 0) { z.ptr[z.n] = rn->grads; z.n16[z.n++] = (long)(rn->n_params / 4); }
    else { r = gr_net_zero_grads(rn); if (r) return r; }
    g_kphase = 1;
    launch_zero_regions(z, c->stream);
    LAUNCHCHK(c);
  }
  g_kphase = 1;
  // R's preparation for this step (weight images and maxima of the parameters Adam just wrote, Dropout noise) depends on nothing G computes and could run on
  // the side stream beside G's forward.  Measured (round 5, same box, interleaved: profiles/r05_ab_prep_overlap_cfg2.txt): the step gets SLOWER, 1.920-1.942 ->
  // 1.960-1.968 ms at cfg2 - the three launches cost 25 us in line, the two event hand-overs and the small kernels' workgroups squeezing in between G's
  // matrix-pipe-filling workgroups cost more.  Ablation build only (GR_PREP_OVERLAP=1); the shipping library runs them in line.
  static const int prep_overlap = GR_KNOB("GR_PREP_OVERLAP", 0);
  const bool prep_side = prep_overlap && c->side_stream != nullptr && gr::g_ktimer == nullptr && rn->capB >= B;
  if (prep_side) {
    HIPCHK(c, hipEventRecord(c->ev_prep_go, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->side_stream, c->ev_prep_go, 0));
    hipStream_t main_stream = c->stream;
    c->stream = c->side_stream;
    g_kphase = 2;
    r = forward_begin(rn, B);
    g_kphase = 1;
    c->stream = main_stream;
    if (r) { rn->begun_B = 0; return r; }
    HIPCHK(c, hipEventRecord(c->ev_prep_done, c->side_stream));
  }
  { PhaseRange pr("G forward"); r = forward_impl(g, noise_dev, B); } if (r) { rn->begun_B = 0; return r; }          // train_r.lua:139
  if (prep_side) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_prep_done, 0));
  const float* images = g->st.back().out;
  if (tm) (void)hipEventRecord(c->ev[1], c->stream);
  g_kphase = 2;
  HeadLaunch hl{};
  r = ensure_batch(rn, B); if (r) return r;                    // (the plan below takes buffer addresses)
  rn->head_fused = c->fused_head && head_plan(rn, B, hl);
  { PhaseRange pr("R forward"); r = forward_impl(rn, images, B); } if (r) { rn->head_fused = false; return r; }            // :146
  if (tm) (void)hipEventRecord(c->ev[2], c->stream);
  g_kphase = 3;
  {
  PhaseRange pr("loss");
  if (rn->head_fused) {
    // fc1's BatchNorm / activation / Dropout, fc2, the criterion (:147,150) and their backward down to fc1's dy: one launch (elem.hip, head_fwd_bwd_kernel)
    if (!c->head_bar) {
      unsigned* bar = nullptr; double* part = nullptr;           // both or neither: a half-made pair would launch the kernel with a null loss_part next time
      if (hipMalloc((void**)&bar, 256) != hipSuccess || hipMalloc((void**)&part, sizeof(double) * 512) != hipSuccess) { if (bar) (void)hipFree(bar); return fail(c, GR_ERR_HIP, "head kernel: allocation failed"); }
      c->head_bar = bar; c->head_loss_part = part;
      HIPCHK(c, hipMemsetAsync(c->head_bar, 0, 256, c->stream));
      c->head_bar_count = 0;
    }
    if (!head_plan(rn, B, hl)) { rn->head_fused = false; return fail(c, GR_ERR_STATE, "head kernel: the plan changed during the forward"); }   // (again: the forward may have re-allocated the Dropout bits)
    hl.n_global = (long)GB * nd; hl.target = noise_dev; hl.loss = c->d_loss; hl.loss_part = c->head_loss_part;
    hl.bar = c->head_bar; hl.bar_base = c->head_bar_count;
    hl.fault = head_fault_dev(c); hl.spin_limit = 0;
    c->head_bar_count += 2u * (unsigned)(hl.C1 / 8);
    if (c->head_fault_inject) { hl.bar_base += 1u << 30; hl.spin_limit = 1 << 10; c->head_fault_inject = 0; }      // (test hook: targets no arrival count reaches)
    launch_head_fwd_bwd(hl, c->stream);
    c->head_unchecked = true;
  } else
  launch_mse(rn->st.back().out, noise_dev, (long)B * nd, (long)GB * nd, c->d_loss, rn->gout_buf, c->stream);  // :147,150
  LAUNCHCHK(c);
  if (c->xchg) { r = small_allreduce(c, c->d_loss, 1, 1); if (r) return r; }
  else if (c->comm) {   // global MSE = sum of the ranks' partial means (same communicator, same stream as the gradient buckets)
    HIPCHK(c, hipEventRecord(c->ev_ready, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->comm_stream, c->ev_ready, 0));
    NCCLCHK(c, ncclAllReduce(c->d_loss, c->d_loss, 1, ncclDouble, ncclSum, c->comm, c->comm_stream));
  }
  }
  if (tm) (void)hipEventRecord(c->ev[3], c->stream);
  // the penalty and the clamp are non-linear in g (train_r.lua:154-165): the SUM over ranks comes first.  It is issued
  // bucket by bucket from inside backward on the comm stream; the compute stream waits for it only here.
  g_kphase = 4;
  { PhaseRange pr(c->comm ? "R backward + all-reduce" : "R backward"); r = backward_impl(rn, images, rn->gout_buf, B, nullptr, /*reduce=*/true); }   // :151
  rn->head_fused = false;
  if (r) return r;
  if (tm) (void)hipEventRecord(c->ev[4], c->stream);

  if (tm) (void)hipEventRecord(c->ev[5], c->stream);
  g_kphase = 5;
  { PhaseRange pr("penalty + clamp + Adam"); r = gr_adam_step(rn, h, t); } g_kphase = 0; if (r) return r;   // :153-170
  if (tm) (void)hipEventRecord(c->ev[6], c->stream);
  if (loss_out || tm) {
    HIPCHK(c, hipMemcpyAsync(c->h_loss, c->d_loss, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (loss_out) *loss_out = *c->h_loss;
    if (tm) for (int i = 0; i < 6; ++i) HIPCHK(c, hipEventElapsedTime(&c->times[i], c->ev[i], c->ev[i + 1]));
    return head_fault_check(c);       // (this call has waited for the stream anyway: a timed-out grid barrier of this or an earlier step surfaces here)
  }
  return GR_OK;
}

// ------------------------------------------------------------------ search
extern "C" int gr_cosine_topk_dev(gr_ctx* c, const float* emb, int64_t N, int d, const int64_t* qrows, int Q, int k,
                                  int64_t* idx_out, float* score_out, int accf) {
  if (!c || !emb || !qrows || !idx_out || N <= 0 || d <= 0 || Q <= 0 || k <= 0) return GR_ERR_INVALID;
  if (k > N) k = (int)N;
  for (int q = 0; q < Q; ++q) if (qrows[q] < 0 || qrows[q] >= N) return fail(c, GR_ERR_INVALID, "query row %lld out of range", (long long)qrows[q]);
  if (k > 1024) return fail(c, GR_ERR_UNSUPPORTED, "k > 1024");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t tail = sizeof(long) * (size_t)Q * (k + 1) + sizeof(float) * (size_t)Q * k + 512;
  const size_t wsb = cosine_topk_workspace_bytes(N, d, Q, k);
  int r = ensure_ws(c, wsb + tail); if (r) return r;
  char* base = (char*)c->ws + ((wsb + 255) & ~(size_t)255);
  // results [idx | scores | status] are contiguous on the device: ONE copy into pinned memory and one wait per search (three
  // copies into pageable memory cost about 30 us of the 0.25 ms a cfg5 search takes)
  long* d_q = (long*)base; long* d_idx = d_q + Q; float* d_sc = (float*)(d_idx + (size_t)Q * k); unsigned* d_status = (unsigned*)(d_sc + (size_t)Q * k);
  const size_t res_bytes = sizeof(long) * (size_t)Q * k + sizeof(float) * (size_t)Q * k + sizeof(unsigned);
  if (res_bytes > c->pin_bytes) {
    if (c->pin) (void)hipHostFree(c->pin);
    c->pin = nullptr; c->pin_bytes = 0;
    HIPCHK(c, hipHostMalloc(&c->pin, res_bytes * 2));
    c->pin_bytes = res_bytes * 2;
  }
  static const bool filter_on = !GR_KNOB_SET("GR_SEARCH_UNFILTERED");
  // A handful of needles (the reference's five): their rows travel in the kernel arguments and the kernels write idx | scores | status
  // straight into the pinned result block (host memory the device can address): no upload, no copy-out - launches, one wait.
  if (filter_on && cosine_topk_small_path(N, d, Q, k)) {
    void* pin_dev = nullptr;
    HIPCHK(c, hipHostGetDevicePointer(&pin_dev, c->pin, 0));
    char* pd = static_cast<char*>(pin_dev);
    long* p_idx = reinterpret_cast<long*>(pd); float* p_sc = reinterpret_cast<float*>(pd + sizeof(long) * (size_t)Q * k);
    unsigned* p_status = reinterpret_cast<unsigned*>(pd + res_bytes - sizeof(unsigned));
    if (!c->pin_done) { HIPCHK(c, hipHostMalloc((void**)&c->pin_done, 64)); memset(c->pin_done, 0, 64); }
    void* done_dev = nullptr;
    HIPCHK(c, hipHostGetDevicePointer(&done_dev, c->pin_done, 0));
    if (++c->search_seq == 0u) c->search_seq = 1u;                                  // never 0: a fresh block reads 0
    const unsigned seq = c->search_seq;
    if (!c->search_state) {       // the sample launch's arrival counter and histogram: zero now, left zero by every search
      HIPCHK(c, hipMalloc((void**)&c->search_state, sizeof(unsigned) * SEARCH_STATE_WORDS));
      HIPCHK(c, hipMemsetAsync(c->search_state, 0, sizeof(unsigned) * SEARCH_STATE_WORDS, c->stream));
    }
    const int lr = launch_cosine_topk(emb, N, d, d_q, Q, k, p_idx, p_sc, accf, c->ws, c->stream, p_status, 0, qrows, c->search_state,
                                      static_cast<unsigned*>(done_dev), seq);
    if (lr < 0) return fail(c, GR_ERR_UNSUPPORTED, "cosine_topk: unsupported size");
    LAUNCHCHK(c);
    // The selection kernel publishes one completion word per needle behind its results (system-scope release): poll them instead of
    // synchronising the stream (measured: 1-3 us of a 0.15 ms search).  Bounded: after 20 ms the stream is synchronised after all (a fault
    // shows up there).
    static const bool poll_on = !GR_KNOB_SET("GR_SEARCH_NO_POLL");
    bool seen = false;
    if (lr == 2 && poll_on) {
      volatile unsigned* dw = c->pin_done;
      const auto t0 = std::chrono::steady_clock::now();
      for (unsigned spins = 0;; ++spins) {
        bool all = true;
        for (int q = 0; q < Q; ++q) if (dw[q] != seq) { all = false; break; }
        if (all) { seen = true; break; }
        if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
        __builtin_ia32_pause();
      }
      std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (!seen) HIPCHK(c, hipStreamSynchronize(c->stream));
    const char* hres = (const char*)c->pin;
    unsigned status; memcpy(&status, hres + res_bytes - sizeof(unsigned), sizeof status);
    if (status == 0) {
      memcpy(idx_out, hres, sizeof(long) * (size_t)Q * k);
      if (score_out) memcpy(score_out, hres + sizeof(long) * (size_t)Q * k, sizeof(float) * (size_t)Q * k);
      return GR_OK;
    }
    c->search_reruns++;       // a candidate list overflowed (adversarial row order): the unfiltered search below decides
  }
  HIPCHK(c, hipMemcpyAsync(d_q, qrows, sizeof(long) * Q, hipMemcpyHostToDevice, c->stream));
  const bool small_failed = filter_on && cosine_topk_small_path(N, d, Q, k);
  for (int unfiltered = (filter_on && !small_failed) ? 0 : 1; unfiltered < 2; ++unfiltered) {
    if (launch_cosine_topk(emb, N, d, d_q, Q, k, d_idx, d_sc, accf, c->ws, c->stream, d_status, unfiltered, qrows)) return fail(c, GR_ERR_UNSUPPORTED, "cosine_topk: unsupported size");
    LAUNCHCHK(c);
    HIPCHK(c, hipMemcpyAsync(c->pin, d_idx, res_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const char* h = (const char*)c->pin;
    unsigned status; memcpy(&status, h + res_bytes - sizeof(unsigned), sizeof status);
    if (status == 0 || unfiltered) {
      memcpy(idx_out, h, sizeof(long) * (size_t)Q * k);
      if (score_out) memcpy(score_out, h + sizeof(long) * (size_t)Q * k, sizeof(float) * (size_t)Q * k);
      break;
    }
    c->search_reruns++;       // 1: the sample-bound filter overflowed (adversarial row order): rerun on every key
  }
  return GR_OK;
}
extern "C" int gr_cosine_topk_host(gr_ctx* c, const float* emb, int64_t N, int d, const int64_t* qrows, int Q, int k,
                                   int64_t* idx_out, float* score_out, int accf) {
  if (!c || !emb || N <= 0 || d <= 0) return GR_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  float* dev = nullptr;
  HIPCHK(c, hipMalloc((void**)&dev, sizeof(float) * (size_t)N * d));
  hipError_t e = hipMemcpyAsync(dev, emb, sizeof(float) * (size_t)N * d, hipMemcpyHostToDevice, c->stream);
  int r = e == hipSuccess ? gr_cosine_topk_dev(c, dev, N, d, qrows, Q, k, idx_out, score_out, accf) : fail(c, GR_ERR_HIP, "upload failed");
  (void)hipStreamSynchronize(c->stream);
  (void)hipFree(dev);
  return r;
}
extern "C" int gr_cosine_similarity_host(gr_ctx* c, const float* a, const float* b, int d, float* out) {
  if (!c || !a || !b || !out || d <= 0) return GR_ERR_INVALID;
  std::vector<float> two((size_t)2 * d);
  memcpy(two.data(), a, sizeof(float) * d); memcpy(two.data() + d, b, sizeof(float) * d);
  int64_t q = 0, idx[2]; float sc[2];
  int r = gr_cosine_topk_host(c, two.data(), 2, d, &q, 1, 2, idx, sc, 0); if (r) return r;
  *out = idx[0] == 1 ? sc[0] : sc[1];   // score of row 1 against needle row 0
  return GR_OK;
}

// ------------------------------------------------------------------ apply_r.lua:197-217 clustering of the recovered noise
extern "C" int gr_kmeans_host(gr_ctx* c, const float* x, int64_t n, int d, int k, int niter, float* cent, float* totalcounts, int32_t* labels) {
  if (!c || !x || !cent || n <= 0 || d <= 0 || k <= 0 || niter < 0) return GR_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t xb = sizeof(float) * (size_t)n * d, cb = sizeof(float) * (size_t)k * d, wsb = kmeans_workspace_bytes(n, d, k);
  int r = ensure_ws(c, wsb + xb + cb + sizeof(float) * 3 * (size_t)k + sizeof(int) * (size_t)n + 1024); if (r) return r;
  char* p = (char*)c->ws + ((wsb + 255) & ~(size_t)255);
  float* dx = (float*)p; p += xb;
  float* dc = (float*)p; p += cb;
  float* dc2 = (float*)p; float* dcnt = dc2 + k; float* dtot = dcnt + k; p += sizeof(float) * 3 * (size_t)k;
  int* dlab = (int*)(((uintptr_t)p + 15) & ~(uintptr_t)15);
  HIPCHK(c, hipMemcpyAsync(dx, x, xb, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dc, cent, cb, hipMemcpyHostToDevice, c->stream));
  if (launch_kmeans(dx, n, d, k, niter, dc, dc2, dcnt, dtot, dlab, c->ws, c->stream)) return fail(c, GR_ERR_UNSUPPORTED, "kmeans: k <= 32 and d <= 256 only");
  LAUNCHCHK(c);
  HIPCHK(c, hipMemcpyAsync(cent, dc, cb, hipMemcpyDeviceToHost, c->stream));
  if (totalcounts) HIPCHK(c, hipMemcpyAsync(totalcounts, dtot, sizeof(float) * k, hipMemcpyDeviceToHost, c->stream));
  if (labels && niter > 0) HIPCHK(c, hipMemcpyAsync(labels, dlab, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GR_OK;
}
extern "C" int gr_cosine_assign_host(gr_ctx* c, const float* x, int64_t n, int d, const float* cent, int k, int take_min, int32_t* labels, float* sims) {
  if (!c || !x || !cent || !labels || !sims || n <= 0 || d <= 0 || k <= 0) return GR_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t xb = sizeof(float) * (size_t)n * d, cb = sizeof(float) * (size_t)k * d;
  int r = ensure_ws(c, xb + cb + sizeof(float) * (size_t)k + (sizeof(int) + sizeof(float)) * (size_t)n + 1024); if (r) return r;
  char* p = (char*)c->ws;
  float* dx = (float*)p; p += xb;
  float* dc = (float*)p; p += cb;
  float* dw = (float*)p; p += sizeof(float) * (size_t)k;
  p = (char*)(((uintptr_t)p + 15) & ~(uintptr_t)15);
  int* dlab = (int*)p; p += sizeof(int) * (size_t)n;
  float* dsim = (float*)p;
  HIPCHK(c, hipMemcpyAsync(dx, x, xb, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dc, cent, cb, hipMemcpyHostToDevice, c->stream));
  if (launch_cosine_assign(dx, n, d, dc, k, take_min, dw, dlab, dsim, c->stream)) return fail(c, GR_ERR_UNSUPPORTED, "cosine_assign: unsupported size");
  LAUNCHCHK(c);
  HIPCHK(c, hipMemcpyAsync(labels, dlab, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(sims, dsim, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GR_OK;
}

// ------------------------------------------------------------------ apply_r.lua:355-372 detectAnomalies' distance
extern "C" int gr_l2_distance_rows_host(gr_ctx* c, const float* a, const float* b, int64_t n, int64_t d, double* out) {
  if (!c || !a || !b || !out || n <= 0 || d <= 0) return GR_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t nb = sizeof(float) * (size_t)n * d;
  int r = ensure_ws(c, 2 * nb + sizeof(double) * (size_t)n + 256); if (r) return r;
  float* da = (float*)c->ws; float* db = da + (size_t)n * d; double* dout = (double*)((char*)c->ws + ((2 * nb + 255) & ~(size_t)255));
  HIPCHK(c, hipMemcpyAsync(da, a, nb, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(db, b, nb, hipMemcpyHostToDevice, c->stream));
  launch_l2_distance_rows(da, db, n, d, dout, c->stream); LAUNCHCHK(c);
  HIPCHK(c, hipMemcpyAsync(out, dout, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GR_OK;
}

// ------------------------------------------------------------------ single-kernel entry points
static int with_prepped(gr_ctx* c, const float* w, int cin, int cout, bool bwd, float** wt) {
  const ConvWeightLayout L = bwd ? conv_weight_layout(cout, cin) : conv_weight_layout(cin, cout);
  HIPCHK(c, hipMalloc((void**)wt, sizeof(float) * L.elems()));
  launch_conv_weight_prep(w, *wt, cin, cout, bwd, c->stream);
  LAUNCHCHK(c);
  return GR_OK;
}
// f16x3: the weight maximum goes to c->amax[2]
static int conv_split_once(gr_ctx* c, const float* w, int cin, int cout, bool bwd, void** ws) {
  HIPCHK(c, hipMalloc(ws, conv_weight_split_bytes(cin, cout, bwd)));
  launch_conv_weight_split(w, *ws, cin, cout, bwd, c->stream, c->conv_mode == 2 ? 2 : 3, c->amax + 2 * AMAX_WORDS);
  LAUNCHCHK(c);
  return GR_OK;
}
extern "C" int gr_conv3_forward_dev(gr_ctx* c, const float* in, const float* w, const float* bias, float* out, int B, int cin, int cout, int h, int wd, int up) {
  if (!c || !in || !w || !out) return GR_ERR_INVALID;
  if (c->conv_mode == 2 && up && conv_up2_supported(cin, cout, h, wd) && !GR_KNOB_SET("GR_NO_UP2")) {
    // the fused up-sampling layer as four 2x2 convolutions (the path a net takes for such a stage in f16x3 mode)
    void* wup = nullptr;
    HIPCHK(c, hipMalloc(&wup, conv_weight_up2_bytes(cin, cout)));
    launch_conv_weight_up2_split(w, wup, cin, cout, c->stream, c->amax + 2 * AMAX_WORDS, true);
    launch_absmax(in, (long)B * cin * (h / 2) * (wd / 2), c->amax, c->stream);
    launch_conv3x3_up2_f16x3(in, wup, bias, out, B, cin, cout, h, wd, c->stream, nullptr, c->amax, c->amax + 2 * AMAX_WORDS, nullptr);
    hipError_t e = hipGetLastError(); (void)hipStreamSynchronize(c->stream); (void)hipFree(wup);
    return e == hipSuccess ? GR_OK : fail(c, GR_ERR_HIP, "conv launch failed: %s", hipGetErrorString(e));
  }
  if (c->conv_mode >= 1 && cout > 4) {
    void* ws = nullptr; int r = conv_split_once(c, w, cin, cout, false, &ws); if (r) return r;
    if (c->conv_mode == 2) launch_absmax(in, (long)B * cin * (up ? (h / 2) * (wd / 2) : h * wd), c->amax, c->stream);
    launch_conv3x3_split(in, ws, bias, out, B, cin, cout, h, wd, up != 0, c->stream, nullptr, c->conv_mode == 2 ? 2 : 3, c->amax, c->amax + 2 * AMAX_WORDS);
    hipError_t e = hipGetLastError(); (void)hipStreamSynchronize(c->stream); (void)hipFree(ws);
    return e == hipSuccess ? GR_OK : fail(c, GR_ERR_HIP, "conv launch failed: %s", hipGetErrorString(e));
  }
  float* wt = nullptr; int r = with_prepped(c, w, cin, cout, false, &wt); if (r) return r;
  launch_conv3x3(in, wt, bias, out, B, cin, cout, h, wd, up != 0, c->stream, w);
  hipError_t e = hipGetLastError(); (void)hipStreamSynchronize(c->stream); (void)hipFree(wt);
  return e == hipSuccess ? GR_OK : fail(c, GR_ERR_HIP, "conv launch failed: %s", hipGetErrorString(e));
}
extern "C" int gr_conv3_backward_data_dev(gr_ctx* c, const float* gout, const float* w, float* gin, int B, int cin, int cout, int h, int wd) {
  if (!c || !gout || !w || !gin) return GR_ERR_INVALID;
  if (c->conv_mode >= 1 && cin > 4) {
    void* ws = nullptr; int r = conv_split_once(c, w, cin, cout, true, &ws); if (r) return r;
    if (c->conv_mode == 2) launch_absmax(gout, (long)B * cout * h * wd, c->amax + AMAX_WORDS, c->stream);
    launch_conv3x3_split(gout, ws, nullptr, gin, B, cout, cin, h, wd, false, c->stream, nullptr, c->conv_mode == 2 ? 2 : 3, c->amax + AMAX_WORDS, c->amax + 2 * AMAX_WORDS);
    hipError_t e = hipGetLastError(); (void)hipStreamSynchronize(c->stream); (void)hipFree(ws);
    return e == hipSuccess ? GR_OK : fail(c, GR_ERR_HIP, "conv launch failed: %s", hipGetErrorString(e));
  }
  float* wt = nullptr; int r = with_prepped(c, w, cin, cout, true, &wt); if (r) return r;
  launch_conv3x3(gout, wt, nullptr, gin, B, cout, cin, h, wd, false, c->stream);
  hipError_t e = hipGetLastError(); (void)hipStreamSynchronize(c->stream); (void)hipFree(wt);
  return e == hipSuccess ? GR_OK : fail(c, GR_ERR_HIP, "conv launch failed: %s", hipGetErrorString(e));
}
extern "C" int gr_conv3_backward_weight_dev(gr_ctx* c, const float* in, const float* gout, float* gw, int B, int cin, int cout, int h, int wd) {
  if (!c || !in || !gout || !gw) return GR_ERR_INVALID;
  int r = ensure_ws(c, conv_wgrad_workspace_bytes(B, cin, cout, h, wd, c->conv_mode)); if (r) return r;
  if (c->conv_mode == 2 && conv_wgrad_is_split(2, cin, wd)) {
    launch_absmax(in, (long)B * cin * h * wd, c->amax, c->stream);
    launch_absmax(gout, (long)B * cout * h * wd, c->amax + AMAX_WORDS, c->stream);
  }
  launch_conv3x3_wgrad(in, gout, gw, c->ws, B, cin, cout, h, wd, c->stream, c->conv_mode, c->amax, c->amax + AMAX_WORDS);
  LAUNCHCHK(c);
  return GR_OK;
}
// Sustained rate of the bare f16x3 inner loop (mfmaloop.hip) on this device: `launches` back-to-back launches (>= 0.3 s of them
// before the timed ones so that the clock settles), HIP events on the ctx stream.  shape 0 = v_mfma_f32_32x32x16_f16 (what the
// convolution kernels issue), 1 = v_mfma_f32_16x16x32_f16.  tflops_out: fp32-accurate TFLOP/s (f16 MFMA rate / 3 products), the
// figure comparable with the 833 TFLOP/s ceiling bench.py prices the f16x3 kernels against.
extern "C" int gr_bench_mfma_loop(gr_ctx* c, int shape, int launches, float* tflops_out) {
  if (!c || !tflops_out || shape < 0 || shape > 1 || launches < 1) return GR_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  int r = ensure_ws(c, mfma_loop_workspace_bytes()); if (r) return r;
  launch_mfma_loop_fill(c->ws, c->stream);
  const int iters = 200;
  for (int i = 0; i < 300; ++i) launch_mfma_loop(shape, c->ws, iters, c->stream);     // ~0.35 s of warm-up under load
  LAUNCHCHK(c);
  hipEvent_t e0, e1; HIPCHK(c, hipEventCreate(&e0)); HIPCHK(c, hipEventCreate(&e1));
  HIPCHK(c, hipEventRecord(e0, c->stream));
  for (int i = 0; i < launches; ++i) launch_mfma_loop(shape, c->ws, iters, c->stream);
  HIPCHK(c, hipEventRecord(e1, c->stream));
  HIPCHK(c, hipEventSynchronize(e1));
  float ms = 0; HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  *tflops_out = (float)(mfma_loop_flops(iters) * launches / (ms * 1e-3) / 1e12 / 3.0);
  return GR_OK;
}
extern "C" int gr_bench_conv3(gr_ctx* c, int which, int B, int cin, int cout, int h, int wd, int iters, float* avg_ms) {
  if (!c || iters < 1 || !avg_ms) return GR_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t nin = (size_t)B * cin * h * wd, nout = (size_t)B * cout * h * wd, nw = (size_t)cin * cout * 9;
  float *x = nullptr, *y = nullptr, *w = nullptr, *wt = nullptr, *gw = nullptr;
  HIPCHK(c, hipMalloc((void**)&x, sizeof(float) * nin)); HIPCHK(c, hipMalloc((void**)&y, sizeof(float) * nout));
  HIPCHK(c, hipMalloc((void**)&w, sizeof(float) * nw)); HIPCHK(c, hipMalloc((void**)&gw, sizeof(float) * nw));
  launch_fill_normal(x, (long)nin, 11, c->stream); launch_fill_normal(y, (long)nout, 12, c->stream); launch_fill_normal(w, (long)nw, 13, c->stream);
  if (GR_KNOB_SET("GR_BENCH_ZERO")) {   // DVFS diagnostic: all-zero operands draw less power (MI355X_MICROARCH.md, DVFS give-back item 1)
    (void)hipMemsetAsync(x, 0, sizeof(float) * nin, c->stream); (void)hipMemsetAsync(y, 0, sizeof(float) * nout, c->stream); (void)hipMemsetAsync(w, 0, sizeof(float) * nw, c->stream);
  }
  (void)hipMemsetAsync(gw, 0, sizeof(float) * nw, c->stream);
  int r = with_prepped(c, w, cin, cout, which == 1, &wt); if (r) return r;
  void* wsp = nullptr;
  const bool split = c->conv_mode >= 1 && which != 2 && (which == 0 ? cout > 4 : cin > 4);
  const int nterm = c->conv_mode == 2 ? 2 : 3;
  if (split) { r = conv_split_once(c, w, cin, cout, which == 1, &wsp); if (r) return r; }
  r = ensure_ws(c, conv_wgrad_workspace_bytes(B, cin, cout, h, wd, c->conv_mode)); if (r) return r;
  // f16x3 scales: taken once outside the timed loop (in a net the producing kernel tracks them), or per launch with GR_BENCH_ABSMAX
  const bool amax_each = GR_KNOB_SET("GR_BENCH_ABSMAX");
  if (c->conv_mode == 2) { launch_absmax(x, (long)nin, c->amax, c->stream); launch_absmax(y, (long)nout, c->amax + AMAX_WORDS, c->stream); }
  void* wup = nullptr;
  if (which == 3) {      // fused up-sampling layer: x is the source plane [B, cin, h/2, wd/2] (a quarter of the buffer), y the output
    if (c->conv_mode != 2 || !conv_up2_supported(cin, cout, h, wd)) return fail(c, GR_ERR_UNSUPPORTED, "up2 bench needs f16x3 mode and a supported shape");
    HIPCHK(c, hipMalloc(&wup, conv_weight_up2_bytes(cin, cout)));
    launch_conv_weight_up2_split(w, wup, cin, cout, c->stream, c->amax + 2 * AMAX_WORDS, true);
  }
  void* xp16 = nullptr; double* statp = nullptr;
  if (which == 4 || which == 5) {   // operand-ready forward (5: with the BatchNorm statistics epilogue): x converted once outside the loop
    if (c->conv_mode != 2 || !conv_p16_supported(B, cin, cout, h, wd)) return fail(c, GR_ERR_UNSUPPORTED, "p16 bench needs f16x3 mode and a supported shape");
    HIPCHK(c, hipMalloc(&xp16, sizeof(float) * nin));
    HIPCHK(c, hipMalloc((void**)&statp, sizeof(double) * 2 * cout * conv_stat_tiles_max(B, h, wd)));
    launch_to_p16(x, xp16, B, cin, h * wd, c->amax, c->stream);
    r = conv_split_once(c, w, cin, cout, false, &wsp); if (r) return r;
  }
  auto run = [&]() {
    if (which == 4 || which == 5) { int st = 0; launch_conv3x3_p16(xp16, wsp, nullptr, y, B, cin, cout, h, wd, c->stream, nullptr, c->amax, c->amax + 2 * AMAX_WORDS, nullptr, which == 5 ? statp : nullptr, which == 5 ? &st : nullptr); return; }
    if (which == 3) { launch_conv3x3_up2_f16x3(x, wup, nullptr, y, B, cin, cout, h, wd, c->stream, nullptr, c->amax, c->amax + 2 * AMAX_WORDS, nullptr); return; }
    if (amax_each && c->conv_mode == 2) { if (which != 1) launch_absmax(x, (long)nin, c->amax, c->stream); if (which != 0) launch_absmax(y, (long)nout, c->amax + AMAX_WORDS, c->stream); }
    if (split && which == 0) launch_conv3x3_split(x, wsp, nullptr, y, B, cin, cout, h, wd, false, c->stream, nullptr, nterm, c->amax, c->amax + 2 * AMAX_WORDS);
    else if (split && which == 1) launch_conv3x3_split(y, wsp, nullptr, x, B, cout, cin, h, wd, false, c->stream, nullptr, nterm, c->amax + AMAX_WORDS, c->amax + 2 * AMAX_WORDS);
    else if (which == 0) launch_conv3x3(x, wt, nullptr, y, B, cin, cout, h, wd, false, c->stream, w);
    else if (which == 1) launch_conv3x3(y, wt, nullptr, x, B, cout, cin, h, wd, false, c->stream);
    else launch_conv3x3_wgrad(x, y, gw, c->ws, B, cin, cout, h, wd, c->stream, c->conv_mode, c->amax, c->amax + AMAX_WORDS);
  };
  for (int i = 0; i < 3; ++i) run();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, c->stream);
  for (int i = 0; i < iters; ++i) run();
  (void)hipEventRecord(e1, c->stream);
  HIPCHK(c, hipEventSynchronize(e1));
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  *avg_ms = ms / iters;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(x); (void)hipFree(y); (void)hipFree(w); (void)hipFree(wt); (void)hipFree(gw); (void)hipFree(wsp); (void)hipFree(wup); (void)hipFree(xp16); (void)hipFree(statp);
  LAUNCHCHK(c);
  return GR_OK;
}

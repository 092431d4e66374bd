--[[ hipnn.lua — LuaJIT FFI binding of libganrev.so for a box that has Torch7 (th / luajit).

  UNTESTED: neither the build container nor the GPU box of this project has luajit or Torch7, so this file has
  never been executed.  It shows the reference-side binding a maintainer would add (see INTEGRATION.md); the Python
  package gan-reverser_amd/ganrev is the tested host mirror of the same ABI.

  Usage inside train_r.lua / apply_r.lua (unchanged otherwise):
      local hipnn = require 'hipnn'
      MODEL_R = hipnn.wrap(MODELS.create_R(IMG_DIMENSIONS, OPT.noiseDim, OPT.noiseMethod, OPT.fixer, false))
      MODEL_G = hipnn.wrap(MODEL_G)          -- after cudnn.convert(MODEL_G, nn)
  The wrapper keeps the nn.Module protocol the scripts use: :forward, :backward, :training, :evaluate,
  :getParameters (host FloatTensors, flat order identical to nn's), :float, :listModules.
]]
local ffi = require 'ffi'
require 'torch'
require 'nn'

ffi.cdef[[
typedef struct { int32_t kind, a, b, c; float p; int32_t flags; } gr_layer_desc;
typedef struct gr_ctx gr_ctx; typedef struct gr_net gr_net;
typedef struct { double lr, beta1, beta2, eps, l1, l2, clamp; } gr_hyper;
int gr_init(int device, gr_ctx** out);
int gr_shutdown(gr_ctx* ctx);
const char* gr_last_error(gr_ctx* ctx);
int gr_net_create(gr_ctx*, const gr_layer_desc*, int n, int c, int h, int w, gr_net** out);
int gr_net_destroy(gr_net*);
int gr_net_out_dim(gr_net*, int*, int*, int*);
int64_t gr_net_param_count(gr_net*);
int gr_net_get_params(gr_net*, float*);  int gr_net_set_params(gr_net*, const float*);
int gr_net_get_grads(gr_net*, float*);   int gr_net_zero_grads(gr_net*);
int gr_net_n_bn(gr_net*);
int gr_net_get_bn_running(gr_net*, int, float*, float*); int gr_net_set_bn_running(gr_net*, int, const float*, const float*);
int gr_net_set_training(gr_net*, int);   int gr_net_set_seed(gr_net*, uint64_t);
int gr_net_forward_host(gr_net*, const float* in_host, int batch, float* out_host);
int gr_net_backward_host(gr_net*, const float* in_host, const float* gout_host, int batch, float* gin_host);
int gr_mse_host(gr_ctx*, const float*, const float*, int64_t n, int64_t n_global, double* loss, float* grad);
int gr_bce_host(gr_ctx*, const float*, const float*, int64_t n, double* loss, float* grad);     /* nn.BCECriterion (train.lua:173) */
int gr_adam_step(gr_net*, const gr_hyper*, int t);
int gr_cosine_topk_host(gr_ctx*, const float* emb, int64_t n, int d, const int64_t* rows, int q, int k,
                        int64_t* idx, float* score, int accumulate_in_float);
int gr_cosine_similarity_host(gr_ctx*, const float* a, const float* b, int d, float* out);
int gr_l2_distance_rows_host(gr_ctx*, const float* a, const float* b, int64_t n, int64_t d, double* out);
int gr_kmeans_host(gr_ctx*, const float* x, int64_t n, int d, int k, int niter, float* centroids_inout, float* total_counts, int32_t* labels);
int gr_cosine_assign_host(gr_ctx*, const float* x, int64_t n, int d, const float* centroids, int k, int take_min, int32_t* labels, float* sims);
int gr_set_conv_mode(gr_ctx*, int mode);   /* 0 exact fp32 MFMA, 1 bf16x6, 2 f16x3 (default) */
int gr_set_tuning(gr_ctx*, const char* key, int value);                    /* "range_guard" 0|1 ... (include/ganrev.h) */
int gr_range_guard_stats(gr_ctx*, int64_t* scans, int64_t* fallbacks);     /* f16x3 passes the range guard sent to bf16x6 */
int gr_range_guard_scan_params(gr_net* net, int* tripped_out);                  /* synchronous scan of a net's weights / BatchNorm scales (device-pointer loops) */
int gr_search_stats(gr_ctx*, int64_t* reruns);
/* fast mode: everything resident on the GPU (INTEGRATION.md section 2) */
int gr_malloc(gr_ctx*, int64_t bytes, void** out_dev);  int gr_free(gr_ctx*, void* dev);
int gr_memcpy_h2d(gr_ctx*, void* dst_dev, const void* src_host, int64_t bytes);
int gr_memcpy_d2h(gr_ctx*, void* dst_host, const void* src_dev, int64_t bytes);
int gr_fill_normal_dev(gr_ctx*, float* dst_dev, int64_t n, uint64_t seed);
int gr_fill_uniform_dev(gr_ctx*, float* dst_dev, int64_t n, float lo, float hi, uint64_t seed);
int gr_net_forward_dev(gr_net*, const float* in_dev, int batch, float* out_dev);
float* gr_net_output_dev(gr_net*);
int gr_net_forward_batched_dev(gr_net*, const float* in_dev, int64_t rows, int batch, float* out_dev);   /* utils/nn_utils.lua:5-33 */
int gr_embed_dev(gr_net* gnet, gr_net* const* rnets, int n_rnets, const float* noise_dev, int64_t rows, int batch,
                 float* images_out_dev, float* const* attr_out_dev);                                       /* apply_r.lua:145-153 */
int gr_cosine_topk_dev(gr_ctx*, const float* emb_dev, int64_t n, int d, const int64_t* rows, int q, int k,
                       int64_t* idx, float* score, int accumulate_in_float);
int gr_synchronize(gr_ctx*);
int gr_adam_reset(gr_net*);
int gr_train_r_step(gr_net* gnet, gr_net* rnet, const float* noise_dev, int batch, int global_batch,
                    const gr_hyper* h, int t, double* loss_out);
]]
local C = ffi.load('ganrev')          -- libganrev.so on LD_LIBRARY_PATH
local hipnn = {}
local ctx

local function context()
   if not ctx then
      local p = ffi.new('gr_ctx*[1]')
      assert(C.gr_init((OPT and OPT.gpu and OPT.gpu >= 0) and OPT.gpu or 0, p) == 0, 'gr_init failed: no gfx950 GPU')
      ctx = p[0]
   end
   return ctx
end
local function check(rc, what) if rc ~= 0 then error(what .. ': ' .. ffi.string(C.gr_last_error(context()))) end end

-- kind numbers of include/ganrev.h
local K = {CONV3=1, BN=2, ELU=3, RELU=4, LEAKYRELU=5, SIGMOID=6, TANH=7, DROPOUT=8, SPATIAL_DROPOUT=9,
           MAXPOOL2=10, UPSAMPLE2=11, VIEW=12, LINEAR=13, FULLCONV3=14, CONVK=15, PRELU=16}

-- nn module -> descriptor rows, in nn.Sequential order (models.lua:104-143, 389-464)
local function describe(m, out, leaves)
   local t = torch.type(m)
   if t == 'nn.Sequential' then for _, c in ipairs(m.modules) do describe(c, out, leaves) end; return end
   if t == 'nn.Copy' then return end
   local d = {kind=0, a=0, b=0, c=0, p=0, flags=0}
   if t == 'nn.SpatialConvolution' or t == 'cudnn.SpatialConvolution' or t == 'nn.SpatialConvolutionMM' then
      assert(m.kW == m.kH and (m.kW == 3 or m.kW == 5) and m.dW == 1 and m.dH == 1 and m.padW == (m.kW - 1) / 2,
             'hipnn: only 3x3 s1 p1 and 5x5 s1 p2 (models.lua:275)')
      if m.kW == 3 then d.kind, d.a, d.b = K.CONV3, m.nInputPlane, m.nOutputPlane
      else d.kind, d.a, d.b, d.c = K.CONVK, m.nInputPlane, m.nOutputPlane, m.kW end
   elseif t == 'nn.SpatialFullConvolution' then d.kind, d.a, d.b = K.FULLCONV3, m.nInputPlane, m.nOutputPlane
   elseif t == 'nn.SpatialBatchNormalization' or t == 'nn.BatchNormalization' then d.kind, d.a = K.BN, m.running_mean:size(1)
   elseif t == 'nn.ELU' then d.kind = K.ELU
   elseif t == 'nn.ReLU' or t == 'cudnn.ReLU' then d.kind = K.RELU
   elseif t == 'nn.LeakyReLU' then d.kind, d.p = K.LEAKYRELU, m.negval
   elseif t == 'nn.PReLU' then       -- models.lua:276: nn.PReLU() = one shared slope, a parameter (weight[1]) in getParameters() order
      assert(m.nOutputPlane == 0, 'hipnn: only nn.PReLU() with one shared slope'); d.kind = K.PRELU
   elseif t == 'nn.Sigmoid' or t == 'cudnn.Sigmoid' then d.kind = K.SIGMOID
   elseif t == 'nn.Tanh' or t == 'cudnn.Tanh' then d.kind = K.TANH
   elseif t == 'nn.Dropout' then
      d.kind, d.p = K.DROPOUT, m.p
      d.flags = (m.v2 and 1 or 0) + ((rawget(m, 'evaluate') ~= nil) and 2 or 0)   -- models.lua:404 overrides evaluate
   elseif t == 'nn.SpatialDropout' then d.kind, d.p = K.SPATIAL_DROPOUT, m.p
   elseif t == 'nn.SpatialMaxPooling' then assert(m.kW == 2 and m.kH == 2 and m.dW == 2); d.kind = K.MAXPOOL2
   elseif t == 'nn.SpatialUpSamplingNearest' then assert(m.scale_factor == 2); d.kind = K.UPSAMPLE2
   elseif t == 'nn.View' then
      d.kind = K.VIEW; d.a = m.size[1]; d.b = m.size:size() > 1 and m.size[2] or 1; d.c = m.size:size() > 2 and m.size[3] or 1
   elseif t == 'nn.Linear' then d.kind, d.a, d.b = K.LINEAR, m.weight:size(2), m.weight:size(1)
   else error('hipnn: no gfx950 kernel for ' .. t) end
   out[#out + 1] = d; leaves[#leaves + 1] = m
end

local Wrapped = torch.class('hipnn.Sequential', 'nn.Module')

function Wrapped:__init(inner)
   nn.Module.__init(self)
   self.inner = inner; self.modules = inner.modules
   self.output = torch.FloatTensor(); self.gradInput = torch.FloatTensor()
   self.train = true
end

function Wrapped:compile(input)
   local c, h, w
   if input:dim() == 4 then c, h, w = input:size(2), input:size(3), input:size(4) else c, h, w = input:size(2), 1, 1 end
   if self.net and self.key == c .. 'x' .. h .. 'x' .. w then return end
   local rows, leaves = {}, {}
   describe(self.inner, rows, leaves)
   local arr = ffi.new('gr_layer_desc[?]', #rows)
   for i, d in ipairs(rows) do arr[i-1].kind, arr[i-1].a, arr[i-1].b, arr[i-1].c, arr[i-1].p, arr[i-1].flags = d.kind, d.a, d.b, d.c, d.p, d.flags end
   local p = ffi.new('gr_net*[1]')
   check(C.gr_net_create(context(), arr, #rows, c, h, w, p), 'gr_net_create')
   self.net, self.key, self.leaves = p[0], c .. 'x' .. h .. 'x' .. w, leaves
   self:pushParams()
end

function Wrapped:getParameters()          -- train_r.lua:122: same flat order as nn (weight, bias per module)
   if not self.flat then self.flat, self.flatGrad = self.inner:getParameters() end
   return self.flat, self.flatGrad
end

function Wrapped:pushParams()
   local flat = self:getParameters()
   check(C.gr_net_set_params(self.net, flat:data()), 'gr_net_set_params')
   local bi = 0
   for _, m in ipairs(self.leaves) do
      if m.running_mean then C.gr_net_set_bn_running(self.net, bi, m.running_mean:float():data(), m.running_var:float():data()); bi = bi + 1 end
   end
end

-- BatchNorm running statistics live on the device and move with every training-mode forward; the nn modules inside
-- `inner` are what torch.save writes (train_r.lua:228-235) and what :evaluate() forwards of a reloaded net use, so they
-- are refreshed from the device after each such forward (2 x nFeature floats per BN layer).
function Wrapped:pullRunningStats()
   local bi = 0
   for _, m in ipairs(self.leaves) do
      if m.running_mean then
         local rm, rv = m.running_mean:float(), m.running_var:float()
         check(C.gr_net_get_bn_running(self.net, bi, rm:data(), rv:data()), 'gr_net_get_bn_running')
         m.running_mean:copy(rm); m.running_var:copy(rv)
         bi = bi + 1
      end
   end
end

-- device parameters -> the host flat storage the nn modules view (needed after fast-mode steps, before torch.save)
function Wrapped:pullParams()
   check(C.gr_net_get_params(self.net, self.flat:data()), 'gr_net_get_params')
   self:pullRunningStats()
end

function Wrapped:updateOutput(input)
   input = input:contiguous()
   self:compile(input)
   check(C.gr_net_set_params(self.net, self.flat:data()), 'gr_net_set_params')   -- host storage is authoritative (compat mode)
   C.gr_net_set_training(self.net, self.train and 1 or 0)
   local B = input:size(1)
   local out = ffi.new('int[3]')
   -- output dims come from the library's shape inference
   C.gr_net_out_dim(self.net, out, out + 1, out + 2)
   if out[1] == 1 and out[2] == 1 then self.output:resize(B, out[0]) else self.output:resize(B, out[0], out[1], out[2]) end
   check(C.gr_net_forward_host(self.net, input:data(), B, self.output:data()), 'gr_net_forward_host')
   if self.train then self:pullRunningStats() end
   return self.output
end

function Wrapped:backward(input, gradOutput)   -- updateGradInput + accGradParameters(scale = 1), train_r.lua:151
   input = input:contiguous(); gradOutput = gradOutput:contiguous()
   self.gradInput:resizeAs(input)
   check(C.gr_net_zero_grads(self.net), 'gr_net_zero_grads')
   check(C.gr_net_backward_host(self.net, input:data(), gradOutput:data(), input:size(1), self.gradInput:data()), 'gr_net_backward_host')
   self.tmpGrad = self.tmpGrad or self.flatGrad:clone()
   check(C.gr_net_get_grads(self.net, self.tmpGrad:data()), 'gr_net_get_grads')
   self.flatGrad:add(self.tmpGrad)              -- accumulate, as nn does into GRAD_PARAMETERS_R
   return self.gradInput
end

function Wrapped:training() self.train = true; self.inner:training(); return self end
function Wrapped:evaluate() self.train = false; self.inner:evaluate(); return self end
function Wrapped:float() return self end
function Wrapped:listModules() return self.inner:listModules() end

function hipnn.wrap(model) return hipnn.Sequential(model) end

-- Fast mode: train_r.lua:138-170 as ONE call with noise, images, parameters and the Adam state resident on the GPU.
--   local step = hipnn.trainer(MODEL_G, MODEL_R, OPT.batchSize, {l1=OPT.R_L1, l2=OPT.R_L2, clamp=OPT.R_clamp}, noiseMethod)
--   for batchIdx = 1, n do local loss = step(batchIdx) ... end ;  MODEL_R:pullParams() before save()
-- Both models must have been compiled by one forward each (so that the nets exist and hold the current parameters).
function hipnn.trainer(G, R, batchSize, pen, noiseMethod)
   assert(G.net and R.net, 'hipnn.trainer: run one forward through G and R first')
   local nd = R.output:size(2)
   local pnoise = ffi.new('void*[1]')
   check(C.gr_malloc(context(), 4 * batchSize * nd, pnoise), 'gr_malloc')
   local noise = ffi.cast('float*', pnoise[0])
   local h = ffi.new('gr_hyper', {1e-3, 0.9, 0.999, 1e-8, pen.l1 or 0, pen.l2 or 1e-4, pen.clamp or 1})   -- optim.adam defaults, train_r.lua:22-24
   local loss = ffi.new('double[1]')
   check(C.gr_net_set_params(R.net, R.flat:data()), 'gr_net_set_params')
   check(C.gr_adam_reset(R.net), 'gr_adam_reset')                  -- OPTSTATE = {adam={R={}}}  train_r.lua:125
   local t = 0
   return function(seed)
      t = t + 1
      if noiseMethod == 'uniform' then check(C.gr_fill_uniform_dev(context(), noise, batchSize * nd, -1, 1, seed), 'fill')
      else check(C.gr_fill_normal_dev(context(), noise, batchSize * nd, seed), 'fill') end
      check(C.gr_train_r_step(G.net, R.net, noise, batchSize, batchSize, h, t, loss), 'gr_train_r_step')
      return loss[0]
   end
end

-- apply_r.lua:396-400 replacement
-- nn.BCECriterion (train.lua:173) over FloatTensors: returns loss, gradInput
function hipnn.bce(input, target)
  local x, t = input:contiguous(), target:contiguous()
  local g = torch.FloatTensor():resizeAs(x)
  local loss = ffi.new('double[1]')
  check(C.gr_bce_host(context(), x:data(), t:data(), x:nElement(), loss, g:data()), 'gr_bce_host')
  return loss[0], g
end

function hipnn.cosineSimilarity(v1, v2)
   local out = ffi.new('float[1]')
   check(C.gr_cosine_similarity_host(context(), v1:contiguous():data(), v2:contiguous():data(), v1:nElement(), out), 'cosine')
   return out[0]
end

-- apply_r.lua:266-282 replacement: returns a LongTensor [#needles x k] of 1-based row indices, best first
function hipnn.cosineTopK(attributes, needles, k)
   local N, d, Q = attributes:size(1), attributes:size(2), #needles
   local rows = ffi.new('int64_t[?]', Q); for i = 1, Q do rows[i-1] = needles[i] - 1 end
   local idx = torch.LongTensor(Q, k); local sc = torch.FloatTensor(Q, k)
   check(C.gr_cosine_topk_host(context(), attributes:contiguous():data(), N, d, rows, Q, k,
                               ffi.cast('int64_t*', idx:data()), sc:data(), 0), 'gr_cosine_topk_host')
   return idx:add(1), sc
end

-- apply_r.lua:145-153 replacement, device-resident: N noise rows are drawn on the GPU (utils/nn_utils.lua:39-51), pushed through G and
-- through every reverser net in `Rs` (MODEL_R, MODEL_R_FIXER) in chunks of batchSize; images and recovered noise never visit the host.
--   local emb = hipnn.embed(MODEL_G, {MODEL_R, MODEL_R_FIXER}, 10000, 512, OPT.noiseMethod, OPT.seed)
--   local idx = emb:cosineTopK(1, {100, 200, 300, 400, 500}, 100)      -- apply_r.lua:266-282 on table 1 (MODEL_R's attributes)
--   local attributes = emb:attributes(1)                              -- FloatTensor copy, when a script still wants one
-- All nets must have been compiled by one forward each and be in :evaluate() mode (apply_r.lua:64,94,103).
function hipnn.embed(G, Rs, N, batchSize, noiseMethod, seed)
   assert(G.net, 'hipnn.embed: run one forward through G first')
   local nd = G.inner:get(1).weight and G.inner:get(1).weight:size(2) or Rs[1].output:size(2)
   local function dev(bytes) local p = ffi.new('void*[1]'); check(C.gr_malloc(context(), bytes, p), 'gr_malloc'); return ffi.cast('float*', p[0]) end
   local noise = dev(4 * N * nd)
   if noiseMethod == 'uniform' then check(C.gr_fill_uniform_dev(context(), noise, N * nd, -1, 1, seed or 1), 'fill')
   else check(C.gr_fill_normal_dev(context(), noise, N * nd, seed or 1), 'fill') end
   local nets = ffi.new('gr_net*[?]', #Rs); local outs = ffi.new('float*[?]', #Rs); local dims = {}
   C.gr_net_set_training(G.net, 0)
   for i, R in ipairs(Rs) do
      assert(R.net, 'hipnn.embed: run one forward through every R first')
      C.gr_net_set_training(R.net, 0)
      local o = ffi.new('int[3]'); C.gr_net_out_dim(R.net, o, o + 1, o + 2)
      dims[i] = o[0] * o[1] * o[2]; nets[i-1] = R.net; outs[i-1] = dev(4 * N * dims[i])
   end
   -- the device-pointer calls inside gr_embed_dev are not range-guarded (include/ganrev.h): scan every net's weights and BatchNorm scales once,
   -- as ganrev.apply_r.embed_dev does - a hostile spread keeps the context on bf16x6 for the whole pipeline.  (gr_net_set_training(net, 0) above
   -- leaves R_FIXER's always-on first Dropout active: GR_DROPOUT_ALWAYS_ON layers ignore the mode, models.lua:399-406.)
   local tripped = ffi.new('int[1]')
   check(C.gr_range_guard_scan_params(G.net, tripped), 'gr_range_guard_scan_params')
   for i = 1, #Rs do check(C.gr_range_guard_scan_params(Rs[i].net, tripped), 'gr_range_guard_scan_params') end
   check(C.gr_embed_dev(G.net, nets, #Rs, noise, N, batchSize, nil, outs), 'gr_embed_dev')
   local emb = {N = N, noise = noise, tables = outs, dims = dims}
   function emb:attributes(i)
      local t = torch.FloatTensor(self.N, self.dims[i])
      check(C.gr_memcpy_d2h(context(), t:data(), self.tables[i-1], 4 * self.N * self.dims[i]), 'gr_memcpy_d2h'); return t
   end
   function emb:cosineTopK(i, needles, k)
      local Q = #needles
      local rows = ffi.new('int64_t[?]', Q); for j = 1, Q do rows[j-1] = needles[j] - 1 end
      local idx = torch.LongTensor(Q, k); local sc = torch.FloatTensor(Q, k)
      check(C.gr_cosine_topk_dev(context(), self.tables[i-1], self.N, self.dims[i], rows, Q, k, ffi.cast('int64_t*', idx:data()), sc:data(), 0), 'gr_cosine_topk_dev')
      return idx:add(1), sc
   end
   function emb:free() C.gr_free(context(), self.noise); for i = 1, #self.dims do C.gr_free(context(), self.tables[i-1]) end end
   return emb
end

-- apply_r.lua:198 replacement: unsup.kmeans(attributes, nbClusters, nbIterations) -> centroids, totalcounts
-- (initial centroids drawn here exactly as unsup does: normal() rows divided by their norm, from Torch's RNG)
function hipnn.kmeans(attributes, k, niter)
   local N, d = attributes:size(1), attributes:size(2)
   local centroids = torch.FloatTensor(k, d):normal()
   for i = 1, k do centroids[i]:div(centroids[i]:norm()) end
   local counts = torch.FloatTensor(k)
   check(C.gr_kmeans_host(context(), attributes:contiguous():data(), N, d, k, niter, centroids:data(), counts:data(), nil), 'gr_kmeans_host')
   return centroids, counts
end

-- apply_r.lua:205-217 replacement: per row the (1-based) cluster with the MINIMUM cosine similarity, as the reference's loop
-- keeps it, and that similarity
function hipnn.assignClusters(attributes, centroids)
   local N, d, k = attributes:size(1), attributes:size(2), centroids:size(1)
   local labels = torch.IntTensor(N); local sims = torch.FloatTensor(N)
   check(C.gr_cosine_assign_host(context(), attributes:contiguous():data(), N, d, centroids:contiguous():data(), k, 1,
                                 labels:data(), sims:data()), 'gr_cosine_assign_host')
   return labels:add(1), sims
end

return hipnn

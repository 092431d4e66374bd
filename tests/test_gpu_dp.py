"""-m gpu: the data-parallel semantics of the HIP path itself, on ONE GPU.

tests/test_dp_gloo.py checks ganrev.parallel's sharding arithmetic with the oracle as the compute.  Here the compute is
libganrev.so: the global batch of 2B is split into two shards, each shard runs gr_train_r_step / the decomposed ABI calls with
global_batch = 2B (the MSE normaliser), its own slice of the dropout noise and its own (per-shard) BatchNorm statistics; the two
flat gradients are SUMMED before penalty / clamp / Adam (train_r.lua:147-165 order) and everything is compared with the oracle
run ONCE on the 2B batch with BatchNorm in two groups (go_net_set_bn_groups(2)).

  test_dp_two_shards_in_process_vs_grouped_oracle   both shards in this process, gradients summed on the host
  test_dp_two_processes_share_the_gpu_vs_grouped_oracle   two rank PROCESSES (tests/dp_rank_worker.py, started by
      tests/conftest.py before this process touches the GPU), gradients reduced through torch.distributed gloo by
      ganrev.parallel.DeviceTrainer.step_decomposed - bench.py's GANREV_ALL_RANKS_ON_DEVICE0 control flow, now asserted.
RCCL itself needs one device per rank: its N > 1 execution is the driver's multi-GPU run (SCALE_rNN.json)."""
import json
import os

import numpy as np
import pytest

import dp_common as D
from helpers import TOL, assert_close, assert_grads_close, maxdiff, pool_layers

pytestmark = pytest.mark.gpu


def _layer_of(R, oR):
    return lambda m: oR.layer_index[id(m)]


def _compile(G, R, dims, nd):
    from ganrev import synth
    G.evaluate(); G.forward(synth.normal((2, nd), 1))
    R.training(); R.forward(synth.uniform((2,) + dims, 2, 0, 1)); R.push_params()
    R._pending_masks = {}


def _pooled_convs(R, oR):
    from helpers import _stage_before_pool
    return {li: oR.layer_index[id(_stage_before_pool(R, m)[0])] for m, li, _ in pool_layers(R, oR)}


def _check_against_oracle(oracle, R, ref, loss, raw_sum, grads, theta, m2, v2, flips_note=""):
    assert abs(loss - ref["loss"]) <= 1e-5 * max(1.0, abs(ref["loss"])), f"global loss {loss} vs oracle {ref['loss']}"
    # the SUM of the shards' gradients, before the non-linear part, against the grouped oracle's raw gradient
    assert_grads_close(R, raw_sum, ref["raw"], 1e-4, 1e-3, f"summed raw gradient {flips_note}")
    # after L2 + clamp (fevalR's return value) and after Adam
    assert_grads_close(R, grads, ref["grads"], 1e-4, 1e-3, f"penalised + clamped gradient {flips_note}")
    well = np.abs(ref["grads"]) > 1e-4                 # entries whose Adam step is well-conditioned (see test_train_r_steps_vs_oracle)
    assert_close(theta[well], ref["theta"][well], TOL, "parameters after Adam")
    assert maxdiff(theta, ref["theta"]) <= 2.1e-3
    assert_close(m2, ref["m"], 1e-5, "adam m"); assert_close(v2, ref["v"], 1e-5, "adam v")


@pytest.mark.parametrize("world,name,dims,nd,B", [pytest.param(*c, id=c[1]) for c in D.WORLD_CASES])
def test_dp_two_shards_in_process_vs_grouped_oracle(ctx, oracle, conv_mode, world, name, dims, nd, B):
    """`world` shards (2, and 8 = cfg4's rank count) of one global batch through the HIP path, one after the other in this process."""
    import ganrev._lib as L
    G, R = D.make_models(dims, nd)
    oG, oR = oracle.from_model(G, (nd, 1, 1)), oracle.from_model(R, dims)
    _compile(G, R, dims, nd)
    gnet, rnet = G._net, R._net
    theta0 = oR.params.copy()
    noise, masks = D.global_inputs(R, _layer_of(R, oR), oR.mask_size, dims, nd, B, world=world)
    GB = B * world
    hyper, free = L.Hyper(), L.Hyper(l1=0.0, l2=0.0, clamp=0.0)          # `free`: no penalty, no clamp -> the step leaves the raw gradient
    zeros = np.zeros_like(theta0)
    pooled = _pooled_convs(R, oR)
    dn = ctx.malloc(4 * B * nd)
    dfdo = ctx.malloc(4 * B * nd)
    dloss = ctx.malloc(64)
    raw, losses, images, preds = [], [], [], []
    idx = {li: [] for li in pooled}; ys = {cl: [] for cl in pooled.values()}
    def shard(a, r):
        return D.shard(a, r, world)
    for r in range(world):
        rnet.set_params(theta0); rnet.set_adam_state(zeros, zeros)
        ctx.upload(shard(noise, r), dn)
        for li, k in masks.items():
            rnet.set_mask(li, shard(k.reshape(GB, -1), r).ravel())
        # (a0) the fused entry point as it ships (round 5: R's head - fc1's pipeline, fc2, the criterion and their backward - in ONE launch,
        #      head_fwd_bwd_kernel): same operations per value as the stage-by-stage path, sums in another order -> compared at the gradient bar, not bit for bit
        loss_head = L.train_r_step(gnet, rnet, dn, B, GB, free, D.T_STEP)
        raw_head = rnet.get_grads()
        rnet.set_params(theta0); rnet.set_adam_state(zeros, zeros)
        for li, k in masks.items():
            rnet.set_mask(li, shard(k.reshape(GB, -1), r).ravel())
        # (a) the fused entry point with the GLOBAL normaliser; no communicator: its Adam sees the local gradient only, so the
        #     penalty-free hyper-parameters keep the raw local gradient readable afterwards.  Stage by stage (fused_head 0): (b) below must reproduce it bit for bit
        ctx.set_tuning("fused_head", 0)
        try:
            losses.append(L.train_r_step(gnet, rnet, dn, B, GB, free, D.T_STEP))
        finally:
            ctx.set_tuning("fused_head", 1)
        raw.append(rnet.get_grads())
        assert abs(loss_head - losses[-1]) <= 1e-6 * max(1.0, abs(losses[-1])), (loss_head, losses[-1])
        assert_grads_close(R, raw_head, raw[-1], 1e-4, 1e-3, f"shard {r}: head kernel vs stage-by-stage step")
        images.append(ctx.download(gnet.lib.gr_net_output_dev(gnet.h), (B,) + dims))
        preds.append(ctx.download(rnet.lib.gr_net_output_dev(rnet.h), (B, nd)))
        for m, li, (c, h, w) in pool_layers(R, oR):
            idx[li].append(rnet.pool_index(li, B * c * (h // 2) * (w // 2)))
            ys[pooled[li]].append(rnet.layer_output(pooled[li], (B * c * h * w,)))
        # (b) the same shard through the individual ABI calls (gr_net_forward_dev, gr_mse_dev with n_global, gr_net_backward_dev):
        #     bit-identical raw gradient and partial loss
        rnet.set_params(theta0)
        for li, k in masks.items():
            rnet.set_mask(li, shard(k.reshape(GB, -1), r).ravel())
        gnet.set_training(False); img_dev = gnet.forward_dev(dn, B)
        rnet.set_training(True); rnet.zero_grads()
        pred_dev = rnet.forward_dev(img_dev, B)
        ctx.check(ctx.lib.gr_mse_dev(ctx.h, L._ptr(pred_dev), L._ptr(dn), B * nd, GB * nd, L._ptr(dloss), L._ptr(dfdo)), "gr_mse_dev")
        rnet.backward_dev(img_dev, dfdo, B)
        assert np.array_equal(rnet.get_grads(), raw[-1]), f"shard {r}: decomposed ABI calls and gr_train_r_step disagree"
        assert float(ctx.download(dloss, (1,), np.float64)[0]) == losses[-1]
    for p in (dn, dfdo, dloss):
        ctx.free(p)
    # SUM on the host (what ncclAllReduce(sum) does between the ranks), then the non-linear part on the reduced gradient
    raw_sum = raw[0].copy()
    for g in raw[1:]:                                                    # rank order, fp32: what a ring / tree all-reduce may re-associate (inside the gradient bar)
        raw_sum += g
    rnet.set_params(theta0); rnet.set_adam_state(zeros, zeros)
    rnet.set_grads(raw_sum)
    rnet.adam_step(hyper, D.T_STEP)                                      # gr_adam_step = penalty + clamp + Adam (train_r.lua:153-170)
    grads, theta = rnet.get_grads(), rnet.get_params()
    m2, v2 = rnet.adam_state()

    dev_index = {li: np.concatenate(v) for li, v in idx.items()}
    dev_y = {cl: np.concatenate(v) for cl, v in ys.items()}
    rep = {}
    ref = D.oracle_grouped_step(oracle, oG, oR, noise, masks, theta0, oracle.GoHyper(), dev_index, dev_y, R, max_flips=16, report=rep, groups=world)
    assert_close(np.concatenate(images), ref["images"], TOL, "G images of all shards")
    assert_close(np.concatenate(preds), ref["preds"], TOL, "recovered noise of all shards (per-shard BatchNorm statistics)")
    _check_against_oracle(oracle, R, ref, float(np.sum(losses)), raw_sum, grads, theta, m2, v2, f"(world {world}, argmax flips {rep.get('flips')})")
    # and the control: ONE batch-statistics group over the 2B batch is a different computation (per-rank BatchNorm is what DP means here)
    oR.set_bn_groups(1)
    oR.params[...] = theta0
    for li, k in masks.items():
        oR.set_mask(li, k)
    single = oR.forward(ref["images"])
    oR.set_bn_groups(world)
    assert maxdiff(single, ref["preds"]) > 10 * TOL, "grouped and ungrouped BatchNorm agree: the case does not separate them"


def test_dp_two_processes_share_the_gpu_vs_grouped_oracle(oracle, dp_children):
    """Process-level variant: the two ranks are separate processes (one gr_ctx each, both on GPU 0), started before this
    process initialised the GPU; their gradients meet in a gloo all-reduce (RCCL refuses two ranks on one device)."""
    outdir = dp_children()                      # waits for the rank processes; raises with their logs when one failed
    for name, dims, nd, B in D.CASES:
        G, R = D.make_models(dims, nd)
        oG, oR = oracle.from_model(G, (nd, 1, 1)), oracle.from_model(R, dims)
        theta0 = oR.params.copy()
        noise, masks = D.global_inputs(R, _layer_of(R, oR), oR.mask_size, dims, nd, B)
        pooled = _pooled_convs(R, oR)
        for mode in D.MODES:
            ranks = [np.load(os.path.join(outdir, f"{name}_{mode}_rank{r}.npz")) for r in range(D.WORLD)]
            for k in ("grads", "theta", "raw_sum", "m", "v"):
                assert np.array_equal(ranks[0][k], ranks[1][k]), f"{name} {mode}: replicas differ in {k}"
            assert float(ranks[0]["loss"]) == float(ranks[1]["loss"])
            dev_index = {li: np.concatenate([rk[f"pool{li}"] for rk in ranks]) for li in pooled}
            dev_y = {cl: np.concatenate([rk[f"y{cl}"] for rk in ranks]) for cl in pooled.values()}
            rep = {}
            ref = D.oracle_grouped_step(oracle, oG, oR, noise, masks, theta0, oracle.GoHyper(), dev_index, dev_y, R, max_flips=16, report=rep, mode=mode)
            assert_close(np.concatenate([rk["preds"] for rk in ranks]), ref["preds"], TOL, f"{name} {mode}: recovered noise")
            rk = ranks[0]
            _check_against_oracle(oracle, R, ref, float(rk["loss"]), rk["raw_sum"], rk["grads"], rk["theta"], rk["m"], rk["v"],
                                  f"[{name} {mode}, argmax flips {rep.get('flips')}]")
    meta = json.load(open(os.path.join(outdir, "meta.json")))
    assert meta["world"] == D.WORLD and sorted(meta["ranks"]) == list(range(D.WORLD))

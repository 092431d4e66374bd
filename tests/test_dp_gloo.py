"""not-gpu: the N>1 path's host logic with torch.distributed gloo, world_size 2 (spawned here on CPU).

Each rank runs ganrev.parallel.train_r_step_decomposed over a TorchDistCommunicator with the ORACLE as the compute
(the HIP library needs a GPU); the result must equal a single-process oracle step on the global batch whose BatchNorm
is evaluated in 2 groups — i.e. shard bounds, the global MSE normaliser, SUM-before-clamp ordering and replica
consistency are what is under test."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DIMS, ND, GB, WORLD = (1, 8, 8), 6, 8, 2


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _setup(seed=5):
    for p in (os.path.join(ROOT, "gan-reverser_amd"), ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from ganrev import models, synth
    from oracle import oracle
    G = models.create_G(DIMS, ND); synth.init_params(G, seed)
    R = models.create_R(DIMS, ND); synth.init_params(R, seed + 1)
    return G, R, oracle.from_model(G, (ND, 1, 1)), oracle.from_model(R, DIMS), oracle, synth


def _masks(R, oR, synth, B, seed):
    out = {}
    for m in R.leaves():
        if m.typename in ("nn.Dropout", "nn.SpatialDropout"):
            li = oR.layer_index[id(m)]
            out[li] = synth.bernoulli_keep((oR.mask_size(li, B),), seed * 131 + li, m.p)
    return out


def _worker(rank, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    G, R, oG, oR, oracle, synth = _setup()
    from ganrev.parallel import TorchDistCommunicator, shard_bounds, train_r_step_decomposed
    comm = TorchDistCommunicator()
    m = np.zeros(oR.n_params, np.float32); v = np.zeros_like(m)
    hyper = oracle.GoHyper()
    losses = []
    for t in (1, 2):
        noise = synth.normal((GB, ND), 50 + t)
        full = _masks(R, oR, synth, GB, t)
        lo, hi = shard_bounds(GB, WORLD, rank)
        for li, k in full.items():
            per = k.size // GB
            oR.set_mask(li, k[lo * per:hi * per])

        def g_forward(z):
            oG.set_training(False)
            return oG.forward(z)

        def r_fwd_bwd(images, z, n_global):
            oR.set_training(True); oR.zero_grads()
            pred = oR.forward(images)
            loss, dfdo = oracle.mse(pred, z, n_global)
            oR.backward(images, dfdo, want_gin=False)
            return loss, oR.grads.copy()

        def update(grad, t_):
            g = grad.copy()
            oracle.penalty_clamp_adam(oR.params, g, m, v, hyper, t_)

        losses.append(train_r_step_decomposed(g_forward, r_fwd_bwd, update, comm, noise, t))
    q.put((rank, losses, oR.params.copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_equals_grouped_single_process_oracle():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(WORLD)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single process, global batch, BatchNorm in WORLD groups
    G, R, oG, oR, oracle, synth = _setup()
    oR.set_bn_groups(WORLD)
    m = np.zeros(oR.n_params, np.float32); v = np.zeros_like(m)
    ref_losses = []
    for t in (1, 2):
        noise = synth.normal((GB, ND), 50 + t)
        for li, k in _masks(R, oR, synth, GB, t).items():
            oR.set_mask(li, k)
        loss, _ = oracle.train_r_step(oG, oR, noise, oracle.GoHyper(), m, v, t)
        ref_losses.append(loss)
    (r0, l0, p0), (r1, l1, p1) = res
    assert np.array_equal(p0, p1), "replicas diverged"
    assert np.allclose(l0, ref_losses, rtol=1e-6, atol=1e-7) and np.allclose(l1, ref_losses, rtol=1e-6, atol=1e-7)
    # gradient sums are re-associated across ranks -> rounding-level differences, amplified by Adam only where |g| ~ 0
    well = np.abs(oR.grads) > 1e-4
    assert np.max(np.abs(p0[well] - oR.params[well])) < 1e-5
    assert np.max(np.abs(p0 - oR.params)) < 4.1e-3


def _search_worker(rank, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    for p in (os.path.join(ROOT, "gan-reverser_amd"), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    from ganrev import synth
    from ganrev.parallel import TorchDistCommunicator, sharded_cosine_topk
    from oracle import oracle
    N, d, k = 4001, 32, 50
    emb = synth.normal((N, d), 77)
    emb[1234] = emb[99]                                     # a duplicate row: a score tie across the two shards
    lo = 0 if rank == 0 else 2000
    hi = 2000 if rank == 0 else N
    idx, sc = sharded_cosine_topk(oracle.cosine_topk, emb[lo:hi], lo, [99, 199, 2999, 4000], k, TorchDistCommunicator())
    q.put((rank, idx, sc))
    dist.barrier(); dist.destroy_process_group()


def test_sharded_search_two_ranks_equals_unsharded():
    """corpus rows split over 2 ranks (unequal shards), local top-k, candidate all-gather, merge: the lists must be
    bit-identical to the unsharded search (oracle as the local search: the HIP library needs a GPU), ties included."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_search_worker, args=(r, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(WORLD)]
    for p in procs:
        p.join(60)
    for p in (os.path.join(ROOT, "gan-reverser_amd"), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    from ganrev import synth
    from oracle import oracle
    emb = synth.normal((4001, 32), 77); emb[1234] = emb[99]
    ridx, rsc = oracle.cosine_topk(emb, [99, 199, 2999, 4000], 50)
    for _, idx, sc in res:
        assert np.array_equal(idx, ridx) and np.array_equal(sc, rsc)

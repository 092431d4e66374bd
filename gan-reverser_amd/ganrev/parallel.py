"""Data parallelism for the train_r step (NEW capability required by BASELINE.json north_star; the reference is
single-GPU: train_r.lua:34,58-62).

One process per GPU.  Rank r takes noise rows [r*B/P, (r+1)*B/P) of the global batch, runs G forward and R
forward/backward on its shard with replicated weights, with the MSE normalised by the GLOBAL element count so that
a plain SUM all-reduce of R's flat gradient reproduces train_r.lua:147-151 on the global batch; the L1/L2 penalty,
the clamp (both non-linear in g, train_r.lua:154-165) and Adam run AFTER the all-reduce, identically on every rank.
BatchNorm uses per-rank batch statistics (the oracle models this as BN evaluated in P groups).

The production path is the fused C entry point gr_train_r_step (all-reduce = RCCL over xGMI inside libganrev.so);
`train_r_step_decomposed` below spells the same phases out one by one over a pluggable communicator so that the
sharding / normaliser / reduction-order logic is testable with torch.distributed's gloo backend on CPU.
"""
import numpy as np


def shard_bounds(global_batch, world, rank):
    """Rows [lo, hi) of the global batch owned by `rank` (equal shards; the reference's batch sizes are multiples of 8)."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by {world} ranks")
    per = global_batch // world
    return rank * per, (rank + 1) * per


class LocalCommunicator:
    """world_size 1."""
    world, rank = 1, 0

    def allreduce_sum(self, arr):
        return arr

    def allreduce_scalar(self, x):
        return x

    def allgather(self, arr):
        return [arr]


class TorchDistCommunicator:
    """torch.distributed (gloo on CPU, or any initialised backend) all-reduce of host arrays: test / control plane."""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        self.world, self.rank = dist.get_world_size(), dist.get_rank()

    def allreduce_sum(self, arr):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.numpy()

    def allreduce_scalar(self, x):
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def allgather(self, arr):
        """Equal-shaped host arrays from every rank, in rank order."""
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr))
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [o.numpy() for o in out]


class RcclCommunicator:
    """The native path: the RCCL communicator owned by the gr_ctx (gr_comm_init); bootstrap id shipped over an
    already-initialised torch.distributed group (any backend)."""

    def __init__(self, ctx, world=None, rank=None):
        import torch.distributed as dist
        self.ctx = ctx
        self.world = world if world is not None else dist.get_world_size()
        self.rank = rank if rank is not None else dist.get_rank()
        uid = [ctx.comm_unique_id() if self.rank == 0 else None]
        if self.world > 1:
            dist.broadcast_object_list(uid, src=0)
        ctx.comm_init(uid[0], self.world, self.rank)

    def close(self):
        self.ctx.comm_destroy()

    # The collectives of the sharded search (gather_needles / sharded_cosine_topk below) on the RCCL communicator: host arrays
    # are staged through device buffers (they are Q x d and Q x k values), the exchange itself is ncclAllReduce / ncclAllGather
    # over xGMI (SURVEY.md 8e names the all-gather of Q * k * 12 bytes per rank).
    def _staged(self, arr):
        arr = np.ascontiguousarray(arr)
        return arr, self.ctx.upload(arr)

    def allreduce_sum(self, arr):
        arr, d = self._staged(np.asarray(arr, np.float32))
        try:
            self.ctx.allreduce(d, arr.size)
            return self.ctx.download(d, arr.shape, np.float32)
        finally:
            self.ctx.free(d)

    def allreduce_scalar(self, x):
        return float(self.allreduce_sum(np.array([x], np.float32))[0])       # (control values only: fp32 on the wire)

    def allgather(self, arr):
        arr, d = self._staged(arr)
        out = self.ctx.malloc(arr.nbytes * self.world)
        try:
            self.ctx.allgather(d, out, arr.nbytes)
            full = self.ctx.download(out, (self.world,) + arr.shape, arr.dtype)
            return [full[r] for r in range(self.world)]
        finally:
            self.ctx.free(d); self.ctx.free(out)


def train_r_step_decomposed(g_forward, r_forward_backward, penalty_clamp_adam, comm, noise_global, t):
    """One iteration of train_r.lua:138-170 over `comm`, phase by phase.

    g_forward(noise_shard) -> images
    r_forward_backward(images, noise_shard, n_global_elems) -> (sum_sq_err / n_global_elems, flat_grad)   [local shard]
    penalty_clamp_adam(flat_grad_reduced, t) -> None          [updates the replica's parameters in place]
    Returns the global un-penalised MSE (train_r.lua:172-176 prints CRITERION_R.output).
    """
    B = noise_global.shape[0]
    lo, hi = shard_bounds(B, comm.world, comm.rank)
    noise = noise_global[lo:hi]
    images = g_forward(noise)                                              # train_r.lua:139
    loss_local, grad = r_forward_backward(images, noise, noise_global.size)  # :146-151, normalised by B*nd (global)
    grad = comm.allreduce_sum(grad)                                        # SUM over ranks, before the non-linear part
    loss = comm.allreduce_scalar(loss_local)
    penalty_clamp_adam(grad, t)                                            # :153-170 on the reduced gradient
    return loss


class DeviceTrainer:
    """train_r.lua:131-170 on one GPU of a data-parallel job, everything resident in HBM.

    step()            -> the production path: one gr_train_r_step call (RCCL buckets overlapped with backward inside).
    step_decomposed() -> the same iteration spelled out through the individual ABI calls, with a pluggable gradient
                         reduction; used to check the fused path and to run the N>1 control flow where RCCL cannot
                         (several ranks on one GPU).
    """

    def __init__(self, ctx, gnet, rnet, hyper, per_gpu_batch, world=1, rank=0, noise_method="normal"):
        from . import _lib as L
        self.L, self.ctx, self.gnet, self.rnet, self.hyper = L, ctx, gnet, rnet, hyper
        self.B, self.world, self.rank, self.t = int(per_gpu_batch), int(world), int(rank), 0
        if noise_method not in ("normal", "uniform"):
            raise ValueError(f"Unknown noise method '{noise_method}'")        # utils/nn_utils.lua:48
        self.noise_method = noise_method
        self.nd = int(np.prod(rnet.out_dims))
        self.noise = ctx.malloc(4 * self.B * self.nd)
        self.dfdo = ctx.malloc(4 * self.B * self.nd)
        self.loss_dev = ctx.malloc(64)

    def close(self):
        for p in (self.noise, self.dfdo, self.loss_dev):
            if p:
                self.ctx.free(p)
        self.noise = self.dfdo = self.loss_dev = None

    def new_noise(self, seed):
        # createNoiseInputs on device (utils/nn_utils.lua:39-51): normal(0, 1) or uniform(-1, 1) - with the uniform method R ends
        # in a Tanh (models.lua:452-454) and can only reach targets in (-1, 1)
        if self.noise_method == "uniform":
            self.ctx.fill_uniform(self.noise, self.B * self.nd, seed, -1.0, 1.0)
        else:
            self.ctx.fill_normal(self.noise, self.B * self.nd, seed)

    def step(self, want_loss=False):
        self.t += 1
        return self.L.train_r_step(self.gnet, self.rnet, self.noise, self.B, self.B * self.world, self.hyper, self.t, want_loss=want_loss)

    def step_decomposed(self, reduce_grads=None, reduce_scalar=None):
        L, g, r, B = self.L, self.gnet, self.rnet, self.B
        self.t += 1
        g.set_training(False)
        images = g.forward_dev(self.noise, B)                                           # train_r.lua:139
        r.set_training(True)
        r.zero_grads()                                                                  # :143
        preds = r.forward_dev(images, B)                                                # :146
        n = B * self.nd
        self.ctx.check(self.ctx.lib.gr_mse_dev(self.ctx.h, L._ptr(preds), L._ptr(self.noise), n, n * self.world,
                                               L._ptr(self.loss_dev), L._ptr(self.dfdo)), "gr_mse_dev")   # :147,150
        r.backward_dev(images, self.dfdo, B)                                            # :151
        if reduce_grads is not None:
            reduce_grads(r)                                                             # SUM over ranks before the non-linear part
        r.adam_step(self.hyper, self.t)                                                 # :153-170
        loss = float(self.ctx.download(self.loss_dev, (1,), np.float64)[0])
        return reduce_scalar(loss) if reduce_scalar is not None else loss


def host_allreduce_grads(dist):
    """Gradient reduction through host memory over an initialised torch.distributed group (gloo): the control-flow stand-in
    for RCCL when several ranks share one GPU (RCCL refuses duplicate devices)."""
    import torch

    def fn(rnet):
        g = rnet.get_grads()
        t = torch.from_numpy(g)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        rnet.set_grads(t.numpy())
    return fn


# ---------------------------------------------------------------------------------------------------------------------
# Sharded similarity search: the corpus rows are independent and top-k merges associatively, so the N rows are
# split over the ranks, every rank scores ITS rows against the needles and keeps its k best, and the P*k candidates are
# merged with the order the single-GPU search defines (score descending, global row index ascending).  A (needle, row)
# score does not depend on which rank computes it, so the merged lists are bit-identical to an unsharded search.
# The only exchanges are the needle vectors (Q x d floats) and the candidates (Q x k x 12 bytes per rank).

def gather_needles(emb_shard, row_offset, needle_rows, comm):
    """The needle vectors emb[needle_rows] on every rank: each needle is contributed by the rank that owns its row, zeros
    elsewhere, and summed (x + 0 is exact)."""
    emb_shard = np.asarray(emb_shard, np.float32)
    out = np.zeros((len(needle_rows), emb_shard.shape[1]), np.float32)
    for i, r in enumerate(needle_rows):
        if row_offset <= r < row_offset + len(emb_shard):
            out[i] = emb_shard[r - row_offset]
    return comm.allreduce_sum(out)


def merge_candidates(idx_lists, score_lists, k):
    """Candidates [P][Q, k'] -> the k best per needle, ordered by (score descending, row index ascending)."""
    idx = np.concatenate(idx_lists, axis=1)
    sc = np.concatenate(score_lists, axis=1)
    Q = idx.shape[0]
    out_i = np.empty((Q, min(k, idx.shape[1])), np.int64)
    out_s = np.empty(out_i.shape, np.float32)
    for q in range(Q):
        order = np.lexsort((idx[q], -sc[q].astype(np.float64)))       # primary: score desc; ties: index asc
        order = order[sc[q][order] > -np.inf][:out_i.shape[1]]
        out_i[q, :len(order)] = idx[q][order]; out_s[q, :len(order)] = sc[q][order]
        out_i[q, len(order):] = -1; out_s[q, len(order):] = -np.inf
    return out_i, out_s


def sharded_cosine_topk(local_topk, emb_shard, row_offset, needle_rows, k, comm):
    """apply_r.lua:266-282 over a corpus whose rows [row_offset, row_offset + len(emb_shard)) live on this rank.
    local_topk(emb, query_rows, k) -> (idx, scores) is the single-device search (Context.cosine_topk).  The needle vectors
    are put in front of the shard so that they are rows of the searched matrix; their own hits are dropped afterwards."""
    emb_shard = np.asarray(emb_shard, np.float32)
    needles = gather_needles(emb_shard, row_offset, needle_rows, comm)
    Q = len(needle_rows)
    aug = np.concatenate([needles, emb_shard])
    kk = min(k + Q, len(aug))
    idx, sc = local_topk(aug, np.arange(Q, dtype=np.int64), kk)
    cand_i = np.full((Q, k), -1, np.int64); cand_s = np.full((Q, k), -np.inf, np.float32)
    for q in range(Q):
        keep = idx[q] >= Q                                          # rows of the shard proper
        ii, ss = idx[q][keep][:k], sc[q][keep][:k]
        cand_i[q, :len(ii)] = ii - Q + row_offset; cand_s[q, :len(ii)] = ss
    gi, gs = comm.allgather(cand_i), comm.allgather(cand_s)
    return merge_candidates(gi, gs, k)

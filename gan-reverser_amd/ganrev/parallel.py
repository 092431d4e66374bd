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

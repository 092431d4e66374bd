"""One rank of the two-process data-parallel parity test (tests/test_gpu_dp.py).  Started by tests/conftest.py as a CHILD
process before the pytest process has touched the GPU (a process that has initialised HIP must not exec another program on this
pool).  Both ranks open GPU 0 (one gr_ctx each); gradients are reduced through torch.distributed gloo by
ganrev.parallel.DeviceTrainer.step_decomposed + host_allreduce_grads - the GANREV_ALL_RANKS_ON_DEVICE0 control flow of bench.py.

python dp_rank_worker.py RANK WORLD PORT OUTDIR   ->   OUTDIR/<case>_<mode>_rank<r>.npz, OUTDIR/meta.json (rank 0)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "gan-reverser_amd"), ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import numpy as np
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dp_common as D
    import ganrev._lib as L
    from ganrev import synth
    from ganrev.parallel import DeviceTrainer, host_allreduce_grads
    assert world == D.WORLD
    ctx = L.Context(0)                                   # every rank on device 0
    for name, dims, nd, B in D.CASES:
        G, R = D.make_models(dims, nd)
        G._ctx = R._ctx = ctx
        G.evaluate(); G.forward(synth.normal((2, nd), 1))
        R.training(); R.forward(synth.uniform((2,) + dims, 2, 0, 1)); R.push_params()
        R._pending_masks = {}
        gnet, rnet = G._net, R._net
        theta0 = rnet.get_params()
        layer_of = {id(m): R._leaf_layer(m) for m in R.leaves()}
        noise, masks = D.global_inputs(R, lambda m: layer_of[id(m)], rnet.mask_size, dims, nd, B)
        GB = B * world
        zeros = np.zeros_like(theta0)
        pools = [(layer_of[id(m)], i) for i, m in enumerate(R.leaves()) if m.typename == "nn.SpatialMaxPooling"]
        for mode in D.MODES:
            ctx.set_conv_mode(mode)
            rnet.set_params(theta0); rnet.set_adam_state(zeros, zeros)
            tr = DeviceTrainer(ctx, gnet, rnet, L.Hyper(), B, world, rank)
            tr.t = D.T_STEP - 1
            ctx.upload(D.shard(noise, rank), tr.noise)
            for li, k in masks.items():
                rnet.set_mask(li, D.shard(k.reshape(GB, -1), rank).ravel())
            store = {}
            reduce_ = host_allreduce_grads(dist)

            def reduce_and_keep(net):
                reduce_(net)
                store["raw_sum"] = net.get_grads()

            def reduce_scalar(x):
                import torch
                t = torch.tensor([x], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                return float(t.item())
            loss = tr.step_decomposed(reduce_and_keep, reduce_scalar)
            out = dict(loss=np.float64(loss), raw_sum=store["raw_sum"], grads=rnet.get_grads(), theta=rnet.get_params(),
                       preds=ctx.download(rnet.lib.gr_net_output_dev(rnet.h), (B, nd)))
            out["m"], out["v"] = rnet.adam_state()
            leaves = R.leaves()
            for li, i in pools:
                conv = next(m for m in reversed(leaves[:i]) if m.typename.endswith("SpatialConvolution"))
                c, h, w = conv.nOutputPlane, *_plane(R, leaves, conv, dims)
                out[f"pool{li}"] = rnet.pool_index(li, B * c * (h // 2) * (w // 2))
                out[f"y{layer_of[id(conv)]}"] = rnet.layer_output(layer_of[id(conv)], (B * c * h * w,))
            np.savez(os.path.join(outdir, f"{name}_{mode}_rank{rank}.npz"), **out)
        gnet.close(); rnet.close()
    seen = [None] * world
    dist.all_gather_object(seen, rank)
    if rank == 0:
        json.dump(dict(world=world, ranks=seen, device=ctx.info()), open(os.path.join(outdir, "meta.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


def _plane(R, leaves, conv, dims):
    """(H, W) of the plane `conv` works on: the input plane halved by every max-pool in front of it."""
    h, w = dims[1], dims[2]
    for m in leaves:
        if m is conv:
            break
        if m.typename == "nn.SpatialMaxPooling":
            h, w = h // 2, w // 2
    return h, w


if __name__ == "__main__":
    main()

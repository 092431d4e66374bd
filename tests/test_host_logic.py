"""not-gpu: the host-side mirror of the reference interface (models.lua / weight-init.lua / utils/nn_utils.lua)."""
import numpy as np
import pytest

import ganrev._lib as L
from ganrev import models, nn, nn_utils, synth, weight_init
from ganrev.parallel import shard_bounds


def test_parameter_counts_match_the_reference_architectures():
    # R = 4 656 928 / 17 275 876 ; G = 2 623 745 / 14 979 587
    def count(m):
        return sum(a.size for x in m.leaves() for a in x.param_arrays())
    assert count(models.create_R((1, 32, 32), 32)) == 4656928
    assert count(models.create_G((1, 32, 32), 32)) == 2623745
    assert count(models.create_R((3, 64, 64), 100)) == 17275876
    assert count(models.create_G((3, 64, 64), 100)) == 14979587


def test_R_layer_list_is_models_lua_389_464():
    R = models.create_R((1, 32, 32), 32)
    names = [m.typename for m in R.modules]
    assert names[0] == "nn.Copy" and names[-1] == "nn.Copy"
    body = names[1:-1]
    blk = ["nn.SpatialConvolution", "nn.SpatialBatchNormalization", "nn.ELU"]
    assert body[:4] == blk + ["nn.Dropout"]
    assert body[8:13] == blk + ["nn.SpatialMaxPooling", "nn.Dropout"]                 # models.lua:419-423
    assert body[21:26] == blk + ["nn.SpatialDropout", "nn.SpatialMaxPooling"]         # models.lua:436-440
    assert body[26:] == ["nn.View", "nn.Linear", "nn.BatchNormalization", "nn.ELU", "nn.Dropout", "nn.Linear"]
    Ru = models.create_R((3, 16, 16), 10, "uniform", fixer=True)
    names = [m.typename for m in Ru.modules]
    assert names[1] == "nn.Dropout" and Ru.modules[1].always_on and not Ru.modules[1].v2   # models.lua:399-406
    assert names[-2] == "nn.Tanh"                                                           # models.lua:452-454
    Ru.evaluate()
    assert Ru.modules[1].train is True          # drop.evaluate = function() end
    with pytest.raises(AssertionError):
        models.create_R((1, 8, 8), 4, "gaussian")                                           # models.lua:390


def test_descriptor_lists_and_stage_grammar():
    R = models.create_R((1, 32, 32), 32)
    descs, index = R._descs((1, 32, 32))
    kinds = [d[0] for d in descs]
    assert len(descs) == 32 and kinds.count(L.CONV3) == 6 and kinds.count(L.BN) == 7 and kinds.count(L.DROPOUT) == 6
    assert descs[0][:3] == (L.CONV3, 1, 64) and descs[-1][:3] == (L.LINEAR, 512, 32)
    view = [d for d in descs if d[0] == L.VIEW][0]
    assert view[1:4] == (8192, 1, 1)
    G = models.create_G((3, 64, 64), 100)
    gd, _ = G._descs((100, 1, 1))
    assert [d[0] for d in gd] == [L.LINEAR, L.BN, L.RELU, L.VIEW, L.UPSAMPLE2, L.CONV3, L.BN, L.RELU, L.UPSAMPLE2, L.CONV3, L.BN,
                                   L.RELU, L.CONV3, L.SIGMOID]
    assert gd[0][1:3] == (100, 512 * 16 * 16) and gd[3][1:4] == (512, 16, 16)


def test_weight_init_semantics():
    # weight-init.lua:52-72: nn.SpatialConvolution / nn.Linear re-initialised with U(+-stdv*sqrt(3)), every bias zeroed;
    # cudnn.SpatialConvolution (G) is NOT matched by the typename test, only its bias is zeroed
    R = models.create_R((1, 32, 32), 32)
    conv = R.modules[1]
    stdv = weight_init.w_init_heuristic(1 * 9, 64 * 9) * np.sqrt(3)
    assert np.all(conv.bias == 0) and np.abs(conv.weight).max() <= stdv + 1e-7 and np.abs(conv.weight).max() > 0.8 * stdv
    G = models.create_G((1, 32, 32), 32)
    gconv = [m for m in G.modules if m.typename == "cudnn.SpatialConvolution"][0]
    assert np.all(gconv.bias == 0)
    assert np.abs(gconv.weight).max() <= 1.0 / np.sqrt(512 * 9) + 1e-7     # constructor default, untouched
    bn = [m for m in G.modules if "BatchNormalization" in m.typename][0]
    assert np.all(bn.bias == 0) and bn.weight.min() >= 0 and bn.weight.max() <= 1


def test_getParameters_makes_views():
    R = models.create_R((1, 8, 8), 4)
    flat, grads = R.getParameters()
    conv = R.modules[1]
    conv.weight[0, 0, 0, 0] = 123.0
    assert flat[0] == 123.0 and flat.size == grads.size
    flat[1] = -7.0
    assert conv.weight[0, 0, 0, 1] == -7.0
    assert R.getParameters()[0] is flat


def test_forwardBatched_chunks_like_nn_utils_lua():
    class Fake:
        calls = []

        def forward(self, x):
            self.calls.append(len(x))
            return np.asarray(x, np.float32) * 2
    m = Fake()
    x = synth.normal((70, 3), 1)
    out = nn_utils.forwardBatched(m, x, 32)
    assert m.calls == [32, 32, 6] and np.array_equal(out, x * 2)           # ceil(70/32) batches, last one ragged
    assert nn_utils.createNoiseInputs(5, 7, "uniform").shape == (5, 7)
    assert np.abs(nn_utils.createNoiseInputs(1000, 8, "uniform")).max() <= 1
    with pytest.raises(ValueError):
        nn_utils.createNoiseInputs(1, 1, "laplace")


def test_synth_is_deterministic_and_sane():
    a, b = synth.normal((1000, 8), 3), synth.normal((1000, 8), 3)
    assert np.array_equal(a, b) and abs(float(a.mean())) < 0.05 and abs(float(a.std()) - 1) < 0.05
    assert not np.array_equal(a, synth.normal((1000, 8), 4))
    k = synth.bernoulli_keep((100000,), 1, 0.25)
    assert abs(k.mean() - 0.75) < 0.01


def test_unsupported_geometries_raise():
    with pytest.raises(L.GanrevError):
        nn.SpatialConvolution(3, 8, 7, 7, 1, 1, 3, 3)
    with pytest.raises(L.GanrevError):
        nn.SpatialConvolution(3, 8, 5, 5, 1, 1, 1, 1)          # 5x5 only with the D network's padding (models.lua:275)
    with pytest.raises(L.GanrevError):
        nn.SpatialFullConvolution(3, 8, 5, 5, 1, 1, 2, 2)
    assert nn.SpatialConvolution(3, 8, 5, 5, 1, 1, 2, 2).weight.shape == (8, 3, 5, 5)
    with pytest.raises(L.GanrevError):
        nn.SpatialMaxPooling(3, 3)
    with pytest.raises(L.GanrevError):
        nn.SpatialUpSamplingNearest(3)


def test_shard_bounds():
    assert [shard_bounds(4096, 8, r) for r in (0, 7)] == [(0, 512), (3584, 4096)]
    with pytest.raises(ValueError):
        shard_bounds(10, 4, 0)


def test_t7_reader_on_hand_assembled_bytes():
    """Torch7 binary serialisation (train_r.lua:68 torch.load; format restated from memory in ganrev/t7.py): a byte string put
    together here field by field - number, string, boolean, nested array table, a 2x3 FloatTensor view with a storage offset
    and non-trivial strides, a repeated reference, an nn object - must read back as the values it encodes."""
    import struct
    from ganrev import t7
    i32 = lambda v: struct.pack("<i", v)
    i64 = lambda v: struct.pack("<q", v)
    num = lambda v: i32(1) + struct.pack("<d", v)
    st = lambda s: i32(2) + i32(len(s)) + s.encode()
    raw = lambda s: i32(len(s)) + s.encode()
    storage = np.arange(10, dtype=np.float32)
    # tensor: 2x3 view of the storage, offset 2 (1-based 3), strides (1, 2): element [i][j] = storage[2 + i + 2j]
    tensor = (i32(4) + i32(5) + raw("V 1") + raw("torch.FloatTensor") + i32(2) + i64(2) + i64(3) + i64(1) + i64(2) + i64(3) +
              i32(4) + i32(6) + raw("V 1") + raw("torch.FloatStorage") + i64(10) + storage.tobytes())
    inner = i32(3) + i32(2) + i32(3) + num(1) + num(10) + num(2) + num(20) + num(3) + num(30)
    module = i32(4) + i32(7) + raw("V 1") + raw("nn.ReLU") + i32(3) + i32(8) + i32(1) + st("inplace") + i32(5) + i32(0)
    blob = (i32(3) + i32(1) + i32(7) +
            st("a") + num(1.5) + st("b") + st("xy") + st("flag") + i32(5) + i32(1) + st("arr") + inner +
            st("t") + tensor + st("again") + i32(4) + i32(5) + st("m") + module)
    o = t7.load(blob)
    assert o["a"] == 1.5 and o["b"] == "xy" and o["flag"] is True and o["arr"] == [10, 20, 30]
    assert o["t"].dtype == np.float32 and np.array_equal(o["t"], [[2, 4, 6], [3, 5, 7]])
    assert o["again"] is o["t"]
    assert o["m"].typename == "nn.ReLU" and o["m"].fields == {"inplace": False}


def test_t7_checkpoint_round_trip():
    """train_r.lua:234 / :68: {R=MODEL_R, opt=OPT} written by ganrev.t7.save_checkpoint and read back: same layers, parameters,
    running statistics, the fixer's always-on dropout, and the opt table; G with its cudnn-free layer set likewise."""
    from ganrev import models, synth, t7
    for make, dims in ((lambda: models.create_R((3, 16, 16), 10, "uniform", True), (3, 16, 16)), (lambda: models.create_G((1, 16, 16), 8), (8, 1, 1))):
        m = make(); synth.init_params(m, 9)
        for j, bn in enumerate([x for x in m.leaves() if hasattr(x, "running_mean")]):
            bn.running_mean[...] = synth.normal(bn.running_mean.shape, 30 + j); bn.running_var[...] = synth.uniform(bn.running_var.shape, 60 + j, 0.5, 2)
        blob = t7.dumps({"R": t7.from_model(m), "opt": {"noiseDim": 10, "noiseMethod": "uniform", "height": 16, "width": 16, "colorSpace": "rgb"}})
        back = t7.load(blob)
        assert back["opt"] == {"noiseDim": 10, "noiseMethod": "uniform", "height": 16, "width": 16, "colorSpace": "rgb"}
        m2 = t7.to_model(back["R"])
        assert m2._descs(dims)[0] == m._descs(dims)[0]                       # same layer descriptors (kinds, sizes, dropout flags)
        assert np.array_equal(m2._flat_host(), m._flat_host())
        for a, b in zip([x for x in m.leaves() if hasattr(x, "running_mean")], [x for x in m2.leaves() if hasattr(x, "running_mean")]):
            assert np.array_equal(a.running_mean, b.running_mean) and np.array_equal(a.running_var, b.running_var)


def test_D_network_structure_parts_and_checkpoint_round_trip():
    """models.create_D2 (models.lua:272-337): layer list, the compiled parts an nn.Concat splits it into, getParameters() views
    across the parts, top-level-only weight-init (weight-init.lua:52: the nested towers keep their constructor draw), and the
    train.lua:256 checkpoint round trip with nn.Concat / nn.PReLU / the 5x5 convolution."""
    from ganrev import t7
    D = models.create_D2((1, 32, 32), seed=4)
    names = [m.typename for m in D.modules]
    assert names == ["nn.Copy", "nn.Sequential", "nn.Sequential", "nn.SpatialMaxPooling", "nn.Concat", "nn.Linear", "nn.PReLU", "nn.Dropout",
                     "nn.Linear", "nn.Sigmoid", "nn.Copy"]
    concat = D.modules[4]
    assert concat.dimension == 2 and concat.size() == 2
    left = [m.typename for m in concat.get(1).leaves()]
    assert left == ["nn.SpatialConvolution", "nn.PReLU", "nn.SpatialDropout", "nn.SpatialMaxPooling", "nn.View", "nn.Linear", "nn.PReLU", "nn.Dropout"]
    conv5 = concat.get(1).leaves()[0]
    assert conv5.weight.shape == (64, 128, 5, 5) and conv5._wshape() == (64, 128, 5, 5)
    assert sum(1 for m in D.leaves() if m.typename == "nn.PReLU") == 9 and all(float(m.weight[0]) == 0.25 for m in D.leaves() if m.typename == "nn.PReLU")
    assert D._is_graph() and [type(p).__name__ for p in D.parts()] == ["Sequential", "Concat", "Sequential"]
    chunks = D._param_chunks()
    assert len(chunks) == 4 and chunks[0][1] == 0 and all(a[2] == b[1] for a, b in zip(chunks, chunks[1:])) and chunks[-1][2] == D._param_count()
    # 1->128 3x3 (+1 slope), 128->128 3x3 (+1): the trunk's share of the flat vector
    assert chunks[0][2] == (128 * 9 + 128 + 1) + (128 * 128 * 9 + 128 + 1)
    flat, grads = D.getParameters()
    conv5.weight[0, 0, 0, 0] = 77.0
    assert flat[chunks[1][1]] == 77.0                      # the 5x5 weights open the second part's slice
    head_linear = D.modules[5]
    assert np.all(head_linear.bias == 0)                   # weight-init reaches the top-level Linear ...
    assert np.abs(conv5.bias).max() > 0                    # ... not the modules nested in the towers (weight-init.lua:52 walks net.modules only)
    descs = D.parts()[1].modules[0]._descs((128, 16, 16))[0]
    assert descs[0][:4] == (L.CONVK, 128, 64, 5) and descs[1][0] == L.PRELU
    raw = t7.load(t7.dumps({"D": t7.from_model(D)}))["D"]
    back = t7.to_model(raw)
    # nn.Concat.size and nn.View.size are torch.LongStorage objects in Torch7's nn (a LongTensor there breaks the reference's
    # first forward of a loaded D: `self.size:resize(...):copy(outs[1]:size())`): the bytes must say so, not merely round-trip
    blob = t7.dumps({"D": t7.from_model(D)})
    cat = [m for m in raw.fields["modules"] if getattr(m, "typename", "") == "nn.Concat"][0]
    assert isinstance(cat.fields["size"], t7.Storage) and cat.fields["size"].dtype == np.int64 and cat.fields["size"].size == 0
    i = blob.index(b"nn.Concat")
    assert blob.index(b"torch.LongStorage", i) < blob.index(b"nn.Sequential", i), "nn.Concat.size is not written as a torch.LongStorage"
    def walk(o):
        yield o
        for m in (o.fields.get("modules") or []):
            yield from walk(m)
    views = [m for m in walk(t7.from_model(D)) if m.typename == "nn.View"]
    assert views and all(isinstance(v.fields["size"], t7.Storage) for v in views)
    assert [m.typename for m in back.leaves()] == [m.typename for m in D.leaves()]
    assert np.array_equal(back._flat_host(), flat) and back.modules[3].dimension == 2
    with pytest.raises(L.GanrevError):
        nn.PReLU(16)                                       # per-channel slopes: not what models.lua builds


def test_module_initialisation_draws_from_one_stream():
    """Torch draws every module's initial parameters from one process-wide generator: equal-shaped modules must not start
    out identical, create_*(seed=...) must reach the modules weight-init.lua does not touch (BatchNorm gammas, G's
    cudnn.SpatialConvolution), and the same seed must reproduce the same model."""
    R = models.create_R((1, 32, 32), 32, seed=3)
    bns = [m for m in R.leaves() if isinstance(m, nn.BatchNormalization) and m.nFeature == 64]
    assert len(bns) == 3 and not np.array_equal(bns[0].weight, bns[1].weight) and not np.array_equal(bns[1].weight, bns[2].weight)
    convs = [m for m in R.leaves() if m.typename == "nn.SpatialConvolution" and m.weight.shape == (64, 64, 3, 3)]
    assert not np.array_equal(convs[0].weight, convs[1].weight)
    G1, G2, G3 = models.create_G((1, 32, 32), 32, seed=1), models.create_G((1, 32, 32), 32, seed=1), models.create_G((1, 32, 32), 32, seed=2)
    c1, c2, c3 = ([m for m in g.leaves() if m.typename == "cudnn.SpatialConvolution"][0].weight for g in (G1, G2, G3))
    assert np.array_equal(c1, c2) and not np.array_equal(c1, c3)


def test_oracle_argmax_hooks_and_lean_backward(oracle):
    """The parity tests' hooks on the oracle: the recorded pool argmax, forcing it (its own argmax reproduces the same
    forward and gradients bit for bit; a re-routed window changes exactly that window's output), and the memory-lean backward
    (same gradients)."""
    from helpers import inject_noise, pool_layers
    dims, nd, B = (1, 16, 16), 8, 5
    R = models.create_R(dims, nd); synth.init_params(R, 3)
    onet = oracle.from_model(R, dims)
    onet.set_training(True)
    x = synth.uniform((B,) + dims, 5, 0, 1)
    gy = synth.normal((B, nd), 9)
    inject_noise(R, onet, B, 7)
    out0 = onet.forward(x)
    onet.zero_grads(); gin0 = onet.backward(x, gy); g0 = onet.grads.copy()
    pools = pool_layers(R, onet)
    assert len(pools) == 2
    idx = [onet.pool_index(li) for _, li, _ in pools]
    assert idx[0].size == B * 64 * 8 * 8 and idx[0].max() <= 3
    for (_, li, _), ix in zip(pools, idx):
        onet.force_pool_index(li, ix)
    onet.set_lean(True)
    assert np.array_equal(onet.forward(x), out0)
    onet.zero_grads(); gin1 = onet.backward(x, gy)
    assert np.array_equal(gin1, gin0) and np.array_equal(onet.grads, g0)
    # re-route one window of the first pool: its output becomes another element of the window
    li = pools[0][1]
    before = onet.forward(x) is not None and onet.layer_output(li).copy()
    pin = onet.layer_output(li - 1).reshape(B, 64, 8, 2, 8, 2)
    forced = idx[0].copy(); forced[0] = (forced[0] + 1) % 4
    onet.force_pool_index(li, forced)
    onet.forward(x)
    after = onet.layer_output(li)
    assert after[0] == pin[0, 0, 0, forced[0] >> 1, 0, forced[0] & 1] and np.array_equal(after[1:], before[1:])
    for _, li, _ in pools:
        onet.force_pool_index(li, None)
    onet.set_lean(False)
    assert np.array_equal(onet.forward(x), out0)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no torch.distributed.run environment must start two ranks itself (one child process per
    rank, this parent never touching a GPU) and relay rank 0's line with n_gpus = 2; with a WORLD_SIZE that contradicts --gpus it
    must refuse.  GANREV_BENCH_DRY_RUN stops each rank after the gloo rendezvous (this box has no GPU)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["GANREV_BENCH_DRY_RUN"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks"] == [[0, 0], [1, 1]]
    # the contract is ONE JSON line on stdout: torch's gloo transport prints "[Gloo] Rank r is connected to ..." on fd 1 when the group forms - bench.py keeps
    # that off stdout (round 6: found with the driver-style launch on the GPU box)
    assert [l for l in r.stdout.splitlines() if l.strip()] == [l for l in r.stdout.splitlines() if l.startswith("{")], r.stdout[:500]
    # ... and the same when the driver starts the ranks itself (python -m torch.distributed.run ... bench.py --gpus 2)
    d = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541",
                        os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert d.returncode == 0, d.stderr[-2000:]
    out_lines = [l for l in d.stdout.splitlines() if l.strip()]
    assert len(out_lines) == 1 and json.loads(out_lines[0])["n_gpus"] == 2, d.stdout[:500]
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=2" in bad.stderr


def test_bench_gpus_8_dry_run_forms_eight_ranks():
    """cfg4's launch shape (BASELINE configs[3]: 8 ranks): `python bench.py --gpus 8` spawns eight rank processes that meet in the gloo control group, each
    bound to its LOCAL_RANK's device, and rank 0 relays ONE line with n_gpus = 8 and the eight (rank, device) pairs; per-rank batch and global batch are the
    weak-scaling ones (256 per GPU at cfg2, 512 at cfg3 = cfg4's 4096).  The dry-run hook stops each rank after the rendezvous: this box has no GPU and the
    8-GPU run itself is the driver's (SCALE_rNN.json)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["GANREV_BENCH_DRY_RUN"] = "1"
    for wl, per, glob in (("cfg2", 256, 2048), ("cfg3", 512, 4096)):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0", "--workload", wl], env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert line["n_gpus"] == 8 and line["ranks"] == [[i, i] for i in range(8)], line
        assert line["scaling"] == "weak" and line["config"]["per_gpu_batch"] == per and line["config"]["global_batch"] == glob, line


def test_sharded_search_eight_ranks_with_ties_across_three_boundaries(oracle):
    """SURVEY 8e at cfg4's rank count: the corpus split over EIGHT unequal shards, local top-k, candidate gather, merge - bit-identical to the unsharded
    search, with one exact-tie group spread over four shards (three boundaries), a duplicate of the needle row in the last shard, and k chosen so that the
    cut falls INSIDE the tie group (global row order decides who is in).  Oracle as the local search (the HIP variant: tests/test_gpu_parity.py)."""
    from helpers import sharded_search_in_process, tied_corpus
    N, d = 4003, 32
    cuts = [0, 300, 811, 1500, 1501, 2400, 3000, 3777, N]            # unequal shards, one of a single row
    bounds = list(zip(cuts[:-1], cuts[1:]))
    needles = [99, 1500, 4002]
    emb, group = tied_corpus(N, d, 77, needles[0], bounds)
    for k in (4, 50):                                                 # k = 4: needle, its duplicate, then 2 of the 4 tied rows
        idx, sc = sharded_search_in_process(oracle.cosine_topk, emb, bounds, needles, k)
        ridx, rsc = oracle.cosine_topk(emb, needles, k)
        assert np.array_equal(idx, ridx) and np.array_equal(sc, rsc)
    ridx, rsc = oracle.cosine_topk(emb, needles, 6)
    assert sorted(ridx[0][:2].tolist()) == [99, bounds[-1][0] + 7] and ridx[0][2:6].tolist() == group      # the case is what it claims to be
    assert len(set(rsc[0][2:6].tolist())) == 1


def test_design_quotes_the_tracked_bench_summary():
    """VERDICT round 5, item 2 (the mis-filed r05 summary: DESIGN quoted 136.4 k img/s while the tracked file held a 3-step run at 50.6 k).  The tracked
    summary of the round - profiles/r06_bench_default.json, the --detail file of an UNPROFILED `python bench.py` - must be a real run (>= 20 timed steps,
    >= 5 warm-up, the headline workload, the arithmetic the line names), and every number DESIGN.md's current-round table quotes must be the number in that
    file (the table is generated from it by tools/update_design_table.py: equal to the printed precision, not merely within a box-to-box spread)."""
    import json, os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "profiles", "r06_bench_default.json")
    assert os.path.exists(path), "profiles/r06_bench_default.json is missing"
    d = json.load(open(path))
    assert d["steps"] >= 20 and d["warmup"] >= 5, (d["steps"], d["warmup"])
    assert d["n_gpus"] == 1 and "batch=256" in d["config"]["workload"] and d["dtype"].startswith("f32 via f16x3")
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - d["config"]["per_gpu_batch"]) < 0.01 * d["config"]["per_gpu_batch"]      # images/s x s/step = batch
    r = d["roofline"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 5e-4 and abs(r["achieved"] - r["algorithmic_gflop_per_launch"] / r["avg_launch_ms"]) < 0.02 * r["achieved"]
    assert r["avg_launch_ms"] * r["launches_per_step"] < d["ms_per_step"]
    assert "cpu_baseline" in d and d["cpu_baseline"]["kind"] == "port" and "cfg3" in d and d["search_cfg5"]["exact_match"] is True
    text = open(os.path.join(root, "DESIGN.md")).read()
    m = re.search(r"<!-- bench-table:begin -->\n(.*?)\n<!-- bench-table:end -->", text, flags=re.S)
    assert m, "DESIGN.md has no bench table"
    rows = [l for l in m.group(1).splitlines() if l.startswith("|") and "`" in l]
    assert len(rows) >= 15, rows

    def lookup(o, dotted):
        for k in dotted.split("."):
            o = o[k]
        return o
    checked = 0
    for l in rows:
        keys = re.search(r"\| `([^`]+)` \|", l).group(1).split(";")          # (a row may carry two keys, "a;b", its value cell then reads "x / y")
        shown = [v.strip() for v in l.rstrip("|").rsplit("|", 1)[1].split(" / ")]
        assert len(keys) == len(shown), l
        for key, txt in zip(keys, shown):
            want = lookup(d, key)
            if isinstance(want, bool) or isinstance(want, str):
                assert txt == str(want), (key, txt, want)
            else:
                digits = len(txt.split(".")[1]) if "." in txt else 0
                assert abs(float(txt) - float(want)) <= 0.5 * 10 ** (-digits) + 1e-12, (key, txt, want)
            checked += 1
    assert checked >= 20
    # the headline quoted in the prose around the table must be the file's too (no second source of truth)
    for k_img in re.findall(r"\*\*(\d{3}\.\d) k img/s", text):
        assert abs(float(k_img) * 1e3 - d["value"]) <= 60, (k_img, d["value"])


def test_bench_headline_is_compact_strict_json():
    """the LAST stdout line of bench.py must be a compact (< 4 KB), strictly valid JSON object carrying the contract's keys,
    `roofline` and `cpu_baseline`; the tables go to a side file.  Built here from a canned full result (a committed copy of round 4's 23 KB line) and from the
    same result with hostile values (NaN, a tripped range guard, failed side legs, pathologically long notes)."""
    import copy, json, os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    full = json.load(open(os.path.join(root, "tests", "golden", "bench_result_canned.json")))
    assert len(json.dumps(full)) > 20000                       # the size that broke the driver
    text = bench.headline_text(full)
    assert len(text) < 4096 and "\n" not in text
    line = json.loads(text, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))      # NaN / Infinity would raise
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "rccl_ranks", "roofline", "cpu_baseline", "f32_row", "cfg3", "search_cfg5"):
        assert k in line, k
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"] and line["config"]["workload"] == full["config"]["workload"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches_per_step", "algorithmic_gflop_per_launch"):
        assert line["roofline"][k] == full["roofline"][k], k
    for k in ("value", "unit", "cores", "kind"):
        assert line["cpu_baseline"][k] == full["cpu_baseline"][k], k
    assert line["cfg3"]["images_per_sec"] == full["cfg3"]["images_per_sec"] and line["cfg3"]["roofline_frac"] == full["cfg3"]["roofline"]["frac"]
    assert line["search_cfg5"]["exact_match"] is True and line["f32_row"]["ms_per_step"] == full["f32_row"]["ms_per_step"]
    assert "model" not in line["config"]
    # hostile variant
    bad = copy.deepcopy(full)
    bad["last_loss"] = float("nan"); bad["roofline"]["traffic"] = float("inf")
    bad["dtype"] = "x" * 3000; bad["cpu_baseline"]["sample"] = "y" * 5000
    bad["gan_step"] = {"error": "RuntimeError: " + "z" * 5000}
    bad["search_cfg5"]["embed"] = {"error": "boom"}
    bad = bench._finite(bad)
    t2 = bench.headline_text(bad)
    l2 = json.loads(t2)
    assert len(t2) < 4096 and l2["roofline"]["traffic"] is None and l2["search_cfg5"]["embed_error"] == "boom" and len(l2["dtype"]) < 120
    json.dumps(bad, allow_nan=False)                           # the detail file is strict JSON too


def test_oracle_act_side_hook_moves_exactly_the_forced_elements():
    """go_net_force_act_side: forcing the side the oracle would take anyway changes nothing; flipping one LeakyReLU input's side
    changes gradInput of the activation by gout * (1 - slope) at that element only (seen through a 1x1 identity-free net:
    LeakyReLU alone)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "gan-reverser_amd"))
    from ganrev import nn, synth
    from oracle import oracle
    net = nn.Sequential(); net.add(nn.LeakyReLU(0.333))
    onet = oracle.from_model(net, (3, 4, 4))
    x = synth.normal((2, 3, 4, 4), 1); g = synth.normal((2, 3, 4, 4), 2)
    onet.forward(x); base = onet.backward(x, g).copy()
    own = (x.reshape(-1) > 0)
    onet.force_act_side(0, own)
    onet.forward(x); assert np.array_equal(onet.backward(x, g), base)
    flipped = own.copy(); flipped[5] = not flipped[5]
    onet.force_act_side(0, flipped)
    onet.forward(x); got = onet.backward(x, g).reshape(-1)
    d = got - base.reshape(-1)
    assert np.count_nonzero(d) == 1 and abs(abs(d[5]) - abs(g.reshape(-1)[5]) * (1 - 0.333)) < 1e-6
    onet.force_act_side(0, None)
    onet.forward(x); assert np.array_equal(onet.backward(x, g), base)


def test_oracle_bce_matches_float64_formula():
    """go_bce against the THNN formula in numpy float64 (EPS = 1e-12, sizeAverage), incl. predictions at exactly 0 and 1."""
    from oracle import oracle
    rng = np.random.default_rng(3)
    x = rng.random(1000).astype(np.float32); t = (rng.random(1000) < 0.5).astype(np.float32)
    x[:2] = (0.0, 1.0); t[:2] = (0.0, 1.0)
    loss, g = oracle.bce(x, t)
    xd, td = x.astype(np.float64), t.astype(np.float64)
    ref = -np.mean(np.log(xd + 1e-12) * td + np.log(1 - xd + 1e-12) * (1 - td))
    gref = (-(td - xd) / ((1 - xd + 1e-12) * (xd + 1e-12)) / x.size).astype(np.float32)
    assert abs(loss - ref) <= 1e-12 * max(1.0, abs(ref))
    assert np.array_equal(g, gref)


def test_adversarial_host_pieces():
    """ganrev.adversarial without a GPU: the penalty / clamp helpers (adversarial.lua:8-28), the option table of train.lua:27-38,
    the optimiser state tables of train.lua:183-193, and the unknown-method
    error text of adversarial.lua:170."""
    from ganrev import adversarial
    theta = np.array([1.0, -2.0, 0.0, 0.5], np.float32)
    g = np.array([0.3, -7.0, 9.0, -0.1], np.float32)
    f = adversarial.l2(theta, g, 1.0, 0.1)                               # f + l2 * ||theta||^2 / 2 ; g += l2 * theta
    assert abs(f - (1.0 + 0.1 * 5.25 / 2)) < 1e-7 and np.allclose(g, [0.4, -7.2, 9.0, -0.05])
    f = adversarial.l1(theta, g, f, 0.5)                                 # f + l1 * |theta|_1 ; g += l1 * sign(theta)
    assert abs(f - (1.0 + 0.2625 + 0.5 * 3.5)) < 1e-6 and np.allclose(g, [0.9, -7.7, 9.0, 0.45])
    adversarial.clamp(g, 5.0)
    assert np.allclose(g, [0.9, -5.0, 5.0, 0.45])
    g0 = g.copy(); adversarial.clamp(g, 0); assert np.array_equal(g, g0)      # clampValue 0: untouched (adversarial.lua:9)
    assert adversarial.l2(theta, g, 2.0, 0) == 2.0 and np.array_equal(g, g0)
    G, D = models.create_G((1, 16, 16), 8), models.create_D2((1, 16, 16))
    env = adversarial.make_env(G, D, (1, 16, 16), batchSize=8, noiseDim=8)
    assert (env.OPT.D_L2, env.OPT.D_clamp, env.OPT.G_clamp, env.OPT.G_L2, env.OPT.D_optmethod) == (1e-4, 1.0, 5.0, 0.0, "adam")   # train.lua:27-38
    assert env.PARAMETERS_D.size == D._param_count() and env.PARAMETERS_G.size == G._param_count() and env.CONFUSION.shape == (2, 2)
    assert all(m.train for m in D.listModules()) and all(m.train for m in G.listModules())                                      # train.lua:133-134
    with pytest.raises(L.GanrevError):
        adversarial.make_env(G, D, (1, 16, 16), batchsize=8)
    assert env.OPTSTATE["sgd"]["D"] == {"learningRate": 0.02, "momentum": 0.0} and env.OPTSTATE["rmsprop"]["G"] == {}         # train.lua:183-193
    env.OPT.D_optmethod = "lbfgs"
    with pytest.raises(L.GanrevError, match="Unknown optimizer method 'lbfgs' chosen for D."):
        adversarial._optimize(env, "D", lambda x: (0.0, x), env.PARAMETERS_D, D)
    from ganrev import train
    imgs = train.synthetic_images(6, (3, 16, 16), 5)
    assert imgs.shape == (6, 3, 16, 16) and imgs.dtype == np.float32 and 0 <= imgs.min() and imgs.max() <= 1 and not np.array_equal(imgs[0], imgs[1])


def test_host_optimisers_follow_their_update_rules():
    """ganrev.optim's five host methods (adversarial.lua:156-171 picks among them) on a quadratic f = |x - c|^2 / 2: each step is
    checked against the update rule spelled out in float64, state carried across calls, and all of them reduce f."""
    from ganrev import optim
    rng = np.random.default_rng(3)
    c = rng.standard_normal(64).astype(np.float32)

    def feval(x):
        return float(0.5 * np.sum((x - c) ** 2)), (x - c).astype(np.float32)

    def run(name, cfg, ref_step, steps=5):
        x = np.zeros(64, np.float32); xr = np.zeros(64, np.float64); st = {}; cfg = dict(cfg)
        f0 = feval(x)[0]
        for t in range(1, steps + 1):
            g = xr - c
            xr = ref_step(xr, g, st, t)
            xo, fs = optim.METHODS[name](feval, x, cfg)
            assert xo is x and len(fs) == 1                                # in place, {f(x)} returned like optim.*
            assert np.allclose(x, xr, rtol=2e-5, atol=2e-6), (name, t, np.abs(x - xr).max())
        assert feval(x)[0] < f0
        return cfg

    def sgd_ref(x, g, st, t):
        st["b"] = g.copy() if "b" not in st else 0.9 * st["b"] + (1 - 0.9) * g           # dampening defaults to the momentum
        return x - 0.02 / (1 + (t - 1) * 0.1) * st["b"]
    cfg = run("sgd", {"learningRate": 0.02, "momentum": 0.9, "learningRateDecay": 0.1}, sgd_ref)
    assert cfg["evalCounter"] == 5 and cfg["dfdx"].shape == (64,)                         # the state lives in the config table (train.lua:189-191)
    run("sgd", {"learningRate": 0.02, "momentum": 0.0}, lambda x, g, st, t: x - 0.02 * g)

    def adagrad_ref(x, g, st, t):
        st["v"] = st.get("v", 0) + g * g
        return x - 1e-3 * g / (np.sqrt(st["v"]) + 1e-10)
    run("adagrad", {}, adagrad_ref)

    def adadelta_ref(x, g, st, t):
        st["v"] = 0.9 * st.get("v", 0) + 0.1 * g * g
        d = np.sqrt(st.get("a", 0) + 1e-6) / np.sqrt(st["v"] + 1e-6) * g
        st["a"] = 0.9 * st.get("a", 0) + 0.1 * d * d
        return x - d
    run("adadelta", {}, adadelta_ref)

    def adamax_ref(x, g, st, t):
        st["m"] = 0.9 * st.get("m", 0) + 0.1 * g
        st["u"] = np.maximum(0.999 * st.get("u", 0), np.abs(g) + 1e-38)
        return x - 2e-3 / (1 - 0.9 ** t) * st["m"] / st["u"]
    run("adamax", {}, adamax_ref)

    def rmsprop_ref(x, g, st, t):
        st["m"] = 0.99 * st.get("m", 0) + 0.01 * g * g
        return x - 1e-2 * g / (np.sqrt(st["m"]) + 1e-8)
    run("rmsprop", {}, rmsprop_ref)
    with pytest.raises(ValueError):
        optim.sgd(feval, np.zeros(64, np.float32), {"nesterov": True})

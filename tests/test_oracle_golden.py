"""not-gpu: the oracle reproduces the committed golden vectors (tests/golden/golden_v1.npz, made by make_golden.py)."""
import os

import numpy as np
import pytest

from golden_cases import CASES, run_oracle_case

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v1.npz"))


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_golden(oracle, name):
    res = run_oracle_case(oracle, CASES[name])
    for k, v in res.items():
        if k == "grads_full":
            continue
        g = GOLD[f"{name}/{k}"]
        if np.asarray(v).dtype.kind in "iu":
            assert np.array_equal(v, g), f"{name}/{k}"
        else:
            # same binary, same thread count -> identical; a different OpenMP schedule may move conv sums by an ulp
            scale = max(1.0, float(np.max(np.abs(g))))
            assert np.max(np.abs(np.asarray(v, np.float64) - g)) <= 1e-5 * scale, f"{name}/{k}"


from golden_cases import DCASES, run_oracle_dcase  # noqa: E402

GOLD_D = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v2_dnet.npz"))


@pytest.mark.parametrize("name", sorted(DCASES))
def test_oracle_matches_dnet_golden(oracle, name):
    """The D network's module types (5x5 convolution, nn.PReLU, nn.Concat: models.lua:272-337), golden_v2_dnet.npz."""
    res = run_oracle_dcase(oracle, DCASES[name])[0]
    for k, v in res.items():
        g = GOLD_D[f"{name}/{k}"]
        scale = max(1.0, float(np.max(np.abs(g))))
        assert np.max(np.abs(np.asarray(v, np.float64) - g)) <= 1e-5 * scale, f"{name}/{k}"


# ---------------------------------------------------------------------------------------------------------------------
# The im2col + blocked-sgemm convolution path (oracle/oracle_mm.c, go_set_conv_impl(1)): THNN SpatialConvolutionMM's structure,
# the CPU baseline bench.py reports.  Pinned here against the direct loops - the parity
# oracle - operator by operator on ragged shapes, and through the golden R / G / train-step cases at the golden tolerance.
@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 3, 5, 6, 7), (3, 8, 16, 8, 8), (1, 1, 64, 32, 32), (2, 64, 64, 16, 16), (5, 17, 9, 12, 20), (2, 128, 3, 16, 16)])
def test_im2col_sgemm_convolution_matches_the_direct_loops(oracle, B, Cin, Cout, H, W):
    from ganrev import synth
    x = synth.normal((B, Cin, H, W), 11)
    w = synth.normal((Cout, Cin, 3, 3), 12) * np.float32(1.0 / np.sqrt(9 * Cin))
    b = synth.normal((Cout,), 13)
    gy = synth.normal((B, Cout, H, W), 14)
    prev = oracle.set_conv_impl("direct")
    try:
        ref = (oracle.conv3_forward(x, w, b), oracle.conv3_backward_data(gy, w), *oracle.conv3_backward_weight(x, gy))
        oracle.set_conv_impl("mm")
        got = (oracle.conv3_forward(x, w, b), oracle.conv3_backward_data(gy, w), *oracle.conv3_backward_weight(x, gy))
    finally:
        oracle.set_conv_impl(prev)
    for name, g, r in zip(("forward", "gradInput", "gradWeight", "gradBias"), got, ref):
        scale = max(1.0, float(np.abs(r).max()))
        d = float(np.abs(g.astype(np.float64) - r).max())
        assert d <= 2e-6 * scale * np.sqrt(9 * Cin + B * H * W if name.startswith("gradW") else 9 * max(Cin, Cout)), f"{name}: {d:.3e} (scale {scale:.3g})"
    # known answers: a centre-tap identity kernel reproduces the input, a shift kernel shifts it (zero padding at the border)
    oracle.set_conv_impl("mm")
    try:
        wi = np.zeros((Cin, Cin, 3, 3), np.float32); wi[np.arange(Cin), np.arange(Cin), 1, 1] = 1
        assert np.array_equal(oracle.conv3_forward(x, wi, None), x)
        ws = np.zeros((Cin, Cin, 3, 3), np.float32); ws[np.arange(Cin), np.arange(Cin), 0, 2] = 1      # out[y][x] = in[y-1][x+1]
        sh = np.zeros_like(x); sh[:, :, 1:, :-1] = x[:, :, :-1, 1:]
        assert np.array_equal(oracle.conv3_forward(x, ws, None), sh)
    finally:
        oracle.set_conv_impl(prev)


@pytest.mark.parametrize("name", [n for n in sorted(CASES) if CASES[n]["kind"] in ("R", "G", "train")][:6])
def test_im2col_sgemm_path_reproduces_the_golden_cases(oracle, name):
    prev = oracle.set_conv_impl("mm")
    try:
        res = run_oracle_case(oracle, CASES[name])
    finally:
        oracle.set_conv_impl(prev)
    for k, v in res.items():
        if k == "grads_full" or np.asarray(v).dtype.kind in "iu":
            continue
        g = GOLD[f"{name}/{k}"]
        scale = max(1.0, float(np.max(np.abs(g))))
        assert np.max(np.abs(np.asarray(v, np.float64) - g)) <= 1e-4 * scale, f"{name}/{k}"

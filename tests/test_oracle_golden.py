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

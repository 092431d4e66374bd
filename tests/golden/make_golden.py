#!/usr/bin/env python3
"""Generates tests/golden/golden_v1.npz.

The reference (Lua/Torch7) cannot run in the build container and holds no test vectors of its own, so these are
outputs of the CPU oracle (oracle/), each cross-checked here against an independent float64 PyTorch-CPU evaluation
before being frozen.  Inputs are NOT stored: they are regenerated from ganrev.synth seeds recorded in CASES.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "gan-reverser_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

from ganrev import models, synth  # noqa: E402
from oracle import oracle  # noqa: E402
from golden_cases import CASES, build_case, run_oracle_case  # noqa: E402
from torch_twin import Twin  # noqa: E402


def main():
    out = {}
    for name, case in CASES.items():
        res = run_oracle_case(oracle, case)
        if case["kind"] in ("R", "G"):
            model, in_dims, x, masks = build_case(case)
            descs, _ = model._descs(tuple(in_dims))
            running = [(m.running_mean.copy(), m.running_var.copy()) for m in model.leaves() if hasattr(m, "running_mean")]
            twin = Twin(descs, in_dims, model._flat_host(), running, case["training"], masks)
            ref = twin.forward(x)
            assert np.abs(ref - res["out"]).max() < 5e-5, (name, np.abs(ref - res["out"]).max())
            if case["kind"] == "R" and case["training"]:
                g = twin.backward(synth.normal(ref.shape, case["seed"] + 9) * np.float32(0.1))
                assert np.abs(g - res["grads_full"]).max() < 3e-4 * max(1, np.abs(g).max()), name
        for k, v in res.items():
            if k != "grads_full":
                out[f"{name}/{k}"] = v
        print(name, {k: getattr(v, "shape", v) for k, v in res.items() if k != "grads_full"})
    np.savez_compressed(os.path.join(HERE, "golden_v1.npz"), **out)
    print("wrote", os.path.join(HERE, "golden_v1.npz"), os.path.getsize(os.path.join(HERE, "golden_v1.npz")), "bytes")


def main_dnet():
    """tests/golden/golden_v2_dnet.npz: the D network's module types (golden_cases.DCASES).  The sequential chain is checked
    against float64 autograd here; the four-part D2 is the same per-part arithmetic composed by helpers.OracleGraph (its parts'
    operators are pinned by tests/test_oracle_vs_torch.py)."""
    from golden_cases import DCASES, run_oracle_dcase
    out = {}
    for name, case in DCASES.items():
        res, twin, model, pairs = run_oracle_dcase(oracle, case)
        if case["kind"] == "chain":
            descs, index = model._descs(tuple(case["dims"]))
            masks = {index[id(m)]: synth.bernoulli_keep((pairs[0][1].mask_size(index[id(m)], case["B"]),), case["seed"] * 131 + index[id(m)], m.p)
                     for m in model.leaves() if m.typename in ("nn.Dropout", "nn.SpatialDropout")}      # what helpers.inject_noise drew
            tw = Twin(descs, case["dims"], model._flat_host(), [], True, masks)
            x = synth.uniform((case["B"],) + tuple(case["dims"]), case["seed"] + 1, 0, 1)
            ref = tw.forward(x)
            assert np.abs(ref - res["out"]).max() < 5e-6, (name, np.abs(ref - res["out"]).max())
            g = tw.backward(synth.normal(ref.shape, case["seed"] + 9))
            assert np.abs(g - res["grads"]).max() < 1e-4 * max(1, np.abs(g).max()), name
        for k, v in res.items():
            out[f"{name}/{k}"] = v
        print(name, {k: getattr(v, "shape", v) for k, v in res.items()})
    path = os.path.join(HERE, "golden_v2_dnet.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    if "--dnet" in sys.argv:
        main_dnet()
    else:
        main()

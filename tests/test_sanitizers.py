"""not-gpu: AddressSanitizer + UndefinedBehaviorSanitizer over the CPU side.

The oracle is C with hand-written index arithmetic (halo handling, ragged planes, group slices, top-k buffers): exactly the code a
sanitizer is for, and the thing every parity claim rests on.  Two legs:
  1. oracle/asan_harness.c, a C program that walks the oracle's operators and nets at ragged sizes, linked against the
     sanitized build (`make -C oracle asan`);
  2. the ctypes-driven host logic - oracle.py's bindings, golden cases, the grouped data-parallel step of tests/dp_common.py's
     oracle side - in a child Python with libasan preloaded and the sanitized library loaded in place of the normal one.
The GPU library is never built with sanitizers (not available on the GPU pool)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle")


@pytest.fixture(scope="module")
def asan_build():
    r = subprocess.run(["make", "-C", ORACLE, "-s", "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return os.path.join(ORACLE, "_asan")


def _env(**extra):
    return dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
                OMP_NUM_THREADS="4", **extra)


def test_oracle_c_harness_under_asan_ubsan(asan_build):
    r = subprocess.run([os.path.join(asan_build, "asan_harness")], capture_output=True, text=True, env=_env(), timeout=600)
    assert r.returncode == 0 and "asan_harness: ok" in r.stdout, (r.stdout + r.stderr)[-4000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]


_CHILD = r"""
import os, sys
sys.path[:0] = [os.path.join({root!r}, "gan-reverser_amd"), {root!r}, os.path.join({root!r}, "tests")]
import numpy as np
from oracle import oracle
assert "_asan" in oracle._SO, oracle._SO
from golden_cases import CASES, run_oracle_case
GOLD = np.load(os.path.join({root!r}, "tests", "golden", "golden_v1.npz"))
for name in ("R_gray8_train", "G_gray32", "search_10k"):
    if name not in CASES:
        continue
    res = run_oracle_case(oracle, CASES[name])
    for k, v in res.items():
        if k == "grads_full":
            continue
        g = GOLD[name + "/" + k]
        if np.asarray(v).dtype.kind in "iu":
            assert np.array_equal(v, g), (name, k)
        else:
            # (the sanitized build is -O1 without FMA contraction: sums round differently from the golden build; memory errors are the point here)
            assert np.max(np.abs(np.asarray(v, np.float64) - g)) <= 2e-4 * max(1.0, float(np.max(np.abs(g)))), (name, k)
# the grouped data-parallel step of the oracle side (tests/dp_common.py), both convolution implementations
import dp_common as D
for impl in ("direct", "mm"):
    oracle.set_conv_impl(impl)
    dims, nd, B = (1, 8, 8), 6, 3
    G, R = D.make_models(dims, nd)
    oG, oR = oracle.from_model(G, (nd, 1, 1)), oracle.from_model(R, dims)
    noise, masks = D.global_inputs(R, lambda m: oR.layer_index[id(m)], oR.mask_size, dims, nd, B)
    ref = D.oracle_grouped_step(oracle, oG, oR, noise, masks, oR.params.copy(), oracle.GoHyper(), R=R)
    assert np.isfinite(ref["loss"]) and np.isfinite(ref["theta"]).all()
oracle.set_conv_impl("direct")
idx, sc = oracle.cosine_topk(np.random.default_rng(1).standard_normal((300, 9)).astype(np.float32), [0, 299], 500)   # k > N
assert idx.shape[1] == 300
print("host logic under asan: ok")
"""


def test_ctypes_host_logic_under_asan_ubsan(asan_build):
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan.so not found next to gcc")
    env = _env(LD_PRELOAD=libasan, GANREV_ORACLE_SO=os.path.join(asan_build, "libganrev_oracle_asan.so"))
    r = subprocess.run([sys.executable, "-c", _CHILD.format(root=ROOT)], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "host logic under asan: ok" in r.stdout, (r.stdout + r.stderr)[-4000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]

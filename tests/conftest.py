import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "gan-reverser_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def ctx():
    import ganrev._lib as L
    return L.default_context()


def _reset_guard(c):
    """The trainer's range guard is context state (a tripped guard keeps the context on bf16x6; a scan's verdict is read by the next
    gr_train_r_step): no test may inherit it from the one before."""
    c.set_tuning("range_guard", 0); c.set_tuning("range_guard", 1)


@pytest.fixture(autouse=True)
def _default_arithmetic_between_gpu_tests(request):
    """Every GPU test starts from the library's defaults on the shared context: f16x3 arithmetic, guard untripped, nothing pending."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import ganrev._lib as L
    c = L.default_context()
    _reset_guard(c)
    first = _DP.setdefault("default_mode", c.conv_mode())         # f16x3 unless GR_CONV_MODE says otherwise
    c.set_conv_mode(first)                                        # (a trainer whose guard tripped leaves its context on bf16x6: by design)
    yield


@pytest.fixture(params=["f32", "bf16x6", "f16x3"])
def conv_mode(request):
    """Run the test once per convolution arithmetic: exact fp32 MFMA, the fp32-accurate 3-term bf16 split (bf16x6) and the
    fp32-accurate 2-term fp16 split of scaled operands (f16x3) that bench.py uses by default.  Same tolerances for all three."""
    import ganrev._lib as L
    c = L.default_context()
    _reset_guard(c)
    prev = c.conv_mode()
    c.set_conv_mode(request.param)
    yield request.param
    _reset_guard(c)
    c.set_conv_mode(prev)


@pytest.fixture(params=["f16x3", "f16x3-p16"])
def f16_path(request):
    """f16x3 arithmetic with the default kernel selection, and with the operand-ready (P16: LDS-DMA convolution kernels fed by the
    8-channel-group pipeline kernels) path forced onto shapes with fewer tiles than the chip has CUs."""
    import ganrev._lib as L
    c = L.default_context()
    _reset_guard(c)
    prev = c.conv_mode()
    c.set_conv_mode("f16x3")
    c.set_tuning("p16_min_tiles", 1 if request.param.endswith("p16") else 128)
    yield request.param
    c.set_tuning("p16_min_tiles", 128)     # the library default (conv.hip g_p16_min_tiles)
    c.set_conv_mode(prev)


# ---------------------------------------------------------------------------------------------------------------------
# Two-process data-parallel parity test (tests/test_gpu_dp.py): the rank processes are started HERE, at the end of collection,
# i.e. before any test - and therefore this process - has initialised the GPU (on this pool a process that has touched HIP
# must not start another program).  They run beside the first tests (3 processes on the card, limit 6) and the test that
# needs them waits for their exit.
_DP = {}


def pytest_collection_finish(session):
    if not any(item.name.startswith("test_dp_two_processes") for item in session.items):
        return
    import socket
    import subprocess
    import tempfile
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = tempfile.mkdtemp(prefix="ganrev_dp_")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), OMP_NUM_THREADS="4")
    worker = os.path.join(ROOT, "tests", "dp_rank_worker.py")
    _DP["out"] = out
    _DP["procs"] = [subprocess.Popen([sys.executable, worker, str(r), "2", str(port), out], env=env,
                                     stdout=open(os.path.join(out, f"rank{r}.log"), "w"), stderr=subprocess.STDOUT) for r in range(2)]


@pytest.fixture(scope="session")
def dp_children():
    def wait(timeout=600):
        if "procs" not in _DP:
            pytest.fail("the data-parallel rank processes were not started (collection hook did not see this test)")
        for r, p in enumerate(_DP["procs"]):
            try:
                rc = p.wait(timeout=timeout)
            except Exception:  # noqa: BLE001
                p.kill()
                rc = "timeout"
            if rc != 0:
                log = open(os.path.join(_DP["out"], f"rank{r}.log")).read()[-3000:]
                pytest.fail(f"data-parallel rank {r} exited with {rc}:\n{log}")
        return _DP["out"]
    yield wait
    for p in _DP.get("procs", []):
        if p.poll() is None:
            p.kill()

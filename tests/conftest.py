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


@pytest.fixture(params=["f32", "bf16x6", "f16x3"])
def conv_mode(request):
    """Run the test once per convolution arithmetic: exact fp32 MFMA, the fp32-accurate 3-term bf16 split (bf16x6) and the
    fp32-accurate 2-term fp16 split of scaled operands (f16x3) that bench.py uses by default.  Same tolerances for all three."""
    import ganrev._lib as L
    c = L.default_context()
    prev = c.conv_mode()
    c.set_conv_mode(request.param)
    yield request.param
    c.set_conv_mode(prev)


@pytest.fixture(params=["f16x3", "f16x3-p16"])
def f16_path(request):
    """f16x3 arithmetic with the default kernel selection, and with the operand-ready (P16: LDS-DMA convolution kernels fed by the
    8-channel-group pipeline kernels) path forced onto shapes with fewer tiles than the chip has CUs."""
    import ganrev._lib as L
    c = L.default_context()
    prev = c.conv_mode()
    c.set_conv_mode("f16x3")
    c.set_tuning("p16_min_tiles", 1 if request.param.endswith("p16") else 128)
    yield request.param
    c.set_tuning("p16_min_tiles", 128)     # the library default (conv.hip g_p16_min_tiles)
    c.set_conv_mode(prev)

"""Does train_r's guard trip on the small default nets of tests/test_gpu_parity.py::test_train_r_reads_and_writes_torch7_checkpoints?"""
import os, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "gan-reverser_amd"))
import numpy as np
import ganrev._lib as L
from ganrev import models, synth, t7, train_r
c = L.default_context()
d = tempfile.mkdtemp()
G = models.create_G((1, 16, 16), 8); synth.init_params(G, 3)
gpath, rpath = os.path.join(d, "g.net"), os.path.join(d, "r.net")
t7.save_checkpoint(gpath, G=G, opt={"noiseDim": 8, "noiseMethod": "normal", "height": 16, "width": 16, "colorSpace": "y"})
print("before", c.conv_mode(), c.range_guard_stats())
G2, R, losses = train_r.main(["--G", gpath, "--save", rpath, "--nbBatches", "3", "--batchSize", "8", "--quiet"])
print("after", c.conv_mode(), c.range_guard_stats())
for name, M in (("G", G2), ("R", R)):
    for m in M.leaves():
        if hasattr(m, "weight") and m.weight is not None and m.weight.ndim >= 2:
            w = np.abs(m.weight.reshape(m.weight.shape[0], -1)).max(1)
            print(name, m.typename, m.weight.shape, "per-out-channel max: min %.3g max %.3g" % (w.min(), w.max()))

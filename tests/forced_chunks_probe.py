"""Helper of test_loss_edges_gpu.py::test_forced_chunk_heights: run as a script in a child process with SFM_CHUNK_ROWS set
(the library reads its tuning overrides once per process).  Compares the fused loss and its gradients with the oracle on one
shape whose passes then have the fewest / the most steps a pass can have (chunks of 4 rows: 8 steps; 28 rows: 32 steps, the
whole width of the step masks) and prints OK."""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import test_loss_gpu as T   # noqa: E402
from util import to_np      # noqa: E402

PKG = "sfm-learner-chainer_amd"
ops = importlib.import_module(PKG + ".ops")
synth = importlib.import_module(PKG + ".synth")


def main():
    dev = torch.device("cuda", 0)
    rows = int(os.environ["SFM_CHUNK_ROWS"])
    for cfg_name, layout in (("ssim_smooth", "hwc"), ("edge_aware", "planar"), ("l1_smooth", "hwc")):
        cfg = T.CONFIGS[cfg_name]
        d = synth.make_inputs(B=2, H=84, W=70, n_src=2, n_scales=2, seed=21)
        ref = T._oracle(d, cfg)
        fl = T._bind(ops, dev, d, cfg, layout=layout)
        T._check_losses(fl.forward_backward(), ref)
        T._check_grads(fl, ref, 2, what="forced %d-row chunks %s" % (rows, cfg_name))
        l_sep = to_np(fl.forward()).copy()
        np.testing.assert_allclose(l_sep, to_np(fl.loss5), rtol=2e-6)
    # the override did take effect: the plan of the first scale has chunks of exactly that height
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import show_plan
    _, scales = show_plan.plan(2, 84, 70, 2, 2)
    assert scales[0]["rows"] == rows, scales
    print("OK rows=%d" % rows)


if __name__ == "__main__":
    main()

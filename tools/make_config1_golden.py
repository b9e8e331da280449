"""Generates tests/golden/config1_oracle.npz: the loss curve and final unseen-view quality of BASELINE config 1's analogue —
1000 Adam iterations of the CPU reference path (torch autograd over oracle/hotpath.py + oracle/training.py) on the closed-form
scene of tests/helpers.py::config1_problem.  The GPU test compares the HIP path's 1000 steps against this fixture instead
of re-running the CPU path on the GPU box (whose host cores are shared and can be 40x slower under load).
usage (build container, CPU): python tools/make_config1_golden.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import helpers  # noqa: E402


def main():
    prob = helpers.config1_problem()
    losses, sd = helpers.config1_oracle_run(prob, log=100)  # (one thread, deterministic algorithms: see the helper)
    psnr, mae, mae_hit = helpers.held_out_quality(prob, sd)
    psnr0, mae0, mae_hit0 = helpers.held_out_quality(prob, prob["sd"])
    out = os.path.join(ROOT, "tests", "golden", "config1_oracle.npz")
    np.savez(out, losses=np.asarray(losses, dtype=np.float64), psnr=psnr, mae=mae, mae_hit=mae_hit, psnr_initial=psnr0, mae_initial=mae0,
             mae_hit_initial=mae_hit0, threads=1, torch_version=np.bytes_(torch.__version__.encode()))
    print("wrote", out, "psnr", psnr, "mae", mae, "last window", float(np.mean(losses[-100:])))


if __name__ == "__main__":
    main()

"""BASELINE config 1 ("1k iters on the CPU PyTorch reference path: plumbing, no GPU") on the CPU oracle.

The reference's double_robot scene is not in the checkout and nerfstudio is not importable, so this is the analogue the
repository can run: 1000 Adam steps (lr 1e-2, eps 1e-15 [REF config_thermal_nerf.py:32-45]) of torch autograd over
oracle/hotpath.py + oracle/training.py — the reference's get_loss_dict [REF thermal_nerf_model.py:277-326], proposal-weight
anneal and proposal update schedule [REF :152-161] — on the closed-form RGB + thermal scene.  tests/test_gpu_training.py
runs the same 1000 steps on the HIP path against this run."""
import numpy as np
import torch

from tests import helpers


def test_config1_one_thousand_cpu_iterations_fit_the_scene(golden_dir):
    torch.set_num_threads(min(8, torch.get_num_threads()))
    prob = helpers.config1_problem()
    assert helpers.CONFIG1["steps"] == 1000
    psnr0, mae0 = helpers.held_out_quality(prob, prob["sd"])
    losses, sd = helpers.config1_oracle_run(prob)
    loss = np.asarray(losses)
    assert np.isfinite(loss).all()
    windows = loss.reshape(10, 100).mean(axis=1)
    # measured (8 / 4 / 3 threads): 0.0545, 0.0115, 0.0080, 0.0040-0.0052, 0.0025-0.0031, 0.0018-0.0022, then 0.001-0.006: with
    # 64-ray batches at a constant lr of 1e-2 the late stage wanders (thread count alone moves the last window by 5x), so the
    # descent is asserted over the first 600 steps and "stays converged" after that
    assert (np.diff(windows[:6]) < 0).all(), windows
    assert windows[5] < 0.06 * windows[0], windows
    assert (windows[6:] < windows[1]).all(), windows
    # the recorded run the GPU test compares the HIP path with (tools/make_config1_golden.py): the same trajectory while
    # rounding has not separated them, the same bands afterwards
    import os

    gold = np.load(os.path.join(golden_dir, "config1_oracle.npz"))
    assert np.abs(loss[:20] - gold["losses"][:20]).max() <= 2e-3 * np.abs(gold["losses"][:20]).max()
    gw = gold["losses"].reshape(10, 100).mean(axis=1)
    assert (np.abs(windows[:6] - gw[:6]) <= 0.5 * gw[:6]).all(), (windows, gw)
    assert (gw[6:] < gw[1]).all()
    psnr1, mae1 = helpers.held_out_quality(prob, sd)
    # (the unseen view is mostly backdrop: its thermal MAE stays near the initial 0.22 and ends anywhere in 0.19 ... 0.225)
    assert abs(psnr1 - float(gold["psnr"])) <= 2.0 and abs(mae1 - float(gold["mae"])) <= 0.05
    assert psnr1 > psnr0 + 2.0, (psnr0, psnr1)            # unseen view: 12.9 -> 15.9 ... 16.2 dB
    assert mae1 <= mae0 + 1e-2, (mae0, mae1)             # 0.220 -> 0.19 ... 0.225 (the view sees mostly backdrop)
    # the sampler's schedule over this horizon: every step below 10, then every second step (update_sched == 1)
    upd = helpers.proposal_updates(1000)
    assert all(upd[:10]) and upd[10:20] == [False, True] * 5 and sum(upd) == 10 + 495

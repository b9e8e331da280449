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
    prob = helpers.config1_problem()
    assert helpers.CONFIG1["steps"] == 1000
    psnr0, mae0, hit0 = helpers.held_out_quality(prob, prob["sd"])
    losses, sd = helpers.config1_oracle_run(prob)  # one thread, deterministic algorithms: reproducible on a host (see the helper)
    loss = np.asarray(losses)
    assert np.isfinite(loss).all()
    windows = loss.reshape(10, 100).mean(axis=1)
    # 100-step means of the recorded run: 0.0545, 0.0116, 0.0077, 0.0037, 0.0023, 0.0018, then 0.0019 ... 0.0030: with 64-ray batches
    # at a constant lr of 1e-2 the late stage wanders, so the descent is asserted over the first 600 steps and "stays converged" after
    assert (np.diff(windows[:6]) < 0).all(), windows
    assert windows[5] < 0.06 * windows[0], windows
    assert (windows[6:] < windows[1]).all(), windows
    # the recorded run the GPU test compares the HIP path with (tools/make_config1_golden.py, same helper): the same trajectory
    # while rounding has not separated them (another host's libm / BLAS kernels may), the same bands afterwards
    import os

    gold = np.load(os.path.join(golden_dir, "config1_oracle.npz"))
    assert np.abs(loss[:20] - gold["losses"][:20]).max() <= 2e-3 * np.abs(gold["losses"][:20]).max()
    # round 6: the fixture's SET of valid fp32 runs (thread counts 1 / 2 / 4 / 8, five orders of the batches' rays) — entry 0 is this
    # very run, and the set's own spread over the descent is what the GPU test's envelopes rest on; `losses_fp64` is the same
    # optimisation in float64: it shares the first steps with every fp32 run and sits at the set's upper edge in windows 2-3
    assert [str(x) for x in gold["run_labels"]][0] == "threads1" and gold["losses_set"].shape == (9, 1000)
    assert np.array_equal(gold["losses_set"][0], gold["losses"])
    ws = gold["losses_set"].reshape(9, 10, 100).mean(axis=2)
    spread = ws.max(axis=0) / ws.min(axis=0)
    assert (spread[:3] < 1.15).all() and (spread[3:6] > 1.15).all() and (spread[3:6] < 1.6).all(), spread
    l64 = gold["losses_fp64"]
    assert l64.shape == (1000,) and np.abs(l64[:20] - gold["losses"][:20]).max() <= 2e-3 * np.abs(l64[:20]).max()
    w64 = l64.reshape(10, 100).mean(axis=1)
    assert (np.abs(ws[:, :3] - w64[:3]) <= 0.08 * w64[:3]).all(), (ws[:, :3], w64[:3])
    gw = gold["losses"].reshape(10, 100).mean(axis=1)
    assert (np.abs(windows[:6] - gw[:6]) <= 0.5 * gw[:6]).all(), (windows, gw)
    assert (gw[6:] < gw[1]).all()
    psnr1, mae1, hit1 = helpers.held_out_quality(prob, sd)
    assert abs(psnr1 - float(gold["psnr"])) <= 1.0 and abs(mae1 - float(gold["mae"])) <= 0.02 and abs(hit1 - float(gold["mae_hit"])) <= 0.02
    # held-out pixels (rays between the training pixels of a training view, 60 % on the sphere): recorded 12.8 -> 16.6 dB,
    # thermal MAE 0.255 -> 0.044 (sphere rays 0.227 -> 0.047).  These fail if the thermal branch does not learn.
    assert psnr1 > psnr0 + 2.5, (psnr0, psnr1)
    assert mae1 < 0.4 * mae0, (mae0, mae1)
    assert hit1 < 0.4 * hit0, (hit0, hit1)
    # the sampler's schedule over this horizon: every step below 10, then every second step (update_sched == 1)
    upd = helpers.proposal_updates(1000)
    assert all(upd[:10]) and upd[10:20] == [False, True] * 5 and sum(upd) == 10 + 495


def test_config1_full_batch_fixture_is_this_oracles_run(golden_dir):
    """tests/golden/config1_batch4096.npz (config 1 at its stated step size: 4096 rays x P=(256,96) x S=48, full-size tables, 30 steps;
    the GPU suite follows it step for step): its first step is reproduced live — the same problem, the same oracle."""
    import os

    gold = np.load(os.path.join(golden_dir, "config1_batch4096.npz"))
    c = helpers.CONFIG1_FULL
    assert gold["losses"].shape == (c["steps"],) and int(gold["rays_per_batch"]) == 4096 and int(gold["S"]) == 48
    assert np.isfinite(gold["losses"]).all() and gold["losses"][-1] < 0.5 * gold["losses"][0]
    prob = helpers.config1_problem(c)
    losses, _ = helpers.config1_oracle_run(prob, steps=1, threads=min(8, torch.get_num_threads()))
    assert abs(losses[0] - gold["losses"][0]) <= 1e-5 * abs(gold["losses"][0]), (losses[0], gold["losses"][0])

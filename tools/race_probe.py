"""Repeat-and-compare probes for stream-ordering bugs (the kind the config-1 statistics found in tn_train_step_fwd): the same work N times,
every result against the first / the per-step median.   usage (GPU box): python tools/race_probe.py [engine|trainer|all] [repeats] [trainer start step] [trainer steps]
  engine : one frame through RayRenderEngine on 1 / 2 / 4 streams with small chunks, `repeats` times each: outputs must be bit-equal
  trainer: `repeats` Trainer runs (HipAdam, default config) of 60 iterations on a SMALL problem (small tables: nothing queued on the side
           streams most of the time, the host far ahead of the device) and on the full tables: per-step losses against the median"""
import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers  # noqa: E402
from thermo_nerf_amd import synthetic  # noqa: E402
from thermo_nerf_amd.engine import RayRenderEngine  # noqa: E402
from thermo_nerf_amd.trainer import RayDataset, Trainer, TrainerConfig  # noqa: E402

DEV = "cuda:0"
what = sys.argv[1] if len(sys.argv) > 1 else "all"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
START = int(sys.argv[3]) if len(sys.argv) > 3 else 0
STEPS = int(sys.argv[4]) if len(sys.argv) > 4 else 60


def engine():
    cm, _, _ = helpers.build("scene", 48, small=False)
    gm = cm.to(DEV).eval()
    o, d = helpers.rays(200, 200, view=1)
    o, d = o.to(DEV), d.to(DEV)
    for streams in (1, 2, 4):
        for chunk in (4096, 9999, 40000):
            eng = RayRenderEngine(gm, chunk=chunk, streams=streams)
            first, bad = None, 0
            for r in range(reps):
                junk = [torch.full((1 << 18,), float("nan"), device=DEV) for _ in range(3)]  # allocator churn between the frames
                del junk
                out = eng.render(o, d)
                cur = {k: v.clone() for k, v in out.items() if torch.is_tensor(v)}
                if first is None:
                    first = cur
                else:
                    bad += any(not torch.equal(cur[k], first[k]) for k in first)
            print(f"engine streams {streams} chunk {chunk}: {bad} of {reps - 1} repeats differ from the first", flush=True)


def trainer():
    V, res = 6, 48
    for small in (True, False):
        runs = []
        for r in range(reps):
            torch.manual_seed(11)
            cm, _, _ = helpers.build("init", 48, small=small, num_images=V, camera_optimizer_mode="SO3xR3")
            cams = synthetic.orbit_cameras(res, res, list(range(V)), num_views=V, elevation_deg=[(-10.0, 20.0, 50.0)[v % 3] for v in range(V)])
            imgs, ths = [], []
            for i in range(V):
                rb = cams.generate_rays(i, device=DEV)
                im, th = synthetic.analytic_scene(rb.origins, rb.directions)
                imgs.append(im)
                ths.append(th)
            ds = RayDataset.from_images(cams, imgs, ths, DEV)
            model = copy.deepcopy(cm).to(DEV)
            tr = Trainer(model, ds, TrainerConfig(train_num_rays_per_batch=256 if small else 1024, seed=0))
            tr.step = START  # 0: the sampler's warm-up (every step updates the proposal networks: per-call path); 5000: five steps in
            losses = []      # six are frozen (the step calls, the deferred table update)
            for _ in range(STEPS):
                loss, _, _ = tr.train_iteration(tr.step)
                tr.step += 1
                losses.append(loss.detach())
            tr.train(0)
            runs.append(torch.stack(losses).cpu().numpy())
        L = np.stack(runs)
        med = np.median(L, axis=0)
        dev = np.abs(L - med) / np.abs(med)
        print(f"trainer small={small}: max relative deviation from the per-step median over {reps} runs, steps 0-9 / 10-19 / 20-29 / 30-59: "
              f"{dev[:, :10].max():.1e} {dev[:, 10:20].max():.1e} {dev[:, 20:30].max():.1e} {dev[:, 30:].max():.1e}; start step {START}; "
              f"runs with a step off by > 10 % in the first 30: {(dev[:, :30] > 0.1).any(axis=1).sum()}", flush=True)


if what in ("engine", "all"):
    engine()
if what in ("trainer", "all"):
    trainer()

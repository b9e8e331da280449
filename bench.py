"""bench.py — rays/sec of the ThermoNeRF rendering hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

Default workload = the configuration BASELINE.json's `metric` is quoted on: a synthetic 800x800 RGB+thermal scene at
192 samples/ray, exact fp32.  A *step* is one pass of the hot path over one frame (640 000 rays, forward-only, eval mode:
proposal sampling 256+96 -> hash-grid+MLP field at S samples/ray -> alpha-composited RGB + thermal + depths), rendered
with eval_num_rays_per_chunk = the frame (one proposal + one field launch per frame).  Rays and weights are resident in
HBM before the timed region.  On rank 0 at N = 1 the line also carries, under `variants`, BASELINE config 2 (S = 64), the
reference config's chunking (eval_num_rays_per_chunk = 65 536, REF config_thermal_nerf.py:30), the opt-in split-precision
(bf16x6, f16x3) and early-termination forms, BASELINE config 4 end to end (the reference's 96-pose 1080p camera path), a
single-GPU proxy of strong scaling over ray shards, BASELINE config 3 as the reference defines it (30 000 consecutive Trainer
iterations on fresh batches, held-out PSNR / thermal MAE, parity on the trained weights) and the fixed-batch training step at
S = 48 and S = 192 — each its own measurement, none of them `value`.

N ranks (--shard weak, default): every rank renders its own frame (view = rank; rays shard with no data-path dependency)
and the rendered pixels (36 B/ray) are all-gathered over RCCL inside the timed region, asynchronously.
--shard frame (BASELINE config 4): ONE 1920x1080 frame, ray-sharded on the reference's chunk boundaries over the N ranks
and all-gathered, every step; `value` = frame rays / frame latency, `scaling` = "strong".

The JSON line also carries
  roofline      the dominant kernel's achieved rate (HIP events on the launch stream inside the timed region) against its
                bound: fp32 MFMA flops for the exact-fp32 field kernel, with the HBM view alongside
  cpu_baseline  the CPU oracle (torch fp32 restatement of the reference path) timed on a bounded sample of the same rays
                on this box's host cores: best torch pool size AND one thread, CPU model stated (rank 0, N = 1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # same guide: v_mfma_f32_32x32x2_f32, dense, f32 in / f32 accumulate (= the FP32 vector rate)
MFMA_16BIT_PEAK_TFLOPS = 2500.0  # same guide: dense bf16 / f16 MFMA (AMD's 5 PF headline includes 2:1 sparsity)
FIELD_FLOPS_PER_SAMPLE = 2 * (32 * 64 + 64 * 16 + 63 * 64 + 64 * 64 + 64 * 3 + 15 * 64 + 64 * 64 + 64 * 1)  # 33 024
PROP_FLOPS_PER_SAMPLE = 2 * (10 * 16 + 16)  # 352
P0, P1 = 256, 96
TORCH_ADAM = False  # --torch-adam: the training variants with torch.optim.Adam(fused=True) and a joined scatter (A/B of round 6's HipAdam + deferred table update)
ATOMIC_SCATTER = False  # --atomic-scatter: config.bucketed_table_scatter = False in the train-step variants (A/B)
REF_CHUNK = 1 << 16  # REF thermo_nerf/thermal_nerf/config_thermal_nerf.py:30


def algorithmic_bytes_per_ray(S: int):
    """SURVEY §8d / BASELINE.md §3: hash-grid corner reads (8 B each) + 36 B in + 36 B out."""
    prop = (P0 + P1) * 5 * 8 * 8
    main = S * 16 * 8 * 8
    return prop, main, prop + main + 72


def algorithmic_flops_per_ray(S: int) -> int:
    return (P0 + P1) * PROP_FLOPS_PER_SAMPLE + FIELD_FLOPS_PER_SAMPLE * S


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks (one per GPU).  Without a launcher (WORLD_SIZE unset) and N > 1, bench.py starts the N ranks itself "
                         "through torch.distributed.run; under a launcher it must equal WORLD_SIZE (default: WORLD_SIZE, else 1)")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--atomic-scatter", action="store_true",
                    help="train steps with config.bucketed_table_scatter = False (global atomics on every level): A/B")
    ap.add_argument("--torch-adam", action="store_true",
                    help="training variants: torch.optim.Adam(fused=True) + joined table scatter instead of HipAdam + deferred table update")
    ap.add_argument("--mode", default="render", choices=["render", "train"],
                    help="render (default): the BASELINE metric.  train: one optimisation step per 'step' "
                         "(BASELINE configs 3/5; with N ranks every rank trains its own scene replica, no collective)")
    ap.add_argument("--ray-batch", default="dataset", choices=["dataset", "patch", "random"],
                    help="--mode train: dataset (default) = BASELINE config 3 as defined: consecutive Trainer iterations, a FRESH "
                         "batch of random pixels per step from the analytic scene's ray table, held-out PSNR / thermal MAE at the "
                         "end; patch / random = ONE fixed batch with random targets re-used every step (one small orbit view = the "
                         "scatter's worst case / random pixels over 8 800x800 views)")
    ap.add_argument("--config3-steps", type=int, default=30000,
                    help="iterations of the train_config3 variant (REF config_thermal_nerf.py:21: 30 000); 0 skips it")
    ap.add_argument("--shard", default="weak", choices=["weak", "frame"],
                    help="N > 1 render: weak = one frame per rank (default); frame = ONE 1920x1080 frame ray-sharded over the "
                         "ranks on chunk boundaries + all-gather (BASELINE config 4, strong scaling)")
    ap.add_argument("--samples", type=int, default=None,
                    help="num_nerf_samples_per_ray: 192 = the metric's configuration (default for render), 64 = config 2, "
                         "48 = the reference default (default for --mode train and --shard frame)")
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--chunk", type=int, default=0,
                    help="rays per launch = eval_num_rays_per_chunk; 0 (default) = the whole frame in one launch pair "
                         "(--shard frame: 65 536, the reference config)")
    ap.add_argument("--weights", default="scene", choices=["init", "stress", "scene"])
    ap.add_argument("--dense-mb", type=int, default=64, help="config.dense_grid_budget_mb (default = the config's default)")
    ap.add_argument("--field-dense-mb", type=int, default=16, help="config.field_dense_grid_budget_mb")
    ap.add_argument("--no-mfma", action="store_true")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16x6", "f16x3"],
                    help="MLP product arithmetic of the field kernel (bf16x6 = six bf16 MFMA products of an exact three-piece split per "
                         "fp32 product; f16x3 = three f16 products of a two-piece split)")
    ap.add_argument("--early-eps", type=float, default=0.0,
                    help="early ray termination threshold (0 = off = the reference's arithmetic; NOT used for `value`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the secondary measurements (profiling runs)")
    ap.add_argument("--cpu-rays", type=int, default=4096, help="rays per CPU-baseline repeat")
    a = ap.parse_args()
    global ATOMIC_SCATTER, TORCH_ADAM
    ATOMIC_SCATTER = bool(a.atomic_scatter)
    TORCH_ADAM = bool(a.torch_adam)
    return a


def cpu_model_name() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform

    return platform.processor() or platform.machine()


def physical_cores() -> int:
    """distinct (socket, core) pairs of /proc/cpuinfo; falls back to half the logical count"""
    try:
        pairs, phys = set(), None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k = k.strip()
            if k == "physical id":
                phys = v.strip()
            elif k == "core id":
                pairs.add((phys, v.strip()))
        if pairs:
            return len(pairs)
    except OSError:
        pass
    return max((os.cpu_count() or 2) // 2, 1)


def cpu_baseline(sd, ocfg, o, d, rays_per_rep: int, S: int):
    """the torch-CPU oracle on a strided sample of the same frame.  `value` is the FASTEST configuration this box
    offers the oracle — torch pool size x rays per oracle call, both probed (its ops are small: a large pool or a large call
    collapses it; round 5 measured 1 048 rays/s at 16 threads x 4 096-ray calls, 1 152 at one thread x 256, 2 493 at 16 x 512 and
    reported the first) — timed over the whole sample; the 4 096-ray-call figure at that pool size, the one-thread figure and the
    all-physical-cores figure are reported beside it, with the CPU model and every probe."""
    from oracle import hotpath as H

    n = min(rays_per_rep, o.shape[0])
    # spread the sample over the frame (every k-th ray) so it is representative of the workload
    idx = torch.linspace(0, o.shape[0] - 1, n).long()
    oc, dc = o[idx].contiguous(), d[idx].contiguous()
    avail = os.cpu_count() or 1
    phys = min(physical_cores(), avail)

    def run(first, count):
        return H.get_outputs(sd, oc[first:first + count], dc[first:first + count], None, ocfg)

    def sweep(call):
        """the whole sample in oracle calls of `call` rays -> (seconds, rgb, thermal)"""
        t = time.perf_counter()
        parts = [run(a, min(call, n - a)) for a in range(0, n, call)]
        return time.perf_counter() - t, torch.cat([p["rgb"] for p in parts]), torch.cat([p["thermal"] for p in parts])

    with torch.no_grad():
        probed = {}  # (threads, rays per call) -> rays/s of one call after a warm-up call at that pool size
        best, cores, call_rays = 0.0, 1, min(512, n)
        for c in [c for c in (1, 8, 16, 32, 64, 128) if c <= avail] or [avail]:
            torch.set_num_threads(c)
            run(0, min(128, n))
            row = 0.0
            for call in [k for k in (256, 512, 1024) if k <= n] or [n]:
                t = time.perf_counter()
                run(n - call, call)
                probed[(c, call)] = call / (time.perf_counter() - t)
                row = max(row, probed[(c, call)])
                if probed[(c, call)] > best:
                    best, cores, call_rays = probed[(c, call)], c, call
            if c > 1 and row < 0.7 * best:
                break  # the pool has collapsed: larger ones are slower still (128 threads: 177-400 rays/s)
        if not any(k[0] == phys for k in probed):  # BASELINE.md §4 asks for the all-physical-cores figure: measured and reported, whatever it is
            torch.set_num_threads(phys)
            run(0, min(128, n))
            t = time.perf_counter()
            run(n - min(512, n), min(512, n))
            probed[(phys, min(512, n))] = min(512, n) / (time.perf_counter() - t)
        phys_key = max((k for k in probed if k[0] == phys), key=lambda k: probed[k])
        # the timed leg: the whole sample in calls of the fastest size at the fastest pool, repeated for >= 10 s
        torch.set_num_threads(cores)
        reps, t_total, rgb, th = 0, 0.0, None, None
        while reps < 1 or (t_total < 10.0 and reps < 10):
            dt, rgb, th = sweep(call_rays)
            t_total += dt
            reps += 1
        # beside it: ONE oracle call over the whole sample at the same pool (what rounds 1-5 reported as `value`) ...
        multi = [c for (c, _) in probed if c > 1]
        torch.set_num_threads(cores if cores > 1 else (min(multi, key=lambda c: abs(c - 16)) if multi else 1))
        big_threads = torch.get_num_threads()
        t = time.perf_counter()
        run(0, n)
        big = n / (time.perf_counter() - t)
        # ... and one thread (BASELINE.md §4 asks for both), at its own best call size
        torch.set_num_threads(1)
        call1 = max((k for k in probed if k[0] == 1), key=lambda k: probed[k])[1] if any(k[0] == 1 for k in probed) else min(256, n)
        n1 = min(2 * call1, n)
        reps1, t1 = 0, 0.0
        while reps1 < 1 or (t1 < 4.0 and reps1 < 4):
            t = time.perf_counter()
            for a in range(0, n1, call1):
                run(a, min(call1, n1 - a))
            t1 += time.perf_counter() - t
            reps1 += 1
        torch.set_num_threads(cores)
    out = {"rgb": rgb, "thermal": th}
    return {"value": n * reps / t_total, "unit": "rays/s", "cores": cores, "call_rays": call_rays, "kind": "port",
            "cpu_model": cpu_model_name(), "host_threads": avail,
            "one_call_of_the_sample": {"value": big, "unit": "rays/s", "cores": big_threads, "call_rays": n,
                                       "sample": "1 x %d rays in ONE oracle call, torch.set_num_threads(%d)" % (n, big_threads)},
            "physical_cores": {"value": probed[phys_key], "unit": "rays/s", "cores": phys, "call_rays": phys_key[1],
                               "sample": "1 x %d rays of the same strided sample, torch.set_num_threads(%d)" % (phys_key[1], phys)},
            "probed_rays_per_s_by_threads_x_call_rays": {"%dx%d" % k: v for k, v in sorted(probed.items())},
            "single_thread": {"value": n1 * reps1 / t1, "unit": "rays/s", "cores": 1, "call_rays": call1,
                              "sample": f"{reps1} x {n1} rays of the same strided sample in calls of {call1}, torch.set_num_threads(1)"},
            "sample": f"{reps} x {n} rays strided over the same 800x800 frame at {S} samples/ray, in oracle calls of {call_rays} rays on "
                      f"{cores} of {avail} host threads (the fastest of the probed threads x call-size grid), torch fp32 CPU oracle"}, idx, out


def load_profile_json(name: str):
    p = os.path.join(ROOT, "profiles", name)
    return json.load(open(p)) if os.path.exists(p) else {}


def train_roofline_from_profiles(roof: dict, samples: int, step_s: float) -> None:
    """What the committed accounting of the step (profiles/train_kernels.json: tools/train_account.sh + tools/line_census.py) adds
    to a training roofline: `traffic` (HBM-side bytes of a whole step from the PMC passes), the critical-path time of every phase
    (a step's kernels run on three streams: no sum of concurrent durations), and the SERIAL-PHASE view — the step is a chain of
    phases that wait on different units, so each is priced against its own bound (forward: line-granular gathers at the measured
    random-line rates; backward: 3 x the MLP flops at the fp32-MFMA peak; scatter: line-atomics at the measured atomic rate +
    the fine levels' records at HBM rate) and the step against their sum.  Every fraction is <= 1 and recomputable from the
    files named in `sources`."""
    prof = load_profile_json("train_kernels.json").get("S%d" % samples)
    if not prof or "critical_path_us_per_step" not in prof:
        return
    roof["traffic"] = prof["hbm_bytes_per_step"]
    roof["traffic_over_algorithmic"] = prof["hbm_bytes_per_step"] / (roof["algorithmic_bytes_per_ray"] * 4096)
    roof["critical_path_us_per_step"] = prof["critical_path_us_per_step"]
    roof["serial_phase_view"] = {
        "phase_bounds_us": prof["phase_bounds_us"], "phase_frac_in_the_profiled_step": prof["phase_frac"],
        "serial_bound_us": prof["serial_bound_us"], "frac": prof["serial_bound_us"] * 1e-6 / step_s,
        "dominant_phase": prof["dominant_phase"], "line_census": prof["line_census"]}
    roof["profile"] = {"period_us": prof["period_us"], "launches_per_step": prof["launches_per_step"], "command": prof["command"],
                       "commit": prof["commit"], "sources": prof["sources"]}


def measure_train_step(dev, samples: int, rays: int = 4096, steps: int = 240, warmup: int = 24, start_step: int = 5000,
                       cpu: bool = True, ray_batch: str = "patch", sustained: bool = False):
    """Secondary measurement (BASELINE configs 3/5): one optimisation step = train-mode forward (tape-free
    final level, config.tape_free_training) + get_metrics_dict/get_loss_dict + backward + Adam(lr 1e-2, eps 1e-15) [REF config_thermal_nerf.py:32-45] on
    `rays` random-target rays, full-size tables, starting at `start_step` (>= proposal_warmup: the proposal networks take
    gradient every 6th step, as in nerfstudio's schedule).  ``ray_batch``: "patch" = the sqrt(rays)^2 image of one orbit view
    (what rounds 1-3 timed: its samples crowd the same table entries, the scatter's worst case); "random" = pixels drawn
    uniformly over all 8 views of an 800x800 orbit, the way nerfstudio's PixelSampler fills a batch."""
    from thermo_nerf_amd import SceneBox, ThermalNerfModel, ThermalNerfModelConfig, synthetic
    from thermo_nerf_amd import training as TR
    from thermo_nerf_amd.rays import RayBundle

    cfg = ThermalNerfModelConfig(num_nerf_samples_per_ray=samples,  # camera_optimizer_mode = SO3xR3, the reference default
                                 bucketed_table_scatter=not ATOMIC_SCATTER)
    model = ThermalNerfModel(cfg, metadata={"thermal": []}, scene_box=SceneBox.unit(), num_train_data=8)
    synthetic.fill_model_(model, "scene")
    sd_cpu = synthetic.model_state_dict_cpu(model) if cpu else None
    model.to(dev).train()
    groups = model.get_param_groups()
    pgroups = [{"params": groups["fields"]}, {"params": groups["proposal_networks"]}, {"params": groups["camera_opt"], "lr": 6e-4}]
    if TORCH_ADAM:  # --torch-adam (A/B): torch's fused Adam on the calling stream, the scatter joined by the backward
        opt = torch.optim.Adam(pgroups, lr=1e-2, eps=1e-15, fused=True)
    else:  # as thermo_nerf_amd.trainer.Trainer: HipAdam, the field's table update deferred to the step's side streams
        from thermo_nerf_amd.optim import HipAdam

        opt = HipAdam(pgroups, lr=1e-2, eps=1e-15, deferred=[model.field.mlp_base.encoder.hash_table])
        model.config.deferred_table_update = True
    g = torch.Generator().manual_seed(0)
    side = int(rays ** 0.5)
    o, d, _ = synthetic.orbit_camera_rays(side, side, view=1)
    o_cpu, d_cpu = o.reshape(-1, 3)[:rays].contiguous(), d.reshape(-1, 3)[:rays].contiguous()
    R = o_cpu.shape[0]
    cam_cpu = torch.randint(0, 8, (R, 1), generator=g)
    batch_cpu = {"image": torch.rand(R, 3, generator=g), "thermal": torch.rand(R, 1, generator=g)}
    if ray_batch == "random":
        o_cpu, d_cpu, cam_cpu = synthetic.random_pixel_rays(rays)
    o, d, cam = o_cpu.to(dev), d_cpu.to(dev), cam_cpu.to(dev)
    batch = {k: v.to(dev) for k, v in batch_cpu.items()}

    def step(i):
        model.set_step(i)
        out = model(RayBundle(origins=o, directions=d, camera_indices=cam))
        loss = TR.total_loss(model.get_loss_dict(out, batch, model.get_metrics_dict(out, batch)))  # as thermo_nerf_amd.trainer.Trainer does
        opt.zero_grad(set_to_none=True)
        TR.backward_total(loss)
        opt.step()

    # (a torch intra-op pool left large by an earlier CPU leg slows the host side of this 77-launch step: timed with the pool at 1)
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    for i in range(warmup):
        step(start_step + i)
    torch.cuda.synchronize()
    # three consecutive windows of steps / 3, each bracketed by a synchronise; ms_per_step = their MEDIAN (the host cores of a
    # GPU box are shared: one slow window no longer decides the line), all three reported
    n_win = 3 if steps >= 30 else 1
    per = max(steps // n_win, 1)
    wins, done = [], 0
    for _ in range(n_win):
        t = time.perf_counter()
        for i in range(per):
            step(start_step + warmup + done + i)
        torch.cuda.synchronize()
        wins.append((time.perf_counter() - t) / per)
        done += per
    steps = done
    # sustained=True: a long run (thousands of steps); the figure is its LAST window — the first few hundred steps of a process are
    # slower on the host (1.4 against 0.9 ms of Python per step until ~500 steps in, tools/train_bench.py --seconds) and on the
    # device (the fit flattens the synthetic fill's field), so a 240-step variant times a transient at S = 48
    dt = wins[-1] if sustained else sorted(wins)[len(wins) // 2]
    torch.set_num_threads(threads)
    _, _, b_all = algorithmic_bytes_per_ray(samples)
    f_all = algorithmic_flops_per_ray(samples)
    # a training step moves ~3x the forward's algorithmic bytes (forward reads + backward read-modify-write
    # of the touched table entries) and does ~3x its MLP flops (forward + dx + dW).  The step is bound by HBM-side atomics
    # and launch chains, so the whole step is priced against both ceilings; the dominant kernel (from the committed rocprof
    # trace of tools/train_bench.py) is named beside it.
    hbm = 3 * b_all * R / dt / 1e9
    tfl = 3 * f_all * R / dt / 1e12
    res = {"what": ("sustained (the last 1 200 of 3 600 consecutive steps on one batch) " if sustained else "") +
                   "train step: forward + losses + backward + Adam, %d rays/step (%s), P=(256,96)+%d samples/ray, "
                   "steps %d.. (proposal nets updated every 6th step), camera optimizer SO3xR3" % (
                       R, "one %dx%d view" % (side, side) if ray_batch == "patch" else "random pixels of 8 800x800 views", samples,
                       start_step),
           "value": R / dt, "unit": "rays/s", "ms_per_step": dt * 1e3, "steps": steps,
           "ms_per_step_windows": [round(w * 1e3, 4) for w in wins],
           "roofline": {"bound": "hbm", "achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_ray": 3 * b_all, "algorithmic_flops_per_ray": 3 * f_all,
                        "mfma_view": {"achieved": tfl, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                      "frac": tfl / MFMA_F32_PEAK_TFLOPS},
                        "kernel": "whole step (all launches between two optimizer steps)"}}
    train_roofline_from_profiles(res["roofline"], samples, dt)
    if cpu:
        from oracle import training as T
        from tests import helpers

        n = 256
        jit = [torch.rand(n, 1, generator=g) for _ in range(3)]
        b = {k: v[:n] for k, v in batch_cpu.items()}
        ocfg = helpers.oracle_config(cfg)
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        T.loss_and_grads(sd_cpu, o_cpu[:64], d_cpu[:64], cam_cpu[:64], {k: v[:64] for k, v in b.items()}, ocfg,
                         [j[:64] for j in jit])
        t = time.perf_counter()
        T.loss_and_grads(sd_cpu, o_cpu[:n], d_cpu[:n], cam_cpu[:n], b, ocfg, jit)
        ct = time.perf_counter() - t
        res["cpu_baseline"] = {"value": n / ct, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
                               "cpu_model": cpu_model_name(),
                               "sample": "1 x %d rays forward+backward (torch autograd over the CPU oracle), no optimizer "
                                         "step" % n}
    from thermo_nerf_amd import _hip

    _hip.join_pending()  # the last step's deferred table update (and the temporaries it holds)
    del model, opt
    torch.cuda.empty_cache()
    return res


CONFIG3_VIEWS = 100  # training views of the analytic room scene (800x800 each, golden-angle spiral over a band of elevations) + 2 held-out views between them
TEMPERATURE_BOUNDS = (33.085, 13.896)  # REF tests/data/thermal/temperature_bounds.json: the fixture's 19.2 degree span


def measure_train_config3(dev, samples: int = 192, steps: int = 30000, window: int = 2500, rays: int = 4096, res: int = 800,
                          budget_s: float = 150.0, oracle_rays: int = 2048, cpu: bool = True):
    """BASELINE config 3 as the reference defines it [REF thermo_nerf/thermal_nerf/config_thermal_nerf.py:17-48]: ``steps``
    CONSECUTIVE iterations of the thermal-nerf method — thermo_nerf_amd.trainer.Trainer (per-group Adam lr 1e-2 / eps 1e-15 with
    nerfstudio's exponential decay to 1e-4 over 200 k steps; SO3xR3 camera optimizer; proposal anneal and update schedule), a
    FRESH batch of ``rays`` random pixels every step (NS PixelSampler: uniform over all training images), P=(256,96)+``samples``
    samples per ray, full-size tables, starting from nerfstudio's initialisation — on a structured scene: the closed-form
    RGB + thermal room scene of thermo_nerf_amd.synthetic (a textured warm/cold sphere inside a textured spherical room: every
    pixel is a point in world space, so parallax pins the geometry as it does for a captured scene — with the direction-only
    backdrop of ``analytic_scene`` the same run parks the density in front of the cameras and held-out views fall to 16 dB,
    tools/config3_fit.py --scene backdrop) seen from CONFIG3_VIEWS cameras at ``res`` x ``res`` (every ray and its target
    pixel resident in HBM before the timed region).  Reports ms/step per ``window`` (does the step slow down as the field
    sharpens?), the sustained rate over the last half, held-out RGB PSNR + thermal MAE (degrees on the reference fixture's span)
    through get_outputs_for_camera_ray_bundle [REF evaluator/evaluator.py:47-106], and the HIP eval render of the trained weights
    against the CPU oracle (the default lane = ray kernels).  ``budget_s``: stop at a window boundary once the timed
    region has used this long (a short GPU lease); the line says how many steps were run."""
    from thermo_nerf_amd import SceneBox, ThermalNerfModel, ThermalNerfModelConfig, synthetic
    from thermo_nerf_amd.cameras import frame_metrics
    from thermo_nerf_amd.trainer import RayDataset, Trainer, TrainerConfig, render_view

    V = CONFIG3_VIEWS
    train_cams = synthetic.spiral_cameras(res, res, V)
    test_cams = synthetic.spiral_cameras(res, res, 2, phase=0.5)

    def truth(cams, i):
        rb = cams.generate_rays(i, device=dev)
        return synthetic.analytic_room_scene(rb.origins, rb.directions)

    imgs, ths = zip(*[truth(train_cams, i) for i in range(V)])
    ds = RayDataset.from_images(train_cams, imgs, ths, dev)
    del imgs, ths
    held = [truth(test_cams, i) for i in range(len(test_cams))]
    cfg = ThermalNerfModelConfig(num_nerf_samples_per_ray=samples, eval_num_rays_per_chunk=res * res,
                                 bucketed_table_scatter=not ATOMIC_SCATTER)
    model = ThermalNerfModel(cfg, metadata={"thermal": []}, scene_box=SceneBox.unit(), num_train_data=V)
    synthetic.fill_model_(model, "init")
    model.to(dev)
    tr = Trainer(model, ds, TrainerConfig(max_num_iterations=steps, train_num_rays_per_batch=rays,
                                          optimizer_impl="torch" if TORCH_ADAM else "hip"))
    t_max, t_min = TEMPERATURE_BOUNDS

    def evaluate():
        rows = [frame_metrics(render_view(model, test_cams, i, dev), held[i][0], held[i][1], t_max, t_min) for i in range(len(test_cams))]
        return {k: sum(r[k] for r in rows) / len(rows) for k in ("psnr", "psnr_thermal", "mae_thermal")}

    q0 = evaluate()
    threads = torch.get_num_threads()
    torch.set_num_threads(1)  # (see measure_train_step)
    tr.train(8)  # untimed: first-use allocations, kernel attribute calls, the caching allocator's growth
    torch.cuda.synchronize()
    windows, done, t0 = [], 8, time.perf_counter()
    prev = t0
    while done < steps:
        n = min(window, steps - done)
        tr.train(n)
        torch.cuda.synchronize()
        now = time.perf_counter()
        windows.append({"steps": [done, done + n], "ms_per_step": (now - prev) / n * 1e3})
        prev, done = now, done + n
        if now - t0 > budget_s:
            break
    torch.set_num_threads(threads)
    half = [w for w in windows if w["steps"][0] >= done // 2] or windows[-1:]
    n_half = sum(w["steps"][1] - w["steps"][0] for w in half)
    ms_half = sum(w["ms_per_step"] * (w["steps"][1] - w["steps"][0]) for w in half) / n_half
    q1 = evaluate()
    _, _, b_all = algorithmic_bytes_per_ray(samples)
    f_all = algorithmic_flops_per_ray(samples)
    res_d = {
        "what": "BASELINE config 3 as defined (REF config_thermal_nerf.py:17-48): %d consecutive Trainer iterations from nerfstudio's "
                "initialisation, a fresh batch of %d random pixels per step over %d views of the analytic RGB+thermal room scene at %dx%d "
                "(%.1f M rays resident in HBM), P=(256,96)+%d samples/ray, full-size tables, SO3xR3 camera optimizer, Adam + "
                "exponential decay" % (done, rays, V, res, res, len(ds) / 1e6, samples),
        "value": rays / (ms_half * 1e-3), "unit": "rays/s", "ms_per_step": ms_half, "steps": done, "steps_requested": steps,
        "sustained_over": "steps %d..%d (the last half)" % (half[0]["steps"][0], done),
        "ms_per_step_by_window": [round(w["ms_per_step"], 4) for w in windows], "window_steps": window,
        "held_out": {"views": len(test_cams), "resolution": [res, res],
                     "rgb_psnr_db": q1["psnr"], "thermal_psnr_db": q1["psnr_thermal"], "thermal_mae_degC": q1["mae_thermal"],
                     "thermal_mae_normalised": q1["mae_thermal"] / (t_max - t_min),
                     "before_training": {"rgb_psnr_db": q0["psnr"], "thermal_mae_degC": q0["mae_thermal"]},
                     "temperature_span_degC": t_max - t_min},
        "roofline": {"bound": "hbm", "achieved": 3 * b_all * rays / (ms_half * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": 3 * b_all * rays / (ms_half * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "mfma_view": {"achieved": 3 * f_all * rays / (ms_half * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": 3 * f_all * rays / (ms_half * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS},
                     "algorithmic_bytes_per_ray": 3 * b_all, "algorithmic_flops_per_ray": 3 * f_all,
                     "kernel": "whole step (all launches between two optimizer steps), last half of the run"}}
    train_roofline_from_profiles(res_d["roofline"], samples, ms_half * 1e-3)
    if cpu:
        # the trained weights through the default eval kernels against the CPU oracle on a strided sample of a held-out frame
        from oracle import hotpath as H
        from tests import helpers

        model.eval()
        rb = test_cams.generate_rays(0, device=dev, flat=True)
        idx = torch.linspace(0, res * res - 1, oracle_rays).long().to(dev)
        with torch.no_grad():
            got = render_view(model, test_cams, 0, dev)
        sd = synthetic.model_state_dict_cpu(model)
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        with torch.no_grad():
            want = H.get_outputs(sd, rb.origins[idx].cpu(), rb.directions[idx].cpu(), None, helpers.oracle_config(cfg),
                                 anneal=float(model.proposal_sampler._anneal))
        torch.set_num_threads(threads)
        e_rgb = float((got["rgb"].reshape(-1, 3)[idx].cpu() - want["rgb"]).abs().mean())
        e_th = float((got["thermal"].reshape(-1, 1)[idx].cpu() - want["thermal"]).abs().mean())
        res_d["trained_weights_parity"] = {
            "what": "HIP eval render (default lane = ray fp32 kernels) of the trained model vs the CPU oracle on the same weights, "
                    "%d rays strided over held-out view 0" % oracle_rays,
            "rgb_mae": e_rgb, "thermal_mae": e_th, "thermal_mae_degC": e_th * (t_max - t_min), "tolerance": 1e-4}
        # the same frame through the opt-in bf16x6 field kernel (exact three-piece split, six products): against the oracle, and
        # against the exact-fp32 kernel's frame
        model.config.mlp_precision = "bf16x6"
        model.invalidate_prepared()
        with torch.no_grad():
            got6 = render_view(model, test_cams, 0, dev)
        model.config.mlp_precision = "f32"
        res_d["trained_weights_parity"]["bf16x6"] = {
            "rgb_mae": float((got6["rgb"].reshape(-1, 3)[idx].cpu() - want["rgb"]).abs().mean()),
            "thermal_mae": float((got6["thermal"].reshape(-1, 1)[idx].cpu() - want["thermal"]).abs().mean()),
            "max_abs_vs_fp32_kernel_rgb": float((got6["rgb"] - got["rgb"]).abs().max()),
            "max_abs_vs_fp32_kernel_thermal": float((got6["thermal"] - got["thermal"]).abs().max()),
            "held_out_rgb_psnr_db": frame_metrics(got6, held[0][0], held[0][1], t_max, t_min)["psnr"],
            "held_out_rgb_psnr_db_fp32_kernel": frame_metrics(got, held[0][0], held[0][1], t_max, t_min)["psnr"]}
    del tr, model, ds
    torch.cuda.empty_cache()
    return res_d


BROADCAST = {}  # N > 1: what distributed.broadcast_model_ moved at the last model build (the line's `rccl.weights_broadcast`)


def build_render(dev, S, chunk, args, want_cpu_sd=False, streams=None):
    """the eval model + engine of one rank.  Under torch.distributed (N > 1) only rank 0 fills the weights; every other rank
    receives them through distributed.broadcast_model_ — the "one broadcast at load" of SURVEY §8e."""
    import torch.distributed as dist

    from thermo_nerf_amd import SceneBox, ThermalNerfModel, ThermalNerfModelConfig, synthetic
    from thermo_nerf_amd.engine import RayRenderEngine

    sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    cfg = ThermalNerfModelConfig(num_nerf_samples_per_ray=S, eval_num_rays_per_chunk=chunk,
                                 dense_grid_budget_mb=args.dense_mb, field_dense_grid_budget_mb=args.field_dense_mb, use_mfma=not args.no_mfma,
                                 mlp_precision=args.precision, early_termination_eps=args.early_eps)
    model = ThermalNerfModel(cfg, metadata={"thermal": []}, scene_box=SceneBox.unit(), num_train_data=8)
    if not sharded or dist.get_rank() == 0:
        synthetic.fill_model_(model, args.weights)  # a counter hash: identical bits on every box
    model.eval()
    sd_cpu = synthetic.model_state_dict_cpu(model) if want_cpu_sd else None
    model = model.to(dev)
    if sharded:
        from thermo_nerf_amd.distributed import broadcast_model_

        torch.cuda.synchronize()
        t = time.perf_counter()
        nbytes = broadcast_model_(model, src=0)
        torch.cuda.synchronize()
        BROADCAST.update(bytes=nbytes, ms=(time.perf_counter() - t) * 1e3)
    return model, cfg, sd_cpu, RayRenderEngine(model, chunk=chunk, streams=streams)


def timed_frames(engine, o, d, out, steps, warmup, after_step=None, barrier=None):
    """W untimed + K timed renders of the resident ray set; returns (elapsed seconds, proposal ms list, field ms list)."""
    def sync():
        if barrier is not None:
            barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        engine.render(o, d, out=out)
        if after_step is not None:
            after_step(out)
    sync()
    engine.timings = []
    t0 = time.perf_counter()
    for _ in range(steps):
        engine.render(o, d, out=out, record_events=True)
        if after_step is not None:
            after_step(out)
    sync()
    elapsed = time.perf_counter() - t0
    prop_ms, main_ms = engine.drain_timings()
    return elapsed, prop_ms, main_ms


def roofline_of(S, n_rays, steps, prop_ms, main_ms, precision, no_mfma, value_per_gpu):
    b_prop, b_main, b_all = algorithmic_bytes_per_ray(S)
    launches = len(main_ms)
    rays_per_launch = n_rays * steps / launches  # a launch processes `chunk` rays (the last chunk of a frame is shorter)
    avg_prop, avg_main = sum(prop_ms) / launches, sum(main_ms) / launches
    dominant = "field_render" if avg_main >= avg_prop else "proposal_sample"
    dom_ms = max(avg_main, avg_prop)
    dom_bytes = (b_main + 36 + 24 + 4 * (S + 1)) if dominant == "field_render" else (b_prop + 24 + 4 * (S + 1) + 8)
    achieved = dom_bytes * rays_per_launch / (dom_ms * 1e-3) / 1e9
    r = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
         "traffic": None, "kernel": dominant, "avg_launch_ms": dom_ms, "rays_per_launch": rays_per_launch,
         "algorithmic_bytes_per_ray": dom_bytes, "proposal_ms": avg_prop, "field_ms": avg_main,
         "path_bytes_per_ray": b_all}
    # the whole path (proposal pass + field pass) against the ceiling that binds it.  For the exact-fp32 kernels that is the
    # fp32-MFMA ceiling (157.3 TFLOP/s / F(S)) at every S: the algorithmic-bytes HBM figure (8 TB/s / B(S)) is a
    # TRAFFIC MODEL, not a ceiling the hardware enforces — ~0.8 of the gathers hit L2 / Infinity Cache (roofline.traffic is ~0.2
    # of the algorithmic bytes), so a path can exceed "8 TB/s of algorithmic bytes" (S=64: 1.05) without skipping work.  It is
    # reported beside the binding fraction, labelled as what it is.
    mfma_ceiling = MFMA_F32_PEAK_TFLOPS * 1e12 / algorithmic_flops_per_ray(S)
    hbm_model = HBM_PEAK_GBS * 1e9 / b_all
    exact = precision == "f32" and not no_mfma
    r["path"] = {"rays_per_s": value_per_gpu, "hbm_algorithmic_traffic_model_rays_per_s": hbm_model,
                 "frac_of_hbm_traffic_model": value_per_gpu / hbm_model}
    if exact:
        r["path"].update({"binding": "mfma", "mfma_ceiling_rays_per_s": mfma_ceiling,
                          "path_frac_of_binding_ceiling": value_per_gpu / mfma_ceiling})
    if dominant == "field_render" and precision == "f32" and not no_mfma:
        # the exact-fp32 field kernel is bound by the matrix pipe, not by HBM (tables sit in L2/MALL; the f32-input MFMA
        # runs at the FP32 vector rate and does not co-execute with VALU work): report THAT roofline and
        # keep the HBM view alongside.  Algorithmic flops = the MLP MACs of the reference's field x 2, per sample.
        tflops = FIELD_FLOPS_PER_SAMPLE * S * rays_per_launch / (dom_ms * 1e-3) / 1e12
        r["hbm"] = {"achieved": r["achieved"], "peak": r["peak"], "unit": "GB/s", "frac": r["frac"]}
        r.update({"bound": "mfma", "achieved": tflops, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                  "frac": tflops / MFMA_F32_PEAK_TFLOPS, "algorithmic_flops_per_ray": FIELD_FLOPS_PER_SAMPLE * S})
    # HBM-side bytes per launch from the committed rocprofv3 PMC pass of this same command (FETCH_SIZE + WRITE_SIZE of the
    # dominant kernel, scaled to this launch size); see profiles/ for the raw counters
    t = load_profile_json("pmc_traffic.json").get("%s@S%d%s" % (dominant, S, "" if precision == "f32" else "_" + precision))
    if t:
        r["traffic"] = (t["fetch_kb"] + t["write_kb"]) * 1024.0 * rays_per_launch / t["rays_per_launch"]
        r["traffic_source"] = t["source"]
    return r


def measure_camera_path(dev, args, S: int = 48):
    """BASELINE config 4 end to end on one GPU [REF thermo_nerf/scripts/render_video_script.py:59-91 -> render/renderer.py:160-201]:
    every pose of the reference's own camera-path fixture (tests/golden/camera_path_facade_2.json: 96 cameras at 1920x1080,
    REF tests/test_renderer.py:65-69) through Cameras.generate_rays (rays made on the device) -> the engine at the reference's
    eval_num_rays_per_chunk = 65 536 -> assembled [H,W,C] frames of all seven outputs in ONE pass (the reference re-renders per
    modality, REF renderer.py:180-183); no image encoding.  Camera centres are scaled into the unit scene box (the path was
    recorded around a real scene; the synthetic weights live in [-1,1]^3)."""
    from thermo_nerf_amd.cameras import get_path_from_json
    from thermo_nerf_amd.rays import RayBundle

    path = os.path.join(ROOT, "tests", "golden", "camera_path_facade_2.json")
    cams = get_path_from_json(json.load(open(path)))
    c2w = cams.camera_to_worlds.clone()
    c2w[:, :3, 3] *= 0.45 / c2w[:, :3, 3].norm(dim=-1).max()
    cams.camera_to_worlds = c2w
    model, cfg, _, _ = build_render(dev, S, REF_CHUNK, args)
    n = cams.height * cams.width

    def frame(i):
        rb = cams.generate_rays(i, device=dev)
        return model.get_outputs_for_camera_ray_bundle(RayBundle(origins=rb.origins, directions=rb.directions))

    frame(0)
    torch.cuda.synchronize()
    per, t0 = [], time.perf_counter()
    for i in range(len(cams)):
        t = time.perf_counter()
        out = frame(i)
        torch.cuda.synchronize()  # a frame is handed on (to the video writer) when it is complete
        per.append(time.perf_counter() - t)
    total = time.perf_counter() - t0
    assert out["rgb"].shape == (cams.height, cams.width, 3) and out["thermal"].shape == (cams.height, cams.width, 1)
    per.sort()
    del model
    torch.cuda.empty_cache()
    return {"what": "config 4 end to end: the %d poses of the reference's camera path at %dx%d, P=(256,96)+%d samples/ray, rays generated "
                    "on the device, eval_num_rays_per_chunk 65 536, all modalities in one pass, frames assembled as [H,W,C], no "
                    "image encoding" % (len(cams), cams.width, cams.height, S),
            "value": n * len(cams) / total, "unit": "rays/s", "frames": len(cams), "rays_per_frame": n, "total_s": total,
            "frame_ms_p50": per[len(per) // 2] * 1e3, "frame_ms_p99": per[min(len(per) - 1, int(0.99 * len(per)))] * 1e3,
            "frame_ms_max": per[-1] * 1e3}


XGMI_LINK_GBS = 153.0  # /opt/skills/guides: 7 point-to-point xGMI links per GPU, ~153 GB/s each


def measure_shard_proxy(dev, args, reps: int = 4):
    """Strong scaling predicted on ONE GPU (the build environment reaches no 8-GPU node): for the metric's 800x800 x S=192 frame
    and BASELINE config 4's 1920x1080 x S=48 frame, render the ray range one rank owns at N = 1, 2, 4, 8 under the fine sharding
    of thermo_nerf_amd.distributed (ray_block: even runs cut on multiples of 64 rays; RayRenderEngine.render_shard: chunk pieces +
    exported depth bounds; apply_depth_bounds) — the first and a middle rank's, the slower of the two counts — and price the
    all-gather of 36 B/ray as a direct exchange over the point-to-point links (each rank sends its shard to every peer on that
    peer's own link: shard bytes / 153 GB/s + 30 us of launch latency; the depth-bound all-reduce is 2 floats per chunk).
    implied_efficiency(N) = T(1) / (N * (T_shard(N) + T_gather(N)))."""
    from thermo_nerf_amd import distributed as D
    from thermo_nerf_amd import synthetic

    res = {}
    for tag, (H_, W_, S) in (("800x800_S192", (800, 800, 192)), ("1080p_S48", (1080, 1920, 48))):
        model, cfg, _, engine = build_render(dev, S, REF_CHUNK, args)
        o3, d3, _ = synthetic.orbit_camera_rays(H_, W_, view=3)
        o, d = o3.reshape(-1, 3).contiguous().to(dev), d3.reshape(-1, 3).contiguous().to(dev)
        n = H_ * W_
        rows, t1 = {}, None
        for N in (1, 2, 4, 8):
            worst = 0.0
            for rank in sorted({0, N // 2}):
                a, b = D.ray_block(n, rank, N)
                oa, da = o[a:b].contiguous(), d[a:b].contiguous()
                out = engine.allocate_outputs(b - a, dev)

                # the segments per tile a frame sharded N ways is rendered with (the same on every rank; 1 at N = 1)
                k_split = engine.shard_sample_split(D.ray_block(n, 0, N)[1]) if N > 1 else 1

                def once(a=a, oa=oa, da=da, out=out, k_split=k_split):
                    _, bounds = engine.render_shard(oa, da, a, n, out=out, sample_split=k_split)
                    engine.apply_depth_bounds(out, a, bounds)

                once()
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(reps):
                    once()
                torch.cuda.synchronize()
                worst = max(worst, (time.perf_counter() - t) / reps)
            shard_rays = D.ray_block(n, 0, N)[1]
            gather = 0.0 if N == 1 else shard_rays * 36 / (XGMI_LINK_GBS * 1e9) + 30e-6
            if N == 1:
                t1 = worst
            rows["N%d" % N] = {"rays_per_rank": shard_rays, "sample_split": k_split, "shard_ms": worst * 1e3, "gather_ms_priced": gather * 1e3,
                               "frame_rays_per_s": n / (worst + gather), "implied_efficiency": t1 / (N * (worst + gather))}
        res[tag] = rows
        del model, engine
        torch.cuda.empty_cache()
    return {"what": "single-GPU proxy of strong scaling: the fine ray shard one rank owns at N = 1, 2, 4, 8 (distributed.ray_block, "
                    "chunk 65 536, one launch pair per shard with per-chunk depth bounds), all-gather priced over the xGMI links; see "
                    "bench.measure_shard_proxy", "frames": res}


def rccl_info(dist, world, dev):
    info = _rccl_info(dist, world, dev)
    if BROADCAST:
        info["weights_broadcast"] = dict(BROADCAST)
    return info


def _rccl_info(dist, world, dev):
    """What the line says about the transport: the backend torch.distributed reports, the world size it sees, every rank's device."""
    name = torch.cuda.get_device_name(dev)
    if world == 1:
        return {"backend": None, "world_size": 1, "devices": [name]}
    names = [None] * world
    dist.all_gather_object(names, "%s (cuda:%d)" % (name, dev.index))
    info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "devices": names}
    try:
        info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:  # pragma: no cover - informational only
        pass
    return info


def measure_sharded_frame(dev, args, world, rank, dist, steps, warmup, S=None, hw=None):
    """BASELINE config 4: ONE 1920x1080 camera-path frame, ray-sharded on the reference's chunk boundaries over the ranks and
    all-gathered inside every step (strong scaling).  Returns the JSON fields on rank 0, None elsewhere."""
    from thermo_nerf_amd import distributed as D
    from thermo_nerf_amd import synthetic

    S = S or args.samples or 48
    H_, W_ = hw or (args.height or 1080, args.width or 1920)
    chunk = args.chunk or REF_CHUNK
    model, cfg, _, engine = build_render(dev, S, chunk, args)
    o3, d3, _ = synthetic.orbit_camera_rays(H_, W_, view=3)
    n_rays = H_ * W_
    if world > 1:  # even shards cut on multiples of 64 rays, not on chunk boundaries (distributed.ray_block)
        r0, r1 = D.ray_block(n_rays, rank, world)
        counts = [D.ray_block(n_rays, r, world)[1] - D.ray_block(n_rays, r, world)[0] for r in range(world)]
    else:
        r0, r1, counts = 0, n_rays, [n_rays]
    o = o3.reshape(-1, 3)[r0:r1].contiguous().to(dev)
    d = d3.reshape(-1, 3)[r0:r1].contiguous().to(dev)
    out = engine.allocate_outputs(max(r1 - r0, 1), dev)
    frame = [None]
    # segments per 64-ray tile of the field pass: what suits ONE rank's run, the same on every rank (distributed.render_frame_sharded_fine)
    k_split = engine.shard_sample_split(counts[0]) if world > 1 else None

    def step():
        if world == 1:
            engine.render(o, d, out=out)
            frame[0] = out
            return
        # this rank's pieces of the reference's chunks; the chunk-wide expected-depth bounds joined by one all-reduce of
        # 2 floats per chunk; then the frame exists (on every rank) once the gather is done: inside the step, not pipelined away
        _, bounds = engine.render_shard(o, d, r0, n_rays, out=out, sample_split=k_split)
        key = torch.stack([bounds[:, 0], -bounds[:, 1]], dim=1).contiguous()
        dist.all_reduce(key, op=dist.ReduceOp.MIN)
        engine.apply_depth_bounds(out, r0, torch.stack([key[:, 0], -key[:, 1]], dim=1).contiguous())
        frame[0] = D.gather_frame({k: v[: r1 - r0] for k, v in out.items()}, H_, W_, counts=counts)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    del model, engine, out
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    _, _, b_all = algorithmic_bytes_per_ray(S)
    value = n_rays * steps / elapsed
    return {
        "metric": "rays/sec (forward-only render, ONE frame ray-sharded over the GPUs) @ %dx%d, %d samples/ray" % (W_, H_, S),
        "value": value, "unit": "rays/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "frame_latency_ms": elapsed / steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "ONE synthetic %dx%d frame, P=(256,96)+%d samples/ray, chunk %d, shards = "
                               "even contiguous ray runs cut on multiples of 64 (rays per rank: %s; %s sample segments per tile; the "
                               "chunk-wide depth bounds all-reduced, 2 floats per chunk), all-gather of 36 B/ray in the timed step" % (
                                   W_, H_, S, chunk, counts if world <= 2 else "%d x ~%d" % (world, counts[0]), k_split or 1),
                   "rays_per_step": n_rays, "parallelism": "one frame ray-sharded x%d + all_gather" % world},
        "roofline": {"bound": "hbm", "achieved": value * b_all / 1e9, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                     "frac": value * b_all / 1e9 / (HBM_PEAK_GBS * world), "traffic": None,
                     "kernel": "whole path, all ranks", "path_bytes_per_ray": b_all}}


LINE_LIMIT = 4096  # bytes: the driver reads the LAST stdout line; round 4's 20.6 KB line did not parse


def _r(x, sig=6):
    """floats to `sig` significant digits (the detail file keeps the full ones); everything else unchanged"""
    if isinstance(x, float):
        return float("%.*g" % (sig, x)) if x == x and abs(x) != float("inf") else None
    return x


def _variant_frac(v: dict):
    """the one fraction a variant is judged by: its roofline's — for a training step SURVEY §8(d)'s "3 x bytes" HBM fraction (the
    MFMA view and the builder's serial-phase number go out under their own keys, compact_line), for a render its path's fraction
    of the binding ceiling"""
    roof = v.get("roofline") or {}
    if "path" in roof and "path_frac_of_binding_ceiling" in roof["path"]:
        return roof["path"]["path_frac_of_binding_ceiling"]
    return roof.get("frac")


def compact_line(full: dict, detail_file: str = "bench_detail.json") -> dict:
    """The ONE line the driver parses: flat, numbers only where numbers do, <= LINE_LIMIT bytes.  Everything `full` holds beyond it
    (the `what` texts, per-window tables, nested samples, censuses) goes to bench_detail.json and to an earlier stdout line.
    tests/test_bench_line.py holds this function to the limit and to a strict-JSON round trip on canned measurements."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")
    line = {k: _r(full[k]) for k in keep if k in full}
    cfg = full.get("config", {})
    line["config"] = {k: cfg[k] for k in ("workload", "rays_per_step_per_gpu", "rays_per_step", "parallelism") if k in cfg}
    line["config"]["workload"] = str(line["config"].get("workload", ""))[:300]
    roof = full.get("roofline") or {}
    r = {k: _r(roof.get(k)) for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "avg_launch_ms", "rays_per_launch",
                                      "algorithmic_flops_per_ray", "algorithmic_bytes_per_ray", "traffic")}
    r["traffic_source"] = str(roof["traffic_source"]).split(" ")[0] if roof.get("traffic_source") else None
    if "path" in roof:
        r["path_frac_of_binding_ceiling"] = _r(roof["path"].get("path_frac_of_binding_ceiling"))
    if "hbm" in roof:
        r["hbm_frac"] = _r(roof["hbm"].get("frac"))
    if "serial_phase_view" in roof:
        r["serial_phase_frac"] = _r(roof["serial_phase_view"].get("frac"))
    for k in ("proposal_ms", "field_ms"):
        if k in roof:
            r[k] = _r(roof[k])
    line["roofline"] = r
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {
            "value": _r(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "cpu_model": cb.get("cpu_model"),
            "call_rays": cb.get("call_rays"), "single_thread": _r((cb.get("single_thread") or {}).get("value")),
            "one_call_4096_rays": _r((cb.get("one_call_of_the_sample") or {}).get("value")),
            "physical_cores": (cb.get("physical_cores") or {}).get("cores"),
            "physical_cores_value": _r((cb.get("physical_cores") or {}).get("value")),
            "sample": str(cb.get("sample", ""))[:160]}
    if "parity" in full:
        line["parity"] = {k: _r(v, 4) for k, v in full["parity"].items() if k in ("rgb_mae", "thermal_mae")}
    if "speedup_vs_cpu" in full:
        line["speedup_vs_cpu"] = _r(full["speedup_vs_cpu"])
    for k in ("held_out", "trained_weights_parity"):  # --mode train lines
        if k in full:
            line[k] = {a: _r(b, 4) for a, b in full[k].items() if isinstance(b, (int, float))}
    if "rccl" in full:
        info = full["rccl"]
        devs = info.get("devices") or []
        line["rccl"] = {"backend": info.get("backend"), "world_size": info.get("world_size"), "rccl_version": info.get("rccl_version"),
                        "devices": sorted({str(d).split(" (cuda")[0] for d in devs}), "device_ids": [
                            int(str(d).split("cuda:")[1].rstrip(")")) for d in devs if "cuda:" in str(d)]}
        if info.get("weights_broadcast"):  # distributed.broadcast_model_ at load: bytes moved from rank 0
            line["rccl"]["weights_broadcast_mb"] = _r(info["weights_broadcast"]["bytes"] / 1e6, 4)
    variants = {}
    for name, v in (full.get("variants") or {}).items():
        if name == "shard_proxy":  # strong scaling predicted on one GPU: efficiency of the N = 8 shard per frame
            for tag, rows in v.get("frames", {}).items():
                variants["shard_proxy_" + tag] = {"n8_shard_ms": _r(rows["N8"]["shard_ms"], 4), "n8_efficiency": _r(rows["N8"]["implied_efficiency"], 4),
                                                  "n4_efficiency": _r(rows["N4"]["implied_efficiency"], 4)}
            continue
        ms = v.get("ms_per_step", v.get("ms_per_frame", v.get("frame_ms_p50")))
        c = {"value": _r(v.get("value"), 5), "ms_per_step": _r(ms, 5)}
        frac = _variant_frac(v)
        if frac is not None:
            c["frac"] = _r(frac, 4)
        roof_v = v.get("roofline") or {}
        if "mfma_view" in roof_v:  # training steps: SURVEY 8(d) "3 x flops" against the fp32-MFMA peak, beside the HBM `frac`
            c["mfma_frac"] = _r(roof_v["mfma_view"].get("frac"), 4)
        if "serial_phase_view" in roof_v:  # phase-by-phase view against measured line rates (profiles/train_kernels.json): not a roofline fraction
            c["serial_phase_frac"] = _r(roof_v["serial_phase_view"].get("frac"), 4)
        if "held_out" in v:  # config 3's quantities: sustained step, held-out quality, parity on the trained weights
            c["steps"] = v.get("steps")
            c["rgb_psnr_db"] = _r(v["held_out"].get("rgb_psnr_db"), 4)
            c["thermal_mae_degC"] = _r(v["held_out"].get("thermal_mae_degC"), 4)
            if "trained_weights_parity" in v:
                c["parity_rgb_mae"] = _r(v["trained_weights_parity"].get("rgb_mae"), 3)
                c["parity_thermal_mae"] = _r(v["trained_weights_parity"].get("thermal_mae"), 3)
        for k in ("rgb_mae", "thermal_mae"):
            if k in v:
                c[k] = _r(v[k], 3)
        if "n_gpus" in v:  # the strong-scaling frame measured by the same ranks (N > 1)
            c["n_gpus"], c["scaling"] = v["n_gpus"], v.get("scaling")
        variants[name] = c
    if variants:
        line["variants"] = variants
    line["detail"] = detail_file
    # the limit is a contract, not a hope: shed the optional parts, least important first, until the line fits
    def fits():
        return len(json.dumps(line, allow_nan=False)) <= LINE_LIMIT

    def shed_variant_fields():
        line["variants"] = {k: {"value": c.get("value")} for k, c in line.get("variants", {}).items()}

    for shed in (lambda: line.get("rccl", {}).pop("device_ids", None), lambda: line.get("cpu_baseline", {}).pop("sample", None),
                 lambda: line["config"].update(workload=line["config"]["workload"][:120]), shed_variant_fields,
                 lambda: line.update(variants={"dropped": len(line.get("variants", {}))})):
        if fits():
            break
        shed()
    return line


def _strict(x):
    """NaN / +-inf are not JSON: null them, recursively (a consumer's strict parser must never be the thing that fails)"""
    if isinstance(x, float):
        return x if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {str(k): _strict(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_strict(v) for v in x]
    return x


def emit(full: dict) -> None:
    """bench_detail.json (next to bench.py and, on a gpurun box, under gpurun_out/ so that it travels back) + the full tree as an
    EARLIER stdout line + the compact line LAST."""
    full = _strict(full)
    text = json.dumps(full, allow_nan=False)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_detail.json"), "w") as f:
                    f.write(text + "\n")
            except OSError:
                pass
    print(json.dumps({"bench_detail": full}, allow_nan=False), flush=True)
    line = compact_line(full)
    out = json.dumps(line, allow_nan=False)
    assert len(out) <= LINE_LIMIT and json.loads(out) == line, len(out)
    print(out, flush=True)


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks through torch.distributed.run (one process per GPU, rendezvous
    on 127.0.0.1 at a free port) with this very command line; rank 0 prints the one line, the children's stdout / stderr pass through."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def resolve_world(args) -> int:
    """--gpus against the launcher's WORLD_SIZE: equal, or one of them absent.  Returns the world size of THIS process, or -1 when
    the ranks have to be started first (launch_ranks); a contradiction is an error, never a silent one-GPU run."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None:
        if args.gpus is None or args.gpus <= 1:
            args.gpus = 1
            return 1
        return -1
    world = int(env_world)
    if args.gpus is not None and args.gpus != world:
        raise SystemExit("bench.py: --gpus %d contradicts the launcher's WORLD_SIZE=%d" % (args.gpus, world))
    args.gpus = world
    return world


def main():
    args = parse()
    world = resolve_world(args)
    if world < 0:
        sys.exit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        # TN_BENCH_BACKEND=gloo: exercise the N > 1 code path with several ranks sharing one GPU (RCCL refuses that);
        # the default is "nccl" = RCCL, one GPU per rank
        backend = os.environ.get("TN_BENCH_BACKEND", "nccl")
        if backend != "nccl":
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    assert torch.cuda.is_available(), "bench.py needs a ROCm GPU (no CPU fallback exists in thermo_nerf_amd)"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    solo = rank == 0 and world == 1

    if args.mode == "train":
        # per-GPU scene assignment (BASELINE config 5): independent replicas, no data-path collective; the timed region is
        # bracketed by barriers like the render mode and `value` is all ranks' rays over the slowest rank's time
        if world > 1:
            dist.barrier()
        if args.ray_batch == "dataset":
            res = measure_train_config3(dev, args.samples or 192, steps=max(args.steps, 16), window=max(min(2500, args.steps // 4), 8),
                                        cpu=(solo and not args.no_cpu_baseline), budget_s=600.0)
        else:
            res = measure_train_step(dev, args.samples or 48, steps=max(args.steps, 1), warmup=max(args.warmup, 1),
                                     cpu=(solo and not args.no_cpu_baseline), ray_batch=args.ray_batch)
        t = torch.tensor([res["ms_per_step"]], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        info = rccl_info(dist, world, dev)
        if rank == 0:
            ms = float(t.item())
            line = {"metric": "rays/sec (train step: forward + losses + backward + Adam) @ 4096 rays/step", "value": world * 4096 / (ms * 1e-3),
                    "rccl": info, **{k: res[k] for k in ("held_out", "ms_per_step_by_window", "window_steps", "sustained_over",
                                                         "trained_weights_parity") if k in res},
                    "unit": "rays/s", "n_gpus": world, "steps": res["steps"], "warmup": max(args.warmup, 1), "ms_per_step": ms,
                    "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                    "config": {"workload": res["what"], "parallelism": "scene replica per rank x%d, no collective" % world},
                    "roofline": res["roofline"]}
            if "cpu_baseline" in res:
                line["cpu_baseline"] = res["cpu_baseline"]
            emit(line)
        if world > 1:
            dist.destroy_process_group()
        return

    from thermo_nerf_amd import synthetic

    if args.shard == "frame":
        line = measure_sharded_frame(dev, args, world, rank, dist, args.steps, args.warmup)
        info = rccl_info(dist, world, dev)
        if rank == 0:
            line["rccl"] = info
            emit(line)
        if world > 1:
            dist.destroy_process_group()
        return

    S = args.samples or 192
    H_, W_ = args.height or 800, args.width or 800
    if args.chunk <= 0:
        args.chunk = H_ * W_
    model, cfg, sd_cpu, engine = build_render(dev, S, args.chunk, args, want_cpu_sd=(solo and not args.no_cpu_baseline))
    o_cpu, d_cpu, _ = synthetic.orbit_camera_rays(H_, W_, view=rank % 8)
    o_cpu, d_cpu = o_cpu.reshape(-1, 3).contiguous(), d_cpu.reshape(-1, 3).contiguous()
    o, d = o_cpu.to(dev), d_cpu.to(dev)
    n_rays = o.shape[0]
    out = engine.allocate_outputs(n_rays, dev)
    pipe = None
    if world > 1:
        from thermo_nerf_amd.distributed import PipelinedFrameGather

        pipe = PipelinedFrameGather(n_rays, world, dev)

    def barrier():
        if pipe is not None:
            pipe.finish()  # every gather of the timed region completes inside it
            dist.barrier()

    # the exchange step of the path (N > 1): rendered pixels (9 floats = 36 B per ray) all-gathered over RCCL/xGMI,
    # asynchronously: the next frame renders while this one is on the links (double-buffered)
    elapsed, prop_ms, main_ms = timed_frames(engine, o, d, out, args.steps, args.warmup,
                                             after_step=(pipe.submit if pipe is not None else None), barrier=barrier)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    info = rccl_info(dist, world, dev)
    strong = {}
    if world > 1 and not args.no_variants:
        # the same ranks on ONE frame (strong scaling): BASELINE config 4's 1920x1080 x S=48 frame and the metric's own 800x800 x
        # S=192 frame — a driver that runs `bench.py --gpus N` sees the weak curve (`value`) and both strong ones
        saved = args.chunk
        args.chunk = 0
        strong["strong_frame_1080p_S48"] = measure_sharded_frame(dev, args, world, rank, dist, max(3, args.steps // 2), 1, S=48, hw=(1080, 1920))
        strong["strong_frame_800_S192"] = measure_sharded_frame(dev, args, world, rank, dist, max(3, args.steps // 2), 1, S=192, hw=(800, 800))
        args.chunk = saved

    if rank == 0:
        value = world * n_rays * args.steps / elapsed
        line = {
            "metric": "rays/sec (forward-only render) @ %dx%d, %d samples/ray" % (W_, H_, S),
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"f32": "f32", "f16x3": "f32 via 3xf16-split MFMA products", "bf16x6": "f32 via 6xbf16-split MFMA products"}[args.precision],
            "data": "synthetic",
            "config": {"workload": "%s: synthetic %dx%d RGB+thermal scene, P=(256,96)+%d samples/ray, chunk %d, "
                                   "forward-only eval, %s weights%s" % (
                                       "BASELINE metric configuration (configs[2]'s 192 samples/ray on the 800x800 frame of configs[1])"
                                       if S == 192 else ("config 2" if S == 64 else "custom"),
                                       W_, H_, S, args.chunk, args.weights,
                                       "" if args.early_eps <= 0 else ", early ray termination eps=%g" % args.early_eps),
                       "rays_per_step_per_gpu": n_rays, "parallelism": "ray-shard x%d (one frame per rank)" % world},
            "roofline": roofline_of(S, n_rays, args.steps, prop_ms, main_ms, args.precision, args.no_mfma, value / world),
            "rccl": info,
        }
        if strong:
            line["variants"] = dict(strong)
        # every GPU measurement first, the CPU oracle last: its torch thread pools (probed up to 128 threads) slow the host side
        # of the 77-launch training step by 30-40 % for the rest of the process (2.65 against 1.83 ms at S=48, same kernels).
        # The GPU frames are sampled now on the rays the baseline will time (a strided sample) and compared afterwards.
        captured = {}
        gidx = None
        if sd_cpu is not None:
            gidx = torch.linspace(0, n_rays - 1, min(args.cpu_rays, n_rays)).long().to(dev)

        def capture(tag, o_):
            if gidx is not None:
                captured[tag] = (o_["rgb"][gidx].cpu(), o_["thermal"][gidx].cpu())

        capture("main", out)
        if solo and not args.no_variants and args.precision == "f32" and args.early_eps <= 0:
            variants = {}

            def quick(eng, reps=3):
                eng.render(o, d, out=out)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(reps):
                    eng.render(o, d, out=out)
                torch.cuda.synchronize()
                return reps * n_rays / (time.perf_counter() - t1)

            # the opt-in split-precision field kernels on the same frame (NOT the headline value)
            for prec, what in (("bf16x6", "field MLP products as 6 bf16 MFMA products of an exact three-piece (24-bit) operand split each, "
                                          "fp32 accumulate: 2^-23 relative per product, fp32's own rounding size (mlp_precision=bf16x6)"),
                               ("f16x3", "field MLP products as 3 f16 MFMA products of a two-piece (22-bit) split each, fp32 accumulate "
                                         "(mlp_precision=f16x3)")):
                model.config.mlp_precision = prec
                quick(engine, 1)
                e_p, p_p, m_p = timed_frames(engine, o, d, out, 4, 1)
                products = {"bf16x6": 6, "f16x3": 3}[prec]
                f_ms = sum(m_p) / len(m_p)
                tfl = products * FIELD_FLOPS_PER_SAMPLE * S * n_rays / (f_ms * 1e-3) / 1e12
                variants[prec] = {"what": what + ", same %d-sample frame" % S, "value": n_rays * 4 / e_p, "unit": "rays/s",
                                  "ms_per_step": e_p / 4 * 1e3, "proposal_ms": sum(p_p) / len(p_p), "field_ms": f_ms,
                                  "roofline": {"bound": "mfma", "unit": "TFLOP/s", "peak": MFMA_16BIT_PEAK_TFLOPS,
                                               "achieved": tfl, "frac": tfl / MFMA_16BIT_PEAK_TFLOPS,
                                               "what": "16-bit MFMA flops actually issued by the field kernel: %d piece products per fp32 "
                                                       "product x the field's %d MLP flops per sample, against the dense bf16 / f16 MFMA peak; "
                                                       "in fp32-equivalent flops: %.1f TFLOP/s (the exact-fp32 MFMA peak is %.1f)" % (
                                                           products, FIELD_FLOPS_PER_SAMPLE, tfl / products, MFMA_F32_PEAK_TFLOPS)}}
                capture(prec, out)
            model.config.mlp_precision = "f32"
            # opt-in early ray termination (wave-wide transmittance vote), exact-fp32 kernels; outputs move by <= eps
            engine.rc.early_stop_transmittance = 1e-3
            v = quick(engine)
            variants["early_termination_1e-3"] = {
                "what": "early_termination_eps=1e-3 (a 64-ray tile stops once every ray's transmittance is below it), same "
                        "%d-sample frame" % S, "value": v, "unit": "rays/s", "ms_per_step": n_rays / v * 1e3}
            capture("early_termination_1e-3", out)
            engine.rc.early_stop_transmittance = 0.0
            # the reference config's chunking: eval_num_rays_per_chunk = 65 536 (two HIP streams alternate the chunks)
            from thermo_nerf_amd.engine import RayRenderEngine

            for tag, fuse in (("reference_chunking_65536", True), ("reference_chunking_65536_launch_per_chunk", False)):
                eng_c = RayRenderEngine(model, chunk=REF_CHUNK, fuse_chunks=fuse)
                e_c, p_c, m_c = timed_frames(eng_c, o, d, out, max(3, args.steps // 2), 1)
                variants[tag] = {
                    "what": "same frame at eval_num_rays_per_chunk = 65 536 (REF config_thermal_nerf.py:30): " + (
                        "the engine's default — the frame's chunks in one launch pair that keeps one expected-depth bound pair per "
                        "chunk (tn_field_render_chunked_fwd): the chunk-by-chunk result bit for bit" if fuse else
                        "one launch pair per chunk, chunks alternating over %d HIP streams (fuse_chunks=False)" % eng_c.num_streams),
                    "value": n_rays * max(3, args.steps // 2) / e_c, "unit": "rays/s",
                    "ms_per_frame": e_c / max(3, args.steps // 2) * 1e3, "launch_pairs_per_frame": len(m_c) // max(3, args.steps // 2)}
            del eng_c
            if S != 64:
                # BASELINE config 2 (64 samples/ray) on the same frame and rays, with its own roofline
                del engine
                torch.cuda.empty_cache()
                m64, _, _, e64 = build_render(dev, 64, args.chunk, args)
                k64 = max(5, args.steps)
                el, p64, f64 = timed_frames(e64, o, d, out, k64, 2)
                v64 = n_rays * k64 / el
                variants["config2_S64"] = {"what": "BASELINE config 2: same 800x800 frame at 64 samples/ray, one launch pair per frame",
                                           "value": v64, "unit": "rays/s", "ms_per_step": el / k64 * 1e3, "steps": k64,
                                           "roofline": roofline_of(64, n_rays, k64, p64, f64, "f32", args.no_mfma, v64)}
                del m64, e64
            del model, out
            torch.cuda.empty_cache()
            cpu_train = sd_cpu is not None
            variants["camera_path_1080p_S48"] = measure_camera_path(dev, args)
            variants["shard_proxy"] = measure_shard_proxy(dev, args)
            if args.config3_steps > 0:  # BASELINE config 3 as the reference defines it (fresh batches, consecutive steps, quality)
                variants["train_config3_S192"] = measure_train_config3(dev, 192, steps=args.config3_steps, cpu=cpu_train)
            variants["train_step_S192"] = measure_train_step(dev, 192, cpu=False)  # (before S48's CPU leg, for the same reason)
            variants["train_step_S192_random_pixels"] = measure_train_step(dev, 192, cpu=False, ray_batch="random")
            # (the long S = 48 run first: the first few hundred S = 48 steps of a process are slower on the host and in the caching
            # allocator — windows of 1.9 / 1.6 / 1.2 ms in a 240-step variant measured cold — and the sustained run reports its LAST third)
            variants["train_sustained_S48_random_pixels"] = measure_train_step(dev, 48, cpu=False, ray_batch="random", steps=3600, sustained=True)
            variants["train_step_S48_random_pixels"] = measure_train_step(dev, 48, cpu=False, ray_batch="random")
            variants["train_step_S48"] = measure_train_step(dev, 48, cpu=cpu_train)
            line["variants"] = variants
        if sd_cpu is not None:
            from tests import helpers  # oracle-side plumbing: only imported on the cpu_baseline leg

            ocfg = helpers.oracle_config(cfg)
            base, idx, want = cpu_baseline(sd_cpu, ocfg, o_cpu, d_cpu, args.cpu_rays, S)
            assert torch.equal(idx, gidx.cpu())
            line["cpu_baseline"] = base
            # matched quality: the GPU frame vs the oracle on the very rays the baseline timed.  NOTE the oracle
            # clips expected_depth per call, the engine per chunk; rgb/thermal are chunk-independent.

            def err(tag):
                g_rgb, g_th = captured[tag]
                return float((g_rgb - want["rgb"]).abs().mean()), float((g_th - want["thermal"]).abs().mean()), g_rgb

            rgb_mae, th_mae, got_rgb = err("main")
            line["parity"] = {"rgb_mae": rgb_mae, "thermal_mae": th_mae,
                              "rgb_psnr_db_vs_oracle": float(10 * torch.log10(1.0 / ((got_rgb - want["rgb"]) ** 2).mean().clamp_min(1e-20)))}
            line["speedup_vs_cpu"] = value / base["value"]
            for tag in ("bf16x6", "f16x3", "early_termination_1e-3"):
                if tag in captured and "variants" in line and tag in line["variants"]:
                    line["variants"][tag]["rgb_mae"], line["variants"][tag]["thermal_mae"], _ = err(tag)
        emit(line)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

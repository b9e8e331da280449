"""Time the training step (forward + losses + backward + Adam) on the GPU.
usage: python tools/train_bench.py [--rays 4096] [--samples 48] [--steps 20 | --seconds 20] [--taped]
--seconds T: run consecutive steps for T seconds (BASELINE config 3 is 30 000 consecutive steps: the sustained figure is the
honest one), report the mean over the first and the second half and the GPU's sclk / power (rocm-smi) before and after."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thermo_nerf_amd import SceneBox, ThermalNerfModel, ThermalNerfModelConfig, synthetic  # noqa: E402
from thermo_nerf_amd import training as TR  # noqa: E402
from thermo_nerf_amd.rays import RayBundle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=48)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--weights", default="scene")
    ap.add_argument("--seconds", type=float, default=0.0, help="sustained run: consecutive steps for this long (overrides --steps)")
    ap.add_argument("--taped", action="store_true", help="config.tape_free_training = False (the round-2 taped forward + chained backward)")
    ap.add_argument("--one-launch", action="store_true", help="config.fused_backward_split = False")
    ap.add_argument("--no-spread", action="store_true", help="config.spread_coarse_scatter = False")
    ap.add_argument("--no-overlap", action="store_true", help="config.overlap_table_scatter = False")
    ap.add_argument("--no-reg-overlap", action="store_true", help="config.overlap_regularisers = False")
    ap.add_argument("--no-opt", action="store_true")
    ap.add_argument("--atomic-scatter", action="store_true", help="config.bucketed_table_scatter = False (global atomics on every level)")
    ap.add_argument("--first-sorted-level", type=int, default=-1, help="config.bucketed_table_scatter = this level (records from it on, "
                    "atomics below) instead of the library's choice")
    ap.add_argument("--sort-rays", action="store_true", help="order the batch's rays by (camera, Morton code of the direction)")
    ap.add_argument("--camera-opt", default="SO3xR3", choices=["off", "SO3xR3"], help="reference default: SO3xR3")
    ap.add_argument("--ray-batch", default="patch", choices=["patch", "random"],
                    help="patch: the sqrt(rays)^2 image of one orbit view (rounds 1-3); random: pixels drawn uniformly over all 8 "
                         "views of an 800x800 orbit, the way nerfstudio's PixelSampler fills a batch")
    ap.add_argument("--kick-ms", type=float, default=0.0,
                    help="probe: queue a spin kernel of this many ms before the timed steps, so that the host starts AHEAD of the device "
                         "(is the slow start of a run a host-bound mode that sustains itself?)")
    ap.add_argument("--no-store-base", action="store_true", help="config.store_base_output = False (the head launches recompute mlp_base)")
    ap.add_argument("--f32-backward", action="store_true", help="config.backward_bf16_pieces = False")
    ap.add_argument("--no-jacobian", action="store_true", help="config.store_position_jacobian = False")
    ap.add_argument("--update-every", type=int, default=5, help="config.proposal_update_every (5 = the reference: the proposal networks take "
                    "gradient on every 6th step after warm-up; 0: on every step; 1000000: never in a run)")
    ap.add_argument("--high-priority-main", action="store_true", help="run the step on a high-priority stream (the step's side "
                    "streams come from the default-priority pool): main-chain kernels are dispatched ahead of the side streams'")
    ap.add_argument("--start-step", type=int, default=5000,
                    help="training step the timed region starts at (>= proposal_warmup: proposal nets update every 6th step)")
    ap.add_argument("--side-priority", type=int, default=0, help="priority of the step's side streams (-1 = high)")
    ap.add_argument("--per-call", action="store_true", help="config.fused_step_calls = False: one ctypes call per launch (rounds 1-5)")
    ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam(fused=True) + joined table scatter (rounds 1-5)")
    ap.add_argument("--joined-table", action="store_true", help="HipAdam, but the table's scatter and Adam joined by the backward / on the calling stream")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    from thermo_nerf_amd import _hip as _H

    _H.SIDE_STREAM_PRIORITY = a.side_priority
    cfg = ThermalNerfModelConfig(num_nerf_samples_per_ray=a.samples, camera_optimizer_mode=a.camera_opt,
                                 bucketed_table_scatter=(a.first_sorted_level if a.first_sorted_level >= 0 else not a.atomic_scatter), tape_free_training=not a.taped,
                                 fused_backward_split=not a.one_launch, spread_coarse_scatter=not a.no_spread,
                                 overlap_table_scatter=not a.no_overlap, overlap_regularisers=(False if a.no_reg_overlap else "auto"),
                                 store_base_output=not a.no_store_base, backward_bf16_pieces=not a.f32_backward,
                                 store_position_jacobian=not a.no_jacobian, proposal_update_every=a.update_every)
    model = ThermalNerfModel(cfg, metadata={"thermal": []}, scene_box=SceneBox.unit(), num_train_data=8)
    synthetic.fill_model_(model, a.weights)
    model.to(dev).train()
    model.config.fused_step_calls = not a.per_call
    groups = model.get_param_groups()
    pg = [{"params": groups["fields"]}, {"params": groups["proposal_networks"]}]
    if "camera_opt" in groups:
        pg.append({"params": groups["camera_opt"], "lr": 6e-4})
    if a.torch_adam:
        opt = torch.optim.Adam(pg, lr=1e-2, eps=1e-15, fused=True)
    else:  # as thermo_nerf_amd.trainer.Trainer: HipAdam, the field's table update left on the step's side streams (round 6)
        from thermo_nerf_amd.optim import HipAdam

        opt = HipAdam(pg, lr=1e-2, eps=1e-15, deferred=[] if a.joined_table else [model.field.mlp_base.encoder.hash_table])
        model.config.deferred_table_update = not a.joined_table
    g = torch.Generator().manual_seed(0)
    side = int(a.rays ** 0.5)
    o, d, _ = synthetic.orbit_camera_rays(side, side, view=1)
    o, d = o.reshape(-1, 3)[: a.rays].contiguous().to(dev), d.reshape(-1, 3)[: a.rays].contiguous().to(dev)
    R = o.shape[0]
    cam = torch.randint(0, 8, (R, 1), generator=g).to(dev)
    if a.ray_batch == "random":
        o, d, cam = (t.to(dev) for t in synthetic.random_pixel_rays(a.rays))
        R = o.shape[0]
    batch = {"image": torch.rand(R, 3, generator=g).to(dev), "thermal": torch.rand(R, 1, generator=g).to(dev)}
    if a.sort_rays:
        # the same batch with its rays ordered by (camera, Morton code of the direction): neighbours in the launch are neighbours in
        # the image (the order of a batch's rays carries no meaning: every loss is a mean over them)
        q = ((d * 0.5 + 0.5).clamp(0, 1) * 1023).long()

        def spread(x):
            x = (x | (x << 16)) & 0x30000FF
            x = (x | (x << 8)) & 0x300F00F
            x = (x | (x << 4)) & 0x30C30C3
            return (x | (x << 2)) & 0x9249249

        key = (cam.view(-1).long() << 30) | spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
        order = torch.argsort(key)
        o, d, cam = o[order].contiguous(), d[order].contiguous(), cam[order].contiguous()
        batch = {k: v[order].contiguous() for k, v in batch.items()}

    phase = [0.0] * 5

    def step(i):
        t0 = time.perf_counter()
        model.set_step(i)
        rb = RayBundle(origins=o, directions=d, camera_indices=cam)
        out = model(rb)
        t1 = time.perf_counter()
        metrics = model.get_metrics_dict(out, batch)
        loss = TR.total_loss(model.get_loss_dict(out, batch, metrics))  # as thermo_nerf_amd.trainer.Trainer does
        t2 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        TR.backward_total(loss)
        t3 = time.perf_counter()
        if not a.no_opt:
            opt.step()
        t4 = time.perf_counter()
        for k, dt in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            phase[k] += dt
        return loss

    def smi():
        """sclk (MHz) and socket power (W) as rocm-smi reports them; None when the tool is not there"""
        import json
        import subprocess

        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20)
            card = next(iter(json.loads(r.stdout).values()))
            sclk = next((v for k, v in card.items() if "sclk" in k.lower()), None)
            power = next((v for k, v in card.items() if "power" in k.lower() and "(w)" in k.lower()), None)
            return f"sclk {sclk}, power {power} W"
        except Exception as e:  # pragma: no cover - informational
            return f"rocm-smi unavailable ({type(e).__name__})"

    if a.high_priority_main:
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
    for i in range(a.warmup):
        step(a.start_step + i)
    torch.cuda.synchronize()
    if a.kick_ms > 0:
        torch.cuda._sleep(int(a.kick_ms * 2.4e6))
    if a.seconds > 0:
        print("before:", smi(), flush=True)
        marks, i, t0 = [], 0, time.perf_counter()
        while True:
            for _ in range(100):
                step(a.start_step + a.warmup + i)
                i += 1
            issued = time.perf_counter() - t0  # the host has queued the window; what is left until the sync is the device's backlog
            torch.cuda.synchronize()
            marks.append((i, time.perf_counter() - t0, issued, tuple(phase)))
            phase[:] = [0.0] * 5
            if marks[-1][1] >= a.seconds:
                break
        print("after: ", smi(), flush=True)
        prev = 0.0
        per = []
        for _, t_end, t_issued, ph in marks:
            per.append(f"{(t_end - prev) * 10:.2f}({(t_end - t_issued) * 1e3:.1f}|" + "/".join(f"{x * 10:.2f}" for x in ph[:4]) + ")")
            prev = t_end
        print("ms/step per 100-step window (device backlog in ms when the host finished queueing it | host ms per step in forward / losses / backward / optimizer):", " ".join(per), flush=True)
        half = next(k for k, m in enumerate(marks) if m[1] >= marks[-1][1] / 2)
        n1, t1 = marks[half][:2]
        n2, t2 = marks[-1][:2]
        first, second = t1 / n1, (t2 - t1) / max(n2 - n1, 1)
        print(f"rays {R} S {a.samples}: sustained {n2} consecutive steps in {t2:.1f} s: first half {first * 1e3:.3f} ms/step, "
              f"second half {second * 1e3:.3f} ms/step ({R / second / 1e6:.3f} M rays/s)")
        return
    t = time.perf_counter()
    for i in range(a.steps):
        loss = step(a.start_step + a.warmup + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / a.steps
    print(f"rays {R} S {a.samples}: {dt * 1e3:.3f} ms/step, {R / dt / 1e6:.3f} M rays/s, loss {loss.item():.5f}")


if __name__ == "__main__":
    main()

"""Cache-line census of one training batch's final-level samples: how many distinct 64-byte table lines the hash-grid gathers of
a step touch — per 64-sample wave (what a gather instruction group presents to the TCP / L2) and overall (the compulsory
traffic) — per level, and the line-granular time bounds that follow from the measured random-line rates
(tools/micro/random_lines.hip -> profiles/micro/round4_random_lines.txt) and the measured atomic rate (tools/micro/atomics.hip).
The hash / index arithmetic is nerfstudio's HashEncoding, restated here in torch integer ops for counting only.
usage (GPU box): python tools/line_census.py [--samples 192] [--rays 4096] [--out gpurun_out/line_census_S192.json]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thermo_nerf_amd import SceneBox, ThermalNerfModel, ThermalNerfModelConfig, synthetic  # noqa: E402
from thermo_nerf_amd.rays import RayBundle  # noqa: E402

L2_RATE, MISS_RATE, ATOMIC_RATE = 267e9, 59e9, 21e9  # lines/s: table in L2 | L2 miss (Infinity Cache / HBM) | fp32 line-atomics


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=192)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = ThermalNerfModelConfig(num_nerf_samples_per_ray=a.samples)
    model = ThermalNerfModel(cfg, metadata={"thermal": []}, scene_box=SceneBox.unit(), num_train_data=8)
    synthetic.fill_model_(model, "scene")
    model.to(dev).train()
    model.set_step(5000)
    o, d, cam = (t.to(dev) for t in synthetic.random_pixel_rays(a.rays))
    with torch.no_grad():
        torch.manual_seed(0)
        out = model(RayBundle(origins=o, directions=d, camera_indices=cam))
    eucl = out["ray_samples_list"][-1].eucl_bins  # [R, S+1]
    mid = 0.5 * (eucl[:, :-1] + eucl[:, 1:])
    pos = (o[:, None, :] + d[:, None, :] * mid[..., None]).reshape(-1, 3)  # ray-major flat order = the kernels' sample order
    # scene contraction (L-inf) and the [0,1] normalisation of NS NerfactoField.get_density
    mag = pos.abs().amax(dim=-1, keepdim=True)
    con = torch.where(mag < 1, pos, (2 - 1 / mag) * pos / mag)
    p = ((con + 2) / 4).clamp(0, 1)
    N = p.shape[0]
    enc = model.field.mlp_base.encoder
    scal = enc.scalings.reshape(-1).tolist()
    T = 1 << cfg.log2_hashmap_size
    wave = torch.arange(N, device=dev) // 64
    primes = (1, 2654435761, 805459861)
    levels = []
    for l, s in enumerate(scal):
        x = p * s
        f, c = torch.floor(x).long(), torch.ceil(x).long()
        lines = []
        for dx in (0, 1):
            for dy in (0, 1):
                for dz in (0, 1):
                    ix = (c if dx else f)[:, 0] * primes[0]
                    iy = (c if dy else f)[:, 1] * primes[1]
                    iz = (c if dz else f)[:, 2] * primes[2]
                    h = ((ix ^ iy ^ iz) % T)
                    lines.append(h // 8)  # 8 entries of 8 bytes per 64-byte line
        lines = torch.stack(lines, dim=1)  # [N, 8]
        glob = torch.unique(lines).numel()
        per_wave = torch.unique(wave[:, None] * (T // 8) + lines).numel()
        # runs of consecutive samples in one cell (what the atomic kernel's segmented scan merges): a new run starts when the cell changes
        cell = (f[:, 0] * 4099 + f[:, 1]) * 4099 + f[:, 2]
        runs = int((cell[1:] != cell[:-1]).sum().item()) + 1
        levels.append({"level": l, "scaling": s, "distinct_lines": glob, "distinct_lines_bytes": glob * 64,
                       "wave_distinct_lines": per_wave, "corner_refs": N * 8, "cell_runs": runs})
    def rate(lv):  # the level's distinct lines fit an XCD's 4 MB L2 beside the others' -> L2 rate, else the miss rate
        return L2_RATE if lv["distinct_lines_bytes"] <= (1 << 20) else MISS_RATE
    gather_s = sum(lv["wave_distinct_lines"] / rate(lv) for lv in levels)
    first_bucketed = next((i for i, lv in enumerate(levels) if lv["scaling"] >= 200.0), len(levels))
    atomic_tx = sum(min(lv["wave_distinct_lines"], lv["cell_runs"] * 4) for lv in levels[:first_bucketed])
    records = sum(lv["cell_runs"] * 4 for lv in levels[first_bucketed:])  # one 20-byte record per (run, x-neighbour corner pair)
    doc = {"samples": N, "rays": a.rays, "samples_per_ray": a.samples, "weights": "scene fill, step 5000, random-pixel batch",
           "levels": levels,
           "rates_lines_per_s": {"table_in_L2": L2_RATE, "L2_miss": MISS_RATE, "fp32_line_atomics": ATOMIC_RATE,
                                 "source": "profiles/micro/round4_random_lines.txt, profiles/micro/round2c_atomics_by_table_size.txt"},
           "gather_pass": {"wave_distinct_lines": sum(lv["wave_distinct_lines"] for lv in levels), "bound_us": gather_s * 1e6,
                           "what": "one pass of the 16-level gathers of the batch (field forward; the pose-gradient re-read in the "
                                   "mlp_base backward launch is a second one): per level, wave-distinct lines / the line rate of "
                                   "where the level's distinct lines live"},
           "scatter_pass": {"atomic_levels": first_bucketed, "atomic_line_transactions": atomic_tx,
                            "atomic_bound_us": atomic_tx / ATOMIC_RATE * 1e6, "bucketed_records": records,
                            "bucketed_bytes": records * 20 * 2,
                            "what": "levels below the bucketing threshold: one line-atomic per (wave, distinct line) after run merging; "
                                    "levels from it: 20-byte records written once and read once"},
           "algorithmic_lines_per_pass": N * 16 * 4}
    doc["step_line_granular_bound_us"] = 2 * doc["gather_pass"]["bound_us"] + doc["scatter_pass"]["atomic_bound_us"] + \
        doc["scatter_pass"]["bucketed_bytes"] / 8e12 * 1e6
    s = json.dumps(doc, indent=1)
    if a.out:
        open(a.out, "w").write(s)
    print(s)


if __name__ == "__main__":
    main()

"""Why did the fixed-batch training variants read 8-11 % slower in BENCH_r04 than in BENCH_r03 (Weak #3)?
Runs bench.measure_train_step (a) first in the process, (b) after the other GPU variants' allocations have come and gone, (c) after
a config-3 run (its 3.3 GB ray table + 30 k steps of clocks / allocator history), (d) after torch.cuda.empty_cache() +
a pause — each with rocm-smi's sclk / power beside it.
usage (GPU box): python tools/train_regress_probe.py [--config3-steps 6000]"""
import argparse
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def smi():
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=20)
        card = next(iter(json.loads(r.stdout).values()))
        keep = {k: v for k, v in card.items() if any(s in k.lower() for s in ("sclk", "mclk", "(w)", "junction", "hotspot"))}
        return json.dumps(keep)
    except Exception as e:  # pragma: no cover
        return "rocm-smi unavailable (%s)" % type(e).__name__


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config3-steps", type=int, default=6000)
    a = ap.parse_args()
    dev = torch.device("cuda:0")

    def step(tag, S=192, rb="random"):
        r = bench.measure_train_step(dev, S, cpu=False, ray_batch=rb)
        print("%-44s S=%d %-6s %.4f ms/step | reserved %.2f GB | %s" % (tag, S, rb, r["ms_per_step"], torch.cuda.memory_reserved() / 2**30, smi()), flush=True)

    step("first in the process")
    step("second (same process)")
    step("first S=48", 48)
    t = time.time()
    r = bench.measure_train_config3(dev, 192, steps=a.config3_steps, cpu=False)
    print("config3 %d steps: %.4f ms/step sustained, windows %s (%.0f s)" % (r["steps"], r["ms_per_step"], r["ms_per_step_by_window"], time.time() - t), flush=True)
    step("right after config 3")
    step("again")
    step("S=48 after config 3", 48)
    torch.cuda.empty_cache()
    time.sleep(20)
    step("after empty_cache + 20 s idle")
    step("patch batch", 192, "patch")


if __name__ == "__main__":
    main()

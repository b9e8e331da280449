"""Second probe of the run-order dependence of the fixed-batch training step (tools/train_regress_probe.py found: HEAD == round 3
on one box; S=48 reads 1.32 ms first in the process and 1.44 after a config-3 run).  Candidates: (a) which entries of torch's
32-stream pool the step's two side streams are (the trio is made once per process, training._SIDE_STREAMS: its position depends on
how many streams were handed out before) — HIP maps streams onto a few hardware queues; (b) the host side (cyclic GC over a
larger heap): the step at S=48 is within 30 % of host-bound.
usage (GPU box): python tools/train_regress_probe2.py"""
import gc
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from thermo_nerf_amd import training as TR  # noqa: E402


def main():
    dev = torch.device("cuda:0")

    def step(tag, S=48):
        r = bench.measure_train_step(dev, S, cpu=False, ray_batch="random")
        trio = next(iter(TR._SIDE_STREAMS.values()), None)
        ids = [hex(s.cuda_stream) for s in trio] if trio else None
        print("%-60s S=%d %.4f ms/step  side streams %s  gc %s" % (tag, S, r["ms_per_step"], ids, gc.get_count()), flush=True)

    step("first in the process")
    step("second")
    held = []
    for k in (1, 1, 1, 1, 4, 8, 16):
        held += [torch.cuda.Stream(device=dev) for _ in range(k)]
        TR._SIDE_STREAMS.clear()
        step("after %d more pool streams handed out (trio re-made)" % len(held))
    TR._SIDE_STREAMS.clear()
    step("S=192, trio re-made", 192)
    t = time.time()
    r = bench.measure_train_config3(dev, 192, steps=3000, cpu=False)
    print("config3 %d steps %.4f ms/step (%.0f s)" % (r["steps"], r["ms_per_step"], time.time() - t), flush=True)
    step("after config 3")
    gc.collect()
    step("after gc.collect()")
    gc.freeze()
    step("after gc.freeze()")
    gc.disable()
    step("with gc disabled")
    gc.enable()
    torch.cuda.empty_cache()
    step("after empty_cache")
    step("S=192 after config 3", 192)


if __name__ == "__main__":
    main()

"""Per-step accounting of the training step from rocprofv3 output, for profiles/train_kernels.json (bench.py's training rooflines).

    python tools/train_account.py phases  <kernel-trace dir> [anchor=field_fwd_taped] [steps=8]
    python tools/train_account.py traffic <fetch pass dir> <write pass dir> [anchor=field_fwd_taped]

phases : the CRITICAL-PATH time of each phase of a step, averaged over the last `steps` whole steps of the trace.  A step's
         kernels run on three streams, so a sum of durations counts concurrent kernels twice; here every instant of the step
         that at least one kernel covers is attributed to exactly one phase — the first phase in PRIORITY order that has a
         kernel running at that instant — and the instants no kernel covers are the `gaps`.  The phases sum to the period.
traffic: HBM-side bytes per step over ALL kernels of the step: 2 x FETCH_SIZE + WRITE_SIZE (KB as reported; the 2 x is the
         gfx950 note of /opt/skills/guides/MI355X_MICROARCH.md), summed over the dispatches between two anchors, from two
         separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass).
Both print one JSON object."""
import csv
import glob
import json
import os
import re
import sqlite3
import sys
from collections import defaultdict

# phase -> kernel-name patterns, in priority order (an instant covered by kernels of two phases goes to the earlier one)
PHASES = [
    # (round 6: with the table update deferred, the NEXT step's preparation — field_prepare, ray_head_fwd, level geometry, the
    # proposal pass — runs on the calling stream beside this step's scatter; those kernels rank below the scatter so that the
    # scatter's critical path is the time until its last kernel ends, whatever runs beside it)
    ("field_backward", r"field_bwd_fused_kernel|field_bwd_reduce_kernel|ray_render_bwd_kernel|linear_chain_bwd|linear_bwd_reduce|density_act_bwd"),
    ("field_forward", r"field_fwd_taped_kernel"),
    ("table_scatter", r"hash_encode_bwd_kernel|spread_reduce_kernel|sort_emit_kernel|sort_owner_kernel|sort_count|fillBufferAligned"),
    ("optimizer_and_pose", r"adam_kernel|multi_tensor_apply_kernel|camera_opt_"),
    ("proposal_pass", r"proposal_kernel|prop_weights_kmajor_kernel|density_fwd_train|density_bwd_train|weights_bwd|sample_pdf|sample_initial|weights_fwd_kernel|ray_head_fwd_kernel|frustum_from_edges_kernel|field_prepare_kernel"),
    ("ray_level_adjoints", r"ray_head_bwd_kernel|frustum_positions_bwd_kernel|color_input_bwd"),
    ("renderers_losses_glue", r".*"),
]


def phase_of(name):
    for i, (_, rx) in enumerate(PHASES):
        if re.search(rx, name):
            return i
    return len(PHASES) - 1


def load_rows(d):
    db = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    return sorted(cur.execute("select name, start, end from kernels"), key=lambda r: r[1])


def phases(d, anchor, steps):
    rows = load_rows(d)
    starts = [i for i, r in enumerate(rows) if anchor in r[0]]
    if len(starts) < steps + 2:
        raise SystemExit(f"anchor {anchor!r} found {len(starts)} times")
    acc = defaultdict(float)
    periods, launches, dur_sum = [], 0, 0.0
    for k in range(len(starts) - 1 - steps, len(starts) - 1):
        step = rows[starts[k]:starts[k + 1]]
        t0, t1 = step[0][1], rows[starts[k + 1]][1]
        periods.append(t1 - t0)
        launches += len(step)
        dur_sum += sum(e - s for _, s, e in step)
        # sweep over the interval end points; in each elementary interval the running kernels decide the phase
        pts = sorted({t0, t1} | {min(max(s, t0), t1) for _, s, _ in step} | {min(max(e, t0), t1) for _, _, e in step})
        for a, b in zip(pts, pts[1:]):
            live = [phase_of(n) for n, s, e in step if s <= a and e >= b]
            acc["gaps" if not live else PHASES[min(live)][0]] += b - a
    n = float(steps)
    out = {name: acc.get(name, 0.0) / n / 1e3 for name, _ in PHASES}
    out["gaps"] = acc.get("gaps", 0.0) / n / 1e3
    return {"critical_path_us_per_step": {k: round(v, 1) for k, v in out.items()},
            "period_us": round(sum(periods) / n / 1e3, 1), "launches_per_step": launches / n,
            "sum_of_kernel_durations_us_per_step": round(dur_sum / n / 1e3, 1), "steps_averaged": steps,
            "phase_priority": [p for p, _ in PHASES],
            "method": "every instant of a step goes to the first phase (in phase_priority order) with a kernel running; tools/train_account.py"}


def per_step_counter(d, counter, anchor):
    total, anchors, seen_first = 0.0, 0, False
    files = sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True))
    per = defaultdict(float)
    names = {}
    for p in files:
        for row in csv.DictReader(open(p)):
            if row["Counter_Name"] != counter:
                continue
            did = int(row["Dispatch_Id"])
            per[did] += float(row["Counter_Value"])
            names[did] = row["Kernel_Name"]
    ids = sorted(per)
    marks = [i for i in ids if anchor in names[i]]
    if len(marks) < 3:
        raise SystemExit(f"{counter}: anchor {anchor!r} found {len(marks)} times in {d}")
    lo, hi = marks[1], marks[-1]  # whole steps between the second and the last anchor
    total = sum(per[i] for i in ids if lo <= i < hi)
    return total / (len(marks) - 2), len(marks) - 2


def traffic(df, dw, anchor):
    f, nf = per_step_counter(df, "FETCH_SIZE", anchor)
    w, nw = per_step_counter(dw, "WRITE_SIZE", anchor)
    return {"fetch_size_kb_per_step": f, "write_size_kb_per_step": w, "steps_averaged": [nf, nw],
            "hbm_bytes_per_step": (2.0 * f + w) * 1024.0,
            "method": "2 x FETCH_SIZE + WRITE_SIZE (KB as reported; 2 x = the gfx950 FETCH_SIZE note of MI355X_MICROARCH.md) summed "
                      "over every dispatch of a step, separate --pmc passes; tools/train_account.py"}


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "phases":
        print(json.dumps(phases(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "field_fwd_taped",
                                int(sys.argv[4]) if len(sys.argv) > 4 else 8)))
    else:
        print(json.dumps(traffic(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "field_fwd_taped")))

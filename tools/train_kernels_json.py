"""profiles/train_kernels.json from the committed kernel traces of the training step (tools/train_profile.sh):
usage: python tools/train_kernels_json.py profiles/<tag>_kernel_trace_train_S48.txt profiles/<tag>_kernel_trace_train_S192.txt [steps=42]
Per configuration: the dominant kernel GROUP (the table-gradient scatter = hash_encode_bwd_kernel + spread_reduce + sort_*),
its time per step, the tape-free field pair (field_fwd_taped_kernel<false> + field_bwd_fused_kernel<*> + reduce + ray_head_*),
all kernel time and launches per step."""
import csv
import json
import os
import re
import sys


def rows_of(path):
    rows = [r for r in csv.reader(l for l in open(path) if not l.startswith("#") and l.strip()) if len(r) >= 5 and r[0] != "kernel"]
    commit = next((l.split("commit", 1)[1].strip() for l in open(path) if l.startswith("# measured at commit")), "unknown")
    return rows, commit


def main():
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 42
    out = {}
    for path in sys.argv[1:3]:
        S = re.search(r"_S(\d+)\.txt$", path).group(1)
        rows, commit = rows_of(path)
        tot = sum(float(r[2]) for r in rows) / steps
        launches = sum(int(r[1]) for r in rows) / steps

        def group(rx):
            sel = [r for r in rows if re.search(rx, r[0])]
            return sum(float(r[2]) for r in sel) / steps, sum(int(r[1]) for r in sel) / steps

        scat_us, scat_calls = group(r"hash_encode_bwd_kernel|spread_reduce_kernel|sort_emit_kernel|sort_owner_kernel")
        field_us, field_calls = group(r"field_fwd_taped_kernel<false>|field_bwd_fused_kernel|field_bwd_reduce_kernel|ray_head_")
        out["S" + S] = {
            "kernel": "table-gradient scatter: hash_encode_bwd_kernel (levels 0-7 + proposal grids; coarsest levels through private "
                      "dense copies + spread_reduce_kernel) + sort_emit_kernel / sort_owner_kernel (levels 8-15)",
            "calls_per_step": scat_calls, "avg_us": scat_us / max(scat_calls, 1e-9), "us_per_step": scat_us,
            "tape_free_field_us_per_step": field_us, "tape_free_field_launches_per_step": field_calls,
            "all_kernels_us_per_step": tot, "launches_per_step": launches,
            "bound": "L2 atomic path (~21 G 64-byte-line fp32 atomics/s, ~1 per clock and XCD) for the atomic part; record streaming "
                     "+ LDS compare-and-swap adds for the bucketed part (tools/micro/atomics*.hip, lds_atomics.hip; profiles/micro/)",
            "source": path, "commit": commit}
    dst = os.path.join(os.path.dirname(os.path.abspath(sys.argv[1])), "train_kernels.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

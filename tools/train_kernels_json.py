"""profiles/train_kernels.json from the per-step accounting of the training step (tools/train_account.sh -> *_train_account_S*.json)
and the cache-line census of a batch (tools/line_census.py -> *_line_census_S*.json):
usage: python tools/train_kernels_json.py <account_S48.json> <account_S192.json> <census_S48.json> <census_S192.json>
Per configuration: the critical-path time of every phase of a step (no double counting of concurrent kernels), the HBM-side bytes
of a whole step from the PMC passes, and the per-phase bounds a step is priced against in bench.py (`serial_phase_view`):
  field forward   line-granular gather bound of the census (wave-distinct lines / measured random-line rates)
  field backward  3 x the field's MLP flops (recompute + dx + dW) at the fp32-MFMA peak
  table scatter   line-atomics of the coarse levels at the measured atomic rate + the fine levels' records at HBM rate"""
import json
import os
import sys

MFMA_F32_PEAK = 157.3e12
FIELD_FLOPS_PER_SAMPLE = 33024


def main():
    out = {}
    for acct_path, census_path in zip(sys.argv[1:3], sys.argv[3:5]):
        acct, census = json.load(open(acct_path)), json.load(open(census_path))
        S = acct["samples_per_ray"]
        assert census["samples_per_ray"] == S
        cp = acct["phases"]["critical_path_us_per_step"]
        n = census["samples"]
        bounds = {"field_forward": census["gather_pass"]["bound_us"],
                  "field_backward": 3.0 * FIELD_FLOPS_PER_SAMPLE * n / MFMA_F32_PEAK * 1e6,
                  "table_scatter": census["scatter_pass"]["atomic_bound_us"] + census["scatter_pass"]["bucketed_bytes"] / 8e12 * 1e6}
        out["S%d" % S] = {
            "critical_path_us_per_step": cp, "period_us": acct["phases"]["period_us"],
            "launches_per_step": acct["phases"]["launches_per_step"],
            "sum_of_kernel_durations_us_per_step": acct["phases"]["sum_of_kernel_durations_us_per_step"],
            "hbm_bytes_per_step": acct["traffic"]["hbm_bytes_per_step"],
            "fetch_size_kb_per_step": acct["traffic"]["fetch_size_kb_per_step"],
            "write_size_kb_per_step": acct["traffic"]["write_size_kb_per_step"],
            "phase_bounds_us": {k: round(v, 1) for k, v in bounds.items()},
            "phase_frac": {k: round(bounds[k] / cp[k], 3) for k in bounds},
            "serial_bound_us": round(sum(bounds.values()), 1),
            "line_census": {"wave_distinct_lines_per_gather_pass": census["gather_pass"]["wave_distinct_lines"],
                            "algorithmic_lines_per_pass": census["algorithmic_lines_per_pass"],
                            "atomic_line_transactions": census["scatter_pass"]["atomic_line_transactions"],
                            "bucketed_records": census["scatter_pass"]["bucketed_records"],
                            "rates_lines_per_s": census["rates_lines_per_s"],
                            "step_line_granular_bound_us": round(census["step_line_granular_bound_us"], 1)},
            "dominant_phase": max(("field_backward", "field_forward", "table_scatter"), key=lambda k: cp[k]),
            "command": acct["command"], "commit": acct["commit"],
            "sources": ["profiles/" + os.path.basename(acct_path), "profiles/" + os.path.basename(census_path)],
            "method": acct["phases"]["method"] + "; " + acct["traffic"]["method"]}
    dst = os.path.join(os.path.dirname(os.path.abspath(sys.argv[1])), "train_kernels.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

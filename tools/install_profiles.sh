#!/bin/bash
# After tools/gpu_profile.sh / train_account.sh / train_pmc.sh / bench.py ran on the GPU box under tag $1 (default round5) and gpurun
# merged gpurun_out/: copy the summaries into profiles/, refresh pmc_traffic.json / train_kernels.json, keep the bench line + detail.
#   tools/install_profiles.sh round5 gpurun_out/r5_bench4.log
tag=${1:-round5}; bench=$2
cd "$(dirname "$0")/.."
cp gpurun_out/${tag}_kernel_trace_*.txt gpurun_out/${tag}_pmc_*.txt gpurun_out/${tag}_timeline_train_S*.txt gpurun_out/${tag}_train_account_S*.json profiles/
python - <<'PY'
import json
old = json.load(open("profiles/pmc_traffic.json")); new = json.load(open("gpurun_out/pmc_traffic.json"))
old.update(new); json.dump(old, open("profiles/pmc_traffic.json", "w"), indent=1)
PY
python tools/train_kernels_json.py profiles/${tag}_train_account_S48.json profiles/${tag}_train_account_S192.json profiles/round4_line_census_S48.json profiles/round4_line_census_S192.json > /dev/null
if [ -n "$bench" ]; then tail -1 "$bench" > profiles/${tag}_bench_line.json; cp gpurun_out/bench_detail.json profiles/${tag}_bench_detail.json; fi
python -m pytest tests/test_profile_stamps.py -q 2>&1 | tail -1

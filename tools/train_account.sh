#!/bin/bash
# Per-step accounting of the training step on the GPU box -> gpurun_out/<tag>_train_account_S<S>.json (+ the kernel trace, the
# one-step timeline and the all-kernel traffic passes it is made of):
#   git rev-parse HEAD > tools/.head_stamp && gpurun -- 'bash tools/train_account.sh <tag> [samples=192] [extra train_bench args]'
tag=${1:-train}
S=${2:-192}
shift; shift
extra="--ray-batch random $*"
repo=$(pwd)
export TMPDIR=/tmp
out=/tmp/acct_${tag}_S${S}
rm -rf $out; mkdir -p $out $repo/gpurun_out
cmd="python $repo/tools/train_bench.py --steps 36 --warmup 6 --samples $S $extra"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- $cmd > $out/trace.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/f -o f -- python $repo/tools/train_bench.py --steps 12 --warmup 3 --samples $S $extra > $out/f.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/w -o w -- python $repo/tools/train_bench.py --steps 12 --warmup 3 --samples $S $extra > $out/w.log 2>&1)
cd $repo
python tools/prof_summary.py $out/trace gpurun_out/${tag}_kernel_trace_train_S${S}.txt \
  "rocprofv3 --kernel-trace --stats -- python tools/train_bench.py --steps 36 --warmup 6 --samples $S $extra (42 steps in the trace)" > /dev/null
python tools/step_timeline.py $out/trace gpurun_out/${tag}_timeline_train_S${S}.txt > /dev/null 2>gpurun_out/${tag}_timeline_err.txt || true
python tools/train_account.py phases $out/trace > $out/phases.json 2>>gpurun_out/${tag}_timeline_err.txt
python tools/train_account.py traffic $out/f $out/w > $out/traffic.json 2>>gpurun_out/${tag}_timeline_err.txt
python - "$out" "$S" "$tag" "$extra" <<'PY'
import json, sys
out, S, tag, extra = sys.argv[1:5]
try:
    stamp = open("tools/.head_stamp").read().strip()
except OSError:
    stamp = "unknown"
ms = [l for l in open(out + "/trace.log") if "ms/step" in l]
doc = {"samples_per_ray": int(S), "command": "python tools/train_bench.py --steps 36 --warmup 6 --samples %s %s" % (S, extra),
       "commit": stamp, "train_bench_line_under_rocprof": ms[-1].strip() if ms else None,
       "phases": json.load(open(out + "/phases.json")), "traffic": json.load(open(out + "/traffic.json"))}
json.dump(doc, open("gpurun_out/%s_train_account_S%s.json" % (tag, S), "w"), indent=1)
print(json.dumps(doc, indent=1))
PY

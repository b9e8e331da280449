#!/bin/bash
# Build ab_<name>.so in the repo root from a given copy of ONE translation unit (default: the working copy) + the regular build's
# other objects, for tools/ab_run.sh:   tools/ab_build.sh <name> <unit.hip> [path/to/variant.hip] [extra hipcc flags...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/thermo_nerf_amd/csrc
name=$1; unit=$2; src=${3:-$CSRC/$unit}; shift; shift; shift || true
objs=""
for f in tn_samplers tn_fields tn_render tn_render_mfma tn_render_h3 tn_prepare tn_train tn_train_fused tn_metrics; do
  [ "$f.hip" = "$unit" ] || objs="$objs $CSRC/build/$f.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-function -Wno-undefined-internal \
  -Wno-pass-failed -I$CSRC "$@" -shared $objs -x hip $src -o $ROOT/ab_$name.so
echo built ab_$name.so

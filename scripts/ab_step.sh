#!/bin/bash
# Whole-step A/B on ONE box: the in-tree library ("head") and variant libraries alternate, ROUNDS times each, and the step
# time of every run is printed.  Run-to-run spread on a box is 0.01-0.03 ms of ~20.5 ms, so differences of 0.1 % show.
#   bash scripts/ab_step.sh ROUNDS head NAME...        NAME = a library built as dbg/libotter_NAME.so
# (OTTER_HIP_LIB selects the shared library otter_amd/_lib.py loads.)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
R=$1; shift
mkdir -p "$ROOT/gpurun_out"
for i in $(seq "$R"); do for v in "$@"; do
  if [ "$v" = head ]; then unset OTTER_HIP_LIB; else export OTTER_HIP_LIB=$ROOT/dbg/libotter_$v.so; fi
  timeout -k 10 200 python "$ROOT/bench.py" --steps 24 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-reference-points > "$ROOT/gpurun_out/ab_$v$i.json" || exit 1
  echo "$v" "$(python -c "import json,sys; print(json.load(open(sys.argv[1]))['ms_per_step'])" "$ROOT/gpurun_out/ab_$v$i.json")"
done; done

#!/bin/bash
# Runs on the GPU box: WRITE_SIZE (and kernel durations) of one bench configuration under rocprofv3
# for a list of YDC_XCD_TILES settings. Usage: tools/pmc_write.sh <tag> <config> <settings...>
TAG=$1; CFG=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
for X in "$@"; do
  OUT=$ROOT/gpurun_out/$TAG/${CFG}_xcd$X; mkdir -p $OUT
  YDC_XCD_TILES=$X timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -- python "$ROOT/bench.py" --config $CFG --steps 10 --warmup 2 --no-cpu-baseline --resident-only --no-extra-configs > /dev/null 2> "$OUT/write.log"
  W=$(find "$OUT/write" -name "*.db" | head -1)
  echo "== $CFG YDC_XCD_TILES=$X"
  python $ROOT/tools/rocprof_summary.py pmc "$W" | grep -v "k_match\|k_finalize\|k_servant" | head -8
  rm -rf "$OUT/write"
done

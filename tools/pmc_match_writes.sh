#!/bin/bash
# Runs on the GPU box: WRITE_SIZE and the duration of k_match_pass on cfg3 / cfg4 under rocprofv3 for a
# list of YDC_TUNE settings — the breakdown of the matching kernel's writes by subtraction
# (cp_every=1: a checkpoint before every block, round 5; cp_every=1024: only the chunk's first;
# fuse_passes=0: no hand-off granules; ...). Usage: tools/pmc_match_writes.sh <tag> <config> "<tune>" ...
TAG=$1; CFG=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
for X in "$@"; do
  OUT=$ROOT/gpurun_out/$TAG/${CFG}_$(echo "$X" | tr ',= ' '___'); mkdir -p $OUT
  YDC_TUNE="$X" timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -- python "$ROOT/bench.py" --config $CFG --steps 10 --warmup 5 --no-cpu-baseline --resident-only --no-extra-configs > "$OUT/line.json" 2> "$OUT/write.log"
  W=$(find "$OUT/write" -name "*.db" | head -1)
  echo "== $CFG YDC_TUNE=$X"
  python $ROOT/tools/rocprof_summary.py pmc "$W" | grep "k_match_pass\|kernel " | head -4
  python -c "
import json,sys
try:
    j=json.load(open('$ROOT/bench_detail.json')); print('   ms_per_step %.4f rounds %s parity_fixture %s' % (j['ms_per_step'], j['stats']['rounds'], j.get('parity_vs_reference_fixture')))
except Exception as e: print('   (no detail: %s)' % e)
"
  rm -rf "$OUT/write"
done

#!/bin/bash
# Runs on the GPU box: rocprofv3 --kernel-trace of a few synchronous batches and the dispatches of
# the last ones with their start times, durations and gaps (tools/rocprof_summary.py timeline).
# Usage: tools/timeline.sh <tag> <dispatches to print> [bench args]
set -u
TAG=${1:-cfg3}; N=${2:-40}; shift 2 || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/timeline_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace -d "$OUT/trace" -- python "$ROOT/bench.py" --steps 12 --warmup 3 --no-cpu-baseline --resident-only --no-extra-configs --no-pipeline "$@" > "$OUT/bench.json" 2> "$OUT/trace.log"
cd "$ROOT"
python tools/rocprof_summary.py timeline "$(find "$OUT/trace" -name "*.db" | head -1)" "$N" | tee "$OUT/timeline.txt"

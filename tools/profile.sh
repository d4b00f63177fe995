#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel-trace stats + the two HBM PMC
# passes (FETCH_SIZE and WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md §PMC slots) of
# bench.py, and summarises them into gpurun_out/. Usage: tools/profile.sh <tag> [bench args]
set -u
TAG=${1:-cfg2}; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -- python "$ROOT/bench.py" --steps 50 --warmup 5 --no-cpu-baseline --resident-only --no-extra-configs "$@" > "$OUT/bench_stats.json" 2> "$OUT/stats.log"
if [ "${YDC_PROFILE_PMC:-1}" = "0" ]; then  # kernel stats only
  cd "$ROOT"
  python tools/rocprof_summary.py stats "$(find "$OUT/stats" -name "*.db" | head -1)" > "$OUT/kernel_stats.txt" 2>&1
  head -20 "$OUT/kernel_stats.txt"
  exit 0
fi
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -- python "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --resident-only --no-extra-configs "$@" > /dev/null 2> "$OUT/fetch.log"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -- python "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --resident-only --no-extra-configs "$@" > /dev/null 2> "$OUT/write.log"
cd "$ROOT"
S=$(find "$OUT/stats" -name "*.db" | head -1); F=$(find "$OUT/fetch" -name "*.db" | head -1); W=$(find "$OUT/write" -name "*.db" | head -1)
python tools/rocprof_summary.py stats "$S" > "$OUT/kernel_stats.txt" 2>&1
python tools/rocprof_summary.py pmc "$F" > "$OUT/pmc_fetch.txt" 2>&1
python tools/rocprof_summary.py pmc "$W" > "$OUT/pmc_write.txt" 2>&1
python tools/rocprof_summary.py hbmjson "$F" "$W" 2.0 1.0 > "$OUT/pmc_hbm.json" 2>&1  # factors: tools/hbm_calib
tail -1 "$OUT/bench_stats.json" | cut -c1-300
head -20 "$OUT/kernel_stats.txt"

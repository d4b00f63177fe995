#!/bin/bash
# Runs on the GPU box (through gpurun): tools/hbm_calib under the two HBM PMC passes ->
# gpurun_out/calib/hbm_calibration.txt (copied to profiles/ by hand).
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/calib
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -- "$ROOT/tools/hbm_calib" > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -- "$ROOT/tools/hbm_calib" > "$OUT/write.log" 2>&1
cd "$ROOT"
F=$(find "$OUT/fetch" -name "*.db" | head -1); W=$(find "$OUT/write" -name "*.db" | head -1)
python tools/rocprof_summary.py calib "$F" "$W" > "$OUT/hbm_calibration.txt" 2>&1
cat "$OUT/hbm_calibration.txt"

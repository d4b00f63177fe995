#!/bin/bash
# Copies the evidence tools/gpu_round_final.sh left under gpurun_out/ into profiles/ (tracked).
# Usage: tools/collect_profiles.sh r03
set -eu
R=${1:?round tag, e.g. r03}
cd "$(dirname "$0")/.."
F=gpurun_out/final
for c in cfg2 driver_line gpus2_one_device cfg2_sync cfg2_shared cfg2_d150 cfg3 cfg4 cfg5 dist1_rccl dist1_ipc; do
  if [ -s $F/bench_$c.json ] && tail -1 $F/bench_$c.json | python -c 'import json,sys; json.loads(sys.stdin.read())' 2>/dev/null; then
    tail -1 $F/bench_$c.json > profiles/${R}_bench_$c.json
  else
    echo "no bench line for $c (kept the previous one)" >&2
  fi
done
for c in cfg2 cfg3 cfg4; do
  P=gpurun_out/prof_$c
  cp $P/kernel_stats.txt profiles/${R}_${c}_kernel_stats.txt
  cp $P/pmc_hbm.json profiles/${R}_${c}_pmc_hbm.json
  { echo "# rocprofv3 --kernel-trace --pmc FETCH_SIZE (separate pass), per kernel: KB per dispatch"; cat $P/pmc_fetch.txt
    echo; echo "# rocprofv3 --kernel-trace --pmc WRITE_SIZE (separate pass)"; cat $P/pmc_write.txt; } > profiles/${R}_${c}_pmc_hbm.txt
  [ -s $F/phase_$c.txt ] && cp $F/phase_$c.txt profiles/${R}_${c}_match_phases.txt
done
[ -s $F/scatter_probe.txt ] && cp $F/scatter_probe.txt profiles/${R}_scatter_probe.txt
[ -s $F/tail_cfg4.txt ] && cp $F/tail_cfg4.txt profiles/${R}_cfg4_match_tail.txt
[ -s $F/front_cfg2.txt ] && cp $F/front_cfg2.txt profiles/${R}_cfg2_front_phases.txt
[ -s $F/walk_groups.txt ] && cp $F/walk_groups.txt profiles/${R}_walk_groups_probe.txt
[ -s $F/issue_probe.txt ] && cp $F/issue_probe.txt profiles/${R}_issue_probe.txt
[ -s $F/fastloop_probe.txt ] && cp $F/fastloop_probe.txt profiles/${R}_fastloop_probe.txt
python tools/rocprof_summary.py hbmtable "cfg2 (100k requests x 2k servants)=profiles/${R}_cfg2_pmc_hbm.json" \
  "cfg3 (1M requests x 8k servants, 4 digests)=profiles/${R}_cfg3_pmc_hbm.json" \
  "cfg4 (4M requests x 16k servants, 4 digests)=profiles/${R}_cfg4_pmc_hbm.json" > profiles/${R}_hbm_utilisation.txt
cp $F/td_native_bench.log profiles/${R}_td_native_bench.txt
[ -s gpurun_out/rccl_1rank_debug.log ] && cp gpurun_out/rccl_1rank_debug.log profiles/${R}_rccl_1rank_debug.log
git status --short profiles

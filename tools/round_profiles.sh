#!/bin/bash
# Runs on the GPU box (through gpurun): the evidence of a round in one go — rocprofv3 kernel stats
# and the two HBM PMC passes of cfg2 / cfg3 / cfg4, kernel stats of the captured streaming step
# (cfg5), the bench lines of every configuration, the TaskDispatcher surface (throughput, heartbeats,
# per-call latency), the small-batch kernel's phase stamps and the launch-latency probe — and
# copies the summaries into profiles/<round>_* (tracked; gpurun_out/ is scratch).
# Usage: tools/round_profiles.sh r05      (then, back home: git add profiles/)
set -u
R=${1:?round tag, e.g. r05}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
F=gpurun_out/final; mkdir -p $F profiles
for c in cfg2 cfg3 cfg4; do
  bash tools/profile.sh $c --config $c > $F/profile_$c.log 2>&1
  P=gpurun_out/prof_$c
  cp $P/kernel_stats.txt profiles/${R}_${c}_kernel_stats.txt
  cp $P/pmc_hbm.json profiles/${R}_${c}_pmc_hbm.json
  { echo "# rocprofv3 --kernel-trace --pmc FETCH_SIZE (separate pass), per kernel: KB per dispatch"; cat $P/pmc_fetch.txt
    echo; echo "# rocprofv3 --kernel-trace --pmc WRITE_SIZE (separate pass)"; cat $P/pmc_write.txt; } > profiles/${R}_${c}_pmc_hbm.txt
done
YDC_PROFILE_PMC=0 bash tools/profile.sh cfg5 --config cfg5 > $F/profile_cfg5.log 2>&1
cp gpurun_out/prof_cfg5/kernel_stats.txt profiles/${R}_cfg5_kernel_stats.txt
python tools/rocprof_summary.py hbmtable "cfg2 (100k requests x 2k servants)=profiles/${R}_cfg2_pmc_hbm.json" \
  "cfg3 (1M requests x 8k servants, 4 digests)=profiles/${R}_cfg3_pmc_hbm.json" \
  "cfg4 (4M requests x 16k servants, 4 digests)=profiles/${R}_cfg4_pmc_hbm.json" > profiles/${R}_hbm_utilisation.txt
# bench lines
timeout 900 python bench.py --steps 20 --warmup 5 > $F/bench_driver_line.json 2> $F/bench_driver_line.err
timeout 600 python bench.py --no-extra-configs --steps 2000 --warmup 100 > $F/bench_cfg2.json 2> $F/bench_cfg2.err
timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --no-pipeline --steps 2000 --warmup 100 > $F/bench_cfg2_sync.json 2> $F/bench_cfg2_sync.err
timeout 300 python bench.py --config cfg3 --no-cpu-baseline --no-extra-configs --steps 200 --warmup 10 > $F/bench_cfg3.json 2> $F/bench_cfg3.err
timeout 300 python bench.py --config cfg4 --no-cpu-baseline --no-extra-configs --steps 100 --warmup 5 > $F/bench_cfg4.json 2> $F/bench_cfg4.err
timeout 300 python bench.py --config cfg5 --steps 1000 --warmup 50 > $F/bench_cfg5.json 2> $F/bench_cfg5.err
timeout 300 python bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline > $F/bench_gpus2_one_device.json 2> $F/bench_gpus2_one_device.err
for c in driver_line cfg2 cfg2_sync cfg3 cfg4 cfg5 gpus2_one_device; do
  if [ -s $F/bench_$c.json ] && tail -1 $F/bench_$c.json | python -c 'import json,sys; json.loads(sys.stdin.read())' 2>/dev/null; then
    tail -1 $F/bench_$c.json > profiles/${R}_bench_$c.json
  else
    echo "no bench line for $c" >&2; tail -3 $F/bench_$c.err >&2
  fi
done
[ -s profiles/${R}_bench_driver_line.json ] && python tools/latency_table.py profiles/${R}_bench_driver_line.json > profiles/${R}_td_latency_table.txt
# the TaskDispatcher surface, natively
{ for a in "wait 2000 10000 50" "wait 2000 100000 20" "heartbeat 16000 1000000 3" "heartbeat 2000 100000 5" \
           "latency 2000 1000" "latency 8000 1000" "latency 16000 1000"; do
    echo "== td_native_bench $a"; timeout 300 ./tools/td_native_bench $a; done
  for a in "concurrent 2000 3000" "concurrent 16000 3000"; do echo "== td_native_bench $a"; timeout 300 ./tools/td_native_bench $a; done
  echo "== td_native_bench latency 2000 1000 (YDC_TUNE=resident=0: one launch per call)"
  YDC_TUNE=resident=0 timeout 300 ./tools/td_native_bench latency 2000 500
  echo "== td_native_bench latency 2000 1000 (YDC_TUNE=packed_tick=0: the reference's double as the key)"
  YDC_TUNE=packed_tick=0 timeout 300 ./tools/td_native_bench latency 2000 500; } > profiles/${R}_td_native_bench.txt 2>&1
# the small-batch kernel: where a launch and a resident command spend their time
{ for a in "2000 1 1" "2000 16 16" "2000 64 16" "8000 16 16" "16000 1 1" "16000 16 16"; do
    YDC_TUNE=resident=0 timeout 120 python tools/tick_probe.py $a 200; echo; done; } > profiles/${R}_tick_phases.txt 2>&1
timeout 120 ./tests/tools/launch_probe 2000 > profiles/${R}_launch_probe.txt 2>&1
timeout 300 python tools/commit_loop.py 300 > profiles/${R}_commit_loop.txt 2>&1
# the walk of the dedicated tier's end on variants of cfg3's pool; the corner of DESIGN 9.7; two queues
timeout 600 python tools/zone_probe.py > profiles/${R}_zone_probe.txt 2>&1
{ timeout 200 python tools/cliff_probe.py; timeout 200 python tools/cliff_probe.py probe; } > profiles/${R}_cliff_probe.txt 2>&1
{ for m in 0 1 3; do timeout 60 ./tests/tools/overlap_probe 200 150 1954 $m; echo; done; } > profiles/${R}_overlap_probe.txt 2>&1
{ timeout 100 ./tests/tools/atomic_probe; timeout 100 ./tests/tools/atomic_probe 1250000; } > profiles/${R}_atomic_probe.txt 2>&1
# the matching kernel's phases (measurement build)
for c in cfg2 cfg3 cfg4; do timeout 300 python tools/phase_probe.py $c 10 > profiles/${R}_${c}_match_phases.txt 2>&1; done
[ -s gpurun_out/rccl_1rank_debug.log ] && cp gpurun_out/rccl_1rank_debug.log profiles/${R}_rccl_1rank_debug.log
mkdir -p gpurun_out/profiles_$R && cp profiles/${R}_* gpurun_out/profiles_$R/
ls profiles | grep "^${R}_" | wc -l

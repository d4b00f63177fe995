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
# bench lines: what the driver parses (the short line) as ..._line.json, everything that was measured
# in that run (bench_detail.json) as ....json. cfg3 / cfg4 with the reference beside them (the
# first 100k requests of the batch: 13 s / 27 s of the host's CPU).
run_bench() {  # <name> <timeout> <bench args...>
  local name=$1 t=$2; shift 2
  rm -f bench_detail.json
  timeout $t python bench.py "$@" > $F/bench_$name.line 2> $F/bench_$name.err
  if [ -s $F/bench_$name.line ] && tail -1 $F/bench_$name.line | python -c 'import json,sys; json.loads(sys.stdin.read())' 2>/dev/null \
     && [ -s bench_detail.json ]; then
    tail -1 $F/bench_$name.line > profiles/${R}_bench_${name}_line.json
    python -c 'import json,sys; json.dump(json.load(open("bench_detail.json")), open(sys.argv[1], "w"))' profiles/${R}_bench_$name.json
  else
    echo "no bench line for $name" >&2; tail -3 $F/bench_$name.err >&2
  fi
}
run_bench driver_line 1200 --steps 20 --warmup 5
run_bench cfg2 600 --no-extra-configs --steps 2000 --warmup 100
run_bench cfg2_sync 300 --no-cpu-baseline --no-extra-configs --no-pipeline --steps 2000 --warmup 100
run_bench cfg3 400 --config cfg3 --no-extra-configs --steps 200 --warmup 10
run_bench cfg4 400 --config cfg4 --no-extra-configs --steps 100 --warmup 5
run_bench cfg5 300 --config cfg5 --steps 1000 --warmup 50
run_bench gpus2_one_device 300 --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline
[ -s profiles/${R}_bench_driver_line.json ] && python tools/latency_table.py profiles/${R}_bench_driver_line.json > profiles/${R}_td_latency_table.txt
# the TaskDispatcher surface, natively
{ for a in "wait 2000 10000 50" "wait 2000 100000 20" "heartbeat 16000 1000000 3" "heartbeat 2000 100000 5" \
           "latency 2000 1000" "latency 8000 1000" "latency 16000 1000"; do
    echo "== td_native_bench $a"; timeout 300 ./tools/td_native_bench $a; done
  for a in "concurrent 2000 3000" "concurrent 16000 3000"; do echo "== td_native_bench $a"; timeout 300 ./tools/td_native_bench $a; done
  # round 6: the 1 s expiration timer running (10^5 and 10^6 live leases), and K parked waiters on a
  # saturated pool with the reference's multi-threaded build under the same workload beside it
  for a in "timer 2000 100000 4" "timer 16000 1000000 5" "parked 300 2"; do echo "== td_native_bench $a"; timeout 600 ./tools/td_native_bench $a; done
  echo "== oracle/_ref/ref_parked_bench 100 2 100 1000 (the reference, same workload)"
  timeout 300 ./oracle/_ref/ref_parked_bench 100 2 100 1000
  echo "== oracle/_ref/ref_parked_bench 20 1 10000"
  timeout 300 ./oracle/_ref/ref_parked_bench 20 1 10000
  echo "== td_native_bench latency 2000 1000 (YDC_TUNE=resident=0: one launch per call)"
  YDC_TUNE=resident=0 timeout 300 ./tools/td_native_bench latency 2000 500
  echo "== td_native_bench latency 2000 1000 (YDC_TUNE=packed_tick=0: the reference's double as the key)"
  YDC_TUNE=packed_tick=0 timeout 300 ./tools/td_native_bench latency 2000 500; } > profiles/${R}_td_native_bench.txt 2>&1
# the small-batch kernel: where a launch and a resident command spend their time
{ for a in "2000 1 1" "2000 16 16" "2000 64 16" "8000 16 16" "16000 1 1" "16000 16 16"; do
    YDC_TUNE=resident=0 timeout 120 python tools/tick_probe.py $a 200; echo; done; } > profiles/${R}_tick_phases.txt 2>&1
timeout 120 ./tests/tools/launch_probe 2000 > profiles/${R}_launch_probe.txt 2>&1
timeout 300 python tools/commit_loop.py 300 > profiles/${R}_commit_loop.txt 2>&1
# the matching kernel's writes by buffer (WRITE_SIZE under YDC_TUNE variants)
{ bash tools/pmc_match_writes.sh ${R}w cfg3 "cp_every=1" "cp_every=4" "cp_every=1024" "cp_every=4,fuse_passes=0"
  bash tools/pmc_match_writes.sh ${R}w cfg4 "cp_every=1" "cp_every=4" "cp_every=1024"; } > profiles/${R}_match_writes.txt 2>&1
# the matching kernel's phases (measurement build)
for c in cfg2 cfg3 cfg4; do timeout 300 python tools/phase_probe.py $c 10 > profiles/${R}_${c}_match_phases.txt 2>&1; done
[ -s gpurun_out/rccl_1rank_debug.log ] && cp gpurun_out/rccl_1rank_debug.log profiles/${R}_rccl_1rank_debug.log
mkdir -p gpurun_out/profiles_$R && cp profiles/${R}_* gpurun_out/profiles_$R/
ls profiles | grep "^${R}_" | wc -l

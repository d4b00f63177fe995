#!/bin/bash
# Runs on the GPU box (through gpurun): the round's evidence — GPU test suite, the bench lines of
# every configuration, rocprofv3 kernel stats + HBM PMC passes for cfg2 / cfg3, phase stamps of the
# matching kernel, the native host class throughput. Everything lands in gpurun_out/final/
# (tools/collect_profiles.sh copies it to profiles/).
O=gpurun_out/final; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
(timeout 1200 python -m pytest tests -m gpu -q --timeout 420 --durations=10 -p no:cacheprovider 2>&1 | tail -25) > $O/pytest.log
timeout 300 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
( time timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_line.json 2> $O/bench_driver_line.err
( time timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 ) > $O/bench_gpus2_one_device.json 2> $O/bench_gpus2_one_device.err
timeout 300 python bench.py --no-pipeline --no-cpu-baseline --no-extra-configs > $O/bench_cfg2_sync.json 2> $O/bench_cfg2_sync.err
timeout 300 python bench.py --shared-ip-frac 0.05 --steps 2000 --warmup 100 --no-cpu-baseline --no-extra-configs > $O/bench_cfg2_shared.json 2> $O/bench_cfg2_shared.err
timeout 300 python bench.py --config cfg3 --steps 500 --warmup 20 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
timeout 300 python bench.py --config cfg4 --steps 200 --warmup 10 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
timeout 300 python bench.py --config cfg5 --steps 1000 --warmup 50 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
timeout 300 python bench.py --digests 150 --steps 5 --warmup 1 > $O/bench_cfg2_d150.json 2> $O/bench_cfg2_d150.err
for t in rccl ipc; do
  YDC_BENCH_FORCE_DIST=1 YDC_BENCH_RCCL_TIMEOUT=100 timeout 260 python bench.py --gpus 1 --steps 500 --warmup 50 --transport $t --no-cpu-baseline > $O/bench_dist1_$t.json 2> $O/bench_dist1_$t.err
done
for c in cfg2 cfg3 cfg4; do
  YDC_LIB=$PWD/yadcc_amd/libydc_probe.so timeout 200 python tools/phase_probe.py $c 20 > $O/phase_$c.txt 2>&1
done
[ -x build/issue_probe ] && timeout 100 build/issue_probe > $O/issue_probe.txt 2>&1
[ -x build/fastloop_probe ] && timeout 100 build/fastloop_probe > $O/fastloop_probe.txt 2>&1
timeout 300 bash tools/profile.sh cfg2 > $O/profile_cfg2.log 2>&1
timeout 400 bash tools/profile.sh cfg3 --config cfg3 > $O/profile_cfg3.log 2>&1
timeout 400 bash tools/profile.sh cfg4 --config cfg4 > $O/profile_cfg4.log 2>&1
{ timeout 120 tools/td_native_bench wait 2000 10000 50; timeout 120 tools/td_native_bench wait 2000 100000 20;
  timeout 200 tools/td_native_bench heartbeat 16000 1000000 3; timeout 120 tools/td_native_bench heartbeat 2000 100000 5; } > $O/td_native_bench.log 2>&1
[ -x tests/tools/scatter_probe ] && timeout 100 tests/tools/scatter_probe 5000000 8 8 > $O/scatter_probe.txt 2>&1
YDC_LIB=$PWD/yadcc_amd/libydc_probe.so timeout 100 python tools/tail_probe.py cfg4 8 > $O/tail_cfg4.txt 2>&1
YDC_LIB=$PWD/yadcc_amd/libydc_probe.so timeout 100 python tools/front_probe.py > $O/front_cfg2.txt 2>&1
YDC_LIB=$PWD/yadcc_amd/libydc_probe.so timeout 100 python tools/walk_probe.py groups > $O/walk_groups.txt 2>&1
cat $O/pytest.log
python - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        e=j.get("end_to_end") or {}
        print(os.path.basename(f), "ms/step %.4f" % j["ms_per_step"], "sync", j.get("ms_per_step_synchronous"), "p99 %.4f" % j["p99_dispatch_latency_ms"],
              "e2e ms %.4f p99 %.4f" % (e.get("ms_per_batch", 0), e.get("p99_ms", 0)), "pageable", (e.get("pageable_buffers") or {}).get("ms_per_batch"), "rounds", j["stats"].get("rounds"),
              "parity", j.get("parity_vs_cpu_baseline"), j.get("parity_vs_oracle"), j.get("transport"))
        print("    ", {k: round(v,1) for k,v in j.get("kernels_us_per_step", {}).items()})
    except Exception as ex:
        print(os.path.basename(f), "ERR", ex); print(open(f.replace('.json','.err')).read()[-1200:])
PY
head -14 gpurun_out/prof_cfg2/kernel_stats.txt; head -12 gpurun_out/prof_cfg3/kernel_stats.txt; cat $O/td_native_bench.log

#!/bin/bash
# Round-3 checkpoint A (through gpurun): GPU tests (multi-process sharded path, mailbox
# transport, bench smoke), phase stamps of k_match_pass, SQ counters, cfg2 bench, group of one.
O=gpurun_out/r3a; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
(timeout 1500 python -m pytest tests -m gpu -q --timeout 420 --durations=15 -p no:cacheprovider 2>&1 | tail -60) > $O/pytest.log
cat $O/pytest.log | tail -40
for c in cfg2 cfg3; do
  YDC_LIB=$PWD/yadcc_amd/libydc_probe.so timeout 200 python tools/phase_probe.py $c 20 > $O/phase_$c.txt 2>&1
done
cat $O/phase_cfg2.txt
timeout 300 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
for t in rccl ipc; do
  YDC_BENCH_FORCE_DIST=1 YDC_BENCH_RCCL_TIMEOUT=100 NCCL_DEBUG=INFO timeout 260 python bench.py --gpus 1 --steps 500 --warmup 50 --transport $t --no-cpu-baseline > $O/bench_dist1_$t.json 2> $O/bench_dist1_$t.err
done
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
for c in cfg2 cfg3; do
  X=""; [ $c = cfg3 ] && X="--config cfg3"
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $ROOT/$O/sq1_$c -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline $X > /dev/null 2> $ROOT/$O/sq1_$c.log
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD -d $ROOT/$O/sq2_$c -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline $X > /dev/null 2> $ROOT/$O/sq2_$c.log
done
cd $ROOT
for c in cfg2 cfg3; do for k in 1 2; do
  D=$(find $O/sq${k}_$c -name "*.db" | head -1)
  [ -n "$D" ] && python tools/rocprof_summary.py pmc "$D" > $O/sq${k}_$c.txt 2>&1
  rm -rf $O/sq${k}_$c
done; done
python - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        e=j.get("end_to_end") or {}
        print(os.path.basename(f), "ms/step %.4f" % j["ms_per_step"], "p99 %.4f" % j["p99_dispatch_latency_ms"],
              "e2e ms %.4f" % e.get("ms_per_batch", 0), "rounds", j["stats"].get("rounds"),
              "parity", j.get("parity_vs_cpu_baseline"), j.get("parity_vs_oracle"), "sharded", j.get("sharded"), j.get("transport_detail"))
        print("    ", {k: round(v,1) for k,v in j.get("kernels_us_per_step", {}).items()})
    except Exception as ex:
        print(os.path.basename(f), "ERR", ex); print(open(f.replace('.json','.err')).read()[-1500:])
PY
grep -h "k_match" $O/sq1_cfg2.txt $O/sq2_cfg2.txt | head -20

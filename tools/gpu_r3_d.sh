#!/bin/bash
# Round-3 checkpoint D: walk mode of the wide kernel, group ranks on the bin sort, full GPU suite.
O=gpurun_out/r3d; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
(timeout 1500 python -m pytest tests -m gpu -q -x --timeout 300 --durations=8 -p no:cacheprovider 2>&1 | tail -30) > $O/pytest.log
tail -22 $O/pytest.log
timeout 300 python bench.py --digests 150 --steps 20 --warmup 2 > $O/bench_cfg2_d150.json 2> $O/bench_cfg2_d150.err
YDC_WIDE=0 timeout 400 python bench.py --digests 150 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_cfg2_d150_thread.json 2> $O/bench_cfg2_d150_thread.err
for t in rccl ipc; do
  YDC_BENCH_FORCE_DIST=1 YDC_BENCH_RCCL_TIMEOUT=100 timeout 260 python bench.py --gpus 1 --steps 500 --warmup 50 --transport $t --no-cpu-baseline > $O/bench_dist1_$t.json 2> $O/bench_dist1_$t.err
done
python - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        e=j.get("end_to_end") or {}
        print(os.path.basename(f), "ms/step %.4f" % j["ms_per_step"], "sync", j.get("ms_per_step_synchronous"), "p99 %.4f" % j["p99_dispatch_latency_ms"],
              "e2e ms %.4f" % e.get("ms_per_batch", 0), "rounds", j["stats"].get("rounds"), "classes", j["stats"]["n_classes"],
              "parity", j.get("parity_vs_cpu_baseline"), j.get("parity_vs_oracle"), j.get("transport"))
        print("    ", {k: round(v,1) for k,v in j.get("kernels_us_per_step", {}).items()})
    except Exception as ex:
        print(os.path.basename(f), "ERR", ex); print(open(f.replace('.json','.err')).read()[-1500:])
PY

#!/bin/bash
# Round-3 checkpoint C: prologue of k_match_pass (level table windows fill the rings), wide kernel (> 256 classes).
O=gpurun_out/r3c; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_binsort_gpu.py tests/test_streaming_gpu.py tests/test_task_dispatcher_gpu.py tests/test_golden_fixtures.py -m gpu -q -x --timeout 300 --durations=8 -p no:cacheprovider 2>&1 | tail -40) > $O/pytest.log
tail -25 $O/pytest.log
YDC_LIB=$PWD/yadcc_amd/libydc_probe.so timeout 200 python tools/phase_probe.py cfg2 20 > $O/phase_cfg2.txt 2>&1
sed -n 16,32p $O/phase_cfg2.txt
timeout 300 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 300 python bench.py --digests 150 --steps 300 --warmup 20 > $O/bench_cfg2_d150.json 2> $O/bench_cfg2_d150.err
YDC_WIDE=0 timeout 300 python bench.py --digests 150 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_cfg2_d150_thread.json 2> $O/bench_cfg2_d150_thread.err
timeout 300 python bench.py --shared-ip-frac 0.05 --steps 2000 --warmup 100 --no-cpu-baseline > $O/bench_cfg2_shared.json 2> $O/bench_cfg2_shared.err
python - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        e=j.get("end_to_end") or {}
        print(os.path.basename(f), "ms/step %.4f" % j["ms_per_step"], "sync", j.get("ms_per_step_synchronous"), "p99 %.4f" % j["p99_dispatch_latency_ms"],
              "e2e ms %.4f p99 %.4f" % (e.get("ms_per_batch", 0), e.get("p99_ms", 0)), "rounds", j["stats"].get("rounds"), "classes", j["stats"]["n_classes"],
              "parity", j.get("parity_vs_cpu_baseline"), e.get("same_placement_as_pageable"))
        print("    ", {k: round(v,1) for k,v in j.get("kernels_us_per_step", {}).items()})
    except Exception as ex:
        print(os.path.basename(f), "ERR", ex); print(open(f.replace('.json','.err')).read()[-1500:])
PY

#!/bin/bash
# Round-3 checkpoint B: level table, pipelined batches, zero-copy host buffers, address aliases, wide ticks.
O=gpurun_out/r3b; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
(timeout 1200 python -m pytest tests -m gpu -q -x --timeout 300 --durations=8 -p no:cacheprovider --deselect tests/test_sharded_gpu.py::test_cfg4_shape_eight_ranks 2>&1 | tail -40) > $O/pytest.log
tail -30 $O/pytest.log
YDC_LIB=$PWD/yadcc_amd/libydc_probe.so timeout 200 python tools/phase_probe.py cfg2 20 > $O/phase_cfg2.txt 2>&1
sed -n 1,32p $O/phase_cfg2.txt
timeout 300 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 300 python bench.py --no-pipeline --no-cpu-baseline > $O/bench_cfg2_sync.json 2> $O/bench_cfg2_sync.err
YDC_HOST_IN=copy timeout 300 python bench.py --no-cpu-baseline --steps 500 > $O/bench_cfg2_hostcopy.json 2> $O/bench_cfg2_hostcopy.err
YDC_LEVEL_TAB=0 timeout 300 python bench.py --no-cpu-baseline --steps 2000 > $O/bench_cfg2_notab.json 2> $O/bench_cfg2_notab.err
timeout 300 python bench.py --config cfg3 --steps 300 --warmup 20 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err
timeout 300 python bench.py --config cfg4 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err
python - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        e=j.get("end_to_end") or {}
        print(os.path.basename(f), "ms/step %.4f" % j["ms_per_step"], "sync", j.get("ms_per_step_synchronous"), "p99 %.4f" % j["p99_dispatch_latency_ms"],
              "e2e ms %.4f p99 %.4f" % (e.get("ms_per_batch", 0), e.get("p99_ms", 0)), "pageable", (e.get("pageable_buffers") or {}).get("ms_per_batch"), "rounds", j["stats"].get("rounds"),
              "parity", j.get("parity_vs_cpu_baseline"), e.get("same_placement_as_pageable"))
        print("    ", {k: round(v,1) for k,v in j.get("kernels_us_per_step", {}).items()})
    except Exception as ex:
        print(os.path.basename(f), "ERR", ex); print(open(f.replace('.json','.err')).read()[-1500:])
PY

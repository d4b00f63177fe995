#!/bin/bash
# Runs on the GPU box (through gpurun): the bin sort's own tests with the host-side verification
# of every sort switched on, then cfg2 with the bin sort and with the radix sort side by side.
O=gpurun_out/bin; mkdir -p $O
(YDC_BINSORT_VERIFY=1 timeout 400 python -m pytest tests/test_binsort_gpu.py -x -q --timeout 150 2>&1 | tail -60) > $O/pytest_bin.log
timeout 120 python bench.py --no-cpu-baseline --steps 2000 --warmup 100 > $O/bench_bin.json 2> $O/bench_bin.err
YDC_BINSORT=0 timeout 120 python bench.py --no-cpu-baseline --steps 2000 --warmup 100 > $O/bench_radix.json 2> $O/bench_radix.err
cat $O/pytest_bin.log
python - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        e=j.get("end_to_end") or {}
        print(os.path.basename(f), "ms/step %.4f" % j["ms_per_step"], "p99 %.4f" % j["p99_dispatch_latency_ms"],
              "e2e ms %.4f" % e.get("ms_per_batch", 0), "rounds", j["stats"].get("rounds"), "radix_passes", j["stats"].get("radix_passes"))
        print("    ", {k: round(v,1) for k,v in j.get("kernels_us_per_step", {}).items()})
    except Exception as ex:
        print(os.path.basename(f), "ERR", ex); print(open(f.replace('.json','.err')).read()[-1500:])
PY

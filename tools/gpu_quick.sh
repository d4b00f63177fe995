#!/bin/bash
# Runs on the GPU box (through gpurun): the whole GPU suite and the cfg2 bench line.
O=gpurun_out/quick; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -x -q --timeout 200 2>&1 | tail -40) > $O/pytest.log
timeout 120 python bench.py --no-cpu-baseline --steps 2000 --warmup 100 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
cat $O/pytest.log
python - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        e=j.get("end_to_end") or {}
        print(os.path.basename(f), "ms/step %.4f" % j["ms_per_step"], "p99 %.4f" % j["p99_dispatch_latency_ms"],
              "e2e ms %.4f" % e.get("ms_per_batch", 0), "rounds", j["stats"].get("rounds"), "radix_passes", j["stats"].get("radix_passes"))
        print("    ", {k: round(v,1) for k,v in j.get("kernels_us_per_step", {}).items()}, j.get("kernel_launches_per_step"))
    except Exception as ex:
        print(os.path.basename(f), "ERR", ex); print(open(f.replace('.json','.err')).read()[-1500:])
PY

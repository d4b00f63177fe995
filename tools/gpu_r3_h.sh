#!/bin/bash
O=gpurun_out/r3h; mkdir -p $O
(timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_streaming_gpu.py -m gpu -q -x --timeout 120 --durations=5 -p no:cacheprovider 2>&1 | tail -16) > $O/pytest.log
tail -14 $O/pytest.log
timeout 200 python bench.py --digests 150 --steps 10 --warmup 1 > $O/bench_cfg2_d150.json 2> $O/bench_cfg2_d150.err
python - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "ms/step %.4f" % j["ms_per_step"], "rounds", j["stats"].get("rounds"), "classes", j["stats"]["n_classes"], "parity", j.get("parity_vs_cpu_baseline"))
        print("    ", {k: round(v,1) for k,v in j.get("kernels_us_per_step", {}).items()})
    except Exception as ex:
        print(os.path.basename(f), "ERR", ex); print(open(f.replace('.json','.err')).read()[-1500:])
PY

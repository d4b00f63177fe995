#!/bin/bash
# Trimmed matching loop (mask-bit loop control, shifted mask copies, chained pairs): parity, then timing.
O=gpurun_out/trim; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_binsort_gpu.py tests/test_streaming_gpu.py tests/test_golden_fixtures.py tests/test_sharded_gpu.py -m gpu -q -x --timeout 300 -p no:cacheprovider 2>&1 | tail -12) > $O/pytest.log
cat $O/pytest.log
timeout 200 python bench.py --resident-only > $O/cfg2.json 2> $O/cfg2.err
timeout 200 python bench.py --config cfg3 --steps 300 --warmup 20 --resident-only > $O/cfg3.json 2> $O/cfg3.err
timeout 200 python bench.py --config cfg4 --steps 100 --warmup 10 --resident-only > $O/cfg4.json 2> $O/cfg4.err
YDC_PAIR=0 timeout 200 python bench.py --config cfg4 --steps 100 --warmup 10 --resident-only > $O/cfg4_nopair.json 2> $O/cfg4_nopair.err
python - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "ms/step %.4f" % j["ms_per_step"], "sync", j.get("ms_per_step_synchronous"), "rounds", j["stats"].get("rounds"), "parity", j.get("parity_vs_cpu_baseline"))
        print("    ", {k: round(v,1) for k,v in j.get("kernels_us_per_step", {}).items()})
    except Exception as ex:
        print(os.path.basename(f), "ERR", ex); print(open(f.replace('.json','.err')).read()[-1500:])
PY

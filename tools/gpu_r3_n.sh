#!/bin/bash
# Staging split + grouped ring refills: parity, then timing.
O=gpurun_out/refill; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_binsort_gpu.py tests/test_streaming_gpu.py tests/test_golden_fixtures.py tests/test_sharded_gpu.py -m gpu -q -x --timeout 300 -p no:cacheprovider 2>&1 | tail -6) > $O/pytest.log
cat $O/pytest.log
timeout 200 python bench.py --resident-only --no-cpu-baseline > $O/cfg2.json 2> $O/cfg2.err
timeout 200 python bench.py --config cfg3 --steps 300 --warmup 20 --resident-only --no-cpu-baseline > $O/cfg3.json 2> $O/cfg3.err
timeout 200 python bench.py --config cfg4 --steps 100 --warmup 10 --resident-only --no-cpu-baseline > $O/cfg4.json 2> $O/cfg4.err
timeout 200 python bench.py --config cfg5 --steps 500 --warmup 50 --resident-only --no-cpu-baseline > $O/cfg5.json 2> $O/cfg5.err
python - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print("%-8s" % os.path.basename(f)[:-5], "ms/step %.4f" % j["ms_per_step"], "rounds", j["stats"].get("rounds"), "parity", j.get("parity_vs_cpu_baseline"), j.get("parity_vs_oracle"), "match %.1f" % j.get("kernels_us_per_step", {}).get("k_match_pass", 0))
    except Exception as ex:
        print(os.path.basename(f), "ERR", ex); print(open(f.replace('.json','.err')).read()[-800:])
PY
for c in cfg3 cfg4; do YDC_LIB=$PWD/yadcc_amd/libydc_probe.so timeout 120 python tools/phase_probe.py $c 10 2>&1 | tail -9; done

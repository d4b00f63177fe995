#!/bin/bash
# Round-3 closing check: full GPU suite, the many-class bench line, cfg2 line.
O=gpurun_out/final; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
(timeout 1200 python -m pytest tests -m gpu -q --timeout 420 --durations=6 -p no:cacheprovider 2>&1 | tail -14) > $O/pytest.log
cat $O/pytest.log
timeout 300 python bench.py --digests 150 --steps 20 --warmup 2 > $O/bench_cfg2_d150.json 2> $O/bench_cfg2_d150.err
timeout 300 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 300 python bench.py --config cfg3 --steps 500 --warmup 20 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
YDC_LIB=$PWD/yadcc_amd/libydc_probe.so timeout 200 python tools/walk_probe.py 150 100000 > $O/walk_probe.txt 2>&1
python - $O <<'PY'
import json,sys,glob,os
for f in ("bench_cfg2.json","bench_cfg2_d150.json","bench_cfg3.json"):
    f=os.path.join(sys.argv[1], f)
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        e=j.get("end_to_end") or {}
        print(os.path.basename(f), "ms/step %.4f" % j["ms_per_step"], "sync", j.get("ms_per_step_synchronous"), "e2e %.4f" % e.get("ms_per_batch",0), "rounds", j["stats"].get("rounds"), "classes", j["stats"]["n_classes"], "parity", j.get("parity_vs_cpu_baseline"), "traffic", j["roofline"]["traffic"])
        print("    ", {k: round(v,1) for k,v in j.get("kernels_us_per_step", {}).items()})
    except Exception as ex:
        print(os.path.basename(f), "ERR", ex); print(open(f.replace('.json','.err')).read()[-1500:])
PY
cat $O/walk_probe.txt

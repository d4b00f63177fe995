#!/bin/bash
# Dense matching kernel (4 waves per SIMD where chunks are long) + trimmed loop: whole GPU suite, bench lines.
O=gpurun_out/dense; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
(timeout 1200 python -m pytest tests -m gpu -q -x --timeout 420 -p no:cacheprovider 2>&1 | tail -8) > $O/pytest.log
cat $O/pytest.log
timeout 200 python bench.py > $O/cfg2.json 2> $O/cfg2.err
timeout 200 python bench.py --config cfg3 --steps 300 --warmup 20 > $O/cfg3.json 2> $O/cfg3.err
timeout 200 python bench.py --config cfg4 --steps 100 --warmup 10 > $O/cfg4.json 2> $O/cfg4.err
YDC_DENSE=0 timeout 200 python bench.py --config cfg4 --steps 100 --warmup 10 --resident-only > $O/cfg4_nodense.json 2> $O/cfg4_nodense.err
python - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        e=j.get("end_to_end") or {}
        print("%-16s" % os.path.basename(f)[:-5], "ms/step %.4f" % j["ms_per_step"], "sync", round(j.get("ms_per_step_synchronous") or 0, 4), "e2e", round(e.get("ms_per_batch", 0), 4), "rounds", j["stats"].get("rounds"), "chunks", j["stats"].get("n_chunks"), "parity", j.get("parity_vs_cpu_baseline"), "roofline", j["roofline"]["achieved"], j["roofline"]["frac"])
        print("    ", {k: round(v,1) for k,v in j.get("kernels_us_per_step", {}).items()})
    except Exception as ex:
        print(os.path.basename(f), "ERR", ex); print(open(f.replace('.json','.err')).read()[-800:])
PY
timeout 100 build/fastloop_probe > $O/fastloop_probe.txt 2>&1; grep "waves/SIMD [02]" $O/fastloop_probe.txt

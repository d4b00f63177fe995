#!/bin/bash
# Runs on the GPU box (through gpurun): the parity tests that cover the kernels + the per-kernel
# figures of cfg2 / cfg3 / cfg4 (HIP events, resident batches) into gpurun_out/<tag>/.
# Usage: tools/gpu_kernel_check.sh <tag> [notests] [ENV=VAL ...]   (extra env for the bench runs)
TAG=${1:-kern}; shift
O=gpurun_out/$TAG; mkdir -p $O
if [ "$1" = notests ]; then shift; else
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_binsort_gpu.py tests/test_streaming_gpu.py tests/test_fuzz_gpu.py tests/test_sharded_gpu.py -m gpu -x -q --timeout 600 2>&1 | tail -15) > $O/pytest.log
cat $O/pytest.log
fi
B="--no-cpu-baseline --no-extra-configs --resident-only"
export YDC_BENCH_GUARD=1
env "$@" timeout 120 python bench.py $B --steps 2000 --warmup 100 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
env "$@" timeout 120 python bench.py $B --config cfg3 --steps 200 --warmup 20 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
env "$@" timeout 120 python bench.py $B --config cfg4 --steps 100 --warmup 10 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
env "$@" YDC_SPLIT_GEN=1 timeout 120 python bench.py $B --config cfg4 --steps 50 --warmup 10 > $O/bench_cfg4_split.json 2> $O/bench_cfg4_split.err
env "$@" YDC_SPLIT_GEN=1 timeout 120 python bench.py $B --config cfg3 --steps 50 --warmup 10 > $O/bench_cfg3_split.json 2> $O/bench_cfg3_split.err
python - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        k=j.get("kernels_us_per_step", {}); l=j.get("kernel_launches_per_step", {})
        print("%-22s ms/step %.4f sync %.4f rounds %s  kernels sum %.1f" % (os.path.basename(f)[6:-5], j["ms_per_step"], j.get("ms_per_step_synchronous") or 0, j["stats"].get("rounds"), sum(k.values())))
        print("     ", "  ".join("%s %.1f(x%d)" % (a[2:], b, l.get(a,1)) for a,b in k.items()))
    except Exception as ex:
        print(os.path.basename(f), "ERR", ex); print(open(f.replace('.json','.err')).read()[-800:])
PY

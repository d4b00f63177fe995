#!/bin/bash
# Runs on the GPU box (through gpurun): the parity tests of the sparse-eligibility walk
# (k_walk_groups, both head layouts) + cfg2 with 150 digests timed + where the walk's time goes.
# Usage: tools/gpu_walk_check.sh <tag>
TAG=${1:-walk}; shift
O=gpurun_out/$TAG; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 600 -k "more_than_256 or group_walk or many_classes or wide" 2>&1 | tail -8) > $O/pytest.log
tail -1 $O/pytest.log
B="--no-cpu-baseline --no-extra-configs --resident-only"
export YDC_BENCH_GUARD=1
timeout 200 python bench.py $B --digests 150 --steps 5 --warmup 2 > $O/bench_d150.json 2> $O/bench_d150.err
YDC_WALK_PACKED=0 timeout 200 python bench.py $B --digests 150 --steps 5 --warmup 2 > $O/bench_d150_unpacked.json 2> $O/bench_d150_unpacked.err
python - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        k=j.get("kernels_us_per_step", {})
        print("%-22s ms/step %.4f rounds %s walk %.1f us" % (os.path.basename(f)[6:-5], j["ms_per_step"], j["stats"].get("rounds"), k.get("k_walk_groups", 0)))
    except Exception as ex:
        print(os.path.basename(f), "ERR", ex); print(open(f.replace('.json','.err')).read()[-800:])
PY
YDC_LIB=yadcc_amd/libydc_probe.so timeout 120 python tools/walk_probe.py groups 2>&1 | tee $O/groups.txt

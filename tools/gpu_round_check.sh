#!/bin/bash
# Runs on the GPU box (through gpurun): GPU test suite + the bench lines of the round, into
# gpurun_out/<tag>/. Usage: tools/gpu_round_check.sh <tag> [quick]
TAG=${1:-check}; MODE=${2:-full}
O=gpurun_out/$TAG; mkdir -p $O
(timeout 600 python -m pytest tests -m gpu -x -q --timeout 120 2>&1 | tail -25) > $O/pytest.log
timeout 300 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 300 python bench.py --shared-ip-frac 0.05 --steps 2000 --warmup 100 > $O/bench_cfg2_shared.json 2> $O/bench_cfg2_shared.err
timeout 300 python bench.py --config cfg3 --steps 300 --warmup 20 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
if [ "$MODE" = full ]; then
timeout 300 python bench.py --config cfg4 --steps 100 --warmup 10 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
timeout 300 python bench.py --config cfg5 --steps 1000 --warmup 50 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
YDC_BENCH_FORCE_DIST=1 YDC_BENCH_RCCL_TIMEOUT=60 timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 200 --warmup 20 > $O/bench_dist1.json 2> $O/bench_dist1.err
fi
cat $O/pytest.log
python - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        e=j.get("end_to_end") or {}
        print(os.path.basename(f), "ms/step %.4f" % j["ms_per_step"], "p99 %.4f" % j["p99_dispatch_latency_ms"],
              "e2e ms %.4f" % e.get("ms_per_batch", 0), "rounds", j["stats"].get("rounds"),
              "parity", j.get("parity_vs_cpu_baseline"), j.get("parity_vs_oracle"), "rccl", j.get("rccl_ranks"))
        print("    ", {k: round(v,1) for k,v in j.get("kernels_us_per_step", {}).items()})
    except Exception as ex:
        print(os.path.basename(f), "ERR", ex); print(open(f.replace('.json','.err')).read()[-1200:])
PY

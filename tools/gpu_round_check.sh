mkdir -p gpurun_out/r2b; O=gpurun_out/r2b
(timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/pytest.log
timeout 300 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 300 python bench.py --config cfg3 --steps 300 --warmup 20 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
timeout 300 python bench.py --config cfg5 --steps 1000 --warmup 50 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
YDC_BENCH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 200 --warmup 20 > $O/bench_dist1.json 2> $O/bench_dist1.err
for cs in 128 256 512 1024; do YDC_CHUNK_SIZE=$cs timeout 200 python bench.py --config cfg3 --steps 200 --warmup 20 --no-cpu-baseline > $O/cfg3_cs$cs.json 2>/dev/null; done
for cs in 64 128; do YDC_CHUNK_SIZE=$cs timeout 200 python bench.py --steps 2000 --warmup 100 --no-cpu-baseline > $O/cfg2_cs$cs.json 2>/dev/null; done
cat $O/pytest.log
for f in $O/bench_cfg2.json $O/bench_cfg3.json $O/bench_cfg5.json $O/bench_dist1.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    keys=("value","ms_per_step","p99_dispatch_latency_ms","end_to_end","parity_vs_cpu_baseline","parity_vs_oracle","rccl_ranks","rccl","kernels_us_per_step","roofline","cpu_baseline")
    print({k:j.get(k) for k in keys if k in j})
except Exception as e:
    print("ERR",e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
for f in $O/cfg3_cs*.json $O/cfg2_cs*.json; do python - "$f" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], j["ms_per_step"], j["stats"]["rounds"], j["kernels_us_per_step"].get("k_match_pass"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done

#!/bin/bash
# Runs on the GPU box (through gpurun): the bench-contract tests and the driver's own command
# lines (N = 1 and N = 2 without a launcher) into gpurun_out/<tag>/. Usage: tools/gpu_bench_check.sh <tag>
TAG=${1:-bench}
O=gpurun_out/$TAG; mkdir -p $O
(timeout 900 python -m pytest tests/test_bench_contract.py -m gpu -x -q --timeout 900 2>&1 | tail -40) > $O/pytest.log
( time timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err
( time timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 ) > $O/bench_gpus2.json 2> $O/bench_gpus2.err
cat $O/pytest.log
tail -n 5 $O/bench_driver.err $O/bench_gpus2.err
python - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "n_gpus", j["n_gpus"], "ms/step %.4f" % j["ms_per_step"], "sync", j.get("ms_per_step_synchronous"),
              "parity", j.get("parity_vs_cpu_baseline"), j.get("parity_vs_oracle"), "transport", j.get("transport"))
        for k, c in (j.get("configs") or {}).items():
            print("   ", k, "ms/step %.4f" % c["ms_per_step"], "p99 %.4f" % c["p99_dispatch_latency_ms"], "fixture", c.get("parity_vs_reference_fixture"), {a: round(b,1) for a,b in (c.get("kernels_us_per_step") or {}).items()})
        if j.get("strong_cfg4"):
            c = j["strong_cfg4"]; print("    strong_cfg4 ms/step %.4f" % c["ms_per_step"], "parity", c.get("parity_vs_oracle"))
        if j.get("steady_state_commit"): print("    steady", j["steady_state_commit"]["ms_per_step"], j["steady_state_commit"]["registry_restored"])
    except Exception as ex:
        print(os.path.basename(f), "ERR", ex)
PY

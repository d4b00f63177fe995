#!/bin/bash
# Runs on the GPU box: the host-class tests on the device + the driver's bench line (td_surface).
TAG=${1:-td}
O=gpurun_out/$TAG; mkdir -p $O
(timeout 900 python -m pytest tests/test_task_dispatcher_gpu.py tests/test_bench_contract.py -m gpu -x -q --timeout 600 2>&1 | tail -30) > $O/pytest.log
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err
cat $O/pytest.log; tail -n 5 $O/bench_driver.err
python - $O <<'PY'
import json,sys,os
j=json.loads(open(os.path.join(sys.argv[1],"bench_driver.json")).read().strip().splitlines()[-1])
print(json.dumps(j.get("td_surface"), indent=1))
PY

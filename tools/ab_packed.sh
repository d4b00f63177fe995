O=gpurun_out/r2i; mkdir -p $O
(timeout 500 python -m pytest tests -m gpu -x -q --timeout 120 2>&1 | tail -8) > $O/pytest.log
(YDC_PACKED_SORT=0 timeout 300 python -m pytest tests/test_gpu_parity.py -x -q --timeout 120 2>&1 | tail -3) > $O/pytest_unpacked.log
cat $O/pytest.log $O/pytest_unpacked.log
for cfg in cfg2 cfg3 cfg4; do for pk in 1 0; do
  st=300; [ $cfg = cfg2 ] && st=3000; [ $cfg = cfg4 ] && st=100
  YDC_PACKED_SORT=$pk timeout 200 python bench.py --config $cfg --steps $st --warmup 20 --no-cpu-baseline > $O/${cfg}_pk$pk.json 2>/dev/null
  python - $O/${cfg}_pk$pk.json $cfg $pk <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], "packed", sys.argv[3], "ms %.4f" % j["ms_per_step"], {k: round(v,1) for k,v in j["kernels_us_per_step"].items()})
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done; done

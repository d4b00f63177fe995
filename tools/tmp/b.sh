run() { echo "$*"; timeout 200 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=j.get('end_to_end') or {}
print(round(j['ms_per_step']*1e3,1), 'p99', round(j.get('p99_dispatch_latency_ms',0)*1e3,1), 'e2e', round(e.get('ms_per_batch',0)*1e3,1), j['stats'].get('rounds'), j['stats'].get('radix_passes'), {k: round(v,1) for k,v in j.get('kernels_us_per_step',{}).items()})"; }
run --config cfg3 --steps 300 --warmup 20
YDC_FUSE_PASSES=0 run --config cfg3 --steps 300 --warmup 20
run --config cfg4 --steps 100 --warmup 10
run --config cfg5 --steps 500 --warmup 50
run --shared-ip-frac 0.05 --steps 1000 --warmup 100

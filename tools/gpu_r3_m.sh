#!/bin/bash
# Pairs on / off with the trimmed loop.
O=gpurun_out/pairab; mkdir -p $O
for cfg in cfg3 cfg4; do for pr in 1 0; do
  YDC_PAIR=$pr timeout 200 python bench.py --config $cfg --steps 150 --warmup 10 --resident-only --no-cpu-baseline > $O/${cfg}_pair$pr.json 2> $O/${cfg}_pair$pr.err
done; done
python - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    j=json.loads(open(f).read().strip().splitlines()[-1])
    print("%-14s" % os.path.basename(f)[:-5], "ms/step %.4f" % j["ms_per_step"], "rounds", j["stats"].get("rounds"), "match %.1f" % j.get("kernels_us_per_step", {}).get("k_match_pass", 0))
PY

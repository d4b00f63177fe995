#!/bin/bash
# Occupancy of the matching kernel: VGPR cap (3 / 4 / 5 waves per SIMD) x chunk count x ring size.
O=gpurun_out/occ; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { # name lib chunks rings cfg steps
  local lib=$2; [ "$lib" != "-" ] && export YDC_LIB=$PWD/build/$lib || unset YDC_LIB
  YDC_TARGET_CHUNKS=$3 YDC_RING_TOTAL=$4 timeout 200 python bench.py --config $5 --steps $6 --warmup 10 --resident-only > $O/$1.json 2> $O/$1.err
}
for cfg in cfg3 cfg4; do
  run ${cfg}_w3_c2048_r2048 - 2048 2048 $cfg 100
  run ${cfg}_w3_c4096_r1024 - 4096 1024 $cfg 100
  run ${cfg}_w4_c2048_r2048 libydc_w4.so 2048 2048 $cfg 100
  run ${cfg}_w4_c4096_r1024 libydc_w4.so 4096 1024 $cfg 100
  run ${cfg}_w5_c4096_r1024 libydc_w5.so 4096 1024 $cfg 100
  run ${cfg}_w5_c8192_r512 libydc_w5.so 8192 512 $cfg 100
done
run cfg2_w3 - 2048 2048 cfg2 2000
run cfg2_w4 libydc_w4.so 2048 2048 cfg2 2000
run cfg2_w4_c4096 libydc_w4.so 4096 1024 cfg2 2000
python - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print("%-28s" % os.path.basename(f)[:-5], "ms/step %.4f" % j["ms_per_step"], "rounds", j["stats"].get("rounds"), "chunks", j["stats"].get("n_chunks"), "parity", j.get("parity_vs_cpu_baseline"), "match %.1f" % j.get("kernels_us_per_step", {}).get("k_match_pass", 0))
    except Exception as ex:
        print(os.path.basename(f), "ERR", ex); print(open(f.replace('.json','.err')).read()[-800:])
PY

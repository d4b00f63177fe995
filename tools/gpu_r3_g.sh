#!/bin/bash
# Round-3: rocprofv3 kernel stats + HBM PMC passes + SQ counters of the HBM-resident batches only.
O=gpurun_out/final; mkdir -p $O
timeout 300 bash tools/profile.sh cfg2 > $O/profile_cfg2.log 2>&1
timeout 400 bash tools/profile.sh cfg3 --config cfg3 > $O/profile_cfg3.log 2>&1
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
for c in cfg2 cfg3; do
  X=""; [ $c = cfg3 ] && X="--config cfg3"
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $ROOT/$O/sq1_$c -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --resident-only $X > /dev/null 2> $ROOT/$O/sq1_$c.log
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD -d $ROOT/$O/sq2_$c -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --resident-only $X > /dev/null 2> $ROOT/$O/sq2_$c.log
done
cd $ROOT
for c in cfg2 cfg3; do for k in 1 2; do
  D=$(find $O/sq${k}_$c -name "*.db" | head -1)
  [ -n "$D" ] && python tools/rocprof_summary.py pmc "$D" > $O/sq${k}_$c.txt 2>&1
  rm -rf $O/sq${k}_$c
done; done
head -10 gpurun_out/prof_cfg2/kernel_stats.txt; head -12 gpurun_out/prof_cfg3/kernel_stats.txt
grep -h "k_match" $O/sq1_cfg2.txt $O/sq2_cfg2.txt

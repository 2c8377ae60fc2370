#!/bin/bash
# HBM traffic of the shipped attention kernel at the cfg2 L0 shape: separate single-counter --pmc passes (MI355X_MICROARCH.md, HBM section)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03t; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$c
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/prof_$c -- python $GRAFT_REPO_ROOT/tools/attn_l0_once.py > /dev/null 2>&1
  f=$(find /tmp/prof_$c -name "*counter_collection.csv" | head -1)
  echo "+ $c" >> $OUT/attn_pmc_v5.txt
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $f | grep -A2 flash_attn >> $OUT/attn_pmc_v5.txt
done
cat $OUT/attn_pmc_v5.txt

#!/bin/bash
# SQ counters of the attention kernels at the L0 shape (one pass of 8 SQ counters + GRBM_GUI_ACTIVE per variant)
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in $2; do
  rm -rf /tmp/pmc_$v
  timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d /tmp/pmc_$v -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/attn_once.py $v > /tmp/pmc_$v.log 2>&1
  f=$(find /tmp/pmc_$v -name "*counter_collection.csv" | head -1)
  echo "== variant $v" >> $OUT/pmc.txt
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $f | grep -A12 flash_attn >> $OUT/pmc.txt 2>&1 || tail -5 /tmp/pmc_$v.log >> $OUT/pmc.txt
done
cat $OUT/pmc.txt

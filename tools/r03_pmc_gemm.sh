#!/bin/bash
# SQ / cache counters of GEMM variants at one shape: r03_pmc_gemm.sh OUTDIR "1 27 vendor" M N K
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
M=${3:-8192}; N=${4:-8192}; K=${5:-8192}
for v in $2; do
  echo "== $v  ($M x $N x $K)" >> $OUT/pmc.txt
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM GRBM_GUI_ACTIVE" \
             "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
             "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr"; do
    i=$((i+1)); rm -rf /tmp/pmc_${v}_$i
    timeout 120 rocprofv3 --pmc $set -d /tmp/pmc_${v}_$i -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/gemm_pmc_once.py $M $N $K $v > /tmp/pmc_${v}_$i.log 2>&1
    f=$(find /tmp/pmc_${v}_$i -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python $GRAFT_REPO_ROOT/tools/pmc_summary.py $f | grep -A12 "gemm_kernel\|Cijk" | grep -v "^--" >> $OUT/pmc.txt 2>&1; else echo "  pass $i failed: $(grep -i "error\|invalid\|not found" /tmp/pmc_${v}_$i.log | head -2)" >> $OUT/pmc.txt; fi
  done
done
cat $OUT/pmc.txt

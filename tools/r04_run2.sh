#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04b; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( time timeout 600 python tools/ab_gemm_r04.py plain 2>&1 | grep -v amdgpu.ids ) > $OUT/gemm_ab_plain.txt 2>&1
( time timeout 300 python -m pytest tests/test_kernels.py -q -m gpu -k "persistent_tile or group_norm or tail_split" 2>&1 | grep -v amdgpu.ids | tail -5 ) > $OUT/pytest_kernels.txt 2>&1
cat $OUT/gemm_ab_plain.txt; tail -3 $OUT/pytest_kernels.txt

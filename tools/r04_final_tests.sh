#!/bin/bash
# round-4 closing evidence, part 1: the cfg2-geometry parity test on its own (its lines are kept), then the whole GPU suite
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04y; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( time timeout 600 python -m pytest tests/test_parity_cfg2.py -q -m gpu -s 2>&1 | grep -v amdgpu.ids | tail -14 ) > $OUT/pytest_parity_cfg2.txt 2>&1
( time timeout 1500 python -m pytest tests -q -m gpu --durations=14 --deselect tests/test_parity_cfg2.py 2>&1 | grep -v amdgpu.ids | tail -40 ) > $OUT/pytest_gpu.txt 2>&1
cat $OUT/pytest_parity_cfg2.txt; tail -22 $OUT/pytest_gpu.txt

#!/bin/bash
# full per-shape forward tables of the round-3 tree and the round-4 tree on ONE box
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04w; mkdir -p $OUT
( cd $GRAFT_REPO_ROOT/build/r03tree && timeout 300 python tools/profile_forward.py 2>&1 | grep -v amdgpu.ids ) > $OUT/forward_r03.txt 2>&1
( cd $GRAFT_REPO_ROOT && timeout 300 python tools/profile_forward.py 2>&1 | grep -v amdgpu.ids ) > $OUT/forward_r04.txt 2>&1
head -2 $OUT/forward_r03.txt; head -2 $OUT/forward_r04.txt

#!/bin/bash
# round 4, fourth GPU call: packed GELU + per-frame GroupNorm fold + refined tile-18 selection: GEMM A/B (GEGLU shapes), forward profile, A/B of the fold
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04d; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( time timeout 300 python -m pytest tests/test_kernels.py -q -m gpu -k "geglu or persistent_tile or group_norm or tail_split or folded_layer" 2>&1 | grep -v amdgpu.ids | tail -5 ) > $OUT/pytest_kernels.txt 2>&1
( timeout 300 python tools/ab_gemm_r04.py plain geglu 2>&1 | grep -v amdgpu.ids ) > $OUT/gemm_ab_geglu.txt 2>&1
( timeout 200 python tools/ab_gemm_astat.py 2>&1 | grep -v amdgpu.ids ) > $OUT/gemm_astat_ab.txt 2>&1
for v in default nofold; do
  unset STAR_GN_NOFOLD
  [ $v = nofold ] && export STAR_GN_NOFOLD=1
  ( timeout 300 python tools/profile_forward.py 2>&1 | grep -v amdgpu.ids ) > $OUT/forward_$v.txt 2>&1
done
unset STAR_GN_NOFOLD
tail -3 $OUT/pytest_kernels.txt; cat $OUT/gemm_ab_geglu.txt; tail -12 $OUT/gemm_astat_ab.txt; for v in default nofold; do echo $v; head -2 $OUT/forward_$v.txt; grep -E "group_norm|843264   2560    320" $OUT/forward_$v.txt | head -3; done

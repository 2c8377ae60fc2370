#!/bin/bash
# round 4, third GPU call: tile 18 as the automatic choice + GroupNorm finalize fold, A/B on one box (forward profiles with each switched off)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04c; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( time timeout 300 python -m pytest tests/test_kernels.py -q -m gpu -k "persistent_tile or group_norm or tail_split or gn" 2>&1 | grep -v amdgpu.ids | tail -5 ) > $OUT/pytest_kernels.txt 2>&1
( timeout 300 python tools/ab_gemm_r04.py plain geglu,rowaff 2>&1 | grep -v amdgpu.ids ) > $OUT/gemm_ab_plain.txt 2>&1
for v in default nofold nopersist default2; do
  unset STAR_GN_NOFOLD STAR_NO_PERSIST
  [ $v = nofold ] && export STAR_GN_NOFOLD=1
  [ $v = nopersist ] && export STAR_NO_PERSIST=1
  ( timeout 300 python tools/profile_forward.py 2>&1 | grep -v amdgpu.ids ) > $OUT/forward_$v.txt 2>&1
done
unset STAR_GN_NOFOLD STAR_NO_PERSIST
( time timeout 600 python -m pytest tests/test_unet.py tests/test_vae.py -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -5 ) > $OUT/pytest_unet.txt 2>&1
tail -3 $OUT/pytest_kernels.txt; cat $OUT/gemm_ab_plain.txt; for v in default nofold nopersist default2; do echo $v; head -2 $OUT/forward_$v.txt; grep -E "group_norm" $OUT/forward_$v.txt | head -2; done; tail -3 $OUT/pytest_unet.txt

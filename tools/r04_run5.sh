#!/bin/bash
# round 4, GPU call: the fused temporal projection + attention kernel (gemm_tq.h): parity on hardware, forward A/B with STAR_NO_TQ=1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04f; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( time timeout 200 python -m pytest tests/test_kernels.py -q -m gpu -k "fused or temporal_attention" 2>&1 | grep -v amdgpu.ids | tail -6 ) > $OUT/pytest_tq.txt 2>&1
if ! grep -q "passed" $OUT/pytest_tq.txt || grep -q "failed" $OUT/pytest_tq.txt; then cat $OUT/pytest_tq.txt; echo "fused-kernel tests not green: stopping"; exit 1; fi
for v in default notq; do
  unset STAR_NO_TQ
  [ $v = notq ] && export STAR_NO_TQ=1
  ( timeout 300 python tools/profile_forward.py 2>&1 | grep -v amdgpu.ids ) > $OUT/forward_$v.txt 2>&1
done
unset STAR_NO_TQ
( time timeout 600 python -m pytest tests/test_unet.py -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -5 ) > $OUT/pytest_unet.txt 2>&1
cat $OUT/pytest_tq.txt | tail -4; for v in default notq; do echo $v; head -2 $OUT/forward_$v.txt; grep -E "temporal_attn|843264    960    320" $OUT/forward_$v.txt | head -3; done; tail -3 $OUT/pytest_unet.txt

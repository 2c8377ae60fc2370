#!/bin/bash
# tile 17 on hardware: its parity tests, the gathered-layer A/B (auto vs forced tiles), and the forward with / without it
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_sched; mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -k "scheduled_tile or conv3x3 or temporal_conv or bias_residual" 2>&1 | tail -4 > $OUT/pytest.txt
timeout 300 python tools/ab_conv_tiles.py 0,2,17 > $OUT/ab_conv.txt 2>&1
for rep in 1 2; do
  STAR_NO_SCHED=1 timeout 300 python tools/profile_forward.py 2>&1 | grep "forward wall" >> $OUT/forward_nosched.txt
  timeout 300 python tools/profile_forward.py 2>&1 | grep "forward wall" >> $OUT/forward_sched.txt
done
timeout 300 python tools/profile_forward.py > $OUT/forward_detail.txt 2>&1
cat $OUT/pytest.txt $OUT/ab_conv.txt; echo "--- no sched"; cat $OUT/forward_nosched.txt; echo "--- sched"; cat $OUT/forward_sched.txt

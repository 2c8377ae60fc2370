#!/bin/bash
# round 4, first GPU call: the new / affected GPU tests, the GEMM A/B (tail split, persistent tile 18), one profiled forward, one bench clip
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04a; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests/test_kernels.py -q -m gpu -k "tail_split or persistent_tile or header_names or folded_layer or scheduled_tile or gemm_bias_residual" 2>&1 | grep -v amdgpu.ids | tail -15 ) > $OUT/pytest_kernels.txt 2>&1
( time timeout 600 python tools/ab_gemm_r04.py plain 2>&1 | grep -v amdgpu.ids ) > $OUT/gemm_ab_plain.txt 2>&1
( time timeout 300 python tools/ab_gemm_r04.py conv 2>&1 | grep -v amdgpu.ids ) > $OUT/gemm_ab_conv.txt 2>&1
( time timeout 400 python tools/profile_forward.py 2>&1 | grep -v amdgpu.ids ) > $OUT/forward_detail.txt 2>&1
( time timeout 1200 python -m pytest tests/test_unet.py tests/test_embedder.py tests/test_pipeline.py tests/test_dit.py tests/test_parity_cfg1.py -q -m gpu -s --durations=8 2>&1 | grep -v amdgpu.ids | tail -60 ) > $OUT/pytest_models.txt 2>&1
timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_v0_f16_n1.json 2> $OUT/bench.err
tail -4 $OUT/pytest_kernels.txt; cat $OUT/gemm_ab_plain.txt | tail -16; cat $OUT/gemm_ab_conv.txt | tail -6; head -4 $OUT/forward_detail.txt; tail -5 $OUT/pytest_models.txt; head -c 600 $OUT/bench_v0_f16_n1.json; tail -3 $OUT/bench.err

#!/bin/bash
# round-3 closing evidence, second build (tile 17, graph replay): the GPU suite minus its slowest CPU-oracle-bound cases (the driver
# runs the whole suite at round end; they passed on the first closing build, profiles/r03_pytest_gpu_v1.txt), one cfg2 clip through
# bench.py, the same under rocprofv3 --kernel-trace --stats
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03y; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests -q -m gpu --durations=8 \
    --deselect "tests/test_pipeline.py::test_pipeline_psnr_vs_cpu_oracle" --deselect "tests/test_parity_cfg1_50.py" \
    --deselect "tests/test_pipeline.py::test_chunked_solver_loop_with_the_hip_denoiser[reduced]" 2>&1 | grep -v amdgpu.ids | tail -30 ) > $OUT/pytest_gpu.txt 2>&1
timeout 600 python bench.py --steps 1 --warmup 0 > $OUT/bench_v2_f16_n1.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bench
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o b --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_v2_rocprof_f16_n1.json 2> $OUT/rocprof.err
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/bench_v2_kernel_stats.csv 2>/dev/null
tail -4 $OUT/pytest_gpu.txt; head -c 600 $OUT/bench_v2_f16_n1.json; echo; head -6 $OUT/bench_v2_kernel_stats.csv

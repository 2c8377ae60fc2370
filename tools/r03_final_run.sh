#!/bin/bash
# round-3 closing evidence: whole GPU test suite, one cfg2 clip through bench.py, the same under rocprofv3 --kernel-trace --stats
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03z; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests -q -m gpu --durations=12 2>&1 | grep -v amdgpu.ids | tail -40 ) > $OUT/pytest_gpu.txt 2>&1
timeout 900 python bench.py --steps 1 --warmup 0 > $OUT/bench_v1_f16_n1.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bench
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o b --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_v1_rocprof_f16_n1.json 2> $OUT/rocprof.err
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/bench_v1_kernel_stats.csv 2>/dev/null
tail -3 $OUT/pytest_gpu.txt; head -c 400 $OUT/bench_v1_f16_n1.json; echo; head -5 $OUT/bench_v1_kernel_stats.csv

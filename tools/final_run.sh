set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/final
timeout 700 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > gpurun_out/final/pytest_gpu.txt
cat gpurun_out/final/pytest_gpu.txt
rm -rf /tmp/prof_bench
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
tail -3 gpurun_out/final/bench.err
cat gpurun_out/final/bench.json | tail -2 | cut -c1-1500
find /tmp/prof_bench -name "*kernel_stats.csv" -exec cp {} gpurun_out/final/kernel_stats.csv \;
ls -la gpurun_out/final

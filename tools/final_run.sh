# (FETCH_SIZE and WRITE_SIZE go in separate single-counter passes: a pass with FETCH_SIZE + TCC_HIT_sum + TCC_MISS_sum produced no output on this pool)
# Round-end GPU run: full GPU test suite, bench under rocprofv3 (kernel trace + stats), PMC traffic passes of the
# dominant kernel.  Usage on the GPU box: bash tools/final_run.sh   (outputs under gpurun_out/final/)
set -x
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out/final
timeout 700 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > gpurun_out/final/pytest_gpu.txt
cat gpurun_out/final/pytest_gpu.txt
rm -rf /tmp/prof_bench /tmp/prof_fetch /tmp/prof_write
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
tail -3 gpurun_out/final/bench.err
tail -2 gpurun_out/final/bench.json | cut -c1-600
find /tmp/prof_bench -name "*kernel_stats.csv" -exec cp {} gpurun_out/final/kernel_stats.csv \;
timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_fetch -- python tools/attn_l0_once.py > /dev/null 2>&1
find /tmp/prof_fetch -name "*counter_collection.csv" -exec cp {} /tmp/fetch.csv \;
timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_write -- python tools/attn_l0_once.py > /dev/null 2>&1
find /tmp/prof_write -name "*counter_collection.csv" -exec cp {} /tmp/write.csv \;
(python tools/pmc_summary.py /tmp/fetch.csv; python tools/pmc_summary.py /tmp/write.csv) > gpurun_out/final/attn_pmc.txt 2>&1
cat gpurun_out/final/attn_pmc.txt
ls -la gpurun_out/final

# Round-2 closing GPU run: the bench line (no profiler), the same command under rocprofv3 --kernel-trace --stats, single-counter
# PMC traffic passes of the spatial self-attention kernel, and the PSNR lines of the full-width parity tests.
# Usage on the GPU box: bash tools/r02_final_run.sh   (outputs under gpurun_out/final/)
set -x
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out/final
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
tail -2 gpurun_out/final/bench.err
cut -c1-700 gpurun_out/final/bench.json
rm -rf /tmp/prof_bench /tmp/prof_fetch /tmp/prof_write
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python bench.py --steps 1 --no-cpu-baseline > gpurun_out/final/bench_prof.json 2> gpurun_out/final/bench_prof.err
cut -c1-300 gpurun_out/final/bench_prof.json
find /tmp/prof_bench -name "*kernel_stats.csv" -exec cp {} gpurun_out/final/kernel_stats.csv \;
head -12 gpurun_out/final/kernel_stats.csv | cut -c1-200
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_fetch -- python tools/attn_l0_once.py > /dev/null 2>&1
find /tmp/prof_fetch -name "*counter_collection.csv" -exec cp {} /tmp/fetch.csv \;
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_write -- python tools/attn_l0_once.py > /dev/null 2>&1
find /tmp/prof_write -name "*counter_collection.csv" -exec cp {} /tmp/write.csv \;
(python tools/pmc_summary.py /tmp/fetch.csv; python tools/pmc_summary.py /tmp/write.csv) > gpurun_out/final/attn_pmc.txt 2>&1
cat gpurun_out/final/attn_pmc.txt
timeout 600 python -m pytest tests/test_parity_cfg1.py tests/test_pipeline.py -q -m gpu -s 2>&1 | grep -i "psnr\|passed\|failed" > gpurun_out/final/parity_psnr.txt
cat gpurun_out/final/parity_psnr.txt

# round 2, GPU call: full-width cfg1 parity, chunked loop, new kernels paths; cfg4 (1 step) and cfg3 (fast) on hardware
set -x
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r02_probe2
mkdir -p $O
timeout 1200 python -m pytest tests/test_parity_cfg1.py tests/test_pipeline.py tests/test_cli.py tests/test_vae.py -q -m gpu -x -s 2>&1 | grep -v amdgpu.ids | tail -25 > $O/pytest_parity.txt
cat $O/pytest_parity.txt
timeout 600 python -m pytest tests/test_kernels.py tests/test_abi.py -q -m gpu -x -k "temporal or rejects or abi or gemm_bias" 2>&1 | tail -4 > $O/pytest_kernels.txt
cat $O/pytest_kernels.txt
timeout 900 python bench.py --config cfg4 --denoise-steps 1 --no-cpu-baseline > $O/bench_cfg4_1step.json 2> $O/bench_cfg4.err; tail -3 $O/bench_cfg4.err; cut -c1-1500 $O/bench_cfg4_1step.json
timeout 900 python bench.py --config cfg3 --no-cpu-baseline > $O/bench_cfg3_fast.json 2> $O/bench_cfg3.err; tail -3 $O/bench_cfg3.err; cut -c1-1500 $O/bench_cfg3_fast.json
ls -la $O

set -x
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out/r02_tests
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -12 > gpurun_out/r02_tests/pytest_gpu.txt
cat gpurun_out/r02_tests/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 > gpurun_out/r02_tests/smoke.txt; cat gpurun_out/r02_tests/smoke.txt

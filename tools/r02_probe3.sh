set -x
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r02_probe3
mkdir -p $O
timeout 900 python bench.py --config cfg4 --denoise-steps 1 --no-cpu-baseline > $O/bench_cfg4_1step.json 2> $O/bench_cfg4.err; tail -3 $O/bench_cfg4.err; cut -c1-2500 $O/bench_cfg4_1step.json
timeout 900 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err; tail -3 $O/bench_cfg2.err; cut -c1-3000 $O/bench_cfg2.json

# Round-2 closing GPU run, part 2: the other single-GPU BASELINE configurations with the closing build (cfg3 `fast`, cfg4 one
# evaluation), and LDS / issue PMC of the rewritten GEMM loop (old build beside it) at 8192^3 and at a short-K layer shape.
set -x
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/final2
mkdir -p $O
timeout 500 python bench.py --config cfg3 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err; cut -c1-400 $O/bench_cfg3.json
timeout 500 python bench.py --config cfg4 --denoise-steps 1 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err; cut -c1-400 $O/bench_cfg4.json
for cfg in "8192 8192 8192" "843264 960 320"; do
  tag=$(echo $cfg | tr ' ' 'x')
  rm -rf /tmp/pmc_a /tmp/pmc_b
  timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/pmc_a -- python tools/gemm_once.py $cfg 0 > /dev/null 2>&1
  find /tmp/pmc_a -name "*counter_collection.csv" -exec cp {} /tmp/pa.csv \;
  timeout 200 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM --kernel-trace --output-format csv -d /tmp/pmc_b -- python tools/gemm_once.py $cfg 0 > /dev/null 2>&1
  find /tmp/pmc_b -name "*counter_collection.csv" -exec cp {} /tmp/pb.csv \;
  (echo "== $cfg (closing build, auto tile)"; python tools/pmc_summary.py /tmp/pa.csv | grep -A9 gemm_kernel; python tools/pmc_summary.py /tmp/pb.csv | grep -A9 gemm_kernel) > $O/pmc_$tag.txt 2>&1
  cat $O/pmc_$tag.txt
done
ls -la $O

# round 2, GPU call 1: gemm8 on hardware -- parity, race screen, A/B against the 2-stage tiles, LDS / issue PMC
set -x
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r02_probe1
mkdir -p $O
timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -x -k "gemm8" 2>&1 | tail -5 > $O/pytest_gemm8.txt
cat $O/pytest_gemm8.txt
timeout 400 python tools/race_gemm8.py 10 > $O/race.txt 2>&1; tail -12 $O/race.txt
timeout 600 python tools/bench_kernels.py --dtype f16 --only gemm,conv --tiles 1,2,20 > $O/kbench.txt 2>&1; python tools/fmt_kbench.py $O/kbench.txt 2>/dev/null | tail -80 || tail -80 $O/kbench.txt
for cfg in "8192 8192 8192" "843264 2560 320" "55296 3840 1280"; do
  tag=$(echo $cfg | tr ' ' 'x')
  rm -rf /tmp/pmc_a /tmp/pmc_b
  timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/pmc_a -- python tools/gemm_once.py $cfg 1,20 > /dev/null 2>&1
  find /tmp/pmc_a -name "*counter_collection.csv" -exec cp {} /tmp/pa.csv \;
  timeout 200 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM --kernel-trace --output-format csv -d /tmp/pmc_b -- python tools/gemm_once.py $cfg 1,20 > /dev/null 2>&1
  find /tmp/pmc_b -name "*counter_collection.csv" -exec cp {} /tmp/pb.csv \;
  (echo "== $cfg"; python tools/pmc_summary.py /tmp/pa.csv; python tools/pmc_summary.py /tmp/pb.csv) > $O/pmc_$tag.txt 2>&1
  cat $O/pmc_$tag.txt
done
ls -la $O

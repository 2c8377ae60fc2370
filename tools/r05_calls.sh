#!/bin/bash
# Round-5 GPU calls: each block is ONE gpurun call; run from the
# repo root on the GPU box (gpurun -- 'bash tools/r05_calls.sh 1').  Outputs under gpurun_out/ (copy what is quoted into profiles/).
set -u
mkdir -p gpurun_out
case "${1:-1}" in
  1)  # MFMA shadow: the other instruction kinds of the K / key loops, and two waves per SIMD
      timeout 60 ./tools/probe/mfma_valu_overlap 20000 1 > gpurun_out/r05_probe_mfma_valu_overlap2.txt 2>&1
      tail -5 gpurun_out/r05_probe_mfma_valu_overlap2.txt ;;
  2)  # what the A-stationary kernel waits for: PMC passes on a torch-free run (counters in their own runs, kernel trace only)
      cd /tmp && export TMPDIR=/tmp
      R=${GRAFT_REPO_ROOT:-/root/repo}
      printf 'gemm 843264 2560 320 37 30\ngemm 843264 960 320 33 30\n' > /tmp/astat.txt
      for pmc in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES" \
                 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES"; do
        tag=$(echo "$pmc" | cut -d' ' -f1)
        timeout 120 rocprofv3 --kernel-trace --pmc $pmc -d $R/gpurun_out/r05_pmc_astat_$tag -- \
          $R/tools/cbench/cbench $R/tools/bench/libstar_hip_bench.so f16 /tmp/astat.txt 2 > $R/gpurun_out/r05_pmc_astat_$tag.log 2>&1
      done
      ls $R/gpurun_out | tail ;;
  4)  # A-stationary kernel without its global stores (hypothesis: in-order vmcnt couples the W-tile wait to the stores' latency)
      timeout 60 ./tools/cbench/cbench tools/bench/libstar_hip_bench.so f16 tools/cbench/astat_nostore.txt 8 > gpurun_out/r05_cbench_astat_nostore.txt 2>&1
      cat gpurun_out/r05_cbench_astat_nostore.txt ;;
  3)  # the tile sweep on the current tree (21 s)
      timeout 90 ./tools/cbench/cbench tools/bench/libstar_hip_bench.so f16 tools/cbench/cfg2_tile_sweep.txt 6 > gpurun_out/r05_cbench_tile_sweep.txt 2>&1
      tail -5 gpurun_out/r05_cbench_tile_sweep.txt ;;
  5)  # A-stationary product (nt stores + staggered start) against the round-4 kernel; the new GPU tests; forward A/B of the GroupNorm
      # statistics in the producers' epilogues (same box, same process order)
      timeout 60 ./tools/cbench/cbench tools/bench/libstar_hip_bench.so f16 tools/cbench/astat_product.txt 8 > gpurun_out/r05_cbench_astat_product.txt 2>&1
      cat gpurun_out/r05_cbench_astat_product.txt
      timeout 900 python -m pytest tests/test_kernels.py tests/test_unet.py tests/test_parity_cfg4.py -m gpu -x -q -k "producer_epilogue or statistics_from or large_group_means or cfg4 or blocks_match or small_unet or gemm_folded or persistent_tile" 2>&1 | tail -15 > gpurun_out/r05_pytest_gpu_new.txt
      cat gpurun_out/r05_pytest_gpu_new.txt
      ( timeout 400 python tools/profile_forward.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r05_forward_detail_f16_gnepi.txt 2>&1
      ( STAR_NO_GNEPI=1 timeout 400 python tools/profile_forward.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r05_forward_detail_f16_nognepi.txt 2>&1
      head -3 gpurun_out/r05_forward_detail_f16_gnepi.txt; grep group_norm gpurun_out/r05_forward_detail_f16_gnepi.txt | head -3
      head -3 gpurun_out/r05_forward_detail_f16_nognepi.txt; grep group_norm gpurun_out/r05_forward_detail_f16_nognepi.txt | head -3 ;;
  6)  # GroupNorm family after the split finalize: forward table + per-kernel rocprof stats of the same forward
      timeout 600 python -m pytest tests/test_kernels.py tests/test_unet.py -m gpu -x -q -k "producer_epilogue or statistics_from or large_group_means" 2>&1 | tail -5 > gpurun_out/r05_pytest_gpu_new2.txt
      cat gpurun_out/r05_pytest_gpu_new2.txt
      ( timeout 400 python tools/profile_forward.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r05_forward_detail_f16_gnepi2.txt 2>&1
      head -3 gpurun_out/r05_forward_detail_f16_gnepi2.txt; grep group_norm gpurun_out/r05_forward_detail_f16_gnepi2.txt | head -3
      R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp && export TMPDIR=/tmp
      timeout 500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05_prof_fwd -o fwd -- python $R/tools/profile_forward.py --reps 1 > $R/gpurun_out/r05_prof_fwd.log 2>&1
      ls $R/gpurun_out/r05_prof_fwd* | head ;;
  7)  # checkpoint: the whole GPU suite + smoke + a short bench line (1 warm-up clip with the per-family table, 2 timed clips)
      ( time timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > gpurun_out/r05_pytest_gpu_v1.txt 2>&1
      tail -8 gpurun_out/r05_pytest_gpu_v1.txt
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
      timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/r05_bench_v1_f16_n1.json 2> gpurun_out/r05_bench_v1.err
      tail -c 600 gpurun_out/r05_bench_v1.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05_bench_v1_f16_n1.json').read().strip().split('\n')[-1])
print({k: d[k] for k in ('value', 'ms_per_step')}, {k: d['roofline'].get(k) for k in ('achieved', 'frac', 'power_ceiling_frac')}, d['roofline'].get('mfma_skeleton_ceiling_on_random_operands', {}).get('TFLOP/s'))
print(d.get('unet_kernel_ms') or d.get('breakdown'))
PY
      ;;
  8)  # LayerNorm row statistics in the producers' epilogues: new GPU tests, then the forward three ways on ONE box
      timeout 900 python -m pytest tests/test_kernels.py tests/test_unet.py tests/test_parity_cfg4.py -m gpu -x -q -k "row_statistics or producer_epilogue or statistics_from or large_group_means or cfg4 or blocks_match or small_unet" 2>&1 | tail -6 > gpurun_out/r05_pytest_gpu_new3.txt
      cat gpurun_out/r05_pytest_gpu_new3.txt
      ( timeout 400 python tools/profile_forward.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r05_forward_detail_f16_gn_ln_epi.txt 2>&1
      ( STAR_NO_LNEPI=1 timeout 400 python tools/profile_forward.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r05_forward_detail_f16_gn_epi_only.txt 2>&1
      ( STAR_NO_LNEPI=1 STAR_NO_GNEPI=1 timeout 400 python tools/profile_forward.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r05_forward_detail_f16_no_epi_stats.txt 2>&1
      for f in gn_ln_epi gn_epi_only no_epi_stats; do echo "== $f"; head -2 gpurun_out/r05_forward_detail_f16_$f.txt; grep -E "group_norm|layer_norm" gpurun_out/r05_forward_detail_f16_$f.txt | head -2; done ;;
  9)  # closing evidence (one box): the whole GPU suite with its parity lines, smoke, the bench line (per-family table from the warm-up
      # clip, power sampled live, operand sweep + live skeleton ceiling, CPU baseline), the same command under rocprofv3 --kernel-trace
      # --stats, then the other configurations on the final tree
      O=gpurun_out/r05z; mkdir -p $O
      ( time timeout 1500 python -m pytest tests -m gpu -q -x -s 2>&1 | grep -v amdgpu.ids | grep -E "dB|rel rms|relative rms|passed|failed|error|skipped" ) > $O/pytest_gpu_final.txt 2>&1
      tail -4 $O/pytest_gpu_final.txt
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
      timeout 900 python bench.py --steps 1 --warmup 1 > $O/bench_final_f16_n1.json 2> $O/bench.err
      head -c 300 $O/bench_final_f16_n1.json; echo
      R=${GRAFT_REPO_ROOT:-/root/repo}; ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_bench &&
        timeout 700 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o b --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-operand-sweep > $R/$O/bench_final_rocprof_f16_n1.json 2> $R/$O/rocprof.err;
        f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1); cp "$f" $R/$O/bench_final_kernel_stats.csv 2>/dev/null )
      head -6 $O/bench_final_kernel_stats.csv
      timeout 400 python bench.py --dtype bf16 --steps 1 --warmup 0 --no-cpu-baseline --no-operand-sweep > $O/bench_bf16_n1.json 2>> $O/bench.err; head -c 200 $O/bench_bf16_n1.json; echo
      timeout 400 python bench.py --config cfg3 --steps 1 --warmup 0 --no-cpu-baseline --no-operand-sweep > $O/bench_cfg3_fast_f16_n1.json 2>> $O/bench.err; head -c 200 $O/bench_cfg3_fast_f16_n1.json; echo
      timeout 1100 python bench.py --config cfg4 --steps 1 --warmup 0 --no-cpu-baseline --no-operand-sweep > $O/bench_cfg4_50eval_f16_n1.json 2>> $O/bench.err; head -c 200 $O/bench_cfg4_50eval_f16_n1.json; echo ;;
  10) # PMC passes (one counter group per run, kernel trace only) on the torch-free harness: attention traffic + MFMA / VALU / LDS / wait
      # counters of the dominant kernels at their cfg2 shapes
      R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp && export TMPDIR=/tmp
      for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES" \
                 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
        tag=$(echo "$pmc" | cut -d' ' -f1)
        timeout 150 rocprofv3 --kernel-trace --pmc $pmc -d $R/gpurun_out/r05_pmc_$tag -- \
          $R/tools/cbench/cbench $R/tools/bench/libstar_hip_bench.so f16 $R/tools/cbench/pmc_shapes.txt 1 > $R/gpurun_out/r05_pmc_$tag.log 2>&1
      done
      python $R/tools/pmc_db_summary.py $R/gpurun_out/r05_pmc_FETCH_SIZE $R/gpurun_out/r05_pmc_WRITE_SIZE $R/gpurun_out/r05_pmc_SQ_INSTS_VALU $R/gpurun_out/r05_pmc_SQ_WAIT_INST_ANY $R/gpurun_out/r05_pmc_SQ_LDS_BANK_CONFLICT $R/gpurun_out/r05_pmc_GRBM_GUI_ACTIVE > $R/gpurun_out/r05_pmc_kernels.txt 2>&1
      head -60 $R/gpurun_out/r05_pmc_kernels.txt
      # one bench clip each way on this box: statistics in the epilogues off (the round-4 forward) / on
      cd $R
      for mode in off on; do
        if [ $mode = off ]; then export STAR_NO_GNEPI=1 STAR_NO_LNEPI=1; else unset STAR_NO_GNEPI STAR_NO_LNEPI; fi
        timeout 400 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-operand-sweep 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('[epilogue statistics $mode] value', round(d['value'],4), 'frames/s, ms_per_step', round(d['ms_per_step'],1), ', L0 attention', round(d['roofline']['achieved'],1), 'TF/s')"
      done > gpurun_out/r05_same_box_bench_stats_off_on.txt 2>&1
      cat gpurun_out/r05_same_box_bench_stats_off_on.txt ;;
  11) # final tree (channel-major 3x3 convs): the whole GPU suite, smoke, the forward table, the bench line and the same command under rocprofv3
      O=gpurun_out/r05f; mkdir -p $O
      ( time timeout 1500 python -m pytest tests -m gpu -q -x -s 2>&1 | grep -v amdgpu.ids | grep -E "dB|rel rms|relative rms|passed|failed|error|skipped" ) > $O/pytest_gpu_final.txt 2>&1
      tail -4 $O/pytest_gpu_final.txt
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
      ( timeout 400 python tools/profile_forward.py 2>&1 | grep -v amdgpu.ids ) > $O/forward_detail_f16_final.txt 2>&1; head -2 $O/forward_detail_f16_final.txt
      timeout 900 python bench.py --steps 1 --warmup 1 > $O/bench_final_f16_n1.json 2> $O/bench.err
      head -c 300 $O/bench_final_f16_n1.json; echo
      R=${GRAFT_REPO_ROOT:-/root/repo}; ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_bench &&
        timeout 700 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o b --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-operand-sweep > $R/$O/bench_final_rocprof_f16_n1.json 2> $R/$O/rocprof.err;
        f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1); cp "$f" $R/$O/bench_final_kernel_stats.csv 2>/dev/null )
      head -4 $O/bench_final_kernel_stats.csv; head -c 200 $O/bench_final_rocprof_f16_n1.json; echo ;;
  12) # the cfg3 chunk-loop golden on hardware; the level-0 3x3 conv's HBM-side traffic after the K-order change; the tile sweep on the final tree
      ( timeout 600 python -m pytest tests/test_parity_cfg3.py -m gpu -q -x -s 2>&1 | grep -v amdgpu.ids | grep -E "dB|passed|failed|error|skipped|Error" ) > gpurun_out/r05_pytest_parity_cfg3.txt 2>&1
      cat gpurun_out/r05_pytest_parity_cfg3.txt
      R=${GRAFT_REPO_ROOT:-/root/repo}; printf 'conv 32 122 216 320 320 2\nconv 32 62 108 640 640 2\n' > /tmp/convs.txt
      ( cd /tmp && export TMPDIR=/tmp && for pmc in FETCH_SIZE WRITE_SIZE; do
          timeout 100 rocprofv3 --kernel-trace --pmc $pmc -d $R/gpurun_out/r05_pmc_conv_$pmc -- $R/tools/cbench/cbench $R/tools/bench/libstar_hip_bench.so f16 /tmp/convs.txt 1 > $R/gpurun_out/r05_pmc_conv_$pmc.log 2>&1; done )
      python tools/pmc_db_summary.py gpurun_out/r05_pmc_conv_FETCH_SIZE gpurun_out/r05_pmc_conv_WRITE_SIZE > gpurun_out/r05_pmc_conv_after.txt 2>&1; cat gpurun_out/r05_pmc_conv_after.txt
      timeout 90 ./tools/cbench/cbench tools/bench/libstar_hip_bench.so f16 tools/cbench/cfg2_tile_sweep.txt 6 > gpurun_out/r05_cbench_tile_sweep_final.txt 2>&1
      grep -v differ gpurun_out/r05_cbench_tile_sweep_final.txt | grep -E "conv|tile=0 " | cut -c1-110 ;;
esac

#!/bin/bash
# Round-6 GPU calls: each block is ONE gpurun call; run from the repo root on the GPU box (gpurun -- 'bash tools/r06_calls.sh 1').
# Outputs under gpurun_out/ (what is quoted is copied into profiles/, index in profiles/README.md).
set -u
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}
BL=$R/tools/bench/libstar_hip_bench.so
case "${1:-1}" in
  19) # tile 17 alone, alternating whole clips after a discarded one (calls 15 / 16: -0.24 % / -0.2 % with it OFF)
      for tag in warm on1 off1 on2 off2 on3 off3; do
        ( case $tag in off*) export STAR_NO_SCHED=1 ;; esac
          timeout 400 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-operand-sweep 2>> gpurun_out/r06_bench_t17_clip.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['steps_detail'][0]; print('%-6s %8.1f ms  %6.1f W  %6.1f MHz  L0 attention %.0f TF/s' % ('$tag', s['ms'], s.get('socket_W',0), s.get('sclk_MHz',0), d['roofline']['achieved']))" )
      done | tee gpurun_out/r06_same_box_tile17_clips.txt ;;
  18) # the composed FF GEMM alone, alternating whole clips after a discarded one (calls 15 / 16 disagree)
      for tag in warm on1 off1 on2 off2 on3 off3; do
        ( case $tag in off*) export STAR_NO_FFPO=1 ;; esac
          timeout 400 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-operand-sweep 2>> gpurun_out/r06_bench_ffpo_clip.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['steps_detail'][0]; print('%-6s %8.1f ms  %6.1f W  %6.1f MHz  L0 attention %.0f TF/s' % ('$tag', s['ms'], s.get('socket_W',0), s.get('sclk_MHz',0), d['roofline']['achieved']))" )
      done | tee gpurun_out/r06_same_box_ffpo_clips.txt ;;
  17) # call 14 again on another box, alternating, after one discarded clip (call 16's box drifted 1.4 % between its first and last baseline)
      for tag in warm on1 off1 on2 off2 on3; do
        ( case $tag in off*) export STAR_NO_FFPO=1 STAR_NO_SCHED320=1 STAR_NO_GN_INPLACE=1 ;; esac
          timeout 400 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-operand-sweep 2>> gpurun_out/r06_bench_same_box2.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['steps_detail'][0]; print('%-6s %8.1f ms  %6.1f W  %6.1f MHz  L0 attention %.0f TF/s' % ('$tag', s['ms'], s.get('socket_W',0), s.get('sclk_MHz',0), d['roofline']['achieved']))" )
      done | tee gpurun_out/r06_same_box_round_steps2.txt ;;
  16) # call 15's two surprises (the composed FF GEMM and tile 17 LOSE as whole clips) on a second box, alone and together
      for sw in NONE STAR_NO_FFPO STAR_NO_SCHED BOTH NONE2; do
        ( case $sw in NONE*) ;; BOTH) export STAR_NO_FFPO=1 STAR_NO_SCHED=1 ;; *) export $sw=1 ;; esac
          timeout 400 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-operand-sweep 2>> gpurun_out/r06_bench_switches2.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['steps_detail'][0]; print('%-18s %8.1f ms  %6.1f W  %6.1f MHz  L0 attention %.0f TF/s' % ('$sw', s['ms'], s.get('socket_W',0), s.get('sclk_MHz',0), d['roofline']['achieved']))" )
      done | tee gpurun_out/r06_same_box_switches2.txt ;;
  15) # every product A/B switch as a WHOLE CLIP on one box (the sustained regime decides, call 14): one clip per setting, baseline first and last
      for sw in NONE STAR_NO_PERSIST STAR_NO_SCHED STAR_NO_ASTAT STAR_NO_TQ STAR_NO_SCHED320 STAR_NO_FFPO NONE2; do
        ( case $sw in NONE*) ;; *) export $sw=1 ;; esac
          timeout 400 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-operand-sweep 2>> gpurun_out/r06_bench_switches.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['steps_detail'][0]; print('%-18s %8.1f ms  %6.1f W  %6.1f MHz  L0 attention %.0f TF/s' % ('$sw', s['ms'], s.get('socket_W',0), s.get('sclk_MHz',0), d['roofline']['achieved']))" )
      done | tee gpurun_out/r06_same_box_switches.txt ;;
  14) # the round's switchable steps on / off on ONE box, whole clips (cfg2, the metric): composed FF GEMM, tile 19, GroupNorm in place
      # (the VAE fixes and tile 18's non-temporal stores have no product switch and are in both)
      for tag in on off on2; do
        if [ $tag = off ]; then export STAR_NO_FFPO=1 STAR_NO_SCHED320=1 STAR_NO_GN_INPLACE=1; else unset STAR_NO_FFPO STAR_NO_SCHED320 STAR_NO_GN_INPLACE; fi
        timeout 400 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-operand-sweep > gpurun_out/r06_bench_same_box_$tag.json 2>> gpurun_out/r06_bench_same_box.err
        python -c "
import json; d=json.loads(open('gpurun_out/r06_bench_same_box_$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['value'],4), round(d['ms_per_step']), d['steps_detail'])"
      done | tee gpurun_out/r06_same_box_round_steps.txt ;;
  13) # the one-read softmax of the VAE's d = 512 attention: unit tests, the VAE tests, the per-shape table and the wall time again
      timeout 900 python -m pytest tests/test_kernels.py tests/test_vae.py tests/test_fullsize.py -m gpu -x -q -k "softmax_rows or vae" 2>&1 | tail -3 | tee gpurun_out/r06_pytest_softmax_vec.txt
      ( timeout 400 python tools/profile_vae.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06_vae_detail_softmax_vec.txt; grep -E "^==|algorithmic|misc" gpurun_out/r06_vae_detail_softmax_vec.txt
      ( timeout 300 python tools/vae_time.py 6 2>&1 | grep -v amdgpu.ids ) | tee gpurun_out/r06_vae_time_softmax_vec.txt ;;
  12) # the counter table again for the shapes whose kernel changed after call 1 (tile 19, composed FF GEMM): review r05 item 6
      timeout 200 ./tools/cbench/cbench $BL f16 tools/cbench/r06_traffic_shapes_final.txt 6 > gpurun_out/r06_traffic_timing_final.txt 2>&1
      cd /tmp && export TMPDIR=/tmp
      P=$R/gpurun_out/r06_pmc_final; rm -rf $P; mkdir -p $P
      i=0
      grep -v '^#' $R/tools/cbench/r06_traffic_shapes_final.txt | grep -v '^\s*$' | while read -r line; do
        echo "$line" > /tmp/one.txt
        timeout 60 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $P/s${i}_FETCH -- $R/tools/cbench/cbench $BL f16 /tmp/one.txt 1 > $P/s${i}_FETCH.log 2>&1
        timeout 60 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $P/s${i}_WRITE -- $R/tools/cbench/cbench $BL f16 /tmp/one.txt 1 > $P/s${i}_WRITE.log 2>&1
        timeout 60 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $P/s${i}_SQ -- $R/tools/cbench/cbench $BL f16 /tmp/one.txt 1 > $P/s${i}_SQ.log 2>&1
        i=$((i+1))
      done
      cd $R
      python tools/pmc_traffic_table.py tools/cbench/r06_traffic_shapes_final.txt gpurun_out/r06_pmc_final gpurun_out/r06_traffic_timing_final.txt > gpurun_out/r06_traffic_table_final.txt 2>&1
      cat gpurun_out/r06_traffic_table_final.txt
      find gpurun_out/r06_pmc_final -name "*.db" -size +2M -delete; du -sh gpurun_out/r06_pmc_final
      # and the RCCL world-1 test of the sharders
      timeout 600 python -m pytest tests/test_parallel.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r06_pytest_rccl_world1.txt ;;
  11) # the probabilities packed round-toward-zero (attn5.h RTZ, bench variant 35): (a) what the two converts cost beside an MFMA (probe),
      # (b) the kernel A/B with socket power / clock beside it at the three self-attention lengths of cfg2, (c) accuracy of both against fp64
      timeout 60 ./tools/probe/mfma_valu_overlap 20000 2 > gpurun_out/r06_probe_cvt_pkrtz.txt 2>&1; cat gpurun_out/r06_probe_cvt_pkrtz.txt
      CBENCH_POWER=1 CBENCH_BATCH_MS=700 timeout 200 ./tools/cbench/cbench $BL f16 - 4 > gpurun_out/r06_attn_rtz_ab.txt 2>&1 <<'SPEC'
attn 32 5 26352 26352 9,35,9,35
attn 32 10 6696 6696 9,35,9,35
attn 32 20 1728 1728 9,35
SPEC
      cat gpurun_out/r06_attn_rtz_ab.txt
      ( timeout 300 python tools/attn_rtz_accuracy.py 9 35 2>&1 | grep -v amdgpu.ids ) | tee gpurun_out/r06_attn_rtz_accuracy.txt ;;
  1)  # (a) attention: one 512-thread workgroup per CU (variant 34) against the product kernel, with socket power / clock beside it;
      # (b) the traffic table: per (family, shape) >= 3 ms of a cfg2 forward, three single-group PMC passes on the torch-free harness
      CBENCH_POWER=1 CBENCH_BATCH_MS=700 timeout 120 ./tools/cbench/cbench $BL f16 - 4 > gpurun_out/r06_attn8_ab.txt 2>&1 <<'SPEC'
attn 32 5 26352 26352 9,34,9,34
attn 32 10 6696 6696 9,34
attn 32 20 1728 1728 9,34
attn 32 5 26352 77 9,34
SPEC
      cat gpurun_out/r06_attn8_ab.txt
      timeout 200 ./tools/cbench/cbench $BL f16 tools/cbench/r06_traffic_shapes.txt 6 > gpurun_out/r06_traffic_timing.txt 2>&1
      cd /tmp && export TMPDIR=/tmp
      P=$R/gpurun_out/r06_pmc; rm -rf $P; mkdir -p $P
      i=0
      grep -v '^#' $R/tools/cbench/r06_traffic_shapes.txt | grep -v '^\s*$' | while read -r line; do
        echo "$line" > /tmp/one.txt
        timeout 60 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $P/s${i}_FETCH -- $R/tools/cbench/cbench $BL f16 /tmp/one.txt 1 > $P/s${i}_FETCH.log 2>&1
        timeout 60 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $P/s${i}_WRITE -- $R/tools/cbench/cbench $BL f16 /tmp/one.txt 1 > $P/s${i}_WRITE.log 2>&1
        timeout 60 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $P/s${i}_SQ -- $R/tools/cbench/cbench $BL f16 /tmp/one.txt 1 > $P/s${i}_SQ.log 2>&1
        i=$((i+1))
      done
      cd $R
      python tools/pmc_traffic_table.py tools/cbench/r06_traffic_shapes.txt gpurun_out/r06_pmc gpurun_out/r06_traffic_timing.txt > gpurun_out/r06_traffic_table.txt 2>&1
      cat gpurun_out/r06_traffic_table.txt
      find gpurun_out/r06_pmc -name "*.db" -size +2M -delete; du -sh gpurun_out/r06_pmc ;;
  2)  # (a) the attention A/B again with the socket's own hwmon beside it; (b) tile 18's walk / store policy and the two-workgroup GEGLU tile;
      # (c) what a B = 2 batched CFG pair could buy at levels 2-3; (d) the new unit tests on hardware
      CBENCH_POWER=1 CBENCH_BATCH_MS=800 timeout 120 ./tools/cbench/cbench $BL f16 - 3 > gpurun_out/r06_attn8_ab_power.txt 2>&1 <<'SPEC'
attn 32 5 26352 26352 9,34,9,34
SPEC
      cat gpurun_out/r06_attn8_ab_power.txt
      timeout 300 ./tools/cbench/cbench $BL f16 tools/cbench/r06_persist_walk.txt 6 > gpurun_out/r06_cbench_persist_walk.txt 2>&1
      grep -v "bit-identical" gpurun_out/r06_cbench_persist_walk.txt | cut -c1-120
      timeout 300 ./tools/cbench/cbench $BL f16 tools/cbench/r06_cfg_batch.txt 6 > gpurun_out/r06_cbench_cfg_batch.txt 2>&1
      cut -c1-100 gpurun_out/r06_cbench_cfg_batch.txt
      timeout 900 python -m pytest tests/test_kernels.py -m gpu -x -q -k "tail_split or frame_interleaved or flash_attention_self or producer_epilogue" 2>&1 | tail -5 > gpurun_out/r06_pytest_new1.txt
      cat gpurun_out/r06_pytest_new1.txt
      # (e) the VAE with / without GroupNorm statistics in the producers' epilogues, same box
      ( timeout 300 python tools/vae_time.py 6; STAR_NO_GNEPI=1 timeout 300 python tools/vae_time.py 6 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_vae_gnepi_ab.txt
      cat gpurun_out/r06_vae_gnepi_ab.txt
      timeout 600 python -m pytest tests/test_vae.py tests/test_parity_cfg4.py -m gpu -x -q -s 2>&1 | grep -E "dB|passed|failed|rror" | tail -8 > gpurun_out/r06_pytest_new2.txt
      cat gpurun_out/r06_pytest_new2.txt ;;
  8)  # the level-1 q | k | v on the persistent tile (ragged last column tile) IN SITU: forward table with / without, same box (the switch
      # STAR_NO_PERSIST_RAGGED lived only for this measurement: -8 % in situ against +3 % in cbench -> dropped)
      ( timeout 400 python tools/profile_forward.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06_forward_detail_f16_ragged18.txt 2>&1
      ( STAR_NO_PERSIST_RAGGED=1 timeout 400 python tools/profile_forward.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06_forward_detail_f16_noragged18.txt 2>&1
      for f in ragged18 noragged18; do echo "== $f"; head -2 gpurun_out/r06_forward_detail_f16_$f.txt; grep -E "214272 +1920 +640" gpurun_out/r06_forward_detail_f16_$f.txt; done ;;
  11) # tile 19 (scheduled 256 x 320) auto-selected against STAR_NO_SCHED320=1, whole forwards on one box, twice each + the kernel tests
      timeout 300 python -m pytest tests/test_kernels.py -m gpu -x -q -k "scheduled_tile or producer_epilogue or tail_split or conv" 2>&1 | tail -2
      for i in 1 2; do
        ( timeout 400 python tools/profile_forward.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06_forward_detail_f16_tile19_$i.txt 2>&1
        ( STAR_NO_SCHED320=1 timeout 400 python tools/profile_forward.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06_forward_detail_f16_notile19_$i.txt 2>&1
      done
      for f in tile19_1 notile19_1 tile19_2 notile19_2; do echo "== $f"; head -2 gpurun_out/r06_forward_detail_f16_$f.txt; done
      python - <<'PY'
import re
def load(f):
    d={}
    for l in open(f):
        t=l.split()
        if len(t)==9 and t[0] in ('gemm','conv3x3','tconv'): d[tuple(t[:5])]=(int(t[5]),float(t[6]))
    return d
a=load('gpurun_out/r06_forward_detail_f16_tile19_2.txt'); b=load('gpurun_out/r06_forward_detail_f16_notile19_2.txt')
tot=0
for k in sorted(a, key=lambda k:-abs(a[k][1]-b.get(k,(0,a[k][1]))[1])):
    if k in b and abs(a[k][1]-b[k][1])>0.03:
        print(k, 'n',a[k][0], 'tile19 %.2f ms  tile2 %.2f ms  (%+.1f %%)'%(a[k][1],b[k][1],(a[k][1]/b[k][1]-1)*100)); tot+=a[k][1]-b[k][1]
print('sum of changes: %.2f ms per forward'%tot)
PY
      ;;
  10) # GroupNorm apply in place + compile-time SiLU: forward table both ways on one box, twice; the VAE both ways (its norms are always in
      # place now: only the statistics switch remains)
      for i in 1 2; do
        ( timeout 400 python tools/profile_forward.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06_forward_detail_f16_gninplace_$i.txt 2>&1
        ( STAR_NO_GN_INPLACE=1 timeout 400 python tools/profile_forward.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06_forward_detail_f16_gnoutofplace_$i.txt 2>&1
      done
      for f in gninplace_1 gnoutofplace_1 gninplace_2 gnoutofplace_2; do echo "== $f"; head -2 gpurun_out/r06_forward_detail_f16_$f.txt; grep -E "group_norm" gpurun_out/r06_forward_detail_f16_$f.txt | cut -c1-100; done
      ( timeout 300 python tools/vae_time.py 6 2>&1 | grep -v amdgpu.ids ) | tee gpurun_out/r06_vae_inplace.txt
      timeout 900 python -m pytest tests/test_unet.py tests/test_vae.py -m gpu -x -q 2>&1 | tail -3 ;;
  9)  # tile 18's non-temporal stores IN SITU (the cbench gain must survive inside a forward): the bench build, product stores against
      # STAR_PERSIST_PLAIN=1 (the round-5 plain stores), same box, twice each
      for i in 1 2; do
        ( timeout 400 python tools/profile_forward.py --lib tools/bench/libstar_hip_bench.so 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06_forward_detail_f16_nt_$i.txt 2>&1
        ( STAR_PERSIST_PLAIN=1 timeout 400 python tools/profile_forward.py --lib tools/bench/libstar_hip_bench.so 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06_forward_detail_f16_plain_$i.txt 2>&1
      done
      for f in nt_1 plain_1 nt_2 plain_2; do echo "== $f"; head -2 gpurun_out/r06_forward_detail_f16_$f.txt; grep -E " (5120 +640|10240 +1280|3840 +1280|4096 +512) +3[37] " gpurun_out/r06_forward_detail_f16_$f.txt | cut -c1-100; done ;;
  7)  # closing evidence (one box): the whole GPU suite with its parity lines, smoke, the forward table, the bench line (per-family table from
      # the warm-up clip -> roofline.families, power sampled live, operand sweep + live skeleton ceiling, CPU baseline), the same command under
      # rocprofv3 --kernel-trace --stats, then the other configurations
      O=gpurun_out/r06f; mkdir -p $O
      ( time timeout 1800 python -m pytest tests -m gpu -q -x -s 2>&1 | grep -v amdgpu.ids | grep -E "dB|rel rms|relative rms|passed|failed|error|skipped|real" ) > $O/pytest_gpu_final.txt 2>&1
      tail -5 $O/pytest_gpu_final.txt
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
      ( timeout 400 python tools/profile_forward.py 2>&1 | grep -v amdgpu.ids ) > $O/forward_detail_f16_final.txt 2>&1; head -2 $O/forward_detail_f16_final.txt
      timeout 1200 python bench.py --steps 1 --warmup 1 > $O/bench_final_f16_n1.json 2> $O/bench.err
      head -c 300 $O/bench_final_f16_n1.json; echo
      ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_bench &&
        timeout 700 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o b --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-operand-sweep > $R/$O/bench_final_rocprof_f16_n1.json 2> $R/$O/rocprof.err;
        f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1); cp "$f" $R/$O/bench_final_kernel_stats.csv 2>/dev/null )
      head -4 $O/bench_final_kernel_stats.csv; head -c 200 $O/bench_final_rocprof_f16_n1.json; echo
      timeout 400 python bench.py --config cfg3 --steps 1 --warmup 0 --no-cpu-baseline --no-operand-sweep > $O/bench_cfg3_fast_f16_n1.json 2>> $O/bench.err; head -c 200 $O/bench_cfg3_fast_f16_n1.json; echo
      timeout 400 python bench.py --dtype bf16 --steps 1 --warmup 0 --no-cpu-baseline --no-operand-sweep > $O/bench_bf16_n1.json 2>> $O/bench.err; head -c 200 $O/bench_bf16_n1.json; echo
      timeout 1100 python bench.py --config cfg4 --steps 1 --warmup 0 --no-cpu-baseline --no-operand-sweep > $O/bench_cfg4_50eval_f16_n1.json 2>> $O/bench.err; head -c 200 $O/bench_cfg4_50eval_f16_n1.json; echo ;;
  5)  # two forwards on two streams at once against one after the other (upper bound of a two-stream CFG pair)
      ( timeout 600 python tools/concurrent_forward.py 3 2>&1 | grep -v amdgpu.ids | tail -3 ) > gpurun_out/r06_concurrent_forward.txt 2>&1
      cat gpurun_out/r06_concurrent_forward.txt ;;
  4)  # the composed FF-out / proj_out GEMM: UNet tests, the forward table both ways on ONE box, the cfg2-geometry parity lines
      ( timeout 1200 python -m pytest tests/test_unet.py tests/test_kernels.py -m gpu -x -q 2>&1 | tail -4 ) > gpurun_out/r06_pytest_unet_ffpo.txt 2>&1
      cat gpurun_out/r06_pytest_unet_ffpo.txt
      ( timeout 400 python tools/profile_forward.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06_forward_detail_f16_ffpo.txt 2>&1
      ( STAR_NO_FFPO=1 timeout 400 python tools/profile_forward.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06_forward_detail_f16_noffpo.txt 2>&1
      for f in ffpo noffpo; do echo "== $f"; head -12 gpurun_out/r06_forward_detail_f16_$f.txt | cut -c1-110; done
      ( timeout 900 python -m pytest tests/test_parity_cfg2.py tests/test_parity_cfg4.py -m gpu -q -x -s 2>&1 | grep -v amdgpu.ids | grep -E "dB|passed|failed|rror" ) > gpurun_out/r06_pytest_parity_ffpo.txt 2>&1
      cat gpurun_out/r06_pytest_parity_ffpo.txt ;;
  3)  # temporal conv: K order x frame-interleaved walk (the STAR_TCONV_KORDER switch lived only for this measurement: it was removed with
      # the experiment -- +2.2 % at level 0, -2 % at levels 2-3 -- see gemm.h stage_post and DESIGN.md section 8) (+ the FETCH_SIZE of the level-0 launch both ways)
      for o in 0 1; do echo "== STAR_TCONV_KORDER=$o"; STAR_TCONV_KORDER=$o timeout 120 ./tools/cbench/cbench $BL f16 tools/cbench/r06_tconv_korder.txt 8 | grep -v differ | cut -c1-120; done > gpurun_out/r06_cbench_tconv_korder.txt 2>&1
      cat gpurun_out/r06_cbench_tconv_korder.txt
      cd /tmp && export TMPDIR=/tmp
      printf 'tconv 32 26352 320 0\ntconv 32 6696 640 0\n' > /tmp/tc.txt
      for o in 0 1; do
        STAR_TCONV_KORDER=$o timeout 60 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/r06_pmc_tconv_k$o -- $R/tools/cbench/cbench $BL f16 /tmp/tc.txt 1 > /dev/null 2>&1
        echo "== STAR_TCONV_KORDER=$o"; python $R/tools/pmc_db_summary.py $R/gpurun_out/r06_pmc_tconv_k$o | grep -v "^==" | cut -c60-200
      done > $R/gpurun_out/r06_pmc_tconv_korder.txt 2>&1
      cat $R/gpurun_out/r06_pmc_tconv_korder.txt; rm -rf $R/gpurun_out/r06_pmc_tconv_k0 $R/gpurun_out/r06_pmc_tconv_k1 ;;
esac

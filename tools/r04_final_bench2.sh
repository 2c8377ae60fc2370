#!/bin/bash
# round-4 closing evidence, part 2 (one box): round-3 tree against round-4 tree (forward profile + one bench clip each, interleaved), the
# closing bench line (--warmup 1: per-family table from the warm-up clip, power sampled live, operand sweep, CPU baseline), the same
# command under rocprofv3 --kernel-trace --stats, the attention kernel's HBM traffic from single-counter PMC passes
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04z; mkdir -p $OUT
R03=$GRAFT_REPO_ROOT/build/r03tree
AB=$OUT/same_box_r03_vs_r04.txt
echo "# round-3 tree (commit 2a1e7e6, built from git archive into build/r03tree) against the round-4 tree on ONE MI355X box, interleaved" > $AB
for rep in 1; do
  for tree in r03 r04; do
    d=$GRAFT_REPO_ROOT; [ $tree = r03 ] && d=$R03
    ( cd $d && timeout 300 python tools/profile_forward.py 2>&1 | grep -v amdgpu.ids ) > $OUT/forward_${tree}_$rep.txt
    head -2 $OUT/forward_${tree}_$rep.txt | sed "s/^/[$tree forward #$rep] /" >> $AB
  done
done
for tree in r03 r04; do
  d=$GRAFT_REPO_ROOT; [ $tree = r03 ] && d=$R03
  extra=""; [ $tree = r04 ] && extra="--no-operand-sweep"
  ( cd $d && timeout 400 python bench.py --steps 1 --warmup 0 --no-cpu-baseline $extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('[$tree bench clip] value', round(d['value'],4), 'frames/s, ms_per_step', round(d['ms_per_step'],1), ', L0 attention', round(d['roofline']['achieved'],1), 'TF/s')" ) >> $AB
done
cat $AB
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --steps 1 --warmup 1 > $OUT/bench_final_f16_n1.json 2> $OUT/bench.err
head -c 500 $OUT/bench_final_f16_n1.json; echo
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bench
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o b --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-operand-sweep > $OUT/bench_final_rocprof_f16_n1.json 2> $OUT/rocprof.err
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/bench_final_kernel_stats.csv 2>/dev/null
head -8 $OUT/bench_final_kernel_stats.csv

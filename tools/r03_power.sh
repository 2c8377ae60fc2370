#!/bin/bash
# average socket power and shader clock (sysfs hwmon, sampled twice a second) while one attention variant runs in a loop
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
ls /sys/class/drm/ > $OUT/sysfs.txt 2>&1
for h in /sys/class/drm/card*/device/hwmon/hwmon*; do echo $h >> $OUT/sysfs.txt; ls $h >> $OUT/sysfs.txt; done
for spec in $2; do
  v=${spec%%:*}; fill=${spec##*:}
  python tools/attn_loop.py $v 5 $fill > /tmp/loop.txt 2>&1 &
  pid=$!
  sleep 2.5
  for i in 1 2 3 4; do
    for h in /sys/class/drm/card*/device/hwmon/hwmon*; do
      echo -n "power_uW $(cat $h/power1_average 2>/dev/null || cat $h/power1_input 2>/dev/null) sclk_Hz $(cat $h/freq1_input 2>/dev/null) ; "
    done; echo; sleep 0.5
  done > /tmp/smi.txt
  wait $pid
  echo "== $spec: $(cat /tmp/loop.txt | tail -1)" >> $OUT/power.txt
  cat /tmp/smi.txt >> $OUT/power.txt
done
cat $OUT/power.txt

#!/bin/bash
# Socket power, shader clock and junction temperature from the GPUs' hwmon nodes, sampled every ~45 ms while a
# short bench runs (profiles/r01_v8_hwmon_power_clock.txt, r01_v22_hwmon_power_clock.txt).  The box exposes the hwmon
# nodes of all GPUs of the host but runs the bench on one: every sample prints the node that draws the most power.
# usage: bash tools/sample_hwmon.sh > out.txt
NODES=$(ls -d /sys/class/drm/card*/device/hwmon/hwmon* 2>/dev/null)
echo "hwmon nodes: $(echo $NODES | wc -w)  power1_cap_uW=$(cat $(echo $NODES | cut -d' ' -f1)/power1_cap 2>/dev/null)"
python3 "$(dirname "$0")/../bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-async > /tmp/sample_hwmon_bench.json 2>/dev/null &
BP=$!
sleep 0.5
for i in $(seq 1 700); do
  best=0; line=""
  for H in $NODES; do
    p=$(cat $H/power1_input 2>/dev/null || echo 0)
    if [ "$p" -gt "$best" ]; then best=$p; line="power_uW=$p sclk_Hz=$(cat $H/freq1_input 2>/dev/null) tj_mC=$(cat $H/temp2_input 2>/dev/null) node=$(basename $H)"; fi
  done
  echo "$i $line"
  sleep 0.02
done
wait $BP

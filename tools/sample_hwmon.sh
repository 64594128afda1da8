#!/bin/bash
# Socket power, shader clock and junction temperature from the GPU's hwmon node, sampled every ~45 ms while a
# short bench runs (profiles/r01_v8_hwmon_power_clock.txt).  usage: bash tools/sample_hwmon.sh > out.txt
H=$(ls -d /sys/class/drm/card*/device/hwmon/hwmon* 2>/dev/null | head -1)
echo "hwmon: $H  power1_cap_uW=$(cat $H/power1_cap 2>/dev/null)"
python3 "$(dirname "$0")/../bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-async > /tmp/sample_hwmon_bench.json 2>/dev/null &
BP=$!
sleep 0.5
for i in $(seq 1 160); do
  echo "$i power_uW=$(cat $H/power1_input 2>/dev/null) sclk_Hz=$(cat $H/freq1_input 2>/dev/null) tj_mC=$(cat $H/temp2_input 2>/dev/null)"
  sleep 0.04
done
wait $BP
tail -c 300 /tmp/sample_hwmon_bench.json | head -c 0

#!/bin/bash
# usage (on the GPU box): tools/power_lab.sh [seconds]   -- every variant of tools/power_lab alone on the chip, rocm-smi sampled ~4 Hz meanwhile
cd "$(dirname "$0")/.."
secs=${1:-4}
for v in ${PL_VARIANTS:-0 1 5 2 3 4}; do
  tools/power_lab $v $secs > /tmp/pl_$v.txt &
  pid=$!
  sleep 1.2
  : > /tmp/pl_smi_$v.txt
  while kill -0 $pid 2>/dev/null; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk clock level" >> /tmp/pl_smi_$v.txt; done
  wait $pid
  w=$(grep "Package Power" /tmp/pl_smi_$v.txt | sed -E 's/.*: ([0-9.]+)$/\1/' | sort -n | awk '{a[NR]=$1} END{print a[int((NR+1)/2)]}')
  c=$(grep "sclk" /tmp/pl_smi_$v.txt | sed -E 's/.*\(([0-9]+)Mhz\).*/\1/' | sort -n | awk '{a[NR]=$1} END{print a[int((NR+1)/2)]}')
  n=$(grep -c "Package Power" /tmp/pl_smi_$v.txt)
  echo "$(cat /tmp/pl_$v.txt)   | ${w} W (median of $n)  sclk ${c} MHz"
done

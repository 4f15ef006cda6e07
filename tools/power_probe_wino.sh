#!/bin/bash
# Package power / shader clock while ONE conv launch runs back to back (tools/winograd_probe.py --hold): the direct
# product kernel against the Winograd probe at the same shape.  usage (inside gpurun): bash tools/power_probe_wino.sh [layer]
# -> gpurun_out/power_probe_wino.txt
mkdir -p gpurun_out
LAYER=${1:-l3}
OUT=gpurun_out/power_probe_wino.txt
{ echo "== idle"; rocm-smi -P -c 2>&1 | grep -i "power\|sclk" | head -4; } > $OUT
for K in direct wino; do
  echo "== $LAYER $K" >> $OUT
  python tools/winograd_probe.py --hold $LAYER,$K,7 > gpurun_out/power_probe_wino_$K.log 2>&1 &
  BP=$!
  sleep 3.5
  for i in $(seq 1 8); do
    if ! kill -0 $BP 2>/dev/null; then break; fi
    rocm-smi -P -c 2>&1 | grep -i "power\|sclk" | tr '\n' ' ' | sed 's/  */ /g' >> $OUT; echo >> $OUT
    sleep 0.3
  done
  wait $BP
  grep HOLD gpurun_out/power_probe_wino_$K.log >> $OUT
done
cat $OUT | cut -c1-220

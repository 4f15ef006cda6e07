#!/bin/bash
# Is the step power-limited?  Samples rocm-smi (average socket power, shader clock) every ~0.2 s while bench.py replays the
# captured step.  usage (inside gpurun): bash tools/power_probe.sh [bench args...]  -> gpurun_out/power_probe.txt
mkdir -p gpurun_out
OUT=gpurun_out/power_probe.txt
{
  echo "== caps"; rocm-smi --showmaxpower --showpower 2>&1 | grep -v "^=\|^$" | head -8
  echo "== idle"; rocm-smi -P -c 2>&1 | grep -i "power\|sclk\|mclk" | head -6
} > $OUT
python bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-roofline --no-reference-api "$@" > gpurun_out/power_probe_bench.json 2>/dev/null &
BP=$!
sleep 12
for i in $(seq 1 25); do
  if ! kill -0 $BP 2>/dev/null; then break; fi
  rocm-smi -P -c 2>&1 | grep -i "power\|sclk" | tr '\n' ' ' | sed 's/  */ /g' >> $OUT; echo >> $OUT
  sleep 0.2
done
wait $BP
tail -1 gpurun_out/power_probe_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench ms/step', d['ms_per_step'])" >> $OUT
cat $OUT | cut -c1-220

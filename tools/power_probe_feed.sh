#!/bin/bash
# Package power and shader clock while the staged MFMA-feed microbenchmark (tools/mfma_feed.py, libiic_probe.so) holds ONE
# stage for a few seconds: what the matrix pipe alone draws, and what feeding it from LDS / L2 adds.
#   make -C iic_amd/csrc probes; gpurun -- bash tools/power_probe_feed.sh      -> gpurun_out/power_probe_feed.txt
mkdir -p gpurun_out
OUT=gpurun_out/power_probe_feed.txt
: > $OUT
for st in 0 1 2 4; do
  python tools/mfma_feed.py --hold $st,32,3,6 > gpurun_out/feed_hold_$st.txt 2>&1 &
  BP=$!
  sleep 3.5
  S=""
  for i in 1 2 3 4 5 6; do
    if ! kill -0 $BP 2>/dev/null; then break; fi
    S="$S $(rocm-smi -P -c 2>&1 | grep -o 'Package Power (W): [0-9.]*\|sclk clock level: [0-9]: ([0-9]*Mhz)' | tr '\n' ' ')|"
    sleep 0.3
  done
  wait $BP
  echo "stage $st: $(tail -1 gpurun_out/feed_hold_$st.txt)" >> $OUT
  echo "   $S" | sed 's/Package Power (W): /P=/g; s/sclk clock level: [0-9]: //g' >> $OUT
done
cat $OUT

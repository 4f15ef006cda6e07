#!/bin/bash
# per-layer weight-gradient time at the north-star shapes, for each IIC_DEBUG value given
for d in "$@"; do
  echo "IIC_DEBUG=$d"
  IIC_DEBUG="$d" python tools/conv_perf.py --iters 10 2>&1 | awk -F"|" '{print $1 "|" $3}' | cut -c1-30,50-90 | tail -12
done

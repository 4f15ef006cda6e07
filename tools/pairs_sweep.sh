#!/bin/bash
# Compute-side scaling model on ONE GPU: the step a rank sees under `bench.py --strong` at
# N = 1 / 2 / 4 / 8 ranks (660 / 330 / 165 / 84 pairs per rank), default tiles vs 128-row conv tiles.
#   gpurun -- bash tools/pairs_sweep.sh     -> gpurun_out/pairs_sweep.txt
mkdir -p gpurun_out
for p in 660 330 165 84; do
  for dbg in "" "iic_debug_bd_ms=2"; do
    IIC_HIP_LIB=dbg IIC_DEBUG="$dbg" python bench.py --pairs $p --steps 20 --warmup 3 --no-cpu-baseline --no-reference-api --no-secondary 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d.get('roofline',{}); print('pairs %4d  IIC_DEBUG=%-20r ms/step %7.3f  pairs/s %8.0f  conv fwd+bwd-data: %6.1f TF/s frac %.3f  %5.2f ms/step in %d launches' % ($p, '$dbg', d['ms_per_step'], d['value'], r.get('achieved',0), r.get('frac',0), r.get('kernel_ms_per_step',0), r.get('launches_timed',0)))"
  done
done | tee gpurun_out/pairs_sweep.txt

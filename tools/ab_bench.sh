#!/bin/bash
# A/B of library debug switches on the default bench step (pair graphs): each argument is one
# IIC_DEBUG value ("" = defaults); prints ms_per_step for each, interleaved twice to expose box drift.
#   gpurun -- bash tools/ab_bench.sh "" "iic_debug_bd_ms=2"
mkdir -p gpurun_out
for rep in 1 2; do
  for dbg in "$@"; do
    IIC_HIP_LIB=dbg IIC_DEBUG="$dbg" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-reference-api --no-secondary $AB_BENCH_ARGS 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('IIC_DEBUG=%-40r ms/step %.3f  value %.0f' % ('$dbg', d['ms_per_step'], d['value']))"
  done
done | tee -a gpurun_out/ab_bench.txt

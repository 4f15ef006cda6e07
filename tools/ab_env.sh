#!/bin/bash
# A/B of environment switches on the default bench step (pair graphs): each argument is one set of assignments
# ("" = defaults, "IIC_HIP_LIB=dbg IIC_DEBUG=iic_debug_wgrad_target_wgs=128", ...); prints ms_per_step
# for each, interleaved twice to expose box drift.
#   gpurun -- bash tools/ab_env.sh "" "IIC_HIP_LIB=dbg IIC_DEBUG=iic_debug_bd_ms=4"
mkdir -p gpurun_out
for rep in 1 2; do
  for e in "$@"; do
    env $e timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-reference-api --no-secondary $AB_BENCH_ARGS 2>gpurun_out/ab_env.err \
      | python -c "import sys,json; L=sys.stdin.readlines(); d=json.loads(L[-1]) if L else {'ms_per_step': float('nan'), 'value': float('nan'), 'config': {}}; print('%-60s ms/step %.3f  value %.0f  loss %s' % ('$e' or '(defaults)', d['ms_per_step'], d['value'], d['config'].get('final_loss')))" \
      || tail -3 gpurun_out/ab_env.err
  done
done | tee -a gpurun_out/ab_env.txt

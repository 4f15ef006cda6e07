#!/bin/bash
# Segmentation configs, short: pairs/s, ms/step, joint ms, grad ms, contraction frac
for cfg in "potsdam3 --T 10" "coco3" "$@"; do
  python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-18s %.1f pairs/s  %.1f ms/step  joint %.2f ms  grad %.2f ms  frac %.3f' % ('$cfg', d['value'], d['ms_per_step'], r['joint_fwd_ms'], r['grad_bwd_ms'], r['frac']))"
done | tee -a gpurun_out/seg_bench.txt

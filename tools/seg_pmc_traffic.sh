#!/bin/bash
# HBM traffic of the segmentation-loss contraction kernels (two PMC passes over tools/seg_kernel_perf.py),
# written as gpurun_out/r02_seg_pmc_traffic_<cfg>.json in the format of tools/pmc_traffic.py
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for cfg in potsdam coco; do
  rm -rf /tmp/spf /tmp/spw
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/spf -o p -- python $R/tools/seg_kernel_perf.py $cfg 2 > /tmp/spf.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/spw -o p -- python $R/tools/seg_kernel_perf.py $cfg 2 > /tmp/spw.log 2>&1
  (cd $R && python tools/pmc_traffic.py /tmp/spf /tmp/spw gpurun_out/r02_seg_pmc_traffic_$cfg.json | grep seg_)
done

#!/bin/bash
# HBM traffic per kernel of one segmentation step (two PMC passes), $1 = bench config args
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ssf /tmp/ssw
B="python $R/bench.py --config $1 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-branch"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/ssf -o p -- $B > /tmp/ssf.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/ssw -o p -- $B > /tmp/ssw.log 2>&1
cd $R && python tools/pmc_traffic.py /tmp/ssf /tmp/ssw gpurun_out/r02_seg_step_traffic.json | head -24

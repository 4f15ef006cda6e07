#!/bin/bash
# Diagnosis of the latency-bound regimes (84 pairs per rank = the 8-rank strong-scaling shape; the 6c configs):
# kernel traces of the replayed two-stream step + sequential eager kernel statistics.
#   gpurun -- bash tools/small_batch_diag.sh   -> gpurun_out/{p84,mnist}_pair_timeline.txt, *_trace.csv.gz, p84seq_kernel_stats.txt
R=$GRAFT_REPO_ROOT
bash tools/pair_timeline.sh p84 --pairs 84 > /dev/null
gzip -c $(find /tmp/tl_p84 -name '*kernel_trace.csv' | head -1) > $R/gpurun_out/p84_trace.csv.gz
bash tools/pair_timeline.sh mnist --config mnist6c > /dev/null
gzip -c $(find /tmp/tl_mnist -name '*kernel_trace.csv' | head -1) > $R/gpurun_out/mnist_trace.csv.gz
bash tools/prof_step.sh p84seq --pairs 84 --no-branch --no-graph > /dev/null
python bench.py --pairs 84 --steps 30 --warmup 5 --no-cpu-baseline --no-reference-api --no-secondary 2>/dev/null | tail -1 > gpurun_out/p84_bench.json
head -c 600 gpurun_out/p84_bench.json; echo
head -30 gpurun_out/p84_pair_timeline.txt

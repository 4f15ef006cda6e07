# Round 6: repeat the one-rank RCCL runs (every launch mode, the 6c and segmentation configs) to catch intermittent failures
# of the process group's watchdog against graph capture (iic_amd/dist.py::_all_reduce_now).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
S="--pairs 66 --steps 3 --warmup 1 --no-roofline --no-cpu-baseline --no-secondary --no-reference-api"
n=0; bad=0
for rep in 1 2 3 4; do
  for mode in "" "IIC_DIST_STAGED=0" "IIC_DIST_GRAPH=0" "IIC_DIST_OVERLAP=1"; do
    n=$((n+1)); env IIC_DIST_FORCE=1 $mode timeout 200 python bench.py --gpus 1 $S > gpurun_out/flaky.out 2> gpurun_out/flaky.err || { bad=$((bad+1)); echo "FAILED: $mode"; grep -v "^frame\|^$" gpurun_out/flaky.err | head -12; }
  done
  for cfg in "mnist6c" "coco3"; do
    n=$((n+1)); IIC_DIST_FORCE=1 timeout 200 python bench.py --config $cfg --gpus 1 --steps 2 --warmup 1 --no-roofline > gpurun_out/flaky.out 2> gpurun_out/flaky.err || { bad=$((bad+1)); echo "FAILED: $cfg"; grep -v "^frame\|^$" gpurun_out/flaky.err | head -12; }
  done
done
echo "one-rank RCCL runs: $n, failed: $bad"

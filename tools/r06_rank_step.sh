# Round 6: what a rank's step costs at N > 1 as far as ONE MI355X can say -- the staged data-parallel step (17 graph segments,
# the raw-joint all-reduce between the loss segments, four bucket all-reduces on the third stream) through RCCL in a one-rank
# group (IIC_DIST_FORCE=1: identities, i.e. everything but the wire time), against the plain N = 1 step, at the weak-scaling batch
# (660 pairs per rank) and at the strong-scaling per-rank batches (330 / 165 / 84); interleaved twice.
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
A="--steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-reference-api --no-secondary"
for rep in 1 2; do
  for p in 660 330 165 84; do
    for mode in "plain" "IIC_DIST_FORCE=1" "IIC_DIST_FORCE=1 IIC_DIST_STAGED=0"; do
      e=$mode; [ "$mode" = "plain" ] && e="IIC_NOOP=1"
      env $e timeout 300 python bench.py --gpus 1 --pairs $p $A 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.readlines() if l.startswith('{')][-1]); print('pairs %4d  %-36s ms/step %7.3f  pairs/s %8.0f  host enqueue %.2f ms/step' % ($p, '$mode', d['ms_per_step'], d['value'], d['config']['host_enqueue_ms_per_step']))"
    done
  done
done | tee gpurun_out/r06_rank_step.txt

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for c in mnist6c cifar6c; do
  cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_$c
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --config $c --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/r06_${c}_prof.log 2>&1
  cd $GRAFT_REPO_ROOT
  python tools/prof_summary.py /tmp/prof_$c 13 "rocprofv3 --kernel-trace --stats -- python bench.py --config $c --no-roofline  (13 steps incl. warm-up)" > gpurun_out/r06_${c}_kernel_stats.txt
  head -34 gpurun_out/r06_${c}_kernel_stats.txt | cut -c1-70,100-150
done

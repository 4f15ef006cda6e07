# usage (inside gpurun): bash tools/dropin_timeline.sh  -> gpurun_out/r04_dropin_timeline.txt : what the chip does during a step
# issued through the reference's own call sequence with graph replay (tools/graphed_perf.py --modes graphed2)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_dropin
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_dropin -o p -- python $R/tools/graphed_perf.py --modes graphed2 --steps 8 > $R/gpurun_out/dropin_tl.log 2>&1
echo "prof rc=$?"; tail -2 $R/gpurun_out/dropin_tl.log | cut -c1-200
cd $R
python tools/pair_timeline.py /tmp/tl_dropin 3 > gpurun_out/r04_dropin_timeline.txt
cat gpurun_out/r04_dropin_timeline.txt
python tools/queue_spans.py /tmp/tl_dropin 14 | tee -a gpurun_out/r04_dropin_timeline.txt

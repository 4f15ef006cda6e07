# usage (inside gpurun): bash tools/prof_step.sh <tag> [bench.py args...]   -> gpurun_out/<tag>_kernel_stats.txt
tag=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-reference-api --no-secondary "$@" > $R/gpurun_out/${tag}_prof.log 2>&1
echo "prof rc=$?"; tail -1 $R/gpurun_out/${tag}_prof.log | cut -c1-300
cd $R
python tools/prof_summary.py /tmp/prof_$tag 7 "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-reference-api --no-secondary $*  (7 steps incl. warm-up)" > gpurun_out/${tag}_kernel_stats.txt
head -40 gpurun_out/${tag}_kernel_stats.txt | cut -c1-150

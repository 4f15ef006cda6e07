# usage (inside gpurun): bash tools/prof_cmd.sh <tag> <steps-for-normalisation> <python args...>  -> gpurun_out/<tag>_kernel_stats.txt
tag=$1; steps=$2; shift; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python "$@" > $R/gpurun_out/${tag}_prof.log 2>&1
echo "prof rc=$?"; grep -v Warn $R/gpurun_out/${tag}_prof.log | tail -3 | cut -c1-300
cd $R
python tools/prof_summary.py /tmp/prof_$tag $steps "rocprofv3 --kernel-trace --stats -- python $*" > gpurun_out/${tag}_kernel_stats.txt
head -30 gpurun_out/${tag}_kernel_stats.txt | cut -c1-160

# usage (inside gpurun): bash tools/pmc_cmd.sh <tag> <steps-divisor> <python script + args...>
#   -> gpurun_out/<tag>_kernel_stats.txt (rocprofv3 --kernel-trace --stats) and gpurun_out/<tag>_pmc_stalls.txt (one SQ pass)
tag=$1; shift; div=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag /tmp/pmc_$tag
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python "$@" > $R/gpurun_out/${tag}_prof.log 2>&1; echo "prof rc=$?"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- python "$@" > $R/gpurun_out/${tag}_pmc.log 2>&1; echo "pmc rc=$?"
cd $R
python tools/prof_summary.py /tmp/prof_$tag $div "rocprofv3 --kernel-trace --stats -- python $*" > gpurun_out/${tag}_kernel_stats.txt
python tools/pmc_stalls.py /tmp/pmc_$tag gpurun_out/${tag}_kernel_stats.txt gpurun_out/${tag}_pmc_stalls.txt | head -30

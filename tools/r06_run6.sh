cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/pmc_cmd.sh r06_bd_abl 1 $GRAFT_REPO_ROOT/tools/conv_perf.py --no-wgrad --iters 5 --frag-ablate 4,16,32 2>&1 | tail -20
grep -v "^FORCE\|^ABLATE" gpurun_out/r06_bd_abl_prof.log | tail -14 | cut -c1-330

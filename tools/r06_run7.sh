cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python tools/wgrad_swp_ab.py 2>&1 | tail -6 | tee gpurun_out/r06_wgrad_swp_ab.txt
rm -f gpurun_out/ab_env.txt
bash tools/ab_env.sh "IIC_HIP_LIB=dbg IIC_DEBUG=iic_debug_wgrad_swp=0" "IIC_HIP_LIB=dbg IIC_DEBUG=iic_debug_wgrad_swp=1"

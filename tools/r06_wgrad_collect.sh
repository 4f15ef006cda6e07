# One gpurun call: the weight-gradient measurements of LAB.md R6.8 / R6.9 -> gpurun_out/r06_wgrad_*.txt, r06_bd_pitch144_ablation.txt
# (build the instrumented library with the bd timing ablations first: make -C iic_amd/csrc dbg ABL=1)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
{ echo "# tools/wgrad_pl_ab.py: K-loop forms of conv_wgrad_dma.hip, interleaved in one process (us per launch incl. the reduce pass)";
  WGRAD_VARIANTS=0,1,2,3,4,5 timeout 600 python tools/wgrad_pl_ab.py 2>&1 | grep -v amdgpu.ids;
  echo; echo "# timing ablations of the planar asm form (iic_debug_wgrad_ablate: 1 no DMA, 2 no k-steps, 4 no wait/barrier, 8 no partial stores, 16 nt stores)";
  WGRAD_VARIANTS=2 WGRAD_ONLY="5g" WGRAD_ABL_VARIANT=2 WGRAD_ABL=0,1,2,3,4,8,11,16 timeout 600 python tools/wgrad_pl_ab.py 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r06_wgrad_planar_ab.txt
{ echo "# tools/wgrad_trace.sh: rocprofv3 --kernel-trace durations of the main kernel (planar asm form) per ablation code, and of the reduce pass";
  for l in "5g l1" "5g l2" "5g l3" "5g l4"; do echo "== $l"; bash tools/wgrad_trace.sh "$l" 0,1,2,3,4,8,11 | tail -9; done; } > gpurun_out/r06_wgrad_trace.txt
{ echo "# tools/conv_perf.py --no-pw --no-wgrad --frag-ablate 512,16: conv_igemm_bd_kernel with A-fragment addresses as for a 144-byte pitch (512; wrong data) / without A reads (16)";
  timeout 600 python tools/conv_perf.py --no-pw --no-wgrad --frag-ablate 512,16 --iters 30 2>&1 | grep -v amdgpu.ids | cut -c1-330; } > gpurun_out/r06_bd_pitch144_ablation.txt
rm -f gpurun_out/ab_bench.txt
bash tools/ab_bench.sh "" "iic_debug_wgrad_planar=0" "iic_debug_wgrad_planar=2" "iic_debug_wgrad_target_wgs=256" "iic_debug_wgrad_target_wgs=192" "iic_debug_wgrad_target_wgs=128" > /dev/null 2>&1
{ echo "# tools/ab_bench.sh: default step (pair graphs, 20 steps), weight-gradient switches, interleaved twice on one box"; cat gpurun_out/ab_bench.txt; } > gpurun_out/r06_wgrad_step_ab.txt
tail -3 gpurun_out/r06_wgrad_planar_ab.txt; tail -4 gpurun_out/r06_wgrad_trace.txt; tail -3 gpurun_out/r06_bd_pitch144_ablation.txt; cat gpurun_out/r06_wgrad_step_ab.txt

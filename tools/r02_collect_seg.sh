# One gpurun call: round-2 artefacts of the segmentation configs (bench lines, kernel statistics,
# contraction-kernel microbenchmark + SQ counter pass).  Results land in gpurun_out/.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
for cfg in "potsdam3" "potsdam3 --T 10" "coco3"; do
  tag=$(echo $cfg | sed 's/ --T /T/')
  python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r02_bench_seg_$tag.json
  cut -c1-200 gpurun_out/r02_bench_seg_$tag.json
done
bash tools/prof_step.sh r02_potsdam --config potsdam3 --no-branch > /dev/null 2>&1     # (one stream: un-stretched kernel durations)
bash tools/prof_step.sh r02_potsdamT10 --config potsdam3 --T 10 --no-branch > /dev/null 2>&1
bash tools/prof_step.sh r02_coco --config coco3 --no-branch > /dev/null 2>&1
( for c in potsdam coco potsdamB potsdamT1; do python tools/seg_kernel_perf.py $c; done ) > gpurun_out/r02_seg_kernel_perf.txt 2>&1
cat gpurun_out/r02_seg_kernel_perf.txt
( echo "# SQ counter pass over tools/seg_kernel_perf.py (shares of SQ_WAVE_CYCLES; LDS conflict cycles / LDS active cycles; MFMA busy cycles per launch, summed over SIMDs)"
  for c in potsdam coco; do echo "## $c"; bash tools/seg_pmc.sh $c 2>&1 | grep "seg_"; done ) > gpurun_out/r02_seg_pmc.txt
cat gpurun_out/r02_seg_pmc.txt

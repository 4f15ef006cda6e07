#!/bin/bash
# Run the reference's six UNCHANGED training scripts on the MI355X through the launcher's machinery
# (tests/test_gpu_script.py::test_real_reference_scripts_drive_the_hip_kernels).  /root/reference does not exist on
# the GPU box, so a read-only copy of its code/ directory is staged in .refstage/ (git-ignored scratch: never
# committed; gpurun ships it with the snapshot) for this one call and removed afterwards:
#   mkdir -p .refstage && cp -r /root/reference/code .refstage/ && gpurun -- bash tools/real_scripts_gpu.sh; rm -rf .refstage
cd "$GRAFT_REPO_ROOT" || exit 1
test -f .refstage/code/scripts/cluster/cluster_sobel.py || { echo "stage the reference first (see header)"; exit 2; }
mkdir -p gpurun_out
rm -f gpurun_out/real_scripts_gpu.txt gpurun_out/script_real_fp32.txt
IIC_REFERENCE="$GRAFT_REPO_ROOT/.refstage" timeout 1500 python -m pytest tests/test_gpu_script.py -q -m gpu -k "real_${1:-}" 2>&1 | tail -25 | tee gpurun_out/real_scripts_pytest.txt

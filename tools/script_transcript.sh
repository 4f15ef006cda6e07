# (inside gpurun) transcript of `python -m iic_amd.run` driving an unchanged-style Python-2 script on the GPU
R=$GRAFT_REPO_ROOT
T=/tmp/iic_tree
rm -rf $T; mkdir -p $T
cd $R
python - <<'PY'
import sys
sys.path.insert(0, ".")
from tests.test_gpu_script import _make_tree
_make_tree("/tmp/iic_tree")
PY
cd /tmp
echo '$ IIC_REFERENCE=/tmp/iic_tree python -m iic_amd.run code.scripts.cluster.mini_sobel --arch ClusterNet5g --batch_sz 24 --input_sz 32'
PYTHONPATH=$R IIC_REFERENCE=$T python -W ignore -m iic_amd.run code.scripts.cluster.mini_sobel --arch ClusterNet5g --batch_sz 24 --input_sz 32 2>&1 | grep -v amdgpu.ids
echo '$ grep libiic /proc/<pid>/maps   (the run above, checked from inside the script process)'
PYTHONPATH=$R IIC_REFERENCE=$T python -W ignore - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, runpy, os
sys.argv = ["iic_amd.run", "code.scripts.cluster.mini_sobel", "--arch", "ClusterNet6c", "--batch_sz", "24", "--input_sz", "24"]
import iic_amd.run as r
import io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
  r.main(sys.argv[1:])
print([l.split()[-1] for l in open("/proc/self/maps") if "libiic_hip" in l][:1])
print(buf.getvalue().splitlines()[-1][:200])
PY

"""tests/golden/script_cluster_sobel.json: the reference's UNCHANGED training script
(/root/reference/code/scripts/cluster/cluster_sobel.py, Python-2 source through iic_amd.py2compat) driving the
reference's OWN modules on the CPU for three epochs -- ClusterNet5g from code/archs, torch.optim.Adam, the oracle's
line-for-line restatements of IID_loss (+ the .clone() modern autograd needs) and sobel_process -- on the synthetic data,
arguments and seeds of the GPU driver (tests/ref_script_gpu_driver.py: 44 images per loader = 5 batches of 24 + a ragged
one of 12 per epoch, lr 1e-3, k = 20, 2 sub-heads, 32 x 32).  VERDICT r4 next #9: a script-level comparison that the
driver's box can check without the reference tree.

    python -m oracle.gen_golden_script          (build container only: needs /root/reference)

The run is repeated with 1, 2, 4 and 8 BLAS threads, because what a script-level gate can be had to be MEASURED: training
from a random initialisation on 44 synthetic images is chaotic -- the loss climbs from -7e-7 to -0.2 within a dozen
steps by amplifying whatever it is given -- and the reference does not reproduce ITSELF across thread counts (float32
summation order): first step identical, from the second step on the per-call losses differ by 4 % ... 300 %, the
first epoch's loss is -0.0028 with one thread and -0.0201 with eight.  "Epoch loss to 1e-3"  does
not exist for this script, for any implementation; what does exist: the first step tightly, and the later epochs inside
the band the reference's own runs span.

Stored per thread count: every IID_loss call's value in call order (2 sub-heads x 6 batches x 3 epochs), the script's
epoch losses (config.epoch_loss) and accuracies; and the arguments.  Checked on the GPU by
tests/test_gpu_script.py::test_script_loop_in_fp32_mode_matches_the_reference_cpu_run (a restatement of the script's batch
loop, cluster_sobel.py:205-272, on the HIP fp32 path: no tree needed) and, where the tree is staged, by the real script.
"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EPOCHS = 3


def run_once(threads):
  out = tempfile.mkdtemp(prefix="iic_script_golden_")
  env = dict(os.environ, PYTHONPATH=ROOT, IIC_DRIVER_FULL=str(EPOCHS + 1), OMP_NUM_THREADS=str(threads),
             MKL_NUM_THREADS=str(threads))
  r = subprocess.run([sys.executable, "-W", "ignore", os.path.join(ROOT, "tests", "ref_script_driver.py"), out, "cluster_sobel"],
                     env=env, cwd=out, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=3000)
  line = [l for l in r.stdout.splitlines() if l.startswith("IIC_DRIVER_RESULT ")]
  assert line, r.stdout[-2000:] + r.stderr[-2000:]
  res = json.loads(line[0][len("IIC_DRIVER_RESULT "):])
  assert res["error"] is None and len(res["epoch_loss"]) == EPOCHS and len(res["loss_calls"]) == 2 * 6 * EPOCHS, res
  return res


def main():
  runs = {}
  argv = None
  for th in (1, 2, 4, 8):
    res = run_once(th)
    argv = [a for a in res["argv"]]
    runs[str(th)] = {"epoch_loss": res["epoch_loss"], "epoch_acc": res["epoch_acc"], "loss_calls": res["loss_calls"]}
    print("threads %d: epoch losses %s" % (th, res["epoch_loss"]))
  argv[argv.index("--out_root") + 1] = "<out_root>"
  fix = {"script": "code/scripts/cluster/cluster_sobel.py", "epochs": EPOCHS, "argv": argv, "runs_by_blas_threads": runs,
         "note": "reference modules on the CPU (float32); seeds random/numpy/torch = 0 before the script module runs"}
  path = os.path.join(ROOT, "tests", "golden", "script_cluster_sobel.json")
  with open(path, "w") as f:
    json.dump(fix, f, indent=1)
  print("-> %s" % path)


if __name__ == "__main__":
  main()

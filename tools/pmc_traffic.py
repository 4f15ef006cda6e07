"""HBM traffic per kernel launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot
share a pass on gfx950: TCC has 4 counter slots, they cost 3 + 2).

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o p -- \
      python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o p -- (same)
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json

Units and corrections (MI355X_MICROARCH.md "HBM", cdna_hip_programming.md §7): both counters
are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced
streaming read (16 B/lane global_load -- what every kernel here issues), so reads are doubled:
    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def load(d, counter):
  f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
  assert f, "no counter_collection.csv under %s" % d
  per = collections.defaultdict(lambda: [0.0, 0])
  for r in csv.DictReader(open(f[0])):
    if r["Counter_Name"] != counter:
      continue
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    per[name][0] += float(r["Counter_Value"])
    per[name][1] += 1
  return per


def _one_time(kernel_name):
  """Launches that only the first step makes (see step_hbm_bytes)."""
  return "FillFunctor<c10::BFloat16>" in kernel_name


def main():
  fd, wd, out = sys.argv[1:4]
  fe, wr = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
  res = {}
  for k in sorted(set(fe) | set(wr)):
    f, nf = fe.get(k, [0.0, 0])
    w, nw = wr.get(k, [0.0, 0])
    n = max(nf, nw, 1)
    res[k] = {"launches": n, "fetch_kib_raw_per_launch": f / max(nf, 1),
              "write_kib_per_launch": w / max(nw, 1),
              "hbm_bytes_per_launch": (2.0 * f / max(nf, 1) + w / max(nw, 1)) * 1024.0}
  conv = {k: v for k, v in res.items() if k.startswith("conv_igemm")}
  tot_l = sum(v["launches"] for v in conv.values())
  summary = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 1",
    "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE counts 64 B per 128-B request)",
    "conv_igemm_hbm_bytes_per_launch": (sum(v["hbm_bytes_per_launch"] * v["launches"] for v in conv.values())
                                        / max(tot_l, 1)),
    "conv_igemm_launches": tot_l,
    # every launch of the command, per step (the command runs --steps 1 --warmup 1 = 2 steps; weight preparation and
    # the first steps' buffer zero-fills included)
    # one-time launches are left out: the PT buffer pool zero-fills every buffer once, at its first hand-out (all of them
    # in the first step: 156 bf16 fills of ~112 MB at the north-star batch, none afterwards -- the 7-step kernel
    # statistics show the same 156); until round 6's last day they were counted in (125.0 instead of 116.2 GB per step)
    "step_hbm_bytes": sum(v["hbm_bytes_per_launch"] * v["launches"] for k, v in res.items() if not _one_time(k))
                      / float(os.environ.get("PMC_STEPS", "2")),
    "one_time_hbm_bytes": sum(v["hbm_bytes_per_launch"] * v["launches"] for k, v in res.items() if _one_time(k)),
    "kernels": res,
  }
  json.dump(summary, open(out, "w"), indent=1)
  for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:16]:
    print("%-60s %5d launches %10.2f MB/launch" % (k[:60], v["launches"], v["hbm_bytes_per_launch"] / 1e6))
  print("conv_igemm average: %.2f MB/launch over %d launches" % (summary["conv_igemm_hbm_bytes_per_launch"] / 1e6, tot_l))


if __name__ == "__main__":
  main()

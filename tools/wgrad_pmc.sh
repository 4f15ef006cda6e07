#!/bin/bash
# L2 request counters of the weight-gradient kernels at the north-star shapes, per IIC_DEBUG value
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for d in "$@"; do
  rm -rf /tmp/wgp
  IIC_DEBUG="$d" timeout 300 rocprofv3 --pmc TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/wgp -o p -- python $R/tools/conv_perf.py --iters 3 > /tmp/wgp.log 2>&1
  echo "IIC_DEBUG=$d"
  python - <<'PY'
import csv, glob, collections, re
f = glob.glob("/tmp/wgp/**/*counter_collection.csv", recursive=True)
per = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
  k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:70]
  if "wgrad_dma" not in k: continue
  k += " grid%s" % r.get("Grid_Size", r.get("Grid_Size_X", ""))
  per[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k, c in sorted(per.items()):
  m = len(n[k])
  print("%-100s n=%3d  req %.3e  read %.3e  hit %.3e  miss %.3e" % (k, m, c["TCC_REQ_sum"] / m, c["TCC_READ_sum"] / m, c["TCC_HIT_sum"] / m, c["TCC_MISS_sum"] / m))
PY
done

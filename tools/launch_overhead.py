import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("IIC_HIP_LIB", "dbg")      # the iic_debug_* switches live in libiic_hip_dbg.so only (make -C iic_amd/csrc dbg)
import torch
from iic_amd import geom, ops, _lib
dev = torch.device("cuda:0")
N, H, cin, cout = 660, 13, 256, 256
spec = geom.ConvSpec(cin, cout, 3, 1, 1)
gf = geom.fwd_geom(spec, N, H, H, 1, 1)
x = torch.randn(N, H + 2, H + 2, cin, device=dev).to(torch.bfloat16)
y = torch.zeros(N, H + 2, H + 2, cout, device=dev, dtype=torch.bfloat16)
w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
wf, wb = ops.weight_prep(w)
for abl in (0, 63):
    ctypes.CDLL(_lib.LIB_PATH).iic_debug_set_ablate(abl)
    for _ in range(5): ops.conv_igemm(gf, x, wf, y)
    torch.cuda.synchronize()
    n = 300
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): ops.conv_igemm(gf, x, wf, y)
    t1 = time.perf_counter(); e1.record()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("ablate %d: host enqueue %.1f us/call, gpu (events) %.1f us/call, wall %.1f us/call" % (abl, (t1-t0)/n*1e6, e0.elapsed_time(e1)*1e3/n, (t2-t0)/n*1e6))
# small tensor op overhead
a = torch.zeros(1024, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(1000): a.add_(1.0)
t1 = time.perf_counter(); torch.cuda.synchronize()
print("torch add_ host enqueue %.1f us/call" % ((t1-t0)/1000*1e6))

"""Staged MFMA-feed microbenchmark: the K loop of conv_igemm_bd_kernel rebuilt one
ingredient at a time (iic_amd/csrc/probe_mfma_feed.hip), at the loop lengths and grid sizes of the
north-star layers.  python tools/mfma_feed.py [--iters 10] -> TF/s per stage + the shader clock held.

  stage 0  MFMAs only        stage 1  + A ds_read_b128 stream     stage 2  + B global ring
  stage 3  + chunk-boundary barrier / LDS-DMA reload              stage 4  + tile epilogue
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("IIC_HIP_LIB", "dbg")      # the iic_debug_* switches live in libiic_hip_dbg.so only (make -C iic_amd/csrc dbg)
import torch

from iic_amd import _lib

FLOP_PER_MFMA = 2 * 32 * 32 * 16

# name, (tap, chunk) iterations per tile, iterations per chunk, workgroups (= 256 x 128 tiles of one launch)
SHAPES = [
  ("layer2 128->128 @25 (1612 tiles, 18 its)", 18, 9, 1612),
  ("layer3 256->256 @13 ( 872 tiles, 36 its)", 36, 9, 872),
  ("layer4 512->512 @7  ( 508 tiles, 72 its)", 72, 9, 508),
  ("one full round, long loop (512 tiles, 288 its)", 288, 9, 512),
  ("four full rounds (2048 tiles, 36 its)", 36, 9, 2048),
]


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--iters", type=int, default=10)
  ap.add_argument("--npix", type=int, default=330)
  ap.add_argument("--vars", type=str, default="0", help="comma list of probe variants (VAR bits of probe_mfma_feed.hip): "
                  "1 B loads L1-hot, 2 setprio around the MFMAs, 4 interleaved issue order, 6 = 2+4, 8 B ring 16 deep")
  ap.add_argument("--hold", type=str, default="", help="stage,variant,shape index,seconds: run that one launch back to back "
                  "for the given time (tools/power_probe_feed.sh samples rocm-smi meanwhile)")
  a = ap.parse_args()
  dev = torch.device("cuda:0")
  L = ctypes.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), "libiic_probe.so"))   # make -C iic_amd/csrc probes
  fn = L.iic_debug_mfma_feed
  fn.restype = ctypes.c_int
  fn.argtypes = [ctypes.c_int] * 6 + [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_void_p]
  gmax = max(s[3] for s in SHAPES)
  wbytes = 1 << 21                            # ~ a 256 -> 256 3x3 layer's fragment operand (1.2 MB), power of two
  wfrag = (torch.randn(wbytes // 2, device=dev) * 0.05).to(torch.bfloat16)
  patch = torch.randn((gmax * 256 + 1024) * 256, device=dev).to(torch.bfloat16)
  out = torch.empty(gmax * 256 * 128, device=dev, dtype=torch.bfloat16)
  clk = torch.zeros(gmax * 4, device=dev, dtype=torch.int64)
  if a.hold:
    import time
    stage, var, si, secs = a.hold.split(",")
    stage, var, si, secs = int(stage), int(var), int(si), float(secs)
    name, nit, cits, grid = SHAPES[si]
    t_end = time.time() + secs
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() < t_end:
      for _ in range(50):
        rc = fn(stage | (var << 4), grid, nit, cits, a.npix, 80 * 1024, wfrag.data_ptr(), wbytes, patch.data_ptr(), out.data_ptr(),
                None, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
      n += 50
      torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    flops = grid * 4.0 * nit * 4 * 8 * FLOP_PER_MFMA
    print("hold: stage %d variant %d %s: %.1f us per launch, %.0f TF/s over %.1f s" % (stage, var, name, us, flops / us / 1e6, secs))
    return
  print("stages: 0 MFMA only | 1 +A LDS reads | 2 +B global ring | 3 +chunk reloads | 4 +epilogue   (random operands)")
  variants = [int(v) for v in a.vars.split(",")]
  for var, (occ_name, lds) in [(v, o) for v in variants for o in (("2 workgroups/CU (2 waves/SIMD)", 80 * 1024),
                                                                    ("1 workgroup/CU (1 wave/SIMD)", 100 * 1024))]:
    print("== variant %d, %s" % (var, occ_name))
    for name, nit, cits, grid in SHAPES:
      row = []
      for stage in range(5):
        def run(c=None):
          rc = fn(stage | (var << 4), grid, nit, cits, a.npix, lds, wfrag.data_ptr(), wbytes, patch.data_ptr(), out.data_ptr(),
                  c, torch.cuda.current_stream().cuda_stream)
          assert rc == 0, rc
        for _ in range(2):
          run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
          run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.iters
        run(clk.data_ptr())
        torch.cuda.synchronize()
        c = clk[:grid * 4].view(grid, 4).cpu().numpy().astype("float64")
        cyc = (c[:, 1] - c[:, 0])
        ticks = (c[:, 3] - c[:, 2])
        ghz = float((cyc.sum() / max(ticks.sum(), 1.0)) * 0.1)     # s_memrealtime = 100 MHz
        flops = grid * 4.0 * nit * 4 * 8 * FLOP_PER_MFMA
        row.append((us, flops / us / 1e6, ghz, float(cyc.mean())))
      print("%-50s" % name + " | ".join("s%d %7.1f us %6.0f TF/s %.2f GHz %6.0f cyc/tile" % ((i,) + r) for i, r in enumerate(row)))
  print("peak: 2500 TF/s at 2.4 GHz = 1042 TF/s per GHz; MFMA-only cycles per tile = its x 4 x 8 x 32 x (waves per SIMD)")


if __name__ == "__main__":
  main()

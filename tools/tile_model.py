"""Paper model of the conv tilings at the north-star shapes (no GPU): how many input pixels a tile
stages per output pixel (L2 -> LDS amplification) and how well the tile count fills 256 CUs, for
the row-major tiles the kernels use today and for 2-D spatial tiles.

  python tools/tile_model.py [--n 660]

Row-major tile of BM output pixels (dense numbering across images): stages every padded pixel from
the first tap of its first row to the last tap of its last row (what iic_conv_geom.NP / NP256
hold).  2-D tile th x tw (3x3, stride 1): stages (th + 2) x (tw + 2).
"""
import argparse
import math

LAYERS = [("layer1 64->64", 49, 64, 64), ("layer2 128->128", 25, 128, 128),
          ("layer3 256->256", 13, 256, 256), ("layer4 512->512", 7, 512, 512)]


def row_major_patch(W, BM):
  """Worst-case padded pixels between tap (0,0) of the first row and tap (2,2) of the last."""
  Wp = W + 2
  worst = 0
  for start in range(W * W):                      # start pixel within an image (dense index)
    y0, x0 = divmod(start, W)
    end = start + BM - 1
    img_jump, e = divmod(end, W * W)
    y1, x1 = divmod(e, W)
    first = y0 * Wp + x0
    last = (img_jump * Wp * Wp) + (y1 + 2) * Wp + x1 + 2
    worst = max(worst, last - first + 1)
  return worst


def fill(tiles, slots):
  rounds = tiles / float(slots)
  return rounds, rounds / math.ceil(rounds)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--n", type=int, default=660)
  a = ap.parse_args()
  print("%-18s %-22s %8s %8s %10s %8s %8s" % ("layer", "tile", "px/tile", "staged", "amplif.", "rounds", "fill"))
  for name, W, cin, cout in LAYERS:
    M = a.n * W * W
    ntile_n = max(1, cout // 128) if cout >= 128 else 1
    slots = 512 if cout >= 128 else 256          # two 4-wave workgroups per CU / one persistent
    for BM in (128, 256):
      np_ = row_major_patch(W, BM)
      tiles = math.ceil(M / BM) * ntile_n
      r, f = fill(tiles, slots)
      print("%-18s %-22s %8d %8d %10.2f %8.2f %8.2f" % (name, "row-major %d" % BM, BM, np_, np_ / BM, r, f))
    for th, tw in ((8, 16), (16, 16), (8, 32), (16, 32), (7, 7), (13, 13)):
      if th > W or tw > W:
        continue
      per_img = math.ceil(W / th) * math.ceil(W / tw)
      useful = W * W / float(per_img * th * tw)   # MFMA rows that are real pixels
      staged = (th + 2) * (tw + 2)
      tiles = a.n * per_img * ntile_n
      r, f = fill(tiles, slots)
      print("%-18s %-22s %8d %8d %10.2f %8.2f %8.2f   (%.0f %% of the tile rows are pixels)" % (
        name, "2-D %dx%d" % (th, tw), th * tw, staged, staged / (th * tw * useful), r, f, 100 * useful))
    print()


if __name__ == "__main__":
  main()

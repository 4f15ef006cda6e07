// Convolution backward-weight, DMA-fed kernels for the 3x3 layers (the bulk of
// /root/reference/code/archs/cluster/residual.py:4-7,19,22; vgg.py:24-26 where the patch fits LDS twice).
// Three generations of the same work split live in this file and produce the same bits:
//   conv_wgrad_dma_kernel   (rounds 2-5)  swizzled 128-byte patch rows; fallback for tap layouts / sizes the planar
//                                         kernels refuse (and the A/B baseline, iic_debug_wgrad_planar = 0)
//   conv_wgrad_pl_kernel    (round 6)     planar patch: read addresses = per-tile base + immediates; the default for
//                                         128-cout tiles
//   conv_wgrad_pl2_kernel   (round 6)     planar patch + per-tile work off the critical path + software-pipelined
//                                         k-steps; the default for 64-cout tiles
//   conv_wgrad_b2d_kernel   (round 6)     block-tiled K for images >= 32 pixels (a K-tile = a 2-D block of output pixels;
//                                         its own summation order): the large-image layers of SegmentationNet10a
// (see the comments at each kernel and LAB.md R6.8).  First generation: same math and work split as
// conv_wgrad.hip,
//   dW[t][co][ci] = sum_m dY[pout(m)][co] * X[pin(m) + tap_off[t]][ci],
// workgroup = COT co x 64 ci x 9 taps over a range of 128-pixel K-tiles, 12 waves
// (2 co halves x 2 ci halves x 3 tap groups), operands read with ds_read_b64_tr_b16 --
// but the K-tile pipeline is rebuilt around LDS-DMA:
//   * the input patch and the dY rows of K-tile kt+1 are fetched by global_load_lds_dwordx4 into
//     the second LDS buffer while K-tile kt is multiplied: no staging registers, no ds_write
//     pass, ONE barrier per K-tile (conv_wgrad.hip: register staging, 75 KB of ds_write_b128 and
//     two barriers per K-tile with the matrix pipe idle in between);
//   * LDS rows are unpadded (DMA writes lane-linear).  A transposing read covers 4 consecutive
//     rows x 64 B per LDS cycle, so the 64-byte unit of a row is XOR-swizzled: 128-B rows
//     (input patch; dY at COT = 64) unit ^= (row >> 1) & 1, 256-B rows (dY at COT = 128)
//     unit ^= row & 3; the DMA applies the swizzle on its source address;
//   * rows past the end of the batch take their dY from pixel 0 of the PT tensor (zero border);
//   * row bookkeeping by (n, y, x) walkers, tables written ahead of their use;
//   * the DMA ring is NBUF deep (K-tiles of BMK = 64 pixels, 4 buffers, where LDS allows): with
//     one K-tile of prefetch the ~2-4 us HBM latency exceeded the ~1-2 us of MFMA work per tile
//     and every tile waited for its data; each wave issues a FIXED number of DMA instructions per
//     tile so that a counted s_waitcnt vmcnt(N) retires exactly tile kt.
#include <type_traits>
#include "common.h"
#include "../../include/iic_hip.h"

#define WD_THREADS 768
#define WD_NTAB 8                               // table ring (>= NBUF + 2, power of two)
#define WD_TAB_BYTES(BMK_) (WD_NTAB * (BMK_) * (4 + 2))

__device__ __forceinline__ void wd_wait_vmcnt(int n) {   // counted wait, n uniform
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
    case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
    case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
    case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
    case 23: asm volatile("s_waitcnt vmcnt(23)" ::: "memory"); break;
    case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // over-waiting is always safe
  }
}

typedef s16x4 __attribute__((address_space(3))) * wd_lds_s16x4_ptr;

__device__ __forceinline__ bf16x8 wd_frag(uint32_t a0, uint32_t a1) {
  union { bf16x8 v; s16x4 h[2]; } u;
  u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wd_lds_s16x4_ptr)(uintptr_t)a0);
  u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wd_lds_s16x4_ptr)(uintptr_t)a1);
  return u.v;
}

// The same read as inline asm.  hipcc treats the ds_read_tr builtin as a possible reader of LDS that
// an in-flight LDS-DMA is writing and puts s_waitcnt vmcnt(0) in front of the first one -- right
// after the next K-tile's DMA was issued, which serialised DMA and MFMA inside the workgroup (the
// stall profile showed its waves parked 34-39 % of the time).  The compiler cannot see through the
// asm, so the DMA stays in flight until the counted wait at the top of the next tile; the price is
// that lgkmcnt has to be waited for by hand (the wait asm in the K loop ties the fragments to it).
// Opt-in (g_wd_asm): it turned out not to change the kernel's time, see there.
__device__ __forceinline__ void wd_tr_issue(uint32_t addr, s16x4& d) {
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(d) : "v"(addr));
}
__device__ __forceinline__ bf16x8 wd_pack(const s16x4 h0, const s16x4 h1) {
  union { bf16x8 v; s16x4 h[2]; } u;
  u.h[0] = h0;
  u.h[1] = h1;
  return u.v;
}

// Row walkers.  Dense numbering (g.MP == 0 or == MY*MX): (n, y, x) advanced by one K-tile at a time, no
// divisions.  Padded numbering (g.MP rows per image, a multiple of 256 > MY*MX: large images, so that no tile
// straddles two images -- iic_amd/geom.py): the walker keeps (n, r = row within the image) in (n, y) and
// divides once per table entry; rows r >= MY*MX are invalid (conv_tile.h igemm_row_pixels: they read the
// image's last pixel and get pout = -1, i.e. a zero dY row).
struct WdWalk {
  int n, y, x;
};
__device__ __forceinline__ bool wd_padded(const iic_conv_geom& g) { return g.MP > 0 && g.MP != g.MY * g.MX; }
__device__ __forceinline__ void wd_walk_init(WdWalk& w, const iic_conv_geom& g, int m) {
  if (wd_padded(g)) {
    w.n = m / g.MP;
    w.y = m - w.n * g.MP;      // r
    w.x = 0;
    return;
  }
  const int plane = g.MY * g.MX;
  w.n = m / plane;
  const int r = m - w.n * plane;
  w.y = r / g.MX;
  w.x = r - w.y * g.MX;
}
__device__ __forceinline__ void wd_walk_advance(WdWalk& w, const iic_conv_geom& g, int d_y, int d_x, int d_rows) {
  if (wd_padded(g)) {
    w.y += d_rows;
    while (w.y >= g.MP) { w.y -= g.MP; ++w.n; }
    return;
  }
  w.x += d_x;
  w.y += d_y;
  if (w.x >= g.MX) { w.x -= g.MX; ++w.y; }
  while (w.y >= g.MY) { w.y -= g.MY; ++w.n; }
}
__device__ __forceinline__ void wd_walk_pixels(const WdWalk& w, const iic_conv_geom& g, int& pin,
                                               int& pout) {
  int n, y, x;
  bool valid;
  if (wd_padded(g)) {
    const int plane = g.MY * g.MX;
    valid = w.n < g.N && w.y < plane;
    n = w.n < g.N ? w.n : g.N - 1;
    const int r = valid ? w.y : plane - 1;
    y = r / g.MX;
    x = r - y * g.MX;
  } else {
    valid = w.n < g.N;
    n = valid ? w.n : g.N - 1;
    y = valid ? w.y : g.MY - 1;
    x = valid ? w.x : g.MX - 1;
  }
  pin = (n * g.in_Hp + y * g.sy + g.oy) * g.in_Wp + x * g.sx + g.ox;
  pout = valid ? (n * g.out_Hp + y * g.ty + g.py) * g.out_Wp + x * g.tx + g.px : -1;
}

// SWP (round 6, 12-wave form only): the fragments of k-step ks + 1 are read while the MFMAs of k-step ks run (two
// register sets, compile-time indices) -- a wave no longer parks on lgkmcnt(0) between issuing its 10 transposing
// reads and its 6 MFMAs eight times per K-tile (r05_pmc_stalls.txt: 35 % of the wave cycles parked, 34 % of the matrix
// pipe busy).  Three waves per SIMD stay (<= 170 registers), unlike the 4-wave PF form.
template <int COT, int WD_BM, int NBUF, bool ASMRD, bool PF, bool SWP = false>
__global__ __launch_bounds__(PF ? 256 : WD_THREADS) void conv_wgrad_dma_kernel(
    const iic_conv_geom g, const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
    float* __restrict__ partials, int nsplit, int num_ktiles, int xb_bytes, int max_tap_off, int abl) {
  constexpr int CS = COT / 64;                  // 32-wide co sub-tiles per wave
  // PF: 4 fat waves (one per SIMD), each with all 9 taps of its (co half, ci half): 11 operand
  // fragments feed 18 MFMAs per k-step (22 transposing reads per 18 MFMAs instead of 10 per 6: the
  // kernel is LDS-read-bound), the next k-step's fragments are read under the current MFMAs.
  constexpr int NTG = PF ? 1 : 3;               // tap groups = waves / 4
  constexpr int TPW = 9 / NTG;                  // taps per wave
  constexpr int NTH = 256 * NTG;                // threads
  constexpr int DROW = COT * 2;                 // dY tile row bytes (256 | 128)
  constexpr int DB = WD_BM * DROW;              // dY tile bytes
  constexpr int DBLK = DB / 1024;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  typedef unsigned char __attribute__((address_space(3))) * lds_u8_ptr;
  unsigned char* const sX = smem_raw;                        // [NBUF][xb_bytes]
  unsigned char* const sD = smem_raw + NBUF * xb_bytes;      // [NBUF][DB]
  int* const s_pout = reinterpret_cast<int*>(sD + NBUF * DB);   // [NTAB][BMK]
  unsigned short* const s_prow = reinterpret_cast<unsigned short*>(s_pout + WD_NTAB * WD_BM);
  const uint32_t sXo = (uint32_t)(uintptr_t)(lds_u8_ptr)sX, sDo = (uint32_t)(uintptr_t)(lds_u8_ptr)sD;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tg = PF ? 0 : wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
  const int l31 = lane & 31;
  const int q = lane >> 4, i16 = lane & 15;
  const int ncit = g.Cin >> 6;
  // XCD-aware work mapping: workgroups are dealt to the 8 XCDs round-robin by linear id.  All
  // (co, ci) tiles of one K-split read the same input patch / dY rows, so they should share an
  // L2: when the split count divides by 8, linear ids L, L+8, L+16, ... (one XCD) walk the tiles
  // of one split before moving to the next split.
  int tile = blockIdx.x, split = blockIdx.y;
  if ((gridDim.y & 7) == 0) {
    const int L = blockIdx.y * gridDim.x + blockIdx.x;
    const int xcd = L & 7, j = L >> 3;
    tile = j % (int)gridDim.x;
    split = xcd + 8 * (j / (int)gridDim.x);
  }
  const int cot = tile / ncit, cit = tile - cot * ncit;
  const int co0 = cot * COT, ci0 = cit * 64;
  const int per = (num_ktiles + nsplit - 1) / nsplit;
  const int kt0 = split * per;
  const int kt1 = min(num_ktiles, kt0 + per);
  const int in_pixels = g.N * g.in_Hp * g.in_Wp;
  const int v_tapoff = g.tap_off[lane & (IIC_MAX_TAPS - 1)];
  const int tfirst = tg * TPW;                   // this wave's taps: tfirst .. tfirst + TPW - 1

  // lane constants of the transposing reads (see conv_wgrad.hip::frag_T)
  const int trow = 8 * (q >> 1) + (i16 >> 2);                         // + 16*ks (+4 for rd=1)
  const int tsub = (2 * (q & 1) + ((i16 & 3) >> 1)) * 16 + (i16 & 1) * 8;   // byte in the 64-B unit
  // dY fragment c: logical 64-B unit wm*CS + c of row k; its swizzle key only depends on trow
  uint32_t aoff[CS];
#pragma unroll
  for (int c = 0; c < CS; ++c) {
    const int unit = (COT == 128) ? ((wm * 2 + c) ^ (trow & 3)) : (wm ^ ((trow >> 1) & 1));
    aoff[c] = trow * DROW + unit * 64 + tsub;
  }
  // X fragment: unit wn of row R, physical unit = wn ^ ((R >> 1) & 1)
  const uint32_t xoff = tsub + wn * 64;
  const int xs32 = wn ? -32 : 32;

  f32x16 acc[TPW][CS];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int c = 0; c < CS; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][c][r] = 0.f;

  if (kt0 < kt1) {
    const int d_y = WD_BM / g.MX, d_x = WD_BM - d_y * g.MX;
    WdWalk wr, w0, w1;
    wd_walk_init(wr, g, kt0 * WD_BM + (tid & (WD_BM - 1)));
    wd_walk_init(w0, g, kt0 * WD_BM);
    wd_walk_init(w1, g, kt0 * WD_BM + WD_BM - 1);
    // first input pixel / block count of the tabulated tiles (uniform), ring in LDS
    int* const s_plo = reinterpret_cast<int*>(s_prow + WD_NTAB * WD_BM);   // [NTAB][2]
    auto tabulate = [&](int kt) {   // table of K-tile kt (where the walkers stand), then advance
      int pin, pout, p0, p1, dummy;
      wd_walk_pixels(w0, g, p0, dummy);
      wd_walk_pixels(w1, g, p1, dummy);
      p0 = __builtin_amdgcn_readfirstlane(p0);
      p1 = __builtin_amdgcn_readfirstlane(p1);
      if (tid < WD_BM) {
        wd_walk_pixels(wr, g, pin, pout);
        s_pout[(kt & (WD_NTAB - 1)) * WD_BM + tid] = pout;
        s_prow[(kt & (WD_NTAB - 1)) * WD_BM + tid] = (unsigned short)(pin - p0);
      }
      wd_walk_advance(wr, g, d_y, d_x, WD_BM);
      wd_walk_advance(w0, g, d_y, d_x, WD_BM);
      wd_walk_advance(w1, g, d_y, d_x, WD_BM);
      if (tid == 0) {
        s_plo[(kt & (WD_NTAB - 1)) * 2] = p0;
        s_plo[(kt & (WD_NTAB - 1)) * 2 + 1] = ((p1 + max_tap_off - p0 + 1) * 128 + 1023) >> 10;
      }
    };
    // DMA of one K-tile: input patch (nblk 1-KB blocks) + dY rows (DBLK blocks) over 12 waves.
    // Every wave issues exactly NI instructions per tile (surplus ones re-fetch the tile's last
    // block: same data to the same address) so that vmcnt counts tiles.
    const int NI = ((xb_bytes >> 10) + DBLK + NTH / 64 - 1) / (NTH / 64);
    auto dma_issue = [&](int buf, int kt) {
      const int tab = kt & (WD_NTAB - 1);
      const int plo = __builtin_amdgcn_readfirstlane(s_plo[tab * 2]);
      const int nblk = __builtin_amdgcn_readfirstlane(s_plo[tab * 2 + 1]);
      unsigned char* const dX = sX + buf * xb_bytes;
      unsigned char* const dD = sD + buf * DB;
      for (int i = 0; i < NI; ++i) {
        int b = wave + i * (NTH / 64);
        b = b < nblk + DBLK ? b : nblk + DBLK - 1;
        if (b < nblk) {
          const int qq = b * 64 + lane;
          const int r = qq >> 3, slot = qq & 7;
          const int ls = slot ^ (((r >> 1) & 1) << 2);
          int p = plo + r;
          p = p < in_pixels ? p : in_pixels - 1;
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void*)(x + ((long)p * g.Cin + ci0 + ls * 8)),
              (__attribute__((address_space(3))) void*)(dX + b * 1024), 16, 0, 0);
        } else {
          const int bd = b - nblk;
          int row, ls;
          if (COT == 128) {
            row = bd * 4 + (lane >> 4);
            ls = (lane & 15) ^ ((row & 3) << 2);
          } else {
            row = bd * 8 + (lane >> 3);
            ls = (lane & 7) ^ (((row >> 1) & 1) << 2);
          }
          int po = s_pout[tab * WD_BM + row];
          po = po < 0 ? 0 : po;                       // pixel 0 = zero border of the PT tensor
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void*)(dy + ((long)po * g.Cout + co0 + ls * 8)),
              (__attribute__((address_space(3))) void*)(dD + bd * 1024), 16, 0, 0);
        }
      }
    };

    // prologue: tables of the first NBUF tiles, DMA of the first NBUF-1
#pragma unroll
    for (int i = 0; i < NBUF; ++i) tabulate(kt0 + i);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i)
      if (kt0 + i < kt1) dma_issue(i, kt0 + i);

    for (int kt = kt0; kt < kt1; ++kt) {
      const int b = (kt - kt0) % NBUF;
      // tiles kt+1 .. kt+NBUF-2 may stay in flight; tile kt must have landed
      const int ahead = min(NBUF - 2, kt1 - 1 - kt);
      wd_wait_vmcnt(ahead * NI);
      __syncthreads();                                    // everyone's share landed; tile kt-1 consumed
      if (kt + NBUF - 1 < kt1 && !(abl & 1)) dma_issue((b + NBUF - 1) % NBUF, kt + NBUF - 1);   // abl 1: timing only
      tabulate(kt + NBUF);
      const unsigned short* prow = s_prow + (kt & (WD_NTAB - 1)) * WD_BM;
      const uint32_t xb = sXo + b * xb_bytes + xoff;
      const uint32_t db = sDo + b * DB;
      // this lane's patch rows of all 8 k-steps up front (one LDS round trip per K-tile)
      int rr0[WD_BM / 16], rr1[WD_BM / 16];
#pragma unroll
      for (int ks = 0; ks < WD_BM / 16; ++ks) {
        rr0[ks] = prow[ks * 16 + trow];
        rr1[ks] = prow[ks * 16 + trow + 4];
      }
      if (!ASMRD && PF) {
        // fragments of k-step ks+1 are read while the MFMAs of k-step ks run (two register sets,
        // the loop is fully unrolled): a wave no longer sits out the LDS round trip of its 10
        // transposing reads once per k-step
        // pipeline unit = (k-step, group of 3 taps): dY fragments of the k-step + 3 input fragments.
        // Unit u+1 is read while unit u's 6 MFMAs run; everything is unrolled (compile-time indices).
        constexpr int NU = (WD_BM / 16) * 3;
        bf16x8 fa[2][CS], fb[2][3];
        auto load_unit = [&](int u) {
          const int ks = u / 3, gq = u - 3 * ks;
          if (gq == 0) {
#pragma unroll
            for (int c = 0; c < CS; ++c)
              fa[ks & 1][c] = wd_frag(db + ks * 16 * DROW + aoff[c], db + (ks * 16 + 4) * DROW + aoff[c]);
          }
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const int toff = __builtin_amdgcn_readlane(v_tapoff, tfirst + 3 * gq + t);
            const int R0 = rr0[ks] + toff, R1 = rr1[ks] + toff;
            fb[u & 1][t] = wd_frag(xb + (R0 << 7) + (R0 & 2) * xs32, xb + (R1 << 7) + (R1 & 2) * xs32);
          }
        };
        load_unit(0);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          if (u + 1 < NU) load_unit(u + 1);
          __builtin_amdgcn_sched_barrier(0);
          const int ks = u / 3, gq = u - 3 * ks;
#pragma unroll
          for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int c = 0; c < CS; ++c)
              acc[3 * gq + t][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][c], fb[u & 1][t],
                                                                         acc[3 * gq + t][c], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else if (SWP) {
        constexpr int NKS = WD_BM / 16;
        bf16x8 fa[2][CS], fb[2][3];
        auto load_ks = [&](int ks) {
#pragma unroll
          for (int c = 0; c < CS; ++c)
            fa[ks & 1][c] = wd_frag(db + ks * 16 * DROW + aoff[c], db + (ks * 16 + 4) * DROW + aoff[c]);
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const int toff = __builtin_amdgcn_readlane(v_tapoff, tfirst + t);
            const int R0 = rr0[ks] + toff, R1 = rr1[ks] + toff;
            fb[ks & 1][t] = wd_frag(xb + (R0 << 7) + (R0 & 2) * xs32, xb + (R1 << 7) + (R1 & 2) * xs32);
          }
        };
        load_ks(0);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          if (ks + 1 < NKS) load_ks(ks + 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int c = 0; c < CS; ++c)
              acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][c], fb[ks & 1][t], acc[t][c], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
  #pragma unroll
        for (int ks = 0; ks < WD_BM / 16; ++ks) {
          bf16x8 a[CS], bfr[3];
          if (ASMRD) {
            s16x4 ah[CS][2], bh[3][2];
  #pragma unroll
            for (int c = 0; c < CS; ++c) {
              wd_tr_issue(db + ks * 16 * DROW + aoff[c], ah[c][0]);
              wd_tr_issue(db + (ks * 16 + 4) * DROW + aoff[c], ah[c][1]);
            }
  #pragma unroll
            for (int t = 0; t < 3; ++t) {
              const int toff = __builtin_amdgcn_readlane(v_tapoff, tfirst + t);
              const int R0 = rr0[ks] + toff, R1 = rr1[ks] + toff;
              wd_tr_issue(xb + (R0 << 7) + (R0 & 2) * xs32, bh[t][0]);
              wd_tr_issue(xb + (R1 << 7) + (R1 & 2) * xs32, bh[t][1]);
            }
            // all reads of the k-step in flight; wait, and make every fragment depend on the wait
            if (CS == 2)
              asm volatile("s_waitcnt lgkmcnt(0)"
                           : "+v"(ah[0][0]), "+v"(ah[0][1]), "+v"(ah[CS - 1][0]), "+v"(ah[CS - 1][1]));
            else
              asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[0][0]), "+v"(ah[0][1]));
            asm volatile("" : "+v"(bh[0][0]), "+v"(bh[0][1]), "+v"(bh[1][0]), "+v"(bh[1][1]),
                              "+v"(bh[2][0]), "+v"(bh[2][1]));
  #pragma unroll
            for (int c = 0; c < CS; ++c) a[c] = wd_pack(ah[c][0], ah[c][1]);
  #pragma unroll
            for (int t = 0; t < 3; ++t) bfr[t] = wd_pack(bh[t][0], bh[t][1]);
          } else {
  #pragma unroll
            for (int c = 0; c < CS; ++c)
              a[c] = wd_frag(db + ks * 16 * DROW + aoff[c], db + (ks * 16 + 4) * DROW + aoff[c]);
  #pragma unroll
            for (int t = 0; t < 3; ++t) {
              const int toff = __builtin_amdgcn_readlane(v_tapoff, tfirst + t);
              const int R0 = rr0[ks] + toff, R1 = rr1[ks] + toff;
              bfr[t] = wd_frag(xb + (R0 << 7) + (R0 & 2) * xs32, xb + (R1 << 7) + (R1 & 2) * xs32);
            }
          }
          __builtin_amdgcn_sched_barrier(0);   // all 10 transposing reads of the k-step in flight
  #pragma unroll
          for (int t = 0; t < 3; ++t)
  #pragma unroll
            for (int c = 0; c < CS; ++c)
              acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[c], bfr[t], acc[t][c], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  // partial[split][t][co][ci]
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    float* dst = partials + (((long)split * g.ntaps + (tfirst + t)) * g.Cout + co0) * g.Cin + ci0;
#pragma unroll
    for (int c = 0; c < CS; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * (COT / 2) + c * 32 + mfma32_row(r, lane);
        const int col = wn * 32 + l31;
        dst[(long)row * g.Cin + col] = acc[t][c][r];
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Round 6, second generation of the K loop ("planar patch").  The ISA of the kernel above spends 34 VALU
// instructions per k-step on the ADDRESSES of its 10 transposing reads (5 per input-patch read: the row's XOR swizzle
// is a function of row + tap offset, so nothing of it is loop-invariant) against 6 MFMAs, and VALU and MFMA share a
// SIMD's issue port: 3 waves x (48 MFMA + ~500 VALU) issue slots per K-tile exceed the 3 x 48 x 8 slots the matrix
// pipe needs (LAB.md R6.8).  Here the 64-channel input patch lives in LDS as TWO PLANES of 64-byte half rows
// (channels 0-31 | 32-63): a wave (one channel half) reads rows R .. R+3 of its plane = 256 consecutive bytes = every
// bank once, with NO swizzle, so a read's address is  plane + 64 * (row + tap offset)  -- the per-row part is
// tabulated (pre-multiplied) once per K-tile and the three taps of a wave's tap row are IMMEDIATE offsets
// (0, 64*TXS, 128*TXS bytes).  The dY fragments' addresses are a per-tile base + immediates.  The DMA applies the layout
// on its source side (a 1-KB block = 16 half rows of one plane); its global addresses are a uniform base + a lane
// constant (input) or + a tabulated row offset (dY).  Same work split, same k order, same MFMA sequence as above:
// bit-identical partial sums.
// XCD-aware work mapping for ANY split count (the first-generation kernels remap only when the split count divides by
// 8): workgroups are dealt to the 8 XCDs round-robin by linear id L, so XCD x runs L = x, x + 8, ...; its j-th workgroup
// takes virtual index v = (workgroups of the XCDs before it) + j, and v walks the (co, ci) tiles of one K-split before it
// moves to the next split -- the tiles of a split read the same input patch / dY rows and now meet in one L2.
__device__ __forceinline__ void wdp_xcd_map(int& tile, int& split) {
  const int tiles = (int)gridDim.x, G = tiles * (int)gridDim.y;
  const int L = (int)blockIdx.y * tiles + (int)blockIdx.x;
  const int xcd = L & 7, j = L >> 3;
  const int q = G >> 3, r = G & 7;
  const int v = xcd * q + (xcd < r ? xcd : r) + j;
  split = v / tiles;
  tile = v - split * tiles;
}

template <int OFF>
__device__ __forceinline__ void wdp_tr_asm(uint32_t addr, s16x4& d) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int OFF, bool ASMRD>
__device__ __forceinline__ bf16x8 wdp_frag(uint32_t a0, uint32_t a1) {
  union { bf16x8 v; s16x4 h[2]; } u;
  if (ASMRD) {
    wdp_tr_asm<OFF>(a0, u.h[0]);
    wdp_tr_asm<OFF>(a1, u.h[1]);
  } else {
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wd_lds_s16x4_ptr)(uintptr_t)(a0 + OFF));
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wd_lds_s16x4_ptr)(uintptr_t)(a1 + OFF));
  }
  return u.v;
}


template <int OFF0, int DOFF, bool ASMRD>
__device__ __forceinline__ bf16x8 wdp_frag_pair(uint32_t a) {     // the two halves at a + OFF0, a + OFF0 + DOFF
  union { bf16x8 v; s16x4 h[2]; } u;
  if (ASMRD) {
    wdp_tr_asm<OFF0>(a, u.h[0]);
    wdp_tr_asm<OFF0 + DOFF>(a, u.h[1]);
  } else {
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wd_lds_s16x4_ptr)(uintptr_t)(a + OFF0));
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wd_lds_s16x4_ptr)(uintptr_t)(a + OFF0 + DOFF));
  }
  return u.v;
}
template <int N, int I = 0, class F>
__device__ __forceinline__ void wdp_unroll(F& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    wdp_unroll<N, I + 1>(f);
  }
}

// NTAB: row-table ring (power of two >= NBUF + 2): 8, or 4 where the last 4 KB decide whether the layout fits
template <int COT, int WD_BM, int NBUF, int TXS, bool ASMRD, int NTAB = WD_NTAB>
__global__ __launch_bounds__(WD_THREADS) void conv_wgrad_pl_kernel(
    const iic_conv_geom g, const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
    float* __restrict__ partials, int nsplit, int num_ktiles, int plane_bytes, int max_tap_off, int abl,
    int band_pitch, int band_stride) {
  constexpr int CS = COT / 64;
  constexpr int NW = WD_THREADS / 64;           // 12 waves: 2 co halves x 2 ci halves x 3 tap rows
  constexpr int DROW = COT * 2;
  constexpr int DB = WD_BM * DROW;
  constexpr int DBLK = DB / 1024;
  constexpr int NKS = WD_BM / 16;
  static_assert((NTAB & (NTAB - 1)) == 0 && NTAB >= NBUF + 2, "table ring: a power of two, at least NBUF + 2 tiles");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  typedef unsigned char __attribute__((address_space(3))) * lds_u8_ptr;
  const int xb_bytes = 2 * plane_bytes;
  unsigned char* const sX = smem_raw;                        // [NBUF][2 planes][plane_bytes]
  unsigned char* const sD = smem_raw + NBUF * xb_bytes;      // [NBUF][DB]
  uint32_t* const s_dyoff = reinterpret_cast<uint32_t*>(sD + NBUF * DB);   // [NTAB][BMK] byte offset of the dY row
  int* const s_pin = reinterpret_cast<int*>(s_dyoff + NTAB * WD_BM);    // [NTAB][BMK] input pixel of the row (tap 0)
  const uint32_t sXo = (uint32_t)(uintptr_t)(lds_u8_ptr)sX, sDo = (uint32_t)(uintptr_t)(lds_u8_ptr)sD;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tg = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
  const int l31 = lane & 31;
  const int q = lane >> 4, i16 = lane & 15;
  const int ncit = g.Cin >> 6;
  int tile, split;
  wdp_xcd_map(tile, split);
  const int cot = tile / ncit, cit = tile - cot * ncit;
  const int co0 = cot * COT, ci0 = cit * 64;
  const int per = (num_ktiles + nsplit - 1) / nsplit;
  const int kt0 = split * per;
  const int kt1 = min(num_ktiles, kt0 + per);
  const int in_pixels = g.N * g.in_Hp * g.in_Wp;
  const int tfirst = tg * 3;                     // this wave's tap row: taps tfirst .. tfirst + 2
  const int toff0 = __builtin_amdgcn_readfirstlane(g.tap_off[tfirst]);

  const int trow = 8 * (q >> 1) + (i16 >> 2);
  const int tsub = (2 * (q & 1) + ((i16 & 3) >> 1)) * 16 + (i16 & 1) * 8;
  uint32_t aoff[CS];
#pragma unroll
  for (int c = 0; c < CS; ++c) {
    const int unit = (COT == 128) ? ((wm * 2 + c) ^ (trow & 3)) : (wm ^ ((trow >> 1) & 1));
    aoff[c] = trow * DROW + unit * 64 + tsub;
  }
  // DMA lane constants: input block = 16 half rows x 4 pieces; dY block = 4 (8) rows x 16 (8) pieces, swizzled
  const uint32_t lane_x = (uint32_t)(lane >> 2) * (uint32_t)(g.Cin * 2) + (lane & 3) * 16;
  const int drow_l = (COT == 128) ? (lane >> 4) : (lane >> 3);
  const uint32_t lane_d = (COT == 128) ? (uint32_t)(((lane & 15) ^ ((drow_l & 3) << 2)) * 16)
                                       : (uint32_t)(((lane & 7) ^ (((drow_l >> 1) & 1) << 2)) * 16);

  f32x16 acc[3][CS];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int c = 0; c < CS; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][c][r] = 0.f;

  if (kt0 < kt1) {
    const int d_y = WD_BM / g.MX, d_x = WD_BM - d_y * g.MX;
    // Row tables: only the first WD_BM threads walk (one row each); the tile's first / last input pixel are rows 0 and
    // WD_BM - 1 of the table itself, so nothing uniform is walked by every wave (first generation: three walkers in all
    // 12 waves).  The table holds ABSOLUTE input pixels; the readers fold the tile's first pixel into their base.
    WdWalk wr;
    if (tid < WD_BM) wd_walk_init(wr, g, kt0 * WD_BM + tid);
    const uint32_t dy_row_bytes = (uint32_t)g.Cout * 2u;
    auto tabulate = [&](int kt) {
      if (tid < WD_BM) {
        int pin, pout;
        wd_walk_pixels(wr, g, pin, pout);
        s_dyoff[(kt & (NTAB - 1)) * WD_BM + tid] = pout < 0 ? 0u : (uint32_t)pout * dy_row_bytes;   // 0 = zero border
        s_pin[(kt & (NTAB - 1)) * WD_BM + tid] = pin;
        wd_walk_advance(wr, g, d_y, d_x, WD_BM);
      }
    };
    const int NI = (2 * (plane_bytes >> 10) + DBLK + NW - 1) / NW;
    const unsigned char* const xg = reinterpret_cast<const unsigned char*>(x) + (long)ci0 * 2;
    const unsigned char* const dg = reinterpret_cast<const unsigned char*>(dy) + (long)co0 * 2;
    auto dma_issue = [&](int buf, int kt) {
      const int tab = kt & (NTAB - 1);
      const int plo = __builtin_amdgcn_readfirstlane(s_pin[tab * WD_BM]);
      const int phi = __builtin_amdgcn_readfirstlane(s_pin[tab * WD_BM + WD_BM - 1]);
      // 16-row blocks per plane.  Banded patch (band_pitch > 0: dilated convolutions / wide images, where a tile's
      // contiguous pixel span is mostly the rows BETWEEN its three tap rows): one band of band_pitch LDS rows per tap
      // row, band ty holding input pixels plo + ty * band_stride + [0, span + 2 * TXS]
      const int nbb = band_pitch > 0 ? (phi - plo + 2 * TXS + 1 + 15) >> 4 : 0;   // blocks per band
      const int nbp = band_pitch > 0 ? 3 * nbb : (phi + max_tap_off - plo + 1 + 15) >> 4;
      unsigned char* const dX = sX + buf * xb_bytes;
      unsigned char* const dD = sD + buf * DB;
      for (int i = 0; i < NI; ++i) {
        int b = wave + i * NW;
        b = b < 2 * nbp + DBLK ? b : 2 * nbp + DBLK - 1;
        if (b < 2 * nbp) {
          const int pl = b >= nbp ? 1 : 0;
          int j = b - pl * nbp;                    // block of the plane
          int r0 = plo + j * 16;
          if (band_pitch > 0) {
            const int band = (j >= nbb ? 1 : 0) + (j >= 2 * nbb ? 1 : 0);
            const int jb = j - band * nbb;
            r0 = plo + band * band_stride + jb * 16;
            j = band * (band_pitch >> 4) + jb;     // LDS block: band ty starts at row ty * band_pitch
          }
          const unsigned char* src;
          uint32_t vo = lane_x;
          if (r0 + 15 < in_pixels) {
            src = xg + ((long)r0 * g.Cin) * 2 + pl * 64;
          } else {                                  // the end of the tensor: rows past it re-read its last pixel
            const int rb = r0 < in_pixels ? r0 : in_pixels - 1;
            int p = r0 + (lane >> 2);
            p = p < in_pixels ? p : in_pixels - 1;
            src = xg + ((long)rb * g.Cin) * 2 + pl * 64;
            vo = (uint32_t)(p - rb) * (uint32_t)(g.Cin * 2) + (lane & 3) * 16;
          }
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)vo),
                                           (__attribute__((address_space(3))) void*)(dX + pl * plane_bytes + j * 1024),
                                           16, 0, 0);
        } else {
          const int bd = b - 2 * nbp;
          const int row = bd * (COT == 128 ? 4 : 8) + drow_l;
          const uint32_t vo = s_dyoff[tab * WD_BM + row] + lane_d;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dg + (size_t)vo),
                                           (__attribute__((address_space(3))) void*)(dD + bd * 1024), 16, 0, 0);
        }
      }
    };

#pragma unroll
    for (int i = 0; i < NBUF; ++i) tabulate(kt0 + i);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i)
      if (kt0 + i < kt1) dma_issue(i, kt0 + i);

    for (int kt = kt0; kt < kt1; ++kt) {
      const int b = (kt - kt0) % NBUF;
      if (!(abl & 4)) {          // abl 4 (timing only, racy): no wait, no barrier
        if (NBUF == 2) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
          const int ahead = min(NBUF - 2, kt1 - 1 - kt);
          wd_wait_vmcnt(ahead * NI);
        }
        __syncthreads();
      }
      if (kt + NBUF - 1 < kt1 && !(abl & 1)) dma_issue((b + NBUF - 1) % NBUF, kt + NBUF - 1);
      tabulate(kt + NBUF);
      const int* prow = s_pin + (kt & (NTAB - 1)) * WD_BM;
      // read addresses of the tile: input rows for tap (ty, 0) of this wave's plane; dY base per co sub-tile
      const int plo = __builtin_amdgcn_readfirstlane(prow[0]);
      const uint32_t xbase = sXo + b * xb_bytes + wn * plane_bytes + ((band_pitch > 0 ? tg * band_pitch : toff0) - plo) * 64 + tsub;
      uint32_t pb0[NKS], pb1[NKS], ab[CS];
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {         // all table reads in flight, one wait
        pb0[ks] = (uint32_t)prow[ks * 16 + trow];
        pb1[ks] = (uint32_t)prow[ks * 16 + trow + 4];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        pb0[ks] = xbase + (pb0[ks] << 6);
        pb1[ks] = xbase + (pb1[ks] << 6);
        asm volatile("" : "+v"(pb0[ks]), "+v"(pb1[ks]));      // keep base + immediate (no re-association)
      }
#pragma unroll
      for (int c = 0; c < CS; ++c) {
        ab[c] = sDo + b * DB + aoff[c];
        asm volatile("" : "+v"(ab[c]));
      }
      auto step = [&](auto KS) {
        constexpr int ks = decltype(KS)::value;
        bf16x8 a[CS], bfr[3];
#pragma unroll
        for (int c = 0; c < CS; ++c) a[c] = wdp_frag_pair<ks * 16 * DROW, 4 * DROW, ASMRD>(ab[c]);
        bfr[0] = wdp_frag<0, ASMRD>(pb0[ks], pb1[ks]);
        bfr[1] = wdp_frag<64 * TXS, ASMRD>(pb0[ks], pb1[ks]);
        bfr[2] = wdp_frag<128 * TXS, ASMRD>(pb0[ks], pb1[ks]);
        if (ASMRD) {       // LDS returns in order: 4 (2) dY reads, then 2 per tap
          if (CS == 2)
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a[0]), "+v"(a[1]), "+v"(bfr[0]));
          else
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a[0]), "+v"(bfr[0]));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < CS; ++c)
          acc[0][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[c], bfr[0], acc[0][c], 0, 0, 0);
        if (ASMRD) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bfr[1]));
#pragma unroll
        for (int c = 0; c < CS; ++c)
          acc[1][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[c], bfr[1], acc[1][c], 0, 0, 0);
        if (ASMRD) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bfr[2]));
#pragma unroll
        for (int c = 0; c < CS; ++c)
          acc[2][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[c], bfr[2], acc[2][c], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      if (!(abl & 2)) wdp_unroll<NKS>(step);      // abl 2 (timing only): DMA + bookkeeping without the k-steps
    }
  }
  if (abl & 8) return;                   // abl 8 (timing only): no partial stores
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    float* dst = partials + (((long)split * g.ntaps + (tfirst + t)) * g.Cout + co0) * g.Cin + ci0;
#pragma unroll
    for (int c = 0; c < CS; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * (COT / 2) + c * 32 + mfma32_row(r, lane);
        const int col = wn * 32 + l31;
        if (abl & 16) __builtin_nontemporal_store(acc[t][c][r], dst + (long)row * g.Cin + col);   // abl 16: nt stores (results correct)
        else dst[(long)row * g.Cin + col] = acc[t][c][r];
      }
  }
}

// Third form of the loop (iic_debug_wgrad_planar = 3, the default): the planar-patch kernel above with the per-tile work
// taken off the critical path between the tile barrier and the first MFMA (LAB.md R6.8: of a 144-us launch at layer 3,
// 61 us are MFMA time, 26 us the k-steps' own inefficiency, 27 us DMA / compute overlap loss, 30 us fixed):
//   * the read addresses of tile kt + 1 are formed at the END of tile kt (their table reads are issued behind the last
//     k-step's fragment reads and land under its MFMAs), so the first fragment reads issue right after the barrier;
//   * the next tile's DMA is issued behind those first reads (its scalar address work runs under their LDS latency);
//   * SWP: the fragment reads of k-step ks + 1 are issued before the MFMAs of k-step ks (two register sets);
//   * row tables by an incremental walker (adds and selects; the first two forms recompute four products per row).
// All transposing reads are inline asm with hand-counted lgkmcnt waits (LDS operations complete in order, so a wait for
// "at most N outstanding" retires everything but the N youngest whatever else the compiler has in flight).
struct WdpRow {
  int n, y, x, pin;
  uint32_t dyo;
};

template <int OFF0, int OFF1>
__device__ __forceinline__ void wdp_tab_read(uint32_t addr, u32x2& d) {      // two table words (dword offsets)
  asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(d) : "v"(addr), "n"(OFF0), "n"(OFF1));
}
template <int N>
__device__ __forceinline__ void wdp_lgkm_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

template <int COT, int WD_BM, int NBUF, int TXS, bool SWP, int NTAB = WD_NTAB>
__global__ __launch_bounds__(WD_THREADS) void conv_wgrad_pl2_kernel(
    const iic_conv_geom g, const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
    float* __restrict__ partials, int nsplit, int num_ktiles, int plane_bytes, int max_tap_off, int abl,
    int band_pitch, int band_stride) {
  constexpr int CS = COT / 64;
  constexpr int NW = WD_THREADS / 64;
  constexpr int DROW = COT * 2;
  constexpr int DB = WD_BM * DROW;
  constexpr int DBLK = DB / 1024;
  constexpr int NKS = WD_BM / 16;
  static_assert((NTAB & (NTAB - 1)) == 0 && NTAB >= NBUF + 2, "table ring: a power of two, at least NBUF + 2 tiles");
  constexpr int NRD = 2 * CS + 6;               // transposing reads per k-step
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  typedef unsigned char __attribute__((address_space(3))) * lds_u8_ptr;
  const int xb_bytes = 2 * plane_bytes;
  unsigned char* const sX = smem_raw;
  unsigned char* const sD = smem_raw + NBUF * xb_bytes;
  uint32_t* const s_dyoff = reinterpret_cast<uint32_t*>(sD + NBUF * DB);
  int* const s_pin = reinterpret_cast<int*>(s_dyoff + NTAB * WD_BM);
  const uint32_t sXo = (uint32_t)(uintptr_t)(lds_u8_ptr)sX, sDo = (uint32_t)(uintptr_t)(lds_u8_ptr)sD;
  const uint32_t sPo = (uint32_t)(uintptr_t)(lds_u8_ptr)reinterpret_cast<unsigned char*>(s_pin);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tg = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
  const int l31 = lane & 31;
  const int q = lane >> 4, i16 = lane & 15;
  const int ncit = g.Cin >> 6;
  int tile, split;
  wdp_xcd_map(tile, split);
  const int cot = tile / ncit, cit = tile - cot * ncit;
  const int co0 = cot * COT, ci0 = cit * 64;
  const int per = (num_ktiles + nsplit - 1) / nsplit;
  const int kt0 = split * per;
  const int kt1 = min(num_ktiles, kt0 + per);
  const int in_pixels = g.N * g.in_Hp * g.in_Wp;
  const int tfirst = tg * 3;
  const int toff0 = __builtin_amdgcn_readfirstlane(g.tap_off[tfirst]);

  const int trow = 8 * (q >> 1) + (i16 >> 2);
  const int tsub = (2 * (q & 1) + ((i16 & 3) >> 1)) * 16 + (i16 & 1) * 8;
  uint32_t aoff[CS];
#pragma unroll
  for (int c = 0; c < CS; ++c) {
    const int unit = (COT == 128) ? ((wm * 2 + c) ^ (trow & 3)) : (wm ^ ((trow >> 1) & 1));
    aoff[c] = trow * DROW + unit * 64 + tsub;
  }
  const uint32_t lane_x = (uint32_t)(lane >> 2) * (uint32_t)(g.Cin * 2) + (lane & 3) * 16;
  const int drow_l = (COT == 128) ? (lane >> 4) : (lane >> 3);
  const uint32_t lane_d = (COT == 128) ? (uint32_t)(((lane & 15) ^ ((drow_l & 3) << 2)) * 16)
                                       : (uint32_t)(((lane & 7) ^ (((drow_l >> 1) & 1) << 2)) * 16);

  f32x16 acc[3][CS];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int c = 0; c < CS; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][c][r] = 0.f;

  if (kt0 < kt1) {
    const bool padded = wd_padded(g);
    const uint32_t dyrb = (uint32_t)g.Cout * 2u;
    // incremental walker constants (dense numbering): one K-tile further = d_y image rows + d_x pixels
    const int d_y = WD_BM / g.MX, d_x = WD_BM - d_y * g.MX;
    const int pin_step = d_y * g.sy * g.in_Wp + d_x * g.sx, pin_xw = g.sy * g.in_Wp - g.MX * g.sx,
              pin_yw = (g.in_Hp - g.MY * g.sy) * g.in_Wp;
    const uint32_t dyo_step = (uint32_t)(d_y * g.ty * g.out_Wp + d_x * g.tx) * dyrb,
                   dyo_xw = (uint32_t)(g.ty * g.out_Wp - g.MX * g.tx) * dyrb,
                   dyo_yw = (uint32_t)((g.out_Hp - g.MY * g.ty) * g.out_Wp) * dyrb;
    const int pin_last = ((g.N - 1) * g.in_Hp + (g.MY - 1) * g.sy + g.oy) * g.in_Wp + (g.MX - 1) * g.sx + g.ox;
    WdpRow wr = {0, 0, 0, 0, 0u};
    if (tid < WD_BM) {
      WdWalk w;
      wd_walk_init(w, g, kt0 * WD_BM + tid);
      wr.n = w.n; wr.y = w.y; wr.x = w.x;
      if (!padded) {
        wr.pin = (w.n * g.in_Hp + w.y * g.sy + g.oy) * g.in_Wp + w.x * g.sx + g.ox;
        wr.dyo = (uint32_t)((w.n * g.out_Hp + w.y * g.ty + g.py) * g.out_Wp + w.x * g.tx + g.px) * dyrb;
      }
    }
    auto tabulate = [&](int kt) {
      if (tid < WD_BM) {
        int pin;
        uint32_t dyo;
        if (padded) {                      // (large images: one division per row and tile)
          WdWalk w = {wr.n, wr.y, 0};
          int pout;
          wd_walk_pixels(w, g, pin, pout);
          dyo = pout < 0 ? 0u : (uint32_t)pout * dyrb;
          wd_walk_advance(w, g, 0, 0, WD_BM);
          wr.n = w.n; wr.y = w.y;
        } else {
          const bool valid = wr.n < g.N;
          pin = valid ? wr.pin : pin_last;
          dyo = valid ? wr.dyo : 0u;       // 0 = pixel 0 of the PT tensor: zero border
          wr.x += d_x; wr.y += d_y; wr.pin += pin_step; wr.dyo += dyo_step;
          if (wr.x >= g.MX) { wr.x -= g.MX; ++wr.y; wr.pin += pin_xw; wr.dyo += dyo_xw; }
          while (wr.y >= g.MY) { wr.y -= g.MY; ++wr.n; wr.pin += pin_yw; wr.dyo += dyo_yw; }
        }
        s_dyoff[(kt & (NTAB - 1)) * WD_BM + tid] = dyo;
        s_pin[(kt & (NTAB - 1)) * WD_BM + tid] = pin;
      }
    };
    const int NI = (2 * (plane_bytes >> 10) + DBLK + NW - 1) / NW;
    const unsigned char* const xg = reinterpret_cast<const unsigned char*>(x) + (long)ci0 * 2;
    const unsigned char* const dg = reinterpret_cast<const unsigned char*>(dy) + (long)co0 * 2;
    auto dma_issue = [&](int buf, int kt) {
      const int tab = kt & (NTAB - 1);
      const int plo = __builtin_amdgcn_readfirstlane(s_pin[tab * WD_BM]);
      const int phi = __builtin_amdgcn_readfirstlane(s_pin[tab * WD_BM + WD_BM - 1]);
      const int nbb = band_pitch > 0 ? (phi - plo + 2 * TXS + 1 + 15) >> 4 : 0;   // banded patch: see conv_wgrad_pl_kernel
      const int nbp = band_pitch > 0 ? 3 * nbb : (phi + max_tap_off - plo + 1 + 15) >> 4;
      unsigned char* const dX = sX + buf * xb_bytes;
      unsigned char* const dD = sD + buf * DB;
      for (int i = 0; i < NI; ++i) {
        int b = wave + i * NW;
        b = b < 2 * nbp + DBLK ? b : 2 * nbp + DBLK - 1;
        if (b < 2 * nbp) {
          const int pl = b >= nbp ? 1 : 0;
          int j = b - pl * nbp;
          int r0 = plo + j * 16;
          if (band_pitch > 0) {
            const int band = (j >= nbb ? 1 : 0) + (j >= 2 * nbb ? 1 : 0);
            const int jb = j - band * nbb;
            r0 = plo + band * band_stride + jb * 16;
            j = band * (band_pitch >> 4) + jb;
          }
          const unsigned char* src;
          uint32_t vo = lane_x;
          if (r0 + 15 < in_pixels) {
            src = xg + ((long)r0 * g.Cin) * 2 + pl * 64;
          } else {
            const int rb = r0 < in_pixels ? r0 : in_pixels - 1;
            int p = r0 + (lane >> 2);
            p = p < in_pixels ? p : in_pixels - 1;
            src = xg + ((long)rb * g.Cin) * 2 + pl * 64;
            vo = (uint32_t)(p - rb) * (uint32_t)(g.Cin * 2) + (lane & 3) * 16;
          }
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)vo),
                                           (__attribute__((address_space(3))) void*)(dX + pl * plane_bytes + j * 1024),
                                           16, 0, 0);
        } else {
          const int bd = b - 2 * nbp;
          const int row = bd * (COT == 128 ? 4 : 8) + drow_l;
          const uint32_t vo = s_dyoff[tab * WD_BM + row] + lane_d;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dg + (size_t)vo),
                                           (__attribute__((address_space(3))) void*)(dD + bd * 1024), 16, 0, 0);
        }
      }
    };

#pragma unroll
    for (int i = 0; i < NBUF; ++i) tabulate(kt0 + i);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i)
      if (kt0 + i < kt1) dma_issue(i, kt0 + i);

    // read addresses of a tile: pb[ks] = (row trow, row trow + 4) of k-step ks in this wave's plane, tap (ty, 0)
    u32x2 pb[NKS];
    uint32_t ab[CS];
    const uint32_t tab_lane = sPo + trow * 4;
    auto prep_issue = [&](int kt) {                 // NKS table reads (inline asm: counted by the callers' waits)
      const uint32_t ta = tab_lane + (kt & (NTAB - 1)) * (WD_BM * 4);
      auto rd = [&](auto KS) {
        constexpr int ks = decltype(KS)::value;
        wdp_tab_read<ks * 16, ks * 16 + 4>(ta, pb[ks]);
      };
      wdp_unroll<NKS>(rd);
    };
    auto prep_finish = [&](int kt, int buf) {       // waits for the table reads, then forms the addresses
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pb[0]));
#pragma unroll
      for (int ks = 1; ks < NKS; ++ks) asm volatile("" : "+v"(pb[ks]));
      const int plo = __builtin_amdgcn_readfirstlane(s_pin[(kt & (NTAB - 1)) * WD_BM]);
      const uint32_t xbase = sXo + buf * xb_bytes + wn * plane_bytes + ((band_pitch > 0 ? tg * band_pitch : toff0) - plo) * 64 + tsub;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        pb[ks][0] = xbase + (pb[ks][0] << 6);
        pb[ks][1] = xbase + (pb[ks][1] << 6);
        asm volatile("" : "+v"(pb[ks]));
      }
#pragma unroll
      for (int c = 0; c < CS; ++c) {
        ab[c] = sDo + buf * DB + aoff[c];
        asm volatile("" : "+v"(ab[c]));
      }
    };
    bf16x8 fa[SWP ? 2 : 1][CS], fb[SWP ? 2 : 1][3];
    auto reads = [&](auto KS) {                     // the NRD transposing reads of k-step ks, dY first, then tap by tap
      constexpr int ks = decltype(KS)::value;
      constexpr int s = SWP ? (ks & 1) : 0;
#pragma unroll
      for (int c = 0; c < CS; ++c) fa[s][c] = wdp_frag_pair<ks * 16 * DROW, 4 * DROW, true>(ab[c]);
      fb[s][0] = wdp_frag<0, true>(pb[ks][0], pb[ks][1]);
      fb[s][1] = wdp_frag<64 * TXS, true>(pb[ks][0], pb[ks][1]);
      fb[s][2] = wdp_frag<128 * TXS, true>(pb[ks][0], pb[ks][1]);
    };
    auto mfmas = [&](auto KS, auto YOUNGER) {       // YOUNGER: LDS operations issued after this k-step's reads
      constexpr int ks = decltype(KS)::value;
      constexpr int s = SWP ? (ks & 1) : 0;
      constexpr int Y = decltype(YOUNGER)::value;
      constexpr int W0 = Y + 4 > 15 ? 15 : Y + 4, W1 = Y + 2 > 15 ? 15 : Y + 2, W2 = Y > 15 ? 15 : Y;
      if (CS == 2) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(fa[s][0]), "+v"(fa[s][1]), "+v"(fb[s][0]) : "n"(W0));
      else asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fa[s][0]), "+v"(fb[s][0]) : "n"(W0));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < CS; ++c)
        acc[0][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][c], fb[s][0], acc[0][c], 0, 0, 0);
      asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fb[s][1]) : "n"(W1));
#pragma unroll
      for (int c = 0; c < CS; ++c)
        acc[1][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][c], fb[s][1], acc[1][c], 0, 0, 0);
      asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fb[s][2]) : "n"(W2));
#pragma unroll
      for (int c = 0; c < CS; ++c)
        acc[2][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][c], fb[s][2], acc[2][c], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    };

    prep_issue(kt0);
    prep_finish(kt0, 0);
    for (int kt = kt0; kt < kt1; ++kt) {
      const int b = (kt - kt0) % NBUF;
      if (!(abl & 4)) {
        if (NBUF == 2) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
          const int ahead = min(NBUF - 2, kt1 - 1 - kt);
          wd_wait_vmcnt(ahead * NI);
        }
        __syncthreads();
      }
      if (!(abl & 2)) reads(std::integral_constant<int, 0>{});
      __builtin_amdgcn_sched_barrier(0);
      if (kt + NBUF - 1 < kt1 && !(abl & 1)) dma_issue((b + NBUF - 1) % NBUF, kt + NBUF - 1);
      tabulate(kt + NBUF);
      __builtin_amdgcn_sched_barrier(0);
      if (!(abl & 2)) {
        auto step = [&](auto KS) {
          constexpr int ks = decltype(KS)::value;
          if constexpr (SWP) {
            if constexpr (ks + 1 < NKS) {
              reads(std::integral_constant<int, ks + 1>{});
              mfmas(KS, std::integral_constant<int, NRD>{});
            } else {
              prep_issue(kt + 1);
              mfmas(KS, std::integral_constant<int, NKS>{});
            }
          } else {
            if constexpr (ks > 0) reads(KS);
            if constexpr (ks + 1 < NKS) {
              mfmas(KS, std::integral_constant<int, 0>{});
            } else {
              prep_issue(kt + 1);
              mfmas(KS, std::integral_constant<int, NKS>{});
            }
          }
        };
        wdp_unroll<NKS>(step);
      } else {
        prep_issue(kt + 1);
      }
      prep_finish(kt + 1, (b + 1) % NBUF);
    }
  }
  if (abl & 8) return;                   // abl 8 (timing only): no partial stores
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    float* dst = partials + (((long)split * g.ntaps + (tfirst + t)) * g.Cout + co0) * g.Cin + ci0;
#pragma unroll
    for (int c = 0; c < CS; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * (COT / 2) + c * 32 + mfma32_row(r, lane);
        const int col = wn * 32 + l31;
        if (abl & 16) __builtin_nontemporal_store(acc[t][c][r], dst + (long)row * g.Cin + col);   // abl 16: nt stores (results correct)
        else dst[(long)row * g.Cin + col] = acc[t][c][r];
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Block-tiled form for large images (round 6, LAB.md R6.13): a K-tile is a 2-D block of bw x bh <= 128 OUTPUT pixels of
// one image instead of 128 consecutive pixels of the row-major numbering.  The input patch of a tile is then the
// (bw + 2 dx) x (bh + 2 dy) pixel sub-image under the block -- 1.5-1.9 x the pixels used instead of the 3.4-7.6 x of a
// contiguous span (banded: 3.4-3.75 x) on 100-200-pixel image rows, dilated taps included -- and EVERY address of the loop
// is a tile base + a per-lane constant: patch row of output (ky, kx), tap (ty, tx) = (ky + ty dy) PW + kx + tx dx, DMA
// source of patch row (py, px) = tile pixel + py in_Wp + px, dY row of (ky, kx) = tile pixel + ky out_Wp + kx.  No row
// tables, no walkers; per tile: a uniform tile decode, the DMA issue and 16 adds for the buffer switch.  Rows of a block
// past the image edge (or past bw x bh) take their dY from pixel 0 of the PT tensor (zero border) and read whatever input
// pixel lies there (finite x 0).  Same work split per wave, same partial-sum layout as the planar kernels; the summation
// ORDER over pixels differs from theirs (so do the last bits).  Stride-1 3 x 3 taps on a regular grid, COT = 128 only.
#define WDB_MAXNI 6
struct wdb_args {
  int bw, bh, nbx, nby;           // block size, blocks per image row / column
  int PW, NPR, drow;              // patch width, patch pixels, tap-row distance in image rows
  int plane_bytes, num_tiles;
};

template <int TXS>
__global__ __launch_bounds__(WD_THREADS) void conv_wgrad_b2d_kernel(
    const iic_conv_geom g, const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
    float* __restrict__ partials, int nsplit, const wdb_args A, int abl) {
  constexpr int COT = 128, CS = 2, NW = WD_THREADS / 64, DROW = COT * 2, WD_BM = 128;
  constexpr int DB = WD_BM * DROW, DBLK = DB / 1024, NKS = WD_BM / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  typedef unsigned char __attribute__((address_space(3))) * lds_u8_ptr;
  const int plane_bytes = A.plane_bytes, xb_bytes = 2 * plane_bytes;
  unsigned char* const sX = smem_raw;                        // [2][2 planes][plane_bytes]
  unsigned char* const sD = smem_raw + 2 * xb_bytes;         // [2][DB]
  const uint32_t sXo = (uint32_t)(uintptr_t)(lds_u8_ptr)sX, sDo = (uint32_t)(uintptr_t)(lds_u8_ptr)sD;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tg = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
  const int l31 = lane & 31;
  const int q = lane >> 4, i16 = lane & 15;
  const int ncit = g.Cin >> 6;
  int tile, split;
  wdp_xcd_map(tile, split);
  const int cot = tile / ncit, cit = tile - cot * ncit;
  const int co0 = cot * COT, ci0 = cit * 64;
  const int per = (A.num_tiles + nsplit - 1) / nsplit;
  const int kt0 = split * per;
  const int kt1 = min(A.num_tiles, kt0 + per);
  const int in_pixels = g.N * g.in_Hp * g.in_Wp;
  const int tfirst = tg * 3;
  const int bw = A.bw, bh = A.bh, PW = A.PW, nrows = bw * bh;

  const int trow = 8 * (q >> 1) + (i16 >> 2);
  const int tsub = (2 * (q & 1) + ((i16 & 3) >> 1)) * 16 + (i16 & 1) * 8;
  uint32_t aoff[CS];
#pragma unroll
  for (int c = 0; c < CS; ++c) aoff[c] = trow * DROW + (((wm * 2 + c) ^ (trow & 3)) * 64) + tsub;
  // this lane's K rows -> patch rows (relative to the buffer): 64 * ((ky + ty drow) PW + kx) in the wave's plane
  // (absolute LDS addresses for buffer 0; the loop moves them to the other buffer and back in place)
  uint32_t pb0[NKS], pb1[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      int k = ks * 16 + trow + 4 * rd;
      k = k < nrows ? k : nrows - 1;              // rows past the block: dY is zero there, any patch row will do
      const int ky = k / bw, kx = k - ky * bw;
      const uint32_t v = sXo + (uint32_t)(((ky + tg * A.drow) * PW + kx) * 64) + wn * plane_bytes + tsub;
      if (rd) pb1[ks] = v; else pb0[ks] = v;
    }
  }
  // DMA roles of this wave's instruction slots (static): patch block (plane, 16 patch pixels) or dY block (4 rows)
  const int nbp = (A.NPR + 15) >> 4;
  const int total = 2 * nbp + DBLK;
  const int NI = (total + NW - 1) / NW;             // <= WDB_MAXNI (host)
  const uint32_t dyrb = (uint32_t)g.Cout * 2u;
  const int drow_l = lane >> 4;
  const uint32_t lane_d = (uint32_t)(((lane & 15) ^ ((drow_l & 3) << 2)) * 16);
  // byte offset from the tile's first input pixel / first dY pixel (0xffffffff: a dY row past the block, never valid);
  // the rare paths (a tile that touches the end of the tensor / the image edge) re-derive (py, px) / (ky, kx)
  uint32_t voff[WDB_MAXNI];
  // (`launder` keeps the rare paths' lane arithmetic out of the loop-invariant registers: the kernel sits at the
  //  3-waves-per-SIMD register limit)
  auto launder = [](int v) { asm volatile("" : "+v"(v)); return v; };
  auto patch_pixel = [&](int j) {                  // pixel offset of this lane's patch row in block j
    int r = j * 16 + (launder(lane) >> 2);
    r = r < A.NPR ? r : A.NPR - 1;
    const int py = r / PW;
    return py * g.in_Wp + (r - py * PW);
  };
#pragma unroll
  for (int i = 0; i < WDB_MAXNI; ++i) {
    int b = wave + i * NW;
    b = b < total ? b : total - 1;
    if (b < 2 * nbp) {
      const int pl = b >= nbp ? 1 : 0, j = b - pl * nbp;
      voff[i] = (uint32_t)patch_pixel(j) * (uint32_t)(g.Cin * 2) + (lane & 3) * 16 + pl * 64;
    } else {
      const int k = (b - 2 * nbp) * 4 + drow_l;
      const int ky = k / bw, kx = k - ky * bw;
      voff[i] = k < nrows ? (uint32_t)(ky * g.out_Wp + kx) * dyrb + lane_d : 0xffffffffu;
    }
  }
  const unsigned char* const xg = reinterpret_cast<const unsigned char*>(x) + (long)ci0 * 2;
  const unsigned char* const dg = reinterpret_cast<const unsigned char*>(dy) + (long)co0 * 2;
  const int tpi = A.nbx * A.nby;                   // tiles per image

  f32x16 acc[3][CS];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int c = 0; c < CS; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][c][r] = 0.f;

  if (kt0 < kt1) {
    // uniform tile cursor (image, block row, block column) of the NEXT tile to fetch
    int cn = kt0 / tpi, crem = kt0 - cn * tpi;
    int cby = crem / A.nbx, cbx = crem - cby * A.nbx;
    auto dma_issue = [&](int buf) {                // fetches the cursor's tile, then advances the cursor
      const int y0 = cby * bh, x0 = cbx * bw;
      const int pin0 = (cn * g.in_Hp + y0 + g.oy) * g.in_Wp + x0 + g.ox;
      const int pout0 = (cn * g.out_Hp + y0 + g.py) * g.out_Wp + x0 + g.px;
      const bool inside = pin0 + (A.NPR / PW) * g.in_Wp + PW < in_pixels;     // (whole patch before the tensor's end)
      const int bhv = min(bh, g.MY - y0), bwv = min(bw, g.MX - x0);
      const bool full = bhv == bh && bwv == bw;
      const uint32_t dbase = (uint32_t)pout0 * dyrb;
      unsigned char* const dX = sX + buf * xb_bytes;
      unsigned char* const dD = sD + buf * DB;
#pragma unroll
      for (int i = 0; i < WDB_MAXNI; ++i) {
        if (i < NI) {
          int b = wave + i * NW;
          b = b < total ? b : total - 1;
          if (b < 2 * nbp) {
            const int pl = b >= nbp ? 1 : 0, j = b - pl * nbp;
            const unsigned char* src = xg + ((long)pin0 * g.Cin) * 2;
            uint32_t vo = voff[i];
            if (!inside) {                         // the last image's last blocks: pixels past the tensor re-read its last one
              int pix = pin0 + patch_pixel(j);
              pix = pix < in_pixels ? pix : in_pixels - 1;
              src = xg;
              vo = (uint32_t)pix * (uint32_t)(g.Cin * 2) + (launder(lane) & 3) * 16 + pl * 64;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)vo),
                                             (__attribute__((address_space(3))) void*)(dX + pl * plane_bytes + j * 1024),
                                             16, 0, 0);
          } else {
            const int bd = b - 2 * nbp;
            bool valid = voff[i] != 0xffffffffu;
            if (!full) {                           // a block at the image's right / bottom edge
              const int k = bd * 4 + (launder(lane) >> 4);
              const int ky = k / bw, kx = k - ky * bw;
              valid = valid && ky < bhv && kx < bwv;
            }
            const uint32_t vo = valid ? dbase + voff[i] : lane_d;      // pixel 0 of the PT tensor: zero border
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dg + (size_t)vo),
                                             (__attribute__((address_space(3))) void*)(dD + bd * 1024), 16, 0, 0);
          }
        }
      }
      if (++cbx == A.nbx) { cbx = 0; if (++cby == A.nby) { cby = 0; ++cn; } }
    };
    dma_issue(0);
    for (int kt = kt0; kt < kt1; ++kt) {
      const int b = (kt - kt0) & 1;
      if (!(abl & 4)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      if (kt + 1 < kt1 && !(abl & 1)) dma_issue(b ^ 1);
      uint32_t ab[CS];
      if (kt > kt0) {                              // the read addresses move to the other patch buffer
        const uint32_t d = b ? (uint32_t)xb_bytes : (uint32_t)(-xb_bytes);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          pb0[ks] += d;
          pb1[ks] += d;
        }
      }
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(pb0[ks]), "+v"(pb1[ks]));
#pragma unroll
      for (int c = 0; c < CS; ++c) {
        ab[c] = sDo + b * DB + aoff[c];
        asm volatile("" : "+v"(ab[c]));
      }
      auto step = [&](auto KS) {
        constexpr int ks = decltype(KS)::value;
        bf16x8 a[CS], bfr[3];
#pragma unroll
        for (int c = 0; c < CS; ++c) a[c] = wdp_frag_pair<ks * 16 * DROW, 4 * DROW, true>(ab[c]);
        bfr[0] = wdp_frag<0, true>(pb0[ks], pb1[ks]);
        bfr[1] = wdp_frag<64 * TXS, true>(pb0[ks], pb1[ks]);
        bfr[2] = wdp_frag<128 * TXS, true>(pb0[ks], pb1[ks]);
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a[0]), "+v"(a[1]), "+v"(bfr[0]));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < CS; ++c)
          acc[0][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[c], bfr[0], acc[0][c], 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bfr[1]));
#pragma unroll
        for (int c = 0; c < CS; ++c)
          acc[1][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[c], bfr[1], acc[1][c], 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bfr[2]));
#pragma unroll
        for (int c = 0; c < CS; ++c)
          acc[2][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[c], bfr[2], acc[2][c], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      if (!(abl & 2)) wdp_unroll<NKS>(step);
    }
  }
  if (abl & 8) return;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    float* dst = partials + (((long)split * g.ntaps + (tfirst + t)) * g.Cout + co0) * g.Cin + ci0;
#pragma unroll
    for (int c = 0; c < CS; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * (COT / 2) + c * 32 + mfma32_row(r, lane);
        const int col = wn * 32 + l31;
        dst[(long)row * g.Cin + col] = acc[t][c][r];
      }
  }
}

static long wdp_plane_bytes(int np) { return (((long)np * 64) + 1023) & ~1023L; }
static long wdp_lds(int np, int cot, int bmk, int nbuf, int ntab) {
  return nbuf * (2 * wdp_plane_bytes(np) + (long)bmk * cot * 2) + (long)ntab * bmk * 8;
}

static long wd_xb_bytes(int np) { return (((long)np * 128) + 1023) & ~1023L; }
static long wd_lds(int np, int cot, int bmk, int nbuf) {
  return nbuf * (wd_xb_bytes(np) + (long)bmk * cot * 2) + WD_TAB_BYTES(bmk) + 64;
}

// 1: transposing reads as inline asm (see wd_tr_issue).  Measured (tools/conv_perf.py, same process):
// the compiler's vmcnt(0) disappears from the K-tile loop, the time does not change (layer1 187 ->
// 181 us, layer2 144 -> 143, layer3 143 -> 145, layer4 169 -> 172): the DMA wait was not what parks
// the waves.  Default stays on the builtin; the switch is kept for the next experiments.
IIC_SWITCH(g_wd_asm, 0, iic_debug_wgrad_asm)
// 1: "fat wave" variant -- 4 waves (one per SIMD) with all 9 taps each, 22 transposing reads per 18
// MFMAs instead of 10 per 6, the next pipeline unit's fragments read under the current MFMAs.
// Measured (tools/wgrad_ab.sh, per launch incl. the reduce pass): layer2-4 within 2 % of the 12-wave
// kernel, layer1 28 % slower -- 27 % less LDS-read traffic buys nothing, i.e. the kernel is not
// LDS-read-bound as round 1 assumed.  Default 0 (12 waves).
// timing ablation (WRONG results): 1 = no DMA after the prologue (compute-only time of the K loop); planar kernel also
// 2 = no k-steps (DMA + bookkeeping only), 4 = no per-tile wait / barrier
IIC_SWITCH(g_wd_ablate, 0, iic_debug_wgrad_ablate)
IIC_SWITCH(g_wd_prefetch, 0, iic_debug_wgrad_prefetch)
// 1: 12-wave kernel with the next k-step's fragments read under the current k-step's MFMAs (template SWP).  Measured
// (tools/wgrad_swp_ab.py, profiles/r06_wgrad_swp_ab.txt): bit-identical, 1.01-1.03 x per launch in isolation (layer 1
// 215.5 -> 209.1 us, layer 3 159.1 -> 155.4), and 36.31 / 36.26 -> 36.38 / 36.34 ms per step interleaved on one box: the
// wave-level lgkmcnt park was not what sets the step.  Default 0; instantiated in the instrumented library only.
IIC_SWITCH(g_wd_swp, 0, iic_debug_wgrad_swp)
// K-loop form where the planar layout fits: 0 = first generation; 1 = planar patch, builtin transposing reads; 2 = planar,
// inline-asm reads (no compiler-inserted vmcnt(0) between the next tile's DMA and this tile's reads); 3 = pipelined form
// (conv_wgrad_pl2_kernel); 4 = pipelined + software-pipelined k-steps; 5 (default) = 4 for 64-cout tiles (115 registers: no
// spills; layer 1: 172 -> 164 us), 2 for 128-cout tiles (form 4 spills there and measured 6 % slower, form 3 equal to form 2:
// tools/wgrad_pl_ab.py, profiles/r06_wgrad_planar_ab.txt).
IIC_SWITCH(g_wd_planar, 5, iic_debug_wgrad_planar)
IIC_SWITCH(g_wd_enabled, 1, iic_debug_enable_wgrad_dma)     // 0: register-staged kernel, 1: DMA kernel, 3: force 64-pixel K-tiles

// K-tile size / ring depth.  Measured at the ClusterNet5g shapes (tools/conv_perf.py): 128-pixel
// tiles with 2 buffers beat 64-pixel tiles with 3-4 buffers by 10-20 % -- the halo makes a 64-row
// patch 44 % larger per row (NP64/64 vs NP/128) and the tile barrier comes twice as often, which
// costs more than the deeper prefetch hides.  The 64-pixel ring serves geometries whose 128-row
// patch does not fit twice in LDS (wide images).  g_wd_enabled = 3 forces it (tests).
static int wd_config(const iic_conv_geom* g, int* bmk, int* nbuf) {
  if (g->ntaps != 9 || g->Cin % 64 != 0 || g->Cout % 64 != 0 || g->NP <= 0 || g->NP > 65535) return 0;
  if (g->MP > 0 && g->MP % 128 != 0) return 0;         // padded numbering: K-tiles must not straddle images
  const int cot = (g->Cout % 128 == 0) ? 128 : 64;
  if (g_wd_enabled != 3 && wd_lds(g->NP, cot, 128, 2) <= 160 * 1024) {
    *bmk = 128;
    *nbuf = 2;
    return 1;
  }
  // padded row numbering = large images (SegmentationNet10a at 200 x 200): measured with the 64-pixel ring
  // (the only one that fits there) 675 / 615 / 1180 us for c2 / c3 / c4 against 644 / 593 / 1127 us on the
  // register-staged kernel -- the walkers support the numbering (tests), the dispatcher keeps the faster kernel
  if (g->MP > 0 && g->MP != g->MY * g->MX && g_wd_enabled != 3) return 0;
  if (g->NP64 > 0) {
    // (2 buffers: the stride-2 layers, whose 64-row patch spans 330-440 input pixels)
    for (int nb = 4; nb >= 2; --nb)
      if (wd_lds(g->NP64, cot, 64, nb) <= 160 * 1024) {
        *bmk = 64;
        *nbuf = nb;
        return 1;
      }
  }
  return 0;
}

// Planar-patch kernels: K-tile size / ring depth / table ring.  Chosen on their own terms (a layout that fits these
// kernels need not fit the first generation's and vice versa): 128-pixel tiles with 2 buffers where they fit (if need be
// with the 4-tile table ring: SegmentationNet10a c3 / c4 fit 160 KB to the byte that way), else the 64-pixel ring.
// A wave's three taps must be one tap row with a fixed x step (1, or 2 = dilation 2); dY row offsets must fit 32 bits.
// Padded row numbering (large images): allowed (g_wd_planar_padded) -- LAB.md R6.12 / profiles/r06_wgrad_seg_ab.txt have
// the per-layer A/B against the register-staged kernel that used to keep these layers (Potsdam c3 / c4: 1.28-1.30 x).
IIC_SWITCH(g_wd_planar_padded, 1, iic_debug_wgrad_planar_padded)
static int wdp_txs(const iic_conv_geom* g) {
  if (g->ntaps != 9) return 0;
  int txs = g->tap_off[1] - g->tap_off[0];
  for (int ty = 0; ty < 3; ++ty)
    for (int tx = 0; tx < 3; ++tx)
      if (g->tap_off[3 * ty + tx] != g->tap_off[3 * ty] + tx * txs) txs = 0;
  return (txs == 1 || txs == 2) ? txs : 0;
}
// band (out): LDS rows per band of the banded patch, 0 = contiguous patch.  Banded: the three tap rows of a tile each
// get their own band of span + 2 * txs + 1 input rows (rounded to the DMA's 16-row blocks) instead of one contiguous
// span of span + max tap offset + 1 rows -- for dilated convolutions and wide images the rows between the tap rows are
// most of that span (SegmentationNet10a c5 at Potsdam: 500 rows per 64-pixel tile contiguous, 3 x 80 banded).  Only
// where the bands do not overlap (tap-row distance >= band) and only for the 128-cout kernel.
IIC_SWITCH(g_wd_banded, 1, iic_debug_wgrad_banded)
static int wdp_band_rows(const iic_conv_geom* g, int np, int txs) {
  int mto = 0;
  for (int i = 0; i < g->ntaps; ++i) mto = g->tap_off[i] > mto ? g->tap_off[i] : mto;
  const int span = np - mto - 1;                    // last - first input pixel (tap 0) of a tile, at most
  return (span + 2 * txs + 1 + 15) & ~15;
}
static int wdp_config(const iic_conv_geom* g, int* bmk, int* nbuf, int* ntab, int* band) {
  *band = 0;
  if (!g_wd_planar || g->ntaps != 9 || g->Cin % 64 != 0 || g->Cout % 64 != 0 || g->NP <= 0 || g->NP > 65535) return 0;
  const int txs = wdp_txs(g);
  if (!txs) return 0;
  if ((long)g->N * g->out_Hp * g->out_Wp * g->Cout * 2 >= (1L << 32)) return 0;
  const bool padded = g->MP > 0 && g->MP != g->MY * g->MX;
  if (padded && !g_wd_planar_padded && g_wd_enabled != 3) return 0;
  const int cot = (g->Cout % 128 == 0) ? 128 : 64;
  const long lim = 160 * 1024;
  if (g_wd_enabled != 3 && (g->MP <= 0 || g->MP % 128 == 0)) {
    for (int nt = 8; nt >= 4; nt >>= 1)
      if (wdp_lds(g->NP, cot, 128, 2, nt) <= lim) { *bmk = 128; *nbuf = 2; *ntab = nt; return 1; }
  }
  if (g->NP64 > 0 && (g->MP <= 0 || g->MP % 64 == 0)) {
    for (int nb = 4; nb >= 3; --nb)
      if (wdp_lds(g->NP64, cot, 64, nb, 8) <= lim) { *bmk = 64; *nbuf = nb; *ntab = 8; return 1; }
    // banded 64-pixel ring: 3 bands of (64 + wraps + 2 txs + 1) rows instead of the contiguous span
    const int bstride = g->tap_off[3] - g->tap_off[0];
    if (g_wd_banded && cot == 128 && bstride > 0 && g->tap_off[6] - g->tap_off[3] == bstride) {
      const int bp = wdp_band_rows(g, g->NP64, txs);
      if (bp <= bstride && 3 * bp < g->NP64) {
        for (int nb = 4; nb >= 2; --nb)
          if (wdp_lds(3 * bp, cot, 64, nb, 8) <= lim) { *bmk = 64; *nbuf = nb; *ntab = 8; *band = bp; return 1; }
      }
    }
    // two buffers of 64-pixel tiles: the patch is mostly halo there (NP64 / 64 = 5-8 rows fetched per row used) and the
    // planar DMA moves it in 64-byte pieces -- measured (profiles/r06_wgrad_seg_ab.txt): 1.12-1.22 x the previous kernel
    // up to 284 halo rows (COCO-Stuff c2 / c5 / c6), 1.04 x at 414 (Potsdam c2), 0.98 x at 420 (Potsdam c6) -- where the
    // banded form above does not apply
    int mto = 0;
    for (int i = 0; i < g->ntaps; ++i) mto = g->tap_off[i] > mto ? g->tap_off[i] : mto;
    if (mto <= 400 || g_wd_enabled == 3) {          // (halo rows, not patch rows: the stride-2 layers' 64-row span is long by itself)
      if (wdp_lds(g->NP64, cot, 64, 2, 8) <= lim) { *bmk = 64; *nbuf = 2; *ntab = 8; return 1; }
      if (wdp_lds(g->NP64, cot, 64, 2, 4) <= lim) { *bmk = 64; *nbuf = 2; *ntab = 4; return 1; }
    }
  }
  return 0;
}

// Block-tiled kernel (conv_wgrad_b2d_kernel): block shape by exhaustive search -- the fewest 128-row tiles per image, then
// the smallest patch.  0 = not applicable.  mode (g_wd_b2d): 2 (default) = wherever it applies (every layer of
// profiles/r06_wgrad_b2d_ab.txt is faster on it: 1.08-1.42 x), 1 = only where the planar kernels would need the 64-pixel
// ring or bands, 0 = off.
IIC_SWITCH(g_wd_b2d, 2, iic_debug_wgrad_b2d)
static int wdb_config(const iic_conv_geom* g, wdb_args* A) {
  if (!g_wd_b2d || !g_wd_enabled || g->ntaps != 9 || g->Cin % 64 != 0 || g->Cout % 128 != 0) return 0;
  if (g->sy != 1 || g->sx != 1 || g->ty != 1 || g->tx != 1) return 0;
  const int txs = wdp_txs(g);
  if (!txs) return 0;
  const int rowoff = g->tap_off[3] - g->tap_off[0];
  if (rowoff <= 0 || rowoff % g->in_Wp != 0 || g->tap_off[6] - g->tap_off[3] != rowoff || g->tap_off[0] != 0) return 0;
  const int drow = rowoff / g->in_Wp;
  if ((long)g->N * g->out_Hp * g->out_Wp * g->Cout * 2 >= (1L << 32)) return 0;
  if ((long)g->N * g->in_Hp * g->in_Wp * g->Cin * 2 >= (1L << 32)) return 0;       // 32-bit patch offsets (end-of-tensor path)
  // small images: blocks within one image waste rows, and at 25 x 25 (ClusterNet5g layer 2: 5 x 25 blocks, 3 % faster
  // alone) the step measured 0.1 ms slower (profiles/r06_wgrad_b2d_ab.txt) -- those stay on the planar kernels
  if (g->MY < 32 || g->MX < 32) return 0;
  // block shape: tiles x (MFMA time of a tile + half its DMA bytes, in units of a 64-KB tile) over the shapes that fit
  double best = -1.0;
  long best_tiles = 0;
  int best_bw = 0;
  for (int bw = 4; bw <= 64 && bw <= g->MX; ++bw) {
    const int bh = 128 / bw;
    if (bh < 1 || bh > g->MY) continue;
    const long tiles = (long)((g->MX + bw - 1) / bw) * ((g->MY + bh - 1) / bh);
    const int patch = (bw + 2 * txs) * (bh + 2 * drow);
    const long plane = wdp_plane_bytes(patch);
    if ((2 * ((patch + 15) >> 4) + 32 + 11) / 12 > WDB_MAXNI) continue;
    if (2L * (2L * plane + 128 * 256) > 160 * 1024) continue;
    const double cost = (double)tiles * (1.0 + 0.5 * ((double)patch * 128.0 + 32768.0) / 65536.0);
    if (best < 0 || cost < best) { best = cost; best_tiles = tiles; best_bw = bw; }
  }
  if (best < 0) return 0;
  A->bw = best_bw; A->bh = 128 / best_bw;
  A->nbx = (g->MX + A->bw - 1) / A->bw; A->nby = (g->MY + A->bh - 1) / A->bh;
  if ((double)g->MY * g->MX < 0.88 * 128.0 * (double)best_tiles) return 0;        // > 12 % idle rows: not worth it
  A->PW = A->bw + 2 * txs; A->drow = drow;
  A->NPR = A->PW * (A->bh + 2 * drow);
  A->plane_bytes = (int)wdp_plane_bytes(A->NPR);
  A->num_tiles = (int)(best_tiles * g->N);
  if (g_wd_b2d == 1) {                             // only where the planar kernels fall off their 128-pixel contiguous layout
    int bmk, nbuf, ntab, band;
    if (wdp_config(g, &bmk, &nbuf, &ntab, &band) && bmk == 128 && band == 0) return 0;
  }
  return 1;
}

#ifdef IIC_DEBUG_HOOKS
// Which weight-gradient kernel / layout a 3 x 3 geometry gets (tests: a layer must not fall off the planar kernels by a
// few bytes of LDS unnoticed -- SegmentationNet10a c3 / c4 did, by 64): 0 = register-staged (conv_wgrad.hip),
// 1 = first-generation DMA kernel, 2 = planar, 3 = planar with the banded patch; + 100 * K-tile pixels + 10000 * ring
// depth + 100000 * table ring.
IIC_HOOK int iic_debug_wgrad_config(const iic_conv_geom* g) {
  int bmk = 0, nbuf = 0, ntab = 0, band = 0;
  if (!g || !g_wd_enabled) return 0;
  wdb_args A;
  if (wdb_config(g, &A)) return 4 + 100 * 128 + 10000 * 2 + 1000000 * A.bw;      // 4 = block-tiled (+ 1e6 * block width)
  if (wdp_config(g, &bmk, &nbuf, &ntab, &band)) return (band > 0 ? 3 : 2) + 100 * bmk + 10000 * nbuf + 100000 * ntab;
  if (wd_config(g, &bmk, &nbuf)) return 1 + 100 * bmk + 10000 * nbuf + 100000 * WD_NTAB;
  return 0;
}
#endif

// used by conv_wgrad.hip's dispatcher
int iic_wgrad_dma_supported(const iic_conv_geom* g) {
  int bmk, nbuf, ntab, band;
  wdb_args A;
  return g_wd_enabled && (wdb_config(g, &A) || wdp_config(g, &bmk, &nbuf, &ntab, &band) || wd_config(g, &bmk, &nbuf));
}

int iic_wgrad_dma_launch(const iic_conv_geom* g, const void* x, const void* dy, float* partials,
                         int nsplit, void* stream) {
  const int cot = (g->Cout % 128 == 0) ? 128 : 64;
  const long M = (long)g->N * (g->MP > 0 ? g->MP : g->MY * g->MX);
  int mto = 0;
  for (int i = 0; i < g->ntaps; ++i) mto = g->tap_off[i] > mto ? g->tap_off[i] : mto;
  dim3 grid((g->Cout / cot) * (g->Cin / 64), nsplit);
  hipStream_t s = (hipStream_t)stream;
  wdb_args BA;
  if (wdb_config(g, &BA)) {                            // block-tiled kernel (large images)
    const long ldsb = 2L * (2L * BA.plane_bytes + 128 * 256);
#define WDB_LAUNCH(TXS_)                                                                          \
  do {                                                                                          \
    static bool attr = false;                                                                   \
    if (!attr) {                                                                                \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_b2d_kernel<TXS_>),    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);        \
      attr = true;                                                                              \
    }                                                                                           \
    hipLaunchKernelGGL((conv_wgrad_b2d_kernel<TXS_>), grid, dim3(WD_THREADS), ldsb, s, *g,      \
                       (const bf16_t*)x, (const bf16_t*)dy, partials, nsplit, BA, g_wd_ablate); \
  } while (0)
    if (wdp_txs(g) == 1) WDB_LAUNCH(1); else WDB_LAUNCH(2);
    return iic_launch_status();
  }
  int pbmk = 0, pnbuf = 0, pntab = 0, pband = 0;
  if (wdp_config(g, &pbmk, &pnbuf, &pntab, &pband)) {  // planar-patch kernels
    const int txs = wdp_txs(g);
    const int bstride = g->tap_off[3] - g->tap_off[0];
    const int np = pband > 0 ? 3 * pband : (pbmk == 64 ? g->NP64 : g->NP);      // LDS rows per plane
    const int kt = (int)((M + pbmk - 1) / pbmk);
    const int plane = (int)wdp_plane_bytes(np);
    const long ldsp = wdp_lds(np, cot, pbmk, pnbuf, pntab);
#define WDP_LAUNCH3(COT_, BMK_, NBUF_, TXS_, ASM_, NTAB_)                                        \
  do {                                                                                          \
    static bool attr = false;                                                                   \
    if (!attr) {                                                                                \
      (void)hipFuncSetAttribute(                                                                \
          reinterpret_cast<const void*>(&conv_wgrad_pl_kernel<COT_, BMK_, NBUF_, TXS_, ASM_, NTAB_>), \
          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                              \
      attr = true;                                                                              \
    }                                                                                           \
    hipLaunchKernelGGL((conv_wgrad_pl_kernel<COT_, BMK_, NBUF_, TXS_, ASM_, NTAB_>), grid,      \
                       dim3(WD_THREADS), ldsp, s, *g, (const bf16_t*)x, (const bf16_t*)dy,      \
                       partials, nsplit, kt, plane, mto, g_wd_ablate, pband, bstride);          \
  } while (0)
#define WDP2_LAUNCH3(COT_, BMK_, NBUF_, TXS_, SWP_, NTAB_)                                       \
  do {                                                                                          \
    static bool attr = false;                                                                   \
    if (!attr) {                                                                                \
      (void)hipFuncSetAttribute(                                                                \
          reinterpret_cast<const void*>(&conv_wgrad_pl2_kernel<COT_, BMK_, NBUF_, TXS_, SWP_, NTAB_>), \
          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                              \
      attr = true;                                                                              \
    }                                                                                           \
    hipLaunchKernelGGL((conv_wgrad_pl2_kernel<COT_, BMK_, NBUF_, TXS_, SWP_, NTAB_>), grid,     \
                       dim3(WD_THREADS), ldsp, s, *g, (const bf16_t*)x, (const bf16_t*)dy,      \
                       partials, nsplit, kt, plane, mto, g_wd_ablate, pband, bstride);          \
  } while (0)
#ifdef IIC_DEBUG_HOOKS
#define WDP_LAUNCH2(COT_, BMK_, NBUF_, TXS_, NTAB_)                                              \
  do {                                                                                          \
    if (g_wd_planar == 4 || (g_wd_planar == 5 && COT_ == 64)) WDP2_LAUNCH3(COT_, BMK_, NBUF_, TXS_, true, NTAB_); \
    else if (g_wd_planar == 3) WDP2_LAUNCH3(COT_, BMK_, NBUF_, TXS_, false, NTAB_);             \
    else if (g_wd_planar == 1) WDP_LAUNCH3(COT_, BMK_, NBUF_, TXS_, false, NTAB_);              \
    else WDP_LAUNCH3(COT_, BMK_, NBUF_, TXS_, true, NTAB_);                                     \
  } while (0)
#else      /* the product library instantiates the default forms only */
#define WDP_LAUNCH2(COT_, BMK_, NBUF_, TXS_, NTAB_)                                              \
  do {                                                                                          \
    if (COT_ == 64) WDP2_LAUNCH3(64, BMK_, NBUF_, TXS_, true, NTAB_);                           \
    else WDP_LAUNCH3(128, BMK_, NBUF_, TXS_, true, NTAB_);                                      \
  } while (0)
#endif
#define WDP_LAUNCH(BMK_, NBUF_, NTAB_)                                                           \
  do {                                                                                          \
    if (cot == 128) { if (txs == 1) WDP_LAUNCH2(128, BMK_, NBUF_, 1, NTAB_); else WDP_LAUNCH2(128, BMK_, NBUF_, 2, NTAB_); } \
    else { if (txs == 1) WDP_LAUNCH2(64, BMK_, NBUF_, 1, NTAB_); else WDP_LAUNCH2(64, BMK_, NBUF_, 2, NTAB_); } \
  } while (0)
    if (pbmk == 128) { if (pntab == 8) WDP_LAUNCH(128, 2, 8); else WDP_LAUNCH(128, 2, 4); }
    else if (pnbuf == 4) WDP_LAUNCH(64, 4, 8);
    else if (pnbuf == 3) WDP_LAUNCH(64, 3, 8);
    else if (pntab == 8) WDP_LAUNCH(64, 2, 8);
    else WDP_LAUNCH(64, 2, 4);
    return iic_launch_status();
  }
  int bmk = 0, nbuf = 0;
  if (!wd_config(g, &bmk, &nbuf)) return IIC_ERR_UNSUPPORTED;
  const int kt = (int)((M + bmk - 1) / bmk);
  const int np = bmk == 64 ? g->NP64 : g->NP;
  const int xb = (int)wd_xb_bytes(np);
  const long lds = wd_lds(np, cot, bmk, nbuf);
#define WD_LAUNCH2(COT_, BMK_, NBUF_, ASM_, PF_, SWP_)                                           \
  do {                                                                                          \
    static bool attr = false;                                                                   \
    if (!attr) {                                                                                \
      (void)hipFuncSetAttribute(                                                                \
          reinterpret_cast<const void*>(&conv_wgrad_dma_kernel<COT_, BMK_, NBUF_, ASM_, PF_, SWP_>), \
          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                              \
      attr = true;                                                                              \
    }                                                                                           \
    hipLaunchKernelGGL((conv_wgrad_dma_kernel<COT_, BMK_, NBUF_, ASM_, PF_, SWP_>), grid,       \
                       dim3(PF_ ? 256 : WD_THREADS), lds, s, *g, (const bf16_t*)x,              \
                       (const bf16_t*)dy,                                                       \
                       partials, nsplit, kt, xb, mto, g_wd_ablate);                             \
  } while (0)
#ifdef IIC_DEBUG_HOOKS
#define WD_LAUNCH(COT_, BMK_, NBUF_)                                                             \
  do {                                                                                          \
    if (g_wd_asm) WD_LAUNCH2(COT_, BMK_, NBUF_, true, false, false);                            \
    else if (g_wd_prefetch) WD_LAUNCH2(COT_, BMK_, NBUF_, false, true, false);                  \
    else if (g_wd_swp) WD_LAUNCH2(COT_, BMK_, NBUF_, false, false, true);                       \
    else WD_LAUNCH2(COT_, BMK_, NBUF_, false, false, false);                                    \
  } while (0)
#else
#define WD_LAUNCH(COT_, BMK_, NBUF_) WD_LAUNCH2(COT_, BMK_, NBUF_, false, false, (g_wd_swp != 0))
#endif
  if (bmk == 64 && nbuf == 4) {
    if (cot == 128) WD_LAUNCH(128, 64, 4); else WD_LAUNCH(64, 64, 4);
  } else if (bmk == 64 && nbuf == 3) {
    if (cot == 128) WD_LAUNCH(128, 64, 3); else WD_LAUNCH(64, 64, 3);
  } else if (bmk == 64) {
    if (cot == 128) WD_LAUNCH(128, 64, 2); else WD_LAUNCH(64, 64, 2);
  } else {
    if (cot == 128) WD_LAUNCH(128, 128, 2); else WD_LAUNCH(64, 128, 2);
  }
  return iic_launch_status();
}

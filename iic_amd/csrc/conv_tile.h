// Pieces shared by the implicit-GEMM conv kernels (conv_igemm.hip, conv_igemm_bd.hip):
// LDS patch staging, XCD-aware tile order, and the tile store with the fused
// accumulate / ReLU-masked residual-gradient epilogue.
#pragma once
#include "common.h"
#include "../../include/iic_hip.h"

// GEMM row -> pixel indices.  Rows are numbered per image with g.MP rows each (0 = the dense
// plane MY*MX); rows >= plane of an image, and rows of images >= N, are invalid: they read the
// image's last pixel (stays inside the tile's patch span) and get pout = -1.
__device__ __forceinline__ void igemm_row_pixels(const iic_conv_geom& g, int m, int& pin, int& pout) {
  const int plane = g.MY * g.MX;
  const int mp = g.MP > 0 ? g.MP : plane;
  int n = m / mp;
  int r = m - n * mp;
  const bool valid = n < g.N && r < plane;
  if (n >= g.N) { n = g.N - 1; r = plane - 1; }
  r = r < plane ? r : plane - 1;
  const int y = r / g.MX, x = r - y * g.MX;
  pin = (n * g.in_Hp + y * g.sy + g.oy) * g.in_Wp + x * g.sx + g.ox;
  pout = valid ? (n * g.out_Hp + y * g.ty + g.py) * g.out_Wp + x * g.tx + g.px : -1;
}
__device__ __forceinline__ int igemm_rows(const iic_conv_geom& g) {
  return g.N * (g.MP > 0 ? g.MP : g.MY * g.MX);
}
// does the tile [m0, m0 + bm) contain invalid rows?
__device__ __forceinline__ bool igemm_tile_has_invalid(const iic_conv_geom& g, int m0, int bm) {
  const int plane = g.MY * g.MX;
  if (m0 + bm > igemm_rows(g)) return true;
  return g.MP > 0 && g.MP != plane && (m0 % g.MP) + bm > plane;
}
static inline long igemm_rows_host(const iic_conv_geom* g) {
  return (long)g->N * (g->MP > 0 ? g->MP : g->MY * g->MX);
}
static inline bool igemm_dense_host(const iic_conv_geom* g) { return g->MP <= 0 || g->MP == g->MY * g->MX; }

#define ROWB 144   // LDS row pitch in bytes: 128 B of data + 16 B pad.  144*r mod 256 visits all
                   // sixteen 16-B slots over 16 consecutive rows => the 16-lane groups of
                   // ds_read_b128 are conflict-free, and every k-step is an IMMEDIATE offset
                   // (ks*32 B) from one per-tap row address: no address VALU in the MFMA loop.

// XCD-aware tile order: blocks b, b+8, b+16.. share an XCD (observed dispatch); give each XCD
// a contiguous range of tiles so neighbouring M-tiles (shared halo, shared weights) hit the
// same L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_tile_index(int b, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, idx = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Stage the input patch (npix pixels x 64 channels) or, in gather mode (1-tap convs), the
// tile rows' own pixels.  NB independent 16-B loads in flight per thread and batch (every batch
// is exposed to one full memory latency: use the largest NB the live registers allow).
template <bool GATHER, int NTHREADS, int NB = 4>
__device__ __forceinline__ void igemm_load_patch(unsigned char* sA, const bf16_t* __restrict__ in,
                                                 int Cin, int c0, int p_lo, int npix,
                                                 int in_pixels, const int* s_pin, int tid) {
  const int n8 = npix * 8;
  for (int base = 0; base < n8; base += NTHREADS * NB) {
    u32x4 v[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int idx = base + u * NTHREADS + tid;
      v[u] = (u32x4){0u, 0u, 0u, 0u};
      if (idx < n8) {
        const long p = GATHER ? (long)s_pin[idx >> 3] : (long)p_lo + (idx >> 3);
        if (p < in_pixels) v[u] = *reinterpret_cast<const u32x4*>(in + (p * Cin + c0 + (idx & 7) * 8));
      }
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int idx = base + u * NTHREADS + tid;
      if (idx < n8) *reinterpret_cast<u32x4*>(sA + (idx >> 3) * ROWB + (idx & 7) * 16) = v[u];
    }
  }
}

// sC [BM][BN + PAD] bf16 (LDS) -> out rows s_pout[row] (skipped when < 0), 16-byte stores.
// accumulate is a flag word (include/iic_hip.h): IIC_ACC_ADD: out += previous contents;
// without IIC_ACC_PREMASK: res_grad/res_act: out += res_grad where res_act > 0;
// with IIC_ACC_PREMASK: out = (value [+ previous] [+ res_grad]) where res_act > 0, else 0 -- the
// gradient leaves already multiplied by the ReLU mask of the activation it belongs to (res_act =
// this conv's INPUT activation), so its consumers do not read that activation again.
//
// RED (fused BatchNorm-backward reduction, include/iic_hip.h iic_conv_igemm_frag_red): the tile
// being stored is a gradient g that a BatchNorm backward consumes next; its two reductions
//   sum g   and   sum g * y      (y = that BatchNorm's input, a PT tensor shaped like `out`;
//                                 RED == 2: also sum g * y2 for the downsample branch's BatchNorm)
// are taken here from the values as they are stored (bf16-rounded), masked by
// (scale*y + shift > 0) when red_coef != nullptr (the ReLU between that BatchNorm and this conv,
// recomputed from y exactly as bn_bwd_reduce does).  The y tile rides along with an MFMA-bound
// kernel instead of costing a separate HBM-bound pass over g and y.  A thread always owns the same
// 8 channels (NTHREADS % (BN/8) == 0), so the partial sums live in registers (TileRed) across
// rows and -- persistent kernels -- across tiles; igemm_red_finish folds them.
struct TileRed {
  float s[8], sy[8], sy2[8];
};
__device__ __forceinline__ void tile_red_zero(TileRed& r) {
#pragma unroll
  for (int i = 0; i < 8; ++i) r.s[i] = r.sy[i] = r.sy2[i] = 0.f;
}

template <int BN, int BM, int NTHREADS, int PAD = 8, int RED = 0, int UB = 4>
__device__ __forceinline__ void igemm_store_tile(const bf16_t* sC, const int* s_pout,
                                                 bf16_t* __restrict__ out,
                                                 const bf16_t* __restrict__ res_grad,
                                                 const bf16_t* __restrict__ res_act, int accumulate,
                                                 int Cout, int n0, int tid,
                                                 const bf16_t* __restrict__ red_y = nullptr,
                                                 const float* __restrict__ red_coef = nullptr,
                                                 const bf16_t* __restrict__ red_y2 = nullptr,
                                                 TileRed* red = nullptr) {
  constexpr int CLD = BN + PAD;
  constexpr int CH = BN / 8;
  constexpr int ITERS = (BM * CH + NTHREADS - 1) / NTHREADS;
  constexpr int U = ITERS < UB ? ITERS : UB;    // rows whose global loads are in flight together
  static_assert(NTHREADS % CH == 0, "a thread must keep its channel chunk");
  const bool add_prev = accumulate & IIC_ACC_ADD, premask = accumulate & IIC_ACC_PREMASK;
  const bool any_in = add_prev || res_grad || res_act;
  const int ch = tid % CH;
  // the ReLU-mask coefficients of the fused reduction: this thread's 8 channels, loaded once
  float msc[RED ? 8 : 1], msh[RED ? 8 : 1];
  if (RED && red_coef) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      msc[RED ? i : 0] = red_coef[n0 + ch * 8 + i];
      msh[RED ? i : 0] = red_coef[Cout + n0 + ch * 8 + i];
    }
  }
  // Rows are processed U at a time: first ALL global loads of the batch are issued (previous
  // contents, residual gradient, mask activation, the reduction's y / y2), then the batch is
  // computed and stored -- one memory latency per batch instead of one per row (the loop used to
  // wait for each row's loads in turn: ~16 exposed latencies per tile in the backward-data epilogue).
  for (int b0 = 0; b0 < ITERS; b0 += U) {
    long o[U];
    uint4 pv[U], gv[U], av[U], yv[U], zv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = tid + (b0 + u) * NTHREADS;
      const int row = idx / CH;
      const int po = (idx < BM * CH) ? s_pout[row] : -1;
      o[u] = po < 0 ? -1 : (long)po * Cout + n0 + ch * 8;
      if (o[u] >= 0) {
        if (add_prev) pv[u] = *reinterpret_cast<const uint4*>(out + o[u]);
        if (res_grad) gv[u] = *reinterpret_cast<const uint4*>(res_grad + o[u]);
        if (res_act) av[u] = *reinterpret_cast<const uint4*>(res_act + o[u]);
        if (RED) yv[u] = *reinterpret_cast<const uint4*>(red_y + o[u]);
        if (RED == 2) zv[u] = *reinterpret_cast<const uint4*>(red_y2 + o[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (o[u] < 0) continue;
      const int row = (tid + (b0 + u) * NTHREADS) / CH;
      uint4 v = *reinterpret_cast<const uint4*>(sC + row * CLD + ch * 8);
      if (any_in) {
        uint32_t vv[4] = {v.x, v.y, v.z, v.w};
        float f[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) { f[2 * i] = bf16lo(vv[i]); f[2 * i + 1] = bf16hi(vv[i]); }
        if (add_prev) {
          const uint32_t oo[4] = {pv[u].x, pv[u].y, pv[u].z, pv[u].w};
#pragma unroll
          for (int i = 0; i < 4; ++i) { f[2 * i] += bf16lo(oo[i]); f[2 * i + 1] += bf16hi(oo[i]); }
        }
        if (premask) {
          if (res_grad) {
            const uint32_t gg[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) { f[2 * i] += bf16lo(gg[i]); f[2 * i + 1] += bf16hi(gg[i]); }
          }
          if (res_act) {
            const uint32_t aa[4] = {av[u].x, av[u].y, av[u].z, av[u].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              if (!(bf16lo(aa[i]) > 0.f)) f[2 * i] = 0.f;
              if (!(bf16hi(aa[i]) > 0.f)) f[2 * i + 1] = 0.f;
            }
          }
        } else if (res_grad) {
          const uint32_t gg[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
          const uint32_t aa[4] = {av[u].x, av[u].y, av[u].z, av[u].w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (bf16lo(aa[i]) > 0.f) f[2 * i] += bf16lo(gg[i]);
            if (bf16hi(aa[i]) > 0.f) f[2 * i + 1] += bf16hi(gg[i]);
          }
        }
        v = make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]),
                       pack_bf16x2(f[6], f[7]));
      }
      *reinterpret_cast<uint4*>(out + o[u]) = v;
      if (RED) {
        const uint32_t gq4[4] = {v.x, v.y, v.z, v.w};
        const uint32_t yy[4] = {yv[u].x, yv[u].y, yv[u].z, yv[u].w};
        float gq[8], yq[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          gq[2 * i] = bf16lo(gq4[i]); gq[2 * i + 1] = bf16hi(gq4[i]);
          yq[2 * i] = bf16lo(yy[i]); yq[2 * i + 1] = bf16hi(yy[i]);
        }
        if (red_coef) {
#pragma unroll
          for (int i = 0; i < 8; ++i)
            if (!(yq[i] * msc[RED ? i : 0] + msh[RED ? i : 0] > 0.f)) gq[i] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { red->s[i] += gq[i]; red->sy[i] += gq[i] * yq[i]; }
        if (RED == 2) {
          const uint32_t zz[4] = {zv[u].x, zv[u].y, zv[u].z, zv[u].w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            red->sy2[2 * i] += gq[2 * i] * bf16lo(zz[i]);
            red->sy2[2 * i + 1] += gq[2 * i + 1] * bf16hi(zz[i]);
          }
        }
      }
    }
  }
}

// Fold the TileRed partials of a workgroup and add them to the exact statistic accumulators:
// lanes of a wave that own the same channel chunk by shuffles, waves through `scratch`
// (NTHREADS/64 x 3 x BN floats of LDS that nobody else is using any more), fixed order.
// red_stats: sums (sum g, sum g*y); red_stats2 (RED == 2): (sum g, sum g*y2).
template <int BN, int NTHREADS, int RED>
__device__ __forceinline__ void igemm_red_finish(TileRed& r, float* scratch, float* red_stats,
                                                 float* red_stats2, int Cout, int n0, int tid) {
  constexpr int CH = BN / 8, NW = NTHREADS / 64;
  static_assert(64 % CH == 0, "channel chunks must divide the wave");
  const int lane = tid & 63, wave = tid >> 6, ch = tid % CH;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int o = CH; o < 64; o <<= 1) {
      r.s[i] += __shfl_xor(r.s[i], o, 64);
      r.sy[i] += __shfl_xor(r.sy[i], o, 64);
      if (RED == 2) r.sy2[i] += __shfl_xor(r.sy2[i], o, 64);
    }
  }
  __syncthreads();                       // every wave is done with the LDS that becomes scratch
  if (lane < CH) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      scratch[(wave * 3 + 0) * BN + ch * 8 + i] = r.s[i];
      scratch[(wave * 3 + 1) * BN + ch * 8 + i] = r.sy[i];
      if (RED == 2) scratch[(wave * 3 + 2) * BN + ch * 8 + i] = r.sy2[i];
    }
  }
  __syncthreads();
  if (tid < BN) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int w = 0; w < NW; ++w) {
      a0 += scratch[(w * 3 + 0) * BN + tid];
      a1 += scratch[(w * 3 + 1) * BN + tid];
      if (RED == 2) a2 += scratch[(w * 3 + 2) * BN + tid];
    }
    const int stripe = blockIdx.x % IIC_STAT_STRIPES;
    iic_stat_add(red_stats, stripe, Cout, n0 + tid, 0, a0);
    iic_stat_add(red_stats, stripe, Cout, n0 + tid, 1, a1);
    if (RED == 2) {
      iic_stat_add(red_stats2, stripe, Cout, n0 + tid, 0, a0);
      iic_stat_add(red_stats2, stripe, Cout, n0 + tid, 1, a2);
    }
  }
}

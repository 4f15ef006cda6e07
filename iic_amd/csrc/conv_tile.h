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
template <int BN, int BM, int NTHREADS, int PAD = 8>
__device__ __forceinline__ void igemm_store_tile(const bf16_t* sC, const int* s_pout,
                                                 bf16_t* __restrict__ out,
                                                 const bf16_t* __restrict__ res_grad,
                                                 const bf16_t* __restrict__ res_act, int accumulate,
                                                 int Cout, int n0, int tid) {
  constexpr int CLD = BN + PAD;
  const bool add_prev = accumulate & IIC_ACC_ADD, premask = accumulate & IIC_ACC_PREMASK;
  for (int idx = tid; idx < BM * (BN / 8); idx += NTHREADS) {
    const int row = idx / (BN / 8), ch = idx - row * (BN / 8);
    const int po = s_pout[row];
    if (po < 0) continue;
    uint4 v = *reinterpret_cast<const uint4*>(sC + row * CLD + ch * 8);
    const long o = (long)po * Cout + n0 + ch * 8;
    if (add_prev || res_grad || res_act) {
      uint32_t vv[4] = {v.x, v.y, v.z, v.w};
      float f[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) { f[2 * i] = bf16lo(vv[i]); f[2 * i + 1] = bf16hi(vv[i]); }
      if (add_prev) {
        const uint4 ov = *reinterpret_cast<const uint4*>(out + o);
        const uint32_t oo[4] = {ov.x, ov.y, ov.z, ov.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { f[2 * i] += bf16lo(oo[i]); f[2 * i + 1] += bf16hi(oo[i]); }
      }
      if (premask) {
        if (res_grad) {
          const uint4 gv = *reinterpret_cast<const uint4*>(res_grad + o);
          const uint32_t gg[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) { f[2 * i] += bf16lo(gg[i]); f[2 * i + 1] += bf16hi(gg[i]); }
        }
        if (res_act) {
          const uint4 av = *reinterpret_cast<const uint4*>(res_act + o);
          const uint32_t aa[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (!(bf16lo(aa[i]) > 0.f)) f[2 * i] = 0.f;
            if (!(bf16hi(aa[i]) > 0.f)) f[2 * i + 1] = 0.f;
          }
        }
      } else if (res_grad) {
        const uint4 gv = *reinterpret_cast<const uint4*>(res_grad + o);
        const uint4 av = *reinterpret_cast<const uint4*>(res_act + o);
        const uint32_t gg[4] = {gv.x, gv.y, gv.z, gv.w}, aa[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (bf16lo(aa[i]) > 0.f) f[2 * i] += bf16lo(gg[i]);
          if (bf16hi(aa[i]) > 0.f) f[2 * i + 1] += bf16hi(gg[i]);
        }
      }
      v = make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]),
                     pack_bf16x2(f[6], f[7]));
    }
    *reinterpret_cast<uint4*>(out + o) = v;
  }
}

// Cluster heads for gfx950: global AvgPool -> Linear -> Softmax(dim=1) per sub-head, fp32.
//
// Replaces /root/reference/code/archs/cluster/net5g.py:31-39,53 (AvgPool2d + view) and
// :69-80 (nn.Linear + nn.Softmax per sub-head); also used for ClusterNet6c's flatten +
// Linear heads (net6c.py:47-59).
//
// All sub-heads are concatenated into ONE exact-fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32):
// logits[n][h*k + j]; the softmax kernel treats every (n, h) segment as a row.  These ops
// are tiny (0.5 GFLOP / step) and launch-latency bound; fp32 keeps the loss inputs at
// reference precision.
#include "common.h"
#include "../../include/iic_hip.h"

// feats[n][c] = mean over the H*W interior pixels of in[n][.][.][c]   (PT bf16 -> fp32)
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const bf16_t* __restrict__ in,
                                                          float* __restrict__ feats, int H, int W,
                                                          int P, int C) {
  const int n = blockIdx.x;
  const int Hp = H + 2 * P, Wp = W + 2 * P;
  const float inv = 1.f / (float)(H * W);
  for (int c2 = threadIdx.x; c2 < C / 2; c2 += blockDim.x) {
    float s0 = 0.f, s1 = 0.f;
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        const uint32_t v = *reinterpret_cast<const uint32_t*>(
            in + (((long)n * Hp + y + P) * Wp + x + P) * C + 2 * c2);
        s0 += bf16lo(v);
        s1 += bf16hi(v);
      }
    feats[(long)n * C + 2 * c2] = s0 * inv;
    feats[(long)n * C + 2 * c2 + 1] = s1 * inv;
  }
}

// din[n][y][x][c] = dfeats[n][c] / (H*W)
// act (nullable): din is zeroed where act <= 0 (the ReLU mask of the pooled activation, so that the
// last block's BatchNorm backward does not have to read it: archs/cluster.py PREMASK)
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dfeats,
                                                          bf16_t* __restrict__ din,
                                                          const bf16_t* __restrict__ act, int H,
                                                          int W, int P, int C) {
  const int n = blockIdx.x;
  const int Hp = H + 2 * P, Wp = W + 2 * P;
  const float inv = 1.f / (float)(H * W);
  for (int c2 = threadIdx.x; c2 < C / 2; c2 += blockDim.x) {
    const uint32_t v = pack_bf16x2(dfeats[(long)n * C + 2 * c2] * inv,
                                   dfeats[(long)n * C + 2 * c2 + 1] * inv);
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        const long o = (((long)n * Hp + y + P) * Wp + x + P) * C + 2 * c2;
        uint32_t w = v;
        if (act) {
          const uint32_t a = *reinterpret_cast<const uint32_t*>(act + o);
          if (!(bf16lo(a) > 0.f)) w &= 0xffff0000u;
          if (!(bf16hi(a) > 0.f)) w &= 0x0000ffffu;
        }
        *reinterpret_cast<uint32_t*>(din + o) = w;
      }
  }
}

// C[m][n] (+)= sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn] (+ bias[n]).  One wave per 32x32 tile.
__global__ __launch_bounds__(64) void gemm_f32_kernel(const float* __restrict__ A, long sam,
                                                      long sak, const float* __restrict__ B,
                                                      long sbk, long sbn,
                                                      const float* __restrict__ bias,
                                                      float* __restrict__ Cm, long scm, int M, int Nn,
                                                      int K, int accumulate) {
  const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  const int lane = threadIdx.x, i = lane & 31, kk = lane >> 5;
  const bool vm = (m0 + i) < M, vn = (n0 + i) < Nn;
  const float* ap = A + (long)(m0 + i) * sam;
  const float* bp = B + (long)(n0 + i) * sbn;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // split-K: blockIdx.z owns an (even-aligned) slice of K and atomically adds into C
  const int nz = gridDim.z;
  int k = 0;
  if (nz > 1) {
    const int per = (((K + nz - 1) / nz) + 1) & ~1;
    k = blockIdx.z * per;
    K = min(K, k + per);
  }
  for (; k + 8 <= K; k += 8) {
    float a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kq = k + 2 * u + kk;
      a[u] = vm ? ap[(long)kq * sak] : 0.f;
      b[u] = vn ? bp[(long)kq * sbk] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
  }
  for (; k < K; k += 2) {
    const int kq = k + kk;
    const float a = (vm && kq < K) ? ap[(long)kq * sak] : 0.f;
    const float b = (vn && kq < K) ? bp[(long)kq * sbk] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  const int col = n0 + i;
  const float bv = (bias && col < Nn) ? bias[col] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + mfma32_row(r, lane);
    if (row < M && col < Nn) {
      float* o = Cm + (long)row * scm + col;
      if (nz > 1) {
        atomicAdd(o, acc[r] + (blockIdx.z == 0 ? bv : 0.f));
      } else {
        float v = acc[r] + bv;
        if (accumulate) v += *o;
        *o = v;
      }
    }
  }
}

// The same product with KW waves per 32x32 tile: wave w owns a contiguous slice of K, the partial tiles
// are folded through LDS in wave order (fixed order: bit-reproducible, no atomics).  For the long-K,
// few-tile products of the VGG-style heads (ClusterNet6c: logits [700 x 250] over K = 4608 is only 176
// tiles -- one wave each left 5/6 of the SIMDs idle for 125 us; round-3 profile r03_6c_kernel_stats).
template <int KW>
__global__ __launch_bounds__(64 * KW) void gemm_f32_kw_kernel(const float* __restrict__ A, long sam, long sak,
                                                               const float* __restrict__ B, long sbk, long sbn,
                                                               const float* __restrict__ bias,
                                                               float* __restrict__ Cm, long scm, int M, int Nn,
                                                               int K, int accumulate) {
  __shared__ float part[KW][32 * 32];
  const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i = lane & 31, kk = lane >> 5;
  const bool vm = (m0 + i) < M, vn = (n0 + i) < Nn;
  const float* ap = A + (long)(m0 + i) * sam;
  const float* bp = B + (long)(n0 + i) * sbn;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int per = (((K + KW - 1) / KW) + 7) & ~7;
  int k = w * per;
  const int kend = min(K, k + per);
  for (; k + 8 <= kend; k += 8) {
    float a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kq = k + 2 * u + kk;
      a[u] = vm ? ap[(long)kq * sak] : 0.f;
      b[u] = vn ? bp[(long)kq * sbk] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
  }
  for (; k < kend; k += 2) {
    const int kq = k + kk;
    const float a = (vm && kq < kend) ? ap[(long)kq * sak] : 0.f;
    const float b = (vn && kq < kend) ? bp[(long)kq * sbk] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) part[w][mfma32_row(r, lane) * 32 + i] = acc[r];
  __syncthreads();
  for (int e = threadIdx.x; e < 32 * 32; e += 64 * KW) {
    const int row = m0 + (e >> 5), col = n0 + (e & 31);
    if (row < M && col < Nn) {
      float v = part[0][e];
#pragma unroll
      for (int q = 1; q < KW; ++q) v += part[q][e];
      if (bias) v += bias[col];
      float* o = Cm + (long)row * scm + col;
      if (accumulate) v += *o;
      *o = v;
    }
  }
}


// LDS-tiled fp32 GEMM (same contract as gemm_f32_kernel): 64 x 64 tile per 4-wave block, K-tiles of 32 staged
// through double-buffered LDS with COALESCED global loads -- the one-wave kernels above read their operands
// with one 4-byte load per lane at the row stride (32 different lines per instruction), which made the long-K
// products of the VGG-style heads 6x slower than the fp32 MFMA rate (ClusterNet6c k = 280: logits
// [700 x 1400] over K = 4608 took 341 us).  AKC / BKC: the operand's unit stride runs along k (else along
// m / n): decides how the 256 threads walk the tile so that a wave's lanes read consecutive addresses.
// gridDim.z > 1: the K-tiles are split between gridDim.z blocks whose partial tiles go to
// ws[z][M][Nn]; gemm_f32_fold_kernel adds them in z order (deterministic, no atomics) -- for launches with
// too few tiles for the chip.  Products and sums are exact fp32 MFMA (32x32x2).
template <bool AKC, bool BKC>
__global__ __launch_bounds__(256) void gemm_f32_tiled_kernel(const float* __restrict__ A, long sam, long sak,
                                                              const float* __restrict__ B, long sbk, long sbn,
                                                              const float* __restrict__ bias,
                                                              float* __restrict__ Cm, long scm, int M, int Nn,
                                                              int K, int accumulate, float* __restrict__ ws) {
  constexpr int TK = 32, LD = 65;
  __shared__ float sA[2][TK * LD], sB[2][TK * LD];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int i = lane & 31, kk = lane >> 5;
  const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  const int ntiles = (K + TK - 1) / TK;
  const int S = gridDim.z, z = blockIdx.z;
  const int per = (ntiles + S - 1) / S;
  const int t0 = z * per, t1 = min(ntiles, t0 + per);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float ra[8], rb[8];
  auto gload = [&](int t) {
    const int k0 = t * TK;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int row, k;
      if (AKC) { k = tid & 31; row = u * 8 + (tid >> 5); } else { row = tid & 63; k = u * 4 + (tid >> 6); }
      ra[u] = (m0 + row < M && k0 + k < K) ? A[(long)(m0 + row) * sam + (long)(k0 + k) * sak] : 0.f;
      if (BKC) { k = tid & 31; row = u * 8 + (tid >> 5); } else { row = tid & 63; k = u * 4 + (tid >> 6); }
      rb[u] = (n0 + row < Nn && k0 + k < K) ? B[(long)(k0 + k) * sbk + (long)(n0 + row) * sbn] : 0.f;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int row, k;
      if (AKC) { k = tid & 31; row = u * 8 + (tid >> 5); } else { row = tid & 63; k = u * 4 + (tid >> 6); }
      sA[buf][k * LD + row] = ra[u];
      if (BKC) { k = tid & 31; row = u * 8 + (tid >> 5); } else { row = tid & 63; k = u * 4 + (tid >> 6); }
      sB[buf][k * LD + row] = rb[u];
    }
  };
  if (t0 < t1) {
    gload(t0);
    lstore(0);
    if (t0 + 1 < t1) gload(t0 + 1);
  }
  for (int t = t0; t < t1; ++t) {
    const int cur = (t - t0) & 1;
    __syncthreads();               // buffer `cur` complete; everyone is done reading the other one
    if (t + 1 < t1) lstore(cur ^ 1);
    if (t + 2 < t1) gload(t + 2);
#pragma unroll
    for (int k2 = 0; k2 < TK / 2; ++k2) {
      const float a = sA[cur][(2 * k2 + kk) * LD + wm * 32 + i];
      const float b = sB[cur][(2 * k2 + kk) * LD + wn * 32 + i];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  }
  const int col = n0 + wn * 32 + i;
  if (S > 1) {
    float* o = ws + (long)z * M * Nn;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * 32 + mfma32_row(r, lane);
      if (row < M && col < Nn) o[(long)row * Nn + col] = acc[r];
    }
    return;
  }
  const float bv = (bias && col < Nn) ? bias[col] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + wm * 32 + mfma32_row(r, lane);
    if (row < M && col < Nn) {
      float* o = Cm + (long)row * scm + col;
      float v = acc[r] + bv;
      if (accumulate) v += *o;
      *o = v;
    }
  }
}

// The same product on the bf16 matrix pipe (round 6): every fp32 operand element as three bf16 terms
// x = h + m + l (24 mantissa bits; the subtractions are exact), a product as the six MFMAs h h', h m', m h', h l', l h',
// m m' (v_mfma_f32_32x32x16_bf16, fp32 accumulate): what is dropped is below 2^-24 of a product, inside the rounding
// of the exact-fp32 kernel above (tests: same 2e-5-of-scale gate against float64).  Unlike the segmentation
// contractions (seg_loss.hip) the split is paid ONCE per element, at staging time -- a thread splits the 4
// consecutive k values it loaded and writes three 8-byte units -- because a GEMM's fragments are always 16-byte
// aligned in k: six bf16 MFMAs (6 x 32 cycles per 16 k) replace eight fp32 ones (8 x 64): 2.7 x less matrix time,
// which is what the long-K head products of the VGG-style nets (ClusterNet6c k = 280: [700 x 1400] over K = 4608,
// three of them per view and step) are made of.  Same tiling, K split and epilogue as gemm_f32_tiled_kernel.
#ifdef IIC_DEBUG_HOOKS
#define GX3_PITCH 80      // bytes per row of a 32-k bf16 tile in LDS (64 + 16: 16-byte aligned fragments)
template <bool AKC, bool BKC>
__global__ __launch_bounds__(256) void gemm_x3_tiled_kernel(const float* __restrict__ A, long sam, long sak,
                                                             const float* __restrict__ B, long sbk, long sbn,
                                                             const float* __restrict__ bias,
                                                             float* __restrict__ Cm, long scm, int M, int Nn,
                                                             int K, int accumulate, float* __restrict__ ws) {
  constexpr int TK = 32, PL = 64 * GX3_PITCH;                 // plane bytes
  __shared__ __attribute__((aligned(16))) unsigned char sA[2][3 * PL], sB[2][3 * PL];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int i = lane & 31, kg = lane >> 5;
  const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  const int ntiles = (K + TK - 1) / TK;
  const int S = gridDim.z, z = blockIdx.z;
  const int per = (ntiles + S - 1) / S;
  const int t0 = z * per, t1 = min(ntiles, t0 + per);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // a thread stages two units per operand and K-tile: unit = (row, 4 consecutive k)
  float ra[2][4], rb[2][4];
  auto unit_of = [&](bool kc, int u, int& row, int& k) {
    if (kc) { const int q = tid + 256 * u; row = q >> 3; k = (q & 7) * 4; }      // lanes walk k (unit stride)
    else { row = tid & 63; k = ((tid >> 6) + 4 * u) * 4; }                       // lanes walk the rows (unit stride)
  };
  auto gload = [&](int t) {
    const int k0 = t * TK;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int row, k;
      unit_of(AKC, u, row, k);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        ra[u][e] = (m0 + row < M && k0 + k + e < K) ? A[(long)(m0 + row) * sam + (long)(k0 + k + e) * sak] : 0.f;
      unit_of(BKC, u, row, k);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        rb[u][e] = (n0 + row < Nn && k0 + k + e < K) ? B[(long)(k0 + k + e) * sbk + (long)(n0 + row) * sbn] : 0.f;
    }
  };
  auto split4 = [&](const float (&v)[4], unsigned char* plane0, int row, int k) {
    uint2 H, Mi, L;
    float a = v[0], b = v[1], c = v[2], d = v[3];
    H.x = pack_bf16x2(a, b); H.y = pack_bf16x2(c, d);
    a -= bf16lo(H.x); b -= bf16hi(H.x); c -= bf16lo(H.y); d -= bf16hi(H.y);
    Mi.x = pack_bf16x2(a, b); Mi.y = pack_bf16x2(c, d);
    a -= bf16lo(Mi.x); b -= bf16hi(Mi.x); c -= bf16lo(Mi.y); d -= bf16hi(Mi.y);
    L.x = pack_bf16x2(a, b); L.y = pack_bf16x2(c, d);
    unsigned char* p = plane0 + row * GX3_PITCH + k * 2;
    *reinterpret_cast<uint2*>(p) = H;
    *reinterpret_cast<uint2*>(p + PL) = Mi;
    *reinterpret_cast<uint2*>(p + 2 * PL) = L;
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int row, k;
      unit_of(AKC, u, row, k);
      split4(ra[u], sA[buf], row, k);
      unit_of(BKC, u, row, k);
      split4(rb[u], sB[buf], row, k);
    }
  };
  if (t0 < t1) {
    gload(t0);
    lstore(0);
    if (t0 + 1 < t1) gload(t0 + 1);
  }
  for (int t = t0; t < t1; ++t) {
    const int cur = (t - t0) & 1;
    __syncthreads();               // buffer `cur` complete; everyone is done reading the other one
    if (t + 1 < t1) lstore(cur ^ 1);
    if (t + 2 < t1) gload(t + 2);
    const unsigned char* pa = sA[cur] + (wm * 32 + i) * GX3_PITCH + kg * 16;
    const unsigned char* pb = sB[cur] + (wn * 32 + i) * GX3_PITCH + kg * 16;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(pa + ks * 32);
      const bf16x8 am = *reinterpret_cast<const bf16x8*>(pa + ks * 32 + PL);
      const bf16x8 al = *reinterpret_cast<const bf16x8*>(pa + ks * 32 + 2 * PL);
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(pb + ks * 32);
      const bf16x8 bm = *reinterpret_cast<const bf16x8*>(pb + ks * 32 + PL);
      const bf16x8 bl = *reinterpret_cast<const bf16x8*>(pb + ks * 32 + 2 * PL);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);      // (smallest terms first)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    }
  }
  const int col = n0 + wn * 32 + i;
  if (S > 1) {
    float* o = ws + (long)z * M * Nn;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * 32 + mfma32_row(r, lane);
      if (row < M && col < Nn) o[(long)row * Nn + col] = acc[r];
    }
    return;
  }
  const float bv = (bias && col < Nn) ? bias[col] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + wm * 32 + mfma32_row(r, lane);
    if (row < M && col < Nn) {
      float* o = Cm + (long)row * scm + col;
      float v = acc[r] + bv;
      if (accumulate) v += *o;
      *o = v;
    }
  }
}

#endif   // IIC_DEBUG_HOOKS (gemm_x3_tiled_kernel)

// 128 x 128 tile form of gemm_f32_tiled_kernel (round 6): four waves, each a 64 x 64 quadrant (2 x 2 MFMA tiles).
// At 64 x 64 a K-tile of 32 is 16 KB of operands for 16 fp32 MFMAs per wave and every A element is re-read by
// ceil(N / 64) column tiles; here a K-tile is 32 KB for 64 MFMAs per wave (half the operand bytes per FLOP, one LDS read
// per MFMA instead of two).  Measured: 3-9 % at the long-K head products of ClusterNet6c k = 280 -- the 64-tile kernel's
// 0.4 of the fp32 MFMA rate is NOT an L2 or matrix-pipe limit (the bf16-split kernel, with a third of the matrix work,
// takes the same time): what is left is the staging path (16-32 bounds-checked scalar loads with 64-bit index
// arithmetic per thread and K-tile, one tile of prefetch).  Launches with too few tiles still split K over gridDim.z
// (fixed-order fold: bit-reproducible).
template <bool AKC, bool BKC>
__global__ __launch_bounds__(256) void gemm_f32_tiled128_kernel(const float* __restrict__ A, long sam, long sak,
                                                                 const float* __restrict__ B, long sbk, long sbn,
                                                                 const float* __restrict__ bias,
                                                                 float* __restrict__ Cm, long scm, int M, int Nn,
                                                                 int K, int accumulate, float* __restrict__ ws) {
  constexpr int TK = 32, LD = 129;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* const sA = reinterpret_cast<float*>(smem_raw);            // [2][TK * LD]
  float* const sB = sA + 2 * TK * LD;                              // [2][TK * LD]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int i = lane & 31, kk = lane >> 5;
  const int m0 = blockIdx.x * 128, n0 = blockIdx.y * 128;
  const int ntiles = (K + TK - 1) / TK;
  const int S = gridDim.z, z = blockIdx.z;
  const int per = (ntiles + S - 1) / S;
  const int t0 = z * per, t1 = min(ntiles, t0 + per);
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float ra[16], rb[16];
  // element u of this thread: k-contiguous operand: k = tid & 31, row = 8 u + (tid >> 5); row-contiguous: row = tid & 127,
  // k = 2 u + (tid >> 7)
  auto rk = [&](bool kc, int u, int& row, int& k) {
    if (kc) { k = tid & 31; row = u * 8 + (tid >> 5); } else { row = tid & 127; k = u * 2 + (tid >> 7); }
  };
  auto gload = [&](int t) {
    const int k0 = t * TK;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      int row, k;
      rk(AKC, u, row, k);
      ra[u] = (m0 + row < M && k0 + k < K) ? A[(long)(m0 + row) * sam + (long)(k0 + k) * sak] : 0.f;
      rk(BKC, u, row, k);
      rb[u] = (n0 + row < Nn && k0 + k < K) ? B[(long)(k0 + k) * sbk + (long)(n0 + row) * sbn] : 0.f;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      int row, k;
      rk(AKC, u, row, k);
      sA[buf * TK * LD + k * LD + row] = ra[u];
      rk(BKC, u, row, k);
      sB[buf * TK * LD + k * LD + row] = rb[u];
    }
  };
  if (t0 < t1) {
    gload(t0);
    lstore(0);
    if (t0 + 1 < t1) gload(t0 + 1);
  }
  for (int t = t0; t < t1; ++t) {
    const int cur = (t - t0) & 1;
    __syncthreads();               // buffer `cur` complete; everyone is done reading the other one
    if (t + 1 < t1) lstore(cur ^ 1);
    if (t + 2 < t1) gload(t + 2);
    const float* pa = sA + cur * TK * LD + wm * 64 + i;
    const float* pb = sB + cur * TK * LD + wn * 64 + i;
#pragma unroll
    for (int k2 = 0; k2 < TK / 2; ++k2) {
      const int kr = (2 * k2 + kk) * LD;
      const float a0 = pa[kr], a1 = pa[kr + 32], b0 = pb[kr], b1 = pb[kr + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int col = n0 + wn * 64 + b * 32 + i;
      if (S > 1) {
        float* o = ws + (long)z * M * Nn;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * 64 + a * 32 + mfma32_row(r, lane);
          if (row < M && col < Nn) o[(long)row * Nn + col] = acc[a][b][r];
        }
      } else {
        const float bv = (bias && col < Nn) ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * 64 + a * 32 + mfma32_row(r, lane);
          if (row < M && col < Nn) {
            float* o = Cm + (long)row * scm + col;
            float v = acc[a][b][r] + bv;
            if (accumulate) v += *o;
            *o = v;
          }
        }
      }
    }
}

__global__ __launch_bounds__(256) void gemm_f32_fold_kernel(const float* __restrict__ ws, int S,
                                                             const float* __restrict__ bias,
                                                             float* __restrict__ Cm, long scm, int M, int Nn,
                                                             int accumulate) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long mn = (long)M * Nn;
  if (idx >= mn) return;
  const int row = (int)(idx / Nn), col = (int)(idx % Nn);
  float v = ws[idx];
  for (int q = 1; q < S; ++q) v += ws[(long)q * mn + idx];
  if (bias) v += bias[col];
  float* o = Cm + (long)row * scm + col;
  if (accumulate) v += *o;
  *o = v;
}

// 1: long-K tiled products on the bf16 pipe through the three-term split (gemm_x3_tiled_kernel); 0: exact-fp32 MFMA.
// Measured (tools/gemm_x3_ab.py, profiles/r06_gemm_ab.txt): same errors against float64 as the fp32 kernel (7e-7), and
// 0.8-1.0 x its time -- at 64 x 64 tiles neither is bound by the matrix pipe: a K-tile is 16 KB of operands for 0.5 us of
// fp32 MFMA work.  What did pay a little: one round of workgroups instead of 1.2 (gemm_tiled_split) and the 128 x 128 tile
// below for the k = 280 products; the split kernel stays in the instrumented library as the record of the experiment.
IIC_SWITCH(g_gemm_x3, 0, iic_debug_gemm_x3)
IIC_SWITCH(g_gemm_t128, 1, iic_debug_gemm_t128)
static bool gemm_tiled_ok(long sam, long sak, long sbk, long sbn, int M, int Nn, int K) {
  return (sak == 1 || sam == 1) && (sbk == 1 || sbn == 1) && K >= 128 && (long)M * Nn >= 64 * 64 * 8;
}
// K-split factor the tiled kernel wants for this product (1 = none): about 1024 blocks, at least 4 K-tiles each
// 128 x 128 tiles (gemm_f32_tiled128_kernel) where both output dimensions are long (measured, profiles/r06_gemm_ab.txt:
// 1.03-1.09 x at the ClusterNet6c k = 280 products, 0.64-0.82 x where one dimension is a short class count)
static bool gemm_tiled_big(int M, int Nn, int K) {
  return g_gemm_t128 && K >= 512 && M >= 512 && Nn >= 512;
}
static int gemm_tiled_split(int M, int Nn, int K) {
  const bool big = gemm_tiled_big(M, Nn, K);
  const int T = big ? 128 : 64;
  const long nt = (long)((M + T - 1) / T) * ((Nn + T - 1) / T);
  const int ktiles = (K + 31) / 32;
  // as many K-split groups as fit ONE round of resident workgroups (two 128-tile workgroups per CU, four 64-tile ones).
  // Round 6: this used to round UP -- [700 x 1400] over 4608 = 242 tiles x 5 groups = 1 210 workgroups on 1 024 slots:
  // a second round for the last 18 %, i.e. twice the time (profiles/r06_gemm_ab.txt)
  int s = (int)((big ? 512 : 1024) / nt);
  if (s > ktiles / 4) s = ktiles / 4;
  if (s > (big ? 8 : 16)) s = big ? 8 : 16;
  return s < 1 ? 1 : s;
}

// one wave per row of k logits
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ logits,
                                                          float* __restrict__ probs, int rows,
                                                          int k) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* x = logits + row * k;
  float m = -INFINITY;
  for (int j = lane; j < k; j += 64) m = fmaxf(m, x[j]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  float s = 0.f;
  for (int j = lane; j < k; j += 64) s += expf(x[j] - m);
  s = wave_sum(s);
  const float inv = 1.f / s;
  for (int j = lane; j < k; j += 64) probs[row * k + j] = expf(x[j] - m) * inv;
}

// dlogits = p * (dp - sum_j dp_j p_j)
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ probs,
                                                          const float* __restrict__ dprobs,
                                                          float* __restrict__ dlogits, int rows,
                                                          int k) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* p = probs + row * k;
  const float* dp = dprobs + row * k;
  float s = 0.f;
  for (int j = lane; j < k; j += 64) s += p[j] * dp[j];
  s = wave_sum(s);
  for (int j = lane; j < k; j += 64) dlogits[row * k + j] = p[j] * (dp[j] - s);
}

// k <= 32: G = next power of two >= k lanes per row, 64 / G rows per wave.  The butterflies run
// over the G lanes of a group only; with the other lanes of the 64-lane version holding 0 / -inf
// that is the same sequence of additions, so the results are bit-identical to the kernels above
// (k = 24: 2 rows per wave, k = 3: 16).
template <int G>
__global__ __launch_bounds__(256) void softmax_fwd_grouped_kernel(const float* __restrict__ logits,
                                                                  float* __restrict__ probs, long rows,
                                                                  int k) {
  constexpr int RPW = 64 / G;
  const int lane = threadIdx.x & 63, j = lane % G;
  const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / G;
  const bool v = row < rows && j < k;
  const float x = v ? logits[row * k + j] : -INFINITY;
  float m = x;
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  const float e = v ? expf(x - m) : 0.f;
  float s = e;
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (v) probs[row * k + j] = e * (1.f / s);
}

template <int G>
__global__ __launch_bounds__(256) void softmax_bwd_grouped_kernel(const float* __restrict__ probs,
                                                                  const float* __restrict__ dprobs,
                                                                  float* __restrict__ dlogits, long rows,
                                                                  int k) {
  constexpr int RPW = 64 / G;
  const int lane = threadIdx.x & 63, j = lane % G;
  const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / G;
  const bool v = row < rows && j < k;
  const float p = v ? probs[row * k + j] : 0.f, dp = v ? dprobs[row * k + j] : 0.f;
  float s = p * dp;
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (v) dlogits[row * k + j] = p * (dp - s);
}

// out[c] (+)= sum_r A[r][c].  A block owns 32 columns; its 8 row groups (r = g, g + 8, ...) are summed
// by 8 x 32 threads -- 128-byte row segments per access -- and folded through LDS in group order (fixed
// order).  (One thread per column walked all rows alone: 114 us for a [700 x 250] matrix.)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ A,
                                                     float* __restrict__ out, int rows, int cols,
                                                     int accumulate) {
  __shared__ float part[8][32];
  const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float s = 0.f;
  if (c < cols)
    for (int r = g; r < rows; r += 8) s += A[(long)r * cols + c];
  part[g][cl] = s;
  __syncthreads();
  if (g == 0 && c < cols) {
    float t = part[0][cl];
#pragma unroll
    for (int q = 1; q < 8; ++q) t += part[q][cl];
    out[c] = accumulate ? out[c] + t : t;
  }
}

extern "C" {

int iic_avgpool_fwd(const void* in_pt, float* feats, int N, int H, int W, int P, int C,
                    void* stream) {
  if (!in_pt || !feats || N <= 0 || (C & 1)) return IIC_ERR_ARG;
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)in_pt, feats, H, W, P, C);
  return iic_launch_status();
}

int iic_avgpool_bwd(const float* dfeats, void* din_pt, int N, int H, int W, int P, int C,
                    const void* mask_act_pt, void* stream) {
  if (!dfeats || !din_pt || N <= 0 || (C & 1)) return IIC_ERR_ARG;
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, dfeats,
                     (bf16_t*)din_pt, (const bf16_t*)mask_act_pt, H, W, P, C);
  return iic_launch_status();
}

long iic_gemm_f32_ws_floats(long sam, long sak, long sbk, long sbn, int M, int Nn, int K) {
  if (M <= 0 || Nn <= 0 || K <= 0 || !gemm_tiled_ok(sam, sak, sbk, sbn, M, Nn, K)) return 0;
  const int S = gemm_tiled_split(M, Nn, K);
  return S > 1 ? (long)S * M * Nn : 0;
}

int iic_gemm_f32_ws(const float* A, long sam, long sak, const float* B, long sbk, long sbn,
                    const float* bias, float* C, long scm, int M, int Nn, int K, int accumulate,
                    float* ws, long ws_floats, void* stream) {
  if (!A || !B || !C || M <= 0 || Nn <= 0 || K <= 0) return IIC_ERR_ARG;
  // LDS-tiled kernel wherever an operand's unit stride allows coalesced tile loads and the product is big
  // enough to matter (the small sub-head GEMMs of ClusterNet5g stay on the one-wave kernels); without a
  // workspace no K split
  if (gemm_tiled_ok(sam, sak, sbk, sbn, M, Nn, K)) {
    int S = ws ? gemm_tiled_split(M, Nn, K) : 1;
    if (S > 1 && (long)S * M * Nn > ws_floats) S = (int)(ws_floats / ((long)M * Nn));
    if (S < 1) S = 1;
    const bool big = gemm_tiled_big(M, Nn, K);
    const int T = big ? 128 : 64;
    dim3 tg((M + T - 1) / T, (Nn + T - 1) / T, S);
    const size_t lds128 = (size_t)4 * 32 * 129 * sizeof(float);
    if (big) {
      static bool attr = false;
      if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_tiled128_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds128);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_tiled128_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds128);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_tiled128_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds128);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_tiled128_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds128);
        attr = true;
      }
    }
#ifdef IIC_DEBUG_HOOKS
#define GT_X3(AK_, BK_)                                                                              \
      if (g_gemm_x3 && K >= 512 && !big)                                                             \
        hipLaunchKernelGGL((gemm_x3_tiled_kernel<AK_, BK_>), tg, dim3(256), 0, (hipStream_t)stream,  \
                           A, sam, sak, B, sbk, sbn, bias, C, scm, M, Nn, K, accumulate, ws);        \
      else
#else
#define GT_X3(AK_, BK_)
#endif
#define GT_LAUNCH(AK_, BK_)                                                                          \
    do {                                                                                             \
      GT_X3(AK_, BK_)                                                                                \
      if (big)                                                                                       \
        hipLaunchKernelGGL((gemm_f32_tiled128_kernel<AK_, BK_>), tg, dim3(256), lds128, (hipStream_t)stream, \
                           A, sam, sak, B, sbk, sbn, bias, C, scm, M, Nn, K, accumulate, ws);        \
      else                                                                                           \
        hipLaunchKernelGGL((gemm_f32_tiled_kernel<AK_, BK_>), tg, dim3(256), 0, (hipStream_t)stream, \
                           A, sam, sak, B, sbk, sbn, bias, C, scm, M, Nn, K, accumulate, ws);        \
    } while (0)
    if (sak == 1) { if (sbk == 1) GT_LAUNCH(true, true); else GT_LAUNCH(true, false); }
    else { if (sbk == 1) GT_LAUNCH(false, true); else GT_LAUNCH(false, false); }
    if (S > 1) {
      const long mn = (long)M * Nn;
      hipLaunchKernelGGL(gemm_f32_fold_kernel, dim3((unsigned)((mn + 255) / 256)), dim3(256), 0,
                         (hipStream_t)stream, ws, S, bias, C, scm, M, Nn, accumulate);
    }
    return iic_launch_status();
  }
  dim3 grid((M + 31) / 32, (Nn + 31) / 32);
  const long tiles = (long)grid.x * grid.y;
  // long K and too few tiles to fill 1024 SIMDs: split K over the waves of a block
  if (K >= 1024 && tiles * 8 <= 2048)
    hipLaunchKernelGGL((gemm_f32_kw_kernel<8>), grid, dim3(512), 0, (hipStream_t)stream, A, sam, sak, B, sbk, sbn,
                       bias, C, scm, M, Nn, K, accumulate);
  else if (K >= 512 && tiles * 4 <= 2048)
    hipLaunchKernelGGL((gemm_f32_kw_kernel<4>), grid, dim3(256), 0, (hipStream_t)stream, A, sam, sak, B, sbk, sbn,
                       bias, C, scm, M, Nn, K, accumulate);
  else
    hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(64), 0, (hipStream_t)stream, A, sam, sak, B, sbk,
                       sbn, bias, C, scm, M, Nn, K, accumulate);
  return iic_launch_status();
}

int iic_gemm_f32(const float* A, long sam, long sak, const float* B, long sbk, long sbn,
                 const float* bias, float* C, long scm, int M, int Nn, int K, int accumulate,
                 void* stream) {
  return iic_gemm_f32_ws(A, sam, sak, B, sbk, sbn, bias, C, scm, M, Nn, K, accumulate, nullptr, 0, stream);
}

/* split-K variant: C must be zero-initialised (or hold the value to accumulate onto); the K
 * range is cut into `splitk` slices whose partial products are added atomically. */
int iic_gemm_f32_splitk(const float* A, long sam, long sak, const float* B, long sbk, long sbn,
                        float* C, long scm, int M, int Nn, int K, int splitk, void* stream) {
  if (!A || !B || !C || M <= 0 || Nn <= 0 || K <= 0 || splitk < 1) return IIC_ERR_ARG;
  dim3 grid((M + 31) / 32, (Nn + 31) / 32, splitk);
  hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(64), 0, (hipStream_t)stream, A, sam, sak, B, sbk,
                     sbn, (const float*)nullptr, C, scm, M, Nn, K, 1);
  return iic_launch_status();
}

int iic_softmax_fwd(const float* logits, float* probs, int rows, int k, void* stream) {
  if (!logits || !probs || rows <= 0 || k <= 0) return IIC_ERR_ARG;
#define SMF(G_)                                                                                   \
  hipLaunchKernelGGL((softmax_fwd_grouped_kernel<G_>), dim3((rows + 4 * (64 / G_) - 1) / (4 * (64 / G_))), \
                     dim3(256), 0, (hipStream_t)stream, logits, probs, (long)rows, k)
  if (k <= 4) SMF(4); else if (k <= 8) SMF(8); else if (k <= 16) SMF(16); else if (k <= 32) SMF(32);
  else
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       logits, probs, rows, k);
  return iic_launch_status();
}

int iic_softmax_bwd(const float* probs, const float* dprobs, float* dlogits, int rows, int k,
                    void* stream) {
  if (!probs || !dprobs || !dlogits || rows <= 0 || k <= 0) return IIC_ERR_ARG;
#define SMB(G_)                                                                                   \
  hipLaunchKernelGGL((softmax_bwd_grouped_kernel<G_>), dim3((rows + 4 * (64 / G_) - 1) / (4 * (64 / G_))), \
                     dim3(256), 0, (hipStream_t)stream, probs, dprobs, dlogits, (long)rows, k)
  if (k <= 4) SMB(4); else if (k <= 8) SMB(8); else if (k <= 16) SMB(16); else if (k <= 32) SMB(32);
  else
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       probs, dprobs, dlogits, rows, k);
  return iic_launch_status();
}

int iic_colsum_f32(const float* A, float* out, int rows, int cols, int accumulate, void* stream) {
  if (!A || !out || rows <= 0 || cols <= 0) return IIC_ERR_ARG;
  hipLaunchKernelGGL(colsum_kernel, dim3((cols + 31) / 32), dim3(256), 0, (hipStream_t)stream, A,
                     out, rows, cols, accumulate);
  return iic_launch_status();
}

}  // extern "C"

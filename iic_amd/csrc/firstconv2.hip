// First-layer convolution of the VGG-style trunks, second generation (round 6): LDS-staged input bands and
// 16-byte global accesses.  Replaces -- like firstconv_fwd_kernel / firstconv_wgrad_kernel of vgg.hip, which stay
// as the path for row widths that are not a multiple of 4 -- the first nn.Conv2d of
// /root/reference/code/archs/cluster/vgg.py:24-26 as configured by net6c.py:16-20 (5x5, pad 2) and net10a.py:21-25
// (3x3, pad 1): fp32 NCHW image in (Cin <= 8), bf16 PT tensor [N][H+2P][W+2P][64] out, exact-fp32 MFMA.
//
// Why: the first generation fetched every MFMA operand element with its own 4-byte global load (18 dependent loads,
// 36 table reads per 32-pixel tile; in the weight gradient a 32-address gather per load because a lane's k index
// selects a different tap), and stored the tile with 32 two-byte stores per lane: 568 / 941 us per launch at
// Potsdam-3 (75 x 4 x 200 x 200) against an fp32-MFMA floor of 100 / 120 us and an HBM floor of 90 us (VERDICT r5
// weak #7, next #7).
//
// Here a workgroup walks BANDS of R output rows of one image:
//   * the band's input rows (R + K - 1 rows x W + 2 pad columns x Cin planes, zero halo) live in LDS, fetched with
//     float4 loads one band ahead (registers) -- every operand element of the band is then an LDS read at
//     table[k] + pixel offset, no bounds checks in the loop;
//   * a tile is 32 (16 in the weight gradient's k dimension) CONSECUTIVE pixels of the band in row-major order, so a
//     24-pixel-wide image (MNIST, CIFAR) fills its tiles (the row-segment tiling of the first generation idled a
//     quarter of every MFMA there);
//   * forward: D[cout][pixel] = W[cout][k] patch[k][pixel] (v_mfma_f32_32x32x2_f32) leaves 4 consecutive couts of one
//     pixel in a lane: packed to 8 bytes, written to a per-wave LDS tile and stored to HBM as 16-byte units (4
//     instructions per tile instead of 32); BatchNorm statistics from the fp32 accumulators, carried in registers
//     to the end of the kernel;
//   * weight gradient: dW[cout][k] = sum_pix dy[pix][cout] patch[pix][k] on v_mfma_f32_16x16x4_f32 -- 16-wide k tiles
//     (K = 36 pads to 48, not 64) -- dy tiles fetched with 16-byte loads into a per-wave LDS tile (pitch 160 B:
//     the four pixels of a k-step on disjoint banks).
// Bit-reproducible: fixed tile -> wave assignment, fixed summation order, statistics through the exact accumulators.
#include "common.h"
#include "../../include/iic_hip.h"

#define FC2_CO 64
#define FC2_NPF_MAX 8         // float4 prefetch units per thread (Cin * (R + K - 1) * W / 4 <= 256 * NPF; NPF = 2 | 4 | 8)
#define FC2_OPITCH 136        // forward output tile: bytes per pixel row in LDS (8 B = 2 banks past 128: 16 consecutive lanes
                              // -- 16 pixels of one k half -- write their 8-byte units to 32 distinct banks)
#define FC2_DPITCH 160        // weight gradient dy tile: bytes per pixel row in LDS

struct fc2_args {
  int N, Cin, H, W, K, pad, P;
  int R;                      // output rows per band
  int RK;                     // R + K - 1
  int PW;                     // patch row pitch in floats (>= W + 2 pad)
  int KT;                     // Cin * K * K
  int bpi;                    // bands per image
  int nunits;                 // float4 units of a band's input rows
  iic_mdiv divW;              // pixel -> (row, column)
  iic_mdiv divW4;             // prefetch unit -> (plane row, float4 column)
  iic_mdiv divRK;             // plane row -> (plane, row)
};

__device__ __forceinline__ float4 fc2_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// LDS layout helpers (all offsets in bytes, 16-byte aligned)
__host__ __device__ __forceinline__ int fc2_align16(int b) { return (b + 15) & ~15; }

// The band pipeline shared by both kernels: per-thread prefetch slots (band-independent decomposition of the
// unit index), issue = global float4 loads of band b into registers, publish = registers -> LDS patch.
template <int NPF>
struct Fc2Prefetch {
  float4 v[NPF];
};

// unit u of a band -> (offset inside the image, LDS float offset, row relative to the band's first output row)
__device__ __forceinline__ void fc2_unit(const fc2_args& a, int u, int& goff, int& loff, int& rr) {
  const int W4 = a.W >> 2;
  const int cr = iic_mdivide(u, a.divW4), x4 = u - cr * W4;
  const int c = iic_mdivide(cr, a.divRK), r = cr - c * a.RK;
  goff = c * a.H * a.W + (r - a.pad) * a.W + 4 * x4;
  loff = cr * a.PW + a.pad + 4 * x4;
  rr = r - a.pad;
}

template <int NPF>
__device__ __forceinline__ void fc2_issue(Fc2Prefetch<NPF>& pf, const fc2_args& a, const float* __restrict__ x, long band) {
  const int n = (int)(band / a.bpi), y0 = (int)(band - (long)n * a.bpi) * a.R;
  const float* base = x + ((long)n * a.Cin * a.H + y0) * a.W;
#pragma unroll
  for (int j = 0; j < NPF; ++j) {
    const int u = threadIdx.x + 256 * j;
    if (u < a.nunits) {
      int goff, loff, rr;
      fc2_unit(a, u, goff, loff, rr);
      const int yy = y0 + rr;
      pf.v[j] = (yy >= 0 && yy < a.H) ? fc2_ld4(base + goff) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

template <int NPF>
__device__ __forceinline__ void fc2_publish(const Fc2Prefetch<NPF>& pf, const fc2_args& a, float* sPatch) {
#pragma unroll
  for (int j = 0; j < NPF; ++j) {
    const int u = threadIdx.x + 256 * j;
    if (u < a.nunits) {
      int goff, loff, rr;
      fc2_unit(a, u, goff, loff, rr);
      float* d = sPatch + loff;
      d[0] = pf.v[j].x; d[1] = pf.v[j].y; d[2] = pf.v[j].z; d[3] = pf.v[j].w;
    }
  }
}

// ------------------------------------------------------------------------------------
// forward.  LDS: sW [KT2][64] f32 | sTab [KT2] int | sPatch [Cin][RK][PW] f32 | sOut [4 waves][32][FC2_OPITCH] | s_red
// ------------------------------------------------------------------------------------
// KSC > 0: the k-step count is a compile-time constant (K*K*Cin = 25 / 36 / 45: the MNIST, Potsdam and COCO first
// layers) -- the wave's weight operands (2 per k-step) and patch offsets then live in REGISTERS for the whole kernel
// and a tile's K loop is one LDS read + two MFMAs per step, with the next tile's patch values read under the current
// tile's MFMAs.  KSC == 0: any K*K*Cin <= 128, weights and offsets read from LDS per step.
template <int KSC, int NPF>
__global__ __launch_bounds__(256, 2) void firstconv_fwd2_kernel(const fc2_args a, const float* __restrict__ x,
                                                             const float* __restrict__ w, bf16_t* __restrict__ out,
                                                             float* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int KT = a.KT, KT2 = (KT + 1) & ~1, KS = KT2 >> 1;
  float* const sW = reinterpret_cast<float*>(smem_raw);
  int* const sTab = reinterpret_cast<int*>(sW + KT2 * FC2_CO);
  float* const sPatch = reinterpret_cast<float*>(smem_raw + fc2_align16(KT2 * FC2_CO * 4 + KT2 * 4));
  const int patch_floats = a.Cin * a.RK * a.PW;
  unsigned char* const sOut = smem_raw + fc2_align16(KT2 * FC2_CO * 4 + KT2 * 4) + fc2_align16(patch_floats * 4);
  float* const s_red = reinterpret_cast<float*>(sOut + 4 * 32 * FC2_OPITCH);      // [4 waves][2 stats][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, kk = lane >> 5;
  const int KK = a.K * a.K;
  for (int idx = tid; idx < KT2 * FC2_CO; idx += 256) {
    const int k = idx >> 6, co = idx & 63;
    sW[idx] = k < KT ? w[co * KT + k] : 0.f;
  }
  for (int k = tid; k < KT2; k += 256) {
    int off = 0;
    if (k < KT) {
      const int c = k / KK, r = k - c * KK, kh = r / a.K, kw = r - kh * a.K;
      off = (c * a.RK + kh) * a.PW + kw;
    }
    sTab[k] = off;
  }
  for (int idx = tid; idx < patch_floats; idx += 256) sPatch[idx] = 0.f;     // halo columns stay zero for good

  Fc2Prefetch<NPF> pf;
  const long nb = (long)a.N * a.bpi;
  const int Hp = a.H + 2 * a.P, Wp = a.W + 2 * a.P;
  constexpr int KR = KSC > 0 ? KSC : 1;
  constexpr bool WREG = KSC > 0 && KSC <= 18;      // (46 weight registers at K*K*Cin = 45 would spill: LDS reads there)
  constexpr int KW = WREG ? KSC : 1;
  float wr0[KW], wr1[KW], bvn[KR], bvm[KR];
  int so0[KR], so1[KR];       // patch offsets of the k-step's two k indices (wave-uniform: scalar registers)
  if (KSC > 0) {
    __syncthreads();
#pragma unroll
    for (int st = 0; st < KR; ++st) {
      const int k = 2 * st + kk;
      if (WREG) {
        wr0[st] = sW[k * FC2_CO + i];
        wr1[st] = sW[k * FC2_CO + 32 + i];
      }
      so0[st] = __builtin_amdgcn_readfirstlane(sTab[2 * st]);
      so1[st] = __builtin_amdgcn_readfirstlane(sTab[2 * st + 1]);
      bvn[st] = bvm[st] = 0.f;
    }
  }
  float s[2][16], ss[2][16];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 16; ++r) s[h][r] = ss[h][r] = 0.f;
  unsigned char* const myOut = sOut + wave * 32 * FC2_OPITCH;

  // one tile: K loop on `cur` (this tile's patch values, KSC > 0) while `nxt` is filled for tile t + 4, then the
  // epilogue.  Two waves share a SIMD (one of each resident workgroup) and run the same phases at the same time:
  // what is not an MFMA is paid twice per pair of tiles, so the epilogue is kept lean -- one divide per tile, 32-bit
  // offsets from a per-band base, lanes past the band end cleared only in the band's last tile.
  long b = blockIdx.x;
  if (b < nb) fc2_issue(pf, a, x, b);
  for (; b < nb; b += gridDim.x) {
    __syncthreads();                       // the previous band's tiles are done with the patch
    fc2_publish(pf, a, sPatch);
    __syncthreads();
    if (b + gridDim.x < nb) fc2_issue(pf, a, x, b + gridDim.x);      // in flight under this band's MFMAs
    const int n = (int)(b / a.bpi), y0 = (int)(b - (long)n * a.bpi) * a.R;
    const int rows = min(a.R, a.H - y0), npix = rows * a.W, ntiles = (npix + 31) >> 5;
    bf16_t* const outb = out + (((long)n * Hp + y0 + a.P) * Wp + a.P) * FC2_CO;      // pixel (y0, 0) of image n
    auto tile_ptr = [&](int t) {
      const int pc = min(t * 32 + i, npix - 1);
      const int ry = iic_mdivide(pc, a.divW), xx = pc - ry * a.W;
      return sPatch + ry * a.PW + xx;
    };
    auto tile = [&](int t, float (&cur)[KR], float (&nxt)[KR]) {
      const float* pb = tile_ptr(t);
      f32x16 acc[2];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[h][r] = 0.f;
      if (KSC > 0) {
        if (t + 4 < ntiles) {              // the next tile's patch values: in flight under this tile's MFMAs
          const float* pbn = tile_ptr(t + 4);
#pragma unroll
          for (int st = 0; st < KR; ++st) nxt[st] = pbn[kk ? so1[st] : so0[st]];
        }
#pragma unroll
        for (int st = 0; st < KR; ++st) {
          const float w0 = WREG ? wr0[st] : sW[(2 * st + kk) * FC2_CO + i];
          const float w1 = WREG ? wr1[st] : sW[(2 * st + kk) * FC2_CO + 32 + i];
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, cur[st], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, cur[st], acc[1], 0, 0, 0);
        }
      }
      // KSC == 0: operands of 6 k-steps are read before their 12 MFMAs (table entry -> patch element is a dependent
      // pair of LDS reads: one pair of latencies per 6 steps instead of per step)
      for (int sb = 0; KSC == 0 && sb < KS; sb += 6) {
        float bv[6], a0[6], a1[6];
#pragma unroll
        for (int s6 = 0; s6 < 6; ++s6) {
          const int k = min(2 * (sb + s6) + kk, KT2 - 1);
          bv[s6] = pb[sTab[k]];
          a0[s6] = sW[k * FC2_CO + i];
          a1[s6] = sW[k * FC2_CO + 32 + i];
        }
#pragma unroll
        for (int s6 = 0; s6 < 6; ++s6) {
          if (sb + s6 < KS) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s6], bv[s6], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s6], bv[s6], acc[1], 0, 0, 0);
          }
        }
      }
      // D[cout = h*32 + 8*(r/4) + 4*kk + r%4][pixel i]
      if (t * 32 + 32 > npix) {   // (the band's last tile: pixels past its end were computed on a clamped address)
        const bool valid = t * 32 + i < npix;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[h][r] = valid ? acc[h][r] : 0.f;
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {       // (pairs: v_pk_add_f32 / v_pk_fma_f32)
          const f32x2 v = {acc[h][r], acc[h][r + 1]};
          f32x2 sv = {s[h][r], s[h][r + 1]}, sq = {ss[h][r], ss[h][r + 1]};
          sv += v;
          sq = __builtin_elementwise_fma(v, v, sq);
          s[h][r] = sv[0]; s[h][r + 1] = sv[1];
          ss[h][r] = sq[0]; ss[h][r + 1] = sq[1];
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 pk;
          pk.x = pack_bf16x2(acc[h][4 * g], acc[h][4 * g + 1]);
          pk.y = pack_bf16x2(acc[h][4 * g + 2], acc[h][4 * g + 3]);
          *reinterpret_cast<uint2*>(myOut + i * FC2_OPITCH + (h * 32 + 8 * g + 4 * kk) * 2) = pk;
        }
      }
      // (the tile buffer is private to this wave: LDS executes a wave's accesses in order)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // this lane stores 16-byte unit ch of pixels px0, px0 + 8, px0 + 16, px0 + 24 of the tile
      const int px0 = lane >> 3, ch = lane & 7;
      uint2 v0[4], v1[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v0[q] = *reinterpret_cast<const uint2*>(myOut + (px0 + 8 * q) * FC2_OPITCH + ch * 16);
        v1[q] = *reinterpret_cast<const uint2*>(myOut + (px0 + 8 * q) * FC2_OPITCH + ch * 16 + 8);
      }
      int pp = t * 32 + px0;
      int py = iic_mdivide(min(pp, npix - 1), a.divW), pxx = min(pp, npix - 1) - py * a.W;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (pp < npix)
          *reinterpret_cast<uint4*>(outb + (py * Wp + pxx) * FC2_CO + ch * 8) = make_uint4(v0[q].x, v0[q].y, v1[q].x, v1[q].y);
        pp += 8;
        pxx += 8;
        while (pxx >= a.W) { pxx -= a.W; ++py; }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads done before the next tile's writes
    };
    if (KSC > 0 && wave < ntiles) {
      const float* pb0 = tile_ptr(wave);
#pragma unroll
      for (int st = 0; st < KR; ++st) bvn[st] = pb0[kk ? so1[st] : so0[st]];
    }
    for (int t = wave; t < ntiles; t += 8) {     // (two tiles per trip: the operand buffers swap roles)
      tile(t, bvn, bvm);
      if (t + 4 < ntiles) tile(t + 4, bvm, bvn);
    }
  }
  if (stats) {
    // sums over the 32 pixel lanes of each half-wave; lane (0, kk) then holds couts h*32 + 8*(r/4) + 4*kk + r%4
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v0 = s[h][r], v1 = ss[h][r];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
          v0 += __shfl_xor(v0, o, 64);
          v1 += __shfl_xor(v1, o, 64);
        }
        if (i == 0) {
          const int co = h * 32 + 8 * (r >> 2) + 4 * kk + (r & 3);
          s_red[(wave * 2 + 0) * 64 + co] = v0;
          s_red[(wave * 2 + 1) * 64 + co] = v1;
        }
      }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, ch = tid & 63;
      float t = 0.f;
      for (int wv = 0; wv < 4; ++wv) t += s_red[(wv * 2 + which) * 64 + ch];
      iic_stat_add(stats, blockIdx.x % IIC_STAT_STRIPES, FC2_CO, ch, which, t);
    }
  }
}

// ------------------------------------------------------------------------------------
// weight gradient.  part[block][co][LD], LD = NKT * 16.
// LDS: sTab [NKT*16] int | sPatch | sDy [4 waves][32][FC2_DPITCH]   (the cross-wave reduction reuses the space)
// ------------------------------------------------------------------------------------
template <int NKT, int NPF>
__global__ __launch_bounds__(256) void firstconv_wgrad2_kernel(const fc2_args a, const float* __restrict__ x,
                                                               const bf16_t* __restrict__ dy, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int LD = NKT * 16;
  int* const sTab = reinterpret_cast<int*>(smem_raw);
  float* const sPatch = reinterpret_cast<float*>(smem_raw + fc2_align16(LD * 4));
  const int patch_floats = a.Cin * a.RK * a.PW;
  unsigned char* const sDy = smem_raw + fc2_align16(LD * 4) + fc2_align16(patch_floats * 4);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kq = lane >> 4;
  const int KK = a.K * a.K;
  for (int k = tid; k < LD; k += 256) {
    int off = 0;
    if (k < a.KT) {
      const int c = k / KK, r = k - c * KK, kh = r / a.K, kw = r - kh * a.K;
      off = (c * a.RK + kh) * a.PW + kw;
    }
    sTab[k] = off;
  }
  for (int idx = tid; idx < patch_floats; idx += 256) sPatch[idx] = 0.f;
  __syncthreads();
  int koff[NKT];
  float kmask[NKT];           // columns past KT read patch element 0 (finite) and are zeroed here
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    koff[t] = sTab[t * 16 + i16];
    kmask[t] = (t * 16 + i16) < a.KT ? 1.f : 0.f;
  }
  Fc2Prefetch<NPF> pf;
  const long nb = (long)a.N * a.bpi;
  const int Hp = a.H + 2 * a.P, Wp = a.W + 2 * a.P;
  f32x4 acc[4][NKT];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int t = 0; t < NKT; ++t) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  unsigned char* const myDy = sDy + wave * 32 * FC2_DPITCH;

  long b = blockIdx.x;
  if (b < nb) fc2_issue(pf, a, x, b);
  for (; b < nb; b += gridDim.x) {
    __syncthreads();
    fc2_publish(pf, a, sPatch);
    __syncthreads();
    if (b + gridDim.x < nb) fc2_issue(pf, a, x, b + gridDim.x);
    const int n = (int)(b / a.bpi), y0 = (int)(b - (long)n * a.bpi) * a.R;
    const int rows = min(a.R, a.H - y0), npix = rows * a.W, ntiles = (npix + 31) >> 5;
    for (int t = wave; t < ntiles; t += 4) {
      // dy rows of the tile's 32 pixels -> LDS (16-byte units; pixels past the band end are zero rows)
      uint4 dv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int u = q * 64 + lane, px = u >> 3, ch = u & 7;
        const int pp = t * 32 + px;
        dv[q] = make_uint4(0u, 0u, 0u, 0u);
        if (pp < npix) {
          const int py = iic_mdivide(pp, a.divW), pxx = pp - py * a.W;
          dv[q] = *reinterpret_cast<const uint4*>(dy + (((long)n * Hp + y0 + py + a.P) * Wp + pxx + a.P) * FC2_CO + ch * 8);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int u = q * 64 + lane, px = u >> 3, ch = u & 7;
        *reinterpret_cast<uint4*>(myDy + px * FC2_DPITCH + ch * 16) = dv[q];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // this lane's pixel of k-step 0: p = t*32 + kq, then +4 per step (clamped: its dy row is zero past the end)
      int p = t * 32 + kq;
      int pc = min(p, npix - 1);
      int ry = iic_mdivide(pc, a.divW), xx = pc - ry * a.W;
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        const float* pb = sPatch + ry * a.PW + xx;
        const unsigned char* drow = myDy + (4 * st + kq) * FC2_DPITCH + i16 * 2;
        float av[4], bv[NKT];
#pragma unroll
        for (int c = 0; c < 4; ++c) av[c] = bf16_to_f32(*reinterpret_cast<const bf16_t*>(drow + c * 32));
#pragma unroll
        for (int tt = 0; tt < NKT; ++tt) bv[tt] = pb[koff[tt]] * kmask[tt];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int tt = 0; tt < NKT; ++tt) acc[c][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c], bv[tt], acc[c][tt], 0, 0, 0);
        // advance the pixel walker by 4 (W >= 4)
        p += 4;
        if (p < npix) {
          xx += 4;
          if (xx >= a.W) { xx -= a.W; ++ry; }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  // cross-wave reduction through LDS in two passes of 32 couts: red[4 waves][32][LD]
  // D[co = c*16 + 4*kq + r][k = tt*16 + i16]
  float* const red = reinterpret_cast<float*>(smem_raw);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __syncthreads();
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
      for (int tt = 0; tt < NKT; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          red[((long)wave * 32 + c2 * 16 + 4 * kq + r) * LD + tt * 16 + i16] = acc[half * 2 + c2][tt][r];
    __syncthreads();
    for (int idx = tid; idx < 32 * LD; idx += 256) {
      float t = 0.f;
      for (int wv = 0; wv < 4; ++wv) t += red[(long)wv * 32 * LD + idx];
      part[((long)blockIdx.x * 64 + half * 32) * LD + idx] = t;
    }
  }
}

IIC_SWITCH(g_fc2_enabled, 1, iic_debug_firstconv_v2)     // 0: first-generation kernels (vgg.hip)

// Band height: the candidate that needs the fewest rounds of 4 tiles (one per wave) per image among those whose input
// rows fit the prefetch slots and leave room for two workgroups per CU.
static int fc2_setup(fc2_args* a, const void* p0, const void* p1, int N, int Cin, int H, int W, int K, int pad, int P,
                     size_t extra_lds, size_t* patch_bytes, int* npf) {
  if (!g_fc2_enabled || (W & 3) || W < 4 || Cin * K * K > 128) return 0;
  if ((long)(H + 2 * P) * (W + 2 * P) * FC2_CO >= (1L << 31)) return 0;      // (32-bit pixel offsets inside an image)
  if ((((uintptr_t)p0) | ((uintptr_t)p1)) & 15) return 0;
  a->N = N; a->Cin = Cin; a->H = H; a->W = W; a->K = K; a->pad = pad; a->P = P;
  a->KT = Cin * K * K;
  a->PW = W + 2 * pad + 1;            // (+1: consecutive patch rows start on different banks)
  static const int cand[] = {1, 2, 4, 8, 12, 16};      // (ties: the smaller band -- less LDS, fewer prefetch registers)
  long best = -1;
  for (int ci = 0; ci < 6; ++ci) {
    const int R = cand[ci], RK = R + K - 1;
    if (R > H && R != 1) continue;
    const long units = (long)Cin * RK * (W >> 2);
    const size_t patch = (size_t)Cin * RK * a->PW * 4;
    if (units > 256L * FC2_NPF_MAX || patch + extra_lds > 72 * 1024) continue;
    if (R > 1 && (long)N * ((H + R - 1) / R) < 2048 && R > 4) continue;      // keep enough bands to balance ~512 resident workgroups
    long rounds = 0;
    for (int y0 = 0; y0 < H; y0 += R) {
      const int rows = H - y0 < R ? H - y0 : R;
      rounds += ((rows * W + 31) / 32 + 3) / 4;
    }
    if (best < 0 || rounds < best) {
      best = rounds;
      a->R = R; a->RK = RK;
      a->bpi = (H + R - 1) / R;
      a->nunits = (int)units;
      *patch_bytes = patch;
    }
  }
  if (best < 0) return 0;
  a->divW = iic_make_mdiv(W);
  a->divW4 = iic_make_mdiv(W >> 2);
  a->divRK = iic_make_mdiv(a->RK);
  const int per = (a->nunits + 255) / 256;
  *npf = per <= 2 ? 2 : (per <= 6 ? per : 8);
  return 1;
}

#define FC2_ATTR(KERNEL)                                                                          \
  do {                                                                                            \
    static bool attr = false;                                                                     \
    if (!attr) {                                                                                  \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&KERNEL),                           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);          \
      attr = true;                                                                                \
    }                                                                                             \
  } while (0)

// IIC_ERR_UNSUPPORTED: the caller (vgg.hip) runs the first-generation kernel
int fc2_fwd_launch(const float* x, const float* w, void* out_pt, float* stats, int N, int Cin, int H, int W, int K,
                   int pad, int P, int max_blocks, void* stream) {
  fc2_args a;
  const int KT2 = (Cin * K * K + 1) & ~1;
  const size_t front = (size_t)fc2_align16(KT2 * FC2_CO * 4 + KT2 * 4);
  const size_t tail = (size_t)4 * 32 * FC2_OPITCH + 4 * 2 * 64 * 4;
  size_t patch = 0;
  int npf = 0;
  if (!fc2_setup(&a, x, out_pt, N, Cin, H, W, K, pad, P, front + tail, &patch, &npf)) return IIC_ERR_UNSUPPORTED;
  const size_t lds = front + fc2_align16((int)patch) + tail;
  const long nb = (long)N * a.bpi;
  // two workgroups per CU are resident (256 registers per lane): more workgroups would only queue behind them and pay
  // the set-up (weights -> LDS -> registers, first band's load latency) a second time
  if (max_blocks > 512) max_blocks = 512;
  const int grid = (int)(nb < max_blocks ? nb : max_blocks);
  hipStream_t s = (hipStream_t)stream;
#define FC2F2(KS_, NPF_)                                                                          \
  do {                                                                                            \
    FC2_ATTR((firstconv_fwd2_kernel<KS_, NPF_>));                                                 \
    hipLaunchKernelGGL((firstconv_fwd2_kernel<KS_, NPF_>), dim3(grid), dim3(256), lds, s, a, x, w, \
                       (bf16_t*)out_pt, stats);                                                   \
  } while (0)
#define FC2F(KS_)                                                                                 \
  do {                                                                                            \
    if (npf == 2) FC2F2(KS_, 2); else if (npf <= 4) FC2F2(KS_, 4); else if (npf == 5) FC2F2(KS_, 5); \
    else if (npf == 6) FC2F2(KS_, 6); else FC2F2(KS_, 8);                                         \
  } while (0)
  const int ks = g_fc2_enabled == 2 ? 0 : KT2 / 2;       // (iic_debug_firstconv_v2(2): the generic variant everywhere)
  switch (ks) {
    case 13: FC2F(13); break;       // 1 x 5 x 5   (MNIST)
    case 18: FC2F(18); break;       // 4 x 3 x 3   (Potsdam)
    case 23: FC2F(23); break;       // 5 x 3 x 3   (COCO-Stuff: RGB + Sobel)
    default: FC2F(0); break;
  }
#undef FC2F
#undef FC2F2
  return iic_launch_status();
}

// *grid_out workgroups wrote partials [block][64][*ld_out]; the caller folds them (fc_wgrad_reduce_kernel, vgg.hip)
int fc2_wgrad_launch(const float* x, const void* dy_pt, float* partials, int N, int Cin, int H, int W, int K, int pad,
                     int P, int max_blocks, int* grid_out, int* ld_out, void* stream) {
  fc2_args a;
  const int KT = Cin * K * K, NKT = (KT + 15) / 16, LD = NKT * 16;
  const size_t front = (size_t)fc2_align16(LD * 4);
  const size_t tail = (size_t)4 * 32 * FC2_DPITCH;
  size_t patch = 0;
  int npf = 0;
  if (!fc2_setup(&a, x, dy_pt, N, Cin, H, W, K, pad, P, front + tail, &patch, &npf)) return IIC_ERR_UNSUPPORTED;
  size_t lds = front + fc2_align16((int)patch) + tail;
  const size_t red = (size_t)4 * 32 * LD * 4;
  if (red > lds) lds = red;
  const long nb = (long)N * a.bpi;
  const int grid = (int)(nb < max_blocks ? nb : max_blocks);
  hipStream_t s = (hipStream_t)stream;
#define FC2W2(NK, NPF_)                                                                           \
  do {                                                                                            \
    FC2_ATTR((firstconv_wgrad2_kernel<NK, NPF_>));                                                \
    hipLaunchKernelGGL((firstconv_wgrad2_kernel<NK, NPF_>), dim3(grid), dim3(256), lds, s, a, x,  \
                       (const bf16_t*)dy_pt, partials);                                           \
  } while (0)
#define FC2W(NK)                                                                                  \
  do {                                                                                            \
    if (npf == 2) FC2W2(NK, 2); else if (npf <= 4) FC2W2(NK, 4); else if (npf <= 6) FC2W2(NK, 6); \
    else FC2W2(NK, 8);                                                                            \
  } while (0)
  switch (NKT) {
    case 1: FC2W(1); break;
    case 2: FC2W(2); break;
    case 3: FC2W(3); break;
    case 4: FC2W(4); break;
    case 5: FC2W(5); break;
    case 6: FC2W(6); break;
    case 7: FC2W(7); break;
    case 8: FC2W(8); break;
    default: return IIC_ERR_UNSUPPORTED;
  }
#undef FC2W
#undef FC2W2
  *grid_out = grid;
  *ld_out = LD;
  return iic_launch_status();
}

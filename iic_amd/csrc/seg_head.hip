// SegmentationNet10a head pieces for gfx950 (fp32): the 1x1 conv with padding = 1 is a plain
// GEMM over the (Hf+2) x (Wf+2) window of the PT feature tensor (its first border ring IS the
// conv's zero padding), followed by Softmax2d and bilinear up-sampling to input_sz.
//
// Replaces /root/reference/code/archs/segmentation/net10a.py:44-59
// (nn.Conv2d(1x1, padding=1, bias=False) + nn.Softmax2d + F.interpolate(bilinear)).
//   seg_window_gather : PT bf16 window -> fp32 [M][C] matrix (A operand of iic_gemm_f32)
//   seg_window_scatter: fp32 [M][C] feature gradient -> PT bf16 interior
//   bilinear_fwd / bwd: [N][Hl][Wl][k] (pixel-major) <-> [N][k][S][S] (NCHW, what the loss takes),
//                       align_corners = False (torch default), backward in gather form.
#include "common.h"
#include "../../include/iic_hip.h"

__global__ __launch_bounds__(256) void seg_window_gather_kernel(const bf16_t* __restrict__ pt,
                                                                float* __restrict__ out, int N,
                                                                int Hw, int Ww, int Hp, int Wp,
                                                                int off, int C) {
  const long total = (long)N * Hw * Ww * (C / 8);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % (C / 8));
    const long m = i / (C / 8);
    const int xx = (int)(m % Ww);
    const long r = m / Ww;
    const int yy = (int)(r % Hw), n = (int)(r / Hw);
    const uint4 v = *reinterpret_cast<const uint4*>(
        pt + (((long)n * Hp + yy + off) * Wp + xx + off) * C + c8 * 8);
    float4 a = make_float4(bf16lo(v.x), bf16hi(v.x), bf16lo(v.y), bf16hi(v.y));
    float4 b = make_float4(bf16lo(v.z), bf16hi(v.z), bf16lo(v.w), bf16hi(v.w));
    float* o = out + m * C + c8 * 8;
    *reinterpret_cast<float4*>(o) = a;
    *reinterpret_cast<float4*>(o + 4) = b;
  }
}

// window rows (yy, xx) with 1 <= yy <= Hw-2, 1 <= xx <= Ww-2 are the PT interior
__global__ __launch_bounds__(256) void seg_window_scatter_kernel(const float* __restrict__ in,
                                                                 bf16_t* __restrict__ pt, int N,
                                                                 int Hw, int Ww, int Hp, int Wp,
                                                                 int off, int C) {
  const int Hi = Hw - 2, Wi = Ww - 2;
  const long total = (long)N * Hi * Wi * (C / 8);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % (C / 8));
    const long q = i / (C / 8);
    const int x = (int)(q % Wi);
    const long r = q / Wi;
    const int y = (int)(r % Hi), n = (int)(r / Hi);
    const long m = ((long)n * Hw + y + 1) * Ww + x + 1;
    const float* s = in + m * C + c8 * 8;
    const float4 a = *reinterpret_cast<const float4*>(s), b = *reinterpret_cast<const float4*>(s + 4);
    *reinterpret_cast<uint4*>(pt + (((long)n * Hp + y + 1 + off) * Wp + x + 1 + off) * C + c8 * 8) =
        make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y),
                   pack_bf16x2(b.z, b.w));
  }
}

__device__ __forceinline__ void bl_src(int d, float scale, int in, int& i0, int& i1, float& lam) {
  float s = ((float)d + 0.5f) * scale - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  lam = s - (float)i0;
}

// out[n][c][y][x] = bilinear(in[n][.][.][c]).  One workgroup per output row (n, y): the two input
// rows it blends are staged in LDS (pixel-major, as they come), then every thread produces outputs
// that are consecutive in x for one class -- coalesced stores of the NCHW result, which is 4x the
// input.  (The first version gathered 4 strided values per output element from global memory.)
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const float* __restrict__ in,
                                                           float* __restrict__ out, int N, int Hl,
                                                           int Wl, int k, int S) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* s0 = reinterpret_cast<float*>(smem_raw);            // [Wl][k] row y0
  float* s1 = s0 + Wl * k;                                   // [Wl][k] row y1
  const float sy = (float)Hl / (float)S, sx = (float)Wl / (float)S;
  const int n = blockIdx.x / S, y = blockIdx.x - n * S;
  int y0, y1;
  float ly;
  bl_src(y, sy, Hl, y0, y1, ly);
  const float* r0 = in + ((long)n * Hl + y0) * Wl * k;
  const float* r1 = in + ((long)n * Hl + y1) * Wl * k;
  for (int i = threadIdx.x; i < Wl * k; i += blockDim.x) {
    s0[i] = r0[i];
    s1[i] = r1[i];
  }
  __syncthreads();
  float* o = out + (long)n * k * S * S + (long)y * S;
  for (int i = threadIdx.x; i < k * S; i += blockDim.x) {
    const int c = i / S, x = i - c * S;
    int x0, x1;
    float lx;
    bl_src(x, sx, Wl, x0, x1, lx);
    const float v00 = s0[x0 * k + c], v01 = s0[x1 * k + c];
    const float v10 = s1[x0 * k + c], v11 = s1[x1 * k + c];
    o[(long)c * S * S + x] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
  }
}

// din[n][yl][xl][c] = sum over the output pixels whose footprint touches (yl, xl).  One workgroup
// per input row (n, yl): threads walk (class, xl) with xl fastest, so the rows of dout they read are
// contiguous in x; the results go through LDS to be written pixel-major.  Same summation order per
// element as the first version (y outer, x inner).
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ dout,
                                                           float* __restrict__ din, int N, int Hl,
                                                           int Wl, int k, int S) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* sres = reinterpret_cast<float*>(smem_raw);          // [Wl][k]
  const float sy = (float)Hl / (float)S, sx = (float)Wl / (float)S;
  const float ry = (float)S / (float)Hl, rx = (float)S / (float)Wl;
  const int n = blockIdx.x / Hl, yl = blockIdx.x - n * Hl;
  // candidate destination rows: src(d) in (yl-1, yl+1)
  int ya = (int)floorf(((float)yl - 1.f + 0.5f) * ry - 0.5f) - 1, yb = (int)ceilf(((float)yl + 1.f + 0.5f) * ry - 0.5f) + 1;
  if (ya < 0) ya = 0;
  if (yb > S - 1) yb = S - 1;
  // row weights of the candidate rows (the same for the whole workgroup), column weights once per
  // thread: the footprint loop itself is then multiply-adds only
  constexpr int BLW = 12;                      // candidate rows / columns per input pixel: < 2 S/Hl + 5
  float wyv[BLW];
#pragma unroll
  for (int j = 0; j < BLW; ++j) {
    const int y = ya + j;
    float wy = 0.f;
    if (y <= yb) {
      int y0, y1;
      float ly;
      bl_src(y, sy, Hl, y0, y1, ly);
      if (y0 == yl) wy += 1.f - ly;
      if (y1 == yl) wy += ly;
    }
    wyv[j] = wy;
  }
  for (int i = threadIdx.x; i < k * Wl; i += blockDim.x) {
    const int c = i / Wl, xl = i - c * Wl;
    int xa = (int)floorf(((float)xl - 1.f + 0.5f) * rx - 0.5f) - 1, xb = (int)ceilf(((float)xl + 1.f + 0.5f) * rx - 0.5f) + 1;
    if (xa < 0) xa = 0;
    if (xb > S - 1) xb = S - 1;
    float wxv[BLW];
#pragma unroll
    for (int j = 0; j < BLW; ++j) {
      const int x = xa + j;
      float wx = 0.f;
      if (x <= xb) {
        int x0, x1;
        float lx;
        bl_src(x, sx, Wl, x0, x1, lx);
        if (x0 == xl) wx += 1.f - lx;
        if (x1 == xl) wx += lx;
      }
      wxv[j] = wx;
    }
    const float* g = dout + ((long)n * k + c) * S * S;
    float acc = 0.f;
#pragma unroll
    for (int jy = 0; jy < BLW; ++jy) {
      if (wyv[jy] == 0.f) continue;
      const float* gr = g + (long)(ya + jy) * S + xa;
#pragma unroll
      for (int jx = 0; jx < BLW; ++jx)
        if (wxv[jx] != 0.f) acc += wyv[jy] * wxv[jx] * gr[jx];
    }
    sres[xl * k + c] = acc;
  }
  __syncthreads();
  float* o = din + ((long)n * Hl + yl) * Wl * k;
  for (int i = threadIdx.x; i < Wl * k; i += blockDim.x) o[i] = sres[i];
}

// ====================================================================================
// Fused head GEMMs on the bf16 PT tensor itself (round 2).  The gather -> fp32 GEMM -> scatter
// chain above moves the 512-channel feature window three times in fp32 (1.5 GB each at the
// Potsdam shapes) through a one-wave-per-tile GEMM with strided scalar loads; these kernels read
// the bf16 window directly (bf16 -> fp32 is exact, products and sums stay fp32 on
// v_mfma_f32_16x16x4_f32), keep W in LDS and write logits / bf16 feature gradients / per-chunk
// weight-gradient partials straight from the accumulators.  C = 256 or 512, k <= 32.
//   m = ((n * Hw) + wy) * Ww + wx   window row,  PT pixel (wy + off, wx + off)
// ====================================================================================
__device__ __forceinline__ f32x4 sh_mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ long sh_row_off(long m, int Hw, int Ww, int Hp, int Wp, int off, int C) {
  const int xx = (int)(m % Ww);
  const long r = m / Ww;
  const int yy = (int)(r % Hw), n = (int)(r / Hw);
  return (((long)n * Hp + yy + off) * Wp + xx + off) * C;
}

#define SH_ROWS 256     // window rows per workgroup (4 waves x 4 tiles of 16)

// logits[m][j] = sum_c x[m][c] * W[j][c].  K order inside a 32-channel group: lane kk owns channels
// 8kk .. 8kk+7 (one 16-byte load), MFMA step s multiplies channel 8kk + s on both operands.
template <int TK>
__global__ __launch_bounds__(256) void seg_head_fwd_kernel(const bf16_t* __restrict__ pt,
                                                           const float* __restrict__ Wm,
                                                           float* __restrict__ logits, long M, int Hw,
                                                           int Ww, int Hp, int Wp, int off, int C, int k) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* sW = reinterpret_cast<float*>(smem_raw);            // [16*TK][C + 4]
  const int PW = C + 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, kk = lane >> 4;
  for (int idx = tid; idx < 16 * TK * (C / 4); idx += 256) {
    const int j = idx / (C / 4), c4 = idx - j * (C / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < k) v = *reinterpret_cast<const float4*>(Wm + (long)j * C + 4 * c4);
    *reinterpret_cast<float4*>(sW + j * PW + 4 * c4) = v;
  }
  __syncthreads();
  const long m_wg = (long)blockIdx.x * SH_ROWS;
  for (int t = 0; t < SH_ROWS / 64; ++t) {
    const long m0 = m_wg + t * 64 + wave * 16;
    if (m0 >= M) break;
    const long mrow = m0 + c < M ? m0 + c : M - 1;            // (clamped rows are not stored)
    const bf16_t* ap = pt + sh_row_off(mrow, Hw, Ww, Hp, Wp, off, C) + 8 * kk;
    f32x4 acc[TK], acc2[TK];                                // two chains per tile (MFMA dependent latency)
#pragma unroll
    for (int tj = 0; tj < TK; ++tj) acc[tj] = acc2[tj] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < C; k0 += 64) {
      uint4 v[2];
      v[0] = *reinterpret_cast<const uint4*>(ap + k0);
      v[1] = k0 + 32 < C ? *reinterpret_cast<const uint4*>(ap + k0 + 32) : make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int hgrp = 0; hgrp < 2; ++hgrp) {
        if (k0 + 32 * hgrp >= C) break;
        const float a[8] = {bf16lo(v[hgrp].x), bf16hi(v[hgrp].x), bf16lo(v[hgrp].y), bf16hi(v[hgrp].y),
                            bf16lo(v[hgrp].z), bf16hi(v[hgrp].z), bf16lo(v[hgrp].w), bf16hi(v[hgrp].w)};
#pragma unroll
        for (int tj = 0; tj < TK; ++tj) {
          const float* wp = sW + (tj * 16 + c) * PW + k0 + 32 * hgrp + 8 * kk;
          const float4 b0 = *reinterpret_cast<const float4*>(wp), b1 = *reinterpret_cast<const float4*>(wp + 4);
          acc[tj] = sh_mfma16(a[0], b0.x, acc[tj]); acc2[tj] = sh_mfma16(a[1], b0.y, acc2[tj]);
          acc[tj] = sh_mfma16(a[2], b0.z, acc[tj]); acc2[tj] = sh_mfma16(a[3], b0.w, acc2[tj]);
          acc[tj] = sh_mfma16(a[4], b1.x, acc[tj]); acc2[tj] = sh_mfma16(a[5], b1.y, acc2[tj]);
          acc[tj] = sh_mfma16(a[6], b1.z, acc[tj]); acc2[tj] = sh_mfma16(a[7], b1.w, acc2[tj]);
        }
      }
    }
#pragma unroll
    for (int tj = 0; tj < TK; ++tj) {
      const int j = tj * 16 + c;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long m = m0 + kk * 4 + r;
        if (m < M && j < k) logits[m * k + j] = acc[tj][r] + acc2[tj][r];
      }
    }
  }
}

// dx[m][c] = sum_j dlog[m][j] * W[j][c] for the window's INTERIOR rows (the ring is the conv's zero
// padding: its PT positions stay zero), rounded to bf16 into the PT gradient.  Rows of the MFMA
// tile = 16 channels, columns = 16 window rows: a lane ends up with 4 consecutive channels of one
// pixel (one 8-byte store).  Waves split the channel range.
__global__ __launch_bounds__(256) void seg_head_bwd_dx_kernel(const float* __restrict__ dlog,
                                                              const float* __restrict__ Wm,
                                                              bf16_t* __restrict__ pt_dx, long M, int Hw,
                                                              int Ww, int Hp, int Wp, int off, int C, int k) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* sW = reinterpret_cast<float*>(smem_raw);            // [k4][C + 16]
  const int PW = C + 16, k4 = (k + 3) & ~3;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, kk = lane >> 4;
  for (int idx = tid; idx < k4 * (C / 4); idx += 256) {
    const int j = idx / (C / 4), c4 = idx - j * (C / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < k) v = *reinterpret_cast<const float4*>(Wm + (long)j * C + 4 * c4);
    *reinterpret_cast<float4*>(sW + j * PW + 4 * c4) = v;
  }
  __syncthreads();
  const int cw = C / 4;                                       // channels per wave
  const long m_wg = (long)blockIdx.x * 64;
  for (int t = 0; t < 4; ++t) {
    const long m0 = m_wg + t * 16;
    if (m0 >= M) break;
    const long mrow = m0 + c < M ? m0 + c : M - 1;
    const int xx = (int)(mrow % Ww);
    const int yy = (int)((mrow / Ww) % Hw);
    const bool interior = (m0 + c < M) && xx >= 1 && xx <= Ww - 2 && yy >= 1 && yy <= Hw - 2;
    bf16_t* op = pt_dx + sh_row_off(mrow, Hw, Ww, Hp, Wp, off, C);
    float b[8];                                               // dlog[m][4s + kk], k4 <= 32
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int j = 4 * s + kk;
      b[s] = (4 * s < k4 && j < k) ? dlog[mrow * k + j] : 0.f;
    }
    for (int n0 = wave * cw; n0 < (wave + 1) * cw; n0 += 64) {
      f32x4 acc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        if (4 * s < k4) {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            acc[u] = sh_mfma16(sW[(4 * s + kk) * PW + n0 + 16 * u + c], b[s], acc[u]);
        }
      }
      if (interior) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          *reinterpret_cast<uint2*>(op + n0 + 16 * u + 4 * kk) =
              make_uint2(pack_bf16x2(acc[u][0], acc[u][1]), pack_bf16x2(acc[u][2], acc[u][3]));
      }
    }
  }
}

// part[chunk][j][c] = sum over the chunk's window rows of dlog[m][j] * x[m][c]; the caller folds the
// chunks in a fixed order (iic_colsum_f32): deterministic, unlike an atomic split-K.  Wave w owns
// channels [w*C/4, (w+1)*C/4); within a 64-channel group, column c of tile u is channel 4c + u
// (one 8-byte load per lane and row quad serves the four tiles).
#define SHW_ROWS 1024
template <int TK>
__global__ __launch_bounds__(256) void seg_head_wgrad_kernel(const float* __restrict__ dlog,
                                                             const bf16_t* __restrict__ pt,
                                                             float* __restrict__ part, long M, int Hw,
                                                             int Ww, int Hp, int Wp, int off, int C, int k) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, kk = lane >> 4;
  const int cw = C / 4;
  const int NG = cw / 64;                                     // 64-channel groups per wave (C = 512: 2)
  const long m_lo = (long)blockIdx.x * SHW_ROWS, m_hi = min(M, m_lo + SHW_ROWS);
  f32x4 acc[2][TK][4];                                        // [group][class tile][channel tile]  (C <= 512)
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int tj = 0; tj < TK; ++tj)
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[g][tj][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (long m0 = m_lo; m0 < m_hi; m0 += 4) {
    const long m = m0 + kk;
    const bool vm = m < m_hi;
    const long mm = vm ? m : m_hi - 1;
    const bf16_t* xp = pt + sh_row_off(mm, Hw, Ww, Hp, Wp, off, C) + wave * cw + 4 * c;
    float a[TK];
#pragma unroll
    for (int tj = 0; tj < TK; ++tj) {
      const int j = tj * 16 + c;
      a[tj] = (vm && j < k) ? dlog[mm * k + j] : 0.f;
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      if (g < NG) {
        const uint2 v = *reinterpret_cast<const uint2*>(xp + 64 * g);
        const float b[4] = {bf16lo(v.x), bf16hi(v.x), bf16lo(v.y), bf16hi(v.y)};
#pragma unroll
        for (int tj = 0; tj < TK; ++tj)
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[g][tj][u] = sh_mfma16(a[tj], b[u], acc[g][tj][u]);
      }
    }
  }
  float* o = part + (long)blockIdx.x * k * C;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    if (g >= NG) break;
#pragma unroll
    for (int tj = 0; tj < TK; ++tj)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = tj * 16 + kk * 4 + r;                 // D row = class
          const int ch = wave * cw + 64 * g + 4 * c + u;      // D column c of tile u
          if (j < k) o[(long)j * C + ch] = acc[g][tj][u][r];
        }
  }
}

static int grid_for(long total) {
  long g = (total + 255) / 256;
  if (g > 16384) g = 16384;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" {

int iic_seg_window_gather(const void* pt, float* out, int N, int Hw, int Ww, int Hp, int Wp, int off,
                          int C, void* stream) {
  if (!pt || !out || N <= 0 || (C & 7) || off < 0 || Hw + off > Hp || Ww + off > Wp) return IIC_ERR_ARG;
  hipLaunchKernelGGL(seg_window_gather_kernel, dim3(grid_for((long)N * Hw * Ww * (C / 8))), dim3(256),
                     0, (hipStream_t)stream, (const bf16_t*)pt, out, N, Hw, Ww, Hp, Wp, off, C);
  return iic_launch_status();
}

int iic_seg_window_scatter(const float* in, void* pt, int N, int Hw, int Ww, int Hp, int Wp, int off,
                           int C, void* stream) {
  if (!pt || !in || N <= 0 || (C & 7) || off < 0 || Hw < 3 || Ww < 3) return IIC_ERR_ARG;
  hipLaunchKernelGGL(seg_window_scatter_kernel, dim3(grid_for((long)N * (Hw - 2) * (Ww - 2) * (C / 8))),
                     dim3(256), 0, (hipStream_t)stream, in, (bf16_t*)pt, N, Hw, Ww, Hp, Wp, off, C);
  return iic_launch_status();
}

int iic_bilinear_fwd(const float* in_nhwc, float* out_nchw, int N, int Hl, int Wl, int k, int S,
                     void* stream) {
  if (!in_nhwc || !out_nchw || N <= 0 || Hl <= 0 || Wl <= 0 || k <= 0 || S <= 0) return IIC_ERR_ARG;
  const size_t lds = (size_t)2 * Wl * k * sizeof(float);
  if (lds > 64 * 1024) return IIC_ERR_UNSUPPORTED;
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bilinear_fwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(N * S), dim3(256), lds, (hipStream_t)stream, in_nhwc,
                     out_nchw, N, Hl, Wl, k, S);
  return iic_launch_status();
}

int iic_bilinear_bwd(const float* dout_nchw, float* din_nhwc, int N, int Hl, int Wl, int k, int S,
                     void* stream) {
  if (!dout_nchw || !din_nhwc || N <= 0 || Hl <= 0 || Wl <= 0 || k <= 0 || S <= 0) return IIC_ERR_ARG;
  const size_t lds = (size_t)Wl * k * sizeof(float);
  if (lds > 64 * 1024) return IIC_ERR_UNSUPPORTED;
  // (the kernel tabulates 12 candidate rows / columns per input pixel: up-sampling factors up to 3.5)
  if (2 * S > 7 * Hl || 2 * S > 7 * Wl) return IIC_ERR_UNSUPPORTED;
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bilinear_bwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(N * Hl), dim3(256), lds, (hipStream_t)stream,
                     dout_nchw, din_nhwc, N, Hl, Wl, k, S);
  return iic_launch_status();
}

/* Fused 10a head on the bf16 PT window (see the kernels): C = 256 or 512, k <= 32.
 * logits [M][k], dlog [M][k] fp32 row-major, M = N*Hw*Ww window rows; w [k][C] fp32.
 * iic_seg_head_wgrad writes iic_seg_head_wgrad_chunks(M) partial matrices [chunk][k][C] (fold them
 * with iic_colsum_f32).  iic_seg_head_bwd_dx writes the interior rows of pt_dx only.            */
// (each of the 4 waves owns C/4 channels in 64-channel groups: C = 256 or 512)
int iic_seg_head_supported(int C, int k) { return C % 256 == 0 && C <= 512 && k >= 1 && k <= 32; }
int iic_seg_head_wgrad_chunks(long M) { return (int)((M + SHW_ROWS - 1) / SHW_ROWS); }

int iic_seg_head_fwd(const void* pt, const float* w, float* logits, int N, int Hw, int Ww, int Hp,
                     int Wp, int off, int C, int k, void* stream) {
  if (!pt || !w || !logits || N <= 0 || off < 0 || Hw + off > Hp || Ww + off > Wp) return IIC_ERR_ARG;
  if (!iic_seg_head_supported(C, k)) return IIC_ERR_UNSUPPORTED;
  const long M = (long)N * Hw * Ww;
  const int tk = (k + 15) / 16;
  const size_t lds = (size_t)16 * tk * (C + 4) * sizeof(float);
  const int grid = (int)((M + SH_ROWS - 1) / SH_ROWS);
#define SH_FWD(TK_)                                                                              \
  do {                                                                                           \
    if (lds > 48 * 1024)                                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&seg_head_fwd_kernel<TK_>),        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);           \
    hipLaunchKernelGGL((seg_head_fwd_kernel<TK_>), dim3(grid), dim3(256), lds, (hipStream_t)stream, \
                       (const bf16_t*)pt, w, logits, M, Hw, Ww, Hp, Wp, off, C, k);              \
  } while (0)
  if (tk == 1) SH_FWD(1); else SH_FWD(2);
  return iic_launch_status();
}

int iic_seg_head_bwd_dx(const float* dlog, const float* w, void* pt_dx, int N, int Hw, int Ww, int Hp,
                        int Wp, int off, int C, int k, void* stream) {
  if (!dlog || !w || !pt_dx || N <= 0 || off < 0 || Hw < 3 || Ww < 3 || Hw + off > Hp || Ww + off > Wp)
    return IIC_ERR_ARG;
  if (!iic_seg_head_supported(C, k)) return IIC_ERR_UNSUPPORTED;
  const long M = (long)N * Hw * Ww;
  const size_t lds = (size_t)((k + 3) & ~3) * (C + 16) * sizeof(float);
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&seg_head_bwd_dx_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(seg_head_bwd_dx_kernel, dim3((int)((M + 63) / 64)), dim3(256), lds,
                     (hipStream_t)stream, dlog, w, (bf16_t*)pt_dx, M, Hw, Ww, Hp, Wp, off, C, k);
  return iic_launch_status();
}

int iic_seg_head_wgrad(const float* dlog, const void* pt, float* partials, int N, int Hw, int Ww,
                       int Hp, int Wp, int off, int C, int k, void* stream) {
  if (!dlog || !pt || !partials || N <= 0 || off < 0 || Hw + off > Hp || Ww + off > Wp) return IIC_ERR_ARG;
  if (!iic_seg_head_supported(C, k)) return IIC_ERR_UNSUPPORTED;
  const long M = (long)N * Hw * Ww;
  const int grid = iic_seg_head_wgrad_chunks(M);
  if (k <= 16)
    hipLaunchKernelGGL((seg_head_wgrad_kernel<1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, dlog,
                       (const bf16_t*)pt, partials, M, Hw, Ww, Hp, Wp, off, C, k);
  else
    hipLaunchKernelGGL((seg_head_wgrad_kernel<2>), dim3(grid), dim3(256), 0, (hipStream_t)stream, dlog,
                       (const bf16_t*)pt, partials, M, Hw, Ww, Hp, Wp, off, C, k);
  return iic_launch_status();
}

}  // extern "C"

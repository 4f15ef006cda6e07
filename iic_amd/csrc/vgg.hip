// VGG-style trunk pieces for gfx950 (ClusterNet6c, SegmentationNet10a):
//   * first-layer convolution from the fp32 NCHW image (Cin <= 8, K = 3 or 5, "same" size)
//     forward (+ BatchNorm statistics) and weight gradient, on exact-fp32 MFMA
//     (v_mfma_f32_32x32x2_f32): K*K*Cin <= 128 is far too short a contraction for the bf16
//     implicit-GEMM kernel, and the input is fp32 at the boundary anyway;
//   * MaxPool2d(kernel 2, stride 2, padding 0) forward / backward on PT tensors.
//
// Replaces the first nn.Conv2d of /root/reference/code/archs/cluster/vgg.py:24-26 as
// configured by net6c.py:16-20 (5x5, pad 2) and net10a.py:21-25 (3x3, pad 1), and
// nn.MaxPool2d(kernel_size=2, stride=2) of vgg.py:19-20.
#include "common.h"
#include "../../include/iic_hip.h"

#define FC_CO 64
#define FC_KMAX 128
#define FC_PERSIST 1024

struct FcTables {           // built once per block in LDS
  int koff[FC_KMAX];        // c*H*W + (kh-p)*W + (kw-p)
  int kdy[FC_KMAX];         // kh - p
  int kdx[FC_KMAX];         // kw - p
};

__device__ __forceinline__ void fc_build_tables(FcTables* t, int Cin, int K, int pad, int H, int W) {
  const int KK = K * K, KT = Cin * KK;
  for (int k = threadIdx.x; k < FC_KMAX; k += blockDim.x) {
    if (k < KT) {
      const int c = k / KK, r = k - c * KK, kh = r / K, kw = r - kh * K;
      t->koff[k] = c * H * W + (kh - pad) * W + (kw - pad);
      t->kdy[k] = kh - pad;
      t->kdx[k] = kw - pad;
    } else {
      t->koff[k] = 0;
      t->kdy[k] = 1 << 20;   // never in bounds
      t->kdx[k] = 0;
    }
  }
}

// ------------------------------------------------------------------------------------
// forward: out PT bf16 [N][H+2P][W+2P][64] interior, stats += sum / sum^2 of the fp32 outputs
//   wave tile = 32 pixels (row segment) x 64 couts; A[i = pixel][k], B[k][j = cout]
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void firstconv_fwd_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ w,
                                                            bf16_t* __restrict__ out,
                                                            float* __restrict__ stats, int N,
                                                            int Cin, int H, int W, int K, int pad,
                                                            int P) {
  __shared__ FcTables tab;
  __shared__ float sW[FC_KMAX * FC_CO];   // [k][co]
  __shared__ float s_red[4][2][2][32];
  const int KT = Cin * K * K;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  fc_build_tables(&tab, Cin, K, pad, H, W);
  for (int i = threadIdx.x; i < FC_KMAX * FC_CO; i += blockDim.x) {
    const int k = i / FC_CO, co = i - k * FC_CO;
    sW[i] = k < KT ? w[co * KT + k] : 0.f;
  }
  __syncthreads();
  const int nseg = (W + 31) / 32;
  const long tiles = (long)N * H * nseg;
  const int KS = (KT + 1) / 2;
  const int i = lane & 31, kk = lane >> 5;
  const int Hp = H + 2 * P, Wp = W + 2 * P;
  float s[2] = {0.f, 0.f}, ss[2] = {0.f, 0.f};
  for (long t = (long)blockIdx.x * 4 + wave; t < tiles; t += (long)gridDim.x * 4) {
    const int seg = (int)(t % nseg);
    const long row = t / nseg;
    const int y = (int)(row % H), n = (int)(row / H);
    const int px = seg * 32 + i;
    const float* xin = x + (long)n * Cin * H * W + (long)y * W + px;
    f32x16 acc[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[h][r] = 0.f;
    // patch values of 6 k-steps are fetched before their MFMAs (the step-by-step loop paid one
    // global round trip per 2 MFMAs)
    for (int sb = 0; sb < KS; sb += 6) {
      float ra[6];
#pragma unroll
      for (int s6 = 0; s6 < 6; ++s6) {
        const int k = min(2 * (sb + s6) + kk, FC_KMAX - 1);  // (tables are zero-padded past KT)
        const int yy = y + tab.kdy[k], xx = px + tab.kdx[k];
        ra[s6] = (sb + s6 < KS && yy >= 0 && yy < H && xx >= 0 && xx < W) ? xin[tab.koff[k]] : 0.f;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s6 = 0; s6 < 6; ++s6) {
        if (sb + s6 < KS) {
          const int k = 2 * (sb + s6) + kk;
          const float b0 = sW[k * FC_CO + i], b1 = sW[k * FC_CO + 32 + i];
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[s6], b0, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[s6], b1, acc[1], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ox = seg * 32 + mfma32_row(r, lane);
      if (ox < W) {
        bf16_t* d = out + (((long)n * Hp + y + P) * Wp + ox + P) * FC_CO;
        d[i] = f32_to_bf16(acc[0][r]);
        d[i + 32] = f32_to_bf16(acc[1][r]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          s[h] += acc[h][r];
          ss[h] += acc[h][r] * acc[h][r];
        }
      }
    }
  }
  if (stats) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      s[h] += __shfl_xor(s[h], 32, 64);
      ss[h] += __shfl_xor(ss[h], 32, 64);
      if (lane < 32) {
        s_red[wave][h][0][lane] = s[h];
        s_red[wave][h][1][lane] = ss[h];
      }
    }
    __syncthreads();
    if (threadIdx.x < 128) {
      const int which = threadIdx.x >> 6, ch = threadIdx.x & 63;
      float t = 0.f;
      for (int wv = 0; wv < 4; ++wv) t += s_red[wv][ch >> 5][which][ch & 31];
      iic_stat_add(stats, blockIdx.x % IIC_STAT_STRIPES, FC_CO, ch, which, t);
    }
  }
}

// ------------------------------------------------------------------------------------
// weight gradient: part[block][co][LD] = sum_pix dy[pix][co] * patch[pix][k]
//   A[i = co][kk = pix] = dy (bf16 PT), B[kk = pix][j = k]; NKT = ceil(KT/32) column tiles
// ------------------------------------------------------------------------------------
template <int NKT>
__global__ __launch_bounds__(256) void firstconv_wgrad_kernel(const float* __restrict__ x,
                                                              const bf16_t* __restrict__ dy,
                                                              float* __restrict__ part, int N,
                                                              int Cin, int H, int W, int K, int pad,
                                                              int P) {
  __shared__ FcTables tab;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* red = reinterpret_cast<float*>(smem_raw);   // [4 waves][64][NKT*32]
  constexpr int LD = NKT * 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  fc_build_tables(&tab, Cin, K, pad, H, W);
  __syncthreads();
  const int nseg = (W + 31) / 32;
  const long tiles = (long)N * H * nseg;
  const int i = lane & 31, kk = lane >> 5;
  const int Hp = H + 2 * P, Wp = W + 2 * P;
  f32x16 acc[2][NKT];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[h][t][r] = 0.f;
  // this lane's patch columns k = t*32 + i
  int koff[NKT], kdy[NKT], kdx[NKT];
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    koff[t] = tab.koff[t * 32 + i];
    kdy[t] = tab.kdy[t * 32 + i];
    kdx[t] = tab.kdx[t * 32 + i];
  }
  for (long tl = (long)blockIdx.x * 4 + wave; tl < tiles; tl += (long)gridDim.x * 4) {
    const int seg = (int)(tl % nseg);
    const long row = tl / nseg;
    const int y = (int)(row % H), n = (int)(row / H);
    const float* xin = x + (long)n * Cin * H * W + (long)y * W;
    const bf16_t* drow = dy + (((long)n * Hp + y + P) * Wp + P) * FC_CO;
    // operands of 8 pixel pairs are fetched before their MFMAs (one load round trip per 8 steps
    // instead of one per step: the loop was latency-bound at 1/7 of the fp32 MFMA rate)
#pragma unroll
    for (int sb = 0; sb < 16; sb += 8) {
      unsigned short ra0[8], ra1[8];
      float rb[8][NKT];
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) {
        const int px = seg * 32 + 2 * (sb + s8) + kk;   // this lane's k-slot pixel
        const bool pv = px < W;
        ra0[s8] = pv ? drow[(long)px * FC_CO + i] : (unsigned short)0;
        ra1[s8] = pv ? drow[(long)px * FC_CO + 32 + i] : (unsigned short)0;
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
          const int yy = y + kdy[t], xx = px + kdx[t];
          rb[s8][t] = (pv && yy >= 0 && yy < H && xx >= 0 && xx < W) ? xin[px + koff[t]] : 0.f;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) {
        const float a0 = bf16_to_f32(ra0[s8]), a1 = bf16_to_f32(ra1[s8]);
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
          acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, rb[s8][t], acc[0][t], 0, 0, 0);
          acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, rb[s8][t], acc[1][t], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        red[((long)wave * 64 + h * 32 + mfma32_row(r, lane)) * LD + t * 32 + i] = acc[h][t][r];
  __syncthreads();
  for (int idx = threadIdx.x; idx < 64 * LD; idx += blockDim.x) {
    float t = 0.f;
    for (int wv = 0; wv < 4; ++wv) t += red[(long)wv * 64 * LD + idx];
    part[(long)blockIdx.x * 64 * LD + idx] = t;
  }
}

__global__ __launch_bounds__(256) void fc_wgrad_reduce_kernel(const float* __restrict__ part,
                                                              int nblocks, int LD, int KT,
                                                              float* __restrict__ dW) {
  __shared__ float red[256];
  const int idx = blockIdx.x;            // co*KT + k
  const int co = idx / KT, k = idx - co * KT;
  float t = 0.f;
  for (int b = threadIdx.x; b < nblocks; b += 256) t += part[((long)b * 64 + co) * LD + k];
  red[threadIdx.x] = t;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) dW[idx] = red[0];
}

// ------------------------------------------------------------------------------------
// MaxPool2d(2, 2, pad 0) on PT tensors.  16 B (8 channels) per lane.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void unpack8v(const uint4 v, float* f) {
  f[0] = bf16lo(v.x); f[1] = bf16hi(v.x); f[2] = bf16lo(v.y); f[3] = bf16hi(v.y);
  f[4] = bf16lo(v.z); f[5] = bf16hi(v.z); f[6] = bf16lo(v.w); f[7] = bf16hi(v.w);
}

// BN = true: `in` is the convolution output y of a pooled stage and the activation a = relu(scale*y + shift) is never
// stored: every element is recomputed as bn_apply_kernel (bn.hip) would have written it -- the same expression, rounded
// to bf16 -- before it enters the window (forward: one full-tensor write and one read less per pooled stage; the
// backward's arg-max sees the same bits).  sc / sh: this thread's 8 channels of coef[0][C], coef[1][C].
__device__ __forceinline__ void bn_relu_bf16_8(float* v, const float* sc, const float* sh) {
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i] * sc[i] + sh[i], 0.f);
  const uint4 r = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                             pack_bf16x2(v[6], v[7]));
  unpack8v(r, v);
}

template <bool BN>
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const bf16_t* __restrict__ in,
                                                           const float* __restrict__ coef,
                                                           bf16_t* __restrict__ out, int H, int W,
                                                           int Pi, int Po, int C) {
  const int Ho = H / 2, Wo = W / 2, c8n = C >> 3;
  const int item = blockIdx.y * blockDim.x + threadIdx.x;
  if (item >= Wo * c8n) return;
  const int xo = item / c8n, c8 = item - xo * c8n;
  const int n = blockIdx.x / Ho, yo = blockIdx.x - n * Ho;
  const int Hpi = H + 2 * Pi, Wpi = W + 2 * Pi, Hpo = Ho + 2 * Po, Wpo = Wo + 2 * Po;
  float m[8], sc[BN ? 8 : 1], sh[BN ? 8 : 1];
  if (BN) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      sc[BN ? i : 0] = coef[c8 * 8 + i];
      sh[BN ? i : 0] = coef[C + c8 * 8 + i];
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) m[i] = -INFINITY;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const long off = (((long)n * Hpi + 2 * yo + (q >> 1) + Pi) * Wpi + 2 * xo + (q & 1) + Pi) * C + c8 * 8;
    float v[8];
    unpack8v(*reinterpret_cast<const uint4*>(in + off), v);
    if (BN) bn_relu_bf16_8(v, sc, sh);
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = fmaxf(m[i], v[i]);
  }
  const long o = (((long)n * Hpo + yo + Po) * Wpo + xo + Po) * C + c8 * 8;
  *reinterpret_cast<uint4*>(out + o) = make_uint4(pack_bf16x2(m[0], m[1]), pack_bf16x2(m[2], m[3]),
                                                  pack_bf16x2(m[4], m[5]), pack_bf16x2(m[6], m[7]));
}

// din[4 window positions] = dout at the FIRST arg-max in scan order (torch), 0 elsewhere.
// Odd trailing rows / columns of the input (not covered by any window) receive 0.
template <bool BN>
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const bf16_t* __restrict__ in,
                                                           const float* __restrict__ coef,
                                                           const bf16_t* __restrict__ dout,
                                                           bf16_t* __restrict__ din, int H, int W,
                                                           int Pi, int Po, int C) {
  const int Ho = H / 2, Wo = W / 2, c8n = C >> 3;
  const int Wc = (W + 1) / 2, Hc = (H + 1) / 2;   // cover odd tails
  const int item = blockIdx.y * blockDim.x + threadIdx.x;
  if (item >= Wc * c8n) return;
  const int xo = item / c8n, c8 = item - xo * c8n;
  const int n = blockIdx.x / Hc, yo = blockIdx.x - n * Hc;
  const int Hpi = H + 2 * Pi, Wpi = W + 2 * Pi, Hpo = Ho + 2 * Po, Wpo = Wo + 2 * Po;
  const bool win = yo < Ho && xo < Wo;
  float v[4][8], g[8], sc[BN ? 8 : 1], sh[BN ? 8 : 1];
  if (BN) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      sc[BN ? i : 0] = coef[c8 * 8 + i];
      sh[BN ? i : 0] = coef[C + c8 * 8 + i];
    }
  }
  long offs[4];
  bool ok[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int yy = 2 * yo + (q >> 1), xx = 2 * xo + (q & 1);
    ok[q] = yy < H && xx < W;
    offs[q] = (((long)n * Hpi + yy + Pi) * Wpi + xx + Pi) * C + c8 * 8;
    if (ok[q] && win) {
      unpack8v(*reinterpret_cast<const uint4*>(in + offs[q]), v[q]);
      if (BN) bn_relu_bf16_8(v[q], sc, sh);
    }
  }
  if (win) {
    const long o = (((long)n * Hpo + yo + Po) * Wpo + xo + Po) * C + c8 * 8;
    unpack8v(*reinterpret_cast<const uint4*>(dout + o), g);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (!ok[q]) continue;
    float r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      r[i] = 0.f;
      if (win) {
        int am = 0;
        float best = v[0][i];
#pragma unroll
        for (int t = 1; t < 4; ++t)
          if (v[t][i] > best) { best = v[t][i]; am = t; }
        if (am == q) r[i] = g[i];
      }
    }
    *reinterpret_cast<uint4*>(din + offs[q]) = make_uint4(pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3]),
                                                         pack_bf16x2(r[4], r[5]), pack_bf16x2(r[6], r[7]));
  }
}

// firstconv2.hip
int fc2_fwd_launch(const float* x, const float* w, void* out_pt, float* stats, int N, int Cin, int H, int W, int K,
                   int pad, int P, int max_blocks, void* stream);
int fc2_wgrad_launch(const float* x, const void* dy_pt, float* partials, int N, int Cin, int H, int W, int K, int pad,
                     int P, int max_blocks, int* grid_out, int* ld_out, void* stream);

extern "C" {

static int fc_check(const void* x, int N, int Cin, int H, int W, int K, int pad) {
  if (!x || N <= 0 || H <= 0 || W <= 0) return IIC_ERR_ARG;
  if (Cin < 1 || Cin * K * K > FC_KMAX || (K != 3 && K != 5) || 2 * pad != K - 1)
    return IIC_ERR_UNSUPPORTED;
  return IIC_OK;
}

int iic_firstconv_fwd(const float* x, const float* w, void* out_pt, float* stats, int N, int Cin,
                      int H, int W, int K, int pad, int P, void* stream) {
  int rc = fc_check(x, N, Cin, H, W, K, pad);
  if (rc) return rc;
  if (!w || !out_pt) return IIC_ERR_ARG;
  // second generation (firstconv2.hip: LDS-staged bands, 16-byte accesses) wherever it applies (W % 4 == 0)
  rc = fc2_fwd_launch(x, w, out_pt, stats, N, Cin, H, W, K, pad, P, FC_PERSIST, stream);
  if (rc != IIC_ERR_UNSUPPORTED) return rc;
  const long tiles = (long)N * H * ((W + 31) / 32);
  int grid = (int)((tiles + 3) / 4);
  if (grid > FC_PERSIST) grid = FC_PERSIST;
  hipLaunchKernelGGL(firstconv_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, w,
                     (bf16_t*)out_pt, stats, N, Cin, H, W, K, pad, P);
  return iic_launch_status();
}

long iic_firstconv_wgrad_partial_floats(void) { return (long)FC_PERSIST * 64 * 128; }

int iic_firstconv_wgrad(const float* x, const void* dy_pt, float* partials, float* dW, int N,
                        int Cin, int H, int W, int K, int pad, int P, void* stream) {
  int rc = fc_check(x, N, Cin, H, W, K, pad);
  if (rc) return rc;
  if (!dy_pt || !partials || !dW) return IIC_ERR_ARG;
  const int KT = Cin * K * K, NKT = (KT + 31) / 32;
  {
    int g2 = 0, ld2 = 0;
    rc = fc2_wgrad_launch(x, dy_pt, partials, N, Cin, H, W, K, pad, P, FC_PERSIST, &g2, &ld2, stream);
    if (rc == IIC_OK) {
      hipLaunchKernelGGL(fc_wgrad_reduce_kernel, dim3(64 * KT), dim3(256), 0, (hipStream_t)stream, partials, g2, ld2,
                         KT, dW);
      return iic_launch_status();
    }
    if (rc != IIC_ERR_UNSUPPORTED) return rc;
  }
  const long tiles = (long)N * H * ((W + 31) / 32);
  int grid = (int)((tiles + 3) / 4);
  if (grid > FC_PERSIST) grid = FC_PERSIST;
  const size_t lds = (size_t)4 * 64 * NKT * 32 * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
#define FCW(NK)                                                                                  \
  do {                                                                                           \
    if (lds > 48 * 1024)                                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&firstconv_wgrad_kernel<NK>),      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);           \
    hipLaunchKernelGGL((firstconv_wgrad_kernel<NK>), dim3(grid), dim3(256), lds, s, x,           \
                       (const bf16_t*)dy_pt, partials, N, Cin, H, W, K, pad, P);                 \
  } while (0)
  switch (NKT) {
    case 1: FCW(1); break;
    case 2: FCW(2); break;
    case 3: FCW(3); break;
    case 4: FCW(4); break;
    default: return IIC_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL(fc_wgrad_reduce_kernel, dim3(64 * KT), dim3(256), 0, s, partials, grid,
                     NKT * 32, KT, dW);
  return iic_launch_status();
}

int iic_maxpool2_fwd(const void* in_pt, void* out_pt, int N, int H, int W, int Pi, int Po, int C,
                     void* stream) {
  if (!in_pt || !out_pt || N <= 0 || H < 2 || W < 2 || (C & 7)) return IIC_ERR_ARG;
  dim3 grid(N * (H / 2), ((W / 2) * (C / 8) + 255) / 256);
  hipLaunchKernelGGL(maxpool2_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)in_pt, (const float*)nullptr, (bf16_t*)out_pt, H, W, Pi, Po, C);
  return iic_launch_status();
}

int iic_maxpool2_bwd(const void* in_pt, const void* dout_pt, void* din_pt, int N, int H, int W,
                     int Pi, int Po, int C, void* stream) {
  if (!in_pt || !dout_pt || !din_pt || N <= 0 || H < 2 || W < 2 || (C & 7)) return IIC_ERR_ARG;
  dim3 grid(N * ((H + 1) / 2), (((W + 1) / 2) * (C / 8) + 255) / 256);
  hipLaunchKernelGGL(maxpool2_bwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)in_pt, (const float*)nullptr, (const bf16_t*)dout_pt, (bf16_t*)din_pt, H, W, Pi,
                     Po, C);
  return iic_launch_status();
}

int iic_bn_relu_maxpool2_fwd(const void* y_pt, const float* coef, void* out_pt, int N, int H, int W, int Pi,
                             int Po, int C, void* stream) {
  if (!y_pt || !coef || !out_pt || N <= 0 || H < 2 || W < 2 || (C & 7)) return IIC_ERR_ARG;
  dim3 grid(N * (H / 2), ((W / 2) * (C / 8) + 255) / 256);
  hipLaunchKernelGGL(maxpool2_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)y_pt, coef, (bf16_t*)out_pt, H, W, Pi, Po, C);
  return iic_launch_status();
}

int iic_bn_relu_maxpool2_bwd(const void* y_pt, const float* coef, const void* dout_pt, void* din_pt, int N,
                             int H, int W, int Pi, int Po, int C, void* stream) {
  if (!y_pt || !coef || !dout_pt || !din_pt || N <= 0 || H < 2 || W < 2 || (C & 7)) return IIC_ERR_ARG;
  dim3 grid(N * ((H + 1) / 2), (((W + 1) / 2) * (C / 8) + 255) / 256);
  hipLaunchKernelGGL(maxpool2_bwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)y_pt, coef, (const bf16_t*)dout_pt, (bf16_t*)din_pt, H, W, Pi, Po, C);
  return iic_launch_status();
}

}  // extern "C"

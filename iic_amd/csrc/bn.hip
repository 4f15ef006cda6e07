// BatchNorm2d / ReLU / residual-add kernels on PT (padded NHWC bf16) tensors for gfx950.
// HBM-bound streaming kernels: 16 B per lane, interior pixels only (zero border preserved).
//
// Replaces nn.BatchNorm2d + nn.ReLU + `out += residual` of
//   /root/reference/code/archs/cluster/residual.py:20-41,56-57 and vgg.py:28-30
// (train mode: biased batch variance, eps 1e-5; running stats momentum 0.1 with the
// unbiased variance, exactly torch.nn.BatchNorm2d).
//
// Forward statistics (sum, sum of squares per channel) are produced by the conv kernel's
// epilogue into IIC_STAT_STRIPES stripes; bn_finalize folds them into per-channel
// scale/shift.  Backward: bn_bwd_reduce (sum g, sum g*y) -> bn_bwd_finalize (c1,c2,c3,
// dgamma, dbeta) -> bn_bwd_apply (dy = c1*g + c2*y + c3), with g = dout * (act > 0).
#include "common.h"
#include "../../include/iic_hip.h"

__device__ __forceinline__ void unpack8(const uint4 v, float* f) {
  f[0] = bf16lo(v.x); f[1] = bf16hi(v.x); f[2] = bf16lo(v.y); f[3] = bf16hi(v.y);
  f[4] = bf16lo(v.z); f[5] = bf16hi(v.z); f[6] = bf16lo(v.w); f[7] = bf16hi(v.w);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]),
                    pack_bf16x2(f[6], f[7]));
}

// Streaming loads of the BatchNorm passes carry the non-temporal hint (`global_load ... nt`: evict-first in L2).
// Every tensor here is 50-400 MB and read once per pass; without the hint the stream pushes the weight fragments
// that the other view's convolution keeps re-reading out of the 4 MB L2s.  Measured on the two-stream step
// (profiles/r05_cache_policy_ab.txt): loads nt -0.5 ms per step, stores nt neutral, both = loads alone; the same
// hint on the convolutions' epilogues / patch DMA and on the weight gradient's DMA is neutral to worse.
template <bool NT>
__device__ __forceinline__ uint4 ld16(const bf16_t* p) {
  if (NT) {
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return make_uint4(v[0], v[1], v[2], v[3]);
  }
  return *reinterpret_cast<const uint4*>(p);
}
// coef layout: [0]=scale [1]=shift [2]=mean [3]=invstd [4]=unbiased batch variance, each [C]
__global__ __launch_bounds__(256) void bn_finalize_kernel(
    float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ running_mean, float* __restrict__ running_var, long long* __restrict__ nbt,
    float* __restrict__ coef, int C, long count, long ucount, float eps, float momentum, int training) {
  // 16 lanes per channel (2 stats x 8 bins of the exact accumulators, common.h); lane16 == 0 writes
  const int lane16 = threadIdx.x & 15;
  const int c = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (blockIdx.x == 0 && threadIdx.x == 0 && training && nbt) *nbt += 1;
  float mean, var, unbf = 0.f;
  if (training) {
    double s, ss;
    iic_stat_collect(stats, IIC_STAT_STRIPES, C, c < C ? c : C - 1, lane16, s, ss);
    if (c >= C || lane16 != 0) return;
    const double m = s / (double)count;
    double v = ss / (double)count - m * m;
    if (v < 0.0) v = 0.0;
    mean = (float)m;
    var = (float)v;
    // ucount: sample count for the unbiased-variance factor; differs from `count` only when the
    // batch holds exact replicas that were forwarded once (replica de-duplication)
    const double unb = ucount > 1 ? v * (double)ucount / (double)(ucount - 1) : v;
    unbf = (float)unb;
    if (running_mean) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbf;
    }
  } else {
    if (c >= C || lane16 != 0) return;
    mean = running_mean[c];
    var = running_var[c];
  }
  const float invstd = rsqrtf(var + eps);
  // (rsqrtf is ~1 ulp on gfx950: no refinement step; the parity tests hold it against 1/sqrt in fp64)
  const float sc = gamma[c] * invstd;
  coef[c] = sc;
  coef[C + c] = beta[c] - mean * sc;
  coef[2 * C + c] = mean;
  coef[3 * C + c] = invstd;
  coef[4 * C + c] = unbf;
}

// Running-statistic updates that iic_bn_finalize was told to skip (running_mean == nullptr),
// applied later from the saved coefficients -- same arithmetic, bit for bit.  Multi-tensor.
#define BNRU_CHUNK 64
struct BnRunTable {
  const float* coef[BNRU_CHUNK];
  float* rm[BNRU_CHUNK];
  float* rv[BNRU_CHUNK];
  long long* nbt[BNRU_CHUNK];
  int C[BNRU_CHUNK];
};
__global__ __launch_bounds__(256) void bn_running_update_kernel(const BnRunTable t, float momentum) {
  const int i = blockIdx.y;
  const int C = t.C[i];
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && t.nbt[i]) *t.nbt[i] += 1;
  if (c >= C) return;
  const float* coef = t.coef[i];
  t.rm[i][c] = (1.f - momentum) * t.rm[i][c] + momentum * coef[2 * C + c];
  t.rv[i][c] = (1.f - momentum) * t.rv[i][c] + momentum * coef[4 * C + c];
}

// out = act( scale*y + shift [+ res] [+ scale2*y2 + shift2] ), grid = (N*H, ceil(W*C/8/256))
template <bool NTL>
__global__ __launch_bounds__(256) void bn_apply_kernel(
    const bf16_t* __restrict__ y, const float* __restrict__ coef, const bf16_t* __restrict__ res,
    const bf16_t* __restrict__ y2, const float* __restrict__ coef2, bf16_t* __restrict__ out,
    int H, int W, int P, int C, int relu) {
  const int c8n = C >> 3;
  const int item = blockIdx.y * blockDim.x + threadIdx.x;
  if (item >= W * c8n) return;
  const int xq = item / c8n, c8 = item - xq * c8n;
  const int n = blockIdx.x / H, yy = blockIdx.x - n * H;
  const int Hp = H + 2 * P, Wp = W + 2 * P;
  const long off = (((long)n * Hp + yy + P) * Wp + xq + P) * C + c8 * 8;
  float v[8], sc[8], sh[8];
  unpack8(ld16<NTL>(y + off), v);
  *reinterpret_cast<float4*>(sc) = *reinterpret_cast<const float4*>(coef + c8 * 8);
  *reinterpret_cast<float4*>(sc + 4) = *reinterpret_cast<const float4*>(coef + c8 * 8 + 4);
  *reinterpret_cast<float4*>(sh) = *reinterpret_cast<const float4*>(coef + C + c8 * 8);
  *reinterpret_cast<float4*>(sh + 4) = *reinterpret_cast<const float4*>(coef + C + c8 * 8 + 4);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = v[i] * sc[i] + sh[i];
  if (res) {
    float r[8];
    unpack8(ld16<NTL>(res + off), r);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += r[i];
  }
  if (y2) {
    float r[8];
    unpack8(ld16<NTL>(y2 + off), r);
    *reinterpret_cast<float4*>(sc) = *reinterpret_cast<const float4*>(coef2 + c8 * 8);
    *reinterpret_cast<float4*>(sc + 4) = *reinterpret_cast<const float4*>(coef2 + c8 * 8 + 4);
    *reinterpret_cast<float4*>(sh) = *reinterpret_cast<const float4*>(coef2 + C + c8 * 8);
    *reinterpret_cast<float4*>(sh + 4) = *reinterpret_cast<const float4*>(coef2 + C + c8 * 8 + 4);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += r[i] * sc[i] + sh[i];
  }
  if (relu) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
  }
  *reinterpret_cast<uint4*>(out + off) = pack8(v);
}

// sums[stripe][0][C] += sum g ; sums[stripe][1][C] += sum g*y   (g = dout * (act > 0))
// block = 256 threads: channel chunk = tid % (C/8), pixel lane = tid / (C/8); grid-stride over rows.
// MASK: 0 = g = dout, 1 = g = dout * (act > 0), 2 = g = dout * (scale*y + shift > 0) with the
// forward coefficients mcoef: act = relu(scale*y + shift) was produced by bn_apply with exactly
// this expression, so the mask is identical and the activation tensor is not read (one tensor
// less).  HAS2: second BatchNorm (downsample branch) sharing g.
template <int MASK, bool HAS2>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    const bf16_t* __restrict__ dout, const bf16_t* __restrict__ act, const bf16_t* __restrict__ y,
    const bf16_t* __restrict__ y2, float* __restrict__ sums, float* __restrict__ sums2,
    const float* __restrict__ mcoef, int N, int H, int W, int P, int C) {
  __shared__ float s_acc[256 * 8];
  const int c8n = C >> 3;
  const int PL = 256 / c8n;
  const int c8 = threadIdx.x % c8n, pl = threadIdx.x / c8n;
  const int Hp = H + 2 * P, Wp = W + 2 * P;
  float sg[8], sgy[8], sgy2[HAS2 ? 8 : 1], msc[MASK == 2 ? 8 : 1], msh[MASK == 2 ? 8 : 1];
#pragma unroll
  for (int i = 0; i < 8; ++i) sg[i] = sgy[i] = 0.f;
  if (HAS2) {
#pragma unroll
    for (int i = 0; i < 8; ++i) sgy2[HAS2 ? i : 0] = 0.f;
  }
  if (MASK == 2) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      msc[MASK == 2 ? i : 0] = mcoef[c8 * 8 + i];
      msh[MASK == 2 ? i : 0] = mcoef[C + c8 * 8 + i];
    }
  }
  const long rows = (long)N * H;
  for (long row = blockIdx.x; row < rows; row += gridDim.x) {
    const int n = (int)(row / H), yy = (int)(row - (long)n * H);
    const long rbase = (((long)n * Hp + yy + P) * Wp + P) * C + c8 * 8;
    for (int xq = pl; xq < W; xq += PL) {
      const long off = rbase + (long)xq * C;
      float g[8], v[8];
      unpack8(*reinterpret_cast<const uint4*>(dout + off), g);
      if (MASK == 1) {
        float a[8];
        unpack8(*reinterpret_cast<const uint4*>(act + off), a);
#pragma unroll
        for (int i = 0; i < 8; ++i) g[i] = a[i] > 0.f ? g[i] : 0.f;
      }
      unpack8(*reinterpret_cast<const uint4*>(y + off), v);
      if (MASK == 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          g[i] = (v[i] * msc[MASK == 2 ? i : 0] + msh[MASK == 2 ? i : 0]) > 0.f ? g[i] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) { sg[i] += g[i]; sgy[i] += g[i] * v[i]; }
      if (HAS2) {
        unpack8(*reinterpret_cast<const uint4*>(y2 + off), v);
#pragma unroll
        for (int i = 0; i < 8; ++i) sgy2[HAS2 ? i : 0] += g[i] * v[i];
      }
    }
  }
  const int stripe = blockIdx.x % IIC_STAT_STRIPES;
  // three LDS reductions over the pixel lanes (sum g, sum g*y, sum g*y2)
  for (int which = 0; which < (HAS2 ? 3 : 2); ++which) {
    const float* src = which == 0 ? sg : (which == 1 || !HAS2 ? sgy : sgy2);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) s_acc[(pl * c8n + c8) * 8 + i] = src[i];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      float t = 0.f;
      for (int p = 0; p < PL; ++p) t += s_acc[(p * c8n + (c >> 3)) * 8 + (c & 7)];
      if (which == 0) {
        iic_stat_add(sums, stripe, C, c, 0, t);
        if (HAS2) iic_stat_add(sums2, stripe, C, c, 0, t);
      } else if (which == 1) {
        iic_stat_add(sums, stripe, C, c, 1, t);
      } else {
        iic_stat_add(sums2, stripe, C, c, 1, t);
      }
    }
  }
}

// bcoef: [0]=c1 [1]=c2 [2]=c3 ; dy = c1*g + c2*y + c3
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(
    float* __restrict__ sums, const float* __restrict__ gamma, const float* __restrict__ coef,
    float* __restrict__ bcoef, float* __restrict__ dgamma, float* __restrict__ dbeta, int C, long count) {
  const int lane16 = threadIdx.x & 15;
  const int c = blockIdx.x * 16 + (threadIdx.x >> 4);
  double s, sy;
  iic_stat_collect(sums, IIC_STAT_STRIPES, C, c < C ? c : C - 1, lane16, s, sy);
  if (c >= C || lane16 != 0) return;
  const double mean = coef[2 * C + c], invstd = coef[3 * C + c];
  const double sgx = (sy - mean * s) * invstd;   // sum g * xhat
  const double c1 = (double)gamma[c] * invstd;
  const double c2 = -c1 * sgx * invstd / (double)count;
  const double c3 = -c1 * s / (double)count - c2 * mean;
  bcoef[c] = (float)c1;
  bcoef[C + c] = (float)c2;
  bcoef[2 * C + c] = (float)c3;
  if (dgamma) dgamma[c] = (float)sgx;
  if (dbeta) dbeta[c] = (float)s;
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const bf16_t* __restrict__ dout, const bf16_t* __restrict__ act, const bf16_t* __restrict__ y,
    const float* __restrict__ bcoef, bf16_t* __restrict__ dy, const bf16_t* __restrict__ y2,
    const float* __restrict__ bcoef2, bf16_t* __restrict__ dy2, const float* __restrict__ mcoef,
    int H, int W, int P, int C) {
  const int c8n = C >> 3;
  const int item = blockIdx.y * blockDim.x + threadIdx.x;
  if (item >= W * c8n) return;
  const int xq = item / c8n, c8 = item - xq * c8n;
  const int n = blockIdx.x / H, yy = blockIdx.x - n * H;
  const int Hp = H + 2 * P, Wp = W + 2 * P;
  const long off = (((long)n * Hp + yy + P) * Wp + xq + P) * C + c8 * 8;
  float g[8], v[8], o[8], k1[8], k2[8], k3[8];
  auto ld8 = [](const float* p, float (&d)[8]) {
    *reinterpret_cast<float4*>(d) = *reinterpret_cast<const float4*>(p);
    *reinterpret_cast<float4*>(d + 4) = *reinterpret_cast<const float4*>(p + 4);
  };
  unpack8(*reinterpret_cast<const uint4*>(dout + off), g);
  unpack8(*reinterpret_cast<const uint4*>(y + off), v);
  if (act) {
    float a[8];
    unpack8(*reinterpret_cast<const uint4*>(act + off), a);
#pragma unroll
    for (int i = 0; i < 8; ++i) g[i] = a[i] > 0.f ? g[i] : 0.f;
  } else if (mcoef) {      // mask from y (see bn_bwd_reduce_kernel)
    ld8(mcoef + c8 * 8, k1);
    ld8(mcoef + C + c8 * 8, k2);
#pragma unroll
    for (int i = 0; i < 8; ++i) g[i] = (v[i] * k1[i] + k2[i]) > 0.f ? g[i] : 0.f;
  }
  ld8(bcoef + c8 * 8, k1);
  ld8(bcoef + C + c8 * 8, k2);
  ld8(bcoef + 2 * C + c8 * 8, k3);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = k1[i] * g[i] + k2[i] * v[i] + k3[i];
  *reinterpret_cast<uint4*>(dy + off) = pack8(o);
  if (y2) {
    unpack8(*reinterpret_cast<const uint4*>(y2 + off), v);
    ld8(bcoef2 + c8 * 8, k1);
    ld8(bcoef2 + C + c8 * 8, k2);
    ld8(bcoef2 + 2 * C + c8 * 8, k3);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = k1[i] * g[i] + k2[i] * v[i] + k3[i];
    *reinterpret_cast<uint4*>(dy2 + off) = pack8(o);
  }
}

// ---------------------------------------------------------------------------------------------
// Second generation of the two backward passes (g_bn_v2, default on).  Same arithmetic and the
// same operation order per element as the kernels above; what changes is how work is laid out:
//   * a thread owns ONE 8-channel chunk and walks a contiguous run of interior pixels (all rows
//     flattened), so its per-channel coefficients live in registers -- the first version re-read 10
//     float4 of coefficients from L1 for every 16 bytes of data in bn_bwd_apply, and left 20 % of its
//     lanes idle in the second pass over a 49/25/13/7-pixel row;
//   * two pixels per iteration, loads first: 4-6 16-byte loads in flight per thread.
// Each block takes a contiguous chunk of pixels; threads of a block step through it PL pixels apart.
// (The forward apply was tried in the same layout and measured no faster -- 2 coefficient vectors
// per 2-3 data accesses are cheap enough from L1 -- so it keeps the row-per-block kernel.)
struct PxWalk {
  int x, yy;
  long off;
};
__device__ __forceinline__ PxWalk px_start(long q, int H, int W, int P, int C, int c8) {
  const long row = q / W;
  PxWalk w;
  w.x = (int)(q - row * W);
  const long n = row / H;
  w.yy = (int)(row - n * H);
  const int Hp = H + 2 * P, Wp = W + 2 * P;
  w.off = ((n * Hp + w.yy + P) * Wp + P + w.x) * C + c8 * 8;
  return w;
}
__device__ __forceinline__ void px_advance(PxWalk& w, int step, int H, int W, int P, int C) {
  w.x += step;
  w.off += (long)step * C;
  while (w.x >= W) {
    w.x -= W;
    w.off += 2L * P * C;                       // skip the right + left border to the next row
    if (++w.yy == H) {
      w.yy = 0;
      w.off += 2L * P * (W + 2 * P) * C;       // skip the bottom + top border rows
    }
  }
}

template <int MASK, bool HAS2, bool NTL>
__global__ __launch_bounds__(256) void bn_bwd_reduce2_kernel(
    const bf16_t* __restrict__ dout, const bf16_t* __restrict__ act, const bf16_t* __restrict__ y,
    const bf16_t* __restrict__ y2, float* __restrict__ sums, float* __restrict__ sums2,
    const float* __restrict__ mcoef, long npx, long per, int H, int W, int P, int C) {
  __shared__ float s_acc[256 * 8];
  const int c8n = C >> 3;
  const int PL = 256 / c8n;
  const int c8 = threadIdx.x % c8n, pl = threadIdx.x / c8n;
  float sg[8], sgy[8], sgy2[HAS2 ? 8 : 1], msc[MASK == 2 ? 8 : 1], msh[MASK == 2 ? 8 : 1];
#pragma unroll
  for (int i = 0; i < 8; ++i) sg[i] = sgy[i] = 0.f;
  if (HAS2) {
#pragma unroll
    for (int i = 0; i < 8; ++i) sgy2[HAS2 ? i : 0] = 0.f;
  }
  if (MASK == 2) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      msc[MASK == 2 ? i : 0] = mcoef[c8 * 8 + i];
      msh[MASK == 2 ? i : 0] = mcoef[C + c8 * 8 + i];
    }
  }
  const long q0 = (long)blockIdx.x * per;
  const long q1 = q0 + per < npx ? q0 + per : npx;
  auto accumulate = [&](const uint4 rg, const uint4 ra, const uint4 ry, const uint4 ry2) {
    float g[8], v[8];
    unpack8(rg, g);
    if (MASK == 1) {
      float a[8];
      unpack8(ra, a);
#pragma unroll
      for (int i = 0; i < 8; ++i) g[i] = a[i] > 0.f ? g[i] : 0.f;
    }
    unpack8(ry, v);
    if (MASK == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        g[i] = (v[i] * msc[MASK == 2 ? i : 0] + msh[MASK == 2 ? i : 0]) > 0.f ? g[i] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { sg[i] += g[i]; sgy[i] += g[i] * v[i]; }
    if (HAS2) {
      unpack8(ry2, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) sgy2[HAS2 ? i : 0] += g[i] * v[i];
    }
  };
  if (q0 + pl < q1) {
    PxWalk wa = px_start(q0 + pl, H, W, P, C, c8);
    long q = q0 + pl;
    for (; q + PL < q1; q += 2 * PL) {          // two pixels per iteration
      PxWalk wb = wa;
      px_advance(wb, PL, H, W, P, C);
      const uint4 z = make_uint4(0, 0, 0, 0);
      const uint4 g0 = ld16<NTL>(dout + wa.off);
      const uint4 g1 = ld16<NTL>(dout + wb.off);
      const uint4 y0 = ld16<NTL>(y + wa.off);
      const uint4 y1 = ld16<NTL>(y + wb.off);
      const uint4 a0 = MASK == 1 ? ld16<NTL>(act + wa.off) : z;
      const uint4 a1 = MASK == 1 ? ld16<NTL>(act + wb.off) : z;
      const uint4 t0 = HAS2 ? ld16<NTL>(y2 + wa.off) : z;
      const uint4 t1 = HAS2 ? ld16<NTL>(y2 + wb.off) : z;
      accumulate(g0, a0, y0, t0);
      accumulate(g1, a1, y1, t1);
      wa = wb;
      px_advance(wa, PL, H, W, P, C);
    }
    if (q < q1) {
      const uint4 z = make_uint4(0, 0, 0, 0);
      accumulate(ld16<NTL>(dout + wa.off),
                 MASK == 1 ? ld16<NTL>(act + wa.off) : z,
                 ld16<NTL>(y + wa.off),
                 HAS2 ? ld16<NTL>(y2 + wa.off) : z);
    }
  }
  const int stripe = blockIdx.x % IIC_STAT_STRIPES;
  for (int which = 0; which < (HAS2 ? 3 : 2); ++which) {
    const float* src = which == 0 ? sg : (which == 1 || !HAS2 ? sgy : sgy2);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) s_acc[(pl * c8n + c8) * 8 + i] = src[i];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      float t = 0.f;
      for (int p = 0; p < PL; ++p) t += s_acc[(p * c8n + (c >> 3)) * 8 + (c & 7)];
      if (which == 0) {
        iic_stat_add(sums, stripe, C, c, 0, t);
        if (HAS2) iic_stat_add(sums2, stripe, C, c, 0, t);
      } else if (which == 1) {
        iic_stat_add(sums, stripe, C, c, 1, t);
      } else {
        iic_stat_add(sums2, stripe, C, c, 1, t);
      }
    }
  }
}

template <int MASK, bool HAS2, bool NTL>
__global__ __launch_bounds__(256) void bn_bwd_apply2_kernel(
    const bf16_t* __restrict__ dout, const bf16_t* __restrict__ act, const bf16_t* __restrict__ y,
    const float* __restrict__ bcoef, bf16_t* __restrict__ dy, const bf16_t* __restrict__ y2,
    const float* __restrict__ bcoef2, bf16_t* __restrict__ dy2, const float* __restrict__ mcoef,
    long npx, long per, int H, int W, int P, int C) {
  const int c8n = C >> 3;
  const int PL = 256 / c8n;
  const int c8 = threadIdx.x % c8n, pl = threadIdx.x / c8n;
  float k1[8], k2[8], k3[8], m1[MASK == 2 ? 8 : 1], m2[MASK == 2 ? 8 : 1];
  float j1[HAS2 ? 8 : 1], j2[HAS2 ? 8 : 1], j3[HAS2 ? 8 : 1];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    k1[i] = bcoef[c8 * 8 + i];
    k2[i] = bcoef[C + c8 * 8 + i];
    k3[i] = bcoef[2 * C + c8 * 8 + i];
    if (MASK == 2) {
      m1[MASK == 2 ? i : 0] = mcoef[c8 * 8 + i];
      m2[MASK == 2 ? i : 0] = mcoef[C + c8 * 8 + i];
    }
    if (HAS2) {
      j1[HAS2 ? i : 0] = bcoef2[c8 * 8 + i];
      j2[HAS2 ? i : 0] = bcoef2[C + c8 * 8 + i];
      j3[HAS2 ? i : 0] = bcoef2[2 * C + c8 * 8 + i];
    }
  }
  const long q0 = (long)blockIdx.x * per;
  const long q1 = q0 + per < npx ? q0 + per : npx;
  auto emit = [&](long off, const uint4 rg, const uint4 ra, const uint4 ry, const uint4 ry2) {
    float g[8], v[8], o[8];
    unpack8(rg, g);
    unpack8(ry, v);
    if (MASK == 1) {
      float a[8];
      unpack8(ra, a);
#pragma unroll
      for (int i = 0; i < 8; ++i) g[i] = a[i] > 0.f ? g[i] : 0.f;
    } else if (MASK == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        g[i] = (v[i] * m1[MASK == 2 ? i : 0] + m2[MASK == 2 ? i : 0]) > 0.f ? g[i] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = k1[i] * g[i] + k2[i] * v[i] + k3[i];
    *reinterpret_cast<uint4*>(dy + off) = pack8(o);
    if (HAS2) {
      unpack8(ry2, v);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        o[i] = j1[HAS2 ? i : 0] * g[i] + j2[HAS2 ? i : 0] * v[i] + j3[HAS2 ? i : 0];
      *reinterpret_cast<uint4*>(dy2 + off) = pack8(o);
    }
  };
  if (q0 + pl >= q1) return;
  PxWalk wa = px_start(q0 + pl, H, W, P, C, c8);
  long q = q0 + pl;
  const uint4 z = make_uint4(0, 0, 0, 0);
  for (; q + PL < q1; q += 2 * PL) {
    PxWalk wb = wa;
    px_advance(wb, PL, H, W, P, C);
    const uint4 g0 = ld16<NTL>(dout + wa.off);
    const uint4 g1 = ld16<NTL>(dout + wb.off);
    const uint4 y0 = ld16<NTL>(y + wa.off);
    const uint4 y1 = ld16<NTL>(y + wb.off);
    const uint4 a0 = MASK == 1 ? ld16<NTL>(act + wa.off) : z;
    const uint4 a1 = MASK == 1 ? ld16<NTL>(act + wb.off) : z;
    const uint4 t0 = HAS2 ? ld16<NTL>(y2 + wa.off) : z;
    const uint4 t1 = HAS2 ? ld16<NTL>(y2 + wb.off) : z;
    emit(wa.off, g0, a0, y0, t0);
    emit(wb.off, g1, a1, y1, t1);
    wa = wb;
    px_advance(wa, PL, H, W, P, C);
  }
  if (q < q1)
    emit(wa.off, ld16<NTL>(dout + wa.off),
         MASK == 1 ? ld16<NTL>(act + wa.off) : z,
         ld16<NTL>(y + wa.off),
         HAS2 ? ld16<NTL>(y2 + wa.off) : z);
}

// 0: first-generation backward kernels; 1 (default): second generation where it measured faster
// (tools/bn_perf.py: every reduce, and the apply unless it reads the activation tensor for its mask
// -- with 3 reads + 1 write per element the deeper prefetch costs occupancy: 161 vs 158 us at
// layer1, against 122 vs 145 us for the mask-from-y apply); 2: second generation everywhere (tests).
#ifdef IIC_DEBUG_HOOKS
static int g_bn_v2 = 1;
static int g_bn_v2_blocks = 1024;
#else
static constexpr int g_bn_v2 = 1;
static constexpr int g_bn_v2_blocks = 1024;
#endif
// 1 (default): the passes' streaming loads carry the non-temporal hint (see ld16); 0: plain loads (A/B runs)
IIC_SWITCH(g_bn_nt, 1, iic_debug_bn_nt)

// chunk of pixels per block: a multiple of 2*PL so that every thread's pair loop stays aligned
// reduce != 0: the reduction kernels end with 2C (3C) integer atomics per block into the exact
// statistic accumulators -- at C >= 256 a 1024-block grid spends more time in those than in its
// HBM stream (measured at 7x7x512: 48.8 us with 1024 blocks, 21.7 us with 256; tools/bn_perf.py)
static void bn_v2_grid(long npx, int C, long& per, int& grid, int reduce = 0) {
  const int PL = 256 / (C >> 3);
  int blocks = g_bn_v2_blocks;
  if (reduce && g_bn_v2_blocks == 1024) blocks = C <= 128 ? 512 : 256;
  long p = (npx + blocks - 1) / blocks;
  p = (p + 2 * PL - 1) / (2 * PL) * (2 * PL);
  per = p;
  grid = (int)((npx + p - 1) / p);
}

extern "C" {

int iic_bn_finalize(float* stats, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, long long* num_batches_tracked, float* coef, int C,
                    long count, long unbiased_count, float eps, float momentum, int training,
                    void* stream) {
  if (!gamma || !beta || !coef || C <= 0) return IIC_ERR_ARG;
  if (training && (!stats || count <= 0)) return IIC_ERR_ARG;
  // the statistic cells are read AND zeroed by 16-channel groups: a ragged last group would race with the
  // clamped lanes of its neighbours (every network on the path has C % 64 == 0)
  if (training && (C & 15)) return IIC_ERR_ARG;
  if (!training && (!running_mean || !running_var)) return IIC_ERR_ARG;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)stream,
                     stats, gamma, beta, running_mean, running_var, num_batches_tracked, coef, C,
                     count, unbiased_count > 0 ? unbiased_count : count, eps, momentum, training);
  return iic_launch_status();
}

long iic_stat_bytes(int C) {
  return C > 0 ? (long)IIC_STAT_STRIPES * C * 2 * IIC_STAT_BINS * (long)sizeof(iic_stat_t) : 0;
}

int iic_bn_running_update(int n, const float* const* coef, float* const* running_mean,
                          float* const* running_var, long long* const* num_batches_tracked,
                          const int* C, float momentum, void* stream) {
  if (n <= 0 || !coef || !running_mean || !running_var || !num_batches_tracked || !C) return IIC_ERR_ARG;
  // one BatchNorm may appear several times (two forwards): chunks run in order on the stream, and
  // within a chunk duplicates would race -- start a new chunk at a repeated running_mean
  int base = 0;
  while (base < n) {
    BnRunTable t;
    int cnt = 0, maxC = 0;
    while (base + cnt < n && cnt < BNRU_CHUNK) {
      const int j = base + cnt;
      bool dup = false;
      for (int k = 0; k < cnt; ++k) dup = dup || t.rm[k] == running_mean[j];
      if (dup) break;
      if (!coef[j] || !running_mean[j] || !running_var[j] || C[j] <= 0) return IIC_ERR_ARG;
      t.coef[cnt] = coef[j]; t.rm[cnt] = running_mean[j]; t.rv[cnt] = running_var[j];
      t.nbt[cnt] = num_batches_tracked[j]; t.C[cnt] = C[j];
      if (C[j] > maxC) maxC = C[j];
      ++cnt;
    }
    for (int k = cnt; k < BNRU_CHUNK; ++k) { t.coef[k] = nullptr; t.rm[k] = nullptr; t.rv[k] = nullptr; t.nbt[k] = nullptr; t.C[k] = 0; }
    hipLaunchKernelGGL(bn_running_update_kernel, dim3((maxC + 255) / 256, cnt), dim3(256), 0,
                       (hipStream_t)stream, t, momentum);
    base += cnt;
  }
  return iic_launch_status();
}

static int check_c(int C) { return (C % 64 == 0 && 256 % (C / 8) == 0 && C <= 2048) ? 0 : 1; }

int iic_bn_apply(const void* y, const float* coef, const void* res, const void* y2,
                 const float* coef2, void* out, int N, int H, int W, int P, int C, int relu,
                 void* stream) {
  if (!y || !coef || !out || N <= 0 || H <= 0 || W <= 0) return IIC_ERR_ARG;
  if (C % 8 != 0) return IIC_ERR_UNSUPPORTED;
  if ((y2 == nullptr) != (coef2 == nullptr)) return IIC_ERR_ARG;
  dim3 grid(N * H, (W * (C / 8) + 255) / 256);
#define BN_APPLY_LAUNCH(L_)                                                                                 \
  hipLaunchKernelGGL((bn_apply_kernel<L_>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)y, coef, \
                     (const bf16_t*)res, (const bf16_t*)y2, coef2, (bf16_t*)out, H, W, P, C, relu)
  if (g_bn_nt) BN_APPLY_LAUNCH(true); else BN_APPLY_LAUNCH(false);
  return iic_launch_status();
}

int iic_bn_bwd_reduce(const void* dout, const void* act, const void* y, const void* y2, float* sums,
                      float* sums2, const float* mask_coef, int N, int H, int W, int P, int C,
                      void* stream) {
  if (!dout || !y || !sums || N <= 0 || (act && mask_coef)) return IIC_ERR_ARG;
  if (check_c(C)) return IIC_ERR_UNSUPPORTED;
  if ((y2 == nullptr) != (sums2 == nullptr)) return IIC_ERR_ARG;
  const int mode = act ? 1 : (mask_coef ? 2 : 0);
  if (g_bn_v2) {
    long per;
    int grid2;
    bn_v2_grid((long)N * H * W, C, per, grid2, 1);
#define BN_RED2_LAUNCH_(M_, H2_, L_)                                                            \
  hipLaunchKernelGGL((bn_bwd_reduce2_kernel<M_, H2_, L_>), dim3(grid2), dim3(256), 0,           \
                     (hipStream_t)stream, (const bf16_t*)dout, (const bf16_t*)act,              \
                     (const bf16_t*)y, (const bf16_t*)y2, sums, sums2, mask_coef,               \
                     (long)N * H * W, per, H, W, P, C)
#define BN_RED2_LAUNCH(M_, H2_)                                  \
  do {                                                           \
    if (g_bn_nt) BN_RED2_LAUNCH_(M_, H2_, true);                 \
    else BN_RED2_LAUNCH_(M_, H2_, false);                        \
  } while (0)
    if (y2) {
      if (mode == 1) BN_RED2_LAUNCH(1, true); else if (mode == 2) BN_RED2_LAUNCH(2, true); else BN_RED2_LAUNCH(0, true);
    } else {
      if (mode == 1) BN_RED2_LAUNCH(1, false); else if (mode == 2) BN_RED2_LAUNCH(2, false); else BN_RED2_LAUNCH(0, false);
    }
    return iic_launch_status();
  }
  long rows = (long)N * H;
  int grid = (int)(rows < 2048 ? rows : 2048);
#define BN_RED_LAUNCH(M_, H2_)                                                                  \
  hipLaunchKernelGGL((bn_bwd_reduce_kernel<M_, H2_>), dim3(grid), dim3(256), 0,                 \
                     (hipStream_t)stream, (const bf16_t*)dout, (const bf16_t*)act,              \
                     (const bf16_t*)y, (const bf16_t*)y2, sums, sums2, mask_coef, N, H, W, P, C)
  if (y2) {
    if (mode == 1) BN_RED_LAUNCH(1, true); else if (mode == 2) BN_RED_LAUNCH(2, true); else BN_RED_LAUNCH(0, true);
  } else {
    if (mode == 1) BN_RED_LAUNCH(1, false); else if (mode == 2) BN_RED_LAUNCH(2, false); else BN_RED_LAUNCH(0, false);
  }
  return iic_launch_status();
}

int iic_bn_bwd_finalize(float* sums, const float* gamma, const float* coef, float* bcoef,
                        float* dgamma, float* dbeta, int C, long count, void* stream) {
  if (!sums || !gamma || !coef || !bcoef || C <= 0 || count <= 0 || (C & 15)) return IIC_ERR_ARG;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0,
                     (hipStream_t)stream, sums, gamma, coef, bcoef, dgamma, dbeta, C, count);
  return iic_launch_status();
}

int iic_bn_bwd_apply(const void* dout, const void* act, const void* y, const float* bcoef, void* dy,
                     const void* y2, const float* bcoef2, void* dy2, const float* mask_coef, int N,
                     int H, int W, int P, int C, void* stream) {
  if (!dout || !y || !bcoef || !dy || N <= 0 || (act && mask_coef)) return IIC_ERR_ARG;
  if (C % 8 != 0) return IIC_ERR_UNSUPPORTED;
  if ((y2 == nullptr) != (bcoef2 == nullptr) || (y2 == nullptr) != (dy2 == nullptr))
    return IIC_ERR_ARG;
  if (g_bn_v2 && !check_c(C) && (g_bn_v2 == 2 || !act)) {
    const int mode = act ? 1 : (mask_coef ? 2 : 0);
    long per;
    int grid2;
    bn_v2_grid((long)N * H * W, C, per, grid2);
#define BN_APP2_LAUNCH_(M_, H2_, L_)                                                             \
  hipLaunchKernelGGL((bn_bwd_apply2_kernel<M_, H2_, L_>), dim3(grid2), dim3(256), 0,            \
                     (hipStream_t)stream, (const bf16_t*)dout, (const bf16_t*)act,              \
                     (const bf16_t*)y, bcoef, (bf16_t*)dy, (const bf16_t*)y2, bcoef2,           \
                     (bf16_t*)dy2, mask_coef, (long)N * H * W, per, H, W, P, C)
#define BN_APP2_LAUNCH(M_, H2_)                                  \
  do {                                                           \
    if (g_bn_nt) BN_APP2_LAUNCH_(M_, H2_, true);                 \
    else BN_APP2_LAUNCH_(M_, H2_, false);                        \
  } while (0)
    if (y2) {
      if (mode == 1) BN_APP2_LAUNCH(1, true); else if (mode == 2) BN_APP2_LAUNCH(2, true); else BN_APP2_LAUNCH(0, true);
    } else {
      if (mode == 1) BN_APP2_LAUNCH(1, false); else if (mode == 2) BN_APP2_LAUNCH(2, false); else BN_APP2_LAUNCH(0, false);
    }
    return iic_launch_status();
  }
  dim3 grid(N * H, (W * (C / 8) + 255) / 256);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, grid, dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dout, (const bf16_t*)act, (const bf16_t*)y, bcoef, (bf16_t*)dy,
                     (const bf16_t*)y2, bcoef2, (bf16_t*)dy2, mask_coef, H, W, P, C);
  return iic_launch_status();
}

#ifdef IIC_DEBUG_HOOKS
/* A/B switches for tools/bn_perf.py and the kernel-generation test (instrumented library only) */
IIC_HOOK void iic_debug_bn_v2(int on, int blocks) {
  g_bn_v2 = on;
  if (blocks > 0) g_bn_v2_blocks = blocks;
}
#endif

}  // extern "C"

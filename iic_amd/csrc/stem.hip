// ClusterNet5g stem for gfx950: conv3x3(Cin<=5 -> 64, pad 1) + BatchNorm + ReLU +
// MaxPool(k2, s2, p1), plus the Sobel pre-op.
//
// Replaces /root/reference/code/archs/cluster/net5g.py:21-26,42-45 and
//          /root/reference/code/utils/cluster/transforms.py:47-96 (sobel_process).
//
// The stem conv output (N x 96 x 96 x 64) is 20 % of all activation elements of the network
// but costs only K = Cin*9 <= 45 MACs per output, so it is never written to HBM: every pass
// that needs it RECOMPUTES it from the fp32 NCHW input (24-73 KB per image, L2 resident) on
// the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32, bitwise an fmaf chain):
//   stem_stats      conv -> per-channel sum / sum^2                       (BN batch stats)
//   stem_apply_pool conv -> BN -> ReLU -> 2x2/2 max-pool -> PT bf16 [N][Ho+2][Wo+2][64]
//   stem_bwd_reduce conv -> BN/ReLU/pool-argmax routing of dpool -> sum g, sum g*y
//   stem_bwd_wgrad  conv -> dy = c1*g + c2*y + c3 -> dW[64][K] += dy^T . patches (MFMA)
// HBM traffic per pass = the input (+ the pooled tensor once), instead of 4 passes over a
// 1.5 GB tensor.  (bwd-data is not needed: the input image has no gradient.)
//
// MFMA operand maps (32x32x2 f32): A[i = lane&31][k = lane>>5], B[k = lane>>5][j = lane&31],
// D[row = (r&3)+8*(r>>2)+4*(lane>>5)][col = lane&31].
#include "common.h"
#include "../../include/iic_hip.h"

#include "stem_common.h"

// The KS input values this lane contributes to the 32-pixel tile (row y, cols x0..x0+31) of one image, loaded
// UNCONDITIONALLY (out-of-image taps read element 0 and are zeroed by a select): with the load under the bounds
// check the compiler emitted branch -> load -> s_waitcnt vmcnt(0) -> 2 MFMAs per k-step, i.e. nine exposed memory
// latencies per tile (SQ: 44 % of the waves' cycles parked on s_waitcnt, matrix pipe 35 % busy).  Separate from
// the MFMAs so that callers can have the NEXT tile's values in flight while this tile multiplies.
template <int CIN>
__device__ __forceinline__ void stem_load_patch(const float* __restrict__ xin, int H, int W, int y, int x0,
                                                int lane, float (&av)[StemK<CIN>::KS]) {
  constexpr int K = StemK<CIN>::K;
  const int px = x0 + (lane & 31);
  const bool hi = (lane >> 5) != 0;
#pragma unroll
  for (int s = 0; s < StemK<CIN>::KS; ++s) {
    const int k0 = 2 * s, k1 = 2 * s + 1;
    const int c = hi ? (k1 / 9) : (k0 / 9);
    const int dy = (hi ? ((k1 % 9) / 3) : ((k0 % 9) / 3)) - 1;
    const int dx = (hi ? (k1 % 3) : (k0 % 3)) - 1;
    const bool kval = hi ? (k1 < K) : true;
    const int yy = y + dy, xx = px + dx;
    const bool ok = kval && yy >= 0 && yy < H && xx >= 0 && xx < W;
    const float v = xin[ok ? ((long)c * H + yy) * W + xx : 0];
    av[s] = ok ? v : 0.f;
  }
}
template <int CIN>
__device__ __forceinline__ void stem_mma(const float (&av)[StemK<CIN>::KS], const float (&wr)[2][StemK<CIN>::KS],
                                         f32x16 (&acc)[2]) {
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[h][r] = 0.f;
#pragma unroll
  for (int s = 0; s < StemK<CIN>::KS; ++s) {
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], wr[0][s], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], wr[1][s], acc[1], 0, 0, 0);
  }
}
// conv outputs for 32 pixels (row y, cols x0..x0+31) x 64 couts of one image.
template <int CIN>
__device__ __forceinline__ void stem_conv_tile(const float* __restrict__ xin, int H, int W, int y,
                                               int x0, int lane,
                                               const float (&wr)[2][StemK<CIN>::KS],
                                               f32x16 (&acc)[2]) {
  float av[StemK<CIN>::KS];
  stem_load_patch<CIN>(xin, H, W, y, x0, lane, av);
  stem_mma<CIN>(av, wr, acc);
}

// ------------------------------------------------------------------------------------
// 1. batch statistics.  persistent grid, 4 waves / block, one 32-pixel tile per wave step.
// ------------------------------------------------------------------------------------
template <int CIN>
__global__ __launch_bounds__(256) void stem_stats_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ w,
                                                         float* __restrict__ stats, int N, int H,
                                                         int W) {
  __shared__ float s_red[4][2][2][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float wr[2][StemK<CIN>::KS];
  stem_load_w<CIN>(w, lane, wr);
  const int nseg = (W + 31) / 32;
  const long tiles = (long)N * H * nseg;
  float s[2] = {0.f, 0.f}, ss[2] = {0.f, 0.f};
  for (long t = (long)blockIdx.x * 4 + wave; t < tiles; t += (long)gridDim.x * 4) {
    const int seg = (int)(t % nseg);
    const long row = t / nseg;
    const int y = (int)(row % H), n = (int)(row / H);
    f32x16 acc[2];
    // (loading the NEXT tile's values before this tile's MFMAs was measured slower: 223 vs 208 us)
    stem_conv_tile<CIN>(x + (long)n * CIN * H * W, H, W, y, seg * 32, lane, wr, acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool ok = seg * 32 + mfma32_row(r, lane) < W;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float v = ok ? acc[h][r] : 0.f;
        s[h] += v;
        ss[h] += v * v;
      }
    }
  }
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
    const int which = threadIdx.x >> 6, ch = threadIdx.x & 63;   // 0: sum, 1: sumsq
    float t = 0.f;
    for (int wv = 0; wv < 4; ++wv) t += s_red[wv][ch >> 5][which][ch & 31];
    iic_stat_add(stats, blockIdx.x % IIC_STAT_STRIPES, STEM_CO, ch, which, t);
  }
}

// ------------------------------------------------------------------------------------
// 2. apply + pool.  block = (n, ho), one wave per 32-column segment, two conv rows.
//    LDS: post-ReLU activations [2][W][64] bf16.
// ------------------------------------------------------------------------------------
template <int CIN>
__global__ __launch_bounds__(512) void stem_apply_pool_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                       const float* __restrict__ coef, bf16_t* __restrict__ out,
                                       int N, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* sP = reinterpret_cast<bf16_t*>(smem_raw);   // [2][W][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Ho = H / 2 + 1, Wo = W / 2 + 1;
  const int n = blockIdx.x / Ho, ho = blockIdx.x - n * Ho;
  float wr[2][StemK<CIN>::KS];
  stem_load_w<CIN>(w, lane, wr);
  const int ch0 = lane & 31;
  const float sc0 = coef[ch0], sc1 = coef[ch0 + 32];
  const float sh0 = coef[STEM_CO + ch0], sh1 = coef[STEM_CO + ch0 + 32];
#pragma unroll
  for (int rs = 0; rs < 2; ++rs) {
    const int y = 2 * ho - 1 + rs;
    if (y < 0 || y >= H) continue;   // uniform per block
    f32x16 acc[2];
    stem_conv_tile<CIN>(x + (long)n * CIN * H * W, H, W, y, wave * 32, lane, wr, acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int px = wave * 32 + mfma32_row(r, lane);
      if (px < W) {
        bf16_t* d = sP + ((long)rs * W + px) * STEM_CO;
        d[ch0] = f32_to_bf16(fmaxf(acc[0][r] * sc0 + sh0, 0.f));
        d[ch0 + 32] = f32_to_bf16(fmaxf(acc[1][r] * sc1 + sh1, 0.f));
      }
    }
  }
  __syncthreads();
  const bool v0 = (2 * ho - 1) >= 0, v1 = (2 * ho) < H;
  for (int item = threadIdx.x; item < Wo * 8; item += blockDim.x) {
    const int wo = item >> 3, c8 = item & 7;
    float m[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = 0.f;   // post-ReLU values are >= 0: 0 == -inf padding
#pragma unroll
    for (int rs = 0; rs < 2; ++rs) {
      if (!(rs == 0 ? v0 : v1)) continue;
#pragma unroll
      for (int cs = 0; cs < 2; ++cs) {
        const int cx = 2 * wo - 1 + cs;
        if (cx < 0 || cx >= W) continue;
        const uint4 v = *reinterpret_cast<const uint4*>(sP + ((long)rs * W + cx) * STEM_CO + c8 * 8);
        const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          m[2 * i] = fmaxf(m[2 * i], bf16lo(vv[i]));
          m[2 * i + 1] = fmaxf(m[2 * i + 1], bf16hi(vv[i]));
        }
      }
    }
    const long o = (((long)n * (Ho + 2) + ho + 1) * (Wo + 2) + wo + 1) * STEM_CO + c8 * 8;
    *reinterpret_cast<uint4*>(out + o) =
        make_uint4(pack_bf16x2(m[0], m[1]), pack_bf16x2(m[2], m[3]), pack_bf16x2(m[4], m[5]),
                   pack_bf16x2(m[6], m[7]));
  }
}

// ------------------------------------------------------------------------------------
// 3/4. backward.  persistent over (n, ho) items; LDS: conv outputs y [2][W][64] fp32.
//   MODE 0: sums[stripe][0][64] += sum g ; [1] += sum g*y
//   MODE 1: dy = c1*g + c2*y + c3 (in LDS), dWpart[block][64][K] += dy^T . patch  (bf16 MFMA)
//   MODE 2 (the product path): BOTH in one recompute pass.  dy is affine in (g, y) with
//     per-channel coefficients, so dW[co][k] = c1[co]*G1 + c2[co]*G2 + c3[co]*G3[k] with
//     G1 = sum g*patch, G2 = sum y*patch, G3 = sum patch over valid pixels: the three GEMMs
//     need no coefficient and run next to the sum g / sum g*y reduction; the coefficients are
//     applied to the 64 x K results afterwards (stem_wgrad_combine_kernel).
// ------------------------------------------------------------------------------------
template <int CIN, int MODE>
__global__ __launch_bounds__(512) void stem_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                const float* __restrict__ coef, const float* __restrict__ bcoef,
                                const bf16_t* __restrict__ dpool, float* __restrict__ outbuf,
                                float* __restrict__ sums2, int N, int H, int W) {
  constexpr int K = StemK<CIN>::K;
  constexpr int NKT = StemK<CIN>::NKT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* sY = reinterpret_cast<float*>(smem_raw);   // [2][W][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int Ho = H / 2 + 1, Wo = W / 2 + 1;
  // MODE 1: dy as bf16, TRANSPOSED [64 co][2 rows x WP pixels] so that an MFMA A fragment
  // (8 consecutive pixels of one channel) is one 16-B LDS read; and the 4 input rows the two
  // conv rows touch, zero padded, as fp32 [CIN][4][WX].
  const int WP = (W + 15) & ~15;
  const int PD = 2 * WP + 8;   // = 8 * odd (WP is a multiple of 16): conflict-free 16-B rows
  const int WX = WP + 10;
  bf16_t* sDt = reinterpret_cast<bf16_t*>(smem_raw + (size_t)2 * W * STEM_CO * sizeof(float));
  float* sXr = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(sDt) +
                                        (((size_t)STEM_CO * PD * 2 + 15) & ~(size_t)15));
  if (MODE >= 1) {
    for (int i = threadIdx.x; i < STEM_CO * PD / 2; i += blockDim.x)
      reinterpret_cast<uint32_t*>(sDt)[i] = 0u;      // pad pixels stay zero forever
  }
  float wr[2][StemK<CIN>::KS];
  stem_load_w<CIN>(w, lane, wr);
  const int ch0 = lane & 31;
  const int c8 = threadIdx.x & 7;   // blockDim % 8 == 0: a thread always owns the same 8 channels
  // per-channel coefficients live in LDS (keeps the MFMA accumulators out of scratch)
  __shared__ float s_cf[5][STEM_CO];
  for (int i = threadIdx.x; i < STEM_CO; i += blockDim.x) {
    s_cf[0][i] = coef[i];
    s_cf[1][i] = coef[STEM_CO + i];
    s_cf[2][i] = MODE == 1 ? bcoef[i] : 0.f;
    s_cf[3][i] = MODE == 1 ? bcoef[STEM_CO + i] : 0.f;
    s_cf[4][i] = MODE == 1 ? bcoef[2 * STEM_CO + i] : 0.f;
  }
  float sg[8], sgy[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) sg[i] = sgy[i] = 0.f;
  f32x16 dacc[2][NKT];                      // MODE 1: dW;  MODE 2: G1 (g . patch)
  f32x16 daccy[MODE == 2 ? 2 : 1][NKT];     // MODE 2: G2 (y . patch)
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 16; ++r) dacc[h][t][r] = 0.f;
#pragma unroll
    for (int h = 0; h < (MODE == 2 ? 2 : 1); ++h)
#pragma unroll
      for (int r = 0; r < 16; ++r) daccy[h][t][r] = 0.f;
  }

  const long items = (long)N * Ho;
  for (long it = blockIdx.x; it < items; it += gridDim.x) {
    const int n = (int)(it / Ho), ho = (int)(it - (long)n * Ho);
    const float* xin = x + (long)n * CIN * H * W;
    __syncthreads();   // previous item's LDS fully consumed
#pragma unroll
    for (int rs = 0; rs < 2; ++rs) {
      const int y = 2 * ho - 1 + rs;
      if (y < 0 || y >= H) continue;
      f32x16 acc[2];
      stem_conv_tile<CIN>(xin, H, W, y, wave * 32, lane, wr, acc);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int px = wave * 32 + mfma32_row(r, lane);
        if (px < W) {
          float* d = sY + ((long)rs * W + px) * STEM_CO;
          d[ch0] = acc[0][r];
          d[ch0 + 32] = acc[1][r];
        }
      }
    }
    __syncthreads();
    const bool rv[2] = {(2 * ho - 1) >= 0, (2 * ho) < H};
    for (int item = threadIdx.x; item < Wo * 8; item += blockDim.x) {
      const int wo = item >> 3;   // (item & 7) == c8
      const uint4 gv = *reinterpret_cast<const uint4*>(
          dpool + (((long)n * (Ho + 2) + ho + 1) * (Wo + 2) + wo + 1) * STEM_CO + c8 * 8);
      const uint32_t gg[4] = {gv.x, gv.y, gv.z, gv.w};
      bool valid[4];
      int lidx[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rs = q >> 1, cx = 2 * wo - 1 + (q & 1);
        valid[q] = rv[rs] && cx >= 0 && cx < W;
        lidx[q] = (rs * W + (valid[q] ? cx : 0)) * STEM_CO + c8 * 8;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float yv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) yv[q] = valid[q] ? sY[lidx[q] + i] : 0.f;
        const int am = window_argmax(yv, valid, s_cf[0][c8 * 8 + i], s_cf[1][c8 * 8 + i]);
        const float g = (i & 1) ? bf16hi(gg[i >> 1]) : bf16lo(gg[i >> 1]);
        if (MODE == 0 || MODE == 2) {
          if (am >= 0) {
            sg[i] += g;
            sgy[i] += g * yv[am];
          }
        }
        if (MODE == 2) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (valid[q])
              sDt[(c8 * 8 + i) * PD + (q >> 1) * WP + (2 * wo - 1 + (q & 1))] =
                  f32_to_bf16(q == am ? g : 0.f);
        }
        if (MODE == 1) {
          const float cb1 = s_cf[2][c8 * 8 + i], cb2 = s_cf[3][c8 * 8 + i], cb3 = s_cf[4][c8 * 8 + i];
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (valid[q])
              sDt[(c8 * 8 + i) * PD + (q >> 1) * WP + (2 * wo - 1 + (q & 1))] =
                  f32_to_bf16(cb1 * (q == am ? g : 0.f) + cb2 * yv[q] + cb3);
        }
      }
    }
    if (MODE >= 1) {
      // input rows 2ho-2 .. 2ho+1 (zero outside the image), columns -1 .. WP+8
      for (int idx = threadIdx.x; idx < CIN * 4 * WX; idx += blockDim.x) {
        const int c = idx / (4 * WX), r = (idx / WX) & 3, xx = idx % WX;
        const int yy = 2 * ho - 2 + r, xg = xx - 1;
        sXr[idx] = (yy >= 0 && yy < H && xg >= 0 && xg < W) ? xin[((long)c * H + yy) * W + xg] : 0.f;
      }
      __syncthreads();
      // dW[co][k] += sum_pix dy[pix][co] * patch[pix][k] on bf16 MFMA (32x32x16, fp32 accumulate):
      //   A[i = co][8 pixels] from sDt, B[8 pixels][j = k] built from sXr.  Wave w owns the pixel
      //   steps {2w, 2w+1} (its 32 columns) of both rows.
      const int i = lane & 31, g5 = lane >> 5;
#pragma unroll
      for (int rs = 0; rs < 2; ++rs) {
        if (!rv[rs]) continue;
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx) {
          const int p0 = (wave * 2 + sidx) * 16 + 8 * g5;      // this lane's 8 pixels start here
          if ((wave * 2 + sidx) * 16 >= WP) continue;
          const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(sDt + (long)i * PD + rs * WP + p0);
          const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(sDt + (long)(i + 32) * PD + rs * WP + p0);
          bf16x8 ay0, ay1;
          if (MODE == 2) {      // y operand straight from the fp32 conv rows (pixel-strided reads)
            union { bf16x8 v; uint32_t u[4]; } y0, y1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int pa = p0 + 2 * e, pb = pa + 1;
              const float* ya = sY + ((long)rs * W + (pa < W ? pa : 0)) * STEM_CO;
              const float* yb = sY + ((long)rs * W + (pb < W ? pb : 0)) * STEM_CO;
              y0.u[e] = pack_bf16x2(pa < W ? ya[i] : 0.f, pb < W ? yb[i] : 0.f);
              y1.u[e] = pack_bf16x2(pa < W ? ya[i + 32] : 0.f, pb < W ? yb[i + 32] : 0.f);
            }
            ay0 = y0.v;
            ay1 = y1.v;
          }
#pragma unroll
          for (int t = 0; t < NKT; ++t) {
            const int k = t * 32 + i;
            union { bf16x8 v; uint32_t u[4]; } bb;
            bb.u[0] = bb.u[1] = bb.u[2] = bb.u[3] = 0u;
            if (k < K) {
              const int c = k / 9, kh = (k % 9) / 3, kw = k % 3;
              // conv row y = 2ho-1+rs reads input row y+kh-1 = (2ho-2) + rs + kh; column px+kw-1
              const float* xr = sXr + (c * 4 + rs + kh) * WX + p0 + kw;
#pragma unroll
              for (int e = 0; e < 4; ++e) bb.u[e] = pack_bf16x2(xr[2 * e], xr[2 * e + 1]);
            }
            dacc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bb.v, dacc[0][t], 0, 0, 0);
            dacc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bb.v, dacc[1][t], 0, 0, 0);
            if (MODE == 2) {
              daccy[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ay0, bb.v, daccy[0][t], 0, 0, 0);
              daccy[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ay1, bb.v, daccy[1][t], 0, 0, 0);
            }
          }
        }
      }
    }
  }

  __syncthreads();
  if (MODE == 0 || MODE == 2) {
    float* red = sY;   // [blockDim][16]
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      red[threadIdx.x * 16 + i] = sg[i];
      red[threadIdx.x * 16 + 8 + i] = sgy[i];
    }
    __syncthreads();
    float* sums = MODE == 2 ? sums2 : outbuf;
    for (int o = threadIdx.x; o < 128; o += blockDim.x) {   // blockDim may be a single wave
      const int which = o >> 6, ch = o & 63;
      float t = 0.f;
      for (int th = (ch >> 3); th < (int)blockDim.x; th += 8) t += red[th * 16 + which * 8 + (ch & 7)];
      iic_stat_add(sums, blockIdx.x % IIC_STAT_STRIPES, STEM_CO, ch, which, t);
    }
    __syncthreads();
  }
  if (MODE >= 1) {
    // reduce the waves' accumulators through LDS, then one partial per block:
    //   MODE 1: [64][LD];  MODE 2: [128][LD] = G1 rows 0..63, G2 rows 64..127
    float* red = sY;   // [nwaves][64][NKT*32]
    constexpr int LD = NKT * 32;
    constexpr int ROWS = MODE == 2 ? 128 : 64;
    float* pout = outbuf + (long)blockIdx.x * ROWS * LD;
#pragma unroll
    for (int pass = 0; pass < (MODE == 2 ? 2 : 1); ++pass) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int t = 0; t < NKT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            red[((long)wave * 64 + h * 32 + mfma32_row(r, lane)) * LD + t * 32 + (lane & 31)] =
                pass == 0 ? dacc[h][t][r] : daccy[MODE == 2 ? h : 0][t][r];
      __syncthreads();
      for (int idx = threadIdx.x; idx < 64 * LD; idx += blockDim.x) {
        float t = 0.f;
        for (int wv = 0; wv < nwaves; ++wv) t += red[(long)wv * 64 * LD + idx];
        pout[(long)pass * 64 * LD + idx] = t;
      }
      __syncthreads();
    }
  }
}

// G3[k = (c, kh, kw)] = sum over images and valid conv positions (y, x) of the zero-padded input
// x[c][y+kh-1][x+kw-1]: every input pixel (yy, xx) counts for tap (kh, kw) iff the conv position
// (yy-kh+1, xx-kw+1) lies inside the image.  out[block][64]: per-block partial sums.
__global__ __launch_bounds__(256) void stem_patch_sums_kernel(const float* __restrict__ x,
                                                              float* __restrict__ out, int N, int CIN,
                                                              int H, int W) {
  // Inclusion-exclusion instead of 9 predicated sums per pixel: tap (kh, kw) covers every input
  // pixel except the last / no / the first row (kh = 0 / 1 / 2) and column (kw likewise), so
  //   G3 = T - R[kh] - C[kw] + X[kh][kw]
  // with T = total, R = excluded-row sum, C = excluded-column sum, X = excluded corner.
  // Block = one (image, channel) plane slice: rows strided over blocks, no divisions per pixel.
  __shared__ float red[9][256];
  const int c = blockIdx.y;
  float T = 0.f, r0 = 0.f, r1 = 0.f, c0 = 0.f, c1 = 0.f, x00 = 0.f, x01 = 0.f, x10 = 0.f, x11 = 0.f;
  // 256 threads = 8 rows in flight x 32 column lanes (latency-bound otherwise: one dependent
  // load per row per block)
  const long rows = (long)N * H;
  const int rl = threadIdx.x >> 5, cl = threadIdx.x & 31;
  for (long row = (long)blockIdx.x * 8 + rl; row < rows; row += (long)gridDim.x * 8) {
    const long n = row / H;
    const int yy = (int)(row - n * H);
    const float* p = x + ((n * CIN + c) * H + yy) * (long)W;
    float t = 0.f;
    for (int xx = cl; xx < W; xx += 32) t += p[xx];
    T += t;
    if (yy == 0) r0 += t;
    if (yy == H - 1) r1 += t;
    if (cl == 0) {
      const float a = p[0], b = p[W - 1];
      c0 += a;
      c1 += b;
      if (yy == 0) { x00 += a; x01 += b; }
      if (yy == H - 1) { x10 += a; x11 += b; }
    }
  }
  const float v[9] = {T, r0, r1, c0, c1, x00, x01, x10, x11};
#pragma unroll
  for (int t = 0; t < 9; ++t) red[t][threadIdx.x] = v[t];
  __syncthreads();
  for (int st = blockDim.x >> 1; st > 0; st >>= 1) {
    if (threadIdx.x < st) {
#pragma unroll
      for (int t = 0; t < 9; ++t) red[t][threadIdx.x] += red[t][threadIdx.x + st];
    }
    __syncthreads();
  }
  if (threadIdx.x < 9) {
    const int kh = threadIdx.x / 3, kw = threadIdx.x % 3;
    // excluded row: kh = 0 -> last row (H-1), kh = 2 -> first row (0); same for columns
    const float R = kh == 0 ? red[2][0] : (kh == 2 ? red[1][0] : 0.f);
    const float C = kw == 0 ? red[4][0] : (kw == 2 ? red[3][0] : 0.f);
    float X = 0.f;
    if (kh == 0 && kw == 0) X = red[8][0];   // row H-1, col W-1
    if (kh == 0 && kw == 2) X = red[7][0];   // row H-1, col 0
    if (kh == 2 && kw == 0) X = red[6][0];   // row 0,   col W-1
    if (kh == 2 && kw == 2) X = red[5][0];   // row 0,   col 0
    // one slot per block (no float atomics: the sum over blocks happens in a fixed order in
    // stem_wgrad_combine_kernel)
    out[(long)blockIdx.x * 64 + c * 9 + threadIdx.x] = red[0][0] - R - C + X;
  }
}

// MODE 2 epilogue: dW[co][k] = c1[co]*sum_b G1 + c2[co]*sum_b G2 + c3[co]*sum_b G3[k]
// (bcoef = [3][64] from bn_bwd_finalize: dy = c1*g + c2*y + c3).
__global__ __launch_bounds__(256) void stem_wgrad_combine_kernel(const float* __restrict__ part,
                                                                 int nblocks, int LD, int K,
                                                                 const float* __restrict__ bcoef,
                                                                 const float* __restrict__ g3,
                                                                 float* __restrict__ dW) {
  __shared__ float red[3][256];
  const int idx = blockIdx.x;            // output element co*K + k
  const int co = idx / K, k = idx - co * K;
  float t1 = 0.f, t2 = 0.f, t3 = 0.f;
  for (int b = threadIdx.x; b < nblocks; b += 256) {
    const float* pb = part + (long)b * 128 * LD;
    t1 += pb[(long)co * LD + k];
    t2 += pb[(long)(64 + co) * LD + k];
  }
  for (int b = threadIdx.x; b < 512; b += 256) t3 += g3[(long)b * 64 + k];   // G3 per-block partials
  red[0][threadIdx.x] = t1;
  red[1][threadIdx.x] = t2;
  red[2][threadIdx.x] = t3;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      red[0][threadIdx.x] += red[0][threadIdx.x + s];
      red[1][threadIdx.x] += red[1][threadIdx.x + s];
      red[2][threadIdx.x] += red[2][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0)
    dW[idx] = bcoef[co] * red[0][0] + bcoef[STEM_CO + co] * red[1][0] + bcoef[2 * STEM_CO + co] * red[2][0];
}

// dW[co][k] = sum_b part[b][co][k]  (k < K), fp32 OIHW flatten.  One block per output
// element group: 256 threads stride the partial blocks, LDS tree reduction.
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ part,
                                                                int nblocks, int LD, int K,
                                                                float* __restrict__ dW) {
  __shared__ float red[256];
  const int idx = blockIdx.x;            // output element co*K + k
  const int co = idx / K, k = idx - co * K;
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
// Sobel pre-op (transforms.py:47-96): grey -> dx, dy (zero padded 3x3 correlations), other
// channels copied in the reference's concat order.  One thread per output pixel.
// ------------------------------------------------------------------------------------
__global__ void sobel_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int C,
                             int Cout, int H, int W, int grey_c, int sob_c, int ncopy,
                             const int4 copy_src, const int4 copy_dst) {
  const long total = (long)N * H * W;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const int xq = (int)(idx % W);
    const long r = idx / W;
    const int y = (int)(r % H), n = (int)(r / H);
    const float* gp = in + ((long)n * C + grey_c) * H * W;
    float v[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const int yy = y + a - 1, xx = xq + b - 1;
        v[a][b] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? gp[(long)yy * W + xx] : 0.f;
      }
    // same accumulation order as a direct 3x3 correlation (row-major taps)
    float dx = 0.f, dy = 0.f;
    dx += v[0][0] * 1.f; dx += v[0][2] * -1.f; dx += v[1][0] * 2.f; dx += v[1][2] * -2.f;
    dx += v[2][0] * 1.f; dx += v[2][2] * -1.f;
    dy += v[0][0] * 1.f; dy += v[0][1] * 2.f; dy += v[0][2] * 1.f; dy += v[2][0] * -1.f;
    dy += v[2][1] * -2.f; dy += v[2][2] * -1.f;
    float* op = out + (long)n * Cout * H * W + (long)y * W + xq;
    op[(long)sob_c * H * W] = dx;
    op[(long)(sob_c + 1) * H * W] = dy;
    const int cs[4] = {copy_src.x, copy_src.y, copy_src.z, copy_src.w};
    const int cd[4] = {copy_dst.x, copy_dst.y, copy_dst.z, copy_dst.w};
    for (int i = 0; i < ncopy; ++i)
      op[(long)cd[i] * H * W] = in[((long)n * C + cs[i]) * H * W + (long)y * W + xq];
  }
}

#define STEM_DISPATCH(CIN_, CALL)            \
  switch (CIN_) {                            \
    case 1: { constexpr int CI = 1; CALL; } break; \
    case 2: { constexpr int CI = 2; CALL; } break; \
    case 3: { constexpr int CI = 3; CALL; } break; \
    case 4: { constexpr int CI = 4; CALL; } break; \
    case 5: { constexpr int CI = 5; CALL; } break; \
    default: return IIC_ERR_UNSUPPORTED;     \
  }

static int stem_check(const void* x, const void* w, int N, int Cin, int H, int W) {
  if (!x || !w || N <= 0) return IIC_ERR_ARG;
  if (Cin < 1 || Cin > 5 || H < 2 || W < 2 || (H & 1) || (W & 1) || W > 256) return IIC_ERR_UNSUPPORTED;
  return IIC_OK;
}

// stem_bwd2.hip
int iic_stem_bwd2_supported(int Cin, int W);
int iic_stem_bwd2_launch(const float* x, const float* w, const float* coef, const void* dpool_pt,
                         float* sums, float* partials, int* nblocks_out, int N, int Cin, int H, int W,
                         void* stream);
IIC_SWITCH(g_stem_bwd2, 1, iic_debug_enable_stem_bwd2)

extern "C" {

int iic_stem_stats(const float* x, const float* w, float* stats, int N, int Cin, int H, int W,
                   void* stream) {
  int rc = stem_check(x, w, N, Cin, H, W);
  if (rc) return rc;
  if (!stats) return IIC_ERR_ARG;
  const long tiles = (long)N * H * ((W + 31) / 32);
  int grid = (int)((tiles + 3) / 4);
  if (grid > STEM_PERSIST_BLOCKS) grid = STEM_PERSIST_BLOCKS;
  STEM_DISPATCH(Cin, hipLaunchKernelGGL(stem_stats_kernel<CI>, dim3(grid), dim3(256), 0,
                                        (hipStream_t)stream, x, w, stats, N, H, W));
  return iic_launch_status();
}

int iic_stem_apply_pool(const float* x, const float* w, const float* coef, void* out_pt, int N,
                        int Cin, int H, int W, void* stream) {
  int rc = stem_check(x, w, N, Cin, H, W);
  if (rc) return rc;
  if (!coef || !out_pt) return IIC_ERR_ARG;
  const int nseg = (W + 31) / 32, Ho = H / 2 + 1;
  const size_t lds = (size_t)2 * W * STEM_CO * sizeof(bf16_t);
  STEM_DISPATCH(Cin, hipLaunchKernelGGL(stem_apply_pool_kernel<CI>, dim3(N * Ho), dim3(64 * nseg),
                                        lds, (hipStream_t)stream, x, w, coef, (bf16_t*)out_pt, N, H,
                                        W));
  return iic_launch_status();
}

static size_t stem_bwd_lds(int Cin, int W, int nseg, int mode) {
  size_t a = (size_t)2 * W * STEM_CO * sizeof(float);
  if (mode >= 1) {
    const int WP = (W + 15) & ~15;
    const int PD = 2 * WP + 8;
    a += (((size_t)STEM_CO * PD * 2 + 15) & ~(size_t)15) +
         (((size_t)Cin * 4 * (WP + 10) * sizeof(float) + 15) & ~(size_t)15);
  }
  size_t b0 = (size_t)64 * nseg * 16 * sizeof(float);
  size_t b1 = (size_t)nseg * 64 * ((Cin * 9 + 31) / 32) * 32 * sizeof(float);
  size_t b = mode == 0 ? b0 : (b1 > b0 ? b1 : b0);
  return a > b ? a : b;
}

int iic_stem_bwd_reduce(const float* x, const float* w, const float* coef, const void* dpool_pt,
                        float* sums, int N, int Cin, int H, int W, void* stream) {
  int rc = stem_check(x, w, N, Cin, H, W);
  if (rc) return rc;
  if (!coef || !dpool_pt || !sums) return IIC_ERR_ARG;
  const int nseg = (W + 31) / 32, Ho = H / 2 + 1;
  long items = (long)N * Ho;
  int grid = (int)(items < STEM_PERSIST_BLOCKS ? items : STEM_PERSIST_BLOCKS);
  const size_t lds = stem_bwd_lds(Cin, W, nseg, 0);
  STEM_DISPATCH(Cin, {
    if (lds > 48 * 1024) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_bwd_kernel<CI, 0>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        return IIC_ERR_UNSUPPORTED;
      }
    }
    hipLaunchKernelGGL((stem_bwd_kernel<CI, 0>), dim3(grid), dim3(64 * nseg), lds,
                       (hipStream_t)stream, x, w, coef, (const float*)nullptr,
                       (const bf16_t*)dpool_pt, sums, (float*)nullptr, N, H, W);
  });
  return iic_launch_status();
}

#define STEM_G3_BLOCKS 512
long iic_stem_wgrad_partial_floats(void) { return (long)STEM_PERSIST_BLOCKS * 128 * 64 + STEM_G3_BLOCKS * 64; }

/* One recompute pass: sums (as iic_stem_bwd_reduce) AND the coefficient-free weight-gradient
 * GEMMs G1 = sum g*patch, G2 = sum y*patch into partials [blocks][128][LD] (+ G3 = sum patch as
 * [512][64] per-block partials at the end of the partials buffer); *nblocks_out = blocks written. */
int iic_stem_bwd_fused(const float* x, const float* w, const float* coef, const void* dpool_pt,
                       float* sums, float* partials, int* nblocks_out, int N, int Cin, int H, int W,
                       void* stream) {
  int rc = stem_check(x, w, N, Cin, H, W);
  if (rc) return rc;
  if (!coef || !dpool_pt || !sums || !partials || !nblocks_out) return IIC_ERR_ARG;
  const int nseg = (W + 31) / 32, Ho = H / 2 + 1;
  long items = (long)N * Ho;
  int grid = (int)(items < STEM_PERSIST_BLOCKS ? items : STEM_PERSIST_BLOCKS);
  *nblocks_out = grid;
  float* g3 = partials + (long)STEM_PERSIST_BLOCKS * 128 * 64;
  hipLaunchKernelGGL(stem_patch_sums_kernel, dim3(STEM_G3_BLOCKS, Cin), dim3(256), 0, (hipStream_t)stream,
                     x, g3, N, Cin, H, W);
  if (g_stem_bwd2 && iic_stem_bwd2_supported(Cin, W))      // register-resident routing (stem_bwd2.hip)
    return iic_stem_bwd2_launch(x, w, coef, dpool_pt, sums, partials, nblocks_out, N, Cin, H, W, stream);
  const size_t lds = stem_bwd_lds(Cin, W, nseg, 2);
  STEM_DISPATCH(Cin, {
    if (lds > 48 * 1024) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_bwd_kernel<CI, 2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        return IIC_ERR_UNSUPPORTED;
      }
    }
    hipLaunchKernelGGL((stem_bwd_kernel<CI, 2>), dim3(grid), dim3(64 * nseg), lds,
                       (hipStream_t)stream, x, w, coef, (const float*)nullptr,
                       (const bf16_t*)dpool_pt, partials, sums, N, H, W);
  });
  return iic_launch_status();
}

/* dW (fp32 OIHW [64][Cin][3][3]) from the partials of iic_stem_bwd_fused and the BatchNorm
 * backward coefficients bcoef = [3][64] (iic_bn_bwd_finalize). */
int iic_stem_wgrad_combine(const float* partials, int nblocks, const float* bcoef, float* dW, int Cin,
                           void* stream) {
  if (!partials || !bcoef || !dW || nblocks <= 0 || Cin < 1 || Cin > 5) return IIC_ERR_ARG;
  const int K = Cin * 9, LD = ((K + 31) / 32) * 32;
  hipLaunchKernelGGL(stem_wgrad_combine_kernel, dim3(64 * K), dim3(256), 0, (hipStream_t)stream,
                     partials, nblocks, LD, K, bcoef, partials + (long)STEM_PERSIST_BLOCKS * 128 * 64, dW);
  return iic_launch_status();
}

int iic_stem_bwd_wgrad(const float* x, const float* w, const float* coef, const float* bcoef,
                       const void* dpool_pt, float* partials, float* dW, int N, int Cin, int H,
                       int W, void* stream) {
  int rc = stem_check(x, w, N, Cin, H, W);
  if (rc) return rc;
  if (!coef || !bcoef || !dpool_pt || !partials || !dW) return IIC_ERR_ARG;
  const int nseg = (W + 31) / 32, Ho = H / 2 + 1;
  long items = (long)N * Ho;
  int grid = (int)(items < STEM_PERSIST_BLOCKS ? items : STEM_PERSIST_BLOCKS);
  const size_t lds = stem_bwd_lds(Cin, W, nseg, 1);
  STEM_DISPATCH(Cin, {
    if (lds > 48 * 1024) {   // static s_cf (1.25 KB) + dynamic must stay <= 160 KB
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_bwd_kernel<CI, 1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        return IIC_ERR_UNSUPPORTED;
      }
    }
    hipLaunchKernelGGL((stem_bwd_kernel<CI, 1>), dim3(grid), dim3(64 * nseg), lds,
                       (hipStream_t)stream, x, w, coef, bcoef, (const bf16_t*)dpool_pt, partials,
                       (float*)nullptr, N, H, W);
  });
  const int K = Cin * 9, LD = ((K + 31) / 32) * 32;
  hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(64 * K), dim3(256), 0,
                     (hipStream_t)stream, partials, grid, LD, K, dW);
  return iic_launch_status();
}

int iic_sobel(const float* imgs, float* out, int N, int C, int H, int W, int include_rgb,
              int using_IR, void* stream) {
  if (!imgs || !out || N <= 0 || H <= 0 || W <= 0) return IIC_ERR_ARG;
  // channel bookkeeping of transforms.py:50-67,83-94
  int grey_c, sob_c, ncopy = 0, Cout;
  int4 cs = make_int4(0, 0, 0, 0), cd = make_int4(0, 0, 0, 0);
  if (!using_IR) {
    if (!include_rgb) { if (C != 1) return IIC_ERR_ARG; grey_c = 0; sob_c = 0; Cout = 2; }
    else { if (C != 4) return IIC_ERR_ARG; grey_c = 3; sob_c = 3; Cout = 5; ncopy = 3;
           cs = make_int4(0, 1, 2, 0); cd = make_int4(0, 1, 2, 0); }
  } else {
    if (!include_rgb) { if (C != 2) return IIC_ERR_ARG; grey_c = 0; sob_c = 0; Cout = 3; ncopy = 1;
                        cs = make_int4(1, 0, 0, 0); cd = make_int4(2, 0, 0, 0); }
    else { if (C != 5) return IIC_ERR_ARG; grey_c = 3; sob_c = 3; Cout = 6; ncopy = 4;
           cs = make_int4(0, 1, 2, 4); cd = make_int4(0, 1, 2, 5); }
  }
  const long total = (long)N * H * W;
  int grid = (int)((total + 255) / 256);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(sobel_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, imgs, out, N, C,
                     Cout, H, W, grey_c, sob_c, ncopy, cs, cd);
  return iic_launch_status();
}

}  // extern "C"

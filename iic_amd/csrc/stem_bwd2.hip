// ClusterNet5g stem backward, register-resident (9*Cin <= 32): conv1 + bn1 + relu + maxpool(2,2,p1)
// of /root/reference/code/archs/cluster/net5g.py:21-26,42-45, gradient w.r.t. conv1.weight and
// the bn1 sums, in ONE recompute pass (same algebra and outputs as stem_bwd_kernel<CIN, 2> in
// stem.hip: sums += (sum g, sum g*y); partials = G1 = sum g*patch, G2 = sum y*patch).
//
// stem_bwd_kernel routes the pooled gradient through LDS (fp32 conv rows out, a VALU loop per
// (window, channel) with 4 LDS reads, transposed bf16 tiles back in) and is bound by that loop.
// Here one wave owns a 32-pixel segment of the two conv rows of a pooled row, with the segment
// origin at an ODD column (x0 = 32*seg - 1).  In the MFMA C layout a lane then holds, for its
// two channels, pixels {4g+8j .. 4g+8j+3}: both columns of every pooling window (2wo-1, 2wo)
// and both rows are in the SAME lane, so BatchNorm + ReLU + arg-max routing happen in
// registers.  The routed gradient and the conv outputs, packed to bf16 in accumulator order, ARE
// the A operands of the dW GEMM (contraction over pixels; the B operand, patch[pixel][k], is
// built from the LDS-staged input rows with the same pixel permutation).  LDS holds only the four
// input rows of the item (double-buffered, next item prefetched).
#include "stem_common.h"

template <int CIN>
__global__ __launch_bounds__(512) void stem_bwd2_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ coef,
                                                        const bf16_t* __restrict__ dpool,
                                                        float* __restrict__ partials,
                                                        float* __restrict__ sums, int N, int H, int W,
                                                        int abl) {
  // abl: timing ablation (wrong results): 1 no conv MFMAs, 2 no routing, 4 no dW MFMAs,
  // 8 no pooled-gradient loads, 16 no input staging
  constexpr int K = StemK<CIN>::K;
  constexpr int KS = StemK<CIN>::KS;
  static_assert(K <= 32, "one 32-wide column tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int l31 = lane & 31, g5 = lane >> 5;
  const int Ho = H / 2 + 1, Wo = W / 2 + 1;
  const int WX = nwaves * 32 + 4;                  // staged row: image columns -2 .. nwaves*32+1
  const int XR = CIN * 4 * WX;                     // floats per input-row buffer
  float* sX = reinterpret_cast<float*>(smem_raw);  // [2][CIN][4][WX]
  float* sRed = sX + 2 * XR;                       // [128][33] end-of-kernel reduction
  float* sW = sRed + 128 * 33;                     // [2][KS][64] conv B fragments (per lane), kept in
                                                   // LDS: 2*KS registers less across the routing code
  const int GR = (Wo * STEM_CO + 7) & ~7;          // bf16 elements per pooled-gradient row buffer
  bf16_t* sG = reinterpret_cast<bf16_t*>(sW + 2 * KS * 64);   // [2][Wo][64] pooled gradient rows
  if (wave == 0) {
    float wr[2][KS];
    stem_load_w<CIN>(w, lane, wr);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int s = 0; s < KS; ++s) sW[(h * KS + s) * 64 + lane] = wr[h][s];
  }
  // BatchNorm coefficients of this lane's two channels (l31, l31 + 32)
  const float sc0 = coef[l31], sc1 = coef[l31 + 32];
  const float sh0 = coef[STEM_CO + l31], sh1 = coef[STEM_CO + l31 + 32];
  // conv A operand: k = 2s + (lane >> 5) -> (c, kh, kw): offset (c*4 + kh)*WX + kw, or -1
  int koff[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int k = 2 * s + g5;
    koff[s] = k < K ? ((k / 9) * 4 + (k % 9) / 3) * WX + (k % 3) : -1;
  }
  // dW B operand: column kcol = l31 -> same decomposition
  const int boff = l31 < K ? ((l31 / 9) * 4 + (l31 % 9) / 3) * WX + (l31 % 3) : -1;

  // column validity of this lane's 16 window columns (item-independent): bit 2m / 2m+1 = first /
  // second column of window slot m = 2j + e, image column wave*32 - 1 + 8j + 4g5 + 2e (+1)
  uint32_t cvb = 0;
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int xc = wave * 32 - 1 + 8 * (m >> 1) + 4 * g5 + 2 * (m & 1);
    if (xc >= 0 && xc < W) cvb |= 1u << (2 * m);
    if (xc + 1 < W) cvb |= 1u << (2 * m + 1);
  }

  f32x16 dg[2], dy[2];     // G1 / G2 accumulators [channel half]: rows = co, cols = k
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 16; ++r) dg[h][r] = dy[h][r] = 0.f;
  float sg[2] = {0.f, 0.f}, sgy[2] = {0.f, 0.f};

  const long items = (long)N * Ho;
  const long it0 = (items * blockIdx.x) / gridDim.x, it1 = (items * (blockIdx.x + 1)) / gridDim.x;

  // The 4 input rows 2ho-2 .. 2ho+1 of an item (zero outside the image) are fetched into
  // registers at the START of the previous item and written to the idle LDS buffer at its END:
  // the global-load latency hides behind that item's MFMA / routing work.
  constexpr int SPT = 2 * CIN + 1;                 // staged floats per thread: XR <= SPT * blockDim always
  // item-independent part of the staging addresses: element idx = (c, r, xs) of the buffer reads
  // input row 2ho-2+r, column xs-2: offset c*H*W + (r-2)*W + xs-2 (+ 2ho*W per item); srow = r,
  // or -1 when the element is padding (column outside the image / idx >= XR)
  int soff[SPT], srow[SPT];
#pragma unroll
  for (int u = 0; u < SPT; ++u) {
    const int idx = u * (int)blockDim.x + (int)threadIdx.x;
    const int c = idx / (4 * WX), rem = idx - c * 4 * WX;
    const int r = rem / WX, xg = rem - r * WX - 2;
    soff[u] = (c * H + r - 2) * W + xg;
    srow[u] = (idx < XR && xg >= 0 && xg < W) ? r : -1;
  }
  // pooled-gradient row of the item: Wo x 64 bf16 = Wo*8 16-byte pieces, staged like the input rows
  // (2-byte gathers straight from global cost ~25 % of the kernel: latency per item, not bytes)
  const int GP = Wo * 8;                           // pieces per item, <= 2 * blockDim
  float stg[SPT];
  u32x4 gst[2];
  auto stage_load = [&](long it) {
    const int n = (int)(it / Ho), ho = (int)(it - (long)n * Ho);
    const float* xin = x + (long)n * CIN * H * W + (long)2 * ho * W;
#pragma unroll
    for (int u = 0; u < SPT; ++u) {
      const int yy = 2 * ho - 2 + srow[u];
      stg[u] = (srow[u] >= 0 && yy >= 0 && yy < H) ? xin[soff[u]] : 0.f;
    }
    const bf16_t* gp = dpool + (((long)n * (Ho + 2) + ho + 1) * (Wo + 2) + 1) * STEM_CO;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int piece = u * (int)blockDim.x + (int)threadIdx.x;
      gst[u] = (u32x4){0u, 0u, 0u, 0u};
      if (piece < GP && !(abl & 8)) gst[u] = *reinterpret_cast<const u32x4*>(gp + (long)piece * 8);
    }
  };
  auto stage_store = [&](int b) {
    float* d = sX + b * XR;
#pragma unroll
    for (int u = 0; u < SPT; ++u) {
      const int idx = u * (int)blockDim.x + (int)threadIdx.x;
      if (idx < XR) d[idx] = stg[u];
    }
    bf16_t* dgp = sG + b * GR;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int piece = u * (int)blockDim.x + (int)threadIdx.x;
      if (piece < GP) *reinterpret_cast<u32x4*>(dgp + piece * 8) = gst[u];
    }
  };

  if (it0 < it1) {
    stage_load(it0);
    stage_store(0);
  }
  __syncthreads();
  for (long it = it0; it < it1; ++it) {
    const int b = (int)((it - it0) & 1);
    const int n = (int)(it / Ho), ho = (int)(it - (long)n * Ho);
    (void)n;
    const float* sx = sX + b * XR;
    const bf16_t* sg_row = sG + b * GR;
    if (it + 1 < it1 && !(abl & 16)) stage_load(it + 1);       // in flight during this item's work
    const bool rv0 = (2 * ho - 1) >= 0, rv1 = (2 * ho) < H;
    const int xl = wave * 32 + l31;               // staged-column index of this lane's conv pixel (kw = 0)

    // ---- conv rows 2ho-1, 2ho for 32 pixels x 64 channels (exact fp32 MFMA) ----------------
    f32x16 acc[2][2];    // [row][channel half]
#pragma unroll
    for (int rs = 0; rs < 2; ++rs) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rs][h][r] = 0.f;
      if (!(abl & 1))
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const float a = koff[s] >= 0 ? sx[koff[s] + rs * WX + xl] : 0.f;
        acc[rs][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, sW[s * 64 + lane], acc[rs][0], 0, 0, 0);
        acc[rs][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, sW[(KS + s) * 64 + lane], acc[rs][1], 0, 0, 0);
      }
    }
    // ---- BN + ReLU + arg-max routing in registers, then the dW GEMMs, in two halves u = 0, 1:
    //      windows j = 2u, 2u+1 fill accumulator registers 8u .. 8u+7 = the 8 k-slots of MFMA u
    //      (k-slot s of lane-half g5 is local pixel 4*g5 + 8*(2u + (s >> 2)) + (s & 3)).
    // Branch-free routing: invalid positions (outside the image) take activation 0, which can
    // never be a positive maximum; "first maximum in scan order wins" (torch max_pool2d) = strict
    // '>' updates; the gradient is routed only when the maximum is positive (ReLU).
    uint32_t cv = cvb;
    asm volatile("" : "+v"(cv));      // keep the validity tests inside the loop (see cvb)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      uint32_t gpk[2][2][4], ypk[2][2][4];   // [row][half][register pair of this u]
      if (abl & 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            gpk[0][h][i] = gpk[1][h][i] = (uint32_t)sg_row[(i + 4 * u) * STEM_CO + l31 + 32 * h];
            ypk[0][h][i] = ypk[1][h][i] = __float_as_uint(acc[0][h][i] + acc[1][h][i + 8 * u]);
          }
      } else
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int j = 2 * u + jj;
          const int r0 = 4 * j + 2 * e;                   // registers r0, r0+1 = the window's columns
          const int m = j * 2 + e;                        // window slot of this lane
          const bool c0 = (cv >> (2 * m)) & 1, c1 = (cv >> (2 * m + 1)) & 1;
          const bool v00 = rv0 && c0, v01 = rv0 && c1, v10 = rv1 && c0, v11 = rv1 && c1;
          // pooled gradient of window wo = wave*16 + 4j + 2g5 + e, channels l31 / l31 + 32 (LDS row)
          const int wo = wave * 16 + 4 * j + 2 * g5 + e;
          const bf16_t* gq = sg_row + (wo < Wo ? wo : 0) * STEM_CO + l31;
          const float gvm[2] = {wo < Wo ? bf16_to_f32(gq[0]) : 0.f, wo < Wo ? bf16_to_f32(gq[32]) : 0.f};
#pragma unroll
          for (int h = 0; h < 2; ++h) {                   // both channels of the lane: same masks
            const float sc = h ? sc1 : sc0, sh = h ? sh1 : sh0;
            const float y00 = acc[0][h][r0], y01 = acc[0][h][r0 + 1];
            const float y10 = acc[1][h][r0], y11 = acc[1][h][r0 + 1];
            const float a00 = v00 ? fmaxf(y00 * sc + sh, 0.f) : 0.f;
            const float a01 = v01 ? fmaxf(y01 * sc + sh, 0.f) : 0.f;
            const float a10 = v10 ? fmaxf(y10 * sc + sh, 0.f) : 0.f;
            const float a11 = v11 ? fmaxf(y11 * sc + sh, 0.f) : 0.f;
            float best = a00, yb = y00;
            int am = 0;
            if (a01 > best) { best = a01; yb = y01; am = 1; }
            if (a10 > best) { best = a10; yb = y10; am = 2; }
            if (a11 > best) { best = a11; yb = y11; am = 3; }
            const float gs = best > 0.f ? gvm[h] : 0.f;
            sg[h] += gs;
            sgy[h] += gs * yb;
            const int pr = jj * 2 + e;
            gpk[0][h][pr] = pack_bf16x2(am == 0 ? gs : 0.f, am == 1 ? gs : 0.f);
            gpk[1][h][pr] = pack_bf16x2(am == 2 ? gs : 0.f, am == 3 ? gs : 0.f);
            ypk[0][h][pr] = pack_bf16x2(c0 ? y00 : 0.f, c1 ? y01 : 0.f);
            ypk[1][h][pr] = pack_bf16x2(c0 ? y10 : 0.f, c1 ? y11 : 0.f);
          }
        }
      // dW GEMMs of this half: A = g / y (channels x pixels), B = patch (pixels x k)
#pragma unroll
      for (int rs = 0; rs < 2; ++rs) {
        if (!(rs ? rv1 : rv0)) continue;          // uniform: the whole row is outside the image
        if (abl & 4) { sg[0] += __uint_as_float(gpk[rs][0][0] ^ ypk[rs][1][1] ^ gpk[rs][1][2] ^ ypk[rs][0][3]); continue; }
        union { bf16x8 v; uint32_t q[4]; } bb;
#pragma unroll
        for (int i = 0; i < 4; ++i) bb.q[i] = 0u;
        if (boff >= 0) {
          // image column of local pixel p is wave*32 - 1 + p; tap kw reads column + kw - 1, staged at
          // index column + kw + 1 = wave*32 + p + kw (kw is inside boff)
          const float* xr = sx + boff + rs * WX + wave * 32 + 4 * g5 + 16 * u;
          bb.q[0] = pack_bf16x2(xr[0], xr[1]);
          bb.q[1] = pack_bf16x2(xr[2], xr[3]);
          bb.q[2] = pack_bf16x2(xr[8], xr[9]);
          bb.q[3] = pack_bf16x2(xr[10], xr[11]);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          union { bf16x8 v; uint32_t q[4]; } ag, ay;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            ag.q[i] = gpk[rs][h][i];
            ay.q[i] = ypk[rs][h][i];
          }
          dg[h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ag.v, bb.v, dg[h], 0, 0, 0);
          dy[h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ay.v, bb.v, dy[h], 0, 0, 0);
        }
      }
    }
    if (it + 1 < it1 && !(abl & 16)) stage_store(b ^ 1);       // buffer b^1 was last read one item ago
    __syncthreads();     // buffer b fully consumed; buffer b^1 fully staged
  }

  // ---- block reduction: partial [128][32] = G1 rows 0..63, G2 rows 64..127; sums via atomics ----
  for (int i = threadIdx.x; i < 128 * 33; i += blockDim.x) sRed[i] = 0.f;
  __syncthreads();
  // waves take turns in index order: a fixed summation order (LDS float atomics would add the
  // waves' accumulators in arrival order and make the partial -- hence dW -- irreproducible)
  for (int wv = 0; wv < nwaves; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = h * 32 + mfma32_row(r, lane);
          sRed[row * 33 + l31] += dg[h][r];
          sRed[(64 + row) * 33 + l31] += dy[h][r];
        }
    }
    __syncthreads();
  }
  float* pout = partials + (long)blockIdx.x * 128 * 32;
  for (int i = threadIdx.x; i < 128 * 32; i += blockDim.x) pout[i] = sRed[(i >> 5) * 33 + (i & 31)];
  __syncthreads();
  // sums: lanes (l31, g5 = 0/1) hold the same channels -> fold halves, then waves through LDS
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    sg[h] += __shfl_xor(sg[h], 32, 64);
    sgy[h] += __shfl_xor(sgy[h], 32, 64);
  }
  float* sS = sRed;      // [nwaves][2][64]
  if (lane < 32) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      sS[(wave * 2 + 0) * 64 + h * 32 + lane] = sg[h];
      sS[(wave * 2 + 1) * 64 + h * 32 + lane] = sgy[h];
    }
  }
  __syncthreads();
  for (int o = threadIdx.x; o < 128; o += blockDim.x) {
    const int which = o >> 6, ch = o & 63;
    float t = 0.f;
    for (int wv = 0; wv < nwaves; ++wv) t += sS[(wv * 2 + which) * 64 + ch];
    iic_stat_add(sums, blockIdx.x % IIC_STAT_STRIPES, STEM_CO, ch, which, t);
  }
}

// used by stem.hip's iic_stem_bwd_fused
int iic_stem_bwd2_supported(int Cin, int W) { return Cin * 9 <= 32 && W + 1 <= 8 * 32; }

int iic_stem_bwd2_launch(const float* x, const float* w, const float* coef, const void* dpool_pt,
                         float* sums, float* partials, int* nblocks_out, int N, int Cin, int H, int W,
                         void* stream) {
  const int nseg = (W + 1 + 31) / 32, Ho = H / 2 + 1;
  const long items = (long)N * Ho;
  int grid = (int)(items < STEM_PERSIST_BLOCKS ? items : STEM_PERSIST_BLOCKS);
  *nblocks_out = grid;
  const size_t xr = (size_t)Cin * 4 * (nseg * 32 + 4) * sizeof(float);
  const size_t lds = 2 * xr + (size_t)128 * 33 * sizeof(float) + (size_t)2 * ((Cin * 9 + 1) / 2) * 64 * sizeof(float) +
                     (size_t)2 * (((W / 2 + 1) * STEM_CO + 7) & ~7) * sizeof(bf16_t);
#define BWD2_LAUNCH(CI_)                                                                          \
  hipLaunchKernelGGL(stem_bwd2_kernel<CI_>, dim3(grid), dim3(64 * nseg), lds, (hipStream_t)stream, x, \
                     w, coef, (const bf16_t*)dpool_pt, partials, sums, N, H, W, iic_debug_get_ablate())
  switch (Cin) {
    case 1: BWD2_LAUNCH(1); break;
    case 2: BWD2_LAUNCH(2); break;
    case 3: BWD2_LAUNCH(3); break;
    default: return IIC_ERR_UNSUPPORTED;
  }
  return iic_launch_status();
}

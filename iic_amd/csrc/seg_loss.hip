// IID segmentation loss for gfx950: the T-shifted per-pixel joint and its gradient.
//
// Replaces, in /root/reference/code/utils/segmentation/IID_losses.py:14-159,
//   perform_affine_tf (identity / axis flips -- the only transforms the published runs use,
//   potsdam.py:189-202), the mask multiplies (:45-47,115-117), the permutes (:51-52,121-122)
//   and the giant-filter F.conv2d(x1, weight=x2_inv, padding=T) (:55,125), whose backward is
//   two more such convolutions.  The k x k x (2T+1)^2 statistics stage reuses iid_loss_kernel.
//
//   R[p][q][i][j] = sum_{n,y,x} x1m[n][i][y+p-T][x+q-T] * x2m[n][j][y][x]
//   x1m = x1 * mask,  x2m = flip(x2) * mask   (zero outside the image)
//
// All arithmetic is exact fp32 on v_mfma_f32_16x16x4_f32 (k <= 48 classes => 16x16 tiles keep
// the matrix core well filled where a 32x32 tile would idle 3/4 of it at k = 15).
//
// seg_joint_kernel : workgroup = (one row shift p, a group of QG column shifts q, a slice of
//   the (n, y) rows); one image row of x2m and the matching shifted row of x1m (+T halo) are
//   staged in LDS as [class][pixel]; the x2m fragment is reused by all QG shifts; accumulators
//   (QG x TK x TK tiles) stay in registers across the whole slice -> one partial per workgroup.
// seg_grad_kernel  : d/dx1m[i][v] = sum_{p,q,j} G[p][q][i][j] x2m[j][v-(t)]   (which = 0)
//                    d/dx2m[j][u] = sum_{p,q,i} G[p][q][i][j] x1m[i][u+(t)]   (which = 1)
//   as a GEMM pixels x classes with K = (q, class) per row shift p; G = gl*dR1 + gnl*dR2.
#include "common.h"
#include "../../include/iic_hip.h"

#define SEG_MAXW 256

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// value of (x * mask) at (n, ch, y, x) with optional flips of the SOURCE tensor (x2 -> x2_inv)
__device__ __forceinline__ float seg_src(const float* __restrict__ src, const float* __restrict__ mask,
                                         int n, int ch, int y, int x, int k, int h, int w, int fx,
                                         int fy) {
  const float m = mask[((long)n * h + y) * w + x];
  const int sy = fy ? h - 1 - y : y, sx = fx ? w - 1 - x : x;
  return src[(((long)n * k + ch) * h + sy) * w + sx] * m;
}

// ------------------------------------------------------------------------------------
// joint.  grid = (2T+1, ceil((2T+1)/QG), S), block = 256.
// part[s][p][q][i][j]
// ------------------------------------------------------------------------------------
template <int TK, int QG>
__global__ __launch_bounds__(256) void seg_joint_kernel(
    const float* __restrict__ x1, const float* __restrict__ x2, const float* __restrict__ mask,
    const int* __restrict__ flips, float* __restrict__ part, int bn, int k, int h, int w, int T) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int nq = 2 * T + 1;
  const int w4 = (w + 3) & ~3;
  const int P1 = (w4 + 2 * T) | 1, P2 = w4 | 1;         // odd pitches: conflict-free columns
  float* sX1 = reinterpret_cast<float*>(smem_raw);          // [16*TK][P1]
  float* sX2 = sX1 + 16 * TK * P1;                          // [16*TK][P2]
  const int p = blockIdx.x, q0 = blockIdx.y * QG, split = blockIdx.z, S = gridDim.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, kk = lane >> 4;
  const long rows = (long)bn * h;
  const long per = (rows + S - 1) / S;
  const long r0 = split * per, r1 = min(rows, r0 + per);

  f32x4 acc[QG][TK][TK];
#pragma unroll
  for (int a = 0; a < QG; ++a)
#pragma unroll
    for (int ti = 0; ti < TK; ++ti)
#pragma unroll
      for (int tj = 0; tj < TK; ++tj) acc[a][ti][tj] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (long r = r0; r < r1; ++r) {
    const int n = (int)(r / h), y = (int)(r - (long)n * h);
    const int y1 = y + p - T;
    if (y1 < 0 || y1 >= h) continue;                 // uniform: the shifted row is all padding
    const int fx = flips[2 * n], fy = flips[2 * n + 1];
    __syncthreads();
    for (int idx = tid; idx < 16 * TK * P2; idx += 256) {
      const int ch = idx / P2, x = idx - ch * P2;
      sX2[idx] = (ch < k && x < w) ? seg_src(x2, mask, n, ch, y, x, k, h, w, fx, fy) : 0.f;
    }
    for (int idx = tid; idx < 16 * TK * P1; idx += 256) {
      const int ch = idx / P1, xx = idx - ch * P1, x = xx - T;
      sX1[idx] = (ch < k && x >= 0 && x < w) ? seg_src(x1, mask, n, ch, y1, x, k, h, w, 0, 0) : 0.f;
    }
    __syncthreads();
    for (int st = wave; st < w4 / 4; st += 4) {
      const int x0 = 4 * st + kk;
      float b[TK];
#pragma unroll
      for (int tj = 0; tj < TK; ++tj) b[tj] = sX2[(tj * 16 + c) * P2 + x0];
#pragma unroll
      for (int a = 0; a < QG; ++a) {
        const int q = q0 + a;
        if (q < nq) {
#pragma unroll
          for (int ti = 0; ti < TK; ++ti) {
            const float av = sX1[(ti * 16 + c) * P1 + x0 + q];   // x + (q - T) + T
#pragma unroll
            for (int tj = 0; tj < TK; ++tj) acc[a][ti][tj] = mfma16(av, b[tj], acc[a][ti][tj]);
          }
        }
      }
    }
  }
  // cross-wave reduction, one shift at a time, through LDS (reuses the row buffers)
  float* red = reinterpret_cast<float*>(smem_raw);          // [4][TK*TK][256]
  constexpr int TT = TK * TK;
#pragma unroll
  for (int a = 0; a < QG; ++a) {
    const int q = q0 + a;
    if (q >= nq) break;
    __syncthreads();
#pragma unroll
    for (int ti = 0; ti < TK; ++ti)
#pragma unroll
      for (int tj = 0; tj < TK; ++tj)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
          red[(wave * TT + ti * TK + tj) * 256 + lane * 4 + rr] = acc[a][ti][tj][rr];
    __syncthreads();
    float* out = part + (((long)split * nq + p) * nq + q) * k * k;
    for (int idx = tid; idx < TT * 256; idx += 256) {
      const int t = idx >> 8, e = idx & 255, ln = e >> 2, rr = e & 3;
      const int i = (t / TK) * 16 + (ln >> 4) * 4 + rr, j = (t % TK) * 16 + (ln & 15);
      if (i < k && j < k)
        out[(long)i * k + j] = red[(0 * TT + t) * 256 + e] + red[(1 * TT + t) * 256 + e] +
                               red[(2 * TT + t) * 256 + e] + red[(3 * TT + t) * 256 + e];
    }
  }
}

// ------------------------------------------------------------------------------------
// gradient.  grid = bn*h (one output row each), block = 256.
//   which = 0: out = d/dx1  (reads x2m rows y - (p-T), columns x - (q-T); G[i = a][j = b])
//   which = 1: out = d/dx2  (reads x1m rows y + (p-T), columns x + (q-T); G[i = b][j = a])
//   G[h][i][j] = gl[h]*dR1[h] + gnl[h]*dR2[h],  h = shift index (stride 0 when collapsed)
// ------------------------------------------------------------------------------------
template <int TK>
__global__ __launch_bounds__(256) void seg_grad_kernel(
    const float* __restrict__ src, const float* __restrict__ mask, const int* __restrict__ flips,
    const float* __restrict__ dR1, const float* __restrict__ dR2, const float* __restrict__ gl,
    const float* __restrict__ gnl, float* __restrict__ out, int bn, int k, int h, int w, int T,
    int which, int shift_stride, int src_is_x2, int QC) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int nq = 2 * T + 1;
  const int w16 = (w + 15) & ~15;
  const int PS = (w16 + 2 * T) | 1;
  constexpr int PG = 16 * TK + 1;
  float* sS = reinterpret_cast<float*>(smem_raw);          // [k][PS]      masked source row
  float* sG = sS + (long)k * PS;                            // [QC*k (pad4)][PG]
  const int n = blockIdx.x / h, y = blockIdx.x - n * h;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, kk = lane >> 4;
  const int sgn = which == 0 ? -1 : 1;
  const int fx = flips[2 * n], fy = flips[2 * n + 1];
  const int ntile = w16 / 16;
  constexpr int MT = 4;                                    // pixel tiles per wave (w <= 256)
  f32x4 acc[MT][TK];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < TK; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int p = 0; p < nq; ++p) {
    const int ys = y + sgn * (p - T);
    if (ys < 0 || ys >= h) continue;                       // uniform
    __syncthreads();
    for (int idx = tid; idx < k * PS; idx += 256) {
      const int ch = idx / PS, xx = idx - ch * PS, x = xx - T;
      float v = 0.f;
      if (x >= 0 && x < w)
        v = src_is_x2 ? seg_src(src, mask, n, ch, ys, x, k, h, w, fx, fy)
                      : seg_src(src, mask, n, ch, ys, x, k, h, w, 0, 0);
      sS[idx] = v;
    }
    for (int qc0 = 0; qc0 < nq; qc0 += QC) {
      const int qn = min(QC, nq - qc0);
      const int KT = qn * k, KT4 = (KT + 3) & ~3;
      __syncthreads();
      for (int idx = tid; idx < KT4 * 16 * TK; idx += 256) {
        const int kidx = idx / (16 * TK), a = idx - kidx * (16 * TK);
        float v = 0.f;
        if (kidx < KT && a < k) {
          const int q = qc0 + kidx / k, b = kidx - (kidx / k) * k;
          const long hh = (long)(p * nq + q) * shift_stride;
          const long e = which == 0 ? ((long)a * k + b) : ((long)b * k + a);
          const float w1 = gl[hh], w2 = gnl ? gnl[hh] : 0.f;
          v = w1 * dR1[hh * k * k + e] + w2 * dR2[hh * k * k + e];
        }
        sG[kidx * PG + a] = v;
      }
      __syncthreads();
      for (int s = 0; s < KT4 / 4; ++s) {
        const int kidx = 4 * s + kk;
        const int q = qc0 + kidx / k, b = kidx - (kidx / k) * k;
        const bool kv = kidx < KT;
        float bv[TK];
#pragma unroll
        for (int t = 0; t < TK; ++t) bv[t] = sG[kidx * PG + t * 16 + c];
        const int xoff = sgn * (q - T) + T;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int tile = wave + 4 * m;
          if (tile < ntile) {
            const float av = kv ? sS[b * PS + tile * 16 + c + xoff] : 0.f;
#pragma unroll
            for (int t = 0; t < TK; ++t) acc[m][t] = mfma16(av, bv[t], acc[m][t]);
          }
        }
      }
    }
  }
  // store: D[row = pixel (lane>>4)*4 + r][col = class lane&15]; x mask; un-flip for x2
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int tile = wave + 4 * m;
    if (tile >= ntile) continue;
#pragma unroll
    for (int t = 0; t < TK; ++t) {
      const int a = t * 16 + c;
      if (a >= k) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int x = tile * 16 + kk * 4 + r;
        if (x < w) {
          const float v = acc[m][t][r] * mask[((long)n * h + y) * w + x];
          const int oy = (src_is_x2 == 0 && which == 1 && fy) ? h - 1 - y : y;
          const int ox = (src_is_x2 == 0 && which == 1 && fx) ? w - 1 - x : x;
          out[(((long)n * k + a) * h + oy) * w + ox] = v;
        }
      }
    }
  }
}


// ====================================================================================
// Streaming versions of the two kernels above (the default path whenever w % 4 == 0, k <= 32 and
// the tensors are 16-byte aligned).  Same tiles, same MFMA order -- identical results -- but the
// row staging no longer serialises on memory latency: the generic kernels fetch one 4-byte element
// per loop trip (a runtime division, a mask load and a source load each), i.e. ~50 dependent
// round trips to L2/HBM per staged row while the matrix core waits (measured: 20 % of the fp32
// MFMA rate at k = 24, T = 10).  Here a thread owns a fixed set of float4 units of the row
// (lane -> 4 pixels, thread row -> channels cr, cr + CR, ...), issues ALL loads of the NEXT row /
// G slice right after publishing the current one to LDS, and only waits for them after the MFMA
// loop: one round trip per row, hidden under the arithmetic.
// ====================================================================================
__device__ __forceinline__ float4 seg_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 seg_rev4(float4 v) { return make_float4(v.w, v.z, v.y, v.x); }
__device__ __forceinline__ void seg_store4(float* d, float4 v, float4 m) {
  d[0] = v.x * m.x; d[1] = v.y * m.y; d[2] = v.z * m.z; d[3] = v.w * m.w;
}
// LDS pitches of the streaming kernels.  Joint: lanes (class c, k index kk) read row c, column
// x + kk: a pitch of 2 * odd spreads the 16 rows over the even banks and kk = 0,1 over even / odd.
// Gradient: lanes read row b0 + kk, column x0 + c: a pitch = 16 (mod 32) puts rows kk = 0,1 on the
// two halves of the banks.
__host__ __device__ __forceinline__ int seg_pitch2(int n) { return (n & 3) == 2 ? n : ((n + 1) & ~3) + 2; }   // >= n, = 2 (mod 4)
__host__ __device__ __forceinline__ int seg_pitch16(int n) { return ((n + 15) & ~31) + 16; }               // >= n, = 16 (mod 32)

// MFMA loop of one staged row: the (column shift, class) pairs of this workgroup's QG shifts are
// packed densely into 16-row MFMA tiles (k = 24: 6 shifts = 144 rows = 9 full tiles, not 6 x 2
// padded ones), lane c of tile ti reads its row at sX1[aoff[ti] + x].  Tiles are issued in groups
// of GT (reads of a group before its MFMAs); groups past the mt tiles in use are skipped by one
// wave-uniform branch each.
template <int TK, int MTMAX, int GT>
__device__ __forceinline__ void seg_joint_ksteps(f32x4 (*acc)[TK], const float* sX1, const float* sX2,
                                                 const int* aoff, int P2, int nst, int mt, int wave,
                                                 int kk, int c) {
  // Fragments of step st + 4 are read while the MFMAs of step st run (left to itself the compiler
  // emits read -> wait -> 2 MFMAs per tile: the LDS latency is then paid 9 times per step).
  float b0[TK], a0[MTMAX], b1[TK], a1[MTMAX];                // two fragment sets, used alternately
  int st = wave;
  if (st >= nst) return;
  auto load = [&](float (&bb)[TK], float (&aa)[MTMAX], int stl) {
    const int x0 = 4 * stl + kk;
#pragma unroll
    for (int tj = 0; tj < TK; ++tj) bb[tj] = sX2[(tj * 16 + c) * P2 + x0];
#pragma unroll
    for (int ti = 0; ti < MTMAX; ++ti) aa[ti] = sX1[aoff[ti] + x0];
  };
  auto step = [&](float (&bc)[TK], float (&ac)[MTMAX], float (&bn)[TK], float (&an)[MTMAX]) {
    load(bn, an, st + 4 < nst ? st + 4 : st);                 // (last step: re-reads itself)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g0 = 0; g0 < MTMAX; g0 += GT) {
      if (g0 < mt) {
#pragma unroll
        for (int ti = 0; ti < GT; ++ti)
          if (g0 + ti < MTMAX) {
#pragma unroll
            for (int tj = 0; tj < TK; ++tj) acc[g0 + ti][tj] = mfma16(ac[g0 + ti], bc[tj], acc[g0 + ti][tj]);
          }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    st += 4;
  };
  load(b0, a0, st);
  while (true) {
    step(b0, a0, b1, a1);
    if (st >= nst) break;
    step(b1, a1, b0, a0);
    if (st >= nst) break;
  }
}

template <int TK, int MTMAX, int LW>
__global__ __launch_bounds__(256, 2) void seg_joint_stream_kernel(
    const float* __restrict__ x1, const float* __restrict__ x2, const float* __restrict__ mask,
    const int* __restrict__ flips, float* __restrict__ part, int bn, int k, int h, int w, int T, int QG) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int CR = 256 / LW, NJ = (16 * TK + CR - 1) / CR;
  const int nq = 2 * T + 1;
  const int w4 = w;                                        // w % 4 == 0 here
  const int P1 = seg_pitch2(w4 + 2 * T), P2 = seg_pitch2(w4);
  float* sX1 = reinterpret_cast<float*>(smem_raw);          // [16*TK][P1]
  float* sX2 = sX1 + 16 * TK * P1;                          // [16*TK][P2]
  // workgroups are dealt round-robin to the 8 XCDs by linear id: keep all (row shift, shift group)
  // workgroups of one row slice on ONE XCD, next to each other in time -- they read the same rows of
  // x1 / x2 (shifted by p), which then come from that XCD's L2 instead of HBM (measured before the
  // remap: 21.5 GB of HBM traffic per launch for 0.6 GB of operands)
  int p = blockIdx.x, grp = blockIdx.y, split = blockIdx.z;
  const int S = gridDim.z;
  if ((S & 7) == 0) {
    const int npg = gridDim.x * gridDim.y;
    const int L = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const int xcd = L & 7, j = L >> 3;
    const int pg = j % npg;
    split = xcd + 8 * (j / npg);
    p = pg % (int)gridDim.x;
    grp = pg / (int)gridDim.x;
  }
  const int q0 = grp * QG;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, kk = lane >> 4;
  const int xl = tid % LW, cr = tid / LW, wq = w >> 2;
  const bool xact = xl < wq;
  const long rows = (long)bn * h;
  const long per = (rows + S - 1) / S;
  const long r0 = split * per, r1 = min(rows, r0 + per);
  const int qn = min(QG, nq - q0);                         // column shifts of this workgroup
  const int mrows = qn * k, mt = (mrows + 15) >> 4;        // packed (shift, class) rows -> row tiles

  f32x4 acc[MTMAX][TK];
  int aoff[MTMAX];
#pragma unroll
  for (int ti = 0; ti < MTMAX; ++ti) {
#pragma unroll
    for (int tj = 0; tj < TK; ++tj) acc[ti][tj] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int m = ti * 16 + c, a = m / k, i = m - a * k;
    // rows past the packed range read the all-zero padding channel (exists whenever a tile is ragged)
    aoff[ti] = m < mrows ? i * P1 + q0 + a : (16 * TK - 1) * P1;
  }

  // halo columns and padding channels stay zero for the whole kernel: only interiors are rewritten
  for (int idx = tid; idx < 16 * TK * (P1 + P2); idx += 256) sX1[idx] = 0.f;

  float4 pre1[NJ], pre2[NJ], m1, m2;
  int pfx = 0;                                             // x flip of the row in flight
  auto next_valid = [&](long r) {
    while (r < r1) {
      const int y1 = (int)(r % h) + p - T;
      if (y1 >= 0 && y1 < h) break;                  // else the shifted row is all padding
      ++r;
    }
    return r;
  };
  auto issue = [&](long r) {
    const int n = (int)(r / h), y = (int)(r - (long)n * h), y1 = y + p - T;
    const int fx = flips[2 * n], fy = flips[2 * n + 1];
    const int sy = fy ? h - 1 - y : y, sx = fx ? wq - 1 - xl : xl;
    pfx = fx;
    if (xact) {
      m2 = seg_ld4(mask + ((long)n * h + y) * w + 4 * xl);
      m1 = seg_ld4(mask + ((long)n * h + y1) * w + 4 * xl);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int ch = cr + CR * j;
        if (ch < k) {
          // (raw values only: any arithmetic on them here would wait for the loads before the MFMA loop)
          pre2[j] = seg_ld4(x2 + (((long)n * k + ch) * h + sy) * w + 4 * sx);
          pre1[j] = seg_ld4(x1 + (((long)n * k + ch) * h + y1) * w + 4 * xl);
        }
      }
    }
  };

  long r = next_valid(r0);
  if (r < r1) issue(r);
  while (r < r1) {
    __syncthreads();                                 // everyone is done reading the previous row
    if (xact) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int ch = cr + CR * j;
        if (ch < k) {
          seg_store4(sX2 + ch * P2 + 4 * xl, pfx ? seg_rev4(pre2[j]) : pre2[j], m2);
          seg_store4(sX1 + ch * P1 + T + 4 * xl, pre1[j], m1);
        }
      }
    }
    __syncthreads();
    const long rn = next_valid(r + 1);
    if (rn < r1) issue(rn);                          // in flight during the MFMA loop below
    seg_joint_ksteps<TK, MTMAX, (TK == 1 ? 4 : (TK == 2 ? 3 : 2))>(acc, sX1, sX2, aoff, P2, w4 / 4, mt, wave, kk, c);
    r = rn;
  }
  // cross-wave reduction, one row tile at a time, through LDS (reuses the row buffers)
  float* red = reinterpret_cast<float*>(smem_raw);          // [4 waves][TK][256]
#pragma unroll
  for (int ti = 0; ti < MTMAX; ++ti) {
    if (ti >= mt) continue;
    __syncthreads();
#pragma unroll
    for (int tj = 0; tj < TK; ++tj)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) red[(wave * TK + tj) * 256 + lane * 4 + rr] = acc[ti][tj][rr];
    __syncthreads();
    for (int idx = tid; idx < TK * 256; idx += 256) {
      const int tj = idx >> 8, e = idx & 255, ln = e >> 2, rr = e & 3;
      const int m = ti * 16 + (ln >> 4) * 4 + rr, j = tj * 16 + (ln & 15);
      if (m < mrows && j < k) {
        const int a = m / k, i = m - a * k;
        float* out = part + (((long)split * nq + p) * nq + q0 + a) * k * k;
        out[(long)i * k + j] = red[(0 * TK + tj) * 256 + e] + red[(1 * TK + tj) * 256 + e] +
                               red[(2 * TK + tj) * 256 + e] + red[(3 * TK + tj) * 256 + e];
      }
    }
  }
}


// ====================================================================================
// The same contractions on the bf16 matrix pipe (round 6): every fp32 operand element is split
// into three bf16 terms  x = h + m + l  (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): 24 mantissa bits, the
// subtractions are exact in fp32) and a product is the six MFMAs  h h', h m', m h', h l', l h', m m'  accumulated in
// fp32 -- what is dropped (m l', l m', l l') is below 2^-24 of the product, i.e. inside the rounding of the exact-fp32
// MFMA path itself (measured on the golden fixtures: the loss moves in its 8th digit).  One v_mfma_f32_16x16x32_bf16
// covers 32 k values in 16 cycles where v_mfma_f32_16x16x4_f32 covers 4 in 32: six of them replace eight fp32 MFMAs at
// 3/8 of the matrix-pipe time; the split costs ~36 VALU + 8 scalar LDS reads per 8-element fragment, paid once per
// fragment, not per product.  That cost decides where this pays (see g_seg_bf16 below): the joint at k <= 16.
// ====================================================================================
__device__ __forceinline__ void seg_split8(const float (&v)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
  union { uint32_t u[4]; bf16x8 v; } H, M, L;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    float a = v[2 * p], b = v[2 * p + 1];
    H.u[p] = pack_bf16x2(a, b);
    a -= bf16lo(H.u[p]);
    b -= bf16hi(H.u[p]);
    M.u[p] = pack_bf16x2(a, b);
    a -= bf16lo(M.u[p]);
    b -= bf16hi(M.u[p]);
    L.u[p] = pack_bf16x2(a, b);
  }
  h = H.v; m = M.v; l = L.v;
}
__device__ __forceinline__ f32x4 seg_mfma6(const bf16x8& ah, const bf16x8& am, const bf16x8& al, const bf16x8& bh,
                                           const bf16x8& bm, const bf16x8& bl, f32x4 c) {
  // (smallest terms first)
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
  return c;
}
// row pitch of the bf16-path joint buffers: steps of 32 pixels read up to 31 columns past the row (zeros)
__host__ __device__ __forceinline__ int seg_w32(int w) { return (w + 31) & ~31; }

// Joint.  Same staging, work split and output layout as seg_joint_stream_kernel; a step is 32 pixels: lane
// (class row / column c, k group kg) holds pixels x0 + 8 kg .. + 7 of its row.
template <int TK, int MTMAX, int LW>
__global__ __launch_bounds__(256, 2) void seg_joint_bf16_kernel(
    const float* __restrict__ x1, const float* __restrict__ x2, const float* __restrict__ mask,
    const int* __restrict__ flips, float* __restrict__ part, int bn, int k, int h, int w, int T, int QG) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int CR = 256 / LW, NJ = (16 * TK + CR - 1) / CR;
  const int nq = 2 * T + 1;
  const int w32 = seg_w32(w);
  const int P1 = seg_pitch2(w32 + 2 * T), P2 = seg_pitch2(w32);
  float* sX1 = reinterpret_cast<float*>(smem_raw);          // [16*TK][P1]
  float* sX2 = sX1 + 16 * TK * P1;                          // [16*TK][P2]
  int p = blockIdx.x, grp = blockIdx.y, split = blockIdx.z;
  const int S = gridDim.z;
  if ((S & 7) == 0) {       // (XCD-aware slice mapping: see seg_joint_stream_kernel)
    const int npg = gridDim.x * gridDim.y;
    const int L = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const int xcd = L & 7, j = L >> 3;
    const int pg = j % npg;
    split = xcd + 8 * (j / npg);
    p = pg % (int)gridDim.x;
    grp = pg / (int)gridDim.x;
  }
  const int q0 = grp * QG;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, kg = lane >> 4;
  const int xl = tid % LW, cr = tid / LW, wq = w >> 2;
  const bool xact = xl < wq;
  const long rows = (long)bn * h;
  const long per = (rows + S - 1) / S;
  const long r0 = split * per, r1 = min(rows, r0 + per);
  const int qn = min(QG, nq - q0);
  const int mrows = qn * k, mt = (mrows + 15) >> 4;
  const int nst = w32 >> 5;

  f32x4 acc[MTMAX][TK];
  int aoff[MTMAX];
#pragma unroll
  for (int ti = 0; ti < MTMAX; ++ti) {
#pragma unroll
    for (int tj = 0; tj < TK; ++tj) acc[ti][tj] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int m = ti * 16 + c, a = m / k, i = m - a * k;
    aoff[ti] = m < mrows ? i * P1 + q0 + a : (16 * TK - 1) * P1;     // (ragged tiles read the all-zero padding channel)
  }
  for (int idx = tid; idx < 16 * TK * (P1 + P2); idx += 256) sX1[idx] = 0.f;

  float4 pre1[NJ], pre2[NJ], m1, m2;
  int pfx = 0;
  auto next_valid = [&](long r) {
    while (r < r1) {
      const int y1 = (int)(r % h) + p - T;
      if (y1 >= 0 && y1 < h) break;
      ++r;
    }
    return r;
  };
  auto issue = [&](long r) {
    const int n = (int)(r / h), y = (int)(r - (long)n * h), y1 = y + p - T;
    const int fx = flips[2 * n], fy = flips[2 * n + 1];
    const int sy = fy ? h - 1 - y : y, sx = fx ? wq - 1 - xl : xl;
    pfx = fx;
    if (xact) {
      m2 = seg_ld4(mask + ((long)n * h + y) * w + 4 * xl);
      m1 = seg_ld4(mask + ((long)n * h + y1) * w + 4 * xl);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int ch = cr + CR * j;
        if (ch < k) {
          pre2[j] = seg_ld4(x2 + (((long)n * k + ch) * h + sy) * w + 4 * sx);
          pre1[j] = seg_ld4(x1 + (((long)n * k + ch) * h + y1) * w + 4 * xl);
        }
      }
    }
  };

  long r = next_valid(r0);
  if (r < r1) issue(r);
  while (r < r1) {
    __syncthreads();
    if (xact) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int ch = cr + CR * j;
        if (ch < k) {
          seg_store4(sX2 + ch * P2 + 4 * xl, pfx ? seg_rev4(pre2[j]) : pre2[j], m2);
          seg_store4(sX1 + ch * P1 + T + 4 * xl, pre1[j], m1);
        }
      }
    }
    __syncthreads();
    const long rn = next_valid(r + 1);
    if (rn < r1) issue(rn);
    for (int st = wave; st < nst; st += 4) {
      const int x0 = 32 * st + 8 * kg;
      bf16x8 bh[TK], bm[TK], bl[TK];
#pragma unroll
      for (int tj = 0; tj < TK; ++tj) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = sX2[(tj * 16 + c) * P2 + x0 + e];
        seg_split8(v, bh[tj], bm[tj], bl[tj]);
      }
#pragma unroll
      for (int ti = 0; ti < MTMAX; ++ti) {
        if (ti < mt) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = sX1[aoff[ti] + x0 + e];
          bf16x8 ah, am, al;
          seg_split8(v, ah, am, al);
#pragma unroll
          for (int tj = 0; tj < TK; ++tj) acc[ti][tj] = seg_mfma6(ah, am, al, bh[tj], bm[tj], bl[tj], acc[ti][tj]);
        }
      }
    }
    r = rn;
  }
  // cross-wave reduction, one row tile at a time, through LDS (reuses the row buffers)
  float* red = reinterpret_cast<float*>(smem_raw);          // [4 waves][TK][256]
#pragma unroll
  for (int ti = 0; ti < MTMAX; ++ti) {
    if (ti >= mt) continue;
    __syncthreads();
#pragma unroll
    for (int tj = 0; tj < TK; ++tj)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) red[(wave * TK + tj) * 256 + lane * 4 + rr] = acc[ti][tj][rr];
    __syncthreads();
    for (int idx = tid; idx < TK * 256; idx += 256) {
      const int tj = idx >> 8, e = idx & 255, ln = e >> 2, rr = e & 3;
      const int m = ti * 16 + (ln >> 4) * 4 + rr, j = tj * 16 + (ln & 15);
      if (m < mrows && j < k) {
        const int a = m / k, i = m - a * k;
        float* out = part + (((long)split * nq + p) * nq + q0 + a) * k * k;
        out[(long)i * k + j] = red[(0 * TK + tj) * 256 + e] + red[(1 * TK + tj) * 256 + e] +
                               red[(2 * TK + tj) * 256 + e] + red[(3 * TK + tj) * 256 + e];
      }
    }
  }
}

// G of one launch, laid out as the gradient kernel's LDS image wants it:
//   Gp[p][r = q*k4 + b][a (pitch 16*TK)] = gl*dR1[p,q][e] + gnl*dR2[p,q][e],  e = a*k+b (which 0) | b*k+a
// k4 = roundup4(k); rows with b >= k and columns a >= k are zero.  grid = (nq, ceil(rowsP*PG/256)).
__global__ __launch_bounds__(256) void seg_gprep_kernel(
    const float* __restrict__ dR1, const float* __restrict__ dR2, const float* __restrict__ gl,
    const float* __restrict__ gnl, float* __restrict__ Gp, int k, int nq, int PG, int rowsP,
    int which, int shift_stride, int k4) {       // k4: classes per shift, padded (4 | 8)
  const int p = blockIdx.x;
  const int idx = blockIdx.y * 256 + threadIdx.x;
  if (idx >= rowsP * PG) return;
  const int r = idx / PG, a = idx - r * PG;
  const int q = r / k4, b = r - q * k4;
  float v = 0.f;
  if (q < nq && b < k && a < k) {
    const long hh = (long)(p * nq + q) * shift_stride;
    const long e = which == 0 ? ((long)a * k + b) : ((long)b * k + a);
    const float w1 = gl[hh], w2 = gnl ? gnl[hh] : 0.f;
    v = w1 * dR1[hh * k * k + e] + w2 * dR2[hh * k * k + e];
  }
  Gp[((long)p * rowsP + r) * PG + a] = v;
}

#define SEG_NG 9     // float4 of G per thread per slice: a slice is at most 256*9*4 floats = 36 KB

// The MFMA loop of one G slice for a wave that owns N units: one G fragment and N source
// fragments per step, all reads issued before the N MFMAs (no per-unit branches in the loop).
template <int TK, int N>
__device__ __forceinline__ void seg_grad_ksteps(f32x4 (&acc)[4 * TK], const float* sS, const float* sG,
                                                int gc, int PS, int k4, int qn, int xoff0, int sgn,
                                                int kk, const int (&ubase)[4 * TK]) {
  constexpr int PG = 16 * TK;
  // (an explicit next-step fragment prefetch, as in the joint kernel, was measured SLOWER here: with
  // only N <= 7 MFMAs per step the extra stepping arithmetic costs more than the LDS latency that
  // the second wave of the SIMD already hides -- 20.9 vs 23.0 ms at k = 24, 4.05 vs 5.3 ms at k = 15)
  for (int ql = 0; ql < qn; ++ql) {
    const int xoff = xoff0 + sgn * ql;                        // wave-uniform column offset
    const float* sGq = sG + ql * k4 * PG;
    for (int b0 = 0; b0 < k4; b0 += 4) {
      const int b = b0 + kk;                                  // class row of this lane's k index
      const int sw = TK == 2 ? ((b & 1) << 4) : 0;            // (k4 % 4 == 0: slice row parity = b parity)
      const float* srow = sS + b * PS + xoff;
      const float bv = sGq[b * PG + (gc ^ sw)];               // gc = the wave's class tile * 16 + c
      float av[N];
#pragma unroll
      for (int u = 0; u < N; ++u) av[u] = srow[ubase[u]];
#pragma unroll
      for (int u = 0; u < N; ++u) acc[u] = mfma16(av[u], bv, acc[u]);
    }
  }
}

// K order of the streaming gradient kernel: (column shift q, class b padded to k4 = roundup4(k)),
// so the four k-lanes of one MFMA step share q (the column offset is wave-uniform) and no lane
// does index arithmetic inside the loop; the padded classes multiply zero rows of G and of the
// source tile.  MFMA work is spread over the waves as (pixel tile, class tile) units, unit =
// wave + 4u: 13 pixel tiles x 2 class tiles (w = 200, k = 24) split 7/7/6/6 instead of 8/6/6/6.
template <int TK, int LW>
__global__ __launch_bounds__(256) void seg_grad_stream_kernel(
    const float* __restrict__ src, const float* __restrict__ mask, const int* __restrict__ flips,
    const float* __restrict__ Gp, float* __restrict__ out, int bn, int k, int h, int w, int T,
    int which, int src_is_x2, int QC, int rowsP) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int CR = 256 / LW, NJ = (16 * TK + CR - 1) / CR;
  constexpr int PG = 16 * TK;
  constexpr int NU = 4 * TK;                               // MFMA units per wave (w <= 256)
  const int nq = 2 * T + 1;
  const int k4 = (k + 3) & ~3;
  const int w16 = (w + 15) & ~15;
  const int PS = seg_pitch16(w16 + 2 * T);
  float* sS = reinterpret_cast<float*>(smem_raw);          // [k4][PS]     masked source row
  float* sG = sS + (((long)k4 * PS + 3) & ~3L);            // [slice rows][PG], 16-byte aligned
  // workgroups are dealt round-robin to the 8 XCDs: give each XCD a contiguous band of output rows
  // so that the 2T+1 source rows a workgroup reads are the ones its L2 neighbours just fetched
  const int per8 = gridDim.x >> 3;                         // grid = 8 * ceil(bn*h / 8)
  const int row = (blockIdx.x & 7) * per8 + (blockIdx.x >> 3);
  if (row >= bn * h) return;
  const int n = row / h, y = row - n * h;
  const int tid = threadIdx.x, lane = tid & 63;
  // the waves owning one MFMA unit more (units = wave + 4u) sit on different SIMDs in the two
  // workgroups a CU holds at a time (dispatch fills a CU's second slot 32 workgroups later)
  const int wave = ((tid >> 6) + 2 * ((blockIdx.x >> 8) & 1)) & 3;
  const int c = lane & 15, kk = lane >> 4;
  const int xl = tid % LW, cr = tid / LW, wq = w >> 2;
  const bool xact = xl < wq;
  const int sgn = which == 0 ? -1 : 1;
  const int fx = src_is_x2 ? flips[2 * n] : 0, fy = src_is_x2 ? flips[2 * n + 1] : 0;
  const int nunit = (w16 / 16) * TK;
  const int nch = (nq + QC - 1) / QC;                      // G slices per row shift
  const int nu = nunit > wave ? (nunit - wave + 3) / 4 : 0;   // units of this wave (uniform)
  f32x4 acc[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int idx = tid; idx < k4 * PS; idx += 256) sS[idx] = 0.f;   // halo + padded classes stay zero

  float4 pres[NJ], pm, preg[SEG_NG];
  auto next_valid = [&](int p) {
    while (p < nq) {
      const int ys = y + sgn * (p - T);
      if (ys >= 0 && ys < h) break;
      ++p;
    }
    return p;
  };
  // item = (row shift p, slice ch_): loads the G slice, plus the source row when ch_ == 0
  auto issue = [&](int p, int ch_) {
    const float* g = Gp + ((long)p * rowsP + (long)ch_ * QC * k4) * PG;
    const int nf4 = min(QC, nq - ch_ * QC) * k4 * (PG / 4);
#pragma unroll
    for (int j = 0; j < SEG_NG; ++j) {
      const int f = tid + 256 * j;
      if (f < nf4) preg[j] = seg_ld4(g + 4 * f);
    }
    if (ch_ == 0 && xact) {
      const int ys = y + sgn * (p - T);
      const int sy = fy ? h - 1 - ys : ys, sx = fx ? wq - 1 - xl : xl;
      pm = seg_ld4(mask + ((long)n * h + ys) * w + 4 * xl);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int ch = cr + CR * j;
        if (ch < k) {
          pres[j] = seg_ld4(src + (((long)n * k + ch) * h + sy) * w + 4 * sx);   // (raw: see the joint kernel)
        }
      }
    }
  };
  // per-unit constants: pixel tile -> column base in the source row, class tile -> column in G
  int ubase[NU], gcol[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int unit = wave + 4 * u;
    const int tile = unit / TK, t = unit - tile * TK;
    ubase[u] = tile * 16 + c;
    gcol[u] = t * 16 + c;
  }

  int p = next_valid(0), ch_ = 0;
  if (p < nq) issue(p, 0);
  while (p < nq) {
    __syncthreads();
    {
      const int nf4 = min(QC, nq - ch_ * QC) * k4 * (PG / 4);
#pragma unroll
      for (int j = 0; j < SEG_NG; ++j) {
        const int f = tid + 256 * j;
        if (f < nf4) {
          const int rr = (4 * f) / PG, a = 4 * f - rr * PG;
          // PG = 32: rows of one k-step group would hit the same banks -> odd rows swap halves
          const int as = TK == 2 ? (a ^ ((rr & 1) << 4)) : a;
          *reinterpret_cast<float4*>(sG + rr * PG + as) = preg[j];
        }
      }
    }
    if (ch_ == 0 && xact) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int ch = cr + CR * j;
        if (ch < k) {
          seg_store4(sS + ch * PS + T + 4 * xl, fx ? seg_rev4(pres[j]) : pres[j], pm);
        }
      }
    }
    __syncthreads();
    int pn = p, cn = ch_ + 1;
    if (cn == nch) { cn = 0; pn = next_valid(p + 1); }
    if (pn < nq) issue(pn, cn);
    {
      const int qc0 = ch_ * QC;
      const int qn = min(QC, nq - qc0);
      const int xoff0 = sgn * (qc0 - T) + T;
      // (a wave's units all use the same class tile: unit = wave + 4u and TK divides 4)
#define SEG_KS(N_) seg_grad_ksteps<TK, N_>(acc, sS, sG, gcol[0], PS, k4, qn, xoff0, sgn, kk, ubase)
      switch (nu) {
        case 1: SEG_KS(1); break;
        case 2: SEG_KS(2); break;
        case 3: SEG_KS(3); break;
        case 4: SEG_KS(4); break;
        case 5: if (TK == 2) SEG_KS(5); break;
        case 6: if (TK == 2) SEG_KS(6); break;
        case 7: if (TK == 2) SEG_KS(7); break;
        case 8: if (TK == 2) SEG_KS(8); break;
        default: break;
      }
#undef SEG_KS
    }
    p = pn; ch_ = cn;
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int unit = wave + 4 * u;
    if (unit >= nunit) continue;
    const int tile = unit / TK, t = unit - tile * TK;
    const int a = t * 16 + c;
    if (a >= k) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int x = tile * 16 + kk * 4 + r;
      if (x < w) {
        const float v = acc[u][r] * mask[((long)n * h + y) * w + x];
        const int oy = (src_is_x2 == 0 && which == 1 && flips[2 * n + 1]) ? h - 1 - y : y;
        const int ox = (src_is_x2 == 0 && which == 1 && flips[2 * n]) ? w - 1 - x : x;
        out[(((long)n * k + a) * h + oy) * w + ox] = v;
      }
    }
  }
}


// Streaming gradient kernel for 33 <= k <= 48 (three class tiles; round 6).  seg_grad_stream_kernel deals (pixel tile,
// class tile) units to the waves so that a wave sees ONE class tile, which needs the tile count to divide 4; here a
// wave owns pixel tiles wave, wave + 4, ... and computes all three class tiles of each: per k-step NPT source
// fragments + 3 G fragments feed 3 NPT MFMAs.  Same K order ((column shift, class padded to 4) inside a row shift),
// same staging and prefetch as the kernel above.
template <int LW>
__global__ __launch_bounds__(256) void seg_grad_stream3_kernel(
    const float* __restrict__ src, const float* __restrict__ mask, const int* __restrict__ flips,
    const float* __restrict__ Gp, float* __restrict__ out, int bn, int k, int h, int w, int T,
    int which, int src_is_x2, int QC, int rowsP) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int TK = 3;
  constexpr int CR = 256 / LW, NJ = (16 * TK + CR - 1) / CR;
  constexpr int PG = 16 * TK;
  constexpr int NPT = 4;                                   // pixel tiles per wave (w <= 256)
  const int nq = 2 * T + 1;
  const int k4 = (k + 3) & ~3;
  const int w16 = (w + 15) & ~15;
  const int PS = seg_pitch16(w16 + 2 * T);
  float* sS = reinterpret_cast<float*>(smem_raw);          // [k4][PS]
  float* sG = sS + (((long)k4 * PS + 3) & ~3L);            // [slice rows][PG]
  const int per8 = gridDim.x >> 3;
  const int row = (blockIdx.x & 7) * per8 + (blockIdx.x >> 3);
  if (row >= bn * h) return;
  const int n = row / h, y = row - n * h;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, kk = lane >> 4;
  const int xl = tid % LW, cr = tid / LW, wq = w >> 2;
  const bool xact = xl < wq;
  const int sgn = which == 0 ? -1 : 1;
  const int fx = src_is_x2 ? flips[2 * n] : 0, fy = src_is_x2 ? flips[2 * n + 1] : 0;
  const int ntile = w16 / 16;
  const int npt = ntile > wave ? (ntile - wave + 3) / 4 : 0;      // pixel tiles of this wave (uniform)
  const int nch = (nq + QC - 1) / QC;
  f32x4 acc[NPT][TK];
#pragma unroll
  for (int u = 0; u < NPT; ++u)
#pragma unroll
    for (int t = 0; t < TK; ++t) acc[u][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int idx = tid; idx < k4 * PS; idx += 256) sS[idx] = 0.f;

  float4 pres[NJ], pm, preg[SEG_NG];
  auto next_valid = [&](int p) {
    while (p < nq) {
      const int ys = y + sgn * (p - T);
      if (ys >= 0 && ys < h) break;
      ++p;
    }
    return p;
  };
  auto issue = [&](int p, int ch_) {
    const float* g = Gp + ((long)p * rowsP + (long)ch_ * QC * k4) * PG;
    const int nf4 = min(QC, nq - ch_ * QC) * k4 * (PG / 4);
#pragma unroll
    for (int j = 0; j < SEG_NG; ++j) {
      const int f = tid + 256 * j;
      if (f < nf4) preg[j] = seg_ld4(g + 4 * f);
    }
    if (ch_ == 0 && xact) {
      const int ys = y + sgn * (p - T);
      const int sy = fy ? h - 1 - ys : ys, sx = fx ? wq - 1 - xl : xl;
      pm = seg_ld4(mask + ((long)n * h + ys) * w + 4 * xl);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int ch = cr + CR * j;
        if (ch < k) pres[j] = seg_ld4(src + (((long)n * k + ch) * h + sy) * w + 4 * sx);
      }
    }
  };
  int ubase[NPT];
#pragma unroll
  for (int u = 0; u < NPT; ++u) ubase[u] = (wave + 4 * u) * 16 + c;

  int p = next_valid(0), ch_ = 0;
  if (p < nq) issue(p, 0);
  while (p < nq) {
    __syncthreads();
    {
      const int nf4 = min(QC, nq - ch_ * QC) * k4 * (PG / 4);
#pragma unroll
      for (int j = 0; j < SEG_NG; ++j) {
        const int f = tid + 256 * j;
        if (f < nf4) *reinterpret_cast<float4*>(sG + 4 * f) = preg[j];
      }
    }
    if (ch_ == 0 && xact) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int ch = cr + CR * j;
        if (ch < k) seg_store4(sS + ch * PS + T + 4 * xl, fx ? seg_rev4(pres[j]) : pres[j], pm);
      }
    }
    __syncthreads();
    int pn = p, cn = ch_ + 1;
    if (cn == nch) { cn = 0; pn = next_valid(p + 1); }
    if (pn < nq) issue(pn, cn);
    {
      const int qc0 = ch_ * QC;
      const int qn = min(QC, nq - qc0);
      const int xoff0 = sgn * (qc0 - T) + T;
      for (int ql = 0; ql < qn; ++ql) {
        const int xoff = xoff0 + sgn * ql;
        const float* sGq = sG + ql * k4 * PG;
        for (int b0 = 0; b0 < k4; b0 += 4) {
          const int b = b0 + kk;
          const float* srow = sS + b * PS + xoff;
          float bv[TK], av[NPT];
#pragma unroll
          for (int t = 0; t < TK; ++t) bv[t] = sGq[b * PG + t * 16 + c];
#pragma unroll
          for (int u = 0; u < NPT; ++u) av[u] = u < npt ? srow[ubase[u]] : 0.f;
#pragma unroll
          for (int u = 0; u < NPT; ++u)
            if (u < npt) {
#pragma unroll
              for (int t = 0; t < TK; ++t) acc[u][t] = mfma16(av[u], bv[t], acc[u][t]);
            }
        }
      }
    }
    p = pn; ch_ = cn;
  }
#pragma unroll
  for (int u = 0; u < NPT; ++u) {
    const int tile = wave + 4 * u;
    if (tile >= ntile) continue;
#pragma unroll
    for (int t = 0; t < TK; ++t) {
      const int a = t * 16 + c;
      if (a >= k) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int x = tile * 16 + kk * 4 + r;
        if (x < w) {
          const float v = acc[u][t][r] * mask[((long)n * h + y) * w + x];
          const int oy = (src_is_x2 == 0 && which == 1 && flips[2 * n + 1]) ? h - 1 - y : y;
          const int ox = (src_is_x2 == 0 && which == 1 && flips[2 * n]) ? w - 1 - x : x;
          out[(((long)n * k + a) * h + oy) * w + ox] = v;
        }
      }
    }
  }
}


// Gradient on the bf16 pipe.  Same staging, unit split and output as seg_grad_stream_kernel; K is ordered (column
// shift q, class b padded to k8 = roundup8(k)) and consumed 32 at a time: lane (pixel / class column c, k group kg)
// holds the 8 classes of ONE octet o = 4 s + kg -> (shift ql = o / K8, classes 8 (o % K8) .. + 7), so the column
// offset is per k group (per lane), not per wave.  Octets past the slice end contribute zeros.
template <int TK, int LW>
__global__ __launch_bounds__(256) void seg_grad_bf16_kernel(
    const float* __restrict__ src, const float* __restrict__ mask, const int* __restrict__ flips,
    const float* __restrict__ Gp, float* __restrict__ out, int bn, int k, int h, int w, int T,
    int which, int src_is_x2, int QC, int rowsP) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int CR = 256 / LW, NJ = (16 * TK + CR - 1) / CR;
  constexpr int PG = 16 * TK;
  constexpr int NU = 4 * TK;
  const int nq = 2 * T + 1;
  const int k8 = (k + 7) & ~7, K8 = k8 >> 3;
  const int w16 = (w + 15) & ~15;
  const int PS = seg_pitch16(w16 + 2 * T);
  float* sS = reinterpret_cast<float*>(smem_raw);          // [k8][PS]     masked source row
  float* sG = sS + (((long)k8 * PS + 3) & ~3L);            // [slice rows][PG]
  const int per8 = gridDim.x >> 3;
  const int row = (blockIdx.x & 7) * per8 + (blockIdx.x >> 3);
  if (row >= bn * h) return;
  const int n = row / h, y = row - n * h;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = (__builtin_amdgcn_readfirstlane(tid >> 6) + 2 * ((blockIdx.x >> 8) & 1)) & 3;
  const int c = lane & 15, kg = lane >> 4;
  const int xl = tid % LW, cr = tid / LW, wq = w >> 2;
  const bool xact = xl < wq;
  const int sgn = which == 0 ? -1 : 1;
  const int fx = src_is_x2 ? flips[2 * n] : 0, fy = src_is_x2 ? flips[2 * n + 1] : 0;
  const int nunit = (w16 / 16) * TK;
  const int nch = (nq + QC - 1) / QC;
  const int nu = nunit > wave ? (nunit - wave + 3) / 4 : 0;
  f32x4 acc[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int idx = tid; idx < k8 * PS; idx += 256) sS[idx] = 0.f;   // halo + padded classes stay zero

  float4 pres[NJ], pm, preg[SEG_NG];
  auto next_valid = [&](int p) {
    while (p < nq) {
      const int ys = y + sgn * (p - T);
      if (ys >= 0 && ys < h) break;
      ++p;
    }
    return p;
  };
  auto issue = [&](int p, int ch_) {
    const float* g = Gp + ((long)p * rowsP + (long)ch_ * QC * k8) * PG;
    const int nf4 = min(QC, nq - ch_ * QC) * k8 * (PG / 4);
#pragma unroll
    for (int j = 0; j < SEG_NG; ++j) {
      const int f = tid + 256 * j;
      if (f < nf4) preg[j] = seg_ld4(g + 4 * f);
    }
    if (ch_ == 0 && xact) {
      const int ys = y + sgn * (p - T);
      const int sy = fy ? h - 1 - ys : ys, sx = fx ? wq - 1 - xl : xl;
      pm = seg_ld4(mask + ((long)n * h + ys) * w + 4 * xl);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int ch = cr + CR * j;
        if (ch < k) pres[j] = seg_ld4(src + (((long)n * k + ch) * h + sy) * w + 4 * sx);
      }
    }
  };
  int ubase[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) ubase[u] = ((wave + 4 * u) / TK) * 16 + c;
  const int gc = ((wave % TK) * 16) + c;                  // (a wave's units all use one class tile: TK divides 4)

  int p = next_valid(0), ch_ = 0;
  if (p < nq) issue(p, 0);
  while (p < nq) {
    __syncthreads();
    {
      const int nf4 = min(QC, nq - ch_ * QC) * k8 * (PG / 4);
#pragma unroll
      for (int j = 0; j < SEG_NG; ++j) {
        const int f = tid + 256 * j;
        if (f < nf4) {
          const int rr = (4 * f) / PG, a = 4 * f - rr * PG;
          const int as = TK == 2 ? (a ^ ((rr & 1) << 4)) : a;
          *reinterpret_cast<float4*>(sG + rr * PG + as) = preg[j];
        }
      }
    }
    if (ch_ == 0 && xact) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int ch = cr + CR * j;
        if (ch < k) seg_store4(sS + ch * PS + T + 4 * xl, fx ? seg_rev4(pres[j]) : pres[j], pm);
      }
    }
    __syncthreads();
    int pn = p, cn = ch_ + 1;
    if (cn == nch) { cn = 0; pn = next_valid(p + 1); }
    if (pn < nq) issue(pn, cn);
    {
      const int qc0 = ch_ * QC;
      const int qn = min(QC, nq - qc0);
      const int xoff0 = sgn * (qc0 - T) + T;
      const int noct = qn * K8;
      // this lane's octet walks o = kg, kg + 4, ...: (ql, oc) kept incrementally (K8 <= 4: at most 4 wraps per step)
      int ql = 0, oc = kg;
      while (oc >= K8) { oc -= K8; ++ql; }
      for (int o0 = 0; o0 < noct; o0 += 4) {
        const bool kv = o0 + kg < noct;
        const int b0 = 8 * oc;
        const float* srow = sS + b0 * PS + xoff0 + sgn * ql;
        const float* grow = sG + (ql * k8 + b0) * PG;
        float gv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) gv[e] = kv ? grow[e * PG + (TK == 2 ? (gc ^ ((e & 1) << 4)) : gc)] : 0.f;
        bf16x8 bh, bm, bl;
        seg_split8(gv, bh, bm, bl);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          if (u < nu) {
            float av[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) av[e] = kv ? srow[e * PS + ubase[u]] : 0.f;
            bf16x8 ah, am, al;
            seg_split8(av, ah, am, al);
            acc[u] = seg_mfma6(ah, am, al, bh, bm, bl, acc[u]);
          }
        }
        oc += 4;
        while (oc >= K8) { oc -= K8; ++ql; }
      }
    }
    p = pn; ch_ = cn;
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int unit = wave + 4 * u;
    if (unit >= nunit) continue;
    const int tile = unit / TK, t = unit - tile * TK;
    const int a = t * 16 + c;
    if (a >= k) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int x = tile * 16 + kg * 4 + r;
      if (x < w) {
        const float v = acc[u][r] * mask[((long)n * h + y) * w + x];
        const int oy = (src_is_x2 == 0 && which == 1 && flips[2 * n + 1]) ? h - 1 - y : y;
        const int ox = (src_is_x2 == 0 && which == 1 && flips[2 * n]) ? w - 1 - x : x;
        out[(((long)n * k + a) * h + oy) * w + ox] = v;
      }
    }
  }
}

extern "C" {

static int seg_tk(int k) { return (k + 15) / 16; }
// streaming kernels: float4 rows (w % 4 == 0, 16-byte aligned tensors), k <= 48.  A/B: iic_debug_seg_stream(0).
IIC_SWITCH(g_seg_stream, 1, iic_debug_seg_stream)
static bool seg_stream_ok(int k, int w, const void* a, const void* b, const void* c) {
  return g_seg_stream && (w & 3) == 0 && k <= 48 &&
         ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c)) & 15) == 0;
}
// bf16-split kernels (three-term operands on v_mfma_f32_16x16x32_bf16).  Measured (tools/seg_bf16_ab.py,
// profiles/r06_seg_bf16_ab.txt; results agree with the exact-fp32 MFMA kernels to 1e-6): the split is done on the
// fly -- 8 scalar LDS reads and ~36 VALU per 8-element fragment -- and that, not the matrix pipe, then sets the time:
//   joint     k = 15: 4.38 -> 3.70 ms (1.18x), k = 3: 1.83 -> 1.10 (1.67x), k = 24: 20.1 -> 19.9 (1.01x)
//   gradient  k = 15: 4.01 -> 5.47 ms (0.73x), k = 24: 20.3 -> 24.2 (0.84x), k = 3: 1.7 -> 3.2 (0.5x)
// (one shared-operand fragment per step in the joint, one per 7 MFMA units in the gradient: the gradient pays a split
// per product group).  Splitting at staging time instead needs the three bf16 planes in LDS with 16-byte-aligned
// fragments for every column shift, i.e. the k index across image ROWS and 8 rows of both maps resident (113 + 98 KB
// at k = 15, w = 128): it does not fit.  Default (1): the joint at k <= 16, T >= 5 -- the only place it pays; the
// gradient stays on the exact-fp32 MFMA.  iic_debug_seg_bf16(0): fp32 everywhere; (2): bf16 split everywhere (A/B, tests).
IIC_SWITCH(g_seg_bf16, 1, iic_debug_seg_bf16)
static bool seg_bf16_joint_ok(int k, int T) { return g_seg_bf16 == 2 || (g_seg_bf16 == 1 && T >= 5 && k <= 16); }
static bool seg_bf16_grad_ok() { return g_seg_bf16 == 2; }
static int seg_qg(int tk) { return tk == 1 ? 21 : (tk == 2 ? 7 : 3); }
// streaming joint kernel: row tiles per workgroup (accumulators) and the column shifts per group that
// fill them -- groups balanced over the 2T+1 shifts
#define SEG_MT1 21
#define SEG_MT2 11
#define SEG_MT3 7      // (round 6: 33 <= k <= 48 -- the reference's 15- / 6-class runs overcluster with k_A = 45 / 36, commands.txt:80,89)
static int seg_stream_groups(int k, int nq, int* qg_out) {
  const int mt = seg_tk(k) == 1 ? SEG_MT1 : (seg_tk(k) == 2 ? SEG_MT2 : SEG_MT3);
  int qgmax = mt * 16 / k;
  if (qgmax < 1) qgmax = 1;
  const int groups = (nq + qgmax - 1) / qgmax;
  if (qg_out) *qg_out = (nq + groups - 1) / groups;
  return groups;
}

int iic_seg_joint_nsplit(int bn, int h, int k, int T) {
  const int nq = 2 * T + 1, tk = seg_tk(k), qg = seg_qg(tk);
  int groups = nq * ((nq + qg - 1) / qg);
  if (tk <= 3) groups = nq * seg_stream_groups(k, nq, nullptr);   // (any split count suits either kernel)
  int s = 1536 / groups;
  if (s < 1) s = 1;
  if (s >= 8) s &= ~7;                        // (multiple of 8: the XCD-aware slice mapping)
  const long rows = (long)bn * h;
  if (s > rows) s = (int)rows;
  return s;
}

int iic_seg_joint_raw(const float* x1, const float* x2, const float* mask, const int* flips,
                      float* partials, int bn, int k, int h, int w, int T, int nsplit,
                      void* stream) {
  if (!x1 || !x2 || !mask || !flips || !partials || bn <= 0 || nsplit <= 0) return IIC_ERR_ARG;
  if (k < 1 || k > 48 || T < 0 || T > 10 || w > SEG_MAXW || h < 1) return IIC_ERR_UNSUPPORTED;
  const int nq = 2 * T + 1, tk = seg_tk(k), qg = seg_qg(tk);
  const int w4 = (w + 3) & ~3;
  const size_t rows_b = (size_t)16 * tk * (((w4 + 2 * T) | 1) + (w4 | 1)) * sizeof(float);
  const size_t red_b = (size_t)4 * tk * tk * 256 * sizeof(float);
  const size_t lds = rows_b > red_b ? rows_b : red_b;
  dim3 grid(nq, (nq + qg - 1) / qg, nsplit);
  hipStream_t s = (hipStream_t)stream;
#define SEGJ(TK_, QG_)                                                                          \
  do {                                                                                          \
    if (lds > 48 * 1024)                                                                        \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&seg_joint_kernel<TK_, QG_>),     \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);          \
    hipLaunchKernelGGL((seg_joint_kernel<TK_, QG_>), grid, dim3(256), lds, s, x1, x2, mask,     \
                       flips, partials, bn, k, h, w, T);                                        \
  } while (0)
#define SEGJS(TK_, MT_, LW_)                                                                    \
  do {                                                                                          \
    if (lds > 48 * 1024)                                                                        \
      (void)hipFuncSetAttribute(                                                                \
          reinterpret_cast<const void*>(&seg_joint_stream_kernel<TK_, MT_, LW_>),               \
          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                \
    hipLaunchKernelGGL((seg_joint_stream_kernel<TK_, MT_, LW_>), sgrid, dim3(256), lds, s, x1,  \
                       x2, mask, flips, partials, bn, k, h, w, T, sqg);                         \
  } while (0)
#define SEGJS_LW(TK_, MT_)                                                                      \
  do {                                                                                          \
    if (w > 128) SEGJS(TK_, MT_, 64); else SEGJS(TK_, MT_, 32);                                 \
  } while (0)
  if (seg_stream_ok(k, w, x1, x2, mask) && seg_bf16_joint_ok(k, T)) {
    const int w32 = seg_w32(w);
    const size_t srows = (size_t)16 * tk * (seg_pitch2(w32 + 2 * T) + seg_pitch2(w32)) * sizeof(float);
    const size_t sred = (size_t)4 * tk * 256 * sizeof(float);
    const size_t lds = srows > sred ? srows : sred;
    int sqg = 1;
    const int sgroups = seg_stream_groups(k, nq, &sqg);
    dim3 sgrid(nq, sgroups, nsplit);
#define SEGJB(TK_, MT_, LW_)                                                                    \
  do {                                                                                          \
    if (lds > 48 * 1024)                                                                        \
      (void)hipFuncSetAttribute(                                                                \
          reinterpret_cast<const void*>(&seg_joint_bf16_kernel<TK_, MT_, LW_>),                 \
          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                \
    hipLaunchKernelGGL((seg_joint_bf16_kernel<TK_, MT_, LW_>), sgrid, dim3(256), lds, s, x1,    \
                       x2, mask, flips, partials, bn, k, h, w, T, sqg);                         \
  } while (0)
    if (tk == 1) { if (w > 128) SEGJB(1, SEG_MT1, 64); else SEGJB(1, SEG_MT1, 32); }
    else { if (w > 128) SEGJB(2, SEG_MT2, 64); else SEGJB(2, SEG_MT2, 32); }
  } else if (seg_stream_ok(k, w, x1, x2, mask)) {
    const size_t srows = (size_t)16 * tk * (seg_pitch2(w + 2 * T) + seg_pitch2(w)) * sizeof(float);
    const size_t sred = (size_t)4 * tk * 256 * sizeof(float);
    const size_t lds = srows > sred ? srows : sred;
    int sqg = 1;
    const int sgroups = seg_stream_groups(k, nq, &sqg);
    dim3 sgrid(nq, sgroups, nsplit);
    if (tk == 1) SEGJS_LW(1, SEG_MT1); else if (tk == 2) SEGJS_LW(2, SEG_MT2); else SEGJS_LW(3, SEG_MT3);
  } else if (tk == 1) SEGJ(1, 21);
  else if (tk == 2) SEGJ(2, 7);
  else SEGJ(3, 3);
  return iic_launch_status();
}

long iic_seg_grad_workspace_bytes(int k, int T) {
  if (k < 1 || k > 48 || T < 0 || T > 10) return 0;
  const int nq = 2 * T + 1, tk = seg_tk(k);
  return (long)nq * nq * ((k + 7) & ~7) * 16 * tk * (long)sizeof(float);      // (classes padded to 8: the bf16 path)
}

int iic_seg_grad(const float* src, const float* mask, const int* flips, const float* dR_loss,
                 const float* dR_loss_no_lamb, const float* g_loss, const float* g_loss_no_lamb,
                 float* out, int bn, int k, int h, int w, int T, int which, int collapsed,
                 float* workspace, void* stream) {
  if (!src || !mask || !flips || !dR_loss || !dR_loss_no_lamb || !g_loss || !out) return IIC_ERR_ARG;
  if (k < 1 || k > 48 || T < 0 || T > 10 || w > SEG_MAXW || bn <= 0 || h < 1) return IIC_ERR_UNSUPPORTED;
  const int nq = 2 * T + 1, tk = seg_tk(k);
  const int w16 = (w + 15) & ~15;
  const int src_is_x2 = which == 0 ? 1 : 0;   // d/dx1 reads x2m ; d/dx2 reads x1m
  hipStream_t s = (hipStream_t)stream;
  if (g_seg_stream && workspace && seg_stream_ok(k, w, src, mask, out) && ((uintptr_t)workspace & 15) == 0 &&
      seg_bf16_grad_ok()) {
    // bf16-split path: as the streaming path below, classes padded to 8 per shift
    const int PG = 16 * tk, k8 = (k + 7) & ~7, rowsP = nq * k8;
    int QC = (256 * SEG_NG * 4 / PG) / k8;
    if (QC < 1) return IIC_ERR_UNSUPPORTED;
    if (QC > nq) QC = nq;
    hipLaunchKernelGGL(seg_gprep_kernel, dim3(nq, (rowsP * PG + 255) / 256), dim3(256), 0, s,
                       dR_loss, dR_loss_no_lamb, g_loss, g_loss_no_lamb, workspace, k, nq, PG, rowsP,
                       which, collapsed ? 0 : 1, k8);
    const int PS = seg_pitch16(w16 + 2 * T);
    const size_t lds = ((((size_t)k8 * PS + 3) & ~(size_t)3) + (size_t)QC * k8 * PG) * sizeof(float);
#define SEGGB(TK_, LW_)                                                                         \
  do {                                                                                          \
    if (lds > 48 * 1024)                                                                        \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&seg_grad_bf16_kernel<TK_, LW_>), \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);          \
    hipLaunchKernelGGL((seg_grad_bf16_kernel<TK_, LW_>), dim3(8 * ((bn * h + 7) / 8)), dim3(256), lds, s, src, \
                       mask, flips, workspace, out, bn, k, h, w, T, which, src_is_x2, QC, rowsP); \
  } while (0)
#define SEGGB_LW(TK_)                                                                           \
  do {                                                                                          \
    if (w > 128) SEGGB(TK_, 64); else if (w > 64) SEGGB(TK_, 32); else SEGGB(TK_, 16);          \
  } while (0)
    if (tk == 1) SEGGB_LW(1); else SEGGB_LW(2);
    return iic_launch_status();
  }
  if (g_seg_stream && workspace && seg_stream_ok(k, w, src, mask, out) && ((uintptr_t)workspace & 15) == 0) {
    // streaming path: G laid out once per launch (workspace), then one workgroup per output row
    const int PG = 16 * tk, k4 = (k + 3) & ~3, rowsP = nq * k4;
    int QC = (256 * SEG_NG * 4 / PG) / k4;    // column shifts per G slice: SEG_NG float4 per thread
    if (QC < 1) return IIC_ERR_UNSUPPORTED;
    if (QC > nq) QC = nq;
    hipLaunchKernelGGL(seg_gprep_kernel, dim3(nq, (rowsP * PG + 255) / 256), dim3(256), 0, s,
                       dR_loss, dR_loss_no_lamb, g_loss, g_loss_no_lamb, workspace, k, nq, PG, rowsP,
                       which, collapsed ? 0 : 1, k4);
    const int PS = seg_pitch16(w16 + 2 * T);
    const size_t lds = ((((size_t)k4 * PS + 3) & ~(size_t)3) + (size_t)QC * k4 * PG) * sizeof(float);
#define SEGGS(TK_, LW_)                                                                         \
  do {                                                                                          \
    if (lds > 48 * 1024)                                                                        \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&seg_grad_stream_kernel<TK_, LW_>), \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);          \
    hipLaunchKernelGGL((seg_grad_stream_kernel<TK_, LW_>), dim3(8 * ((bn * h + 7) / 8)), dim3(256), lds, s, src, \
                       mask, flips, workspace, out, bn, k, h, w, T, which, src_is_x2, QC, rowsP); \
  } while (0)
#define SEGGS_LW(TK_)                                                                           \
  do {                                                                                          \
    if (w > 128) SEGGS(TK_, 64); else if (w > 64) SEGGS(TK_, 32); else SEGGS(TK_, 16);          \
  } while (0)
    if (tk == 1) SEGGS_LW(1);
    else if (tk == 2) SEGGS_LW(2);
    else {
#define SEGG3(LW_)                                                                              \
  do {                                                                                          \
    if (lds > 48 * 1024)                                                                        \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&seg_grad_stream3_kernel<LW_>),   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);          \
    hipLaunchKernelGGL((seg_grad_stream3_kernel<LW_>), dim3(8 * ((bn * h + 7) / 8)), dim3(256), lds, s, src, \
                       mask, flips, workspace, out, bn, k, h, w, T, which, src_is_x2, QC, rowsP); \
  } while (0)
      if (w > 128) SEGG3(64); else if (w > 64) SEGG3(32); else SEGG3(16);
    }
    return iic_launch_status();
  }
  const int PS = (w16 + 2 * T) | 1, PG = 16 * tk + 1;
  int QC = (40 * 1024) / (k * PG * 4);        // G slice kept in LDS per chunk of column shifts
  if (QC < 1) QC = 1;
  if (QC > nq) QC = nq;
  const size_t lds = ((size_t)k * PS + (size_t)((QC * k + 3) & ~3) * PG) * sizeof(float);
#define SEGG(TK_)                                                                               \
  do {                                                                                          \
    if (lds > 48 * 1024)                                                                        \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&seg_grad_kernel<TK_>),           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);          \
    hipLaunchKernelGGL((seg_grad_kernel<TK_>), dim3(bn * h), dim3(256), lds, s, src, mask,      \
                       flips, dR_loss, dR_loss_no_lamb, g_loss, g_loss_no_lamb, out, bn, k, h,  \
                       w, T, which, collapsed ? 0 : 1, src_is_x2, QC);                          \
  } while (0)
  if (tk == 1) SEGG(1);
  else if (tk == 2) SEGG(2);
  else SEGG(3);
  return iic_launch_status();
}

}  // extern "C"

// Fused IID (mutual-information) clustering loss for gfx950.
//
// Replaces /root/reference/code/utils/cluster/IID_losses.py:6-47 (≈35 elementwise /
// reduce launches per call in the reference) with three launches for ALL sub-heads:
//
//   1. iid_joint_kernel   R_part[s][h] = sum_{n in split s} z[h][n,:]^T z'[h][n,:]
//                         exact fp32 on v_mfma_f32_32x32x2_f32 (bitwise an fmaf chain);
//                         raw (un-normalised) R is additive over samples, so this is the
//                         quantity that is all-reduced over ranks.
//   2. iid_loss_kernel    R -> symmetrise -> normalise -> marginals (before clamp) ->
//                         clamp -> loss, loss_no_lamb  AND the closed-form dLoss/dR for
//                         both outputs, in float64 (k*k <= 78k elements, negligible cost).
//   3. iid_grad_kernel    dz = g . z' dR^T ,  dz' = g . z dR   (fp32 MFMA again).
//
// Gradient derivation (clamp semantics of IID_losses.py:17-19 = zero gradient through
// clamped entries, marginals taken before clamping) is restated and checked against the
// reference's autograd in oracle/iid_oracle.py::loss_and_grad_from_raw_np.
#include "common.h"
#include "../../include/iic_hip.h"

// ---------------------------------------------------------------------------------
// 1. raw joint.  grid = (nsplit, TI*TJ, H), block = 64 (one wave per 32x32 tile).
//    A[i][kk] = z[n0+kk][i0+i]   (lane l: i = l&31, kk = l>>5)  -> coalesced over i
//    B[kk][j] = z'[n0+kk][j0+j]
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void iid_joint_kernel(
    const float* __restrict__ z, const float* __restrict__ zt, float* __restrict__ part,
    int bn, int k, long head_stride, long ld, int TJ) {
  const int split = blockIdx.x, nsplit = gridDim.x;
  const int ti = blockIdx.y / TJ, tj = blockIdx.y % TJ;
  const int h = blockIdx.z, H = gridDim.z;
  const int lane = threadIdx.x;
  const int c = lane & 31, kk = lane >> 5;
  const float* zh = z + (long)h * head_stride;
  const float* zth = zt + (long)h * head_stride;
  // split the sample range in pairs so every MFMA step has both k-slots
  const int pairs = (bn + 1) >> 1;
  const int per = (pairs + nsplit - 1) / nsplit;
  const int p0 = split * per;
  const int p1 = min(pairs, p0 + per);
  const int i = ti * 32 + c, j = tj * 32 + c;
  const bool vi = i < k, vj = j < k;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  int p = p0;
  for (; p + 4 <= p1; p += 4) {
    float a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int n = 2 * (p + u) + kk;
      const bool vn = n < bn;
      a[u] = (vn && vi) ? zh[(long)n * ld + i] : 0.f;
      b[u] = (vn && vj) ? zth[(long)n * ld + j] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
  }
  for (; p < p1; ++p) {
    const int n = 2 * p + kk;
    const bool vn = n < bn;
    const float a = (vn && vi) ? zh[(long)n * ld + i] : 0.f;
    const float b = (vn && vj) ? zth[(long)n * ld + j] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  float* out = part + ((long)split * H + h) * k * k;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = ti * 32 + mfma32_row(r, lane);
    const int col = tj * 32 + c;
    if (row < k && col < k) out[(long)row * k + col] = acc[r];
  }
}

// ---------------------------------------------------------------------------------
// 2. loss + dLoss/dR from the raw joint.  grid = H, block = 1024.
//    part : [nparts][H][k][k] fp32 raw joints (summed here in float64)
//    ws   : [H][k][k] float64 scratch (symmetrised raw joint)
//    out  : loss[H], loss_no_lamb[H] (fp32), dR1[H][k][k], dR2[H][k][k] (fp32)
//           dR1 = dloss/dR, dR2 = dloss_no_lamb/dR
// ---------------------------------------------------------------------------------
#define IID_MAXK 1024
__global__ __launch_bounds__(1024) void iid_loss_kernel(
    const float* __restrict__ part, int nparts, int k, double lamb, double eps,
    double* __restrict__ ws, float* __restrict__ loss_out, float* __restrict__ loss_nl_out,
    float* __restrict__ dR1, float* __restrict__ dR2, int detach_norm) {
  __shared__ double red[32];
  __shared__ double s_pi[IID_MAXK];   // marginal (row sum of P; == col sum, P symmetric)
  __shared__ double s_rc[IID_MAXK];   // row sum of clamped P
  const int h = blockIdx.x, H = gridDim.x;
  const int tid = threadIdx.x, nt = blockDim.x;
  const long kk2 = (long)k * k;
  double* Ps = ws + (long)h * kk2;
  // pass 1: symmetrised raw joint (IID_losses.py:44) and its sum (:45)
  double s_loc = 0.0;
  for (long idx = tid; idx < kk2; idx += nt) {
    const int i = (int)(idx / k), j = (int)(idx % k);
    double a = 0.0, b = 0.0;
    for (int p = 0; p < nparts; ++p) {
      const float* R = part + ((long)p * H + h) * kk2;
      a += (double)R[(long)i * k + j];
      b += (double)R[(long)j * k + i];
    }
    const double v = 0.5 * (a + b);
    Ps[idx] = v;
    s_loc += v;
  }
  const double S = block_sum_d(s_loc, red);   // includes __syncthreads: Ps visible block-wide
  const double invS = 1.0 / S;
  // pass 2: marginals BEFORE clamping (:12-14) + row sums of the clamped joint
  {
    const int lane = tid & 63, w = tid >> 6, nw = nt >> 6;
    for (int i = w; i < k; i += nw) {   // one wave per row, lanes stride the columns
      double r = 0.0, rc = 0.0;
      for (int j = lane; j < k; j += 64) {
        const double P = Ps[(long)i * k + j] * invS;
        r += P;
        rc += (P < eps) ? eps : P;
      }
      r = wave_sum_d(r);
      rc = wave_sum_d(rc);
      if (lane == 0) {
        s_pi[i] = r;
        s_rc[i] = rc;
      }
    }
  }
  __syncthreads();
  // pass 3: the two losses (:21-31) and sum(dP o P) for the normaliser's gradient
  double l1 = 0.0, l2 = 0.0, g1 = 0.0, g2 = 0.0;
  for (long idx = tid; idx < kk2; idx += nt) {
    const int i = (int)(idx / k), j = (int)(idx % k);
    const double P = Ps[idx] * invS;
    const bool mP = !(P < eps);
    const double Pc = mP ? P : eps;
    const double pi = s_pi[i], pj = s_pi[j];
    const bool mi = !(pi < eps), mj = !(pj < eps);
    const double lP = log(Pc), li = log(mi ? pi : eps), lj = log(mj ? pj : eps);
    l1 -= Pc * (lP - lamb * lj - lamb * li);
    l2 -= Pc * (lP - lj - li);
    const double rowi = mi ? s_rc[i] / pi : 0.0;   // d/dp_i term (zero if p_i was clamped)
    const double colj = mj ? s_rc[j] / pj : 0.0;
    const double d1 = (mP ? -(lP - lamb * lj - lamb * li) - 1.0 : 0.0) + lamb * (rowi + colj);
    const double d2 = (mP ? -(lP - lj - li) - 1.0 : 0.0) + (rowi + colj);
    g1 += d1 * P;
    g2 += d2 * P;
  }
  l1 = block_sum_d(l1, red);
  l2 = block_sum_d(l2, red);
  g1 = block_sum_d(g1, red);
  g2 = block_sum_d(g2, red);
  if (detach_norm) g1 = g2 = 0.0;   // normaliser taken as a constant (segmentation, collapsed:
                                    // `current_norm = float(p_i_j.sum())`, IID_losses.py:60)
  if (tid == 0) {
    loss_out[h] = (float)l1;
    loss_nl_out[h] = (float)l2;
  }
  // pass 4: dR = (sym(dP) - sum(dP o P)) / S.  P, p_i == p_j symmetric => dP symmetric.
  float* o1 = dR1 + (long)h * kk2;
  float* o2 = dR2 + (long)h * kk2;
  for (long idx = tid; idx < kk2; idx += nt) {
    const int i = (int)(idx / k), j = (int)(idx % k);
    const double P = Ps[idx] * invS;
    const bool mP = !(P < eps);
    const double Pc = mP ? P : eps;
    const double pi = s_pi[i], pj = s_pi[j];
    const bool mi = !(pi < eps), mj = !(pj < eps);
    const double lP = log(Pc), li = log(mi ? pi : eps), lj = log(mj ? pj : eps);
    const double rowi = mi ? s_rc[i] / pi : 0.0;
    const double colj = mj ? s_rc[j] / pj : 0.0;
    const double d1 = (mP ? -(lP - lamb * lj - lamb * li) - 1.0 : 0.0) + lamb * (rowi + colj);
    const double d2 = (mP ? -(lP - lj - li) - 1.0 : 0.0) + (rowi + colj);
    o1[idx] = (float)((d1 - g1) * invS);
    o2[idx] = (float)((d2 - g2) * invS);
  }
}

// ---------------------------------------------------------------------------------
// 2b. the same k x k stage for LARGE k (over-clustering heads: k = 280 at ClusterNet6c CIFAR-20,
//     examples/commands.txt): one 1024-thread block per head is 5 blocks on a 256-CU chip and
//     6 float64 logarithms per entry per pass -- 808 us per step at H = 5, k = 280.  Four
//     launches of (G, H) blocks instead, the same arithmetic in the same float64, the per-row
//     logarithms taken once per row; every sum in a fixed order (per-block partials folded by
//     index), so the result does not depend on scheduling.
//     extra workspace after the [H][k][k] doubles: psum[H][G], part4[H][G][4], row[H][4][k]
//     (marginal, clamped-row-sum, log(clamped marginal), row term).
// ---------------------------------------------------------------------------------
#define IID_BIG_G 32
__global__ __launch_bounds__(256) void iid_big_sym_kernel(const float* __restrict__ part, int nparts, int k,
                                                          double* __restrict__ ws, double* __restrict__ psum) {
  __shared__ double red[32];
  const int g = blockIdx.x, G = gridDim.x, h = blockIdx.y, H = gridDim.y;
  const long kk2 = (long)k * k;
  const long per = (kk2 + G - 1) / G;
  const long e0 = g * per, e1 = min(kk2, e0 + per);
  double* Ps = ws + (long)h * kk2;
  double s_loc = 0.0;
  for (long idx = e0 + threadIdx.x; idx < e1; idx += 256) {
    const int i = (int)(idx / k), j = (int)(idx % k);
    double a = 0.0, b = 0.0;
    for (int p = 0; p < nparts; ++p) {
      const float* R = part + ((long)p * H + h) * kk2;
      a += (double)R[(long)i * k + j];
      b += (double)R[(long)j * k + i];
    }
    const double v = 0.5 * (a + b);
    Ps[idx] = v;
    s_loc += v;
  }
  const double t = block_sum_d(s_loc, red);
  if (threadIdx.x == 0) psum[h * G + g] = t;
}

// one wave per row: marginals before clamping, row sums of the clamped joint, per-row logarithm and row term
__global__ __launch_bounds__(256) void iid_big_rows_kernel(const double* __restrict__ ws, const double* __restrict__ psum,
                                                           int G, int k, double eps, double* __restrict__ rows) {
  const int h = blockIdx.y;
  const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= k) return;
  double S = 0.0;
  for (int g = 0; g < G; ++g) S += psum[h * G + g];
  const double invS = 1.0 / S;
  const double* Ps = ws + (long)h * k * k;
  double r = 0.0, rc = 0.0;
  for (int j = lane; j < k; j += 64) {
    const double P = Ps[(long)i * k + j] * invS;
    r += P;
    rc += (P < eps) ? eps : P;
  }
  r = wave_sum_d(r);
  rc = wave_sum_d(rc);
  if (lane == 0) {
    double* o = rows + (long)h * 4 * k;
    const bool mi = !(r < eps);
    o[i] = r;
    o[k + i] = rc;
    o[2 * k + i] = log(mi ? r : eps);
    o[3 * k + i] = mi ? rc / r : 0.0;
  }
}

// PASS 0: loss terms and sum(dP o P) per block -> part4;  PASS 1: fold part4, write the losses and dR
template <int PASS>
__global__ __launch_bounds__(256) void iid_big_terms_kernel(const double* __restrict__ ws, const double* __restrict__ psum,
                                                            const double* __restrict__ rows, double* __restrict__ part4,
                                                            int k, double lamb, double eps, int detach_norm,
                                                            float* __restrict__ loss_out, float* __restrict__ loss_nl_out,
                                                            float* __restrict__ dR1, float* __restrict__ dR2) {
  __shared__ double red[32];
  const int g = blockIdx.x, G = gridDim.x, h = blockIdx.y;
  const long kk2 = (long)k * k;
  const long per = (kk2 + G - 1) / G;
  const long e0 = g * per, e1 = min(kk2, e0 + per);
  double S = 0.0;
  for (int q = 0; q < G; ++q) S += psum[h * G + q];
  const double invS = 1.0 / S;
  const double* Ps = ws + (long)h * kk2;
  const double* li_ = rows + (long)h * 4 * k + 2 * k;
  const double* rt_ = rows + (long)h * 4 * k + 3 * k;
  double l1 = 0.0, l2 = 0.0, g1 = 0.0, g2 = 0.0;
  if (PASS == 1) {
    for (int q = 0; q < G; ++q) {
      const double* p4 = part4 + ((long)h * G + q) * 4;
      l1 += p4[0];
      l2 += p4[1];
      g1 += p4[2];
      g2 += p4[3];
    }
    if (detach_norm) g1 = g2 = 0.0;
    if (g == 0 && threadIdx.x == 0) {
      loss_out[h] = (float)l1;
      loss_nl_out[h] = (float)l2;
    }
  }
  for (long idx = e0 + threadIdx.x; idx < e1; idx += 256) {
    const int i = (int)(idx / k), j = (int)(idx % k);
    const double P = Ps[idx] * invS;
    const bool mP = !(P < eps);
    const double Pc = mP ? P : eps;
    const double lP = log(Pc), li = li_[i], lj = li_[j];
    const double rowi = rt_[i], colj = rt_[j];
    const double d1 = (mP ? -(lP - lamb * lj - lamb * li) - 1.0 : 0.0) + lamb * (rowi + colj);
    const double d2 = (mP ? -(lP - lj - li) - 1.0 : 0.0) + (rowi + colj);
    if (PASS == 0) {
      l1 -= Pc * (lP - lamb * lj - lamb * li);
      l2 -= Pc * (lP - lj - li);
      g1 += d1 * P;
      g2 += d2 * P;
    } else {
      dR1[(long)h * kk2 + idx] = (float)((d1 - g1) * invS);
      dR2[(long)h * kk2 + idx] = (float)((d2 - g2) * invS);
    }
  }
  if (PASS == 0) {
    l1 = block_sum_d(l1, red);
    l2 = block_sum_d(l2, red);
    g1 = block_sum_d(g1, red);
    g2 = block_sum_d(g2, red);
    if (threadIdx.x == 0) {
      double* p4 = part4 + ((long)h * G + g) * 4;
      p4[0] = l1;
      p4[1] = l2;
      p4[2] = g1;
      p4[3] = g2;
    }
  }
}

// ---------------------------------------------------------------------------------
// 3. input gradients.  grid = (ceil(bn/32), ceil(k/32), 2*H), block = 64.
//    which = 0:  dz [n][a] = sum_b G[a][b] z'[n][b]
//    which = 1:  dz'[n][a] = sum_b G[b][a] z [n][b]   (G symmetric, so same read)
//    G = gl[h]*dR1 + gnl[h]*dR2 combined on the fly (upstream grads stay on device).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void iid_grad_kernel(
    const float* __restrict__ z, const float* __restrict__ zt,
    const float* __restrict__ dR1, const float* __restrict__ dR2,
    const float* __restrict__ gl, const float* __restrict__ gnl,
    float* __restrict__ dz, float* __restrict__ dzt, int bn, int k, long head_stride, long ld) {
  const int h = blockIdx.z >> 1, which = blockIdx.z & 1;
  const int n0 = blockIdx.x * 32, a0 = blockIdx.y * 32;
  const int lane = threadIdx.x, c = lane & 31, kk = lane >> 5;
  const float* in = (which == 0 ? zt : z) + (long)h * head_stride;
  float* out = (which == 0 ? dz : dzt) + (long)h * head_stride;
  const float* G1 = dR1 + (long)h * k * k;
  const float* G2 = dR2 + (long)h * k * k;
  const float w1 = gl ? gl[h] : 1.f;
  const float w2 = gnl ? gnl[h] : 0.f;
  const int n = n0 + c, a = a0 + c;
  const bool vn = n < bn, va = a < k;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int b = 0; b < k; b += 2) {
    const int bb = b + kk;
    const bool vb = bb < k;
    // A[m = n][kk = b]; B[kk = b][col = a] = G[a][b] (which=0) or G[b][a] (which=1)
    const float av = (vn && vb) ? in[(long)n * ld + bb] : 0.f;
    float bv = 0.f;
    if (va && vb) {
      const long gi = which == 0 ? ((long)a * k + bb) : ((long)bb * k + a);
      bv = w1 * G1[gi] + w2 * G2[gi];
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = n0 + mfma32_row(r, lane);
    if (row < bn && a < k) out[(long)row * ld + a] = acc[r];
  }
}

// ---------------------------------------------------------------------------------
// C ABI  (declared in include/iic_hip.h)
// ---------------------------------------------------------------------------------
extern "C" {

int iic_iid_nsplit(int bn) {
  int s = (bn + 127) / 128;
  if (s < 1) s = 1;
  if (s > 16) s = 16;
  return s;
}

#define IID_BIG_MIN_K 96
long iic_iid_workspace_bytes(int H, int k) {
  return ((long)H * k * k + (long)H * IID_BIG_G * 5 + (long)H * 4 * k) * (long)sizeof(double);
}

static int iid_loss_launch(const float* partials, int nparts, int H, int k, double lamb, double eps, void* workspace,
                           float* loss, float* loss_no_lamb, float* dR1, float* dR2, int detach_norm,
                           hipStream_t st) {
  if (k < IID_BIG_MIN_K) {
    hipLaunchKernelGGL(iid_loss_kernel, dim3(H), dim3(1024), 0, st, partials, nparts, k, lamb, eps,
                       (double*)workspace, loss, loss_no_lamb, dR1, dR2, detach_norm);
    return iic_launch_status();
  }
  double* ws = (double*)workspace;
  double* psum = ws + (long)H * k * k;
  double* part4 = psum + (long)H * IID_BIG_G;
  double* rows = part4 + (long)H * IID_BIG_G * 4;
  const dim3 grid(IID_BIG_G, H);
  hipLaunchKernelGGL(iid_big_sym_kernel, grid, dim3(256), 0, st, partials, nparts, k, ws, psum);
  hipLaunchKernelGGL(iid_big_rows_kernel, dim3((k + 3) / 4, H), dim3(256), 0, st, ws, psum, IID_BIG_G, k, eps, rows);
  hipLaunchKernelGGL(iid_big_terms_kernel<0>, grid, dim3(256), 0, st, ws, psum, rows, part4, k, lamb, eps,
                     detach_norm, loss, loss_no_lamb, dR1, dR2);
  hipLaunchKernelGGL(iid_big_terms_kernel<1>, grid, dim3(256), 0, st, ws, psum, rows, part4, k, lamb, eps,
                     detach_norm, loss, loss_no_lamb, dR1, dR2);
  return iic_launch_status();
}

int iic_iid_joint_raw(const float* z, const float* zt, float* partials, int H, int bn, int k,
                      long head_stride, long ld, int nsplit, void* stream) {
  if (!z || !zt || !partials || H <= 0 || bn <= 0 || k <= 0 || nsplit <= 0 || ld < k) return IIC_ERR_ARG;
  const int T = (k + 31) / 32;
  dim3 grid(nsplit, T * T, H);
  hipLaunchKernelGGL(iid_joint_kernel, grid, dim3(64), 0, (hipStream_t)stream, z, zt, partials, bn,
                     k, head_stride, ld, T);
  return iic_launch_status();
}

int iic_iid_loss_from_joint(const float* partials, int nparts, int H, int k, double lamb,
                            double eps, void* workspace, float* loss, float* loss_no_lamb,
                            float* dR_loss, float* dR_loss_no_lamb, void* stream) {
  if (!partials || !workspace || !loss || !loss_no_lamb || !dR_loss || !dR_loss_no_lamb)
    return IIC_ERR_ARG;
  if (H <= 0 || k <= 0 || k > IID_MAXK || nparts <= 0) return IIC_ERR_ARG;
  return iid_loss_launch(partials, nparts, H, k, lamb, eps, workspace, loss, loss_no_lamb, dR_loss,
                         dR_loss_no_lamb, 0, (hipStream_t)stream);
}

// Same k x k stage for the segmentation losses: H = number of shifts (uncollapsed, one joint
// per shift, normaliser differentiable) or H = 1 with the shifts summed as `nparts`
// (collapsed, detach_norm = 1).
int iic_seg_loss_from_joint(const float* partials, int nparts, int H, int k, double lamb,
                            double eps, void* workspace, float* loss, float* loss_no_lamb,
                            float* dR_loss, float* dR_loss_no_lamb, int detach_norm, void* stream) {
  if (!partials || !workspace || !loss || !loss_no_lamb || !dR_loss || !dR_loss_no_lamb)
    return IIC_ERR_ARG;
  if (H <= 0 || k <= 0 || k > IID_MAXK || nparts <= 0) return IIC_ERR_ARG;
  return iid_loss_launch(partials, nparts, H, k, lamb, eps, workspace, loss, loss_no_lamb, dR_loss,
                         dR_loss_no_lamb, detach_norm, (hipStream_t)stream);
}

int iic_iid_grad(const float* z, const float* zt, const float* dR_loss,
                 const float* dR_loss_no_lamb, const float* g_loss, const float* g_loss_no_lamb,
                 float* dz, float* dzt, int H, int bn, int k, long head_stride, long ld,
                 void* stream) {
  if (!z || !zt || !dR_loss || !dR_loss_no_lamb || !dz || !dzt) return IIC_ERR_ARG;
  if (H <= 0 || bn <= 0 || k <= 0 || ld < k) return IIC_ERR_ARG;
  dim3 grid((bn + 31) / 32, (k + 31) / 32, 2 * H);
  hipLaunchKernelGGL(iid_grad_kernel, grid, dim3(64), 0, (hipStream_t)stream, z, zt, dR_loss,
                     dR_loss_no_lamb, g_loss, g_loss_no_lamb, dz, dzt, bn, k, head_stride, ld);
  return iic_launch_status();
}

}  // extern "C"

// General affine warp + whole-batch integer shift of the segmentation loss's second view:
//   perform_affine_tf          /root/reference/code/utils/segmentation/transforms.py:131-143
//                              (F.affine_grid + F.grid_sample, bilinear, zero padding)
//   random_translation_multiple  transforms.py:145-165 (zero pad + crop = shift with zero fill)
// Every published run uses identity / x-flip matrices and no sparse shift; those stay folded into
// the joint / gradient kernels' index arithmetic (seg_loss.hip).  This file is the general case.
//
// The host converts theta (normalised coordinates) to a pixel-space matrix per sample
// (iic_amd/seg_losses.py::_pixel_matrices): source pixel (ix, iy) = M * (ox, oy, 1).
//   out[n][k][oy][ox] = 0                                   if (oy + sy, ox + sx) is outside the image
//                     = bilinear(x[n][k], M_n * (ox + sx, oy + sy, 1))  otherwise (zeros outside)
// NCHW fp32 (the loss inputs are the fp32 softmax maps).  HBM-bound: one read + one write per element.
#include "common.h"
#include "../../include/iic_hip.h"

struct WarpTap {
  int o00, o01, o10, o11;      // offsets inside one (n, k) plane, -1 = outside
  float w00, w01, w10, w11;
};

__device__ __forceinline__ bool warp_taps(const float* __restrict__ M, int n, int oy, int ox, int H, int W,
                                          int sx, int sy, WarpTap& t) {
  const int qx = ox + sx, qy = oy + sy;
  if (qx < 0 || qx >= W || qy < 0 || qy >= H) return false;
  const float* m = M + (long)n * 6;
  // coordinates in double: the reference's float64 runs are the parity target
  const double fx = (double)m[0] * qx + (double)m[1] * qy + (double)m[2];
  const double fy = (double)m[3] * qx + (double)m[4] * qy + (double)m[5];
  const double x0d = floor(fx), y0d = floor(fy);
  if (x0d < -1.0 || x0d >= (double)W || y0d < -1.0 || y0d >= (double)H) {
    t.o00 = t.o01 = t.o10 = t.o11 = -1;
    t.w00 = t.w01 = t.w10 = t.w11 = 0.f;
    return true;
  }
  const int x0 = (int)x0d, y0 = (int)y0d;
  const float ax = (float)(fx - x0d), ay = (float)(fy - y0d);
  const bool vx0 = x0 >= 0, vx1 = x0 + 1 < W, vy0 = y0 >= 0, vy1 = y0 + 1 < H;
  t.o00 = (vx0 && vy0) ? y0 * W + x0 : -1;
  t.o01 = (vx1 && vy0) ? y0 * W + x0 + 1 : -1;
  t.o10 = (vx0 && vy1) ? (y0 + 1) * W + x0 : -1;
  t.o11 = (vx1 && vy1) ? (y0 + 1) * W + x0 + 1 : -1;
  t.w00 = (1.f - ax) * (1.f - ay);
  t.w01 = ax * (1.f - ay);
  t.w10 = (1.f - ax) * ay;
  t.w11 = ax * ay;
  return true;
}

__global__ __launch_bounds__(256) void affine_warp_fwd_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ M,
                                                              float* __restrict__ out, int N, int K, int H,
                                                              int W, int sx, int sy) {
  const long hw = (long)H * W;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)N * hw) return;
  const int n = (int)(i / hw);
  const int r = (int)(i - (long)n * hw);
  const int oy = r / W, ox = r - oy * W;
  WarpTap t;
  const bool inside = warp_taps(M, n, oy, ox, H, W, sx, sy, t);
  const float* xp = x + (long)n * K * hw;
  float* op = out + (long)n * K * hw + r;
  for (int k = 0; k < K; ++k) {
    float v = 0.f;
    if (inside) {
      if (t.o00 >= 0) v += t.w00 * xp[t.o00];
      if (t.o01 >= 0) v += t.w01 * xp[t.o01];
      if (t.o10 >= 0) v += t.w10 * xp[t.o10];
      if (t.o11 >= 0) v += t.w11 * xp[t.o11];
    }
    op[(long)k * hw] = v;
    xp += hw;
  }
}

// dx (zeroed here) += scatter of dout through the same taps.  Float atomics: the order of the
// (at most a few) contributions to one source pixel is not fixed -- the general-affine path is
// the one place of the library whose gradient is not bit-reproducible; no published run uses it.
__global__ __launch_bounds__(256) void affine_warp_bwd_kernel(const float* __restrict__ dout,
                                                              const float* __restrict__ M,
                                                              float* __restrict__ dx, int N, int K, int H,
                                                              int W, int sx, int sy) {
  const long hw = (long)H * W;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)N * hw) return;
  const int n = (int)(i / hw);
  const int r = (int)(i - (long)n * hw);
  const int oy = r / W, ox = r - oy * W;
  WarpTap t;
  if (!warp_taps(M, n, oy, ox, H, W, sx, sy, t)) return;
  const float* gp = dout + (long)n * K * hw + r;
  float* dp = dx + (long)n * K * hw;
  for (int k = 0; k < K; ++k) {
    const float g = gp[(long)k * hw];
    if (g != 0.f) {
      if (t.o00 >= 0) atomicAdd(dp + t.o00, t.w00 * g);
      if (t.o01 >= 0) atomicAdd(dp + t.o01, t.w01 * g);
      if (t.o10 >= 0) atomicAdd(dp + t.o10, t.w10 * g);
      if (t.o11 >= 0) atomicAdd(dp + t.o11, t.w11 * g);
    }
    dp += hw;
  }
}

extern "C" {

int iic_affine_warp_fwd(const float* x, const float* pixel_mats, float* out, int N, int K, int H, int W,
                        int shift_x, int shift_y, void* stream) {
  if (!x || !pixel_mats || !out || N <= 0 || K <= 0 || H <= 0 || W <= 0) return IIC_ERR_ARG;
  const long total = (long)N * H * W;
  hipLaunchKernelGGL(affine_warp_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, x, pixel_mats, out, N, K, H, W, shift_x, shift_y);
  return iic_launch_status();
}

int iic_affine_warp_bwd(const float* dout, const float* pixel_mats, float* dx, int N, int K, int H, int W,
                        int shift_x, int shift_y, void* stream) {
  if (!dout || !pixel_mats || !dx || N <= 0 || K <= 0 || H <= 0 || W <= 0) return IIC_ERR_ARG;
  const long total = (long)N * H * W;
  if (iic_zero_async(dx, (size_t)total * K * sizeof(float), (hipStream_t)stream) != IIC_OK) return IIC_ERR_LAUNCH;
  hipLaunchKernelGGL(affine_warp_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, dout, pixel_mats, dx, N, K, H, W, shift_x, shift_y);
  return iic_launch_status();
}

}  // extern "C"

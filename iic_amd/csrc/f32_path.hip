// Exact-fp32 path of the residual / VGG trunks (SURVEY.md §8c parity tier T2).
//
// The production path stores activations in bf16 and feeds bf16 operands to the MFMA convolutions;
// what separates it from the fp32 reference is therefore rounding, amplified chaotically by 33
// batch-statistics BatchNorm layers -- whole-net comparisons can only use aggregates.  This file is
// the SAME dataflow in plain fp32: PT tensors hold floats, every kernel below has the contract of
// its bf16 counterpart (same geometry descriptor, same epilogue flags, same exact statistic
// accumulators), written as straightforward one-thread-per-output loops.  It exists for ONE purpose:
// `iic_amd.ops.fp32_mode()` runs the unmodified host orchestration (autograd Functions, BatchNorm
// semantics, pre-masked gradient chain, heads, loss, optimiser) through it, so that a whole
// ClusterNet5g train step can be held to ~1e-4 against the reference's fp32 golden
// (tests/test_gpu_net.py::test_net5g_fp32_mode_vs_reference_golden).  It is not tuned and is never
// selected by the product path (bf16) -- correctness instrument, not a fallback.
//
// Replaces, in fp32: nn.Conv2d fwd / bwd-data / bwd-weight, nn.BatchNorm2d + ReLU + residual add
// (residual.py:20-41), nn.MaxPool2d(2, 2, padding=1) (net5g.py:26), nn.AvgPool2d (net5g.py:31-39).
#include "common.h"
#include "conv_tile.h"
#include "../../include/iic_hip.h"

// ---- convolution: out[pout(m)][co] = sum_t sum_ci in[pin(m)+tap_off[t]][ci] * W(co, ci, tap_w[t])
// w: the fp32 OIHW parameter itself.  transposed = 0: O = co, I = ci (forward);  1: O = ci, I = co
// (backward-data: the geometry's output channels are the parameter's input channels).
__global__ __launch_bounds__(256) void f32_conv_kernel(const iic_conv_geom g, const float* __restrict__ in,
                                                       const float* __restrict__ w, int wtaps, int transposed,
                                                       float* __restrict__ out, float* __restrict__ stats,
                                                       const float* __restrict__ res_grad,
                                                       const float* __restrict__ res_act, int accumulate) {
  const long total = (long)igemm_rows(g) * g.Cout;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int co = (int)(idx % g.Cout);
  const int m = (int)(idx / g.Cout);
  int pin, pout;
  igemm_row_pixels(g, m, pin, pout);
  if (pout < 0) return;
  float acc = 0.f;
  for (int t = 0; t < g.ntaps; ++t) {
    const float* xp = in + ((long)pin + g.tap_off[t]) * g.Cin;
    const int tw = g.tap_w[t];
    for (int ci = 0; ci < g.Cin; ++ci) {
      const float wv = transposed ? w[((long)ci * g.Cout + co) * wtaps + tw] : w[((long)co * g.Cin + ci) * wtaps + tw];
      acc = fmaf(xp[ci], wv, acc);
    }
  }
  if (stats) {      // every element adds itself: exact accumulators make the order irrelevant
    iic_stat_add(stats, blockIdx.x % IIC_STAT_STRIPES, g.Cout, co, 0, acc);
    iic_stat_add(stats, blockIdx.x % IIC_STAT_STRIPES, g.Cout, co, 1, acc * acc);
  }
  const long o = (long)pout * g.Cout + co;
  float f = acc;
  const bool add_prev = accumulate & IIC_ACC_ADD, premask = accumulate & IIC_ACC_PREMASK;
  if (add_prev) f += out[o];
  if (premask) {
    if (res_grad) f += res_grad[o];
    if (res_act && !(res_act[o] > 0.f)) f = 0.f;
  } else if (res_grad) {
    if (res_act[o] > 0.f) f += res_grad[o];
  }
  out[o] = f;
}

// ---- weight gradient: dW[co][ci][t] (OIHW) (+)= sum_m dy[pout(m)][co] * x[pin(m)+tap_off[t]][ci]
__global__ __launch_bounds__(256) void f32_wgrad_kernel(const iic_conv_geom g, const float* __restrict__ x,
                                                        const float* __restrict__ dy, float* __restrict__ dW,
                                                        int wtaps, int accumulate) {
  const long total = (long)g.Cout * g.Cin * g.ntaps;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int t = (int)(idx % g.ntaps);
  const int ci = (int)((idx / g.ntaps) % g.Cin);
  const int co = (int)(idx / ((long)g.ntaps * g.Cin));
  const int M = igemm_rows(g);
  float acc = 0.f;
  for (int m = 0; m < M; ++m) {
    int pin, pout;
    igemm_row_pixels(g, m, pin, pout);
    if (pout < 0) continue;
    acc = fmaf(dy[(long)pout * g.Cout + co], x[((long)pin + g.tap_off[t]) * g.Cin + ci], acc);
  }
  const long o = ((long)co * g.Cin + ci) * wtaps + g.tap_w[t];
  dW[o] = accumulate ? dW[o] + acc : acc;
}

// ---- BatchNorm streaming kernels on fp32 PT tensors (contracts of bn.hip) ------------------------
__device__ __forceinline__ long f32_pt_off(long e, int H, int W, int P, int C, int& c) {
  c = (int)(e % C);
  long px = e / C;
  const int x = (int)(px % W);
  px /= W;
  const int y = (int)(px % H);
  const long n = px / H;
  return ((n * (H + 2 * P) + y + P) * (W + 2 * P) + x + P) * C + c;
}

__global__ __launch_bounds__(256) void f32_bn_apply_kernel(const float* __restrict__ y, const float* __restrict__ coef,
                                                           const float* __restrict__ res, const float* __restrict__ y2,
                                                           const float* __restrict__ coef2, float* __restrict__ out,
                                                           long total, int H, int W, int P, int C, int relu) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  int c;
  const long o = f32_pt_off(e, H, W, P, C, c);
  float v = y[o] * coef[c] + coef[C + c];
  if (res) v += res[o];
  if (y2) v += y2[o] * coef2[c] + coef2[C + c];
  out[o] = relu ? fmaxf(v, 0.f) : v;
}

// g = dout [* (act > 0)] [* (scale*y + shift > 0)]
__device__ __forceinline__ float f32_masked(const float* dout, const float* act, const float* y,
                                            const float* mcoef, long o, int c, int C) {
  float g = dout[o];
  if (act && !(act[o] > 0.f)) g = 0.f;
  if (mcoef && !(y[o] * mcoef[c] + mcoef[C + c] > 0.f)) g = 0.f;
  return g;
}

__global__ __launch_bounds__(256) void f32_bn_bwd_reduce_kernel(const float* __restrict__ dout,
                                                                const float* __restrict__ act,
                                                                const float* __restrict__ y,
                                                                const float* __restrict__ y2, float* __restrict__ sums,
                                                                float* __restrict__ sums2,
                                                                const float* __restrict__ mcoef, long total, int H,
                                                                int W, int P, int C) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  int c;
  const long o = f32_pt_off(e, H, W, P, C, c);
  const float g = f32_masked(dout, act, y, mcoef, o, c, C);
  const int stripe = blockIdx.x % IIC_STAT_STRIPES;
  iic_stat_add(sums, stripe, C, c, 0, g);
  iic_stat_add(sums, stripe, C, c, 1, g * y[o]);
  if (y2) {
    iic_stat_add(sums2, stripe, C, c, 0, g);
    iic_stat_add(sums2, stripe, C, c, 1, g * y2[o]);
  }
}

__global__ __launch_bounds__(256) void f32_bn_bwd_apply_kernel(const float* __restrict__ dout,
                                                               const float* __restrict__ act,
                                                               const float* __restrict__ y,
                                                               const float* __restrict__ bcoef, float* __restrict__ dy,
                                                               const float* __restrict__ y2,
                                                               const float* __restrict__ bcoef2, float* __restrict__ dy2,
                                                               const float* __restrict__ mcoef, long total, int H,
                                                               int W, int P, int C) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  int c;
  const long o = f32_pt_off(e, H, W, P, C, c);
  const float g = f32_masked(dout, act, y, mcoef, o, c, C);
  dy[o] = bcoef[c] * g + bcoef[C + c] * y[o] + bcoef[2 * C + c];
  if (y2) dy2[o] = bcoef2[c] * g + bcoef2[C + c] * y2[o] + bcoef2[2 * C + c];
}

// ---- pools ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void f32_avgpool_fwd_kernel(const float* __restrict__ in, float* __restrict__ feats,
                                                              int N, int H, int W, int P, int C) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)N * C) return;
  const int c = (int)(idx % C);
  const long n = idx / C;
  float s = 0.f;
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) s += in[((n * (H + 2 * P) + y + P) * (W + 2 * P) + x + P) * C + c];
  feats[idx] = s * (1.f / (float)(H * W));
}

__global__ __launch_bounds__(256) void f32_avgpool_bwd_kernel(const float* __restrict__ dfeats, float* __restrict__ din,
                                                              const float* __restrict__ act, long total, int H, int W,
                                                              int P, int C) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  int c;
  const long o = f32_pt_off(e, H, W, P, C, c);
  const long n = e / ((long)C * W * H);
  float v = dfeats[n * C + c] * (1.f / (float)(H * W));
  if (act && !(act[o] > 0.f)) v = 0.f;
  din[o] = v;
}

// nn.MaxPool2d(kernel_size=2, stride=2, padding=1): out[ho][wo] = max over rows 2ho-1, 2ho and
// columns 2wo-1, 2wo inside the image.  PT in (P = 1) -> PT out (P = 1).
__global__ __launch_bounds__(256) void f32_maxpool_fwd_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                              long total, int H, int W, int C) {
  const int Ho = H / 2 + 1, Wo = W / 2 + 1;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % C);
  long px = e / C;
  const int wo = (int)(px % Wo);
  px /= Wo;
  const int ho = (int)(px % Ho);
  const long n = px / Ho;
  float m = -INFINITY;
  for (int dy = 0; dy < 2; ++dy)
    for (int dx = 0; dx < 2; ++dx) {
      const int y = 2 * ho - 1 + dy, x = 2 * wo - 1 + dx;
      if (y < 0 || y >= H || x < 0 || x >= W) continue;
      m = fmaxf(m, in[((n * (H + 2) + y + 1) * (W + 2) + x + 1) * C + c]);
    }
  out[((n * (Ho + 2) + ho + 1) * (Wo + 2) + wo + 1) * C + c] = m;
}

// din (zeroed by the caller... overwritten here: every input pixel belongs to exactly one window)
// = dout of its window if it is the window's FIRST maximum in scan order (torch's rule), else 0.
__global__ __launch_bounds__(256) void f32_maxpool_bwd_kernel(const float* __restrict__ in, const float* __restrict__ dout,
                                                              float* __restrict__ din, long total, int H, int W,
                                                              int C) {
  const int Ho = H / 2 + 1, Wo = W / 2 + 1;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % C);
  long px = e / C;
  const int x = (int)(px % W);
  px /= W;
  const int y = (int)(px % H);
  const long n = px / H;
  const int ho = (y + 1) / 2, wo = (x + 1) / 2;
  // the window's arg-max, first maximum in scan order (strict > update): torch's routing rule
  float best = -INFINITY;
  int by = -1, bx = -1;
  for (int dy = 0; dy < 2; ++dy)
    for (int dx = 0; dx < 2; ++dx) {
      const int yy = 2 * ho - 1 + dy, xx = 2 * wo - 1 + dx;
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
      const float v = in[((n * (H + 2) + yy + 1) * (W + 2) + xx + 1) * C + c];
      if (by < 0 || v > best) { best = v; by = yy; bx = xx; }
    }
  const bool first = (by == y && bx == x);
  din[((n * (H + 2) + y + 1) * (W + 2) + x + 1) * C + c] =
      first ? dout[((n * (Ho + 2) + ho + 1) * (Wo + 2) + wo + 1) * C + c] : 0.f;
}

// nn.MaxPool2d(2, 2) of the VGG-style trunks (vgg.py:31-32): PT (border Pi) -> PT (border Po)
__global__ __launch_bounds__(256) void f32_maxpool2_fwd_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                               long total, int H, int W, int Pi, int Po, int C) {
  const int Ho = H / 2, Wo = W / 2;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % C);
  long px = e / C;
  const int xo = (int)(px % Wo);
  px /= Wo;
  const int yo = (int)(px % Ho);
  const long n = px / Ho;
  float m = -INFINITY;
  for (int q = 0; q < 4; ++q)
    m = fmaxf(m, in[((n * (H + 2 * Pi) + 2 * yo + (q >> 1) + Pi) * (W + 2 * Pi) + 2 * xo + (q & 1) + Pi) * C + c]);
  out[((n * (Ho + 2 * Po) + yo + Po) * (Wo + 2 * Po) + xo + Po) * C + c] = m;
}

__global__ __launch_bounds__(256) void f32_maxpool2_bwd_kernel(const float* __restrict__ in, const float* __restrict__ dout,
                                                               float* __restrict__ din, long total, int H, int W,
                                                               int Pi, int Po, int C) {
  const int Ho = H / 2, Wo = W / 2;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % C);
  long px = e / C;
  const int x = (int)(px % W);
  px /= W;
  const int y = (int)(px % H);
  const long n = px / H;
  const int yo = y / 2, xo = x / 2;
  float g = 0.f;
  if (yo < Ho && xo < Wo) {      // odd trailing rows / columns belong to no window
    float best = -INFINITY;
    int bq = -1;
    for (int q = 0; q < 4; ++q) {
      const float v = in[((n * (H + 2 * Pi) + 2 * yo + (q >> 1) + Pi) * (W + 2 * Pi) + 2 * xo + (q & 1) + Pi) * C + c];
      if (bq < 0 || v > best) { best = v; bq = q; }
    }
    if (2 * yo + (bq >> 1) == y && 2 * xo + (bq & 1) == x)
      g = dout[((n * (Ho + 2 * Po) + yo + Po) * (Wo + 2 * Po) + xo + Po) * C + c];
  }
  din[((n * (H + 2 * Pi) + y + Pi) * (W + 2 * Pi) + x + Pi) * C + c] = g;
}

// SegmentationNet10a head (net10a.py:44-59): window of the PT feature map <-> fp32 matrix
__global__ __launch_bounds__(256) void f32_window_gather_kernel(const float* __restrict__ pt, float* __restrict__ out,
                                                                long total, int Hw, int Ww, int Hp, int Wp, int off,
                                                                int C) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % C);
  const long m = e / C;
  const int xx = (int)(m % Ww);
  const long r = m / Ww;
  const int yy = (int)(r % Hw);
  const long n = r / Hw;
  out[e] = pt[((n * Hp + yy + off) * Wp + xx + off) * C + c];
}
__global__ __launch_bounds__(256) void f32_window_scatter_kernel(const float* __restrict__ in, float* __restrict__ pt,
                                                                 long total, int Hw, int Ww, int Hp, int Wp, int off,
                                                                 int C) {
  const int Hi = Hw - 2, Wi = Ww - 2;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % C);
  const long q = e / C;
  const int x = (int)(q % Wi);
  const long r = q / Wi;
  const int y = (int)(r % Hi);
  const long n = r / Hi;
  pt[((n * Hp + y + 1 + off) * Wp + x + 1 + off) * C + c] = in[((n * Hw + y + 1) * Ww + x + 1) * C + c];
}

// NCHW fp32 image -> PT fp32 (interior only; the buffer's border stays zero)
__global__ __launch_bounds__(256) void f32_nchw_to_pt_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                             long total, int C, int H, int W, int P) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int xx = (int)(e % W);
  long r = e / W;
  const int y = (int)(r % H);
  r /= H;
  const int c = (int)(r % C);
  const long n = r / C;
  out[((n * (H + 2 * P) + y + P) * (W + 2 * P) + xx + P) * C + c] = x[e];
}

#define F32_GRID(total) dim3((unsigned)(((total) + 255) / 256)), dim3(256), 0, (hipStream_t)stream

extern "C" {

int iic_f32_conv(const iic_conv_geom* g, const float* in, const float* w_oihw, int wtaps, int transposed,
                 float* out, float* stats, const float* res_grad, const float* res_act, int accumulate,
                 void* stream) {
  if (!g || !in || !w_oihw || !out || wtaps <= 0) return IIC_ERR_ARG;
  if (!(accumulate & IIC_ACC_PREMASK) && (res_grad == nullptr) != (res_act == nullptr)) return IIC_ERR_ARG;
  const long total = igemm_rows_host(g) * g->Cout;
  if (total <= 0) return IIC_ERR_ARG;
  hipLaunchKernelGGL(f32_conv_kernel, F32_GRID(total), *g, in, w_oihw, wtaps, transposed, out, stats, res_grad,
                     res_act, accumulate);
  return iic_launch_status();
}

int iic_f32_wgrad(const iic_conv_geom* g, const float* x, const float* dy, float* dW_oihw, int wtaps,
                  int accumulate, void* stream) {
  if (!g || !x || !dy || !dW_oihw || wtaps <= 0) return IIC_ERR_ARG;
  const long total = (long)g->Cout * g->Cin * g->ntaps;
  hipLaunchKernelGGL(f32_wgrad_kernel, F32_GRID(total), *g, x, dy, dW_oihw, wtaps, accumulate);
  return iic_launch_status();
}

int iic_f32_bn_apply(const float* y, const float* coef, const float* res, const float* y2, const float* coef2,
                     float* out, int N, int H, int W, int P, int C, int relu, void* stream) {
  if (!y || !coef || !out || N <= 0 || (y2 == nullptr) != (coef2 == nullptr)) return IIC_ERR_ARG;
  const long total = (long)N * H * W * C;
  hipLaunchKernelGGL(f32_bn_apply_kernel, F32_GRID(total), y, coef, res, y2, coef2, out, total, H, W, P, C, relu);
  return iic_launch_status();
}

int iic_f32_bn_bwd_reduce(const float* dout, const float* act, const float* y, const float* y2, float* sums,
                          float* sums2, const float* mask_coef, int N, int H, int W, int P, int C,
                          void* stream) {
  if (!dout || !y || !sums || N <= 0 || (y2 == nullptr) != (sums2 == nullptr)) return IIC_ERR_ARG;
  const long total = (long)N * H * W * C;
  hipLaunchKernelGGL(f32_bn_bwd_reduce_kernel, F32_GRID(total), dout, act, y, y2, sums, sums2, mask_coef, total, H, W,
                     P, C);
  return iic_launch_status();
}

int iic_f32_bn_bwd_apply(const float* dout, const float* act, const float* y, const float* bcoef, float* dy,
                         const float* y2, const float* bcoef2, float* dy2, const float* mask_coef, int N, int H,
                         int W, int P, int C, void* stream) {
  if (!dout || !y || !bcoef || !dy || N <= 0) return IIC_ERR_ARG;
  if ((y2 == nullptr) != (bcoef2 == nullptr) || (y2 == nullptr) != (dy2 == nullptr)) return IIC_ERR_ARG;
  const long total = (long)N * H * W * C;
  hipLaunchKernelGGL(f32_bn_bwd_apply_kernel, F32_GRID(total), dout, act, y, bcoef, dy, y2, bcoef2, dy2, mask_coef,
                     total, H, W, P, C);
  return iic_launch_status();
}

int iic_f32_avgpool_fwd(const float* in_pt, float* feats, int N, int H, int W, int P, int C, void* stream) {
  if (!in_pt || !feats || N <= 0) return IIC_ERR_ARG;
  hipLaunchKernelGGL(f32_avgpool_fwd_kernel, F32_GRID((long)N * C), in_pt, feats, N, H, W, P, C);
  return iic_launch_status();
}

int iic_f32_avgpool_bwd(const float* dfeats, float* din_pt, int N, int H, int W, int P, int C,
                        const float* mask_act_pt, void* stream) {
  if (!dfeats || !din_pt || N <= 0) return IIC_ERR_ARG;
  const long total = (long)N * H * W * C;
  hipLaunchKernelGGL(f32_avgpool_bwd_kernel, F32_GRID(total), dfeats, din_pt, mask_act_pt, total, H, W, P, C);
  return iic_launch_status();
}

int iic_f32_maxpool_s2p1_fwd(const float* in_pt, float* out_pt, int N, int H, int W, int C, void* stream) {
  if (!in_pt || !out_pt || N <= 0) return IIC_ERR_ARG;
  const long total = (long)N * (H / 2 + 1) * (W / 2 + 1) * C;
  hipLaunchKernelGGL(f32_maxpool_fwd_kernel, F32_GRID(total), in_pt, out_pt, total, H, W, C);
  return iic_launch_status();
}

int iic_f32_maxpool_s2p1_bwd(const float* in_pt, const float* dout_pt, float* din_pt, int N, int H, int W, int C,
                             void* stream) {
  if (!in_pt || !dout_pt || !din_pt || N <= 0) return IIC_ERR_ARG;
  const long total = (long)N * H * W * C;
  hipLaunchKernelGGL(f32_maxpool_bwd_kernel, F32_GRID(total), in_pt, dout_pt, din_pt, total, H, W, C);
  return iic_launch_status();
}

int iic_f32_maxpool2_fwd(const float* in_pt, float* out_pt, int N, int H, int W, int Pi, int Po, int C,
                         void* stream) {
  if (!in_pt || !out_pt || N <= 0 || H < 2 || W < 2) return IIC_ERR_ARG;
  const long total = (long)N * (H / 2) * (W / 2) * C;
  hipLaunchKernelGGL(f32_maxpool2_fwd_kernel, F32_GRID(total), in_pt, out_pt, total, H, W, Pi, Po, C);
  return iic_launch_status();
}

int iic_f32_maxpool2_bwd(const float* in_pt, const float* dout_pt, float* din_pt, int N, int H, int W, int Pi,
                         int Po, int C, void* stream) {
  if (!in_pt || !dout_pt || !din_pt || N <= 0 || H < 2 || W < 2) return IIC_ERR_ARG;
  const long total = (long)N * H * W * C;
  hipLaunchKernelGGL(f32_maxpool2_bwd_kernel, F32_GRID(total), in_pt, dout_pt, din_pt, total, H, W, Pi, Po, C);
  return iic_launch_status();
}

int iic_f32_window_gather(const float* pt, float* out, int N, int Hw, int Ww, int Hp, int Wp, int off, int C,
                          void* stream) {
  if (!pt || !out || N <= 0 || off < 0 || Hw + off > Hp || Ww + off > Wp) return IIC_ERR_ARG;
  const long total = (long)N * Hw * Ww * C;
  hipLaunchKernelGGL(f32_window_gather_kernel, F32_GRID(total), pt, out, total, Hw, Ww, Hp, Wp, off, C);
  return iic_launch_status();
}

int iic_f32_window_scatter(const float* in, float* pt, int N, int Hw, int Ww, int Hp, int Wp, int off, int C,
                           void* stream) {
  if (!pt || !in || N <= 0 || off < 0 || Hw < 3 || Ww < 3) return IIC_ERR_ARG;
  const long total = (long)N * (Hw - 2) * (Ww - 2) * C;
  hipLaunchKernelGGL(f32_window_scatter_kernel, F32_GRID(total), in, pt, total, Hw, Ww, Hp, Wp, off, C);
  return iic_launch_status();
}

int iic_f32_nchw_to_pt(const float* x_nchw, float* out_pt, int N, int C, int H, int W, int P, void* stream) {
  if (!x_nchw || !out_pt || N <= 0) return IIC_ERR_ARG;
  const long total = (long)N * C * H * W;
  hipLaunchKernelGGL(f32_nchw_to_pt_kernel, F32_GRID(total), x_nchw, out_pt, total, C, H, W, P);
  return iic_launch_status();
}

}  // extern "C"

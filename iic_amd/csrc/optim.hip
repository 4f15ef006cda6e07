// Multi-tensor Adam for gfx950 -- replaces torch.optim.Adam as configured by
// /root/reference/code/utils/cluster/general.py:5-9 + cluster_sobel.py:149,272
// (betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad; torch's update formula).
// HBM-bound: 4 reads + 3 writes of 4 B per parameter; up to ADAM_CHUNK tensors per launch
// (pointer table in the kernel arguments), grid-stride inside each tensor.
#include "common.h"
#include "../../include/iic_hip.h"
#include <math.h>

#define ADAM_CHUNK 48
struct AdamTable {
  float* p[ADAM_CHUNK];
  const float* g[ADAM_CHUNK];
  const float* g2[ADAM_CHUNK];   // second gradient source (nullable): g_total = g + g2
  float* m[ADAM_CHUNK];
  float* v[ADAM_CHUNK];
  long n[ADAM_CHUNK];
};

__global__ __launch_bounds__(256) void adam_kernel(const AdamTable t, float beta1, float beta2,
                                                   float eps, float step_size, float inv_sqrt_bc2) {
  const int ti = blockIdx.y;
  float* __restrict__ p = t.p[ti];
  const float* __restrict__ g = t.g[ti];
  const float* __restrict__ g2 = t.g2[ti];
  float* __restrict__ m = t.m[ti];
  float* __restrict__ v = t.v[ti];
  const long n = t.n[ti];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = (g ? g[i] : 0.f) + (g2 ? g2[i] : 0.f);
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    p[i] -= step_size * (mi / denom);
  }
}

// Device-side step count: `steps_done` (int32 in HBM) holds the number of updates already applied
// to this group of tensors; the bias corrections are derived from it on the device (same double
// arithmetic as the host variant), so the launch arguments never change from step to step and
// the whole optimiser step can live in a captured HIP graph.  adam_count_kernel bumps the
// counter after the last chunk of the group.
__global__ __launch_bounds__(256) void adam_dev_kernel(const AdamTable t, float lr, float beta1,
                                                       float beta2, float eps,
                                                       const int* __restrict__ steps_done,
                                                       const float* __restrict__ lr_dev) {
  if (lr_dev) lr = lr_dev[0];       // learning rate kept on the device: update_lr() between graph replays
  const double step = (double)(steps_done[0] + 1);
  const double bc1 = 1.0 - pow((double)beta1, step);
  const double bc2 = 1.0 - pow((double)beta2, step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  const int ti = blockIdx.y;
  float* __restrict__ p = t.p[ti];
  const float* __restrict__ g = t.g[ti];
  const float* __restrict__ g2 = t.g2[ti];
  float* __restrict__ m = t.m[ti];
  float* __restrict__ v = t.v[ti];
  const long n = t.n[ti];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = (g ? g[i] : 0.f) + (g2 ? g2[i] : 0.f);
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    p[i] -= step_size * (mi / denom);
  }
}
__global__ void adam_count_kernel(int* steps_done) { steps_done[0] += 1; }

static void adam_fill_table(AdamTable& t, int base, int n, float* const* params,
                            const float* const* grads, const float* const* grads2, float* const* exp_avg,
                            float* const* exp_avg_sq, const long* numel, int* cnt_out, long* gx_out) {
  const int cnt = (n - base) < ADAM_CHUNK ? (n - base) : ADAM_CHUNK;
  long maxn = 0;
  for (int i = 0; i < ADAM_CHUNK; ++i) {
    const int j = i < cnt ? base + i : base;   // pad with a duplicate of n == 0 work
    t.p[i] = params[j]; t.g[i] = grads[j]; t.g2[i] = grads2 ? grads2[j] : nullptr;
    t.m[i] = exp_avg[j]; t.v[i] = exp_avg_sq[j];
    t.n[i] = i < cnt ? numel[j] : 0;
    if (t.n[i] > maxn) maxn = t.n[i];
  }
  long gx = (maxn + 255) / 256;
  if (gx > 256) gx = 256;
  if (gx < 1) gx = 1;
  *cnt_out = cnt;
  *gx_out = gx;
}

extern "C" int iic_adam_step_devlr(int n, float* const* params, const float* const* grads,
                                   const float* const* grads2, float* const* exp_avg, float* const* exp_avg_sq,
                                   const long* numel, const float* lr_dev, float lr, float beta1, float beta2,
                                   float eps, int* steps_done, void* stream) {
  if (n <= 0 || !params || !grads || !exp_avg || !exp_avg_sq || !numel || !steps_done)
    return IIC_ERR_ARG;
  for (int base = 0; base < n; base += ADAM_CHUNK) {
    AdamTable t;
    int cnt;
    long gx;
    adam_fill_table(t, base, n, params, grads, grads2, exp_avg, exp_avg_sq, numel, &cnt, &gx);
    hipLaunchKernelGGL(adam_dev_kernel, dim3((unsigned)gx, cnt), dim3(256), 0, (hipStream_t)stream,
                       t, lr, beta1, beta2, eps, (const int*)steps_done, lr_dev);
  }
  hipLaunchKernelGGL(adam_count_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, steps_done);
  return iic_launch_status();
}

extern "C" int iic_adam_step_dev(int n, float* const* params, const float* const* grads,
                                 const float* const* grads2, float* const* exp_avg, float* const* exp_avg_sq, const long* numel,
                                 float lr, float beta1, float beta2, float eps, int* steps_done,
                                 void* stream) {
  return iic_adam_step_devlr(n, params, grads, grads2, exp_avg, exp_avg_sq, numel, nullptr, lr, beta1, beta2, eps,
                             steps_done, stream);
}

extern "C" int iic_adam_step(int n, float* const* params, const float* const* grads,
                             const float* const* grads2, float* const* exp_avg, float* const* exp_avg_sq, const long* numel,
                             float lr, float beta1, float beta2, float eps, int step, void* stream) {
  if (n <= 0 || !params || !grads || !exp_avg || !exp_avg_sq || !numel || step < 1)
    return IIC_ERR_ARG;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  for (int base = 0; base < n; base += ADAM_CHUNK) {
    AdamTable t;
    int cnt;
    long gx;
    adam_fill_table(t, base, n, params, grads, grads2, exp_avg, exp_avg_sq, numel, &cnt, &gx);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)gx, cnt), dim3(256), 0, (hipStream_t)stream, t,
                       beta1, beta2, eps, step_size, inv_sqrt_bc2);
  }
  return iic_launch_status();
}

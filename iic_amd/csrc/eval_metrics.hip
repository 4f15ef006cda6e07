// Device-side evaluation counts.
//
// Replaces the k_pred x k_gt masked sums of
//   /root/reference/code/utils/cluster/eval_metrics.py:18-24 (_original_match) and :42-46
//   (_hungarian_match): `int(((flat_preds == c1) * (flat_targets == c2)).sum())` per pair -- 1 400
//   tiny kernels each ending in a host sync at k = 140, gt_k = 10 -- by ONE contingency-matrix
//   kernel, and `int((preds == targets).sum())` of _acc (:69) by a counting kernel.
// Integer work: per-workgroup LDS histogram (k_pred * k_gt <= 16384 bins) merged with 64-bit
// global atomics; labels outside [0, k) match no pair, exactly like the reference's comparisons.
#include "common.h"
#include "../../include/iic_hip.h"

#define EV_MAXBINS 16384

__global__ __launch_bounds__(256) void contingency_kernel(const long long* __restrict__ preds,
                                                          const long long* __restrict__ targets,
                                                          long n, int kp, int kt,
                                                          unsigned long long* __restrict__ counts) {
  __shared__ unsigned int bins[EV_MAXBINS];
  const int nb = kp * kt;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) bins[i] = 0u;
  __syncthreads();
  // a workgroup handles < 2^32 samples (grid-stride over at most n / gridDim), 32-bit bins suffice
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long long p = preds[i], t = targets[i];
    if (p >= 0 && p < kp && t >= 0 && t < kt) atomicAdd(&bins[(int)p * kt + (int)t], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nb; i += blockDim.x)
    if (bins[i]) atomicAdd(&counts[i], (unsigned long long)bins[i]);
}

__global__ __launch_bounds__(256) void count_equal_kernel(const long long* __restrict__ a,
                                                          const long long* __restrict__ b, long n,
                                                          unsigned long long* __restrict__ out) {
  unsigned int c = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    c += (a[i] == b[i]) ? 1u : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, (unsigned long long)c);
}

extern "C" {

int iic_contingency(const long long* preds, const long long* targets, long n, int k_pred, int k_gt,
                    long long* counts, void* stream) {
  if (!preds || !targets || !counts || n < 0 || k_pred <= 0 || k_gt <= 0) return IIC_ERR_ARG;
  if ((long)k_pred * k_gt > EV_MAXBINS) return IIC_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (iic_zero_async(counts, sizeof(long long) * k_pred * k_gt, s) != IIC_OK) return IIC_ERR_LAUNCH;
  if (n == 0) return IIC_OK;
  long blocks = (n + 255) / 256;
  int grid = (int)(blocks < 1024 ? blocks : 1024);
  hipLaunchKernelGGL(contingency_kernel, dim3(grid), dim3(256), 0, s, preds, targets, n, k_pred, k_gt,
                     (unsigned long long*)counts);
  return iic_launch_status();
}

int iic_count_equal(const long long* a, const long long* b, long n, long long* count, void* stream) {
  if (!a || !b || !count || n < 0) return IIC_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (iic_zero_async(count, sizeof(long long), s) != IIC_OK) return IIC_ERR_LAUNCH;
  if (n == 0) return IIC_OK;
  long blocks = (n + 255) / 256;
  int grid = (int)(blocks < 1024 ? blocks : 1024);
  hipLaunchKernelGGL(count_equal_kernel, dim3(grid), dim3(256), 0, s, a, b, n, (unsigned long long*)count);
  return iic_launch_status();
}

}  // extern "C"

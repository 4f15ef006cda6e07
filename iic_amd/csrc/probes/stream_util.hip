// Measurement helper (not part of the operator API, include/iic_hip.h): HIP streams restricted to a subset of the
// compute units, for the round-4 experiment "the two views of a step on disjoint halves of the chip instead of
// time-sharing all 256 CUs" (iic_amd/graph.py: IIC_PAIR_CUMASK, tools/ab_bench.sh).  torch wraps the handle with
// torch.cuda.ExternalStream; the stream is created by the same HIP runtime instance the kernels are launched with.
#include "../common.h"
#include <hip/hip_ext.h>

extern "C" {

// mask: `words` 32-bit words, bit i = CU i of the logical numbering (dealt round-robin over the 8 XCDs on MI355X).
void* iic_debug_stream_create_cumask(const uint32_t* mask, int words) {
  hipStream_t s = nullptr;
  if (!mask || words < 1) return nullptr;
  if (hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return (void*)s;
}

// "Matrix token" (round-4 experiment; its Python hook around the conv / weight-gradient launches lived in commit 088dc05): a device-side lock taken by a one-wave kernel in front of
// every matrix-bound launch and dropped by another behind it, so that the two views' streams never run two matrix-bound
// kernels at once and fall into anti-phase (one view's convolution beside the other view's BatchNorm passes).  The wait is
// bounded (timeout_us): a lost token degrades to the unsynchronised schedule, it cannot hang the queue.
static int* g_token = nullptr;

__global__ void token_acquire_kernel(int* lock, long long timeout_ticks) {
  if (threadIdx.x != 0) return;
  const long long t0 = wall_clock64();            // 100 MHz
  while (atomicCAS(lock, 0, 1) != 0) {
    __builtin_amdgcn_s_sleep(64);
    if (wall_clock64() - t0 > timeout_ticks) break;
  }
}
__global__ void token_release_kernel(int* lock) {
  if (threadIdx.x == 0) atomicExch(lock, 0);
}

int iic_debug_token_init() {
  if (!g_token) {
    if (hipMalloc(&g_token, 256) != hipSuccess) { g_token = nullptr; return 0; }
    (void)hipMemset(g_token, 0, 256);
    (void)hipDeviceSynchronize();
  }
  return 1;
}
int iic_debug_token_acquire(void* stream, int timeout_us) {
  if (!g_token) return IIC_ERR_ARG;
  hipLaunchKernelGGL(token_acquire_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, g_token, (long long)timeout_us * 100);
  return iic_launch_status();
}
int iic_debug_token_release(void* stream) {
  if (!g_token) return IIC_ERR_ARG;
  hipLaunchKernelGGL(token_release_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, g_token);
  return iic_launch_status();
}

int iic_debug_stream_destroy(void* s) {
  return s && hipStreamDestroy((hipStream_t)s) == hipSuccess ? IIC_OK : IIC_ERR_ARG;
}

}  // extern "C"

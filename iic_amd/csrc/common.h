// Shared device helpers for the gfx950 (MI355X / CDNA4) IIC kernels.
// Wave = 64 lanes; MFMA fragment maps follow /opt/skills/guides/cdna_hip_programming.md §3.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define IIC_OK 0
#define IIC_ERR_ARG (-1)
#define IIC_ERR_LAUNCH (-2)
#define IIC_ERR_UNSUPPORTED (-3)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

typedef uint16_t bf16_t;  // storage type for bf16 in HBM / LDS

static inline int iic_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? IIC_OK : IIC_ERR_LAUNCH;
}

// Zero-fill as a KERNEL node rather than hipMemsetAsync: inside a captured HIP graph, ROCm 7.0's
// packet-capture replay did not keep a small memset node ordered before the kernel that follows
// it (measured: the stem's 64-float accumulator kept its previous contents from the second replay
// on -- tools/graph_debug2.py); a kernel node is ordered like every other launch.
static __global__ void iic_zero_words_kernel(uint32_t* __restrict__ p, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    p[i] = 0u;
}
static inline int iic_zero_async(void* p, size_t bytes, hipStream_t s) {
  if (bytes & 3) return IIC_ERR_ARG;
  const long n = (long)(bytes >> 2);
  long g = (n + 255) / 256;
  if (g > 1024) g = 1024;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(iic_zero_words_kernel, dim3((unsigned)g), dim3(256), 0, s, (uint32_t*)p, n);
  return IIC_OK;
}

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// round-to-nearest-even, NaN stays NaN (matches torch's float->bfloat16): gfx950 converts in
// hardware (v_cvt_pk_bf16_f32, one instruction per PAIR instead of ~6 VALU per value).
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  return __builtin_bit_cast(bf16_t, (__bf16)f);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf16lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// Accumulator row of register r for the 32x32 MFMA C/D map (col = lane & 31).
__device__ __forceinline__ int mfma32_row(int r, int lane) {
  return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide sum of a double; `red` is LDS scratch of >= 32 doubles. All threads get the result.
__device__ __forceinline__ double block_sum_d(double v, double* red) {
  v = wave_sum_d(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

// Shared device helpers for the gfx950 (MI355X / CDNA4) IIC kernels.
// Wave = 64 lanes; MFMA fragment maps follow /opt/skills/guides/cdna_hip_programming.md §3.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define IIC_OK 0
#define IIC_ERR_ARG (-1)
#define IIC_ERR_LAUNCH (-2)
#define IIC_ERR_UNSUPPORTED (-3)

// ------------------------------------------------------------------------------------------
// Measurement switches.  The product library (libiic_hip.so) has NO mutable global state besides its
// kernel-attribute caches: every A/B switch below is a compile-time constant there and no iic_debug_*
// symbol exists.  `make dbg` builds the same sources with -DIIC_DEBUG_HOOKS into libiic_hip_dbg.so, where
// the switches are variables with exported setters (tests marked `hooks`, tools/*.py; IIC_HIP_LIB=dbg).
// ------------------------------------------------------------------------------------------
#ifdef IIC_DEBUG_HOOKS
#define IIC_HOOK extern "C" __attribute__((visibility("default")))
#define IIC_SWITCH(var, dflt, setter) \
  static int var = dflt;              \
  IIC_HOOK void setter(int v) { var = v; }
IIC_HOOK int iic_debug_get_ablate(void);
#else
#define IIC_SWITCH(var, dflt, setter) static constexpr int var = dflt;
static inline constexpr int iic_debug_get_ablate(void) { return 0; }
#endif

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

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

// ------------------------------------------------------------------------------------------
// Exact, order-independent accumulation of per-channel statistics (BatchNorm sums and their
// backward counterparts).  Workgroups used to atomicAdd float partial sums into 32 stripes: the
// result depended on the arrival order, so training was not reproducible run to run (and a
// 33-BatchNorm net amplifies one flipped ulp into visibly different losses).  Now every partial
// (a float, computed in a fixed order inside its workgroup) is added EXACTLY as a fixed-point
// integer: a cell is 7 int64 bins of 24 bits spacing (bin b holds multiples of 2^(LSB0 + 24 b));
// a 24-bit mantissa shifted by < 24 bits fits one bin with 15 bits of carry headroom, integer
// atomics are associative, and the finaliser folds stripes and bins in a fixed order.  Covers
// |x| from 2^-96 (smaller partials lose low bits, deterministically) to 2^72; non-finite or
// larger partials raise the cell's poison counter (bin 7) and the decoded sum is NaN, so a
// diverged run still shows up as NaN.  Layout: [stripe][C][2 stats][8] int64 -- one 128-byte
// line per (stripe, channel).
// ------------------------------------------------------------------------------------------
#define IIC_STAT_BINS 8
#define IIC_STAT_LSB0 (-96)
#define IIC_STAT_SPACING 24
typedef long long iic_stat_t;
__device__ __forceinline__ void iic_stat_add(float* stats, int stripe, int C, int c, int which, float t) {
  iic_stat_t* cell = reinterpret_cast<iic_stat_t*>(stats) + (((long)stripe * C + c) * 2 + which) * IIC_STAT_BINS;
  const uint32_t u = __float_as_uint(t);
  const int e = (int)((u >> 23) & 0xffu);
  int m = (int)(u & 0x7fffffu) | (e ? 0x800000 : 0);
  if (e == 0xff) { atomicAdd(reinterpret_cast<unsigned long long*>(cell + 7), 1ull); return; }
  int pos = (e ? e : 1) - 150 - IIC_STAT_LSB0;       // lsb of the mantissa, relative to bin 0's lsb
  if (pos < 0) { m = pos > -24 ? (m >> -pos) : 0; pos = 0; }
  if (m == 0) return;
  const int b = pos / IIC_STAT_SPACING, sh = pos - b * IIC_STAT_SPACING;
  if (b > 6) { atomicAdd(reinterpret_cast<unsigned long long*>(cell + 7), 1ull); return; }
  long long v = (long long)m << sh;
  if (u >> 31) v = -v;
  atomicAdd(reinterpret_cast<unsigned long long*>(cell + b), (unsigned long long)v);
}
// Finaliser side.  Called by 16 consecutive lanes per channel: lane16 = which * 8 + bin.  Folds the
// stripes (exact int64 sums), re-zeroes them, and returns in EVERY one of the 16 lanes the decoded
// sums of stat 0 and stat 1 (double, bins combined from the highest down: a fixed order).
__device__ __forceinline__ void iic_stat_collect(float* stats, int nstripes, int C, int c, int lane16,
                                                 double& v0, double& v1) {
  iic_stat_t* p = reinterpret_cast<iic_stat_t*>(stats) + ((long)c * 2) * IIC_STAT_BINS + lane16;
  long long S = 0;
  for (int st = 0; st < nstripes; ++st) {
    iic_stat_t* q = p + (long)st * C * 2 * IIC_STAT_BINS;
    S += *q;
    *q = 0;
  }
  const int bin = lane16 & 7;
  const double term = bin < 7 ? ldexp((double)S, IIC_STAT_LSB0 + IIC_STAT_SPACING * bin) : 0.0;
  const int lane = threadIdx.x & 63, base = lane & ~15;
  double a0 = 0.0, a1 = 0.0;
#pragma unroll
  for (int b = 6; b >= 0; --b) {
    a0 += __shfl(term, base + b, 64);
    a1 += __shfl(term, base + 8 + b, 64);
  }
  const long long p0 = __shfl(S, base + 7, 64), p1 = __shfl(S, base + 15, 64);
  v0 = p0 ? __builtin_nan("") : a0;
  v1 = p1 ? __builtin_nan("") : a1;
}

// floor(n / d) for 0 <= n < 2^31 as a multiply-shift (64-bit product); the pair is made on the host
struct iic_mdiv {
  unsigned mul;
  int sh;
};
static inline iic_mdiv iic_make_mdiv(int d) {
  iic_mdiv r;
  int l = 0;
  while ((1L << l) < d) ++l;
  r.sh = 31 + l;
  r.mul = (unsigned)(((1ULL << r.sh) + (unsigned long long)d - 1) / (unsigned long long)d);
  if (d == 1) { r.mul = 1u << 31; r.sh = 31; }
  return r;
}
__device__ __forceinline__ int iic_mdivide(int n, const iic_mdiv& d) {
  return (int)(((unsigned long long)(unsigned)n * d.mul) >> d.sh);
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

// BatchNorm batch statistics of the ClusterNet5g stem WITHOUT a convolution pass (round 4).
//
// The stem's conv1 output is never materialised (stem.hip: every pass recomputes it), so its batch statistics
// (/root/reference/code/archs/cluster/net5g.py:21-24: conv1 -> bn1 in train mode) used to cost a whole recompute pass,
// stem_stats_kernel: 64 x K MACs per pixel on the fp32 matrix cores, 210 us per view at 660 x 96 x 96.  But conv1 is
// LINEAR in the K = Cin * 9 values of a pixel's zero-padded 3x3 patch p:  y_c = w_c . p, hence
//     sum_pixels y_c   = w_c . S,            S[k]    = sum_pixels p[k]
//     sum_pixels y_c^2 = w_c^T G w_c,        G[k][l] = sum_pixels p[k] p[l]      (the patches' Gram matrix)
// and S, G do not depend on the weights: K + K (K + 1) / 2 = 189 sums for Cin = 2 instead of 64 x 18 MACs per pixel -- a
// VALU pass at the speed the input arrives.  stem_gram_kernel accumulates them (per-thread fp32 partials over its pixels,
// block sums added EXACTLY into the fixed-point cells of common.h, so the result is independent of the block order);
// stem_gram_finalize_kernel decodes them and evaluates mean / variance per channel in double, then does exactly what
// bn_finalize_kernel does with them (coefficients, running statistics with the unbiased variance).
//
// Same mathematical quantity as before, different rounding: the old pass summed fp32 conv outputs (each an fmaf chain),
// this one combines fp32 patch-product partials in double; both sit ~1e-6 relative from the real-number value
// (tests/test_gpu_kernels.py::test_stem_gram_statistics_match_the_convolution_pass).  Cin <= 2 (the Gram of Cin = 5
// would be 1 080 accumulators per thread); other inputs keep stem_stats_kernel.
#include "common.h"
#include "../../include/iic_hip.h"

#include "stem_common.h"

#define SG_RB 8            // conv rows per tile (the LDS band holds SG_RB + 2 input rows)
#define SG_THREADS 256

template <int CIN> struct StemGram {
  static constexpr int K = CIN * 9;
  static constexpr int NG = K * (K + 1) / 2;
  static constexpr int NV = NG + K;             // Gram (upper triangle, row-major) then the K patch sums
  static constexpr int NC = (NV + 1) / 2;       // statistic "channels": value v sits in cell (v >> 1, v & 1)
};

template <int CIN>
__global__ __launch_bounds__(SG_THREADS) void stem_gram_kernel(const float* __restrict__ x, float* __restrict__ gstats,
                                                               int N, int H, int W) {
  constexpr int K = StemGram<CIN>::K, NG = StemGram<CIN>::NG, NV = StemGram<CIN>::NV, NC = StemGram<CIN>::NC;
  extern __shared__ float sg_band[];            // [SG_RB + 2][CIN][W + 2], zero-padded
  __shared__ float s_red[SG_THREADS / 64][NV];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Wp = W + 2;
  const int bands = (H + SG_RB - 1) / SG_RB;
  const long tiles = (long)N * bands;
  float acc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) acc[v] = 0.f;
  for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int n = (int)(t / bands), b = (int)(t - (long)n * bands);
    const int y0 = b * SG_RB;
    const int rows = min(SG_RB, H - y0);
    const float* xin = x + (long)n * CIN * H * W;
    __syncthreads();                            // the previous tile's readers are done
    for (int idx = tid; idx < (rows + 2) * CIN * Wp; idx += SG_THREADS) {
      const int r = idx / (CIN * Wp), rem = idx - r * (CIN * Wp);
      const int c = rem / Wp, xx = rem - c * Wp;
      const int yy = y0 - 1 + r, xi = xx - 1;
      const bool ok = yy >= 0 && yy < H && xi >= 0 && xi < W;
      const float v = xin[ok ? ((long)c * H + yy) * W + xi : 0];
      sg_band[idx] = ok ? v : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < rows * W; i += SG_THREADS) {
      const int ry = i / W, cx = i - ry * W;
      float p[K];
#pragma unroll
      for (int c = 0; c < CIN; ++c)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) p[c * 9 + kh * 3 + kw] = sg_band[((ry + kh) * CIN + c) * Wp + cx + kw];
      int idx = 0;
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int l = k; l < K; ++l) {
          acc[idx] = fmaf(p[k], p[l], acc[idx]);
          ++idx;
        }
#pragma unroll
      for (int k = 0; k < K; ++k) acc[NG + k] += p[k];
    }
  }
  // block sum of the NV values (fixed order: lanes by butterfly, then waves 0..3), exact accumulation across blocks
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const float s = wave_sum(acc[v]);
    if (lane == 0) s_red[wave][v] = s;
  }
  __syncthreads();
  const int stripe = blockIdx.x % IIC_STAT_STRIPES;
  for (int v = tid; v < NV; v += SG_THREADS) {
    float s = 0.f;
#pragma unroll
    for (int wv = 0; wv < SG_THREADS / 64; ++wv) s += s_red[wv][v];
    iic_stat_add(gstats, stripe, NC, v >> 1, v & 1, s);
  }
}

// One block.  Thread v < NV folds value v over the stripes (exact int64 sums per bin, cells re-zeroed) and decodes it;
// thread c < 64 then evaluates channel c.  coef layout as bn_finalize_kernel: scale, shift, mean, invstd, unbiased variance.
template <int CIN>
__global__ __launch_bounds__(SG_THREADS) void stem_gram_finalize_kernel(
    float* __restrict__ gstats, const float* __restrict__ w, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var,
    long long* __restrict__ nbt, float* __restrict__ coef, long count, long ucount, float eps, float momentum) {
  constexpr int K = StemGram<CIN>::K, NG = StemGram<CIN>::NG, NV = StemGram<CIN>::NV, NC = StemGram<CIN>::NC;
  __shared__ double sv[NV];
  const int tid = threadIdx.x;
  if (tid == 0 && nbt) *nbt += 1;
  for (int v = tid; v < NV; v += SG_THREADS) {
    iic_stat_t* cell = reinterpret_cast<iic_stat_t*>(gstats) + ((long)(v >> 1) * 2 + (v & 1)) * IIC_STAT_BINS;
    long long S[IIC_STAT_BINS];
#pragma unroll
    for (int b = 0; b < IIC_STAT_BINS; ++b) S[b] = 0;
    for (int st = 0; st < IIC_STAT_STRIPES; ++st) {
      iic_stat_t* q = cell + (long)st * NC * 2 * IIC_STAT_BINS;
#pragma unroll
      for (int b = 0; b < IIC_STAT_BINS; ++b) {
        S[b] += q[b];
        q[b] = 0;
      }
    }
    double a = 0.0;
#pragma unroll
    for (int b = IIC_STAT_BINS - 2; b >= 0; --b) a += ldexp((double)S[b], IIC_STAT_LSB0 + IIC_STAT_SPACING * b);
    sv[v] = S[IIC_STAT_BINS - 1] ? __builtin_nan("") : a;     // poison counter: a non-finite partial was added
  }
  __syncthreads();
  if (tid >= STEM_CO) return;
  const int c = tid;
  double wk[K];
#pragma unroll
  for (int k = 0; k < K; ++k) wk[k] = (double)w[c * K + k];
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int k = 0; k < K; ++k) s1 += wk[k] * sv[NG + k];
  int idx = 0;
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int l = k; l < K; ++l) {
      s2 += (l == k ? 1.0 : 2.0) * wk[k] * wk[l] * sv[idx];
      ++idx;
    }
  const double m = s1 / (double)count;
  double v = s2 / (double)count - m * m;
  if (v < 0.0) v = 0.0;
  const float mean = (float)m, var = (float)v;
  const double unb = ucount > 1 ? v * (double)ucount / (double)(ucount - 1) : v;
  const float unbf = (float)unb;
  if (running_mean) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbf;
  }
  const float invstd = rsqrtf(var + eps);
  const float sc = gamma[c] * invstd;
  coef[c] = sc;
  coef[STEM_CO + c] = beta[c] - mean * sc;
  coef[2 * STEM_CO + c] = mean;
  coef[3 * STEM_CO + c] = invstd;
  coef[4 * STEM_CO + c] = unbf;
}

static long sg_lds_bytes(int Cin, int W) { return (long)(SG_RB + 2) * Cin * (W + 2) * (long)sizeof(float); }

extern "C" {

/* 1 if the stem's batch statistics of this input can be taken from the patch Gram matrix (iic_stem_gram). */
int iic_stem_gram_supported(int Cin, int H, int W) {
  return (Cin == 1 || Cin == 2) && H >= 1 && W >= 1 && sg_lds_bytes(Cin, W) <= 48 * 1024;
}

/* Size of the accumulator iic_stem_gram adds into (exact fixed-point cells, zero it once: the finaliser re-zeroes it). */
long iic_stem_gram_bytes(int Cin) {
  if (Cin == 1) return iic_stat_bytes(StemGram<1>::NC);
  if (Cin == 2) return iic_stat_bytes(StemGram<2>::NC);
  return 0;
}

int iic_stem_gram(const float* x, float* gstats, int N, int Cin, int H, int W, void* stream) {
  if (!x || !gstats || N < 1) return IIC_ERR_ARG;
  if (!iic_stem_gram_supported(Cin, H, W)) return IIC_ERR_UNSUPPORTED;
  const long tiles = (long)N * ((H + SG_RB - 1) / SG_RB);
  const int grid = (int)(tiles < STEM_PERSIST_BLOCKS ? tiles : STEM_PERSIST_BLOCKS);
  const long lds = sg_lds_bytes(Cin, W);
  hipStream_t s = (hipStream_t)stream;
  if (Cin == 1) hipLaunchKernelGGL(stem_gram_kernel<1>, dim3(grid), dim3(SG_THREADS), lds, s, x, gstats, N, H, W);
  else hipLaunchKernelGGL(stem_gram_kernel<2>, dim3(grid), dim3(SG_THREADS), lds, s, x, gstats, N, H, W);
  return iic_launch_status();
}

/* Train-mode iic_bn_finalize of the stem's BatchNorm from the Gram accumulator and the conv weights w [64][Cin*9]
 * (fp32 OIHW): coef [5][64]; running_mean / running_var / num_batches_tracked may be NULL (updated later through
 * iic_bn_running_update, or not tracked). */
int iic_stem_gram_finalize(float* gstats, const float* w, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, long long* num_batches_tracked, float* coef,
                           int Cin, long count, long ucount, float eps, float momentum, void* stream) {
  if (!gstats || !w || !gamma || !beta || !coef || count < 1 || (running_mean == nullptr) != (running_var == nullptr))
    return IIC_ERR_ARG;
  if (Cin != 1 && Cin != 2) return IIC_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (Cin == 1)
    hipLaunchKernelGGL(stem_gram_finalize_kernel<1>, dim3(1), dim3(SG_THREADS), 0, s, gstats, w, gamma, beta, running_mean,
                       running_var, num_batches_tracked, coef, count, ucount, eps, momentum);
  else
    hipLaunchKernelGGL(stem_gram_finalize_kernel<2>, dim3(1), dim3(SG_THREADS), 0, s, gstats, w, gamma, beta, running_mean,
                       running_var, num_batches_tracked, coef, count, ucount, eps, momentum);
  return iic_launch_status();
}

}  // extern "C"

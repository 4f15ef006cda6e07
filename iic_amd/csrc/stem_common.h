// Shared pieces of the ClusterNet5g stem kernels (stem.hip, stem_bwd2.hip).
#pragma once
#include "common.h"
#include "../../include/iic_hip.h"

#define STEM_CO 64
#define STEM_PERSIST_BLOCKS 1024

template <int CIN> struct StemK {
  static constexpr int K = CIN * 9;
  static constexpr int KS = (K + 1) / 2;       // MFMA k-steps (2 k per step)
  static constexpr int NKT = (K + 31) / 32;    // 32-wide column tiles of the dW GEMM
};

template <int CIN>
__device__ __forceinline__ void stem_load_w(const float* __restrict__ w, int lane,
                                            float (&wr)[2][StemK<CIN>::KS]) {
  constexpr int K = StemK<CIN>::K;
  const int j = lane & 31, kk = lane >> 5;
#pragma unroll
  for (int s = 0; s < StemK<CIN>::KS; ++s) {
    const int k = 2 * s + kk;
#pragma unroll
    for (int h = 0; h < 2; ++h) wr[h][s] = k < K ? w[(j + 32 * h) * K + k] : 0.f;
  }
}

// Routing of the pooled gradient to the conv grid for one (window, channel): returns the
// LDS index (0..3 -> (rs,cs)) of the arg-max of relu(bn(y)) in scan order (first max wins,
// as torch's max_pool2d), or -1 when the max is not positive (ReLU kills the gradient).
__device__ __forceinline__ int window_argmax(const float yv[4], const bool valid[4], float sc,
                                             float sh) {
  float best = -1.f;
  int bi = -1;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (!valid[q]) continue;
    const float a = fmaxf(yv[q] * sc + sh, 0.f);   // fp32 compare, like the fp32 reference
    if (a > best) { best = a; bi = q; }
  }
  return best > 0.f ? bi : -1;
}


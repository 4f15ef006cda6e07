// Convolution backward-weight for gfx950 on bf16 MFMA (fp32 accumulate).
//
//   dW[t][co][ci] = sum_m dY[pout(m)][co] * X[pin(m) + tap_off[t]][ci]
//
// Replaces cuDNN's bwd-filter behind nn.Conv2d in
//   /root/reference/code/archs/cluster/residual.py:4-7,19,22,54-55.
//
// GEMM view: M = co, N = ci, K = output pixels m.  The contraction runs over PIXELS, but the
// activations are pixel-major/channel-minor (PT layout), i.e. K is the strided dimension of
// both operands.  The MFMA wants 8 consecutive k per lane, so the [pixel][channel] LDS tiles
// are read with gfx950's transposing LDS load ds_read_b64_tr_b16: every lane supplies the
// address of a 4-channel (8-byte) row segment, a 16-lane group covers a [4 pixel][16 channel]
// block and each lane receives one channel's 4 pixels.  Pixel rows are addressed individually,
// so the tap shift (pin(m) + tap_off) is just a different row address into the SAME input
// patch the forward kernel stages -- all taps are produced from one staged patch per K-tile.
// A scalar-gather variant (USE_TR = false) with identical k-slot labelling exists as a
// semantic cross-check of the transpose-load mapping.
//
// workgroup = 64 (co) x 64 (ci) x all taps, 4 waves as 2 x 2, wave = 32 x 32 x ntaps
// accumulators (<= 9 taps per pass: 144 fp32 registers); K-tiles of 128 pixels are split
// across workgroups, partial sums go to [split][tap][co][ci] fp32 and are reduced (and
// transposed to the OIHW parameter layout) by conv_wgrad_reduce_kernel.
#include "common.h"
#include "../../include/iic_hip.h"

#define BM 128
#define NTHREADS 256
#define NTP 9   // taps per pass

__device__ __forceinline__ int swz(int row, int chunk) { return row * 8 + (chunk ^ ((row >> 1) & 7)); }

typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;

template <bool USE_TR>
__device__ __forceinline__ bf16x8 read_frag_T(const unsigned char* tile, const int* s_pin, int p_lo,
                                              bool via_lp, int toff, int ks, int colhalf, int lane) {
  // Returns, for MFMA lane (i = lane&31, g = lane>>5), the 8 values
  //   tile[row(ks*16 + 8g + j)][colhalf*32 + i],  j = 0..7
  // where row(k) = k (via_lp == false) or s_pin[k] - p_lo + toff (input patch).
  union { bf16x8 v; s16x4 h[2]; uint16_t e[8]; } u;
  if (USE_TR) {
    const int q = lane >> 4, i16 = lane & 15;
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int k = ks * 16 + 8 * (q >> 1) + 4 * rd + (i16 >> 2);
      const int row = via_lp ? (s_pin[k] - p_lo + toff) : k;
      const int chunk = colhalf * 4 + 2 * (q & 1) + ((i16 & 3) >> 1);
      const unsigned char* a = tile + swz(row, chunk) * 16 + (i16 & 1) * 8;
      u.h[rd] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)a);
    }
  } else {
    const int i = lane & 31, g5 = lane >> 5;
    const int col = colhalf * 32 + i;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = ks * 16 + 8 * g5 + j;
      const int row = via_lp ? (s_pin[k] - p_lo + toff) : k;
      u.e[j] = reinterpret_cast<const uint16_t*>(tile)[swz(row, col >> 3) * 8 + (col & 7)];
    }
  }
  return u.v;
}

template <bool USE_TR>
__global__ __launch_bounds__(NTHREADS, 2) void conv_wgrad_kernel(
    const iic_conv_geom g, const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
    float* __restrict__ partials, int nsplit, int num_ktiles, int lds_x_bytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* sX = smem_raw;                    // [NP][128 B]   input patch, this WG's 64 ci
  unsigned char* sD = smem_raw + lds_x_bytes;      // [BM][128 B]   dY rows, this WG's 64 co
  int* s_pin = reinterpret_cast<int*>(sD + BM * 128);
  int* s_pout = s_pin + BM;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31;
  const int ncit = g.Cin >> 6;
  const int cot = blockIdx.x / ncit, cit = blockIdx.x - cot * ncit;
  const int co0 = cot * 64, ci0 = cit * 64;
  const int split = blockIdx.y;
  const int per = (num_ktiles + nsplit - 1) / nsplit;
  const int kt0 = split * per;
  const int kt1 = min(num_ktiles, kt0 + per);
  const long M = (long)g.N * g.MY * g.MX;
  const long in_pixels = (long)g.N * g.in_Hp * g.in_Wp;
  const int np8 = g.NP * 8;
  uint4* sX4 = reinterpret_cast<uint4*>(sX);
  uint4* sD4 = reinterpret_cast<uint4*>(sD);

  for (int t0 = 0; t0 < g.ntaps; t0 += NTP) {
    const int tcount = min(NTP, g.ntaps - t0);
    f32x16 acc[NTP];
#pragma unroll
    for (int t = 0; t < NTP; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    for (int kt = kt0; kt < kt1; ++kt) {
      const long m0 = (long)kt * BM;
      __syncthreads();   // previous K-tile fully consumed
      if (tid < BM) {
        long m = m0 + tid;
        const bool valid = m < M;
        if (!valid) m = M - 1;
        const int plane = g.MY * g.MX;
        const int n = (int)(m / plane);
        const int r = (int)(m - (long)n * plane);
        const int y = r / g.MX, xx = r - y * g.MX;
        s_pin[tid] = (n * g.in_Hp + y * g.sy + g.oy) * g.in_Wp + xx * g.sx + g.ox;
        s_pout[tid] = valid ? (n * g.out_Hp + y * g.ty + g.py) * g.out_Wp + xx * g.tx + g.px : -1;
      }
      __syncthreads();
      const int p_lo = s_pin[0];
      // input patch (64 channels of this ci tile)
      for (int base = 0; base < np8; base += NTHREADS * 4) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = base + u * NTHREADS + tid;
          const long p = (long)p_lo + (idx >> 3);
          v[u] = make_uint4(0, 0, 0, 0);
          if (idx < np8 && p < in_pixels)
            v[u] = *reinterpret_cast<const uint4*>(x + (p * g.Cin + ci0 + (idx & 7) * 8));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = base + u * NTHREADS + tid;
          if (idx < np8) sX4[swz(idx >> 3, idx & 7)] = v[u];
        }
      }
      // dY rows (64 channels of this co tile); rows past M are zero => contribute nothing
      {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = u * NTHREADS + tid;      // BM*8 = 1024 = 4*256
          const int po = s_pout[idx >> 3];
          v[u] = make_uint4(0, 0, 0, 0);
          if (po >= 0)
            v[u] = *reinterpret_cast<const uint4*>(dy + ((long)po * g.Cout + co0 + (idx & 7) * 8));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = u * NTHREADS + tid;
          sD4[swz(idx >> 3, idx & 7)] = v[u];
        }
      }
      __syncthreads();
#pragma unroll 1
      for (int ks = 0; ks < BM / 16; ++ks) {
        const bf16x8 a = read_frag_T<USE_TR>(sD, s_pin, p_lo, false, 0, ks, wm, lane);
#pragma unroll
        for (int t = 0; t < NTP; ++t) {
          if (t < tcount) {
            const bf16x8 b = read_frag_T<USE_TR>(sX, s_pin, p_lo, true, g.tap_off[t0 + t], ks, wn, lane);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
          }
        }
      }
    }
    // partial[split][t][co][ci]
#pragma unroll
    for (int t = 0; t < NTP; ++t) {
      if (t < tcount) {
        float* dst = partials + (((long)split * g.ntaps + (t0 + t)) * g.Cout + co0) * g.Cin + ci0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * 32 + mfma32_row(r, lane);
          const int col = wn * 32 + l31;
          dst[(long)row * g.Cin + col] = acc[t][r];
        }
      }
    }
  }
}

// dW[co][ci][t] (fp32 OIHW) (+)= sum_s partial[s][t][co][ci]
__global__ void conv_wgrad_reduce_kernel(const float* __restrict__ partials, int nsplit, int T,
                                         int Co, int Ci, float* __restrict__ dW, int accumulate) {
  const long per = (long)T * Co * Ci;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per;
       i += (long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) s += partials[(long)sp * per + i];
    const int ci = (int)(i % Ci);
    const long r = i / Ci;
    const int co = (int)(r % Co), t = (int)(r / Co);
    const long o = ((long)co * Ci + ci) * T + t;
    dW[o] = accumulate ? dW[o] + s : s;
  }
}

// Probe: what does ds_read_b64_tr_b16 return?  LDS holds u16 value = element index; lane l
// supplies byte address 8*l; out[l][0..3] = returned elements.  (tests/test_gpu_probe.py)
__global__ void probe_tr16_kernel(uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[64 * 4];
  const int l = threadIdx.x;
  for (int i = l; i < 256; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(&lds[l * 4]));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}

extern "C" {

int iic_conv_wgrad_nsplit(const iic_conv_geom* g) {
  const long M = (long)g->N * g->MY * g->MX;
  const int kt = (int)((M + BM - 1) / BM);
  const int tiles = (g->Cout / 64) * (g->Cin / 64);
  int ns = 512 / (tiles > 0 ? tiles : 1);
  if (ns < 1) ns = 1;
  if (ns > kt) ns = kt;
  return ns;
}

int iic_conv_wgrad(const iic_conv_geom* g, const void* x, const void* dy, float* partials,
                   int nsplit, int use_tr, void* stream) {
  if (!g || !x || !dy || !partials || nsplit < 1) return IIC_ERR_ARG;
  if (g->Cin % 64 != 0 || g->Cout % 64 != 0 || g->ntaps < 1 || g->ntaps > IIC_MAX_TAPS)
    return IIC_ERR_UNSUPPORTED;
  const long M = (long)g->N * g->MY * g->MX;
  const int kt = (int)((M + BM - 1) / BM);
  const int lx = (g->NP * 128 + 15) & ~15;
  const long lds = (long)lx + BM * 128 + 2 * BM * 4;
  if (lds > 160 * 1024) return IIC_ERR_UNSUPPORTED;
  dim3 grid((g->Cout / 64) * (g->Cin / 64), nsplit);
  hipStream_t s = (hipStream_t)stream;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<true>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<false>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  if (use_tr)
    hipLaunchKernelGGL(conv_wgrad_kernel<true>, grid, dim3(NTHREADS), lds, s, *g, (const bf16_t*)x,
                       (const bf16_t*)dy, partials, nsplit, kt, lx);
  else
    hipLaunchKernelGGL(conv_wgrad_kernel<false>, grid, dim3(NTHREADS), lds, s, *g,
                       (const bf16_t*)x, (const bf16_t*)dy, partials, nsplit, kt, lx);
  return iic_launch_status();
}

int iic_conv_wgrad_reduce(const float* partials, int nsplit, int T, int Cout, int Cin, float* dW,
                          int accumulate, void* stream) {
  if (!partials || !dW || nsplit < 1) return IIC_ERR_ARG;
  const long per = (long)T * Cout * Cin;
  int grid = (int)((per + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                     partials, nsplit, T, Cout, Cin, dW, accumulate);
  return iic_launch_status();
}

int iic_probe_tr16(void* out_u16_64x4, void* stream) {
  hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream,
                     (uint16_t*)out_u16_64x4);
  return iic_launch_status();
}

}  // extern "C"

// im2col-free implicit-GEMM convolution for gfx950: bf16 operands, fp32 accumulate on
// v_mfma_f32_32x32x16_bf16.  One kernel = forward conv AND backward-data (the geometry
// descriptor says which pixels a GEMM row reads / writes and which taps it uses).
//
// Replaces the cuDNN kernels behind nn.Conv2d in
//   /root/reference/code/archs/cluster/residual.py:4-7,19,22,54-55  (3x3 s1/s2, 1x1 s2)
//   /root/reference/code/archs/cluster/vgg.py:24-26                 (5x5, dilated 3x3)
//
// Data layout (HBM): activations are "PT" tensors, bf16 [N][H+2P][W+2P][C] with a zero
// border, so no tap ever needs a bounds check and a tap is a constant pixel offset.
// Weights arrive as bf16 [tap][Cout][Cin] (iic_weight_prep).
//
// Work decomposition: GEMM M = output pixels (flat n,y,x), N = Cout, K = taps x Cin.
//   workgroup = 128 (M) x BN (64|128) tile, 4 waves as 2(M) x 2(N), wave tile 64 x BN/2
//   per 64-channel chunk of Cin: the union of input pixels the tile's 128 rows touch over
//   ALL taps (a contiguous span of the padded-flat pixel index) is staged ONCE in LDS
//   ("patch", 128 B per pixel) and every tap reads it at a shifted pixel index -- so L2
//   sees each input byte ~span/128 times instead of ntaps times;
//   the [BN][64] weight slice of each (tap, chunk) is double-buffered in LDS.
// LDS rows are 128 B; 16-B chunks are XOR-swizzled with (row>>1)&7 so that the 16-lane
// groups of ds_read_b128 hit 16 distinct 16-B slots (conflict-free for unit-stride rows).
#include "common.h"
#include "../../include/iic_hip.h"

#define BM 128
#define NTHREADS 256

__device__ __forceinline__ int swz(int row, int chunk) { return row * 8 + (chunk ^ ((row >> 1) & 7)); }

template <int BPASS>
__device__ __forceinline__ void igemm_load_b(u32x4 (&breg)[BPASS], const iic_conv_geom& g,
                                             const bf16_t* __restrict__ w, int n0, int it, int tid) {
  const int chunk = it / g.ntaps, tap = it - chunk * g.ntaps;
  const bf16_t* wt = w + ((long)g.tap_w[tap] * g.Cout + n0) * g.Cin + chunk * 64;
#pragma unroll
  for (int u = 0; u < BPASS; ++u) {
    const int idx = u * NTHREADS + tid;
    breg[u] = *reinterpret_cast<const u32x4*>(wt + (long)(idx >> 3) * g.Cin + (idx & 7) * 8);
  }
}
template <int BPASS>
__device__ __forceinline__ void igemm_store_b(const u32x4 (&breg)[BPASS], u32x4* dst, int tid) {
#pragma unroll
  for (int u = 0; u < BPASS; ++u) {
    const int idx = u * NTHREADS + tid;
    dst[swz(idx >> 3, idx & 7)] = breg[u];
  }
}
__device__ __forceinline__ void igemm_load_patch(uint4* sA, const bf16_t* __restrict__ in, int Cin,
                                                 int c0, int p_lo, int np8, long in_pixels, int tid) {
  for (int base = 0; base < np8; base += NTHREADS * 4) {
    uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0, v2 = v0, v3 = v0;
    const int i0 = base + tid, i1 = i0 + NTHREADS, i2 = i1 + NTHREADS, i3 = i2 + NTHREADS;
    const long q0 = (long)p_lo + (i0 >> 3), q1 = (long)p_lo + (i1 >> 3);
    const long q2 = (long)p_lo + (i2 >> 3), q3 = (long)p_lo + (i3 >> 3);
    if (i0 < np8 && q0 < in_pixels) v0 = *reinterpret_cast<const uint4*>(in + (q0 * Cin + c0 + (i0 & 7) * 8));
    if (i1 < np8 && q1 < in_pixels) v1 = *reinterpret_cast<const uint4*>(in + (q1 * Cin + c0 + (i1 & 7) * 8));
    if (i2 < np8 && q2 < in_pixels) v2 = *reinterpret_cast<const uint4*>(in + (q2 * Cin + c0 + (i2 & 7) * 8));
    if (i3 < np8 && q3 < in_pixels) v3 = *reinterpret_cast<const uint4*>(in + (q3 * Cin + c0 + (i3 & 7) * 8));
    if (i0 < np8) sA[swz(i0 >> 3, i0 & 7)] = v0;
    if (i1 < np8) sA[swz(i1 >> 3, i1 & 7)] = v1;
    if (i2 < np8) sA[swz(i2 >> 3, i2 & 7)] = v2;
    if (i3 < np8) sA[swz(i3 >> 3, i3 & 7)] = v3;
  }
}

template <int BN>
__global__ __launch_bounds__(NTHREADS, 2) void conv_igemm_kernel(
    const iic_conv_geom g, const bf16_t* __restrict__ in, const bf16_t* __restrict__ w,
    bf16_t* __restrict__ out, float* __restrict__ stats, const bf16_t* __restrict__ res_grad,
    const bf16_t* __restrict__ res_act, int accumulate, int num_mtiles, int lds_a_bytes) {
  constexpr int NS = BN / 64;          // 32-wide N sub-tiles per wave
  constexpr int BPASS = BN * 8 / NTHREADS;  // uint4 per thread per weight tile
  constexpr int CLD = BN + 8;          // epilogue tile row stride (bf16 elements)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint4* sA = reinterpret_cast<uint4*>(smem_raw);
  uint4* sB = reinterpret_cast<uint4*>(smem_raw + lds_a_bytes);
  int* s_pin = reinterpret_cast<int*>(sB + 2 * BN * 8);
  int* s_pout = s_pin + BM;
  float* s_red = reinterpret_cast<float*>(s_pout + BM);   // [2(wm)][2][BN]
  bf16_t* sC = reinterpret_cast<bf16_t*>(smem_raw);        // epilogue reuse of sA

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, g5 = lane >> 5;

  // XCD-aware tile order: blocks b, b+8, b+16.. share an XCD (observed dispatch); give each
  // XCD a contiguous range of tiles so neighbouring M-tiles (shared halo, shared weights)
  // hit the same L2.  Bijective for any grid size.
  const int nt = g.Cout / BN;
  const int nwg = num_mtiles * nt;
  int tix;
  {
    const int b = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = b & 7, idx = b >> 3;
    tix = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int mtile = tix / nt, ntile = tix % nt;
  const int n0 = ntile * BN;
  const long M = (long)g.N * g.MY * g.MX;
  const long m0 = (long)mtile * BM;
  const long in_pixels = (long)g.N * g.in_Hp * g.in_Wp;

  // ---- per-row pixel indices --------------------------------------------------------
  if (tid < BM) {
    long m = m0 + tid;
    const bool valid = m < M;
    if (!valid) m = M - 1;
    const int plane = g.MY * g.MX;
    const int n = (int)(m / plane);
    const int r = (int)(m - (long)n * plane);
    const int y = r / g.MX, x = r - y * g.MX;
    s_pin[tid] = (n * g.in_Hp + y * g.sy + g.oy) * g.in_Wp + x * g.sx + g.ox;
    s_pout[tid] = valid ? (n * g.out_Hp + y * g.ty + g.py) * g.out_Wp + x * g.tx + g.px : -1;
  }
  __syncthreads();
  const int p_lo = s_pin[0];
  int lp[2];
#pragma unroll
  for (int ms = 0; ms < 2; ++ms) lp[ms] = s_pin[wm * 64 + ms * 32 + l31] - p_lo;

  f32x16 acc[2][NS];
#pragma unroll
  for (int ms = 0; ms < 2; ++ms)
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ms][ns][r] = 0.f;

  const int nchunks = g.Cin >> 6;
  const int NIT = nchunks * g.ntaps;
  const int np8 = g.NP * 8;

  // ---- prologue ------------------------------------------------------------------------
  u32x4 breg[BPASS];
  igemm_load_patch(sA, in, g.Cin, 0, p_lo, np8, in_pixels, tid);
  igemm_load_b<BPASS>(breg, g, w, n0, 0, tid);
  igemm_store_b<BPASS>(breg, reinterpret_cast<u32x4*>(sB), tid);
  __syncthreads();

  // ---- main loop over (chunk, tap) -------------------------------------------------------
  for (int it = 0; it < NIT; ++it) {
    const int chunk = it / g.ntaps, tap = it - chunk * g.ntaps;
    const bool has_next = it + 1 < NIT;
    if (has_next) igemm_load_b<BPASS>(breg, g, w, n0, it + 1, tid);   // in flight during the MFMAs below
    const int toff = g.tap_off[tap];
    const uint4* bB = sB + (it & 1) * BN * 8;
    const int pa0 = lp[0] + toff, pa1 = lp[1] + toff;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int ch = 2 * ks + g5;
      bf16x8 a[2], b[NS];
      a[0] = __builtin_bit_cast(bf16x8, sA[swz(pa0, ch)]);
      a[1] = __builtin_bit_cast(bf16x8, sA[swz(pa1, ch)]);
#pragma unroll
      for (int ns = 0; ns < NS; ++ns)
        b[ns] = __builtin_bit_cast(bf16x8, bB[swz(wn * (BN / 2) + ns * 32 + l31, ch)]);
#pragma unroll
      for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int ns = 0; ns < NS; ++ns)
          acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ms], b[ns], acc[ms][ns], 0, 0, 0);
    }
    if (has_next) {
      if (tap + 1 == g.ntaps) {        // next iteration starts a new channel chunk
        __syncthreads();               // everyone is done reading the patch
        igemm_load_patch(sA, in, g.Cin, (chunk + 1) * 64, p_lo, np8, in_pixels, tid);
      }
      igemm_store_b<BPASS>(breg, reinterpret_cast<u32x4*>(sB + ((it + 1) & 1) * BN * 8), tid);
      __syncthreads();
    }
  }

  // ---- epilogue ------------------------------------------------------------------------
  const bool tail = (m0 + BM > M);
  if (stats) {
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
      float s = 0.f, ss = 0.f;
#pragma unroll
      for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[ms][ns][r];
          if (tail && s_pout[wm * 64 + ms * 32 + mfma32_row(r, lane)] < 0) v = 0.f;
          s += v;
          ss += v * v;
        }
      s += __shfl_xor(s, 32, 64);
      ss += __shfl_xor(ss, 32, 64);
      if (lane < 32) {
        const int col = wn * (BN / 2) + ns * 32 + lane;
        s_red[(wm * 2 + 0) * BN + col] = s;
        s_red[(wm * 2 + 1) * BN + col] = ss;
      }
    }
  }
  __syncthreads();   // all waves finished reading sA/sB; s_red complete
  if (stats && tid < BN) {
    float* st = stats + (long)(blockIdx.x % IIC_STAT_STRIPES) * 2 * g.Cout;
    atomicAdd(st + n0 + tid, s_red[0 * BN + tid] + s_red[2 * BN + tid]);
    atomicAdd(st + g.Cout + n0 + tid, s_red[1 * BN + tid] + s_red[3 * BN + tid]);
  }
#pragma unroll
  for (int ms = 0; ms < 2; ++ms)
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + ms * 32 + mfma32_row(r, lane);
        const int col = wn * (BN / 2) + ns * 32 + l31;
        sC[row * CLD + col] = f32_to_bf16(acc[ms][ns][r]);
      }
  __syncthreads();
  for (int idx = tid; idx < BM * (BN / 8); idx += NTHREADS) {
    const int row = idx / (BN / 8), ch = idx - row * (BN / 8);
    const int po = s_pout[row];
    if (po < 0) continue;
    uint4 v = *reinterpret_cast<const uint4*>(sC + row * CLD + ch * 8);
    const long o = (long)po * g.Cout + n0 + ch * 8;
    if (accumulate || res_grad) {
      uint32_t vv[4] = {v.x, v.y, v.z, v.w};
      float f[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) { f[2 * i] = bf16lo(vv[i]); f[2 * i + 1] = bf16hi(vv[i]); }
      if (accumulate) {
        const uint4 ov = *reinterpret_cast<const uint4*>(out + o);
        const uint32_t oo[4] = {ov.x, ov.y, ov.z, ov.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { f[2 * i] += bf16lo(oo[i]); f[2 * i + 1] += bf16hi(oo[i]); }
      }
      if (res_grad) {
        const uint4 gv = *reinterpret_cast<const uint4*>(res_grad + o);
        const uint4 av = *reinterpret_cast<const uint4*>(res_act + o);
        const uint32_t gg[4] = {gv.x, gv.y, gv.z, gv.w}, aa[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (bf16lo(aa[i]) > 0.f) f[2 * i] += bf16lo(gg[i]);
          if (bf16hi(aa[i]) > 0.f) f[2 * i + 1] += bf16hi(gg[i]);
        }
      }
      v = make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]),
                     pack_bf16x2(f[6], f[7]));
    }
    *reinterpret_cast<uint4*>(out + o) = v;
  }
}

// fp32 OIHW -> bf16 [T][Co][Ci] and [T][Ci][Co]
__global__ void weight_prep_kernel(const float* __restrict__ w, bf16_t* __restrict__ wf,
                                   bf16_t* __restrict__ wb, int Co, int Ci, int T) {
  const long total = (long)T * Co * Ci;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    {  // wf[t][co][ci]
      const int ci = (int)(i % Ci);
      const long r = i / Ci;
      const int co = (int)(r % Co), t = (int)(r / Co);
      wf[i] = f32_to_bf16(w[((long)co * Ci + ci) * T + t]);
    }
    if (wb) {  // wb[t][ci][co]
      const int co = (int)(i % Co);
      const long r = i / Co;
      const int ci = (int)(r % Ci), t = (int)(r / Ci);
      wb[i] = f32_to_bf16(w[((long)co * Ci + ci) * T + t]);
    }
  }
}

static int pick_bn(int Cout) { return (Cout % 128 == 0) ? 128 : 64; }

static long lds_a_bytes_for(const iic_conv_geom* g, int BN) {
  long a = (long)g->NP * 128;
  long c = (long)BM * (BN + 8) * 2;
  long m = a > c ? a : c;
  return (m + 15) & ~15L;
}

extern "C" {

long iic_conv_lds_bytes(const iic_conv_geom* g, int BN) {
  if (BN == 0) BN = pick_bn(g->Cout);
  return lds_a_bytes_for(g, BN) + 2L * BN * 128 + 2L * BM * 4 + 4L * BN * 4;
}

int iic_conv_igemm(const iic_conv_geom* g, const void* in, const void* w, void* out, float* stats,
                   const void* res_grad, const void* res_act, int accumulate, void* stream) {
  if (!g || !in || !w || !out) return IIC_ERR_ARG;
  if (g->Cin % 64 != 0 || g->Cout % 64 != 0 || g->ntaps < 1 || g->ntaps > IIC_MAX_TAPS)
    return IIC_ERR_UNSUPPORTED;
  if ((res_grad == nullptr) != (res_act == nullptr)) return IIC_ERR_ARG;
  const int BN = pick_bn(g->Cout);
  const long M = (long)g->N * g->MY * g->MX;
  if (M <= 0 || g->NP <= 0) return IIC_ERR_ARG;
  const int mt = (int)((M + BM - 1) / BM);
  const int grid = mt * (g->Cout / BN);
  const long lds = iic_conv_lds_bytes(g, BN);
  if (lds > 160 * 1024) return IIC_ERR_UNSUPPORTED;
  const int la = (int)lds_a_bytes_for(g, BN);
  hipStream_t s = (hipStream_t)stream;
  if (BN == 128) {
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<128>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr = true;
    }
    hipLaunchKernelGGL(conv_igemm_kernel<128>, dim3(grid), dim3(NTHREADS), lds, s, *g,
                       (const bf16_t*)in, (const bf16_t*)w, (bf16_t*)out, stats,
                       (const bf16_t*)res_grad, (const bf16_t*)res_act, accumulate, mt, la);
  } else {
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<64>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr = true;
    }
    hipLaunchKernelGGL(conv_igemm_kernel<64>, dim3(grid), dim3(NTHREADS), lds, s, *g,
                       (const bf16_t*)in, (const bf16_t*)w, (bf16_t*)out, stats,
                       (const bf16_t*)res_grad, (const bf16_t*)res_act, accumulate, mt, la);
  }
  return iic_launch_status();
}

int iic_weight_prep(const float* w_oihw, void* w_fwd, void* w_bwd, int Cout, int Cin, int T,
                    void* stream) {
  if (!w_oihw || !w_fwd || Cout <= 0 || Cin <= 0 || T <= 0) return IIC_ERR_ARG;
  const long total = (long)T * Cout * Cin;
  int grid = (int)((total + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(weight_prep_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_oihw,
                     (bf16_t*)w_fwd, (bf16_t*)w_bwd, Cout, Cin, T);
  return iic_launch_status();
}

int iic_version(void) { return 1; }

}  // extern "C"

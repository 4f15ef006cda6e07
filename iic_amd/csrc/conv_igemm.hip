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
// LDS rows are 128 B of data at a 144-B pitch (see ROWB): conflict-free ds_read_b128 and
// immediate-offset k-steps.  1-tap convs (1x1 stride-2 downsample) skip the patch and gather
// the 128 rows' own pixels.
#include "common.h"
#include "conv_tile.h"
#include "../../include/iic_hip.h"

template <int BN, bool GATHER, bool ABL, int BM>
__global__ __launch_bounds__(BM * 2, BM == 128 ? 2 : 1) void conv_igemm_kernel(
    const iic_conv_geom g, const bf16_t* __restrict__ in, const bf16_t* __restrict__ w,
    bf16_t* __restrict__ out, float* __restrict__ stats, const bf16_t* __restrict__ res_grad,
    const bf16_t* __restrict__ res_act, int accumulate, int num_mtiles, int lds_a_bytes,
    int ablate_arg) {
  const int ablate = ABL ? ablate_arg : 0;
  constexpr int NTHREADS = BM * 2;          // 4 (BM=128) or 8 (BM=256) waves: (BM/64) x 2
  constexpr int WM = BM / 64;
  constexpr int NS = BN / 64;               // 32-wide N sub-tiles per wave
  constexpr int BPASS = BN * 8 / NTHREADS;  // 16-B pieces per thread per weight tile
  constexpr int CLD = BN + 8;               // epilogue tile row stride (bf16 elements)
  // 16-B patch pieces per thread prefetched in registers across a chunk (0 = reload in place;
  // the BN = 64 convs of this network have a single 64-channel chunk or small patches)
  constexpr int PATCH_PF_MAX = (BN == 128) ? 8 : 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* sA = smem_raw;                          // [NP][ROWB]
  unsigned char* sB = smem_raw + lds_a_bytes;            // [2][BN][ROWB]
  int* s_pin = reinterpret_cast<int*>(sB + 2 * BN * ROWB);
  int* s_pout = s_pin + BM;
  float* s_red = reinterpret_cast<float*>(s_pout + BM);   // [WM][2][BN]
  bf16_t* sC = reinterpret_cast<bf16_t*>(smem_raw);        // epilogue reuse of sA

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, g5 = lane >> 5;

  const int nt = g.Cout / BN;
  const int tix = xcd_tile_index(blockIdx.x, num_mtiles * nt);
  const int mtile = tix / nt, ntile = tix - mtile * nt;
  const int n0 = ntile * BN;
  const int m0 = mtile * BM;
  const int in_pixels = g.N * g.in_Hp * g.in_Wp;

  // tap tables live in VGPRs (lane t holds tap t) and are broadcast with v_readlane: the
  // main loop has no scalar-memory load on its critical path.
  const int v_tapoff = g.tap_off[lane & (IIC_MAX_TAPS - 1)];
  const int v_tapw = g.tap_w[lane & (IIC_MAX_TAPS - 1)];

  // ---- per-row pixel indices --------------------------------------------------------
  if (tid < BM) {
    int pin, pout;
    igemm_row_pixels(g, m0 + tid, pin, pout);
    s_pin[tid] = pin;
    s_pout[tid] = pout;
  }
  __syncthreads();
  const int p_lo = s_pin[0];
  const int npix = GATHER ? BM : (BM == 128 ? g.NP : g.NP256);
  // byte address (within sA) of this lane's A rows at tap offset 0, k-chunk g5
  int arow[2];
#pragma unroll
  for (int ms = 0; ms < 2; ++ms) {
    const int row = wm * 64 + ms * 32 + l31;
    arow[ms] = (GATHER ? row : (s_pin[row] - p_lo)) * ROWB + g5 * 16;
  }
  const int brow = (wn * (BN / 2) + l31) * ROWB + g5 * 16;   // within one weight buffer

  f32x16 acc[2][NS];
#pragma unroll
  for (int ms = 0; ms < 2; ++ms)
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ms][ns][r] = 0.f;

  const int nchunks = g.Cin >> 6;
  const int ntaps = g.ntaps;
  const int NIT = nchunks * ntaps;
  // per-thread weight-tile pieces: row co = tid>>3 (+32 per pass), 16-B piece tid&7
  const int wrow = tid >> 3, wpc = tid & 7;
  const bf16_t* wbase = w + ((long)n0 + wrow) * g.Cin + wpc * 8;
  const long wtap_stride = (long)g.Cout * g.Cin;
  const long wrow32 = (long)(NTHREADS / 8) * g.Cin;
  unsigned char* const sBw = sB + wrow * ROWB + wpc * 16;   // this thread's store slot, pass 0

  // weight tile of flat iteration (tap, chunk) -> registers
  auto bload = [&](u32x4(&R)[BPASS], int tap, int chunk) {
    const int tw = __builtin_amdgcn_readlane(v_tapw, tap);
    const bf16_t* wt = wbase + (long)tw * wtap_stride + chunk * 64;
#pragma unroll
    for (int u = 0; u < BPASS; ++u) R[u] = *reinterpret_cast<const u32x4*>(wt + u * wrow32);
  };
  auto bstore = [&](const u32x4(&R)[BPASS], int b) {
    unsigned char* dst = sBw + b * (BN * ROWB);
#pragma unroll
    for (int u = 0; u < BPASS; ++u) *reinterpret_cast<u32x4*>(dst + u * (NTHREADS / 8) * ROWB) = R[u];
  };
  // patch of the NEXT chunk held in registers while the current chunk computes
  const int n8 = npix * 8;
  const bool patch_pf = (BN == 128) && (n8 <= PATCH_PF_MAX * NTHREADS);
  u32x4 P[PATCH_PF_MAX];
  auto pload = [&](int c0) {
#pragma unroll
    for (int u = 0; u < PATCH_PF_MAX; ++u) {
      const int idx = u * NTHREADS + tid;
      P[u] = (u32x4){0u, 0u, 0u, 0u};
      if (idx < n8) {
        const int p = GATHER ? s_pin[idx >> 3] : p_lo + (idx >> 3);
        if (p < in_pixels)
          P[u] = *reinterpret_cast<const u32x4*>(in + ((long)p * g.Cin + c0 + (idx & 7) * 8));
      }
    }
  };
  auto pstore = [&]() {
#pragma unroll
    for (int u = 0; u < PATCH_PF_MAX; ++u) {
      const int idx = u * NTHREADS + tid;
      if (idx < n8) *reinterpret_cast<u32x4*>(sA + (idx >> 3) * ROWB + (idx & 7) * 16) = P[u];
    }
  };

  // ---- prologue ------------------------------------------------------------------------
  u32x4 R0[BPASS], R1[BPASS];
  int tap2 = 0, chunk2 = 0;                    // (tap, chunk) of flat iteration it + 2
  auto advance = [&](int& t, int& c) { if (++t == ntaps) { t = 0; ++c; } };
  if (!(ablate & 16))
    igemm_load_patch<GATHER, NTHREADS>(sA, in, g.Cin, 0, p_lo, npix, in_pixels, s_pin, tid);
  bload(R0, 0, 0);
  bstore(R0, 0);
  advance(tap2, chunk2);
  if (NIT > 1) bload(R1, tap2, chunk2);        // B(1) in flight
  advance(tap2, chunk2);
  __syncthreads();

  // ---- main loop: flat (chunk, tap) iterations, weight tiles prefetched TWO ahead ----------
  int tap = 0, chunk = 0;
  auto body = [&](u32x4(&Rnext)[BPASS], u32x4(&Rfree)[BPASS], int it, int buf) {
    // Rnext holds B(it+1) (loaded one iteration ago); Rfree is loaded with B(it+2) now.
    const bool last_tap = (tap + 1 == ntaps);
    if (it + 2 < NIT && !(ablate & 2)) bload(Rfree, tap2, chunk2);
    if (tap == 0 && chunk + 1 < nchunks && patch_pf && !(ablate & 16)) pload((chunk + 1) * 64);
    const int toffb = GATHER ? 0 : __builtin_amdgcn_readlane(v_tapoff, tap) * ROWB;
    const unsigned char* pa0 = sA + arow[0] + toffb;
    const unsigned char* pa1 = sA + arow[1] + toffb;
    const unsigned char* pb = sB + buf * (BN * ROWB) + brow;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 a[2], b[NS];
      if (!(ablate & 8)) {
        a[0] = *reinterpret_cast<const bf16x8*>(pa0 + ks * 32);
        a[1] = *reinterpret_cast<const bf16x8*>(pa1 + ks * 32);
#pragma unroll
        for (int ns = 0; ns < NS; ++ns)
          b[ns] = *reinterpret_cast<const bf16x8*>(pb + ns * 32 * ROWB + ks * 32);
      } else {
        a[0] = a[1] = __builtin_bit_cast(bf16x8, Rnext[0]);
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) b[ns] = __builtin_bit_cast(bf16x8, Rnext[0]);
      }
      if (!(ablate & 1)) {
#pragma unroll
        for (int ms = 0; ms < 2; ++ms)
#pragma unroll
          for (int ns = 0; ns < NS; ++ns)
            acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ms], b[ns], acc[ms][ns], 0, 0, 0);
      } else {
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) asm volatile("" ::"v"(a[ms]));
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) asm volatile("" ::"v"(b[ns]));
      }
    }
    if (it + 1 < NIT) {
      if (last_tap) {                        // next iteration starts a new channel chunk
        if (!(ablate & 4)) __syncthreads();  // everyone is done reading the patch
        if (!(ablate & 16)) {
          if (patch_pf) pstore();
          else igemm_load_patch<GATHER, NTHREADS>(sA, in, g.Cin, (chunk + 1) * 64, p_lo, npix, in_pixels, s_pin, tid);
        }
      }
      // B(it+1) (fetched one iteration ago) -> the idle LDS buffer.  (Storing it at the START of
      // the iteration instead was measured ~8 % slower: the ds_writes then contend with the
      // fragment reads feeding the MFMAs.)
      if (!(ablate & 2)) bstore(Rnext, buf ^ 1);
      if (!(ablate & 4)) __syncthreads();
    }
    advance(tap, chunk);
    advance(tap2, chunk2);
  };
  for (int it = 0; it < NIT; it += 2) {
    body(R1, R0, it, 0);
    if (it + 1 < NIT) body(R0, R1, it + 1, 1);
  }

  // ---- epilogue ------------------------------------------------------------------------
  if (ABL && (ablate & 32)) return;
  const bool tail = igemm_tile_has_invalid(g, m0, BM);
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
    const int stripe = blockIdx.x % IIC_STAT_STRIPES;
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int q = 0; q < WM; ++q) {
      a0 += s_red[(q * 2 + 0) * BN + tid];
      a1 += s_red[(q * 2 + 1) * BN + tid];
    }
    iic_stat_add(stats, stripe, g.Cout, n0 + tid, 0, a0);
    iic_stat_add(stats, stripe, g.Cout, n0 + tid, 1, a1);
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
  igemm_store_tile<BN, BM, NTHREADS>(sC, s_pout, out, res_grad, res_act, accumulate, g.Cout, n0, tid);
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

IIC_SWITCH(g_ablate, 0, iic_debug_set_ablate)
IIC_SWITCH(g_force_bm, 0, iic_debug_force_bm)
#ifdef IIC_DEBUG_HOOKS
IIC_HOOK int iic_debug_get_ablate(void) { return g_ablate; }
#endif

static int pick_bn(int Cout) { return (Cout % 128 == 0) ? 128 : 64; }

static int geom_gather(const iic_conv_geom* g) { return g->ntaps == 1; }

static long lds_a_bytes_for(const iic_conv_geom* g, int BN, int BM) {
  long a = (long)(geom_gather(g) ? BM : (BM == 128 ? g->NP : g->NP256)) * ROWB;
  long c = (long)BM * (BN + 8) * 2;
  long m = a > c ? a : c;
  return (m + 15) & ~15L;
}

extern "C" {

static long lds_total(const iic_conv_geom* g, int BN, int BM) {
  return lds_a_bytes_for(g, BN, BM) + 2L * BN * ROWB + 2L * BM * 4 + (BM / 64) * 2L * BN * 4;
}

// largest M tile whose LDS footprint fits (256 rows halve the weight-tile traffic per FLOP)
static int pick_bm(const iic_conv_geom* g, int BN) {
  if (g_force_bm == 128 || g_force_bm == 256) return g_force_bm;
  const long M = igemm_rows_host(g);
  // measured (tools/conv_perf.py --bm): one 8-wave workgroup per CU loses the overlap two
  // independent 4-wave workgroups give; 256-row tiles stay opt-in (iic_debug_force_bm).
  (void)M;
  return 128;
}

long iic_conv_lds_bytes(const iic_conv_geom* g, int BN) {
  if (BN == 0) BN = pick_bn(g->Cout);
  return lds_total(g, BN, pick_bm(g, BN));
}

int iic_conv_igemm(const iic_conv_geom* g, const void* in, const void* w, void* out, float* stats,
                   const void* res_grad, const void* res_act, int accumulate, void* stream) {
  if (!g || !in || !w || !out) return IIC_ERR_ARG;
  if (g->Cin % 64 != 0 || g->Cout % 64 != 0 || g->ntaps < 1 || g->ntaps > IIC_MAX_TAPS)
    return IIC_ERR_UNSUPPORTED;
  if (!(accumulate & IIC_ACC_PREMASK) && (res_grad == nullptr) != (res_act == nullptr)) return IIC_ERR_ARG;
  const int BN = pick_bn(g->Cout);
  const long M = igemm_rows_host(g);
  if (M <= 0 || g->NP <= 0) return IIC_ERR_ARG;
  if (M >= (1L << 31) || (long)g->N * g->in_Hp * g->in_Wp >= (1L << 31)) return IIC_ERR_UNSUPPORTED;
  const int BMv = pick_bm(g, BN);
  const int mt = (int)((M + BMv - 1) / BMv);
  const int grid = mt * (g->Cout / BN);
  const long lds = lds_total(g, BN, BMv);
  if (lds > 160 * 1024) return IIC_ERR_UNSUPPORTED;
  const int la = (int)lds_a_bytes_for(g, BN, BMv);
  hipStream_t s = (hipStream_t)stream;
#define IGEMM_LAUNCH3(BN_, GA_, AB_, BM_)                                                         \
  do {                                                                                           \
    static bool attr = false;                                                                    \
    if (!attr) {                                                                                 \
      (void)hipFuncSetAttribute(                                                                 \
          reinterpret_cast<const void*>(&conv_igemm_kernel<BN_, GA_, AB_, BM_>),                 \
          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                               \
      attr = true;                                                                               \
    }                                                                                            \
    hipLaunchKernelGGL((conv_igemm_kernel<BN_, GA_, AB_, BM_>), dim3(grid), dim3(BM_ * 2), lds,  \
                       s, *g, (const bf16_t*)in, (const bf16_t*)w, (bf16_t*)out, stats,          \
                       (const bf16_t*)res_grad, (const bf16_t*)res_act, accumulate, mt, la,      \
                       g_ablate);                                                                \
  } while (0)
#define IGEMM_LAUNCH(BN_, GA_)                                                                   \
  do {                                                                                           \
    if (g_ablate) IGEMM_LAUNCH3(BN_, GA_, true, 128);                                            \
    else if (BMv == 256) IGEMM_LAUNCH3(BN_, GA_, false, 256);                                    \
    else IGEMM_LAUNCH3(BN_, GA_, false, 128);                                                    \
  } while (0)
  const bool ga = geom_gather(g);
  if (BN == 128) {
    if (ga) IGEMM_LAUNCH(128, true); else IGEMM_LAUNCH(128, false);
  } else {
    if (ga) IGEMM_LAUNCH(64, true); else IGEMM_LAUNCH(64, false);
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

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
#include "conv_tile.h"
#include "../../include/iic_hip.h"

#define BM 128      // K-tile: 128 output pixels
#define TPG 3       // taps per wave (tap group)
#define NTG 3       // tap groups per pass => 9 taps per pass
#define PFX 4       // patch 16-B pieces per thread prefetched in registers (NP <= 384 at 768 threads)

typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;

// MFMA operand fragment for lane (i = lane&31, g = lane>>5): the 8 values
//   tile[row_j][col], j = 0..7, rows = the 8 k-slots 8g..8g+7 of this k-step.
// USE_TR: two ds_read_b64_tr_b16; lane (q = lane>>4, i16 = lane&15) supplies the address of
// row 8*(q>>1) + 4*rd + (i16>>2), 8-byte segment (i16&3) of the 16-column block (q&1);
// a0/a1 are that lane's two row byte-addresses (column offset included).
// !USE_TR (semantic cross-check of the transpose-load model): 8 scalar LDS reads.
typedef uint16_t __attribute__((address_space(3))) * lds_u16_ptr;
// a0/a1/tile are 32-bit LDS byte addresses (keeps all address arithmetic in one VGPR each).
template <bool USE_TR>
__device__ __forceinline__ bf16x8 frag_T(uint32_t a0, uint32_t a1, uint32_t tile, int pitch,
                                         const int* rows8, int col) {
  union { bf16x8 v; s16x4 h[2]; uint16_t e[8]; } u;
  if (USE_TR) {
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(uintptr_t)a0);
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(uintptr_t)a1);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      u.e[j] = *(lds_u16_ptr)(uintptr_t)(tile + rows8[j] * pitch + col * 2);
  }
  return u.v;
}

// Workgroup = COT (64|128) output channels x 64 input channels x all taps.
//   non-gather: 12 waves = 2 (co halves) x 2 (ci halves) x 3 tap groups; a wave owns
//               (COT/2) co x 32 ci x 3 taps  (48 or 96 fp32 accumulators)
//   gather (1-tap convs): 4 waves, one tap.
// K-tiles (128 pixels) are software-pipelined: the next tile's input patch and dY rows are
// fetched into registers while the current tile is multiplied, then stored to LDS.
template <bool USE_TR, bool GATHER, int COT>
__global__ __launch_bounds__(GATHER ? 256 : 768) void conv_wgrad_kernel(
    const iic_conv_geom g, const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
    float* __restrict__ partials, int nsplit, int num_ktiles, int lds_x_bytes) {
  constexpr int NT = GATHER ? 256 : 768;
  constexpr int CS = COT / 64;                 // 32-wide co sub-tiles per wave
  constexpr int DPITCH = COT == 64 ? 144 : 288;  // dY tile row pitch (bytes)
  constexpr int DPIECES = BM * (COT / 8);      // 16-B pieces of the dY tile
  constexpr int PDN = (DPIECES + NT - 1) / NT;
  constexpr int MYT = GATHER ? 1 : TPG;        // taps per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* sX = smem_raw;                    // [NP][ROWB]   input patch, this WG's 64 ci
  unsigned char* sD = smem_raw + lds_x_bytes;      // [BM][DPITCH] dY rows,    this WG's COT co
  int* s_pin = reinterpret_cast<int*>(sD + BM * DPITCH);   // [2][BM]
  int* s_pout = s_pin + 2 * BM;                            // [2][BM]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tg = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
  const int l31 = lane & 31, g5 = lane >> 5;
  const int q = lane >> 4, i16 = lane & 15;
  const int ncit = g.Cin >> 6;
  // XCD-aware work mapping (as in conv_wgrad_dma.hip): workgroups are dealt to the 8 XCDs round-robin
  // by linear id, and all (co, ci) tiles of one K-split read the same input patch / dY rows -- when
  // the split count divides by 8, one XCD walks the tiles of a split before moving to its next
  // split, so those re-reads hit its L2 (measured at the 10a shapes: 2.6 GB of HBM traffic per
  // launch for ~1 GB of operands before the remap)
  int tile = blockIdx.x, split = blockIdx.y;
  if ((gridDim.y & 7) == 0) {
    const int L = blockIdx.y * gridDim.x + blockIdx.x;
    const int xcd = L & 7, j = L >> 3;
    tile = j % (int)gridDim.x;
    split = xcd + 8 * (j / (int)gridDim.x);
  }
  const int cot = tile / ncit, cit = tile - cot * ncit;
  const int co0 = cot * COT, ci0 = cit * 64;
  const int per = (num_ktiles + nsplit - 1) / nsplit;
  const int kt0 = split * per;
  const int kt1 = min(num_ktiles, kt0 + per);
  const int in_pixels = g.N * g.in_Hp * g.in_Wp;
  const int npix = GATHER ? BM : g.NP;
  const int n8 = npix * 8;
  const bool pf = (n8 <= PFX * NT);
  const int v_tapoff = g.tap_off[lane & (IIC_MAX_TAPS - 1)];

  // lane constants of the transposing reads
  const int trow = 8 * (q >> 1) + (i16 >> 2);                         // + 16*ks (+4 for rd=1)
  const int tsub = (2 * (q & 1) + ((i16 & 3) >> 1)) * 16 + (i16 & 1) * 8;
  typedef unsigned char __attribute__((address_space(3))) * lds_u8_ptr;
  const uint32_t sXo = (uint32_t)(uintptr_t)(lds_u8_ptr)sX, sDo = (uint32_t)(uintptr_t)(lds_u8_ptr)sD;
  const uint32_t aBase = sDo + trow * DPITCH + wm * (COT / 2) * 2 + tsub;
  const int tcolB = wn * 64 + tsub;

  auto rowinfo = [&](int kt, int b) {
    if (tid < BM) {
      int pin, pout;
      igemm_row_pixels(g, kt * BM + tid, pin, pout);
      s_pin[b * BM + tid] = pin;
      s_pout[b * BM + tid] = pout;
    }
  };
  u32x4 PX[PFX], PD[PDN];
  // `launder` hides tid from loop-invariant code motion: the per-piece address arithmetic is
  // recomputed where used instead of living in ~40 registers across the MFMA loop.
  auto launder = [](int v) { asm volatile("" : "+v"(v)); return v; };
  auto gload_d = [&](int b) {
    const int tl = launder(tid);
#pragma unroll
    for (int u = 0; u < PDN; ++u) {
      const int idx = u * NT + tl;
      PD[u] = (u32x4){0u, 0u, 0u, 0u};        // rows past M are zero => contribute nothing
      if (idx < DPIECES) {
        const int po = s_pout[b * BM + idx / (COT / 8)];
        if (po >= 0)
          PD[u] = *reinterpret_cast<const u32x4*>(
              dy + ((long)po * g.Cout + co0 + (idx % (COT / 8)) * 8));
      }
    }
  };
  auto lstore_d = [&]() {
    const int tl = launder(tid);
#pragma unroll
    for (int u = 0; u < PDN; ++u) {
      const int idx = u * NT + tl;
      if (idx < DPIECES)
        *reinterpret_cast<u32x4*>(sD + (idx / (COT / 8)) * DPITCH + (idx % (COT / 8)) * 16) = PD[u];
    }
  };
  auto gload_x = [&](int b) {
    const int p_lo = s_pin[b * BM];
    const int tl = launder(tid);
#pragma unroll
    for (int u = 0; u < PFX; ++u) {
      const int idx = u * NT + tl;
      PX[u] = (u32x4){0u, 0u, 0u, 0u};
      if (idx < n8) {
        const int p = GATHER ? s_pin[b * BM + (idx >> 3)] : p_lo + (idx >> 3);
        if (p < in_pixels)
          PX[u] = *reinterpret_cast<const u32x4*>(x + ((long)p * g.Cin + ci0 + (idx & 7) * 8));
      }
    }
  };
  auto lstore_x = [&]() {
    const int tl = launder(tid);
#pragma unroll
    for (int u = 0; u < PFX; ++u) {
      const int idx = u * NT + tl;
      if (idx < n8) *reinterpret_cast<u32x4*>(sX + (idx >> 3) * ROWB + (idx & 7) * 16) = PX[u];
    }
  };
  auto load_x_sync_big = [&](int b) {   // patches too large for the register prefetch
    const int p_lo = s_pin[b * BM];
    for (int base = 0; base < n8; base += NT * 4) {
      u32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = base + u * NT + tid;
        v[u] = (u32x4){0u, 0u, 0u, 0u};
        if (idx < n8) {
          const int p = p_lo + (idx >> 3);
          if (p < in_pixels)
            v[u] = *reinterpret_cast<const u32x4*>(x + ((long)p * g.Cin + ci0 + (idx & 7) * 8));
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = base + u * NT + tid;
        if (idx < n8) *reinterpret_cast<u32x4*>(sX + (idx >> 3) * ROWB + (idx & 7) * 16) = v[u];
      }
    }
  };

  // Tap batches (9 taps each) are a grid dimension (blockIdx.z): a 5x5 convolution used to walk its three
  // batches one after the other inside every workgroup, re-streaming all K-tiles of its split each time;
  // with one batch per workgroup the launch has 3x the workgroups at a third of the split count -- a third
  // of the split-K partials to write and to reduce (ClusterNet6c: 210 -> 70 MB per launch).
  for (int t0 = (int)blockIdx.z * (GATHER ? 1 : NTG * TPG), pass = 0; pass < 1; ++pass) {
    const int tfirst = t0 + tg * MYT;                      // this wave's first tap
    const int tcount = max(0, min(MYT, g.ntaps - tfirst));
    f32x16 acc[MYT][CS];
#pragma unroll
    for (int t = 0; t < MYT; ++t)
#pragma unroll
      for (int c = 0; c < CS; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][c][r] = 0.f;

    // ---- prologue: first tile of this split -----------------------------------------------
    __syncthreads();
    if (kt0 < kt1) {
      rowinfo(kt0, 0);
      __syncthreads();
      gload_d(0);
      if (pf) { gload_x(0); lstore_x(); } else load_x_sync_big(0);
      lstore_d();
      if (kt0 + 1 < kt1) rowinfo(kt0 + 1, 1);
      __syncthreads();
    }
    for (int kt = kt0; kt < kt1; ++kt) {
      const int b = (kt - kt0) & 1;
      const bool more = kt + 1 < kt1;
      if (more) {                              // next tile in flight during the MFMAs below
        gload_d(b ^ 1);
        if (pf) gload_x(b ^ 1);
      }
      const int p_lo = s_pin[b * BM];
      const int* spin = s_pin + b * BM;
#pragma unroll 2
      for (int ks = 0; ks < BM / 16; ++ks) {
        int rows8[8], rowsA[8];
        uint32_t b0, b1;
        {
          const int k0 = ks * 16 + trow, k1 = k0 + 4;
          const int r0 = GATHER ? k0 : spin[k0] - p_lo;
          const int r1 = GATHER ? k1 : spin[k1] - p_lo;
          b0 = sXo + r0 * ROWB + tcolB;
          b1 = sXo + r1 * ROWB + tcolB;
        }
        if (!USE_TR) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int k = ks * 16 + 8 * g5 + j;
            rows8[j] = GATHER ? k : spin[k] - p_lo;
            rowsA[j] = k;
          }
        }
        bf16x8 a[CS];
#pragma unroll
        for (int c = 0; c < CS; ++c)
          a[c] = frag_T<USE_TR>(aBase + ks * 16 * DPITCH + c * 64, aBase + (ks * 16 + 4) * DPITCH + c * 64,
                                sDo, DPITCH, rowsA, wm * (COT / 2) + c * 32 + l31);
#pragma unroll
        for (int t = 0; t < MYT; ++t) {
          if (t < tcount) {
            const int toff = GATHER ? 0 : __builtin_amdgcn_readlane(v_tapoff, tfirst + t);
            int rows8t[8];
            if (!USE_TR) {
#pragma unroll
              for (int j = 0; j < 8; ++j) rows8t[j] = rows8[j] + toff;
            }
            const bf16x8 bf = frag_T<USE_TR>(b0 + toff * ROWB, b1 + toff * ROWB, sXo, ROWB, rows8t,
                                             wn * 32 + l31);
#pragma unroll
            for (int c = 0; c < CS; ++c)
              acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[c], bf, acc[t][c], 0, 0, 0);
          }
        }
      }
      __syncthreads();                        // tile kt fully consumed
      if (more) {
        lstore_d();
        if (pf) lstore_x(); else load_x_sync_big(b ^ 1);
        if (kt + 2 < kt1) rowinfo(kt + 2, b);
        __syncthreads();
      }
    }
    // partial[split][t][co][ci]
#pragma unroll
    for (int t = 0; t < MYT; ++t) {
      if (t < tcount) {
        float* dst = partials + (((long)split * g.ntaps + (tfirst + t)) * g.Cout + co0) * g.Cin + ci0;
#pragma unroll
        for (int c = 0; c < CS; ++c)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = wm * (COT / 2) + c * 32 + mfma32_row(r, lane);
            const int col = wn * 32 + l31;
            dst[(long)row * g.Cin + col] = acc[t][c][r];
          }
      }
    }
  }
}

// dW[co][ci][t] (fp32 OIHW) (+)= sum_s partial[s][t][co][ci]
// 8 independent accumulators keep 8 loads in flight per thread (the split loop is otherwise a
// serial latency chain); reads are coalesced over (t, co, ci), the transposing write is small.
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ partials,
                                                                int nsplit, int T, int Co, int Ci,
                                                                float* __restrict__ dW,
                                                                int accumulate) {
  const long per = (long)T * Co * Ci;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per;
       i += (long)gridDim.x * blockDim.x) {
    float a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = 0.f;
    int sp = 0;
    for (; sp + 8 <= nsplit; sp += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += partials[(long)(sp + u) * per + i];
    }
    for (; sp < nsplit; ++sp) a[0] += partials[(long)sp * per + i];
    const float s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
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

// conv_wgrad_dma.hip: LDS-DMA double-buffered variant for the stride-1 3x3 layers
int iic_wgrad_dma_supported(const iic_conv_geom* g);
int iic_wgrad_dma_launch(const iic_conv_geom* g, const void* x, const void* dy, float* partials,
                         int nsplit, void* stream);

extern "C" {

static int wgrad_cot(const iic_conv_geom* g) { return (g->Cout % 128 == 0) ? 128 : 64; }

// Workgroups a weight-gradient launch aims for (tiles x K-splits); a 12-wave workgroup fills a CU.  256 (one per CU, rounds
// 1-3) means every CU has to take one, and in the two-stream step a CU whose LDS the other view's kernel holds delays the
// launch's last workgroup; 224 also writes 1/8 less split-K partial traffic.  Measured on three boxes, default step,
// interleaved (tools/ab_env.sh, profiles/r04_wgrad_target_wgs.txt): 224 / 192 / 128 are 0.2-0.35 ms per step faster than 256,
// 160 and 96 slower (K-tile quantisation), 64 much slower.
#ifdef IIC_DEBUG_HOOKS
static int g_wgrad_target_wgs = 224;
IIC_HOOK void iic_debug_wgrad_target_wgs(int v) { g_wgrad_target_wgs = v > 0 ? v : 224; }
#else
static constexpr int g_wgrad_target_wgs = 224;
#endif

int iic_conv_wgrad_nsplit(const iic_conv_geom* g) {
  const long M = igemm_rows_host(g);
  const int kt = (int)((M + BM - 1) / BM);
  const int batches = g->ntaps == 1 ? 1 : (g->ntaps + NTG * TPG - 1) / (NTG * TPG);
  const int tiles = (g->Cout / wgrad_cot(g)) * (g->Cin / 64) * batches;
  int ns = g_wgrad_target_wgs / (tiles > 0 ? tiles : 1);   // default: one workgroup (12 waves) per CU
  if (ns < 1) ns = 1;
  if (ns > kt) ns = kt;
  return ns;
}

int iic_conv_wgrad(const iic_conv_geom* g, const void* x, const void* dy, float* partials,
                   int nsplit, int use_tr, void* stream) {
  if (!g || !x || !dy || !partials || nsplit < 1) return IIC_ERR_ARG;
  if (g->Cin % 64 != 0 || g->Cout % 64 != 0 || g->ntaps < 1 || g->ntaps > IIC_MAX_TAPS)
    return IIC_ERR_UNSUPPORTED;
  const long M = igemm_rows_host(g);
  if (M >= (1L << 31) || (long)g->N * g->in_Hp * g->in_Wp >= (1L << 31)) return IIC_ERR_UNSUPPORTED;
  if (use_tr && iic_wgrad_dma_supported(g)) return iic_wgrad_dma_launch(g, x, dy, partials, nsplit, stream);
  const int kt = (int)((M + BM - 1) / BM);
  const bool ga = g->ntaps == 1;
  const int cot = wgrad_cot(g);
  const int lx = ((ga ? BM : g->NP) * ROWB + 15) & ~15;
  const long lds = (long)lx + BM * (cot == 64 ? 144 : 288) + 4 * BM * 4;
  if (lds > 160 * 1024) return IIC_ERR_UNSUPPORTED;
  dim3 grid((g->Cout / cot) * (g->Cin / 64), nsplit, ga ? g->ntaps : (g->ntaps + NTG * TPG - 1) / (NTG * TPG));
  hipStream_t s = (hipStream_t)stream;
#define WGRAD_LAUNCH(TR_, GA_, COT_)                                                            \
  do {                                                                                          \
    static bool attr = false;                                                                   \
    if (!attr) {                                                                                \
      (void)hipFuncSetAttribute(                                                                \
          reinterpret_cast<const void*>(&conv_wgrad_kernel<TR_, GA_, COT_>),                    \
          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                              \
      attr = true;                                                                              \
    }                                                                                           \
    hipLaunchKernelGGL((conv_wgrad_kernel<TR_, GA_, COT_>), grid, dim3(GA_ ? 256 : 768), lds,   \
                       s, *g, (const bf16_t*)x, (const bf16_t*)dy, partials, nsplit, kt, lx);   \
  } while (0)
#define WGRAD_LAUNCH2(TR_, GA_)                                                                 \
  do {                                                                                          \
    if (cot == 128) WGRAD_LAUNCH(TR_, GA_, 128); else WGRAD_LAUNCH(TR_, GA_, 64);               \
  } while (0)
  if (use_tr) { if (ga) WGRAD_LAUNCH2(true, true); else WGRAD_LAUNCH2(true, false); }
  else        { if (ga) WGRAD_LAUNCH2(false, true); else WGRAD_LAUNCH2(false, false); }
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

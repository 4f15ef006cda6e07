// Persistent, DMA-fed implicit-GEMM convolution for the 64 -> 64 channel 3x3 layers (ResNet
// layer1 of ClusterNet5g: /root/reference/code/archs/cluster/residual.py:4-7,19,22 at 64 planes;
// forward and backward-data).  These layers sit at the HBM/MFMA ridge (AI 288):
// 0.4 GB of activations move per launch for 117 GFLOP, so the kernel is organised around the
// memory pipeline instead of around the MFMA loop:
//   * one persistent workgroup per CU (8 waves) walks a contiguous range of 256-row tiles;
//   * the input patch of tile t+1 is fetched by LDS-DMA (global_load_lds_dwordx4: no VGPRs, every
//     piece in flight at once) into the second patch buffer while tile t computes, and the
//     output rows of tile t-1 are stored while tile t computes => HBM reads, HBM writes and MFMA
//     overlap inside ONE workgroup; two barriers per tile;
//   * the whole weight operand of the wave (9 taps x 64 k x 32 couts, MFMA B-fragment order, see
//     iic_weight_prep_frag) is loaded ONCE into 144 VGPRs and stays there for every tile;
//   * LDS patch rows are 128 B, unpadded (DMA writes lane-linear), with the 16-byte slot XOR-ed by
//     (row >> 1) & 7: the DMA applies the swizzle on its SOURCE address, ds_read_b128 of 16
//     consecutive rows is conflict-free;
//   * BatchNorm statistics are accumulated in registers across all tiles, one atomic pass at the
//     end of the kernel.
#include "common.h"
#include "conv_tile.h"
#include "../../include/iic_hip.h"

#define P64_BM 256
#define P64_BN 64
#define P64_THREADS 512
#define P64_NT 9
#define P64_CLD P64_BN               // unpadded epilogue tile: LDS is the scarce resource here
#define P64_SC_BYTES(BM_) ((BM_) * P64_CLD * 2)   // 32768 at 256 rows
#define P64_NTAB 4                   // row tables are written two tiles ahead of their use
#define P64_TAB_BYTES(BM_) (P64_NTAB * (BM_) * (4 + 2))   // s_pout (int) + s_prow (u16)

// (n, y, x) of a GEMM row, advanced by one tile (256 rows) at a time: no per-tile divisions.
struct P64Walk {
  int n, y, x;
};
__device__ __forceinline__ void p64_walk_init(P64Walk& w, const iic_conv_geom& g, int m) {
  const int plane = g.MY * g.MX;
  w.n = m / plane;
  const int r = m - w.n * plane;
  w.y = r / g.MX;
  w.x = r - w.y * g.MX;
}
__device__ __forceinline__ void p64_walk_advance(P64Walk& w, const iic_conv_geom& g, int d_y, int d_x) {
  w.x += d_x;
  w.y += d_y;
  if (w.x >= g.MX) { w.x -= g.MX; ++w.y; }
  while (w.y >= g.MY) { w.y -= g.MY; ++w.n; }
}
// input / output pixel of the row; rows past the end repeat the last row and are never stored
__device__ __forceinline__ void p64_walk_pixels(const P64Walk& w, const iic_conv_geom& g, int& pin,
                                                int& pout) {
  const bool valid = w.n < g.N;
  const int n = valid ? w.n : g.N - 1, y = valid ? w.y : g.MY - 1, x = valid ? w.x : g.MX - 1;
  pin = (n * g.in_Hp + y * g.sy + g.oy) * g.in_Wp + x * g.sx + g.ox;
  pout = valid ? (n * g.out_Hp + y * g.ty + g.py) * g.out_Wp + x * g.tx + g.px : -1;
}

// ABL: timing-ablation build (WRONG results): 1 = only the first patch is fetched, 2 = no output
// stores, 4 = one tap instead of nine.
// RED: fused BatchNorm-backward reduction over the stored rows (conv_tile.h); the partial sums
// stay in registers across all tiles of the workgroup.
// NW = 8: two waves per SIMD, wave tile 64 rows x 32 couts (one A fragment read per MFMA: the LDS read
// pipe is as busy as the matrix pipe).  NW = 4 ("wide", iic_debug_p64_wide(1)): one wave per SIMD with 64 rows x
// 64 couts -- every A fragment feeds two MFMAs (half the LDS read traffic), the whole 9 x 64 x 64 weight operand
// sits in 288 registers of the wave, the A fragments of tap t+1 are read while the MFMAs of tap t run.  Measured
// (round 3, tools/p64_phases.py, 660 x 49 x 49, forward): NW = 8 155-165 us, wide 186-199 us -- a single wave per
// SIMD does not cover its own LDS latency (K loop 9.7 k cycles per tile against 5.9 k + 1.6 k of barrier wait).
// Where a tile's 10.3 k cycles go at NW = 8 (matrix pipe alone: 4.6 k): K loop 5.9 k, barrier B 1.6 k, row tables +
// DMA issue 1.4 k, accumulators -> LDS 1.0 k, waiting for the patch 0.3 k (the DMA is hidden); the residual
// epilogue of backward-data adds 5.5 k (its 64 KB of residual-gradient / mask loads per tile are exposed).
// BM = 128 (with NW = 4, wave tiles 64 x 32 as at NW = 8): half-height tiles, TWO workgroups per CU -- the serial
// phases of one workgroup's tile loop (row tables + DMA issue, barriers, accumulators -> LDS, stores: 40 % of a
// tile's cycles with the matrix pipe idle, tools/p64_phases.py) run under the other workgroup's K loop.
template <int ABL, int RED, int NW, int BM>
__global__ __launch_bounds__(NW * 64) void conv_igemm_p64_kernel(
    const iic_conv_geom g, const bf16_t* __restrict__ in, const unsigned char* __restrict__ wfrag,
    bf16_t* __restrict__ out, float* __restrict__ stats, const bf16_t* __restrict__ res_grad,
    const bf16_t* __restrict__ res_act, int accumulate, int num_tiles, int pb_bytes, int max_tap_off,
    const bf16_t* __restrict__ red_y, const float* __restrict__ red_coef,
    const bf16_t* __restrict__ red_y2, float* __restrict__ red_stats, float* __restrict__ red_stats2,
    int g_spread, unsigned long long* __restrict__ prof) {
  // PROF (ABL bit 8, results CORRECT): wave 0 sums the cycles (s_memtime) its workgroup spends in each phase of
  // the tile loop into prof[blockIdx][8]: wait for patch + barrier A | store of tile t-1 | DMA issue + row
  // tables | K loop | barrier B | accumulators -> LDS; [6] = tiles, [7] = whole loop (tools/p64_phases.py)
  constexpr bool PROF = (ABL & 8) != 0;
  constexpr bool WIDE = NW * 64 == BM;          // one 64-row group per wave, all 64 couts
  constexpr int NWM = BM / 64;                  // 64-row groups of the tile
  constexpr int SC_BYTES = BM * P64_CLD * 2, TAB_ROWS = P64_NTAB * BM;
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, t_a = 0, t_b = 0, t_loop = 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* const sP0 = smem_raw;                    // patch buffer 0  [rows][128 B]
  unsigned char* const sP1 = smem_raw + pb_bytes;         // patch buffer 1
  bf16_t* const sC = reinterpret_cast<bf16_t*>(smem_raw + 2 * pb_bytes);   // [256][64]
  int* const s_pout = reinterpret_cast<int*>(smem_raw + 2 * pb_bytes + SC_BYTES);  // [4][256]
  // patch row (input pixel - first input pixel of the tile) of every tile row, < NP256 <= 65535
  unsigned short* const s_prow = reinterpret_cast<unsigned short*>(s_pout + TAB_ROWS);
  float* const s_red = reinterpret_cast<float*>(sC);      // [4 wm][2][64], after the last store

  // (WIDE / NWM / SC_BYTES / TAB_ROWS: see the top of the kernel)
  constexpr int NTH = NW * 64;
  constexpr int NCO = WIDE ? 2 : 1;             // 32-wide cout fragments per wave
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = WIDE ? wave : wave >> 1, wn = WIDE ? 0 : wave & 1;
  const int l31 = lane & 31, g5 = lane >> 5;
  const int M = g.N * g.MY * g.MX;
  const int in_pixels = g.N * g.in_Hp * g.in_Wp;

  // contiguous tile range of this workgroup; workgroups of one XCD own neighbouring ranges
  const int G = gridDim.x;
  const int wl = xcd_tile_index(blockIdx.x, G);
  const int t0 = (int)(((long)wl * num_tiles) / G), t1 = (int)(((long)(wl + 1) * num_tiles) / G);
  if (t0 >= t1) return;

  const int v_tapoff = g.tap_off[lane & (IIC_MAX_TAPS - 1)];
  const int d_y = BM / g.MX, d_x = BM - d_y * g.MX;

  // walkers: `wr` = this thread's table row (row tid & 255 of the next tile to tabulate),
  // `w0` / `wl255` = first / last row of that tile (uniform): p_lo and the span the tile needs
  P64Walk wr, w0, w255;
  p64_walk_init(wr, g, t0 * BM + (tid & (BM - 1)));
  p64_walk_init(w0, g, t0 * BM);
  p64_walk_init(w255, g, t0 * BM + BM - 1);
  int plo_q[3], nblk_q[3];      // [0] = tile being tabulated next - 2 ... rotating queue
  auto tabulate = [&](int t) {  // writes table t (the tile all three walkers point at), advances
    int pin, pout, p0, p255, dummy;
    p64_walk_pixels(w0, g, p0, dummy);
    p64_walk_pixels(w255, g, p255, dummy);
    p0 = __builtin_amdgcn_readfirstlane(p0);
    p255 = __builtin_amdgcn_readfirstlane(p255);
    if (tid < BM) {
      p64_walk_pixels(wr, g, pin, pout);
      s_pout[(t & (P64_NTAB - 1)) * BM + tid] = pout;
      s_prow[(t & (P64_NTAB - 1)) * BM + tid] = (unsigned short)(pin - p0);
    }
    p64_walk_advance(wr, g, d_y, d_x);
    p64_walk_advance(w0, g, d_y, d_x);
    p64_walk_advance(w255, g, d_y, d_x);
    plo_q[0] = plo_q[1];
    nblk_q[0] = nblk_q[1];
    plo_q[1] = plo_q[2];
    nblk_q[1] = nblk_q[2];
    plo_q[2] = p0;
    // 1-KB blocks covering the rows this tile reads: (p255 + max tap offset - p0 + 1) x 128 B
    nblk_q[2] = ((p255 + max_tap_off - p0 + 1) * 128 + 1023) >> 10;
  };
  // LDS-DMA of a patch: piece q -> LDS byte q*16 (row q>>3, physical slot q&7); the source is
  // the logical slot (q&7) ^ ((row>>1)&7) of pixel plo + row.
  auto dma_issue = [&](unsigned char* dst, int plo, int nblk) {
    for (int blk = wave; blk < nblk; blk += NW) {         // 1-KB blocks, wave-uniform
      const int q = blk * 64 + lane;
      const int r = q >> 3;
      const int ls = (q & 7) ^ ((r >> 1) & 7);
      int p = plo + r;
      p = p < in_pixels ? p : in_pixels - 1;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(in + ((long)p * 64 + ls * 8)),
          (__attribute__((address_space(3))) void*)(dst + blk * 1024), 16, 0, 0);
    }
  };

  // ---- prologue: tables of the first two tiles, first patch in flight, resident weights -------
  tabulate(t0);
  tabulate(t0 + 1);
  // queue now: [1] = tile t0, [2] = tile t0 + 1
  dma_issue(sP0, plo_q[1], nblk_q[1]);
  u32x4 Bw[P64_NT][NCO][4];
#pragma unroll
  for (int tap = 0; tap < P64_NT; ++tap)
#pragma unroll
    for (int c = 0; c < NCO; ++c) {
      const unsigned char* p = wfrag + ((long)g.tap_w[tap] * 2 + wn + c) * 4096 + lane * 16;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) Bw[tap][c][ks] = *reinterpret_cast<const u32x4*>(p + ks * 1024);
    }
  f32x2 st_s[NCO], st_ss[NCO];                   // BN statistics of column (wn + c)*32 + l31
#pragma unroll
  for (int c = 0; c < NCO; ++c) {
    st_s[c] = f32x2{0.f, 0.f};
    st_ss[c] = f32x2{0.f, 0.f};
  }
  TileRed tr;
  if (RED) tile_red_zero(tr);

  if (PROF) t_loop = __builtin_readcyclecounter();
  for (int t = t0; t < t1; ++t) {
    const int par = (t - t0) & 1;
    unsigned char* const sP = par ? sP1 : sP0;
    if (PROF) t_a = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of patch t has landed
    __syncthreads();                                    // A: patch t + tile t-1 in sC are complete
    if (PROF) { t_b = __builtin_readcyclecounter(); ph[0] += t_b - t_a; t_a = t_b; }
    // a plain store (no residual / previous contents / fused reduction: every forward launch) is spread
    // over the taps of the K loop below instead: issued in one go, 32 KB of stores per workgroup back up in
    // the CU's memory pipeline at the speed HBM drains them, with the matrix pipe idle meanwhile (measured:
    // compute-only 108 us + stores 34 us + patch reads 22 us = the full 170 us, i.e. no overlap)
    const bool spread = RED == 0 && g_spread && !res_grad && !res_act && !accumulate;
    if (t > t0 && !(ABL & 2) && !spread)
      igemm_store_tile<P64_BN, BM, NTH, 0, RED, (RED ? 1 : 4)>(sC, s_pout + ((t - 1) & (P64_NTAB - 1)) * BM,
                                                            out, res_grad, res_act, accumulate, P64_BN, 0, tid,
                                                            red_y, red_coef, red_y2, &tr);
    if (PROF) { t_b = __builtin_readcyclecounter(); ph[1] += t_b - t_a; t_a = t_b; }
    // queue: [2] = tile t + 1 (tabulated one iteration ago)
    if (t + 1 < t1 && !(ABL & 1)) dma_issue(par ? sP0 : sP1, plo_q[2], nblk_q[2]);
    tabulate(t + 2);
    if (PROF) { t_b = __builtin_readcyclecounter(); ph[2] += t_b - t_a; t_a = t_b; }
    // ---- tile t: 9 taps x 4 k-steps x (2 A reads, 2 MFMAs) --------------------------------
    int R0[2];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
      R0[ms] = s_prow[(t & (P64_NTAB - 1)) * BM + wm * 64 + ms * 32 + l31];
    f32x16 acc[2][NCO];
    constexpr int NTAPS = (ABL & 4) ? 1 : P64_NT;
    auto load_a = [&](int tap, bf16x8 (&a)[2][4]) {
      const int toff = __builtin_amdgcn_readlane(v_tapoff, tap);
#pragma unroll
      for (int ms = 0; ms < 2; ++ms) {
        const int R = R0[ms] + toff;
        const int key = (R >> 1) & 7;
        const int base = R * 128 + (((g5 ^ key) & 1) << 4);
        const int kk = (key >> 1) << 5;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          a[ms][ks] = *reinterpret_cast<const bf16x8*>(sP + base + ((ks << 5) ^ kk));
      }
    };
    bf16x8 abuf[WIDE ? 2 : 1][2][4];
    if (WIDE) load_a(0, abuf[0]);
#pragma unroll
    for (int tap = 0; tap < NTAPS; ++tap) {
      if (WIDE) {
        if (tap + 1 < NTAPS) load_a(tap + 1, abuf[(tap + 1) & 1]);     // under this tap's 16 MFMAs
      } else {
        load_a(tap, abuf[0]);
      }
      bf16x8 (&a)[2][4] = abuf[WIDE ? (tap & 1) : 0];
      if (spread && t > t0 && !(ABL & 2)) {
        constexpr int SIT = BM * 8 / NTH;           // 16-byte pieces per thread and tile (4 | 8)
        constexpr int PER = (SIT + P64_NT - 2) / (P64_NT - 1);
#pragma unroll
        for (int q = 0; q < PER; ++q) {
          const int it = (tap - 1) * PER + q;
          if (tap >= 1 && it < SIT) {
            const int idx = tid + it * NTH;
            const int row = idx >> 3, ch = idx & 7;
            const int po = s_pout[((t - 1) & (P64_NTAB - 1)) * BM + row];
            if (po >= 0)
              *reinterpret_cast<uint4*>(out + (long)po * P64_BN + ch * 8) =
                  *reinterpret_cast<const uint4*>(sC + row * P64_CLD + ch * 8);
          }
        }
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int c = 0; c < NCO; ++c) {
          const bf16x8 b = __builtin_bit_cast(bf16x8, Bw[tap][c][ks]);
          if (tap == 0 && ks == 0) {          // first MFMA of the tile: C operand = 0 (no zero-fill)
            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[0][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][ks], b, z, 0, 0, 0);
            acc[1][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][ks], b, z, 0, 0, 0);
          } else {
            acc[0][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][ks], b, acc[0][c], 0, 0, 0);
            acc[1][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][ks], b, acc[1][c], 0, 0, 0);
          }
        }
    }
    if (stats) {
      if ((t + 1) * BM > M) {           // last tile: rows past the end do not count
#pragma unroll
        for (int ms = 0; ms < 2; ++ms)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (t * BM + wm * 64 + ms * 32 + mfma32_row(r, lane) >= M) {
#pragma unroll
              for (int c = 0; c < NCO; ++c) acc[ms][c][r] = 0.f;
            }
      }
#pragma unroll
      for (int c = 0; c < NCO; ++c)
#pragma unroll
        for (int ms = 0; ms < 2; ++ms)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const f32x2 v = {acc[ms][c][r], acc[ms][c][r + 1]};
            st_s[c] += v;
            st_ss[c] += v * v;
          }
    }
    if (PROF) { t_b = __builtin_readcyclecounter(); ph[3] += t_b - t_a; t_a = t_b; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // B: the stores of tile t-1 have read sC
    if (PROF) { t_b = __builtin_readcyclecounter(); ph[4] += t_b - t_a; t_a = t_b; }
#pragma unroll
    for (int c = 0; c < NCO; ++c)
#pragma unroll
      for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          sC[(wm * 64 + ms * 32 + mfma32_row(r, lane)) * P64_CLD + (wn + c) * 32 + l31] = f32_to_bf16(acc[ms][c][r]);
    if (PROF) { t_b = __builtin_readcyclecounter(); ph[5] += t_b - t_a; }
  }
  if (PROF && prof && tid == 0) {
    unsigned long long* q = prof + (long)blockIdx.x * 8;
#pragma unroll
    for (int i = 0; i < 6; ++i) q[i] = ph[i];
    q[6] = (unsigned long long)(t1 - t0);
    q[7] = __builtin_readcyclecounter() - t_loop;
  }

  // ---- drain: last tile's rows, then the statistics -------------------------------------------
  __syncthreads();
  if (!(ABL & 2))
    igemm_store_tile<P64_BN, BM, NTH, 0, RED, (RED ? 1 : 4)>(sC, s_pout + ((t1 - 1) & (P64_NTAB - 1)) * BM, out,
                                                          res_grad, res_act, accumulate, P64_BN, 0, tid,
                                                          red_y, red_coef, red_y2, &tr);
  if (RED)
    igemm_red_finish<P64_BN, NTH, RED>(tr, reinterpret_cast<float*>(sC), red_stats, red_stats2, P64_BN,
                                               0, tid);
  if (stats) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCO; ++c) {
      float s1 = st_s[c][0] + st_s[c][1], s2 = st_ss[c][0] + st_ss[c][1];
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (lane < 32) {
        s_red[(wm * 2 + 0) * P64_BN + (wn + c) * 32 + lane] = s1;
        s_red[(wm * 2 + 1) * P64_BN + (wn + c) * 32 + lane] = s2;
      }
    }
    __syncthreads();
    if (tid < P64_BN) {
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int q = 0; q < NWM; ++q) {
        a0 += s_red[(q * 2 + 0) * P64_BN + tid];
        a1 += s_red[(q * 2 + 1) * P64_BN + tid];
      }
      const int stripe = blockIdx.x % IIC_STAT_STRIPES;
      iic_stat_add(stats, stripe, P64_BN, tid, 0, a0);
      iic_stat_add(stats, stripe, P64_BN, tid, 1, a1);
    }
  }
}

#ifdef IIC_DEBUG_HOOKS
static unsigned long long* g_p64_prof = nullptr;   // ablate 8: per-workgroup phase cycle sums go here
IIC_HOOK void iic_debug_p64_prof(void* buf) { g_p64_prof = (unsigned long long*)buf; }
#else
static constexpr unsigned long long* g_p64_prof = nullptr;
#endif
IIC_SWITCH(g_p64_spread, 1, iic_debug_p64_spread)   // 1: plain output stores spread over the K loop's taps (see the kernel)
IIC_SWITCH(g_p64_wide, 0, iic_debug_p64_wide)       // 1: four "wide" waves per workgroup (64 x 64 wave tiles), 0: eight 64 x 32 waves
IIC_SWITCH(g_p64_grid, 0, iic_debug_p64_grid)       // tests: force a small persistent grid (many tiles per workgroup)

static int p64_num_cus() {
  if (g_p64_grid > 0) return g_p64_grid;
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      n = v;
    else
      n = 256;
  }
  return n;
}

static long p64_pb_bytes(const iic_conv_geom* g, int bm = P64_BM) {
  return (((long)(bm == 128 ? g->NP : g->NP256) * 128) + 1023) & ~1023L;
}
// Half-height tiles with two workgroups per CU (BM = 128: the kernel is written for it) would let one
// workgroup's serial phases run under the other's K loop, but do not fit: at layer1 of ClusterNet5g a 128-row
// tile still spans 340 patch rows (42.5 KB; 474 rows at 256), so two double-buffered workgroups need 208 KB.

// used by conv_igemm_bd.hip's dispatcher
int iic_p64_supported(const iic_conv_geom* g) {
  if (g->Cin != 64 || g->Cout != 64 || g->ntaps != P64_NT || g->NP256 <= 0 || g->NP256 > 65535) return 0;
  if (!igemm_dense_host(g)) return 0;       // the row walkers assume the dense row numbering
  return 2 * p64_pb_bytes(g) + P64_SC_BYTES(P64_BM) + P64_TAB_BYTES(P64_BM) <= 160 * 1024;
}

int iic_p64_launch(const iic_conv_geom* g, const void* in, const void* wfrag, void* out, float* stats,
                   const void* res_grad, const void* res_act, int accumulate, const void* red_y,
                   const float* red_coef, const void* red_y2, float* red_stats, float* red_stats2,
                   void* stream) {
  const int red = red_y ? (red_y2 ? 2 : 1) : 0;
  const long M = (long)g->N * g->MY * g->MX;
  if (M <= 0) return IIC_ERR_ARG;
  if (M >= (1L << 31) - P64_BM || (long)g->N * g->in_Hp * g->in_Wp >= (1L << 31)) return IIC_ERR_UNSUPPORTED;
  const int bm = P64_BM;
  const int nt = (int)((M + bm - 1) / bm);
  const int pb = (int)p64_pb_bytes(g, bm);
  const long lds = 2L * pb + P64_SC_BYTES(bm) + P64_TAB_BYTES(bm);
  int mto = 0;
  for (int i = 0; i < g->ntaps; ++i) mto = g->tap_off[i] > mto ? g->tap_off[i] : mto;
  const int ncu = p64_num_cus();
  const int grid = nt < ncu ? nt : ncu;
#define P64_LAUNCH3(AB_, RD_, NW_, BM_)                                                                     \
  do {                                                                                                     \
    static bool attr = false;                                                                              \
    if (!attr) {                                                                                           \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_p64_kernel<AB_, RD_, NW_, BM_>), \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                   \
      attr = true;                                                                                         \
    }                                                                                                      \
    hipLaunchKernelGGL((conv_igemm_p64_kernel<AB_, RD_, NW_, BM_>), dim3(grid), dim3(NW_ * 64), lds,       \
                       (hipStream_t)stream, *g, (const bf16_t*)in, (const unsigned char*)wfrag,       \
                       (bf16_t*)out, stats, (const bf16_t*)res_grad, (const bf16_t*)res_act,          \
                       accumulate, nt, pb, mto, (const bf16_t*)red_y, red_coef,                       \
                       (const bf16_t*)red_y2, red_stats, red_stats2, g_p64_spread, g_p64_prof);       \
  } while (0)
#define P64_LAUNCH2(AB_, RD_)                                              \
  do {                                                                     \
    if (g_p64_wide) P64_LAUNCH3(AB_, RD_, 4, 256);                         \
    else P64_LAUNCH3(AB_, RD_, 8, 256);                                    \
  } while (0)
#define P64_LAUNCH(AB_) P64_LAUNCH2(AB_, 0)
  if (red == 1) { P64_LAUNCH2(0, 1); return iic_launch_status(); }
  if (red == 2) { P64_LAUNCH2(0, 2); return iic_launch_status(); }
  switch (iic_debug_get_ablate()) {
#ifdef IIC_BD_ABLATIONS
    // timing-ablation / phase-profile instantiations (tools/p64_phases.py): `make -C iic_amd/csrc ABL=1` only
    case 1: P64_LAUNCH(1); break;
    case 2: P64_LAUNCH(2); break;
    case 3: P64_LAUNCH(3); break;
    case 4: P64_LAUNCH(4); break;
    case 7: P64_LAUNCH(7); break;
    case 8: P64_LAUNCH(8); break;
#endif
    default: P64_LAUNCH(0); break;
  }
  return iic_launch_status();
}

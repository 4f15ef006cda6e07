// "Weights-direct" implicit-GEMM convolution for gfx950 (Cout % 128 == 0): the second
// generation of conv_igemm.hip's kernel for the deep layers.
//
// Replaces the cuDNN kernels behind nn.Conv2d in
//   /root/reference/code/archs/cluster/residual.py:4-7,19,22,54-55  (3x3 s1/s2, 1x1 s2)
//   /root/reference/code/archs/cluster/vgg.py:24-26                 (5x5, dilated 3x3)
//
// What conv_igemm_kernel pays per 16 MFMAs per wave: a 16 KB weight tile written to LDS by
// ds_write_b128 (13 LDS cycles each), 16 ds_read_b128 and a workgroup barrier.  Here
//   * the weights never touch LDS: iic_weight_prep_frag stores them in MFMA B-fragment order
//     ([tap][64-ch chunk][32 couts][k-step][lane][8 bf16]), so one global_load_dwordx4 per lane
//     IS a B operand and a wave-instruction reads 1 KB contiguous from L2.  A ring of the next
//     iteration's 8 fragments is kept in flight in registers;
//   * the input patch (all pixels a 256-row tile touches over all taps, 64 channels) is the only
//     LDS resident, read-only for a whole chunk => barriers only at chunk boundaries
//     (every ntaps*4 k-steps) instead of every tap;
//   * wave tile 128(M) x 64(N): 4 A-fragment reads + 2 B loads feed 8 MFMAs (LDS bytes per FLOP
//     halved again); workgroup = 256 x 128 tile, 4 waves as 2(M) x 2(N), 2 workgroups per CU.
// Geometry, PT layout, fused BN statistics / residual-gradient epilogue: as conv_igemm.hip.
#include "common.h"
#include "conv_tile.h"
#include "../../include/iic_hip.h"

// B-fragment loads as inline asm: invisible to hipcc's vmcnt bookkeeping, so that the K loop can keep
// exactly 8 of them in flight behind hand-counted waits while its LDS reads sit BETWEEN the MFMAs (a
// compiler-visible load would be waited for conservatively at the loop header, see bd_bwait).
__device__ __forceinline__ void bd_bload(u32x4& d, const unsigned char* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory");
}
// wait until at most N vector-memory operations of this wave are outstanding; `d` (the fragment the
// wait is for) becomes opaque here, so no consumer of it can be scheduled above the wait
template <int N>
__device__ __forceinline__ void bd_bwait(u32x4& d) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(d) : "i"(N) : "memory");
}

// Block tiling (round 6, LAB.md R6.14; the weight gradient's form: conv_wgrad_dma.hip conv_wgrad_b2d_kernel): a 256-row tile
// is a 2-D block of bw x bh OUTPUT pixels of one image instead of 256 consecutive rows of the row-major numbering, and
// its LDS patch the (bw + max tap dx) x (bh + max tap dy) sub-image under it -- for 100-200-pixel image rows and dilated
// taps 1.3-1.6 x the pixels used instead of 2.7-2.8 x, so the 64-KB patches of SegmentationNet10a's c2 / c5 / c6 shrink to
// 41-51 KB (two workgroups per CU again).  Everything behind the patch loader works in PATCH coordinates: the row
// table holds ky * PW + kx, the tap offsets become iy * PW + ix, the swizzle key counts patch rows.  Which tile computes
// an output row changes; the row's own accumulation order does not: outputs are bit-identical to the row-major tiles.
struct bd_blk {
  int bw, bh, nbx, nby;      // bw == 0: row-major tiles
  int PW, npix, mul;         // patch width, patch pixels, ceil(65536 / PW) (patch row / PW by multiply-shift)
};

#define BD_BM 256
#define BD_BN 128
#define BD_THREADS 256
#define BD_PROF_SLOTS 16
#ifndef BD_STORE_UB
#define BD_STORE_UB 8      // rows per thread whose epilogue loads are in flight together (16 per tile)
#endif
#ifndef BD_RELOAD_NB
#define BD_RELOAD_NB 4     // 16-B patch pieces in flight per thread at a chunk boundary (8 spills)
#endif

// ABL: timing-ablation build (tools/conv_perf.py --ablate; results are WRONG by design):
//   1 = no patch reload at chunk boundaries, 2 = B fragments always from the same address
//   (L2-hot), 4 = no epilogue, 8 = no chunk-boundary barriers either, 16 = no A-fragment LDS reads,
//   32 = no global stores of the tile, 64 = no BatchNorm statistics.
// DMA: the patch is fetched by LDS-DMA (global_load_lds_dwordx4: no staging VGPRs, every piece of
// a chunk in flight at once instead of 4-piece batches that each wait a full memory latency) into
// unpadded 128-byte rows whose 16-byte slot is XOR-ed with a 3-bit key of the row -- the swizzle is
// applied on the DMA's SOURCE address; A-fragment addresses then cost ~2 VALU per read instead of
// an immediate offset.  The key is (D >> 1) & 7 with D = p - J * (p / in_Wp), p the row's pixel
// index and J = in_Wp - MX the padding a GEMM row run skips at the end of an image row: D counts
// pixels the way consecutive GEMM rows visit them, so the 16 lanes of a ds_read_b128 group (rows
// of consecutive m, any tap) see 16 consecutive D = 16 distinct (row parity, slot) pairs = all 64
// banks, whereas a key on the raw row index collided across every row end (SQ_LDS_BANK_CONFLICT
// was 51 % of the LDS cycles at 13- and 25-pixel rows).  !DMA: register-staged loads into
// 144-byte-pitch rows (first version).
// MS: 32-row MFMA sub-tiles per wave: 4 = 256-row workgroup tile (2 workgroups per CU), 2 = 128-row
// tile (half the patch and accumulators: 3 workgroups per CU -- more tiles in flight to hide the
// per-tile prologue / epilogue where the K loop is short, i.e. few input channels).
// RED: fused BatchNorm-backward reduction over the stored tile (conv_tile.h), 0 = off.
// WN: 64-cout column groups of the workgroup tile.  2: 128 couts, waves 2 (rows) x 2 (cols), MS*64 rows;
// 1: 64 couts (layers with Cout = 64: the backward-data of a 64 -> 128 convolution, SegmentationNet10a c2), the
// four waves stacked along the rows, MS*128 rows -- the wave tile stays MS x 2 MFMA blocks either way.
template <bool GATHER, int ABL, bool DMA, int MS, int RED, int WN = 2>
__device__ __forceinline__ void bd_tile(
    const iic_conv_geom& g, const bf16_t* __restrict__ in, const unsigned char* __restrict__ wfrag,
    bf16_t* __restrict__ out, float* __restrict__ stats, const bf16_t* __restrict__ res_grad,
    const bf16_t* __restrict__ res_act, int accumulate, int lds_a_bytes,
    int dense_key, const bf16_t* __restrict__ red_y, const float* __restrict__ red_coef,
    const bf16_t* __restrict__ red_y2, float* __restrict__ red_stats, float* __restrict__ red_stats2,
    unsigned long long* __restrict__ prof, int stagger, int blk_in_class, int nwg_class, int m_base,
    const bd_blk& B) {
  constexpr int BNT = WN * 64, NWM = 4 / WN;   // tile couts, wave row groups
  constexpr bool CANBLK = !GATHER && DMA && MS * 32 * (4 / WN) == 256;      // 256-row tiles of either width
  const bool blk = CANBLK && B.bw > 0;         // uniform
  constexpr int CLD = BNT + 8;
  constexpr bool PROF = (ABL & 128) != 0;
  constexpr bool NEWORD = (ABL & 256) == 0;     // ABL bit 256: the round-2 K-loop order (A/B runs)
  // ABL bit 512 (timing only, WRONG results): A-fragment addresses as for a 144-byte row pitch -- one add per tap and
  // sub-tile, immediate k-step offsets -- while the DMA still writes the swizzled 128-byte rows: what the K loop would
  // cost without its ~70 VALU of swizzle arithmetic per tap (LAB.md R6.9)
  constexpr bool SWZ = DMA && (ABL & 512) == 0;
  // PROF (ABL bit 128, results CORRECT): wave 0 stamps s_memtime at the phase boundaries of its tile
  // into prof[blockIdx][BD_PROF_SLOTS] (tools/bd_timeline.py decodes them)
  unsigned long long t_stamp[8];
  unsigned long long t_bsum = 0, t_b0 = 0;
  int t_nb = 0;
  if (PROF) t_stamp[0] = __builtin_readcyclecounter();
  // Stagger (stagger > 0 cycles): the two workgroups that share a CU start together and do identical
  // work, so their prologues, chunk-boundary reloads and epilogues coincide instead of hiding behind
  // each other's MFMA loops.  The workgroup in the odd threadgroup slot of its CU (HW_ID.TG_ID) of the
  // FIRST dispatch round waits `stagger` cycles once; later workgroups inherit the offset from the
  // slot they take over.
  if (stagger > 0 && (int)blockIdx.x < 2 * 256) {
    const unsigned hw = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 16 << 6 | 4);     // HW_ID.TG_ID
    if (hw & 1u) {
      const unsigned long long t_go = __builtin_readcyclecounter() + (unsigned long long)stagger;
      while (__builtin_readcyclecounter() < t_go) __builtin_amdgcn_s_sleep(16);
    }
  }
  constexpr int BM = MS * 32 * NWM, WR = MS * 32;     // workgroup tile rows, rows per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* sA = smem_raw;                                   // [NP256][ROWB]
  int* s_pin = reinterpret_cast<int*>(smem_raw + lds_a_bytes);     // [256]
  int* s_pout = s_pin + BM;                                     // [256]
  float* s_red = reinterpret_cast<float*>(s_pout + BM);         // [NWM wm][2][BNT]  (512 floats)
  unsigned char* s_key = reinterpret_cast<unsigned char*>(s_red + NWM * 2 * BNT);   // DMA: [npix] swizzle keys (after 512 floats)
  bf16_t* sC = reinterpret_cast<bf16_t*>(smem_raw);                // epilogue reuse of sA

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = WN == 2 ? wave >> 1 : wave, wn = WN == 2 ? wave & 1 : 0;
  const int l31 = lane & 31, g5 = lane >> 5;

  const int nt = g.Cout / BNT;
  const int tix = xcd_tile_index(blk_in_class, nwg_class);
  const int mtile = tix / nt, ntile = tix - mtile * nt;
  const int n0 = ntile * BNT;
  const int m0 = m_base + mtile * BM;
  const int in_pixels = g.N * g.in_Hp * g.in_Wp;

  int v_tapoff = g.tap_off[lane & (IIC_MAX_TAPS - 1)];
  const int v_tapw = g.tap_w[lane & (IIC_MAX_TAPS - 1)];
  // block tiling: this tile's image and block, its first input pixel (tap 0), whether the block is whole
  int blk_glo = 0;
  bool blk_full = true;
  if (blk) {
    const int tpi = B.nbx * B.nby;
    const int bn = mtile / tpi, brem = mtile - bn * tpi;
    const int by = brem / B.nbx, bx = brem - by * B.nbx;
    const int y0 = by * B.bh, x0 = bx * B.bw;
    blk_glo = (bn * g.in_Hp + y0 + g.oy) * g.in_Wp + x0 + g.ox;
    blk_full = y0 + B.bh <= g.MY && x0 + B.bw <= g.MX && B.bw * B.bh == BM;
    v_tapoff = (v_tapoff / g.in_Wp) * B.PW + v_tapoff % g.in_Wp;        // tap offsets in patch rows
    if (tid < BM) {
      int ky = tid / B.bw;
      const int kx = tid - ky * B.bw;
      const int y = y0 + ky, x = x0 + kx;
      const bool valid = ky < B.bh && y < g.MY && x < g.MX;
      ky = ky < B.bh ? ky : B.bh - 1;                                    // (rows past the block: any patch row)
      s_pin[tid] = ky * B.PW + kx;
      s_pout[tid] = valid ? (bn * g.out_Hp + y + g.py) * g.out_Wp + x + g.px : -1;
    }
  } else if (tid < BM) {
    int pin, pout;
    igemm_row_pixels(g, m0 + tid, pin, pout);
    s_pin[tid] = pin;
    s_pout[tid] = pout;
  }
  __syncthreads();
  const int p_lo = blk ? 0 : s_pin[0];
  const int npix = GATHER ? BM : (blk ? B.npix : (BM >= 192 ? g.NP256 : g.NP));    // (192-row tiles: bound checked by the host)
  // swizzle key of a pixel (see the header): D = p - J * (p / in_Wp); J must be even so that D keeps
  // the row parity (the 128-B half of the 256-B bank window is the physical row parity)
  // (block tiling: the same in patch coordinates -- row pitch PW, J = PW - bw, even by the host's choice)
  const int keyw = blk ? B.PW : g.in_Wp;
  const int jskip = blk ? B.PW - B.bw
                        : ((dense_key && !GATHER && g.sx == 1 && ((g.in_Wp - g.MX) & 1) == 0) ? g.in_Wp - g.MX : 0);
  auto dense_of = [&](int p) { return p - jskip * (p / keyw); };
  int arow[MS];     // !DMA: byte offset of the lane's row (tap 0) + k-chunk; DMA: patch row index
  int drow[MS];     // DMA: D of the lane's row at tap offset 0
#pragma unroll
  for (int ms = 0; ms < MS; ++ms) {
    const int row = wm * WR + ms * 32 + l31;
    const int pr = GATHER ? row : (s_pin[row] - p_lo);
    arow[ms] = SWZ ? pr : pr * ROWB + g5 * 16;
    drow[ms] = (SWZ && !GATHER) ? dense_of(s_pin[row]) : pr;
  }
  if (DMA && !GATHER && jskip != 0) {     // (without a skip the key is a function of r alone: no table)
    for (int r = tid; r < npix; r += BD_THREADS) s_key[r] = (unsigned char)((dense_of(p_lo + r) >> 1) & 7);
    __syncthreads();
  }
  // per-tap increment of D: tap_off = dy * in_Wp + dx  ->  dy * (in_Wp - J) + dx
  const int v_tapd = v_tapoff - jskip * (v_tapoff / keyw);
  // DMA patch loader: 1-KB blocks over the 4 waves; piece q -> LDS byte q*16 (row q>>3, physical
  // slot q&7), source = logical slot (q&7) ^ ((row>>1)&7) of the row's pixel
  const int nblk = (npix * 128 + 1023) >> 10;
  auto dma_patch = [&](int c0) {
    for (int kb = wave; kb < nblk; kb += BD_THREADS / 64) {
      const int q = kb * 64 + lane;
      const int r = q >> 3;
      const int ls = (q & 7) ^ (GATHER ? ((r >> 1) & 7)
                                       : (jskip != 0 ? (int)s_key[r < npix ? r : npix - 1]
                                                     : (((p_lo + r) >> 1) & 7)));
      long p = GATHER ? (long)s_pin[r < BM ? r : BM - 1] : (long)p_lo + r;
      if (blk) {                                   // patch row r = (py, px) of the sub-image under the block
        const int rc = r < npix ? r : npix - 1;
        const int py = (rc * B.mul) >> 16;
        p = (long)blk_glo + py * g.in_Wp + (rc - py * B.PW);
      }
      p = p < in_pixels ? p : in_pixels - 1;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(in + (p * g.Cin + c0 + ls * 8)),
          (__attribute__((address_space(3))) void*)(sA + kb * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  // DMA layout: byte address of A fragment (row R, k-step ks, lane half g5) = R*128 + ((2ks+g5) ^ key)*16,
  // key = (D >> 1) & 7 of the row
  auto a_base = [&](int R, int D) { return R * 128 + (((g5 ^ (D >> 1)) & 1) << 4); };
  auto a_kk = [&](int D) { return ((D >> 2) & 3) << 5; };

  f32x16 acc[MS][2];
#pragma unroll
  for (int ms = 0; ms < MS; ++ms)
#pragma unroll
    for (int ns = 0; ns < 2; ++ns)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ms][ns][r] = 0.f;

  const int nchunks = g.Cin >> 6;
  const int ntaps = g.ntaps;
  const int NIT = nchunks * ntaps;
  // this wave's B fragments of iteration (tw, chunk): 2 x 4 KB contiguous
  const long frag_it = (long)(g.Cout >> 5) * 4096;      // bytes per (tap, chunk)
  const unsigned char* wb0 = wfrag + (long)((n0 + wn * 64) >> 5) * 4096 + lane * 16;
  auto frag_ptr = [&](int tap, int chunk) {
    const int tw = __builtin_amdgcn_readlane(v_tapw, tap);
    return wb0 + ((long)tw * nchunks + chunk) * frag_it;
  };

  // ---- prologue ------------------------------------------------------------------------
  if (PROF) t_stamp[1] = __builtin_readcyclecounter();
  u32x4 Bc[4][2];
  {
    const unsigned char* p = frag_ptr(0, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int ns = 0; ns < 2; ++ns)
        if (NEWORD) bd_bload(Bc[ks][ns], p + ns * 4096 + ks * 1024);
        else Bc[ks][ns] = *reinterpret_cast<const u32x4*>(p + ns * 4096 + ks * 1024);
  }
  if (DMA) {
    dma_patch(0);
  } else {
    // the accumulators are not live yet => 16 pieces (64 VGPRs) in flight per thread
    igemm_load_patch<GATHER, BD_THREADS, 16>(sA, in, g.Cin, 0, p_lo, npix, in_pixels, s_pin, tid);
  }
  __syncthreads();
  if (PROF) t_stamp[2] = __builtin_readcyclecounter();

  // ---- main loop: flat (chunk, tap) iterations; barriers only when the chunk changes ------
  // Software pipeline, explicit so that the register allocator cannot fold it away:
  //   A fragments double-buffered (a[ks & 1]): k-step ks issues the LDS reads of k-step ks + 1
  //   (of the NEXT tap when ks == 3) before its own 8 MFMAs;
  //   B fragments: right after its MFMAs, k-step ks re-fills Bc[ks] for the next iteration, so 6
  //   newer loads are always in flight behind the one being waited for (vmcnt(6)).
  // sched_barrier(0) pins each group: without it the scheduler sinks all 8 loads to the end of
  // the iteration and recycles the B registers as A temporaries (=> vmcnt(0) every iteration).
  int tap = 0, chunk = 0;
  bf16x8 a[2][MS];
  int pcur[MS], kcur[MS];       // kcur: DMA only (k-step XOR term of the row's swizzle)
#pragma unroll
  for (int ms = 0; ms < MS; ++ms) {
    const int t0 = GATHER ? 0 : __builtin_amdgcn_readlane(v_tapoff, 0);
    if (SWZ) {
      const int R = arow[ms] + t0;
      const int D = drow[ms] + (GATHER ? 0 : __builtin_amdgcn_readlane(v_tapd, 0));
      pcur[ms] = a_base(R, D);
      kcur[ms] = a_kk(D);
      a[0][ms] = *reinterpret_cast<const bf16x8*>(sA + pcur[ms] + kcur[ms]);
    } else {
      pcur[ms] = arow[ms] + t0 * ROWB;
      kcur[ms] = 0;
      a[0][ms] = *reinterpret_cast<const bf16x8*>(sA + pcur[ms]);
    }
  }
  for (int it = 0; it < NIT; ++it) {
    int tn = tap + 1, cn = chunk;
    if (tn == ntaps) { tn = 0; ++cn; }
    const bool more = (it + 1 < NIT);
    if (!more) { tn = tap; cn = chunk; }   // the ring always reloads: no branch around a load
    const unsigned char* nb = (ABL & 2) ? frag_ptr(0, 0) : frag_ptr(tn, cn);
    const int toffn = GATHER ? 0 : __builtin_amdgcn_readlane(v_tapoff, tn);
    const int tdn = GATHER ? 0 : __builtin_amdgcn_readlane(v_tapd, tn);
    int pnext[MS], knext[MS];
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) {
      if (SWZ) {
        const int R = arow[ms] + toffn;
        const int D = drow[ms] + tdn;
        pnext[ms] = a_base(R, D);
        knext[ms] = a_kk(D);
      } else {
        pnext[ms] = arow[ms] + toffn * ROWB;
        knext[ms] = 0;
      }
    }
    auto a_read = [&](int ks, int nxt, int cur, int ms) {
      if (!(ABL & 16)) {
        if (SWZ)
          a[nxt][ms] = (ks < 3) ? *reinterpret_cast<const bf16x8*>(sA + pcur[ms] + (((ks + 1) << 5) ^ kcur[ms]))
                                : *reinterpret_cast<const bf16x8*>(sA + pnext[ms] + knext[ms]);
        else
          a[nxt][ms] = (ks < 3) ? *reinterpret_cast<const bf16x8*>(sA + pcur[ms] + (ks + 1) * 32)
                                : *reinterpret_cast<const bf16x8*>(sA + pnext[ms]);
      } else {
        a[nxt][ms] = a[cur][ms];
      }
    };
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (NEWORD) {
        // Round 3 (tools/mfma_feed.py): [4 reads][8 MFMAs][2 loads] blocks leave the matrix pipe idle
        // while a wave works through its feeder block; one LDS read between consecutive MFMAs keeps
        // both pipes issuing (+5...14 % in the staged microbenchmark at these loop lengths).
        // ns-major MFMA order: Bc[ks][0] is free after the first MS MFMAs and is re-filled there.
        // In flight at the top of a k-step: the 8 loads of the next four k-steps, oldest first.
        bd_bwait<7>(Bc[ks][0]);
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 b0 = __builtin_bit_cast(bf16x8, Bc[ks][0]);
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) {
          acc[ms][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][ms], b0, acc[ms][0], 0, 0, 0);
          a_read(ks, nxt, cur, ms);
          __builtin_amdgcn_sched_barrier(0);
        }
        bd_bwait<6>(Bc[ks][1]);
        bd_bload(Bc[ks][0], nb + ks * 1024);
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 b1 = __builtin_bit_cast(bf16x8, Bc[ks][1]);
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
          acc[ms][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][ms], b1, acc[ms][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        bd_bload(Bc[ks][1], nb + 4096 + ks * 1024);
        __builtin_amdgcn_sched_barrier(0);
        continue;
      }
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) a_read(ks, nxt, cur, ms);
      __builtin_amdgcn_sched_barrier(0);   // reads of the NEXT k-step issue before these MFMAs
      const bf16x8 b0 = __builtin_bit_cast(bf16x8, Bc[ks][0]);
      const bf16x8 b1 = __builtin_bit_cast(bf16x8, Bc[ks][1]);
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) {
        acc[ms][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][ms], b0, acc[ms][0], 0, 0, 0);
        acc[ms][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][ms], b1, acc[ms][1], 0, 0, 0);
      }
      Bc[ks][0] = *reinterpret_cast<const u32x4*>(nb + ks * 1024);
      Bc[ks][1] = *reinterpret_cast<const u32x4*>(nb + 4096 + ks * 1024);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) {
      pcur[ms] = pnext[ms];
      kcur[ms] = knext[ms];
    }
    if (more && tn == 0 && !(ABL & 8)) {       // next iteration starts a new channel chunk
      if (PROF) t_b0 = __builtin_readcyclecounter();
      __syncthreads();           // everyone is done reading the patch
      if (!(ABL & 1)) {
        if (DMA)
          dma_patch(cn * 64);
        else
          igemm_load_patch<GATHER, BD_THREADS, BD_RELOAD_NB>(sA, in, g.Cin, cn * 64, p_lo, npix, in_pixels, s_pin, tid);
      }
      __syncthreads();
      if (PROF) { t_bsum += __builtin_readcyclecounter() - t_b0; ++t_nb; }
#pragma unroll
      for (int ms = 0; ms < MS; ++ms)
        a[0][ms] = *reinterpret_cast<const bf16x8*>(sA + pcur[ms] + (SWZ ? kcur[ms] : 0));
    }
    tap = tn;
    chunk = cn;
  }

  // the ring re-fills unconditionally (the last iteration's loads are never used): hipcc does not know
  // about them and would hand their destination registers to the epilogue while they are in flight
  if (NEWORD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // ---- epilogue ------------------------------------------------------------------------
  if (PROF) t_stamp[3] = __builtin_readcyclecounter();
  if (ABL & 4) {
    float t = 0.f;   // keep every accumulator live
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
      for (int ns = 0; ns < 2; ++ns)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[ms][ns][r];
    if (t == 123.f) out[0] = 1;
    return;
  }
  const bool tail = blk ? !blk_full : igemm_tile_has_invalid(g, m0, BM);
  if (stats && !(ABL & 64)) {
    if (tail) {       // rows past the end / in the row padding do not count
#pragma unroll
      for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (s_pout[wm * WR + ms * 32 + mfma32_row(r, lane)] < 0) {
            acc[ms][0][r] = 0.f;
            acc[ms][1][r] = 0.f;
          }
    }
#pragma unroll
    for (int ns = 0; ns < 2; ++ns) {
      f32x2 s2 = {0.f, 0.f}, ss2 = {0.f, 0.f};      // packed fp32 adds / fmas: half the VALU count
#pragma unroll
      for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 v = {acc[ms][ns][r], acc[ms][ns][r + 1]};
          s2 += v;
          ss2 += v * v;
        }
      float s = s2[0] + s2[1], ss = ss2[0] + ss2[1];
      s += __shfl_xor(s, 32, 64);
      ss += __shfl_xor(ss, 32, 64);
      if (lane < 32) {
        const int col = wn * 64 + ns * 32 + lane;
        s_red[(wm * 2 + 0) * BNT + col] = s;
        s_red[(wm * 2 + 1) * BNT + col] = ss;
      }
    }
  }
  __syncthreads();   // all waves finished reading sA; s_red complete
  if (stats && tid < BNT && !(ABL & 64)) {
    const int stripe = blockIdx.x % IIC_STAT_STRIPES;
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int q = 0; q < NWM; ++q) {
      a0 += s_red[(q * 2 + 0) * BNT + tid];
      a1 += s_red[(q * 2 + 1) * BNT + tid];
    }
    iic_stat_add(stats, stripe, g.Cout, n0 + tid, 0, a0);
    iic_stat_add(stats, stripe, g.Cout, n0 + tid, 1, a1);
  }
#pragma unroll
  for (int ms = 0; ms < MS; ++ms)
#pragma unroll
    for (int ns = 0; ns < 2; ++ns)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * WR + ms * 32 + mfma32_row(r, lane);
        const int col = wn * 64 + ns * 32 + l31;
        sC[row * CLD + col] = f32_to_bf16(acc[ms][ns][r]);
      }
  __syncthreads();
  if (PROF) t_stamp[4] = __builtin_readcyclecounter();
  TileRed tr;
  if (RED) tile_red_zero(tr);
  if (!(ABL & 32))
    igemm_store_tile<BNT, BM, BD_THREADS, 8, RED, (BM == 256 ? BD_STORE_UB : 4)>(sC, s_pout, out, res_grad, res_act, accumulate,
                                                                 g.Cout, n0, tid, red_y, red_coef, red_y2, &tr);
  if (RED)
    igemm_red_finish<BNT, BD_THREADS, RED>(tr, reinterpret_cast<float*>(smem_raw), red_stats, red_stats2,
                                             g.Cout, n0, tid);
  if (PROF && prof && tid == 0) {
    unsigned long long* q = prof + (long)blockIdx.x * BD_PROF_SLOTS;
    const unsigned long long t_end = __builtin_readcyclecounter();
    q[0] = t_stamp[0]; q[1] = t_stamp[1]; q[2] = t_stamp[2]; q[3] = t_stamp[3]; q[4] = t_stamp[4];
    q[5] = t_end; q[6] = t_bsum; q[7] = (unsigned long long)t_nb;
    q[8] = (unsigned long long)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4)     // HW_REG_HW_ID
           | ((unsigned long long)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) << 32);  // XCC_ID
    q[9] = (unsigned long long)tix;
    q[10] = __builtin_amdgcn_s_memrealtime();
  }
}

// Launch shape.  Tiles are equal work, so a launch runs in whole rounds over the chip's workgroup slots
// (2 per CU): 1 612 layer-2 tiles take 4 rounds for 3.15 rounds of work, 872 layer-3 tiles 2 for 1.70
// (tools/bd_timeline.py: slot occupancy 0.79 / 0.85).  MS2 != 0: the tiles of the last, partial round are
// MS2 * 64 rows high instead of 256, sized so that they still fit ONE round (host: the split below): the
// first n_big workgroups take 256-row tiles of rows [0, m_split), the others MS2-tiles of the rest.
// Results per output row are unchanged (same K order); only the grouping of the statistics partials differs.
template <bool GATHER, int ABL, bool DMA, int MS, int RED, int MS2, int WN = 2>
__global__ __launch_bounds__(BD_THREADS, ((MS == 4 || MS2 != 0 || WN == 1) ? 2 : 3)) void conv_igemm_bd_kernel(
    const iic_conv_geom g, const bf16_t* __restrict__ in, const unsigned char* __restrict__ wfrag,
    bf16_t* __restrict__ out, float* __restrict__ stats, const bf16_t* __restrict__ res_grad,
    const bf16_t* __restrict__ res_act, int accumulate, int num_mtiles, int lds_a_bytes,
    int dense_key, const bf16_t* __restrict__ red_y, const float* __restrict__ red_coef,
    const bf16_t* __restrict__ red_y2, float* __restrict__ red_stats, float* __restrict__ red_stats2,
    unsigned long long* __restrict__ prof, int stagger, int n_big, int m_split, const bd_blk B) {
  const bd_blk B0 = {0, 0, 0, 0, 0, 0, 0};
  if (WN == 1) {
    bd_tile<GATHER, ABL, DMA, MS, RED, 1>(g, in, wfrag, out, stats, res_grad, res_act, accumulate, lds_a_bytes,
                                          dense_key, red_y, red_coef, red_y2, red_stats, red_stats2, prof, stagger,
                                          (int)blockIdx.x, num_mtiles * (g.Cout / 64), 0, B);
  } else if (MS2 == 0 || (int)blockIdx.x < n_big) {
    bd_tile<GATHER, ABL, DMA, MS, RED>(g, in, wfrag, out, stats, res_grad, res_act, accumulate, lds_a_bytes,
                                       dense_key, red_y, red_coef, red_y2, red_stats, red_stats2, prof, stagger,
                                       (int)blockIdx.x, MS2 == 0 ? num_mtiles * (g.Cout / BD_BN) : n_big, 0, MS2 == 0 ? B : B0);
  } else {
    bd_tile<GATHER, ABL, DMA, (MS2 == 0 ? MS : MS2), RED>(
        g, in, wfrag, out, stats, res_grad, res_act, accumulate, lds_a_bytes, dense_key, red_y, red_coef, red_y2,
        red_stats, red_stats2, prof, stagger, (int)blockIdx.x - n_big, (int)gridDim.x - n_big, m_split, B0);
  }
}

// fp32 OIHW -> bf16 MFMA-B-fragment order.  mode 0 (forward operand): GEMM N = Cout, K = Cin;
// mode 1 (backward-data operand): N = Cin, K = Cout.
//   out[tap][kchunk][n/32][ks][lane][e] = W[n = j*32 + (lane & 31)][k = kchunk*64 + ks*16 + (lane>>5)*8 + e][tap]
__global__ void weight_prep_frag_kernel(const float* __restrict__ w, bf16_t* __restrict__ o, int Co,
                                        int Ci, int T, int mode) {
  const int Nn = mode ? Ci : Co, Kk = mode ? Co : Ci;
  const int n32 = Nn >> 5, nchunks = Kk >> 6;
  const long total = (long)T * Co * Ci;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63), ks = (int)((i >> 9) & 3);
    long r = i >> 11;
    const int j = (int)(r % n32);
    r /= n32;
    const int kc = (int)(r % nchunks), t = (int)(r / nchunks);
    const int n = j * 32 + (lane & 31), k = kc * 64 + ks * 16 + (lane >> 5) * 8 + e;
    const int co = mode ? k : n, ci = mode ? n : k;
    o[i] = f32_to_bf16(w[((long)co * Ci + ci) * T + t]);
  }
}

// conv_igemm_p64.hip: persistent DMA-fed kernel for the 64 -> 64 channel 3x3 layers
int iic_p64_supported(const iic_conv_geom* g);
int iic_p64_launch(const iic_conv_geom* g, const void* in, const void* wfrag, void* out, float* stats,
                   const void* res_grad, const void* res_act, int accumulate, const void* red_y,
                   const float* red_coef, const void* red_y2, float* red_stats, float* red_stats2,
                   void* stream);
// conv_igemm_pw.hip: persistent kernel for the stride-1 multi-tap launches (round 4)
int iic_pw_supported(const iic_conv_geom* g);
int iic_pw_launch(const iic_conv_geom* g, const void* in, const void* wfrag, void* out, float* stats,
                  const void* res_grad, const void* res_act, int accumulate, const void* red_y,
                  const float* red_coef, const void* red_y2, float* red_stats, float* red_stats2, void* stream);
// also take LDS footprints that leave room for only one workgroup per CU (large-image segmentation layers: still
// 15-20 % faster than conv_igemm_kernel)
IIC_SWITCH(g_bd_one_wg, 1, iic_debug_bd_one_wg)
IIC_SWITCH(g_p64_enabled, 1, iic_debug_enable_p64)
IIC_SWITCH(g_p64_red, 0, iic_debug_p64_red)      // 1: allow the fused reduction in the persistent kernel (tests, A/B)

extern "C" {

#ifdef IIC_DEBUG_HOOKS
static unsigned long long* g_bd_prof = nullptr;   // iic_debug_set_ablate(128): per-workgroup phase stamps go here
IIC_HOOK void iic_debug_bd_prof(void* buf) { g_bd_prof = (unsigned long long*)buf; }
IIC_HOOK int iic_debug_bd_prof_slots(void) { return BD_PROF_SLOTS; }
#else
static constexpr unsigned long long* g_bd_prof = nullptr;
#endif
IIC_SWITCH(g_bd_stagger, 0, iic_debug_bd_stagger)   // cycles per tap by which odd-slot workgroups of the first round start late (0 = off)
IIC_SWITCH(g_bd_dma, 1, iic_debug_bd_dma)           // 1: LDS-DMA patch loads (128-B swizzled rows), 0: register-staged (144-B rows)

// ms: 4 = 256-row tiles, 2 = 128-row tiles (the kernel's MS)
// (wn: the kernel's WN -- 1 = 64-cout tiles of ms*128 rows)
static long bd_lds_a(const iic_conv_geom* g, int ms, int wn = 2) {
  const long bm = ms * 32 * (4 / wn);
  const long npix = g->ntaps == 1 ? bm : (bm >= 192 ? g->NP256 : g->NP);
  long a = g_bd_dma ? ((npix * 128 + 1023) & ~1023L) : npix * ROWB;
  long c = bm * (wn * 64 + 8) * 2;
  long m = a > c ? a : c;
  return (m + 15) & ~15L;
}

// input pixel of GEMM row m (conv_tile.h igemm_row_pixels, host side)
static long bd_row_pin_host(const iic_conv_geom* g, long m) {
  const long plane = (long)g->MY * g->MX, mp = g->MP > 0 ? g->MP : plane;
  long n = m / mp, r = m - n * mp;
  if (n >= g->N) { n = g->N - 1; r = plane - 1; }
  r = r < plane ? r : plane - 1;
  const long y = r / g->MX, x = r - y * g->MX;
  return (n * g->in_Hp + y * g->sy + g->oy) * g->in_Wp + x * g->sx + g->ox;
}
// 1: last partial round in smaller tiles. Per launch alone -5 % (layer 2 / 3), but inside the two-stream step the other
// view already fills those tails: 38.0 vs 37.7 ms/step (r03 A/B) => off
IIC_SWITCH(g_bd_mixed, 0, iic_debug_bd_mixed)
IIC_SWITCH(g_bd_dense_key, 1, iic_debug_bd_dense_key)   // 0: swizzle key from the raw pixel index (A/B: conflicts at row ends)
static long bd_key_bytes(const iic_conv_geom* g, int ms, int wn = 2) {    // swizzle-key table of the DMA patch (1 B / row)
  const int jskip = (g_bd_dense_key && g->ntaps > 1 && g->sx == 1 && ((g->in_Wp - g->MX) & 1) == 0) ? g->in_Wp - g->MX : 0;
  return (g_bd_dma && jskip != 0) ? (((long)(ms * 32 * (4 / wn) >= 192 ? g->NP256 : g->NP) + 15) & ~15L) : 0;
}
static long bd_lds_total(const iic_conv_geom* g, int ms, int wn = 2) {
  return bd_lds_a(g, ms, wn) + 2L * ms * 32 * (4 / wn) * 4 + 4L * BD_BN * 4 + bd_key_bytes(g, ms, wn);
}
// 64-cout tiles (kernel WN = 1, 256 rows): layers whose Cout is an odd multiple of 64
IIC_SWITCH(g_bd_w1, 1, iic_debug_bd_w1)
static bool bd_w1_ok(const iic_conv_geom* g) {
  return g_bd_w1 && g_bd_dma && g->Cout % 64 == 0 && g->Cout % BD_BN != 0 && g->ntaps > 1 && g->NP256 > 0 &&
         bd_lds_total(g, 2, 1) <= 160 * 1024;
}
// Tile height per geometry.  g_bd_ms: 0 = heuristic, 2 / 4 = forced (A/B runs, tests).
IIC_SWITCH(g_bd_ms, 0, iic_debug_bd_ms)
static int bd_pick_ms(const iic_conv_geom* g) {
  const bool ok4 = (g->ntaps == 1 || g->NP256 > 0) && bd_lds_total(g, 4) <= (g_bd_one_wg ? 160 : 80) * 1024;
  const bool ok2 = (g->ntaps == 1 || g->NP > 0) && bd_lds_total(g, 2) <= 160 * 1024;
  if (g_bd_ms == 4) return ok4 ? 4 : (ok2 ? 2 : 0);
  if (g_bd_ms == 2) return ok2 ? 2 : (ok4 ? 4 : 0);
  // small launches (a rank's share of the batch under strong scaling: tools/pairs_sweep.sh): when the 256-row tiles
  // fill less than half of the chip's 512 workgroup slots, 128-row tiles (3 workgroups per CU) halve the time a
  // launch is held up by its single round (660 / 4 images: conv family 7.58 -> 7.08 ms per step)
  if (ok4 && ok2 && g->ntaps > 1) {
    const long M = igemm_rows_host(g);
    if (((M + 255) / 256) * (g->Cout / BD_BN) * 2 <= 512) return 2;
  }
  return ok4 ? 4 : (ok2 ? 2 : 0);
}

static int bd_block_config(const iic_conv_geom* g, bd_blk* B, int wn);
/* 1 if iic_conv_igemm_frag can run this geometry (else use iic_conv_igemm). */
int iic_conv_igemm_frag_supported(const iic_conv_geom* g) {
  if (!g) return 0;
  if (g_p64_enabled && iic_p64_supported(g)) return 1;
  if (g->Cin % 64 != 0 || g->ntaps < 1 || g->ntaps > IIC_MAX_TAPS) return 0;
  // (images too wide for any row-major patch -- > ~290 pixels at dilation 2 -- still run here on block tiles)
  bd_blk B;
  if (g->Cout % BD_BN != 0) return (bd_w1_ok(g) || (g->Cout % 64 == 0 && bd_block_config(g, &B, 1))) ? 1 : 0;
  return bd_pick_ms(g) != 0 || bd_block_config(g, &B, 2) != 0;
}

int iic_conv_igemm_frag_red(const iic_conv_geom* g, const void* in, const void* wfrag, void* out,
                            float* stats, const void* res_grad, const void* res_act, int accumulate,
                            const void* red_y, const float* red_coef, const void* red_y2,
                            float* red_stats, float* red_stats2, void* stream);

/* 1 if iic_conv_igemm_frag_red can fuse a reduction into this geometry's launch. */
// Block tiling of the 256-row tiles (see bd_blk): stride-1 multi-tap launches on images of at least 32 pixels whose
// row-major 256-row patch is too big for two workgroups per CU, where the sub-image patch is not.  Block shape: fewest
// tiles x (MFMA time + half the patch bytes), over shapes whose patch keeps two workgroups per CU.
IIC_SWITCH(g_bd_blk, 1, iic_debug_bd_blk)       // 0: row-major tiles only; 2: block tiles wherever they apply (A/B)
IIC_SWITCH(g_bd_blk_bw, 0, iic_debug_bd_blk_bw)  // > 0: force this block width where it is valid (shape sweeps)
static int bd_block_config(const iic_conv_geom* g, bd_blk* B, int wn) {      // wn: the kernel's WN (tile couts / 64)
  B->bw = 0;
  if (!g_bd_blk || !g_bd_dma || g->ntaps < 2 || g->Cout % (wn * 64) != 0 || g->Cin % 64 != 0) return 0;
  if (g->sy != 1 || g->sx != 1 || g->ty != 1 || g->tx != 1 || g->MY < 32 || g->MX < 32 || g->NP256 <= 0) return 0;
  int mix = 0, miy = 0;
  for (int t = 0; t < g->ntaps; ++t) {
    const int iy = g->tap_off[t] / g->in_Wp, ix = g->tap_off[t] % g->in_Wp;
    if (ix > 8 || iy > 8) return 0;               // (taps on a small grid)
    mix = ix > mix ? ix : mix;
    miy = iy > miy ? iy : miy;
  }
  if (mix & 1) return 0;                          // swizzle key: the row-end skip of a patch row must be even
  double best = -1.0;
  int best_bw = 0;
  long best_tiles = 0;
  for (int bw = 8; bw <= 64 && bw <= g->MX; ++bw) {
    const int bh = BD_BM / bw;
    if (bh < 2 || bh > g->MY) continue;
    if (g_bd_blk_bw > 0 && bw != g_bd_blk_bw) continue;
    const long tiles = (long)((g->MX + bw - 1) / bw) * ((g->MY + bh - 1) / bh);
    const long npix = (long)(bw + mix) * (bh + miy);
    const long a = (npix * 128 + 1023) & ~1023L;
    const long c = (long)BD_BM * (wn * 64 + 8) * 2;
    const long tot = (a > c ? a : c) + 2L * BD_BM * 4 + 4L * BD_BN * 4 + ((npix + 15) & ~15L);
    if (tot > 80 * 1024 || npix >= 65536 / (bw + mix)) continue;       // two workgroups per CU; multiply-shift division
    const double cost = (double)tiles * (1.0 + 0.5 * (double)npix * 128.0 / 32768.0);
    if (best < 0 || cost < best) { best = cost; best_bw = bw; best_tiles = tiles; }
  }
  if (best < 0) return 0;
  const int bw = best_bw, bh = BD_BM / bw;
  if (g_bd_blk_bw == 0 && (double)g->MY * g->MX < 0.88 * (double)BD_BM * (double)best_tiles) return 0;      // > 12 % idle rows
  const int npix = (bw + mix) * (bh + miy);
  // worth it where the row-major patch costs the second workgroup of a CU or is much larger
  if (g_bd_blk != 2 && !(bd_lds_total(g, wn == 2 ? 4 : 2, wn) > 80 * 1024 || npix * 10 < g->NP256 * 7)) return 0;
  B->bw = bw; B->bh = bh;
  B->nbx = (g->MX + bw - 1) / bw; B->nby = (g->MY + bh - 1) / bh;
  B->PW = bw + mix; B->npix = npix; B->mul = (65536 + B->PW - 1) / B->PW;
  return 1;
}

int iic_conv_igemm_red_supported(const iic_conv_geom* g) {
  if (!g || !iic_conv_igemm_frag_supported(g)) return 0;
  if (g_p64_enabled && iic_p64_supported(g)) return g_p64_red;
  return g->ntaps > 1;
}

int iic_conv_igemm_frag(const iic_conv_geom* g, const void* in, const void* wfrag, void* out,
                        float* stats, const void* res_grad, const void* res_act, int accumulate,
                        void* stream) {
  return iic_conv_igemm_frag_red(g, in, wfrag, out, stats, res_grad, res_act, accumulate, nullptr, nullptr,
                                 nullptr, nullptr, nullptr, stream);
}

int iic_conv_igemm_frag_red(const iic_conv_geom* g, const void* in, const void* wfrag, void* out,
                            float* stats, const void* res_grad, const void* res_act, int accumulate,
                            const void* red_y, const float* red_coef, const void* red_y2,
                            float* red_stats, float* red_stats2, void* stream) {
  if (!g || !in || !wfrag || !out) return IIC_ERR_ARG;
  if ((red_y == nullptr) != (red_stats == nullptr) || (red_y2 == nullptr) != (red_stats2 == nullptr) ||
      (red_y2 && !red_y) || (red_coef && !red_y))
    return IIC_ERR_ARG;
  const int red = red_y ? (red_y2 ? 2 : 1) : 0;
  if (!(accumulate & IIC_ACC_PREMASK) && (res_grad == nullptr) != (res_act == nullptr)) return IIC_ERR_ARG;
  if (!iic_conv_igemm_frag_supported(g)) return IIC_ERR_UNSUPPORTED;
  if (g_p64_enabled && iic_p64_supported(g)) {
    // the persistent kernel stores tile t-1 right before tile t's MFMA loop: loads of y there
    // stall every wave once per tile (measured +120 us per launch) -- callers keep the separate
    // reduction pass for the 64 -> 64 layers (iic_conv_igemm_red_supported)
    if (red && !g_p64_red) return IIC_ERR_UNSUPPORTED;
    return iic_p64_launch(g, in, wfrag, out, stats, res_grad, res_act, accumulate, red_y, red_coef, red_y2,
                          red_stats, red_stats2, stream);
  }
  if (iic_debug_get_ablate() == 0 && g_bd_dma && g_bd_ms == 0 && iic_pw_supported(g))
    return iic_pw_launch(g, in, wfrag, out, stats, res_grad, res_act, accumulate, red_y, red_coef, red_y2, red_stats,
                         red_stats2, stream);
  const long M = igemm_rows_host(g);
  if (M <= 0) return IIC_ERR_ARG;
  if (M >= (1L << 31) || (long)g->N * g->in_Hp * g->in_Wp >= (1L << 31)) return IIC_ERR_UNSUPPORTED;
  if (g->Cout % BD_BN != 0) {       // 64-cout tiles
    bd_blk B1 = {0, 0, 0, 0, 0, 0, 0};
    const bool blocked1 = bd_block_config(g, &B1, 1) != 0;
    const int mt1 = blocked1 ? g->N * B1.nbx * B1.nby : (int)((M + 255) / 256);
    const int grid1 = mt1 * (g->Cout / 64);
    int la1 = (int)bd_lds_a(g, 2, 1);
    long lds1 = bd_lds_total(g, 2, 1);
    if (blocked1) {
      const long a = ((long)B1.npix * 128 + 1023) & ~1023L, c = (long)BD_BM * (64 + 8) * 2;
      la1 = (int)(((a > c ? a : c) + 15) & ~15L);
      lds1 = la1 + 2L * BD_BM * 4 + 4L * BD_BN * 4 + ((B1.npix + 15) & ~15L);
    }
    hipStream_t s1 = (hipStream_t)stream;
#define BD_LAUNCH_W1(RD_)                                                                              \
  do {                                                                                                 \
    static bool attr = false;                                                                          \
    if (!attr) {                                                                                       \
      (void)hipFuncSetAttribute(                                                                       \
          reinterpret_cast<const void*>(&conv_igemm_bd_kernel<false, 0, true, 2, RD_, 0, 1>),          \
          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                     \
      attr = true;                                                                                     \
    }                                                                                                  \
    hipLaunchKernelGGL((conv_igemm_bd_kernel<false, 0, true, 2, RD_, 0, 1>), dim3(grid1),              \
                       dim3(BD_THREADS), lds1, s1, *g, (const bf16_t*)in, (const unsigned char*)wfrag, \
                       (bf16_t*)out, stats, (const bf16_t*)res_grad, (const bf16_t*)res_act,           \
                       accumulate, mt1, la1, g_bd_dense_key, (const bf16_t*)red_y, red_coef,           \
                       (const bf16_t*)red_y2, red_stats, red_stats2, (unsigned long long*)nullptr, 0,  \
                       0, 0, B1);                                                                      \
  } while (0)
    if (red == 0) BD_LAUNCH_W1(0); else if (red == 1) BD_LAUNCH_W1(1); else BD_LAUNCH_W1(2);
    return iic_launch_status();
  }
  bd_blk BB = {0, 0, 0, 0, 0, 0, 0};
  const bool blocked = iic_debug_get_ablate() == 0 && bd_block_config(g, &BB, 2) != 0;
  const int ms = blocked ? 4 : bd_pick_ms(g);
  if (ms == 0) return IIC_ERR_UNSUPPORTED;      // (only block tiles fit and an ablation build / switch turned them off)
  const int bm = ms * 64;
  const int mt = blocked ? g->N * BB.nbx * BB.nby : (int)((M + bm - 1) / bm);
  const int nt = g->Cout / BD_BN;
  int grid = mt * nt;
  int la = (int)bd_lds_a(g, ms);
  long lds = bd_lds_total(g, ms);
  if (blocked) {
    const long a = ((long)BB.npix * 128 + 1023) & ~1023L, c = (long)BD_BM * (BD_BN + 8) * 2;
    la = (int)(((a > c ? a : c) + 15) & ~15L);
    lds = la + 2L * BD_BM * 4 + 4L * BD_BN * 4 + ((BB.npix + 15) & ~15L);
  }
  // last partial round in smaller tiles (see conv_igemm_bd_kernel)
  int ms2 = 0, n_big = 0, m_split = 0;
  if (g_bd_mixed && !blocked && ms == 4 && g->ntaps > 1 && g_bd_dma && iic_debug_get_ablate() == 0) {
    const int slots = (lds <= 80 * 1024 ? 2 : 1) * 256;
    const int per_round = slots / nt;                       // m-tiles per round
    const int full = per_round > 0 ? mt / per_round : 0;    // whole rounds of 256-row tiles
    if (full >= 1 && mt % per_round != 0) {
      const long rem = M - (long)full * per_round * 256;
      for (int c = 2; c <= 3 && ms2 == 0; ++c)
        if ((rem + 64 * c - 1) / (64 * c) <= per_round) ms2 = c;
      if (ms2 != 0) {
        // the small tiles stage their patch in the 256-row tile's LDS image: every span must fit it
        // (128-row tiles at multiples of 128 rows are covered by g->NP by construction)
        int max_tap = 0;
        for (int t = 0; t < g->ntaps; ++t) max_tap = g->tap_off[t] > max_tap ? g->tap_off[t] : max_tap;
        const long bound = ms2 == 2 ? g->NP : g->NP256;
        for (long m0 = (long)full * per_round * 256; m0 < M && ms2 != 0; m0 += 64 * ms2) {
          const long m1 = (m0 + 64 * ms2 < M ? m0 + 64 * ms2 : M) - 1;
          if (bd_row_pin_host(g, m1) - bd_row_pin_host(g, m0) + max_tap + 1 > bound) ms2 = 0;
        }
      }
      if (ms2 != 0) {
        n_big = full * per_round * nt;
        m_split = full * per_round * 256;
        grid = n_big + (int)((rem + 64 * ms2 - 1) / (64 * ms2)) * nt;
      }
    }
  }
  hipStream_t s = (hipStream_t)stream;
#define BD_LAUNCH5(GA_, AB_, DM_, MS_, RD_, M2_)                                                  \
  do {                                                                                           \
    static bool attr = false;                                                                    \
    if (!attr) {                                                                                 \
      (void)hipFuncSetAttribute(                                                                 \
          reinterpret_cast<const void*>(&conv_igemm_bd_kernel<GA_, AB_, DM_, MS_, RD_, M2_>),    \
          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                               \
      attr = true;                                                                               \
    }                                                                                            \
    hipLaunchKernelGGL((conv_igemm_bd_kernel<GA_, AB_, DM_, MS_, RD_, M2_>), dim3(grid),         \
                       dim3(BD_THREADS), lds, s, *g, (const bf16_t*)in,                          \
                       (const unsigned char*)wfrag, (bf16_t*)out, stats, (const bf16_t*)res_grad, \
                       (const bf16_t*)res_act, accumulate, mt, la, g_bd_dense_key,               \
                       (const bf16_t*)red_y, red_coef, (const bf16_t*)red_y2, red_stats,         \
                       red_stats2, g_bd_prof, g_bd_stagger * (g->ntaps > 1 ? g->ntaps : 0),      \
                       n_big, m_split, BB);                                                      \
  } while (0)
#define BD_LAUNCH4(GA_, AB_, DM_, MS_, RD_)                                                       \
  do {                                                                                           \
    if (AB_ == 0 && !GA_ && DM_ && MS_ == 4 && ms2 == 2) BD_LAUNCH5(false, 0, true, 4, RD_, 2);  \
    else if (AB_ == 0 && !GA_ && DM_ && MS_ == 4 && ms2 == 3) BD_LAUNCH5(false, 0, true, 4, RD_, 3); \
    else BD_LAUNCH5(GA_, AB_, DM_, MS_, RD_, 0);                                                 \
  } while (0)
#define BD_LAUNCH3(GA_, AB_, DM_, MS_)                                                            \
  do {                                                                                           \
    if ((AB_ != 0 && AB_ != 128 && AB_ != 256) || GA_ || red == 0) {                                           \
      if (red != 0) return IIC_ERR_UNSUPPORTED;                                                  \
      BD_LAUNCH4(GA_, AB_, DM_, MS_, 0);                                                         \
    } else if (red == 1) BD_LAUNCH4(false, (AB_ == 128 || AB_ == 256 ? AB_ : 0), DM_, MS_, 1);                 \
    else BD_LAUNCH4(false, (AB_ == 128 || AB_ == 256 ? AB_ : 0), DM_, MS_, 2);                                 \
  } while (0)
#define BD_LAUNCH2(GA_, AB_, DM_)                                                                 \
  do {                                                                                           \
    if (ms == 4) BD_LAUNCH3(GA_, AB_, DM_, 4); else BD_LAUNCH3(GA_, AB_, DM_, 2);                \
  } while (0)
#define BD_LAUNCH(GA_, AB_)                                                                       \
  do {                                                                                           \
    if (g_bd_dma) BD_LAUNCH2(GA_, AB_, true); else BD_LAUNCH2(GA_, AB_, false);                  \
  } while (0)
  if (g->ntaps == 1) BD_LAUNCH(true, 0);
  else switch (iic_debug_get_ablate()) {
#ifdef IIC_BD_ABLATIONS
    // timing-ablation and phase-stamp instantiations (tools/conv_perf.py --frag-ablate, tools/bd_timeline.py): NOT in the
    // product library -- `make -C iic_amd/csrc ABL=1` builds them in (13 more instantiations of the kernel)
    case 1: BD_LAUNCH(false, 1); break;
    case 2: BD_LAUNCH(false, 2); break;
    case 3: BD_LAUNCH(false, 3); break;
    case 4: BD_LAUNCH(false, 4); break;
    case 7: BD_LAUNCH(false, 7); break;
    case 15: BD_LAUNCH(false, 15); break;
    case 16: BD_LAUNCH(false, 16); break;
    case 31: BD_LAUNCH(false, 31); break;
    case 32: BD_LAUNCH(false, 32); break;
    case 64: BD_LAUNCH(false, 64); break;
    case 96: BD_LAUNCH(false, 96); break;
    case 128: BD_LAUNCH(false, 128); break;
    case 256: BD_LAUNCH(false, 256); break;
    case 512: BD_LAUNCH(false, 512); break;
#endif
    default: BD_LAUNCH(false, 0); break;
  }
  return iic_launch_status();
}

// All weight operands of a network in ONE launch (jobs in device memory, see iic_weight_prep_job): a train step
// re-lays ~70 convolution weights into 1-2 operand layouts each after the optimiser step -- 136 launches of
// 5-7 us per ClusterNet5g step, 290 per ClusterNet6c two-head step, every one in front of its convolution on
// the forward's critical path.  A block finds its job by binary search over the jobs' first-block indices.
__global__ __launch_bounds__(256) void weight_prep_multi_kernel(const iic_weight_prep_job* __restrict__ jobs,
                                                                int njobs) {
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((long long)blockIdx.x >= jobs[mid].first_block) lo = mid; else hi = mid - 1;
  }
  const iic_weight_prep_job j = jobs[lo];
  const float* __restrict__ w = j.w;
  bf16_t* __restrict__ o = reinterpret_cast<bf16_t*>(j.out);
  const int Co = j.Cout, Ci = j.Cin, T = j.T, mode = j.mode;
  const long total = (long)T * Co * Ci;
  const long nblk = (total + 2047) >> 11;            // 8 elements per thread
  const long b = (long)blockIdx.x - j.first_block;
  if (b >= nblk) return;
  for (long i = b * 256 + threadIdx.x; i < total; i += nblk * 256) {
    if (mode <= 1) {         // MFMA B-fragment order (weight_prep_frag_kernel)
      const int Nn = mode ? Ci : Co, Kk = mode ? Co : Ci;
      const int n32 = Nn >> 5, nchunks = Kk >> 6;
      const int e = (int)(i & 7), lane = (int)((i >> 3) & 63), ks = (int)((i >> 9) & 3);
      long r = i >> 11;
      const int jj = (int)(r % n32);
      r /= n32;
      const int kc = (int)(r % nchunks), t = (int)(r / nchunks);
      const int n = jj * 32 + (lane & 31), k = kc * 64 + ks * 16 + (lane >> 5) * 8 + e;
      const int co = mode ? k : n, ci = mode ? n : k;
      o[i] = f32_to_bf16(w[((long)co * Ci + ci) * T + t]);
    } else if (mode == 2) {  // [T][Co][Ci]
      const int ci = (int)(i % Ci);
      const long r = i / Ci;
      const int co = (int)(r % Co), t = (int)(r / Co);
      o[i] = f32_to_bf16(w[((long)co * Ci + ci) * T + t]);
    } else {                 // [T][Ci][Co]
      const int co = (int)(i % Co);
      const long r = i / Co;
      const int ci = (int)(r % Ci), t = (int)(r / Ci);
      o[i] = f32_to_bf16(w[((long)co * Ci + ci) * T + t]);
    }
  }
}

long iic_weight_prep_multi_blocks(int Cout, int Cin, int T) { return ((long)T * Cout * Cin + 2047) >> 11; }

int iic_weight_prep_multi(const iic_weight_prep_job* jobs_dev, int njobs, long total_blocks, void* stream) {
  if (!jobs_dev || njobs <= 0 || total_blocks <= 0 || total_blocks >= (1L << 31)) return IIC_ERR_ARG;
  hipLaunchKernelGGL(weight_prep_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                     jobs_dev, njobs);
  return iic_launch_status();
}

int iic_weight_prep_frag(const float* w_oihw, void* w_frag, int Cout, int Cin, int T, int bwd,
                         void* stream) {
  if (!w_oihw || !w_frag || Cout <= 0 || Cin <= 0 || T <= 0) return IIC_ERR_ARG;
  const int Nn = bwd ? Cin : Cout, Kk = bwd ? Cout : Cin;
  if (Nn % 32 != 0 || Kk % 64 != 0) return IIC_ERR_UNSUPPORTED;
  const long total = (long)T * Cout * Cin;
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(weight_prep_frag_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                     w_oihw, (bf16_t*)w_frag, Cout, Cin, T, bwd ? 1 : 0);
  return iic_launch_status();
}

}  // extern "C"

// Go / no-go probe (round 5): Winograd F(2x2, 3x3) for the stride-1 3x3 convolutions of
// ClusterNet5g's layers 2-4 (/root/reference/code/archs/cluster/residual.py:4-7,19,22), as ONE fused gfx950
// kernel -- input transform, 16 Winograd-domain GEMMs, output transform, BatchNorm statistics.  NOT part of
// the product library (libiic_probe.so; tools/winograd_probe.py drives it against conv_igemm_bd / _pw).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A       per 4 x 4 input tile d -> 2 x 2 output tile Y
//
// Design notes (what a CUDA-shaped port would not do):
//   * The input transform runs on the MATRIX pipe, not the VALU.  bf16 has no packed add on gfx950, so a VALU
//     transform costs ~14 VALU issues per main MFMA (unpack, two add stages, pack) and would bound the kernel.
//     Instead a tile's 16 pixels x 16 channels are read with ONE ds_read_b64_tr_b16 (the transposing LDS read:
//     lane = channel, 4 values = the 4 pixels of a tile row, 16-lane group = tile row) -- exactly the A operand
//     of v_mfma_f32_16x16x16_bf16 with k = the tile's 16 pixels -- and multiplied by the constant 16 x 16
//     Kronecker matrix B^T (x) B^T (entries 0 / +-1: exact): D[channel][position] in fp32, rounded ONCE to bf16
//     (v_cvt_pk_bf16_f32) and written to LDS as the main MFMAs' A operand.  5 issue slots per tile and k-step
//     instead of 64; the price is 16 matrix-pipe cycles per tile and k-step beside the 16 of the main product.
//   * Accumulators bound the tile: 16 positions x (tiles x couts) fp32 must live in registers, 64 K of them per
//     CU => 64 tiles x 64 couts per workgroup, ONE wave per SIMD with 256 accumulator registers.  Wave w owns
//     Winograd row xi = w (4 positions x 64 tiles x 64 couts): 8 A reads + 8 B loads per 16 MFMAs.
//   * Weights (G g G^T, transformed in fp32 from the master weights, rounded once to bf16) are laid out by the
//     host in MFMA B-fragment order per position and streamed L2 -> registers one k-step ahead.
//   * Raw input patch: LDS-DMA (global_load_lds_dwordx4) in 32-channel chunks, double-buffered; the 16-byte slot
//     pair of a pixel is XOR-ed with its image-row parity on the DMA's source side so that the two tile rows a
//     half-wave of the transposing read touches land in different banks.
//   * Output transform: nu-reduction in registers, xi-reduction across the four waves through LDS (fp32).
#include "../common.h"
#include "../conv_tile.h"

#define WN_T 64                       // tiles per workgroup
#define WN_CO 64                      // couts per workgroup
#define WN_VPITCH 1296                // bytes per (position, k-half) row of V: (64 + 15) slots x 16 B, padded
#define WN_VBYTES (32 * WN_VPITCH)    // one V buffer: 16 positions x 2 k-halves
#define WN_ZBYTES (4 * 2 * 64 * 64 * 4)
#define WN_MAXDMA 10                  // 1-KB DMA blocks per wave and 32-channel chunk (raw buffer <= 40 KB)

struct wino_geom {
  int N, H, W, Hp, Wp, Cin, Cout, TH, TW, tiles_img, Mtiles, in_pixels, rawb;   // rawb: bytes per raw buffer
  iic_mdiv d_tiles_img, d_TW, d_Wp, d_Hp;
};

typedef s16x4 __attribute__((address_space(3))) * wn_lds_s16x4_ptr;
typedef __attribute__((ext_vector_type(2))) unsigned int wn_u32x2;

__device__ __forceinline__ s16x4 wn_tr_read(uint32_t addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((wn_lds_s16x4_ptr)(uintptr_t)addr);
}

// B^T of F(2,3): rows xi, columns r
__device__ __forceinline__ int wn_bt(int xi, int r) {
  // [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]
  const int tab = (xi == 0) ? ((r == 0) - (r == 2)) : (xi == 1) ? ((r == 1) + (r == 2))
                : (xi == 2) ? ((r == 2) - (r == 1)) : ((r == 1) - (r == 3));
  return tab;
}

// ABL (timing ablations, results WRONG): 1 no input transform, 2 no main MFMAs, 4 no stage-2 epilogue.
// DBG: workgroup 0 copies its LDS image (after the first k-step's transform), its accumulators and Z to `dbg`.
template <bool STATS, int ABL, bool DBG, bool PROF = false>
__global__ __launch_bounds__(256, 1) void wino_fwd_kernel(const wino_geom g, const bf16_t* __restrict__ in,
                                                          const unsigned char* __restrict__ ufrag,
                                                          bf16_t* __restrict__ out, float* __restrict__ stats,
                                                          unsigned char* __restrict__ dbg) {
  constexpr int abl = ABL;
  // PROF: wave 0 sums s_memtime deltas per phase into dbg[blockIdx][16] (u64): 0 set-up, 1 prologue, then per k-step
  // 2 top wait + barrier, 3 issue (weights, DMA, A reads, transposing reads) + A latency, 4 main MFMAs, 5 transform;
  // 6 stage 1, 7 stage 2, 8 whole kernel, 9 k-steps
  unsigned long long pt[10];
  unsigned long long tp = 0;
  if (PROF) {
#pragma unroll
    for (int i = 0; i < 10; ++i) pt[i] = 0;
    tp = __builtin_readcyclecounter();
    pt[8] = tp;
  }
  auto stamp = [&](int slot) {
    if (PROF) {
      const unsigned long long now = __builtin_readcyclecounter();
      pt[slot] += now - tp;
      tp = now;
    }
  };
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // LDS map: [raw 0][raw 1][V 0][V 1][tables]; the epilogue's Z (128 KB) overlays raw + V
  unsigned char* raw0 = smem;
  unsigned char* v0 = smem + 2 * g.rawb;
  const int data_bytes = 2 * g.rawb + 2 * WN_VBYTES;
  const int tab_off = data_bytes > WN_ZBYTES ? data_bytes : WN_ZBYTES;
  int* s_tpin = reinterpret_cast<int*>(smem + tab_off);      // [64] top-left input pixel of the tile's 4 x 4 patch
  int* s_tout = s_tpin + 64;                                 // [64] output pixel (a = b = 0)
  int* s_tflag = s_tout + 64;                                // [64] 1 valid | 2 row a = 1 exists | 4 column b = 1 exists
  float* s_red = reinterpret_cast<float*>(s_tflag + 64);     // [4 waves][2][64]
  unsigned char* s_ypar = reinterpret_cast<unsigned char*>(s_red + 4 * 2 * 64);   // [npix] image-row parity per patch pixel

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g5 = lane >> 5;

  const int nt = g.Cout / WN_CO;
  const int tix = xcd_tile_index((int)blockIdx.x, (int)gridDim.x);
  const int mtile = tix / nt, ntile = tix - mtile * nt;
  const int m0 = mtile * WN_T, n0 = ntile * WN_CO;

  if (tid < 64) {
    int m = m0 + tid;
    const bool valid = m < g.Mtiles;
    if (!valid) m = g.Mtiles - 1;
    const int n = iic_mdivide(m, g.d_tiles_img);
    const int rem = m - n * g.tiles_img;
    const int ti = iic_mdivide(rem, g.d_TW), tj = rem - ti * g.TW;
    s_tpin[tid] = (n * g.Hp + 2 * ti) * g.Wp + 2 * tj;
    s_tout[tid] = (n * g.Hp + 2 * ti + 1) * g.Wp + 2 * tj + 1;
    s_tflag[tid] = (valid ? 1 : 0) | ((2 * ti + 1 < g.H) ? 2 : 0) | ((2 * tj + 1 < g.W) ? 4 : 0);
  }
  __syncthreads();
  const int p_lo = s_tpin[0];
  const int npix = s_tpin[63] + 3 * g.Wp + 3 - p_lo + 1;
  for (int r = tid; r < npix; r += 256) {
    const int P = p_lo + r;
    const int row = iic_mdivide(P, g.d_Wp);             // global padded row index
    const int img = iic_mdivide(row, g.d_Hp);
    s_ypar[r] = (unsigned char)((row - img * g.Hp) & 1);
  }
  __syncthreads();

  // ---- raw patch DMA: 32 channels (64 B per pixel = 4 slots) of chunk c into raw buffer `buf` ----
  // Inline asm, so that hipcc's vmcnt bookkeeping does not see it (a visible LDS-DMA made it wait for every load in
  // flight -- the next k-step's weights included -- in front of the next LDS read).  Lane's source offsets are
  // chunk-independent: computed once (WN_MAXDMA 1-KB blocks per wave and chunk).
  const int nblk = (npix * 4 + 63) >> 6;
  const int njw = (nblk - wave + 3) >> 2;            // this wave's blocks: wave, wave + 4, ...
  uint32_t doff[WN_MAXDMA];
#pragma unroll
  for (int j = 0; j < WN_MAXDMA; ++j) {
    const int q = (wave + 4 * j) * 64 + lane;
    int r = q >> 2;
    r = r < npix ? r : npix - 1;
    const int ls = (q & 3) ^ (2 * (int)s_ypar[r]);
    long p = (long)p_lo + r;
    p = p < g.in_pixels ? p : g.in_pixels - 1;
    doff[j] = (uint32_t)((p * g.Cin + ls * 8) * 2);
  }
  const uint32_t raw_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)raw0;
  auto dma_chunk = [&](int c, int buf) {
    const unsigned char* src = reinterpret_cast<const unsigned char*>(in) + c * 64;
#pragma unroll
    for (int j = 0; j < WN_MAXDMA; ++j) {
      if (j < njw) {
        const uint32_t dst = raw_lds + buf * g.rawb + (wave + 4 * j) * 1024;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :: "v"(doff[j]), "s"(src), "s"(dst) : "memory");
      }
    }
  };

  // ---- transform constants ----
  // transposing read: lane (q4 = tile row, x = pixel of the row, seg = 8-byte channel segment)
  const int q4 = lane >> 4, i16 = lane & 15, tx = i16 >> 2, seg = i16 & 3;
  const uint32_t lds0 = raw_lds;
  uint32_t LO[2];
#pragma unroll
  for (int kp = 0; kp < 2; ++kp)
    LO[kp] = lds0 + (q4 * g.Wp + tx) * 64 + (((kp ^ (q4 & 1)) * 2 + (seg >> 1)) * 16) + (seg & 1) * 8;
  // Kronecker operand: B[k = 4 r + x][j = position 4 xi + nu], lane (j = lane & 15, r = lane >> 4), element x
  s16x4 kron;
  {
    const int pos = lane & 15, xi = pos >> 2, nu = pos & 3, r = lane >> 4;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const int v = wn_bt(xi, r) * wn_bt(nu, x);
      kron[x] = (short)(v > 0 ? 0x3F80 : (v < 0 ? (short)0xBF80 : 0));
    }
  }
  // this wave's 16 tiles: byte offset of the tile's top-left pixel in a raw buffer
  int sb[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) sb[u] = __builtin_amdgcn_readfirstlane((s_tpin[wave * 16 + u] - p_lo) * 64);
  // V write: lane (pos = lane & 15, gq = lane >> 4 -> channels 4 gq .. 4 gq + 3) of tile wave * 16 + u
  const uint32_t vbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)v0;
  const uint32_t VW = vbase + ((lane & 15) * 2 + (q4 >> 1)) * WN_VPITCH + (wave * 16 + (lane & 15)) * 16 + (q4 & 1) * 8;
  // A-fragment read: position 4 wave + nu, tile block mb: row (pos, g5), slot mb * 32 + l31 + pos
  const uint32_t AR = vbase + ((wave * 4) * 2 + g5) * WN_VPITCH + (l31 + wave * 4) * 16;

  // The transform of this wave's 16 tiles, in two parts so that the LDS latency of the transposing reads hides under
  // the main MFMAs issued between them:
  //   tr_issue: 16 ds_read_b64_tr_b16 (compiler-visible: hipcc waits for them in front of the asm block);
  //   tr_run:   ONE asm block, software-pipelined by hand -- per tile  MFMA(u) | cvt, cvt, ds_write of tile u - 2.
  // Why asm: the 256 accumulators of the main product own every AGPR, and hipcc gives ALL MFMAs of a function
  // AGPR destinations once it needs any -- it evicted an accumulator to VGPRs around every transform (32 copies
  // per k-step), serialised the 16 small MFMAs on one destination and read each result back with 4
  // v_accvgpr_read.  In asm the small MFMA writes VGPRs (v[240:251], three rotating sets; C = inline 0).  The
  // hazard hipcc would have padded (gfx950: 8 wait states between a 4-pass MFMA's VGPR result and a VALU read) is
  // kept by construction: 8 instructions, two of them MFMAs of >= 16 cycles each, separate a result from its cvt.
  auto tr_issue = [&](s16x4 (&t)[16], int rawsel, int kp) {
#pragma unroll
    for (int u = 0; u < 16; ++u) t[u] = wn_tr_read(LO[kp] + (uint32_t)(rawsel * g.rawb + sb[u]));
  };
  // register names: D sets v[240:243] (24x), v[244:247], v[248:251]; pack pairs v[252:253], v[254:255]
  auto tr_run = [&](const s16x4 (&t)[16], uint32_t vaddr) {
    asm volatile(
        "v_mfma_f32_16x16x16_bf16 v[240:243], %[t0], %[k], 0\n\t"
        "v_mfma_f32_16x16x16_bf16 v[244:247], %[t1], %[k], 0\n\t"
        "s_nop 7\n\t"
        // tile u: MFMA into set u % 3; convert + store tile u - 2 (set (u - 2) % 3)
        "v_mfma_f32_16x16x16_bf16 v[248:251], %[t2], %[k], 0\n\t"
        "v_cvt_pk_bf16_f32 v252, v240, v241\n\t" "v_cvt_pk_bf16_f32 v253, v242, v243\n\t" "ds_write_b64 %[va], v[252:253] offset:0\n\t"
        "v_mfma_f32_16x16x16_bf16 v[240:243], %[t3], %[k], 0\n\t"
        "v_cvt_pk_bf16_f32 v254, v244, v245\n\t" "v_cvt_pk_bf16_f32 v255, v246, v247\n\t" "ds_write_b64 %[va], v[254:255] offset:16\n\t"
        "v_mfma_f32_16x16x16_bf16 v[244:247], %[t4], %[k], 0\n\t"
        "v_cvt_pk_bf16_f32 v252, v248, v249\n\t" "v_cvt_pk_bf16_f32 v253, v250, v251\n\t" "ds_write_b64 %[va], v[252:253] offset:32\n\t"
        "v_mfma_f32_16x16x16_bf16 v[248:251], %[t5], %[k], 0\n\t"
        "v_cvt_pk_bf16_f32 v254, v240, v241\n\t" "v_cvt_pk_bf16_f32 v255, v242, v243\n\t" "ds_write_b64 %[va], v[254:255] offset:48\n\t"
        "v_mfma_f32_16x16x16_bf16 v[240:243], %[t6], %[k], 0\n\t"
        "v_cvt_pk_bf16_f32 v252, v244, v245\n\t" "v_cvt_pk_bf16_f32 v253, v246, v247\n\t" "ds_write_b64 %[va], v[252:253] offset:64\n\t"
        "v_mfma_f32_16x16x16_bf16 v[244:247], %[t7], %[k], 0\n\t"
        "v_cvt_pk_bf16_f32 v254, v248, v249\n\t" "v_cvt_pk_bf16_f32 v255, v250, v251\n\t" "ds_write_b64 %[va], v[254:255] offset:80\n\t"
        "v_mfma_f32_16x16x16_bf16 v[248:251], %[t8], %[k], 0\n\t"
        "v_cvt_pk_bf16_f32 v252, v240, v241\n\t" "v_cvt_pk_bf16_f32 v253, v242, v243\n\t" "ds_write_b64 %[va], v[252:253] offset:96\n\t"
        "v_mfma_f32_16x16x16_bf16 v[240:243], %[t9], %[k], 0\n\t"
        "v_cvt_pk_bf16_f32 v254, v244, v245\n\t" "v_cvt_pk_bf16_f32 v255, v246, v247\n\t" "ds_write_b64 %[va], v[254:255] offset:112\n\t"
        "v_mfma_f32_16x16x16_bf16 v[244:247], %[t10], %[k], 0\n\t"
        "v_cvt_pk_bf16_f32 v252, v248, v249\n\t" "v_cvt_pk_bf16_f32 v253, v250, v251\n\t" "ds_write_b64 %[va], v[252:253] offset:128\n\t"
        "v_mfma_f32_16x16x16_bf16 v[248:251], %[t11], %[k], 0\n\t"
        "v_cvt_pk_bf16_f32 v254, v240, v241\n\t" "v_cvt_pk_bf16_f32 v255, v242, v243\n\t" "ds_write_b64 %[va], v[254:255] offset:144\n\t"
        "v_mfma_f32_16x16x16_bf16 v[240:243], %[t12], %[k], 0\n\t"
        "v_cvt_pk_bf16_f32 v252, v244, v245\n\t" "v_cvt_pk_bf16_f32 v253, v246, v247\n\t" "ds_write_b64 %[va], v[252:253] offset:160\n\t"
        "v_mfma_f32_16x16x16_bf16 v[244:247], %[t13], %[k], 0\n\t"
        "v_cvt_pk_bf16_f32 v254, v248, v249\n\t" "v_cvt_pk_bf16_f32 v255, v250, v251\n\t" "ds_write_b64 %[va], v[254:255] offset:176\n\t"
        "v_mfma_f32_16x16x16_bf16 v[248:251], %[t14], %[k], 0\n\t"
        "v_cvt_pk_bf16_f32 v252, v240, v241\n\t" "v_cvt_pk_bf16_f32 v253, v242, v243\n\t" "ds_write_b64 %[va], v[252:253] offset:192\n\t"
        "v_mfma_f32_16x16x16_bf16 v[240:243], %[t15], %[k], 0\n\t"
        "v_cvt_pk_bf16_f32 v254, v244, v245\n\t" "v_cvt_pk_bf16_f32 v255, v246, v247\n\t" "ds_write_b64 %[va], v[254:255] offset:208\n\t"
        "s_nop 7\n\t"
        "v_cvt_pk_bf16_f32 v252, v248, v249\n\t" "v_cvt_pk_bf16_f32 v253, v250, v251\n\t" "ds_write_b64 %[va], v[252:253] offset:224\n\t"
        "s_nop 7\n\t" "s_nop 7\n\t"
        "v_cvt_pk_bf16_f32 v254, v240, v241\n\t" "v_cvt_pk_bf16_f32 v255, v242, v243\n\t" "ds_write_b64 %[va], v[254:255] offset:240\n\t"
        :
        : [t0] "v"(t[0]), [t1] "v"(t[1]), [t2] "v"(t[2]), [t3] "v"(t[3]), [t4] "v"(t[4]), [t5] "v"(t[5]), [t6] "v"(t[6]),
          [t7] "v"(t[7]), [t8] "v"(t[8]), [t9] "v"(t[9]), [t10] "v"(t[10]), [t11] "v"(t[11]), [t12] "v"(t[12]),
          [t13] "v"(t[13]), [t14] "v"(t[14]), [t15] "v"(t[15]), [k] "v"(kron), [va] "v"(vaddr)
        : "memory", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252",
          "v253", "v254", "v255");
  };

  // ---- weights: [pos][k-step][Cout / 32][lane][8] bf16 ----
  const int KS = g.Cin >> 4, NB = g.Cout >> 5;
  const unsigned char* ub = ufrag + ((long)(n0 >> 5) << 10) + lane * 16;
  auto bptr = [&](int nu, int ks) { return ub + ((long)((wave * 4 + nu) * KS + ks) * NB << 10); };

  f32x16 acc[4][2][2];
#pragma unroll
  for (int nu = 0; nu < 4; ++nu)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nu][mb][nb][r] = 0.f;

  // Weight fragments one k-step ahead, as inline asm: hipcc must not count them (it does not see the LDS-DMA pieces issued
  // behind them, so its vmcnt(N) in front of the MFMAs -- N = loads it knows to be younger -- made even k-steps wait for
  // the loads they had just issued).  The k-step's top waits vmcnt(0) for everything issued one k-step earlier.
  u32x4 B0[4][2], B1[4][2];
  auto bload = [&](u32x4 (&B)[4][2], int ks) {
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      const unsigned char* p = bptr(nu, ks);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(B[nu][0]) : "v"(p) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(B[nu][1]) : "v"(p) : "memory");
    }
  };
  auto top_wait = [&](u32x4 (&B)[4][2]) {       // everything this wave has in flight: weights of this k-step, DMA pieces, V stores
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                 : "+v"(B[0][0]), "+v"(B[0][1]), "+v"(B[1][0]), "+v"(B[1][1]), "+v"(B[2][0]), "+v"(B[2][1]), "+v"(B[3][0]),
                   "+v"(B[3][1]) :: "memory");
  };
  // A fragments of a k-step: all 8 reads issued back to back (asm: left to itself hipcc issues them two at a time in
  // front of the MFMAs that use them and exposes the LDS latency four times per k-step -- 1 400 cycles per k-step
  // measured for 512 cycles of MFMAs), then the 16 transposing reads of the next transform (compiler-visible), then ONE
  // wait that lets exactly those 16 younger reads stay in flight.
  auto a_issue = [&](bf16x8 (&a)[4][2], int vsel) {
    const uint32_t base = AR + vsel * WN_VBYTES;
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[nu][mb]) : "v"(base), "i"(nu * (2 * WN_VPITCH + 16) + mb * 512));
  };
  auto a_wait16 = [&](bf16x8 (&a)[4][2]) {      // 16 younger LDS reads may still be in flight
    asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[2][0]),
                 "+v"(a[2][1]), "+v"(a[3][0]), "+v"(a[3][1]) :: "memory");
  };
  auto a_wait0 = [&](bf16x8 (&a)[4][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[2][0]),
                 "+v"(a[2][1]), "+v"(a[3][0]), "+v"(a[3][1]) :: "memory");
  };
  auto mma = [&](const bf16x8 (&a)[4][2], u32x4 (&B)[4][2]) {
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          acc[nu][mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[nu][mb], __builtin_bit_cast(bf16x8, B[nu][nb]),
                                                                  acc[nu][mb][nb], 0, 0, 0);
  };

  // ---- prologue ----
  stamp(0);
  const int NCH = g.Cin >> 5;
  s16x4 t[16];
  bf16x8 a[4][2];
  dma_chunk(0, 0);
  bload(B0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  tr_issue(t, 0, 0);
  tr_run(t, VW);
  stamp(1);

  // The last chunk's look-ahead work (DMA of chunk NCH, transform of k-step 2 NCH, weights of k-step 2 NCH) is
  // done on clamped indices instead of being branched around: it lands in buffers nobody reads again.
  for (int c = 0; c < NCH; ++c) {
    const int cn = c + 1 < NCH ? c + 1 : c;
    // ---- k-step 2c: V[0], B0 ----  (lgkmcnt: the transform's ds_writes are invisible to hipcc)
    top_wait(B0);
    __syncthreads();
    stamp(2);
    if (DBG && c == 0 && blockIdx.x == 0) {       // checkpoint 1: raw[0], V[0], tables as they sit in LDS
      const int total = tab_off + 64 * 3 * 4 + 4 * 2 * 64 * 4 + npix;
      for (int i = tid; i < (total + 3) / 4; i += 256)
        reinterpret_cast<uint32_t*>(dbg)[i] = reinterpret_cast<const uint32_t*>(smem)[i];
    }
    bload(B1, 2 * c + 1);
    dma_chunk(cn, (c + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
    if (!(abl & 2)) a_issue(a, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (!(abl & 1)) tr_issue(t, c & 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (!(abl & 2)) { if (abl & 1) a_wait0(a); else a_wait16(a); }
    stamp(3);
    if (!(abl & 2)) mma(a, B0);
    __builtin_amdgcn_sched_barrier(0);
    stamp(4);
    if (!(abl & 1)) tr_run(t, VW + WN_VBYTES);
    __builtin_amdgcn_sched_barrier(0);
    stamp(5);
    // ---- k-step 2c + 1: V[1], B1 ----
    top_wait(B1);
    __syncthreads();
    stamp(2);
    bload(B0, 2 * cn);
    __builtin_amdgcn_sched_barrier(0);
    if (!(abl & 2)) a_issue(a, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (!(abl & 1)) tr_issue(t, (c + 1) & 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (!(abl & 2)) { if (abl & 1) a_wait0(a); else a_wait16(a); }
    stamp(3);
    if (!(abl & 2)) mma(a, B1);
    __builtin_amdgcn_sched_barrier(0);
    stamp(4);
    if (!(abl & 1)) tr_run(t, VW);
    __builtin_amdgcn_sched_barrier(0);
    stamp(5);
    if (PROF) pt[9] += 2;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();          // every wave is done with raw / V: the region becomes Z
  stamp(2);

  if (DBG && blockIdx.x == 0) {                   // checkpoint 2: the accumulators, [wave][lane][nu][mb][nb][16]
    float* d2 = reinterpret_cast<float*>(dbg + 256 * 1024) + (long)tid * 256;
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r) d2[((nu * 2 + mb) * 2 + nb) * 16 + r] = acc[nu][mb][nb][r];
  }

  // ---- output transform, stage 1: nu-reduction in registers -> Z[xi][b][tile][cout] (fp32, LDS) ----
  float* Z = reinterpret_cast<float*>(smem);
  {
    // one lane base + compile-time offsets: every store is ds_write_b32 base, value offset:imm
    float* zb = Z + (wave * 2 * 64 + 4 * g5) * 64 + l31;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float m0v = acc[0][mb][nb][r], m1v = acc[1][mb][nb][r], m2v = acc[2][mb][nb][r], m3v = acc[3][mb][nb][r];
          const int off = (mb * 32 + (r & 3) + 8 * (r >> 2)) * 64 + nb * 32;      // row mfma32_row(r, lane) - 4 g5, column nb * 32
          zb[off] = m0v + m1v + m2v;
          zb[64 * 64 + off] = m1v - m2v - m3v;
        }
  }
  __syncthreads();
  stamp(6);
  if (DBG && blockIdx.x == 0) {                   // checkpoint 3: Z
    for (int i = tid; i < WN_ZBYTES / 4; i += 256)
      reinterpret_cast<uint32_t*>(dbg + 1024 * 1024)[i] = reinterpret_cast<const uint32_t*>(smem)[i];
  }
  if (abl & 4) return;

  // ---- stage 2: xi-reduction, 2 x 2 pixels x 8 couts per item; stores + BatchNorm statistics ----
  float s8[8], ss8[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s8[i] = ss8[i] = 0.f;
  const int cg = tid & 7;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int t = (tid >> 3) + it * 32;
    float z[4][2][8];
#pragma unroll
    for (int xi = 0; xi < 4; ++xi)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const f32x4* p = reinterpret_cast<const f32x4*>(Z + ((xi * 2 + b) * 64 + t) * 64 + cg * 8);
        const f32x4 lo = p[0], hi = p[1];
#pragma unroll
        for (int i = 0; i < 4; ++i) { z[xi][b][i] = lo[i]; z[xi][b][4 + i] = hi[i]; }
      }
    const int fl = s_tflag[t];
    const long po = s_tout[t];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const bool ok = (fl & 1) && (a == 0 || (fl & 2)) && (b == 0 || (fl & 4));
        if (!ok) continue;
        float y[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
          y[i] = a == 0 ? (z[0][b][i] + z[1][b][i] + z[2][b][i]) : (z[1][b][i] - z[2][b][i] - z[3][b][i]);
        if (STATS) {
#pragma unroll
          for (int i = 0; i < 8; ++i) { s8[i] += y[i]; ss8[i] += y[i] * y[i]; }
        }
        const uint4 v = make_uint4(pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]), pack_bf16x2(y[4], y[5]),
                                   pack_bf16x2(y[6], y[7]));
        *reinterpret_cast<uint4*>(out + ((po + a * g.Wp + b) * g.Cout + n0 + cg * 8)) = v;
      }
  }
  if (STATS && stats) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) {
        s8[i] += __shfl_xor(s8[i], o, 64);
        ss8[i] += __shfl_xor(ss8[i], o, 64);
      }
    if (lane < 8) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        s_red[(wave * 2 + 0) * 64 + cg * 8 + i] = s8[i];
        s_red[(wave * 2 + 1) * 64 + cg * 8 + i] = ss8[i];
      }
    }
    __syncthreads();
    if (tid < 64) {
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { a0 += s_red[(w * 2 + 0) * 64 + tid]; a1 += s_red[(w * 2 + 1) * 64 + tid]; }
      const int stripe = blockIdx.x % IIC_STAT_STRIPES;
      iic_stat_add(stats, stripe, g.Cout, n0 + tid, 0, a0);
      iic_stat_add(stats, stripe, g.Cout, n0 + tid, 1, a1);
    }
  }
  if (PROF) {
    stamp(7);
    if (tid == 0) {
      unsigned long long* q = reinterpret_cast<unsigned long long*>(dbg) + (long)blockIdx.x * 16;
      pt[8] = __builtin_readcyclecounter() - pt[8];
#pragma unroll
      for (int i = 0; i < 10; ++i) q[i] = pt[i];
      q[10] = __builtin_amdgcn_s_memrealtime();
    }
  }
}

// G g G^T in fp32 from the fp32 OIHW master weights, rounded once to bf16, in MFMA B-fragment order:
//   out[pos][ks][n / 32][lane][e] = U[n = (n/32)*32 + (lane & 31)][k = ks*16 + (lane >> 5)*8 + e][pos]
// mode 0: forward operand (GEMM N = Cout, K = Cin); mode 1: backward-data operand (N = Cin, K = Cout, taps rotated by 180 degrees).
__global__ void wino_weight_prep_kernel(const float* __restrict__ w, bf16_t* __restrict__ o, int Co, int Ci, int mode) {
  const int Nn = mode ? Ci : Co, Kk = mode ? Co : Ci;
  const int NB = Nn >> 5, KS = Kk >> 4;
  const long total = 16L * Co * Ci;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
    long r = i >> 9;
    const int nb = (int)(r % NB);
    r /= NB;
    const int ks = (int)(r % KS), pos = (int)(r / KS);
    const int n = nb * 32 + (lane & 31), k = ks * 16 + (lane >> 5) * 8 + e;
    const int co = mode ? k : n, ci = mode ? n : k;
    const float* g = w + ((long)co * Ci + ci) * 9;
    float gg[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) gg[a][b] = mode ? g[(2 - a) * 3 + (2 - b)] : g[a * 3 + b];
    const int xi = pos >> 2, nu = pos & 3;
    // row xi of G: [1,0,0], [.5,.5,.5], [.5,-.5,.5], [0,0,1]
    float t[3];
#pragma unroll
    for (int b = 0; b < 3; ++b)
      t[b] = xi == 0 ? gg[0][b] : xi == 1 ? 0.5f * (gg[0][b] + gg[1][b] + gg[2][b])
           : xi == 2 ? 0.5f * (gg[0][b] - gg[1][b] + gg[2][b]) : gg[2][b];
    const float u = nu == 0 ? t[0] : nu == 1 ? 0.5f * (t[0] + t[1] + t[2]) : nu == 2 ? 0.5f * (t[0] - t[1] + t[2]) : t[2];
    o[i] = f32_to_bf16(u);
  }
}

extern "C" {

static int wn_make_geom(wino_geom* g, int N, int H, int W, int Cin, int Cout) {
  if (N <= 0 || H < 2 || W < 2 || Cin % 32 != 0 || Cout % WN_CO != 0) return IIC_ERR_UNSUPPORTED;
  g->N = N; g->H = H; g->W = W; g->Hp = H + 2; g->Wp = W + 2; g->Cin = Cin; g->Cout = Cout;
  g->TH = (H + 1) / 2; g->TW = (W + 1) / 2;
  g->tiles_img = g->TH * g->TW;
  const long mt = (long)N * g->tiles_img;
  const long px = (long)N * g->Hp * g->Wp;
  if (mt >= (1L << 31) || px >= (1L << 31)) return IIC_ERR_UNSUPPORTED;
  g->Mtiles = (int)mt;
  g->in_pixels = (int)px;
  g->d_tiles_img = iic_make_mdiv(g->tiles_img);
  g->d_TW = iic_make_mdiv(g->TW);
  g->d_Wp = iic_make_mdiv(g->Wp);
  g->d_Hp = iic_make_mdiv(g->Hp);
  // widest pixel span of a 64-tile workgroup tile (the first tiles of an image pair see every alignment after
  // tiles_img * 64 / gcd tiles; scan them all: a few thousand iterations)
  long npmax = 0;
  for (long m0 = 0; m0 < mt; m0 += WN_T) {
    long m1 = m0 + WN_T - 1 < mt ? m0 + WN_T - 1 : mt - 1;
    auto pin = [&](long m) {
      const long n = m / g->tiles_img, rem = m % g->tiles_img, ti = rem / g->TW, tj = rem % g->TW;
      return (n * g->Hp + 2 * ti) * g->Wp + 2 * tj;
    };
    const long span = pin(m1) + 3 * g->Wp + 3 - pin(m0) + 1;
    npmax = span > npmax ? span : npmax;
    if (m0 > 4L * g->tiles_img * WN_T) break;       // the pattern repeats after tiles_img workgroup tiles
  }
  g->rawb = (int)(((npmax * 64 + 1023) / 1024) * 1024);
  return IIC_OK;
}

static long wn_lds_bytes(const wino_geom* g, long npmax_pixels) {
  const long data = 2L * g->rawb + 2L * WN_VBYTES;
  const long tab_off = data > WN_ZBYTES ? data : WN_ZBYTES;
  return tab_off + 64 * 3 * 4 + 4 * 2 * 64 * 4 + ((npmax_pixels + 15) & ~15L);
}

/* LDS bytes of a launch (<= 160 KB or the geometry is unsupported); < 0: unsupported. */
long iic_probe_wino_lds_bytes(int N, int H, int W, int Cin, int Cout) {
  wino_geom g;
  if (wn_make_geom(&g, N, H, W, Cin, Cout) != IIC_OK) return -1;
  return wn_lds_bytes(&g, g.rawb / 64);
}

/* Transformed weights in fragment order: 16 * Cout * Cin bf16.  bwd != 0: the backward-data operand. */
int iic_probe_wino_weight_prep(const float* w_oihw, void* ufrag, int Cout, int Cin, int bwd, void* stream) {
  if (!w_oihw || !ufrag || Cout % 32 != 0 || Cin % 32 != 0) return IIC_ERR_ARG;
  const long total = 16L * Cout * Cin;
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(wino_weight_prep_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_oihw, (bf16_t*)ufrag, Cout, Cin,
                     bwd ? 1 : 0);
  return iic_launch_status();
}

/* Stride-1 pad-1 3x3 convolution, PT tensors (bf16 [N][H+2][W+2][C], zero border): in -> out interior, and the
 * BatchNorm statistics of the output into `stats` (exact accumulators, may be null).
 * dbg: null, or >= 1.5 MB of device memory for workgroup 0's checkpoints (LDS image at 0, accumulators at 256 KB,
 * Z at 1 MB).  abl: timing ablations (results WRONG): 1 no input transform, 2 no main MFMAs, 4 no stage-2 epilogue. */
int iic_probe_wino_fwd(const void* in, const void* ufrag, void* out, float* stats, int N, int H, int W, int Cin, int Cout,
                       void* dbg, int abl, void* stream) {
  if (!in || !ufrag || !out) return IIC_ERR_ARG;
  wino_geom g;
  const int rc = wn_make_geom(&g, N, H, W, Cin, Cout);
  if (rc != IIC_OK) return rc;
  const long lds = wn_lds_bytes(&g, g.rawb / 64);
  if (lds > 160 * 1024 || g.rawb > WN_MAXDMA * 4 * 1024) return IIC_ERR_UNSUPPORTED;
  if ((long)g.in_pixels * Cin * 2 >= (1L << 32)) return IIC_ERR_UNSUPPORTED;      // 32-bit DMA source offsets
  const int grid = ((g.Mtiles + WN_T - 1) / WN_T) * (Cout / WN_CO);
  hipStream_t s = (hipStream_t)stream;
#define WN_LAUNCH(ST_, AB_, DB_)                                                                                       \
  do {                                                                                                               \
    static bool attr = false;                                                                                        \
    if (!attr) {                                                                                                     \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_fwd_kernel<ST_, AB_, DB_>),                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                             \
      attr = true;                                                                                                   \
    }                                                                                                                \
    hipLaunchKernelGGL((wino_fwd_kernel<ST_, AB_, DB_>), dim3(grid), dim3(256), lds, s, g, (const bf16_t*)in,         \
                       (const unsigned char*)ufrag, (bf16_t*)out, stats, (unsigned char*)dbg);                       \
  } while (0)
  if (dbg && abl == 8) {            // phase profile: dbg = grid x 16 u64
    if (!stats) return IIC_ERR_ARG;
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_fwd_kernel<true, 0, false, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr = true;
    }
    hipLaunchKernelGGL((wino_fwd_kernel<true, 0, false, true>), dim3(grid), dim3(256), lds, s, g, (const bf16_t*)in,
                       (const unsigned char*)ufrag, (bf16_t*)out, stats, (unsigned char*)dbg);
  } else if (dbg) { if (!stats || abl) return IIC_ERR_ARG; WN_LAUNCH(true, 0, true); }
  else if (!stats) { if (abl) return IIC_ERR_ARG; WN_LAUNCH(false, 0, false); }
  else switch (abl) {
    case 0: WN_LAUNCH(true, 0, false); break;
    case 1: WN_LAUNCH(true, 1, false); break;
    case 2: WN_LAUNCH(true, 2, false); break;
    case 3: WN_LAUNCH(true, 3, false); break;
    case 4: WN_LAUNCH(true, 4, false); break;
    case 7: WN_LAUNCH(true, 7, false); break;
    default: return IIC_ERR_ARG;
  }
  return iic_launch_status();
}

}  // extern "C"

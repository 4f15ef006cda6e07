// Third generation of the weights-direct implicit-GEMM convolution (round 3): the multi-tap,
// Cin % 64 == 0, Cout % 128 == 0 launches of conv_igemm_bd.hip -- i.e. the stride-1 3x3 convolutions of
// layers 2-4, forward and backward-data
//   /root/reference/code/archs/cluster/residual.py:4-7,19,22   (conv3x3, BasicBlock.conv1 / conv2)
//   /root/reference/code/archs/cluster/vgg.py:24-26            (VGG-style stages, dilated 3x3)
// Same contract, same B-fragment weight operand, same epilogue (fused BatchNorm statistics, residual
// gradient / pre-mask, fused BatchNorm-backward reduction) -- what changes is how the K loop is fed.
// tools/bd_timeline.py (per-workgroup s_memtime stamps) on the round-2 kernel at 660 images:
//   a 256 x 128 tile spends 13 / 12 / 19 % of its cycles (layer 2 / 3 / 4) stalled at chunk boundaries
//   (barrier -> LDS-DMA of the next 64-channel patch -> vmcnt(0) -> barrier, 4.4-5.6 k cycles each), and
//   tools/mfma_feed.py: [4 LDS reads][8 MFMAs][2 loads] blocks leave the matrix pipe idle while a wave
//   works through its feeder block (interleaving them: +5...14 % at these loop lengths).
// Here
//   * the patch is double-buffered in HALF chunks (32 channels, 64-byte rows: the same LDS footprint as
//     one 64-channel patch): while the waves run the taps of one half, the other buffer is filled by
//     LDS-DMA pieces issued one per tap iteration, so a half boundary costs one barrier and no memory
//     wait.  K order: (chunk, half, tap, 2 k-steps);
//   * B fragments and the DMA are issued as inline asm, invisible to hipcc's vmcnt bookkeeping, behind
//     hand-counted waits (exactly 8 B loads in flight, the one needed is the oldest; a DMA piece in the
//     window only makes vmcnt(7) over-wait by one entry: safe); hipcc would otherwise drain the queue
//     at the loop header and in front of the first LDS read after every DMA;
//   * one LDS read between consecutive MFMAs (ns-major MFMA order so that the first B fragment of a
//     k-step is free, and re-filled, after MS MFMAs);
//   * no swizzle-key table and no integer divisions in the set-up: rows are located with host-computed
//     multiply-shift reciprocals (bd2_div).
// LDS image of a half-chunk patch: row R (pixel p_lo + R) at R * 64 bytes, its four 16-byte slots
// (8 channels each) XOR-ed with key = (D >> 2) & 3, D = the dense pixel count of conv_igemm_bd.hip --
// the 16 lanes of a ds_read_b128 group (16 consecutive D) then cover all 64 banks except where two
// pixels 2 apart straddle an image-row end (2-way).  The swizzle is applied on the DMA's SOURCE address.
#include <type_traits>

#include "../common.h"
#include "../conv_tile.h"
#include "../../../include/iic_hip.h"

#define B2_BN 128
#define B2_THREADS 256
#define B2_PROF_SLOTS 16

struct bd2_div {      // floor(n / d) for 0 <= n < 2^31:  (n * mul) >> sh  (64-bit product)
  unsigned mul;
  int sh;
};
static inline bd2_div bd2_make_div(int d) {
  bd2_div r;
  int l = 0;
  while ((1L << l) < d) ++l;
  r.sh = 31 + l;
  r.mul = (unsigned)(((1ULL << r.sh) + (unsigned long long)d - 1) / (unsigned long long)d);   // ceil(2^sh / d) <= 2^32
  if (d == 1) { r.mul = 1u << 31; r.sh = 31; }
  return r;
}
__device__ __forceinline__ int bd2_divide(int n, const bd2_div& d) {
  return (int)(((unsigned long long)(unsigned)n * d.mul) >> d.sh);
}

struct bd2_args {
  bd2_div d_rows;      // by rows per image (g.MP or the plane)
  bd2_div d_mx;        // by g.MX
  bd2_div d_wp;        // by g.in_Wp
  int jskip;           // dense-count skip per image row (0: key from the raw pixel index)
  int max_tap;         // largest tap offset
  int buf_bytes;       // one half-chunk patch buffer (multiple of 1024)
  int lds_a_bytes;     // patch buffers / epilogue staging region
  int num_mtiles;
  int pieces;          // DMA pieces (1 KB) per wave and tap iteration (mode 1)
  int mode;            // 0: next half's patch as one burst at the start of a half, 1: one piece per iteration
};

__device__ __forceinline__ void b2_bload(u32x4& d, const unsigned char* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void b2_bwait(u32x4& d) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(d) : "i"(N) : "memory");
}
// one LDS-DMA piece: 64 lanes x 16 bytes from per-lane global addresses to lds_dst + lane * 16
// (lds_dst wave-uniform; M0 is compiler-reserved: saved and restored inside the statement)
__device__ __forceinline__ void b2_dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

template <int MS, int RED, bool PROF>
__global__ __launch_bounds__(B2_THREADS, 2) void conv_igemm_bd2_kernel(
    const iic_conv_geom g, const bd2_args A, const bf16_t* __restrict__ in,
    const unsigned char* __restrict__ wfrag, bf16_t* __restrict__ out, float* __restrict__ stats,
    const bf16_t* __restrict__ res_grad, const bf16_t* __restrict__ res_act, int accumulate,
    const bf16_t* __restrict__ red_y, const float* __restrict__ red_coef, const bf16_t* __restrict__ red_y2,
    float* __restrict__ red_stats, float* __restrict__ red_stats2, unsigned long long* __restrict__ prof) {
  constexpr int CLD = B2_BN + 8;
  constexpr int BM = MS * 64, WR = MS * 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* sA = smem_raw;                                       // [2][buf_bytes]
  int* s_pin = reinterpret_cast<int*>(smem_raw + A.lds_a_bytes);      // [BM]
  int* s_pout = s_pin + BM;                                           // [BM]
  float* s_red = reinterpret_cast<float*>(s_pout + BM);               // [2 wm][2][128]
  bf16_t* sC = reinterpret_cast<bf16_t*>(smem_raw);                   // epilogue staging (aliases the patch)

  unsigned long long t_stamp[5];
  unsigned long long t_bsum = 0, t_b0 = 0;
  int t_nb = 0;
  if (PROF) t_stamp[0] = __builtin_readcyclecounter();

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, g5 = lane >> 5;

  const int nt = g.Cout / B2_BN;
  const int tix = xcd_tile_index(blockIdx.x, A.num_mtiles * nt);
  const int mtile = tix / nt, ntile = tix - mtile * nt;
  const int n0 = ntile * B2_BN;
  const int m0 = mtile * BM;
  const int in_pixels = g.N * g.in_Hp * g.in_Wp;
  const int plane = g.MY * g.MX;

  // ---- set-up: row -> pixel tables (conv_tile.h igemm_row_pixels with multiply-shift divisions) ----
  int my_pin = 0;
  if (tid < BM) {
    const int m = m0 + tid;
    int n = bd2_divide(m, A.d_rows);
    int r = m - n * (g.MP > 0 ? g.MP : plane);
    const bool valid = n < g.N && r < plane;
    if (n >= g.N) { n = g.N - 1; r = plane - 1; }
    r = r < plane ? r : plane - 1;
    const int y = bd2_divide(r, A.d_mx), x = r - y * g.MX;
    my_pin = (n * g.in_Hp + y * g.sy + g.oy) * g.in_Wp + x * g.sx + g.ox;
    s_pin[tid] = my_pin;
    s_pout[tid] = valid ? (n * g.out_Hp + y * g.ty + g.py) * g.out_Wp + x * g.tx + g.px : -1;
  }
  // span of the tile's patch: rows are visited in non-decreasing pixel order except for clamped
  // (invalid) rows, which repeat an earlier pixel => max over the tile's rows
  int pmax = my_pin;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) pmax = max(pmax, __shfl_xor(pmax, o, 64));
  int* s_max = reinterpret_cast<int*>(s_red);      // 4 ints of the (still unused) reduction scratch
  if (lane == 0) s_max[wave] = pmax;
  __syncthreads();
  const int p_lo = s_pin[0];
  const int p_hi = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
  int npix = p_hi - p_lo + A.max_tap + 1;
  npix = min(npix, A.buf_bytes >> 6);
  const int jskip = A.jskip;
  auto dense_of = [&](int p) { return p - jskip * bd2_divide(p, A.d_wp); };

  const int v_tapoff = g.tap_off[lane & (IIC_MAX_TAPS - 1)];
  const int v_tapw = g.tap_w[lane & (IIC_MAX_TAPS - 1)];
  const int v_tapd = v_tapoff - jskip * bd2_divide(v_tapoff, A.d_wp);

  int arow[MS], drow[MS];
#pragma unroll
  for (int ms = 0; ms < MS; ++ms) {
    const int row = wm * WR + ms * 32 + l31;
    arow[ms] = (s_pin[row] - p_lo) << 6;            // byte offset of the lane's patch row at tap offset 0
    drow[ms] = dense_of(s_pin[row]);
  }

  const int nchunks = g.Cin >> 6;
  const int ntaps = g.ntaps;
  const int NIT = nchunks * 2 * ntaps;              // (chunk, half, tap) iterations of 2 k-steps, even
  const int nblk = (npix * 64 + 1023) >> 10;        // 1 KB DMA blocks per half-chunk patch
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)sA;

  // one DMA block (blk) of half-chunk (c, h) into buffer hb
  auto dma_block = [&](int blk, int c, int h, int hb) {
    const int q = blk * 64 + lane;
    int r = q >> 2;
    r = r < npix ? r : npix - 1;
    int p = p_lo + r;
    p = p < in_pixels ? p : in_pixels - 1;
    const int ls = (q ^ (dense_of(p) >> 2)) & 3;
    b2_dma16(in + ((long)p * g.Cin + c * 64 + h * 32 + ls * 8),
             (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)(hb * A.buf_bytes + blk * 1024))));
  };

  // ---- prologue ------------------------------------------------------------------------
  if (PROF) t_stamp[1] = __builtin_readcyclecounter();
  const long frag_it = (long)(g.Cout >> 5) * 4096;      // bytes per (tap, chunk)
  const unsigned char* wb0 = wfrag + (long)((n0 + wn * 64) >> 5) * 4096 + lane * 16;
  auto frag_ptr = [&](int tap, int half, int chunk) {
    const int tw = __builtin_amdgcn_readlane(v_tapw, tap);
    return wb0 + ((long)tw * nchunks + chunk) * frag_it + half * 2048;
  };
  auto advance = [&](int& t, int& h, int& c) {
    if (++t == ntaps) {
      t = 0;
      if (++h == 2) { h = 0; ++c; }
    }
  };
  u32x4 Bc[4][2];        // ring position 2 * (iteration parity) + k-step
  int t2 = 0, h2 = 0, c2 = 0;                           // iteration j + 2 (B pointer)
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const unsigned char* p = frag_ptr(t2, h2, c2);
#pragma unroll
    for (int kl = 0; kl < 2; ++kl)
#pragma unroll
      for (int ns = 0; ns < 2; ++ns) b2_bload(Bc[2 * par + kl][ns], p + ns * 4096 + kl * 1024);
    advance(t2, h2, c2);
  }
  for (int blk = wave; blk < nblk; blk += B2_THREADS / 64) dma_block(blk, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (PROF) t_stamp[2] = __builtin_readcyclecounter();

  f32x16 acc[MS][2];
#pragma unroll
  for (int ms = 0; ms < MS; ++ms)
#pragma unroll
    for (int ns = 0; ns < 2; ++ns)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ms][ns][r] = 0.f;

  // A-fragment address of (row, tap) in buffer hb: hb * buf + (arow + tapoff * 64) + (((g5 ^ (D >> 2)) & 3) << 4),
  // second k-step: ^ 32
  auto a_addr = [&](int ms, int toff, int td, int hb) {
    const int D = drow[ms] + td;
    return hb * A.buf_bytes + arow[ms] + (toff << 6) + (((g5 ^ (D >> 2)) & 3) << 4);
  };
  bf16x8 a[2][MS];
  int pcur[MS];
  int tap = 0, half = 0, chunk = 0;                      // iteration j
  {
    const int toff = __builtin_amdgcn_readlane(v_tapoff, 0), td = __builtin_amdgcn_readlane(v_tapd, 0);
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) {
      pcur[ms] = a_addr(ms, toff, td, 0);
      a[0][ms] = *reinterpret_cast<const bf16x8*>(sA + pcur[ms]);
    }
  }

  // ---- main loop -------------------------------------------------------------------------
  auto body = [&](auto parsel) {
    constexpr int PAR = decltype(parsel)::value;
    int t1 = tap, h1 = half, c1 = chunk;                 // iteration j + 1 (next A addresses)
    advance(t1, h1, c1);
    const bool more = c1 < nchunks;
    const bool more2 = c2 < nchunks;
    // (past the end the ring and the speculative reads re-use the last valid iteration: no branches)
    const unsigned char* nb = more2 ? frag_ptr(t2, h2, c2) : frag_ptr(tap, half, chunk);
    int pnext[MS];
    {
      const int tq = more ? t1 : tap;
      const int toff = __builtin_amdgcn_readlane(v_tapoff, tq), td = __builtin_amdgcn_readlane(v_tapd, tq);
      const int hb = more ? ((c1 * 2 + h1) & 1) : ((chunk * 2 + half) & 1);
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) pnext[ms] = a_addr(ms, toff, td, hb);
    }
#pragma unroll
    for (int kl = 0; kl < 2; ++kl) {
      u32x4& B0 = Bc[2 * PAR + kl][0];
      u32x4& B1 = Bc[2 * PAR + kl][1];
      b2_bwait<7>(B0);
      __builtin_amdgcn_sched_barrier(0);
      const bf16x8 b0 = __builtin_bit_cast(bf16x8, B0);
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) {
        acc[ms][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kl][ms], b0, acc[ms][0], 0, 0, 0);
        a[kl ^ 1][ms] = (kl == 0) ? *reinterpret_cast<const bf16x8*>(sA + (pcur[ms] ^ 32))
                                  : *reinterpret_cast<const bf16x8*>(sA + pnext[ms]);
        __builtin_amdgcn_sched_barrier(0);
      }
      b2_bwait<6>(B1);
      b2_bload(B0, nb + kl * 1024);
      __builtin_amdgcn_sched_barrier(0);
      const bf16x8 b1 = __builtin_bit_cast(bf16x8, B1);
#pragma unroll
      for (int ms = 0; ms < MS; ++ms)
        acc[ms][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kl][ms], b1, acc[ms][1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      b2_bload(B1, nb + 4096 + kl * 1024);
      __builtin_amdgcn_sched_barrier(0);
    }
    // The NEXT half-chunk's patch.  vmcnt retires in order, so every B load issued behind a DMA piece
    // waits for that piece: spread over the iterations (one piece each, first version) every piece stalled
    // the ring for (DMA latency - ring look-ahead) -- +38 k cycles per layer-3 tile.  Issued as ONE burst
    // in the first iteration of a half the ring stalls once per half; the boundary's vmcnt(8) still
    // covers every piece (>= 2 iterations of B loads follow).  mode 1 (A/B): the spread issue.
    {
      const int Hn = chunk * 2 + half + 1;
      if (Hn < 2 * nchunks) {
        if (A.mode == 0) {
          if (tap == 0)
            for (int blk = wave; blk < nblk; blk += B2_THREADS / 64) dma_block(blk, Hn >> 1, Hn & 1, Hn & 1);
        } else if (tap + 2 < ntaps) {
          for (int i = 0; i < A.pieces; ++i) {
            const int blk = wave + 4 * (tap * A.pieces + i);
            if (blk < nblk) dma_block(blk, Hn >> 1, Hn & 1, Hn & 1);
          }
        }
      }
    }
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) pcur[ms] = pnext[ms];
    if (more && t1 == 0) {            // the next iteration starts a new half-chunk: swap buffers
      if (PROF) t_b0 = __builtin_readcyclecounter();
      asm volatile("s_waitcnt vmcnt(8)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (PROF) { t_bsum += __builtin_readcyclecounter() - t_b0; ++t_nb; }
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) a[0][ms] = *reinterpret_cast<const bf16x8*>(sA + pcur[ms]);
    }
    tap = t1; half = h1; chunk = c1;
    if (more2) advance(t2, h2, c2);
  };
  for (int it = 0; it < NIT; it += 2) {
    body(std::integral_constant<int, 0>());
    body(std::integral_constant<int, 1>());
  }
  // the ring re-fills unconditionally: drain it before hipcc re-uses its registers
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue (as conv_igemm_bd.hip) ------------------------------------------------------
  if (PROF) t_stamp[3] = __builtin_readcyclecounter();
  const bool tail = igemm_tile_has_invalid(g, m0, BM);
  __syncthreads();                  // s_max (aliasing s_red) is dead, everyone has left the K loop
  if (stats) {
    if (tail) {
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
      f32x2 s2 = {0.f, 0.f}, ss2 = {0.f, 0.f};
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
        s_red[(wm * 2 + 0) * B2_BN + col] = s;
        s_red[(wm * 2 + 1) * B2_BN + col] = ss;
      }
    }
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
  if (stats && tid < B2_BN) {
    const int stripe = blockIdx.x % IIC_STAT_STRIPES;
    iic_stat_add(stats, stripe, g.Cout, n0 + tid, 0, s_red[0 * B2_BN + tid] + s_red[2 * B2_BN + tid]);
    iic_stat_add(stats, stripe, g.Cout, n0 + tid, 1, s_red[1 * B2_BN + tid] + s_red[3 * B2_BN + tid]);
  }
  if (PROF) t_stamp[4] = __builtin_readcyclecounter();
  TileRed tr;
  if (RED) tile_red_zero(tr);
  igemm_store_tile<B2_BN, BM, B2_THREADS, 8, RED, 8>(sC, s_pout, out, res_grad, res_act, accumulate, g.Cout, n0,
                                                    tid, red_y, red_coef, red_y2, &tr);
  if (RED)
    igemm_red_finish<B2_BN, B2_THREADS, RED>(tr, reinterpret_cast<float*>(smem_raw), red_stats, red_stats2,
                                             g.Cout, n0, tid);
  if (PROF && prof && tid == 0) {
    unsigned long long* q = prof + (long)blockIdx.x * B2_PROF_SLOTS;
    const unsigned long long t_end = __builtin_readcyclecounter();
    q[0] = t_stamp[0]; q[1] = t_stamp[1]; q[2] = t_stamp[2]; q[3] = t_stamp[3]; q[4] = t_stamp[4];
    q[5] = t_end; q[6] = t_bsum; q[7] = (unsigned long long)t_nb;
    q[8] = (unsigned long long)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4)
           | ((unsigned long long)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) << 32);
    q[9] = (unsigned long long)tix;
    q[10] = __builtin_amdgcn_s_memrealtime();
  }
}

static int g_bd2_enabled = 0;   // off: measured slower than the round-2 kernel + interleaved K loop (LAB.md §8, profiles/r03_bd2_*)
extern "C" void iic_debug_bd2(int v) { g_bd2_enabled = v; }
static int g_bd2_mode = 0;
extern "C" void iic_debug_bd2_mode(int v) { g_bd2_mode = v; }

static long bd2_buf_bytes(const iic_conv_geom* g, int ms) {
  const long npix = ms == 4 ? g->NP256 : (ms == 2 ? g->NP : 0);
  return npix > 0 ? ((npix * 64 + 1023) & ~1023L) : 0;
}
static long bd2_lds_a(const iic_conv_geom* g, int ms) {
  const long a = 2 * bd2_buf_bytes(g, ms), c = (long)ms * 64 * (B2_BN + 8) * 2;
  return ((a > c ? a : c) + 15) & ~15L;
}
static long bd2_lds_total(const iic_conv_geom* g, int ms) {
  return bd2_lds_a(g, ms) + 2L * ms * 64 * 4 + 4L * B2_BN * 4;
}

/* 1 if conv_igemm_bd2_kernel takes this geometry at tile height ms (4 = 256 rows). */
int iic_bd2_supported(const iic_conv_geom* g, int ms) {
  if (!g_bd2_enabled || !g) return 0;
  if (g->Cin % 64 != 0 || g->Cout % B2_BN != 0 || g->ntaps < 4 || g->ntaps > IIC_MAX_TAPS) return 0;
  if (ms != 4 || g->NP256 <= 0) return 0;
  if (bd2_lds_total(g, ms) > 80 * 1024) return 0;          // two workgroups per CU
  // every DMA piece must be issued while >= 2 iterations of its half remain: pieces per wave and iteration
  const long nblk = bd2_buf_bytes(g, ms) >> 10;
  const long per_wave = (nblk + 3) / 4;
  const long pieces = (per_wave + (g->ntaps - 2) - 1) / (g->ntaps - 2);
  return pieces <= 4;
}

int iic_bd2_launch(const iic_conv_geom* g, const void* in, const void* wfrag, void* out, float* stats,
                   const void* res_grad, const void* res_act, int accumulate, const void* red_y,
                   const float* red_coef, const void* red_y2, float* red_stats, float* red_stats2,
                   int dense_key, unsigned long long* prof, void* stream) {
  const int ms = 4;
  if (!iic_bd2_supported(g, ms)) return IIC_ERR_UNSUPPORTED;
  const long M = igemm_rows_host(g);
  if (M <= 0) return IIC_ERR_ARG;
  if (M >= (1L << 31) || (long)g->N * g->in_Hp * g->in_Wp >= (1L << 31)) return IIC_ERR_UNSUPPORTED;
  bd2_args A;
  const int plane = g->MY * g->MX;
  A.d_rows = bd2_make_div(g->MP > 0 ? g->MP : plane);
  A.d_mx = bd2_make_div(g->MX);
  A.d_wp = bd2_make_div(g->in_Wp);
  A.jskip = (dense_key && g->sx == 1) ? g->in_Wp - g->MX : 0;
  A.max_tap = 0;
  for (int t = 0; t < g->ntaps; ++t) A.max_tap = g->tap_off[t] > A.max_tap ? g->tap_off[t] : A.max_tap;
  A.buf_bytes = (int)bd2_buf_bytes(g, ms);
  A.lds_a_bytes = (int)bd2_lds_a(g, ms);
  const int bm = ms * 64;
  A.num_mtiles = (int)((M + bm - 1) / bm);
  const long per_wave = ((A.buf_bytes >> 10) + 3) / 4;
  A.pieces = (int)((per_wave + (g->ntaps - 2) - 1) / (g->ntaps - 2));
  A.mode = g_bd2_mode;
  const int grid = A.num_mtiles * (g->Cout / B2_BN);
  const long lds = bd2_lds_total(g, ms);
  const int red = red_y ? (red_y2 ? 2 : 1) : 0;
  hipStream_t s = (hipStream_t)stream;
#define B2_LAUNCH(RD_, PF_)                                                                           \
  do {                                                                                               \
    static bool attr = false;                                                                        \
    if (!attr) {                                                                                     \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_bd2_kernel<4, RD_, PF_>),  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);             \
      attr = true;                                                                                   \
    }                                                                                                \
    hipLaunchKernelGGL((conv_igemm_bd2_kernel<4, RD_, PF_>), dim3(grid), dim3(B2_THREADS), lds, s,   \
                       *g, A, (const bf16_t*)in, (const unsigned char*)wfrag, (bf16_t*)out, stats,   \
                       (const bf16_t*)res_grad, (const bf16_t*)res_act, accumulate,                  \
                       (const bf16_t*)red_y, red_coef, (const bf16_t*)red_y2, red_stats, red_stats2, \
                       prof);                                                                        \
  } while (0)
  if (prof) {
    if (red == 0) B2_LAUNCH(0, true); else if (red == 1) B2_LAUNCH(1, true); else B2_LAUNCH(2, true);
  } else {
    if (red == 0) B2_LAUNCH(0, false); else if (red == 1) B2_LAUNCH(1, false); else B2_LAUNCH(2, false);
  }
  return iic_launch_status();
}

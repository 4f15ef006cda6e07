// Persistent weights-direct implicit-GEMM convolution for gfx950 (round 4): the stride-1 multi-tap,
// Cin % 64 == 0, Cout % 128 == 0 launches of conv_igemm_bd.hip -- the 3x3 convolutions of layers 2-4,
// forward and backward-data, and the VGG-style stages whose 256-row patch fits --
//   /root/reference/code/archs/cluster/residual.py:4-7,19,22   (conv3x3, BasicBlock.conv1 / conv2)
//   /root/reference/code/archs/cluster/vgg.py:24-26            (VGG-style stages, dilated 3x3)
// Same contract, same B-fragment weight operand (iic_weight_prep_frag), same results per output element
// (same K order) as conv_igemm_bd_kernel; what changes is everything AROUND the MFMA loop.
//
// What the round-3 measurements said about conv_igemm_bd_kernel (LAB.md section 8.1, profiles/r03_bd_timeline_*):
// inside the K loop two co-resident waves keep a SIMD's matrix pipe 0.95 busy, but a 256 x 128 tile spends
// 29-48 % of its cycles outside it (row tables 3 k, prologue 5 k, 3.6 k per chunk boundary, epilogue 13-21 k),
// and while ONE workgroup of a CU is in such a phase the other's waves run alone on their SIMDs at only
// ~0.62 of the pipe: every (tap, chunk) iteration opened with ~65 address / pointer instructions in a row
// (sched_barrier-pinned), a bubble the partner wave normally hides.
// Here
//   * one workgroup walks several tiles (persistent; grid = workgroup slots of the chip, every workgroup keeps ONE
//     128-cout column block and its XCD a contiguous run of row tiles): the patch of tile t+1 (chunk 0) is
//     fetched by LDS-DMA while tile t's epilogue runs, row -> pixel arithmetic is done per lane with
//     multiply-shift divisions (no row tables, no set-up barrier), BatchNorm statistics and the fused
//     reduction's partial sums stay in registers across tiles and reach the exact accumulators once per launch;
//   * the epilogue is per WAVE: a wave converts one 32 x 64 block of its accumulators at a time through a
//     private 4.5 KB staging area (no workgroup barrier, not aliased with the patch) into 16-byte row stores
//     with the fused reads -- waves drift apart and a wave's store phase overlaps the MFMAs of the others;
//   * the K loop spreads the next tap's address arithmetic over the gaps between this tap's MFMAs (two
//     address sets, ping-pong: no register rotation), the B-fragment pointer is a scalar base + immediate
//     offsets (no per-load VALU), so a wave that is alone on its SIMD still issues MFMAs back to back.
// LDS: patch (NP256 rows x 128 B) + 4 x 4.5 KB staging + 1 KB of swizzle keys  => two workgroups per CU up to
// NP256 = 476.
#include <type_traits>

#include "common.h"
#include "conv_tile.h"
#include "../../include/iic_hip.h"

#define PW_THREADS 256
#define PW_STG_LD 72                          // staging row pitch (bf16): 64 columns + 8 pad
#define PW_STG_BYTES (32 * PW_STG_LD * 2)     // 4608 B per wave
#define PW_KEYS 512                           // swizzle keys per buffer (>= NP256)
#define PW_PROF_SLOTS 32      // 0..7 sums, 8..23 tile stamps, 24 / 25 s_memrealtime (100 MHz) at start / end

struct pw_div {      // floor(n / d) for 0 <= n < 2^31:  (n * mul) >> sh  (64-bit product)
  unsigned mul;
  int sh;
};
static inline pw_div pw_make_div(int d) {
  pw_div r;
  int l = 0;
  while ((1L << l) < d) ++l;
  r.sh = 31 + l;
  r.mul = (unsigned)(((1ULL << r.sh) + (unsigned long long)d - 1) / (unsigned long long)d);
  if (d == 1) { r.mul = 1u << 31; r.sh = 31; }
  return r;
}
__device__ __forceinline__ int pw_divide(int n, const pw_div& d) {
  return (int)(((unsigned long long)(unsigned)n * d.mul) >> d.sh);
}

struct pw_args {
  pw_div d_rows;       // by rows per image (g.MP or the plane)
  pw_div d_mx;         // by g.MX
  pw_div d_wp;         // by g.in_Wp
  int rows_per_img;    // g.MP > 0 ? g.MP : plane
  int plane;           // g.MY * g.MX
  int jskip;           // dense-count skip per image row (0: key from the raw pixel index)
  int npix;            // patch rows (g.NP256)
  int patch_bytes;     // npix * 128 rounded up to 1 KB
  int mt;              // 256-row tiles
  int M;               // GEMM rows
  int in_pixels;       // N * in_Hp * in_Wp
  int pad_rows;        // 1: g.MP pads the per-image row count (rows >= plane are invalid)
  int dbg;             // timing experiments (results WRONG): 1 = every B fragment from the same 8 KB (L1-hot)
};

// B fragment: 16 bytes per lane from (scalar base + per-lane offset + immediate); inline asm so that hipcc's
// vmcnt bookkeeping never sees the ring (conv_igemm_bd.hip bd_bload)
template <int OFF>
__device__ __forceinline__ void pw_bload(u32x4& d, unsigned voff, const unsigned char* sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(d) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void pw_bwait(u32x4& d) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(d) : "n"(N) : "memory");
}
// one LDS-DMA piece: 64 lanes x 16 bytes from per-lane global addresses to lds_dst + lane * 16 (lds_dst wave-uniform;
// M0 is compiler-reserved: saved and restored inside the statement).  Inline asm: hipcc would wait vmcnt(0) in front
// of the first LDS read after a DMA it knows about.
__device__ __forceinline__ void pw_dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

// GEMM row -> (input pixel, its padded image-row index p / in_Wp, output pixel or -1); conv_tile.h
// igemm_row_pixels with multiply-shift divisions
__device__ __forceinline__ void pw_row(const iic_conv_geom& g, const pw_args& A, int m, int& pin, int& prow,
                                       int& pout) {
  int n = pw_divide(m, A.d_rows);
  int r = m - n * A.rows_per_img;
  const bool valid = n < g.N && r < A.plane;
  if (n >= g.N) { n = g.N - 1; r = A.plane - 1; }
  r = r < A.plane ? r : A.plane - 1;
  const int y = pw_divide(r, A.d_mx), x = r - y * g.MX;
  prow = n * g.in_Hp + y * g.sy + g.oy;
  pin = prow * g.in_Wp + x * g.sx + g.ox;
  pout = valid ? (n * g.out_Hp + y * g.ty + g.py) * g.out_Wp + x * g.tx + g.px : -1;
}
// One 32-row x 64-column block of a wave's output tile: sW (the wave's staging area, [32][PW_STG_LD] bf16) -> `out`
// rows po[u] (pass u covers rows u * 8 + lane / 8; < 0: skipped), 16-byte stores, with the fused epilogue of
// conv_tile.h igemm_store_tile (same arithmetic, same order): accumulate flags, residual gradient / ReLU mask,
// pre-masked gradients, and the BatchNorm-backward reduction RED over the values as stored.  A lane owns the
// 8 channels col0 + (lane & 7) * 8 ... for the whole launch (TileRed carried across blocks and tiles).
template <int RED, int U>
__device__ __forceinline__ void pw_store_pass(const bf16_t* sW, int u0, const int (&po)[4], bf16_t* __restrict__ out,
                                               const bf16_t* __restrict__ res_grad, const bf16_t* __restrict__ res_act,
                                               int accumulate, int Cout, int col0, int lane,
                                               const bf16_t* __restrict__ red_y, bool red_mask,
                                               const float (&msc)[8], const float (&msh)[8],
                                               const bf16_t* __restrict__ red_y2, TileRed& red) {
  const bool add_prev = accumulate & IIC_ACC_ADD, premask = accumulate & IIC_ACC_PREMASK;
  const bool any_in = add_prev || res_grad || res_act;
  const int ch = lane & 7, rb = lane >> 3;
  long o[U];
  uint4 pv[U], gv[U], av[U], yv[U], zv[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    o[u] = po[u0 + u] < 0 ? -1 : (long)po[u0 + u] * Cout + col0 + ch * 8;
    if (o[u] >= 0) {
      if (add_prev) pv[u] = *reinterpret_cast<const uint4*>(out + o[u]);
      if (res_grad) gv[u] = *reinterpret_cast<const uint4*>(res_grad + o[u]);
      if (res_act) av[u] = *reinterpret_cast<const uint4*>(res_act + o[u]);
      if (RED) yv[u] = *reinterpret_cast<const uint4*>(red_y + o[u]);
      if (RED == 2) zv[u] = *reinterpret_cast<const uint4*>(red_y2 + o[u]);
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (o[u] < 0) continue;
    const int row = (u0 + u) * 8 + rb;
    uint4 v = *reinterpret_cast<const uint4*>(sW + row * PW_STG_LD + ch * 8);
    if (any_in) {
      uint32_t vv[4] = {v.x, v.y, v.z, v.w};
      float f[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) { f[2 * i] = bf16lo(vv[i]); f[2 * i + 1] = bf16hi(vv[i]); }
      if (add_prev) {
        const uint32_t oo[4] = {pv[u].x, pv[u].y, pv[u].z, pv[u].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { f[2 * i] += bf16lo(oo[i]); f[2 * i + 1] += bf16hi(oo[i]); }
      }
      if (premask) {
        if (res_grad) {
          const uint32_t gg[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
#pragma unroll
          for (int i = 0; i < 4; ++i) { f[2 * i] += bf16lo(gg[i]); f[2 * i + 1] += bf16hi(gg[i]); }
        }
        if (res_act) {
          const uint32_t aa[4] = {av[u].x, av[u].y, av[u].z, av[u].w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (!(bf16lo(aa[i]) > 0.f)) f[2 * i] = 0.f;
            if (!(bf16hi(aa[i]) > 0.f)) f[2 * i + 1] = 0.f;
          }
        }
      } else if (res_grad) {
        const uint32_t gg[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
        const uint32_t aa[4] = {av[u].x, av[u].y, av[u].z, av[u].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (bf16lo(aa[i]) > 0.f) f[2 * i] += bf16lo(gg[i]);
          if (bf16hi(aa[i]) > 0.f) f[2 * i + 1] += bf16hi(gg[i]);
        }
      }
      v = make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]),
                     pack_bf16x2(f[6], f[7]));
    }
    *reinterpret_cast<uint4*>(out + o[u]) = v;
    if (RED) {
      const uint32_t gq4[4] = {v.x, v.y, v.z, v.w};
      const uint32_t yy[4] = {yv[u].x, yv[u].y, yv[u].z, yv[u].w};
      float gq[8], yq[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        gq[2 * i] = bf16lo(gq4[i]); gq[2 * i + 1] = bf16hi(gq4[i]);
        yq[2 * i] = bf16lo(yy[i]); yq[2 * i + 1] = bf16hi(yy[i]);
      }
      if (red_mask) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (!(yq[i] * msc[i] + msh[i] > 0.f)) gq[i] = 0.f;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) { red.s[i] += gq[i]; red.sy[i] += gq[i] * yq[i]; }
      if (RED == 2) {
        const uint32_t zz[4] = {zv[u].x, zv[u].y, zv[u].z, zv[u].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          red.sy2[2 * i] += gq[2 * i] * bf16lo(zz[i]);
          red.sy2[2 * i + 1] += gq[2 * i + 1] * bf16hi(zz[i]);
        }
      }
    }
  }
}

// PROF (results correct): wave 0 sums s_memtime per phase over its tiles into prof[blockIdx][PW_PROF_SLOTS]:
// 0 waiting for the patch (tile top), 1 K loop incl. boundaries, 2 boundaries, 3 epilogue, 4 everything, 5 tiles,
// 6 boundary count, 7 XCC id
template <int RED, bool PROF>
__global__ __launch_bounds__(PW_THREADS, 2) void conv_igemm_pw_kernel(
    const iic_conv_geom g, const pw_args A, const bf16_t* __restrict__ in, const unsigned char* __restrict__ wfrag,
    bf16_t* __restrict__ out, float* __restrict__ stats, const bf16_t* __restrict__ res_grad,
    const bf16_t* __restrict__ res_act, int accumulate, const bf16_t* __restrict__ red_y,
    const float* __restrict__ red_coef, const bf16_t* __restrict__ red_y2, float* __restrict__ red_stats,
    float* __restrict__ red_stats2, unsigned long long* __restrict__ prof, int stagger) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, g5 = lane >> 5;
  unsigned char* sA = smem_raw;                                                    // [patch_bytes]
  bf16_t* sW = reinterpret_cast<bf16_t*>(smem_raw + A.patch_bytes + wave * PW_STG_BYTES);   // this wave's staging
  unsigned char* s_key = smem_raw + A.patch_bytes + 4 * PW_STG_BYTES;              // [2][PW_KEYS]
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)sA;

  unsigned long long t_all = 0, t_wait = 0, t_k = 0, t_b = 0, t_e = 0, t0 = 0, t1 = 0;
  int n_tiles = 0, n_b = 0;
  unsigned long long t_line[4][4];      // absolute stamps of the first four tiles: top, K loop start, K loop end, epilogue end
  unsigned long long rt0 = 0;
  if (PROF) {
    rt0 = __builtin_amdgcn_s_memrealtime();
    t_all = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) t_line[i][j] = 0;
  }

  // ---- tile schedule: block b runs on XCD b & 7 (observed dispatch; speed only).  Every workgroup keeps one
  // column block; the XCDs that share a column block split the row tiles into contiguous runs, and the
  // workgroups of an XCD walk their run side by side (neighbouring tiles share halo rows and weights in L2).
  const int nt = g.Cout >> 7;                          // 1, 2, 4 or 8 (host)
  const int xcd = blockIdx.x & 7, kx = blockIdx.x >> 3, Gx = gridDim.x >> 3;
  const int ntile = xcd & (nt - 1), per = 8 / nt, xr = xcd / nt;
  const int mt_lo = (int)((long)A.mt * xr / per), mt_hi = (int)((long)A.mt * (xr + 1) / per);
  const int n0 = ntile << 7;
  int mtile = mt_lo + kx;
  if (mtile >= mt_hi) return;
  // Stagger experiment (iic_debug_pw_stagger, cycles): all workgroups start together and do identical work, so the
  // two that share a CU reach their epilogues -- and the whole chip its HBM burst -- at the same moment.  The
  // workgroup in an odd threadgroup slot of its CU (HW_ID.TG_ID) starts `stagger` cycles late.
  const unsigned hw_id = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
  if (stagger > 0 && ((hw_id >> 16) & 1u)) {
    const unsigned long long t_go = __builtin_readcyclecounter() + (unsigned long long)stagger;
    while (__builtin_readcyclecounter() < t_go) __builtin_amdgcn_s_sleep(32);
  }

  const int v_tapoff = g.tap_off[lane & (IIC_MAX_TAPS - 1)];
  const int v_tapw = g.tap_w[lane & (IIC_MAX_TAPS - 1)];
  const int v_tapd = v_tapoff - A.jskip * pw_divide(v_tapoff, A.d_wp);
  const int nchunks = g.Cin >> 6, ntaps = g.ntaps, NIT = nchunks * ntaps;
  const int nblk = A.patch_bytes >> 10;
  const long frag_it = (long)(g.Cout >> 5) * 4096;      // bytes per (tap, chunk)
  const unsigned boff = (unsigned)(((n0 + wn * 64) >> 5) * 4096 + lane * 16);    // this lane's fragment offset
  auto frag_base = [&](int tap, int chunk) {             // scalar: + 4096 so that all 8 immediates fit [-4096, 3072]
    const int tw = __builtin_amdgcn_readlane(v_tapw, tap);
    return wfrag + ((A.dbg & 1) ? 0L : ((long)tw * nchunks + chunk) * frag_it) + 4096;
  };

  // swizzle keys of a tile's patch rows (conv_igemm_bd.hip: key = (D >> 1) & 7, D the dense pixel count)
  auto tile_plo = [&](int mtl) {
    int pin, prow, pout;
    pw_row(g, A, mtl << 8, pin, prow, pout);
    return __builtin_amdgcn_readfirstlane(pin);
  };
  auto write_keys = [&](int p_lo_t, int kb) {
    for (int r = tid; r < PW_KEYS; r += PW_THREADS) {
      const int p = p_lo_t + r;
      const int prow = pw_divide(p, A.d_wp);
      const int D = p - A.jskip * prow;
      const int key = (D >> 1) & 7;
      s_key[kb * PW_KEYS + r] = (unsigned char)key;
    }
  };
  // piece q -> LDS byte q * 16 (row q >> 3, physical slot q & 7), source = logical slot (q & 7) ^ key(row)
  auto dma_patch = [&](int p_lo_t, int c0, int kb) {
    for (int blk = wave; blk < nblk; blk += PW_THREADS / 64) {
      const int q = blk * 64 + lane;
      const int r = q >> 3;
      const int ls = (q & 7) ^ ((int)s_key[kb * PW_KEYS + r] & 7);
      long p = (long)p_lo_t + r;
      p = p < A.in_pixels ? p : A.in_pixels - 1;
      pw_dma16(in + (p * g.Cin + c0 + ls * 8),
               (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)(blk * 1024))));
    }
  };

  // BatchNorm statistics / fused-reduction partials carried across the tiles of this workgroup
  float st_s[2] = {0.f, 0.f}, st_ss[2] = {0.f, 0.f};
  TileRed tr;
  tile_red_zero(tr);
  const bool red_mask = RED && red_coef != nullptr;

  int kb = 0;
  int p_lo = tile_plo(mtile);
  write_keys(p_lo, 0);
  __syncthreads();
  dma_patch(p_lo, 0, 0);

  for (;;) {
    const int m0 = mtile << 8;
    const int mnext = mtile + Gx;
    const bool has_next = mnext < mt_hi;
    if (PROF) {
      t0 = __builtin_readcyclecounter();
#pragma unroll
      for (int i = 0; i < 4; ++i) if (n_tiles == i) t_line[i][0] = t0;
    }

    // ---- this lane's four rows: patch row index at tap offset 0 and dense count ----
    int arow[4], drow[4];
#pragma unroll
    for (int ms = 0; ms < 4; ++ms) {
      int pin, prow, pout;
      pw_row(g, A, m0 + wm * 128 + ms * 32 + l31, pin, prow, pout);
      arow[ms] = pin - p_lo;
      drow[ms] = pin - A.jskip * prow;
    }
    int p_lo_next = 0;
    if (has_next) {
      p_lo_next = tile_plo(mnext);
      write_keys(p_lo_next, kb ^ 1);          // (that buffer's last reader was the previous tile's DMA issue)
    }
    auto tap_p = [&](int ms, int toff, int td) {
      const int R = arow[ms] + toff, D = drow[ms] + td;
      return (R << 7) + ((((D >> 1) ^ g5) & 1) << 4);
    };
    auto tap_k = [&](int ms, int td) { return (((drow[ms] + td) >> 2) & 3) << 5; };

    f32x16 acc[4][2];
#pragma unroll
    for (int ms = 0; ms < 4; ++ms)
#pragma unroll
      for (int ns = 0; ns < 2; ++ns)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ms][ns][r] = 0.f;

    // ---- B ring of the first iteration, then wait for the patch (the DMA pieces are older than the ring) ----
    u32x4 Bc[4][2];
    {
      const unsigned char* sb = frag_base(0, 0);
      pw_bload<-4096>(Bc[0][0], boff, sb); pw_bload<0>(Bc[0][1], boff, sb);
      pw_bload<-3072>(Bc[1][0], boff, sb); pw_bload<1024>(Bc[1][1], boff, sb);
      pw_bload<-2048>(Bc[2][0], boff, sb); pw_bload<2048>(Bc[2][1], boff, sb);
      pw_bload<-1024>(Bc[3][0], boff, sb); pw_bload<3072>(Bc[3][1], boff, sb);
    }
    asm volatile("s_waitcnt vmcnt(8)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (PROF) {
      t1 = __builtin_readcyclecounter(); t_wait += t1 - t0; t0 = t1;
#pragma unroll
      for (int i = 0; i < 4; ++i) if (n_tiles == i) t_line[i][1] = t1;
    }

    int pa[2][4], ka[2][4];           // A-fragment address sets of the current / next tap (ping-pong), LDS byte addresses
    bf16x8 a[2][4];
    auto lds16 = [](int addr) {
      return *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>((size_t)(unsigned)addr);
    };
    {
      const int toff = __builtin_amdgcn_readlane(v_tapoff, 0), td = __builtin_amdgcn_readlane(v_tapd, 0);
#pragma unroll
      for (int ms = 0; ms < 4; ++ms) {
        pa[0][ms] = (int)lds0 + tap_p(ms, toff, td);
        ka[0][ms] = tap_k(ms, td);
        a[0][ms] = lds16(pa[0][ms] + ka[0][ms]);
      }
    }

    // ---- K loop: flat (chunk, tap) iterations of 4 k-steps; barriers only when the chunk changes ----
    // Scalars of the NEXT iteration (its tap's offsets, its B-fragment base, whether it exists / opens a new chunk)
    // are computed one iteration ahead, inside the MFMA gaps of k-step 2, so that an iteration opens with an MFMA.
    int tap_n = 0, chunk_n = 0;         // the next iteration's (tap, chunk); == the current one past the end
    bool more_n = false, bnd_n = false;
    const unsigned char* nb = nullptr;
    int toffn = 0, tdn = 0;
    auto advance = [&](int t, int c, int& t2, int& c2, bool& more2, bool& bnd2) {
      t2 = t + 1; c2 = c;
      if (t2 == ntaps) { t2 = 0; ++c2; }
      more2 = c2 < nchunks;
      bnd2 = more2 && t2 == 0;
      if (!more2) { t2 = t; c2 = c; }      // the ring always reloads, the speculative reads stay inside the patch
    };
    advance(0, 0, tap_n, chunk_n, more_n, bnd_n);
    nb = frag_base(tap_n, chunk_n);
    toffn = __builtin_amdgcn_readlane(v_tapoff, tap_n);
    tdn = __builtin_amdgcn_readlane(v_tapd, tap_n);
    auto body = [&](auto parsel) {
      constexpr int P = decltype(parsel)::value, Q = P ^ 1;
      int tap_2 = 0, chunk_2 = 0, toff2 = 0, td2 = 0;
      bool more_2 = false, bnd_2 = false;
      const unsigned char* nb2 = nullptr;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
        // in flight at the top of a k-step: the 8 loads of the next four k-steps, oldest first
        pw_bwait<7>(Bc[ks][0]);
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 b0 = __builtin_bit_cast(bf16x8, Bc[ks][0]);
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) {
          acc[ms][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][ms], b0, acc[ms][0], 0, 0, 0);
          a[nxt][ms] = (ks < 3) ? lds16(pa[P][ms] + (((ks + 1) << 5) ^ ka[P][ms])) : lds16(pa[Q][ms] + ka[Q][ms]);
          __builtin_amdgcn_sched_barrier(0);
        }
        pw_bwait<6>(Bc[ks][1]);
        if (ks == 0) pw_bload<-4096>(Bc[0][0], boff, nb);
        if (ks == 1) pw_bload<-3072>(Bc[1][0], boff, nb);
        if (ks == 2) pw_bload<-2048>(Bc[2][0], boff, nb);
        if (ks == 3) pw_bload<-1024>(Bc[3][0], boff, nb);
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 b1 = __builtin_bit_cast(bf16x8, Bc[ks][1]);
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) {
          acc[ms][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][ms], b1, acc[ms][1], 0, 0, 0);
          // the NEXT tap's addresses, one piece per MFMA gap (needed by k-step 3's reads)
          if (ks == 0) pa[Q][ms] = (int)lds0 + tap_p(ms, toffn, tdn);
          if (ks == 1) ka[Q][ms] = tap_k(ms, tdn);
          // the scalars of the iteration after the next
          if (ks == 2 && ms == 0) advance(tap_n, chunk_n, tap_2, chunk_2, more_2, bnd_2);
          if (ks == 2 && ms == 1) nb2 = frag_base(tap_2, chunk_2);
          if (ks == 2 && ms == 2) {
            toff2 = __builtin_amdgcn_readlane(v_tapoff, tap_2);
            td2 = __builtin_amdgcn_readlane(v_tapd, tap_2);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (ks == 0) pw_bload<0>(Bc[0][1], boff, nb);
        if (ks == 1) pw_bload<1024>(Bc[1][1], boff, nb);
        if (ks == 2) pw_bload<2048>(Bc[2][1], boff, nb);
        if (ks == 3) pw_bload<3072>(Bc[3][1], boff, nb);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (bnd_n) {       // the next iteration starts a new channel chunk
        if (PROF) t1 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();           // everyone is done reading the patch
        dma_patch(p_lo, chunk_n * 64, kb);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (PROF) { t_b += __builtin_readcyclecounter() - t1; ++n_b; }
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) a[0][ms] = lds16(pa[Q][ms] + ka[Q][ms]);
      }
      tap_n = tap_2; chunk_n = chunk_2; more_n = more_2; bnd_n = bnd_2;
      nb = nb2; toffn = toff2; tdn = td2;
    };
    for (int it = 0; it < NIT; it += 2) {
      body(std::integral_constant<int, 0>());
      if (it + 1 < NIT) body(std::integral_constant<int, 1>());
    }
    // the ring re-fills unconditionally (the last iteration's loads are never used): drain it before hipcc hands
    // their destination registers to the epilogue
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();              // every wave has left the K loop: the patch buffer is free
    if (has_next) dma_patch(p_lo_next, 0, kb ^ 1);     // lands while the epilogue runs
    if (PROF) {
      t1 = __builtin_readcyclecounter(); t_k += t1 - t0; t0 = t1;
#pragma unroll
      for (int i = 0; i < 4; ++i) if (n_tiles == i) t_line[i][2] = t1;
    }

    // ---- epilogue, per wave ----
    // valid rows of the tile are [0, vlimit): the launch's last tile, and -- with a padded per-image row count
    // (g.MP, a multiple of 256: a tile never straddles two images) -- the rows past the image's plane
    int vlimit = A.M - m0;
    if (A.pad_rows) {
      const int r0 = m0 - pw_divide(m0, A.d_rows) * A.rows_per_img;
      vlimit = min(vlimit, A.plane - r0);
    }
    if (stats) {
      // (the accumulators themselves stay untouched: invalid rows are skipped by the store, masked only here)
      const int lim = vlimit - wm * 128;
      auto sums = [&](auto masked) {
        constexpr bool MK = decltype(masked)::value;
#pragma unroll
        for (int ns = 0; ns < 2; ++ns) {
          f32x2 s2 = {0.f, 0.f}, ss2 = {0.f, 0.f};
#pragma unroll
          for (int ms = 0; ms < 4; ++ms)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              f32x2 v = {acc[ms][ns][r], acc[ms][ns][r + 1]};
              if (MK) {
                const int row = ms * 32 + mfma32_row(r, lane);
                v[0] = row < lim ? v[0] : 0.f;
                v[1] = row + 1 < lim ? v[1] : 0.f;
              }
              s2 += v;
              ss2 += v * v;
            }
          st_s[ns] += s2[0] + s2[1];
          st_ss[ns] += ss2[0] + ss2[1];
        }
      };
      if (lim < 128) sums(std::true_type()); else sums(std::false_type());
    }
    // the ReLU-mask coefficients of the fused reduction: this lane's 8 channels (re-read per tile, L2-hot: 16
    // registers that would otherwise be carried through the K loop)
    float msc[8], msh[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { msc[i] = 0.f; msh[i] = 0.f; }
    if (RED && red_coef) {
      const float4* c4 = reinterpret_cast<const float4*>(red_coef + n0 + wn * 64 + (lane & 7) * 8);
      const float4* h4 = reinterpret_cast<const float4*>(red_coef + g.Cout + n0 + wn * 64 + (lane & 7) * 8);
      const float4 c0 = c4[0], c1 = c4[1], h0 = h4[0], h1 = h4[1];
      msc[0] = c0.x; msc[1] = c0.y; msc[2] = c0.z; msc[3] = c0.w; msc[4] = c1.x; msc[5] = c1.y; msc[6] = c1.z; msc[7] = c1.w;
      msh[0] = h0.x; msh[1] = h0.y; msh[2] = h0.z; msh[3] = h0.w; msh[4] = h1.x; msh[5] = h1.y; msh[6] = h1.z; msh[7] = h1.w;
    }
#pragma unroll
    for (int ms = 0; ms < 4; ++ms) {
      // accumulators -> bf16 -> staging ([row][col], rows of the 32x32 C layout, two rows per conversion)
#pragma unroll
      for (int ns = 0; ns < 2; ++ns)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const uint32_t pk = pack_bf16x2(acc[ms][ns][r], acc[ms][ns][r + 1]);
          const int row = mfma32_row(r, lane), col = ns * 32 + l31;
          sW[row * PW_STG_LD + col] = (bf16_t)(pk & 0xffffu);
          sW[(row + 1) * PW_STG_LD + col] = (bf16_t)(pk >> 16);
        }
      int po[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int pin, prow;
        pw_row(g, A, m0 + wm * 128 + ms * 32 + u * 8 + (lane >> 3), pin, prow, po[u]);
      }
      // passes whose global loads are in flight together: 2 while most accumulators are still live, then 4
      if (ms == 0) {
        pw_store_pass<RED, 2>(sW, 0, po, out, res_grad, res_act, accumulate, g.Cout, n0 + wn * 64, lane, red_y, red_mask,
                              msc, msh, red_y2, tr);
        pw_store_pass<RED, 2>(sW, 2, po, out, res_grad, res_act, accumulate, g.Cout, n0 + wn * 64, lane, red_y, red_mask,
                              msc, msh, red_y2, tr);
      } else {
        pw_store_pass<RED, 4>(sW, 0, po, out, res_grad, res_act, accumulate, g.Cout, n0 + wn * 64, lane, red_y, red_mask,
                              msc, msh, red_y2, tr);
      }
    }
    if (PROF) {
      t1 = __builtin_readcyclecounter(); t_e += t1 - t0;
#pragma unroll
      for (int i = 0; i < 4; ++i) if (n_tiles == i) t_line[i][3] = t1;
      ++n_tiles;
    }
    if (!has_next) break;
    mtile = mnext;
    p_lo = p_lo_next;
    kb ^= 1;
  }

  // ---- once per launch: the carried sums reach the exact accumulators ----
  const int stripe = blockIdx.x % IIC_STAT_STRIPES;
  if (stats) {
#pragma unroll
    for (int ns = 0; ns < 2; ++ns) {
      float s = st_s[ns], ss = st_ss[ns];
      s += __shfl_xor(s, 32, 64);
      ss += __shfl_xor(ss, 32, 64);
      if (lane < 32) {
        const int col = n0 + wn * 64 + ns * 32 + lane;
        iic_stat_add(stats, stripe, g.Cout, col, 0, s);
        iic_stat_add(stats, stripe, g.Cout, col, 1, ss);
      }
    }
  }
  if (RED) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) {
        tr.s[i] += __shfl_xor(tr.s[i], o, 64);
        tr.sy[i] += __shfl_xor(tr.sy[i], o, 64);
        if (RED == 2) tr.sy2[i] += __shfl_xor(tr.sy2[i], o, 64);
      }
    }
    if (lane < 8) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int col = n0 + wn * 64 + lane * 8 + i;
        iic_stat_add(red_stats, stripe, g.Cout, col, 0, tr.s[i]);
        iic_stat_add(red_stats, stripe, g.Cout, col, 1, tr.sy[i]);
        if (RED == 2) {
          iic_stat_add(red_stats2, stripe, g.Cout, col, 0, tr.s[i]);
          iic_stat_add(red_stats2, stripe, g.Cout, col, 1, tr.sy2[i]);
        }
      }
    }
  }
  if (PROF && prof && tid == 0) {
    unsigned long long* q = prof + (long)blockIdx.x * PW_PROF_SLOTS;
    q[0] = t_wait; q[1] = t_k; q[2] = t_b; q[3] = t_e;
    q[4] = __builtin_readcyclecounter() - t_all;
    q[5] = (unsigned long long)n_tiles; q[6] = (unsigned long long)n_b;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) q[8 + i * 4 + j] = t_line[i][j];
    q[24] = rt0;
    q[25] = __builtin_amdgcn_s_memrealtime();
    q[7] = (unsigned long long)hw_id | ((unsigned long long)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) << 32);   // HW_ID, XCC_ID
  }
}

IIC_SWITCH(g_pw_enabled, 1, iic_debug_enable_pw)
IIC_SWITCH(g_pw_stagger, 0, iic_debug_pw_stagger)             // start offset (cycles) of the workgroups in odd threadgroup slots (A/B)
IIC_SWITCH(g_pw_min_tiles10, 25, iic_debug_pw_min_tiles10)    // take a launch only if it has >= this many tiles per workgroup slot (x 10)
IIC_SWITCH(g_pw_dbg, 0, iic_debug_pw_dbg)                     // timing experiment (results WRONG): 1 = every B fragment from the same 8 KB
IIC_SWITCH(g_pw_one_wg, 0, iic_debug_pw_one_wg)               // 1: pad the LDS request so that only one workgroup fits a CU (A/B)
#ifdef IIC_DEBUG_HOOKS
static unsigned long long* g_pw_prof = nullptr;
IIC_HOOK void iic_debug_pw_prof(void* buf) { g_pw_prof = (unsigned long long*)buf; }
IIC_HOOK int iic_debug_pw_prof_slots(void) { return PW_PROF_SLOTS; }
IIC_HOOK int iic_debug_pw_grid(const iic_conv_geom* g);
#else
static constexpr unsigned long long* g_pw_prof = nullptr;
#endif

static long pw_lds_bytes(const iic_conv_geom* g) {
  const long patch = ((long)g->NP256 * 128 + 1023) & ~1023L;
  return patch + 4 * PW_STG_BYTES + 2 * PW_KEYS;
}

static int pw_num_cus();
static int pw_supported_shape(const iic_conv_geom* g);
int iic_pw_supported(const iic_conv_geom* g) {
  if (!g || !g_pw_enabled || !pw_supported_shape(g)) return 0;
  const long M = igemm_rows_host(g);
  if (g_pw_min_tiles10 > 0 && ((M + 255) / 256) * (g->Cout / 128) * 10 < (long)g_pw_min_tiles10 * 2 * pw_num_cus()) return 0;
  return 1;
}
static int pw_supported_shape(const iic_conv_geom* g) {
  if (!g) return 0;
  if (g->ntaps < 2 || g->ntaps > IIC_MAX_TAPS || g->Cin % 64 != 0 || g->Cout % 128 != 0) return 0;
  const int nt = g->Cout / 128;
  if (nt != 1 && nt != 2 && nt != 4 && nt != 8) return 0;
  if (g->NP256 <= 0 || g->NP256 > PW_KEYS) return 0;
  if (pw_lds_bytes(g) > 80 * 1024) return 0;      // two workgroups per CU (larger patches: conv_igemm_bd_kernel)
  const long M = igemm_rows_host(g);
  if (M <= 0 || M + 256 >= (1L << 31) || (long)g->N * g->in_Hp * g->in_Wp >= (1L << 31)) return 0;
  return 1;
}

static int pw_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

// workgroups of a launch: at most two per CU, a multiple of 8, and no more per XCD than its longest run of tiles
static int pw_grid(const iic_conv_geom* g, long lds) {
  const long M = igemm_rows_host(g);
  const int mt = (int)((M + 255) / 256), nt = g->Cout / 128, per = 8 / nt;
  int longest = 0;
  for (int xr = 0; xr < per; ++xr) {
    const int len = (int)((long)mt * (xr + 1) / per) - (int)((long)mt * xr / per);
    longest = len > longest ? len : longest;
  }
  const int slots = (lds <= 80 * 1024 ? 2 : 1) * pw_num_cus();
  int gx = slots / 8;
  if (gx > longest) gx = longest;
  if (gx < 1) gx = 1;
  return gx * 8;
}
#ifdef IIC_DEBUG_HOOKS
IIC_HOOK int iic_debug_pw_grid(const iic_conv_geom* g) { return g ? pw_grid(g, pw_lds_bytes(g)) : 0; }
#endif

int iic_pw_launch(const iic_conv_geom* g, const void* in, const void* wfrag, void* out, float* stats,
                  const void* res_grad, const void* res_act, int accumulate, const void* red_y,
                  const float* red_coef, const void* red_y2, float* red_stats, float* red_stats2, void* stream) {
  if (!iic_pw_supported(g)) return IIC_ERR_UNSUPPORTED;
  const long M = igemm_rows_host(g);
  pw_args A;
  A.plane = g->MY * g->MX;
  A.rows_per_img = g->MP > 0 ? g->MP : A.plane;
  A.d_rows = pw_make_div(A.rows_per_img);
  A.d_mx = pw_make_div(g->MX);
  A.d_wp = pw_make_div(g->in_Wp);
  A.jskip = (g->sx == 1 && ((g->in_Wp - g->MX) & 1) == 0) ? g->in_Wp - g->MX : 0;
  A.npix = g->NP256;
  A.patch_bytes = (int)(((long)g->NP256 * 128 + 1023) & ~1023L);
  A.mt = (int)((M + 255) / 256);
  A.M = (int)M;
  A.in_pixels = g->N * g->in_Hp * g->in_Wp;
  A.pad_rows = (g->MP > 0 && g->MP != A.plane) ? 1 : 0;
  A.dbg = g_pw_dbg;
  long lds = pw_lds_bytes(g);
  const int grid = pw_grid(g, lds);
  if (g_pw_one_wg) lds = 96 * 1024;
  const int red = red_y ? (red_y2 ? 2 : 1) : 0;
  hipStream_t s = (hipStream_t)stream;
#define PW_LAUNCH(RD_, PR_)                                                                                   \
  do {                                                                                                        \
    static bool attr = false;                                                                                 \
    if (!attr) {                                                                                              \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_pw_kernel<RD_, PR_>),               \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                      \
      attr = true;                                                                                            \
    }                                                                                                         \
    hipLaunchKernelGGL((conv_igemm_pw_kernel<RD_, PR_>), dim3(grid), dim3(PW_THREADS), lds, s, *g, A,         \
                       (const bf16_t*)in, (const unsigned char*)wfrag, (bf16_t*)out, stats,                   \
                       (const bf16_t*)res_grad, (const bf16_t*)res_act, accumulate, (const bf16_t*)red_y,     \
                       red_coef, (const bf16_t*)red_y2, red_stats, red_stats2, g_pw_prof, g_pw_stagger);      \
  } while (0)
#ifdef IIC_DEBUG_HOOKS
  if (g_pw_prof) {
    if (red == 0) PW_LAUNCH(0, true); else if (red == 1) PW_LAUNCH(1, true); else PW_LAUNCH(2, true);
    return iic_launch_status();
  }
#endif
  if (red == 0) PW_LAUNCH(0, false); else if (red == 1) PW_LAUNCH(1, false); else PW_LAUNCH(2, false);
  return iic_launch_status();
}

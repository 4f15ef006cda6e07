// Staged reproduction of conv_igemm_bd_kernel's K loop (measurement tool, tools/mfma_feed.py; not on
// the product path).  Answers "where do the MFMA issue slots go": the same 128 x 64 wave tile
// (8 independent 32x32 accumulators), register budget (launch bounds 256 threads x 2 workgroups
// per CU = 2 waves per SIMD) and instruction stream as the product kernel, switched on one
// ingredient at a time:
//   stage 0  MFMAs only (operands fixed in registers)
//   stage 1  + the A-fragment stream: 4 ds_read_b128 per k-step, issued one k-step ahead
//   stage 2  + the B-fragment ring: 2 global_load_dwordx4 per k-step, 8 loads in flight, vmcnt(6)
//   stage 3  + chunk boundaries: every `chunk_its` iterations barrier -> LDS-DMA patch reload ->
//            vmcnt(0) -> barrier
//   stage 4  + tile epilogue: accumulators -> bf16 -> LDS -> 16-byte global stores (256 x 128 tile)
// Every workgroup also stamps s_memtime / s_memrealtime so that the tool can report the shader
// clock the chip actually held (cycles per 100 MHz tick).
#include "../common.h"
#include <type_traits>

#define MF_THREADS 256

// VAR bits: 1 = every B load from the same 8 KB (L1-hot: separates issue cost from L2 latency / bandwidth);
//   2 = s_setprio(1) around the MFMA cluster; 4 = no order pinning: reads / loads interleaved with the MFMAs
//   (sched_group_barrier) instead of [reads][8 MFMAs][loads] blocks; 8 = B ring two iterations deep (16 loads);
//   16 = per-workgroup rotation of the walk over the weight buffer
template <int STAGE, int VAR>
__global__ __launch_bounds__(MF_THREADS, 2) void mfma_feed_kernel(
    const unsigned char* __restrict__ wfrag, long wfrag_bytes, const bf16_t* __restrict__ patch,
    bf16_t* __restrict__ out, int nit, int chunk_its, int npix, unsigned long long* __restrict__ clk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, g5 = lane >> 5;
  unsigned long long t0 = 0, r0 = 0;
  if (clk) { t0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime(); }

  const int nblk = (npix * 128 + 1023) >> 10;
  auto dma_patch = [&](int c0) {
    for (int blk = wave; blk < nblk; blk += MF_THREADS / 64) {
      const int q = blk * 64 + lane;
      const int r = q >> 3;
      const int ls = (q & 7) ^ ((r >> 1) & 7);
      const long p = (long)blockIdx.x * 256 + r;       // neighbouring workgroups overlap like conv tiles
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(patch + (p * 256 + c0 + ls * 8)),
          (__attribute__((address_space(3))) void*)(smem + blk * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  if (STAGE >= 1) {
    dma_patch(0);
    __syncthreads();
  }

  f32x16 acc[4][2];
#pragma unroll
  for (int ms = 0; ms < 4; ++ms)
#pragma unroll
    for (int ns = 0; ns < 2; ++ns)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ms][ns][r] = 0.f;

  // B fragments: this wave's 2 x 4 KB per iteration, walking a weight buffer laid out as the product's
  const long frag_it = 8 * 4096;        // bytes per iteration across the 4 n-pairs of a 256-cout layer
  const unsigned char* wb0 = wfrag + (long)((blockIdx.x & 1) * 2 + wn) * 8192 + lane * 16;
  const long wmask = wfrag_bytes - 1;      // power of two
  // VAR 16: every workgroup starts its walk over the weight buffer somewhere else (all workgroups of a
  // launch otherwise read the SAME fragment lines at the same moment: one hot spot in each L2)
  const int rot = (VAR & 16) ? (int)((blockIdx.x * 2654435761u) >> 20) : 0;
  auto frag_ptr = [&](int it) { return (VAR & 1) ? wb0 : wb0 + (((long)(it + rot) * frag_it) & wmask); };
  constexpr int DEPTH = (VAR & 8) ? 2 : 1;
  u32x4 Bc[DEPTH][4][2];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    const unsigned char* p = frag_ptr(d);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int ns = 0; ns < 2; ++ns) {
        if (VAR & 32) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(Bc[d][ks][ns]) : "v"(p + ns * 4096 + ks * 1024) : "memory");
        else Bc[d][ks][ns] = *reinterpret_cast<const u32x4*>(p + ns * 4096 + ks * 1024);
      }
  }
  if (VAR & 32)      // (hipcc must never see a pending load of its own on these registers: it would wait for it in the loop)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(Bc[0][0][0]), "+v"(Bc[0][0][1]), "+v"(Bc[0][1][0]), "+v"(Bc[0][1][1]),
                 "+v"(Bc[0][2][0]), "+v"(Bc[0][2][1]), "+v"(Bc[0][3][0]), "+v"(Bc[0][3][1]) : : "memory");
  // A fragments: row R of the patch at 128-byte pitch, 16-byte slot XOR-swizzled by (R >> 1) & 7
  int arow[4];
#pragma unroll
  for (int ms = 0; ms < 4; ++ms) arow[ms] = wm * 128 + ms * 32 + l31;
  auto a_addr = [&](int R, int ks) { return R * 128 + ((((2 * ks + g5) ^ (R >> 1)) & 7) << 4); };
  bf16x8 a[2][4];
  int pc[4];
#pragma unroll
  for (int ms = 0; ms < 4; ++ms) {
    pc[ms] = arow[ms];
    if (STAGE >= 1) a[0][ms] = *reinterpret_cast<const bf16x8*>(smem + a_addr(pc[ms], 0));
    else a[0][ms] = __builtin_bit_cast(bf16x8, Bc[0][ms][0]);
    a[1][ms] = a[0][ms];
  }
  int since = 0;
  auto body = [&](int it, auto dsel) {
    constexpr int D = decltype(dsel)::value;
    const unsigned char* nb = frag_ptr(it + DEPTH);
    int pn[4];
    const int toff = ((it + 1) % 9) * 3;         // next tap's row offset (stays inside the patch)
#pragma unroll
    for (int ms = 0; ms < 4; ++ms) pn[ms] = arow[ms] + toff;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (VAR & 32) {
        // hand-ordered k-step: B loads as inline asm (invisible to hipcc's vmcnt bookkeeping) with exact
        // counted waits -- 8 loads always in flight, the one needed is the oldest --, the next k-step's
        // A reads between the first four MFMAs, ns-major MFMA order so that b0 is free after MFMA 4
        u32x4& B0 = Bc[D][ks][0];
        u32x4& B1 = Bc[D][ks][1];
        if (STAGE >= 2) asm volatile("s_waitcnt vmcnt(7)" : "+v"(B0) : : "memory");
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 b0 = __builtin_bit_cast(bf16x8, B0);
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) {
          acc[ms][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][ms], b0, acc[ms][0], 0, 0, 0);
          if (STAGE >= 1)
            a[nxt][ms] = (ks < 3) ? *reinterpret_cast<const bf16x8*>(smem + a_addr(pc[ms], ks + 1))
                                  : *reinterpret_cast<const bf16x8*>(smem + a_addr(pn[ms], 0));
          __builtin_amdgcn_sched_barrier(0);
        }
        if (STAGE >= 2) {
          asm volatile("s_waitcnt vmcnt(6)" : "+v"(B1) : : "memory");
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(B0) : "v"(nb + ks * 1024) : "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 b1 = __builtin_bit_cast(bf16x8, B1);
#pragma unroll
        for (int ms = 0; ms < 4; ++ms)
          acc[ms][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][ms], b1, acc[ms][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (STAGE >= 2)
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(B1) : "v"(nb + 4096 + ks * 1024) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        continue;
      }
      if (STAGE >= 1) {
#pragma unroll
        for (int ms = 0; ms < 4; ++ms)
          a[nxt][ms] = (ks < 3) ? *reinterpret_cast<const bf16x8*>(smem + a_addr(pc[ms], ks + 1))
                                : *reinterpret_cast<const bf16x8*>(smem + a_addr(pn[ms], 0));
      }
      if (!(VAR & 4)) __builtin_amdgcn_sched_barrier(0);
      const bf16x8 b0 = __builtin_bit_cast(bf16x8, Bc[D][ks][0]);
      const bf16x8 b1 = __builtin_bit_cast(bf16x8, Bc[D][ks][1]);
      if (VAR & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ms = 0; ms < 4; ++ms) {
        acc[ms][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][ms], b0, acc[ms][0], 0, 0, 0);
        acc[ms][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][ms], b1, acc[ms][1], 0, 0, 0);
      }
      if (VAR & 2) __builtin_amdgcn_s_setprio(0);
      if (STAGE >= 2) {
        Bc[D][ks][0] = *reinterpret_cast<const u32x4*>(nb + ks * 1024);
        Bc[D][ks][1] = *reinterpret_cast<const u32x4*>(nb + 4096 + ks * 1024);
      }
      if (VAR & 4) {
        // 1 LDS read : 2 MFMAs, the two global loads after the 4th and 8th MFMA
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (STAGE >= 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
          if (STAGE >= 2 && (q & 1)) __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ms = 0; ms < 4; ++ms) pc[ms] = pn[ms];
    if (STAGE >= 3 && ++since == chunk_its && it + 1 < nit) {
      since = 0;
      __syncthreads();
      dma_patch(((it / chunk_its) & 3) * 64);
      __syncthreads();
#pragma unroll
      for (int ms = 0; ms < 4; ++ms) a[0][ms] = *reinterpret_cast<const bf16x8*>(smem + a_addr(pc[ms], 0));
    }
  };
  for (int it = 0; it < nit; it += DEPTH) {
    body(it, std::integral_constant<int, 0>());
    if (DEPTH == 2) body(it + 1, std::integral_constant<int, DEPTH - 1>());
  }

  if (VAR & 32) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // asm loads still in flight own their registers
  if (STAGE >= 4) {
    constexpr int CLD = 128 + 8;
    bf16_t* sC = reinterpret_cast<bf16_t*>(smem);
    __syncthreads();
#pragma unroll
    for (int ms = 0; ms < 4; ++ms)
#pragma unroll
      for (int ns = 0; ns < 2; ++ns)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * 128 + ms * 32 + mfma32_row(r, lane);
          const int col = wn * 64 + ns * 32 + l31;
          sC[row * CLD + col] = f32_to_bf16(acc[ms][ns][r]);
        }
    __syncthreads();
    for (int i = tid; i < 256 * 16; i += MF_THREADS) {
      const int row = i >> 4, ch = i & 15;
      *reinterpret_cast<uint4*>(out + ((long)blockIdx.x * 256 + row) * 128 + ch * 8) =
          *reinterpret_cast<const uint4*>(sC + row * CLD + ch * 8);
    }
  } else {
    float t = 0.f;
#pragma unroll
    for (int ms = 0; ms < 4; ++ms)
#pragma unroll
      for (int ns = 0; ns < 2; ++ns)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[ms][ns][r];
    if (t == 123.456f) out[0] = 1;
  }
  if (clk && tid == 0) {
    clk[blockIdx.x * 4 + 0] = t0;
    clk[blockIdx.x * 4 + 1] = __builtin_readcyclecounter();
    clk[blockIdx.x * 4 + 2] = r0;
    clk[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
  }
}

// stage 0..4 as above; lds_bytes also sets the occupancy (<= 80 KB: two workgroups per CU, more: one).
// wfrag: >= 64 KB and a multiple of 32 KB; patch: (grid * 256 + npix) rows of 256 bf16; out: grid * 256 * 128 bf16;
// clk: grid * 4 uint64 or NULL.
extern "C" int iic_debug_mfma_feed(int stage_var, int grid, int nit, int chunk_its, int npix, int lds_bytes,
                                   const void* wfrag, long wfrag_bytes, const void* patch, void* out,
                                   void* clk, void* stream) {
  if (grid <= 0 || nit <= 0 || chunk_its <= 0 || npix < 256 + 32 || lds_bytes < npix * 128 ||
      lds_bytes < 256 * 136 * 2 || lds_bytes > 160 * 1024 || wfrag_bytes < 65536 || (wfrag_bytes & (wfrag_bytes - 1)))
    return IIC_ERR_ARG;
  if (nit & 1) return IIC_ERR_ARG;        // (the two-deep ring variant steps two iterations at a time)
  const int stage = stage_var & 15, var = stage_var >> 4;
  hipStream_t s = (hipStream_t)stream;
#define MF_LAUNCH(ST_, VA_)                                                                          \
  do {                                                                                               \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_feed_kernel<ST_, VA_>),            \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);               \
    hipLaunchKernelGGL((mfma_feed_kernel<ST_, VA_>), dim3(grid), dim3(MF_THREADS), lds_bytes, s,     \
                       (const unsigned char*)wfrag, wfrag_bytes, (const bf16_t*)patch, (bf16_t*)out, \
                       nit, chunk_its, npix, (unsigned long long*)clk);                              \
  } while (0)
#define MF_STAGES(VA_)                                                                               \
  switch (stage) {                                                                                   \
    case 0: MF_LAUNCH(0, VA_); break;                                                                \
    case 1: MF_LAUNCH(1, VA_); break;                                                                \
    case 2: MF_LAUNCH(2, VA_); break;                                                                \
    case 3: MF_LAUNCH(3, VA_); break;                                                                \
    case 4: MF_LAUNCH(4, VA_); break;                                                                \
    default: return IIC_ERR_ARG;                                                                     \
  }
  switch (var) {
    case 0: MF_STAGES(0); break;
    case 1: MF_STAGES(1); break;
    case 2: MF_STAGES(2); break;
    case 4: MF_STAGES(4); break;
    case 6: MF_STAGES(6); break;
    case 8: MF_STAGES(8); break;
    case 16: MF_STAGES(16); break;
    case 20: MF_STAGES(20); break;
    case 32: MF_STAGES(32); break;
    default: return IIC_ERR_ARG;
  }
  return iic_launch_status();
}

// SegmentationNet10a head pieces for gfx950 (fp32): the 1x1 conv with padding = 1 is a plain
// GEMM over the (Hf+2) x (Wf+2) window of the PT feature tensor (its first border ring IS the
// conv's zero padding), followed by Softmax2d and bilinear up-sampling to input_sz.
//
// Replaces /root/reference/code/archs/segmentation/net10a.py:44-59
// (nn.Conv2d(1x1, padding=1, bias=False) + nn.Softmax2d + F.interpolate(bilinear)).
//   seg_window_gather : PT bf16 window -> fp32 [M][C] matrix (A operand of iic_gemm_f32)
//   seg_window_scatter: fp32 [M][C] feature gradient -> PT bf16 interior
//   bilinear_fwd / bwd: [N][Hl][Wl][k] (pixel-major) <-> [N][k][S][S] (NCHW, what the loss takes),
//                       align_corners = False (torch default), backward in gather form.
#include "common.h"
#include "../../include/iic_hip.h"

__global__ __launch_bounds__(256) void seg_window_gather_kernel(const bf16_t* __restrict__ pt,
                                                                float* __restrict__ out, int N,
                                                                int Hw, int Ww, int Hp, int Wp,
                                                                int off, int C) {
  const long total = (long)N * Hw * Ww * (C / 8);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % (C / 8));
    const long m = i / (C / 8);
    const int xx = (int)(m % Ww);
    const long r = m / Ww;
    const int yy = (int)(r % Hw), n = (int)(r / Hw);
    const uint4 v = *reinterpret_cast<const uint4*>(
        pt + (((long)n * Hp + yy + off) * Wp + xx + off) * C + c8 * 8);
    float4 a = make_float4(bf16lo(v.x), bf16hi(v.x), bf16lo(v.y), bf16hi(v.y));
    float4 b = make_float4(bf16lo(v.z), bf16hi(v.z), bf16lo(v.w), bf16hi(v.w));
    float* o = out + m * C + c8 * 8;
    *reinterpret_cast<float4*>(o) = a;
    *reinterpret_cast<float4*>(o + 4) = b;
  }
}

// window rows (yy, xx) with 1 <= yy <= Hw-2, 1 <= xx <= Ww-2 are the PT interior
__global__ __launch_bounds__(256) void seg_window_scatter_kernel(const float* __restrict__ in,
                                                                 bf16_t* __restrict__ pt, int N,
                                                                 int Hw, int Ww, int Hp, int Wp,
                                                                 int off, int C) {
  const int Hi = Hw - 2, Wi = Ww - 2;
  const long total = (long)N * Hi * Wi * (C / 8);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % (C / 8));
    const long q = i / (C / 8);
    const int x = (int)(q % Wi);
    const long r = q / Wi;
    const int y = (int)(r % Hi), n = (int)(r / Hi);
    const long m = ((long)n * Hw + y + 1) * Ww + x + 1;
    const float* s = in + m * C + c8 * 8;
    const float4 a = *reinterpret_cast<const float4*>(s), b = *reinterpret_cast<const float4*>(s + 4);
    *reinterpret_cast<uint4*>(pt + (((long)n * Hp + y + 1 + off) * Wp + x + 1 + off) * C + c8 * 8) =
        make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y),
                   pack_bf16x2(b.z, b.w));
  }
}

__device__ __forceinline__ void bl_src(int d, float scale, int in, int& i0, int& i1, float& lam) {
  float s = ((float)d + 0.5f) * scale - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  lam = s - (float)i0;
}

// out[n][c][y][x] = bilinear(in[n][.][.][c])
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const float* __restrict__ in,
                                                           float* __restrict__ out, int N, int Hl,
                                                           int Wl, int k, int S) {
  const long total = (long)N * k * S * S;
  const float sy = (float)Hl / (float)S, sx = (float)Wl / (float)S;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % S);
    long r = i / S;
    const int y = (int)(r % S);
    r /= S;
    const int c = (int)(r % k), n = (int)(r / k);
    int y0, y1, x0, x1;
    float ly, lx;
    bl_src(y, sy, Hl, y0, y1, ly);
    bl_src(x, sx, Wl, x0, x1, lx);
    const float* b = in + (long)n * Hl * Wl * k + c;
    const float v00 = b[((long)y0 * Wl + x0) * k], v01 = b[((long)y0 * Wl + x1) * k];
    const float v10 = b[((long)y1 * Wl + x0) * k], v11 = b[((long)y1 * Wl + x1) * k];
    out[i] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
  }
}

// din[n][yl][xl][c] = sum over the output pixels whose footprint touches (yl, xl)
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ dout,
                                                           float* __restrict__ din, int N, int Hl,
                                                           int Wl, int k, int S) {
  const long total = (long)N * Hl * Wl * k;
  const float sy = (float)Hl / (float)S, sx = (float)Wl / (float)S;
  const float ry = (float)S / (float)Hl, rx = (float)S / (float)Wl;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % k);
    long r = i / k;
    const int xl = (int)(r % Wl);
    r /= Wl;
    const int yl = (int)(r % Hl), n = (int)(r / Hl);
    // candidate destination range: src(d) in (l-1, l+1)
    int ya = (int)floorf(((float)yl - 1.f + 0.5f) * ry - 0.5f) - 1, yb = (int)ceilf(((float)yl + 1.f + 0.5f) * ry - 0.5f) + 1;
    int xa = (int)floorf(((float)xl - 1.f + 0.5f) * rx - 0.5f) - 1, xb = (int)ceilf(((float)xl + 1.f + 0.5f) * rx - 0.5f) + 1;
    if (ya < 0) ya = 0;
    if (xa < 0) xa = 0;
    if (yb > S - 1) yb = S - 1;
    if (xb > S - 1) xb = S - 1;
    const float* g = dout + ((long)n * k + c) * S * S;
    float acc = 0.f;
    for (int y = ya; y <= yb; ++y) {
      int y0, y1;
      float ly;
      bl_src(y, sy, Hl, y0, y1, ly);
      float wy = 0.f;
      if (y0 == yl) wy += 1.f - ly;
      if (y1 == yl) wy += ly;
      if (wy == 0.f) continue;
      for (int x = xa; x <= xb; ++x) {
        int x0, x1;
        float lx;
        bl_src(x, sx, Wl, x0, x1, lx);
        float wx = 0.f;
        if (x0 == xl) wx += 1.f - lx;
        if (x1 == xl) wx += lx;
        if (wx != 0.f) acc += wy * wx * g[(long)y * S + x];
      }
    }
    din[i] = acc;
  }
}

static int grid_for(long total) {
  long g = (total + 255) / 256;
  if (g > 16384) g = 16384;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" {

int iic_seg_window_gather(const void* pt, float* out, int N, int Hw, int Ww, int Hp, int Wp, int off,
                          int C, void* stream) {
  if (!pt || !out || N <= 0 || (C & 7) || off < 0 || Hw + off > Hp || Ww + off > Wp) return IIC_ERR_ARG;
  hipLaunchKernelGGL(seg_window_gather_kernel, dim3(grid_for((long)N * Hw * Ww * (C / 8))), dim3(256),
                     0, (hipStream_t)stream, (const bf16_t*)pt, out, N, Hw, Ww, Hp, Wp, off, C);
  return iic_launch_status();
}

int iic_seg_window_scatter(const float* in, void* pt, int N, int Hw, int Ww, int Hp, int Wp, int off,
                           int C, void* stream) {
  if (!pt || !in || N <= 0 || (C & 7) || off < 0 || Hw < 3 || Ww < 3) return IIC_ERR_ARG;
  hipLaunchKernelGGL(seg_window_scatter_kernel, dim3(grid_for((long)N * (Hw - 2) * (Ww - 2) * (C / 8))),
                     dim3(256), 0, (hipStream_t)stream, in, (bf16_t*)pt, N, Hw, Ww, Hp, Wp, off, C);
  return iic_launch_status();
}

int iic_bilinear_fwd(const float* in_nhwc, float* out_nchw, int N, int Hl, int Wl, int k, int S,
                     void* stream) {
  if (!in_nhwc || !out_nchw || N <= 0 || Hl <= 0 || Wl <= 0 || k <= 0 || S <= 0) return IIC_ERR_ARG;
  hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(grid_for((long)N * k * S * S)), dim3(256), 0,
                     (hipStream_t)stream, in_nhwc, out_nchw, N, Hl, Wl, k, S);
  return iic_launch_status();
}

int iic_bilinear_bwd(const float* dout_nchw, float* din_nhwc, int N, int Hl, int Wl, int k, int S,
                     void* stream) {
  if (!dout_nchw || !din_nhwc || N <= 0 || Hl <= 0 || Wl <= 0 || k <= 0 || S <= 0) return IIC_ERR_ARG;
  hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(grid_for((long)N * Hl * Wl * k)), dim3(256), 0,
                     (hipStream_t)stream, dout_nchw, din_nhwc, N, Hl, Wl, k, S);
  return iic_launch_status();
}

}  // extern "C"

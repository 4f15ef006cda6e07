// GPU-side paired augmentation for the clustering scripts.
//
// Replaces the per-sample PIL pipeline the reference builds in
//   /root/reference/code/utils/cluster/transforms.py:107-217 (sobel_make_transforms, default
//   branch) out of torchvision 0.2.1 transforms:
//     RandomCrop -> Resize (PIL BILINEAR) -> [RandomHorizontalFlip -> ColorJitter] ->
//     custom_greyscale_to_tensor (:12-25)
// with ONE kernel: a workgroup produces one output image entirely in LDS (crop 84x84x3 ->
// 96x96x3 is < 52 KB).  The arithmetic is PIL's, reproduced bit for bit (specification and
// cross-check: oracle/augment_oracle.py::np_pipeline):
//   * resize: Pillow's two-pass convolution resampling -- horizontal then vertical pass, 22-bit
//     fixed-point weights (computed on the host exactly like precompute_coeffs /
//     normalize_coeffs_8bpc), 8-bit rounding after each pass;
//   * brightness / contrast / saturation: ImageEnhance = ImagingBlend(degenerate, image, factor):
//     (float)in1 + factor * (float)(in2 - in1) in float32 WITHOUT fused multiply-add, truncated,
//     clipped; contrast's degenerate is the rounded mean of the L image, saturation's the L image;
//   * hue: Pillow's RGB -> HSV -> RGB round trip (float / double mix of Convert.c) with the uint8
//     wrap-around hue shift of torchvision's adjust_hue;
//   * grey: L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16; to_tensor: value / 255 in float32
//     (256-entry table from the host).
// The same kernel serves the greyscale scripts' pipelines (transforms.py:220-330, mode "L" images,
// CH = 1): optional RandomRotation first (PIL Image.rotate, NEAREST = ImagingTransformAffine's
// 16.16 fixed-point inverse mapping, zero fill; the coefficients come from the host), a crop size
// chosen per image out of a few (one resampling table per size), ToTensor at the end.  On an L
// image ColorJitter's saturation and hue are identities (ImageEnhance.Color blends the image with
// itself; adjust_hue returns L images unchanged).
// The random parameters (rotation, crop size and offsets, flip, jitter factors and their shuffled
// order) are inputs: iic_amd/augment.py draws them with torchvision's distributions.
#include "common.h"
#include "../../include/iic_hip.h"

// PIL's C code runs as separate IEEE multiplies and adds (x86-64 builds without FMA): a fused
// a*b+c rounds once instead of twice and flips truncations.  hipcc contracts by default
// (-ffp-contract=fast; the *_rn intrinsics are plain operators in its headers too), so contraction is
// switched off for this translation unit and every step below is written as its own operation.
#pragma clang fp contract(off)

#define AUG_PREC 22
#define AUG_IP 20      // ints per output image:  src, x0, y0, flip, nops, op[4], hue_delta, table,
                       //                         rotate?, a0..a5 (16.16 fixed point),
                       //                         cutout box (left | upper << 16), (right | lower << 16)
                       //                         in crop coordinates (custom_cutout, transforms.py:28-44:
                       //                         img.paste(0, box) on the cropped image; right == left: none)
#define AUG_FP 4       // floats per output image: factor of op 0 (brightness), 1 (contrast), 2 (saturation), -

__device__ __forceinline__ int aug_luma(int r, int g, int b) {
  return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16;
}
// ImagingBlend for one channel value
__device__ __forceinline__ int aug_blend(int in1, int in2, float alpha) {
  const float t = (float)in1 + alpha * (float)(in2 - in1);
  if (t <= 0.f) return 0;
  if (t >= 255.f) return 255;
  return (int)t;
}
__device__ __forceinline__ int aug_clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// Pillow Convert.c rgb2hsv_row / hsv2rgb_row with the hue shift in between
__device__ __forceinline__ void aug_hue(int& r, int& g, int& b, int delta) {
  const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
  int uh = 0, us = 0;
  const int uv = maxc;
  if (minc != maxc) {
    const float cr = (float)(maxc - minc);
    const float s = cr / (float)maxc;
    const float rc = (float)(maxc - r) / cr;
    const float gc = (float)(maxc - g) / cr;
    const float bc = (float)(maxc - b) / cr;
    float h;
    if (r == maxc) h = (float)((double)bc - (double)gc);
    else if (g == maxc) h = (float)((2.0 + (double)rc) - (double)bc);
    else h = (float)((4.0 + (double)gc) - (double)rc);
    const double x = (double)h / 6.0 + 1.0;     // in [5/6, 11/6]
    h = (float)(x - floor(x));                                       // fmod(x, 1.0), exact
    uh = aug_clip8((int)((double)h * 255.0));
    us = aug_clip8((int)((double)s * 255.0));
  }
  uh = (uh + delta) & 255;
  if (us == 0) {
    r = g = b = uv;
    return;
  }
  const double hh = (double)uh * 6.0 / 255.0;
  const int i = (int)floor(hh);
  const double f = (double)(float)(hh - (double)i);
  const double fs = (double)(float)((double)us / 255.0);
  const double v = (double)uv;
  const int p = aug_clip8((int)round(v * (1.0 - fs)));
  const int q = aug_clip8((int)round(v * (1.0 - fs * f)));
  const int t = aug_clip8((int)round(v * (1.0 - fs * (1.0 - f))));
  switch (i % 6) {
    case 0: r = uv; g = t; b = p; break;
    case 1: r = q; g = uv; b = p; break;
    case 2: r = p; g = uv; b = t; break;
    case 3: r = p; g = q; b = uv; break;
    case 4: r = t; g = p; b = uv; break;
    default: r = uv; g = p; b = q; break;
  }
}

struct AugTabs {           // one entry per crop size: Pillow coefficient table of crop -> S
  int crop[IIC_AUG_MAX_TABLES], ksize[IIC_AUG_MAX_TABLES], boff[IIC_AUG_MAX_TABLES], koff[IIC_AUG_MAX_TABLES];
};

template <int CH, bool INC_RGB>
__global__ __launch_bounds__(256) void augment_kernel(
    const uint8_t* __restrict__ imgs, int H, int W, const int* __restrict__ iparams,
    const float* __restrict__ fparams, const AugTabs tabs, const int* __restrict__ bounds,
    const int* __restrict__ kk, int S, const float* __restrict__ lut, float* __restrict__ out,
    const float* __restrict__ norm) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ int s_red[4];
  const int n = blockIdx.x, tid = threadIdx.x;
  const int* ip = iparams + (long)n * AUG_IP;
  const float* fp = fparams + (long)n * AUG_FP;
  const int src = ip[0], x0 = ip[1], y0 = ip[2], flip = ip[3], nops = ip[4], hdelta = ip[9];
  const int tab = ip[10], rot = ip[11];
  const int a0 = ip[12], a1 = ip[13], a2 = ip[14], a3 = ip[15], a4 = ip[16], a5 = ip[17];
  const int bx0 = ip[18] & 0xffff, by0 = (ip[18] >> 16) & 0xffff;
  const int bx1 = ip[19] & 0xffff, by1 = (ip[19] >> 16) & 0xffff;
  const int crop = tabs.crop[tab], KS = tabs.ksize[tab];
  const int* bnd = bounds + tabs.boff[tab] * 2;
  const int* kt = kk + tabs.koff[tab];
  uint8_t* sB = smem_raw;                                    // [crop][S][CH] after the horizontal pass
  uint8_t* sC = smem_raw + ((crop * S * CH + 15) & ~15);     // [S][S][CH]
  const uint8_t* im = imgs + (long)src * H * W * CH;

  // ---- horizontal pass (global [-> rotation gather] -> sB)
  for (int idx = tid; idx < crop * S * CH; idx += 256) {
    const int c = idx % CH, xo = (idx / CH) % S, y = idx / (CH * S);
    const int xmin = bnd[xo * 2], cnt = bnd[xo * 2 + 1];
    int ss = 1 << (AUG_PREC - 1);
    const int Y = y0 + y, X = x0 + xmin;
    const bool cut_row = y >= by0 && y < by1;       // cutout: crop pixels inside the box read as 0
    if (rot) {
      int xx = a2 + a1 * Y + a0 * X, yy = a5 + a4 * Y + a3 * X;
      for (int k = 0; k < cnt; ++k) {
        const int xin = xx >> 16, yin = yy >> 16;
        int v = (xin >= 0 && xin < W && yin >= 0 && yin < H) ? (int)im[((long)yin * W + xin) * CH + c] : 0;
        if (cut_row && xmin + k >= bx0 && xmin + k < bx1) v = 0;
        ss += v * kt[xo * KS + k];
        xx += a0;
        yy += a3;
      }
    } else {
      const uint8_t* row = im + ((long)Y * W + X) * CH + c;
      for (int k = 0; k < cnt; ++k) {
        int v = (int)row[k * CH];
        if (cut_row && xmin + k >= bx0 && xmin + k < bx1) v = 0;
        ss += v * kt[xo * KS + k];
      }
    }
    sB[idx] = (uint8_t)aug_clip8(ss >> AUG_PREC);
  }
  __syncthreads();
  // ---- vertical pass (sB -> sC)
  for (int idx = tid; idx < S * S * CH; idx += 256) {
    const int c = idx % CH, xo = (idx / CH) % S, yo = idx / (CH * S);
    const int ymin = bnd[yo * 2], cnt = bnd[yo * 2 + 1];
    int ss = 1 << (AUG_PREC - 1);
    for (int k = 0; k < cnt; ++k) ss += (int)sB[((ymin + k) * S + xo) * CH + c] * kt[yo * KS + k];
    sC[idx] = (uint8_t)aug_clip8(ss >> AUG_PREC);
  }
  __syncthreads();
  // ---- ColorJitter ops in their shuffled order (pointwise on whole pixels, in place)
  for (int o = 0; o < nops; ++o) {
    const int op = ip[5 + o];
    if (CH == 1 && op >= 2) continue;                // saturation / hue: identities on an L image
    int mean = 0;
    if (op == 1) {                                   // contrast: rounded mean of the L image
      int part = 0;
      for (int px = tid; px < S * S; px += 256)
        part += CH == 1 ? (int)sC[px] : aug_luma(sC[px * 3], sC[px * 3 + 1], sC[px * 3 + 2]);
#pragma unroll
      for (int sft = 32; sft > 0; sft >>= 1) part += __shfl_xor(part, sft, 64);
      if ((tid & 63) == 0) s_red[tid >> 6] = part;
      __syncthreads();
      const int tot = s_red[0] + s_red[1] + s_red[2] + s_red[3];
      mean = (int)((double)tot / (double)(S * S) + 0.5);
      __syncthreads();
    }
    const float alpha = op < 3 ? fp[op] : 0.f;
    for (int px = tid; px < S * S; px += 256) {
      if (CH == 1) {
        sC[px] = (uint8_t)aug_blend(op == 0 ? 0 : mean, sC[px], alpha);
        continue;
      }
      int r = sC[px * 3], g = sC[px * 3 + 1], b = sC[px * 3 + 2];
      if (op == 0) {
        r = aug_blend(0, r, alpha); g = aug_blend(0, g, alpha); b = aug_blend(0, b, alpha);
      } else if (op == 1) {
        r = aug_blend(mean, r, alpha); g = aug_blend(mean, g, alpha); b = aug_blend(mean, b, alpha);
      } else if (op == 2) {
        const int L = aug_luma(r, g, b);
        r = aug_blend(L, r, alpha); g = aug_blend(L, g, alpha); b = aug_blend(L, b, alpha);
      } else {
        aug_hue(r, g, b, hdelta);
      }
      sC[px * 3] = (uint8_t)r; sC[px * 3 + 1] = (uint8_t)g; sC[px * 3 + 2] = (uint8_t)b;
    }
    __syncthreads();
  }
  // ---- custom_greyscale_to_tensor / ToTensor (+ the horizontal flip, which commutes with the ops above)
  constexpr int C = CH == 1 ? 1 : (INC_RGB ? 4 : 1);
  float* on = out + (long)n * C * S * S;
  for (int px = tid; px < S * S; px += 256) {
    const int y = px / S, x = px - y * S;
    const int sp = (y * S + (flip ? S - 1 - x : x)) * CH;
    // norm (nullable): torchvision Normalize after the tensor conversion, (v - mean[c]) / std[c]
    // as two float32 operations (t.sub_(m).div_(s)); norm = [C means][C stds]
    if (CH == 1) {
      float v = lut[sC[sp]];
      if (norm) v = (v - norm[0]) / norm[1];
      on[px] = v;
      continue;
    }
    const int r = sC[sp], g = sC[sp + 1], b = sC[sp + 2];
    if (INC_RGB) {
      float vr = lut[r], vg = lut[g], vb = lut[b];
      if (norm) {
        vr = (vr - norm[0]) / norm[C + 0];
        vg = (vg - norm[1]) / norm[C + 1];
        vb = (vb - norm[2]) / norm[C + 2];
      }
      on[px] = vr;
      on[S * S + px] = vg;
      on[2 * S * S + px] = vb;
    }
    float vl = lut[aug_luma(r, g, b)];
    if (norm) vl = (vl - norm[C - 1]) / norm[2 * C - 1];
    on[(C - 1) * S * S + px] = vl;
  }
}

extern "C" {

int iic_augment(const void* imgs_u8, int B, int H, int W, int channels, const int* iparams,
                const float* fparams, int N, const int* tables_host, int n_tables,
                const int* bounds, const int* kk, int S, const float* lut, float* out,
                int include_rgb, const float* norm, void* stream) {
  if (!imgs_u8 || !iparams || !fparams || !tables_host || !bounds || !kk || !lut || !out) return IIC_ERR_ARG;
  if (B <= 0 || N <= 0 || S <= 0 || H <= 0 || W <= 0) return IIC_ERR_ARG;
  if (channels != 1 && channels != 3) return IIC_ERR_UNSUPPORTED;
  if (n_tables < 1 || n_tables > IIC_AUG_MAX_TABLES) return IIC_ERR_ARG;
  AugTabs tabs;
  int max_crop = 0;
  for (int t = 0; t < IIC_AUG_MAX_TABLES; ++t) {
    const int* e = tables_host + 4 * (t < n_tables ? t : 0);
    if (e[0] <= 0 || e[0] > H || e[0] > W || e[1] <= 0 || e[2] < 0 || e[3] < 0) return IIC_ERR_ARG;
    tabs.crop[t] = e[0]; tabs.ksize[t] = e[1]; tabs.boff[t] = e[2]; tabs.koff[t] = e[3];
    max_crop = e[0] > max_crop ? e[0] : max_crop;
  }
  const size_t lds = ((size_t)(max_crop * S * channels + 15) & ~(size_t)15) + (size_t)S * S * channels;
  if (lds > 150 * 1024) return IIC_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
#define AUG_LAUNCH(CH_, RGB_)                                                                     \
  do {                                                                                           \
    static bool attr = false; /* one-time: raise the dynamic LDS limit (static s_red on top) */   \
    if (!attr) {                                                                                 \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&augment_kernel<CH_, RGB_>),         \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) !=         \
          hipSuccess) {                                                                          \
        (void)hipGetLastError();                                                                 \
        return IIC_ERR_UNSUPPORTED;                                                              \
      }                                                                                          \
      attr = true;                                                                               \
    }                                                                                            \
    hipLaunchKernelGGL((augment_kernel<CH_, RGB_>), dim3(N), dim3(256), lds, s,                   \
                       (const uint8_t*)imgs_u8, H, W, iparams, fparams, tabs, bounds, kk, S, lut, \
                       out, norm);                                                               \
  } while (0)
  if (channels == 1) AUG_LAUNCH(1, false);
  else if (include_rgb) AUG_LAUNCH(3, true);
  else AUG_LAUNCH(3, false);
  return iic_launch_status();
}

}  // extern "C"

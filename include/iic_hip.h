/* libiic_hip.so -- C ABI of the MI355X-native (gfx950) IIC training hot path.
 *
 * The reference (xu-ji/IIC, PyTorch 0.4.1) has no FFI layer: its operator API is a
 * set of Python call signatures.  This library sits beneath Python
 * shims that reproduce those signatures (the iic_amd Python package); every entry point below names
 * the reference code it replaces.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless
 *     a parameter is documented as host;
 *   - the caller owns every buffer (workspace sizes via *_bytes helpers);
 *   - kernels are enqueued asynchronously on `stream` (a hipStream_t passed as void*),
 *     no hidden synchronisation or allocation;
 *   - return 0 on success, negative on error (IIC_ERR_*); nothing throws.
 *
 * Activation tensor format "PT" (padded tile): bf16 [N][H+2P][W+2P][C] with a ZERO
 * border of P pixels.  Kernels only ever write interior pixels, so a buffer zeroed
 * once keeps its border.  C must be a multiple of 64 for the MFMA conv kernels.
 */
#ifndef IIC_HIP_H
#define IIC_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* The prototypes below are the library's WHOLE exported surface: libiic_hip.so is built with -fvisibility=hidden
 * and only what this header declares is visible (tests/test_cabi_cpu.py holds `nm -D` to it, both ways). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define IIC_OK 0
#define IIC_ERR_ARG (-1)
#define IIC_ERR_LAUNCH (-2)
#define IIC_ERR_UNSUPPORTED (-3)

int iic_version(void);

/* ---------------------------------------------------------------------------------
 * IID clustering loss -- replaces code/utils/cluster/IID_losses.py:6-47
 * (IID_loss + compute_joint).  z / zt: post-softmax fp32, element (h, n, i) at
 * z[h*head_stride + n*ld + i] (ld >= k: sub-heads may be interleaved per sample).  Two-phase so that the raw joint can be all-reduced over ranks
 *  between phase 1 and phase 2.
 * ------------------------------------------------------------------------------- */
int iic_iid_nsplit(int bn);                       /* recommended sample-splits          */
long iic_iid_workspace_bytes(int H, int k);       /* float64 scratch for phase 2        */
/* phase 1: partials[s][h][k][k] = sum over sample-split s of z^T z' (exact fp32 MFMA) */
int iic_iid_joint_raw(const float* z, const float* zt, float* partials, int H, int bn, int k,
                      long head_stride, long ld, int nsplit, void* stream);
/* phase 2: sum partials -> symmetrise, normalise, marginals, clamp (eps), the two losses
 * (IID_losses.py:21-31) and dLoss/dR, dLossNoLamb/dR ([H][k][k] fp32 each).            */
int iic_iid_loss_from_joint(const float* partials, int nparts, int H, int k, double lamb,
                            double eps, void* workspace, float* loss, float* loss_no_lamb,
                            float* dR_loss, float* dR_loss_no_lamb, void* stream);
/* phase 3: dz = (g*dR1 + gnl*dR2) applied to z' (and z for dz'); g_* are DEVICE arrays
 * [H] of upstream gradients (NULL => 1 and 0).                                        */
int iic_iid_grad(const float* z, const float* zt, const float* dR_loss,
                 const float* dR_loss_no_lamb, const float* g_loss, const float* g_loss_no_lamb,
                 float* dz, float* dzt, int H, int bn, int k, long head_stride, long ld,
                 void* stream);

/* ---------------------------------------------------------------------------------
 * IID segmentation losses -- replace code/utils/segmentation/IID_losses.py:14-159
 * (IID_segmentation_loss, IID_segmentation_loss_uncollapsed) incl. perform_affine_tf for the
 * identity / axis-flip transforms (transforms.py:131-143; `flips` = int32 [bn][2]: flip x, y).
 * x1, x2: fp32 NCHW [bn][k][h][w] (post-softmax), mask fp32 [bn][h][w], T = half_T_side_dense.
 *   partials[s][p][q][i][j] = sum over row-slice s of x1m[i](y+p-T, x+q-T) * x2m[j](y, x)
 * Exact fp32 MFMA (16x16x4).  k <= 48, T <= 10, w <= 256.  Rows with w % 4 == 0, k <= 32 and
 * 16-byte aligned tensors take the streaming kernels (float4 rows prefetched under the MFMA loop);
 * anything else the element-wise generic ones -- same results (bit-identical joint).
 * ------------------------------------------------------------------------------- */
int iic_seg_joint_nsplit(int bn, int h, int k, int T);
int iic_seg_joint_raw(const float* x1, const float* x2, const float* mask, const int* flips,
                      float* partials, int bn, int k, int h, int w, int T, int nsplit,
                      void* stream);
/* k x k stage: uncollapsed = one joint per shift (H = (2T+1)^2, nparts = nsplit);
 * collapsed = shifts summed (H = 1, nparts = nsplit*(2T+1)^2, detach_norm = 1, :60).
 * Workspace / outputs as iic_iid_loss_from_joint.                                          */
int iic_seg_loss_from_joint(const float* partials, int nparts, int H, int k, double lamb,
                            double eps, void* workspace, float* loss, float* loss_no_lamb,
                            float* dR_loss, float* dR_loss_no_lamb, int detach_norm, void* stream);
/* which = 0: out = dLoss/dx1 (src = x2);  which = 1: out = dLoss/dx2 (src = x1).
 * g_loss / g_loss_no_lamb: DEVICE arrays [H] of upstream gradients per shift ([1] if collapsed).
 * workspace: iic_seg_grad_workspace_bytes(k, T) bytes, 16-byte aligned, caller-owned scratch of
 * this launch (the per-shift gradient matrices laid out for the streaming kernel); NULL selects
 * the generic kernel (any w, k <= 48), which needs none.                                     */
long iic_seg_grad_workspace_bytes(int k, int T);
int iic_seg_grad(const float* src, const float* mask, const int* flips, const float* dR_loss,
                 const float* dR_loss_no_lamb, const float* g_loss, const float* g_loss_no_lamb,
                 float* out, int bn, int k, int h, int w, int T, int which, int collapsed,
                 float* workspace, void* stream);
/* General case of the second view's warp -- perform_affine_tf (code/utils/segmentation/
 * transforms.py:131-143: affine_grid + grid_sample, bilinear, zero padding) followed by the
 * whole-batch integer shift of random_translation_multiple (transforms.py:145-165; IID_losses.py:
 * 101-104).  x, out, dout, dx: fp32 NCHW [N][K][H][W]; pixel_mats: fp32 [N][6], source pixel
 * (ix, iy) = M * (ox + shift_x, oy + shift_y, 1) (the host derives M from theta in normalised
 * coordinates); output pixels whose shifted position leaves the image are 0.  Identity / flip
 * matrices without shift never come here: they are index arithmetic inside iic_seg_joint_raw /
 * iic_seg_grad (flips).  iic_affine_warp_bwd overwrites dx.                                 */
int iic_affine_warp_fwd(const float* x, const float* pixel_mats, float* out, int N, int K, int H, int W,
                        int shift_x, int shift_y, void* stream);
int iic_affine_warp_bwd(const float* dout, const float* pixel_mats, float* dx, int N, int K, int H, int W,
                        int shift_x, int shift_y, void* stream);

/* ---------------------------------------------------------------------------------
 * Convolution as an im2col-free implicit GEMM on bf16 MFMA (fp32 accumulate).
 * Replaces the cuDNN conv fwd / bwd-data / bwd-weight the reference reaches through
 * nn.Conv2d in code/archs/cluster/residual.py:4-7,19,22,54-55, net5g.py:21-23,
 * vgg.py:24-26.  One kernel serves forward and backward-data: the geometry maps GEMM
 * row m=(n,y,x) to an input pixel and an output pixel, and lists the taps.
 * ------------------------------------------------------------------------------- */
#define IIC_MAX_TAPS 32
typedef struct {
  int32_t N, MY, MX;            /* GEMM rows M = N*MY*MX, m -> (n, y, x)                       */
  int32_t in_Hp, in_Wp, Cin;    /* input  PT tensor dims (padded)                              */
  int32_t sy, sx, oy, ox;       /* input pixel of row m, tap offset 0:                         */
                                /*   (n*in_Hp + y*sy+oy)*in_Wp + x*sx+ox                       */
  int32_t out_Hp, out_Wp, Cout; /* output PT tensor dims (padded)                              */
  int32_t ty, tx, py, px;       /* output pixel (n*out_Hp + y*ty+py)*out_Wp + x*tx+px          */
  int32_t ntaps;
  int32_t tap_off[IIC_MAX_TAPS];/* >=0, added to the input pixel index                         */
  int32_t tap_w[IIC_MAX_TAPS];  /* which [Cout][Cin] slice of the weight tensor the tap uses   */
  int32_t NP;                   /* LDS patch pixels per 128-row tile (max input span + 1)      */
  int32_t NP256;                /* same for 256-row tiles (0 = unknown: 128-row tiles only)    */
  int32_t NP64;                 /* same for 64-row tiles  (0 = unknown)                         */
  int32_t MP;                   /* GEMM rows per image: 0 = MY*MX (dense), else a multiple of   */
                                /* 256 >= MY*MX; rows r >= MY*MX of an image are invalid (never */
                                /* stored, no statistics, zero weight gradient): tiles of large */
                                /* images then never straddle two images                        */
} iic_conv_geom;

long iic_conv_lds_bytes(const iic_conv_geom* g, int BN);
/* out[pout(m)][co] = sum_t sum_ci in[pin(m)+tap_off[t]][ci] * w[tap_w[t]][co][ci]
 * w: bf16 [wtaps][Cout][Cin].  stats (nullable): a statistics accumulator of iic_stat_bytes(Cout)
 * bytes, zero-initialised once by the caller (the finalisers re-zero it); += per-channel sum /
 * sum of squares of the fp32 accumulators (BatchNorm batch stats).  The accumulator is opaque:
 * [IIC_STAT_STRIPES][Cout][2][8] int64 fixed-point bins, added to with integer atomics, so the
 * result is exact and independent of the order in which workgroups arrive -- bit-reproducible
 * training (csrc/common.h).  `float*` in the signatures below is that opaque buffer.
 * res_grad/res_act (nullable, PT like out): out += res_grad where res_act > 0 (fused
 * ReLU-masked residual gradient).  accumulate is a flag word: IIC_ACC_ADD: out += previous
 * contents; IIC_ACC_PREMASK changes the meaning of res_grad / res_act (each nullable on its
 * own): out = (value [+ previous] [+ res_grad]) where res_act > 0, else 0 -- a backward-data
 * launch hands its gradient over already multiplied by the ReLU mask of the activation it
 * belongs to (res_act = the conv's input activation, archs/cluster.py PREMASK).           */
#define IIC_STAT_STRIPES 32
long iic_stat_bytes(int C);
#define IIC_ACC_ADD 1
#define IIC_ACC_PREMASK 2
int iic_conv_igemm(const iic_conv_geom* g, const void* in, const void* w, void* out,
                   float* stats, const void* res_grad, const void* res_act, int accumulate,
                   void* stream);
/* bwd-weight: partial[s][t][co][ci] (fp32) = sum over the rows of split s of
 * dy[pout(m)][co] * x[pin(m)+tap_off[t]][ci]; the geometry is the FORWARD geometry.
 * use_tr != 0 selects ds_read_b64_tr_b16 operand reads (fast); 0 = scalar LDS gathers. */
int iic_conv_wgrad_nsplit(const iic_conv_geom* g);
int iic_conv_wgrad(const iic_conv_geom* g, const void* x, const void* dy, float* partials,
                   int nsplit, int use_tr, void* stream);
/* dW[co][ci][kh][kw] (fp32 OIHW, the nn.Conv2d parameter layout) (+)= sum_s partial[s][t][co][ci] */
int iic_conv_wgrad_reduce(const float* partials, int nsplit, int T, int Cout, int Cin, float* dW,
                          int accumulate, void* stream);
/* fp32 OIHW parameter -> bf16 [T][Co][Ci] (forward operand) and [T][Ci][Co] (bwd-data operand) */
int iic_weight_prep(const float* w_oihw, void* w_fwd, void* w_bwd, int Cout, int Cin, int T,
                    void* stream);

/* Second-generation kernel for Cout % 128 == 0 (same contract as iic_conv_igemm, same
 * reference call sites): the weight operand is read straight from L2 in MFMA B-fragment order
 * and never staged in LDS; 256 x 128 workgroup tiles, barriers only per 64-channel chunk.
 *   w_frag[tap][k/64][n/32][ks][lane][e] = W[n = (n/32)*32 + (lane & 31)]
 *                                           [k = (k/64)*64 + ks*16 + (lane >> 5)*8 + e][tap]
 * with (n, k) = (cout, cin) for the forward operand (bwd = 0) and (cin, cout) for the
 * backward-data operand (bwd = 1).  iic_conv_igemm_frag_supported: 1 if the geometry can run
 * here (needs NP256), else callers use iic_conv_igemm with the row-major operand.            */
int iic_conv_igemm_frag_supported(const iic_conv_geom* g);
int iic_conv_igemm_frag(const iic_conv_geom* g, const void* in, const void* w_frag, void* out,
                        float* stats, const void* res_grad, const void* res_act, int accumulate,
                        void* stream);
/* Same launch with a fused BatchNorm-backward reduction over the tile it stores (backward-data
 * launches): the gradient g this conv produces is the upstream gradient of a BatchNorm whose
 * input y (PT, shaped like `out`) is known, so the two sums that BatchNorm's backward needs,
 *   red_stats += (sum g, sum g*y)     [red_y2 / red_stats2: (sum g, sum g*y2), downsample branch]
 * (the quantities of iic_bn_bwd_reduce, residual.py:20-41 backward) are taken in the epilogue from
 * the stored (bf16-rounded) values -- masked with (scale*y + shift > 0) when red_coef (that
 * BatchNorm's forward coefficients) is given, i.e. a ReLU sits between it and this conv -- instead
 * of in a separate HBM-bound pass over g and y.  Single-launch geometries only (stride 1).     */
int iic_conv_igemm_red_supported(const iic_conv_geom* g);
int iic_conv_igemm_frag_red(const iic_conv_geom* g, const void* in, const void* w_frag, void* out,
                            float* stats, const void* res_grad, const void* res_act, int accumulate,
                            const void* red_y, const float* red_coef, const void* red_y2,
                            float* red_stats, float* red_stats2, void* stream);
int iic_weight_prep_frag(const float* w_oihw, void* w_frag, int Cout, int Cin, int T, int bwd,
                         void* stream);
/* Every weight operand of a network in ONE launch (what the per-parameter calls above do once per conv and
 * layout after each optimiser step): `jobs_dev` = njobs records in DEVICE memory, sorted by first_block
 * (job i owns blocks [first_block, first_block + iic_weight_prep_multi_blocks(Cout, Cin, T))),
 * total_blocks = the sum.  mode: 0 / 1 = B-fragment order forward / backward-data operand
 * (iic_weight_prep_frag), 2 = [T][Co][Ci], 3 = [T][Ci][Co] (iic_weight_prep's two outputs).               */
typedef struct iic_weight_prep_job {
  const float* w;        /* fp32 OIHW parameter */
  void* out;             /* bf16 operand */
  long long first_block;
  int32_t Cout, Cin, T, mode;
} iic_weight_prep_job;
long iic_weight_prep_multi_blocks(int Cout, int Cin, int T);
int iic_weight_prep_multi(const iic_weight_prep_job* jobs_dev, int njobs, long total_blocks, void* stream);

/* ---------------------------------------------------------------------------------
 * BatchNorm2d (train / eval), ReLU, residual add -- replaces nn.BatchNorm2d / nn.ReLU /
 * `out += residual` in residual.py:20-41,56-57, vgg.py:28-30.
 * Statistics: stats stripes produced by iic_conv_igemm; finalised per channel here.
 * ------------------------------------------------------------------------------- */
/* coef[0..4][C]: scale, shift, mean, invstd, unbiased batch variance.  use_running: eval() with
 * track_running_stats.  running_* nullable (track_running_stats=False).  Re-zeroes stats.
 * unbiased_count (0 = count): sample count of the unbiased running_var factor n/(n-1); differs
 * from `count` only under replica de-duplication (cluster_sobel.py:215-226 replicates imgs_curr
 * num_dataloaders times; forwarding the unique images once leaves mean / biased variance
 * unchanged, only this factor sees the true batch size).                                    */
int iic_bn_finalize(float* stats, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, long long* num_batches_tracked, float* coef, int C,
                    long count, long unbiased_count, float eps, float momentum, int training,
                    void* stream);
/* Deferred running-statistic update: what iic_bn_finalize does to running_mean / running_var /
 * num_batches_tracked (torch.nn.BatchNorm2d, momentum 0.1, unbiased variance; residual.py:20,23)
 * when it is given them, applied later from the coefficients it saved (coef row 2 = batch mean,
 * row 4 = unbiased batch variance).  Used when two forwards of one step run concurrently on two
 * streams and must not read-modify-write the same running statistics.  n BatchNorms per call;
 * the pointer arrays are HOST arrays of device pointers.                                     */
int iic_bn_running_update(int n, const float* const* coef, float* const* running_mean,
                          float* const* running_var, long long* const* num_batches_tracked,
                          const int* C, float momentum, void* stream);
/* out = relu( scale*y+shift  [+ res]  [+ scale2*y2+shift2] ) on PT interiors.            */
int iic_bn_apply(const void* y, const float* coef, const void* res, const void* y2,
                 const float* coef2, void* out, int N, int H, int W, int P, int C, int relu,
                 void* stream);
/* sums[stripe][2][C] += sum g, sum g*y  with g = dout * (act > 0) (act nullable => g = dout);
 * second BN (y2/sums2) optional (downsample branch shares g).
 * mask_coef (nullable, exclusive with act): the forward coef of THIS BatchNorm when its
 * activation was act = relu(scale*y + shift) with nothing added: the mask is then recomputed
 * as (scale*y + shift > 0) and the activation tensor is not read at all.                  */
int iic_bn_bwd_reduce(const void* dout, const void* act, const void* y, const void* y2,
                      float* sums, float* sums2, const float* mask_coef, int N, int H, int W, int P,
                      int C, void* stream);
/* from sums -> bcoef[0..2][C] = c1,c2,c3 (dy = c1*g + c2*y + c3), dgamma, dbeta. Re-zeroes sums. */
int iic_bn_bwd_finalize(float* sums, const float* gamma, const float* coef, float* bcoef,
                        float* dgamma, float* dbeta, int C, long count, void* stream);
int iic_bn_bwd_apply(const void* dout, const void* act, const void* y, const float* bcoef,
                     void* dy, const void* y2, const float* bcoef2, void* dy2,
                     const float* mask_coef, int N, int H, int W, int P, int C, void* stream);

/* ---------------------------------------------------------------------------------
 * Stem: conv3x3(Cin<=5 -> 64, pad 1, no bias) + BN + ReLU + MaxPool(k2,s2,p1), computed
 * from the fp32 NCHW input with exact-fp32 MFMA and RECOMPUTED in every pass instead of
 * materialising the 96x96x64 tensor.  Replaces net5g.py:21-26,42-45.
 * ------------------------------------------------------------------------------- */
int iic_stem_stats(const float* x, const float* w, float* stats, int N, int Cin, int H, int W,
                   void* stream);
int iic_stem_apply_pool(const float* x, const float* w, const float* coef, void* out_pt, int N,
                        int Cin, int H, int W, void* stream);
int iic_stem_bwd_reduce(const float* x, const float* w, const float* coef, const void* dpool_pt,
                        float* sums, int N, int Cin, int H, int W, void* stream);
long iic_stem_wgrad_partial_floats(void);
int iic_stem_bwd_wgrad(const float* x, const float* w, const float* coef, const float* bcoef,
                       const void* dpool_pt, float* partials, float* dW, int N, int Cin, int H,
                       int W, void* stream);
/* Fused stem backward (one recompute pass instead of two): `sums` as iic_stem_bwd_reduce AND the
 * coefficient-free weight-gradient GEMMs G1 = sum g*patch, G2 = sum y*patch (+ G3 = sum patch)
 * into `partials` (iic_stem_wgrad_partial_floats() floats).  dy = c1*g + c2*y + c3 is affine
 * with per-channel coefficients, so after iic_bn_bwd_finalize
 *   dW[co][k] = c1[co]*G1[co][k] + c2[co]*G2[co][k] + c3[co]*G3[k]   (iic_stem_wgrad_combine). */
int iic_stem_bwd_fused(const float* x, const float* w, const float* coef, const void* dpool_pt,
                       float* sums, float* partials, int* nblocks_out, int N, int Cin, int H, int W,
                       void* stream);
int iic_stem_wgrad_combine(const float* partials, int nblocks, const float* bcoef, float* dW, int Cin,
                           void* stream);
/* First-layer convolution of the VGG-style trunks from the fp32 NCHW image (Cin*K*K <= 128,
 * K = 3 (pad 1) or 5 (pad 2), 64 output channels) -- replaces the first nn.Conv2d of
 * code/archs/cluster/vgg.py:24-26 (net6c.py:16-20, net10a.py:21-25).  Exact fp32 MFMA.
 * out: PT bf16 [N][H+2P][W+2P][64]; stats as for iic_conv_igemm (nullable).               */
int iic_firstconv_fwd(const float* x, const float* w, void* out_pt, float* stats, int N, int Cin,
                      int H, int W, int K, int pad, int P, void* stream);
long iic_firstconv_wgrad_partial_floats(void);
int iic_firstconv_wgrad(const float* x, const void* dy_pt, float* partials, float* dW, int N,
                        int Cin, int H, int W, int K, int pad, int P, void* stream);
/* nn.MaxPool2d(kernel_size=2, stride=2) (vgg.py:19-20) on PT tensors; backward routes to the
 * first arg-max in scan order like torch.                                                 */
int iic_maxpool2_fwd(const void* in_pt, void* out_pt, int N, int H, int W, int Pi, int Po, int C,
                     void* stream);
int iic_maxpool2_bwd(const void* in_pt, const void* dout_pt, void* din_pt, int N, int H, int W,
                     int Pi, int Po, int C, void* stream);
/* The same pool behind a BatchNorm + ReLU whose output is never stored (the pooled stages of vgg.py:19-30): y_pt is the
 * convolution output (PT bf16), coef the BatchNorm's forward coefficients (iic_bn_finalize); every element enters the
 * window as a = bf16(relu(scale*y + shift)) -- bit for bit what iic_bn_apply would have written.  Forward: out = maxpool(a)
 * (one full-tensor write and one read less per pooled stage); backward: din (gradient w.r.t. a) = dout at the first
 * arg-max of a in scan order, 0 elsewhere.                                                                            */
int iic_bn_relu_maxpool2_fwd(const void* y_pt, const float* coef, void* out_pt, int N, int H, int W, int Pi,
                             int Po, int C, void* stream);
int iic_bn_relu_maxpool2_bwd(const void* y_pt, const float* coef, const void* dout_pt, void* din_pt, int N,
                             int H, int W, int Pi, int Po, int C, void* stream);
/* Sobel pre-op -- replaces code/utils/cluster/transforms.py:47-96 (grey channel -> dx,dy;
 * other channels copied through in the reference's order).                              */
int iic_sobel(const float* imgs, float* out, int N, int C, int H, int W, int include_rgb,
              int using_IR, void* stream);

/* ---------------------------------------------------------------------------------
 * Heads: AvgPool(global) + Linear + Softmax(dim=1) per sub-head -- replaces
 * net5g.py:31-39,53,69-80.  fp32 throughout (feeds the loss).
 * ------------------------------------------------------------------------------- */
int iic_avgpool_fwd(const void* in_pt, float* feats, int N, int H, int W, int P, int C,
                    void* stream);
int iic_avgpool_bwd(const float* dfeats, void* din_pt, int N, int H, int W, int P, int C,
                    const void* mask_act_pt /* nullable: din = 0 where this activation <= 0 */,
                    void* stream);
/* C[m][n] (+)= sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn] (+ bias[n]); exact fp32 MFMA      */
int iic_gemm_f32(const float* A, long sam, long sak, const float* B, long sbk, long sbn,
                 const float* bias, float* C, long scm, int M, int Nn, int K, int accumulate,
                 void* stream);
/* Same product with a caller-provided fp32 workspace of iic_gemm_f32_ws_floats(...) elements (0 = none needed):
 * launches with too few output tiles for the chip split K between blocks, partial tiles go to the workspace
 * and are folded in a fixed order (deterministic).  iic_gemm_f32 == this with ws = NULL (no split).          */
long iic_gemm_f32_ws_floats(long sam, long sak, long sbk, long sbn, int M, int Nn, int K);
int iic_gemm_f32_ws(const float* A, long sam, long sak, const float* B, long sbk, long sbn,
                    const float* bias, float* C, long scm, int M, int Nn, int K, int accumulate,
                    float* ws, long ws_floats, void* stream);
int iic_gemm_f32_splitk(const float* A, long sam, long sak, const float* B, long sbk, long sbn,
                        float* C, long scm, int M, int Nn, int K, int splitk, void* stream);
/* SegmentationNet10a head (net10a.py:44-59): PT feature window <-> fp32 matrix for the 1x1
 * conv (padding 1) GEMM, and bilinear up-sampling (align_corners=False) fwd / bwd between
 * pixel-major [N][Hl][Wl][k] and NCHW [N][k][S][S].                                        */
int iic_seg_window_gather(const void* pt, float* out, int N, int Hw, int Ww, int Hp, int Wp, int off,
                          int C, void* stream);
int iic_seg_window_scatter(const float* in, void* pt, int N, int Hw, int Ww, int Hp, int Wp, int off,
                           int C, void* stream);
/* Fused 10a head on the bf16 PT feature window (net10a.py:44-59: the 1x1 conv with padding 1, its
 * input gradient and its weight gradient) -- the window is read / written in place, W stays fp32,
 * products and sums are exact fp32 (v_mfma_f32_16x16x4_f32).  Supported: C = 256 or 512, k <= 32
 * (iic_seg_head_supported); otherwise the gather + iic_gemm_f32 + scatter chain above.
 * logits, dlog: fp32 [M][k] row-major, M = N*Hw*Ww; w: fp32 [k][C].
 * iic_seg_head_bwd_dx writes the window's interior rows of pt_dx (bf16) only -- the ring is the
 * conv's zero padding.  iic_seg_head_wgrad writes iic_seg_head_wgrad_chunks(M) partial matrices
 * [chunk][k][C]; fold them in order with iic_colsum_f32 (deterministic, no atomics).            */
int iic_seg_head_supported(int C, int k);
int iic_seg_head_wgrad_chunks(long M);
int iic_seg_head_fwd(const void* pt, const float* w, float* logits, int N, int Hw, int Ww, int Hp,
                     int Wp, int off, int C, int k, void* stream);
int iic_seg_head_bwd_dx(const float* dlog, const float* w, void* pt_dx, int N, int Hw, int Ww, int Hp,
                        int Wp, int off, int C, int k, void* stream);
int iic_seg_head_wgrad(const float* dlog, const void* pt, float* partials, int N, int Hw, int Ww,
                       int Hp, int Wp, int off, int C, int k, void* stream);
int iic_bilinear_fwd(const float* in_nhwc, float* out_nchw, int N, int Hl, int Wl, int k, int S,
                     void* stream);
int iic_bilinear_bwd(const float* dout_nchw, float* din_nhwc, int N, int Hl, int Wl, int k, int S,
                     void* stream);
int iic_softmax_fwd(const float* logits, float* probs, int rows, int k, void* stream);
int iic_softmax_bwd(const float* probs, const float* dprobs, float* dlogits, int rows, int k,
                    void* stream);
int iic_colsum_f32(const float* A, float* out, int rows, int cols, int accumulate, void* stream);

/* ---------------------------------------------------------------------------------
 * Adam -- replaces torch.optim.Adam as used by code/utils/cluster/general.py:5-9
 * (betas (0.9,0.999), eps 1e-8, no weight decay, no amsgrad).  Multi-tensor: `n` tensors.
 * ptrs are HOST arrays of device pointers.
 * ------------------------------------------------------------------------------- */
/* grads2: optional second gradient source (NULL, or a host array whose entries may be NULL): the
 * update uses grads[i] + grads2[i] -- the two views of a step may accumulate their parameter
 * gradients separately (iic_amd.ops.branch) and are summed here instead of by extra kernels;
 * grads[i] may then be NULL as well (a tensor that only the second view touched).           */
int iic_adam_step(int n, float* const* params, const float* const* grads, const float* const* grads2,
                  float* const* exp_avg, float* const* exp_avg_sq, const long* numel, float lr,
                  float beta1, float beta2, float eps, int step, void* stream);
/* Same update with the step count kept on the DEVICE: `steps_done` (int32) = updates already
 * applied to these tensors; the kernel derives the bias corrections from it and a trailing
 * 1-thread kernel increments it.  No launch argument changes between steps, so the optimiser
 * step can be part of a captured HIP graph (iic_amd.graph).                                  */
int iic_adam_step_dev(int n, float* const* params, const float* const* grads,
                      const float* const* grads2, float* const* exp_avg, float* const* exp_avg_sq,
                      const long* numel, float lr, float beta1, float beta2, float eps,
                      int* steps_done, void* stream);
/* ... and with the learning rate read from DEVICE memory too (lr_dev != NULL; `lr` is then ignored): the
 * reference scales param_groups[i]["lr"] in place between epochs (update_lr, general.py:12-23), and a
 * captured step must see the new rate without being re-captured.                                        */
int iic_adam_step_devlr(int n, float* const* params, const float* const* grads,
                        const float* const* grads2, float* const* exp_avg, float* const* exp_avg_sq,
                        const long* numel, const float* lr_dev, float lr, float beta1, float beta2,
                        float eps, int* steps_done, void* stream);

/* ---------------------------------------------------------------------------------
 * Exact-fp32 path (SURVEY.md 8c parity tier T2; csrc/f32_path.hip).  The same operators as above
 * on fp32 PT tensors -- same geometry descriptors, epilogue flags and statistic accumulators --
 * as plain one-thread-per-output kernels.  Selected only by iic_amd.ops.fp32_mode(): it lets a whole
 * network be compared with the reference's fp32 results to ~1e-4; never used by the product path.
 * Weights are the fp32 OIHW parameters themselves (wtaps = kh*kw); transposed = 1 for the
 * backward-data geometries (the geometry's output channels are the parameter's input channels).
 * ------------------------------------------------------------------------------- */
int iic_f32_conv(const iic_conv_geom* g, const float* in, const float* w_oihw, int wtaps, int transposed,
                 float* out, float* stats, const float* res_grad, const float* res_act, int accumulate,
                 void* stream);
int iic_f32_wgrad(const iic_conv_geom* g, const float* x, const float* dy, float* dW_oihw, int wtaps,
                  int accumulate, void* stream);
int iic_f32_bn_apply(const float* y, const float* coef, const float* res, const float* y2, const float* coef2,
                     float* out, int N, int H, int W, int P, int C, int relu, void* stream);
int iic_f32_bn_bwd_reduce(const float* dout, const float* act, const float* y, const float* y2, float* sums,
                          float* sums2, const float* mask_coef, int N, int H, int W, int P, int C,
                          void* stream);
int iic_f32_bn_bwd_apply(const float* dout, const float* act, const float* y, const float* bcoef, float* dy,
                         const float* y2, const float* bcoef2, float* dy2, const float* mask_coef, int N, int H,
                         int W, int P, int C, void* stream);
int iic_f32_avgpool_fwd(const float* in_pt, float* feats, int N, int H, int W, int P, int C, void* stream);
int iic_f32_avgpool_bwd(const float* dfeats, float* din_pt, int N, int H, int W, int P, int C,
                        const float* mask_act_pt, void* stream);
/* nn.MaxPool2d(2, 2, padding=1) of the ClusterNet5g stem (net5g.py:26), PT (P = 1) in and out */
int iic_f32_maxpool_s2p1_fwd(const float* in_pt, float* out_pt, int N, int H, int W, int C, void* stream);
int iic_f32_maxpool_s2p1_bwd(const float* in_pt, const float* dout_pt, float* din_pt, int N, int H, int W, int C,
                             void* stream);
int iic_f32_nchw_to_pt(const float* x_nchw, float* out_pt, int N, int C, int H, int W, int P, void* stream);
/* nn.MaxPool2d(2, 2) of the VGG-style trunks and the SegmentationNet10a head's window copies */
int iic_f32_maxpool2_fwd(const float* in_pt, float* out_pt, int N, int H, int W, int Pi, int Po, int C,
                         void* stream);
int iic_f32_maxpool2_bwd(const float* in_pt, const float* dout_pt, float* din_pt, int N, int H, int W, int Pi,
                         int Po, int C, void* stream);
int iic_f32_window_gather(const float* pt, float* out, int N, int Hw, int Ww, int Hp, int Wp, int off, int C,
                          void* stream);
int iic_f32_window_scatter(const float* in, float* pt, int N, int Hw, int Ww, int Hp, int Wp, int off, int C,
                           void* stream);

/* one-time device probes used by the test-suite (documented in DESIGN.md) */
int iic_probe_tr16(void* out_u16_64x4, void* stream);

/* ---------------------------------------------------------------------------------
 * Evaluation counts -- replaces the per-(cluster, class) masked sums with
 * a host sync each of code/utils/cluster/eval_metrics.py:18-24 (_original_match), :42-46
 * (_hungarian_match) and the equality count of _acc (:69).
 * preds / targets: int64 [n] (torch.long, as the reference's flat_preds / flat_targets);
 * counts int64 [k_pred][k_gt] (zeroed here): counts[c1][c2] = #{i : preds[i]==c1, targets[i]==c2}.
 * Labels outside [0, k) match nothing.  k_pred * k_gt <= 16384.
 * ------------------------------------------------------------------------------- */
int iic_contingency(const long long* preds, const long long* targets, long n, int k_pred, int k_gt,
                    long long* counts, void* stream);
int iic_count_equal(const long long* a, const long long* b, long n, long long* count, void* stream);

/* ---------------------------------------------------------------------------------
 * Paired augmentation on the GPU -- replaces the per-sample PIL pipelines
 * of code/utils/cluster/transforms.py that the DataLoaders of code/utils/cluster/data.py:223-335
 * run on the host:
 *   :107-217 sobel_make_transforms, default branch (RGB sources, channels = 3): RandomCrop ->
 *            Resize(BILINEAR) -> [RandomHorizontalFlip -> ColorJitter] -> custom_greyscale_to_tensor (:12-25)
 *   :220-330 greyscale_make_transforms (mode "L" sources, channels = 1): [RandomRotation] ->
 *            crop (one of several sizes) -> Resize -> [flip] -> [ColorJitter] -> ToTensor
 * Results are bit-identical to PIL's (oracle/augment_oracle.py).
 * imgs_u8  uint8 [B][H][W][channels] (HWC, as the datasets hold them), resident in HBM.
 * iparams  int32 [N][20]: source image, crop x0, crop y0, flip, n_ops, op[4] (0 brightness,
 *          1 contrast, 2 saturation, 3 hue -- in application order), hue shift (uint8 wrap),
 *          table index, rotate flag, a0..a5 = PIL's inverse rotation matrix in 16.16 fixed point
 *          (source x = (a2 + a1*y + a0*x) >> 16, source y = (a5 + a4*y + a3*x) >> 16), then the
 *          cutout box of custom_cutout (transforms.py:28-44) in crop coordinates as
 *          (left | upper << 16), (right | lower << 16); right == left = no cutout.
 * norm     float [2][C] (means, then stds) of torchvision Normalize (--demean), or NULL.
 * fparams  float [N][4]: factor of brightness, contrast, saturation; [3] = hue factor (not read:
 *          the kernel uses the uint8 increment in iparams[9]).
 * tables_host  HOST int32 [n_tables][4] = (crop size, taps per output = row pitch of its kk table,
 *          first row of its bounds table, first int of its kk table), n_tables <= 8;
 * bounds   int32 [rows][2] (first tap, tap count), kk int32: Pillow's 22-bit resampling
 *          coefficients of crop -> S per table (square crops: both passes use the same table).
 * lut      float [256] = v / 255 as torch computes it.
 * out      float [N][C][S][S]; channels 3: C = 4 (R,G,B,grey) with include_rgb else 1 (grey);
 *          channels 1: C = 1.
 * ------------------------------------------------------------------------------- */
#define IIC_AUG_MAX_TABLES 8
int iic_augment(const void* imgs_u8, int B, int H, int W, int channels, const int* iparams,
                const float* fparams, int N, const int* tables_host, int n_tables,
                const int* bounds, const int* kk, int S, const float* lut, float* out,
                int include_rgb, const float* norm, void* stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* IIC_HIP_H */

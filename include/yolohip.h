/*
 * yolohip.h — C ABI of libyolohip.so, the MI355X (gfx950) kernels behind the
 * yl-jiang/YOLOSeries detection hot path.
 *
 * The reference has no FFI layer of its own (it is pure PyTorch);
 * every entry point below replaces a Python op sequence of the reference, cited
 * as  file:line  relative to the reference tree.  The Python mirror in
 * yoloseries_amd/ binds these with ctypes.
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer owned by the caller unless the name
 *    says host.  Kernels are only enqueued on `stream`; nothing here
 *    allocates, frees, synchronises or copies to the host, so every call is
 *    legal inside hipStreamBeginCapture.
 *  - Activations are NHWC bf16 (raw uint16 storage), `ld` = elements between
 *    consecutive pixels, so a channel slice of a wider buffer is (ptr+coff, ld).
 *    Channel counts and channel offsets of activation operands are multiples of 8
 *    (16-byte vector access).
 *  - Return value: 0 on success, negative yh_status otherwise;
 *    yh_last_error() gives a thread-local message.
 */
#ifndef YOLOHIP_H
#define YOLOHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* yh_stream;          /* hipStream_t */
typedef uint16_t yh_bf16;         /* raw bfloat16 bits */

enum yh_status {
    YH_OK = 0,
    YH_EINVAL = -1,               /* bad shape / alignment / null pointer */
    YH_ELAUNCH = -2,              /* hipLaunch reported an error */
    YH_EUNSUPPORTED = -3
};

const char* yh_last_error(void);
int yh_version(void);
/* number of compute units of the current device (for workspace sizing) */
int yh_device_cus(void);

/* ------------------------------------------------------------------------ *
 * Implicit-GEMM convolution (MFMA 32x32x16 bf16, fp32 accumulate)
 * replaces nn.Conv2d inside ConvBnAct / Detect:
 *   utils/layer_tools.py:82-94 (ConvBnAct), :454-470 (Detect),
 *   torch.cat / nn.Upsample in the neck: models/normal/yolov5s.py:101-114
 * ------------------------------------------------------------------------ */

/* One channel segment of the (virtually concatenated) conv input. */
typedef struct yh_seg {
    const yh_bf16* ptr;   /* first channel of this segment at pixel 0          */
    int32_t ld;           /* elements per pixel of the underlying buffer        */
    int32_t C;            /* channels in this segment (multiple of 8)           */
    int32_t ups;          /* 1: buffer is (H/2,W/2), read with nearest-2x upsample */
    int32_t _pad;
} yh_seg;

#define YH_CONV_FWD   0   /* hi = ho*stride - pad + kh                          */
#define YH_CONV_DGRAD 1   /* hi = (ho + pad - kh)/stride, only when divisible   */

#define YH_ACT_NONE 0
#define YH_ACT_SILU 1

typedef struct yh_conv_desc {
    yh_seg   seg[2];      /* input = concat(seg[0], seg[1]) along channels      */
    int32_t  nseg;
    int32_t  mode;        /* YH_CONV_FWD / YH_CONV_DGRAD                        */
    int32_t  B, Ho, Wo;   /* output pixel grid; rows M = B*Ho*Wo                */
    int32_t  Hi, Wi;      /* logical input grid (of the concatenated input)     */
    int32_t  KH, KW, stride, pad;
    const yh_bf16* w;     /* packed weights [Npad][KH*KW*Ctot], k = tap*Ctot+c  */
    int32_t  N;           /* real output channels                               */
    int32_t  Npad;        /* rows allocated in w (multiple of 128, zero filled) */
    /* epilogue: v = acc (+bias[n]); if scale: v = v*scale[n]+shift[n];
     *           act; out = v (+res) (+out if accumulate)                        */
    const float* bias;    /* [N] or NULL                                         */
    const float* scale;   /* [N] or NULL  (folded BN, inference)                 */
    const float* shift;   /* [N] or NULL                                         */
    int32_t  act;
    int32_t  accumulate;  /* 1: out += result (gradient accumulation)            */
    yh_bf16* out0; int32_t ld0;   /* columns [0,nsplit)  -> out0[m*ld0 + n]          */
    int32_t  nsplit;              /* multiple of 8; == N when a single destination   */
    yh_bf16* out1; int32_t ld1;   /* columns [nsplit,N)  -> out1[m*ld1 + n-nsplit]   */
    const yh_bf16* res; int32_t ldr;  /* residual added after act to out0 columns, or NULL */
    /* per-channel batch statistics of the stored (bf16-rounded) pre-activation,
     * as per-block partial sums: stats[(blk*2+0)*Npad + n] = sum, [(blk*2+1)*Npad+n] = sumsq,
     * blk in [0, yh_conv_stat_blocks()).  NULL: not collected.                   */
    float*   stats;
    /* launch tuning (0 = library default): output-channel tile width 32/64/128, channels per k-step and the cap on
     * persistent blocks along the pixel axis.  Results are identical for every setting except the number (and so the
     * summation grouping) of the statistics rows; yh_conv_stat_blocks() honours both.            */
    int32_t  tile_n;
    int32_t  grid_cap;
    int32_t  tile_k;      /* channels per k-step, 32 or 64 (64 needs whole 64-channel blocks in every segment but the last and the
                           * 128-wide tile on the register-staged kernel)                                               */
    int32_t  algo;        /* kernel family: 0 library default, 1 register-staged (conv_v2_kernel), 2..4 LDS-DMA ring
                           * (conv_v3_kernel) with a 256x128 / 128x128 / 128x64 tile, 5 the 3x3 halo kernel (conv_halo_kernel), 6 its
                           * 160-channel-wide variant (conv_halo160_kernel: N % 160 == 0, no statistics), when the shape is eligible;
                           * 7 the stride-2 data-gradient kernel (conv_dg2_kernel: DGRAD of a 3x3 / s2 / p1 layer with even output
                           * dims and gz channels in a multiple of 32: the four parity classes from one LDS patch of gz);
                           * 8 the 3x3 / stride-1 patch kernel for small channel counts (conv_p3_kernel: <= 128 channels in a multiple
                           * of 32 in, <= 128 out, forward with statistics or data gradient with the fused reduction);
                           * 9 the 80-channel halo kernel (conv_h80_kernel: 3x3 / s1 / p1 with exactly 80 input channels and N % 80 == 0,
                           * no statistics: 256-pixel x 80-channel tiles on 16x16x32 MFMAs, reduction over the flattened (tap, channel)
                           * index — no padding of N or K);
                           * 10 the pointwise kernel (conv_pw_kernel: 1x1 / s1 / p0 forward with exactly 80, 160 or 320 input channels and
                           * N % 80 == 0, no statistics: pixel tiles and the weight tile whole in LDS, no k loop over memory);
                           * (11: retired in round 5 — the wave-private forward kernel lost to tile quantisation on every BASELINE shape,
                           * profiles/r04_step_experiments.txt k);
                           * 12 the 80 -> 160 channel tap kernel (conv_c80_kernel: 3x3 / s1 or s2 / p1 forward with exactly 80 input and 160
                           * output channels, inference epilogue into one destination, no residual: 256 pixels x 160 channels per
                           * workgroup, one tap per stage — no padding of N or K; YOLOv5x's stage-1 downsampling layer);
                           * 13 the training pointwise kernel (conv_pt_kernel: 1x1 / s1 / p0, forward or data gradient, 128 or 256 input
                           * channels in one segment or in two equal halves (either may be upsampled), one destination; plain /
                           * statistics / generic / fused-reduction epilogues: 32-pixel tiles whole in LDS through a four-deep ring with
                           * counted waits, the 128 x C weight tile resident, no k loop over memory);
                           * 14 the LDS-DMA ring kernel with a 256 x 256 tile (8 waves of 128 x 64, two stages of 64 channels: half the
                           * staged bytes per MFMA of the 256 x 128 tile): N a multiple of 256, whole 64-channel blocks in every segment */
    /* DGRAD only — fused BatchNorm+SiLU backward reduction of the layer whose output gradient this launch writes
     * (it must be the LAST writer of that gradient: out0 covers exactly the producer's N channels; with `accumulate` the earlier
     * contributions already in out0 are added first and the sums are taken over the rounded total):
     * bnr_z = the producer's raw conv output (same pixel grid / channels as out0), bnr_ws = its scale | shift (stride bnr_C),
     * bnr_part = slab [yh_conv_bnr_rows()][2][N] receiving sum(dz), sum(dz*z) per block, dz = g * silu'(z*scale+shift).
     * Replaces yh_bn_silu_bwd_reduce for that layer; yh_bn_bwd_finalize consumes the slab.  NULL: off.          */
    const yh_bf16* bnr_z; int32_t bnr_ldz; int32_t bnr_C;
    const float* bnr_ws;
    float*   bnr_part;
} yh_conv_desc;
/* number of partial-sum rows the conv kernel writes for this shape */
int yh_conv_stat_blocks(const yh_conv_desc* d);
int yh_conv_igemm(const yh_conv_desc* d, yh_stream stream);
/* name of the kernel instantiation yh_conv_igemm launches for this descriptor ("conv_v2_kernel<128, 2, 2, 2, 1>"),
 * as rocprofv3 prints it: lets bench.py report per-kernel numbers that line up with the profiler's */
int yh_conv_kernel_name(const yh_conv_desc* d, char* buf, int buflen);
/* rows of the bnr_part slab for this data-gradient descriptor; 0 = fused reduction not available for it */
int yh_conv_bnr_rows(const yh_conv_desc* d);

/* Weight gradient: dW[n][tap*Ctot + coff_k + c] += sum_m gy[m][n] * X[src(m,tap)][c]
 * (fp32 atomics into a zeroed packed buffer).  One launch per input segment.
 * replaces the autograd backward of nn.Conv2d (train_yolov5.py:337). */
typedef struct yh_wgrad_desc {
    const yh_bf16* gy; int32_t ldg;   /* [M][ldg] gradient of the conv output      */
    int32_t  N;                       /* output channels                            */
    yh_seg   seg;                     /* ONE input segment                          */
    int32_t  coff_k;                  /* channel offset of this segment inside Ctot */
    int32_t  Ctot;
    int32_t  B, Ho, Wo, Hi, Wi, KH, KW, stride, pad;
    float*   dw;                      /* [N][KH*KW*Ctot] fp32, accumulated          */
    int32_t  splits;                  /* split of the M (pixel) reduction, >=1      */
    int32_t  tile_k;                  /* launch tuning: 64 = 64-pixel k-steps on the wide 64-row tilings (0 / 32: default);
                                         128 = the general 128-column tiling also where KH*KW*C <= 384 (needs >= 128 columns);
                                         on the general tiling: 32 = 32-pixel k-steps (two blocks per CU), 35 = four waves of
                                         64 x 64 on 32-pixel k-steps; 40 = the patch form (conv_wgp_kernel: 3x3 layers with 16 / 32 /
                                         64 input channels and <= 64 outputs: the input patch of a pixel region staged once in LDS,
                                         persistent blocks, `splits` caps their number) where yh_conv_wgrad_patch_ok();
                                         129 = wave-private 128 x 128 tiles + stream-K (conv_wgs_kernel: input channels a multiple
                                         of 32, >= 64 outputs, B*Ho*Wo a multiple of 32; `splits` = workgroups, one per CU) where
                                         yh_conv_wgrad_wave_tiles() > 0 */
    /* optional workspace of >= yh_conv_wgrad_ws_bytes() bytes (16-byte aligned, caller-owned, may be shared by launches on ONE
     * stream): the split-M partial tiles are written there with plain stores and summed into dw by a second kernel in split
     * order — bit-reproducible, and faster than the fp32 atomics of the default form (NULL), which are bound by the atomic rate
     * of L2.  dw is read-modify-written by that kernel: launches that share dw columns must be stream-ordered. */
    float*   partial;
    uint64_t partial_bytes;
    /* Fused BatchNorm+SiLU backward for a layer WITHOUT a data gradient (the stem: its input is the image): with bn_z != NULL
     * `gy` is not gz but ga, the gradient w.r.t. the layer's ACTIVATION, and the kernel forms gz = gamma*invstd*(dz - c1 - xhat*c2),
     * dz = ga*silu'(z*scale+shift), in its operand loader — the arithmetic of yh_bn_silu_bwd_apply, rounded to bf16 exactly as that
     * pass stores it — so gz is never written or re-read (YOLOv5s stem at batch 64: 420 MB each way) and the pass disappears from
     * the end of the backward's critical path.  bn_z = raw conv output [M][bn_ldz], bn_ws = scale|shift|mean|invstd (stride N),
     * bn_gamma [N], bn_coef = mean(dz)|mean(dz*xhat) (yh_bn_bwd_finalize).  Layers with N <= 64 * ... tilings listed in
     * conv_wgrad.hip (wide tilings of up to 256 im2col columns); others return YH_EINVAL.                                      */
    const yh_bf16* bn_z; int32_t bn_ldz; int32_t reserved0;
    const float* bn_ws; const float* bn_gamma; const float* bn_coef;
} yh_wgrad_desc;
int yh_conv_wgrad(const yh_wgrad_desc* d, yh_stream stream);
size_t yh_conv_wgrad_ws_bytes(const yh_wgrad_desc* d);
int yh_conv_wgrad_patch_ok(const yh_wgrad_desc* d);
int yh_conv_wgrad_patch_name(const yh_wgrad_desc* d, char* buf, int buflen);   /* instantiation of the patch form, profiler spelling */
const char* yh_conv_wgrad_kernel_name(int N, int Kseg);   /* instantiation used for a layer (tile_k 0), profiler spelling */
const char* yh_conv_wgrad_kernel_name2(int N, int Kseg, int tile_k);
/* number of (out-channel x im2col-column) tiles the kernel uses for a layer; callers size `splits` so that
 * tiles*splits is about one resident wave of blocks */
int yh_conv_wgrad_tiles(int N, int Kseg);
int yh_conv_wgrad_tiles2(int N, int Kseg, int tile_k);

/* ------------------------------------------------------------------------ *
 * BatchNorm (training statistics) + SiLU, forward and backward
 * replaces nn.BatchNorm2d(eps=1e-3, momentum=0.03) + nn.SiLU
 *   utils/layer_tools.py:87-91
 * ------------------------------------------------------------------------ */
/* reduce conv partial sums -> mean/invstd/scale/shift, update running stats.
 * ws layout (fp32, 4*C): scale | shift | mean | invstd                        */
int yh_bn_finalize(const float* stats, int nblk, int ldstat, int C, int64_t count,
                   const float* gamma, const float* beta,
                   float* running_mean, float* running_var, int64_t* num_batches,
                   float eps, float momentum, float* ws, yh_stream stream);
/* inference fold: scale = gamma/sqrt(rv+eps), shift = beta - rm*scale         */
int yh_bn_fold(const float* gamma, const float* beta, const float* rm, const float* rv,
               float eps, int C, float* scale, float* shift, yh_stream stream);
/* BatchNorm in evaluation mode inside a differentiable forward (nn.BatchNorm2d.eval() with autograd on): the workspace the
 * BN+SiLU passes read — scale | shift | mean | invstd, 4*C floats like yh_bn_finalize's — from the RUNNING statistics; the running
 * statistics are not updated.  The backward of such a layer is yh_bn_bwd_finalize with its `coef` output zeroed (no batch-mean
 * terms: gz = gamma * invstd * dz).  utils/layer_tools.py:90-91 under model.eval() */
int yh_bn_frozen(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, int C, float* ws, yh_stream stream);
/* the same for every BatchNorm of a network in one launch: `items_dev` is a table of nitems entries in DEVICE memory
 * (the pointers inside it are device pointers; built once per program by the caller)                                 */
typedef struct yh_bn_fold_item {
    const float* gamma; const float* beta; const float* rm; const float* rv;
    float* scale; float* shift;
    float eps; int32_t C;
} yh_bn_fold_item;
int yh_bn_fold_batch(const yh_bn_fold_item* items_dev, int nitems, yh_stream stream);
/* out = silu(y*scale+shift) (+res) ; all bf16 NHWC slices                      */
int yh_bn_silu_apply(const yh_bf16* y, int ldy, const float* ws, int C, int64_t M,
                     yh_bf16* out, int ldo, const yh_bf16* res, int ldr, yh_stream stream);
/* pass 1 of the backward: partial sums of gz and gz*y per channel (gz = ga*silu'(bn(y)))
 * part layout: [nblk][2][C], nblk = yh_ew_blocks(M)                            */
int yh_ew_blocks(int64_t M);
int yh_bn_silu_bwd_reduce(const yh_bf16* ga, int ldga, const yh_bf16* y, int ldy,
                          const float* ws, int C, int64_t M, float* part, yh_stream stream);
/* combine partials (fp64): sum(gz*xhat) = invstd*(sum(gz*y) - mean*sum(gz)) with mean/invstd from ws
 * -> dgamma, dbeta and coefficients; coef layout (fp32, 2*C): mean(gz) | mean(gz*xhat) */
int yh_bn_bwd_finalize(const float* part, int nblk, int C, int64_t M, const float* ws,
                       float* dgamma, float* dbeta, float* coef, yh_stream stream);
/* pass 2: gy = gamma*invstd*(gz - c1 - xhat*c2); optional residual pass-through
 * gres (op)= ga                                                                */
int yh_bn_silu_bwd_apply(const yh_bf16* ga, int ldga, const yh_bf16* y, int ldy,
                         const float* ws, const float* gamma, const float* coef,
                         int C, int64_t M, yh_bf16* gy, int ldgy,
                         yh_bf16* gres, int ldgres, int gres_accumulate, yh_stream stream);
/* Stacked ConvBnAct layers — ONE convolution whose output channels belong to several BatchNorm + SiLU modules (C3 runs
 * cba1 and cba2 on the same input: utils/layer_tools.py:106-114, stacked into one GEMM here): the apply passes of all parts
 * as one launch over whole rows of y.  Part i covers the columns [C_0 + .. + C_{i-1}, + C_i) of y (and of gy in the backward).
 * The passes read: forward ws, C, out, ldo; backward ws, C, ga, ldga, gamma, coef (as in yh_bn_silu_bwd_apply).  No residual. */
#define YH_BN_MAX_PARTS 4
typedef struct yh_bn_part {
    float*         ws;       /* [4*C] scale | shift | mean | invstd of this part (written by the finalize, read by the passes) */
    int32_t        C;        /* channels of the part, multiple of 8                                            */
    int32_t        ldo;      /* forward pass: row pitch of out                                                 */
    yh_bf16*       out;      /* forward pass: activation of this part (its own tensor / slice)                 */
    const yh_bf16* ga;       /* backward pass: gradient w.r.t. this part's activation                          */
    int32_t        ldga;
    int32_t        nblk;     /* finalize: rows of `slab`                                                       */
    const float*   gamma;    /* BatchNorm weight of the part (forward finalize, backward pass)                 */
    float*         coef;     /* [2*C]: written by the backward finalize, read by the backward pass             */
    const float*   slab;     /* finalize: partial sums of THIS part — forward [nblk][2][ldslab] at its first channel
                                (yh_conv_desc.stats + channel offset), backward [nblk][2][C]                   */
    int32_t        ldslab;   /* forward finalize: row pitch of slab (Npad of the convolution)                  */
    float          eps;
    const float*   beta;     /* forward finalize */
    float*         running_mean;
    float*         running_var;
    int64_t*       num_batches;
    float          momentum;
    int32_t        _pad;
    float*         dgamma;   /* backward finalize */
    float*         dbeta;
} yh_bn_part;
/* yh_bn_finalize / yh_bn_bwd_finalize of every part in one launch (the kernels are latency bound: a launch per part costs the
 * same ~6 us as one for all) */
int yh_bn_finalize_parts(const yh_bn_part* parts, int nparts, int64_t count, yh_stream stream);
int yh_bn_bwd_finalize_parts(const yh_bn_part* parts, int nparts, int64_t M, yh_stream stream);
int yh_bn_silu_apply_parts(const yh_bf16* y, int ldy, int64_t M, const yh_bn_part* parts, int nparts, yh_stream stream);
int yh_bn_silu_bwd_apply_parts(const yh_bf16* y, int ldy, int64_t M, const yh_bn_part* parts, int nparts,
                               yh_bf16* gy, int ldgy, yh_stream stream);
/* column sums of a bf16 matrix (bias gradient of Detect): out[c] += sum_m g[m][c] */
int yh_colsum(const yh_bf16* g, int ldg, int C, int64_t M, float* part, float* out, yh_stream stream);

/* ------------------------------------------------------------------------ *
 * SPPF max-pool 5x5 s1 p2 (utils/layer_tools.py:270-288), NHWC bf16
 * idx: int8 per element, window position (0..24) of the first maximum
 * ------------------------------------------------------------------------ */
int yh_maxpool5_fwd(const yh_bf16* x, int ldx, int B, int H, int W, int C,
                    yh_bf16* out, int ldo, int8_t* idx, yh_stream stream);
int yh_maxpool5_bwd(const yh_bf16* gout, int ldgo, const int8_t* idx, int B, int H, int W, int C,
                    yh_bf16* gin, int ldgi, int accumulate, yh_stream stream);
/* FastSPP's chain x2 = mp(x), x3 = mp(x2), x4 = mp(x3) (utils/layer_tools.py:282-288) in one launch per direction for maps of
 * <= 1920 pixels (yh_sppf_pool3_ok; 20 x 20 at 640^2 input, 40 x 40 at 1280^2): the map of one image x 32 / 16 / 8 channels (maps of up to
 * 480 / 960 / 1920 pixels) lives in LDS, the pools run
 * separably.  Bit-identical to three yh_maxpool5_fwd / yh_maxpool5_bwd launches (arg-max rule, summation order, rounding points).
 * Backward: g1..g3 = gradients of x2..x4 as they stand before the pools' backward (ld ldg), gx (+)= the chain's gradient w.r.t. x;
 * the intermediate sums g(x3)', g(x2)' are not written back. */
int yh_sppf_pool3_ok(int H, int W, int C);
int yh_sppf_pool3_fwd(const yh_bf16* x, int ldx, int B, int H, int W, int C, yh_bf16* o1, yh_bf16* o2, yh_bf16* o3, int ldo,
                      int8_t* i1, int8_t* i2, int8_t* i3, yh_stream stream);
int yh_sppf_pool3_bwd(const yh_bf16* g1, const yh_bf16* g2, const yh_bf16* g3, int ldg, const int8_t* i1, const int8_t* i2, const int8_t* i3,
                      int B, int H, int W, int C, yh_bf16* gx, int ldx, int accumulate, yh_stream stream);
/* gradient of nearest-2x upsample: glo (op)= sum of the 2x2 block of ghi        */
int yh_upsample2_bwd(const yh_bf16* ghi, int ldh, int B, int Hlo, int Wlo, int C,
                     yh_bf16* glo, int ldl, int accumulate, yh_stream stream);
/* input image NCHW fp32 (B,3,H,W) -> space-to-depth NHWC bf16 (B,H/2,W/2,16):
 * channel = (dy*2+dx)*3 + c, channels 12..15 zero.  The 6x6/s2/p2 stem conv of
 * models/normal/yolov5s.py:16 becomes a 3x3/s1/p1 conv on this tensor.          */
int yh_input_s2d(const float* x, int B, int Cin, int H, int W, yh_bf16* out, yh_stream stream);
int yh_fill_u32(void* p, uint32_t v, int64_t n_words, yh_stream stream);

/* ------------------------------------------------------------------------ *
 * Parameter arena: gather/scatter between the fp32 master parameters and the
 * packed bf16 weight images the conv kernels read.
 * ------------------------------------------------------------------------ */
/* dst_bf16[i] = idx[i] >= 0 ? bf16(src[idx[i]]) : 0                             */
int yh_pack_bf16(const float* src, const int32_t* idx, int64_t n, yh_bf16* dst, yh_stream stream);
/* dst[i] = idx[i] >= 0 ? src[idx[i]] : 0   (packed fp32 grads -> parameter layout) */
int yh_gather_f32(const float* src, const int32_t* idx, int64_t n, float* dst, yh_stream stream);
/* SGD with momentum/nesterov on a flat arena; group[i] selects lr/wd.
 * torch.optim.SGD semantics (train_yolov5.py:258-280):
 *   g += wd*p ; buf = first ? g : mom*buf + g ; g = nesterov ? g + mom*buf : buf ; p -= lr*g */
int yh_sgd_step(float* p, const float* g, float* buf, const uint8_t* group, int64_t n,
                const float* lr, const float* wd, int ngroups, float momentum, int nesterov,
                int first_step, const float* grad_scale, yh_stream stream);
/* the same step with every per-step scalar in DEVICE memory: scal[8] = lr[3] | wd[3] | momentum | first-step flag (!= 0).
 * No value of the step is a launch argument, so a captured hipGraph replays correctly while the host changes lr / momentum
 * between replays (warm-up, train_yolov5.py:437-456); at most 3 parameter groups. */
int yh_sgd_step_dev(float* p, const float* g, float* buf, const uint8_t* group, int64_t n,
                    const float* scal, int nesterov, const float* grad_scale, yh_stream stream);
/* EMA with the decay read from device memory (trainer/ema_model.py:20-28: the decay ramps with the update count) */
int yh_ema_update_dev(float* ema, const float* p, int64_t n, const float* decay_dev, yh_stream stream);
/* advance the device-resident EMA update counter by one and write the decay of this update,
 * decay = ratio * (1 - exp(-counter / tau)) evaluated in double (trainer/ema_model.py:12); then yh_ema_update_dev(.., decay, ..).
 * Keeps the whole train step free of host-computed scalars (hipGraph replay). */
int yh_ema_advance(int64_t* counter, float* decay, double ratio, double tau, yh_stream stream);
/* sum of squares of a flat fp32 buffer (for clip_grad_norm_, train_yolov5.py:344) */
int yh_sumsq(const float* x, int64_t n, float* part, float* out, yh_stream stream);
/* clip coefficient min(1, max_norm/(sqrt(sumsq)+1e-6)) as a device scalar for yh_sgd_step's grad_scale */
int yh_clip_scale(const float* sumsq, float max_norm, float* out, yh_stream stream);
/* EMA: e = d*e + (1-d)*p over a flat arena (trainer/ema_model.py:20-28)         */
int yh_ema_update(float* ema, const float* p, int64_t n, float decay, yh_stream stream);

/* ------------------------------------------------------------------------ *
 * YOLOv5 loss (loss/yolov5_loss.py:30-235)
 * ------------------------------------------------------------------------ */
typedef struct yh_v5loss_desc {
    int32_t B, maxbox, num_class, num_anchor, num_stage;
    int32_t H[4], W[4];           /* stage feature-map sizes                       */
    float   img_size0, img_size1; /* hyp['input_img_size'][0], [1]                */
    float   anchors[4][3][2];     /* pixels, anchors[stage][a] = (w,h)            */
    float   anchor_thr;           /* hyp['anchor_match_thr']                      */
    float   cls_smooth, cls_pos_weight, cof_pos_weight;
    int32_t use_focal; float focal_gamma, focal_alpha;
    float   iou_scale, cof_scale, cls_scale;
    int32_t pred_is_f32;          /* 0: bf16 predictions, 1: fp32                  */
    int32_t ldp[4];               /* elements per cell of each prediction buffer   */
    int32_t targets_xywhn;        /* 1: targets already hold (cx,cy,w,h)/img size — the argument
                                     convention of YOLOV5Loss.match (loss/yolov5_loss.py:142)   */
} yh_v5loss_desc;

/* workspace sizes in bytes */
size_t yh_v5loss_ws_bytes(const yh_v5loss_desc* d);
/* Target assignment for all stages (YOLOV5Loss.match, loss/yolov5_loss.py:142-214).
 * targets: [B][maxbox][6] fp32 (xmin,ymin,xmax,ymax,cls,img_id), padding rows -1.
 * Per stage s, with cap = 5*A*B*maxbox rows:
 *   count[s]                    int32
 *   tbox  [s][cap][4] fp32      (x-cx, y-cy, w, h) grid units
 *   tidx  [s][cap][5] int32     (cls, img, anchor, gy, gx)                       */
int yh_v5_assign(const yh_v5loss_desc* d, const float* targets,
                 int32_t* count, float* tbox, int32_t* tidx, void* ws, yh_stream stream);
/* Forward loss. preds[s]: [B][H][W][ldp] with channel a*(5+nc)+e.
 * balances: [num_stage] fp64 device state (python floats in the reference), updated in place.
 * result (fp32[8]): tot, iou, cof, cls, tar_nums, 0,0,0  (iou/cof/cls already x B)
 * saved: opaque per-call state for the backward, yh_v5loss_saved_bytes()         */
size_t yh_v5loss_saved_bytes(const yh_v5loss_desc* d);
int yh_v5_loss_fwd(const yh_v5loss_desc* d, const void* const* preds, const float* targets,
                   double* balances, float* result, void* saved, void* ws, yh_stream stream);
/* Backward: gpreds[s] same geometry/dtype as preds[s], fully overwritten.
 * gout: device pointer to d(loss)/d(tot) scalar.                                 */
int yh_v5_loss_bwd(const yh_v5loss_desc* d, const void* const* preds, const float* gout,
                   const void* saved, void* const* gpreds, void* ws, yh_stream stream);

/* ------------------------------------------------------------------------ *
 * YOLOX loss with SimOTA assignment (loss/yolox_loss.py:11-458)
 * ------------------------------------------------------------------------ */
typedef struct yh_yolox_desc {
    int32_t B, maxbox, num_class, num_stage;
    int32_t H[4], W[4], ldp[4];   /* per stage: feature map size, elements per cell ([x,y,w,h,obj,cls...] order) */
    int32_t pred_is_f32;
    float   img_size0;            /* hyp['input_img_size'][0]; stride = img_size0 / H                          */
    int32_t use_focal; float focal_gamma, focal_alpha;
    int32_t use_l1;
    float   iou_scale, cls_scale, cof_scale, l1_scale;
    float   cls_smooth, cls_pos_weight, cof_pos_weight;
    int32_t iou_type;             /* 0 iou, 1 giou, 2 ciou (YOLOXLoss.iou_loss :378-415)                        */
    int32_t topk; float center_radius;
    float   cls_cost_const;       /* class part of the SimOTA cost: constant in the reference (:111-147)        */
} yh_yolox_desc;
size_t yh_yolox_saved_bytes(const yh_yolox_desc* d);
size_t yh_yolox_ws_bytes(const yh_yolox_desc* d);
int yh_yolox_layout(const yh_yolox_desc* d, int64_t* out8);
/* targets_xywh: [B][maxbox][6] (cx,cy,w,h,cls,img) pixels, padding rows cls < 0 (the reference converts the
 * caller's xyxy tensor in place, :42).  result fp32[8]: tot, iou, l1, cls, cof, fg_nums, tar_nums, 0           */
int yh_yolox_loss_fwd(const yh_yolox_desc* d, const void* const* preds, const float* targets_xywh,
                      double* balances, float* result, void* saved, void* ws, yh_stream stream);
int yh_yolox_loss_bwd(const yh_yolox_desc* d, const void* const* preds, const float* targets_xywh, const float* gout,
                      const void* saved, void* const* gpreds, yh_stream stream);

/* Box utilities (utils/bbox_tools.py:164-339) fp32.  yh_iou_matrix: (n1, n2) IoU of xyxy boxes; eps_clamp > 0: union clamped from
 * below (gpu_iou, utils/bbox_tools.py:164-190), 0: no clamp on the union (numba_iou :12-35: 0 / 0 = NaN), < 0: no clamp on the
 * intersection sides either (the evaluators' bbox_iou, trainer/eval_yolov5.py:237-258, trainer/eval_yolox.py) */
int yh_iou_matrix(const float* b1, int n1, const float* b2, int n2, float eps_clamp, float* out, yh_stream stream);
/* kind: 0 giou 1 diou 2 ciou ; pairwise (N,) ; grad (optional) d out/d b1 [N][4] */
int yh_iou_pairwise(int kind, const float* b1, const float* b2, int n, float* out, float* grad_b1, yh_stream stream);

/* ------------------------------------------------------------------------ *
 * Inference post-processing (trainer/eval_yolov5.py:182-317, utils/nms.py:10-27)
 * ------------------------------------------------------------------------ */
typedef struct yh_decode_desc {
    int32_t B, num_class, num_anchor, num_stage;
    int32_t H[4], W[4];
    float   stride[4];
    float   anchors[4][3][2];     /* pixels */
    int32_t pred_is_f32;
    int32_t ldp[4];
    int32_t yolox;                /* 0: v5 decode, 1: YOLOX decode (eval_yolox.py:123-150) */
} yh_decode_desc;
/* Full decode to (B, sum(A*H*W), 5+nc) fp32, the tensor do_inference returns.    */
int yh_decode_full(const yh_decode_desc* d, const void* const* preds, float* out, yh_stream stream);
/* Fused decode + candidate filter, order preserving.
 *  cand: [B][cap][6] fp32 (xmin,ymin,xmax,ymax,conf,cls) ; ncand: [B] int32 (may exceed cap: overflow)
 *  conf_ge: obj >= conf_thr ; cls_gt: cls_conf > cls_thr (v5) / obj*max>=conf & cls>=thr (yolox) */
int yh_decode_filter(const yh_decode_desc* d, const void* const* preds, float conf_thr, float cls_thr,
                     float* cand, int32_t* ncand, int cap, void* ws, yh_stream stream);
/*  ws: yh_decode_filter_ws_bytes(d) bytes, 16-byte aligned: the image is spread over blocks of 64 pixels (rows staged through LDS
 *  with coalesced loads), their candidates are ordered by a second, per-image pass.  ws == NULL: one workgroup per image walks
 *  the predictions (same result; for small heads).                                                                     */
size_t yh_decode_filter_ws_bytes(const yh_decode_desc* d);
/* The same filter applied to an already decoded (B, N, 5+nc) fp32 tensor — the argument of
 * YOLOV5Evaluator.numba_nms (trainer/eval_yolov5.py:261-286); used after TTA merging.  `yolox`: 0 YOLOv5 single label,
 * 1 YOLOX, 2 YOLOv5 multi-label (hyp['mutil_label'], :276-279: one candidate per (prediction, class) with cls*obj >= cls_thr,
 * in (prediction, class) order; ncand may exceed cap — rows past cap are counted, not stored: the caller re-runs with more),
 * 3 YOLOX multi-label (trainer/eval_yolox.py:218-221: the same among the predictions with obj * max(cls) >= conf_thr). */
int yh_filter_decoded(const float* dec, int B, int N, int num_class, float conf_thr, float cls_thr, int yolox,
                      float* cand, int32_t* ncand, int cap, yh_stream stream);
/* Greedy NMS per image on candidate lists, selection order = reference order.
 *  class_aware: add cls*4096 to the box before IoU (hyp['agnostic'] == True in the reference)
 *  thr_inclusive: 1 -> suppress when iou >= thr (numba_nms), 0 -> iou > thr (gpu_nms)
 *  merge_filter: postprocess_bbox filter (eval_yolov5.py:306-315)
 *  out: [B][max_keep][6] ; nkeep [B] ; keep_idx [B][max_keep] index into the candidate list
 *  Candidates are sorted once by (score descending, index ascending) and consumed in chunks of 64 whose mutual suppression is a
 *  64 x 64 bit matrix built with wave shuffles; scores must be non-negative (non-positive scores are never selected).
 *  ws: yh_nms_ws_bytes(B, cap) bytes, 16-byte aligned (sort keys, class-offset boxes in candidate and in sorted order, flags) */
size_t yh_nms_ws_bytes(int B, int cap);
int yh_nms_batched(const float* cand, const int32_t* ncand, int B, int cap,
                   float iou_thr, int class_aware, int thr_inclusive, int max_keep, int merge_filter,
                   float* out, int32_t* nkeep, int32_t* keep_idx, void* ws, yh_stream stream);

/* ------------------------------------------------------------------------
 * Program executor: replay a pre-built list of launches with one call (the engine's forward / backward programs; replaces the
 * host loop over ~700 calls of a train step: train_yolov5.py:327-350 spends that time inside torch's dispatcher instead).
 * A command is an entry point of this library (index from yh_exec_op) with its arguments widened to 8-byte slots — pointers
 * and integers as (u)int64, float / double as the bit pattern of a double — in declaration order, the stream slot last (its
 * value is replaced by streams[cmd.stream]); or YH_CMD_EVENT_RECORD / YH_CMD_STREAM_WAIT with a hipEvent_t in slot 0.
 * Executed in order; on failure returns the callee's status and, in *failed, the index of the command.              */
#define YH_CMD_SLOTS 16
#define YH_CMD_EVENT_RECORD (-1)
#define YH_CMD_STREAM_WAIT  (-2)
typedef struct yh_cmd {
    int32_t  op;            /* yh_exec_op() index, or YH_CMD_EVENT_RECORD / YH_CMD_STREAM_WAIT */
    int32_t  nslots;        /* argument count incl. the stream */
    int32_t  stream;        /* index into yh_exec's streams[] */
    int32_t  reserved;
    uint64_t slots[YH_CMD_SLOTS];
} yh_cmd;
int yh_exec_op(const char* name, int* nargs);
int yh_exec(const yh_cmd* cmds, int n, const yh_stream* streams, int nstreams, int* failed);

#ifdef __cplusplus
}
#endif
#endif /* YOLOHIP_H */

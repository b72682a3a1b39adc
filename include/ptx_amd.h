/*
 * ptx_amd.h -- C ABI of libptx_amd.so, the MI355X (gfx950) forward-pass engine for the
 * pretorched-x video model-zoo hot path.
 *
 * The reference (alexandonian/pretorched-x) is pure Python on top of torch.nn: it has no FFI of
 * its own, so there is no reference-side binding to mirror symbol-for-symbol (SURVEY.md F1,
 * section 8b last row).  Each entry point below replaces the ATen op(s) the reference invokes at
 * the cited call site; INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - plain C: raw device pointers, sizes, POD descriptors, a hipStream_t passed as void*.
 *   - every function returns an int status (PTX_OK == 0); nothing throws across the ABI;
 *     ptx_last_error() returns a thread-local message for the last non-zero status.
 *   - all kernels are asynchronous on the given stream; the library never synchronises, never
 *     allocates device memory, and keeps no per-call state (workspace is passed in), so calls
 *     from different host threads on different streams are safe and everything is capturable
 *     into a hipGraph.
 *   - activations are fp32, channels-last:  NDHWC  [N][T][H][W][ld]  with ld >= C, ld % 4 == 0
 *     and channels [C, ld) kept zero.  2-D tensors are the T == 1 case.
 *   - arithmetic: fp32 operands, fp32 accumulate on v_mfma_f32_32x32x2_f32 /
 *     v_mfma_f32_16x16x4_f32 (exact fp32, k-ordered fma chain).
 */
#ifndef PTX_AMD_H
#define PTX_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTX_OK 0
#define PTX_ERR_INVALID 1      /* bad descriptor / null pointer / misaligned */
#define PTX_ERR_UNSUPPORTED 2  /* shape not supported by the requested tile configuration */
#define PTX_ERR_HIP 3          /* a HIP runtime call failed (message has hipGetErrorString) */
#define PTX_ERR_WORKSPACE 4    /* workspace too small */

typedef void* ptx_stream_t; /* hipStream_t */

/* epilogue / prologue flags of ptx_conv3d_fwd */
#define PTX_EPI_RELU 1u      /* ReLU on the output            (resnet3D.py:131,135,142)          */
#define PTX_EPI_RES_ADD 2u   /* + residual, same shape as y   (resnet3D.py:141 `out += residual`) */
#define PTX_EPI_RES_PADA 4u  /* + shortcut-A residual: strided subsample of `res`, zero channels
                                above res_C              (resnet3D.py:65-74, nonlocalnet.py:322) */
#define PTX_EPI_RES_UP 64u   /* with RES_PADA: the residual is nearest-UPsampled (index = out >> res_s*, res_s* =
                                log2 factor) and channel-truncated (res_C >= Co): the BigGAN-deep GBlock
                                skip `up(x[:, :out_channels])`                                            */
#define PTX_F16_OPERANDS 128u /* ptx_conv3d_fwd: x and w_packed hold IEEE halfs (fp16 MFMA, fp32 accumulate, fp32 bias /
                                residual / output).  The descriptor's Ci, ldx and Kc then count 32-bit WORDS, i.e.
                                channel PAIRS (channels % 8 == 0); pack with ptx_pack_desc.f16.  Runs on the
                                ".../f16" tile configurations (BigGAN generator, BASELINE config 5)            */
#define PTX_ACT_OUT_F16 0x100 /* ptx_affine_act_upsample: OR into `act` -- y is written as halfs (ldy in halfs,
                                multiple of 8): the cBN -> ReLU -> upsample pass feeds an fp16-operand conv      */
#define PTX_F16X3_OPERANDS 0x8000u /* ptx_conv3d_fwd / ptx_conv3d_dual_fwd: fp32-ACCURATE products on the fp16 matrix
                                cores.  x stays fp32 (same layout as the default path); w_packed is packed with
                                ptx_pack_desc.f16 == 2: every 8-channel block of a filter row is stored as 8 "hi" halfs then 8
                                "lo" halfs (hi = half(w), lo = half((w - hi) * 2^12), scaled so it stays a normal half; the same 32 bytes).  The kernel splits the
                                activations the same way in registers and issues a_hi.b_lo + a_lo.b_hi + a_hi.b_hi as three
                                v_mfma_f32_32x32x16_f16 with fp32 accumulate: each half product is exact in fp32, the dropped
                                a_lo.b_lo term is <= 2^-22 of the product, so results match the fp32 path to ~1e-6 relative
                                (measured: same |dlogits| vs the CPU reference as fp32 MFMA) at 3 / 16 of its matrix-core
                                time.  Needs Kc % 8 == 0, groups == 1 and operand magnitudes inside the half range (< 65504).
                                Precision contract per operand value v: relative error 2^-22 over the NORMAL half range
                                6.1e-5 <= |v| < 65504 -- the lo halves are kept scaled by 2^12 and their cross terms
                                accumulate separately (folded back exactly after the k-loop), so lo does not go subnormal
                                before hi does.
                                Runs on the ".../x3" tile configurations.                                          */
#define PTX_SPLITK_FUSED 0x10000u /* split_k > 1 without the second (reduce) launch: the workspace then starts with 64 KiB of
                                tile arrival counters -- which the CALLER zeroes once (ptx_conv3d_workspace_bytes includes
                                them when the descriptor carries this flag) and every launch leaves at zero -- followed by
                                the partial slabs; the last split block to finish a tile sums the partials in split order
                                (bit-identical to the separate reduce kernel), applies the epilogue and writes y.  Grids
                                with more than 16384 tiles and the fused generator-stage epilogue fall back to the reduce
                                kernel (same workspace layout).                                                     */
/* Fused generator stage (BigGAN-deep GBlock: cBN -> ReLU -> [nearest x2] -> conv, BASELINE config 5), on the
 * fp16-operand tiles through ptx_conv3d_fused_fwd.  The class-conditional BN that FOLLOWS a conv is applied in that
 * conv's epilogue as a per-sample affine (tables from ptx_cbn_fold), the upsampling that precedes a conv is done by
 * its loader, so no cBN / ReLU / upsample pass touches HBM and activations stay halfs between the convs. */
#define PTX_EPI_OUT_F16 0x200u  /* y is written as halfs (ldy counts halfs, multiple of 8; Co even)                 */
#define PTX_EPI_AFFINE 0x400u   /* v = v * scale[n][co] + shift[n][co] after bias (+ skip), before ReLU / tanh;
                                   n = sample of the output row; tables in ptx_conv_fused_ext                      */
#define PTX_EPI_DUAL_RAW 0x800u /* additionally store the pre-affine value as halfs to ext->y_raw (the next block's
                                   skip operand)                                                                   */
#define PTX_RES_F16 0x1000u     /* the residual / skip operand `res` holds halfs (ldr counts halfs)               */
#define PTX_PRO_UP2 0x2000u     /* the input is read through a nearest 2x upsample in H and W: desc.Hi / Wi are the
                                   UPSAMPLED extents the filter slides over, x stores [N][1][Hi/2][Wi/2][ldx]      */
#define PTX_EPI_TANH 0x4000u    /* tanh on the output (the generator's image conv)                                 */
#define PTX_PRO_RELU 8u      /* ptx_linear_fwd only: ReLU on the input while loading (trn.py:39-45) */
#define PTX_EPI_ACCUM 16u    /* ptx_linear_fwd only: y += result (trn.py:110 stack(...).sum(0))   */

const char* ptx_version(void);
const char* ptx_last_error(void);

/* --------------------------------------------------------------------------------------------
 * Convolution (implicit GEMM on fp32 MFMA) with fused bias(+folded BN) + residual + ReLU.
 * Replaces: conv3d -> batch_norm(eval) [-> relu] [-> add_ -> relu] as issued by
 *   resnet3D.py:125-143 (Bottleneck), :93-106 (BasicBlock), :153-155 (stem), :176-185 (shortcut B),
 *   r2plus1d.py:85-88 (factored convs), nonlocalnet.py:86-111 (g/theta/phi/W pointwise convs),
 *   and any torch.matmul-shaped contraction expressed as a 1x1x1 conv (trn.py:39-45).
 * ------------------------------------------------------------------------------------------ */
typedef struct ptx_conv3d_desc {
    int32_t N, Ti, Hi, Wi;   /* input positions                                   */
    int32_t Ci, ldx;         /* input channels, input channel stride (floats)     */
    int32_t To, Ho, Wo;      /* output positions                                  */
    int32_t Co, ldy;         /* output channels, output row stride (>= Co rounded up to 4) */
    int32_t kT, kH, kW;      /* filter taps                                       */
    int32_t sT, sH, sW;      /* strides                                           */
    int32_t pT, pH, pW;      /* zero padding                                      */
    int32_t Kc;              /* packed-weight K extent per tap  (>= Ci / groups, multiple of 4)       */
    int32_t Co_pad;          /* packed-weight row count per tap (>= Co, multiple of 128)              */
    uint32_t flags;          /* PTX_EPI_RELU | PTX_EPI_RES_ADD | PTX_EPI_RES_PADA */
    /* residual operand (PTX_EPI_RES_ADD: ldr only; PTX_EPI_RES_PADA: all fields) */
    int32_t ldr, res_C, res_T, res_H, res_W, res_sT, res_sH, res_sW;
    /* second activation source of ptx_conv3d_dual_fwd (ignored by ptx_conv3d_fwd): x2 is NDHWC
     * [N][x2_T][x2_H][x2_W][x2_ld] with x2_C channels, sampled at (to*x2_sT, ho*x2_sH, wo*x2_sW) */
    int32_t x2_C, x2_ld, x2_T, x2_H, x2_W, x2_sT, x2_sH, x2_sW;
    /* grouped convolution (resnext3D.py:85-92, cardinality 32): 0 or 1 = dense.  Output channel co reads
     * the input channels [g*Ci/groups, (g+1)*Ci/groups), g = co / (Co/groups); the packed filter has
     * Kc >= Ci/groups columns per tap (pack it with ptx_pack_desc.Ci = Ci/groups).  Ci/groups must be a
     * multiple of 4.  Runs on the direct (VALU) tile configurations, and on MFMA tiles whose N extent divides
     * Co/groups (the tile then reads only its group's input columns). */
    int32_t groups;
} ptx_conv3d_desc;

/* number of compiled tile configurations, and a printable name "BMxBNxBK/WMxWN/mfmaMT" */
int ptx_conv3d_num_configs(void);
const char* ptx_conv3d_config_name(int config);
/* 1 if `config` can run `desc` (tile K-step divides Kc, etc.), else 0 */
int ptx_conv3d_config_supported(const ptx_conv3d_desc* desc, int config);
/* heuristic default configuration for a problem (never fails for a valid descriptor) */
int ptx_conv3d_pick_config(const ptx_conv3d_desc* desc, int* split_k);
/* bytes of fp32 partial-sum workspace needed when split_k > 1 (0 otherwise) */
size_t ptx_conv3d_workspace_bytes(const ptx_conv3d_desc* desc, int split_k);
/*
 * y[m][co] = epilogue( sum_{tap,c} x[pos(m,tap)][c] * w_packed[tap][co][c] + bias[co] )
 * x: NDHWC input; w_packed/bias: from ptx_pack_conv_weight; res: residual operand or NULL;
 * y: NDHWC output with row stride ldy; columns [0, round_up(Co,4)) of every row are written
 * ([Co, round_up(Co,4)) as zero), columns beyond are left untouched -- so y may point at a channel
 * slice of a wider tensor (torch.cat(dim=1) of branch outputs: slowfast.py:145-151).
 * config < 0 -> ptx_conv3d_pick_config.
 */
int ptx_conv3d_fwd(const ptx_conv3d_desc* desc, const float* x, const float* w_packed,
                   const float* bias, const float* res, float* y, void* workspace,
                   size_t workspace_bytes, int config, int split_k, ptx_stream_t stream);
/*
 * Two-source pointwise conv: the K axis is the concatenation of x's channels and x2's channels,
 *   y[m][co] = epilogue( sum_c x[m][c] w[co][c] + sum_c2 x2[pos2(m)][c2] w[co][Kc + c2] + bias[co] )
 * (1x1x1, unit stride on x; x2 is a strided gather).  This is a bottleneck's last conv with its
 * shortcut-B branch folded in: conv3 -> bn3, downsample conv -> bn, add, relu
 * (resnet3D.py:135-142 with :176-185) as ONE GEMM -- no residual tensor is written or re-read.
 * w_packed: [1][Co_pad][Kc + Kc2] built by two ptx_pack_conv_weight calls (ld_k / k_off / bias_accumulate).
 */
int ptx_conv3d_dual_fwd(const ptx_conv3d_desc* desc, const float* x, const float* x2,
                        const float* w_packed, const float* bias, float* y, void* workspace,
                        size_t workspace_bytes, int config, int split_k, ptx_stream_t stream);

/*
 * Chained convolutions: conv -> BN -> ReLU -> 1x1x1 conv -> BN (-> + residual) -> ReLU in ONE launch,
 *   y[m][co2] = epi2( sum_c1 relu?( conv(x, w_packed)[m][c1] + bias[c1] ) * w2_packed[co2][c1] + bias2[co2] (+ res[m][co2]) )
 * -- the [M][N1] result of the first conv never reaches HBM: a workgroup parks its [BM][N1] tile in LDS and feeds it to
 * the second GEMM as the A operand.  Replaces (a) a bottleneck's tail, conv2 (3x3x3) -> bn2 -> relu -> conv3 (1x1x1) ->
 * bn3 -> `out += residual` -> relu (resnet3D.py:129-142), and (b) the pointwise pairs of the (2+1)D networks: a
 * "1x1x1" SpatioTemporalConv is spatial_conv (1x1x1) -> bn -> relu -> temporal_conv (1x1x1) through
 * M = floor(Cin Cout / (Cin + Cout)) mid channels (r2plus1d.py:68-88), followed by the block's own BN (+ residual) + ReLU.
 *   conv: any dense fp32 conv descriptor (taps, strides, padding as ptx_conv3d_fwd; flags within PTX_EPI_RELU: the ReLU
 *         BETWEEN the two convs; ldy is ignored); N1 = conv->Co must fit ONE N tile of the chained tile configuration
 *         (32 / 64 / 128 -- ptx_conv3d_chain_supported).  A strided pointwise PAIR is expressed by composing the strides
 *         in `conv` (the tail's positions are the final output positions).
 *   tail: a 1x1x1 / unit-stride descriptor over the first conv's OUTPUT positions (N, Ti/Hi/Wi = conv's N, To/Ho/Wo),
 *         Ci = conv->Co, Kc >= round_up(Ci, 4) (its packed filter [Co_pad][Kc], ptx_pack_conv_weight), flags within
 *         PTX_EPI_RELU | PTX_EPI_RES_ADD (res: same shape as y, row stride ldr), ldy = row stride of y.
 * Arithmetic: both GEMMs on v_mfma_f32_32x32x2_f32 / 16x16x4_f32 in the same k order as the unfused launches on tiles of
 * the same MFMA shape -- bit-identical to ptx_conv3d_fwd followed by ptx_conv3d_fwd.  config < 0: ptx_conv3d_chain_pick_config.
 * Split operands: PTX_F16X3_OPERANDS on BOTH descriptors (both filters packed with ptx_pack_desc.f16 == 2, Kc % 8 == 0)
 * runs both GEMMs as three fp16 MFMAs per product block on the ".../chain/x3" configurations -- the parked tile stays
 * fp32 and is split at fragment-read time like an activation; bit-identical to the two x3 launches.  The flag on one
 * descriptor only is PTX_ERR_INVALID; a configuration of the other operand kind is PTX_ERR_UNSUPPORTED.
 */
int ptx_conv3d_chain_num_configs(void);
const char* ptx_conv3d_chain_config_name(int config);
int ptx_conv3d_chain_supported(const ptx_conv3d_desc* conv, const ptx_conv3d_desc* tail, int config);
int ptx_conv3d_chain_pick_config(const ptx_conv3d_desc* conv, const ptx_conv3d_desc* tail);
int ptx_conv3d_chain_fwd(const ptx_conv3d_desc* conv, const ptx_conv3d_desc* tail, const float* x, const float* w_packed,
                         const float* bias, const float* w2_packed, const float* bias2, const float* res, float* y,
                         int config, ptx_stream_t stream);

/* Small-Cin STEM convolution with split operands (PTX_F16X3_OPERANDS), read straight from a channels-last input whose
 * positions are 16 bytes (Ci <= 4, ldx == 4) -- `conv1` of the ResNet3D family (resnet3D.py:153), the 2-D ResNet / I3D
 * stems, the (1,7,7) spatial stem of R2Plus1D (r2plus1d.py:73-88).  A workgroup stages the input patch of one temporal
 * tap ((R-1)*sH + kH rows of (Wo-1)*sW + 8 positions) once and serves all kH x kW taps from it: ~6x less L2 -> LDS
 * traffic than the kW-folded implicit GEMM and no fold pass.  desc: Ci <= 4, ldx = 4, Kc = 32, kW <= 8, stride_w <= 2,
 * symmetric padding, flags within PTX_F16X3_OPERANDS | PTX_EPI_RELU.  x: the output of ptx_ncdhw_to_split4 (16-byte
 * positions holding (hi4 | lo4) halfs of 4 channels).  w_packed: ptx_pack_conv_weight with fold_kw = 1, Ci = 4
 * (channel 3 zero), Kc = 32, f16 = 2.  ptx_conv_stem_x3_supported: 1 if the descriptor can run here.                  */
/* x [N][C][S] fp32, C <= 4  ->  y [N][S] positions of 16 bytes: halfs (hi c0..c3 | lo c0..c3), hi = half(v),
 * lo = half((v - hi) * 2^12) (the scaled lo of PTX_F16X3_OPERANDS); missing channels are zero.  The NCDHW ->
 * channels-last edge of a split-operand stem.                                                                        */
int ptx_ncdhw_to_split4(const float* x, void* y, int32_t N, int32_t C, int64_t S, ptx_stream_t stream);
int ptx_conv_stem_x3_supported(const ptx_conv3d_desc* desc);
int ptx_conv_stem_x3_fwd(const ptx_conv3d_desc* desc, const float* x, const float* w_packed, const float* bias, float* y,
                         ptx_stream_t stream);

/* The PLANAR split-operand stem (conv_stem_x3.hip, round 4): same convolution, same descriptor and same arithmetic as
 * ptx_conv_stem_x3_fwd for the RGB stems with stride_w == 2 (Ci <= 3, kW <= 7, Wi a multiple of 8) -- `conv1` of the
 * ResNet3D family (resnet3D.py:153), the 2-D ResNet / I3D stems, the (1,7,7) spatial stem of R2Plus1D (r2plus1d.py:73-88)
 * -- on 21 % fewer matrix instructions: the input is six half planes per frame (ptx_ncdhw_to_split_planes:
 * [N][T][c0 c1 c2 hi | c0 c1 c2 lo][H][W], lo scaled by 2^12) and the K = 8 operand is one (kh, channel) run of 8 columns,
 * so a filter row is 3 operands instead of 4.  w_stem: ptx_pack_stem_x3p_weight of the ptx_pack_conv_weight(fold_kw = 1,
 * Ci = 4, Kc = 32, f16 = 2) filter ptx_conv_stem_x3_fwd takes (BatchNorm already folded there), ptx_stem_x3p_weight_elems
 * floats; bias: [Co_pad] of the same pack.  ptx_conv_stem_x3p_supported: 1 if the descriptor can run here (a subset of
 * ptx_conv_stem_x3_supported).                                                                                          */
int ptx_ncdhw_to_split_planes(const float* x, void* y, int32_t N, int32_t C, int32_t T, int32_t H, int32_t W, ptx_stream_t stream);
int ptx_conv_stem_x3p_supported(const ptx_conv3d_desc* desc);
size_t ptx_stem_x3p_weight_elems(const ptx_conv3d_desc* desc);
int ptx_pack_stem_x3p_weight(const ptx_conv3d_desc* desc, const float* w_x3, float* w_stem, ptx_stream_t stream);
int ptx_conv_stem_x3p_fwd(const ptx_conv3d_desc* desc, const void* x, const float* w_stem, const float* bias, float* y,
                          ptx_stream_t stream);

/* RGB STEM convolution on the fp32 matrix cores, read straight from the caller's NCDHW tensor -- `conv1` + `bn1` + ReLU of
 * the ResNet3D family (resnet3D.py:153-155 / :204-206), the 2-D ResNet stem (torchvision_models.py), the (1,7,7) spatial
 * stem of R2Plus1D (r2plus1d.py:73-88), the SAME-padded I3D stem.  No fold / layout pass: a workgroup owns 256 consecutive
 * outputs of one output frame, LDS-DMAs the three channel planes of the input patch of a temporal tap once (16-byte
 * pieces of the NCDHW rows, zero outside the image) and serves all kH x 7 taps from it; K = 21 runs as 11
 * v_mfma_f32_32x32x2_f32 per tap.  fp32 operands, fp32 accumulate.
 * desc: Ci = 3, kW = 7, 2 <= kH <= 8, stride_w <= 2, symmetric or SAME padding (front pad in pT/pH/pW), flags within
 * PTX_EPI_RELU, a patch of at most 12288 floats; Kc is ignored.  ldx = ROW PITCH of x in floats (0 = Wi): a multiple of
 * 4, >= Wi, with the floats [Wi, ldx) of every row zero (ptx_pad_rows makes such a copy for widths that are not multiples
 * of 4).  x: [N][3][T][H][pitch] fp32 with element strides (stride_n, stride_c, stride_t) -- multiples of
 * 4, so `input[:, :, ::step]` (torchvision_models.py frame sub-sampling) is a stride, not a copy.
 * w_stem: ptx_pack_stem_f32_weight of the folded K-major filter ([kT*kH][Co_pad][Kc], k = kw*3 + c: ptx_pack_conv_weight
 * with fold_kw = 1, which also folds the BatchNorm), ptx_stem_f32_weight_elems floats.  bias: [Co_pad] from the same pack.
 * y: [N][To][Ho][Wo][ldy] fp32.  ptx_conv_stem_f32_supported: 1 if the descriptor / strides can run here (else use
 * ptx_fold_kw_* + ptx_conv3d_fwd).                                                                                   */
int ptx_conv_stem_f32_supported(const ptx_conv3d_desc* desc, int64_t stride_n, int64_t stride_c, int64_t stride_t);
size_t ptx_stem_f32_weight_elems(const ptx_conv3d_desc* desc);
int ptx_pack_stem_f32_weight(const ptx_conv3d_desc* desc, const float* w_folded, int32_t Kc, float* w_stem,
                             ptx_stream_t stream);
int ptx_conv_stem_f32_fwd(const ptx_conv3d_desc* desc, const float* x, int64_t stride_n, int64_t stride_c, int64_t stride_t,
                          const float* w_stem, const float* bias, float* y, ptx_stream_t stream);

/* 3x3x3 BODY convolution on the fp32 matrix cores, input patch resident in LDS -- `conv2` of every Bottleneck
 * (resnet3D.py:117,129-131: conv3x3x3 -> bn2 -> relu) and both convs of a BasicBlock (resnet3D.py:86-104), the layers that
 * hold 48 % of config 2's FLOPs.  A workgroup owns 256 (or 64) consecutive outputs of one frame x 64 channels; per
 * (temporal tap, 16-channel chunk) it stages the halo'd input patch ONCE by LDS-DMA and serves all nine (kh, kw) taps from
 * it; filter slices stream through a three-slot ring (csrc/conv_body_f32.hip).
 *   desc: dense fp32 conv, kT in {1, 3}, kH = kW = 3, unit strides, padding (kT / 2, 1, 1), Ci a multiple of 16 (ldx >= Ci,
 *         multiple of 4), Co_pad a multiple of 64, flags within PTX_EPI_RELU | PTX_EPI_RES_ADD (res: same shape as y, row
 *         stride ldr), a halo'd patch of at most 472 positions.
 *   shape: 0 = tall tiles (4 waves x 64 rows x 64 columns) + square tiles for the tail of a frame; 1 = square tiles only
 *          (2 x 2 waves x 32 x 32: small frames).  ptx_conv_body_f32_supported answers per (desc, shape).
 *   w_body: ptx_pack_conv_body_f32_weight(desc, w_packed) from the ptx_pack_conv_weight image of the same filter
 *          ([tap][Co_pad][Kc], BN folded) -- ptx_conv_body_f32_weight_elems floats.
 * The same entry points also take the TEMPORAL half of a SpatioTemporalConv (r2plus1d.py:84-88) on a T-STACKED tile (shape 0
 * only): kT in {3, 5, 7}, kH = kW = 1, unit strides, padding (kT / 2, 0, 0); a workgroup owns the same 32 positions of 8
 * consecutive output frames and stages the 8 + kT - 1 input frames they touch once per 16-channel chunk.  Channels are walked
 * in 16-wide chunks: ceil(Ci / 16) * 16 <= min(Kc, ldx), the columns [Ci, that) of x finite (their filter rows are zero).
 * Arithmetic: fp32 operands, fp32 accumulate; the k order differs from ptx_conv3d_fwd's tiles (fp32 reorder noise, 1e-6). */
int ptx_conv_body_f32_supported(const ptx_conv3d_desc* desc, int shape);
size_t ptx_conv_body_f32_weight_elems(const ptx_conv3d_desc* desc);
int ptx_pack_conv_body_f32_weight(const ptx_conv3d_desc* desc, const float* w_packed, float* w_body, ptx_stream_t stream);
int ptx_conv_body_f32_fwd(const ptx_conv3d_desc* desc, const float* x, const float* w_body, const float* bias, const float* res,
                          float* y, int shape, ptx_stream_t stream);
/* The body kernel with a CHAINED 1x1x1 tail -- a Bottleneck's `conv2 -> bn2 -> relu -> conv3 -> bn3 -> out += residual -> relu`
 * (resnet3D.py:129-142) in ONE launch, as ptx_conv3d_chain_fwd does on the implicit-GEMM tiles: the [rows][<= 64] result of the
 * 3x3x3 conv is parked in LDS 32 rows at a time and fed to the tail as its A operand; the tail's filter fragments come from
 * global memory (L2-resident).
 *   conv: as ptx_conv_body_f32_fwd, Co <= 64, flags within PTX_EPI_RELU (the ReLU BETWEEN the convs), ldy ignored.
 *   tail: 1x1x1 / unit stride over conv's output positions, Ci = conv->Co, Co_pad a multiple of 64, flags within
 *         PTX_EPI_RELU | PTX_EPI_RES_ADD (res: same shape as y, row stride ldr), ldy = row stride of y.
 *   w_tail: ptx_pack_conv_body_tail_f32_weight(tail, w2_packed) from the ptx_pack_conv_weight image of the tail's filter. */
int ptx_conv_body_chain_f32_supported(const ptx_conv3d_desc* conv, const ptx_conv3d_desc* tail, int shape);
size_t ptx_conv_body_tail_f32_weight_elems(const ptx_conv3d_desc* tail);
int ptx_pack_conv_body_tail_f32_weight(const ptx_conv3d_desc* tail, const float* w2_packed, float* w_tail, ptx_stream_t stream);
int ptx_conv_body_chain_f32_fwd(const ptx_conv3d_desc* conv, const ptx_conv3d_desc* tail, const float* x, const float* w_body,
                                const float* bias, const float* w_tail, const float* bias2, const float* res, float* y, int shape,
                                ptx_stream_t stream);

/* Operands of the fused generator-stage epilogue (see PTX_EPI_AFFINE / PTX_EPI_DUAL_RAW). */
typedef struct ptx_conv_fused_ext {
    const float* scale;   /* [N][ld_affine] per-sample, per-output-channel scale (ptx_cbn_fold)  */
    const float* shift;   /* [N][ld_affine]                                                      */
    int32_t ld_affine;
    int32_t ld_raw;       /* row stride of y_raw in halfs                                        */
    void* y_raw;          /* [M][ld_raw] halfs: the pre-affine output                            */
} ptx_conv_fused_ext;
/* ptx_conv3d_fwd for fp16-operand descriptors (PTX_F16_OPERANDS) with the fused-stage flags: x / w_packed hold
 * halfs, res halfs (PTX_RES_F16) or fp32, y halfs (PTX_EPI_OUT_F16) or fp32.  ext may be NULL when neither
 * PTX_EPI_AFFINE nor PTX_EPI_DUAL_RAW is set. */
int ptx_conv3d_fused_fwd(const ptx_conv3d_desc* desc, const void* x, const void* w_packed, const float* bias,
                         const void* res, void* y, const ptx_conv_fused_ext* ext, void* workspace,
                         size_t workspace_bytes, int config, int split_k, ptx_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * Weight packing: fold eval-mode BatchNorm (+ conv bias) into the filter and re-lay it out
 * K-major for the MFMA B operand.  One-off at load time.
 * Replaces: the per-forward batch_norm after every conv (resnet3D.py:127,131,135; 154; 184).
 *   scale[co] = gamma[co] / sqrt(var[co] + eps)         (1 when gamma == NULL)
 *   w_packed[tap][co][k] = w[co][c][kt][kh][kw] * scale[co]
 *   bias_out[co]         = beta[co] + (conv_bias[co] - mean[co]) * scale[co]
 * fold_kw == 0: tap = (kt*kH + kh)*kW + kw, k = c            -> [kT*kH*kW][Co_pad][Kc]
 * fold_kw == 1: tap = kt*kH + kh,          k = kw*Ci + c     -> [kT*kH   ][Co_pad][Kc]
 *               (small-Cin stem: the kW taps are folded into the channel axis, see
 *                ptx_fold_kw_ncdhw)
 * Rows co >= Co and columns k >= K are written as zero.
 * ------------------------------------------------------------------------------------------ */
typedef struct ptx_pack_desc {
    int32_t Co, Ci, kT, kH, kW;
    int32_t Kc, Co_pad;
    int32_t fold_kw;
    /* K-concatenated packing (ptx_conv3d_dual_fwd): row stride of w_packed in floats (0 = Kc), first
     * column written, and whether bias_out is accumulated into instead of overwritten */
    int32_t ld_k, k_off, bias_accumulate;
    /* Grouped filters packed as block-diagonal SUPER-groups (0 or 1 = off): w is [Co][Ci/sub_groups][taps],
     * i.e. sub_groups real groups of Ci/sub_groups input channels share one packed row of Ci columns, with
     * zeros outside a row's own group.  A cardinality-32 conv of group width 4 (resnext3D.py:85-92) becomes
     * a 32-wide grouped conv (desc.groups = 32/8) that the MFMA tiles can run -- 8x padded work, but
     * coalesced LDS-staged operands instead of per-lane 16-byte gathers.  co_per_super = output channels
     * of one super-group (rows [s*co_per_super, (s+1)*co_per_super) share the input columns of super-group s). */
    int32_t sub_groups, co_per_super;
    /* 1: w_packed is written as IEEE halfs (Kc counts halfs, multiple of 8; ptx_packed_weight_elems counts
     * halfs) for PTX_F16_OPERANDS convs.  2: split halfs for PTX_F16X3_OPERANDS convs -- Kc, ld_k, k_off count
     * channels as in the fp32 layout (multiples of 8) and the buffer has the fp32 layout's size, but every 8-channel
     * block of a row holds 8 hi halfs then 8 lo halfs (hi = half(w), lo = half((w - hi) * 2^12): the SCALED lo of
     * PTX_F16X3_OPERANDS, whose cross terms the kernels accumulate separately and fold back by 2^-12).  bias_out stays fp32. */
    int32_t f16;
} ptx_pack_desc;

size_t ptx_packed_weight_elems(const ptx_pack_desc* desc);
int ptx_pack_conv_weight(const ptx_pack_desc* desc, const float* w /* [Co][Ci][kT][kH][kW] */,
                         const float* conv_bias, const float* bn_gamma, const float* bn_beta,
                         const float* bn_mean, const float* bn_var, float bn_eps,
                         float* w_packed, float* bias_out /* [Co_pad] */, ptx_stream_t stream);

/* Weight-change detection for the host-side cache of packed filters (the reference re-reads its parameters
 * on every forward, resnet3D.py:125-143, so edits such as `m.weight.data.fill_(1)`, resnet3D.py:199-201, take
 * effect immediately): *out += a 64-bit position-sensitive sum over the bit patterns of n fp32 tensors,
 * table[i] = (device pointer, element count) as int64 pairs in device memory; the caller zeroes *out first. */
int ptx_checksum_f32(const int64_t* table, int32_t n, uint64_t* out, ptx_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * Layout transforms at the API edge (the reference's tensors are NCDHW, torchvision_models.py:448).
 * ------------------------------------------------------------------------------------------ */
/* x [N][C][S] -> y [N][S][ld]  (S = T*H*W; channels [C, ld) zero-filled) */
int ptx_ncdhw_to_ndhwc(const float* x, float* y, int32_t N, int32_t C, int64_t S, int32_t ld,
                       ptx_stream_t stream);
/* x [N][S][ld] -> y [N][C][S] */
int ptx_ndhwc_to_ncdhw(const float* x, float* y, int32_t N, int32_t C, int64_t S, int32_t ld,
                       ptx_stream_t stream);
/*
 * Small-Cin stem input: x NCDHW [N][C][T][H][W]  ->  y [N][T][H][Wo][ld] with
 *   y[n][t][h][wo][kw*C + c] = x[n][c][t][h][wo*sW - pW + kw]   (0 outside the image / k >= kW*C)
 * so that the stem (resnet3D.py:153, Conv3d 3->64 k7 s(1,2,2) p3) becomes a (kT,kH,1) conv over
 * ld = 24 "channels" with 96-byte contiguous rows.
 */
int ptx_fold_kw_ncdhw(const float* x, float* y, int32_t N, int32_t C, int32_t T, int32_t H,
                      int32_t W, int32_t kW, int32_t sW, int32_t pW, int32_t Wo, int32_t ld,
                      ptx_stream_t stream);
/* Same, reading a strided view of the clip: element (n,c,t,h,w) lives at
 * x[n*stride_n + c*stride_c + t*stride_t + h*W + w] (rows stay contiguous).  Temporal subsampling
 * `input[:, :, ::tau]` (slowfast.py:228,345,393-394) is stride_t = tau*H*W with T = ceil(T_full/tau). */
int ptx_fold_kw_strided(const float* x, float* y, int32_t N, int32_t C, int32_t T, int32_t H, int32_t W,
                        int64_t stride_n, int64_t stride_c, int64_t stride_t, int32_t kW, int32_t sW,
                        int32_t pW, int32_t Wo, int32_t ld, ptx_stream_t stream);

/* y[r][0..W) = x[r][0..W), y[r][W..ld) = 0 for r < rows: gives image rows whose width is not a multiple of 4 floats the
 * 16-byte pitch the direct stem reads (ptx_conv_stem_f32_fwd with desc.ldx = ld); the reference takes any T/H/W
 * (torchvision_models.py:448, adaptive pooling).                                                                       */
int ptx_pad_rows(const float* x, float* y, int64_t rows, int32_t W, int32_t ld, ptx_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * Pre-processing edge: the tensor half of TransformImage (transforms/utils.py:72-75) on decoded
 * uint8 frames, on the device:  ToTensor (u8 -> f32, /255), ToSpaceBGR (swap channels 0 and 2),
 * ToRange255 (*255), Normalize ((v - mean[c]) / std[c]) -- the same fp32 operations in the same
 * order, so results are bit-identical to the reference's CPU tensors.
 * frames: [N][T][H][W][C] uint8 (decoded video frames, channel-interleaved), C <= 4.
 * ------------------------------------------------------------------------------------------ */
typedef struct ptx_norm_desc {
    float mean[4], std[4];   /* per OUTPUT channel (after the optional BGR swap) */
    int32_t swap_rb;         /* input_space == 'BGR'  (utils.py:73)              */
    int32_t to_255;          /* max(input_range) == 255 (utils.py:74)            */
} ptx_norm_desc;
/* -> y NCDHW fp32 [N][C][T][H][W]: exactly the tensor the reference models take */
int ptx_frames_u8_to_ncdhw(const uint8_t* frames, float* y, int32_t N, int32_t T, int32_t H, int32_t W,
                           int32_t C, const ptx_norm_desc* norm, ptx_stream_t stream);
/* -> the stem's kW-folded operand (see ptx_fold_kw_ncdhw) in one pass: normalise + fold, no fp32
 * NCDHW tensor is materialised.  frame_step: use every frame_step-th frame (T = frames used). */
int ptx_fold_kw_frames_u8(const uint8_t* frames, float* y, int32_t N, int32_t C, int32_t T, int32_t H,
                          int32_t W, int32_t frame_step, int32_t T_full, int32_t kW, int32_t sW, int32_t pW,
                          int32_t Wo, int32_t ld, const ptx_norm_desc* norm, ptx_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * Pooling and head.
 * ------------------------------------------------------------------------------------------ */
#define PTX_POOL_SAME 1u      /* To/Ho/Wo are given (TF "SAME": ceil(in/stride)); pT/pH/pW are the FRONT
                                 pads, the back pads are whatever the output extent implies        */
#define PTX_POOL_PAD_ZERO 2u  /* padded taps contribute 0 (F.pad then MaxPool3d) instead of -inf     */
typedef struct ptx_pool3d_desc {
    int32_t N, Ti, Hi, Wi, C, ld; /* input NDHWC, ld = input row stride */
    int32_t To, Ho, Wo;
    int32_t kT, kH, kW, sT, sH, sW, pT, pH, pW;
    int32_t ldy;                  /* output row stride; 0 = same as ld.  Columns [0, round_up(C,4)) are
                                     written, so y may be a channel slice of a concatenated tensor */
    uint32_t flags;               /* PTX_POOL_* */
} ptx_pool3d_desc;
/* max_pool3d with -inf padding (resnet3D.py:156: MaxPool3d k3 s2 p1; slowfast.py:123 (1,3,3)) */
int ptx_maxpool3d_fwd(const ptx_pool3d_desc* desc, const float* x, float* y, ptx_stream_t stream);
/* Class-conditional BatchNorm folded to a per-sample affine (BigGAN `ccbn`: F.batch_norm with the stored
 * statistics, then * (1 + gain(y)) + bias(y)):
 *   scale[n][c] = m[n][c] / sqrt(var[c] + eps),  shift[n][c] = bias[n][c] - mean[c] * scale[n][c]
 * with m = 1 + gain[n][c] (plus_one != 0) or gain[n][c] itself (plain BN: pass gamma with ld_gain = 0).
 * gain / bias rows have stride ld_gain / ld_bias (0 = one row shared by every sample); scale / shift rows
 * have stride ld_out. */
int ptx_cbn_fold(const float* gain, const float* bias, const float* mean, const float* var, float eps,
                 float* scale, float* shift, int32_t N, int32_t C, int32_t ld_gain, int32_t ld_bias,
                 int32_t ld_out, int32_t plus_one, ptx_stream_t stream);
/* y[n][h][w][c] = act(x[n][h/up][w/up][c] * scale[n][c] + shift[n][c]): the generator's
 * cBN -> ReLU -> nearest-upsample stage in one HBM pass (x NHWC row stride ldx, y row stride ldy;
 * scale/shift rows have stride ld_scale, so every cBN of the network can share one folded table; ld_scale = 0:
 * one row for all samples, i.e. a plain eval-mode BN -> ReLU ahead of a conv, pre_act_resnet3D.py:41-47).
 * act: 0 none, 1 ReLU, 2 tanh (output layer).  scale/shift may be NULL (identity affine). */
int ptx_affine_act_upsample(const float* x, float* y, const float* scale, const float* shift, int32_t ld_scale,
                            int32_t N, int32_t H, int32_t W, int32_t C, int32_t ldx, int32_t ldy, int32_t up,
                            int32_t act, ptx_stream_t stream);
/* Non-local 'concatenation' affinity (nonlocalnet.py:213-243): the 1x1 conv over cat([theta_i, phi_j]) is
 * a[i] + b[j] with a = theta . w[:ci], b = phi . w[ci:], so
 *   f[n][i][j] = relu(a[n][i] + b[n][j]) / cols        (f row stride ldf; columns [cols, ldf) zeroed) */
int ptx_outer_sum_relu(const float* a, const float* b, float* f, int32_t batch, int32_t rows, int32_t cols,
                       int32_t ldf, ptx_stream_t stream);
/* y[r][0..cols) = x[r][0..cols) for r < rows (row strides ldx / ldy, all multiples of 4): places a
 * tensor into a channel slice of another -- torch.cat(dim=1) plumbing (slowfast.py:145, 395) */
int ptx_copy2d(const float* x, float* y, int64_t rows, int32_t cols, int64_t ldx, int64_t ldy,
               ptx_stream_t stream);
/* y[o][j][i] = mean_{d<k} x[o][j*stride + d][i]   for j < (T - k)/stride + 1   (x [outer][T][inner]):
 * temporal average windows / mean over the remaining time steps of a per-frame head */
int ptx_window_mean(const float* x, float* y, int32_t outer, int32_t T, int32_t inner, int32_t k,
                    int32_t stride, ptx_stream_t stream);
/* adaptive_avg_pool3d(1): x [N][S][ld] (channels-last) or [N][C][S] (channels_first != 0)
 * -> y [N][C]   (torchvision_models.py:461) */
int ptx_global_avgpool(const float* x, float* y, int32_t N, int32_t C, int64_t S, int32_t ld,
                       int32_t channels_first, ptx_stream_t stream);
/* y[m][j] = act_out( sum_k act_in(x[m][k]) * w[j][k] + b[j] ), small M (GEMV-class, weight-
 * bandwidth bound): last_linear (torchvision_models.py:463) and the TRN relation MLP at small
 * batch (trn.py:39-45).  flags: PTX_PRO_RELU | PTX_EPI_RELU | PTX_EPI_ACCUM.  b may be NULL. */
int ptx_linear_fwd(const float* x, const float* w, const float* b, float* y, int32_t M, int32_t K,
                   int32_t Nout, int32_t ldx, int32_t ldy, uint32_t flags, ptx_stream_t stream);

/* TRN relation MLPs over frame subsets (trn.py:39-45 inside MultiScaleRelation.forward :101-110) without
 * materialising the gathered inputs.  x: per-frame features [B][T][frame_len] (video stride ldx).
 * A descriptor names n_sets frame subsets of n_frames frames each; output row r*B + b is
 *   y[r*B + b][j] = act_out( sum_k act_in(concat_f x[b][idx[r][f]][:])[k] * w[j][k] + b[j] ),  K = n_frames*frame_len
 * so the subsets that share one Relation's weights (same scale) run as ONE launch reading w once. */
#define PTX_REL_MAX_SETS 8
#define PTX_REL_MAX_FRAMES 16
typedef struct ptx_relation_desc {
    int32_t B, n_sets, n_frames, frame_len;
    int32_t idx[PTX_REL_MAX_SETS][PTX_REL_MAX_FRAMES];
} ptx_relation_desc;
int ptx_relation_linear_fwd(const ptx_relation_desc* desc, const float* x, int32_t ldx, const float* w, const float* b,
                            float* y, int32_t Nout, int32_t ldy, uint32_t flags, ptx_stream_t stream);
/* Second Linear of those relations, summed over the subsets (trn.py:110 `stack(output).sum(0)`), by
 * linearity:  y[b][j] (+)= sum_k (sum_r x[r*B + b][k]) * w[j][k] + n_sets * b[j]     (flags: EPI_ACCUM, EPI_RELU) */
int ptx_linear_setsum_fwd(const float* x, const float* w, const float* b, float* y, int32_t B, int32_t n_sets,
                          int32_t K, int32_t Nout, int32_t ldx, int32_t ldy, uint32_t flags, ptx_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * Non-local block pieces (nonlocalnet.py:143-166).
 * ------------------------------------------------------------------------------------------ */
/* The whole attention core of a non-local block in ONE launch, flash-style (online softmax): the [Nq][Nk]
 * affinity f = theta^T phi (nonlocalnet.py:156) is never written to memory.
 *   mode PTX_NL_SOFTMAX: y[b][i][:] = sum_j softmax_j(theta[b][i] . phi[b][j]) g[b][j][:]   (embedded_gaussian :143-166,
 *                        gaussian :168-190 with theta = phi = x; no 1/sqrt(d) scaling, as upstream)
 *   mode PTX_NL_SCALE:   y[b][i][:] = sum_j (theta[b][i] . phi[b][j] / Nk) g[b][j][:]        (dot_product :192-211)
 * theta [batch][Nq][ld_theta] (d channels used), phi [batch][Nk][ld_phi] (d), g [batch][Nk][ld_g] (dv),
 * y [batch][Nq][ld_y] (dv written); channels-last rows, so the operands may be channel slices of one fused
 * theta|phi|g projection and Nk != Nq covers `sub_sample` (max-pooled phi / g, :126-131) and BigGAN's pooled keys.
 * fp32 MFMA throughout; differs from softmax-then-matmul only by fp32 summation order (+ v_exp_f32 rounding).
 *   mode PTX_NL_SCALE | PTX_NL_RELU: y[b][i][:] = sum_j (relu(theta[b][i] . phi[b][j]) / Nk) g[b][j][:] -- the 'concatenation'
 *                        affinity (:213-243): the 1x1 conv over cat([theta_i, phi_j]) is a_i + b_j, i.e. the dot product of
 *                        the 2-vectors theta' = (a_i, 1) and phi' = (1, b_j) (d = 4 rows: two live columns, two zero)
 * Limits: d, dv multiples of 4, d <= 1024 (ptx_nonlocal_supported); d > 512 (gaussian mode on 1024 channels: theta = x)
 * streams the theta fragments from global memory instead of registers; larger d: ptx_bgemm_nt + ptx_softmax_rows. */
#define PTX_NL_SOFTMAX 0
#define PTX_NL_SCALE 1
#define PTX_NL_F16 2   /* OR into mode (softmax, d <= 64): both matmuls on v_mfma_f32_16x16x16_f16 -- theta / phi / g / P are rounded
                          to halfs in registers, accumulators / softmax statistics / y stay fp32.  The BigGAN generator's
                          self-attention under its fp16 plan (BASELINE config 5); the video nets keep the fp32 kernel. */
#define PTX_NL_RELU 8  /* OR into PTX_NL_SCALE: P = relu(S) / Nk (concatenation mode)                                               */
#define PTX_NL_X3 4    /* OR into mode: fp32-ACCURATE split operands on the same fp16 MFMAs (theta / phi / g / P as (hi, lo) half
                          pairs, a.b = hi.lo + lo.hi + hi.hi, fp32 accumulate) -- the attention of a plan compiled with
                          Engine.precision = "x3" (PTX_F16X3_OPERANDS convs).  Exclusive with PTX_NL_F16.  The lo halves of
                          theta / phi / g are UNSCALED here (unlike the convs): 22 bits per product for |v| >= 2^-3, fewer
                          below (lo goes subnormal) -- activations entering an NL block are O(1); the softmax weights, which
                          are not (P ~ 1 / Nk), are split as 2^12 P with the factor folded into 1 / l.  d > 512 runs the
                          exact fp32 kernel.                                                                            */
#define PTX_NL_OUT_F16 16 /* OR into PTX_NL_F16: y is written as halfs (ld_y / bs_y count halfs, ld_y a multiple of 4): the generator's
                             attention output feeds a half conv (ptx_conv1x1_skip_f16_fwd)                                    */
typedef struct ptx_nonlocal_desc {
    int32_t batch, Nq, Nk, d, dv;
    int32_t ld_theta, ld_phi, ld_g, ld_y;        /* row strides (floats)   */
    int64_t bs_theta, bs_phi, bs_g, bs_y;        /* batch strides (floats) */
    int32_t mode;
} ptx_nonlocal_desc;
int ptx_nonlocal_supported(const ptx_nonlocal_desc* desc);
int ptx_nonlocal_fwd(const ptx_nonlocal_desc* desc, const float* theta, const float* phi, const float* g, float* y,
                     ptx_stream_t stream);
/* Batched C[b] = op(A[b] x B[b]^T): A [batch][M][lda] (row-major, K contiguous),
 * B [batch][Nn][ldb] (row-major, K contiguous), C [batch][M][ldc]; fp32 MFMA.
 * Used for f = theta^T phi  (nonlocalnet.py:156) and y = softmax(f) g  (:160). */
int ptx_bgemm_nt(const float* A, const float* B, float* C, int32_t batch, int32_t M, int32_t Nn,
                 int32_t K, int32_t lda, int32_t ldb, int32_t ldc, int64_t strideA, int64_t strideB,
                 int64_t strideC, ptx_stream_t stream);
/* in-place row softmax over the last dim: x [rows][ld], `cols` valid entries (nonlocalnet.py:157);
 * scale_only != 0 -> x /= cols instead (dot_product mode, :204-205) */
int ptx_softmax_rows(float* x, int64_t rows, int32_t cols, int32_t ld, int32_t scale_only,
                     ptx_stream_t stream);
/* y [N][S][ld] (channels-last) -> yt [N][ld_c][S_pad]: per-sample transpose used to present g as
 * the K-contiguous B operand of y = f g */
int ptx_transpose_last2(const float* x, float* y, int32_t batch, int32_t R, int32_t Cc, int32_t ldx,
                        int32_t ldy, ptx_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * fp16 kernels of the generator stage designed for the fp16 matrix cores (gen_stage_f16.hip; BASELINE config 5).
 * BigGAN-deep has no source in the reference snapshot: the layer definitions follow Brock et al. 2019,
 * appendix B (BigGAN-deep generator); parity of everything below is UNPINNED.
 *
 * ptx_rgb_conv3x3_f16_fwd -- the generator's output layer in one launch:
 *     y[n][h][w][co] = tanh( bias[co] + sum_{kh,kw,c} relu(x[n][h+kh-1][w+kw-1][c] * scale[n][c] + shift[n][c]) * w[co][c][kh][kw] )
 * i.e. BN (folded to a per-sample affine by ptx_cbn_fold) -> ReLU -> conv3x3(C -> 3, zero padding of the ACTIVATED
 * map) -> tanh (PTX_EPI_TANH in flags; 0 = no tanh).  x: halfs [N][H][W][ldx], the RAW output of the last GBlock;
 * y: fp32 [N][H][W][ldy] (3 live columns; with ldy >= 4 a zero 4th column is written: one 16-byte store per pixel).
 * The nine taps sit in the N axis of ONE GEMM over the input positions (K = C, N = 27 -> 32), the shifted partial sums
 * are combined through LDS: the input is read once, no activated copy of it exists.
 * ------------------------------------------------------------------------------------------ */
typedef struct ptx_rgb_conv_desc {
    int32_t N, H, W, C;      /* C = 32, 64 or 128                                      */
    int32_t ldx;             /* channel stride of x in halfs (multiple of 8, >= C)     */
    int32_t ldy;             /* row stride of y in floats (3, or a multiple of 4)      */
    int32_t ld_affine;       /* row stride of scale / shift in floats (multiple of 4)  */
    uint32_t flags;          /* PTX_EPI_TANH or 0                                      */
} ptx_rgb_conv_desc;
int ptx_rgb_conv3x3_f16_supported(const ptx_rgb_conv_desc* desc);
size_t ptx_rgb_conv_weight_elems(int32_t C);                 /* halfs of the packed filter */
/* w [3][C][3][3] fp32 (torch Conv2d layout) -> the kernel's B-fragment order, halfs */
int ptx_pack_rgb_conv_weight(const float* w, int32_t C, void* w_packed, ptx_stream_t stream);
int ptx_rgb_conv3x3_f16_fwd(const ptx_rgb_conv_desc* desc, const void* x, const float* scale, const float* shift,
                            const void* w_packed, const float* bias /* [3] or NULL */, float* y, ptx_stream_t stream);

/* ptx_conv3x3_f16_fwd -- the 3x3 convs of a GBlock (conv2 / conv3 of BigGAN-deep's bottleneck, 64 / 128 / 256 channels in
 * and out: the 32^2 ... 256^2 stages) from ONE staged input patch per 8 x 32 output tile:
 *     y = half( relu?( (conv3x3(up2?(x)) + bias) * scale[n] + shift[n] ) )
 * Same descriptor, operands and flag semantics as ptx_conv3d_fused_fwd restricted to what ptx_conv3x3_f16_supported
 * accepts: PTX_F16_OPERANDS | PTX_EPI_OUT_F16 [| PTX_PRO_UP2] [| PTX_EPI_AFFINE] [| PTX_EPI_RELU], kT = 1, 3 x 3, unit
 * stride, pad 1, Ci == Co (64, 128 or 256 channels), at least 32 columns, no residual; w_packed / bias from
 * ptx_pack_conv_weight (f16 = 1).
 * ext may be NULL without PTX_EPI_AFFINE.  No workspace, no tile configuration. */
/* ptx_conv1x1_skip_f16_fwd -- the closing 1x1 conv of a GBlock (conv4 of BigGAN-deep's bottleneck: C/4 -> C' channels) with
 * everything the next block reads produced in its epilogue:
 *     v = conv1x1(x) + bias + skip;   y_raw = half(v)  [PTX_EPI_DUAL_RAW];   y = half(relu?(v * scale[n] + shift[n]))  [PTX_EPI_AFFINE]
 * (without PTX_EPI_AFFINE: y = half(relu?(v))).  skip = res halfs (PTX_RES_F16): same shape (PTX_EPI_RES_ADD) or nearest-
 * upsampled by 2^res_sH and channel-truncated (PTX_EPI_RES_PADA | PTX_EPI_RES_UP, res_sH == res_sW in 0..1, res_sT = 0).
 * Same descriptor / ext conventions as ptx_conv3d_fused_fwd; accepted: 64 / 128 / 256 input channels, Co a multiple of
 * 128, at least 32 columns, halfs in and out. */
int ptx_conv1x1_skip_f16_supported(const ptx_conv3d_desc* desc);
int ptx_conv1x1_skip_f16_fwd(const ptx_conv3d_desc* desc, const void* x, const void* w_packed, const float* bias, const void* res,
                             void* y, const ptx_conv_fused_ext* ext, ptx_stream_t stream);
/* ptx_conv1x1_pro_f16_fwd -- the opening 1x1 conv of a GBlock (conv1: C -> C/4) with the block's cBN1 + ReLU applied to its
 * INPUT fragments, so the previous block's closing conv stores one tensor (the raw sum) instead of two:
 *     y = half(relu?(scale[n] * (W . relu(x * scale_in[n] + shift_in[n]) + bias) + shift[n]))
 * x: RAW halfs.  ext_in carries the input affine (scale / shift / ld_affine over the K input channels; the other fields are
 * ignored), ext the output affine (PTX_EPI_AFFINE) as for the other fused stages.  Accepted: K a multiple of 128 in
 * 128..2048, Co = 64 / 128 / 256 / 512, H * W a multiple of 256, halfs in and out. */
int ptx_conv1x1_pro_f16_supported(const ptx_conv3d_desc* desc);
int ptx_conv1x1_pro_f16_fwd(const ptx_conv3d_desc* desc, const void* x, const ptx_conv_fused_ext* ext_in, const void* w_packed,
                            const float* bias, void* y, const ptx_conv_fused_ext* ext, ptx_stream_t stream);
int ptx_conv3x3_f16_supported(const ptx_conv3d_desc* desc);
int ptx_conv3x3_f16_fwd(const ptx_conv3d_desc* desc, const void* x, const void* w_packed, const float* bias, void* y,
                        const ptx_conv_fused_ext* ext, ptx_stream_t stream);

/* ============================================================================================
 * EXPERIMENTAL entry points (PTX_EXPERIMENTAL_API).  Built, tested and exported, but NOT on the default path: each was
 * measured SLOWER than what the engine runs by default on MI355X and is reachable only through an opt-in environment
 * switch.  Kept as reproducible negative results (numbers and A/B logs under profiles/); their signatures may change or
 * disappear without a version bump, and no reference-side binding should be written against them.
 *   ptx_conv_program_*            conv programs (PTX_PROGRAM=1|auto): config 2 -3.7 %, config 3 -13 % vs the launches
 *                                 (profiles/r05_program_probe.txt, r05_program_engine_ab.txt)
 *   ptx_nonlocal_workspace_bytes,
 *   ptx_nonlocal_ws_fwd           stream-K attention (PTX_NL_STREAMK=1): 246 vs 233 us at 8 clips x N = 1568
 *                                 (profiles/r05_attention_streamk.txt)
 * ============================================================================================ */
#define PTX_EXPERIMENTAL_API /* marks the declarations below; expands to nothing */

/* --------------------------------------------------------------------------------------------
 * Conv PROGRAM: an ordered list of fp32 convolutions (bias / ReLU / same-shape residual epilogues, optional second source)
 * executed by ONE persistent launch -- the small-M tail of a video ResNet, where a launch per conv is bound by tile
 * quantisation on 256 CUs and per-launch ramps rather than by the matrix cores.  Replaces the op sequence of whole
 * bottlenecks:  conv1 -> bn1 -> relu -> conv2 -> bn2 -> relu -> conv3 -> bn3 -> += residual -> relu, block after block
 * (resnet3D.py:125-143 with shortcut B :175-185; the six factored GEMMs of a (2+1)D bottleneck, r2plus1d.py:68-88).
 * The tiles of all stages sit on one queue -- in wavefront order over (stage, clip group), so tiles of several stages are
 * runnable at any time -- persistent workgroups take them in order and a tile waits only for the ROW TILES of the producing
 * stages it reads (per-row-tile completion counters); no grid barrier exists, the launch completes for any grid size
 * (csrc/conv_program.hip).  Arithmetic and k-order are those of
 * ptx_conv3d_fwd on the same tile and split: results are bit-identical to the stage-by-stage launches.
 *   stage.tile     index into ptx_conv_program_tile_name (the tile shapes compiled into the program kernel), < 0: library's pick
 *   stage.split_k  <= 0: library's pick.  Split-K stages reduce in-launch (last arriver, split order).
 * Rules: every stage writes its own buffer (no reuse inside a program); a stage that reads an earlier stage's output reads
 * it with the producer's row stride; flags other than PTX_EPI_RELU | PTX_EPI_RES_ADD, grouped and fp16 / split-operand convs
 * are refused (PTX_ERR_UNSUPPORTED: keep those as their own launches).
 * Use: ptx_conv_program_plan (sizes) -> allocate `workspace_bytes` of device workspace (256-byte aligned) and `image_bytes`
 * of host + device memory -> ptx_conv_program_build fills the HOST image (it embeds the stage pointers and workspace
 * addresses) -> copy it to the device once -> ptx_conv_program_fwd per forward (asynchronous; one memset node + one kernel).
 * ptx_conv_program_error synchronises the stream and reads the program's error word (a dependency wait that ran out of
 * polls -- a bug or a wedged device, never a data-dependent condition): code4 = {code, waiting stage, queue index, producer}.
 * ------------------------------------------------------------------------------------------ */
typedef struct ptx_conv_stage {
    ptx_conv3d_desc desc;
    const float* x;          /* input  (may be an earlier stage's y)                         */
    const float* x2;         /* second source (ptx_conv3d_dual_fwd semantics) or NULL        */
    const float* w_packed;
    const float* bias;       /* or NULL                                                      */
    const float* res;        /* PTX_EPI_RES_ADD operand (may be an earlier stage's y) or NULL */
    float* y;
    int32_t tile;
    int32_t split_k;
} ptx_conv_stage;

typedef struct ptx_conv_program_info {
    int32_t n_stages, total_items;   /* queue length = sum over stages of tiles x splits              */
    int32_t ctrl_words;              /* 32-bit words at the head of the workspace zeroed per launch   */
    int32_t lds_bytes;               /* dynamic LDS of the program kernel                             */
    int32_t launches_replaced;       /* conv + split-K reduce launches the program stands for         */
    int32_t n_chunks;                /* (stage, clip group) runs of the queue                         */
    uint64_t image_bytes, workspace_bytes;
} ptx_conv_program_info;

PTX_EXPERIMENTAL_API int ptx_conv_program_num_tiles(void);
PTX_EXPERIMENTAL_API const char* ptx_conv_program_tile_name(int tile);
PTX_EXPERIMENTAL_API int ptx_conv_program_plan(const ptx_conv_stage* stages, int32_t n, ptx_conv_program_info* info);
/* the plan as text, one line per stage: tile, split, tiles, halo rows, producers ("<stage>:x|res|x2") -- host only */
PTX_EXPERIMENTAL_API int ptx_conv_program_describe(const ptx_conv_stage* stages, int32_t n, char* text, size_t text_bytes);
PTX_EXPERIMENTAL_API int ptx_conv_program_build(const ptx_conv_stage* stages, int32_t n, void* workspace, size_t workspace_bytes,
                           void* image_host, size_t image_bytes, ptx_conv_program_info* info);
PTX_EXPERIMENTAL_API int ptx_conv_program_fwd(const ptx_conv_program_info* info, const void* image_dev, void* workspace, int32_t wgs_per_cu,
                         ptx_stream_t stream);
PTX_EXPERIMENTAL_API int ptx_conv_program_error(const void* workspace, int32_t* code4, ptx_stream_t stream);
/* diagnostic: the same launch with a phase clock -- 8 x uint64 per queue item written by the workgroup that ran it: the
 * 100 MHz wall clock at [0] item taken, [1] dependencies complete, [2] tile computed and drained, [3] published;
 * [4] CU id | workgroup << 32; [5] stage | tile << 32; [6] 1 + split slice | last-arriver << 32 (scripts/gpu_prog_probe.py) */
PTX_EXPERIMENTAL_API int ptx_conv_program_trace_fwd(const ptx_conv_program_info* info, const void* image_dev, void* workspace, int32_t wgs_per_cu,
                               void* trace, size_t trace_bytes, ptx_stream_t stream);

/* The same operator over a caller-provided scratch buffer, which unlocks the STREAM-K form for long sequences whose
 * 64-query tiles do not fill the chip evenly (nonlocalnet.py:143-166 at N = 1568: 25 tiles x 8 clips = 200 workgroups on 256
 * CUs): the (query tile, key tile) units of one clip are cut into 32 (<= 32 query tiles) or 64 equal chunks, one workgroup
 * each; partial (O, max, sum) blocks meet in the workspace and a second launch folds them in chunk order.  The split is a
 * function of the per-sample extents only -- a clip's bits do not depend on the batch it arrives in.
 * ptx_nonlocal_workspace_bytes: bytes this descriptor's stream-K form needs (0: the plain kernels cover it).  With
 * workspace == NULL, too small or not 16-byte aligned, ptx_nonlocal_ws_fwd runs exactly what ptx_nonlocal_fwd runs. */
PTX_EXPERIMENTAL_API size_t ptx_nonlocal_workspace_bytes(const ptx_nonlocal_desc* desc);
PTX_EXPERIMENTAL_API int ptx_nonlocal_ws_fwd(const ptx_nonlocal_desc* desc, const float* theta, const float* phi, const float* g, float* y,
                        void* workspace, size_t workspace_bytes, ptx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PTX_AMD_H */

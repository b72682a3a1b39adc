// conv_igemm_kernel.h -- the implicit-GEMM convolution kernel template of libptx_amd (gfx950), shared by
// conv_igemm.hip (every tile configuration of ptx_conv3d_fwd & co.) and conv_chain.hip (the chained pointwise tail).
// See conv_igemm.hip for the design notes.
#pragma once
#include "ptx_common.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace ptx {

struct ConvArgs {
    const float* x;
    const float* w;
    const float* bias;
    const float* res;
    float* y;
    float* partial;
    unsigned* counters;    // PTX_SPLITK_FUSED: one arrival counter per output tile (zero between launches)
    int N, Ti, Hi, Wi, ldx, kA;
    int To, Ho, Wo, Co, ldy;
    int ncol;             // output columns written per row = round_up(Co, 4) <= ldy (ldy is the row stride:
                          // a conv may write a channel slice of a wider, concatenated tensor)
    int kT, kH, kW, sT, sH, sW, pT, pH, pW;
    int ldw, kB, w_rows, M;
    long long w_tap_stride;
    unsigned flags;
    int ldr, res_C, res_T, res_H, res_W, res_sT, res_sH, res_sW;
    int m_tiles, n_tiles, split_k, kchunks;
    long long bs_x, bs_w, bs_y;   // batched-GEMM strides (elements); 0 for a plain conv
    int unit_pointwise;   // 1x1x1 / stride 1 / pad 0: skip the position decode
    int k_live;           // live (possibly non-zero) K columns per tap = desc.Ci
    int prune_analytic;   // tap-plane pruning from the tile's row span in scalar arithmetic (no workgroup reduction)
    int tiles_per_plane;  // > 0: frame-fastest tile order (temporal L2 reuse), = Ho*Wo/BM
    // dual-source pointwise conv (ptx_conv3d_dual_fwd): K chunks [0, kc1) read x, chunks [kc1, kchunks) read x2
    const float* x2;
    int dual, ldx2, kA2, kc1, wcol2, T2, H2, W2, s2T, s2H, s2W;
    unsigned x2_bytes;
    int groups, cig, cog;  // grouped conv: input / output channels per group
    int f16;               // A / B operands are halfs; K extents count 32-bit words
    unsigned dv_hw[2];     // KWR tiles: fast division by the halo'd run length Wo + kW - 1
    int x3;                // fp32 A split into half (hi, lo) pairs on the fly, B packed as (hi8 | lo8) blocks: 3 f16 MFMAs
    unsigned x_bytes, w_bytes, y_bytes, r_bytes;   // extents of one batch item (buffer-resource bounds)
    // fused generator stage (ptx_conv3d_fused_fwd, fp16-operand tiles): a per-sample affine after bias (+ skip) -- the
    // NEXT layer's class-conditional BN folded to scale/shift tables -- halfs out, a second pre-affine output, a
    // half-precision skip operand, and an input read through a nearest 2x upsample in H and W
    const float* aff_scale;
    const float* aff_shift;
    int ld_aff, pps;       // table row stride; output positions per sample (sample of row m = m / pps)
    void* y_raw;           // second output: the pre-affine value, halfs, row stride ld_raw
    int ld_raw;
    unsigned raw_bytes, aff_bytes;
    int up2, Hp, Wp;       // up2: (Hi, Wi) are the UPSAMPLED extents the filter slides over, (Hp, Wp) the stored ones
    unsigned dv_wo[2], dv_ho[2], dv_to[2];   // fast division by Wo / Ho / To (mul, shift): the epilogue's row decode
    // chained pointwise tail (ptx_conv3d_chain_fwd, CHAIN tiles): y = epi2( relu?(conv(x, w) + bias) . w2^T + bias2 ).
    // The [BM x Co] result of THIS conv (Co <= BN: one N tile) never leaves the workgroup -- it is parked in LDS as the
    // A operand of a second, 1x1x1 GEMM over N2 = Co2 output channels whose filter tiles stream through the B stages.
    // In chain mode y / ldy / ncol-of-the-output / res / ldr / y_bytes / r_bytes describe the FINAL tensor.
    const float* w2;         // [w2_rows][ldw2]: K-major packed filter of the pointwise conv (K = this conv's Co)
    const float* bias2;
    int ldw2, kB2, w2_rows, Co2, ncol2;
    unsigned w2_bytes, flags2;   // flags2: PTX_EPI_RELU | PTX_EPI_RES_ADD of the tail
};

// Tap-plane pruning without the workgroup OR-reduction (three barriers in every workgroup's prologue): legal when every
// output coordinate of the T and H axes has at least one tap inside the image -- then the union of the rows' tap ranges over
// a tile's raster span follows from the span's two ends (conv_igemm_tile).  PTX_PRUNE_ANALYTIC=0: the reduction everywhere.
inline int prune_analytic_ok(const ConvArgs& a) {
    const char* e = getenv("PTX_PRUNE_ANALYTIC");       // (read per launch: the GPU test flips it inside one process)
    const bool on = !(e && atoi(e) == 0);
    auto axis_ok = [](int k, int pad, int s, int in, int out) {
        return pad >= 0 && pad <= k - 1 && s >= 1 && out >= 1 && in - 1 + pad - (out - 1) * s >= 0;
    };
    return (on && a.kT * a.kH > 1 && !a.up2 && axis_ok(a.kT, a.pT, a.sT, a.Ti, a.To) && axis_ok(a.kH, a.pH, a.sH, a.Hi, a.Ho)) ? 1 : 0;
}

// n / d for n < 2^31 without the ~30-instruction integer division sequence: d == 1 -> mul == 0; else
// l = ceil(log2 d), mul = ceil(2^(31+l) / d) (< 2^32), q = umulhi(n, mul) >> (l - 1).  Exact: the rounding error of
// mul adds less than 2^-l <= 1/d to n / d.
inline void fastdiv_make(unsigned d, unsigned (&out)[2]) {
    if (d <= 1) { out[0] = 0; out[1] = 0; return; }
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    out[0] = (unsigned)(((1ull << (31 + l)) + d - 1) / d);
    out[1] = l - 1;
}
template <class DV>      // unsigned[2] in any address space (kernel argument, or a constant-address-space stage table)
__device__ __forceinline__ unsigned fastdiv(unsigned n, const DV& dv) {
    return dv[0] ? (__umulhi(n, dv[0]) >> dv[1]) : n;
}

constexpr unsigned kFusedEpiFlags = PTX_EPI_OUT_F16 | PTX_EPI_AFFINE | PTX_EPI_DUAL_RAW | PTX_RES_F16 | PTX_EPI_TANH;
// ConvArgs::flags only (never in a descriptor): PTX_SKIP_EARLY=0 in the environment -- A/B switch of fused_stage_skip_prefetch
constexpr unsigned kNoSkipEarly = 0x40000000u;

// Step barrier.  hipcc may schedule LDS reads of the NEXT buffer above a plain __syncthreads() when it
// sees no aliasing store in this thread (observed on the LDS-DMA variant, whose only LDS writers are
// other waves' buffer_load...lds): pin the order for both the optimiser and the machine scheduler.
// The reads that follow take their base offsets through `post_barrier_offsets`, an asm volatile that
// is ordered after the barrier and that the reads depend on (cdna_hip_programming.md 5.7 item 3).
__device__ __forceinline__ void step_barrier() {
    __syncthreads();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void post_barrier_offsets(int& a, int& b) {
    asm volatile("; ds_reads of the next LDS buffer depend on these" : "+v"(a), "+v"(b)::"memory");
}

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
template <int MT> struct Mfma;
template <> struct Mfma<32> {
    using acc_t = f32x16;
    static constexpr int NACC = 16;
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    // fp16 operands: the same 16-byte fragment holds 8 halfs = K 16 per lane group pair, one instruction
    static __device__ __forceinline__ acc_t mma16(f32x4 a, f32x4 b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int r, int lane) {
        return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    }
};
template <> struct Mfma<16> {
    using acc_t = f32x4;
    static constexpr int NACC = 4;
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ acc_t mma16(f32x4 a, f32x4 b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int r, int lane) { return (lane >> 4) * 4 + r; }
};

// bias + residual + ReLU for one output element (shared with the split-K reduce kernel)
__device__ __forceinline__ float conv_epilogue(const ConvArgs& p, float v, int m, int co) {
    if (p.bias) v += p.bias[co];
    if (p.flags & PTX_EPI_RES_ADD) {
        v += p.res[(size_t)m * p.ldr + co];
    } else if (p.flags & PTX_EPI_RES_PADA) {
        if (co < ((p.flags & PTX_EPI_RES_UP) ? p.Co : p.res_C)) {
            const int wo = m % p.Wo;
            int t = m / p.Wo;
            const int ho = t % p.Ho;
            t /= p.Ho;
            const int to = t % p.To;
            const int n = t / p.To;
            const bool up = (p.flags & PTX_EPI_RES_UP) != 0;
            const int rt = up ? to >> p.res_sT : to * p.res_sT, rh = up ? ho >> p.res_sH : ho * p.res_sH,
                      rw = up ? wo >> p.res_sW : wo * p.res_sW;
            const size_t pos = (((size_t)n * p.res_T + rt) * p.res_H + rh) * p.res_W + rw;
            v += p.res[pos * p.ldr + co];
        }
    }
    if (p.flags & PTX_EPI_RELU) v = fmaxf(v, 0.f);
    return v;
}

// ------------------------------------------------------------------------------------------
// Fused generator-stage epilogue (fp16-operand tiles, ptx_conv3d_fused_fwd).  For one accumulator element:
//     v    = acc + bias[co] (+ skip)                  skip: same-shape add, or the nearest-upsampled, channel-truncated
//                                                     GBlock skip up(x[:, :Co]); fp32 or halfs (PTX_RES_F16)
//     raw  = v                                        -> y_raw as halfs (PTX_EPI_DUAL_RAW: the next block's skip operand)
//     v    = v * scale[n][co] + shift[n][co]          (PTX_EPI_AFFINE: the NEXT layer's class-conditional BN, folded)
//     v    = relu(v) | tanh(v)
//     y    = v as halfs (PTX_EPI_OUT_F16) or fp32
// so the cBN -> ReLU (-> upsample) passes between a generator block's convs never touch HBM: the producer applies
// the consumer's normalisation, and the consumer's loader does the upsampling (PTX_PRO_UP2).
// Half outputs are stored two columns per lane: neighbouring lanes hold neighbouring columns of the accumulator
// tile, so an xor-1 lane exchange turns two 2-byte stores into one 4-byte store.
// ------------------------------------------------------------------------------------------
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float fused_skip(const ConvArgs& p, int m, int co) {
    const bool r16 = (p.flags & PTX_RES_F16) != 0;
    size_t idx;
    if (p.flags & PTX_EPI_RES_ADD) {
        idx = (size_t)m * p.ldr + co;
    } else {
        if (co >= ((p.flags & PTX_EPI_RES_UP) ? p.Co : p.res_C)) return 0.f;
        unsigned t = fastdiv((unsigned)m, p.dv_wo);
        const int wo = m - (int)t * p.Wo;
        unsigned t2 = fastdiv(t, p.dv_ho);
        const int ho = (int)t - (int)t2 * p.Ho;
        const int n = (int)fastdiv(t2, p.dv_to);
        const int to = (int)t2 - n * p.To;
        const bool up = (p.flags & PTX_EPI_RES_UP) != 0;
        const int rt = up ? to >> p.res_sT : to * p.res_sT, rh = up ? ho >> p.res_sH : ho * p.res_sH,
                  rw = up ? wo >> p.res_sW : wo * p.res_sW;
        idx = ((((size_t)n * p.res_T + rt) * p.res_H + rh) * p.res_W + rw) * p.ldr + co;
    }
    return r16 ? (float)reinterpret_cast<const _Float16*>(p.res)[idx] : p.res[idx];
}

// bias + skip -> raw; affine + activation -> out   (shared with the split-K reduce kernel)
__device__ __forceinline__ void fused_value(const ConvArgs& p, float acc, int m, int co, float bias, float sc, float sh,
                                            float& raw, float& out) {
    float v = acc + bias;
    if (p.flags & (PTX_EPI_RES_ADD | PTX_EPI_RES_PADA)) v += fused_skip(p, m, co);
    raw = v;
    if (p.flags & PTX_EPI_AFFINE) v = fmaf(v, sc, sh);
    if (p.flags & PTX_EPI_RELU) v = fmaxf(v, 0.f);
    if (p.flags & PTX_EPI_TANH) v = tanhf(v);
    out = v;
}

// The epilogue runs row-major through LDS: a wave parks one MT-row block of its accumulator tile (fp32, [MT][WTN + 4])
// in its own slice of the (now idle) tile buffers, then every lane owns 8 consecutive channels of one output row:
// the skip operand, the scale / shift rows and both outputs move as 16-byte accesses over whole 64..256-byte row
// segments.  (Straight from the MFMA layout -- one column per lane -- the same work was 2-byte skip loads and
// 4-byte stores: config-5's 1x1 convs, which are all epilogue, ran at 1.6-2.1 TB/s.)
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));

// How many of a row block's passes get their skip operand early: 16 bytes per lane and pass stay live across the k-loop, and
// the 8-wave 128x128 tile has ten registers to spare below the 128 that keep two workgroups on a CU.
constexpr int kSkipEarlyPasses = 2;
// The skip operand of a fused generator stage, requested BEFORE the k-loop (half residuals only: one 16-byte load per lane
// and pass).  A block's last conv is a short-K pointwise GEMM (K = 64 .. 512 channels): issued in the epilogue, the skip's
// HBM round trip came after the operand tiles' one -- two dependent latencies per workgroup on a kernel that is nothing but
// memory traffic (2.6-3.0 TB/s on the upsampled skips); here both are in flight together.  Same (i, pass) order and the
// same address arithmetic as fused_stage_epilogue, which consumes `rq`.
template <class MF, int TM, int TN, int WTM, int WTN, int MT>
__device__ __forceinline__ void fused_stage_skip_prefetch(const ConvArgs& p, f32x4* rq /* [TM][NPE] */, int m0, int n0, int wm, int wn,
                                                          int lane) {
    constexpr unsigned kOOB = 0x80000000u;
    constexpr int LPR = WTN / 8, RPP = 64 / LPR, NP = (MT + RPP - 1) / RPP;
    constexpr int NPE = NP < kSkipEarlyPasses ? NP : kSkipEarlyPasses;
    const bool res_gather = (p.flags & PTX_EPI_RES_PADA) != 0, res_up = (p.flags & PTX_EPI_RES_UP) != 0;
    const int res_lim = res_gather ? (res_up ? p.Co : p.res_C) : p.ncol;
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res), 0, p.r_bytes, 0x00020000);
    const int co8 = n0 + wn * WTN + (lane % LPR) * 8;
    const bool c_lo = co8 < p.ncol;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mrow = m0 + wm * WTM + i * MT;
#pragma unroll
        for (int ps = 0; ps < NPE; ++ps) {
            const int row = ps * RPP + lane / LPR;
            const int m = mrow + row;
            const bool ok0 = row < MT && m < p.M && c_lo;
            unsigned pos = (unsigned)m;
            if (res_gather) {
                const unsigned t = fastdiv((unsigned)m, p.dv_wo);
                const int wo = m - (int)t * p.Wo;
                const unsigned t2 = fastdiv(t, p.dv_ho);
                const int ho = (int)t - (int)t2 * p.Ho;
                const int n = (int)fastdiv(t2, p.dv_to);
                const int to = (int)t2 - n * p.To;
                const int rt = res_up ? to >> p.res_sT : to * p.res_sT, rh = res_up ? ho >> p.res_sH : ho * p.res_sH,
                          rw = res_up ? wo >> p.res_sW : wo * p.res_sW;
                pos = (unsigned)(((n * p.res_T + rt) * p.res_H + rh) * p.res_W + rw);
            }
            const unsigned e = pos * (unsigned)p.ldr + (unsigned)co8;
            const bool q0 = ok0 && co8 < res_lim;
            rq[i * NPE + ps] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, q0 ? e * 2u : kOOB, 0, 0));
        }
    }
}

template <class MF, int TM, int TN, int WTM, int WTN, int MT>
__device__ __forceinline__ void fused_stage_epilogue(const ConvArgs& p, typename MF::acc_t (&acc)[TM][TN], int m0, int n0,
                                                     int wm, int wn, int lane, float* smem, int wave, const f32x4* rq, bool have_rq) {
    constexpr int NACC = MF::NACC;
    constexpr unsigned kOOB = 0x80000000u;
    constexpr int LDT = WTN + 4;                 // staged row stride (floats): 16-byte aligned rows, rows 4 banks apart
    constexpr int LPR = WTN / 8;                 // lanes per output row (8 channels each)
    constexpr int RPP = 64 / LPR;                // rows per pass
    constexpr int NP = (MT + RPP - 1) / RPP;
    static_assert(WTN % 8 == 0 && 64 % LPR == 0, "wave tile width");
    const bool out16 = (p.flags & PTX_EPI_OUT_F16) != 0, dual = (p.flags & PTX_EPI_DUAL_RAW) != 0;
    const bool affine = (p.flags & PTX_EPI_AFFINE) != 0;
    const bool res_same = (p.flags & PTX_EPI_RES_ADD) != 0, res_gather = (p.flags & PTX_EPI_RES_PADA) != 0;
    const bool res_up = (p.flags & PTX_EPI_RES_UP) != 0, r16 = (p.flags & PTX_RES_F16) != 0;
    const bool has_res = res_same || res_gather;
    const int res_lim = res_gather ? (res_up ? p.Co : p.res_C) : p.ncol;
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_raw =
        __builtin_amdgcn_make_buffer_rsrc(dual ? p.y_raw : (void*)p.y, 0, dual ? p.raw_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(has_res ? p.res : p.y), 0, has_res ? p.r_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_sc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(affine ? p.aff_scale : p.y), 0, affine ? p.aff_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_sh = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(affine ? p.aff_shift : p.y), 0, affine ? p.aff_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.bias ? p.bias : p.y), 0, p.bias ? (unsigned)p.w_rows * 4u : 0u, 0x00020000);
    auto ld4 = [&](const __amdgpu_buffer_rsrc_t rs, unsigned off) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    };
    float* Ls = smem + wave * (MT * LDT);
    const int cl = (lane % LPR) * 8;             // this lane's first column inside the wave tile (row-major phase)
    const int co8 = n0 + wn * WTN + cl;          // ... its first output channel
    const bool c_lo = co8 < p.ncol, c_hi = co8 + 4 < p.ncol;     // ncol is a multiple of 4: two 4-channel halves
    const f32x4 b0 = ld4(rs_b, c_lo ? (unsigned)co8 * 4u : kOOB), b1 = ld4(rs_b, c_hi ? (unsigned)(co8 + 4) * 4u : kOOB);
    __syncthreads();                             // every wave is done reading the operand tiles
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mrow = m0 + wm * WTM + i * MT;
        // ---- park the row block: MFMA layout (lane = column) -> LDS ----
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < NACC; ++r) Ls[MF::row(r, lane) * LDT + j * MT + (lane % MT)] = acc[i][j][r];
        // the block's rows usually belong to ONE sample: its scale / shift are then loaded once per block
        const int n_lo = (int)((unsigned)mrow / (unsigned)p.pps), n_hi = (int)((unsigned)min(mrow + MT - 1, p.M - 1) / (unsigned)p.pps);
        const bool one = n_lo == n_hi;
        f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sc1 = sc0, sh0 = {0.f, 0.f, 0.f, 0.f}, sh1 = sh0;
        if (affine && one && mrow < p.M) {
            const unsigned off = ((unsigned)n_lo * (unsigned)p.ld_aff + (unsigned)co8) * 4u;
            sc0 = ld4(rs_sc, c_lo ? off : kOOB); sc1 = ld4(rs_sc, c_hi ? off + 16u : kOOB);
            sh0 = ld4(rs_sh, c_lo ? off : kOOB); sh1 = ld4(rs_sh, c_hi ? off + 16u : kOOB);
        }
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const int row = ps * RPP + lane / LPR;
            const int m = mrow + row;
            const bool rok = row < MT && m < p.M;
            const bool ok0 = rok && c_lo, ok1 = rok && c_hi;
            // ---- skip operand: 8 channels of one (possibly upsampled-from) position ----
            f32x4 k0 = {0.f, 0.f, 0.f, 0.f}, k1 = k0;
            if (has_res) {
                unsigned pos = (unsigned)m;
                if (res_gather) {
                    const unsigned t = fastdiv((unsigned)m, p.dv_wo);
                    const int wo = m - (int)t * p.Wo;
                    const unsigned t2 = fastdiv(t, p.dv_ho);
                    const int ho = (int)t - (int)t2 * p.Ho;
                    const int n = (int)fastdiv(t2, p.dv_to);
                    const int to = (int)t2 - n * p.To;
                    const int rt = res_up ? to >> p.res_sT : to * p.res_sT, rh = res_up ? ho >> p.res_sH : ho * p.res_sH,
                              rw = res_up ? wo >> p.res_sW : wo * p.res_sW;
                    pos = (unsigned)(((n * p.res_T + rt) * p.res_H + rh) * p.res_W + rw);
                }
                const unsigned e = pos * (unsigned)p.ldr + (unsigned)co8;
                const bool q0 = ok0 && co8 < res_lim, q1 = ok1 && co8 + 4 < res_lim;
                if (r16) {               // 8 halfs = one 16-byte load (channels beyond res_lim inside it are masked below)
                    const half8_t h = (have_rq && ps < kSkipEarlyPasses)
                                          ? __builtin_bit_cast(half8_t, rq[i * (NP < kSkipEarlyPasses ? NP : kSkipEarlyPasses) + ps])   // requested before the k-loop
                                              : __builtin_bit_cast(half8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_res, q0 ? e * 2u : kOOB, 0, 0));
                    k0 = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
                    k1 = q1 ? f32x4{(float)h[4], (float)h[5], (float)h[6], (float)h[7]} : f32x4{0.f, 0.f, 0.f, 0.f};
                } else {
                    k0 = ld4(rs_res, q0 ? e * 4u : kOOB);
                    k1 = ld4(rs_res, q1 ? e * 4u + 16u : kOOB);
                }
            }
            f32x4 s0 = sc0, s1 = sc1, t0 = sh0, t1 = sh1;
            if (affine && !one) {
                const unsigned n = (unsigned)m / (unsigned)p.pps;
                const unsigned off = (n * (unsigned)p.ld_aff + (unsigned)co8) * 4u;
                s0 = ld4(rs_sc, ok0 ? off : kOOB); s1 = ld4(rs_sc, ok1 ? off + 16u : kOOB);
                t0 = ld4(rs_sh, ok0 ? off : kOOB); t1 = ld4(rs_sh, ok1 ? off + 16u : kOOB);
            }
            const float* lrow = Ls + (row < MT ? row : 0) * LDT + cl;
            f32x4 v0 = *reinterpret_cast<const f32x4*>(lrow) + b0 + k0;
            f32x4 v1 = *reinterpret_cast<const f32x4*>(lrow + 4) + b1 + k1;
            const f32x4 r0 = v0, r1 = v1;
            if (affine) { v0 = v0 * s0 + t0; v1 = v1 * s1 + t1; }
            if (p.flags & PTX_EPI_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v0[e] = fmaxf(v0[e], 0.f); v1[e] = fmaxf(v1[e], 0.f); }
            }
            if (p.flags & PTX_EPI_TANH) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v0[e] = tanhf(v0[e]); v1[e] = tanhf(v1[e]); }
            }
            if (out16) {
                const half8_t h = {(_Float16)v0[0], (_Float16)v0[1], (_Float16)v0[2], (_Float16)v0[3],
                                   (_Float16)v1[0], (_Float16)v1[1], (_Float16)v1[2], (_Float16)v1[3]};
                // ldy % 8 == 0 halfs and co8 % 8 == 0: the 16-byte store stays inside the row (pad columns get
                // the affine of zero -- finite, and multiplied by zero filter columns downstream)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, h), rs_y,
                                                       ok0 ? ((unsigned)m * (unsigned)p.ldy + (unsigned)co8) * 2u : kOOB, 0, 0);
            } else {
                const unsigned off = ((unsigned)m * (unsigned)p.ldy + (unsigned)co8) * 4u;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v0), rs_y, ok0 ? off : kOOB, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v1), rs_y, ok1 ? off + 16u : kOOB, 0, 0);
            }
            if (dual) {
                const half8_t h = {(_Float16)r0[0], (_Float16)r0[1], (_Float16)r0[2], (_Float16)r0[3],
                                   (_Float16)r1[0], (_Float16)r1[1], (_Float16)r1[2], (_Float16)r1[3]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, h), rs_raw,
                                                       ok0 ? ((unsigned)m * (unsigned)p.ld_raw + (unsigned)co8) * 2u : kOOB, 0, 0);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Row-major epilogue for the fp32 tiles (REPI): bias + residual + ReLU with 16-byte accesses.
// Straight from the MFMA layout a lane owns ONE column of 16 rows: the residual arrives as 16 four-byte loads and the
// output leaves as 16 four-byte stores per accumulator tile -- the HBM-bound pointwise convs (layer1's 1x1x1 convs sit on
// the fp32 ridge) ran at 4.3 TB/s on them.  Here a wave parks one MT-row block of its accumulators in
// its own LDS slice ([MT][WTN + 4] floats), then every lane owns 4 consecutive channels of one row: residual, bias and
// output move as 128-byte row segments, a quarter of the memory instructions.
// `res4`: the residual values of the tile, requested by the caller in the SAME (i, pass) order before the accumulators are
// final, so their latency hides under the k-loop's tail.
// ------------------------------------------------------------------------------------------
template <int WTN, int MT>
struct RowEpi {
    static constexpr int LDT = WTN + 4;          // staged row stride (floats): 16-byte aligned rows
    static constexpr int LPR = WTN / 4;          // lanes per output row (4 channels each)
    static constexpr int RPP = 64 / LPR;         // rows per pass
    static constexpr int NP = (MT + RPP - 1) / RPP;
    static constexpr int FLOATS = MT * LDT;      // LDS floats per wave
    static_assert(WTN % 4 == 0 && 64 % LPR == 0 && LPR <= 64, "wave tile width");
};

template <class MF, int TM, int TN, int WTM, int WTN, int MT, int COH = 0>
__device__ __forceinline__ void rowmajor_load_residual(f32x4* res4 /* [TM][NP] */, const __amdgpu_buffer_rsrc_t rs_r,
                                                       bool has_res, int mbase, int co_base, int M, int ncol, int ldr, int lane) {
    using RE = RowEpi<WTN, MT>;
    constexpr unsigned kOOB = 0x80000000u;
    const int co4 = co_base + (lane % RE::LPR) * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int ps = 0; ps < RE::NP; ++ps) {
            const int row = ps * RE::RPP + lane / RE::LPR;
            const int m = mbase + i * MT + row;
            const bool ok = has_res && row < MT && m < M && co4 < ncol;
            const unsigned off = ((unsigned)m * (unsigned)ldr + (unsigned)co4) * 4u;
            res4[i * RE::NP + ps] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_r, ok ? off : kOOB, 0, COH));
        }
}

template <class MF, int TM, int TN, int WTM, int WTN, int MT, int COH = 0>
__device__ __forceinline__ void rowmajor_store_tile(typename MF::acc_t (&acc)[TM][TN], const f32x4* res4 /* [TM][NP] */,
                                                    float* Ls, const float* bias, const __amdgpu_buffer_rsrc_t rs_y, bool relu,
                                                    int mbase, int co_base, int M, int ncol, int ldy, int lane) {
    using RE = RowEpi<WTN, MT>;
    constexpr unsigned kOOB = 0x80000000u;
    const int cl = (lane % RE::LPR) * 4;
    const int co4 = co_base + cl;
    const bool c_ok = co4 < ncol;                 // ncol is a multiple of 4
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (bias && c_ok) b4 = *reinterpret_cast<const f32x4*>(bias + co4);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        // park the row block: MFMA layout (lane = column) -> LDS.  The slice is private to the wave and a wave's LDS
        // operations execute in order, so no barrier separates the stores from the loads below (or from the next block's)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < MF::NACC; ++r) Ls[MF::row(r, lane) * RE::LDT + j * MT + (lane % MT)] = acc[i][j][r];
#pragma unroll
        for (int ps = 0; ps < RE::NP; ++ps) {
            const int row = ps * RE::RPP + lane / RE::LPR;
            const int m = mbase + i * MT + row;
            const bool ok = c_ok && row < MT && m < M;
            f32x4 v = *reinterpret_cast<const f32x4*>(Ls + (row < MT ? row : 0) * RE::LDT + cl) + b4 + res4[i * RE::NP + ps];
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            const unsigned off = ((unsigned)m * (unsigned)ldy + (unsigned)co4) * 4u;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rs_y,
                                                   ok ? off : kOOB, 0, COH);
        }
    }
}

// F16: the A / B operands are IEEE halfs.  Everything that MOVES data (buffer loads, LDS-DMA, swizzle, tap
// pruning, K tails) works on 32-bit words and does not care; the descriptor then counts channel PAIRS.  Only
// the fragment -> MFMA step differs: the 16-byte fragment a lane reads is 8 halfs, consumed by ONE
// v_mfma_f32_32x32x16_f16 / 16x16x32_f16 instead of four fp32 MFMAs.  Accumulators, epilogue and output stay fp32.
// X3 ("split" operands, PTX_F16X3_OPERANDS): fp32-accurate products on the fp16 matrix cores.  An fp32 value v is the
// exact sum of two halfs up to 2^-22 |v|: hi = half(v), lo = half(v - hi).  a.b = a_hi.b_hi + a_hi.b_lo + a_lo.b_hi
// (+ a_lo.b_lo, <= 2^-22 |a.b|, dropped): every product of two halfs is exact in the fp32 accumulator, so the result
// carries ~22 mantissa bits per product -- the same class as the fp32 fma chain -- at 3 x 32x32x16_f16 MFMAs per 16 k
// (96 clk) instead of 8 x 32x32x2_f32 (512 clk).  The activations stay fp32 in HBM and in the LDS tile (so every other
// kernel, the residual and the epilogue are untouched); a lane splits its 8-channel A fragment in registers (24 VALU
// per 3 x TN MFMAs).  The filter is split once, at pack time (ptx_pack_desc.f16 == 2: each 8-channel block of a row is
// stored as 8 hi halfs then 8 lo halfs -- the same 32 bytes), so B fragments are two 16-byte reads, no VALU.
// SCALED lo: lo is stored as half((v - hi) * 2^12).  Unscaled it is ~2^-11 |v| and falls into the subnormal halfs as soon
// as |v| < 2^-3 (measured: 3e-5 relative conv error at |x| ~ 1e-3, 2e-4 at 1e-4); scaled it is a normal half whenever hi
// is one.  The two cross terms then carry a factor 2^12 and accumulate in their own accumulator, folded back once after
// the k-loop (acc += 2^-12 acc2; powers of two: exact).  22 bits per product for every operand in the normal half range
// (6.1e-5 <= |v| < 65504), at the price of a second accumulator tile.
// KWR ("kw reuse", KWR = 3): for stride-1 filters of width 3 whose M tile is a whole number of output rows, the A tile
// staged per (kt, kh, channel chunk) is the HALO'D input run -- Wo + 2 positions per output row -- and the three kw taps
// read their fragments from it at row offsets +0 / +1 / +2, each against its own B (filter) tile of the same stage.  A
// moves through L2 -> LDS once instead of three times: with the matrix work per k-step 5x (x3) to 16x (f16) shorter than
// on the fp32 cores, that traffic (8-11 TB/s sustained by the LDS-DMA path) is what bounds these convs.
// CHAIN: the conv's output tile feeds a second, pointwise conv inside the same workgroup (see ConvArgs::w2): conv ->
// BN -> ReLU -> 1x1x1 conv -> BN (-> + residual) -> ReLU in ONE launch.  Serves a bottleneck's tail (3x3x3 conv2 ->
// conv3 + residual, resnet3D.py:129-142) and the pointwise pairs of the (2+1)D nets (a "1x1x1" SpatioTemporalConv is two
// pointwise GEMMs through M mid channels, r2plus1d.py:68-88): the intermediate tensor's HBM round trip, one launch and the
// second conv's per-tile prologue disappear, and the HBM-bound tail (residual read + 4x wider write) overlaps the
// MFMA-bound body of the co-resident workgroups.  fp32 tiles with 2-stage LDS-DMA only.
// Phase clock of a workgroup (diagnostic build only: scripts/micro/build_timeline.sh compiles conv_igemm.hip with
// -DPTX_IGEMM_TIMELINE into a SEPARATE library; the product library carries none of it).  Thread 0 writes the 100 MHz wall clock
// at: 0 entry, 1 operand tables built, 2 first tile landed, 3 last k-step done, 4 row-major epilogue past its wait, 5 stores
// retired; 6 = __smid().
// Last arriver of a split-K tile: y = epilogue(sum of the tile's partial slabs in SPLIT ORDER -- the order of
// splitk_reduce_kernel, so fused and separate reductions are bit-identical).  The slabs were written write-through (sc1) by
// other workgroups of this launch, possibly on other XCDs: read them (and a residual another stage of a conv program may
// just have produced) with sc1 loads -- no fence (cdna_hip_programming.md Guideline 16, R1).
template <class Args>
__device__ __forceinline__ void splitk_reduce_tile(const Args& p, int BM, int BN, int tile, int tid, int nthreads) {
    constexpr unsigned kOOB = 0x80000000u;
    constexpr int kSc1 = 16;
    const int m0 = (tile / p.n_tiles) * BM, n0 = (tile % p.n_tiles) * BN;
    const size_t slab = (size_t)p.M * p.ncol;
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (unsigned)((size_t)p.M * p.ldy * 4), 0x00020000);
    const bool has_res = (p.flags & PTX_EPI_RES_ADD) != 0;
    const __amdgpu_buffer_rsrc_t rs_r =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res), 0, has_res ? p.r_bytes : 0u, 0x00020000);
    const bool relu = (p.flags & PTX_EPI_RELU) != 0;
#pragma unroll 1
    for (int e = tid * 4; e < BM * BN; e += nthreads * 4) {
        const int m = m0 + e / BN, co = n0 + e % BN;
        const bool ok = co < p.ncol && m < p.M;
        const unsigned poff = ok ? (unsigned)(((size_t)m * p.ncol + co) * 4) : kOOB;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int z = 0; z < p.split_k; ++z) {
            const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(p.partial + (size_t)z * slab, 0, p.y_bytes, 0x00020000);
            const f32x4 u = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_p, poff, 0, kSc1));
            v = z == 0 ? u : v + u;
        }
        if (p.bias && ok) v += *reinterpret_cast<const f32x4*>(p.bias + co);
        const f32x4 r = __builtin_bit_cast(
            f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_r, (ok && has_res) ? (unsigned)(((size_t)m * p.ldr + co) * 4) : kOOB, 0, kSc1));
        v += r;
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rs_y,
                                               ok ? (unsigned)(((size_t)m * p.ldy + co) * 4) : kOOB, 0, kSc1);
    }
}

#ifdef PTX_IGEMM_TIMELINE
static __device__ unsigned long long* g_ig_tl = nullptr;
#define PTX_IG_TL(k)                                                                                                   \
    do {                                                                                                               \
        if (threadIdx.x == 0 && g_ig_tl)                                                                               \
            g_ig_tl[(size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + (k)] =             \
                (k) == 6 ? (unsigned long long)__smid() : (unsigned long long)wall_clock64();                          \
    } while (0)
#define PTX_IG_TL_END()                                                                                                \
    do {                                                                                                               \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                               \
        PTX_IG_TL(5);                                                                                                  \
    } while (0)
#else
#define PTX_IG_TL(k) do {} while (0)
#define PTX_IG_TL_END() do {} while (0)
#endif

// The body of one workgroup's tile.  `tile` = m_tile * n_tiles + n_tile, `zb` the batched-GEMM item, `zs` the split-K slice,
// `smem` the workgroup's dynamic LDS.  conv_igemm_kernel (one tile per workgroup, below) and conv_program_kernel
// (conv_program.hip: a persistent workgroup walking a queue of tiles of SEVERAL convs in one launch) both run it.
// COH: cache-policy bits (`aux`) of every access to data ANOTHER workgroup of the same launch may have produced or will
// consume -- activation / residual loads, output and split-K partial stores.  0 for a plain launch; 16 (sc1: L1 bypass on
// loads, write-through on stores) inside a conv program, whose stages hand tensors to each other without a kernel boundary
// (cdna_hip_programming.md Guideline 16: sc1 stores + drained flag on the producer, sc1 loads on the consumer).
// Filters and biases are read-only for the whole launch and keep the default policy.
// Args: ConvArgs as a kernel argument (plain launch), or the same struct read through a constant-address-space reference
// (conv program: the stage table in global memory is never written during the launch, so its fields are scalar loads the
// compiler may repeat instead of keeping ~120 SGPRs alive across the k-loop -- a by-value copy spilled 269 of them).
template <int BM, int BN, int BK, int WM, int WN, int MT, bool KTAIL, bool K22, bool DMA, int NSTAGE, bool F16 = false,
          bool X3 = false, int KWR = 0, bool CHAIN = false, bool REPI = false, int COH = 0, class Args = ConvArgs>
__device__ __forceinline__ void conv_igemm_tile(const Args& p, const int tile, const int zb, const int zs, float* smem) {
    PTX_IG_TL(0);
    PTX_IG_TL(6);
    static_assert(!CHAIN || (DMA && NSTAGE == 2 && !F16 && !K22), "chained tail: fp32 / split-operand 2-stage LDS-DMA tiles");
    static_assert(!(CHAIN && KWR && REPI), "kw-reuse chained tiles keep the column-wise tail epilogue");
    static_assert(!CHAIN || BN >= BK, "chained tail: the parked tile is cut into BN / BK k-chunks");
    static_assert(!REPI || (DMA && !F16 && !K22 && KWR == 0), "row-major epilogue: fp32-output LDS-DMA tiles (fp32 or split operands)");
    static_assert(KWR == 0 || (KWR == 3 && DMA && NSTAGE == 2 && !K22), "kw-reuse tiles: 3-wide filters, 2-stage LDS-DMA");
    static_assert(!F16 || !K22, "the K22 stem path is fp32 only");
    static_assert(!X3 || (DMA && !F16 && !K22), "split operands: LDS-DMA tiles");
    using MF = Mfma<MT>;
    using acc_t = typename MF::acc_t;
    constexpr int NT = 64 * WM * WN;         // threads per workgroup (4 or 8 waves)
    static_assert(NT >= 256 && NT <= 512, "4 to 8 waves per workgroup");
    // Register-staged tiles are padded by 4 floats per row (conflict-free ds_read_b128).  DMA tiles
    // (buffer_load ... lds) must be lane-linear, i.e. unpadded: the bank-conflict fix moves into an
    // XOR swizzle of the 16-byte slot index that is applied to the per-lane SOURCE address when
    // loading and to the fragment read address (cdna_hip_programming.md rule 21).
    constexpr int LDK = DMA ? BK : BK + 4;
    static_assert(!DMA || (BK == 64 || BK == 32 || BK == 16), "DMA staging: BK 64, 32 or 16");
    static_assert(!(DMA && K22) || (BK == 32 && NSTAGE == 2), "the LDS-DMA K22 stem tile stages 32-float rows, 2 buffers");
    static_assert(NSTAGE == 2 || (NSTAGE >= 3 && NSTAGE <= 6 && DMA), "deeper rings need DMA staging");
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / MT, TN = WTN / MT;
    static_assert(TM * MT * WM == BM && TN * MT * WN == BN, "tile must split evenly");
    constexpr int KG = 64 / MT;              // lane groups along k
    constexpr int KPL = X3 ? 8 : 4;          // 32-bit words of K a lane consumes per sub-step
    constexpr int KSUB = BK / (KPL * KG);    // sub-steps per k-step
    static_assert(KSUB * KPL * KG == BK, "BK must be a multiple of KPL*KG");
    constexpr int F4R = BK / 4;              // float4 per tile row
    constexpr int KW_T = KWR ? KWR : 1;      // kw taps served by one staged A tile
    // LDS rows of the A image: BM/Wo runs of Wo + 2 positions (Wo >= 8), rounded so the image is whole 1-KiB DMA pieces
    constexpr int AR = KWR ? (BM + BM / 4 + 15) / 16 * 16 : BM;
    constexpr int ASTG = AR * (DMA ? BK : BK + 4), BSTG = KW_T * BN * (DMA ? BK : BK + 4);   // floats per stage
    constexpr int A_F4 = AR * F4R, B_F4 = BN * F4R;
    constexpr int A_IT = (A_F4 + NT - 1) / NT, B_IT = (B_F4 + NT - 1) / NT;
    // swizzle: physical 16-B slot = logical slot ^ ((row >> SWS) & (F4R - 1)); with 256-B rows (F4R 16)
    // SWS = 0, with 128-B rows (F4R 8) SWS = 1, with 64-B rows (F4R 4) SWS = 2 -- any 16 distinct
    // rows of a ds_read_b128 lane group then cover all 16 slots of the 256-B bank row.
    constexpr int SWS = (F4R == 16) ? 0 : (F4R == 8) ? 1 : 2;
    auto swz_col = [&](int idx) -> int {       // logical channel column fetched by staging slot idx
        const int row = idx / F4R, ps = idx % F4R;
        return (DMA ? (ps ^ ((row >> SWS) & (F4R - 1))) : ps) * 4;
    };

    float* As = smem;                        // [NSTAGE][AR][LDK]
    float* Bs = smem + NSTAGE * ASTG;        // [NSTAGE][KW_T][BN][LDK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    const int n_tile = tile % p.n_tiles;
    int m_tile = tile / p.n_tiles;
    if (p.tiles_per_plane > 0) {
        // temporal filters: visit the frames of a spatial band back to back (frame index fastest) so
        // the kT-frame input window of the band stays in the XCD's L2 instead of being re-fetched
        // once per frame; only the dispatch ORDER changes, a tile still covers BM consecutive rows
        const int clip_tiles = p.To * p.tiles_per_plane;
        const int c = m_tile / clip_tiles, r = m_tile - c * clip_tiles;
        m_tile = c * clip_tiles + (r % p.To) * p.tiles_per_plane + r / p.To;
    }
    const int m0 = m_tile * BM, n0 = n_tile * BN;

    // grouped conv: this N tile lies inside one group and reads only that group's input columns
    const float* __restrict__ xg = p.x + (size_t)zb * p.bs_x + (p.groups > 1 ? (n0 / p.cog) * p.cig : 0);
    const float* __restrict__ wg = p.w + (size_t)zb * p.bs_w;

    // ---- per-thread operand rows (tap independent, computed once) ----
    // fp32 MFMA shares the FP32 datapath with VALU, so VALU work in the k-loop costs MFMA cycles:
    // everything per-lane is hoisted here.  For each A row this thread stages:
    //   a_off  = byte offset of the row's CENTRE tap (kt,kh,kw) = (pT,pH,pW) -- always inside the
    //            image for a valid row -- plus this thread's channel column; kOOB for rows >= M;
    //   a_mask = separable validity bitmasks: bit kt | bit 8+kh | bit 16+kw set iff that tap
    //            coordinate lands inside the image.
    // A k-step then needs 4 VALU per load: and, cmp (mask test), add (uniform tap offset), cndmask.
    constexpr unsigned kOOB = 0x80000000u;
    unsigned a_off[A_IT], a_mask[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int idx = tid + NT * i;
        const int row = idx / F4R;
        const int col = swz_col(idx);
        if constexpr (KWR) {
            // LDS row -> (output row-run r, position j of its halo'd run): input column j - pW of that run's image row
            const int HW = p.Wo + KWR - 1;
            const unsigned r = fastdiv((unsigned)row, p.dv_hw);
            const int j = row - (int)r * HW;
            const int m = m0 + (int)r * p.Wo;                       // first output position of the run
            const int wcol = j - p.pW;
            const bool ok = (idx < A_F4) && ((int)r * p.Wo < BM) && (m < p.M) && ((unsigned)wcol < (unsigned)p.Wi);
            const unsigned mm = (m < p.M) ? (unsigned)m : 0u;
            const unsigned q1 = fastdiv(mm, p.dv_wo);
            const unsigned q2 = fastdiv(q1, p.dv_ho);
            const int ho = (int)(q1 - q2 * (unsigned)p.Ho);
            const int n = (int)fastdiv(q2, p.dv_to);
            const int to = (int)q2 - n * p.To;
            const int tc = to * p.sT, hc = ho * p.sH;
            auto tap_range = [](int c, int pad, int k, int extent) -> unsigned {
                const int lo = max(0, pad - c), hi = min(k - 1, extent - 1 + pad - c);
                return hi >= lo ? (((2u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u;
            };
            unsigned mask = tap_range(tc, p.pT, p.kT, p.Ti) | (tap_range(hc, p.pH, p.kH, p.Hi) << 8) | (0xFFu << 16);
            unsigned cpos = (unsigned)(((n * p.Ti + tc) * p.Hi + hc) * p.Wi + wcol);
            if constexpr (F16) {
                if (p.up2) {        // upsampling loader: wcol / hc are upsampled coordinates, the tensor stores half of them
                    cpos = (unsigned)(((n * p.Ti + tc) * p.Hp + (hc >> 1)) * p.Wp + (wcol >> 1));
                    mask |= (unsigned)(hc & 1) << 24;
                }
            }
            a_off[i] = ok ? (cpos * (unsigned)p.ldx + (unsigned)col) * 4u : kOOB;
            a_mask[i] = ok ? mask : 0u;
            continue;
        }
        const int m = m0 + row;
        const bool ok = (idx < A_F4) && (m < p.M);
        if (p.unit_pointwise) {      // 1x1x1, stride 1, no padding: input position == output position
            a_off[i] = ok ? ((unsigned)m * (unsigned)p.ldx + (unsigned)col) * 4u : kOOB;
            a_mask[i] = ok ? 0x00010101u : 0u;
            continue;
        }
        // row decode with multiply-shift divisions (a plain `/` is a ~35-instruction sequence: at 8 rows per
        // thread the decode used to cost more VALU than a short-K conv's whole k-loop)
        const unsigned mm = ok ? (unsigned)m : 0u;
        const unsigned q1 = fastdiv(mm, p.dv_wo);
        const int wo = (int)(mm - q1 * (unsigned)p.Wo);
        const unsigned q2 = fastdiv(q1, p.dv_ho);
        const int ho = (int)(q1 - q2 * (unsigned)p.Ho);
        const int n = (int)fastdiv(q2, p.dv_to);
        const int to = (int)q2 - n * p.To;
        const int tc = to * p.sT, hc = ho * p.sH, wc = wo * p.sW;      // centre-tap input coordinate
        // taps k in [lo, hi] of an axis land inside the image: closed form instead of a loop over the filter extent
        auto tap_range = [](int c, int pad, int k, int extent) -> unsigned {
            const int lo = max(0, pad - c), hi = min(k - 1, extent - 1 + pad - c);
            return hi >= lo ? (((2u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u;
        };
        const unsigned mask = tap_range(tc, p.pT, p.kT, p.Ti) | (tap_range(hc, p.pH, p.kH, p.Hi) << 8) |
                              (tap_range(wc, p.pW, p.kW, p.Wi) << 16);
        unsigned cpos = (unsigned)(((n * p.Ti + tc) * p.Hi + hc) * p.Wi + wc);
        unsigned par = 0;
        if constexpr (F16) {
            if (p.up2) {     // stored position of the centre tap; tap offsets then depend on the parity of (hc, wc)
                cpos = (unsigned)(((n * p.Ti + tc) * p.Hp + (hc >> 1)) * p.Wp + (wc >> 1));
                par = ((unsigned)(hc & 1) << 24) | ((unsigned)(wc & 1) << 25);
            }
        }
        a_off[i] = ok ? (cpos * (unsigned)p.ldx + (unsigned)col) * 4u : kOOB;
        a_mask[i] = ok ? (mask | par) : 0u;
    }
    unsigned a_off2[A_IT];      // second activation source (strided gather), dual-source convs only
#pragma unroll
    for (int i = 0; i < A_IT; ++i) a_off2[i] = kOOB;
    if (p.dual) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int idx = tid + NT * i;
            const int m = m0 + idx / F4R;
            if ((idx < A_F4) && (m < p.M)) {
                const unsigned q1 = fastdiv((unsigned)m, p.dv_wo);
                const int wo = (int)((unsigned)m - q1 * (unsigned)p.Wo);
                const unsigned q2 = fastdiv(q1, p.dv_ho);
                const int ho = (int)(q1 - q2 * (unsigned)p.Ho);
                const int n = (int)fastdiv(q2, p.dv_to);
                const int to = (int)q2 - n * p.To;
                const unsigned pos2 = (unsigned)(((n * p.T2 + to * p.s2T) * p.H2 + ho * p.s2H) * p.W2 + wo * p.s2W);
                a_off2[i] = (pos2 * (unsigned)p.ldx2 + (unsigned)swz_col(idx)) * 4u;
            }
        }
    }
    unsigned b_off[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int idx = tid + NT * i;
        const int row = idx / F4R;
        const int col = swz_col(idx);
        const bool ok = idx < B_F4 && (n0 + row) < p.w_rows;
        b_off[i] = ok ? ((unsigned)(n0 + row) * (unsigned)p.ldw + (unsigned)col) * 4u : kOOB;
    }

    // ---- block-uniform tap pruning: a (kt, kh) tap plane that lands in the zero padding for EVERY
    // row of this tile is skipped.  Exact for any tile (also when it straddles frames or clips): the
    // per-row validity bitmasks are OR-reduced over the workgroup (wave shuffles + one LDS word). ----
    int kt_lo = 0, kt_hi = p.kT - 1, kh_lo = 0, kh_hi = p.kH - 1;
    if (!KWR && p.prune_analytic) {
        // The same ranges WITHOUT a workgroup reduction (round 6): the tile's rows m0 .. m1 are a contiguous raster span, so
        // the output frames / output rows they touch are one interval (or, when the span wraps a clip / frame boundary or is
        // longer than one, everything), and a tap axis' valid range is monotone in the output coordinate: lo from the largest
        // coordinate, hi from the smallest.  Scalar arithmetic, identical in every wave; the host sets `prune_analytic` only
        // when every output coordinate has a non-empty tap range on both axes (then this equals the OR of the row masks).
        const unsigned um0 = (unsigned)m0, um1 = (unsigned)(min(m0 + BM, p.M) - 1);
        const unsigned r0 = fastdiv(um0, p.dv_wo), r1 = fastdiv(um1, p.dv_wo);          // (frame, row) index of both ends
        const unsigned f0 = fastdiv(r0, p.dv_ho), f1 = fastdiv(r1, p.dv_ho);            // frame index n * To + to
        const int h0 = (int)(r0 - f0 * (unsigned)p.Ho), h1 = (int)(r1 - f1 * (unsigned)p.Ho);
        const bool h_one = f0 == f1;                                                    // (then h0 <= h1)
        const int h_min = h_one ? h0 : 0, h_max = h_one ? h1 : p.Ho - 1;
        const unsigned n0_ = fastdiv(f0, p.dv_to), n1_ = fastdiv(f1, p.dv_to);
        const int t0 = (int)(f0 - n0_ * (unsigned)p.To), t1 = (int)(f1 - n1_ * (unsigned)p.To);
        const bool t_one = n0_ == n1_;
        const int t_min = t_one ? t0 : 0, t_max = t_one ? t1 : p.To - 1;
        kt_lo = max(0, p.pT - t_max * p.sT);
        kt_hi = min(p.kT - 1, p.Ti - 1 + p.pT - t_min * p.sT);
        kh_lo = max(0, p.pH - h_max * p.sH);
        kh_hi = min(p.kH - 1, p.Hi - 1 + p.pH - h_min * p.sH);
    } else if (p.kT * p.kH > 1) {
        unsigned m_or = 0;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) m_or |= a_mask[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m_or |= (unsigned)__shfl_xor((int)m_or, o, 64);
        unsigned* scratch = reinterpret_cast<unsigned*>(smem);     // tile buffers are not live yet
        if (tid == 0) scratch[0] = 0u;
        __syncthreads();
        if (lane == 0) atomicOr(&scratch[0], m_or);
        __syncthreads();
        m_or = scratch[0];
        __syncthreads();                                            // before the first tile lands here
        const unsigned mt = m_or & 0xFFu, mh = (m_or >> 8) & 0xFFu;
        kt_lo = mt ? __builtin_ctz(mt) : 1;
        kt_hi = mt ? 31 - __builtin_clz(mt) : 0;
        kh_lo = mh ? __builtin_ctz(mh) : 1;
        kh_hi = mh ? 31 - __builtin_clz(mh) : 0;
    }
    // The block iterates the PRUNED k-space {kt_lo..kt_hi} x {kh_lo..kh_hi} x kW x kchunks, tap-major,
    // channel-chunk-minor; split-K slices that space evenly.  (kt, kh, kw, ch) is the next k-step to
    // LOAD and is advanced with a few scalar compares -- no divisions, no data-dependent branches.
    const int nkt = max(kt_hi - kt_lo + 1, 0), nkh = max(kh_hi - kh_lo + 1, 0);
    const int kw_ext = KWR ? 1 : p.kW;            // KWR: one staged A tile serves all kw taps
    const int total_steps = nkt * nkh * kw_ext * p.kchunks;
    const int per_split = (total_steps + p.split_k - 1) / p.split_k;
    const int s_begin = min(zs * per_split, total_steps);
    const int my_steps = min(s_begin + per_split, total_steps) - s_begin;
    int kt, kh, kw, ch;
    {
        ch = s_begin % p.kchunks;
        int t = s_begin / p.kchunks;
        kw = t % kw_ext;
        t /= kw_ext;
        const int d = max(nkh, 1);
        kh = kh_lo + t % d;
        kt = kt_lo + t / d;
    }
    auto advance = [&]() {
        ++ch;
        const bool c1 = ch == p.kchunks;
        ch = c1 ? 0 : ch;
        kw += c1 ? 1 : 0;
        const bool c2 = kw == kw_ext;
        kw = c2 ? 0 : kw;
        kh += c2 ? 1 : 0;
        const bool c3 = kh > kh_hi;
        kh = c3 ? kh_lo : kh;
        kt += c3 ? 1 : 0;
    };

    f32x4 ra[DMA ? 1 : A_IT], rb[DMA ? 1 : B_IT];
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    // Branch-free operand loads through buffer resources: an element that must read as zero
    // (tap outside the image, row >= M, channel tail) gets a byte offset >= kOOB >= num_records,
    // for which the hardware returns 0 without touching memory.  (Extents are validated < 2 GiB.)
    const __amdgpu_buffer_rsrc_t rsrc_x =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xg), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wg), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x2 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dual ? p.x2 : xg), 0, p.dual ? p.x2_bytes : 0u, 0x00020000);

    // issue the global loads of k-step (kt, kh, kw, ch) into registers; `valid` == false turns
    // every load into an OOB (zero, no traffic) access
    auto load_tiles = [&](bool valid, int dbuf = 0) {
        const int tap = (kt * p.kH + kh) * p.kW + kw;
        // dual-source convs: chunks >= kc1 read the second activation tensor (its own channel
        // offset, base offsets and extent; weights continue at column wcol2)
        const bool use2 = p.dual && ch >= p.kc1;
        const int c0 = (use2 ? ch - p.kc1 : ch) * BK;
        const int wc0 = use2 ? p.wcol2 + c0 : c0;
        // uniform: tap selector for the mask test, signed byte offset of the tap from the centre
        const unsigned sel = valid ? ((1u << kt) | (1u << (8 + kh)) | (1u << (16 + kw))) : 0xFFFFFFFFu;
        const int kw_rel = KWR ? 0 : kw - p.pW;       // KWR rows carry their own input column
        const unsigned s_off =
            (unsigned)(((((kt - p.pT) * p.Hi + (kh - p.pH)) * p.Wi + kw_rel) * p.ldx + c0) * 4);
        // nearest-2x upsampled input: the tap's stored offset is floor((par + k - p) / 2) rows / columns from the
        // centre's, i.e. one of two uniform values per axis, selected by the row's parity bits
        unsigned up_h0 = 0, up_h1 = 0, up_w0 = 0, up_w1 = 0;
        bool up2 = false;
        if constexpr (F16) {
            up2 = p.up2 != 0;
            if (up2) {
                const int row_b = p.Wp * p.ldx * 4, col_b = p.ldx * 4;
                up_h0 = (unsigned)(((kh - p.pH) >> 1) * row_b + c0 * 4);
                up_h1 = (unsigned)(((kh - p.pH + 1) >> 1) * row_b + c0 * 4);
                up_w0 = KWR ? 0u : (unsigned)(((kw - p.pW) >> 1) * col_b);
                up_w1 = KWR ? 0u : (unsigned)(((kw - p.pW + 1) >> 1) * col_b);
            }
        }
        auto issue_a = [&](const __amdgpu_buffer_rsrc_t rs, const unsigned (&base)[A_IT], unsigned soff, int klim) {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                bool ok = (a_mask[i] & sel) == sel;
                if (KTAIL) ok = ok && (c0 + swz_col(tid + NT * i)) < klim;
                unsigned off = base[i] + soff;
                if constexpr (F16) {
                    if (up2) off = base[i] + ((a_mask[i] & (1u << 24)) ? up_h1 : up_h0) + ((a_mask[i] & (1u << 25)) ? up_w1 : up_w0);
                }
                if constexpr (DMA) {
                    // wave-uniform LDS destination: this wave's 1-KiB chunk of the tile image
                    if ((A_F4 % NT == 0) || (wave_u * 64 + NT * i < A_F4))
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(
                            rs, (lds_ptr_t)(As + dbuf * ASTG + (wave_u * 64 + NT * i) * 4), 16, ok ? off : kOOB, 0, 0, COH);
                } else {
                    ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : kOOB, 0, COH));
                }
            }
        };
        if (use2)
            issue_a(rsrc_x2, a_off2, (unsigned)(c0 * 4), p.kA2);
        else
            issue_a(rsrc_x, a_off, s_off, p.kA);
#pragma unroll
        for (int kw2 = 0; kw2 < KW_T; ++kw2) {
            const unsigned s_woff = valid ? (unsigned)(((size_t)(tap + kw2) * p.w_tap_stride + wc0) * 4) : kOOB;
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                unsigned off = b_off[i] + s_woff;
                if (KTAIL) off = (wc0 + swz_col(tid + NT * i)) < p.kB ? off : kOOB;
                if constexpr (DMA) {
                    if ((B_F4 % NT == 0) || (wave_u * 64 + NT * i < B_F4))
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(
                            rsrc_w, (lds_ptr_t)(Bs + dbuf * BSTG + kw2 * BN * LDK + (wave_u * 64 + NT * i) * 4), 16, off, 0, 0, 0);
                } else {
                    rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, off, 0, 0));
                }
            }
        }
    };

    auto store_tiles = [&](int buf) {
        float* Ab = As + buf * ASTG;
        float* Bb = Bs + buf * BSTG;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int idx = tid + NT * i;
            if (idx < A_F4) *reinterpret_cast<f32x4*>(Ab + (idx / F4R) * LDK + (idx % F4R) * 4) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int idx = tid + NT * i;
            if (idx < B_F4) *reinterpret_cast<f32x4*>(Bb + (idx / F4R) * LDK + (idx % F4R) * 4) = rb[i];
        }
    };

    acc_t acc[TM][TN];
    acc_t acc2[X3 ? TM : 1][X3 ? TN : 1];        // split operands: the 2^12-scaled cross terms hi.lo' + lo'.hi
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < MF::NACC; ++r) {
                acc[i][j][r] = 0.f;
                if constexpr (X3) acc2[i][j][r] = 0.f;
            }

    // KWR: output row m_local of the tile lives at LDS row m_local + 2 * (m_local / Wo) (+ kw for tap kw)
    int a_lrow[KWR ? TM : 1];
    if constexpr (KWR) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int ml = wm * WTM + i * MT + (lane % MT);
            a_lrow[i] = ml + (KWR - 1) * (int)fastdiv((unsigned)ml, p.dv_wo);
        }
    }
    const int frag_off_a = KWR ? 0 : (wm * WTM + (lane % MT)) * LDK + (DMA ? 0 : (lane / MT) * 4);
    const int frag_off_b = (wn * WTN + (lane % MT)) * LDK + (DMA ? 0 : (lane / MT) * 4);
    const int frag_sw = ((lane % MT) >> SWS) & (F4R - 1);       // DMA: row swizzle of this lane's rows

    // ---- residual prefetch: for same-shape residual adds with few accumulator tiles per wave the
    // residual values are requested BEFORE the k-loop, so their HBM latency hides under it ----
    const bool to_partial = p.split_k > 1;
    const bool res_add = !to_partial && (p.flags & PTX_EPI_RES_ADD) && !(F16 && (p.flags & kFusedEpiFlags));
    constexpr bool kResEarly = (TM * TN * MF::NACC) <= 16;   // keeps multi-tile waves (stem) under 128 regs
    // fp16-operand tiles with any fused-stage flag take their own epilogue below (kFusedEpiFlags)
    const bool fused_epi = F16 && p.split_k <= 1 && (p.flags & kFusedEpiFlags) != 0;
    const __amdgpu_buffer_rsrc_t rsrc_r =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res), 0, res_add ? p.r_bytes : 0u, 0x00020000);
    float rv[TM][TN][MF::NACC];
    auto load_residual = [&](int i, int j) {
        const int co = n0 + wn * WTN + j * MT + (lane % MT);
        const int mrow = m0 + wm * WTM + i * MT;
#pragma unroll
        for (int r = 0; r < MF::NACC; ++r) {
            const int m = mrow + MF::row(r, lane);
            const unsigned off = ((unsigned)m * (unsigned)p.ldr + (unsigned)co) * 4u;
            rv[i][j][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                        rsrc_r, (res_add && co < p.ncol && m < p.M) ? off : kOOB, 0, COH));
        }
    };
    // (REPI tiles fetch the residual row-major in their own epilogue; the column-wise prefetch only serves the paths that
    // fall back to the column-wise epilogue: split-K partials and shortcut-A residuals)
    const bool repi_fast = REPI && !CHAIN && !to_partial && !(p.flags & PTX_EPI_RES_PADA);
    if (kResEarly && !repi_fast) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) load_residual(i, j);
    }

    // fused generator stages (fp16 tiles): the half skip operand, requested now so that its round trip overlaps the
    // operand tiles' (fused_stage_skip_prefetch)
    constexpr int kSkipNP0 = F16 ? (MT + 64 / (WTN / 8) - 1) / (64 / (WTN / 8)) : 1;
    constexpr int kSkipNP = kSkipNP0 < kSkipEarlyPasses ? kSkipNP0 : kSkipEarlyPasses;
    f32x4 skip_rq[F16 ? TM * kSkipNP : 1];
    bool skip_early = false;
    if constexpr (F16) {
        skip_early = fused_epi && (p.flags & (PTX_EPI_RES_ADD | PTX_EPI_RES_PADA)) != 0 && (p.flags & PTX_RES_F16) != 0 &&
                     !(p.flags & kNoSkipEarly);
        if (skip_early) fused_stage_skip_prefetch<MF, TM, TN, WTM, WTN, MT>(p, skip_rq, m0, n0, wm, wn, lane);
    }

    // fragment registers, rotated across sub-steps.  The slot sequence must close on itself at the
    // step boundary with compile-time indices: 2 slots for an even sub-step count, KSUB for odd.
    static_assert(KW_T * KSUB >= 2, "at least two sub-steps per k-step");
    // K22 (kW-folded stem: only 21 of the 24 k of a chunk carry data): the last sub-step covers
    // k = 16..21 with two 8-byte reads per row -- lane group g gets (16+2g, 17+2g) and (20+2g, 21+2g)
    // -- and 3 MFMAs pairing (16,18) (17,19) (20,22); the pair (21,23) is all padding and is dropped:
    // 11 instead of 12 MFMAs per tap.
    static_assert(!K22 || ((BK == 24 || (DMA && BK == 32)) && MT == 32), "K22 is the 32x32x2 stem path (BK 24, or 32 under DMA)");
    // live sub-steps of a k-step: the LDS-DMA K22 tile stages 32-float rows of which 22 carry data -- sub-steps 0, 1
    // (k 0..15), the 3-MFMA sub-step 2 (k 16..21), nothing for k 24..31
    constexpr int KLIVE = (K22 && DMA) ? 3 : KW_T * KSUB;      // KWR: the sub-steps of the three kw taps follow each other
    constexpr int NSLOT = (KLIVE % 2) ? KLIVE : 2;
    constexpr int NF = X3 ? 2 : 1;           // 16-byte reads per operand row per sub-step
    f32x4 fa[NSLOT][TM][NF], fb[NSLOT][TN][NF];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    auto read_frags = [&](int buf, int s_, int slot, int offa, int offb) {
        const int kw_i = KWR ? s_ / KSUB : 0;
        const int ks = KWR ? s_ % KSUB : s_;
        // A row base (floats from As) and its 16-byte-slot swizzle key for wave-tile row block i
        auto a_row = [&](int i, int& sw) -> int {
            if constexpr (KWR) {
                const int lr = a_lrow[i] + kw_i;
                sw = (lr >> SWS) & (F4R - 1);
                return buf * ASTG + offa + lr * LDK;
            } else {
                sw = frag_sw;
                return buf * ASTG + offa + i * MT * LDK;
            }
        };
        const float* Bk = Bs + buf * BSTG + kw_i * BN * LDK + offb;
        if (K22 && ks == KLIVE - 1) {
            // floats (16 + 2g, 17 + 2g) and (20 + 2g, 21 + 2g) of the row; under DMA they sit in the swizzled
            // 16-byte slots 4 and 5 (offa / offb then carry no lane-group term)
            const int g = lane / MT;
            const int lo_off = DMA ? ((4 ^ frag_sw) * 4 + 2 * g) : (16 + 2 * g - 4 * g);
            const int hi_off = DMA ? ((5 ^ frag_sw) * 4 + 2 * g) : (20 + 2 * g - 4 * g);
            const float* Ab = As + buf * ASTG + offa;
            const float* Bb = Bk;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const f32x2 lo = *reinterpret_cast<const f32x2*>(Ab + i * MT * LDK + lo_off);
                const f32x2 hi = *reinterpret_cast<const f32x2*>(Ab + i * MT * LDK + hi_off);
                fa[slot][i][0] = f32x4{lo.x, lo.y, hi.x, hi.y};
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const f32x2 lo = *reinterpret_cast<const f32x2*>(Bb + j * MT * LDK + lo_off);
                const f32x2 hi = *reinterpret_cast<const f32x2*>(Bb + j * MT * LDK + hi_off);
                fb[slot][j][0] = f32x4{lo.x, lo.y, hi.x, hi.y};
            }
            return;
        }
        if constexpr (X3) {
            // lane group g owns the 8 channels of block b = ks * KG + g: swizzled 16-byte slots 2b and 2b + 1
            // (A: floats 8b..8b+3 | 8b+4..8b+7;  B: 8 hi halfs | 8 lo halfs)
            const int b2 = (ks * KG + lane / MT) * 2;
            const int k0 = ((b2 ^ frag_sw) * 4), k1 = (((b2 + 1) ^ frag_sw) * 4);
            const float* Bb = Bk;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                int sw;
                const float* Ar = As + a_row(i, sw);
                fa[slot][i][0] = *reinterpret_cast<const f32x4*>(Ar + ((b2 ^ sw) * 4));
                fa[slot][i][NF - 1] = *reinterpret_cast<const f32x4*>(Ar + (((b2 + 1) ^ sw) * 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                fb[slot][j][0] = *reinterpret_cast<const f32x4*>(Bb + j * MT * LDK + k0);
                fb[slot][j][NF - 1] = *reinterpret_cast<const f32x4*>(Bb + j * MT * LDK + k1);
            }
            return;
        }
        const int koff = DMA ? (((ks * KG + lane / MT) ^ frag_sw) * 4) : ks * 4 * KG;
        const float* Bb = Bk + koff;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            int sw;
            const float* Ar = As + a_row(i, sw);
            fa[slot][i][0] = *reinterpret_cast<const f32x4*>(Ar + (DMA ? (((ks * KG + lane / MT) ^ sw) * 4) : ks * 4 * KG));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[slot][j][0] = *reinterpret_cast<const f32x4*>(Bb + j * MT * LDK);
    };
    auto mma_frags = [&](int slot, int nr) {
        if constexpr (F16) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = MF::mma16(fa[slot][i][0], fb[slot][j][0], acc[i][j]);
            return;
        }
        if constexpr (X3) {
            // split the lane's 8 fp32 A values into (hi, lo) halfs: v_cvt_pk_f16_f32 (round to nearest even),
            // lo = half(v - float(hi)) -- the difference is exact in fp32
            typedef float f32x8 __attribute__((ext_vector_type(8)));
            f32x4 ahi[TM], alo[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const f32x4 r0 = fa[slot][i][0], r1 = fa[slot][i][NF - 1];
                const f32x8 v = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#ifdef PTX_X3_NOCONV_EXPERIMENT      // timing experiment only (garbage results): what the loop costs without the split VALU
                ahi[i] = r0; alo[i] = r1; continue;
#endif
                const half8 h = __builtin_convertvector(v, half8);
                ahi[i] = __builtin_bit_cast(f32x4, h);
                // v - float(hi) as ONE mixed-precision fma per value (v_fma_mix_f32 reads the half in place:
                // -1.0 * hi + v), instead of v_cvt_f32_f16 + v_sub_f32
                f32x8 d;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float hp = ahi[i][e];          // two packed halfs
                    float d0, d1;
                    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d0) : "v"(hp), "v"(v[2 * e]));
                    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d1) : "v"(hp), "v"(v[2 * e + 1]));
                    d[2 * e] = d0 * 4096.f;              // scaled lo: stays a normal half whenever hi is one
                    d[2 * e + 1] = d1 * 4096.f;
                }
                alo[i] = __builtin_bit_cast(f32x4, __builtin_convertvector(d, half8));
            }
            // term-major order: consecutive MFMAs write different accumulators
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc2[i][j] = MF::mma16(ahi[i], fb[slot][j][NF - 1], acc2[i][j]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc2[i][j] = MF::mma16(alo[i], fb[slot][j][0], acc2[i][j]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = MF::mma16(ahi[i], fb[slot][j][0], acc[i][j]);
            return;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r < nr) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = MF::mma(fa[slot][i][0][r], fb[slot][j][0][r], acc[i][j]);
            }
        }
    };

    // ---- main loop.  Register-staged double buffering with the k-step software-pipelined inside
    // the wave so the MFMA stream never drains at the step boundary:
    //   sub-step 0        : MFMAs on fragments already in registers, next fragments requested
    //   sub-step STORE_KS : tile s+1 (loaded during step s-1) is written to the other LDS buffer and
    //                       the global loads of tile s+2 are issued -- under this step's MFMAs
    //   last sub-step     : lgkmcnt(0) + barrier, first fragments of step s+1 requested, then the
    //                       last MFMAs of step s (their fragments were fetched before the barrier)
    // Safety with 2 LDS buffers: after the barrier of step s-1 nobody reads buffer (s-1)&1 again
    // (its last fragments were completed before that barrier), so step s may overwrite it; those
    // writes complete (lgkmcnt(0)) before the barrier of step s, after which step s+1 reads them.
    constexpr int STORE_KS = KSUB >= 3 ? 1 : 0;
    PTX_IG_TL(1);
    if constexpr (DMA && NSTAGE >= 3) {
        // N-stage LDS-DMA ring: tile s+NSTAGE is requested right after the barrier of step s, i.e.
        // NSTAGE-1 k-steps before it is read.  Every wave issues exactly NPS DMA instructions per
        // step, so a counted `s_waitcnt vmcnt((NSTAGE-2)*NPS)` before the raw barrier retires tile s+1
        // while the younger tiles stay in flight (a plain __syncthreads() would drain everything --
        // cdna_hip_programming.md section 5).
        static_assert(A_F4 % NT == 0 && B_F4 % NT == 0, "uniform DMA count per wave");
        constexpr int NPS = A_IT + B_IT;
        static_assert((NSTAGE - 2) * NPS < 64, "vmcnt field");
        auto ring_barrier = [&]() {
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"((NSTAGE - 2) * NPS) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        if (my_steps > 0) {
#pragma unroll
            for (int t = 0; t < NSTAGE - 1; ++t) {
                load_tiles(t < my_steps, t);
                advance();
            }
            ring_barrier();
            PTX_IG_TL(2);
            int offa = frag_off_a, offb = frag_off_b;
            post_barrier_offsets(offa, offb);
            load_tiles(NSTAGE - 1 < my_steps, NSTAGE - 1);
            advance();
            read_frags(0, 0, 0, offa, offb);
            int cur = 0;
            for (int it = 0; it < my_steps; ++it) {
                const int nxt = cur == NSTAGE - 1 ? 0 : cur + 1;
#pragma unroll
                for (int ks = 0; ks < KSUB; ++ks) {
                    if (ks == KSUB - 1) {
                        ring_barrier();
                        post_barrier_offsets(offa, offb);
                        load_tiles(it + NSTAGE < my_steps, cur);
                        advance();
                        read_frags(nxt, 0, (ks + 1) % NSLOT, offa, offb);
                    } else {
                        read_frags(cur, ks + 1, (ks + 1) % NSLOT, offa, offb);
                    }
                    mma_frags(ks % NSLOT, 4);
                }
                cur = nxt;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the trailing (all-OOB) DMAs
        }
    } else
    if constexpr (DMA) {
        // LDS-DMA staging: no staging registers, no ds_write.  The DMA of tile s+2 is issued right
        // after the barrier of step s (which frees buffer s&1: its last fragments were read before
        // the barrier) and has a whole k-step to land before the barrier of step s+1 drains it
        // (__syncthreads() carries the vmcnt(0) for the pending LDS writes).
        if (my_steps > 0) {
            load_tiles(true, 0);
            advance();
            step_barrier();
            PTX_IG_TL(2);
            int offa = frag_off_a, offb = frag_off_b;
            post_barrier_offsets(offa, offb);
            load_tiles(my_steps > 1, 1);
            advance();
            read_frags(0, 0, 0, offa, offb);
            for (int it = 0; it < my_steps; ++it) {
                const int buf = it & 1;
#pragma unroll
                for (int ks = 0; ks < KLIVE; ++ks) {
                    if (ks == KLIVE - 1) {
                        step_barrier();
                        post_barrier_offsets(offa, offb);
                        load_tiles(it + 2 < my_steps, buf);
                        advance();
                        read_frags(buf ^ 1, 0, (ks + 1) % NSLOT, offa, offb);
                    } else {
                        read_frags(buf, ks + 1, (ks + 1) % NSLOT, offa, offb);
                    }
                    mma_frags(ks % NSLOT, (K22 && ks == KLIVE - 1) ? 3 : 4);
                }
            }
        }
    } else
    if (my_steps > 0) {
        load_tiles(true);
        advance();
        store_tiles(0);
        load_tiles(my_steps > 1);
        advance();
        step_barrier();
        int offa = frag_off_a, offb = frag_off_b;
        post_barrier_offsets(offa, offb);
        read_frags(0, 0, 0, offa, offb);
        for (int it = 0; it < my_steps; ++it) {
            const int buf = it & 1;
#pragma unroll
            for (int ks = 0; ks < KSUB; ++ks) {
                if (ks == KSUB - 1) {
                    step_barrier();
                    post_barrier_offsets(offa, offb);
                    read_frags(buf ^ 1, 0, (ks + 1) % NSLOT, offa, offb);
                } else {
                    read_frags(buf, ks + 1, (ks + 1) % NSLOT, offa, offb);
                }
                if (ks == STORE_KS) {
                    store_tiles(buf ^ 1);
                    load_tiles(it + 2 < my_steps);
                    advance();
                }
                mma_frags(ks % NSLOT, (K22 && ks == KSUB - 1) ? 3 : 4);
            }
        }
    }

    PTX_IG_TL(3);
    if constexpr (CHAIN) {
        // ================= chained pointwise tail =================
        // P: the intermediate tile as the A operand of the second GEMM -- [BN / BK k-chunks][BM][BK] floats, 16-byte slots
        // XOR-swizzled by row exactly like a DMA-staged A tile, so the fragment reads are the main loop's.  It aliases
        // the (now idle) A stages when they are big enough, else it sits behind the tile buffers.
        constexpr int KC_MAX = BN / BK;                               // k-chunks of the tail's K axis (K = N1 <= BN)
        // kw-reuse tiles: their stages (a halo'd A image + three filter tiles) are bigger than P, so P takes the WHOLE tile
        // area and the tail's own filter stages (one [BN][BK] tile each) sit right behind it
        static_assert(!KWR || BM * BN <= NSTAGE * (ASTG + BSTG), "kw-reuse chained tile: P must fit the idle stages");
        constexpr bool kAlias = KWR ? true : BM * BN <= NSTAGE * ASTG;
        float* P = kAlias ? smem : smem + NSTAGE * (ASTG + BSTG);
        float* Bt = KWR ? smem + BM * BN : Bs;                        // the tail's filter stages
        constexpr int BTS = KWR ? BN * LDK : BSTG;                    // ... and their stride
        // the main loop leaves its trailing (all-OOB, zero-writing) DMAs in flight and other waves may still be reading
        // the last stage: drain both before P and the B stages are written
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        auto fold_x3 = [&]() {      // split operands: acc += 2^-12 acc2 (exact scaling), as the unchained epilogue does
            if constexpr (X3) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < MF::NACC; ++r) acc[i][j][r] = fmaf(acc2[i][j][r], 1.0f / 4096.0f, acc[i][j][r]);
            }
        };
        auto clear_acc = [&]() {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < MF::NACC; ++r) {
                        acc[i][j][r] = 0.f;
                        if constexpr (X3) acc2[i][j][r] = 0.f;
                    }
        };
        fold_x3();
        {
            const bool relu1 = (p.flags & PTX_EPI_RELU) != 0;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n1 = wn * WTN + j * MT + (lane % MT);       // intermediate channel = k of the tail
                const float b1 = (p.bias && n1 < p.ncol) ? p.bias[n1] : 0.f;
                const int chunk = n1 / BK, kk = n1 % BK;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < MF::NACC; ++r) {
                        const int ml = wm * WTM + i * MT + MF::row(r, lane);
                        float v = acc[i][j][r] + b1;
                        v = relu1 ? fmaxf(v, 0.f) : v;
                        v = n1 < p.Co ? v : 0.f;                      // columns beyond N1 multiply zero filter columns anyway
                        const int slot = (kk >> 2) ^ ((ml >> SWS) & (F4R - 1));
                        P[chunk * (BM * BK) + ml * BK + slot * 4 + (kk & 3)] = v;
                    }
            }
        }
        // filter tiles of the tail: rows n2 of chunk nc, columns kc * BK .. + BK of w2
        unsigned b2_off[B_IT];
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int idx = tid + NT * i;
            const int row = idx / F4R;
            b2_off[i] = idx < B_F4 ? ((unsigned)row * (unsigned)p.ldw2 + (unsigned)swz_col(idx)) * 4u : kOOB;
        }
        const __amdgpu_buffer_rsrc_t rsrc_w2 =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w2), 0, p.w2_bytes, 0x00020000);
        const int kc2 = min((p.kB2 + BK - 1) / BK, KC_MAX);          // live k-chunks (uniform)
        const int nc2 = (p.ncol2 + BN - 1) / BN;                      // N chunks of the final output
        const int steps2 = nc2 * kc2;
        auto load_b2 = [&](int s2, int dbuf) {
            const bool valid = s2 < steps2;
            const int nc = s2 / kc2, kc = s2 - nc * kc2;
            const unsigned base = (unsigned)(((size_t)nc * BN * p.ldw2 + (size_t)kc * BK) * 4);
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                const int idx = tid + NT * i;
                const bool ok = valid && b2_off[i] != kOOB && (nc * BN + idx / F4R) < p.w2_rows && (kc * BK + swz_col(idx)) < p.kB2;
                // (the offset is a plain local: indexing an array inside the builtin's voffset operand makes hipcc's host
                // pass drop the kernel stub -- DESIGN.md 3.8 toolchain note)
                unsigned off = b2_off[i] + base;
                off = ok ? off : kOOB;
                if ((B_F4 % NT == 0) || (wave_u * 64 + NT * i < B_F4))
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        rsrc_w2, (lds_ptr_t)(Bt + dbuf * BTS + (wave_u * 64 + NT * i) * 4), 16, off, 0, 0, 0);
            }
        };
        const bool res2 = (p.flags2 & PTX_EPI_RES_ADD) != 0, relu2 = (p.flags2 & PTX_EPI_RELU) != 0;
        const __amdgpu_buffer_rsrc_t rsrc_r2 =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res), 0, res2 ? p.r_bytes : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc_y2 = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
        load_b2(0, 0);
        step_barrier();                                               // P complete, tile 0 landed
        // (kw-reuse main loops address A rows through a_lrow: the parked tile is indexed by the plain tile row again)
        int offa = KWR ? (wm * WTM + (lane % MT)) * LDK : frag_off_a, offb = frag_off_b;
        post_barrier_offsets(offa, offb);
        load_b2(1, 1);
        auto read_frags2 = [&](int kc, int bufb, int ks, int slot) {
            if constexpr (X3) {
                // 8-channel blocks: two 16-byte slots per lane group (P: floats 8b .. 8b+7; filter: 8 hi | 8 lo halfs)
                const int b2 = (ks * KG + lane / MT) * 2;
                const int k0 = (b2 ^ frag_sw) * 4, k1 = ((b2 + 1) ^ frag_sw) * 4;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    fa[slot][i][0] = *reinterpret_cast<const f32x4*>(P + kc * (BM * BK) + offa + i * MT * LDK + k0);
                    fa[slot][i][NF - 1] = *reinterpret_cast<const f32x4*>(P + kc * (BM * BK) + offa + i * MT * LDK + k1);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    fb[slot][j][0] = *reinterpret_cast<const f32x4*>(Bt + bufb * BTS + offb + j * MT * LDK + k0);
                    fb[slot][j][NF - 1] = *reinterpret_cast<const f32x4*>(Bt + bufb * BTS + offb + j * MT * LDK + k1);
                }
                return;
            }
            const int koff = ((ks * KG + lane / MT) ^ frag_sw) * 4;
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[slot][i][0] = *reinterpret_cast<const f32x4*>(P + kc * (BM * BK) + offa + i * MT * LDK + koff);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[slot][j][0] = *reinterpret_cast<const f32x4*>(Bt + bufb * BTS + offb + j * MT * LDK + koff);
        };
        // REPI: the tail's epilogue runs row-major through a per-wave LDS slice behind P and the tile stages
        using RE2 = RowEpi<WTN, MT>;
        float* Ep = smem + NSTAGE * (ASTG + BSTG) + (kAlias ? 0 : BM * BN) + wave_u * RE2::FLOATS;
        f32x4 res4c[REPI ? TM * RE2::NP : 1];
        int nc = 0, kc = 0;
        for (int s2 = 0; s2 < steps2; ++s2) {
            const int buf = s2 & 1;
            if (REPI && kc == 0) {
                clear_acc();
                if constexpr (REPI)
                    rowmajor_load_residual<MF, TM, TN, WTM, WTN, MT>(res4c, rsrc_r2, res2, m0 + wm * WTM, nc * BN + wn * WTN, p.M,
                                                                     p.ncol2, p.ldr, lane);
            }
            if (!REPI && kc == 0) {
                // a new 64-wide (BN) slice of the output: clear the accumulators, request its residual values -- their
                // HBM latency hides under the slice's MFMAs
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
#pragma unroll
                        for (int r = 0; r < MF::NACC; ++r) {
                            acc[i][j][r] = 0.f;
                            if constexpr (X3) acc2[i][j][r] = 0.f;
                        }
                        const int co = nc * BN + wn * WTN + j * MT + (lane % MT);
                        const int mrow = m0 + wm * WTM + i * MT;
#pragma unroll
                        for (int r = 0; r < MF::NACC; ++r) {
                            const int m = mrow + MF::row(r, lane);
                            const unsigned off = ((unsigned)m * (unsigned)p.ldr + (unsigned)co) * 4u;
                            rv[i][j][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                                        rsrc_r2, (res2 && co < p.ncol2 && m < p.M) ? off : kOOB, 0, 0));
                        }
                    }
            }
            read_frags2(kc, buf, 0, 0);
#pragma unroll
            for (int ks = 0; ks < KSUB; ++ks) {
                if (ks + 1 < KSUB) read_frags2(kc, buf, ks + 1, (ks + 1) % NSLOT);
                mma_frags(ks % NSLOT, 4);
            }
            step_barrier();                                           // stage `buf` is free; tile s2 + 1 has landed
            post_barrier_offsets(offa, offb);
            load_b2(s2 + 2, buf);
            ++kc;
            if (kc == kc2) fold_x3();
            if (REPI && kc == kc2) {
                if constexpr (REPI)
                    rowmajor_store_tile<MF, TM, TN, WTM, WTN, MT>(acc, res4c, Ep, p.bias2, rsrc_y2, relu2, m0 + wm * WTM,
                                                                  nc * BN + wn * WTN, p.M, p.ncol2, p.ldy, lane);
                kc = 0;
                ++nc;
            }
            if (!REPI && kc == kc2) {
                // ---- epilogue of output slice nc: bias2 + residual + ReLU ----
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int co = nc * BN + wn * WTN + j * MT + (lane % MT);
                    const bool co_ok = co < p.ncol2;
                    const float bv = (p.bias2 && co_ok) ? p.bias2[co] : 0.f;
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int mrow = m0 + wm * WTM + i * MT;
#pragma unroll
                        for (int r = 0; r < MF::NACC; ++r) {
                            const int m = mrow + MF::row(r, lane);
                            float v = acc[i][j][r] + bv + rv[i][j][r];
                            v = relu2 ? fmaxf(v, 0.f) : v;
                            const unsigned off = ((unsigned)m * (unsigned)p.ldy + (unsigned)co) * 4u;
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc_y2,
                                                                  (co_ok && m < p.M) ? off : kOOB, 0, 0);
                        }
                    }
                }
                kc = 0;
                ++nc;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the trailing (all-OOB) DMAs of the tail
        return;
    }
    if constexpr (X3) {           // fold the scaled cross terms back: acc += 2^-12 acc2 (exact scaling)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < MF::NACC; ++r) acc[i][j][r] = fmaf(acc2[i][j][r], 1.0f / 4096.0f, acc[i][j][r]);
    }
    if constexpr (F16) {
        if (fused_epi) {          // generator stage: per-sample affine / halfs out / dual output / half skip / tanh
            fused_stage_epilogue<MF, TM, TN, WTM, WTN, MT>(p, acc, m0, n0, wm, wn, lane, smem, wave_u, skip_rq, skip_early);
            return;
        }
    }
    // ---- epilogue: bias + residual + ReLU, branch-free through buffer resources (out-of-range
    // stores are dropped, out-of-range loads read 0); residual values of a tile are requested in
    // one batch before they are consumed. ----
    // split-K partial slabs are dense [M][ncol]; the final tensor has row stride ldy
    if constexpr (REPI && !CHAIN) {
        if (repi_fast) {
            using RE = RowEpi<WTN, MT>;
            const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y + (size_t)zb * p.bs_y, 0, p.y_bytes, 0x00020000);
            f32x4 res4[TM * RE::NP];
            rowmajor_load_residual<MF, TM, TN, WTM, WTN, MT, COH>(res4, rsrc_r, res_add, m0 + wm * WTM, n0 + wn * WTN, p.M, p.ncol, p.ldr, lane);
            // every wave is done with the operand tiles and the trailing (zero-writing) DMAs have landed: the tile buffers
            // become the waves' parking slices
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TM * RE::NP) : "memory");      // (the residual loads just issued may stay in flight)
            __syncthreads();
            PTX_IG_TL(4);
            rowmajor_store_tile<MF, TM, TN, WTM, WTN, MT, COH>(acc, res4, smem + wave_u * RE::FLOATS, p.bias, rs_y,
                                                          (p.flags & PTX_EPI_RELU) != 0, m0 + wm * WTM, n0 + wn * WTN, p.M, p.ncol,
                                                          p.ldy, lane);
            PTX_IG_TL_END();
            return;
        }
    }
    bool partial_stored = false;
    if constexpr (REPI && !CHAIN) {
        if (to_partial) {
            // split-K partial tile: the same row-major path, no bias / residual / ReLU, into this split's dense [M][ncol]
            // slab -- 16-byte WRITE-THROUGH stores (sc1): the slab is read once, by another workgroup (the reduce kernel or
            // the last arriver below), so it need not stay in this XCD's L2, and 4-byte sc1 stores would cost one fabric
            // write each (MI355X_MICROARCH.md, stores of each flavour)
            using RE = RowEpi<WTN, MT>;
            const __amdgpu_buffer_rsrc_t rs_p =
                __builtin_amdgcn_make_buffer_rsrc(p.partial + (size_t)zs * p.M * p.ncol, 0, p.y_bytes, 0x00020000);
            f32x4 res4[TM * RE::NP];
#pragma unroll
            for (int i = 0; i < TM * RE::NP; ++i) res4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            rowmajor_store_tile<MF, TM, TN, WTM, WTN, MT, 16>(acc, res4, smem + wave_u * RE::FLOATS, nullptr, rs_p, false,
                                                              m0 + wm * WTM, n0 + wn * WTN, p.M, p.ncol, p.ncol, lane);
            partial_stored = true;
        }
    }
    if (!partial_stored) {
    float* ybase = to_partial ? p.partial + (size_t)zs * p.M * p.ncol : p.y + (size_t)zb * p.bs_y;
    const unsigned ldo = to_partial ? (unsigned)p.ncol : (unsigned)p.ldy;
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(ybase, 0, p.y_bytes, 0x00020000);
    const bool res_pada = !to_partial && (p.flags & PTX_EPI_RES_PADA);
    const bool res_up = (p.flags & PTX_EPI_RES_UP) != 0;
    const bool relu = !to_partial && (p.flags & PTX_EPI_RELU);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int co = n0 + wn * WTN + j * MT + (lane % MT);
        const bool co_ok = co < p.ncol;
        const float bv = (!to_partial && p.bias && co_ok) ? p.bias[co] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mrow = m0 + wm * WTM + i * MT;
            if (!kResEarly) load_residual(i, j);
#pragma unroll
            for (int r = 0; r < MF::NACC; ++r) {
                const int m = mrow + MF::row(r, lane);
                float v = acc[i][j][r] + bv + rv[i][j][r];
                if (res_pada && co < (res_up ? p.Co : p.res_C) && m < p.M) {     // shortcut A (BasicBlock / NL nets only)
                    const int wo = m % p.Wo;
                    int t = m / p.Wo;
                    const int ho = t % p.Ho;
                    t /= p.Ho;
                    const int to = t % p.To;
                    const int n = t / p.To;
                    const int rt = res_up ? to >> p.res_sT : to * p.res_sT, rh = res_up ? ho >> p.res_sH : ho * p.res_sH,
                              rw = res_up ? wo >> p.res_sW : wo * p.res_sW;
                    const size_t pos = (((size_t)n * p.res_T + rt) * p.res_H + rh) * p.res_W + rw;
                    v += p.res[pos * p.ldr + co];
                }
                v = relu ? fmaxf(v, 0.f) : v;
                const unsigned off = ((unsigned)m * ldo + (unsigned)co) * 4u;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc_y,
                                                      (co_ok && m < p.M) ? off : kOOB, 0, COH);
            }
        }
    }
    }   // !partial_stored
    // ---- split-K, fused reduction (PTX_SPLITK_FUSED): the LAST split block to finish a tile sums the partial tiles of
    // all splits -- in split order, so the result is bit-identical to the separate reduce kernel and independent of
    // which block happens to be last -- applies the epilogue and writes y.  No second launch.  Hand-off: write-through
    // partial stores (above; the column-wise path of non-REPI tiles is followed by a release fence instead), every wave
    // drains, the workgroup meets, ONE returning agent-scope atomic per workgroup on the tile's arrival counter; the last
    // arriver reads the slabs with sc1 loads and leaves the counter at zero.  (Round 2's version bracketed the counter with
    // two __threadfence() -- buffer_wbl2 + buffer_inv across 8 XCD L2s per workgroup -- and measured SLOWER than the
    // reduce launch: 33 -> 54 us on layer4's 3x3x3.)
    PTX_IG_TL_END();
    if constexpr (COH == 0)
    if (to_partial && p.counters) {
        if (!partial_stored) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");       // plain b32 stores: write the L2 back
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* flag = reinterpret_cast<unsigned*>(smem);
        if (tid == 0) {
            const unsigned prev = __hip_atomic_fetch_add(p.counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned last = prev == (unsigned)p.split_k - 1u ? 1u : 0u;
            if (last) __hip_atomic_store(p.counters + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            flag[0] = last;
        }
        __syncthreads();
        if (!flag[0]) return;
        if (!(p.flags & PTX_EPI_RES_PADA)) {
            splitk_reduce_tile(p, BM, BN, tile, tid, NT);
            return;
        }
        // shortcut-A residual (a strided gather with zero channels above res_C): the element-wise epilogue of the reduce kernel
        constexpr unsigned kOOBr = 0x80000000u;
        const size_t slab = (size_t)p.M * p.ncol;
#pragma unroll 1
        for (int e = tid * 4; e < BM * BN; e += NT * 4) {
            const int m = m0 + e / BN, co = n0 + e % BN;
            if (co < p.ncol && m < p.M) {
                const unsigned poff = (unsigned)(((size_t)m * p.ncol + co) * 4);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
                for (int z = 0; z < p.split_k; ++z) {
                    const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(p.partial + (size_t)z * slab, 0, p.y_bytes, 0x00020000);
                    const f32x4 u = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_p, poff < kOOBr ? poff : kOOBr, 0, 16));
                    v = z == 0 ? u : v + u;
                }
                f32x4 o;
                o.x = conv_epilogue(p, v.x, m, co);
                o.y = conv_epilogue(p, v.y, m, co + 1);
                o.z = conv_epilogue(p, v.z, m, co + 2);
                o.w = conv_epilogue(p, v.w, m, co + 3);
                *reinterpret_cast<f32x4*>(p.y + (size_t)m * p.ldy + co) = o;
            }
        }
    }
}

// one tile per workgroup: the plain launch
template <int BM, int BN, int BK, int WM, int WN, int MT, bool KTAIL, bool K22, bool DMA, int NSTAGE, bool F16 = false,
          bool X3 = false, int KWR = 0, bool CHAIN = false, bool REPI = false>
__global__ void __launch_bounds__(64 * WM * WN) conv_igemm_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    conv_igemm_tile<BM, BN, BK, WM, WN, MT, KTAIL, K22, DMA, NSTAGE, F16, X3, KWR, CHAIN, REPI, 0>(
        p, xcd_remap(blockIdx.x, p.m_tiles * p.n_tiles), (int)blockIdx.y, (int)blockIdx.z, smem);
}

// conv_igemm.hip: descriptor validation shared by every conv entry point
int validate_desc(const ptx_conv3d_desc* d);
// conv_igemm.hip: descriptor + tensors -> kernel arguments (tile-independent half), then the tile-dependent half
int make_conv_args(const ptx_conv3d_desc* d, const float* x, const float* x2, const float* w_packed, const float* bias,
                   const float* res, float* y, const ptx_conv_fused_ext* ext, ConvArgs& a);
int finalize_conv_args(ConvArgs& a, int BM, int BN, int BK, int kwr, bool direct, int split_k, int batch);

}  // namespace ptx

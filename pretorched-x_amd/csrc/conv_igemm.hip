// Implicit-GEMM 3-D convolution on the gfx950 fp32 matrix cores, with fused
// bias(+folded BN) + residual + ReLU epilogue.  Channels-last activations (NDHWC), K-major
// packed filters.  One kernel template serves every conv of the hot path:
// the 7x7x7 stem (after the kW fold), 3x3x3 body, 1x1x1 pointwise / shortcut-B, the (1,k,k) and
// (k,1,1) factored convs, the two-source "conv3 + shortcut" GEMM and -- in batched mode -- the
// non-local block's NT matmuls.
//
// GEMM view:  M = N*To*Ho*Wo output positions, N = Co, K = taps * Ci.
//   A[m][k]  gathered on the fly from x (zero outside the image): for one filter tap the BK
//            channels of a row are contiguous in NDHWC, so every lane moves one 16-byte piece.
//   B[n][k]  = w_packed[tap][co][c]  (K contiguous).
// Staging, two flavours per tile shape:
//   * LDS-DMA (default): `buffer_load ... lds` writes 16 B per lane straight into a lane-linear,
//     UNPADDED LDS image; ds_read_b128 bank conflicts are avoided by XOR-swizzling the 16-byte slot
//     index on the per-lane SOURCE address and on the fragment read address.  2..4-stage ring.
//   * register-staged: global -> VGPR -> LDS ([rows][BK+4] floats), used by the BK = 24 stem tile.
// Every load is branch-free: anything that must read as zero (padding taps, rows >= M, channel
// tails) gets a byte offset >= num_records of the buffer resource, for which the hardware returns /
// writes 0 without touching memory.  Per-row base offsets and tap-validity bitmasks are hoisted
// out of the k-loop (4 VALU per activation load, 1 per filter load).
// The k-step is software-pipelined inside the wave: the barrier sits before the last sub-step,
// whose fragments are already in registers; the next tile's first fragments are requested under
// those MFMAs.  Taps that are padding for EVERY row of the tile are skipped (exact, OR of the row
// masks over the workgroup); split-K slices the pruned k-space.
//
// MFMA: v_mfma_f32_32x32x2_f32 (or 16x16x4): lane l supplies A[row = l % MT][k = l / MT] and
// B[k = l / MT][col = l % MT].  A lane's ds_read_b128 returns 4 consecutive k of its row; the
// r-th element of every lane forms one MFMA, i.e. the hardware k index is a fixed permutation of
// the logical one -- harmless because A and B use the same permutation.
// fp32 in, fp32 accumulate, bit-exact k-ordered fma chain (cdna_hip_programming.md section 3).
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
};

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
__device__ __forceinline__ unsigned fastdiv(unsigned n, const unsigned (&dv)[2]) {
    return dv[0] ? (__umulhi(n, dv[0]) >> dv[1]) : n;
}

constexpr unsigned kFusedEpiFlags = PTX_EPI_OUT_F16 | PTX_EPI_AFFINE | PTX_EPI_DUAL_RAW | PTX_RES_F16 | PTX_EPI_TANH;

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

template <class MF, int TM, int TN, int WTM, int WTN, int MT>
__device__ __forceinline__ void fused_stage_epilogue(const ConvArgs& p, typename MF::acc_t (&acc)[TM][TN], int m0, int n0,
                                                     int wm, int wn, int lane, float* smem, int wave) {
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
                    const half8_t h = __builtin_bit_cast(half8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_res, q0 ? e * 2u : kOOB, 0, 0));
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
template <int BM, int BN, int BK, int WM, int WN, int MT, bool KTAIL, bool K22, bool DMA, int NSTAGE, bool F16 = false,
          bool X3 = false, int KWR = 0>
__global__ void __launch_bounds__(64 * WM * WN) conv_igemm_kernel(const ConvArgs p) {
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

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                        // [NSTAGE][AR][LDK]
    float* Bs = smem + NSTAGE * ASTG;        // [NSTAGE][KW_T][BN][LDK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    const int tile = xcd_remap(blockIdx.x, p.m_tiles * p.n_tiles);
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
    const int zb = blockIdx.y;
    const int zs = blockIdx.z;

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
    if (p.kT * p.kH > 1) {
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
                            rs, (lds_ptr_t)(As + dbuf * ASTG + (wave_u * 64 + NT * i) * 4), 16, ok ? off : kOOB, 0, 0, 0);
                } else {
                    ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : kOOB, 0, 0));
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
                                                        rsrc_r, (res_add && co < p.ncol && m < p.M) ? off : kOOB, 0, 0));
        }
    };
    if (kResEarly) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) load_residual(i, j);
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
            fused_stage_epilogue<MF, TM, TN, WTM, WTN, MT>(p, acc, m0, n0, wm, wn, lane, smem, wave_u);
            return;
        }
    }
    // ---- epilogue: bias + residual + ReLU, branch-free through buffer resources (out-of-range
    // stores are dropped, out-of-range loads read 0); residual values of a tile are requested in
    // one batch before they are consumed. ----
    // split-K partial slabs are dense [M][ncol]; the final tensor has row stride ldy
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
                                                      (co_ok && m < p.M) ? off : kOOB, 0, 0);
            }
        }
    }
    // ---- split-K, fused reduction (PTX_SPLITK_FUSED): the LAST split block to finish a tile sums the partial tiles of
    // all splits -- in split order, so the result is bit-identical to the separate reduce kernel and independent of
    // which block happens to be last -- applies the epilogue and writes y.  No second launch.  Release / acquire:
    // device-scope fences around one atomic arrival counter per tile; the last block leaves the counter at zero.
    if (to_partial && p.counters) {
        __threadfence();
        __syncthreads();
        unsigned* flag = reinterpret_cast<unsigned*>(smem);
        if (tid == 0) {
            const unsigned prev = atomicAdd(p.counters + tile, 1u);
            const unsigned last = prev == (unsigned)p.split_k - 1u ? 1u : 0u;
            if (last) atomicExch(p.counters + tile, 0u);
            flag[0] = last;
        }
        __syncthreads();
        if (!flag[0]) return;
        __threadfence();
        const size_t slab = (size_t)p.M * p.ncol;
        // a plain loop over the tile's elements, 4 columns per thread (tiny code: this tail is instantiated per tile shape)
#pragma unroll 1
        for (int e = tid * 4; e < BM * BN; e += NT * 4) {
            const int m = m0 + e / BN, co = n0 + e % BN;
            if (co < p.ncol && m < p.M) {
                const float* src = p.partial + (size_t)m * p.ncol + co;
                f32x4 v = *reinterpret_cast<const f32x4*>(src);
#pragma unroll 1
                for (int z = 1; z < p.split_k; ++z) v += *reinterpret_cast<const f32x4*>(src + z * slab);
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

// y = epilogue(sum over splits of partial)
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const ConvArgs p) {
    const size_t total4 = (size_t)p.M * p.ncol / 4;
    const size_t slab = (size_t)p.M * p.ncol;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        f32x4 v = *reinterpret_cast<const f32x4*>(p.partial + i * 4);
        for (int z = 1; z < p.split_k; ++z) {
            const f32x4 u = *reinterpret_cast<const f32x4*>(p.partial + z * slab + i * 4);
            v += u;
        }
        const size_t e = i * 4;
        const int m = (int)(e / p.ncol);
        const int co = (int)(e - (size_t)m * p.ncol);
        if (p.flags & kFusedEpiFlags) {       // generator stage (see fused_stage_epilogue)
            const float acc4[4] = {v.x, v.y, v.z, v.w};
            float raw[4], out[4];
            const int n = m / p.pps;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool aff = (p.flags & PTX_EPI_AFFINE) != 0;
                const float sc = aff ? p.aff_scale[(size_t)n * p.ld_aff + co + c] : 1.f;
                const float sh = aff ? p.aff_shift[(size_t)n * p.ld_aff + co + c] : 0.f;
                fused_value(p, acc4[c], m, co + c, p.bias ? p.bias[co + c] : 0.f, sc, sh, raw[c], out[c]);
            }
            typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
            if (p.flags & PTX_EPI_OUT_F16)
                *reinterpret_cast<half4_t*>(reinterpret_cast<_Float16*>(p.y) + (size_t)m * p.ldy + co) =
                    half4_t{(_Float16)out[0], (_Float16)out[1], (_Float16)out[2], (_Float16)out[3]};
            else
                *reinterpret_cast<f32x4*>(p.y + (size_t)m * p.ldy + co) = f32x4{out[0], out[1], out[2], out[3]};
            if (p.flags & PTX_EPI_DUAL_RAW)
                *reinterpret_cast<half4_t*>(reinterpret_cast<_Float16*>(p.y_raw) + (size_t)m * p.ld_raw + co) =
                    half4_t{(_Float16)raw[0], (_Float16)raw[1], (_Float16)raw[2], (_Float16)raw[3]};
            continue;
        }
        f32x4 o;
        o.x = conv_epilogue(p, v.x, m, co);
        o.y = conv_epilogue(p, v.y, m, co + 1);
        o.z = conv_epilogue(p, v.z, m, co + 2);
        o.w = conv_epilogue(p, v.w, m, co + 3);
        *reinterpret_cast<f32x4*>(p.y + (size_t)m * p.ldy + co) = o;
    }
}

// ------------------------------------------------------------------------------------------
// tile configurations
// ------------------------------------------------------------------------------------------
typedef int (*launch_fn)(const ConvArgs&, dim3, hipStream_t);

template <int BM, int BN, int BK, int WM, int WN, int MT, bool KTAIL, bool K22, bool DMA, int NSTAGE, bool F16 = false,
          bool X3 = false, int KWR = 0>
static int launch_one(const ConvArgs& a, dim3 grid, hipStream_t st) {
    // fp16 tiles: the fused epilogue parks one MT-row block per wave ([MT][BN / WN + 4] floats) in the tile buffers
    constexpr size_t lds_tiles = (size_t)NSTAGE * ((KWR ? (BM + BM / 4 + 15) / 16 * 16 : BM) + (KWR ? KWR : 1) * BN) * (DMA ? BK : BK + 4) * sizeof(float);
    constexpr size_t lds_epi = F16 ? (size_t)WM * WN * MT * (BN / WN + 4) * sizeof(float) : 0;
    constexpr size_t lds = lds_tiles > lds_epi ? lds_tiles : lds_epi;
    auto kern = conv_igemm_kernel<BM, BN, BK, WM, WN, MT, KTAIL, K22, DMA, NSTAGE, F16, X3, KWR>;
    static bool attr_set[64] = {};   // per device; benign race (idempotent call)
    int dev = 0;
    PTX_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        PTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, st, a);
    return hip_check(hipGetLastError(), "conv_igemm launch");
}

// KTAIL instantiation only when the K extent of either operand is not a multiple of BK
template <int BM, int BN, int BK, int WM, int WN, int MT, bool F16, bool X3, int KWR>
static int launch_cfg_kwr(const ConvArgs& a, dim3 grid, hipStream_t st) {
    if ((a.kA % BK) || (a.kB % BK))
        return launch_one<BM, BN, BK, WM, WN, MT, true, false, true, 2, F16, X3, KWR>(a, grid, st);
    return launch_one<BM, BN, BK, WM, WN, MT, false, false, true, 2, F16, X3, KWR>(a, grid, st);
}

template <int BM, int BN, int BK, int WM, int WN, int MT, bool DMA, int NSTAGE, bool F16 = false, bool X3 = false>
static int launch_cfg(const ConvArgs& a, dim3 grid, hipStream_t st) {
    if constexpr (X3) {
        if ((a.kA % BK) || (a.kB % BK) || (a.dual && ((a.kA2 % BK) || (a.wcol2 % BK))))
            return launch_one<BM, BN, BK, WM, WN, MT, true, false, true, NSTAGE, false, true>(a, grid, st);
        return launch_one<BM, BN, BK, WM, WN, MT, false, false, true, NSTAGE, false, true>(a, grid, st);
    }
    if constexpr (F16) {
        if ((a.kA % BK) || (a.kB % BK))
            return launch_one<BM, BN, BK, WM, WN, MT, true, false, DMA, NSTAGE, true>(a, grid, st);
        return launch_one<BM, BN, BK, WM, WN, MT, false, false, DMA, NSTAGE, true>(a, grid, st);
    }
    if constexpr (BK == 24 && MT == 32 && !DMA) {
        // kW-folded stem: one 24-wide chunk per tap of which at most 22 columns are live
        if (a.k_live <= 22 && a.kA == 24 && a.kB == 24)
            return launch_one<BM, BN, BK, WM, WN, MT, false, true, false, 2>(a, grid, st);
    }
    if constexpr (BK == 32 && MT == 32 && DMA && NSTAGE == 2 && BN == 64) {
        // kW-folded stem on 32-float rows (fold ld = 32): LDS-DMA staging, 11 MFMAs per tap as on the BK = 24 tiles
        if (a.k_live <= 22 && a.kA == 32 && a.kB == 32 && !a.dual && a.groups <= 1)
            return launch_one<BM, BN, BK, WM, WN, MT, false, true, true, 2>(a, grid, st);
    }
    if ((a.kA % BK) || (a.kB % BK) || (a.dual && ((a.kA2 % BK) || (a.wcol2 % BK))))
        return launch_one<BM, BN, BK, WM, WN, MT, true, false, DMA, NSTAGE>(a, grid, st);
    return launch_one<BM, BN, BK, WM, WN, MT, false, false, DMA, NSTAGE>(a, grid, st);
}

// ------------------------------------------------------------------------------------------
// Direct (VALU) convolution for NARROW outputs (Co <= 16: SlowFast's fast pathway, lateral convs,
// Inception reduction branches).  On the MFMA tiles above an 8-channel output wastes 3/4 .. 7/8 of
// every 32- or 64-wide N tile; the fp32 vector ALUs have half the matrix peak but waste nothing:
// one lane owns P consecutive output positions x CO channels in registers, streams its input
// pixels with 16-byte buffer loads (out-of-image taps read zero) and multiplies them with filter
// taps that are UNIFORM across the wave -- the filter never touches LDS or a VGPR-indexed load.
// Same operands (NDHWC input, K-major packed filter with BN folded) and the same epilogue as the
// implicit-GEMM kernel, so it is just another tile configuration for the tuner.
// ------------------------------------------------------------------------------------------
// wait for the scalar loads issued by inline asm; the in/out operands tie every consumer to this point
template <int CO> __device__ __forceinline__ void sload_fence(f32x4 (&w)[CO]);
template <> __device__ __forceinline__ void sload_fence<4>(f32x4 (&w)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(w[0]), "+s"(w[1]), "+s"(w[2]), "+s"(w[3]));
}
template <> __device__ __forceinline__ void sload_fence<8>(f32x4 (&w)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+s"(w[0]), "+s"(w[1]), "+s"(w[2]), "+s"(w[3]), "+s"(w[4]), "+s"(w[5]), "+s"(w[6]), "+s"(w[7]));
}
template <> __device__ __forceinline__ void sload_fence<16>(f32x4 (&w)[16]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+s"(w[0]), "+s"(w[1]), "+s"(w[2]), "+s"(w[3]), "+s"(w[4]), "+s"(w[5]), "+s"(w[6]), "+s"(w[7]),
                   "+s"(w[8]), "+s"(w[9]), "+s"(w[10]), "+s"(w[11]), "+s"(w[12]), "+s"(w[13]), "+s"(w[14]), "+s"(w[15]));
}

template <int CO, int P>
__global__ void __launch_bounds__(256) conv_direct_kernel(const ConvArgs p, const float* __restrict__ wq,
                                                          const float* __restrict__ bias, int segs_per_row,
                                                          long long total_segs, int seg_tiles) {
    const int st = blockIdx.x % seg_tiles, nt = blockIdx.x / seg_tiles;
    const int n0 = nt * CO;
    // grouped conv: this tile's CO output channels lie in one group and read only its input channels
    const unsigned xcol = p.groups > 1 ? (unsigned)((n0 / p.cog) * p.cig) * 4u : 0u;
    const long long seg = (long long)st * 256 + threadIdx.x;
    const bool seg_ok = seg < total_segs;
    const long long sg = seg_ok ? seg : 0;
    const int ws = (int)(sg % segs_per_row);
    long long r = sg / segs_per_row;
    const int ho = (int)(r % p.Ho);
    r /= p.Ho;
    const int to = (int)(r % p.To);
    const int n = (int)(r / p.To);
    const int wo0 = ws * P;
    constexpr unsigned kOOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    float acc[P][CO];
#pragma unroll
    for (int j = 0; j < P; ++j)
#pragma unroll
        for (int c = 0; c < CO; ++c) acc[j][c] = 0.f;
    const int kext = (p.k_live + 3) & ~3;                 // live K columns per tap, rounded to the 16-byte loads
    for (int kt = 0; kt < p.kT; ++kt) {
        const int ti = to * p.sT - p.pT + kt;
        const bool t_ok = seg_ok && ti >= 0 && ti < p.Ti;
        for (int kh = 0; kh < p.kH; ++kh) {
            const int hi = ho * p.sH - p.pH + kh;
            const bool h_ok = t_ok && hi >= 0 && hi < p.Hi;
            const unsigned rowpos = (unsigned)((n * p.Ti + ti) * p.Hi + hi) * (unsigned)p.Wi;
            for (int kw = 0; kw < p.kW; ++kw) {
                const int tap = (kt * p.kH + kh) * p.kW + kw;
                unsigned off[P];
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    const int wi = (wo0 + j) * p.sW - p.pW + kw;
                    const bool ok = h_ok && wi >= 0 && wi < p.Wi && (wo0 + j) < p.Wo;
                    off[j] = ok ? (rowpos + (unsigned)wi) * (unsigned)p.ldx * 4u + xcol : kOOB;
                }
                // filter taps are wave-uniform: fetch them with SCALAR loads (s_load_dwordx4 -> SGPRs).  As vector
                // loads of one address they made the kernel texture-addresser bound (6 ms for the fast stem).
                const float* wt = wq + ((size_t)tap * p.w_rows + n0) * p.ldw;      // uniform
                for (int k = 0; k < kext; k += 4) {
                    f32x4 a[P];
#pragma unroll
                    for (int j = 0; j < P; ++j)
                        a[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                             rsrc_x, off[j] == kOOB ? kOOB : off[j] + (unsigned)k * 4u, 0, 0));
                    f32x4 wreg[CO];
#pragma unroll
                    for (int c = 0; c < CO; ++c) {
                        const unsigned boff = __builtin_amdgcn_readfirstlane((unsigned)(c * p.ldw + k) * 4u);
                        asm volatile("s_load_dwordx4 %0, %1, %2" : "=s"(wreg[c]) : "s"(wt), "s"(boff));
                    }
                    sload_fence<CO>(wreg);
#pragma unroll
                    for (int c = 0; c < CO; ++c) {
                        const f32x4 wv = wreg[c];
#pragma unroll
                        for (int j = 0; j < P; ++j) {
                            float v = acc[j][c];
                            v = fmaf(a[j].x, wv.x, v);
                            v = fmaf(a[j].y, wv.y, v);
                            v = fmaf(a[j].z, wv.z, v);
                            v = fmaf(a[j].w, wv.w, v);
                            acc[j][c] = v;
                        }
                    }
                }
            }
        }
    }
    if (!seg_ok) return;
    const bool relu = (p.flags & PTX_EPI_RELU) != 0, res_add = (p.flags & PTX_EPI_RES_ADD) != 0;
    const long long m0 = (((long long)n * p.To + to) * p.Ho + ho) * p.Wo + wo0;
#pragma unroll
    for (int j = 0; j < P; ++j) {
        if (wo0 + j >= p.Wo) break;
        float* yrow = p.y + (size_t)(m0 + j) * p.ldy + n0;
        const float* rrow = res_add ? p.res + (size_t)(m0 + j) * p.ldr + n0 : nullptr;
#pragma unroll
        for (int c = 0; c < CO; c += 4) {
            if (n0 + c >= p.ncol) break;
            f32x4 o = {acc[j][c], acc[j][c + 1], acc[j][c + 2], acc[j][c + 3]};
            if (bias) o += *reinterpret_cast<const f32x4*>(bias + n0 + c);
            if (res_add) o += *reinterpret_cast<const f32x4*>(rrow + c);
            if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
            *reinterpret_cast<f32x4*>(yrow + c) = o;
        }
    }
}

template <int CO, int P>
static int launch_direct(const ConvArgs& a, dim3 grid, hipStream_t st) {
    if (a.dual || a.bs_x || a.bs_w || a.bs_y || grid.y != 1)
        return fail(PTX_ERR_UNSUPPORTED, "direct conv: dual-source / batched-GEMM launches use the MFMA tiles");
    if (a.flags & PTX_EPI_RES_PADA) return fail(PTX_ERR_UNSUPPORTED, "direct conv: shortcut-A / upsampled residuals use the MFMA tiles");
    if ((a.flags & PTX_EPI_RES_ADD) && (a.ldr % 4 || ((uintptr_t)a.res & 15)))
        return fail(PTX_ERR_UNSUPPORTED, "direct conv: misaligned residual");
    if (a.ldy % 4 || a.ldw % 4) return fail(PTX_ERR_UNSUPPORTED, "direct conv: misaligned rows");
    if (a.groups > 1 && (a.cog % CO || a.cig % 4))
        return fail(PTX_ERR_UNSUPPORTED, "direct conv: %d-channel tiles do not divide the %d output channels of a group",
                    CO, a.cog);
    const int segs_per_row = cdiv(a.Wo, P);
    const long long total = (long long)a.N * a.To * a.Ho * segs_per_row;
    const long long seg_tiles = (total + 255) / 256;
    const long long blocks = seg_tiles * cdiv(a.ncol, CO);
    if (blocks > 0x7fffffffLL) return fail(PTX_ERR_INVALID, "direct conv: grid too large");
    hipLaunchKernelGGL((conv_direct_kernel<CO, P>), dim3((unsigned)blocks), dim3(256), 0, st, a, a.w, a.bias, segs_per_row,
                       total, (int)seg_tiles);
    return hip_check(hipGetLastError(), "conv_direct launch");
}

struct ConvConfig {
    int BM, BN, BK, WM, WN, MT;
    const char* name;
    launch_fn launch;
    bool direct;      // VALU kernel: no split-K, own grid
    bool f16;         // fp16 operands (PTX_F16_OPERANDS)
    bool x3;          // split fp32 operands on the fp16 matrix cores (PTX_F16X3_OPERANDS)
    int kwr;          // kw-reuse tile (3-wide stride-1 filters, BM a whole number of output rows)
};

#define PTX_CFG(BM, BN, BK, WM, WN, MT) \
    { BM, BN, BK, WM, WN, MT, #BM "x" #BN "x" #BK "/" #WM "x" #WN "/m" #MT, launch_cfg<BM, BN, BK, WM, WN, MT, false, 2>, false, false, false, 0 }
#define PTX_CFG_DMA(BM, BN, BK, WM, WN, MT) \
    { BM, BN, BK, WM, WN, MT, #BM "x" #BN "x" #BK "/" #WM "x" #WN "/m" #MT "/dma", launch_cfg<BM, BN, BK, WM, WN, MT, true, 2>, false, false, false, 0 }
#define PTX_CFG_DMA3(BM, BN, BK, WM, WN, MT) \
    { BM, BN, BK, WM, WN, MT, #BM "x" #BN "x" #BK "/" #WM "x" #WN "/m" #MT "/dma3", launch_cfg<BM, BN, BK, WM, WN, MT, true, 3>, false, false, false, 0 }
#define PTX_CFG_DMA4(BM, BN, BK, WM, WN, MT) \
    { BM, BN, BK, WM, WN, MT, #BM "x" #BN "x" #BK "/" #WM "x" #WN "/m" #MT "/dma4", launch_cfg<BM, BN, BK, WM, WN, MT, true, 4>, false, false, false, 0 }

#define PTX_CFG_DIRECT(BM, BN, BK, CO, P) \
    { BM, BN, BK, 4, 1, 0, #BM "x" #BN "x" #BK "/direct", launch_direct<CO, P>, true, false, false, 0 }
#define PTX_CFG_F16(BM, BN, BK, WM, WN, MT) \
    { BM, BN, BK, WM, WN, MT, #BM "x" #BN "x" #BK "/" #WM "x" #WN "/m" #MT "/dma/f16", \
      launch_cfg<BM, BN, BK, WM, WN, MT, true, 2, true>, false, true, false, 0 }
#define PTX_CFG_X3(BM, BN, BK, WM, WN, MT) \
    { BM, BN, BK, WM, WN, MT, #BM "x" #BN "x" #BK "/" #WM "x" #WN "/m" #MT "/dma/x3", \
      launch_cfg<BM, BN, BK, WM, WN, MT, true, 2, false, true>, false, false, true, 0 }
#define PTX_CFG_KWR_X3(BM, BN, BK, WM, WN, MT) \
    { BM, BN, BK, WM, WN, MT, #BM "x" #BN "x" #BK "/" #WM "x" #WN "/m" #MT "/dma/kwr/x3", \
      launch_cfg_kwr<BM, BN, BK, WM, WN, MT, false, true, 3>, false, false, true, 3 }
#define PTX_CFG_KWR_F16(BM, BN, BK, WM, WN, MT) \
    { BM, BN, BK, WM, WN, MT, #BM "x" #BN "x" #BK "/" #WM "x" #WN "/m" #MT "/dma/kwr/f16", \
      launch_cfg_kwr<BM, BN, BK, WM, WN, MT, true, false, 3>, false, true, false, 3 }
#define PTX_CFG_X3R(BM, BN, BK, WM, WN, MT, NS) \
    { BM, BN, BK, WM, WN, MT, #BM "x" #BN "x" #BK "/" #WM "x" #WN "/m" #MT "/dma" #NS "/x3", \
      launch_cfg<BM, BN, BK, WM, WN, MT, true, NS, false, true>, false, false, true, 0 }

static const ConvConfig kConfigs[] = {
    PTX_CFG(128, 128, 32, 2, 2, 32),  // 0  large M, Co >= 128
    PTX_CFG(128, 64, 32, 2, 2, 32),   // 1  Co == 64
    PTX_CFG(64, 64, 32, 2, 2, 32),    // 2  small M
    PTX_CFG(256, 64, 32, 4, 1, 32),   // 3  Co == 64, tall
    PTX_CFG(64, 128, 32, 2, 2, 32),   // 4  small M, wide
    PTX_CFG(112, 64, 32, 1, 4, 16),   // 5  M = 2^k * 49 (7 x 16 rows), Co == 64
    PTX_CFG(128, 64, 24, 2, 2, 32),   // 6  kW-folded stem (K chunk = 24)
    PTX_CFG(256, 64, 24, 4, 1, 32),   // 7  kW-folded stem, tall
    PTX_CFG(112, 128, 32, 1, 4, 16),  // 8  M = 2^k * 49, Co >= 128
    PTX_CFG(64, 64, 16, 2, 2, 32),    // 9  ragged channel counts ((2+1)D), small K chunk
    PTX_CFG(128, 64, 16, 2, 2, 32),   // 10 ragged channel counts, larger M
    // 8-wave workgroups: big tiles (fewer bytes moved per MFMA) at 4 waves per SIMD
    PTX_CFG(128, 128, 32, 2, 4, 32),  // 11
    PTX_CFG(128, 128, 32, 4, 2, 32),  // 12
    PTX_CFG(128, 64, 32, 4, 2, 32),   // 13
    PTX_CFG(256, 64, 32, 8, 1, 32),   // 14
    PTX_CFG(256, 128, 32, 4, 2, 32),  // 15
    PTX_CFG(256, 64, 24, 8, 1, 32),   // 16 stem
    PTX_CFG(128, 64, 24, 4, 2, 32),   // 17 stem
    PTX_CFG(224, 64, 32, 7, 1, 32),   // 18 M = 2^k * 49 (7 x 32 rows), 7 waves
    PTX_CFG(32, 64, 32, 2, 2, 16),    // 19 small M: 32-row tiles (wave 16x32 on 16x16x4 MFMA)
    PTX_CFG(32, 128, 32, 2, 2, 16),   // 20 small M, wide
    PTX_CFG(128, 64, 16, 4, 2, 32),   // 21 short K, 8 waves
    PTX_CFG(128, 128, 16, 4, 2, 32),  // 22 short K, 8 waves, wide
    PTX_CFG(64, 128, 16, 2, 2, 32),   // 23 short K, wide
    // LDS-DMA staging (buffer_load ... lds, swizzled lane-linear tiles)
    PTX_CFG_DMA(64, 64, 32, 2, 2, 32),    // 24
    PTX_CFG_DMA(128, 64, 32, 4, 2, 32),   // 25
    PTX_CFG_DMA(128, 128, 32, 4, 2, 32),  // 26
    PTX_CFG_DMA(64, 128, 32, 2, 2, 32),   // 27
    PTX_CFG_DMA(64, 64, 16, 2, 2, 32),    // 28
    PTX_CFG_DMA(128, 64, 16, 2, 2, 32),   // 29
    PTX_CFG_DMA(32, 64, 32, 2, 2, 16),    // 30
    PTX_CFG_DMA(256, 64, 32, 8, 1, 32),   // 31
    PTX_CFG_DMA(64, 128, 16, 2, 2, 32),   // 32
    PTX_CFG_DMA(128, 128, 16, 4, 2, 32),  // 33
    PTX_CFG_DMA(128, 64, 16, 4, 2, 32),   // 34
    PTX_CFG_DMA(32, 128, 32, 2, 2, 16),   // 35
    PTX_CFG_DMA(112, 64, 32, 1, 4, 16),   // 36
    PTX_CFG_DMA(64, 32, 32, 2, 2, 16),    // 37 (wave 32x16)
    PTX_CFG_DMA(32, 32, 32, 2, 2, 16),    // 38 tiny tiles for very small M
    // 48 / 96-wide N tiles: the (2+1)D mid-channel counts are 144 * 2^k (r2plus1d.py:68-69)
    PTX_CFG_DMA(64, 48, 32, 4, 1, 16),    // 39
    PTX_CFG_DMA(128, 48, 32, 8, 1, 16),   // 40
    PTX_CFG_DMA(64, 96, 32, 2, 2, 16),    // 41
    PTX_CFG_DMA(128, 96, 32, 4, 2, 16),   // 42
    PTX_CFG_DMA(32, 96, 32, 2, 2, 16),    // 43
    // 3-stage DMA ring (tiles requested two k-steps ahead, counted vmcnt + raw barrier)
    PTX_CFG_DMA3(64, 64, 16, 2, 2, 32),   // 44
    PTX_CFG_DMA3(64, 64, 32, 2, 2, 32),   // 45
    PTX_CFG_DMA3(128, 64, 32, 4, 2, 32),  // 46
    PTX_CFG_DMA3(32, 64, 32, 2, 2, 16),   // 47
    PTX_CFG_DMA3(128, 128, 32, 4, 2, 32), // 48
    PTX_CFG_DMA3(64, 128, 16, 2, 2, 32),  // 49
    PTX_CFG_DMA3(128, 64, 16, 2, 2, 32),  // 50
    // 4-stage rings for small grids (layer3 / layer4: <= 2 workgroups per CU, latency bound per k-step)
    PTX_CFG_DMA4(32, 64, 32, 2, 2, 16),   // 51
    PTX_CFG_DMA4(64, 64, 32, 2, 2, 32),   // 52
    PTX_CFG_DMA4(32, 128, 32, 2, 2, 16),  // 53
    PTX_CFG_DMA4(64, 64, 16, 2, 2, 32),   // 54
    PTX_CFG_DMA4(64, 128, 32, 2, 2, 32),  // 55
    // BK = 64: half as many barriers per MFMA for long-K problems on small grids
    PTX_CFG_DMA(64, 64, 64, 2, 2, 32),    // 56
    PTX_CFG_DMA(32, 64, 64, 2, 2, 16),    // 57
    PTX_CFG_DMA(64, 128, 64, 2, 2, 32),   // 58
    PTX_CFG_DMA(32, 128, 64, 2, 2, 16),   // 59
    PTX_CFG_DMA(128, 64, 64, 4, 2, 32),   // 60
    // direct VALU kernels for narrow outputs: <rows per workgroup> x <channels> x <K granule>; the "x24" names are
    // offered to the kW-folded stems (K chunk 24), the "x4" names to everything else
    PTX_CFG_DIRECT(1024, 8, 24, 8, 4),    // 61 fast-pathway stem (3 -> 8): 4 positions x 8 channels per lane
    PTX_CFG_DIRECT(512, 8, 24, 8, 2),     // 62
    PTX_CFG_DIRECT(1024, 8, 4, 8, 4),     // 63
    PTX_CFG_DIRECT(512, 8, 4, 8, 2),      // 64
    PTX_CFG_DIRECT(512, 16, 4, 16, 2),    // 65
    PTX_CFG_DIRECT(256, 16, 4, 16, 1),    // 66
    PTX_CFG_DIRECT(256, 8, 4, 8, 1),      // 67
    // 16-wide N tiles on 16x16x4 MFMA for narrow outputs (Co <= 16): 4x less padded work than a 64-wide tile
    PTX_CFG(256, 16, 32, 8, 1, 16),       // 68
    PTX_CFG(128, 16, 32, 4, 1, 16),       // 69
    PTX_CFG(256, 32, 32, 8, 1, 16),       // 70
    PTX_CFG(128, 32, 32, 4, 1, 16),       // 71
    PTX_CFG_DIRECT(1024, 4, 4, 4, 4),     // 72 group width 4 (ResNeXt3D layer1, cardinality 32)
    PTX_CFG_DIRECT(512, 4, 4, 4, 2),      // 73
    // fp16 operands (BigGAN generator, config 5): LDS-DMA tiles; BK counts 32-bit words = channel pairs
    PTX_CFG_F16(128, 128, 32, 4, 2, 32),  // 74
    PTX_CFG_F16(128, 64, 32, 4, 2, 32),   // 75
    PTX_CFG_F16(64, 64, 32, 2, 2, 32),    // 76
    PTX_CFG_F16(64, 128, 32, 2, 2, 32),   // 77
    PTX_CFG_F16(128, 128, 16, 4, 2, 32),  // 78
    PTX_CFG_F16(64, 64, 16, 2, 2, 32),    // 79
    PTX_CFG_F16(32, 64, 32, 2, 2, 16),    // 80
    PTX_CFG_F16(64, 32, 32, 2, 2, 16),    // 81 narrow outputs (the 3-channel image conv)
    PTX_CFG_F16(256, 128, 32, 4, 2, 32),  // 82
    // bigger per-wave tiles (64x64: 4 fragment reads feed 4 MFMAs) for the LDS-read-bound generator convs
    PTX_CFG_F16(128, 128, 32, 2, 2, 32),  // 83
    PTX_CFG_F16(256, 64, 32, 4, 1, 32),   // 84 hidden width 64 (the 128^2 / 256^2 stages)
    PTX_CFG_F16(128, 64, 32, 2, 2, 32),   // 85
    PTX_CFG_F16(256, 16, 32, 8, 1, 16),   // 86 the 3-channel image conv: 16-wide N on 16x16x32 MFMA
    PTX_CFG_F16(128, 16, 32, 4, 1, 16),   // 87
    // split fp32 operands (x3): 3 f16 MFMAs per 16 k; BK 32 / 64 on 32x32x16, BK 64 on 16x16x32
    PTX_CFG_X3(128, 128, 32, 4, 2, 32),   // 88
    PTX_CFG_X3(128, 64, 32, 4, 2, 32),    // 89
    PTX_CFG_X3(256, 64, 32, 8, 1, 32),    // 90 stem (folded rows of 32 floats)
    PTX_CFG_X3(64, 64, 32, 2, 2, 32),     // 91
    PTX_CFG_X3(64, 128, 32, 2, 2, 32),    // 92
    PTX_CFG_X3(64, 64, 64, 2, 2, 32),     // 93
    PTX_CFG_X3(32, 64, 64, 2, 2, 16),     // 94 small M
    PTX_CFG_X3(32, 128, 64, 2, 2, 16),    // 95 small M, wide
    PTX_CFG_X3(128, 128, 32, 2, 2, 32),   // 96 64x64 per wave: 8 fragment reads feed 12 MFMAs
    PTX_CFG_X3(256, 128, 32, 4, 2, 32),   // 97
    PTX_CFG_X3(128, 64, 32, 2, 2, 32),    // 98
    PTX_CFG_X3(64, 32, 64, 2, 2, 16),     // 99
    PTX_CFG_X3(256, 64, 32, 4, 1, 32),    // 100 stem, 64x64 per wave
    PTX_CFG_X3(128, 64, 64, 4, 2, 32),    // 101
    // deeper LDS-DMA rings: with the matrix work per k-step 5x shorter, one k-step of prefetch no longer covers L2 / HBM latency
    PTX_CFG_X3R(256, 64, 32, 8, 1, 32, 3),   // 102
    PTX_CFG_X3R(256, 64, 32, 4, 1, 32, 3),   // 103
    PTX_CFG_X3R(128, 64, 32, 4, 2, 32, 3),   // 104
    PTX_CFG_X3R(128, 64, 32, 4, 2, 32, 4),   // 105
    PTX_CFG_X3R(128, 128, 32, 4, 2, 32, 3),  // 106
    PTX_CFG_X3R(64, 128, 32, 2, 2, 32, 3),   // 107
    PTX_CFG_X3R(64, 64, 32, 2, 2, 32, 4),    // 108
    PTX_CFG_X3R(128, 64, 32, 2, 2, 32, 4),   // 109
    PTX_CFG_X3R(32, 64, 64, 2, 2, 16, 3),    // 110
    PTX_CFG_X3R(128, 128, 32, 2, 2, 32, 3),  // 111
    // kw-reuse tiles: 3-wide stride-1 filters, BM = whole output rows (224 = 4 x 56 = 8 x 28 = 16 x 14; 128 / 256 for 2^k widths)
    PTX_CFG_KWR_X3(224, 64, 32, 7, 1, 32),    // 112
    PTX_CFG_KWR_X3(224, 64, 16, 7, 1, 32),    // 113
    PTX_CFG_KWR_X3(224, 128, 16, 7, 1, 32),   // 114
    PTX_CFG_KWR_X3(128, 64, 32, 4, 2, 32),    // 115
    PTX_CFG_KWR_X3(112, 64, 32, 1, 4, 16),    // 116
    PTX_CFG_KWR_X3(256, 64, 32, 8, 1, 32),    // 117
    PTX_CFG_KWR_X3(256, 64, 16, 8, 1, 32),    // 118
    PTX_CFG_KWR_F16(256, 64, 32, 8, 1, 32),   // 119
    PTX_CFG_KWR_F16(256, 64, 16, 8, 1, 32),   // 120
    PTX_CFG_KWR_F16(256, 64, 32, 4, 1, 32),   // 121
    PTX_CFG_KWR_F16(128, 128, 32, 4, 2, 32),  // 122
    PTX_CFG_KWR_F16(256, 128, 16, 4, 2, 32),  // 123
    PTX_CFG_KWR_F16(128, 64, 32, 4, 2, 32),   // 124
    PTX_CFG_KWR_F16(256, 16, 32, 8, 1, 16),   // 125 the 3-channel image conv
    PTX_CFG_KWR_F16(128, 128, 16, 4, 2, 32),  // 126
};
constexpr int kNumConfigs = sizeof(kConfigs) / sizeof(kConfigs[0]);

// Default tiles are resolved BY NAME (once): inserting or reordering kConfigs entries can then never make a heuristic
// pick a tile of the wrong operand kind (an fp32 tile for an x3-packed filter computes garbage without an error).
// A name this build does not have is a build defect: it aborts at the first lookup instead of returning a wrong tile.
static int cfg_named(const char* name) {
    for (int i = 0; i < kNumConfigs; ++i)
        if (!strcmp(kConfigs[i].name, name)) return i;
    fprintf(stderr, "libptx_amd: default tile configuration \"%s\" is not compiled into this build\n", name);
    abort();
}
#define PTX_TILE(name) ([]() -> int { static const int idx = cfg_named(name); return idx; }())

static int validate_desc(const ptx_conv3d_desc* d) {
    if (!d) return fail(PTX_ERR_INVALID, "conv3d: null descriptor");
    if (d->N <= 0 || d->Ti <= 0 || d->Hi <= 0 || d->Wi <= 0 || d->To <= 0 || d->Ho <= 0 || d->Wo <= 0 ||
        d->Ci <= 0 || d->Co <= 0)
        return fail(PTX_ERR_INVALID, "conv3d: non-positive extent");
    if (d->ldx < d->Ci || d->ldx % 4 || d->ldy < d->Co || d->ldy % 4)
        return fail(PTX_ERR_INVALID, "conv3d: channel strides must be >= C and multiples of 4 (ldx=%d ldy=%d)",
                    d->ldx, d->ldy);
    if (d->kT <= 0 || d->kH <= 0 || d->kW <= 0 || d->sT <= 0 || d->sH <= 0 || d->sW <= 0 || d->pT < 0 ||
        d->pH < 0 || d->pW < 0)
        return fail(PTX_ERR_INVALID, "conv3d: bad filter geometry");
    if (d->kT > 8 || d->kH > 8 || d->kW > 8)
        return fail(PTX_ERR_UNSUPPORTED, "conv3d: filter extents above 8 are not supported (%d,%d,%d)", d->kT, d->kH, d->kW);
    const int groups = d->groups > 1 ? d->groups : 1;
    if (d->groups < 0 || d->Ci % groups || d->Co % groups || (groups > 1 && (d->Ci / groups) % 4))
        return fail(PTX_ERR_INVALID, "conv3d: groups=%d must divide Ci=%d and Co=%d, with Ci/groups a multiple of 4",
                    d->groups, d->Ci, d->Co);
    if (d->Kc < d->Ci / groups || d->Kc % 4 || d->Co_pad < (d->Co + 3) / 4 * 4)
        return fail(PTX_ERR_INVALID, "conv3d: packed weight extents Kc=%d Co_pad=%d do not cover Ci/groups=%d Co=%d",
                    d->Kc, d->Co_pad, d->Ci / groups, d->Co);
    // output extent must match the conv arithmetic: symmetric padding p, or TF-"SAME" (out = ceil(in/stride),
    // p = the FRONT pad floor(total/2); the back pad is implied -- out-of-range taps read zero either way)
    auto extent_ok = [](int in, int out, int k, int s, int p) {
        if (out == (in + 2 * p - k) / s + 1) return true;
        const int same = (in + s - 1) / s;
        const int total = std::max((same - 1) * s + k - in, 0);
        return out == same && p == total / 2;
    };
    if (!extent_ok(d->Ti, d->To, d->kT, d->sT, d->pT) || !extent_ok(d->Hi, d->Ho, d->kH, d->sH, d->pH) ||
        !extent_ok(d->Wi, d->Wo, d->kW, d->sW, d->pW))
        return fail(PTX_ERR_INVALID, "conv3d: output extent (%d,%d,%d) matches neither symmetric padding (%d,%d,%d) "
                    "nor SAME geometry", d->To, d->Ho, d->Wo, d->pT, d->pH, d->pW);
    if ((int64_t)d->N * d->To * d->Ho * d->Wo > 0x7fffffffLL || (int64_t)d->N * d->Ti * d->Hi * d->Wi > 0x7fffffffLL)
        return fail(PTX_ERR_INVALID, "conv3d: more than 2^31 positions");
    if ((d->flags & PTX_EPI_RES_ADD) && (d->flags & PTX_EPI_RES_PADA))
        return fail(PTX_ERR_INVALID, "conv3d: RES_ADD and RES_PADA are exclusive");
    if ((d->flags & (kFusedEpiFlags | PTX_PRO_UP2)) && !(d->flags & PTX_F16_OPERANDS))
        return fail(PTX_ERR_UNSUPPORTED, "conv3d: the fused generator-stage flags (halfs out, affine, dual output, half skip, "
                    "tanh, upsampling loader) run on the fp16-operand tiles only");
    if (d->flags & PTX_PRO_UP2) {
        if (d->kT != 1 || d->Ti != 1 || d->sH != 1 || d->sW != 1 || (d->Hi & 1) || (d->Wi & 1) || d->groups > 1)
            return fail(PTX_ERR_INVALID, "conv3d: PTX_PRO_UP2 needs a unit-stride 2-D conv over even (upsampled) extents");
    }
    if ((d->flags & PTX_EPI_RELU) && (d->flags & PTX_EPI_TANH)) return fail(PTX_ERR_INVALID, "conv3d: RELU and TANH are exclusive");
    if (d->flags & PTX_F16X3_OPERANDS) {
        if (d->flags & PTX_F16_OPERANDS) return fail(PTX_ERR_INVALID, "conv3d: PTX_F16_OPERANDS and PTX_F16X3_OPERANDS are exclusive");
        if (d->Kc % 8 || d->groups > 1)
            return fail(PTX_ERR_INVALID, "conv3d: split operands (PTX_F16X3_OPERANDS) need a dense conv whose filter rows hold "
                        "whole 8-channel (hi8 | lo8) blocks (Kc %% 8 == 0, got %d)", d->Kc);
    }
    return PTX_OK;
}

}  // namespace ptx

using namespace ptx;

extern "C" int ptx_conv3d_num_configs(void) { return kNumConfigs; }

extern "C" const char* ptx_conv3d_config_name(int config) {
    if (config < 0 || config >= kNumConfigs) return "invalid";
    return kConfigs[config].name;
}

extern "C" int ptx_conv3d_config_supported(const ptx_conv3d_desc* d, int config) {
    if (validate_desc(d) != PTX_OK || config < 0 || config >= kNumConfigs) return 0;
    return 1;   // K / M / N tails are all guarded inside the kernel
}

extern "C" int ptx_conv3d_pick_config(const ptx_conv3d_desc* d, int* split_k) {
    // Defaults distilled from Engine.autotune runs on MI355X (profiles/r01_*); the tuner refines them.
    if (split_k) *split_k = 1;
    if (validate_desc(d) != PTX_OK) return 0;
    const int64_t M = (int64_t)d->N * d->To * d->Ho * d->Wo;
    const int taps = d->kT * d->kH * d->kW;
    const int ncol = (d->Co + 3) / 4 * 4;
    int cfg;
    if (d->flags & PTX_F16_OPERANDS) {
        const int64_t Mrows = (int64_t)d->N * d->To * d->Ho * d->Wo;
        return ncol <= 16 ? PTX_TILE("256x16x32/8x1/m16/dma/f16") : ncol <= 32 ? PTX_TILE("64x32x32/2x2/m16/dma/f16")
               : (Mrows < 8192 ? PTX_TILE("32x64x32/2x2/m16/dma/f16")
                               : (ncol >= 128 ? PTX_TILE("128x128x32/4x2/m32/dma/f16") : PTX_TILE("128x64x32/4x2/m32/dma/f16")));
    }
    if (d->flags & PTX_F16X3_OPERANDS) {           // split operands: defaults; the tuner refines them
        if (M < 8192) cfg = ncol >= 128 ? PTX_TILE("32x128x64/2x2/m16/dma/x3") : PTX_TILE("32x64x64/2x2/m16/dma/x3");
        else if (ncol >= 128 && cdiv64(M, 128) * cdiv(ncol, 128) >= 2 * kNumCU) cfg = PTX_TILE("128x128x32/2x2/m32/dma/x3");
        else if (M >= 256 * 1024 && ncol <= 64) cfg = PTX_TILE("256x64x32/4x1/m32/dma/x3");
        else cfg = ncol >= 128 ? PTX_TILE("64x128x32/2x2/m32/dma/x3") : PTX_TILE("64x64x32/2x2/m32/dma/x3");
        const ConvConfig& c = kConfigs[cfg];
        const int64_t blocks = cdiv64(M, c.BM) * cdiv(ncol, c.BN);
        const int steps = taps * cdiv(d->Kc, c.BK);
        int sk = 1;
        if (blocks < 2 * kNumCU) {
            sk = (int)((3 * kNumCU + blocks - 1) / blocks);
            if (sk > 8) sk = 8;
            while (sk > 1 && steps / sk < 8) --sk;
        }
        if (split_k) *split_k = sk;
        return cfg;
    }
    if (d->groups > 1) {                           // grouped conv: direct tiles sized to the group's output width
        const int cog = d->Co / d->groups;
        if (cog % 32 == 0) return PTX_TILE("64x32x32/2x2/m16/dma");              // 64x32x32 MFMA tile inside one group
        return cog % 16 == 0 ? PTX_TILE("512x16x4/direct") : cog % 8 == 0 ? PTX_TILE("512x8x4/direct") : PTX_TILE("512x4x4/direct");
    }
    if (d->Kc == 24) {
        cfg = M >= 256 * 1024 ? PTX_TILE("256x64x24/8x1/m32") : PTX_TILE("128x64x24/2x2/m32");            // kW-folded stem: 8-wave 256x64x24 (register staged)
    } else if (ncol % 48 == 0 && ncol % 64 != 0) {
        cfg = PTX_TILE("64x48x32/4x1/m16/dma");                                  // (2+1)D mid widths 144 * 2^k: 48-wide N tiles
    } else if (M < 8192) {
        cfg = ncol >= 128 ? PTX_TILE("32x128x32/2x2/m16/dma") : PTX_TILE("32x64x32/2x2/m16/dma");             // small M: 32-row tiles on 16x16x4 MFMA (DMA)
    } else if (ncol >= 128 && taps * d->Kc > 128 && cdiv64(M, 128) * cdiv(ncol, 128) >= 4 * kNumCU) {
        cfg = PTX_TILE("128x128x32/4x2/m32/dma");                                  // wide output, big grid: 8-wave 128x128 (DMA)
    } else {
        cfg = PTX_TILE("64x64x16/2x2/m32/dma");                                  // default: 64x64x16 DMA tiles, 8 workgroups per CU
    }
    const ConvConfig& c = kConfigs[cfg];
    const int64_t blocks = cdiv64(M, c.BM) * cdiv(ncol, c.BN);
    const int steps = taps * cdiv(d->Kc, c.BK);
    int sk = 1;
    if (blocks < 2 * kNumCU) {
        sk = (int)((3 * kNumCU + blocks - 1) / blocks);
        if (sk > 8) sk = 8;
        while (sk > 1 && steps / sk < 12) --sk;
    }
    if (split_k) *split_k = sk;
    return cfg;
}

constexpr size_t kCounterBytes = 65536;       // PTX_SPLITK_FUSED: 16384 tile counters ahead of the partial slabs

extern "C" size_t ptx_conv3d_workspace_bytes(const ptx_conv3d_desc* d, int split_k) {
    if (!d || split_k <= 1) return 0;
    return ((d->flags & PTX_SPLITK_FUSED) ? kCounterBytes : 0) +
           (size_t)split_k * d->N * d->To * d->Ho * d->Wo * ((d->Co + 3) / 4 * 4) * sizeof(float);
}

namespace ptx {
int launch_conv(ConvArgs& a, int config, int split_k, int batch, void* workspace, size_t workspace_bytes,
                hipStream_t st) {
    const ConvConfig& c = kConfigs[config];
    if ((a.f16 != 0) != c.f16)
        return fail(PTX_ERR_UNSUPPORTED, "conv3d: fp16-operand problems run on the /f16 tile configurations only (and vice versa)");
    if ((a.x3 != 0) != c.x3)
        return fail(PTX_ERR_UNSUPPORTED, "conv3d: split-operand problems run on the /x3 tile configurations only (and vice versa)");
    if (c.kwr) {
        if (a.kW != c.kwr || a.sW != 1 || a.dual || batch > 1 || a.groups > 1 || a.Wo < 8 || c.BM % a.Wo ||
            a.Wo != a.Wi + 2 * a.pW - a.kW + 1)
            return fail(PTX_ERR_UNSUPPORTED, "conv3d: a kw-reuse tile needs a dense %d-wide stride-1 filter and a %d-row tile made of "
                        "whole output rows (Wo = %d >= 8)", c.kwr, c.BM, a.Wo);
        fastdiv_make((unsigned)(a.Wo + c.kwr - 1), a.dv_hw);
    }
    if (a.groups > 1 && !c.direct && (a.cog % c.BN || a.dual || batch > 1))
        return fail(PTX_ERR_UNSUPPORTED, "conv3d: an MFMA tile must divide the %d output channels of a group", a.cog);
    a.m_tiles = cdiv(a.M, c.BM);
    fastdiv_make((unsigned)a.Wo, a.dv_wo);      // every launch path (conv, dual, batched GEMM) decodes rows with these
    fastdiv_make((unsigned)a.Ho, a.dv_ho);
    fastdiv_make((unsigned)a.To, a.dv_to);
    {
        static int t_inner = -1;
        if (t_inner < 0) { const char* e = getenv("PTX_T_INNER"); t_inner = e ? atoi(e) : 0; }   // measured: no gain (stem is MFMA-bound), off by default
        const int plane = a.Ho * a.Wo;
        const int64_t in_clip_bytes = (int64_t)a.Ti * a.Hi * a.Wi * a.ldx * 4;
        // worth it only when one clip's input exceeds the per-XCD L2 (4 MiB) and the filter spans frames
        a.tiles_per_plane = (t_inner && a.kT > 1 && a.To > 1 && plane % c.BM == 0 && in_clip_bytes > (8 << 20))
                                ? plane / c.BM : 0;
    }
    a.ncol = (a.Co + 3) / 4 * 4;
    a.n_tiles = cdiv(a.ncol, c.BN);
    a.kchunks = cdiv(std::max(a.kA, a.kB), c.BK);
    if (a.dual) {
        a.kc1 = cdiv(a.kA, c.BK);
        a.kchunks = a.kc1 + cdiv(a.kA2, c.BK);
    }
    const int steps_total = a.kT * a.kH * a.kW * a.kchunks;
    if (split_k < 1) split_k = 1;
    if (split_k > steps_total) split_k = steps_total;
    if (batch > 1 || c.direct) split_k = 1;
    a.split_k = split_k;
    {
        const uint64_t yb = (uint64_t)a.M * a.ldy * ((a.flags & PTX_EPI_OUT_F16) ? 2ull : 4ull);
        uint64_t rb = 0;
        if (a.flags & PTX_EPI_RES_ADD) rb = (uint64_t)a.M * a.ldr * 4ull;
        if (yb >= 0x80000000ull || rb >= 0x80000000ull)
            return fail(PTX_ERR_UNSUPPORTED, "conv3d: output / residual of one launch must be < 2 GiB; split the batch");
        a.y_bytes = (unsigned)yb;
        a.r_bytes = (unsigned)rb;
        if (a.f16 && (a.flags & kFusedEpiFlags)) {           // fused epilogue: residual extent in its own element size
            const uint64_t esz = (a.flags & PTX_RES_F16) ? 2ull : 4ull;
            uint64_t rtot = 0;
            if (a.flags & PTX_EPI_RES_ADD) rtot = (uint64_t)a.M * a.ldr * esz;
            if (a.flags & PTX_EPI_RES_PADA) rtot = (uint64_t)a.N * a.res_T * a.res_H * a.res_W * a.ldr * esz;
            if (rtot >= 0x80000000ull) return fail(PTX_ERR_UNSUPPORTED, "conv3d: skip operand of one launch must be < 2 GiB");
            a.r_bytes = (unsigned)rtot;
        }
        if (split_k > 1) a.y_bytes = (unsigned)std::min<uint64_t>((uint64_t)a.M * a.ncol * 4ull, 0x7fffffffull);   // fp32 partial slab
    }
    a.partial = nullptr;
    a.counters = nullptr;
    a.unit_pointwise = (a.kT * a.kH * a.kW == 1 && a.sT == 1 && a.sH == 1 && a.sW == 1 && a.pT == 0 && a.pH == 0 &&
                        a.pW == 0 && a.Ti == a.To && a.Hi == a.Ho && a.Wi == a.Wo && !a.up2) ? 1 : 0;
    bool fused_reduce = false;
    if (split_k > 1) {
        const size_t head = (a.flags & PTX_SPLITK_FUSED) ? kCounterBytes : 0;
        const size_t need = head + (size_t)split_k * a.M * a.ncol * sizeof(float);
        if (!workspace || workspace_bytes < need)
            return fail(PTX_ERR_WORKSPACE, "conv3d: split_k=%d needs %zu workspace bytes, got %zu", split_k, need,
                        workspace_bytes);
        a.partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + head);
        // the fused generator-stage epilogue (halfs out, dual output ...) keeps the separate reduce kernel
        fused_reduce = head != 0 && (size_t)a.m_tiles * a.n_tiles * 4 <= kCounterBytes && !(a.flags & kFusedEpiFlags);
        if (fused_reduce) a.counters = static_cast<unsigned*>(workspace);
    }
    if ((int64_t)a.m_tiles * a.n_tiles > 0x7fffffffLL) return fail(PTX_ERR_INVALID, "conv3d: grid too large");
    dim3 grid((unsigned)(a.m_tiles * a.n_tiles), (unsigned)batch, (unsigned)split_k);
    int s = c.launch(a, grid, st);
    if (s != PTX_OK) return s;
    if (split_k > 1 && !fused_reduce) {
        const size_t total4 = (size_t)a.M * a.ncol / 4;
        unsigned blocks = (unsigned)std::min<size_t>((total4 + 255) / 256, (size_t)kNumCU * 8);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, a);
        return hip_check(hipGetLastError(), "splitk_reduce launch");
    }
    return PTX_OK;
}
}  // namespace ptx

namespace ptx {
// y[M][ldy] = relu?( x[M][ldx] . w[Nout][K]^T + b ) on the implicit-GEMM tiles: a Linear layer's weight [out][in] already IS
// the K-major "packed filter" of a 1x1x1 conv with one tap.  ptx_linear_fwd routes here once M leaves the GEMV class
// (the skinny kernel re-reads the weights per group of rows: BigGAN's 64-row class-conditional GEMMs ran at 0.1 TB/s).
int linear_gemm(const float* x, const float* w, const float* b, float* y, int M, int K, int Nout, int ldx, int ldy,
                unsigned flags, hipStream_t st) {
    ConvArgs a{};
    a.x = x; a.w = w; a.bias = b; a.res = nullptr; a.y = y;
    a.N = 1; a.Ti = 1; a.Hi = 1; a.Wi = M; a.ldx = ldx; a.kA = K;
    a.To = 1; a.Ho = 1; a.Wo = M; a.Co = Nout; a.ldy = ldy; a.k_live = K;
    a.kT = a.kH = a.kW = 1; a.sT = a.sH = a.sW = 1; a.pT = a.pH = a.pW = 0;
    a.ldw = K; a.kB = K; a.w_rows = Nout; a.w_tap_stride = 0;
    a.M = M; a.flags = flags & PTX_EPI_RELU;
    a.groups = 1; a.cig = K; a.cog = Nout; a.pps = M;
    a.x_bytes = (unsigned)((uint64_t)M * ldx * 4ull);
    a.w_bytes = (unsigned)((uint64_t)Nout * K * 4ull);
    const int config = M <= 32 ? PTX_TILE("32x64x32/2x2/m16/dma") : PTX_TILE("64x64x32/2x2/m32/dma");          // 32x64x32 / 64x64x32 LDS-DMA tiles
    return launch_conv(a, config, 1, 1, nullptr, 0, st);
}
}  // namespace ptx

static int conv3d_common(const ptx_conv3d_desc* d, const float* x, const float* x2, const float* w_packed,
                         const float* bias, const float* res, float* y, void* workspace, size_t workspace_bytes,
                         int config, int split_k, ptx_stream_t stream, const ptx_conv_fused_ext* ext = nullptr) {
    int s = validate_desc(d);
    if (s != PTX_OK) return s;
    if ((d->flags & (PTX_EPI_AFFINE | PTX_EPI_DUAL_RAW)) && !ext)
        return fail(PTX_ERR_INVALID, "conv3d: PTX_EPI_AFFINE / PTX_EPI_DUAL_RAW take their operands through ptx_conv3d_fused_fwd");
    if (!x || !w_packed || !y) return fail(PTX_ERR_INVALID, "conv3d: null tensor pointer");
    if ((d->flags & (PTX_EPI_RES_ADD | PTX_EPI_RES_PADA)) && !res)
        return fail(PTX_ERR_INVALID, "conv3d: residual flag set but res == NULL");
    if (((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)y | (uintptr_t)res | (uintptr_t)workspace) & 15)
        return fail(PTX_ERR_INVALID, "conv3d: pointers must be 16-byte aligned");
    if (config >= kNumConfigs) return fail(PTX_ERR_INVALID, "conv3d: config %d out of range", config);
    if (config < 0) {
        int sk = 1;
        config = ptx_conv3d_pick_config(d, &sk);
        if (split_k <= 0) split_k = sk;
    }
    if (split_k <= 0) split_k = 1;
    if (d->flags & PTX_PRO_RELU)
        return fail(PTX_ERR_UNSUPPORTED, "conv3d: PTX_PRO_RELU is only implemented by ptx_linear_fwd");
    if (d->flags & PTX_EPI_RES_ADD) {
        if (d->ldr < (d->Co + 3) / 4 * 4)
            return fail(PTX_ERR_INVALID, "conv3d: residual stride %d does not cover Co=%d", d->ldr, d->Co);
    }
    if ((d->flags & PTX_EPI_RES_UP) && !(d->flags & PTX_EPI_RES_PADA))
        return fail(PTX_ERR_INVALID, "conv3d: RES_UP modifies RES_PADA");
    if (d->flags & PTX_EPI_RES_PADA) {
        if (d->flags & PTX_EPI_RES_UP) {      // upsampled, channel-truncated skip: res has >= Co channels;
                                              // res_s* hold log2 of the upsampling factor (0..4)
            if (d->res_sT < 0 || d->res_sH < 0 || d->res_sW < 0 || d->res_sT > 4 || d->res_sH > 4 || d->res_sW > 4)
                return fail(PTX_ERR_INVALID, "conv3d: RES_UP takes log2 factors 0..4 in res_s*");
            if (d->res_C > d->ldr || d->res_C < d->Co || ((d->To - 1) >> d->res_sT) >= d->res_T ||
                ((d->Ho - 1) >> d->res_sH) >= d->res_H || ((d->Wo - 1) >> d->res_sW) >= d->res_W)
                return fail(PTX_ERR_INVALID, "conv3d: upsampled residual geometry out of range");
        } else if (d->res_C > d->ldr || d->res_C > d->Co || (d->To - 1) * d->res_sT >= d->res_T ||
                   (d->Ho - 1) * d->res_sH >= d->res_H || (d->Wo - 1) * d->res_sW >= d->res_W) {
            return fail(PTX_ERR_INVALID, "conv3d: shortcut-A residual geometry out of range");
        }
    }
    ConvArgs a{};
    a.x = x; a.w = w_packed; a.bias = bias; a.res = res; a.y = y;
    a.N = d->N; a.Ti = d->Ti; a.Hi = d->Hi; a.Wi = d->Wi; a.ldx = d->ldx; a.kA = d->ldx;
    a.To = d->To; a.Ho = d->Ho; a.Wo = d->Wo; a.Co = d->Co; a.ldy = d->ldy; a.k_live = d->Ci;
    a.kT = d->kT; a.kH = d->kH; a.kW = d->kW; a.sT = d->sT; a.sH = d->sH; a.sW = d->sW;
    a.pT = d->pT; a.pH = d->pH; a.pW = d->pW;
    a.ldw = d->Kc; a.kB = d->Kc; a.w_rows = d->Co_pad; a.w_tap_stride = (long long)d->Co_pad * d->Kc;
    a.M = d->N * d->To * d->Ho * d->Wo;
    a.flags = d->flags;
    {
        const int up = (d->flags & PTX_PRO_UP2) ? 2 : 1;       // the stored input is (Hi/2, Wi/2) behind an upsampling loader
        const uint64_t xb = (uint64_t)d->N * d->Ti * (d->Hi / up) * (d->Wi / up) * d->ldx * 4ull;
        const uint64_t wb = (uint64_t)d->kT * d->kH * d->kW * d->Co_pad * d->Kc * 4ull;
        if (xb >= 0x80000000ull || wb >= 0x80000000ull)
            return fail(PTX_ERR_UNSUPPORTED, "conv3d: input (%llu B) and packed filter (%llu B) must each be < 2 GiB "
                        "(32-bit buffer offsets); split the batch", (unsigned long long)xb, (unsigned long long)wb);
        a.x_bytes = (unsigned)xb;
        a.w_bytes = (unsigned)wb;
    }
    a.f16 = (d->flags & PTX_F16_OPERANDS) ? 1 : 0;
    a.x3 = (d->flags & PTX_F16X3_OPERANDS) ? 1 : 0;
    if (a.f16 && (x2 || d->groups > 1)) return fail(PTX_ERR_UNSUPPORTED, "conv3d: fp16 operands: single-source dense convs only");
    a.up2 = (d->flags & PTX_PRO_UP2) ? 1 : 0;
    a.Hp = a.up2 ? d->Hi / 2 : d->Hi;
    a.Wp = a.up2 ? d->Wi / 2 : d->Wi;
    a.pps = d->To * d->Ho * d->Wo;
    if (d->flags & PTX_EPI_OUT_F16) {
        if (d->Co % 2 || d->ldy % 8 || ((uintptr_t)y & 15))
            return fail(PTX_ERR_INVALID, "conv3d: PTX_EPI_OUT_F16 needs an even Co and ldy (halfs) %% 8 == 0");
    }
    if (d->flags & PTX_EPI_AFFINE) {
        if (!ext->scale || !ext->shift || ext->ld_affine < d->Co || ext->ld_affine % 4 ||
            (((uintptr_t)ext->scale | (uintptr_t)ext->shift) & 15))
            return fail(PTX_ERR_INVALID, "conv3d: PTX_EPI_AFFINE needs 16-byte aligned scale / shift tables, ld_affine >= Co, %% 4 == 0");
        a.aff_scale = ext->scale; a.aff_shift = ext->shift; a.ld_aff = ext->ld_affine;
        a.aff_bytes = (unsigned)std::min<uint64_t>(((uint64_t)(d->N - 1) * ext->ld_affine + d->Co) * 4ull, 0x7fffffffull);
    }
    if ((d->flags & PTX_RES_F16) && (d->ldr % 8 || ((uintptr_t)res & 15)))
        return fail(PTX_ERR_INVALID, "conv3d: a half-precision skip operand needs ldr (halfs) %% 8 == 0");
    if (d->flags & PTX_EPI_DUAL_RAW) {
        const int nc = (d->Co + 3) / 4 * 4;
        if (!ext->y_raw || ext->ld_raw < nc || ext->ld_raw % 8 || d->Co % 2 || ((uintptr_t)ext->y_raw & 15))
            return fail(PTX_ERR_INVALID, "conv3d: PTX_EPI_DUAL_RAW needs y_raw (halfs) with ld_raw >= round_up(Co, 4), ld_raw %% 8 == 0");
        const uint64_t rawb = (uint64_t)d->N * d->To * d->Ho * d->Wo * ext->ld_raw * 2ull;
        if (rawb >= 0x80000000ull) return fail(PTX_ERR_UNSUPPORTED, "conv3d: y_raw of one launch must be < 2 GiB");
        a.y_raw = ext->y_raw; a.ld_raw = ext->ld_raw; a.raw_bytes = (unsigned)rawb;
    }
    a.groups = d->groups > 1 ? d->groups : 1;
    a.cig = d->Ci / a.groups; a.cog = d->Co / a.groups;
    if (a.groups > 1) {
        if (x2) return fail(PTX_ERR_INVALID, "conv3d_dual: grouped convolutions have one source");
        a.k_live = a.cig;
        a.kA = a.cig;          // K extent of the A operand per tap: the group's input channels only
    }
    a.ldr = d->ldr; a.res_C = d->res_C; a.res_T = d->res_T; a.res_H = d->res_H; a.res_W = d->res_W;
    a.res_sT = d->res_sT; a.res_sH = d->res_sH; a.res_sW = d->res_sW;
    if (x2) {
        if (d->kT * d->kH * d->kW != 1 || d->sT != 1 || d->sH != 1 || d->sW != 1 || d->pT || d->pH || d->pW)
            return fail(PTX_ERR_INVALID, "conv3d_dual: the first source must be a unit-stride 1x1x1 conv");
        if (d->flags & (PTX_EPI_RES_ADD | PTX_EPI_RES_PADA))
            return fail(PTX_ERR_INVALID, "conv3d_dual: the second source replaces the residual operand");
        if (d->x2_C <= 0 || d->x2_ld < d->x2_C || d->x2_ld % 4 || d->x2_sT <= 0 || d->x2_sH <= 0 || d->x2_sW <= 0 ||
            (d->To - 1) * d->x2_sT >= d->x2_T || (d->Ho - 1) * d->x2_sH >= d->x2_H || (d->Wo - 1) * d->x2_sW >= d->x2_W)
            return fail(PTX_ERR_INVALID, "conv3d_dual: second-source geometry out of range");
        if ((uintptr_t)x2 & 15) return fail(PTX_ERR_INVALID, "conv3d_dual: x2 must be 16-byte aligned");
        const uint64_t xb2 = (uint64_t)d->N * d->x2_T * d->x2_H * d->x2_W * d->x2_ld * 4ull;
        const int kc2 = (d->flags & PTX_F16X3_OPERANDS) ? (d->x2_C + 7) / 8 * 8 : (d->x2_C + 3) / 4 * 4;
        if (xb2 >= 0x80000000ull) return fail(PTX_ERR_UNSUPPORTED, "conv3d_dual: x2 must be < 2 GiB");
        a.dual = 1; a.x2 = x2; a.x2_bytes = (unsigned)xb2;
        a.ldx2 = d->x2_ld; a.kA2 = d->x2_ld; a.wcol2 = d->Kc;
        a.T2 = d->x2_T; a.H2 = d->x2_H; a.W2 = d->x2_W; a.s2T = d->x2_sT; a.s2H = d->x2_sH; a.s2W = d->x2_sW;
        // packed filter rows hold [Kc | Kc2] columns
        a.ldw = d->Kc + kc2; a.kB = a.ldw; a.w_tap_stride = (long long)d->Co_pad * a.ldw;
        a.w_bytes = (unsigned)((uint64_t)d->Co_pad * a.ldw * 4ull);
        a.k_live = d->Ci + d->x2_C;
    }
    return launch_conv(a, config, split_k, 1, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int ptx_conv3d_fwd(const ptx_conv3d_desc* d, const float* x, const float* w_packed, const float* bias,
                              const float* res, float* y, void* workspace, size_t workspace_bytes, int config,
                              int split_k, ptx_stream_t stream) {
    return conv3d_common(d, x, nullptr, w_packed, bias, res, y, workspace, workspace_bytes, config, split_k, stream);
}

extern "C" int ptx_conv3d_dual_fwd(const ptx_conv3d_desc* d, const float* x, const float* x2, const float* w_packed,
                                   const float* bias, float* y, void* workspace, size_t workspace_bytes, int config,
                                   int split_k, ptx_stream_t stream) {
    if (!x2) return fail(PTX_ERR_INVALID, "conv3d_dual: x2 == NULL");
    return conv3d_common(d, x, x2, w_packed, bias, nullptr, y, workspace, workspace_bytes, config, split_k, stream);
}

extern "C" int ptx_bgemm_nt(const float* A, const float* B, float* C, int32_t batch, int32_t M, int32_t Nn,
                            int32_t K, int32_t lda, int32_t ldb, int32_t ldc, int64_t strideA, int64_t strideB,
                            int64_t strideC, ptx_stream_t stream) {
    if (!A || !B || !C) return fail(PTX_ERR_INVALID, "bgemm: null pointer");
    if (batch <= 0 || M <= 0 || Nn <= 0 || K <= 0) return fail(PTX_ERR_INVALID, "bgemm: non-positive extent");
    const int k4 = (K + 3) / 4 * 4;
    if (lda % 4 || ldb % 4 || ldc % 4 || lda < k4 || ldb < k4 || ldc < Nn || strideA % 4 || strideB % 4 ||
        strideC % 4)
        return fail(PTX_ERR_INVALID, "bgemm: leading dims / strides must be multiples of 4 and cover the extents");
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) return fail(PTX_ERR_INVALID, "bgemm: misaligned pointer");
    if (batch > 65535) return fail(PTX_ERR_INVALID, "bgemm: batch > 65535");
    ConvArgs a{};
    a.x = A; a.w = B; a.bias = nullptr; a.res = nullptr; a.y = C;
    a.N = 1; a.Ti = 1; a.Hi = 1; a.Wi = M; a.ldx = lda; a.kA = k4;
    a.To = 1; a.Ho = 1; a.Wo = M; a.Co = Nn; a.ldy = ldc; a.k_live = K;
    a.kT = a.kH = a.kW = 1; a.sT = a.sH = a.sW = 1; a.pT = a.pH = a.pW = 0;
    a.ldw = ldb; a.kB = k4; a.w_rows = Nn; a.w_tap_stride = 0;
    a.M = M; a.flags = 0;
    a.bs_x = strideA; a.bs_w = strideB; a.bs_y = strideC;
    {
        const uint64_t xb = (uint64_t)M * lda * 4ull, wb = (uint64_t)Nn * ldb * 4ull;
        if (xb >= 0x80000000ull || wb >= 0x80000000ull)
            return fail(PTX_ERR_UNSUPPORTED, "bgemm: one batch item of A / B must be < 2 GiB");
        a.x_bytes = (unsigned)xb;
        a.w_bytes = (unsigned)wb;
    }
    // C columns [Nn, ldc) are written as zero (B rows >= Nn are read as zero)
    // LDS-DMA tiles: 128x128 (8 waves) when that still yields >= 2 workgroups per CU, else 64x64
    const int64_t blocks128 = cdiv64(M, 128) * cdiv(ldc, 128) * batch;
    const int config = (ldc >= 128 && blocks128 >= 2 * kNumCU) ? PTX_TILE("128x128x32/4x2/m32/dma") : PTX_TILE("64x64x32/2x2/m32/dma");
    return launch_conv(a, config, 1, batch, nullptr, 0, (hipStream_t)stream);
}

extern "C" int ptx_conv3d_fused_fwd(const ptx_conv3d_desc* d, const void* x, const void* w_packed, const float* bias,
                                    const void* res, void* y, const ptx_conv_fused_ext* ext, void* workspace,
                                    size_t workspace_bytes, int config, int split_k, ptx_stream_t stream) {
    if (!d || !(d->flags & PTX_F16_OPERANDS))
        return fail(PTX_ERR_INVALID, "conv3d_fused: an fp16-operand descriptor (PTX_F16_OPERANDS) is required");
    return conv3d_common(d, static_cast<const float*>(x), nullptr, static_cast<const float*>(w_packed), bias,
                         static_cast<const float*>(res), static_cast<float*>(y), workspace, workspace_bytes, config, split_k,
                         stream, ext);
}

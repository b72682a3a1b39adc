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
#include "conv_igemm_kernel.h"

namespace ptx {

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
          bool X3 = false, int KWR = 0, bool REPI = false>
static int launch_one(const ConvArgs& a, dim3 grid, hipStream_t st) {
    // fp16 tiles: the fused epilogue parks one MT-row block per wave ([MT][BN / WN + 4] floats) in the tile buffers
    constexpr size_t lds_tiles = (size_t)NSTAGE * ((KWR ? (BM + BM / 4 + 15) / 16 * 16 : BM) + (KWR ? KWR : 1) * BN) * (DMA ? BK : BK + 4) * sizeof(float);
    // (the row-major fp32 epilogue, REPI, parks the same per-wave block)
    constexpr size_t lds_epi = (F16 || REPI) ? (size_t)WM * WN * MT * (BN / WN + 4) * sizeof(float) : 0;
    constexpr size_t lds = lds_tiles > lds_epi ? lds_tiles : lds_epi;
    auto kern = conv_igemm_kernel<BM, BN, BK, WM, WN, MT, KTAIL, K22, DMA, NSTAGE, F16, X3, KWR, false, REPI>;
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

// fp32 LDS-DMA tiles with the row-major epilogue (".../re"): bias + residual + ReLU through 16-byte accesses
template <int BM, int BN, int BK, int WM, int WN, int MT, int NSTAGE, bool X3 = false>
static int launch_cfg_re(const ConvArgs& a, dim3 grid, hipStream_t st) {
    if ((a.kA % BK) || (a.kB % BK) || (a.dual && ((a.kA2 % BK) || (a.wcol2 % BK))))
        return launch_one<BM, BN, BK, WM, WN, MT, true, false, true, NSTAGE, false, X3, 0, true>(a, grid, st);
    return launch_one<BM, BN, BK, WM, WN, MT, false, false, true, NSTAGE, false, X3, 0, true>(a, grid, st);
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

#define PTX_CFG_RE(BM, BN, BK, WM, WN, MT) \
    { BM, BN, BK, WM, WN, MT, #BM "x" #BN "x" #BK "/" #WM "x" #WN "/m" #MT "/dma/re", launch_cfg_re<BM, BN, BK, WM, WN, MT, 2>, false, false, false, 0 }
#define PTX_CFG_RE3(BM, BN, BK, WM, WN, MT) \
    { BM, BN, BK, WM, WN, MT, #BM "x" #BN "x" #BK "/" #WM "x" #WN "/m" #MT "/dma3/re", launch_cfg_re<BM, BN, BK, WM, WN, MT, 3>, false, false, false, 0 }
#define PTX_CFG_X3RE(BM, BN, BK, WM, WN, MT, NS) \
    { BM, BN, BK, WM, WN, MT, #BM "x" #BN "x" #BK "/" #WM "x" #WN "/m" #MT "/dma" #NS "/re/x3", \
      launch_cfg_re<BM, BN, BK, WM, WN, MT, NS, true>, false, false, true, 0 }
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
    // 144-wide N tiles: the (2+1)D mid widths are 144 * 2^k (r2plus1d.py:68-69); one workgroup covers a whole 144-column
    // group, so the A tile crosses L2 -> LDS once instead of three times (48-wide tiles) and a wave's A fragment feeds
    // 9 MFMA column blocks
    PTX_CFG_DMA(64, 144, 32, 4, 1, 16),       // 127
    PTX_CFG_DMA(128, 144, 32, 8, 1, 16),      // 128
    // row-major epilogue (16-byte residual loads / output stores through a per-wave LDS transpose): the HBM-bound
    // pointwise convs -- layer1's 1x1x1 convs, the (2+1)D temporal / pointwise convs, the conv3 + shortcut GEMMs
    PTX_CFG_RE(128, 64, 16, 2, 2, 32),        // 129
    PTX_CFG_RE3(128, 64, 16, 2, 2, 32),       // 130
    PTX_CFG_RE(64, 64, 16, 2, 2, 32),         // 131
    PTX_CFG_RE3(64, 64, 16, 2, 2, 32),        // 132
    PTX_CFG_RE(64, 64, 32, 2, 2, 32),         // 133
    PTX_CFG_RE(128, 64, 32, 4, 2, 32),        // 134
    PTX_CFG_RE(128, 128, 16, 4, 2, 32),       // 135
    PTX_CFG_RE(128, 128, 32, 4, 2, 32),       // 136
    PTX_CFG_RE(64, 128, 16, 2, 2, 32),        // 137
    PTX_CFG_RE(64, 128, 32, 2, 2, 32),        // 138
    PTX_CFG_RE(32, 64, 32, 2, 2, 16),         // 139
    PTX_CFG_RE(32, 64, 64, 2, 2, 16),         // 140
    PTX_CFG_RE(32, 128, 32, 2, 2, 16),        // 141
    PTX_CFG_RE(64, 32, 32, 2, 2, 16),         // 142
    PTX_CFG_RE(112, 64, 32, 1, 4, 16),        // 143
    // ... and for the split-operand tiles, whose shorter matrix work leaves the pointwise convs even more epilogue-bound
    // ("/dma2/re/x3" = 2-stage)
    PTX_CFG_X3RE(128, 128, 32, 4, 2, 32, 2),  // 144
    PTX_CFG_X3RE(128, 128, 32, 4, 2, 32, 3),  // 145
    PTX_CFG_X3RE(128, 64, 32, 4, 2, 32, 3),   // 146
    PTX_CFG_X3RE(64, 128, 32, 2, 2, 32, 3),   // 147
    PTX_CFG_X3RE(64, 64, 32, 2, 2, 32, 2),    // 148
    PTX_CFG_X3RE(64, 64, 64, 2, 2, 32, 2),    // 149
    PTX_CFG_X3RE(32, 64, 64, 2, 2, 16, 3),    // 150
    PTX_CFG_X3RE(32, 128, 64, 2, 2, 16, 2),   // 151
    PTX_CFG_X3RE(128, 64, 32, 2, 2, 32, 2),   // 152
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

int validate_desc(const ptx_conv3d_desc* d) {
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

#ifdef PTX_IGEMM_TIMELINE
extern "C" int ptx_igemm_timeline(void* buf) {     // diagnostic build only: 8 x u64 per workgroup, or NULL to switch off
    return hip_check(hipMemcpyToSymbol(HIP_SYMBOL(g_ig_tl), &buf, sizeof(buf)), "ptx_igemm_timeline");
}
#endif

extern "C" int ptx_conv3d_config_supported(const ptx_conv3d_desc* d, int config) {
    if (validate_desc(d) != PTX_OK || config < 0 || config >= kNumConfigs) return 0;
    // the refusals of launch_conv, decided from the descriptor alone -- so a stale tuned-table entry is dropped when a plan
    // is COMPILED and never reaches a launch (K / M / N tails are all guarded inside the kernel)
    const ConvConfig& c = kConfigs[config];
    if (((d->flags & PTX_F16_OPERANDS) != 0) != (c.f16 != 0)) return 0;          // operand kind <-> tile kind
    if (((d->flags & PTX_F16X3_OPERANDS) != 0) != (c.x3 != 0)) return 0;
    const int groups = d->groups > 1 ? d->groups : 1;
    const bool dual = d->x2_C > 0;
    if (c.kwr) {                                                                  // kw-reuse tiles: whole output rows of a
        // dense stride-1 filter of THAT width (desc.Wi is the extent the filter slides over, upsampled or not)
        if (d->kW != c.kwr || d->sW != 1 || dual || groups > 1 || d->Wo < 8 || c.BM % d->Wo ||
            d->Wo != d->Wi + 2 * d->pW - d->kW + 1)
            return 0;
    }
    if (groups > 1 && !c.direct && ((d->Co / groups) % c.BN || dual)) return 0;  // an MFMA tile stays inside one group
    return 1;
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
// Tile-dependent half of the kernel arguments (tile counts, K chunks, split, extents): shared by the plain launch below and
// by conv_program.hip, which runs the same tile bodies from a persistent workgroup.
int finalize_conv_args(ConvArgs& a, int BM, int BN, int BK, int kwr, bool direct, int split_k, int batch) {
    if (kwr) {
        if (a.kW != kwr || a.sW != 1 || a.dual || batch > 1 || a.groups > 1 || a.Wo < 8 || BM % a.Wo ||
            a.Wo != a.Wi + 2 * a.pW - a.kW + 1)
            return fail(PTX_ERR_UNSUPPORTED, "conv3d: a kw-reuse tile needs a dense %d-wide stride-1 filter and a %d-row tile made of "
                        "whole output rows (Wo = %d >= 8)", kwr, BM, a.Wo);
        fastdiv_make((unsigned)(a.Wo + kwr - 1), a.dv_hw);
    }
    if (a.groups > 1 && !direct && (a.cog % BN || a.dual || batch > 1))
        return fail(PTX_ERR_UNSUPPORTED, "conv3d: an MFMA tile must divide the %d output channels of a group", a.cog);
    a.m_tiles = cdiv(a.M, BM);
    fastdiv_make((unsigned)a.Wo, a.dv_wo);      // every launch path (conv, dual, batched GEMM) decodes rows with these
    fastdiv_make((unsigned)a.Ho, a.dv_ho);
    fastdiv_make((unsigned)a.To, a.dv_to);
    {
        static int t_inner = -1;
        if (t_inner < 0) { const char* e = getenv("PTX_T_INNER"); t_inner = e ? atoi(e) : 0; }   // measured: no gain (stem is MFMA-bound), off by default
        const int plane = a.Ho * a.Wo;
        const int64_t in_clip_bytes = (int64_t)a.Ti * a.Hi * a.Wi * a.ldx * 4;
        // worth it only when one clip's input exceeds the per-XCD L2 (4 MiB) and the filter spans frames
        a.tiles_per_plane = (t_inner && a.kT > 1 && a.To > 1 && plane % BM == 0 && in_clip_bytes > (8 << 20))
                                ? plane / BM : 0;
    }
    a.ncol = (a.Co + 3) / 4 * 4;
    a.n_tiles = cdiv(a.ncol, BN);
    a.kchunks = cdiv(std::max(a.kA, a.kB), BK);
    if (a.dual) {
        a.kc1 = cdiv(a.kA, BK);
        a.kchunks = a.kc1 + cdiv(a.kA2, BK);
    }
    const int steps_total = a.kT * a.kH * a.kW * a.kchunks;
    if (split_k < 1) split_k = 1;
    if (split_k > steps_total) split_k = steps_total;
    if (batch > 1 || direct) split_k = 1;
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
    a.prune_analytic = prune_analytic_ok(a);
    return PTX_OK;
}

int launch_conv(ConvArgs& a, int config, int split_k, int batch, void* workspace, size_t workspace_bytes,
                hipStream_t st) {
    const ConvConfig& c = kConfigs[config];
    if ((a.f16 != 0) != c.f16)
        return fail(PTX_ERR_UNSUPPORTED, "conv3d: fp16-operand problems run on the /f16 tile configurations only (and vice versa)");
    if ((a.x3 != 0) != c.x3)
        return fail(PTX_ERR_UNSUPPORTED, "conv3d: split-operand problems run on the /x3 tile configurations only (and vice versa)");
    {
        const int fs = finalize_conv_args(a, c.BM, c.BN, c.BK, c.kwr, c.direct, split_k, batch);
        if (fs != PTX_OK) return fs;
        split_k = a.split_k;
    }
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

namespace ptx {
// Descriptor + tensors -> the tile-independent half of the kernel arguments (validation included).  Shared by every conv
// entry point here and by conv_program.hip.
int make_conv_args(const ptx_conv3d_desc* d, const float* x, const float* x2, const float* w_packed, const float* bias,
                   const float* res, float* y, const ptx_conv_fused_ext* ext, ConvArgs& a) {
    int s = validate_desc(d);
    if (s != PTX_OK) return s;
    if ((d->flags & (PTX_EPI_AFFINE | PTX_EPI_DUAL_RAW)) && !ext)
        return fail(PTX_ERR_INVALID, "conv3d: PTX_EPI_AFFINE / PTX_EPI_DUAL_RAW take their operands through ptx_conv3d_fused_fwd");
    if (!x || !w_packed || !y) return fail(PTX_ERR_INVALID, "conv3d: null tensor pointer");
    if ((d->flags & (PTX_EPI_RES_ADD | PTX_EPI_RES_PADA)) && !res)
        return fail(PTX_ERR_INVALID, "conv3d: residual flag set but res == NULL");
    if (((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)y | (uintptr_t)res) & 15)
        return fail(PTX_ERR_INVALID, "conv3d: pointers must be 16-byte aligned");
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
    a = ConvArgs{};
    a.x = x; a.w = w_packed; a.bias = bias; a.res = res; a.y = y;
    a.N = d->N; a.Ti = d->Ti; a.Hi = d->Hi; a.Wi = d->Wi; a.ldx = d->ldx; a.kA = d->ldx;
    a.To = d->To; a.Ho = d->Ho; a.Wo = d->Wo; a.Co = d->Co; a.ldy = d->ldy; a.k_live = d->Ci;
    a.kT = d->kT; a.kH = d->kH; a.kW = d->kW; a.sT = d->sT; a.sH = d->sH; a.sW = d->sW;
    a.pT = d->pT; a.pH = d->pH; a.pW = d->pW;
    a.ldw = d->Kc; a.kB = d->Kc; a.w_rows = d->Co_pad; a.w_tap_stride = (long long)d->Co_pad * d->Kc;
    a.M = d->N * d->To * d->Ho * d->Wo;
    a.flags = d->flags;
    {
        static int skip_early = -1;          // PTX_SKIP_EARLY=0: the fused stages fetch their skip operand in the epilogue (A/B)
        if (skip_early < 0) { const char* e = getenv("PTX_SKIP_EARLY"); skip_early = (e && atoi(e) == 0) ? 0 : 1; }
        if (!skip_early) a.flags |= kNoSkipEarly;
    }
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
    return PTX_OK;
}
}  // namespace ptx

static int conv3d_common(const ptx_conv3d_desc* d, const float* x, const float* x2, const float* w_packed,
                         const float* bias, const float* res, float* y, void* workspace, size_t workspace_bytes,
                         int config, int split_k, ptx_stream_t stream, const ptx_conv_fused_ext* ext = nullptr) {
    ConvArgs a;
    const int s = make_conv_args(d, x, x2, w_packed, bias, res, y, ext, a);
    if (s != PTX_OK) return s;
    if ((uintptr_t)workspace & 15) return fail(PTX_ERR_INVALID, "conv3d: pointers must be 16-byte aligned");
    if (config >= kNumConfigs) return fail(PTX_ERR_INVALID, "conv3d: config %d out of range", config);
    if (config < 0) {
        int sk = 1;
        config = ptx_conv3d_pick_config(d, &sk);
        if (split_k <= 0) split_k = sk;
    }
    if (split_k <= 0) split_k = 1;
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

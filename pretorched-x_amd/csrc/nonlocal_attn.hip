// Fused non-local (space-time self-attention) block core on the gfx950 fp32 matrix cores:
//     y = softmax(theta . phi^T) . g          (embedded_gaussian / gaussian, nonlocalnet.py:143-190)
//     y = (theta . phi^T / Nk) . g            (dot_product, nonlocalnet.py:192-211)
// in ONE launch, flash-style: the [Nq, Nk] affinity `f` (78.7 MB per block at N = 1568, 315 MB at the
// reference's usual N = 3136) never exists in HBM -- per 16-key tile it lives in 4 accumulator registers.
//
// Decomposition.  A wave owns 16 query rows for the whole kernel: their theta rows sit in registers
// (D/4 VGPRs, loaded once), their output accumulator O[16][DV] sits in DV/4 accumulator registers.  The 4
// waves of a workgroup (64 queries) share the key / value tiles -- 16 keys per step -- which arrive by LDS-DMA
// (buffer_load ... lds, one 1-KiB piece per wave-instruction, double buffered, one barrier per tile).
//
// Per tile, per wave, on v_mfma_f32_16x16x4_f32 (exact fp32):
//   S^T = phi_tile . theta^T   A = phi rows (keys) from LDS (ds_read_b128: 4 consecutive d per lane, the same
//                              hardware-k permutation on both operands), B = theta from registers.  Computing
//                              the TRANSPOSED tile puts one query per lane column (query = lane & 15, keys =
//                              4 * (lane >> 4) + r): the row maximum / sum of the online softmax is 3 in-lane
//                              ops + 2 cross-lane steps, and ...
//   O  += P . g_tile           ... P is already laid out as the A operand of the second MFMA (A[row = lane % 16]
//                              [k = lane / 16] is one float per lane): MFMA r contracts keys {r, 4+r, 8+r, 12+r},
//                              its B operand is g[4 * (lane >> 4) + r][channels] read straight from the LDS tile
//                              (ds_read_b128 = the operands of 4 MFMAs, conflict-free without a swizzle).
// No shuffle, no LDS round trip between the two matmuls.  Online softmax (running max m, running sum l, O
// rescaled by exp(m_old - m_new) only when some query's maximum moved).
#include "ptx_common.h"
#include <cfloat>
#include <cstdlib>

namespace ptx {

struct NlArgs {
    const float* theta;
    const float* phi;
    const float* g;
    float* y;
    int batch, Nq, Nk, d, dv;
    int ld_t, ld_p, ld_g, ld_y;
    long long bs_t, bs_p, bs_g, bs_y;
    int q_tiles, scale_only;
    int relu;               // PTX_NL_RELU: P = relu(S) / Nk (the 'concatenation' affinity, nonlocalnet.py:213-243)
    int out_f16;            // PTX_NL_OUT_F16: y holds halfs (ld_y / bs_y count halfs): the generator's fp16 plan feeds it to a half conv
    unsigned p_bytes, g_bytes, t_bytes;
    // stream-K (SK kernels): chunks per clip, the partial slots ([batch][sk_chunks][2] x ws_slot floats: a [64][DV] block of
    // unnormalised outputs, then 64 running maxima and 64 running sums)
    int sk_chunks;
    float* ws;
    long long ws_slot;
};

// (inline asm with AMDGPU register constraints must live in a __device__ function: inside the __global__
// template body the host pass silently drops the whole kernel stub)
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// one 1-KiB LDS-DMA piece: 16 bytes per lane from `voffset` of the buffer to lds_base + 16 * lane
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, float* lds_base, unsigned voffset) {
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)lds_base, 16, voffset, 0, 0, 0);
}
__device__ __forceinline__ void tile_barrier(int& a, int& b) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    asm volatile("; ds_reads of this tile depend on these" : "+v"(a), "+v"(b)::"memory");
}

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16h(half4_t a, half4_t b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ half4_t to_half4(float a, float b, float c, float d) {
    return half4_t{(_Float16)a, (_Float16)b, (_Float16)c, (_Float16)d};
}

// F16 (PTX_NL_F16, the BigGAN generator's fp16 plan): the same data movement and register layouts, but both matmuls
// run on v_mfma_f32_16x16x16_f16 -- the 4 consecutive d (resp. the 4 keys 4*gq .. 4*gq+3) a lane already holds as
// floats ARE one K = 16 operand once rounded to halfs, so 4 fp32 MFMAs (128 clk) become one f16 MFMA (16 clk).
// Accumulators, softmax statistics and the output stay fp32.
// X3 (PTX_NL_X3, Engine.precision = "x3"): fp32-ACCURATE on the same fp16 MFMAs -- every operand is split into
// (hi, lo) halfs in registers and a.b = hi.lo + lo.hi + hi.hi (conv_igemm.hip, X3): three f16 MFMAs instead of four
// fp32 ones, each 4x shorter.  F16 and X3 are exclusive (MODE 1 / 2).
__device__ __forceinline__ void split4(float a, float b, float c, float d, half4_t& hi, half4_t& lo) {
    hi = to_half4(a, b, c, d);
    lo = to_half4(a - (float)hi[0], b - (float)hi[1], c - (float)hi[2], d - (float)hi[3]);
}

// TG ("theta from global", D > 512: the 'gaussian' mode at the reference's widths, theta = x with C = 1024 channels,
// nonlocalnet.py:168-190): a wave's 16 theta rows no longer fit its registers (D / 4 VGPRs), so the B fragments of
// S^T = phi . theta^T are fetched with 16-byte buffer loads inside the d loop (L2-resident: 64 queries x 4 KiB per
// workgroup) instead of living in registers.  Same arithmetic and summation order as the register variant.
// KS ("key split", 2): 8 waves per workgroup -- wave group g = wave / 4 takes keys [16g, 16g + 16) of every 32-key tile for
// the SAME 64 queries, each group with its own online-softmax state, merged once at the end through LDS.  Why: a wave is a
// serial chain of Nk / 16 tiles x 128 MFMAs, and config 3's layer2 blocks launch only batch x ceil(1568 / 64) = 200
// workgroups -- 800 waves on 1024 SIMDs, one wave per SIMD, nothing to cover its softmax / LDS / barrier bubbles (measured:
// MFMA pipe 57 % busy, 53 % of the fp32 peak).  Split, every active SIMD holds two waves of half the length.
// SK ("stream-K", with KS = 2): the (query tile, key tile) units of ONE clip -- q_tiles x n_tiles of them, query tile major --
// are cut into p.sk_chunks equal contiguous chunks, one workgroup each, WHATEVER the batch (a clip's split points are a
// function of its own extents: its bits do not change with the batch it arrives in).  Why: config 3's layer2 blocks are 25
// query tiles x 8 clips = 200 workgroups at one per CU (128 KiB of LDS) on 256 CUs; 32 chunks per clip are 256 workgroups of
// 38.3 units instead of 200 of 49.  A chunk covers at most two query tiles (q_tiles <= sk_chunks): per segment the
// workgroup runs the ordinary loop over its key range and parks (unnormalised O, m, l) of its 64 queries in a workspace slot;
// nl_sk_combine_kernel folds the 2-3 pieces of every query tile in chunk order (a fixed order: deterministic).
template <int D, int DV, bool SOFTMAX, int MODE = 0, bool TG = false, int KS = 1, bool SK = false>
__global__ void __launch_bounds__(256 * KS) nl_attention_kernel(const NlArgs p) {
    constexpr bool F16 = MODE == 1, X3 = MODE == 2;
    static_assert(!TG || MODE == 0, "theta-from-global is an fp32 variant");
    static_assert(KS == 1 || (KS == 2 && !TG), "key split: two wave groups");
    static_assert(!SK || (KS == 2 && SOFTMAX && !F16), "stream-K: the key-split softmax kernel (fp32 / split operands)");
    constexpr int TK = 16 * KS;              // keys per tile
    constexpr int NW = 4 * KS;               // waves per workgroup
    constexpr int QJ = D / 16;               // 16-wide d steps (one ds_read_b128 + 4 MFMAs each)
    constexpr int CB = DV / 64;              // 64-channel output super-blocks (one ds_read_b128 + 4 MFMAs per key group)
    constexpr int F4R = D / 4;               // 16-byte slots per key row
    constexpr int NKP = TK * D / 256;        // 1-KiB DMA pieces of a key tile
    constexpr int NVP = TK * DV / 256;
    static_assert(D % 16 == 0 && DV % 64 == 0, "tile extents");
    constexpr unsigned kOOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                        // [2][TK][D]   (16-byte slots XOR-swizzled by row)
    float* Vs = smem + 2 * TK * D;           // [2][TK][DV]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_tiles = (p.Nk + TK - 1) / TK;
    const int tile = xcd_remap(blockIdx.x, (SK ? p.sk_chunks : p.q_tiles) * p.batch);     // one clip's tiles share an XCD's L2
    const int b = tile / (SK ? p.sk_chunks : p.q_tiles);
    int qt = tile - b * p.q_tiles;           // (SK: set per segment)
    // SK: this workgroup's units [u, u_end) of the clip's q_tiles x n_tiles stream; seg counts its segments (query tiles)
    const int chunk = SK ? tile - b * p.sk_chunks : 0;
    int u = 0, u_end = 0, seg = 0, t_lo = 0, t_hi = n_tiles;
    if constexpr (SK) {
        const long long U = (long long)p.q_tiles * n_tiles;
        u = (int)((chunk * U) / p.sk_chunks);
        u_end = (int)(((chunk + 1) * U) / p.sk_chunks);
        if (u >= u_end) return;              // (more chunks than units: nothing to do, nothing the combine pass reads)
    }
    const int c0 = blockIdx.y * DV;          // output-channel chunk of this workgroup
    const int n = lane & 15, gq = lane >> 4;
    const int grp = wave >> 2, wq = wave & 3;                        // key group (KS == 2), query sub-tile of the wave
    int q = qt * 64 + wq * 16 + n;           // this lane's query (as MFMA column / A row)

    const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.theta + (size_t)b * p.bs_t), 0, p.t_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.phi + (size_t)b * p.bs_p), 0, p.p_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.g + (size_t)b * p.bs_g), 0, p.g_bytes, 0x00020000);

    do {      // SK: one pass per segment (a query tile and this chunk's key range of it); otherwise exactly one pass
    if constexpr (SK) {
        qt = u / n_tiles;
        t_lo = u - qt * n_tiles;
        t_hi = min(n_tiles, t_lo + (u_end - u));
        q = qt * 64 + wq * 16 + n;
    }
    // ---- theta rows of this wave's 16 queries -> registers (the B operand of S^T = phi . theta^T) ----
    auto load_theta = [&](int j) -> f32x4 {
        const int col = 16 * j + 4 * gq;
        const unsigned off = ((unsigned)q * (unsigned)p.ld_t + (unsigned)col) * 4u;
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                             rs_t, (q < p.Nq && col < p.d) ? off : kOOB, 0, 0));
    };
    f32x4 qf[TG ? 1 : QJ];
    if constexpr (!TG) {
#pragma unroll
        for (int j = 0; j < QJ; ++j) qf[j] = load_theta(j);
    }
    half4_t qh[QJ], ql[X3 ? QJ : 1];
    if constexpr (F16) {
#pragma unroll
        for (int j = 0; j < QJ; ++j) qh[j] = to_half4(qf[j][0], qf[j][1], qf[j][2], qf[j][3]);
    }
    if constexpr (X3) {
#pragma unroll
        for (int j = 0; j < QJ; ++j) split4(qf[j][0], qf[j][1], qf[j][2], qf[j][3], qh[j], ql[j]);
    }

    // ---- per-lane DMA source offsets (tile independent) ----
    auto swz = [&](int row, int slot) -> int {
        return F4R >= 16 ? (slot ^ (row & 15)) : F4R == 8 ? (slot ^ ((row >> 1) & 7)) : (slot ^ ((row >> 2) & 3));
    };
    constexpr int KPW = (NKP + NW - 1) / NW, VPW = (NVP + NW - 1) / NW;       // pieces per wave
    unsigned koff[KPW], voff[VPW];
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        const int f = (wave + NW * i) * 256 + lane * 4;           // float index in the linear [TK][D] image
        const int row = f / D, slot = (f % D) / 4;
        const int col = swz(row, slot) * 4;                       // logical column this physical slot holds
        koff[i] = col < p.d ? ((unsigned)row * (unsigned)p.ld_p + (unsigned)col) * 4u : kOOB;
    }
#pragma unroll
    for (int i = 0; i < VPW; ++i) {
        const int f = (wave + NW * i) * 256 + lane * 4;
        const int row = f / DV, col = c0 + f % DV;
        voff[i] = col < p.dv ? ((unsigned)row * (unsigned)p.ld_g + (unsigned)col) * 4u : kOOB;
    }
    auto issue_tile = [&](int t, int buf) {
        const unsigned kbase = (unsigned)t * TK * (unsigned)p.ld_p * 4u, vbase = (unsigned)t * TK * (unsigned)p.ld_g * 4u;
#pragma unroll
        for (int i = 0; i < KPW; ++i)
            if (wave + NW * i < NKP)
                dma16(rs_p, Ks + buf * TK * D + (wave + NW * i) * 256, koff[i] == kOOB ? kOOB : koff[i] + kbase);
#pragma unroll
        for (int i = 0; i < VPW; ++i)
            if (wave + NW * i < NVP)
                dma16(rs_g, Vs + buf * TK * DV + (wave + NW * i) * 256, voff[i] == kOOB ? kOOB : voff[i] + vbase);
    };

    f32x4 O[CB][4];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int e = 0; e < 4; ++e) O[cb][e] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const float inv_nk = 1.0f / (float)p.Nk;

    issue_tile(t_lo, 0);
    // fragment read offsets (floats): phi row = n, slot (4j + gq) ^ swizzle;  g row = 4*gq + r, slot cb*16 + n
    const int k_row_off = (16 * grp + n) * D;                     // (16 * grp is a multiple of every swizzle period)
    const int v_row_off = (16 * grp + 4 * gq) * DV + n * 4;
    for (int t = t_lo; t < t_hi; ++t) {
        const int buf = (t - t_lo) & 1;
        // tile t landed: every wave waits for its own DMA pieces, then the barrier; buffer buf^1 is free (its last
        // readers passed this barrier too).  The fragment offsets go through an asm the compiler cannot move
        // above the barrier, so no ds_read of the new tile is scheduled early (cdna_hip_programming.md 5.7).
        int ko = k_row_off, vo = v_row_off;
        tile_barrier(ko, vo);
        if (t + 1 < t_hi) issue_tile(t + 1, buf ^ 1);
        const float* Kb = Ks + buf * TK * D + ko;
        const float* Vb = Vs + buf * TK * DV + vo;

        // ---- S^T tile: two accumulators break the 40-cycle dependent-MFMA chain ----
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < QJ; ++j) {
            const f32x4 kf = *reinterpret_cast<const f32x4*>(Kb + swz(n, 4 * j + gq) * 4);
            if constexpr (F16) {
                const half4_t kh = to_half4(kf[0], kf[1], kf[2], kf[3]);
                if (j & 1) s1 = mfma16h(kh, qh[j], s1);
                else       s0 = mfma16h(kh, qh[j], s0);
                continue;
            }
            if constexpr (X3) {
                half4_t kh, kl;
                split4(kf[0], kf[1], kf[2], kf[3], kh, kl);
                if (j & 1) { s1 = mfma16h(kh, ql[j], s1); s1 = mfma16h(kl, qh[j], s1); s1 = mfma16h(kh, qh[j], s1); }
                else       { s0 = mfma16h(kh, ql[j], s0); s0 = mfma16h(kl, qh[j], s0); s0 = mfma16h(kh, qh[j], s0); }
                continue;
            }
            const f32x4 qv = TG ? load_theta(j) : qf[TG ? 0 : j];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (j & 1) s1 = mfma16(kf[e], qv[e], s1);
                else       s0 = mfma16(kf[e], qv[e], s0);
            }
        }
        f32x4 s = s0 + s1;                    // s[r] = S[q = n][key = 16t + 4*gq + r]

        float pr[4];
        if constexpr (SOFTMAX) {
            const int key0 = t * TK + 16 * grp + 4 * gq;
#pragma unroll
            for (int r = 0; r < 4; ++r) s[r] = (key0 + r < p.Nk) ? s[r] : -INFINITY;
            float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const bool moved = m_new > m_run;
            // a wave (key group) that has seen no valid key yet keeps m = -inf: exp(-inf - (-inf)) would be NaN, so the
            // reference point of the exponentials is 0 until a key arrives (alpha = p = 0)
            const float m_ref = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __expf(m_run - m_ref);         // exp(-inf) = 0 on the first tile
#pragma unroll
            for (int r = 0; r < 4; ++r) pr[r] = __expf(s[r] - m_ref);
            l_run = l_run * alpha + ((pr[0] + pr[1]) + (pr[2] + pr[3]));
            m_run = m_new;
            if (__any(moved)) {               // O rows are queries 4*gq + r: fetch their alpha from lane (4*gq + r)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = __shfl(alpha, 4 * gq + r, 64);
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                        for (int e = 0; e < 4; ++e) O[cb][e][r] *= a;
                }
            }
        } else {
#pragma unroll
            // keys >= Nk read as zero rows: no mask needed.  Split operands: the 1 / Nk factor moves to the epilogue -- P = S / Nk
            // would sit around 1e-2, where the lo half of a split goes subnormal and the product keeps ~17 bits instead of 22
            for (int r = 0; r < 4; ++r) {
                const float v = p.relu ? fmaxf(s[r], 0.f) : s[r];
                pr[r] = X3 ? v : v * inv_nk;
            }
        }

        // ---- O += P . g_tile ----
        if constexpr (F16) {
            // A = P[q = n][keys 4*gq + 0..3] (this lane's pr[]), B_e = g[keys 4*gq + 0..3][channel cb*64 + 4n + e]: the e-th
            // component of the four 16-byte reads below -- one MFMA per output-channel quarter
            const half4_t ph = to_half4(pr[0], pr[1], pr[2], pr[3]);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                f32x4 vf[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) vf[r] = *reinterpret_cast<const f32x4*>(Vb + r * DV + cb * 64);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    O[cb][e] = mfma16h(ph, to_half4(vf[0][e], vf[1][e], vf[2][e], vf[3][e]), O[cb][e]);
            }
            continue;
        }
        if constexpr (X3) {
            // softmax weights are <= 1 and mostly ~1 / Nk: unscaled, their lo half (~2^-11 P) is a SUBNORMAL half for
            // P < 0.125 and the P . g products keep 13-14 bits.  Split 2^12 P instead (hi <= 4096, lo normal down to
            // P ~ 3e-5); the factor is a power of two and leaves with 1 / l in the epilogue.
            constexpr float kPS = SOFTMAX ? 4096.f : 1.f;
            half4_t ph, pl;
            split4(pr[0] * kPS, pr[1] * kPS, pr[2] * kPS, pr[3] * kPS, ph, pl);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                f32x4 vf[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) vf[r] = *reinterpret_cast<const f32x4*>(Vb + r * DV + cb * 64);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    half4_t vh, vl;
                    split4(vf[0][e], vf[1][e], vf[2][e], vf[3][e], vh, vl);
                    O[cb][e] = mfma16h(ph, vl, O[cb][e]);
                    O[cb][e] = mfma16h(pl, vh, O[cb][e]);
                    O[cb][e] = mfma16h(ph, vh, O[cb][e]);
                }
            }
            continue;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const f32x4 vf = *reinterpret_cast<const f32x4*>(Vb + r * DV + cb * 64);
#pragma unroll
                for (int e = 0; e < 4; ++e) O[cb][e] = mfma16(pr[r], vf[e], O[cb][e]);
            }
        }
    }

    if constexpr (KS == 2) {
        // ---- merge the two key groups: group 1 parks (m, l, O) lane-linearly in the (idle) tile buffers, group 0 folds
        // them in:  M = max(m0, m1),  l = l0 e^(m0 - M) + l1 e^(m1 - M),  O likewise per query row ----
        __syncthreads();                                  // every wave is done with the last tile
        float* Mx = smem + wq * (64 * (CB * 16 + 2));     // per query sub-tile: [CB * 16 + 2][64 lanes]
        if (grp == 1) {
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Mx[((cb * 4 + e) * 4 + r) * 64 + lane] = O[cb][e][r];
            Mx[(CB * 16) * 64 + lane] = m_run;
            Mx[(CB * 16 + 1) * 64 + lane] = l_run;
        }
        __syncthreads();
        if constexpr (!SK) {
            if (grp == 1) return;
        }
        if (!SK || grp == 0) {
            const float m1 = Mx[(CB * 16) * 64 + lane], l1 = Mx[(CB * 16 + 1) * 64 + lane];
            float a0 = 1.f, a1 = 1.f;
            if constexpr (SOFTMAX) {
                const float M = fmaxf(m_run, m1);         // finite: group 0 always owns at least one valid key
                a0 = __expf(m_run - M);
                a1 = __expf(m1 - M);                      // exp(-inf) = 0 when group 1 saw no key (its O and l are 0)
                l_run = l_run * a0 + l1 * a1;
                m_run = M;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float b0 = SOFTMAX ? __shfl(a0, 4 * gq + r, 64) : 1.f, b1 = SOFTMAX ? __shfl(a1, 4 * gq + r, 64) : 1.f;
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        O[cb][e][r] = O[cb][e][r] * b0 + Mx[((cb * 4 + e) * 4 + r) * 64 + lane] * b1;
            }
        }
    }
    if constexpr (SK) {
        // ---- park this segment's (O, m, l): slot (clip, chunk, segment), rows = the tile's 64 queries ----
        if (grp == 0) {
            l_run += __shfl_xor(l_run, 16, 64);
            l_run += __shfl_xor(l_run, 32, 64);
            float* slot = p.ws + ((size_t)(b * p.sk_chunks + chunk) * 2 + seg) * (size_t)p.ws_slot;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ql_ = wq * 16 + 4 * gq + r;
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    const f32x4 o = {O[cb][0][r], O[cb][1][r], O[cb][2][r], O[cb][3][r]};
                    *reinterpret_cast<f32x4*>(slot + (size_t)ql_ * DV + cb * 64 + 4 * n) = o;
                }
            }
            if (gq == 0) {
                slot[64 * DV + wq * 16 + n] = m_run;
                slot[64 * DV + 64 + wq * 16 + n] = l_run;
            }
        }
        u += t_hi - t_lo;
        ++seg;
        __syncthreads();                                  // the merge buffer aliases the tile stages of the next segment
        continue;
    }
    // ---- epilogue: 1 / l per query, 16-byte stores (lane: query 4*gq + r, channels cb*64 + 4*n .. +3) ----
    float inv = 1.f;
    if constexpr (SOFTMAX) {
        l_run += __shfl_xor(l_run, 16, 64);
        l_run += __shfl_xor(l_run, 32, 64);
        inv = 1.0f / l_run;
        if constexpr (X3) inv *= 1.0f / 4096.0f;      // the 2^12 carried by the split softmax weights
    }
    float* yb = p.y + (size_t)b * p.bs_y;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float iv = SOFTMAX ? __shfl(inv, 4 * gq + r, 64) : (X3 ? inv_nk : 1.f);
        const int qo = qt * 64 + wq * 16 + 4 * gq + r;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const int ch = c0 + cb * 64 + 4 * n;
            if (qo < p.Nq && ch < p.dv) {
                const f32x4 o = {O[cb][0][r] * iv, O[cb][1][r] * iv, O[cb][2][r] * iv, O[cb][3][r] * iv};
                if (p.out_f16) {
                    _Float16* yh = reinterpret_cast<_Float16*>(p.y) + (size_t)b * p.bs_y + (size_t)qo * p.ld_y + ch;
                    *reinterpret_cast<half4_t*>(yh) = to_half4(o[0], o[1], o[2], o[3]);
                } else {
                    *reinterpret_cast<f32x4*>(yb + (size_t)qo * p.ld_y + ch) = o;
                }
            }
        }
    }
    } while (SK && u < u_end);
}

// most pieces one query tile can be cut into by the stream-K split: the combine kernel keeps that many partials in registers,
// and the host refuses (falls back to the plain kernel) any (chunks, query tiles) pair that could exceed it
constexpr int kNlSkMaxPieces = 6;

// Stream-K combine: query tile (b, qt) <- its pieces, in chunk order.  Piece k holds unnormalised rows O_k, running maxima m_k
// and running sums l_k of its key range:  M = max_k m_k,  y = (sum_k O_k e^(m_k - M)) / (sum_k l_k e^(m_k - M)) * scale
// (scale = 2^-12 under split operands: the factor their softmax weights carry).  One workgroup per query tile, 16-byte items.
template <int DV>
__global__ void __launch_bounds__(256) nl_sk_combine_kernel(const NlArgs p, const int n_tiles, const float scale) {
    // a workgroup = 256 / (DV / 4) query rows of one tile, one 16-byte item per thread: every load of a thread is independent
    // of the others (one memory round trip), and a clip's 25 tiles x 16 row groups fill the chip
    constexpr int F4 = DV / 4, ROWS = 256 / F4, PARTS = 64 / ROWS;
    constexpr int kMaxPieces = kNlSkMaxPieces;       // nl_sk_chunks refuses any split with more pieces per query tile
    const int tile = xcd_remap(blockIdx.x / PARTS, p.q_tiles * p.batch), part = blockIdx.x % PARTS;
    const int b = tile / p.q_tiles, qt = tile - b * p.q_tiles;
    const long long U = (long long)p.q_tiles * n_tiles;
    const int u_lo = qt * n_tiles, u_hi = u_lo + n_tiles;
    // the pieces of a query tile are consecutive chunks (U >= sk_chunks: no empty chunk in between)
    int c_first = -1, np = 0;
    for (int c = 0; c < p.sk_chunks; ++c) {
        const int lo = (int)((c * U) / p.sk_chunks), hi = (int)(((c + 1) * U) / p.sk_chunks);
        if (hi <= lo || lo >= u_hi || hi <= u_lo) continue;
        if (c_first < 0) c_first = c;
        ++np;
    }
    np = min(np, kMaxPieces);
    const int row = part * ROWS + (int)threadIdx.x / F4, ch = ((int)threadIdx.x % F4) * 4;
    const int qo = qt * 64 + row;
    if (qo >= p.Nq || ch >= p.dv || np == 0) return;
    float m[kMaxPieces], l[kMaxPieces];
    f32x4 o[kMaxPieces];
#pragma unroll
    for (int k = 0; k < kMaxPieces; ++k) {
        const int c = c_first + min(k, np - 1);
        const int lo = (int)((c * U) / p.sk_chunks);
        const int seg = (lo / n_tiles == qt) ? 0 : 1;     // the chunk's first segment is the query tile it starts in
        const float* pc = p.ws + ((size_t)(b * p.sk_chunks + c) * 2 + seg) * (size_t)p.ws_slot;
        m[k] = k < np ? pc[64 * DV + row] : -INFINITY;
        l[k] = pc[64 * DV + 64 + row];
        o[k] = *reinterpret_cast<const f32x4*>(pc + (size_t)row * DV + ch);
    }
    float M = m[0];
#pragma unroll
    for (int k = 1; k < kMaxPieces; ++k) M = fmaxf(M, m[k]);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float lsum = 0.f;
#pragma unroll
    for (int k = 0; k < kMaxPieces; ++k) {            // chunk order: a fixed summation order
        const float w = k < np ? __expf(m[k] - M) : 0.f;
        acc += o[k] * w;
        lsum += l[k] * w;
    }
    const float iv = scale / lsum;
    *reinterpret_cast<f32x4*>(p.y + (size_t)b * p.bs_y + (size_t)qo * p.ld_y + ch) = acc * iv;
}

// DS ("d split"): short sequences (Nq <= 512: the layer3 blocks of the video nets, N = 196 / 392 at C = 1024).  There the
// kernel above is a handful of long serial chains -- config 3's layer3: batch x 4 query tiles x 2 dv chunks = 64 workgroups,
// each 13 key tiles x 192 dependent-ish MFMAs = 55 us for 0.6 GFLOP (11 TFLOP/s).  What is scarce is not arithmetic but
// chain length, so the work of ONE 16-query group is spread over the NW waves of a workgroup along the CHANNEL axes:
//   * wave w computes the partial S^T tile over its d slice [w D/NW, (w+1) D/NW)  (D / (16 NW) fragment reads x 4 MFMAs),
//   * the partial tiles (4 floats per lane) meet in LDS, every wave adds them in the same order -> identical S, identical
//     online-softmax state in every wave (recomputed redundantly: 4 exps per lane),
//   * wave w multiplies P into ITS dv slice [w DV/NW, ...) of the output: no S recompute across dv chunks any more.
// Per tile and wave: D/(4 NW) + DV/(4 NW) MFMAs instead of D/4 + DV/4 -- an 8-fold shorter chain at D = DV = 512 -- for two
// barriers per tile.  A workgroup streams every key / value row of its clip through LDS (N x (D + DV) x 4 B = 0.8 MB at
// N = 196): L2-resident and irrelevant at these lengths, prohibitive at N = 1568, hence the Nq bound.
template <int D, int DV, bool SOFTMAX, int NW>
__global__ void __launch_bounds__(64 * NW) nl_attention_ds_kernel(const NlArgs p) {
    constexpr int DW = D / NW, VW = DV / NW;         // this wave's slice of d / dv
    constexpr int QJ = DW / 16, CB = VW / 64;
    constexpr int F4R = D / 4;
    constexpr int NKP = 16 * D / 256, NVP = 16 * DV / 256;        // 1-KiB DMA pieces of a 16-key tile
    static_assert(DW % 16 == 0 && VW % 64 == 0 && F4R >= 16, "slice extents");
    constexpr unsigned kOOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                                // [2][16][D]   (16-byte slots XOR-swizzled by row)
    float* Vs = smem + 2 * 16 * D;                   // [2][16][DV]
    float* Xs = Vs + 2 * 16 * DV;                    // [2][NW][64 lanes][4]: the waves' partial S^T tiles

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = xcd_remap(blockIdx.x, p.q_tiles * p.batch);   // q_tiles counts 16-query groups here
    const int b = tile / p.q_tiles, qg = tile - b * p.q_tiles;
    const int n = lane & 15, gq = lane >> 4;
    const int q = qg * 16 + n;

    const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.theta + (size_t)b * p.bs_t), 0, p.t_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.phi + (size_t)b * p.bs_p), 0, p.p_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.g + (size_t)b * p.bs_g), 0, p.g_bytes, 0x00020000);

    f32x4 qf[QJ];                                    // theta[q][this wave's d slice]
#pragma unroll
    for (int j = 0; j < QJ; ++j) {
        const int col = DW * wave + 16 * j + 4 * gq;
        const unsigned off = ((unsigned)q * (unsigned)p.ld_t + (unsigned)col) * 4u;
        qf[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_t, (q < p.Nq && col < p.d) ? off : kOOB, 0, 0));
    }
    constexpr int KPW = (NKP + NW - 1) / NW, VPW = (NVP + NW - 1) / NW;
    unsigned koff[KPW], voff[VPW];
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        const int f = (wave + NW * i) * 256 + lane * 4;
        const int row = f / D, slot = (f % D) / 4;
        const int col = (slot ^ (row & 15)) * 4;
        koff[i] = col < p.d ? ((unsigned)row * (unsigned)p.ld_p + (unsigned)col) * 4u : kOOB;
    }
#pragma unroll
    for (int i = 0; i < VPW; ++i) {
        const int f = (wave + NW * i) * 256 + lane * 4;
        const int row = f / DV, col = f % DV;
        voff[i] = col < p.dv ? ((unsigned)row * (unsigned)p.ld_g + (unsigned)col) * 4u : kOOB;
    }
    auto issue_tile = [&](int t, int buf) {
        const unsigned kbase = (unsigned)t * 16u * (unsigned)p.ld_p * 4u, vbase = (unsigned)t * 16u * (unsigned)p.ld_g * 4u;
#pragma unroll
        for (int i = 0; i < KPW; ++i)
            if (wave + NW * i < NKP)
                dma16(rs_p, Ks + buf * 16 * D + (wave + NW * i) * 256, koff[i] == kOOB ? kOOB : koff[i] + kbase);
#pragma unroll
        for (int i = 0; i < VPW; ++i)
            if (wave + NW * i < NVP)
                dma16(rs_g, Vs + buf * 16 * DV + (wave + NW * i) * 256, voff[i] == kOOB ? kOOB : voff[i] + vbase);
    };

    f32x4 O[CB][4];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int e = 0; e < 4; ++e) O[cb][e] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const float inv_nk = 1.0f / (float)p.Nk;
    const int n_tiles = (p.Nk + 15) / 16;
    issue_tile(0, 0);
    const int k_row_off = n * D, v_row_off = 4 * gq * DV + VW * wave + n * 4;
    for (int t = 0; t < n_tiles; ++t) {
        const int buf = t & 1;
        int ko = k_row_off, vo = v_row_off;
        tile_barrier(ko, vo);                        // tile t landed; buffer buf^1 and X[buf^1] are free
        if (t + 1 < n_tiles) issue_tile(t + 1, buf ^ 1);
        const float* Kb = Ks + buf * 16 * D + ko;
        const float* Vb = Vs + buf * 16 * DV + vo;
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < QJ; ++j) {
            const f32x4 kf = *reinterpret_cast<const f32x4*>(Kb + (((DW * wave) / 4 + 4 * j + gq) ^ n) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (j & 1) s1 = mfma16(kf[e], qf[j][e], s1);
                else       s0 = mfma16(kf[e], qf[j][e], s0);
            }
        }
        // partial S^T tiles meet in LDS; every wave adds them in wave order (bit-identical S everywhere)
        float* Xb = Xs + buf * NW * 256;
        *reinterpret_cast<f32x4*>(Xb + wave * 256 + lane * 4) = s0 + s1;
        __syncthreads();
        f32x4 s = *reinterpret_cast<const f32x4*>(Xb + lane * 4);
#pragma unroll
        for (int w = 1; w < NW; ++w) s += *reinterpret_cast<const f32x4*>(Xb + w * 256 + lane * 4);

        float pr[4];
        if constexpr (SOFTMAX) {
            const int key0 = t * 16 + 4 * gq;
#pragma unroll
            for (int r = 0; r < 4; ++r) s[r] = (key0 + r < p.Nk) ? s[r] : -INFINITY;
            float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);        // finite from tile 0 on: every 16-key tile of a wave holds a valid key
            const bool moved = m_new > m_run;
            const float alpha = __expf(m_run - m_new);
#pragma unroll
            for (int r = 0; r < 4; ++r) pr[r] = __expf(s[r] - m_new);
            l_run = l_run * alpha + ((pr[0] + pr[1]) + (pr[2] + pr[3]));
            m_run = m_new;
            if (__any(moved)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = __shfl(alpha, 4 * gq + r, 64);
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                        for (int e = 0; e < 4; ++e) O[cb][e][r] *= a;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) pr[r] = (p.relu ? fmaxf(s[r], 0.f) : s[r]) * inv_nk;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const f32x4 vf = *reinterpret_cast<const f32x4*>(Vb + r * DV + cb * 64);
#pragma unroll
                for (int e = 0; e < 4; ++e) O[cb][e] = mfma16(pr[r], vf[e], O[cb][e]);
            }
        }
    }
    float inv = 1.f;
    if constexpr (SOFTMAX) {
        l_run += __shfl_xor(l_run, 16, 64);
        l_run += __shfl_xor(l_run, 32, 64);
        inv = 1.0f / l_run;
    }
    float* yb = p.y + (size_t)b * p.bs_y;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float iv = SOFTMAX ? __shfl(inv, 4 * gq + r, 64) : 1.f;
        const int qo = qg * 16 + 4 * gq + r;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const int ch = VW * wave + cb * 64 + 4 * n;
            if (qo < p.Nq && ch < p.dv) {
                const f32x4 o = {O[cb][0][r] * iv, O[cb][1][r] * iv, O[cb][2][r] * iv, O[cb][3][r] * iv};
                *reinterpret_cast<f32x4*>(yb + (size_t)qo * p.ld_y + ch) = o;
            }
        }
    }
}

template <int D, int DV, int NW>
static int launch_nl_ds(NlArgs a, hipStream_t st) {
    constexpr size_t lds = ((size_t)2 * 16 * (D + DV) + (size_t)2 * NW * 256) * sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS budget");
    a.q_tiles = cdiv(a.Nq, 16);                      // one 16-query group per workgroup
    const dim3 grid((unsigned)(a.q_tiles * a.batch));
    auto launch = [&](auto kernel) -> int {
        static bool attr_set[64] = {};
        int dev = 0;
        PTX_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {
            PTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
        hipLaunchKernelGGL(kernel, grid, dim3(64 * NW), lds, st, a);
        return hip_check(hipGetLastError(), "nonlocal attention (d-split) launch");
    };
    return a.scale_only ? launch(nl_attention_ds_kernel<D, DV, false, NW>) : launch(nl_attention_ds_kernel<D, DV, true, NW>);
}

template <int D, int DV, bool SOFTMAX, int MODE = 0, bool TG = false, int KS = 1>
static int launch_nl_mode(const NlArgs& a, hipStream_t st) {
    constexpr size_t lds_t = (size_t)2 * 16 * KS * (D + DV) * sizeof(float);
    constexpr size_t lds_m = KS == 2 ? (size_t)4 * 64 * ((DV / 64) * 16 + 2) * sizeof(float) : 0;     // the groups' merge buffer
    constexpr size_t lds = lds_t > lds_m ? lds_t : lds_m;
    static_assert(lds <= 160 * 1024, "LDS budget");
    const dim3 grid((unsigned)(a.q_tiles * a.batch), (unsigned)cdiv(a.dv, DV));
    static bool attr_set[64] = {};   // per device; benign race (idempotent call)
    int dev = 0;
    PTX_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        PTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(nl_attention_kernel<D, DV, SOFTMAX, MODE, TG, KS>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL((nl_attention_kernel<D, DV, SOFTMAX, MODE, TG, KS>), grid, dim3(256 * KS), lds, st, a);
    return hip_check(hipGetLastError(), "nonlocal attention launch");
}

template <int D, int DV>
static int launch_nl(const NlArgs& a, hipStream_t st) {
    return a.scale_only ? launch_nl_mode<D, DV, false>(a, st) : launch_nl_mode<D, DV, true>(a, st);
}
// key-split variants (8 waves, two key groups): chosen when the plain grid leaves the chip at one wave per SIMD
template <int D, int DV>
static int launch_nl_ks(const NlArgs& a, hipStream_t st) {
    return a.scale_only ? launch_nl_mode<D, DV, false, 0, false, 2>(a, st) : launch_nl_mode<D, DV, true, 0, false, 2>(a, st);
}
template <int D, int DV>
static int launch_nl_ks_x3(const NlArgs& a, hipStream_t st) {
    return a.scale_only ? launch_nl_mode<D, DV, false, 2, false, 2>(a, st) : launch_nl_mode<D, DV, true, 2, false, 2>(a, st);
}
template <int D, int DV>
static int launch_nl_tg(const NlArgs& a, hipStream_t st) {
    return a.scale_only ? launch_nl_mode<D, DV, false, 0, true>(a, st) : launch_nl_mode<D, DV, true, 0, true>(a, st);
}
template <int D, int DV>
static int launch_nl_x3(const NlArgs& a, hipStream_t st) {
    return a.scale_only ? launch_nl_mode<D, DV, false, 2>(a, st) : launch_nl_mode<D, DV, true, 2>(a, st);
}

// stream-K launch pair: the chunked attention kernel, then the combine pass (same stream)
template <int D, int DV, int MODE>
static int launch_nl_sk(const NlArgs& a, hipStream_t st) {
    constexpr size_t lds_t = (size_t)2 * 32 * (D + DV) * sizeof(float);
    constexpr size_t lds_m = (size_t)4 * 64 * ((DV / 64) * 16 + 2) * sizeof(float);
    constexpr size_t lds = lds_t > lds_m ? lds_t : lds_m;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = nl_attention_kernel<D, DV, true, MODE, false, 2, true>;
    static bool attr_set[64] = {};   // per device; benign race (idempotent call)
    int dev = 0;
    PTX_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        PTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(a.sk_chunks * a.batch)), dim3(512), lds, st, a);
    int rc = hip_check(hipGetLastError(), "nonlocal attention (stream-K) launch");
    if (rc != PTX_OK) return rc;
    hipLaunchKernelGGL((nl_sk_combine_kernel<DV>), dim3((unsigned)(a.q_tiles * a.batch * (DV / 16))), dim3(256), 0, st, a,
                       cdiv(a.Nk, 32), MODE == 2 ? 1.0f / 4096.0f : 1.0f);
    return hip_check(hipGetLastError(), "nonlocal attention (stream-K combine) launch");
}

// Chunks per clip of the stream-K form, or 0 when the plain kernels run: a function of PER-SAMPLE extents and the mode only.
// Covered: softmax attention on the key-split tiles (64 < d <= 256, dv <= 256, Nq > 512, Nk >= 512), 8 <= query tiles <= 64.
// OFF unless PTX_NL_STREAMK=1 (read at every call): measured SLOWER than the plain key-split kernel on MI355X -- at 8 clips x
// N = 1568 247 us against 233 (scripts/gpu_r05_attn_scaling.py, profiles/r05_attention_streamk.txt).  The plain kernel's 25
// workgroups of a clip walk the key tiles in lock-step (the first one pulls a tile into the XCD's L2, 24 hit it); 32 chunks
// start at 32 different key offsets, the clip's whole 3.2 MB of keys / values is live at once and a step's LDS-DMA no longer
// returns within one tile of prefetch: 4.7 -> 5.7 us per 32-key step, more than the 22 % of steps the chunks save.
static int nl_sk_chunks(const ptx_nonlocal_desc* d) {
    // the three knobs are read at EVERY call (tests and tuning sessions flip them inside one process)
    const char* e = getenv("PTX_NL_STREAMK");
    const char* ek = getenv("PTX_NL_KSPLIT");
    const char* ed = getenv("PTX_NL_DS");
    const int env = e ? atoi(e) : 0;
    const int ks_env = ek ? atoi(ek) : -1;
    const int ds_env = ed ? atoi(ed) : 1;
    if (!env || ks_env == 0) return 0;
    if (d->mode & (PTX_NL_SCALE | PTX_NL_F16 | PTX_NL_OUT_F16 | PTX_NL_RELU)) return 0;
    if (!(d->d > 64 && d->d <= 256 && d->dv <= 256) || d->Nk < 512) return 0;
    if (d->Nq <= 512 && ds_env) return 0;
    const int q_tiles = cdiv(d->Nq, 64);
    if (q_tiles < 8 || q_tiles > 64) return 0;
    const int chunks = q_tiles <= 32 ? 32 : 64;
    // a query tile's units span at most chunks / q_tiles whole chunks plus one partial chunk at either end
    if (chunks / q_tiles + 2 > kNlSkMaxPieces) return 0;
    return chunks;
}
static size_t nl_sk_slot_floats(const ptx_nonlocal_desc* d) { return (size_t)64 * (d->dv <= 128 ? 128 : 256) + 128; }

}  // namespace ptx

using namespace ptx;

extern "C" size_t ptx_nonlocal_workspace_bytes(const ptx_nonlocal_desc* d) {
    if (!d || !ptx_nonlocal_supported(d)) return 0;
    const int chunks = nl_sk_chunks(d);
    return chunks ? (size_t)d->batch * chunks * 2 * nl_sk_slot_floats(d) * sizeof(float) : 0;
}

extern "C" int ptx_nonlocal_supported(const ptx_nonlocal_desc* d) {
    if (!d) return 0;
    return d->batch > 0 && d->Nq > 0 && d->Nk > 0 && d->d > 0 && d->dv > 0 && d->d % 4 == 0 && d->dv % 4 == 0 && d->d <= 1024 &&
           d->batch * (int64_t)((d->Nq + 63) / 64) <= 0x7fffffffLL;
}

extern "C" int ptx_nonlocal_fwd(const ptx_nonlocal_desc* d, const float* theta, const float* phi, const float* g, float* y,
                                ptx_stream_t stream) {
    return ptx_nonlocal_ws_fwd(d, theta, phi, g, y, nullptr, 0, stream);
}

extern "C" int ptx_nonlocal_ws_fwd(const ptx_nonlocal_desc* d, const float* theta, const float* phi, const float* g, float* y,
                                   void* workspace, size_t workspace_bytes, ptx_stream_t stream) {
    if (!d || !theta || !phi || !g || !y) return fail(PTX_ERR_INVALID, "nonlocal: null pointer");
    if (!ptx_nonlocal_supported(d))
        return fail(PTX_ERR_UNSUPPORTED, "nonlocal: need d, dv multiples of 4 and d <= 1024 (d=%d dv=%d); use the "
                    "ptx_bgemm_nt / ptx_softmax_rows path", d->d, d->dv);
    if (d->ld_theta < d->d || d->ld_phi < d->d || d->ld_g < d->dv || d->ld_y < d->dv || d->ld_theta % 4 || d->ld_phi % 4 ||
        d->ld_g % 4 || d->ld_y % 4 || d->bs_theta % 4 || d->bs_phi % 4 || d->bs_g % 4 || d->bs_y % 4)
        return fail(PTX_ERR_INVALID, "nonlocal: row / batch strides must be multiples of 4 floats and cover the extents");
    if (((uintptr_t)theta | (uintptr_t)phi | (uintptr_t)g | (uintptr_t)y) & 15) return fail(PTX_ERR_INVALID, "nonlocal: misaligned pointer");
    if ((d->mode & ~(PTX_NL_SCALE | PTX_NL_F16 | PTX_NL_X3 | PTX_NL_RELU | PTX_NL_OUT_F16)) || ((d->mode & PTX_NL_F16) && (d->mode & PTX_NL_X3)) ||
        ((d->mode & PTX_NL_RELU) && !(d->mode & PTX_NL_SCALE)))
        return fail(PTX_ERR_INVALID, "nonlocal: unknown mode %d", d->mode);
    if ((d->mode & PTX_NL_OUT_F16) && !(d->mode & PTX_NL_F16))
        return fail(PTX_ERR_UNSUPPORTED, "nonlocal: PTX_NL_OUT_F16 (halfs out) goes with PTX_NL_F16, the generator's fp16 plan");
    const uint64_t tb = (uint64_t)d->Nq * d->ld_theta * 4ull, pb = (uint64_t)d->Nk * d->ld_phi * 4ull, gb = (uint64_t)d->Nk * d->ld_g * 4ull;
    if (tb >= 0x80000000ull || pb >= 0x80000000ull || gb >= 0x80000000ull)
        return fail(PTX_ERR_UNSUPPORTED, "nonlocal: one batch item of theta / phi / g must be < 2 GiB");
    NlArgs a{};
    a.theta = theta; a.phi = phi; a.g = g; a.y = y;
    a.batch = d->batch; a.Nq = d->Nq; a.Nk = d->Nk; a.d = d->d; a.dv = d->dv;
    a.ld_t = d->ld_theta; a.ld_p = d->ld_phi; a.ld_g = d->ld_g; a.ld_y = d->ld_y;
    a.bs_t = d->bs_theta; a.bs_p = d->bs_phi; a.bs_g = d->bs_g; a.bs_y = d->bs_y;
    a.q_tiles = cdiv(d->Nq, 64);
    a.scale_only = (d->mode & PTX_NL_SCALE) != 0;
    a.relu = (d->mode & PTX_NL_RELU) != 0;
    a.out_f16 = (d->mode & PTX_NL_OUT_F16) != 0;
    a.t_bytes = (unsigned)tb; a.p_bytes = (unsigned)pb; a.g_bytes = (unsigned)gb;
    hipStream_t st = (hipStream_t)stream;
    if (d->mode & PTX_NL_F16) {          // fp16-operand MFMAs: the generator's self-attention shape family only
        if (a.scale_only || d->d > 64)
            return fail(PTX_ERR_UNSUPPORTED, "nonlocal: PTX_NL_F16 covers softmax attention with d <= 64 (d=%d)", d->d);
        return d->dv <= 64 ? launch_nl_mode<64, 64, true, 1>(a, st) : launch_nl_mode<64, 256, true, 1>(a, st);
    }
    // d > 512 ('gaussian' mode at C = 1024): theta fragments come from global memory; fp32 MFMAs whatever the plan's
    // precision (the split-operand mode is fp32-accurate by contract, so the exact kernel is a valid stand-in)
    // short sequences: the d-split kernel (one 16-query group per workgroup, the channel axes spread over its waves).  Chosen
    // from PER-SAMPLE extents only, like everything below: a clip's bits must not depend on the batch it arrives in.
    // Exact fp32 MFMAs also under a split-operand plan (fp32-accurate by contract, as for d > 512).  PTX_NL_DS=0: off (A/B).
    const int ds_env = getenv("PTX_NL_DS") ? atoi(getenv("PTX_NL_DS")) : 1;          // read per call, like nl_sk_chunks
    if (ds_env && d->Nq <= 512 && d->d > 64 && d->d <= 512 && d->dv <= 512) {
        if (d->d <= 256 && d->dv <= 256) return launch_nl_ds<256, 256, 4>(a, st);
        return launch_nl_ds<512, 512, 8>(a, st);
    }
    if (d->d > 512) return launch_nl_tg<1024, 128>(a, st);
    // key split (two wave groups over the key tiles): when the plain grid is below two workgroups per CU -- i.e. one wave
    // per SIMD -- and there are enough keys to split.  PTX_NL_KSPLIT=0 / 1 forces it off / on (A/B runs).
    // The choice depends on PER-SAMPLE extents only (never on the batch): a clip's logits must not change bits with the
    // batch size or a rank's shard size (the reference is batch-independent, nonlocalnet.py:143-166).  At large
    // grids the two variants measure the same (8 waves per CU either way), so the key split is simply the kernel for
    // 64 < d <= 256 and Nk >= 128.
    const int ks_env = getenv("PTX_NL_KSPLIT") ? atoi(getenv("PTX_NL_KSPLIT")) : -1;
    const bool ksplit = ks_env != 0 && a.Nk >= 128;
    // stream-K over a caller-provided workspace (ptx_nonlocal_workspace_bytes): long sequences whose query tiles do not fill
    // the chip evenly.  Without a (large enough, 16-byte aligned) workspace the plain key-split kernel below runs.
    const int sk = nl_sk_chunks(d);
    if (sk && workspace && workspace_bytes >= ptx_nonlocal_workspace_bytes(d) && !((uintptr_t)workspace & 15)) {
        a.sk_chunks = sk;
        a.ws = static_cast<float*>(workspace);
        a.ws_slot = (long long)nl_sk_slot_floats(d);
        if (d->mode & PTX_NL_X3) return d->dv <= 128 ? launch_nl_sk<256, 128, 2>(a, st) : launch_nl_sk<256, 256, 2>(a, st);
        return d->dv <= 128 ? launch_nl_sk<256, 128, 0>(a, st) : launch_nl_sk<256, 256, 0>(a, st);
    }
    if (ksplit && d->d > 64 && d->d <= 256) {
        if (d->mode & PTX_NL_X3) return d->dv <= 128 ? launch_nl_ks_x3<256, 128>(a, st) : launch_nl_ks_x3<256, 256>(a, st);
        return d->dv <= 128 ? launch_nl_ks<256, 128>(a, st) : launch_nl_ks<256, 256>(a, st);
    }
    if (d->mode & PTX_NL_X3) {           // split operands: the same tile family as the fp32 kernel
        if (d->d <= 32 && d->dv <= 128 && d->dv > 64) return launch_nl_x3<32, 128>(a, st);
        if (d->d <= 64) return d->dv <= 64 ? launch_nl_x3<64, 64>(a, st) : launch_nl_x3<64, 256>(a, st);
        if (d->d <= 256) return d->dv <= 128 ? launch_nl_x3<256, 128>(a, st) : launch_nl_x3<256, 256>(a, st);
        return launch_nl_x3<512, 256>(a, st);
    }
    // smallest compiled (D, DV) covering the problem; dv > DV is split over blockIdx.y (S recomputed per chunk)
    if (d->d <= 32 && d->dv <= 128 && d->dv > 64) return launch_nl<32, 128>(a, st);
    if (d->d <= 64) return d->dv <= 64 ? launch_nl<64, 64>(a, st) : launch_nl<64, 256>(a, st);
    if (d->d <= 256) return d->dv <= 128 ? launch_nl<256, 128>(a, st) : launch_nl<256, 256>(a, st);
    return launch_nl<512, 256>(a, st);
}

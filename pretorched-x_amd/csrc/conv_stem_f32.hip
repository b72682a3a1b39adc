// RGB stem convolution on the fp32 matrix cores, straight from the caller's NCDHW tensor:
// Conv3d(3, 64, 7, stride (1,2,2), pad 3) + BN + ReLU of the ResNet3D family (resnet3D.py:153-155), the 2-D ResNet stem
// (torchvision_models.py), the (1,7,7) spatial stem of the (2+1)D nets (r2plus1d.py:73-88), the SAME-padded I3D stem.
//
// Why a dedicated kernel.  On the generic implicit-GEMM tiles the stem is a (7,7,1) conv over a kW-FOLDED copy of the
// input (ptx_fold_kw_ncdhw: 77 MB -> 308 MB at config 2), and every (kt, kh) tap re-stages its A tile global -> VGPR ->
// LDS behind a barrier: the fp32 MFMA pipe sits at 75 %, plus the 82 us fold pass.  Here a workgroup (4
// waves) owns 256 consecutive outputs of one output frame, stages the INPUT PATCH of a temporal tap once with LDS-DMA --
// three channel planes of PR rows x PC floats, exactly the caller's NCDHW rows (16-byte pieces, zero outside the image) --
// and serves all kH x 7 taps of that frame from it.  No fold, no layout pass: the kernel reads the user tensor.  LDS per
// workgroup = the patch + three 5.5 KiB filter slots (53 KB at config 2), 147 VGPRs: THREE workgroups share a CU and cover
// each other's frame-change DMA wait, barriers, prologue and epilogue -- what took the kernel from 1.76 to 1.55 ms
// (one 8-wave workgroup per CU could not be scheduled out of its bubbles).
//
// K axis of one (kt, kh) tap: 21 = 3 channels x 7 kw, issued as 11 v_mfma_f32_32x32x2_f32 (k = 2 per instruction; lanes
// 0-31 hold k0, lanes 32-63 hold k1 =: g).  Pairing keeps the g-dependence of the A address a constant:
//     j = 3c + p (j < 9): (c, kw = 2p + g)         -- one float apart
//     j = 9             : (c = g, kw = 6)          -- one plane apart
//     j = 10            : (c = 2, kw = 6) | zero
// so a fragment is one ds_read_b32 at (patch row r*sH + kh, column wo*sW + shift + kw) of plane c.  The filter comes
// from ptx_pack_stem_f32_weight in exactly that order ([tap][64-channel tile][11][2][64] floats, conflict-free reads).
// Arithmetic: fp32 operands, fp32 accumulate -- the reference's own.
#include "ptx_common.h"
#include <algorithm>

namespace ptx {

struct StemF32Args {
    const float* x;       // NCDHW (strides below, in floats)
    const float* w;       // [kT*kH][w_tiles][11][2][64]
    const float* bias;
    float* y;             // [N][To][Ho][Wo][ldy]
    int N, Ti, Hi, Wi, To, Ho, Wo, ldy, ncol;
    int pitch;            // floats per input row (>= Wi, multiple of 4; [Wi, pitch) are zero)
    int kT, kH, sT, sH, sW, pT, pH;
    int sn, sc, st;       // batch / channel / frame strides of x
    int PR, PC, plane;    // patch rows, floats per patch row (multiple of 4), PR * PC
    int shift, wbase;     // patch column 0 is input column wbase (<= 0, multiple of 4); a window starts at wo*sW + shift
    int tiles_per_frame, n_tiles, n_pieces, w_tiles;
    unsigned flags;
    unsigned x_bytes, w_bytes, y_bytes;
    unsigned dv_wo[2];
    unsigned dv_plane4[2], dv_pc4[2];     // fast division by plane / 4 and PC / 4 (the prologue's patch-piece decode)
    // longest-first tile order inside an XCD's chunk (0 = plain order): output frames grouped by their number of valid
    // temporal taps, most taps first -- the workgroups that finish a launch are the short ones
    int lpt, n_classes, groups_per_xcd;
    unsigned char cls_start[10];      // first entry of class k in `order` (n_classes + 1 entries)
    unsigned char order[64];          // output frames sorted by (taps desc, frame asc)
};

constexpr int kF32Waves = 4;                       // 4 waves x 64 rows; up to three workgroups share a CU
constexpr int kF32NT = 64 * kF32Waves;
constexpr int kF32Rows = 64 * kF32Waves;           // outputs per workgroup
constexpr int kF32BN = 64;                         // output channels per workgroup
constexpr int kF32PatchMax = 12288;                // floats of the patch buffer: 48 KiB
constexpr int kF32K2 = 11;                         // MFMAs per (kt, kh) tap
constexpr int kF32BTile = kF32K2 * 2 * kF32BN;     // floats of one (tap, channel tile) filter block: 5.5 KiB

__device__ __forceinline__ unsigned f32_fdiv(unsigned n, const unsigned (&dv)[2]) {
    return dv[0] ? (__umulhi(n, dv[0]) >> dv[1]) : n;
}
static inline void f32_fdiv_make(unsigned d, unsigned (&out)[2]) {
    if (d <= 1) { out[0] = 0; out[1] = 0; return; }
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    out[0] = (unsigned)(((1ull << (31 + l)) + d - 1) / d);
    out[1] = l - 1;
}

// the whole patch of temporal tap KT, global -> LDS: piece q = tid + 256 i lands at As + 16 q; lanes past the patch stay
// out of the DMA (nothing is written for them), pieces outside the image read as zero
#define PTX_STEM_DMA_PATCH(KT)                                                                                         \
    do {                                                                                                               \
        const unsigned fbase_ = (unsigned)(n * p.sn + (t_first + (KT)) * p.st) * 4u;                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < NP; ++i_)                                                              \
            if (tid + kF32NT * i_ < p.n_pieces)                                                                        \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(As + (wave * 64 + kF32NT * i_) * 4), 16,   \
                                                         a_src[i_] == kOOB ? kOOB : a_src[i_] + fbase_, 0, 0, 0);      \
    } while (0)

// Phase clock of a workgroup (diagnostic build only: scripts/micro/build_timeline.sh, -DPTX_STEM_TIMELINE; the product library
// carries none of it): thread 0 writes the 100 MHz wall clock at 0 entry, 1 requests issued, 2 first patch landed, 3 last
// step done, 5 stores retired; 6 = __smid().
#ifdef PTX_STEM_TIMELINE
__device__ unsigned long long* g_stem_tl = nullptr;
#define PTX_STEM_TL(k)                                                                                                 \
    do {                                                                                                               \
        if (threadIdx.x == 0 && g_stem_tl)                                                                             \
            g_stem_tl[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (k)] =                                       \
                (k) == 6 ? (unsigned long long)__smid() : (unsigned long long)wall_clock64();                          \
    } while (0)
#else
#define PTX_STEM_TL(k) do {} while (0)
#endif

// NP: 16-byte patch pieces per thread (4, 8 or 12).  STAGE: the next frame's patch waits in registers (2 workgroups per CU);
// else it is LDS-DMA'd at the frame change, exposed, and a third workgroup per CU covers the wait (<= 168 registers).
template <int NP, bool STAGE>
__global__ void __launch_bounds__(kF32NT, 2) conv_stem_f32_kernel(const StemF32Args p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    PTX_STEM_TL(0);
    PTX_STEM_TL(6);
    float* As = smem;                                   // [3 * plane]: the patch of the current temporal tap
    float* Bs = smem + 3 * p.plane;                     // [3][kF32BTile]: filter tiles of steps s, s + 1, s + 2
    constexpr unsigned kOOB = 0x80000000u;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // tile order: the frames of one band of rows follow each other (temporal L2 reuse of the kT-frame window), an XCD
    // owns a contiguous chunk of the list
    const int nt = blockIdx.y;
    int to, t_;                                         // output frame, (n, band) group
    if (p.lpt) {
        const int xcd = blockIdx.x % kNumXCD, l = blockIdx.x / kNumXCD;      // workgroup b runs on XCD b % 8
        int k = 0;
        while (k + 1 < p.n_classes && l >= (int)p.cls_start[k + 1] * p.groups_per_xcd) ++k;
        const int first = p.cls_start[k], cnt = p.cls_start[k + 1] - first;
        const int idx = l - first * p.groups_per_xcd;
        const int grp = idx / cnt;
        to = p.order[first + (idx - grp * cnt)];
        t_ = xcd * p.groups_per_xcd + grp;
    } else {
        const int tile = xcd_remap(blockIdx.x, p.n_tiles);
        to = tile % p.To;
        t_ = tile / p.To;
    }
    const int band = t_ % p.tiles_per_frame;
    const int n = t_ / p.tiles_per_frame;
    const int m0 = band * kF32Rows;                     // first output (raster index inside the frame)
    const int ho_a = (int)f32_fdiv((unsigned)m0, p.dv_wo);
    const int h_base = ho_a * p.sH - p.pH;

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);

    // ---- per-thread sources of the patch pieces (frame independent): piece q = tid + 256 i of the three planes ----
    unsigned a_src[12];       // (sized 12, used to NP: a template-dependent extent here makes hipcc's host pass drop the kernel stub)
    // (multiply-shift divisions: two plain integer divisions per piece were ~1000 VALU instructions per thread, and VALU
    //  work shares the fp32 datapath with the co-resident workgroups' MFMAs -- the phase clock, scripts/gpu_stem_timeline.py,
    //  showed 22 us of a 177-us workgroup life before the first request went out, 9.6 of 34 us on the (1,7,7) stem)
    const int pc4 = p.PC >> 2, plane4 = p.plane >> 2;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int q = tid + kF32NT * i;
        const int c = (int)f32_fdiv((unsigned)q, p.dv_plane4);
        const int rem = q - c * plane4;
        const int pr = (int)f32_fdiv((unsigned)rem, p.dv_pc4);
        const int h = h_base + pr, w = (rem - pr * pc4) * 4 + p.wbase;
        const bool ok = c < 3 && (unsigned)h < (unsigned)p.Hi && (unsigned)w < (unsigned)p.Wi;
        a_src[i] = ok ? (unsigned)((c * p.sc + h * p.pitch + w) * 4) : kOOB;
    }
    unsigned b_src[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) b_src[i] = (unsigned)((nt * kF32BTile + (tid + kF32NT * i) * 4) * 4);

    // ---- valid temporal taps (uniform): frames outside the clip contribute nothing ----
    const int t_first = to * p.sT - p.pT;
    const int kt_lo = max(0, -t_first), kt_hi = min(p.kT - 1, p.Ti - 1 - t_first);
    const int n_kt = kt_hi - kt_lo + 1;

    // the patch of the NEXT frame waits in registers (global -> VGPR while this frame computes, VGPR -> LDS at the frame
    // change): one patch buffer instead of two keeps the workgroup at 66 KiB of LDS, so two of them share a CU and fill
    // each other's barrier / frame-change / epilogue bubbles
    f32x4 preg[NP];
    auto fetch_patch = [&](int kt) {
        const unsigned fbase = (unsigned)(n * p.sn + (t_first + kt) * p.st) * 4u;
#pragma unroll
        for (int i = 0; i < NP; ++i)
            preg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, a_src[i] == kOOB ? kOOB : a_src[i] + fbase, 0, 0));
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int i = 0; i < NP; ++i)
            if (tid + kF32NT * i < p.n_pieces) *reinterpret_cast<f32x4*>(As + (tid + kF32NT * i) * 4) = preg[i];
    };
    auto issue_b = [&](int buf, int kt, int kh) {
        const unsigned tbase = (unsigned)((kt * p.kH + kh) * p.w_tiles * kF32BTile * 4);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (tid + kF32NT * i < kF32BTile / 4)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(Bs + buf * kF32BTile + (wave * 64 + kF32NT * i) * 4), 16,
                                                         b_src[i] + tbase, 0, 0, 0);
    };

    // ---- this lane's output rows: ml = m0 + wave * 64 + i * 32 + lane % 32 -> (ho, wo) ----
    const int g = lane >> 5, l32 = lane & 31;
    const int frame_out = p.Ho * p.Wo;
    int a_row[2];            // float offset of (patch row (ho - ho_a) * sH, column wo * sW + shift)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ml = m0 + wave * 64 + i * 32 + l32;
        const int mm = ml < frame_out ? ml : m0;
        const int ho = (int)f32_fdiv((unsigned)mm, p.dv_wo);
        const int wo = mm - ho * p.Wo;
        a_row[i] = ((ho - ho_a) * p.sH) * p.PC + wo * p.sW + p.shift;
    }
    // K pairing (see the header): five address flavours per row tile, fixed for the whole kernel -- a step only adds its
    // kh offset, so the loop spends well under one VALU instruction per MFMA on addresses
    int a_base[2][5];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a_base[i][0] = a_row[i] + g;                      // j = 0..2 : plane 0, kw = 2p + g
        a_base[i][1] = a_row[i] + g + p.plane;            // j = 3..5 : plane 1
        a_base[i][2] = a_row[i] + g + 2 * p.plane;        // j = 6..8 : plane 2
        a_base[i][3] = a_row[i] + g * p.plane + 6;        // j = 9    : (c = g, kw = 6)
        a_base[i][4] = a_row[i] + 2 * p.plane + 6;        // j = 10   : (c = 2, kw = 6) | g = 1 multiplies a zero
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- main loop: one step per (kt, kh) tap.  The fragment registers of k-pair group q are refilled for step s + 1 right
    // after the MFMAs of group q of step s have issued, so inside a frame no MFMA waits on LDS.  Needs filter tile s + 1
    // landed at barrier(s): three filter slots.
    const int n_steps = n_kt * p.kH;
    float fa[2][kF32K2], fb[2][kF32K2];
    auto load_group = [&](int q, const float* Ab, const float* Bb) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int j = q * 3 + t;
            if (j < kF32K2) {
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i][j] = j < 9 ? Ab[a_base[i][j / 3] + 2 * (j % 3)] : Ab[a_base[i][j - 6]];
                fb[0][j] = Bb[j * 2 * kF32BN];
                fb[1][j] = Bb[j * 2 * kF32BN + 32];
            }
        }
    };
    auto mma_group = [&](int q) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int j = q * 3 + t;
            if (j < kF32K2) {
                float a0 = fa[0][j], a1 = fa[1][j];
                if (j == 10) { a0 = g ? 0.f : a0; a1 = g ? 0.f : a1; }
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, fb[0][j], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, fb[1][j], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, fb[0][j], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, fb[1][j], acc[1][1], 0, 0, 0);
            }
        }
    };
    if (n_steps > 0) {
        // prologue: the first patch (through registers), filter tiles 0 and 1; then the fragments of step 0
        if (STAGE) fetch_patch(kt_lo);
        else PTX_STEM_DMA_PATCH(kt_lo);
        issue_b(0, kt_lo, 0);
        if (n_steps > 1) issue_b(1, kt_lo, 1);           // (kH >= 2)
        if (STAGE) store_patch();
        PTX_STEM_TL(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        PTX_STEM_TL(2);
        asm volatile("; LDS reads stay below the barrier" : "+v"(a_base[0][0]), "+v"(a_base[1][0])::"memory");
#pragma unroll
        for (int q = 0; q < 4; ++q) load_group(q, As, Bs + g * kF32BN + l32);
        int ik = 0, kh = 0, slot = 0;                     // state of step s: frame, row tap, filter slot
        for (int s = 0; s < n_steps; ++s) {
            if (s > 0) {
                // filter tile s + 1 (issued during step s - 1) has landed for everyone; slot (s + 2) % 3 is free: its last
                // reads were issued during step s - 2 and consumed during step s - 1
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            int kh1 = kh + 1, ik1 = ik;                   // (kt, kh) of steps s + 1 and s + 2
            if (kh1 == p.kH) { kh1 = 0; ++ik1; }
            int kh2 = kh1 + 1, ik2 = ik1;
            if (kh2 == p.kH) { kh2 = 0; ++ik2; }
            const int slot1 = slot == 2 ? 0 : slot + 1;
            const int slot2 = slot1 == 2 ? 0 : slot1 + 1;
            if (s + 2 < n_steps) issue_b(slot2, kt_lo + ik2, kh2);
            if (STAGE && kh == 0 && ik + 1 < n_kt) fetch_patch(kt_lo + ik + 1);      // lands in registers during this frame
            const bool more = s + 1 < n_steps;
            const bool same_frame = kh1 != 0;
            const float* Bb = Bs + slot1 * kF32BTile + g * kF32BN + l32;
            const bool prefetch = more && same_frame;
            const float* Ab = As + kh1 * p.PC;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                mma_group(q);
                __builtin_amdgcn_sched_barrier(0);
                if (prefetch) load_group(q, Ab, Bb);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more && !same_frame) {
                // frame change: every wave is done with the old patch; the staged one moves VGPR -> LDS
                __syncthreads();
                if (STAGE) store_patch();
                else {
                    PTX_STEM_DMA_PATCH(kt_lo + ik1);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __syncthreads();
                asm volatile("; LDS reads stay below the barrier" : "+v"(a_base[0][0]), "+v"(a_base[1][0])::"memory");
#pragma unroll
                for (int q = 0; q < 4; ++q) load_group(q, As, Bb);
            }
            kh = kh1; ik = ik1; slot = slot1;
        }
    }

    // ---- epilogue: bias (+ folded BN) + ReLU; lane = output channel, 16 rows per accumulator tile.  Outputs of a tile are
    // consecutive in (n, to, ho, wo) raster order, so the row index is arithmetic ----
    PTX_STEM_TL(3);
    const bool relu = (p.flags & PTX_EPI_RELU) != 0;
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const int m_frame = (n * p.To + to) * frame_out;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int co = nt * kF32BN + j * 32 + l32;
        const bool co_ok = co < p.ncol;
        const float bv = (p.bias && co_ok) ? p.bias[co] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // accumulator element r of this lane belongs to tile row (r & 3) + 8 * (r >> 2) + 4 * g
                const int ml = m0 + wave * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                float v = acc[i][j][r] + bv;
                v = relu ? fmaxf(v, 0.f) : v;
                const unsigned off = ((unsigned)(m_frame + ml) * (unsigned)p.ldy + (unsigned)co) * 4u;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_y, (co_ok && ml < frame_out) ? off : kOOB, 0, 0);
            }
        }
    }
#ifdef PTX_STEM_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PTX_STEM_TL(5);
#endif
}


template <int NP, bool STAGE>
static int launch_stem_f32(const StemF32Args& a, dim3 grid, size_t lds, ptx_stream_t stream) {
    static bool attr_set[64] = {};
    int dev = 0;
    PTX_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        PTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_stem_f32_kernel<NP, STAGE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)((kF32PatchMax + 3 * kF32BTile) * sizeof(float))));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_stem_f32_kernel<NP, STAGE>), grid, dim3(kF32NT), lds, (hipStream_t)stream, a);
    return PTX_OK;
}

// folded K-major filter [tap][Co_pad][Kc] (k = kw * 3 + c, ptx_pack_conv_weight with fold_kw = 1) -> the stem's
// [tap][Co_pad / 64][11][2][64] blocks
__global__ void __launch_bounds__(256) pack_stem_f32_kernel(const float* __restrict__ wf, float* __restrict__ out, int taps,
                                                            int Co_pad, int Kc, int total) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int col = i & 63;
        int r = i >> 6;
        const int g = r & 1;
        r >>= 1;
        const int j = r % kF32K2;
        r /= kF32K2;
        const int tiles = Co_pad / kF32BN;
        const int nt = r % tiles;
        const int tap = r / tiles;
        int c, kw;
        if (j < 9) { c = j / 3; kw = 2 * (j % 3) + g; }
        else if (j == 9) { c = g; kw = 6; }
        else { c = 2; kw = 6; }
        const bool zero = j == 10 && g == 1;
        out[i] = zero ? 0.f : wf[((size_t)tap * Co_pad + nt * kF32BN + col) * Kc + kw * 3 + c];
    }
}

struct StemF32Geom {
    int PR, PC, shift, wbase, tiles_per_frame;
};

static bool stem_f32_geom(const ptx_conv3d_desc* d, StemF32Geom* g) {
    const int frame = d->Ho * d->Wo;
    // rows a kF32Rows-output raster span can touch
    const int nrows = std::min(d->Ho, (kF32Rows - 1 + d->Wo - 1) / d->Wo + 1);
    g->PR = (nrows - 1) * d->sH + d->kH;
    g->wbase = -((d->pW + 3) / 4 * 4);
    g->shift = -g->wbase - d->pW;
    g->PC = ((d->Wo - 1) * d->sW + g->shift + d->kW + 3) / 4 * 4;
    g->tiles_per_frame = cdiv(frame, kF32Rows);
    return (int64_t)3 * g->PR * g->PC <= kF32PatchMax;
}

}  // namespace ptx

using namespace ptx;

#ifdef PTX_STEM_TIMELINE
extern "C" int ptx_stem_f32_timeline(void* buf) {  // diagnostic build only: 8 x u64 per workgroup, or NULL to switch off
    return hip_check(hipMemcpyToSymbol(HIP_SYMBOL(g_stem_tl), &buf, sizeof(buf)), "ptx_stem_f32_timeline");
}
#endif

extern "C" int ptx_conv_stem_f32_supported(const ptx_conv3d_desc* d, int64_t stride_n, int64_t stride_c, int64_t stride_t) {
    if (!d) return 0;
    if (d->flags & ~PTX_EPI_RELU) return 0;
    if (d->Ci != 3 || d->kW != 7 || d->kT < 1 || d->kT > 8 || d->kH < 2 || d->kH > 8 || d->groups > 1 || d->Co_pad % kF32BN) return 0;
    if (d->sW < 1 || d->sW > 2 || d->sH < 1 || d->sT < 1 || d->Wo < 1 || d->Ho < 1 || d->To < 1 || d->pW < 0 || d->pW > 8) return 0;
    const int pitch = d->ldx > 0 ? d->ldx : d->Wi;             // floats per input row
    if (pitch % 4 || pitch < d->Wi || stride_n % 4 || stride_c % 4 || stride_t % 4) return 0;   // 16-byte DMA pieces of NCDHW rows
    // a plane inside its channel, a channel inside its sample: the buffer descriptor's range is derived from these
    const int64_t chan_extent = (int64_t)(d->Ti - 1) * stride_t + (int64_t)d->Hi * pitch;
    if (stride_t < (int64_t)d->Hi * pitch || stride_c < chan_extent || stride_n < 2 * stride_c + chan_extent) return 0;
    StemF32Geom g;
    if (!stem_f32_geom(d, &g)) return 0;
    if (((int64_t)(d->N - 1) * stride_n + 3 * stride_c) * 4 >= 0x80000000LL ||
        (int64_t)d->N * d->To * d->Ho * d->Wo * d->ldy * 4 >= 0x80000000LL)
        return 0;
    auto extent_ok = [](int in, int out, int k, int s, int p) {
        if (out == (in + 2 * p - k) / s + 1) return true;
        const int same = (in + s - 1) / s;
        const int total = std::max((same - 1) * s + k - in, 0);
        return out == same && p == total / 2;
    };
    if (!extent_ok(d->Wi, d->Wo, d->kW, d->sW, d->pW) || !extent_ok(d->Hi, d->Ho, d->kH, d->sH, d->pH) ||
        !extent_ok(d->Ti, d->To, d->kT, d->sT, d->pT))
        return 0;
    return 1;
}

extern "C" size_t ptx_stem_f32_weight_elems(const ptx_conv3d_desc* d) {
    if (!d || d->Co_pad <= 0 || d->Co_pad % kF32BN || d->kT <= 0 || d->kH <= 0) return 0;
    return (size_t)d->kT * d->kH * (d->Co_pad / kF32BN) * kF32BTile;
}

extern "C" int ptx_pack_stem_f32_weight(const ptx_conv3d_desc* d, const float* w_folded, int Kc, float* w_stem, ptx_stream_t stream) {
    if (!d || !w_folded || !w_stem) return fail(PTX_ERR_INVALID, "pack_stem_f32: null pointer");
    if (d->Ci != 3 || d->kW != 7 || Kc < 21 || d->Co_pad % kF32BN || d->kT <= 0 || d->kH <= 0)
        return fail(PTX_ERR_UNSUPPORTED, "pack_stem_f32: a folded 3-channel, kW = 7 filter with rows of >= 21 floats");
    const size_t total = ptx_stem_f32_weight_elems(d);
    if (total >= (1ull << 31)) return fail(PTX_ERR_UNSUPPORTED, "pack_stem_f32: filter too large");
    hipLaunchKernelGGL(pack_stem_f32_kernel, dim3((unsigned)std::min<size_t>(cdiv(total, 256), 4096)), dim3(256), 0, (hipStream_t)stream,
                       w_folded, w_stem, d->kT * d->kH, d->Co_pad, Kc, (int)total);
    return hip_check(hipGetLastError(), "pack_stem_f32 launch");
}

extern "C" int ptx_conv_stem_f32_fwd(const ptx_conv3d_desc* d, const float* x, int64_t stride_n, int64_t stride_c, int64_t stride_t,
                                     const float* w_stem, const float* bias, float* y, ptx_stream_t stream) {
    if (!d || !x || !w_stem || !y) return fail(PTX_ERR_INVALID, "conv_stem_f32: null pointer");
    if (((uintptr_t)x | (uintptr_t)w_stem | (uintptr_t)y) & 15) return fail(PTX_ERR_INVALID, "conv_stem_f32: pointers must be 16-byte aligned");
    if (!ptx_conv_stem_f32_supported(d, stride_n, stride_c, stride_t))
        return fail(PTX_ERR_UNSUPPORTED, "conv_stem_f32: needs a 3-channel NCDHW input (row pitch and strides multiples of 4 floats), kW == 7, "
                    "stride_w <= 2, symmetric or SAME padding, only the ReLU epilogue, and an input patch of at most %d floats", kF32PatchMax);
    if (d->ldy < d->Co || d->ldy % 4) return fail(PTX_ERR_INVALID, "conv_stem_f32: bad output stride");
    StemF32Geom g;
    stem_f32_geom(d, &g);
    StemF32Args a{};
    a.x = x; a.w = w_stem; a.bias = bias; a.y = y;
    a.N = d->N; a.Ti = d->Ti; a.Hi = d->Hi; a.Wi = d->Wi; a.To = d->To; a.Ho = d->Ho; a.Wo = d->Wo; a.ldy = d->ldy;
    a.ncol = (d->Co + 3) / 4 * 4;
    a.kT = d->kT; a.kH = d->kH; a.sT = d->sT; a.sH = d->sH; a.sW = d->sW; a.pT = d->pT; a.pH = d->pH;
    a.sn = (int)stride_n; a.sc = (int)stride_c; a.st = (int)stride_t;
    a.pitch = d->ldx > 0 ? d->ldx : d->Wi;
    a.PR = g.PR; a.PC = g.PC; a.plane = g.PR * g.PC; a.shift = g.shift; a.wbase = g.wbase;
    a.tiles_per_frame = g.tiles_per_frame;
    a.n_tiles = d->N * d->To * g.tiles_per_frame;
    a.n_pieces = 3 * a.plane / 4;                           // 16-byte pieces of the patch (PC % 4 == 0)
    a.w_tiles = d->Co_pad / kF32BN;
    a.flags = d->flags;
    a.x_bytes = (unsigned)(((uint64_t)(d->N - 1) * stride_n + 3ull * stride_c) * 4ull);
    a.w_bytes = (unsigned)(ptx_stem_f32_weight_elems(d) * 4ull);
    a.y_bytes = (unsigned)((uint64_t)d->N * d->To * d->Ho * d->Wo * d->ldy * 4ull);
    f32_fdiv_make((unsigned)d->Wo, a.dv_wo);
    f32_fdiv_make((unsigned)(a.plane / 4), a.dv_plane4);
    f32_fdiv_make((unsigned)(a.PC / 4), a.dv_pc4);
    {
        const int groups = d->N * g.tiles_per_frame;
        static const bool lpt_env = !(getenv("PTX_STEM_F32_LPT") && atoi(getenv("PTX_STEM_F32_LPT")) == 0);
        if (lpt_env && d->To <= 64 && d->To > 1 && groups % kNumXCD == 0) {
            int taps[64], idx[64];
            for (int to = 0; to < d->To; ++to) {
                const int t0 = to * d->sT - d->pT;
                taps[to] = std::max(0, std::min(d->kT - 1, d->Ti - 1 - t0) - std::max(0, -t0) + 1);
                idx[to] = to;
            }
            std::stable_sort(idx, idx + d->To, [&](int x, int y) { return taps[x] > taps[y]; });
            int nc = 0;
            for (int i = 0; i < d->To; ++i) {
                a.order[i] = (unsigned char)idx[i];
                if (i == 0 || taps[idx[i]] != taps[idx[i - 1]]) {
                    if (nc < 9) a.cls_start[nc++] = (unsigned char)i;
                    else { nc = 99; break; }
                }
            }
            if (nc > 1 && nc <= 9) {          // (one class: nothing to reorder)
                a.cls_start[nc] = (unsigned char)d->To;
                a.n_classes = nc;
                a.groups_per_xcd = groups / kNumXCD;
                a.lpt = 1;
            }
        }
    }
    const size_t lds = (size_t)(3 * a.plane + 3 * kF32BTile) * sizeof(float);
    const int np = cdiv(a.n_pieces, kF32NT);                // pieces per thread: 4 / 8 / 12
    // default: the patch is LDS-DMA'd at the frame change and three workgroups per CU cover each other's waits (config 2:
    // 1.54 ms); PTX_STEM_F32_MODE=2 stages the next patch through registers instead (two workgroups per CU: 1.62 ms)
    static const int mode_env = getenv("PTX_STEM_F32_MODE") ? atoi(getenv("PTX_STEM_F32_MODE")) : 1;
    const bool stage = mode_env == 2;
    const dim3 grid((unsigned)a.n_tiles, (unsigned)cdiv(a.ncol, kF32BN));
    int rc;
    if (stage) rc = np <= 4 ? launch_stem_f32<4, true>(a, grid, lds, stream) : np <= 8 ? launch_stem_f32<8, true>(a, grid, lds, stream)
                                                                                      : launch_stem_f32<12, true>(a, grid, lds, stream);
    else rc = np <= 4 ? launch_stem_f32<4, false>(a, grid, lds, stream) : np <= 8 ? launch_stem_f32<8, false>(a, grid, lds, stream)
                                                                                  : launch_stem_f32<12, false>(a, grid, lds, stream);
    if (rc != PTX_OK) return rc;
    return hip_check(hipGetLastError(), "conv_stem_f32 launch");
}

// 3x3x3 body convolution on the fp32 matrix cores with the input patch RESIDENT in LDS:
// Conv3d(C, Co, 3, stride 1, pad 1) + BN (+ residual) + ReLU of the ResNet3D family -- `conv2` of every Bottleneck
// (resnet3D.py:117,129-131: conv3x3x3 -> bn2 -> relu) and both convs of a BasicBlock (resnet3D.py:86-104) -- the layers that
// hold 48 % of config 2's FLOPs (C3 / C11 / C17 ...).
//
// Why a dedicated kernel.  On the generic implicit-GEMM tiles every one of the 27 taps re-stages its A tile
// global -> LDS behind a barrier (27 x the activation through L2, ~37 GB/s of LDS fill per CU) and the launches sit at 78 % of
// the MFMA pipe where the stem, with its patch-resident design, sits at 87 %.  Here a workgroup owns BM consecutive output
// positions of ONE output frame (raster order) x 64 output channels and walks K in PHASES (kt, 16-channel chunk):
//   * the input PATCH of a phase -- the PR x PC halo'd positions of input frame t + kt - 1 the tile's outputs can touch, 16
//     channels each -- is brought in ONCE by LDS-DMA (16-byte pieces of the NDHWC rows, zero outside the image) and serves
//     all nine (kh, kw) taps of that frame: 6x less LDS fill than the generic tile at BM = 256;
//   * the filter slice of a step (one tap x 16 channels x 64 columns = 4 KiB) comes through a three-slot ring, requested two
//     steps ahead; the fragments of step s + 1 are read while step s multiplies (one barrier per step, as in the stem);
//   * LDS per workgroup = patch (<= 29.7 KiB) + ring (12 KiB): THREE workgroups share a CU and cover each other's phase-change
//     DMA waits, barriers, prologue and epilogue (what took the stem from 1.76 to 1.55 ms).
// LDS images.  Patch: [4 channel groups][NPOS positions][4 floats] -- lane (row l, half g) reads the 16 bytes of group 2q + g at
// its position with ONE ds_read_b128: channels 8q + 4g + {0..3} = the k0 / k1 operands of four v_mfma_f32_32x32x2_f32
// (pairs (8q + i, 8q + 4 + i)); consecutive lanes are consecutive positions, so the 16-lane groups of a b128 read are
// conflict-free.  Filter slice: [2 octets][2 halves][64 columns][4 floats] in exactly that pairing (ptx_pack_conv_body_f32_weight).
// Arithmetic: fp32 operands, fp32 accumulate -- the reference's own; the k order differs from the generic tiles' (channel
// pairs (i, i + 4) inside an octet), so results agree with them to fp32 reorder noise (1e-6 relative), not bit for bit.
//
// Two tile shapes share the skeleton (template Shape): the TALL shape, 4 waves x (64 rows x 64 columns), BM = 256, for
// frames large enough to give >= 3 workgroups per CU; the SQUARE shape, 2 x 2 waves x (32 x 32), BM = 64, for the tail of a
// frame (a tall tile there would leave three of four waves idle on ONE SIMD's worth of work) and for small frames.  A launch
// is a list of tall tiles followed by square tiles (conv_body_f32_kernel) or square tiles only (conv_body_f32_sq_kernel).
#include "ptx_common.h"
#include <algorithm>

namespace ptx {

constexpr int kB3NT = 256;                 // threads per workgroup (4 waves)
constexpr int kB3BN = 64;                  // output channels per workgroup
constexpr int kB3CK = 16;                  // channels per phase
constexpr int kB3Slot = kB3CK * kB3BN;     // floats of one filter slice (tap x 16 channels x 64 columns): 4 KiB
constexpr int kB3PosMax = 472;             // positions of the largest patch the tall shape may stage (x 64 B = 29.5 KiB)

struct BodyF32Args {
    const float* x;        // NDHWC, ldx floats per position
    const float* w;        // [n tile][kt][chunk][kh][kw][2][2][64][4]
    const float* bias;
    const float* res;      // same shape as y (row stride ldr) or null
    float* y;              // [N][T][H][W][ldy]
    int N, T, H, W, C, ldx, ldy, ldr, ncol;
    int kT, pT;            // 3 / 1 (or 1 / 0: a (1,3,3) conv)
    int chunks;            // C / 16
    int PC;                // patch columns = W + 2
    int tall_per_frame;    // tall tiles (256 outputs) per frame
    int sq_per_frame;      // square tiles (64 outputs) per frame, covering [tall_per_frame * 256, H * W)
    int n_tall, n_sq;      // tiles of the whole launch
    int PR_tall, PR_sq;    // patch rows of the two shapes
    unsigned flags;
    unsigned x_bytes, w_bytes, y_bytes, r_bytes;
    unsigned dv_w[2], dv_pc[2], dv_npos_tall[2], dv_npos_sq[2];
    // chained 1x1x1 tail (ptx_conv_body_chain_f32_fwd): y = epi2(relu?(conv(x) + bias) . w2 + bias2 (+ res)); then `y`, `ldy`,
    // `ncol`, `res`, `ldr`, `y_bytes`, `r_bytes` above describe the TAIL's output and `flags` its epilogue
    const float* w2;       // [N2 / 64][8 octets][2][64][4]
    const float* bias2;
    int ncol1;             // columns of the first conv (<= 64)
    unsigned flags1;       // PTX_EPI_RELU between the two convs
    unsigned w2_bytes;
    // T-stacked tile of a (kT,1,1) temporal conv (conv_tstack_f32_kernel): tiles of 8 frames x 32 positions
    int HW, t_tiles, s_tiles;
};

constexpr int kB3ParkStride = 32 * 4 + 4;            // floats between the 4-channel groups of a parked 32-row tile (+4: bank spread)
constexpr int kB3Park = 16 * kB3ParkStride;          // floats of one parked [32 rows][64 channels] tile: 8.25 KiB

__device__ __forceinline__ unsigned b3_fdiv(unsigned n, const unsigned (&dv)[2]) {
    return dv[0] ? (__umulhi(n, dv[0]) >> dv[1]) : n;
}
static inline void b3_fdiv_make(unsigned d, unsigned (&out)[2]) {
    if (d <= 1) { out[0] = 0; out[1] = 0; return; }
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    out[0] = (unsigned)(((1ull << (31 + l)) + d - 1) / d);
    out[1] = l - 1;
}

// One tile.  WM x WN waves, each RT x CT MFMA tiles of 32 x 32; NP = 16-byte patch pieces per thread.
template <int WM, int WN, int RT, int CT, int NP, bool CHAIN>
__device__ __forceinline__ void conv_body_tile(const BodyF32Args& p, float* smem, const int n, const int to, const int m0, const int nt,
                                               const int PR, const unsigned (&dv_npos)[2]) {
    static_assert(WM * WN == 4, "four waves");
    static_assert(WN * CT * 32 == kB3BN, "a workgroup covers 64 output channels");
    constexpr unsigned kOOB = 0x80000000u;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    constexpr int BM = WM * RT * 32;
    const int NPOS = PR * p.PC;                         // positions of the patch
    float* As = smem;                                   // [4][NPOS][4]
    float* Bs = smem + 16 * NPOS;                       // [3][kB3Slot]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int g = lane >> 5, l32 = lane & 31;
    const int frame = p.H * p.W;
    const int ho_a = (int)b3_fdiv((unsigned)m0, p.dv_w);
    const int h_base = ho_a - 1;                        // input row of patch row 0

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);

    // ---- per-thread sources of the patch pieces (phase independent): piece q = tid + 256 i = group * NPOS + position ----
    unsigned a_src[NP];
    const int n_pieces = 4 * NPOS;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int q = tid + kB3NT * i;
        const int grp = (int)b3_fdiv((unsigned)q, dv_npos);
        const int pos = q - grp * NPOS;
        const int pr = (int)b3_fdiv((unsigned)pos, p.dv_pc);
        const int h = h_base + pr, w = pos - pr * p.PC - 1;
        const bool ok = grp < 4 && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
        a_src[i] = ok ? (unsigned)(((h * p.W + w) * p.ldx + grp * 4) * 4) : kOOB;
    }
    const unsigned b_src = (unsigned)(tid * 16);

    // ---- valid temporal taps (uniform): frames outside the clip contribute nothing ----
    const int t_first = to - p.pT;
    const int kt_lo = max(0, -t_first), kt_hi = min(p.kT - 1, p.T - 1 - t_first);
    const int n_phases = (kt_hi - kt_lo + 1) * p.chunks;
    const int n_steps = n_phases * 9;

    auto dma_patch = [&](int ph) {
        const int ikt = ph / p.chunks, ch = ph - ikt * p.chunks;
        const unsigned fbase = (unsigned)(((n * p.T + t_first + kt_lo + ikt) * frame) * p.ldx + ch * kB3CK) * 4u;
#pragma unroll
        for (int i = 0; i < NP; ++i)
            if (tid + kB3NT * i < n_pieces)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(As + (wave * 64 + kB3NT * i) * 4), 16,
                                                         a_src[i] == kOOB ? kOOB : a_src[i] + fbase, 0, 0, 0);
    };
    // filter slice of step s = (phase, tap): [nt][kt][chunk][tap] blocks of kB3Slot floats
    auto dma_b = [&](int slot, int s) {
        const int ph = s / 9, tap = s - ph * 9;
        const int ikt = ph / p.chunks, ch = ph - ikt * p.chunks;
        const unsigned tbase = (unsigned)((((nt * p.kT + kt_lo + ikt) * p.chunks + ch) * 9 + tap) * (kB3Slot * 4));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(Bs + slot * kB3Slot + wave * 256), 16, b_src + tbase, 0, 0, 0);
    };

    // ---- this lane's rows: row tile i covers tile rows (wm * RT + i) * 32 + l32 ----
    int a_row[RT];           // float offset of (group g, position of the output + tap (0, 0)) inside the patch
    bool any_valid = false;
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int ml = m0 + (wm * RT + i) * 32 + l32;
        const int mm = ml < frame ? ml : m0;
        const int ho = (int)b3_fdiv((unsigned)mm, p.dv_w);
        const int wo = mm - ho * p.W;
        a_row[i] = (g * NPOS + (ho - ho_a) * p.PC + wo) * 4;
        any_valid |= (m0 + (wm * RT + i) * 32) < frame;
    }
    // (a wave whose rows all lie beyond the frame still takes part in the DMA and the barriers, but multiplies nothing)
    const bool active = __builtin_amdgcn_readfirstlane((int)any_valid) != 0;

    f32x16 acc[RT][CT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragments of one step: two octets (q) x RT row tiles / CT column tiles, 16 bytes each
    f32x4 fa[2][RT], fb[2][CT];
    const int b_lane = (g * 64 + wn * CT * 32 + l32) * 4;           // + q * 512 + j * 128 floats
    auto load_group = [&](int q, const float* Ab, const float* Bb) {
#pragma unroll
        for (int i = 0; i < RT; ++i) fa[q][i] = *reinterpret_cast<const f32x4*>(Ab + a_row[i] + q * 8 * NPOS);
#pragma unroll
        for (int j = 0; j < CT; ++j) fb[q][j] = *reinterpret_cast<const f32x4*>(Bb + b_lane + q * 512 + j * 128);
    };
    auto mma_group = [&](int q) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < CT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][i][k], fb[q][j][k], acc[i][j], 0, 0, 0);
    };

    if (n_steps > 0) {
        dma_patch(0);
        dma_b(0, 0);
        dma_b(1, 1);                                    // (n_steps >= 9)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        asm volatile("; LDS reads stay below the barrier" : "+v"(a_row[0])::"memory");
        if (active) {
            load_group(0, As, Bs);
            load_group(1, As, Bs);
        }
        int tap = 0, slot = 0, ph = 0;
        for (int s = 0; s < n_steps; ++s) {
            if (s > 0) {
                // filter slice s + 1 (requested during step s - 1) has landed for everyone; slot (s + 2) % 3 is free: its last
                // reads were issued during step s - 2 and consumed during step s - 1
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            const int slot1 = slot == 2 ? 0 : slot + 1;
            const int slot2 = slot1 == 2 ? 0 : slot1 + 1;
            if (s + 2 < n_steps) dma_b(slot2, s + 2);
            const bool more = s + 1 < n_steps;
            const int tap1 = tap == 8 ? 0 : tap + 1;
            const bool same_phase = tap1 != 0;
            const int kh1 = tap1 / 3, kw1 = tap1 - kh1 * 3;
            const float* Ab = As + (kh1 * p.PC + kw1) * 4;
            const float* Bb = Bs + slot1 * kB3Slot;
            const bool prefetch = more && same_phase && active;
            if (active) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    mma_group(q);
                    __builtin_amdgcn_sched_barrier(0);
                    if (prefetch) load_group(q, Ab, Bb);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (more && !same_phase) {
                // phase change: every wave is done with the old patch; the next one is LDS-DMA'd in place (exposed: the
                // co-resident workgroups cover the wait)
                __syncthreads();
                dma_patch(ph + 1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                asm volatile("; LDS reads stay below the barrier" : "+v"(a_row[0])::"memory");
                if (active) {
                    load_group(0, As, Bb);
                    load_group(1, As, Bb);
                }
                ++ph;
            }
            tap = tap1;
            slot = slot1;
        }
    }

    if (CHAIN) {
        // ---- chained tail: the [rows][64] result of this conv (bias + ReLU applied) is PARKED in LDS 32 rows at a time and fed
        // to the 1x1x1 tail as its A operand; the tail's filter fragments come straight from global memory (64 KiB, L2-resident,
        // the same for every workgroup); one 64-column chunk of the tail's output at a time: bias2 (+ residual) + ReLU, stored.
        const bool relu1 = (p.flags1 & PTX_EPI_RELU) != 0;
        const bool relu2 = (p.flags & PTX_EPI_RELU) != 0;
        const bool has_res = (p.flags & PTX_EPI_RES_ADD) != 0;
        const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res ? p.res : p.y), 0, p.r_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w2), 0, p.w2_bytes, 0x00020000);
        const int m_frame = (n * p.T + to) * frame;
        const int n_chunks2 = (p.ncol + 63) >> 6;
        float* Pk = smem + wm * kB3Park;               // the parked tile of this wave ROW (shared by its WN waves)
        __syncthreads();                                // every wave is done with the patch and the ring: the region is free
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            if (i > 0) {
                if (WN > 1) __syncthreads();
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int j = 0; j < CT; ++j) {
                const int c = (wn * CT + j) * 32 + l32;
                const float bv = (p.bias && c < p.ncol1) ? p.bias[c] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r] + bv;
                    v = relu1 ? fmaxf(v, 0.f) : v;
                    Pk[(c >> 2) * kB3ParkStride + ((r & 3) + 8 * (r >> 2) + 4 * g) * 4 + (c & 3)] = v;
                }
            }
            if (WN > 1) __syncthreads();
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const bool tile_ok = (m0 + (wm * RT + i) * 32) < frame;
            if (tile_ok) {
                for (int nc = wn; nc < n_chunks2; nc += WN) {
                    // the residual rows of this chunk first: their latency hides behind the 64 MFMAs below
                    float rv[2][16];
                    if (has_res) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int co = nc * 64 + j * 32 + l32;
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int ml = m0 + (wm * RT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                                const unsigned off = ((unsigned)(m_frame + ml) * (unsigned)p.ldr + (unsigned)co) * 4u;
                                rv[j][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, (co < p.ncol && ml < frame) ? off : kOOB, 0, 0));
                            }
                        }
                    }
                    f32x16 acc2[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;
                    const unsigned wbase = (unsigned)(((nc * 8 * 2 + g) * 64 + l32) * 16);
                    // filter fragments two octets ahead of the MFMAs that use them (L2 latency ~ one octet's 512 MFMA cycles)
                    f32x4 fb0[2], fb1[2], fb2_[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        fb0[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w2, wbase + j * 512, 0, 0));
                        fb1[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w2, wbase + 2048 + j * 512, 0, 0));
                    }
#pragma unroll 1
                    for (int q = 0; q < 8; ++q) {           // (rolled: unrolled, hipcc hoists all sixteen filter loads -- 64 registers)
                        const f32x4 fa2 = *reinterpret_cast<const f32x4*>(Pk + (2 * q + g) * kB3ParkStride + l32 * 4);
                        const unsigned nxt = q + 2 < 8 ? wbase + (q + 2) * 2048 : kOOB;
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            fb2_[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w2, nxt == kOOB ? kOOB : nxt + j * 512, 0, 0));
#pragma unroll
                        for (int k = 0; k < 4; ++k)
#pragma unroll
                            for (int j = 0; j < 2; ++j)
                                acc2[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa2[k], fb0[j][k], acc2[j], 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            fb0[j] = fb1[j];
                            fb1[j] = fb2_[j];
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int co = nc * 64 + j * 32 + l32;
                        const bool co_ok = co < p.ncol;
                        const float bv = (p.bias2 && co_ok) ? p.bias2[co] : 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int ml = m0 + (wm * RT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                            float v = acc2[j][r] + bv;
                            if (has_res) v += rv[j][r];
                            v = relu2 ? fmaxf(v, 0.f) : v;
                            const unsigned off = ((unsigned)(m_frame + ml) * (unsigned)p.ldy + (unsigned)co) * 4u;
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_y, (co_ok && ml < frame) ? off : kOOB, 0, 0);
                        }
                    }
                }
            }
        }
        return;
    }
    // ---- epilogue: bias (+ folded BN) (+ residual) + ReLU; lane = output channel, 16 rows per accumulator tile ----
    if (!active) return;
    const bool relu = (p.flags & PTX_EPI_RELU) != 0;
    const bool has_res = (p.flags & PTX_EPI_RES_ADD) != 0;
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res ? p.res : p.y), 0, p.r_bytes, 0x00020000);
    const int m_frame = (n * p.T + to) * frame;
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        const int co = nt * kB3BN + (wn * CT + j) * 32 + l32;
        const bool co_ok = co < p.ncol;
        const float bv = (p.bias && co_ok) ? p.bias[co] : 0.f;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            float rv[16];
            if (has_res) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = m0 + (wm * RT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                    const unsigned off = ((unsigned)(m_frame + ml) * (unsigned)p.ldr + (unsigned)co) * 4u;
                    rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, (co_ok && ml < frame) ? off : kOOB, 0, 0));
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // accumulator element r of this lane belongs to tile row (r & 3) + 8 * (r >> 2) + 4 * g
                const int ml = m0 + (wm * RT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                float v = acc[i][j][r] + bv;
                if (has_res) v += rv[r];
                v = relu ? fmaxf(v, 0.f) : v;
                const unsigned off = ((unsigned)(m_frame + ml) * (unsigned)p.ldy + (unsigned)co) * 4u;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_y, (co_ok && ml < frame) ? off : kOOB, 0, 0);
            }
        }
    }
    (void)BM;
}

// Tile list of a launch: blockIdx.x < n_tall -> tall tiles (frames in XCD-contiguous chunks), then the square tiles.
// blockIdx.y = 64-channel column tile.  (Two plain kernels rather than one template: hipcc's host pass dropped the stub of
// the second instantiation.)
template <bool CHAIN>
__device__ __forceinline__ void conv_body_square(const BodyF32Args& p, float* smem, int b, int nt) {
    const int tile = xcd_remap(b, p.n_sq);
    const int f = tile / p.sq_per_frame, k = tile - f * p.sq_per_frame;
    const int n = f / p.T, to = f - n * p.T;
    conv_body_tile<2, 2, 1, 1, 8, CHAIN>(p, smem, n, to, p.tall_per_frame * 256 + k * 64, nt, p.PR_sq, p.dv_npos_sq);
}

template <bool CHAIN>
__device__ __forceinline__ void conv_body_mixed(const BodyF32Args& p, float* smem) {
    const int b = blockIdx.x;
    if (b < p.n_tall) {
        const int tile = xcd_remap(b, p.n_tall);
        const int f = tile / p.tall_per_frame, k = tile - f * p.tall_per_frame;        // frame (n, to), tile inside it
        const int n = f / p.T, to = f - n * p.T;
        conv_body_tile<4, 1, 2, 2, 8, CHAIN>(p, smem, n, to, k * 256, blockIdx.y, p.PR_tall, p.dv_npos_tall);
    } else {
        conv_body_square<CHAIN>(p, smem, b - p.n_tall, blockIdx.y);
    }
}

__global__ void __launch_bounds__(kB3NT, 2) conv_body_f32_kernel(const BodyF32Args p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    conv_body_mixed<false>(p, smem);
}

__global__ void __launch_bounds__(kB3NT, 2) conv_body_f32_sq_kernel(const BodyF32Args p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    conv_body_square<false>(p, smem, blockIdx.x, blockIdx.y);
}

// the same two launches with the chained 1x1x1 tail (one column tile: the tail needs the whole intermediate row)
__global__ void __launch_bounds__(kB3NT, 3) conv_body_chain_f32_kernel(const BodyF32Args p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    conv_body_mixed<true>(p, smem);
}

__global__ void __launch_bounds__(kB3NT, 3) conv_body_chain_f32_sq_kernel(const BodyF32Args p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    conv_body_square<true>(p, smem, blockIdx.x, 0);
}

// ---- (kT,1,1) temporal convolutions on the same skeleton: the T-STACKED tile (VERDICT r5 #2b) ----
// Conv3d(C, Co, (kT,1,1), stride 1, pad (kT/2,0,0)) + BN (+ residual) + ReLU: the temporal half of a SpatioTemporalConv
// (r2plus1d.py:84-88) -- config 3's (7,1,1) stem conv (110 -> 64 channels over 8 x 32 x 56 x 56 positions, 20 % of the net's
// FLOPs) and the (3,1,1) halves of its stride-1 body convs.  On the generic tiles the 128 rows of a workgroup lie in ONE frame,
// so each of the kT taps stages a different input frame and the tensor goes through L2 kT times (FETCH + WRITE = 1.43x the
// algorithmic bytes, profiles/r05_pmc_traffic_cfg3.json).  Here a workgroup owns the SAME 32 positions of 8 CONSECUTIVE output
// frames (256 rows x 64 output channels); the patch of a phase (16 channels) holds those positions of the 8 + kT - 1 input
// frames the tile touches and serves all kT taps: tap kt reads the patch kt frames (= 32 positions) further down.  LDS fill
// per MFMA drops to (8 + kT - 1) / (8 kT) of the generic tile's, tiles adjacent in T follow each other in an XCD's chunk (their
// 6-frame overlap hits L2).  Wave w owns output frames 2w and 2w + 1 (one 32-row MFMA tile each): a (frame, tap) whose input
// frame lies outside the clip is SKIPPED per row tile (wave-uniform), so the zero padding costs no MFMAs.
// Measured (profiles/r06_tstack_ab.txt): 116 TF on config 3's stem conv against 124.6 TF for the best generic tile -- the conv
// is MFMA-bound, not fill-bound, and 256-row x 49-step workgroups quantise worse (12.25 tiles per CU).  The tuner keeps the
// generic tiles there; the kernel stays as an execution it may pick per problem ("body:" keys).
// Filter slices: [n tile][chunk][kt][2][2][64][4] (the body pack with taps = kT); K order as in the body kernel.
constexpr int kTsF = 8;                    // output frames per tile
constexpr int kTsS = 32;                   // positions per frame slab

template <int NP>
__device__ __forceinline__ void conv_tstack_tile(const BodyF32Args& p, float* smem, const int n, const int t0, const int s0, const int nt) {
    constexpr unsigned kOOB = 0x80000000u;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    constexpr int RT = 2, CT = 2;
    const int NPOS = (kTsF + p.kT - 1) * kTsS;          // positions of the patch
    float* As = smem;                                   // [4][NPOS][4]
    float* Bs = smem + 16 * NPOS;                       // [3][kB3Slot]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, l32 = lane & 31;

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);

    // ---- per-thread sources of the patch pieces (phase independent): piece q = tid + 256 i = group * NPOS + frame * 32 + position
    unsigned a_src[NP];
    const int n_pieces = 4 * NPOS;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int q = tid + kB3NT * i;
        const int grp = (int)b3_fdiv((unsigned)q, p.dv_npos_tall);
        const int pos = q - grp * NPOS;
        const int t = t0 - p.pT + (pos >> 5), s = s0 + (pos & 31);
        const bool ok = grp < 4 && (unsigned)t < (unsigned)p.T && s < p.HW;
        a_src[i] = ok ? (unsigned)((((n * p.T + t) * p.HW + s) * p.ldx + grp * 4) * 4) : kOOB;
    }
    const unsigned b_src = (unsigned)(tid * 16);
    const int n_steps = p.chunks * p.kT;

    auto dma_patch = [&](int ph) {
        const unsigned cbase = (unsigned)(ph * kB3CK * 4);
#pragma unroll
        for (int i = 0; i < NP; ++i)
            if (tid + kB3NT * i < n_pieces)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(As + (wave * 64 + kB3NT * i) * 4), 16,
                                                         a_src[i] == kOOB ? kOOB : a_src[i] + cbase, 0, 0, 0);
    };
    // filter slice of step s = chunk * kT + kt: [nt][chunk][kt] blocks of kB3Slot floats
    auto dma_b = [&](int slot, int s) {
        const unsigned tbase = (unsigned)((nt * n_steps + s) * (kB3Slot * 4));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(Bs + slot * kB3Slot + wave * 256), 16, b_src + tbase, 0, 0, 0);
    };

    // ---- this wave's rows: row tile i = output frame t0 + 2 wave + i, positions s0 + l32; its valid taps [lo, hi] ----
    int a_row[RT], kt_lo[RT], kt_hi[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int f = wave * RT + i, t = t0 + f;
        a_row[i] = (g * NPOS + f * kTsS + l32) * 4;
        kt_lo[i] = t < p.T ? max(0, p.pT - t) : 1;
        kt_hi[i] = t < p.T ? min(p.kT - 1, p.T - 1 - t + p.pT) : 0;
    }

    f32x16 acc[RT][CT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 fa[2][RT], fb[2][CT];
    const int b_lane = (g * 64 + l32) * 4;              // + q * 512 + j * 128 floats
    auto load_group = [&](int q, const float* Ab, const float* Bb) {
#pragma unroll
        for (int i = 0; i < RT; ++i) fa[q][i] = *reinterpret_cast<const f32x4*>(Ab + a_row[i] + q * 8 * NPOS);
#pragma unroll
        for (int j = 0; j < CT; ++j) fb[q][j] = *reinterpret_cast<const f32x4*>(Bb + b_lane + q * 512 + j * 128);
    };
    // (one wave-uniform branch per row tile: the two column tiles of a row tile alternate, so no MFMA waits on its own result)
    auto mma_group = [&](int q, bool d0, bool d1) {
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            if (i == 0 ? d0 : d1) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int j = 0; j < CT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][i][k], fb[q][j][k], acc[i][j], 0, 0, 0);
            }
        }
    };

    dma_patch(0);
    dma_b(0, 0);
    dma_b(1, 1);                                        // (n_steps >= kT >= 3)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    asm volatile("; LDS reads stay below the barrier" : "+v"(a_row[0])::"memory");
    load_group(0, As, Bs);
    load_group(1, As, Bs);
    int tap = 0, slot = 0, ph = 0;
    for (int s = 0; s < n_steps; ++s) {
        if (s > 0) {
            // filter slice s + 1 (requested during step s - 1) has landed for everyone; slot (s + 2) % 3 is free
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        const int slot1 = slot == 2 ? 0 : slot + 1;
        const int slot2 = slot1 == 2 ? 0 : slot1 + 1;
        if (s + 2 < n_steps) dma_b(slot2, s + 2);
        const bool more = s + 1 < n_steps;
        const int tap1 = tap == p.kT - 1 ? 0 : tap + 1;
        const bool same_phase = tap1 != 0;
        const float* Ab = As + tap1 * (kTsS * 4);
        const float* Bb = Bs + slot1 * kB3Slot;
        const bool prefetch = more && same_phase;
        const bool d0 = tap >= kt_lo[0] && tap <= kt_hi[0], d1 = tap >= kt_lo[1] && tap <= kt_hi[1];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            mma_group(q, d0, d1);
            __builtin_amdgcn_sched_barrier(0);
            if (prefetch) load_group(q, Ab, Bb);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more && !same_phase) {
            // phase change: every wave is done with the old patch; the next chunk's is LDS-DMA'd in place (exposed: the
            // co-resident workgroups cover most of the wait -- profiles/r06_tstack_ab.txt)
            __syncthreads();
            dma_patch(ph + 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            asm volatile("; LDS reads stay below the barrier" : "+v"(a_row[0])::"memory");
            load_group(0, As, Bb);
            load_group(1, As, Bb);
            ++ph;
        }
        tap = tap1;
        slot = slot1;
    }

    // ---- epilogue: bias (+ folded BN) (+ residual) + ReLU; lane = output channel, 16 rows per accumulator tile ----
    const bool relu = (p.flags & PTX_EPI_RELU) != 0;
    const bool has_res = (p.flags & PTX_EPI_RES_ADD) != 0;
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res ? p.res : p.y), 0, p.r_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int t = t0 + wave * RT + i;
        if (t >= p.T) continue;                         // (wave-uniform)
        const int m_row0 = (n * p.T + t) * p.HW + s0;   // output row of position s0 of this frame
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            const int co = nt * kB3BN + j * 32 + l32;
            const bool co_ok = co < p.ncol;
            const float bv = (p.bias && co_ok) ? p.bias[co] : 0.f;
            float rv[16];
            if (has_res) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int sl = (r & 3) + 8 * (r >> 2) + 4 * g;
                    const unsigned off = ((unsigned)(m_row0 + sl) * (unsigned)p.ldr + (unsigned)co) * 4u;
                    rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, (co_ok && s0 + sl < p.HW) ? off : kOOB, 0, 0));
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int sl = (r & 3) + 8 * (r >> 2) + 4 * g;
                float v = acc[i][j][r] + bv;
                if (has_res) v += rv[r];
                v = relu ? fmaxf(v, 0.f) : v;
                const unsigned off = ((unsigned)(m_row0 + sl) * (unsigned)p.ldy + (unsigned)co) * 4u;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_y, (co_ok && s0 + sl < p.HW) ? off : kOOB, 0, 0);
            }
        }
    }
}

// tile list of a T-stacked launch: frames fastest (the tiles of one position slab follow each other), an XCD owns a contiguous
// chunk; blockIdx.y = 64-channel column tile
__global__ void __launch_bounds__(kB3NT, 2) conv_tstack_f32_kernel(const BodyF32Args p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tile = xcd_remap(blockIdx.x, p.n_tall);
    const int tt = tile % p.t_tiles, r = tile / p.t_tiles;
    const int st = r % p.s_tiles, n = r / p.s_tiles;
    conv_tstack_tile<8>(p, smem, n, tt * kTsF, st * kTsS, blockIdx.y);
}

// generic packed filter [tap][Co_pad][Kc] (ptx_pack_conv_weight: BN scale folded in) -> [nt][kt][chunk][kh * 3 + kw][q][g][64][4]
// (taps = 9; taps = 1 / kT = 1 / chunks = 4 packs a pointwise filter as the chained tail's [N2 / 64][8 octets][2][64][4] image)
__global__ void __launch_bounds__(256) pack_body_f32_kernel(const float* __restrict__ wp, float* __restrict__ out, int kT, int Co_pad,
                                                            int Kc, int chunks, int taps, int total) {
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int i = idx & 3;
        int r = idx >> 2;
        const int col = r & 63;
        r >>= 6;
        const int g = r & 1;
        r >>= 1;
        const int q = r & 1;
        r >>= 1;
        const int tap = r % taps;
        r /= taps;
        const int ch = r % chunks;
        r /= chunks;
        const int kt = r % kT;
        const int nt = r / kT;
        const int c = ch * kB3CK + 8 * q + 4 * g + i;
        const int co = nt * kB3BN + col;
        out[idx] = (co < Co_pad && c < Kc) ? wp[((size_t)(kt * taps + tap) * Co_pad + co) * Kc + c] : 0.f;
    }
}

struct BodyGeom {
    int tall_per_frame, sq_per_frame, PR_tall, PR_sq;
};

// rows of the input a raster span of `bm` outputs starting anywhere in a frame of width W can touch, + the 3-row halo
static int body_patch_rows(int H, int W, int bm) { return std::min(H, (bm - 1 + W - 1) / W + 1) + 2; }

static bool body_geom(const ptx_conv3d_desc* d, int shape, BodyGeom* g) {
    const int frame = d->Ho * d->Wo;
    g->tall_per_frame = shape == 0 ? frame / 256 : 0;
    g->sq_per_frame = cdiv(frame - g->tall_per_frame * 256, 64);
    g->PR_tall = body_patch_rows(d->Hi, d->Wi, 256);
    g->PR_sq = body_patch_rows(d->Hi, d->Wi, 64);
    const int PC = d->Wi + 2;
    if (g->tall_per_frame > 0 && g->PR_tall * PC > kB3PosMax) return false;
    return g->PR_sq * PC <= kB3PosMax;
}

}  // namespace ptx

using namespace ptx;

// shape 0: tall tiles + square tails; shape 1: square tiles only
// a (kT,1,1) temporal conv the T-stacked tile serves (shape 0 only): kT in {3, 5, 7}, stride 1, pad (kT/2,0,0)
static bool tstack_problem(const ptx_conv3d_desc* d) { return d->kH == 1 && d->kW == 1 && d->kT > 1; }

static int tstack_supported(const ptx_conv3d_desc* d, int shape) {
    if (shape != 0) return 0;
    if (d->groups > 1 || (d->kT != 3 && d->kT != 5 && d->kT != 7)) return 0;
    if (d->sT != 1 || d->sH != 1 || d->sW != 1 || d->pH != 0 || d->pW != 0 || d->pT != d->kT / 2) return 0;
    if (d->To != d->Ti || d->Ho != d->Hi || d->Wo != d->Wi || d->N < 1 || d->Ti < 1 || d->Hi < 1 || d->Wi < 1) return 0;
    // channels walk in 16-wide chunks: the columns [Ci, 16 * chunks) must exist in x (zero or finite: their filter rows are zero)
    const int kc16 = cdiv(d->Ci, kB3CK) * kB3CK;
    if (d->Ci < 1 || d->Kc < kc16 || d->ldx < kc16 || d->ldx % 4 || d->Co < 1 || d->Co_pad % kB3BN) return 0;
    if (d->ldy < (d->Co + 3) / 4 * 4 || d->ldy % 4) return 0;
    if ((d->flags & PTX_EPI_RES_ADD) && (d->ldr < d->Co || d->ldr % 4)) return 0;
    const int64_t M = (int64_t)d->N * d->Ti * d->Hi * d->Wi;
    if (M * d->ldx * 4 >= 0x80000000LL || M * d->ldy * 4 >= 0x80000000LL || M * std::max(d->ldr, 1) * 4 >= 0x80000000LL) return 0;
    if ((int64_t)d->N * cdiv(d->Ti, kTsF) * cdiv(d->Hi * d->Wi, kTsS) > 0x7fffffffLL) return 0;
    if ((int64_t)(d->Co_pad / kB3BN) * kc16 / kB3CK * d->kT * kB3Slot * 4 >= 0x80000000LL) return 0;
    return 1;
}

extern "C" int ptx_conv_body_f32_supported(const ptx_conv3d_desc* d, int shape) {
    if (!d || shape < 0 || shape > 1) return 0;
    if (d->flags & ~(PTX_EPI_RELU | PTX_EPI_RES_ADD)) return 0;
    if (tstack_problem(d)) return tstack_supported(d, shape);
    if (d->groups > 1 || d->kH != 3 || d->kW != 3 || (d->kT != 3 && d->kT != 1)) return 0;
    if (d->sT != 1 || d->sH != 1 || d->sW != 1 || d->pH != 1 || d->pW != 1 || d->pT != d->kT / 2) return 0;
    if (d->To != d->Ti || d->Ho != d->Hi || d->Wo != d->Wi || d->N < 1 || d->Ti < 1 || d->Hi < 1 || d->Wi < 1) return 0;
    if (d->Ci < kB3CK || d->Ci % kB3CK || d->Kc < d->Ci || d->ldx < d->Ci || d->ldx % 4 || d->Co < 1 || d->Co_pad % kB3BN) return 0;
    if (d->ldy < (d->Co + 3) / 4 * 4 || d->ldy % 4) return 0;
    if ((d->flags & PTX_EPI_RES_ADD) && (d->ldr < d->Co || d->ldr % 4)) return 0;
    BodyGeom g;
    if (!body_geom(d, shape, &g)) return 0;
    if (shape == 0 && g.tall_per_frame == 0) return 0;
    const int64_t M = (int64_t)d->N * d->Ti * d->Hi * d->Wi;
    if (M * d->ldx * 4 >= 0x80000000LL || M * d->ldy * 4 >= 0x80000000LL || M * std::max(d->ldr, 1) * 4 >= 0x80000000LL) return 0;
    if ((int64_t)d->N * d->Ti * (g.tall_per_frame + g.sq_per_frame) > 0x7fffffffLL) return 0;
    return 1;
}

extern "C" size_t ptx_conv_body_f32_weight_elems(const ptx_conv3d_desc* d) {
    if (d && tstack_problem(d)) {
        if (d->Co_pad <= 0 || d->Co_pad % kB3BN || d->Ci <= 0) return 0;
        return (size_t)(d->Co_pad / kB3BN) * cdiv(d->Ci, kB3CK) * d->kT * kB3Slot;
    }
    if (!d || d->Co_pad <= 0 || d->Co_pad % kB3BN || d->Ci <= 0 || d->Ci % kB3CK || (d->kT != 1 && d->kT != 3)) return 0;
    return (size_t)(d->Co_pad / kB3BN) * d->kT * (d->Ci / kB3CK) * 9 * kB3Slot;
}

extern "C" int ptx_pack_conv_body_f32_weight(const ptx_conv3d_desc* d, const float* w_packed, float* w_body, ptx_stream_t stream) {
    if (!d || !w_packed || !w_body) return fail(PTX_ERR_INVALID, "pack_conv_body_f32: null pointer");
    const size_t total = ptx_conv_body_f32_weight_elems(d);
    if (total && tstack_problem(d)) {
        // [n tile][chunk][kt][2][2][64][4]: the same image with the kT temporal taps as the "taps" of a phase
        if (d->Kc < d->Ci || total >= (1ull << 31)) return fail(PTX_ERR_UNSUPPORTED, "pack_conv_body_f32: bad (kT,1,1) filter extents");
        hipLaunchKernelGGL(pack_body_f32_kernel, dim3((unsigned)std::min<size_t>(cdiv64((int64_t)total, 256), 4096)), dim3(256), 0, (hipStream_t)stream,
                           w_packed, w_body, 1, d->Co_pad, d->Kc, cdiv(d->Ci, kB3CK), d->kT, (int)total);
        return hip_check(hipGetLastError(), "pack_conv_body_f32 launch");
    }
    if (!total || d->kH != 3 || d->kW != 3 || d->Kc < d->Ci)
        return fail(PTX_ERR_UNSUPPORTED, "pack_conv_body_f32: a (1|3)x3x3 or (kT,1,1) filter, Ci a multiple of 16, Co_pad a multiple of 64");
    if (total >= (1ull << 31)) return fail(PTX_ERR_UNSUPPORTED, "pack_conv_body_f32: filter too large");
    hipLaunchKernelGGL(pack_body_f32_kernel, dim3((unsigned)std::min<size_t>(cdiv64((int64_t)total, 256), 4096)), dim3(256), 0, (hipStream_t)stream,
                       w_packed, w_body, d->kT, d->Co_pad, d->Kc, d->Ci / kB3CK, 9, (int)total);
    return hip_check(hipGetLastError(), "pack_conv_body_f32 launch");
}

static int launch_body(bool tall, bool chain, const BodyF32Args& a, dim3 grid, size_t lds, ptx_stream_t stream) {
    static bool attr_set[64] = {};
    int dev = 0;
    PTX_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        const int max_lds = (int)((16 * kB3PosMax + 3 * kB3Slot) * sizeof(float));
        PTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_body_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
        PTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_body_f32_sq_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
        PTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_body_chain_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
        PTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_body_chain_f32_sq_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    if (tall && chain) hipLaunchKernelGGL(conv_body_chain_f32_kernel, grid, dim3(kB3NT), lds, (hipStream_t)stream, a);
    else if (tall) hipLaunchKernelGGL(conv_body_f32_kernel, grid, dim3(kB3NT), lds, (hipStream_t)stream, a);
    else if (chain) hipLaunchKernelGGL(conv_body_chain_f32_sq_kernel, grid, dim3(kB3NT), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(conv_body_f32_sq_kernel, grid, dim3(kB3NT), lds, (hipStream_t)stream, a);
    return PTX_OK;
}

// geometry / pointer-independent part of the launch arguments of `d` on `shape`; returns the dynamic LDS bytes
static size_t body_make_args(const ptx_conv3d_desc* d, int shape, BodyF32Args& a, BodyGeom& g) {
    body_geom(d, shape, &g);
    a.N = d->N; a.T = d->Ti; a.H = d->Hi; a.W = d->Wi; a.C = d->Ci; a.ldx = d->ldx;
    a.kT = d->kT; a.pT = d->pT;
    a.chunks = d->Ci / kB3CK;
    a.PC = d->Wi + 2;
    a.tall_per_frame = g.tall_per_frame; a.sq_per_frame = g.sq_per_frame;
    a.PR_tall = g.PR_tall; a.PR_sq = g.PR_sq;
    const int frames = d->N * d->Ti;
    a.n_tall = frames * g.tall_per_frame;
    a.n_sq = frames * g.sq_per_frame;
    const uint64_t M = (uint64_t)d->N * d->Ti * d->Hi * d->Wi;
    a.x_bytes = (unsigned)(M * d->ldx * 4ull);
    a.w_bytes = (unsigned)(ptx_conv_body_f32_weight_elems(d) * 4ull);
    b3_fdiv_make((unsigned)d->Wi, a.dv_w);
    b3_fdiv_make((unsigned)a.PC, a.dv_pc);
    b3_fdiv_make((unsigned)(a.PR_tall * a.PC), a.dv_npos_tall);
    b3_fdiv_make((unsigned)(a.PR_sq * a.PC), a.dv_npos_sq);
    const int npos = std::max(g.tall_per_frame ? a.PR_tall * a.PC : 0, a.PR_sq * a.PC);
    return (size_t)(16 * npos + 3 * kB3Slot) * sizeof(float);
}

// the T-stacked launch of a (kT,1,1) conv (tstack_supported)
static int launch_tstack(const ptx_conv3d_desc* d, const float* x, const float* w_body, const float* bias, const float* res, float* y,
                         ptx_stream_t stream) {
    BodyF32Args a{};
    a.x = x; a.w = w_body; a.bias = bias; a.res = (d->flags & PTX_EPI_RES_ADD) ? res : nullptr; a.y = y;
    a.N = d->N; a.T = d->Ti; a.H = d->Hi; a.W = d->Wi; a.C = d->Ci; a.ldx = d->ldx;
    a.ldy = d->ldy; a.ldr = d->ldr > 0 ? d->ldr : d->ldy;
    a.ncol = (d->Co + 3) / 4 * 4;
    a.kT = d->kT; a.pT = d->pT;
    a.chunks = cdiv(d->Ci, kB3CK);
    a.HW = d->Hi * d->Wi;
    a.t_tiles = cdiv(d->Ti, kTsF);
    a.s_tiles = cdiv(a.HW, kTsS);
    a.n_tall = d->N * a.t_tiles * a.s_tiles;
    a.flags = d->flags;
    const uint64_t M = (uint64_t)d->N * d->Ti * a.HW;
    a.x_bytes = (unsigned)(M * d->ldx * 4ull);
    a.w_bytes = (unsigned)(ptx_conv_body_f32_weight_elems(d) * 4ull);
    a.y_bytes = (unsigned)(M * d->ldy * 4ull);
    a.r_bytes = (unsigned)(M * a.ldr * 4ull);
    const int npos = (kTsF + d->kT - 1) * kTsS;
    b3_fdiv_make((unsigned)npos, a.dv_npos_tall);
    const size_t lds = (size_t)(16 * npos + 3 * kB3Slot) * sizeof(float);     // 40 KiB at kT = 7: three / four workgroups per CU
    static bool attr_set[64] = {};
    int dev = 0;
    PTX_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        PTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_tstack_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)((16 * (kTsF + 6) * kTsS + 3 * kB3Slot) * sizeof(float))));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(conv_tstack_f32_kernel, dim3((unsigned)a.n_tall, (unsigned)cdiv(a.ncol, kB3BN)), dim3(kB3NT), lds, (hipStream_t)stream, a);
    return hip_check(hipGetLastError(), "conv_tstack_f32 launch");
}

extern "C" int ptx_conv_body_f32_fwd(const ptx_conv3d_desc* d, const float* x, const float* w_body, const float* bias, const float* res,
                                     float* y, int shape, ptx_stream_t stream) {
    if (!d || !x || !w_body || !y) return fail(PTX_ERR_INVALID, "conv_body_f32: null pointer");
    if (((uintptr_t)x | (uintptr_t)w_body | (uintptr_t)y | (uintptr_t)res) & 15) return fail(PTX_ERR_INVALID, "conv_body_f32: pointers must be 16-byte aligned");
    if (!ptx_conv_body_f32_supported(d, shape))
        return fail(PTX_ERR_UNSUPPORTED, "conv_body_f32: needs a dense (1|3)x3x3 / stride 1 / pad (kT/2,1,1) conv, Ci a multiple of 16, Co_pad a "
                    "multiple of 64, bias / ReLU / same-shape residual epilogue, and an input patch of at most %d positions (shape %d) -- or a "
                    "(3|5|7)x1x1 / stride 1 / pad (kT/2,0,0) conv on shape 0", kB3PosMax, shape);
    if ((d->flags & PTX_EPI_RES_ADD) && !res) return fail(PTX_ERR_INVALID, "conv_body_f32: PTX_EPI_RES_ADD without a residual");
    if (tstack_problem(d)) return launch_tstack(d, x, w_body, bias, res, y, stream);
    BodyGeom g;
    BodyF32Args a{};
    const size_t lds = body_make_args(d, shape, a, g);
    a.x = x; a.w = w_body; a.bias = bias; a.res = (d->flags & PTX_EPI_RES_ADD) ? res : nullptr; a.y = y;
    a.ldy = d->ldy; a.ldr = d->ldr > 0 ? d->ldr : d->ldy;
    a.ncol = (d->Co + 3) / 4 * 4;
    a.flags = d->flags;
    const uint64_t M = (uint64_t)d->N * d->Ti * d->Hi * d->Wi;
    a.y_bytes = (unsigned)(M * d->ldy * 4ull);
    a.r_bytes = (unsigned)(M * a.ldr * 4ull);
    const dim3 grid((unsigned)(a.n_tall + a.n_sq), (unsigned)cdiv(a.ncol, kB3BN));
    int rc = launch_body(g.tall_per_frame > 0, false, a, grid, lds, stream);
    if (rc != PTX_OK) return rc;
    return hip_check(hipGetLastError(), "conv_body_f32 launch");
}

// ---- chained: conv (body kernel) -> bias -> ReLU -> 1x1x1 tail -> bias2 (-> + residual) -> ReLU in ONE launch ----
extern "C" int ptx_conv_body_chain_f32_supported(const ptx_conv3d_desc* d, const ptx_conv3d_desc* t, int shape) {
    if (!d || !t) return 0;
    if ((d->flags & ~PTX_EPI_RELU) || (t->flags & ~(PTX_EPI_RELU | PTX_EPI_RES_ADD))) return 0;
    if (tstack_problem(d)) return 0;                    // (the T-stacked tile has no chained form)
    ptx_conv3d_desc dd = *d;
    dd.ldy = (d->Co + 3) / 4 * 4;                       // the first conv's output never reaches memory
    if (!ptx_conv_body_f32_supported(&dd, shape)) return 0;
    if (d->Co > kB3BN) return 0;                        // the tail needs the whole intermediate row in one workgroup
    if (t->kT != 1 || t->kH != 1 || t->kW != 1 || t->sT != 1 || t->sH != 1 || t->sW != 1 || t->pT || t->pH || t->pW || t->groups > 1) return 0;
    if (t->N != d->N || t->Ti != d->To || t->Hi != d->Ho || t->Wi != d->Wo || t->To != d->To || t->Ho != d->Ho || t->Wo != d->Wo) return 0;
    if (t->Ci != d->Co || t->Kc < t->Ci || t->Co < 1 || t->Co_pad % kB3BN || t->Co_pad < t->Co) return 0;
    if (t->ldy < (t->Co + 3) / 4 * 4 || t->ldy % 4) return 0;
    if ((t->flags & PTX_EPI_RES_ADD) && (t->ldr < t->Co || t->ldr % 4)) return 0;
    const int64_t M = (int64_t)d->N * d->Ti * d->Hi * d->Wi;
    if (M * t->ldy * 4 >= 0x80000000LL || M * std::max(t->ldr, 1) * 4 >= 0x80000000LL) return 0;
    return 1;
}

extern "C" size_t ptx_conv_body_tail_f32_weight_elems(const ptx_conv3d_desc* t) {
    if (!t || t->Co_pad <= 0 || t->Co_pad % kB3BN) return 0;
    return (size_t)(t->Co_pad / kB3BN) * 8 * 2 * 64 * 4;
}

extern "C" int ptx_pack_conv_body_tail_f32_weight(const ptx_conv3d_desc* t, const float* w_packed, float* w_tail, ptx_stream_t stream) {
    if (!t || !w_packed || !w_tail) return fail(PTX_ERR_INVALID, "pack_conv_body_tail_f32: null pointer");
    const size_t total = ptx_conv_body_tail_f32_weight_elems(t);
    if (!total || t->kT != 1 || t->kH != 1 || t->kW != 1 || t->Ci > 64 || t->Kc < t->Ci)
        return fail(PTX_ERR_UNSUPPORTED, "pack_conv_body_tail_f32: a 1x1x1 filter over <= 64 channels, Co_pad a multiple of 64");
    hipLaunchKernelGGL(pack_body_f32_kernel, dim3((unsigned)std::min<size_t>(cdiv64((int64_t)total, 256), 4096)), dim3(256), 0, (hipStream_t)stream,
                       w_packed, w_tail, 1, t->Co_pad, t->Kc, 4, 1, (int)total);
    return hip_check(hipGetLastError(), "pack_conv_body_tail_f32 launch");
}

extern "C" int ptx_conv_body_chain_f32_fwd(const ptx_conv3d_desc* d, const ptx_conv3d_desc* t, const float* x, const float* w_body,
                                           const float* bias, const float* w_tail, const float* bias2, const float* res, float* y, int shape,
                                           ptx_stream_t stream) {
    if (!d || !t || !x || !w_body || !w_tail || !y) return fail(PTX_ERR_INVALID, "conv_body_chain_f32: null pointer");
    if (((uintptr_t)x | (uintptr_t)w_body | (uintptr_t)w_tail | (uintptr_t)y | (uintptr_t)res) & 15)
        return fail(PTX_ERR_INVALID, "conv_body_chain_f32: pointers must be 16-byte aligned");
    if (!ptx_conv_body_chain_f32_supported(d, t, shape))
        return fail(PTX_ERR_UNSUPPORTED, "conv_body_chain_f32: a body conv (ptx_conv_body_f32_supported) of at most 64 output channels with a ReLU-only "
                    "epilogue, followed by a dense 1x1x1 / unit-stride conv over its output positions (bias / ReLU / same-shape residual)");
    if ((t->flags & PTX_EPI_RES_ADD) && !res) return fail(PTX_ERR_INVALID, "conv_body_chain_f32: PTX_EPI_RES_ADD without a residual");
    ptx_conv3d_desc dd = *d;
    dd.ldy = (d->Co + 3) / 4 * 4;
    BodyGeom g;
    BodyF32Args a{};
    size_t lds = body_make_args(&dd, shape, a, g);
    lds = std::max(lds, (size_t)4 * kB3Park * sizeof(float));          // four parked tiles (tall) / two (square)
    a.x = x; a.w = w_body; a.bias = bias; a.res = (t->flags & PTX_EPI_RES_ADD) ? res : nullptr; a.y = y;
    a.w2 = w_tail; a.bias2 = bias2;
    a.ncol1 = (d->Co + 3) / 4 * 4;
    a.flags1 = d->flags;
    a.flags = t->flags;
    a.ldy = t->ldy; a.ldr = t->ldr > 0 ? t->ldr : t->ldy;
    a.ncol = (t->Co + 3) / 4 * 4;
    const uint64_t M = (uint64_t)d->N * d->Ti * d->Hi * d->Wi;
    a.y_bytes = (unsigned)(M * t->ldy * 4ull);
    a.r_bytes = (unsigned)(M * a.ldr * 4ull);
    a.w2_bytes = (unsigned)(ptx_conv_body_tail_f32_weight_elems(t) * 4ull);
    const dim3 grid((unsigned)(a.n_tall + a.n_sq), 1u);
    int rc = launch_body(g.tall_per_frame > 0, true, a, grid, lds, stream);
    if (rc != PTX_OK) return rc;
    return hip_check(hipGetLastError(), "conv_body_chain_f32 launch");
}

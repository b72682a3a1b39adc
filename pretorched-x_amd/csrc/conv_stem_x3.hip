// Small-Cin stem convolution with split operands, straight from a channels-last input whose positions are 16 bytes
// (C <= 4 padded to 4): Conv3d(3, 64, 7, stride (1,2,2), pad 3) of the ResNet3D family (resnet3D.py:153), the I3D and
// 2-D ResNet stems, the (1,7,7) spatial stem of the (2+1)D nets.
//
// Why a dedicated kernel.  On the generic implicit-GEMM tiles the stem is a (7,7,1) conv over a kW-FOLDED copy of the
// input (21 live floats per position in a 128-byte row): every (kt, kh) tap re-stages its 256-row A tile through
// L2 -> LDS -- 10.8 GB per launch at config 2 -- and with split operands (3 fp16 MFMAs per product block, DESIGN.md
// 3.3) that stream, not the matrix cores, bounds the kernel (1.05 ms, MFMA 28 % busy).  Here a workgroup owns R output
// rows of one output frame, stages the INPUT PATCH of a temporal tap once -- (R-1)*sH + kH rows x (Wo-1)*sW + 8 positions
// of 16 bytes, 48 KB for R = 4 -- and serves all kH x kW taps of that frame from it: the K axis of one MFMA block is a
// (kw pair) x (4 channels) = 8 consecutive floats of the patch row, i.e. two adjacent positions, so fragments are plain
// 32-byte reads at (row 2r + kh, position 2w + kw).  L2 -> LDS traffic drops ~6x; the fold pass disappears (the input
// conversion writes 103 MB instead of 411 MB).
//
// Arithmetic: identical to the x3 tiles -- a = hi + lo halfs, a.b = hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16,
// fp32 accumulate; the filter is packed by ptx_pack_conv_weight with fold_kw = 1, Ci = 4 (channel 3 zero), Kc = 32,
// f16 = 2: row k = kw * 4 + c, 8-channel blocks as (hi8 | lo8).  The INPUT arrives already split
// (ptx_ncdhw_to_split4: a 16-byte position = (hi4 | lo4) halfs of its 4 channels), so the two positions of a kw pair are
// regrouped into the (hi8, lo8) operands by register naming alone: no VALU between the LDS read and the MFMA.
#include "ptx_common.h"
#include <algorithm>

namespace ptx {

struct StemArgs {
    const float* x;       // [N][Ti][Hi][Wi] positions of 16 bytes: (hi4 | lo4) halfs (ptx_ncdhw_to_split4)
    const float* w;       // [kT*kH][Co_pad][32 words] split halfs
    const float* bias;
    float* y;             // [N][To][Ho][Wo][ldy]
    int N, Ti, Hi, Wi, To, Ho, Wo, Co, ldy, ncol;
    int kT, kH, sT, sH, sW, pT, pH, pW;
    int R, PR, PC;        // output rows per tile; patch rows / positions per row
    int h_tiles, n_tiles; // tiles along Ho; total tiles = N * To * h_tiles
    int n_pieces;         // 1-KiB DMA pieces of one patch
    int w_rows;
    unsigned flags;
    unsigned x_bytes, w_bytes, y_bytes;
    unsigned dv_wo[2];    // fast division by Wo
};

__device__ __forceinline__ unsigned fdiv(unsigned n, const unsigned (&dv)[2]) {
    return dv[0] ? (__umulhi(n, dv[0]) >> dv[1]) : n;
}
static inline void fdiv_make(unsigned d, unsigned (&out)[2]) {
    if (d <= 1) { out[0] = 0; out[1] = 0; return; }
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    out[0] = (unsigned)(((1ull << (31 + l)) + d - 1) / d);
    out[1] = l - 1;
}

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mma16(f32x4 a, f32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}
constexpr int kStemWaves = 7;                 // 7 waves x 64 rows = 448 output positions per workgroup
constexpr int kStemNT = 64 * kStemWaves;
constexpr int kStemRows = 64 * kStemWaves;
constexpr int kStemBN = 64;                   // output channels per workgroup
constexpr int kStemPatchMax = 3072;           // positions (16 B each) of one patch buffer: 48 KiB
constexpr int kStemBTile = kStemBN * 32;      // floats of one (kt, kh) filter tile: 8 KiB
constexpr int kStemPiecesPerWave = (kStemPatchMax / 64 + kStemWaves - 1) / kStemWaves;   // 7

__global__ void __launch_bounds__(kStemNT) conv_stem_x3_kernel(const StemArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                                   // [2][kStemPatchMax][4]
    float* Bs = smem + 2 * kStemPatchMax * 4;           // [2][64][32]
    constexpr unsigned kOOB = 0x80000000u;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // tile order: the frames of one row band follow each other (temporal L2 reuse of the 7-frame window), an XCD owns
    // a contiguous chunk of the list
    const int tile = xcd_remap(blockIdx.x, p.n_tiles);
    const int n0 = blockIdx.y * kStemBN;
    const int to = tile % p.To;
    int t_ = tile / p.To;
    const int ht = t_ % p.h_tiles;
    const int n = t_ / p.h_tiles;
    const int ho0 = ht * p.R;

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);

    // ---- per-lane DMA sources of the patch pieces this wave moves (frame independent) ----
    unsigned a_src[kStemPiecesPerWave];
    const int h_base = ho0 * p.sH - p.pH;
#pragma unroll
    for (int i = 0; i < kStemPiecesPerWave; ++i) {
        const int q = (wave + kStemWaves * i) * 64 + lane;          // patch position
        const unsigned pr = (unsigned)q / (unsigned)p.PC;           // (PC is not a power of two; 7 divisions per thread, once)
        const int pc = q - (int)pr * p.PC;
        const int h = h_base + (int)pr, w = pc - p.pW;
        const bool ok = q < p.PR * p.PC && (unsigned)h < (unsigned)p.Hi && (unsigned)w < (unsigned)p.Wi;
        a_src[i] = ok ? (unsigned)((h * p.Wi + w) * 16) : kOOB;
    }
    // filter tile: slot idx -> row idx / 8, 16-byte slot (idx % 8) holding logical slot (idx % 8) ^ ((row >> 1) & 7)
    unsigned b_src[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + kStemNT * i;
        const int row = idx >> 3, ps = idx & 7;
        const int lslot = ps ^ ((row >> 1) & 7);
        b_src[i] = (idx < kStemBN * 8 && n0 + row < p.w_rows) ? (unsigned)(((n0 + row) * 32 + lslot * 4) * 4) : kOOB;
    }

    // ---- valid temporal taps (uniform): frames outside the clip contribute nothing ----
    const int t_first = to * p.sT - p.pT;
    int kt_lo = max(0, -t_first), kt_hi = min(p.kT - 1, p.Ti - 1 - t_first);
    const int n_kt = kt_hi - kt_lo + 1;

    auto issue_a_piece = [&](int buf, int i, int kt) {
        if (wave + kStemWaves * i < p.n_pieces) {
            const unsigned fbase = (unsigned)(((n * p.Ti + t_first + kt) * p.Hi) * p.Wi) * 16u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(As + buf * kStemPatchMax * 4 + (wave + kStemWaves * i) * 256), 16,
                                                     a_src[i] == kOOB ? kOOB : a_src[i] + fbase, 0, 0, 0);
        }
    };
    auto issue_b = [&](int buf, int kt, int kh) {
        const unsigned tbase = (unsigned)((kt * p.kH + kh) * p.w_rows * 32 * 4);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (wave * 64 + kStemNT * i < kStemBN * 8)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(Bs + buf * kStemBTile + (wave * 64 + kStemNT * i) * 4), 16,
                                                         b_src[i] == kOOB ? kOOB : b_src[i] + tbase, 0, 0, 0);
    };

    // ---- this lane's output rows: ml = wave * 64 + i * 32 + lane % 32 -> (r, wo) ----
    const int g = lane >> 5, l32 = lane & 31;
    int a_row[2];            // float offset of (patch row r * sH, position wo * sW) -- kh / kw offsets are added per step
    bool row_ok[2];
    int m_out[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ml = wave * 64 + i * 32 + l32;
        const unsigned r = fdiv((unsigned)ml, p.dv_wo);
        const int wo = ml - (int)r * p.Wo;
        row_ok[i] = (int)r < p.R && ho0 + (int)r < p.Ho;
        const int rr = row_ok[i] ? (int)r : 0, ww = row_ok[i] ? wo : 0;
        a_row[i] = ((rr * p.sH) * p.PC + ww * p.sW) * 4;
        m_out[i] = ((n * p.To + to) * p.Ho + ho0 + rr) * p.Wo + ww;
    }
    const int b_rowoff = l32 * 32;                       // filter row of this lane inside a 32-row block
    const int b_sw = (l32 >> 1) & 7;

    f32x16 acc[2][2], acc2[2][2];            // acc2: the 2^12-scaled cross terms (scaled lo halves, conv_igemm.hip X3)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; acc2[i][j][r] = 0.f; }

    if (n_kt > 0) {
        // prologue: the first patch (all pieces) and the first filter tile
#pragma unroll
        for (int i = 0; i < kStemPiecesPerWave; ++i) issue_a_piece(0, i, kt_lo);
        issue_b(0, kt_lo, 0);
        int bbuf = 0;
        for (int ik = 0; ik < n_kt; ++ik) {
            const int kt = kt_lo + ik;
            const int abuf = ik & 1;
            for (int kh = 0; kh < p.kH; ++kh) {
                // everything issued so far has landed, and every wave is done with the buffers about to be refilled
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                int off_a0 = a_row[0], off_a1 = a_row[1];
                asm volatile("; LDS reads of this step depend on these" : "+v"(off_a0), "+v"(off_a1)::"memory");
                // refill: the next filter tile, and this step's share of the next frame's patch
                const bool last_kh = kh == p.kH - 1;
                if (!last_kh) issue_b(bbuf ^ 1, kt, kh + 1);
                else if (ik + 1 < n_kt) issue_b(bbuf ^ 1, kt + 1, 0);
                if (ik + 1 < n_kt) {
                    // kStemPiecesPerWave pieces spread over the kH steps of this frame
                    const int per = (kStemPiecesPerWave + p.kH - 1) / p.kH;
#pragma unroll
                    for (int i = 0; i < kStemPiecesPerWave; ++i)
                        if (i / per == kh) issue_a_piece(abuf ^ 1, i, kt + 1);
                }
                const float* Ab = As + abuf * kStemPatchMax * 4 + kh * p.PC * 4;
                const float* Bb = Bs + bbuf * kStemBTile + b_rowoff;
#pragma unroll
                for (int j16 = 0; j16 < 2; ++j16) {
                    // K block of this lane group: kw pair (4 j16 + 2 g, + 1) = two adjacent patch positions = 8 floats
                    const int kwo = (4 * j16 + 2 * g) * 4;
                    f32x4 ahi[2], alo[2];
                    {
                        const f32x4 r0 = *reinterpret_cast<const f32x4*>(Ab + off_a0 + kwo);
                        const f32x4 r1 = *reinterpret_cast<const f32x4*>(Ab + off_a0 + kwo + 4);
                        const f32x4 q0 = *reinterpret_cast<const f32x4*>(Ab + off_a1 + kwo);
                        const f32x4 q1 = *reinterpret_cast<const f32x4*>(Ab + off_a1 + kwo + 4);
                        // position = (hi4 | lo4): the pair's hi halves / lo halves form the two K = 8 operands
                        ahi[0] = f32x4{r0.x, r0.y, r1.x, r1.y}; alo[0] = f32x4{r0.z, r0.w, r1.z, r1.w};
                        ahi[1] = f32x4{q0.x, q0.y, q1.x, q1.y}; alo[1] = f32x4{q0.z, q0.w, q1.z, q1.w};
                    }
                    const int b2 = (j16 * 2 + g) * 2;
                    f32x4 bhi[2], blo[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        bhi[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * 32 + ((b2 ^ b_sw) * 4));
                        blo[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * 32 + (((b2 + 1) ^ b_sw) * 4));
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc2[i][j] = mma16(ahi[i], blo[j], acc2[i][j]);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc2[i][j] = mma16(alo[i], bhi[j], acc2[i][j]);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = mma16(ahi[i], bhi[j], acc[i][j]);
                }
                bbuf ^= 1;
            }
        }
    }

    // ---- epilogue: bias (+ folded BN) + ReLU; lane = output channel, 16 rows per accumulator tile ----
    const bool relu = (p.flags & PTX_EPI_RELU) != 0;
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int co = n0 + j * 32 + l32;
        const bool co_ok = co < p.ncol;
        const float bv = (p.bias && co_ok) ? p.bias[co] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // accumulator element r of this lane belongs to tile row (r & 3) + 8 * (r >> 2) + 4 * g: fetch that
                // row's output index / validity from the lane that owns it as an A row
                const int trow = (r & 3) + 8 * (r >> 2) + 4 * g;
                const int m = __shfl(m_out[i], trow, 64);
                const int ok = __shfl((int)row_ok[i], trow, 64);
                float v = fmaf(acc2[i][j][r], 1.0f / 4096.0f, acc[i][j][r]) + bv;
                v = relu ? fmaxf(v, 0.f) : v;
                const unsigned off = ((unsigned)m * (unsigned)p.ldy + (unsigned)co) * 4u;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_y, (co_ok && ok) ? off : kOOB, 0, 0);
            }
        }
    }
}


// ---- the PLANAR variant (round 4; VERDICT r3 #6: "issue fewer MFMAs") -------------------------------------------------------
// Of the 32 K slots of a (kt, kh) row above only 21 are live: channel 3 of every position is zero and so is kw = 7.  Here the
// input is SIX half planes per frame (c0 c1 c2 hi, then lo: ptx_ncdhw_to_split_planes) and the K = 8 operand of a lane group
// is one (kh, channel) CHUNK: 8 consecutive columns of one plane's patch row -- kw' = 0..7 with the filter shifted so the
// window starts on an even column (kw' = kw + (pW & 1); the unused end is a zero weight).  A row is 3 chunks instead of 4,
// two rows are 3 MFMA blocks instead of 4, a 7-row tap is 11 instead of 14: 21 % fewer v_mfma_f32_32x32x16_f16, a patch of
// 12 instead of 16 bytes per pixel, a filter stream of 48 instead of 64 bytes per row and output channel, and one barrier per
// TWO rows.  The 16 bytes of a chunk sit at a 4-byte-aligned LDS address (column 2 wo + const halfs): four ds_read_b32 per
// operand, conflict-free (consecutive lanes, consecutive dwords).  Arithmetic identical to the kernel above.
struct StemPArgs {
    const _Float16* x;    // [N][Ti][6][Hi][Wi] halfs: planes c0 c1 c2 hi, c0 c1 c2 lo (lo scaled by 2^12)
    const _Float16* w;    // [kT][NS][w_tiles][3 blocks][64 co][4 slots][8 halfs] (ptx_pack_stem_x3p_weight, slots swizzled)
    const float* bias;
    float* y;             // [N][To][Ho][Wo][ldy]
    int N, Ti, Hi, Wi, To, Ho, Wo, ldy, ncol;
    int kT, kH, sT, sH, pT, pH;
    int R, PR, PCW;       // output rows per tile; patch rows; halfs per patch row (a multiple of 8)
    int P2;               // a window starts at input column 2 wo - P2 (P2 = pW rounded up to even)
    int NS;               // steps per temporal tap: (kH + 1) / 2 row pairs
    int h_tiles, n_tiles, n_pieces, w_tiles;
    unsigned flags;
    unsigned x_bytes, w_bytes, y_bytes;
    unsigned dv_wo[2], dv_pp[2], dv_pc8[2];     // fast division by Wo, by the pieces of a plane, by the pieces of a patch row
};

constexpr int kPPatchBytes = 49152;                    // one patch buffer
constexpr int kPBTile = 3 * 64 * 64;                   // bytes of one step's filter tile: 3 blocks x 64 channels x (2 chunks x (hi8 | lo8))
constexpr int kPPiecesPerWave = (kPPatchBytes / 1024 + kStemWaves - 1) / kStemWaves;   // 7

__device__ __forceinline__ f32x4 lds_chunk(const char* p) {       // 16 bytes at a 4-byte-aligned LDS address
    const unsigned* q = reinterpret_cast<const unsigned*>(p);
    return __builtin_bit_cast(f32x4, uint4{q[0], q[1], q[2], q[3]});
}

__global__ void __launch_bounds__(kStemNT) conv_stem_x3p_kernel(const StemPArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smemp[];
    char* As = smemp;                                   // [2][kPPatchBytes]
    char* Bs = smemp + 2 * kPPatchBytes;                // [2][kPBTile]
    constexpr unsigned kOOB = 0x80000000u;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = xcd_remap(blockIdx.x, p.n_tiles);
    const int nt = blockIdx.y, n0 = nt * kStemBN;
    const int to = tile % p.To;
    int t_ = tile / p.To;
    const int ht = t_ % p.h_tiles;
    const int n = t_ / p.h_tiles;
    const int ho0 = ht * p.R;

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.w), 0, p.w_bytes, 0x00020000);

    // ---- per-lane DMA sources of the patch pieces this wave moves (frame independent): piece q = 8 columns of one plane row ----
    unsigned a_src[kPPiecesPerWave];
    const int h_base = ho0 * p.sH - p.pH;
    const int pc8 = p.PCW >> 3, per_plane = p.PR * pc8;
#pragma unroll
    for (int i = 0; i < kPPiecesPerWave; ++i) {
        const int q = (wave + kStemWaves * i) * 64 + lane;
        const int pl = (int)fdiv((unsigned)q, p.dv_pp);        // (multiply-shift: prologue VALU work competes with the MFMAs of
        const int rem = q - pl * per_plane;                   //  the CU's other waves, scripts/gpu_stem_timeline.py)
        const int pr = (int)fdiv((unsigned)rem, p.dv_pc8);
        const int h = h_base + pr, w = (rem - pr * pc8) * 8 - 8;            // patch column 0 is input column -8
        const bool ok = pl < 6 && (unsigned)h < (unsigned)p.Hi && (unsigned)w < (unsigned)p.Wi;
        a_src[i] = ok ? (unsigned)(((pl * p.Hi + h) * p.Wi + w) * 2) : kOOB;
    }
    const unsigned frame_bytes = (unsigned)(6 * p.Hi * p.Wi * 2);

    // ---- valid temporal taps (uniform) ----
    const int t_first = to * p.sT - p.pT;
    const int kt_lo = max(0, -t_first), kt_hi = min(p.kT - 1, p.Ti - 1 - t_first);
    const int n_kt = kt_hi - kt_lo + 1;

    auto issue_a_piece = [&](int buf, int i, int kt) {
        if (wave + kStemWaves * i < p.n_pieces) {
            const unsigned fbase = (unsigned)(n * p.Ti + t_first + kt) * frame_bytes;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(As + buf * kPPatchBytes + (wave + kStemWaves * i) * 1024), 16,
                                                     a_src[i] == kOOB ? kOOB : a_src[i] + fbase, 0, 0, 0);
        }
    };
    auto issue_b = [&](int buf, int kt, int s) {                     // 768 pieces of 16 bytes, stored contiguously (pre-swizzled)
        const unsigned tbase = (unsigned)(((kt * p.NS + s) * p.w_tiles + nt) * kPBTile);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (wave * 64 + kStemNT * i < kPBTile / 16)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(Bs + buf * kPBTile + (wave * 64 + kStemNT * i) * 16), 16,
                                                         tbase + (unsigned)((tid + kStemNT * i) * 16), 0, 0, 0);
    };

    // ---- this lane's output positions: ml = wave * 64 + i * 32 + lane % 32 -> (r, wo) ----
    const int g = lane >> 5, l32 = lane & 31;
    int a_row[2];            // byte offset of (patch row r * sH, column 2 wo - P2 + 8) inside a plane
    bool row_ok[2];
    int m_out[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ml = wave * 64 + i * 32 + l32;
        const unsigned r = fdiv((unsigned)ml, p.dv_wo);
        const int wo = ml - (int)r * p.Wo;
        row_ok[i] = (int)r < p.R && ho0 + (int)r < p.Ho;
        const int rr = row_ok[i] ? (int)r : 0, ww = row_ok[i] ? wo : 0;
        a_row[i] = ((rr * p.sH) * p.PCW + 2 * ww - p.P2 + 8) * 2;
        m_out[i] = ((n * p.To + to) * p.Ho + ho0 + rr) * p.Wo + ww;
    }
    // chunk of (block b, lane group g) inside a row pair: chunks run (kh0,c0) (kh0,c1) (kh0,c2) (kh0+1,c0) (kh0+1,c1) (kh0+1,c2)
    const int plane_b = p.PR * p.PCW * 2, row_b = p.PCW * 2, lo_b = 3 * plane_b;
    const int o_blk0 = g ? plane_b : 0;
    const int o_blk1 = g ? row_b : 2 * plane_b;
    const int o_blk2 = g ? row_b + 2 * plane_b : row_b + plane_b;
    const int o_blk1_last = 2 * plane_b;     // odd kH: the second row of the last pair does not exist -- group 1 multiplies a zero
                                             // filter chunk by (finite) data it re-reads from group 0's address
    const int b_lane = (l32 * 4) * 16;       // this lane's filter row inside a 32-channel half block: 4 slots of 16 bytes
    const int b_sw = (l32 >> 2) & 3;         // 64-byte rows: physical slot = logical ^ ((row >> 2) & 3)  (row = 32 j + l32: same bits)

    f32x16 acc[2][2], acc2[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; acc2[i][j][r] = 0.f; }

    if (n_kt > 0) {
#pragma unroll
        for (int i = 0; i < kPPiecesPerWave; ++i) issue_a_piece(0, i, kt_lo);
        issue_b(0, kt_lo, 0);
        int bbuf = 0;
        const bool odd = (p.kH & 1) != 0;
        for (int ik = 0; ik < n_kt; ++ik) {
            const int kt = kt_lo + ik;
            const int abuf = ik & 1;
            for (int s = 0; s < p.NS; ++s) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                int off_a0 = a_row[0], off_a1 = a_row[1];
                asm volatile("; LDS reads of this step depend on these" : "+v"(off_a0), "+v"(off_a1)::"memory");
                const bool last_s = s == p.NS - 1;
                if (!last_s) issue_b(bbuf ^ 1, kt, s + 1);
                else if (ik + 1 < n_kt) issue_b(bbuf ^ 1, kt + 1, 0);
                if (ik + 1 < n_kt) {
                    const int per = (kPPiecesPerWave + p.NS - 1) / p.NS;
#pragma unroll
                    for (int i = 0; i < kPPiecesPerWave; ++i)
                        if (i / per == s) issue_a_piece(abuf ^ 1, i, kt + 1);
                }
                const char* Ab = As + abuf * kPPatchBytes + (2 * s) * row_b;
                const char* Bb = Bs + bbuf * kPBTile + b_lane;
                const bool half_step = last_s && odd;
#pragma unroll
                for (int blk = 0; blk < 3; ++blk) {
                    if (blk == 2 && half_step) break;                      // (uniform)
                    const int o = blk == 0 ? o_blk0 : blk == 1 ? (half_step ? o_blk1_last : o_blk1) : o_blk2;
                    f32x4 ahi[2], alo[2];
                    ahi[0] = lds_chunk(Ab + off_a0 + o);
                    alo[0] = lds_chunk(Ab + off_a0 + o + lo_b);
                    ahi[1] = lds_chunk(Ab + off_a1 + o);
                    alo[1] = lds_chunk(Ab + off_a1 + o + lo_b);
                    f32x4 bhi[2], blo[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const char* bj = Bb + (blk * 64 + j * 32) * 64;
                        bhi[j] = *reinterpret_cast<const f32x4*>(bj + (((2 * g) ^ b_sw) << 4));
                        blo[j] = *reinterpret_cast<const f32x4*>(bj + (((2 * g + 1) ^ b_sw) << 4));
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc2[i][j] = mma16(ahi[i], blo[j], acc2[i][j]);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc2[i][j] = mma16(alo[i], bhi[j], acc2[i][j]);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = mma16(ahi[i], bhi[j], acc[i][j]);
                }
                bbuf ^= 1;
            }
        }
    }

    // ---- epilogue: as the kernel above ----
    const bool relu = (p.flags & PTX_EPI_RELU) != 0;
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int co = n0 + j * 32 + l32;
        const bool co_ok = co < p.ncol;
        const float bv = (p.bias && co_ok) ? p.bias[co] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int trow = (r & 3) + 8 * (r >> 2) + 4 * g;
                const int m = __shfl(m_out[i], trow, 64);
                const int ok = __shfl((int)row_ok[i], trow, 64);
                float v = fmaf(acc2[i][j][r], 1.0f / 4096.0f, acc[i][j][r]) + bv;
                v = relu ? fmaxf(v, 0.f) : v;
                const unsigned off = ((unsigned)m * (unsigned)p.ldy + (unsigned)co) * 4u;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_y, (co_ok && ok) ? off : kOOB, 0, 0);
            }
        }
    }
}

// w_x3: ptx_pack_conv_weight(fold_kw = 1, Ci = 4, Kc = 32, f16 = 2) = [kT*kH][Co_pad][64 halfs], 8-channel blocks (hi8 | lo8) of
// k = kw * 4 + c  ->  the planar stem's step tiles.  One thread per destination half.
__global__ void __launch_bounds__(256) pack_stem_x3p_kernel(const _Float16* __restrict__ src, _Float16* __restrict__ dst, int kT, int kH,
                                                            int kW, int Co_pad, int w_tiles, int NS, int shift, long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int kwp = (int)(i & 7);
        long long t = i >> 3;
        const int phys = (int)(t & 3);
        t >>= 2;
        const int co_in = (int)(t & 63);
        t >>= 6;
        const int blk = (int)(t % 3);
        t /= 3;
        const int nt = (int)(t % w_tiles);
        t /= w_tiles;
        const int s = (int)(t % NS), kt = (int)(t / NS);
        const int slot = phys ^ ((co_in >> 2) & 3);              // logical slot this physical one holds
        const int gch = slot >> 1, lo = slot & 1;
        const int qs = blk * 2 + gch;                            // chunk inside the row pair
        const int kh = 2 * s + qs / 3, c = qs % 3;
        const int kw = kwp - shift;
        const int co = nt * 64 + co_in;
        _Float16 v = (_Float16)0.f;
        if (kh < kH && kw >= 0 && kw < kW && co < Co_pad)
            v = src[((size_t)(kt * kH + kh) * Co_pad + co) * 64 + (kw >> 1) * 16 + lo * 8 + (kw & 1) * 4 + c];
        dst[i] = v;
    }
}

// fp32 [N][C][T][H][W] -> halfs [N][T][6][H][W]: planes c0 c1 c2 hi, c0 c1 c2 lo.  A thread converts 8 columns of one row;
// a block of 256 threads is TY rows x TX column groups (TX = the power of two >= W / 8), rows = (n, t, h) triples decoded with
// 32-bit arithmetic (64-bit divisions per thread made the first version run at 2.9 TB/s)
__global__ void __launch_bounds__(256) ncdhw_to_split_planes_kernel(const float* __restrict__ x, _Float16* __restrict__ y, int C, int T,
                                                                    int H, int W8, int rows, int tx_log2) {
    const size_t plane = (size_t)H * W8 * 8, S = (size_t)T * plane;
    const int tx = threadIdx.x & ((1 << tx_log2) - 1), ty = threadIdx.x >> tx_log2, TY = 256 >> tx_log2;
    for (unsigned row = blockIdx.x * TY + ty; row < (unsigned)rows; row += gridDim.x * TY) {      // row = (n * T + t) * H + h
        const unsigned nt = row / (unsigned)H, h = row - nt * (unsigned)H;
        const unsigned n = nt / (unsigned)T, t = nt - n * (unsigned)T;
        for (int w8 = tx; w8 < W8; w8 += 1 << tx_log2) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                half8 hi, lo;
                if (c < C) {
                    const float* xs = x + ((size_t)n * C + c) * S + (size_t)t * plane + ((size_t)h * W8 + w8) * 8;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(xs), v1 = *reinterpret_cast<const f32x4*>(xs + 4);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float v = e < 4 ? v0[e] : v1[e - 4];
                        const _Float16 hh = (_Float16)v;
                        hi[e] = hh;
                        lo[e] = (_Float16)((v - (float)hh) * 4096.f);      // scaled lo (conv_igemm.hip, X3)
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { hi[e] = (_Float16)0.f; lo[e] = (_Float16)0.f; }
                }
                _Float16* yp = y + ((size_t)nt * 6 + c) * plane + ((size_t)h * W8 + w8) * 8;
                *reinterpret_cast<half8*>(yp) = hi;
                *reinterpret_cast<half8*>(yp + 3 * plane) = lo;
            }
        }
    }
}

}  // namespace ptx

using namespace ptx;

extern "C" int ptx_conv_stem_x3_supported(const ptx_conv3d_desc* d) {
    if (!d) return 0;
    if (!(d->flags & PTX_F16X3_OPERANDS) || (d->flags & ~(PTX_F16X3_OPERANDS | PTX_EPI_RELU | PTX_SPLITK_FUSED))) return 0;
    if (d->Ci < 1 || d->Ci > 4 || d->ldx != 4 || d->Kc != 32 || d->kW < 1 || d->kW > 8 || d->kT < 1 || d->kT > 8 || d->kH < 1 ||
        d->kH > 8 || d->groups > 1 || d->Co_pad % 128)
        return 0;
    if (d->sW < 1 || d->sW > 2 || d->sH < 1 || d->sT < 1 || d->Wo < 1 || d->Wo > kStemRows) return 0;
    const int R = std::min(kStemRows / d->Wo, d->Ho);
    const int PR = (R - 1) * d->sH + d->kH, PC = (d->Wo - 1) * d->sW + 8;
    if (R < 1 || (int64_t)PR * PC > kStemPatchMax) return 0;
    if ((int64_t)d->N * d->Ti * d->Hi * d->Wi * 16 >= 0x80000000LL || (int64_t)d->N * d->To * d->Ho * d->Wo * d->ldy * 4 >= 0x80000000LL)
        return 0;
    // output extents: symmetric padding p, or TF-"SAME" (out = ceil(in / stride), p = the FRONT pad floor(total / 2); the
    // back pad is implied -- taps beyond the image read zero either way), as ptx_conv3d_fwd accepts them
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

extern "C" int ptx_conv_stem_x3_fwd(const ptx_conv3d_desc* d, const float* x, const float* w_packed, const float* bias, float* y,
                                    ptx_stream_t stream) {
    if (!d || !x || !w_packed || !y) return fail(PTX_ERR_INVALID, "conv_stem_x3: null pointer");
    if (((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)y) & 15) return fail(PTX_ERR_INVALID, "conv_stem_x3: pointers must be 16-byte aligned");
    if (!ptx_conv_stem_x3_supported(d))
        return fail(PTX_ERR_UNSUPPORTED, "conv_stem_x3: needs a split-operand (PTX_F16X3_OPERANDS) stem: Ci <= 4 stored as 4-channel "
                    "positions (ldx == 4), kW <= 8 folded into Kc == 32, stride_w <= 2, symmetric or SAME padding, Wo <= %d and an input "
                    "patch of at most %d positions", kStemRows, kStemPatchMax);
    if (d->ldy < d->Co || d->ldy % 4) return fail(PTX_ERR_INVALID, "conv_stem_x3: bad output stride");
    StemArgs a{};
    a.x = x; a.w = w_packed; a.bias = bias; a.y = y;
    a.N = d->N; a.Ti = d->Ti; a.Hi = d->Hi; a.Wi = d->Wi; a.To = d->To; a.Ho = d->Ho; a.Wo = d->Wo; a.Co = d->Co; a.ldy = d->ldy;
    a.ncol = (d->Co + 3) / 4 * 4;
    a.kT = d->kT; a.kH = d->kH; a.sT = d->sT; a.sH = d->sH; a.sW = d->sW; a.pT = d->pT; a.pH = d->pH; a.pW = d->pW;
    a.R = std::min(kStemRows / d->Wo, d->Ho);
    a.PR = (a.R - 1) * d->sH + d->kH;
    a.PC = (d->Wo - 1) * d->sW + 8;
    a.h_tiles = cdiv(d->Ho, a.R);
    a.n_tiles = d->N * d->To * a.h_tiles;
    a.n_pieces = cdiv(a.PR * a.PC, 64);
    a.w_rows = d->Co_pad;
    a.flags = d->flags;
    a.x_bytes = (unsigned)((uint64_t)d->N * d->Ti * d->Hi * d->Wi * 16ull);
    a.w_bytes = (unsigned)((uint64_t)d->kT * d->kH * d->Co_pad * 32 * 4ull);
    a.y_bytes = (unsigned)((uint64_t)d->N * d->To * d->Ho * d->Wo * d->ldy * 4ull);
    fdiv_make((unsigned)d->Wo, a.dv_wo);
    constexpr size_t lds = (size_t)(2 * kStemPatchMax * 4 + 2 * kStemBTile) * sizeof(float);
    static bool attr_set[64] = {};
    int dev = 0;
    PTX_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        PTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stem_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const dim3 grid((unsigned)a.n_tiles, (unsigned)cdiv(a.ncol, kStemBN));
    hipLaunchKernelGGL(conv_stem_x3_kernel, grid, dim3(kStemNT), lds, (hipStream_t)stream, a);
    return hip_check(hipGetLastError(), "conv_stem_x3 launch");
}

// ---- the planar variant: same descriptor as ptx_conv_stem_x3_fwd, narrower domain ----
static bool stem_x3p_geometry(const ptx_conv3d_desc* d, int& R, int& PR, int& PCW, int& P2) {
    R = std::min(kStemRows / d->Wo, d->Ho);
    PR = (R - 1) * d->sH + d->kH;
    P2 = d->pW + (d->pW & 1);
    PCW = (2 * (d->Wo - 1) - P2 + 16 + 7) / 8 * 8;
    return R >= 1 && (int64_t)6 * PR * PCW * 2 <= kPPatchBytes;
}

extern "C" int ptx_conv_stem_x3p_supported(const ptx_conv3d_desc* d) {
    if (!ptx_conv_stem_x3_supported(d)) return 0;
    if (d->Ci > 3 || d->sW != 2 || d->Wi % 8 || d->kW > 7 || d->pW < 0 || d->pW + (d->pW & 1) > 8) return 0;
    int R, PR, PCW, P2;
    if (!stem_x3p_geometry(d, R, PR, PCW, P2)) return 0;
    return (int64_t)d->N * d->Ti * 6 * d->Hi * d->Wi * 2 < 0x80000000LL;
}

extern "C" size_t ptx_stem_x3p_weight_elems(const ptx_conv3d_desc* d) {         // in floats (4-byte units)
    if (!d || d->kT < 1 || d->kH < 1 || d->Co < 1) return 0;
    const int w_tiles = cdiv((d->Co + 3) / 4 * 4, kStemBN);
    return (size_t)d->kT * ((d->kH + 1) / 2) * w_tiles * (kPBTile / 4);
}

extern "C" int ptx_pack_stem_x3p_weight(const ptx_conv3d_desc* d, const float* w_x3, float* w_stem, ptx_stream_t stream) {
    if (!d || !w_x3 || !w_stem) return fail(PTX_ERR_INVALID, "pack_stem_x3p_weight: null pointer");
    if (!ptx_conv_stem_x3p_supported(d)) return fail(PTX_ERR_UNSUPPORTED, "pack_stem_x3p_weight: not a planar split-operand stem");
    const int w_tiles = cdiv((d->Co + 3) / 4 * 4, kStemBN), NS = (d->kH + 1) / 2;
    const long long total = (long long)ptx_stem_x3p_weight_elems(d) * 2;
    const unsigned blocks = (unsigned)std::min<long long>((total + 255) / 256, (long long)kNumCU * 16);
    hipLaunchKernelGGL(pack_stem_x3p_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const _Float16*>(w_x3),
                       reinterpret_cast<_Float16*>(w_stem), d->kT, d->kH, d->kW, d->Co_pad, w_tiles, NS, d->pW & 1, total);
    return hip_check(hipGetLastError(), "pack_stem_x3p_weight launch");
}

extern "C" int ptx_ncdhw_to_split_planes(const float* x, void* y, int32_t N, int32_t C, int32_t T, int32_t H, int32_t W, ptx_stream_t stream) {
    if (!x || !y) return fail(PTX_ERR_INVALID, "ncdhw_to_split_planes: null pointer");
    if (N <= 0 || C <= 0 || C > 3 || T <= 0 || H <= 0 || W <= 0 || W % 8 || ((uintptr_t)x & 15) || ((uintptr_t)y & 15))
        return fail(PTX_ERR_INVALID, "ncdhw_to_split_planes: 1..3 channels, W a multiple of 8, 16-byte aligned pointers");
    if ((int64_t)N * T * H > 0x7fffffffLL) return fail(PTX_ERR_INVALID, "ncdhw_to_split_planes: too many rows");
    const int rows = N * T * H;
    int tx_log2 = 0;
    while ((1 << tx_log2) < W / 8 && tx_log2 < 8) ++tx_log2;
    const int TY = 256 >> tx_log2;
    const unsigned blocks = (unsigned)std::min(cdiv(rows, TY), kNumCU * 32);
    hipLaunchKernelGGL(ncdhw_to_split_planes_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, static_cast<_Float16*>(y), C, T, H, W / 8, rows,
                       tx_log2);
    return hip_check(hipGetLastError(), "ncdhw_to_split_planes launch");
}

extern "C" int ptx_conv_stem_x3p_fwd(const ptx_conv3d_desc* d, const void* x, const float* w_stem, const float* bias, float* y,
                                     ptx_stream_t stream) {
    if (!d || !x || !w_stem || !y) return fail(PTX_ERR_INVALID, "conv_stem_x3p: null pointer");
    if (((uintptr_t)x | (uintptr_t)w_stem | (uintptr_t)y) & 15) return fail(PTX_ERR_INVALID, "conv_stem_x3p: pointers must be 16-byte aligned");
    if (!ptx_conv_stem_x3p_supported(d))
        return fail(PTX_ERR_UNSUPPORTED, "conv_stem_x3p: needs a split-operand stem ptx_conv_stem_x3_fwd accepts with Ci <= 3, stride_w == 2, kW <= 7, "
                    "an input width that is a multiple of 8 and a six-plane patch of at most %d bytes", kPPatchBytes);
    if (d->ldy < d->Co || d->ldy % 4) return fail(PTX_ERR_INVALID, "conv_stem_x3p: bad output stride");
    StemPArgs a{};
    a.x = static_cast<const _Float16*>(x); a.w = reinterpret_cast<const _Float16*>(w_stem); a.bias = bias; a.y = y;
    a.N = d->N; a.Ti = d->Ti; a.Hi = d->Hi; a.Wi = d->Wi; a.To = d->To; a.Ho = d->Ho; a.Wo = d->Wo; a.ldy = d->ldy;
    a.ncol = (d->Co + 3) / 4 * 4;
    a.kT = d->kT; a.kH = d->kH; a.sT = d->sT; a.sH = d->sH; a.pT = d->pT; a.pH = d->pH;
    stem_x3p_geometry(d, a.R, a.PR, a.PCW, a.P2);
    a.NS = (d->kH + 1) / 2;
    a.h_tiles = cdiv(d->Ho, a.R);
    a.n_tiles = d->N * d->To * a.h_tiles;
    a.n_pieces = cdiv(6 * a.PR * (a.PCW / 8), 64);
    a.w_tiles = cdiv(a.ncol, kStemBN);
    a.flags = d->flags;
    a.x_bytes = (unsigned)((uint64_t)d->N * d->Ti * 6 * d->Hi * d->Wi * 2ull);
    a.w_bytes = (unsigned)(ptx_stem_x3p_weight_elems(d) * 4ull);
    a.y_bytes = (unsigned)((uint64_t)d->N * d->To * d->Ho * d->Wo * d->ldy * 4ull);
    fdiv_make((unsigned)d->Wo, a.dv_wo);
    fdiv_make((unsigned)(a.PR * (a.PCW / 8)), a.dv_pp);
    fdiv_make((unsigned)(a.PCW / 8), a.dv_pc8);
    constexpr size_t lds = (size_t)(2 * kPPatchBytes + 2 * kPBTile);
    static bool attr_set[64] = {};
    int dev = 0;
    PTX_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        PTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stem_x3p_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const dim3 grid((unsigned)a.n_tiles, (unsigned)a.w_tiles);
    hipLaunchKernelGGL(conv_stem_x3p_kernel, grid, dim3(kStemNT), lds, (hipStream_t)stream, a);
    return hip_check(hipGetLastError(), "conv_stem_x3p launch");
}

// EXPERIMENT, not part of the product (round 3): conv_stem_x3.hip with 32-row wave tiles and W-cut blocks so that two
// 80-KB workgroups share a CU.  Same-box A/B against the shipped kernel (scripts/gpu_stem_x3_ab.py): config-2 stem
// 0.848 vs 0.841 ms, I3D stem 0.470 vs 0.452 ms, (2+1)D spatial stem 0.098 vs 0.098 ms -- no gain; the 0.727 ms first read
// off another box was box-to-box spread.  Kept as the evidence for DESIGN.md 3.7.  Build:
//   cd pretorched-x_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -w -c ../../scripts/micro/conv_stem_x3_2wg.hip -o /tmp/stem2wg.o
//   hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scripts/micro/libptx_stem2wg.so conv_igemm.o conv_chain.o pack_layout.o pool_head.o nonlocal_attn.o conv_stem_f32.o /tmp/stem2wg.o
// Small-Cin stem convolution with split operands, straight from a channels-last input whose positions are 16 bytes
// (C <= 4 padded to 4): Conv3d(3, 64, 7, stride (1,2,2), pad 3) of the ResNet3D family (resnet3D.py:153), the I3D and
// 2-D ResNet stems, the (1,7,7) spatial stem of the (2+1)D nets.
//
// Why a dedicated kernel.  On the generic implicit-GEMM tiles the stem is a (7,7,1) conv over a kW-FOLDED copy of the
// input (21 live floats per position in a 128-byte row): every (kt, kh) tap re-stages its 256-row A tile through
// L2 -> LDS -- 10.8 GB per launch at config 2 -- and with split operands (3 fp16 MFMAs per product block, DESIGN.md
// 3.3) that stream, not the matrix cores, bounds the kernel (1.05 ms, MFMA 28 % busy).  Here a workgroup owns an
// R x CW block of output positions of one output frame (224 = 7 waves x 32 rows), stages the INPUT PATCH of a temporal
// tap once -- (R-1)*sH + kH rows x (CW-1)*sW + 8 positions of 16 bytes -- and serves all kH x kW taps of that frame
// from it: the K axis of one MFMA block is a (kw pair) x (4 channels) = 8 consecutive floats of the patch row, i.e. two
// adjacent positions, so fragments are plain 32-byte reads at (row sH r + kh, position sW c + kw).  L2 -> LDS traffic
// drops ~6x; the fold pass disappears (the input conversion writes 103 MB instead of 411 MB).
//
// Tile shape (round 3).  The first version gave a workgroup 448 positions (4 whole 112-wide rows, 7 waves x 64 rows x 64
// channels): two 48-KB patch buffers and 128 accumulator registers per wave -- ONE workgroup per CU, whose patch prologue,
// 49 barrier-separated steps and 411 MB worth of store epilogue ran back to back with nothing to overlap them
// (profiles/r03_stem_x3_probe.txt: with DMA, LDS reads and barriers removed the launch still took 0.73 of 0.97 ms).  Now a
// wave owns 32 rows x 64 channels (64 accumulator registers) and the block is cut along W when two patch buffers of the
// full width would not leave room for a second workgroup (112-wide rows: 4 rows x 56 columns, 2 x 24.5 KB): two workgroups
// per CU, one computing while the other waits for a barrier, its patch or its stores.
//
// Arithmetic: identical to the x3 tiles -- a = hi + lo halfs, a.b = hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16,
// fp32 accumulate; the filter is packed by ptx_pack_conv_weight with fold_kw = 1, Ci = 4 (channel 3 zero), Kc = 32,
// f16 = 2: row k = kw * 4 + c, 8-channel blocks as (hi8 | lo8).  The INPUT arrives already split
// (ptx_ncdhw_to_split4: a 16-byte position = (hi4 | lo4) halfs of its 4 channels), so the two positions of a kw pair are
// regrouped into the (hi8, lo8) operands by register naming alone: no VALU between the LDS read and the MFMA.
#include "../../pretorched-x_amd/csrc/ptx_common.h"
#include <algorithm>

namespace ptx {

struct StemArgs {
    const float* x;       // [N][Ti][Hi][Wi] positions of 16 bytes: (hi4 | lo4) halfs (ptx_ncdhw_to_split4)
    const float* w;       // [kT*kH][Co_pad][32 words] split halfs
    const float* bias;
    float* y;             // [N][To][Ho][Wo][ldy]
    int N, Ti, Hi, Wi, To, Ho, Wo, Co, ldy, ncol;
    int kT, kH, sT, sH, sW, pT, pH, pW;
    int R, CW, PR, PC;    // output rows / columns per tile; patch rows / positions per row
    int h_tiles, w_tiles, n_tiles; // tiles along Ho / Wo; total tiles = N * To * h_tiles * w_tiles
    int n_pieces;         // 1-KiB DMA pieces of one patch
    int w_rows;
    unsigned flags;
    unsigned x_bytes, w_bytes, y_bytes;
    unsigned dv_cw[2];    // fast division by CW
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
constexpr int kStemWaves = 7;                 // 7 waves x 32 rows = 224 output positions per workgroup
constexpr int kStemNT = 64 * kStemWaves;
constexpr int kStemRows = 32 * kStemWaves;
constexpr int kStemBN = 64;                   // output channels per workgroup
constexpr int kStemPatchMax = 2048;           // positions (16 B each) of one patch buffer: 32 KiB -- two of them + two filter
                                              // tiles = 80 KiB, i.e. two workgroups per CU
constexpr int kStemBTile = kStemBN * 32;      // floats of one (kt, kh) filter tile: 8 KiB
constexpr int kStemPiecesPerWave = (kStemPatchMax / 64 + kStemWaves - 1) / kStemWaves;   // 5

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
    const int wt = t_ % p.w_tiles;
    t_ /= p.w_tiles;
    const int ht = t_ % p.h_tiles;
    const int n = t_ / p.h_tiles;
    const int ho0 = ht * p.R, wo0 = wt * p.CW;

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);

    // ---- per-lane DMA sources of the patch pieces this wave moves (frame independent) ----
    unsigned a_src[kStemPiecesPerWave];
    const int h_base = ho0 * p.sH - p.pH, w_base = wo0 * p.sW - p.pW;
#pragma unroll
    for (int i = 0; i < kStemPiecesPerWave; ++i) {
        const int q = (wave + kStemWaves * i) * 64 + lane;          // patch position
        const unsigned pr = (unsigned)q / (unsigned)p.PC;           // (PC is not a power of two; 5 divisions per thread, once)
        const int pc = q - (int)pr * p.PC;
        const int h = h_base + (int)pr, w = w_base + pc;
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

    // ---- this lane's output row: ml = wave * 32 + lane % 32 -> (r, c) of the R x CW block ----
    const int g = lane >> 5, l32 = lane & 31;
    int a_row;               // float offset of (patch row r * sH, position c * sW) -- kh / kw offsets are added per step
    bool row_ok;
    int m_out;
    {
        const int ml = wave * 32 + l32;
        const unsigned r = fdiv((unsigned)ml, p.dv_cw);
        const int c = ml - (int)r * p.CW;
        row_ok = (int)r < p.R && ho0 + (int)r < p.Ho && wo0 + c < p.Wo;
        const int rr = row_ok ? (int)r : 0, cc = row_ok ? c : 0;
        a_row = ((rr * p.sH) * p.PC + cc * p.sW) * 4;
        m_out = ((n * p.To + to) * p.Ho + ho0 + rr) * p.Wo + wo0 + cc;
    }
    const int b_rowoff = l32 * 32;                       // filter row of this lane inside a 32-row block
    const int b_sw = (l32 >> 1) & 7;

    f32x16 acc[2], acc2[2];                  // acc2: the 2^12-scaled cross terms (scaled lo halves, conv_igemm.hip X3)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[j][r] = 0.f; acc2[j][r] = 0.f; }

    if (n_kt > 0) {
        // prologue: the first patch (all pieces) and the first filter tile
#pragma unroll
        for (int i = 0; i < kStemPiecesPerWave; ++i) issue_a_piece(0, i, kt_lo);
        issue_b(0, kt_lo, 0);
        // pieces of the next frame's patch are spread over the kH steps of this one
        const int per = (kStemPiecesPerWave + p.kH - 1) / p.kH;
        int piece_kh[kStemPiecesPerWave];
#pragma unroll
        for (int i = 0; i < kStemPiecesPerWave; ++i) piece_kh[i] = i / per;
        int bbuf = 0;
        for (int ik = 0; ik < n_kt; ++ik) {
            const int kt = kt_lo + ik;
            const int abuf = ik & 1;
            for (int kh = 0; kh < p.kH; ++kh) {
                // everything issued so far has landed, and every wave is done with the buffers about to be refilled
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                int off_a = a_row;
                asm volatile("; LDS reads of this step depend on this" : "+v"(off_a)::"memory");
                // refill: the next filter tile, and this step's share of the next frame's patch
                const bool last_kh = kh == p.kH - 1;
                if (!last_kh) issue_b(bbuf ^ 1, kt, kh + 1);
                else if (ik + 1 < n_kt) issue_b(bbuf ^ 1, kt + 1, 0);
                if (ik + 1 < n_kt) {
#pragma unroll
                    for (int i = 0; i < kStemPiecesPerWave; ++i)
                        if (piece_kh[i] == kh) issue_a_piece(abuf ^ 1, i, kt + 1);
                }
                const float* Ab = As + abuf * kStemPatchMax * 4 + kh * p.PC * 4;
                const float* Bb = Bs + bbuf * kStemBTile + b_rowoff;
#pragma unroll
                for (int j16 = 0; j16 < 2; ++j16) {
                    // K block of this lane group: kw pair (4 j16 + 2 g, + 1) = two adjacent patch positions = 8 floats
                    const int kwo = (4 * j16 + 2 * g) * 4;
                    const f32x4 r0 = *reinterpret_cast<const f32x4*>(Ab + off_a + kwo);
                    const f32x4 r1 = *reinterpret_cast<const f32x4*>(Ab + off_a + kwo + 4);
                    // position = (hi4 | lo4): the pair's hi halves / lo halves form the two K = 8 operands
                    const f32x4 ahi = f32x4{r0.x, r0.y, r1.x, r1.y}, alo = f32x4{r0.z, r0.w, r1.z, r1.w};
                    const int b2 = (j16 * 2 + g) * 2;
                    f32x4 bhi[2], blo[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        bhi[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * 32 + ((b2 ^ b_sw) * 4));
                        blo[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * 32 + (((b2 + 1) ^ b_sw) * 4));
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc2[j] = mma16(ahi, blo[j], acc2[j]);
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc2[j] = mma16(alo, bhi[j], acc2[j]);
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[j] = mma16(ahi, bhi[j], acc[j]);
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
        for (int r = 0; r < 16; ++r) {
            // accumulator element r of this lane belongs to tile row (r & 3) + 8 * (r >> 2) + 4 * g: fetch that
            // row's output index / validity from the lane that owns it as an A row
            const int trow = (r & 3) + 8 * (r >> 2) + 4 * g;
            const int m = __shfl(m_out, trow, 64);
            const int ok = __shfl((int)row_ok, trow, 64);
            float v = fmaf(acc2[j][r], 1.0f / 4096.0f, acc[j][r]) + bv;
            v = relu ? fmaxf(v, 0.f) : v;
            const unsigned off = ((unsigned)m * (unsigned)p.ldy + (unsigned)co) * 4u;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_y, (co_ok && ok) ? off : kOOB, 0, 0);
        }
    }
}

}  // namespace ptx

using namespace ptx;

// The R x CW block of output positions one workgroup owns: as many whole (part-)rows as 224 lanes of A rows hold, cut
// along W into the fewest column tiles whose input patch fits one 32-KiB buffer.
struct StemGeom { int R, CW, PR, PC, w_tiles; };
static bool stem_geom(const ptx_conv3d_desc* d, StemGeom& g) {
    for (g.w_tiles = 1; g.w_tiles <= 64; ++g.w_tiles) {
        g.CW = cdiv(d->Wo, g.w_tiles);
        if (g.CW > kStemRows) continue;
        g.R = std::min(kStemRows / g.CW, d->Ho);
        g.PR = (g.R - 1) * d->sH + d->kH;
        g.PC = (g.CW - 1) * d->sW + 8;
        if ((int64_t)g.PR * g.PC <= kStemPatchMax) return true;
    }
    return false;
}

extern "C" int ptx_conv_stem_x3_supported(const ptx_conv3d_desc* d) {
    if (!d) return 0;
    if (!(d->flags & PTX_F16X3_OPERANDS) || (d->flags & ~(PTX_F16X3_OPERANDS | PTX_EPI_RELU | PTX_SPLITK_FUSED))) return 0;
    if (d->Ci < 1 || d->Ci > 4 || d->ldx != 4 || d->Kc != 32 || d->kW < 1 || d->kW > 8 || d->kT < 1 || d->kT > 8 || d->kH < 1 ||
        d->kH > 8 || d->groups > 1 || d->Co_pad % 128)
        return 0;
    if (d->sW < 1 || d->sW > 2 || d->sH < 1 || d->sT < 1 || d->Wo < 1 || d->Ho < 1) return 0;
    StemGeom gm;
    if (!stem_geom(d, gm)) return 0;
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
                    "positions (ldx == 4), kW <= 8 folded into Kc == 32, stride_w <= 2, symmetric or SAME padding and an input patch of at "
                    "most %d positions per column tile", kStemPatchMax);
    if (d->ldy < d->Co || d->ldy % 4) return fail(PTX_ERR_INVALID, "conv_stem_x3: bad output stride");
    StemArgs a{};
    a.x = x; a.w = w_packed; a.bias = bias; a.y = y;
    a.N = d->N; a.Ti = d->Ti; a.Hi = d->Hi; a.Wi = d->Wi; a.To = d->To; a.Ho = d->Ho; a.Wo = d->Wo; a.Co = d->Co; a.ldy = d->ldy;
    a.ncol = (d->Co + 3) / 4 * 4;
    a.kT = d->kT; a.kH = d->kH; a.sT = d->sT; a.sH = d->sH; a.sW = d->sW; a.pT = d->pT; a.pH = d->pH; a.pW = d->pW;
    StemGeom gm;
    stem_geom(d, gm);
    a.R = gm.R; a.CW = gm.CW; a.PR = gm.PR; a.PC = gm.PC;
    a.h_tiles = cdiv(d->Ho, a.R);
    a.w_tiles = gm.w_tiles;
    a.n_tiles = d->N * d->To * a.h_tiles * a.w_tiles;
    a.n_pieces = cdiv(a.PR * a.PC, 64);
    a.w_rows = d->Co_pad;
    a.flags = d->flags;
    a.x_bytes = (unsigned)((uint64_t)d->N * d->Ti * d->Hi * d->Wi * 16ull);
    a.w_bytes = (unsigned)((uint64_t)d->kT * d->kH * d->Co_pad * 32 * 4ull);
    a.y_bytes = (unsigned)((uint64_t)d->N * d->To * d->Ho * d->Wo * d->ldy * 4ull);
    fdiv_make((unsigned)a.CW, a.dv_cw);
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

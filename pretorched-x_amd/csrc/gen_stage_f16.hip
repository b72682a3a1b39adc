// fp16 kernels of the generator's fused stage that are DESIGNED for the fp16 matrix cores (round 4; VERDICT r3 #5: the
// generic implicit-GEMM tiles are fp32-shaped -- a barrier and a round of per-piece address math every 32-64 k, which the
// 16x shorter f16 MFMAs no longer cover).  BigGAN-deep (Brock et al. 2019, appendix B; no source in the reference snapshot,
// SURVEY.md F2 -- parity of everything here is UNPINNED and the tolerance is the builder's).
//
//   ptx_rgb_conv3x3_f16_fwd   the output layer  BN -> ReLU -> conv3x3(ch -> 3) -> tanh  in one launch
//   ptx_conv3x3_f16_fwd       a GBlock's 3x3 convs (64 / 128 / 256 channels) from one staged input patch
//   ptx_conv1x1_skip_f16_fwd  a GBlock's closing 1x1 conv: + skip (upsampled, channel-truncated) and BOTH outputs the next block reads
//   ptx_conv1x1_pro_f16_fwd   a GBlock's opening 1x1 conv with the block's cBN1 + ReLU applied to its input fragments
//
// ---- the image conv ----------------------------------------------------------------------------------------------------
// 3 output channels waste 13/16 of the narrowest MFMA tile, and as an implicit GEMM the layer re-stages its input nine
// times (0.75 ms at 256 x 256 x 128 x 64 images: 1.4 TB/s, 58 VALU + SALU instructions per MFMA).  Put the TAPS in the N
// axis instead: per input position p the kernel computes
//       Z[p][tap * 3 + co] = sum_c relu(x[p][c] * scale[n][c] + shift[n][c]) * w[co][c][tap]          (K = ch, N = 27 -> 32)
// -- ONE v_mfma_f32_32x32x16_f16 per 32 positions and 16 channels, the input read exactly once (plus the tile halo), the
// BatchNorm + ReLU of the output layer applied to the A fragments in registers (v_pk_fma_f16 / v_pk_max_f16), so the
// activated copy of the last feature map never exists -- and then
//       y[h][w][co] = tanh(bias[co] + sum_{kh,kw} Z[(h + kh - 1, w + kw - 1)][(kh * 3 + kw) * 3 + co])
// from the Z tile in LDS (taps outside the image are skipped: the reference pads the ACTIVATED map with zeros).
// Roofline: HBM -- 2 B x ch per position in, 16 B out (1.14 GB at config 5: 0.19 ms at 6 TB/s).
#include "ptx_common.h"
#include <cstdlib>

namespace ptx {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

struct RgbArgs {
    const _Float16* x;      // [N][H][W][ldx] halfs, RAW (pre-BN) feature map
    const float* scale;     // [N][ld_aff] per-sample affine of the output layer's BN (ptx_cbn_fold)
    const float* shift;
    const _Float16* w;      // packed by ptx_pack_rgb_conv_weight: [C/16][2][32][8] halfs
    const float* bias;      // [3] or NULL
    float* y;               // [N][H][W][ldy] fp32, 3 live channels (a 4th column, when ldy >= 4, is written as 0)
    int N, H, W, C, ldx, ld_aff, ldy, tiles_h, tiles_w;
    unsigned x_bytes, flags;
};

constexpr int kRgbTH = 8, kRgbTW = 32;                       // outputs per workgroup: 8 rows x 32 columns, one per thread
constexpr int kRgbPW = kRgbTW + 2, kRgbPH = kRgbTH + 2;       // halo'd patch
constexpr int kRgbPos = kRgbPW * kRgbPH;                      // 340 input positions
constexpr int kRgbTiles = (kRgbPos + 31) / 32;                // 11 MFMA row tiles
constexpr int kRgbZld = 33;                                   // Z row pitch (floats): consecutive positions -> consecutive banks

template <int KS>                                             // C = 16 * KS input channels
__global__ void __launch_bounds__(256, 2) rgb_conv3x3_f16_kernel(const RgbArgs p) {
    __shared__ float Z[kRgbTiles * 32 * kRgbZld];
    constexpr unsigned kOOB = 0x80000000u;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, kg = lane >> 5;
    int t = xcd_remap(blockIdx.x, gridDim.x);                 // neighbouring tiles (shared halos) on one XCD
    const int tw = t % p.tiles_w;
    t /= p.tiles_w;
    const int th = t % p.tiles_h, n = t / p.tiles_h;
    const int h0 = th * kRgbTH, w0 = tw * kRgbTW;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.x), 0, p.x_bytes, 0x00020000);

    // ---- A fragments of this wave's row tiles: requested first, everything else hides under their latency ----
    constexpr int TPW = (kRgbTiles + 3) / 4;                  // 3 row tiles per wave (the last wave has 2)
    h8 a[TPW][KS];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int tile = wave + 4 * i;
        const int pos = tile * 32 + m;
        const int pr = pos / kRgbPW, pc = pos - pr * kRgbPW;
        const int h = h0 - 1 + pr, w = w0 - 1 + pc;
        const bool ok = tile < kRgbTiles && pos < kRgbPos && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
        const unsigned off = ok ? (unsigned)((((n * p.H + h) * p.W + w) * p.ldx + 8 * kg) * 2) : kOOB;
#pragma unroll
        for (int j = 0; j < KS; ++j)
            a[i][j] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? off + 32u * j : kOOB, 0, 0));
    }
    // ---- B fragments (the 27 live (tap, co) columns of 32) and this lane's slice of the per-sample affine, as halfs ----
    h8 bf[KS], sc[KS], sh[KS];
    const float* scn = p.scale + (size_t)n * p.ld_aff;
    const float* shn = p.shift + (size_t)n * p.ld_aff;
#pragma unroll
    for (int j = 0; j < KS; ++j) {
        bf[j] = *reinterpret_cast<const h8*>(p.w + ((size_t)(j * 2 + kg) * 32 + m) * 8);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(scn + 16 * j + 8 * kg), s1 = *reinterpret_cast<const f32x4*>(scn + 16 * j + 8 * kg + 4);
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(shn + 16 * j + 8 * kg), t1 = *reinterpret_cast<const f32x4*>(shn + 16 * j + 8 * kg + 4);
        sc[j] = h8{(_Float16)s0[0], (_Float16)s0[1], (_Float16)s0[2], (_Float16)s0[3], (_Float16)s1[0], (_Float16)s1[1], (_Float16)s1[2], (_Float16)s1[3]};
        sh[j] = h8{(_Float16)t0[0], (_Float16)t0[1], (_Float16)t0[2], (_Float16)t0[3], (_Float16)t1[0], (_Float16)t1[1], (_Float16)t1[2], (_Float16)t1[3]};
    }
    const h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int tile = wave + 4 * i;
        if (tile >= kRgbTiles) break;                         // (wave-uniform)
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            // BN + ReLU on the fragment: one rounding per element (fused multiply-add in fp16), like the half the producing
            // conv's epilogue used to store
            h8 v = __builtin_elementwise_fma(a[i][j], sc[j], sh[j]);
            v = __builtin_elementwise_max(v, zero);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(v, bf[j], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) Z[(tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg) * kRgbZld + m] = acc[r];
    }
    __syncthreads();
    // ---- one output position per thread: the nine shifted partial sums ----
    const int r_ = tid >> 5, c_ = tid & 31;
    const int h = h0 + r_, w = w0 + c_;
    if (h >= p.H || w >= p.W) return;
    float s0 = p.bias ? p.bias[0] : 0.f, s1 = p.bias ? p.bias[1] : 0.f, s2 = p.bias ? p.bias[2] : 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const bool hok = (unsigned)(h + kh - 1) < (unsigned)p.H;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            if (hok && (unsigned)(w + kw - 1) < (unsigned)p.W) {       // padding taps: the activated map is zero there
                const float* z = Z + ((r_ + kh) * kRgbPW + c_ + kw) * kRgbZld + (kh * 3 + kw) * 3;
                s0 += z[0];
                s1 += z[1];
                s2 += z[2];
            }
        }
    }
    if (p.flags & PTX_EPI_TANH) { s0 = tanhf(s0); s1 = tanhf(s1); s2 = tanhf(s2); }
    float* yo = p.y + ((size_t)(n * p.H + h) * p.W + w) * p.ldy;
    if (p.ldy >= 4) *reinterpret_cast<f32x4*>(yo) = f32x4{s0, s1, s2, 0.f};
    else { yo[0] = s0; yo[1] = s1; yo[2] = s2; }
}

// w [3][C][3][3] fp32 -> [C/16][2][32][8] halfs: the B fragment lane (column j = lane % 32, k group kg = lane / 32) of
// k-step js reads is the 16 bytes at ((js * 2 + kg) * 32 + j) * 8;  column j = (kh * 3 + kw) * 3 + co, columns 27..31 zero
__global__ void __launch_bounds__(256) pack_rgb_conv_kernel(const float* __restrict__ w, _Float16* __restrict__ out, int C) {
    const int total = C * 32;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int e = i & 7, j = (i >> 3) & 31, kg = (i >> 8) & 1, js = i >> 9;
        const int c = js * 16 + kg * 8 + e;
        float v = 0.f;
        if (j < 27) {
            const int co = j % 3, tap = j / 3;
            v = w[((size_t)co * C + c) * 9 + tap];
        }
        out[i] = (_Float16)v;
    }
}


// ---- the 3x3 convs of a GBlock (C -> C channels, C = 64 / 128; [nearest 2x] -> conv -> + bias -> cBN affine -> ReLU -> halfs) -------
// On the generic tiles a 3x3 conv over 64 channels is 18 k-steps of 2 (f16) MFMAs per wave, each behind a barrier and its own
// round of staging address math: 0.53 ms at 256 x 256 x 64 images where HBM needs 0.18 and the matrix cores 0.12-0.19.  Here a
// workgroup (4 waves) owns 8 x 32 output positions x ALL output channels and stages the halo'd INPUT PATCH once -- 10 x 34
// positions x 64 channels = 42.5 KB by LDS-DMA, 16-byte slots XOR-swizzled by position so the fragment reads are conflict
// free; with PTX_PRO_UP2 every patch position is fetched from its source pixel (h / 2, w / 2): the upsampled map never exists
// -- and serves all nine taps from it.  Only the filter streams: 8 KiB tiles ([64][64 halfs] per tap at C = 64, [128][32 halfs]
// per half tap at C >= 128 -- 16 v_mfma_f32_32x32x16_f16 per wave and step either way) through a four-slot ring, requested three
// steps ahead and waited for with a COUNTED vmcnt and a bare s_barrier; the fragments of step s + 1 are read from LDS while
// step s multiplies, group by group into the registers the MFMAs have just consumed.  The product is computed TRANSPOSED
// (filter rows as the A operand): a lane then owns 4 consecutive output channels of one position, so the epilogue -- affine
// tables from LDS, ReLU, 4 halfs per ds_write_b64 into a position-major tile, 16-byte row-major copy-out -- stores whole
// 128 / 256-byte pixels.  76.5 / 77 KB of LDS: two workgroups per CU cover each other's patch load and epilogue.
struct C3Args {
    const _Float16* x;      // [N][Hs][Ws][ldx] halfs (Hs = H / 2 under PTX_PRO_UP2)
    const _Float16* w;      // ptx_pack_conv_weight(f16 = 1): [9][Co_pad][Kc] halfs
    const float* bias;      // [C] or NULL
    const float* scale;     // [N][ld_aff] or NULL (no affine)
    const float* shift;
    _Float16* y;            // [N][H][W][ldy] halfs
    int N, H, W, Hs, Ws, ldx, ldy, ld_aff, Kc, tap_stride, tiles_h, tiles_w;
    unsigned x_bytes, w_bytes, flags;
};

constexpr int kC3TH = 8, kC3TW = 32, kC3PW = kC3TW + 2, kC3Pos = (kC3TH + 2) * kC3PW;      // 340 patch positions
constexpr int kC3PatchPieces = (kC3Pos * 8 + 255) / 256 * 256;                              // 16-byte pieces, padded to whole rounds
constexpr int kC3PatchBytes = kC3PatchPieces * 16;                                          // 45056

// Phase clock of a workgroup (diagnostic build only: scripts/micro/build_timeline.sh compiles this file with -DPTX_C3_TIMELINE
// into a SEPARATE library; the product library carries none of it).  Thread 0 writes the 100 MHz wall clock at: 0 entry, 1 requests
// issued, 2 patch + first filter tile landed, 3 last MFMA issued, 4 output tile parked in LDS, 5 stores retired; 6 = __smid().
#ifdef PTX_C3_TIMELINE
__device__ unsigned long long* g_c3_tl = nullptr;
#define PTX_C3_TL(k)                                                                                                   \
    do {                                                                                                               \
        if (threadIdx.x == 0 && g_c3_tl) g_c3_tl[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (k)] = (k) == 6 ? (unsigned long long)__smid() : (unsigned long long)wall_clock64(); \
    } while (0)
#else
#define PTX_C3_TL(k) do {} while (0)
#endif

// NCH: 64-channel input chunks (Ci = 64 NCH); CT: 32-channel output tiles of ONE workgroup (Co_wg = 32 CT: 64 or 128 --
// wider outputs are cut along blockIdx.y, each part re-staging the patch: C = 256 runs as <4, 4> x 2)
template <int NCH, int CT, bool UP2>
__global__ void __launch_bounds__(256, 2) conv3x3_f16_kernel(const C3Args p) {
    constexpr int C = 32 * CT;                     // output channels of this workgroup
    constexpr int KH = CT == 2 ? 64 : 32;          // input channels of one filter tile = one step (16 MFMAs per wave either way)
    constexpr int SPT = 64 / KH;                   // steps per (tap, 64-channel chunk)
    constexpr int NJ = KH / 16;                    // MFMA k-groups per step
    constexpr int NBUF = 4;                        // filter ring: tile s + 1 is read while s multiplies and s + 2, s + 3 fly
    constexpr int BT_BYTES = C * KH * 2;           // one filter tile: C rows x KH halfs = 8 KiB
    constexpr int BT_IT = BT_BYTES / 16 / 256;     // its 16-byte pieces per thread (2)
    constexpr int RS = KH / 8;                     // 16-byte slots per filter row (8 or 4)
    constexpr int PA_IT = kC3PatchPieces / 256;    // patch pieces per thread (11)
    constexpr int TP = C + 8;                      // pitch (halfs) of the output tile in LDS: conflict-free ds_write_b64
    constexpr int SPC = 9 * SPT;                   // steps per input chunk
    constexpr int NS = SPC * NCH;
    static_assert(BT_BYTES == 8192 && BT_IT == 2, "one step = an 8 KiB filter tile");
    constexpr unsigned kOOB = 0x80000000u;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    PTX_C3_TL(0);
    PTX_C3_TL(6);
    char* Pa = smem;                               // the patch chunk, later the output tile
    char* Bt = smem + kC3PatchBytes;               // [NBUF][BT_BYTES]
    float* Sc = reinterpret_cast<float*>(smem + kC3PatchBytes + NBUF * BT_BYTES);     // [C] scale, [C] shift' = bias * scale + shift
    float* Sh = Sc + C;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, kg = lane >> 5;
    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int tw = t % p.tiles_w;
    t /= p.tiles_w;
    const int th = t % p.tiles_h, n = t / p.tiles_h;
    const int h0 = th * kC3TH, w0 = tw * kC3TW;
    const int co_base = blockIdx.y * C;            // first output channel of this workgroup
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.w), 0, p.w_bytes, 0x00020000);

    // affine tables of this sample: requested FIRST (vmcnt retires in order -- asked for after the patch they would make the
    // table write wait for the whole patch), parked in LDS once the prologue's wait has passed
    f32x4 t_sc = {1.f, 1.f, 1.f, 1.f}, t_sf = {0.f, 0.f, 0.f, 0.f}, t_bs = {0.f, 0.f, 0.f, 0.f};
    if (tid < C / 4) {
        const int c4 = tid * 4;
        if (p.scale) {
            t_sc = *reinterpret_cast<const f32x4*>(p.scale + (size_t)n * p.ld_aff + co_base + c4);
            t_sf = *reinterpret_cast<const f32x4*>(p.shift + (size_t)n * p.ld_aff + co_base + c4);
        }
        if (p.bias) t_bs = *reinterpret_cast<const f32x4*>(p.bias + co_base + c4);
    }

    // ---- DMA sources (byte offsets), fixed for the tile: piece q = wave * 64 + 256 i + lane lands at Pa + 16 q ----
    unsigned pa_src[PA_IT];
#pragma unroll
    for (int i = 0; i < PA_IT; ++i) {
        const int q = tid + 256 * i;
        const int pos = q >> 3, ps = q & 7;
        const int slot = ps ^ ((pos >> 1) & 7);                       // logical 8-channel slot this physical slot holds
        const int pr = pos / kC3PW, pc = pos - pr * kC3PW;
        const int h = h0 - 1 + pr, w = w0 - 1 + pc;
        const bool ok = pos < kC3Pos && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
        const int sh_ = UP2 ? (h >> 1) : h, sw_ = UP2 ? (w >> 1) : w;
        pa_src[i] = ok ? (unsigned)((((n * p.Hs + sh_) * p.Ws + sw_) * p.ldx + slot * 8) * 2) : kOOB;
    }
    // filter tile: piece q -> row q / RS, physical slot q % RS holding logical slot (q % RS) ^ swz(row).  128-byte rows (KH = 64)
    // swizzle by (row >> 1) & 7, 64-byte rows (KH = 32) by (row >> 2) & 3: either way the 16 lanes ds_read_b128 serves per cycle
    // (rows l32 of one logical slot) cover the sixteen 16-byte columns of the 256-byte bank line once
    unsigned bt_src[2];      // (a literal extent: a template-dependent one indexed inside the voffset operand of the LDS-DMA builtin
                             //  makes hipcc's host pass drop the kernel's stub -- DESIGN.md 3.8)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = tid + 256 * i;
        const int co = q / RS, ps = q % RS;
        const int swz = RS == 8 ? (co >> 1) & 7 : (co >> 2) & 3;
        bt_src[i] = (unsigned)(((co_base + co) * p.Kc + (ps ^ swz) * 8) * 2);
    }
    auto issue_patch = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < PA_IT; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(Pa + (wave * 64 + 256 * i) * 16), 16,
                                                     pa_src[i] == kOOB ? kOOB : pa_src[i] + (unsigned)chunk * 128u, 0, 0, 0);
    };
    // step s = (chunk * 9 + tap) * SPT + half: the filter's rows [tap][co][chunk * 64 + half * KH ...]
    auto issue_filter = [&](int chunk, int r) {                       // r = tap * SPT + half, the step inside the chunk
        const int tap = r / SPT, hf = r - tap * SPT, buf = (chunk * SPC + r) % NBUF;
        const unsigned base = (unsigned)(tap * p.tap_stride) * 2u + (unsigned)(chunk * 128 + hf * KH * 2);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(Bt + buf * BT_BYTES + (wave * 64 + 256 * i) * 16), 16,
                                                     bt_src[i] + base, 0, 0, 0);
    };
    issue_patch(0);
    issue_filter(0, 0);
    issue_filter(0, 1);
    issue_filter(0, 2);
    PTX_C3_TL(1);

    f32x16 acc[CT][2];
#pragma unroll
    for (int a = 0; a < CT; ++a)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][i][r] = 0.f;
    // fragment addresses: filter row co = 32 a + l32 (step independent), patch position of output (row 2 wave + i, column l32)
    int b_off[CT], b_sw[CT];
#pragma unroll
    for (int a = 0; a < CT; ++a) {
        const int co = 32 * a + l32;
        b_off[a] = co * KH * 2;
        b_sw[a] = RS == 8 ? (co >> 1) & 7 : (co >> 2) & 3;
    }
    int p_base = (2 * wave) * kC3PW + l32;

    // The fragments of step s + 1 are requested while step s multiplies: group j's registers are refilled right after group
    // j's MFMAs have issued, so inside an input chunk no MFMA waits on an LDS read issued behind the same barrier (measured with
    // the phase clock, scripts/gpu_c3_timeline.py: read-then-multiply per tap took 2.4x the tap's MFMA time).
    h8 xa[NJ][2], wb[NJ][CT];
    auto load_group = [&](int j, int r, const char* Bb) {            // fragments of k-group j of step r (inside its chunk)
        const int tap = r / SPT, hf = r - tap * SPT;
        const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pos = p_base + (i + kh) * kC3PW + kw;
            xa[j][i] = *reinterpret_cast<const h8*>(Pa + pos * 128 + (((hf * RS + 2 * j + kg) ^ ((pos >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int a = 0; a < CT; ++a) wb[j][a] = *reinterpret_cast<const h8*>(Bb + b_off[a] + (((2 * j + kg) ^ b_sw[a]) << 4));
    };
    auto mma_group = [&](int j) {
#pragma unroll
        for (int a = 0; a < CT; ++a)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[a][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[j][a], xa[j][i], acc[a][i], 0, 0, 0);   // C^T[co][position]
    };

    // prologue: everything but filter tile 2 has landed -> tables to LDS, fragments of step 0
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BT_IT) : "memory");
    if (tid < C / 4) {
        const int c4 = tid * 4;
        *reinterpret_cast<f32x4*>(Sc + c4) = t_sc;
        *reinterpret_cast<f32x4*>(Sh + c4) = f32x4{t_bs[0] * t_sc[0] + t_sf[0], t_bs[1] * t_sc[1] + t_sf[1], t_bs[2] * t_sc[2] + t_sf[2],
                                                   t_bs[3] * t_sc[3] + t_sf[3]};
    }
    __syncthreads();
    PTX_C3_TL(2);
    asm volatile("; LDS reads stay below the barrier" : "+v"(p_base)::"memory");
#pragma unroll
    for (int j = 0; j < NJ; ++j) load_group(j, 0, Bt);

#pragma unroll 1
    for (int chunk = 0; chunk < NCH; ++chunk) {
        const int s0 = chunk * SPC;                // (SPC % NBUF != 0: ring slots are taken from the running step index)
#pragma unroll
        for (int r = 0; r < SPC; ++r) {
            const int s = s0 + r;
            if (r > 0) {
                // filter tile s + 1 has landed for everyone (s + 2 may still fly); every wave is done with slot (s - 1) % NBUF.
                // A bare s_barrier: __syncthreads() fences with vmcnt(0) lgkmcnt(0), which would wait for tile s + 2 and for the
                // fragments just requested.  Nothing is WRITTEN to LDS by a wave inside the loop (the DMA is counted by vmcnt).
                if (s + 2 < NS) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(BT_IT) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            }
            if (r + 3 < SPC) issue_filter(chunk, r + 3);            // (issued between MFMA groups instead: same time, measured)
            else if (chunk + 1 < NCH) issue_filter(chunk + 1, r + 3 - SPC);
            const bool prefetch = r + 1 < SPC;     // (the next chunk's first fragments wait for its patch, below)
            const char* Bn = Bt + ((s + 1) % NBUF) * BT_BYTES;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                mma_group(j);
                __builtin_amdgcn_sched_barrier(0);
                if (prefetch) load_group(j, r + 1, Bn);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (chunk + 1 < NCH) {
            __syncthreads();                       // every wave is done with this patch chunk
            issue_patch(chunk + 1);                // exposed; the CU's other workgroup covers it
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            asm volatile("; LDS reads stay below the barrier" : "+v"(p_base)::"memory");
            const char* Bn = Bt + ((s0 + SPC) % NBUF) * BT_BYTES;
#pragma unroll
            for (int j = 0; j < NJ; ++j) load_group(j, 0, Bn);
        }
    }

    // ---- epilogue: lane = position (2 wave + i, l32); element r of acc[a][i] = channel 32 a + 8 (r >> 2) + 4 kg + (r & 3) ----
    PTX_C3_TL(3);
    __syncthreads();                               // all fragment reads done: the patch area becomes the output tile
    const bool relu = (p.flags & PTX_EPI_RELU) != 0;
    _Float16* T = reinterpret_cast<_Float16*>(Pa);
#pragma unroll
    for (int a = 0; a < CT; ++a)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = 32 * a + 8 * g + 4 * kg;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(Sc + co), sf = *reinterpret_cast<const f32x4*>(Sh + co);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                h4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[a][i][4 * g + e] * sc[e] + sf[e];
                    v = relu ? fmaxf(v, 0.f) : v;
                    o[e] = (_Float16)v;
                }
                *reinterpret_cast<h4*>(T + ((2 * wave + i) * 32 + l32) * TP + co) = o;
            }
        }
    __syncthreads();
    PTX_C3_TL(4);
    constexpr int SL = C / 8;                      // 16-byte pieces per pixel
    const size_t y_img = (size_t)n * p.H * p.W * p.ldy;      // element offset of this image
#pragma unroll
    for (int it = 0; it < SL; ++it) {
        const int q = tid + 256 * it;
        const int pos = q / SL, sl = q - pos * SL;
        const int h = h0 + (pos >> 5), w = w0 + (pos & 31);
        if (h < p.H && w < p.W) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(T + pos * TP + sl * 8);
            *reinterpret_cast<f32x4*>(p.y + y_img + ((size_t)h * p.W + w) * p.ldy + co_base + sl * 8) = v;
        }
    }
#ifdef PTX_C3_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PTX_C3_TL(5);
#endif
}


// ---- the closing 1x1 conv of a GBlock (conv4: C/4 -> C' channels, + skip, two outputs) ----------------------------------------
//     v     = conv1x1(t) + bias + up?(x_block[:, :C'])          (the block's output; x_block halfs, nearest 2x when the block upsamples)
//     y_raw = half(v)                                            (the next block's skip operand)              [PTX_EPI_DUAL_RAW]
//     y     = half(relu(v * scale[n] + shift[n]))                (the next block's cBN1 + ReLU, folded)       [PTX_EPI_AFFINE]
// K is 64 ... 256 channels: one to four k-steps per output tile, after which the generic tile spends its time in a 16-row-at-a-
// time epilogue -- 0.69 ms at 256 x 256 x 128 where the bytes (0.54 GB in + 0.27 GB skip + 1.07 GB out) need 0.34.  Same skeleton
// as the 3x3 kernel with one tap and no halo: 8 x 32 positions x 128 output channels per workgroup (wider outputs over
// blockIdx.y), the input tile and the [128][64] filter tile by LDS-DMA, transposed product.  The epilogue keeps fp32 until the
// single rounding: half of the tile at a time is parked as floats (128 positions x 132), then every thread handles 16-byte
// pieces of whole pixels -- skip read, raw store, affine + ReLU, activated store, all coalesced.
struct C1Args {
    const _Float16* x;      // [N][H][W][ldx] halfs
    const _Float16* w;      // [Co_pad][Kc] halfs (ptx_pack_conv_weight, f16 = 1)
    const float* bias;
    const _Float16* res;    // skip operand [N][rH][rW][ldr] halfs, read at (h >> ush, w >> ush), or NULL
    const float* scale;     // [N][ld_aff] or NULL
    const float* shift;
    _Float16* y;            // main output [N][H][W][ldy]
    _Float16* y_raw;        // PTX_EPI_DUAL_RAW: pre-affine output [N][H][W][ld_raw]
    int N, H, W, ldx, ldy, ld_raw, ld_aff, Kc, rH, rW, ldr, ush, tiles_h, tiles_w;
    unsigned x_bytes, w_bytes, r_bytes, y_bytes, raw_bytes, flags;
};

constexpr int kC1TP = 132;                                      // pitch (floats) of the parked half tile
constexpr int kC1Lds = 128 * kC1TP * 4 + 3 * 128 * 4;           // parked tile (overlays the operand tiles) + bias / scale / shift

template <int NCH>
__global__ void __launch_bounds__(256, 2) conv1x1_skip_f16_kernel(const C1Args p) {
    constexpr int CO = 128, CT = 4;
    constexpr int A_BYTES = 256 * 128, BT_BYTES = CO * 128;
    static_assert(A_BYTES + 2 * BT_BYTES <= 128 * kC1TP * 4, "operand tiles under the parked tile");
    constexpr unsigned kOOB = 0x80000000u;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Pa = smem;                               // [256 positions][64 halfs], swizzled
    char* Bt = smem + A_BYTES;                     // [2][128][64 halfs]
    float* Tb = reinterpret_cast<float*>(smem + 128 * kC1TP * 4);     // bias | scale | shift of this workgroup's 128 channels

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, kg = lane >> 5;
    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int tw = t % p.tiles_w;
    t /= p.tiles_w;
    const int th = t % p.tiles_h, n = t / p.tiles_h;
    const int h0 = th * kC3TH, w0 = tw * kC3TW;
    const int co_base = blockIdx.y * CO;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.w), 0, p.w_bytes, 0x00020000);

    unsigned pa_src[8], bt_src[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = tid + 256 * i;
        const int pos = q >> 3, ps = q & 7;
        const int h = h0 + (pos >> 5), w = w0 + (pos & 31);
        const bool ok = h < p.H && w < p.W;
        pa_src[i] = ok ? (unsigned)((((n * p.H + h) * p.W + w) * p.ldx + (ps ^ ((pos >> 1) & 7)) * 8) * 2) : kOOB;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = tid + 256 * i;
        const int co = q >> 3, ps = q & 7;
        bt_src[i] = (unsigned)(((co_base + co) * p.Kc + (ps ^ ((co >> 1) & 7)) * 8) * 2);
    }
    auto issue_a = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(Pa + (wave * 64 + 256 * i) * 16), 16,
                                                     pa_src[i] == kOOB ? kOOB : pa_src[i] + (unsigned)chunk * 128u, 0, 0, 0);
    };
    auto issue_b = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(Bt + (chunk & 1) * BT_BYTES + (wave * 64 + 256 * i) * 16), 16,
                                                     bt_src[i] + (unsigned)chunk * 128u, 0, 0, 0);
    };
    issue_a(0);
    issue_b(0);
    // the epilogue's per-thread pieces: 16 bytes = 8 channels of one pixel; half hp of the tile holds pieces (pl, sl) = (q >> 4, q & 15),
    // q = tid + 256 it.  Their pixel offsets are fixed for the tile, and the SKIP operand of half 0 is requested now -- its round
    // trip hides under the operand DMA and the MFMAs instead of sitting in the epilogue (a load inside the copy-out loop was
    // ~1 us of exposed latency per piece: the first version of this kernel ran at 3.2 TB/s)
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.res ? p.res : p.x), 0, p.res ? p.r_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc(p.y_raw ? p.y_raw : p.y, 0, p.y_raw ? p.raw_bytes : 0u, 0x00020000);
    auto skip_off = [&](int hp, int it) -> unsigned {
        const int q = tid + 256 * it;
        const int pl = q >> 4, sl = q & 15;
        const int h = h0 + 4 * hp + (pl >> 5), w = w0 + (pl & 31);
        return (h < p.H && w < p.W) ? (unsigned)((((n * p.rH + (h >> p.ush)) * p.rW + (w >> p.ush)) * p.ldr + co_base + sl * 8) * 2) : kOOB;
    };
    f32x4 rq0[8], rq1[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) rq0[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, skip_off(0, it), 0, 0);
    if (tid < 3 * CO / 4) {                        // bias | scale | shift -> LDS
        const int which = tid / (CO / 4), c4 = (tid - which * (CO / 4)) * 4;
        f32x4 v = which == 1 ? f32x4{1.f, 1.f, 1.f, 1.f} : f32x4{0.f, 0.f, 0.f, 0.f};
        if (which == 0 && p.bias) v = *reinterpret_cast<const f32x4*>(p.bias + co_base + c4);
        if (which == 1 && p.scale) v = *reinterpret_cast<const f32x4*>(p.scale + (size_t)n * p.ld_aff + co_base + c4);
        if (which == 2 && p.shift) v = *reinterpret_cast<const f32x4*>(p.shift + (size_t)n * p.ld_aff + co_base + c4);
        *reinterpret_cast<f32x4*>(Tb + which * CO + c4) = v;
    }
    f32x16 acc[CT][2];
#pragma unroll
    for (int a = 0; a < CT; ++a)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][i][r] = 0.f;
    int b_off[CT], b_sw[CT], a_off[2], a_sw[2];
#pragma unroll
    for (int a = 0; a < CT; ++a) {
        const int co = 32 * a + l32;
        b_off[a] = co * 128;
        b_sw[a] = (co >> 1) & 7;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pos = (2 * wave + i) * 32 + l32;
        a_off[i] = pos * 128;
        a_sw[i] = (pos >> 1) & 7;
    }
#pragma unroll 1
    for (int s = 0; s < NCH; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                           // input chunk s and filter tile s landed
        if (s + 1 < NCH) issue_b(s + 1);
        const char* Bb = Bt + (s & 1) * BT_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int slot = 2 * j + kg;
            h8 xa[2], wb[CT];
#pragma unroll
            for (int i = 0; i < 2; ++i) xa[i] = *reinterpret_cast<const h8*>(Pa + a_off[i] + ((slot ^ a_sw[i]) << 4));
#pragma unroll
            for (int a = 0; a < CT; ++a) wb[a] = *reinterpret_cast<const h8*>(Bb + b_off[a] + ((slot ^ b_sw[a]) << 4));
#pragma unroll
            for (int a = 0; a < CT; ++a)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[a][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[a], xa[i], acc[a][i], 0, 0, 0);
        }
        if (s + 1 < NCH) {
            __syncthreads();                       // every wave is done with this input chunk
            issue_a(s + 1);
        }
    }
    // ---- epilogue, half a tile (4 output rows = two waves' accumulators) at a time ----
    const bool affine = (p.flags & PTX_EPI_AFFINE) != 0, dual = (p.flags & PTX_EPI_DUAL_RAW) != 0, relu = (p.flags & PTX_EPI_RELU) != 0;
    float* T = reinterpret_cast<float*>(smem);
    auto park = [&](int hp) {
        if ((wave >> 1) == hp) {
#pragma unroll
            for (int a = 0; a < CT; ++a)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int pl = ((2 * wave + i) - 4 * hp) * 32 + l32;
                        *reinterpret_cast<f32x4*>(T + pl * kC1TP + 32 * a + 8 * g + 4 * kg) =
                            f32x4{acc[a][i][4 * g], acc[a][i][4 * g + 1], acc[a][i][4 * g + 2], acc[a][i][4 * g + 3]};
                    }
        }
    };
    auto copy_out = [&](int hp, const f32x4 (&rq)[8]) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int q = tid + 256 * it;
            const int pl = q >> 4, sl = q & 15;
            const int h = h0 + 4 * hp + (pl >> 5), w = w0 + (pl & 31);
            const bool ok = h < p.H && w < p.W;
            const unsigned px = (unsigned)((n * p.H + h) * p.W + w);
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(T + pl * kC1TP + sl * 8), v1 = *reinterpret_cast<const f32x4*>(T + pl * kC1TP + sl * 8 + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(Tb + sl * 8), b1 = *reinterpret_cast<const f32x4*>(Tb + sl * 8 + 4);
            const h8 r = __builtin_bit_cast(h8, rq[it]);           // zeros without a skip operand (empty descriptor)
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (e < 4 ? v0[e] + b0[e] : v1[e - 4] + b1[e - 4]) + (float)r[e];
            if (dual || !affine) {
                h8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (_Float16)((!affine && relu) ? fmaxf(v[e], 0.f) : v[e]);
                if (dual) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(f32x4, o), rs_q, ok ? (px * (unsigned)p.ld_raw + co_base + sl * 8) * 2u : kOOB, 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(f32x4, o), rs_y, ok ? (px * (unsigned)p.ldy + co_base + sl * 8) * 2u : kOOB, 0, 0);
            }
            if (affine) {
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(Tb + CO + sl * 8), s1 = *reinterpret_cast<const f32x4*>(Tb + CO + sl * 8 + 4);
                const f32x4 t0 = *reinterpret_cast<const f32x4*>(Tb + 2 * CO + sl * 8), t1 = *reinterpret_cast<const f32x4*>(Tb + 2 * CO + sl * 8 + 4);
                h8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float u = v[e] * (e < 4 ? s0[e] : s1[e - 4]) + (e < 4 ? t0[e] : t1[e - 4]);
                    o[e] = (_Float16)(relu ? fmaxf(u, 0.f) : u);
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(f32x4, o), rs_y, ok ? (px * (unsigned)p.ldy + co_base + sl * 8) * 2u : kOOB, 0, 0);
            }
        }
    };
    __syncthreads();                               // the operand tiles are no longer read
    park(0);
#pragma unroll
    for (int it = 0; it < 8; ++it) rq1[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, skip_off(1, it), 0, 0);     // half 1's skip: under half 0's copy-out
    __syncthreads();
    copy_out(0, rq0);
    __syncthreads();                               // half 0 has been read
    park(1);
    __syncthreads();
    copy_out(1, rq1);
}


// ---- the opening 1x1 conv of a GBlock (conv1: C -> C/4) with cBN1 + ReLU in its loader ------------------------------------------
//     y = half(relu?(scale2[n] * (W . relu(x * scale1[n] + shift1[n]) + bias) + shift2[n]))
// x is the RAW half output of the previous block, so that block's closing conv stores ONE tensor instead of two (VERDICT r3
// #5b: the activated copy was a third of conv4's bytes).  The activation is the big operand here (K = 256 ... 2048 channels per
// position) and every byte of it is used once: it never touches LDS.  A lane loads its B-operand fragments (8 channels of one
// position) straight from global memory, one 64-channel chunk ahead, applies the per-sample affine + ReLU as packed half math
// (v_pk_fma_f16 / v_pk_max_f16; the tables sit in LDS as halfs) and feeds the MFMA; only the filter chunk [Co][64] is shared
// through LDS (DMA'd, double-buffered, one barrier per chunk).  Transposed product and pixel-major epilogue as in the kernels
// above.  NPT = 32-position tiles per wave (2, or 1 when 256 output channels need the accumulator registers).
struct P1Args {
    const _Float16* x;      // [M][ldx] halfs, raw
    const _Float16* w;      // [Co_pad][Kc] halfs
    const float* bias;
    const float* scale1;    // [N][ld1]: the block's cBN1, applied to the INPUT
    const float* shift1;
    const float* scale2;    // [N][ld2] or NULL: the affine that follows the conv (cBN2)
    const float* shift2;
    _Float16* y;            // [M][ldy] halfs
    int M, HW, K, ldx, ldy, ld1, ld2, Kc, nch;
    unsigned x_bytes, w_bytes, y_bytes, flags;
};

template <int CT, int NPT>
__global__ void __launch_bounds__(256, 2) conv1x1_pro_f16_kernel(const P1Args p) {
    constexpr int CO = 32 * CT, POS = 128 * NPT;              // output channels / positions of one workgroup
    constexpr int BT_BYTES = CO * 128, BT_IT = CO * 8 / 256;  // filter chunk [CO][64 halfs]; its 16-byte pieces per thread
    constexpr int TP = CO + 8;                                // pitch (halfs) of the output tile
    constexpr int REGION = (POS * TP * 2 > 2 * BT_BYTES) ? POS * TP * 2 : 2 * BT_BYTES;
    constexpr unsigned kOOB = 0x80000000u;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Bt = smem;                                          // [2][BT_BYTES]; later the output tile
    _Float16* T1 = reinterpret_cast<_Float16*>(smem + REGION);            // [K] scale1 | [K] shift1, halfs
    float* T2 = reinterpret_cast<float*>(smem + REGION + 4 * p.K);        // [CO] scale2 | [CO] bias * scale2 + shift2

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, kg = lane >> 5;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = tile * POS;
    const int n = m0 / p.HW;                                  // one sample per workgroup (HW % POS == 0)
    const int co_base = blockIdx.y * CO;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.w), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);

    unsigned xo[2];                                          // (sized by its maximum, NPT <= 2)
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int m = m0 + (wave * NPT + i) * 32 + l32;
        xo[i] = m < p.M ? (unsigned)((m * p.ldx + kg * 8) * 2) : kOOB;
    }
    static_assert(BT_IT <= 4 || CT == 8, "filter pieces");
    unsigned bt_src8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i >= BT_IT) break;
        const int q = tid + 256 * i;
        const int co = q >> 3, ps = q & 7;
        bt_src8[i] = (unsigned)(((co_base + co) * p.Kc + (ps ^ ((co >> 1) & 7)) * 8) * 2);
    }
    auto issue_b = [&](int c) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < BT_IT)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(Bt + (c & 1) * BT_BYTES + (wave * 64 + 256 * i) * 16), 16,
                                                         bt_src8[i] + (unsigned)c * 128u, 0, 0, 0);
    };
    // ---- tables -> LDS: the input affine as halfs (K of them), the output affine with the bias folded in ----
    for (int k4 = tid * 4; k4 < p.K; k4 += 1024) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const f32x4 s = *reinterpret_cast<const f32x4*>(p.scale1 + (size_t)n * p.ld1 + k4), t = *reinterpret_cast<const f32x4*>(p.shift1 + (size_t)n * p.ld1 + k4);
        *reinterpret_cast<h4*>(T1 + k4) = h4{(_Float16)s[0], (_Float16)s[1], (_Float16)s[2], (_Float16)s[3]};
        *reinterpret_cast<h4*>(T1 + p.K + k4) = h4{(_Float16)t[0], (_Float16)t[1], (_Float16)t[2], (_Float16)t[3]};
    }
    if (tid < CO / 4) {
        const int c4 = tid * 4;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sf = {0.f, 0.f, 0.f, 0.f}, bs = {0.f, 0.f, 0.f, 0.f};
        if (p.scale2) {
            sc = *reinterpret_cast<const f32x4*>(p.scale2 + (size_t)n * p.ld2 + co_base + c4);
            sf = *reinterpret_cast<const f32x4*>(p.shift2 + (size_t)n * p.ld2 + co_base + c4);
        }
        if (p.bias) bs = *reinterpret_cast<const f32x4*>(p.bias + co_base + c4);
        *reinterpret_cast<f32x4*>(T2 + c4) = sc;
        *reinterpret_cast<f32x4*>(T2 + CO + c4) = f32x4{bs[0] * sc[0] + sf[0], bs[1] * sc[1] + sf[1], bs[2] * sc[2] + sf[2], bs[3] * sc[3] + sf[3]};
    }
    issue_b(0);
    h8 xa0[NPT][4], xa1[NPT][4];                             // the activation fragments of chunk c (even) / c + 1 (odd)
    auto load_x = [&](h8 (&xa)[NPT][4], int c) {
#pragma unroll
        for (int i = 0; i < NPT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                xa[i][j] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rs_x, xo[i] == kOOB ? kOOB : xo[i] + (unsigned)(c * 128 + j * 32), 0, 0));
    };
    load_x(xa0, 0);
    f32x16 acc[CT][NPT];
#pragma unroll
    for (int a = 0; a < CT; ++a)
#pragma unroll
        for (int i = 0; i < NPT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][i][r] = 0.f;
    int b_off[CT], b_sw[CT];
#pragma unroll
    for (int a = 0; a < CT; ++a) {
        const int co = 32 * a + l32;
        b_off[a] = co * 128;
        b_sw[a] = (co >> 1) & 7;
    }
    const h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    auto compute = [&](const h8 (&xa)[NPT][4], int c) {
        const char* Bb = Bt + (c & 1) * BT_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int slot = 2 * j + kg;
            const h8 sc = *reinterpret_cast<const h8*>(T1 + c * 64 + slot * 8), sh = *reinterpret_cast<const h8*>(T1 + p.K + c * 64 + slot * 8);
            h8 v[NPT], wb[CT];
#pragma unroll
            for (int i = 0; i < NPT; ++i) v[i] = __builtin_elementwise_max(__builtin_elementwise_fma(xa[i][j], sc, sh), zero);
#pragma unroll
            for (int a = 0; a < CT; ++a) wb[a] = *reinterpret_cast<const h8*>(Bb + b_off[a] + ((slot ^ b_sw[a]) << 4));
#pragma unroll
            for (int a = 0; a < CT; ++a)
#pragma unroll
                for (int i = 0; i < NPT; ++i) acc[a][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[a], v[i], acc[a][i], 0, 0, 0);
        }
    };
#pragma unroll 1
    for (int c = 0; c < p.nch; c += 2) {                      // (nch is even)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                      // filter chunk c landed, slot (c + 1) & 1 is free, tables visible
        issue_b(c + 1);
        load_x(xa1, c + 1);
        compute(xa0, c);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (c + 2 < p.nch) {
            issue_b(c + 2);
            load_x(xa0, c + 2);
        }
        compute(xa1, c + 1);
    }
    // ---- epilogue ----
    __syncthreads();                                          // the filter ring becomes the output tile
    const bool relu = (p.flags & PTX_EPI_RELU) != 0;
    _Float16* T = reinterpret_cast<_Float16*>(smem);
#pragma unroll
    for (int a = 0; a < CT; ++a)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = 32 * a + 8 * g + 4 * kg;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(T2 + co), sf = *reinterpret_cast<const f32x4*>(T2 + CO + co);
#pragma unroll
            for (int i = 0; i < NPT; ++i) {
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                h4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[a][i][4 * g + e] * sc[e] + sf[e];
                    v = relu ? fmaxf(v, 0.f) : v;
                    o[e] = (_Float16)v;
                }
                *reinterpret_cast<h4*>(T + ((wave * NPT + i) * 32 + l32) * TP + co) = o;
            }
        }
    __syncthreads();
    constexpr int SL = CO / 8;
#pragma unroll
    for (int it = 0; it < POS * SL / 256; ++it) {
        const int q = tid + 256 * it;
        const int pl = q / SL, sl = q - pl * SL;
        const int m = m0 + pl;
        const f32x4 v = *reinterpret_cast<const f32x4*>(T + pl * TP + sl * 8);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs_y, m < p.M ? (unsigned)((m * p.ldy + co_base + sl * 8) * 2) : kOOB, 0, 0);
    }
}

}  // namespace ptx

using namespace ptx;

extern "C" int ptx_rgb_conv3x3_f16_supported(const ptx_rgb_conv_desc* d) {
    if (!d) return 0;
    if (d->N <= 0 || d->H <= 0 || d->W <= 0) return 0;
    if (d->C != 32 && d->C != 64 && d->C != 128) return 0;                        // compiled K extents (ch of BigGAN-deep: 128)
    if (d->ldx < d->C || d->ldx % 8 || d->ld_affine < d->C || d->ld_affine % 4 || d->ldy < 3) return 0;
    if (d->ldy >= 4 && d->ldy % 4) return 0;
    if (d->flags & ~PTX_EPI_TANH) return 0;
    if ((uint64_t)d->N * d->H * d->W * d->ldx * 2ull >= 0x80000000ull) return 0;  // 32-bit buffer offsets
    const int64_t tiles = (int64_t)d->N * cdiv(d->H, kRgbTH) * cdiv(d->W, kRgbTW);
    return tiles <= 0x7fffffffLL;
}

extern "C" size_t ptx_rgb_conv_weight_elems(int32_t C) { return C > 0 ? (size_t)C * 32 : 0; }

extern "C" int ptx_pack_rgb_conv_weight(const float* w, int32_t C, void* w_packed, ptx_stream_t stream) {
    if (!w || !w_packed) return fail(PTX_ERR_INVALID, "pack_rgb_conv: null pointer");
    if (C <= 0 || C % 16) return fail(PTX_ERR_INVALID, "pack_rgb_conv: C must be a positive multiple of 16 (got %d)", C);
    hipLaunchKernelGGL(pack_rgb_conv_kernel, dim3((unsigned)cdiv(C * 32, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       static_cast<_Float16*>(w_packed), C);
    return hip_check(hipGetLastError(), "pack_rgb_conv launch");
}

extern "C" int ptx_rgb_conv3x3_f16_fwd(const ptx_rgb_conv_desc* d, const void* x, const float* scale, const float* shift,
                                       const void* w_packed, const float* bias, float* y, ptx_stream_t stream) {
    if (!d || !x || !scale || !shift || !w_packed || !y) return fail(PTX_ERR_INVALID, "rgb_conv3x3_f16: null pointer");
    if (!ptx_rgb_conv3x3_f16_supported(d))
        return fail(PTX_ERR_UNSUPPORTED, "rgb_conv3x3_f16: C in {32, 64, 128}, ldx %% 8 == 0, ld_affine %% 4 == 0, a feature map "
                    "below 2 GiB (C=%d ldx=%d ldy=%d)", d->C, d->ldx, d->ldy);
    if (((uintptr_t)x | (uintptr_t)scale | (uintptr_t)shift | (uintptr_t)w_packed | (uintptr_t)y) & 15)
        return fail(PTX_ERR_INVALID, "rgb_conv3x3_f16: pointers must be 16-byte aligned");
    RgbArgs a{};
    a.x = static_cast<const _Float16*>(x); a.scale = scale; a.shift = shift; a.w = static_cast<const _Float16*>(w_packed);
    a.bias = bias; a.y = y;
    a.N = d->N; a.H = d->H; a.W = d->W; a.C = d->C; a.ldx = d->ldx; a.ld_aff = d->ld_affine; a.ldy = d->ldy;
    a.tiles_h = cdiv(d->H, kRgbTH); a.tiles_w = cdiv(d->W, kRgbTW);
    a.x_bytes = (unsigned)((uint64_t)d->N * d->H * d->W * d->ldx * 2ull);
    a.flags = d->flags;
    const dim3 grid((unsigned)(d->N * a.tiles_h * a.tiles_w));
    const hipStream_t st = (hipStream_t)stream;
    if (d->C == 128) hipLaunchKernelGGL(rgb_conv3x3_f16_kernel<8>, grid, dim3(256), 0, st, a);
    else if (d->C == 64) hipLaunchKernelGGL(rgb_conv3x3_f16_kernel<4>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(rgb_conv3x3_f16_kernel<2>, grid, dim3(256), 0, st, a);
    return hip_check(hipGetLastError(), "rgb_conv3x3_f16 launch");
}

#ifdef PTX_C3_TIMELINE
extern "C" int ptx_c3_timeline(void* buf) {        // diagnostic build only: 8 x u64 per workgroup, or NULL to switch off
    return hip_check(hipMemcpyToSymbol(HIP_SYMBOL(g_c3_tl), &buf, sizeof(buf)), "ptx_c3_timeline");
}
#endif

extern "C" int ptx_conv3x3_f16_supported(const ptx_conv3d_desc* d) {
    if (!d) return 0;
    const unsigned need = PTX_F16_OPERANDS | PTX_EPI_OUT_F16;
    const unsigned may = need | PTX_PRO_UP2 | PTX_EPI_AFFINE | PTX_EPI_RELU;
    if ((d->flags & need) != need || (d->flags & ~may)) return 0;
    const int C = 2 * d->Ci;                                   // fp16 descriptors count 32-bit words (channel pairs)
    if ((C != 64 && C != 128 && C != 256) || d->Co != C || d->groups > 1) return 0;
    if (d->Wi < kC3TW) return 0;                                // 8 x 32 tiles: at 16 columns half of every tile is padding (measured:
                                                                // 0.033 vs 0.028 ms for the generic tile at 16 x 16 x 256)
    if (d->kT != 1 || d->kH != 3 || d->kW != 3 || d->sT != 1 || d->sH != 1 || d->sW != 1 || d->pT != 0 || d->pH != 1 || d->pW != 1) return 0;
    if (d->Ti != 1 || d->To != 1 || d->Ho != d->Hi || d->Wo != d->Wi || d->N <= 0 || d->Hi <= 0 || d->Wi <= 0) return 0;
    if ((d->flags & PTX_PRO_UP2) && ((d->Hi | d->Wi) & 1)) return 0;
    if (2 * d->ldx < C || (2 * d->ldx) % 8 || d->ldy < C || d->ldy % 8 || 2 * d->Kc < C || (2 * d->Kc) % 8 || d->Co_pad < C) return 0;
    const int Hs = (d->flags & PTX_PRO_UP2) ? d->Hi / 2 : d->Hi, Ws = (d->flags & PTX_PRO_UP2) ? d->Wi / 2 : d->Wi;
    if ((uint64_t)d->N * Hs * Ws * d->ldx * 4ull >= 0x80000000ull) return 0;
    if ((uint64_t)9 * d->Co_pad * d->Kc * 4ull >= 0x80000000ull) return 0;
    return (int64_t)d->N * cdiv(d->Hi, kC3TH) * cdiv(d->Wi, kC3TW) <= 0x7fffffffLL;
}

template <int NCH, int CT, bool UP2>
static int launch_c3(const C3Args& a, dim3 grid, hipStream_t st) {
    constexpr size_t lds = kC3PatchBytes + 4 * 8192 + 2 * (32 * CT) * sizeof(float);       // patch + four filter tiles + tables
    static_assert(lds <= 80 * 1024, "two workgroups per CU");
    static bool attr_set[64] = {};
    int dev = 0;
    PTX_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        PTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_f16_kernel<NCH, CT, UP2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_f16_kernel<NCH, CT, UP2>), grid, dim3(256), lds, st, a);
    return hip_check(hipGetLastError(), "conv3x3_f16 launch");
}

extern "C" int ptx_conv3x3_f16_fwd(const ptx_conv3d_desc* d, const void* x, const void* w_packed, const float* bias, void* y,
                                   const ptx_conv_fused_ext* ext, ptx_stream_t stream) {
    if (!d || !x || !w_packed || !y) return fail(PTX_ERR_INVALID, "conv3x3_f16: null pointer");
    if (!ptx_conv3x3_f16_supported(d))
        return fail(PTX_ERR_UNSUPPORTED, "conv3x3_f16: a unit-stride 3x3 conv with pad 1 over halfs, Ci == Co in {64, 128, 256}, at least 32 columns, halfs out, "
                    "flags within F16_OPERANDS | OUT_F16 | PRO_UP2 | EPI_AFFINE | EPI_RELU");
    if ((d->flags & PTX_EPI_AFFINE) && (!ext || !ext->scale || !ext->shift || ext->ld_affine < d->Co || ext->ld_affine % 4))
        return fail(PTX_ERR_INVALID, "conv3x3_f16: PTX_EPI_AFFINE needs scale / shift tables with a row stride that is a multiple of 4");
    if (((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)y | (uintptr_t)bias) & 15) return fail(PTX_ERR_INVALID, "conv3x3_f16: misaligned pointer");
    const bool up2 = (d->flags & PTX_PRO_UP2) != 0;
    C3Args a{};
    a.x = static_cast<const _Float16*>(x); a.w = static_cast<const _Float16*>(w_packed); a.bias = bias;
    a.scale = (d->flags & PTX_EPI_AFFINE) ? ext->scale : nullptr;
    a.shift = (d->flags & PTX_EPI_AFFINE) ? ext->shift : nullptr;
    a.ld_aff = (d->flags & PTX_EPI_AFFINE) ? ext->ld_affine : 0;
    a.y = static_cast<_Float16*>(y);
    a.N = d->N; a.H = d->Hi; a.W = d->Wi; a.Hs = up2 ? d->Hi / 2 : d->Hi; a.Ws = up2 ? d->Wi / 2 : d->Wi;
    a.ldx = 2 * d->ldx; a.ldy = d->ldy; a.Kc = 2 * d->Kc; a.tap_stride = d->Co_pad * a.Kc;
    a.tiles_h = cdiv(d->Hi, kC3TH); a.tiles_w = cdiv(d->Wi, kC3TW);
    a.x_bytes = (unsigned)((uint64_t)d->N * a.Hs * a.Ws * a.ldx * 2ull);
    a.w_bytes = (unsigned)((uint64_t)9 * d->Co_pad * a.Kc * 2ull);
    a.flags = d->flags;
    const dim3 grid((unsigned)(d->N * a.tiles_h * a.tiles_w), (unsigned)(d->Co <= 128 ? 1 : d->Co / 128));
    const hipStream_t st = (hipStream_t)stream;
    if (d->Co == 64) return up2 ? launch_c3<1, 2, true>(a, grid, st) : launch_c3<1, 2, false>(a, grid, st);
    if (d->Co == 128) return up2 ? launch_c3<2, 4, true>(a, grid, st) : launch_c3<2, 4, false>(a, grid, st);
    return up2 ? launch_c3<4, 4, true>(a, grid, st) : launch_c3<4, 4, false>(a, grid, st);
}

extern "C" int ptx_conv1x1_skip_f16_supported(const ptx_conv3d_desc* d) {
    if (!d) return 0;
    const unsigned need = PTX_F16_OPERANDS | PTX_EPI_OUT_F16;
    const unsigned may = need | PTX_EPI_AFFINE | PTX_EPI_RELU | PTX_EPI_DUAL_RAW | PTX_RES_F16 | PTX_EPI_RES_ADD | PTX_EPI_RES_PADA | PTX_EPI_RES_UP;
    if ((d->flags & need) != need || (d->flags & ~may)) return 0;
    const int K = 2 * d->Ci;
    if ((K != 64 && K != 128 && K != 256) || d->Co < 128 || d->Co % 128 || d->groups > 1) return 0;
    if (d->kT != 1 || d->kH != 1 || d->kW != 1 || d->sT != 1 || d->sH != 1 || d->sW != 1 || d->pT || d->pH || d->pW) return 0;
    if (d->Ti != 1 || d->To != 1 || d->Ho != d->Hi || d->Wo != d->Wi || d->N <= 0 || d->Hi <= 0 || d->Wi < kC3TW) return 0;
    if (2 * d->ldx < K || (2 * d->ldx) % 8 || d->ldy < d->Co || d->ldy % 8 || 2 * d->Kc < K || (2 * d->Kc) % 8 || d->Co_pad < d->Co) return 0;
    const bool has_res = (d->flags & (PTX_EPI_RES_ADD | PTX_EPI_RES_PADA)) != 0;
    if (has_res) {
        if (!(d->flags & PTX_RES_F16) || d->ldr % 8) return 0;                     // halfs only (fp32 skips keep the generic tiles)
        if ((d->flags & PTX_EPI_RES_ADD) && (d->flags & PTX_EPI_RES_PADA)) return 0;
        if (d->flags & PTX_EPI_RES_PADA) {
            if (!(d->flags & PTX_EPI_RES_UP) || d->res_sT != 0 || d->res_sH != d->res_sW || d->res_sH < 0 || d->res_sH > 1) return 0;
            if (d->res_C < d->Co || d->ldr < d->res_C || d->res_T != 1 || ((d->Hi - 1) >> d->res_sH) >= d->res_H || ((d->Wi - 1) >> d->res_sW) >= d->res_W) return 0;
        } else if (d->ldr < d->Co) return 0;
    } else if (d->flags & (PTX_RES_F16 | PTX_EPI_RES_UP)) return 0;
    if ((d->flags & PTX_EPI_DUAL_RAW) && !(d->flags & PTX_EPI_AFFINE)) return 0;
    if ((uint64_t)d->N * d->Hi * d->Wi * d->ldy * 2ull >= 0x80000000ull) return 0;                 // 32-bit buffer offsets on every operand
    if (has_res && (uint64_t)d->N * ((d->flags & PTX_EPI_RES_PADA) ? (uint64_t)d->res_H * d->res_W : (uint64_t)d->Hi * d->Wi) * d->ldr * 2ull >= 0x80000000ull) return 0;
    if ((uint64_t)d->N * d->Hi * d->Wi * d->ldx * 4ull >= 0x80000000ull || (uint64_t)d->Co_pad * d->Kc * 4ull >= 0x80000000ull) return 0;
    return (int64_t)d->N * cdiv(d->Hi, kC3TH) * cdiv(d->Wi, kC3TW) <= 0x7fffffffLL;
}

template <int NCH>
static int launch_c1(const C1Args& a, dim3 grid, hipStream_t st) {
    static_assert(kC1Lds <= 80 * 1024, "two workgroups per CU");
    static bool attr_set[64] = {};
    int dev = 0;
    PTX_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        PTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_skip_f16_kernel<NCH>), hipFuncAttributeMaxDynamicSharedMemorySize, kC1Lds));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv1x1_skip_f16_kernel<NCH>), grid, dim3(256), kC1Lds, st, a);
    return hip_check(hipGetLastError(), "conv1x1_skip_f16 launch");
}

extern "C" int ptx_conv1x1_skip_f16_fwd(const ptx_conv3d_desc* d, const void* x, const void* w_packed, const float* bias, const void* res,
                                        void* y, const ptx_conv_fused_ext* ext, ptx_stream_t stream) {
    if (!d || !x || !w_packed || !y) return fail(PTX_ERR_INVALID, "conv1x1_skip_f16: null pointer");
    if (!ptx_conv1x1_skip_f16_supported(d))
        return fail(PTX_ERR_UNSUPPORTED, "conv1x1_skip_f16: a 1x1 conv over halfs with 64 / 128 / 256 input channels, a multiple of 128 output "
                    "channels, at least 32 columns, halfs out, an optional half skip operand (same shape, or nearest-upsampled and channel-truncated)");
    const bool has_res = (d->flags & (PTX_EPI_RES_ADD | PTX_EPI_RES_PADA)) != 0;
    if (has_res && !res) return fail(PTX_ERR_INVALID, "conv1x1_skip_f16: residual flag set but res == NULL");
    if ((d->flags & PTX_EPI_AFFINE) && (!ext || !ext->scale || !ext->shift || ext->ld_affine < d->Co || ext->ld_affine % 4))
        return fail(PTX_ERR_INVALID, "conv1x1_skip_f16: PTX_EPI_AFFINE needs scale / shift tables with a row stride that is a multiple of 4");
    if ((d->flags & PTX_EPI_DUAL_RAW) && (!ext || !ext->y_raw || ext->ld_raw < d->Co || ext->ld_raw % 8))
        return fail(PTX_ERR_INVALID, "conv1x1_skip_f16: PTX_EPI_DUAL_RAW needs ext->y_raw with a row stride that is a multiple of 8 halfs");
    if (((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)y | (uintptr_t)bias | (uintptr_t)res | (uintptr_t)(ext ? ext->y_raw : nullptr)) & 15)
        return fail(PTX_ERR_INVALID, "conv1x1_skip_f16: misaligned pointer");
    if ((d->flags & PTX_EPI_DUAL_RAW) && (uint64_t)d->N * d->Hi * d->Wi * ext->ld_raw * 2ull >= 0x80000000ull)
        return fail(PTX_ERR_UNSUPPORTED, "conv1x1_skip_f16: the raw output of one launch must be < 2 GiB");
    C1Args a{};
    a.x = static_cast<const _Float16*>(x); a.w = static_cast<const _Float16*>(w_packed); a.bias = bias;
    a.res = has_res ? static_cast<const _Float16*>(res) : nullptr;
    a.scale = (d->flags & PTX_EPI_AFFINE) ? ext->scale : nullptr;
    a.shift = (d->flags & PTX_EPI_AFFINE) ? ext->shift : nullptr;
    a.ld_aff = (d->flags & PTX_EPI_AFFINE) ? ext->ld_affine : 0;
    a.y = static_cast<_Float16*>(y);
    a.y_raw = (d->flags & PTX_EPI_DUAL_RAW) ? static_cast<_Float16*>(ext->y_raw) : nullptr;
    a.ld_raw = (d->flags & PTX_EPI_DUAL_RAW) ? ext->ld_raw : 0;
    a.N = d->N; a.H = d->Hi; a.W = d->Wi; a.ldx = 2 * d->ldx; a.ldy = d->ldy; a.Kc = 2 * d->Kc;
    a.ldr = d->ldr;
    if (d->flags & PTX_EPI_RES_PADA) { a.rH = d->res_H; a.rW = d->res_W; a.ush = d->res_sH; }
    else { a.rH = d->Hi; a.rW = d->Wi; a.ush = 0; }
    a.tiles_h = cdiv(d->Hi, kC3TH); a.tiles_w = cdiv(d->Wi, kC3TW);
    a.x_bytes = (unsigned)((uint64_t)d->N * d->Hi * d->Wi * a.ldx * 2ull);
    a.w_bytes = (unsigned)((uint64_t)d->Co_pad * a.Kc * 2ull);
    a.r_bytes = has_res ? (unsigned)((uint64_t)d->N * a.rH * a.rW * a.ldr * 2ull) : 0u;
    a.y_bytes = (unsigned)((uint64_t)d->N * d->Hi * d->Wi * a.ldy * 2ull);
    a.raw_bytes = (d->flags & PTX_EPI_DUAL_RAW) ? (unsigned)((uint64_t)d->N * d->Hi * d->Wi * a.ld_raw * 2ull) : 0u;
    a.flags = d->flags;
    const dim3 grid((unsigned)(d->N * a.tiles_h * a.tiles_w), (unsigned)(d->Co / 128));
    const hipStream_t st = (hipStream_t)stream;
    const int K = 2 * d->Ci;
    return K == 64 ? launch_c1<1>(a, grid, st) : K == 128 ? launch_c1<2>(a, grid, st) : launch_c1<4>(a, grid, st);
}

// ---- ptx_conv1x1_pro_f16: descriptor = ptx_conv3d_desc of the 1x1 conv; the INPUT affine travels in ext_in (scale / shift /
// ld_affine of the block's cBN1), the output affine in ext as for the other fused stages ----
extern "C" int ptx_conv1x1_pro_f16_supported(const ptx_conv3d_desc* d) {
    if (!d) return 0;
    const unsigned need = PTX_F16_OPERANDS | PTX_EPI_OUT_F16;
    const unsigned may = need | PTX_EPI_AFFINE | PTX_EPI_RELU;
    if ((d->flags & need) != need || (d->flags & ~may)) return 0;
    const int K = 2 * d->Ci;
    if (K < 128 || K > 2048 || K % 128 || d->groups > 1) return 0;                  // whole PAIRS of 64-channel chunks
    if (d->Co != 64 && d->Co != 128 && d->Co != 256 && d->Co != 512) return 0;
    if (d->kT != 1 || d->kH != 1 || d->kW != 1 || d->sT != 1 || d->sH != 1 || d->sW != 1 || d->pT || d->pH || d->pW) return 0;
    if (d->Ti != 1 || d->To != 1 || d->Ho != d->Hi || d->Wo != d->Wi || d->N <= 0) return 0;
    if (((int64_t)d->Hi * d->Wi) % 256) return 0;                                   // a workgroup's positions belong to one sample
    if (2 * d->ldx < K || (2 * d->ldx) % 8 || d->ldy < d->Co || d->ldy % 8 || 2 * d->Kc < K || (2 * d->Kc) % 8 || d->Co_pad < d->Co) return 0;
    const uint64_t M = (uint64_t)d->N * d->Hi * d->Wi;
    if (M * d->ldx * 4ull >= 0x80000000ull || M * d->ldy * 2ull >= 0x80000000ull || (uint64_t)d->Co_pad * d->Kc * 4ull >= 0x80000000ull) return 0;
    return 1;
}

template <int CT, int NPT>
static int launch_p1(const P1Args& a, dim3 grid, hipStream_t st) {
    constexpr int CO = 32 * CT, POS = 128 * NPT;
    constexpr int region = (POS * (CO + 8) * 2 > 2 * CO * 128) ? POS * (CO + 8) * 2 : 2 * CO * 128;
    const size_t lds = (size_t)region + 4 * (size_t)a.K + 2 * CO * sizeof(float);
    if (lds > 80 * 1024) return fail(PTX_ERR_UNSUPPORTED, "conv1x1_pro_f16: %zu bytes of LDS", lds);
    static bool attr_set[64] = {};
    int dev = 0;
    PTX_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        PTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_pro_f16_kernel<CT, NPT>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv1x1_pro_f16_kernel<CT, NPT>), grid, dim3(256), lds, st, a);
    return hip_check(hipGetLastError(), "conv1x1_pro_f16 launch");
}

extern "C" int ptx_conv1x1_pro_f16_fwd(const ptx_conv3d_desc* d, const void* x, const ptx_conv_fused_ext* ext_in, const void* w_packed,
                                       const float* bias, void* y, const ptx_conv_fused_ext* ext, ptx_stream_t stream) {
    if (!d || !x || !w_packed || !y || !ext_in || !ext_in->scale || !ext_in->shift) return fail(PTX_ERR_INVALID, "conv1x1_pro_f16: null pointer");
    if (!ptx_conv1x1_pro_f16_supported(d))
        return fail(PTX_ERR_UNSUPPORTED, "conv1x1_pro_f16: a 1x1 conv over halfs, K a multiple of 128 in 128..2048, Co in {64, 128, 256, 512}, "
                    "H * W a multiple of 256, halfs out");
    const int K = 2 * d->Ci;
    if (ext_in->ld_affine < K || ext_in->ld_affine % 4) return fail(PTX_ERR_INVALID, "conv1x1_pro_f16: input affine row stride must cover K and be a multiple of 4");
    if ((d->flags & PTX_EPI_AFFINE) && (!ext || !ext->scale || !ext->shift || ext->ld_affine < d->Co || ext->ld_affine % 4))
        return fail(PTX_ERR_INVALID, "conv1x1_pro_f16: PTX_EPI_AFFINE needs scale / shift tables with a row stride that is a multiple of 4");
    if (((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)y | (uintptr_t)bias | (uintptr_t)ext_in->scale | (uintptr_t)ext_in->shift) & 15)
        return fail(PTX_ERR_INVALID, "conv1x1_pro_f16: misaligned pointer");
    P1Args a{};
    a.x = static_cast<const _Float16*>(x); a.w = static_cast<const _Float16*>(w_packed); a.bias = bias;
    a.scale1 = ext_in->scale; a.shift1 = ext_in->shift; a.ld1 = ext_in->ld_affine;
    a.scale2 = (d->flags & PTX_EPI_AFFINE) ? ext->scale : nullptr;
    a.shift2 = (d->flags & PTX_EPI_AFFINE) ? ext->shift : nullptr;
    a.ld2 = (d->flags & PTX_EPI_AFFINE) ? ext->ld_affine : 0;
    a.y = static_cast<_Float16*>(y);
    a.M = d->N * d->Hi * d->Wi; a.HW = d->Hi * d->Wi; a.K = K; a.ldx = 2 * d->ldx; a.ldy = d->ldy; a.Kc = 2 * d->Kc; a.nch = K / 64;
    a.x_bytes = (unsigned)((uint64_t)a.M * a.ldx * 2ull);
    a.w_bytes = (unsigned)((uint64_t)d->Co_pad * a.Kc * 2ull);
    a.y_bytes = (unsigned)((uint64_t)a.M * a.ldy * 2ull);
    a.flags = d->flags;
    const hipStream_t st = (hipStream_t)stream;
    if (d->Co == 64) return launch_p1<2, 2>(a, dim3((unsigned)(a.M / 256), 1), st);
    if (d->Co == 128) return launch_p1<4, 2>(a, dim3((unsigned)(a.M / 256), 1), st);
    return launch_p1<8, 1>(a, dim3((unsigned)(a.M / 128), (unsigned)(d->Co / 256)), st);
}

// Load-time weight packing (BN fold + K-major relayout) and the layout transforms at the API edge.
// All of these are HBM-bound byte movers: coalesced 16-byte accesses on the channels-last side,
// LDS tile transposes where both sides cannot be contiguous at once.
#include "ptx_common.h"
#include <algorithm>

namespace ptx {

thread_local char g_last_error[512] = "";
char* last_error_buf() { return g_last_error; }

// ---------------------------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float bn_scale(const float* gamma, const float* var, float eps, int co) {
    return gamma ? gamma[co] * (1.0f / sqrtf(var[co] + eps)) : 1.0f;
}

__global__ void __launch_bounds__(256) pack_weight_kernel(ptx_pack_desc d, const float* __restrict__ w,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ var, float eps,
                                                          float* __restrict__ out, size_t total) {
    const int taps_hw = d.kH * d.kW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int k = (int)(i % d.Kc);
        size_t t = i / d.Kc;
        const int co = (int)(t % d.Co_pad);
        const int tap = (int)(t / d.Co_pad);
        int kt, kh, kw, c;
        bool valid = co < d.Co;
        if (d.fold_kw) {
            kt = tap / d.kH;
            kh = tap % d.kH;
            kw = k / d.Ci;
            c = k % d.Ci;
            valid = valid && k < d.kW * d.Ci;
        } else {
            kt = tap / taps_hw;
            const int r = tap % taps_hw;
            kh = r / d.kW;
            kw = r % d.kW;
            c = k;
            valid = valid && k < d.Ci;
        }
        int ci_src = d.Ci;                 // channels per filter row in the source tensor
        if (d.sub_groups > 1 && valid) {
            // block-diagonal super-group: row co keeps only the columns of its own real group
            ci_src = d.Ci / d.sub_groups;
            const int cog = d.co_per_super / d.sub_groups;      // output channels per real group
            const int g_row = (co % d.co_per_super) / cog, g_col = c / ci_src;
            valid = g_row == g_col;
            c -= g_col * ci_src;
        }
        float v = 0.f;
        if (valid) {
            const size_t src = ((((size_t)co * ci_src + c) * d.kT + kt) * d.kH + kh) * d.kW + kw;
            v = w[src] * bn_scale(gamma, var, eps, co);
        }
        const size_t ld = d.ld_k > 0 ? (size_t)d.ld_k : (size_t)d.Kc;
        const size_t dst = ((size_t)tap * d.Co_pad + co) * ld + d.k_off + k;
        if (d.f16 == 2) {
            // split operands (PTX_F16X3_OPERANDS): the 8-channel block of column k holds 8 hi halfs then 8 lo halfs
            const size_t row_h = (((size_t)tap * d.Co_pad + co) * ld + d.k_off) * 2;      // row start, in halfs
            const _Float16 hi = (_Float16)v;
            const _Float16 lo = (_Float16)((v - (float)hi) * 4096.f);       // scaled lo: a normal half whenever hi is one
            _Float16* o = reinterpret_cast<_Float16*>(out) + row_h + (size_t)(k >> 3) * 16 + (k & 7);
            o[0] = hi;
            o[8] = lo;
        } else if (d.f16) reinterpret_cast<_Float16*>(out)[dst] = (_Float16)v;     // round-to-nearest-even
        else out[dst] = v;
    }
}

__global__ void pack_bias_kernel(int Co, int Co_pad, const float* __restrict__ conv_bias,
                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                 const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                 float* __restrict__ out, int accumulate) {
    const int co = blockIdx.x * blockDim.x + threadIdx.x;
    if (co >= Co_pad) return;
    float b = 0.f;
    if (co < Co) {
        const float s = bn_scale(gamma, var, eps, co);
        const float cb = conv_bias ? conv_bias[co] : 0.f;
        const float mu = mean ? mean[co] : 0.f;
        b = (beta ? beta[co] : 0.f) + (cb - mu) * s;
    }
    out[co] = accumulate ? out[co] + b : b;
}

// ---------------------------------------------------------------------------------------------
// [N][C][S] <-> [N][S][ld] tile transposes (32 x 32 floats through LDS, +1 pad)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ncs_to_nsc_kernel(const float* __restrict__ x, float* __restrict__ y, int C,
                                                         long long S, int ld) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const long long s0 = (long long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const float* xn = x + (size_t)n * C * S;
    float* yn = y + (size_t)n * S * ld;
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int c = c0 + ty + j;
        const long long s = s0 + tx;
        tile[ty + j][tx] = (c < C && s < S) ? xn[(size_t)c * S + s] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const long long s = s0 + ty + j;
        const int c = c0 + tx;
        if (s < S && c < ld) yn[(size_t)s * ld + c] = tile[tx][ty + j];   // zero beyond C by construction
    }
}

__global__ void __launch_bounds__(256) nsc_to_ncs_kernel(const float* __restrict__ x, float* __restrict__ y, int C,
                                                         long long S, int ld) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const long long s0 = (long long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* xn = x + (size_t)n * S * ld;
    float* yn = y + (size_t)n * C * S;
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const long long s = s0 + ty + j;
        const int c = c0 + tx;
        tile[ty + j][tx] = (s < S && c < C) ? xn[(size_t)s * ld + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int c = c0 + ty + j;
        const long long s = s0 + tx;
        if (c < C && s < S) yn[(size_t)c * S + s] = tile[tx][ty + j];
    }
}

// y[b][c][r] = x[b][r][c]; rows r in [R, ldy) of y are zero-filled
__global__ void __launch_bounds__(256) transpose_last2_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                              int R, int Cc, int ldx, int ldy) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* xb = x + (size_t)b * R * ldx;
    float* yb = y + (size_t)b * Cc * ldy;
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int r = r0 + ty + j, c = c0 + tx;
        tile[ty + j][tx] = (r < R && c < Cc) ? xb[(size_t)r * ldx + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int c = c0 + ty + j, r = r0 + tx;
        if (c < Cc && r < ldy) yb[(size_t)c * ldy + r] = tile[tx][ty + j];
    }
}

// ---------------------------------------------------------------------------------------------
// small-Cin stem: NCDHW -> [N][T][H][Wo][ld] with the kW taps folded into the channel axis.
// One workgroup per input row (n, t, h): the C channel rows are staged in LDS with coalesced
// loads, then the Wo x ld output row (contiguous in memory) is written as float4s.  A thread's
// float4 column q is fixed, so its four (kw, c) pairs are decoded once.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) fold_kw_kernel(const float* __restrict__ x, float* __restrict__ y, int C,
                                                      int T, int H, int W, long long stride_n, long long stride_c,
                                                      long long stride_t, int kW, int sW, int pW, int Wo, int ld) {
    extern __shared__ float rowbuf[];      // [C][W]
    const int h = blockIdx.x % H;
    const int t = (blockIdx.x / H) % T;
    const int n = blockIdx.x / (H * T);
    const float* xin = x + (size_t)n * stride_n + (size_t)t * stride_t + (size_t)h * W;
    for (int i = threadIdx.x; i < C * W; i += 256) {
        const int c = i / W, w = i - c * W;
        rowbuf[i] = xin[(size_t)c * stride_c + w];
    }
    __syncthreads();
    const int f4r = ld / 4;
    const int rows_per_pass = 256 / f4r;
    const int q = threadIdx.x % f4r;
    const int r0 = threadIdx.x / f4r;
    if (r0 >= rows_per_pass) return;
    int off[4], kwp[4];
    bool live[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = q * 4 + e;
        const int kw = k / C, c = k - kw * C;
        live[e] = k < kW * C;
        kwp[e] = kw - pW;
        off[e] = c * W + kw - pW;          // + wo * sW gives the LDS index of (c, wo*sW - pW + kw)
    }
    float* yrow = y + (size_t)blockIdx.x * Wo * ld;
    for (int wo = r0; wo < Wo; wo += rows_per_pass) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int wi = wo * sW + kwp[e];                  // input column of this tap
            v[e] = (live[e] && wi >= 0 && wi < W) ? rowbuf[off[e] + wo * sW] : 0.f;
        }
        f32x4 o = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(yrow + (size_t)wo * ld + q * 4) = o;
    }
}

// ---------------------------------------------------------------------------------------------
// uint8 frames [N][T][H][W][C] -> normalised fp32 (TransformImage's tensor half, utils.py:72-75).
// The fp32 operations and their order are the reference's (ToTensor /255, ToRange255 *255,
// Normalize (v - mean) / std); the _rn intrinsics keep the compiler from contracting them into FMAs
// or reciprocal multiplies, so the result is bit-identical to the CPU tensors.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float normalise_u8(unsigned char u, float mean, float stdv, int to_255) {
    float v = __fdiv_rn((float)u, 255.0f);
    if (to_255) v = __fmul_rn(v, 255.0f);
    return __fdiv_rn(__fsub_rn(v, mean), stdv);
}

__global__ void __launch_bounds__(256) frames_u8_to_ncdhw_kernel(const unsigned char* __restrict__ f, float* __restrict__ y,
                                                                 size_t total, int C, long long THW, ptx_norm_desc nd) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t s = e % (size_t)THW;              // (t, h, w)
        const size_t nc = e / (size_t)THW;
        const int c = (int)(nc % C);
        const size_t n = nc / C;
        const int cin = (nd.swap_rb && (c == 0 || c == 2)) ? 2 - c : c;
        y[e] = normalise_u8(f[(n * THW + s) * C + cin], nd.mean[c], nd.std[c], nd.to_255);
    }
}

// one block per (n, t, h) input row: W*C bytes -> normalised floats in LDS (kept channel-interleaved, so
// the folded row of output column wo is the contiguous run rowbuf[(wo*sW - pW)*C ...][0, kW*C)).
__global__ void __launch_bounds__(256) fold_kw_frames_u8_kernel(const unsigned char* __restrict__ f, float* __restrict__ y,
                                                                int C, int T, int H, int W, int frame_step, int T_full,
                                                                int kW, int sW, int pW, int Wo, int ld, ptx_norm_desc nd) {
    extern __shared__ float rowbuf[];      // [W][C]
    const int h = blockIdx.x % H;
    const int t = (blockIdx.x / H) % T;
    const int n = blockIdx.x / (H * T);
    const unsigned char* fin = f + ((((size_t)n * T_full + (size_t)t * frame_step) * H + h) * W) * C;
    for (int i = threadIdx.x; i < W * C; i += 256) {
        const int w = i / C, c = i - w * C;
        const int cin = (nd.swap_rb && (c == 0 || c == 2)) ? 2 - c : c;
        rowbuf[i] = normalise_u8(fin[w * C + cin], nd.mean[c], nd.std[c], nd.to_255);
    }
    __syncthreads();
    const int f4r = ld / 4;
    const int rows_per_pass = 256 / f4r;
    const int q = threadIdx.x % f4r;
    const int r0 = threadIdx.x / f4r;
    if (r0 >= rows_per_pass) return;
    float* yrow = y + (size_t)blockIdx.x * Wo * ld;
    for (int wo = r0; wo < Wo; wo += rows_per_pass) {
        const int base = (wo * sW - pW) * C;           // rowbuf index of folded column 0 (may be negative)
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = q * 4 + e;
            const int idx = base + k;
            v[e] = (k < kW * C && idx >= 0 && idx < W * C) ? rowbuf[idx] : 0.f;
        }
        f32x4 o = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(yrow + (size_t)wo * ld + q * 4) = o;
    }
}

static unsigned grid_for(size_t work_items) {
    size_t b = (work_items + 255) / 256;
    const size_t cap = (size_t)kNumCU * 8;
    if (b > cap) b = cap;
    if (b == 0) b = 1;
    return (unsigned)b;
}

// ---------------------------------------------------------------------------------------------
// parameter checksum: a 64-bit, position-sensitive sum over the bit patterns of a SET of fp32 tensors
// (table of (device pointer, element count) pairs).  The host compares it between forwards to notice
// weight edits that bypass torch's version counters (`p.data.fill_(..)`); HBM-bound, one launch.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) checksum_f32_kernel(const long long* __restrict__ table, int n,
                                                           unsigned long long* __restrict__ out) {
    const int t = blockIdx.y;
    const unsigned* __restrict__ p = reinterpret_cast<const unsigned*>(table[2 * t]);
    const long long cnt = table[2 * t + 1];
    unsigned long long acc = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < cnt; i += (long long)gridDim.x * 256)
        acc += (unsigned long long)p[i] * (unsigned long long)(2654435761u * (unsigned)i | 1u);
    acc *= (unsigned long long)(2 * t + 1);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)acc, o, 64);
        const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(acc >> 32), o, 64);
        acc += ((unsigned long long)hi << 32) | lo;
    }
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}

}  // namespace ptx

using namespace ptx;

// PTX_SOURCE_SHA256: sha256 of csrc/*.hip, csrc/*.h and include/ptx_amd.h, computed by build.py (source_hash) and passed on
// this file's command line -- `ptx_version()` names the source the loaded binary was compiled from
#ifndef PTX_SOURCE_SHA256
#define PTX_SOURCE_SHA256 "unstamped"
#endif
extern "C" const char* ptx_version(void) { return "ptx_amd 0.5.0 (gfx950, fp32 MFMA) src:" PTX_SOURCE_SHA256; }
extern "C" const char* ptx_last_error(void) { return last_error_buf(); }

extern "C" size_t ptx_packed_weight_elems(const ptx_pack_desc* d) {
    if (!d) return 0;
    const size_t taps = d->fold_kw ? (size_t)d->kT * d->kH : (size_t)d->kT * d->kH * d->kW;
    return taps * d->Co_pad * (d->ld_k > 0 ? d->ld_k : d->Kc);
}

extern "C" int ptx_pack_conv_weight(const ptx_pack_desc* d, const float* w, const float* conv_bias,
                                    const float* bn_gamma, const float* bn_beta, const float* bn_mean,
                                    const float* bn_var, float bn_eps, float* w_packed, float* bias_out,
                                    ptx_stream_t stream) {
    if (!d || !w || !w_packed || !bias_out) return fail(PTX_ERR_INVALID, "pack: null pointer");
    if (d->Co <= 0 || d->Ci <= 0 || d->kT <= 0 || d->kH <= 0 || d->kW <= 0)
        return fail(PTX_ERR_INVALID, "pack: non-positive extent");
    if (d->ld_k < 0 || d->k_off < 0 || (d->ld_k > 0 && (d->k_off + d->Kc > d->ld_k || d->ld_k % 4 || d->k_off % 4)))
        return fail(PTX_ERR_INVALID, "pack: bad K-concatenation window (ld_k=%d k_off=%d Kc=%d)", d->ld_k, d->k_off, d->Kc);
    if (d->ld_k == 0 && d->k_off != 0) return fail(PTX_ERR_INVALID, "pack: k_off needs ld_k");
    if (d->sub_groups > 1 && (d->fold_kw || d->Ci % d->sub_groups || d->co_per_super <= 0 ||
                              d->co_per_super % d->sub_groups || d->Co % d->co_per_super))
        return fail(PTX_ERR_INVALID, "pack: super-group packing needs sub_groups | Ci, sub_groups | co_per_super | Co, no kW fold");
    const int keff = d->fold_kw ? d->kW * d->Ci : d->Ci;
    if (d->f16 == 1 && (d->Kc % 8 || d->ld_k || d->k_off))
        return fail(PTX_ERR_INVALID, "pack: fp16 filters need Kc %% 8 == 0 and no K-concatenation window");
    if (d->f16 == 2 && (d->Kc % 8 || d->ld_k % 8 || d->k_off % 8 || d->sub_groups > 1))
        return fail(PTX_ERR_INVALID, "pack: split (hi8 | lo8) filters need Kc, ld_k and k_off %% 8 == 0, no super-groups");
    if (d->f16 < 0 || d->f16 > 2) return fail(PTX_ERR_INVALID, "pack: f16 must be 0 (fp32), 1 (halfs) or 2 (split halfs)");
    if (d->Kc < keff || d->Kc % 4 || d->Co_pad < d->Co || d->Co_pad % 128)
        return fail(PTX_ERR_INVALID, "pack: Kc=%d must cover K=%d (multiple of 4); Co_pad=%d must cover Co=%d (multiple of 128)",
                    d->Kc, keff, d->Co_pad, d->Co);
    if ((bn_gamma != nullptr) != (bn_var != nullptr))
        return fail(PTX_ERR_INVALID, "pack: bn_gamma and bn_var must be given together");
    const size_t taps = d->fold_kw ? (size_t)d->kT * d->kH : (size_t)d->kT * d->kH * d->kW;
    const size_t total = taps * d->Co_pad * d->Kc;     // elements written by this call (one K window)
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(grid_for(total)), dim3(256), 0, st, *d, w, bn_gamma, bn_var, bn_eps,
                       w_packed, total);
    PTX_HIP(hipGetLastError());
    hipLaunchKernelGGL(pack_bias_kernel, dim3(cdiv(d->Co_pad, 256)), dim3(256), 0, st, d->Co, d->Co_pad, conv_bias,
                       bn_gamma, bn_beta, bn_mean, bn_var, bn_eps, bias_out, d->bias_accumulate);
    return hip_check(hipGetLastError(), "pack launch");
}

static int check_layout_args(const void* x, const void* y, int N, int C, int64_t S, int ld) {
    if (!x || !y) return fail(PTX_ERR_INVALID, "layout: null pointer");
    if (N <= 0 || C <= 0 || S <= 0 || ld < C || ld % 4) return fail(PTX_ERR_INVALID, "layout: bad extents");
    if (N > 65535 || cdiv(ld, 32) > 65535) return fail(PTX_ERR_INVALID, "layout: N or C too large for the grid");
    return PTX_OK;
}

// C <= 4 channels -> 16-byte positions (the split-operand stem's input): every thread gathers its position's channels
// from C coalesced planes and writes one float4 -- the 32 x 32 tile transpose above spends 29 of its 32 channel rows
// on padding for an RGB clip (1.6 TB/s; this one streams at the HBM rate)
__global__ void __launch_bounds__(256) ncs_to_ns4_kernel(const float* __restrict__ x, float* __restrict__ y, int C,
                                                         long long S, long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long n = i / S, sp = i - n * S;
        const float* xn = x + (size_t)n * C * S + sp;
        f32x4 v = {xn[0], 0.f, 0.f, 0.f};
        if (C > 1) v.y = xn[S];
        if (C > 2) v.z = xn[2 * S];
        if (C > 3) v.w = xn[3 * S];
        *reinterpret_cast<f32x4*>(y + (size_t)i * 4) = v;
    }
}

// The split-operand stem's input format: one 16-byte position = 4 channels as (hi4 | lo4) halfs, hi = half(v),
// lo = half(v - hi) -- the split the x3 tiles do in registers, done once here, in the pass that leaves NCDHW anyway.
typedef _Float16 half8_s __attribute__((ext_vector_type(8)));
__global__ void __launch_bounds__(256) ncs_to_split4_kernel(const float* __restrict__ x, float* __restrict__ y, int C,
                                                            long long S, long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long n = i / S, sp = i - n * S;
        const float* xn = x + (size_t)n * C * S + sp;
        float v[4] = {xn[0], 0.f, 0.f, 0.f};
        if (C > 1) v[1] = xn[S];
        if (C > 2) v[2] = xn[2 * S];
        if (C > 3) v[3] = xn[3 * S];
        half8_s o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const _Float16 h = (_Float16)v[c];
            o[c] = h;
            o[4 + c] = (_Float16)((v[c] - (float)h) * 4096.f);      // scaled lo (conv_igemm.hip, X3)
        }
        *reinterpret_cast<half8_s*>(y + (size_t)i * 4) = o;
    }
}

extern "C" int ptx_ncdhw_to_split4(const float* x, void* y, int32_t N, int32_t C, int64_t S, ptx_stream_t stream) {
    if (!x || !y) return fail(PTX_ERR_INVALID, "ncdhw_to_split4: null pointer");
    if (N <= 0 || C <= 0 || C > 4 || S <= 0 || (((uintptr_t)y) & 15)) return fail(PTX_ERR_INVALID, "ncdhw_to_split4: 1..4 channels, 16-byte aligned output");
    const long long total = (long long)N * S;
    const unsigned blocks = (unsigned)std::min<long long>((total + 255) / 256, (long long)kNumCU * 32);
    hipLaunchKernelGGL(ncs_to_split4_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, static_cast<float*>(y), C, (long long)S, total);
    return hip_check(hipGetLastError(), "ncdhw_to_split4 launch");
}

extern "C" int ptx_ncdhw_to_ndhwc(const float* x, float* y, int32_t N, int32_t C, int64_t S, int32_t ld,
                                  ptx_stream_t stream) {
    int s = check_layout_args(x, y, N, C, S, ld);
    if (s) return s;
    if (C <= 4 && ld == 4 && (((uintptr_t)y) & 15) == 0) {
        const long long total = (long long)N * S;
        const unsigned blocks = (unsigned)std::min<long long>((total + 255) / 256, (long long)kNumCU * 32);
        hipLaunchKernelGGL(ncs_to_ns4_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, C, (long long)S, total);
        return hip_check(hipGetLastError(), "ncdhw_to_ndhwc launch");
    }
    dim3 grid((unsigned)cdiv64(S, 32), (unsigned)cdiv(ld, 32), (unsigned)N);
    hipLaunchKernelGGL(ncs_to_nsc_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, C, (long long)S, ld);
    return hip_check(hipGetLastError(), "ncdhw_to_ndhwc launch");
}

extern "C" int ptx_ndhwc_to_ncdhw(const float* x, float* y, int32_t N, int32_t C, int64_t S, int32_t ld,
                                  ptx_stream_t stream) {
    int s = check_layout_args(x, y, N, C, S, ld);
    if (s) return s;
    dim3 grid((unsigned)cdiv64(S, 32), (unsigned)cdiv(C, 32), (unsigned)N);
    hipLaunchKernelGGL(nsc_to_ncs_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, C, (long long)S, ld);
    return hip_check(hipGetLastError(), "ndhwc_to_ncdhw launch");
}

extern "C" int ptx_transpose_last2(const float* x, float* y, int32_t batch, int32_t R, int32_t Cc, int32_t ldx,
                                   int32_t ldy, ptx_stream_t stream) {
    if (!x || !y) return fail(PTX_ERR_INVALID, "transpose: null pointer");
    if (batch <= 0 || batch > 65535 || R <= 0 || Cc <= 0 || ldx < Cc || ldy < R)
        return fail(PTX_ERR_INVALID, "transpose: bad extents");
    dim3 grid((unsigned)cdiv(ldy, 32), (unsigned)cdiv(Cc, 32), (unsigned)batch);
    hipLaunchKernelGGL(transpose_last2_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, R, Cc, ldx, ldy);
    return hip_check(hipGetLastError(), "transpose_last2 launch");
}

static int check_fold_args(const void* x, const void* y, int N, int C, int T, int H, int W, int kW, int sW, int pW, int Wo,
                           int ld);

extern "C" int ptx_fold_kw_ncdhw(const float* x, float* y, int32_t N, int32_t C, int32_t T, int32_t H, int32_t W,
                                 int32_t kW, int32_t sW, int32_t pW, int32_t Wo, int32_t ld, ptx_stream_t stream) {
    const int64_t plane = (int64_t)T * H * W;
    return ptx_fold_kw_strided(x, y, N, C, T, H, W, (int64_t)C * plane, plane, (int64_t)H * W, kW, sW, pW, Wo, ld, stream);
}

static int check_fold_args(const void* x, const void* y, int N, int C, int T, int H, int W, int kW, int sW, int pW, int Wo,
                           int ld) {
    if (!x || !y) return fail(PTX_ERR_INVALID, "fold_kw: null pointer");
    if (N <= 0 || C <= 0 || T <= 0 || H <= 0 || W <= 0 || kW <= 0 || sW <= 0 || pW < 0 || Wo <= 0)
        return fail(PTX_ERR_INVALID, "fold_kw: non-positive extent");
    if (ld < kW * C || ld % 4) return fail(PTX_ERR_INVALID, "fold_kw: ld=%d must cover kW*C=%d and be a multiple of 4", ld, kW * C);
    {
        const int same = (W + sW - 1) / sW;              // TF-"SAME": pW is the front pad floor(total/2)
        const int total = std::max((same - 1) * sW + kW - W, 0);
        if (Wo != (W + 2 * pW - kW) / sW + 1 && !(Wo == same && pW == total / 2))
            return fail(PTX_ERR_INVALID, "fold_kw: Wo mismatch");
    }
    if ((uintptr_t)y & 15) return fail(PTX_ERR_INVALID, "fold_kw: misaligned output");
    if (ld / 4 > 256 || (size_t)C * W * sizeof(float) > 64 * 1024)
        return fail(PTX_ERR_UNSUPPORTED, "fold_kw: row of %d x %d floats does not fit the LDS staging buffer", C, W);
    if ((int64_t)N * T * H > 0x7fffffffLL) return fail(PTX_ERR_INVALID, "fold_kw: too many rows");
    return PTX_OK;
}

extern "C" int ptx_fold_kw_strided(const float* x, float* y, int32_t N, int32_t C, int32_t T, int32_t H, int32_t W,
                                   int64_t stride_n, int64_t stride_c, int64_t stride_t, int32_t kW, int32_t sW,
                                   int32_t pW, int32_t Wo, int32_t ld, ptx_stream_t stream) {
    int s = check_fold_args(x, y, N, C, T, H, W, kW, sW, pW, Wo, ld);
    if (s) return s;
    if (stride_n <= 0 || stride_c < (int64_t)H * W || stride_t < (int64_t)H * W)
        return fail(PTX_ERR_INVALID, "fold_kw: strides must be at least one H x W plane");
    hipLaunchKernelGGL(fold_kw_kernel, dim3((unsigned)(N * T * H)), dim3(256), (size_t)C * W * sizeof(float),
                       (hipStream_t)stream, x, y, C, T, H, W, (long long)stride_n, (long long)stride_c,
                       (long long)stride_t, kW, sW, pW, Wo, ld);
    return hip_check(hipGetLastError(), "fold_kw launch");
}

static int check_norm(const ptx_norm_desc* nd, int C) {
    if (!nd) return fail(PTX_ERR_INVALID, "frames_u8: null norm descriptor");
    if (C <= 0 || C > 4) return fail(PTX_ERR_INVALID, "frames_u8: C=%d must be 1..4", C);
    for (int c = 0; c < C; ++c)
        if (!(nd->std[c] != 0.f)) return fail(PTX_ERR_INVALID, "frames_u8: std[%d] must be non-zero", c);
    if (nd->swap_rb && C < 3) return fail(PTX_ERR_INVALID, "frames_u8: BGR swap needs 3 channels");
    return PTX_OK;
}

extern "C" int ptx_frames_u8_to_ncdhw(const uint8_t* frames, float* y, int32_t N, int32_t T, int32_t H, int32_t W,
                                      int32_t C, const ptx_norm_desc* norm, ptx_stream_t stream) {
    if (!frames || !y) return fail(PTX_ERR_INVALID, "frames_u8: null pointer");
    if (N <= 0 || T <= 0 || H <= 0 || W <= 0) return fail(PTX_ERR_INVALID, "frames_u8: non-positive extent");
    int s = check_norm(norm, C);
    if (s) return s;
    const long long THW = (long long)T * H * W;
    const size_t total = (size_t)N * C * THW;
    hipLaunchKernelGGL(frames_u8_to_ncdhw_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, frames, y,
                       total, C, THW, *norm);
    return hip_check(hipGetLastError(), "frames_u8_to_ncdhw launch");
}

extern "C" int ptx_fold_kw_frames_u8(const uint8_t* frames, float* y, int32_t N, int32_t C, int32_t T, int32_t H,
                                     int32_t W, int32_t frame_step, int32_t T_full, int32_t kW, int32_t sW, int32_t pW,
                                     int32_t Wo, int32_t ld, const ptx_norm_desc* norm, ptx_stream_t stream) {
    int s = check_fold_args(frames, y, N, C, T, H, W, kW, sW, pW, Wo, ld);
    if (s) return s;
    s = check_norm(norm, C);
    if (s) return s;
    if (frame_step <= 0 || T_full <= 0 || (int64_t)(T - 1) * frame_step >= T_full)
        return fail(PTX_ERR_INVALID, "fold_kw_frames_u8: T=%d frames at step %d do not fit T_full=%d", T, frame_step, T_full);
    hipLaunchKernelGGL(fold_kw_frames_u8_kernel, dim3((unsigned)(N * T * H)), dim3(256), (size_t)C * W * sizeof(float),
                       (hipStream_t)stream, frames, y, C, T, H, W, frame_step, T_full, kW, sW, pW, Wo, ld, *norm);
    return hip_check(hipGetLastError(), "fold_kw_frames_u8 launch");
}

// y[r][0..W) = x[r][0..W), y[r][W..ld) = 0: gives rows whose length is not a multiple of 4 floats a 16-byte pitch
__global__ void __launch_bounds__(256) pad_rows_kernel(const float* __restrict__ x, float* __restrict__ y, size_t total,
                                                       int W, int ld) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t r = e / (size_t)ld;
        const int c = (int)(e - r * (size_t)ld);
        y[e] = c < W ? x[r * (size_t)W + c] : 0.f;
    }
}

extern "C" int ptx_pad_rows(const float* x, float* y, int64_t rows, int32_t W, int32_t ld, ptx_stream_t stream) {
    if (!x || !y) return fail(PTX_ERR_INVALID, "pad_rows: null pointer");
    if (rows <= 0 || W <= 0 || ld < W) return fail(PTX_ERR_INVALID, "pad_rows: bad extents (rows=%lld W=%d ld=%d)", (long long)rows, W, ld);
    const size_t total = (size_t)rows * ld;
    hipLaunchKernelGGL(pad_rows_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, total, W, ld);
    return hip_check(hipGetLastError(), "pad_rows launch");
}

extern "C" int ptx_checksum_f32(const int64_t* table, int32_t n, uint64_t* out, ptx_stream_t stream) {
    if (!table || !out || n <= 0) return fail(PTX_ERR_INVALID, "checksum: null pointer / empty table");
    if (n > 65535) return fail(PTX_ERR_UNSUPPORTED, "checksum: more than 65535 tensors");
    hipLaunchKernelGGL(checksum_f32_kernel, dim3(32, (unsigned)n), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const long long*>(table), n, reinterpret_cast<unsigned long long*>(out));
    return hip_check(hipGetLastError(), "checksum launch");
}

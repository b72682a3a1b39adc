// HBM-bound pieces of the path: 3-D max pooling, global average pooling, the small-M linear
// layers (classifier head, TRN relation MLP) and the non-local block's row softmax.
// Channels-last everywhere, so a wave's 64 lanes always touch 64 consecutive float4/float.
#include "ptx_common.h"
#include <cfloat>

namespace ptx {

// ---------------------------------------------------------------------------------------------
// max_pool3d, -inf padding (every window of the reference's k3 s2 p1 pool has >= 1 valid tap)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) maxpool3d_kernel(ptx_pool3d_desc d, const float* __restrict__ x,
                                                        float* __restrict__ y, size_t total4) {
    const int f4r = (d.C + 3) / 4;
    const int ldy = d.ldy ? d.ldy : d.ld;
    const bool pad_zero = (d.flags & PTX_POOL_PAD_ZERO) != 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const int q = (int)(i % f4r);
        size_t pos = i / f4r;
        const int wo = (int)(pos % d.Wo);
        size_t t1 = pos / d.Wo;
        const int ho = (int)(t1 % d.Ho);
        t1 /= d.Ho;
        const int to = (int)(t1 % d.To);
        const int n = (int)(t1 / d.To);
        const int t_lo = max(0, to * d.sT - d.pT), t_hi = min(d.Ti, to * d.sT - d.pT + d.kT);
        const int h_lo = max(0, ho * d.sH - d.pH), h_hi = min(d.Hi, ho * d.sH - d.pH + d.kH);
        const int w_lo = max(0, wo * d.sW - d.pW), w_hi = min(d.Wi, wo * d.sW - d.pW + d.kW);
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int t = t_lo; t < t_hi; ++t)
            for (int h = h_lo; h < h_hi; ++h) {
                const float* row = x + ((((size_t)n * d.Ti + t) * d.Hi + h) * d.Wi) * d.ld + q * 4;
                for (int w = w_lo; w < w_hi; ++w) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(row + (size_t)w * d.ld);
                    m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y);
                    m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
                }
            }
        if (pad_zero && ((t_hi - t_lo) != d.kT || (h_hi - h_lo) != d.kH || (w_hi - w_lo) != d.kW)) {
            m.x = fmaxf(m.x, 0.f); m.y = fmaxf(m.y, 0.f); m.z = fmaxf(m.z, 0.f); m.w = fmaxf(m.w, 0.f);
        }
        // pad channels [C, round_up(C,4)) hold zeros on input, so they stay zero on output
        *reinterpret_cast<f32x4*>(y + pos * ldy + q * 4) = m;
    }
}

// kW = 3 along W with stride 1 or 2 (resnet3D.py:156's MaxPool3d(3,2,1); the SAME-padded 3x3x3 / (1,3,3) pools of the
// Inception modules): a thread produces WSEG consecutive outputs of one (n, to, ho, 4-channel group) and slides along W,
// so an input column shared by neighbouring windows is reduced over (t, h) ONCE: ((WSEG-1)*SW + 3) * kT * kH loads for
// WSEG outputs -- 19 per output at stride 2, 11 at stride 1 (WSEG = 8) -- instead of 27.  Any front pad < 3, explicit
// (SAME) output extents and zero-valued padding (PTX_POOL_PAD_ZERO: a clipped window also sees a 0) are handled.
template <int WSEG, int SW>
__global__ void __launch_bounds__(256) maxpool3d_slide_kernel(ptx_pool3d_desc d, const float* __restrict__ x,
                                                              float* __restrict__ y, size_t total) {
    constexpr int NCOL = (WSEG - 1) * SW + 3;
    const int f4r = (d.C + 3) / 4;
    const int ldy = d.ldy ? d.ldy : d.ld;
    const int segs = (d.Wo + WSEG - 1) / WSEG;
    const bool pad_zero = (d.flags & PTX_POOL_PAD_ZERO) != 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int q = (int)(i % f4r);
        size_t r = i / f4r;
        const int seg = (int)(r % segs);
        r /= segs;
        const int ho = (int)(r % d.Ho);
        r /= d.Ho;
        const int to = (int)(r % d.To);
        const int n = (int)(r / d.To);
        const int t_lo = max(0, to * d.sT - d.pT), t_hi = min(d.Ti, to * d.sT - d.pT + d.kT);
        const int h_lo = max(0, ho * d.sH - d.pH), h_hi = min(d.Hi, ho * d.sH - d.pH + d.kH);
        const bool th_clipped = (t_hi - t_lo) != d.kT || (h_hi - h_lo) != d.kH;
        const float* base = x + ((size_t)n * d.Ti * d.Hi * d.Wi) * d.ld + q * 4;
        const int wo0 = seg * WSEG;
        const int w0 = wo0 * SW - d.pW;                       // input column of cm[0]
        f32x4 cm[NCOL];
        unsigned live = 0;                                    // columns inside the input that a live window consumes
#pragma unroll
        for (int c = 0; c < NCOL; ++c) {
            cm[c] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            const int w = w0 + c;
            if (w >= 0 && w < d.Wi && wo0 + (c < 3 ? 0 : (c - 2 + SW - 1) / SW) < d.Wo) live |= 1u << c;
        }
        // (t, h) outside, the columns inside: NCOL independent 16-byte loads in flight per row
        for (int t = t_lo; t < t_hi; ++t)
            for (int h = h_lo; h < h_hi; ++h) {
                const float* row = base + (((ptrdiff_t)t * d.Hi + h) * d.Wi + w0) * (ptrdiff_t)d.ld;
                f32x4 v[NCOL];
#pragma unroll
                for (int c = 0; c < NCOL; ++c)
                    if (live >> c & 1) v[c] = *reinterpret_cast<const f32x4*>(row + (ptrdiff_t)c * d.ld);
#pragma unroll
                for (int c = 0; c < NCOL; ++c)
                    if (live >> c & 1) {
                        cm[c].x = fmaxf(cm[c].x, v[c].x); cm[c].y = fmaxf(cm[c].y, v[c].y);
                        cm[c].z = fmaxf(cm[c].z, v[c].z); cm[c].w = fmaxf(cm[c].w, v[c].w);
                    }
            }
        float* yrow = y + ((((size_t)n * d.To + to) * d.Ho + ho) * d.Wo) * ldy + q * 4;
#pragma unroll
        for (int j = 0; j < WSEG; ++j) {
            const int wo = wo0 + j;
            if (wo < d.Wo) {
                const f32x4 a = cm[j * SW], b = cm[j * SW + 1], c = cm[j * SW + 2];
                f32x4 o;
                o.x = fmaxf(a.x, fmaxf(b.x, c.x)); o.y = fmaxf(a.y, fmaxf(b.y, c.y));
                o.z = fmaxf(a.z, fmaxf(b.z, c.z)); o.w = fmaxf(a.w, fmaxf(b.w, c.w));
                const int wl = wo * SW - d.pW;
                if (pad_zero && (th_clipped || wl < 0 || wl + 3 > d.Wi)) {
                    o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
                }
                *reinterpret_cast<f32x4*>(yrow + (size_t)wo * ldy) = o;
            }
        }
    }
}

// 3 x 3 windows along (H, W) with equal stride S in {1, 2}: a thread produces an HSEG x WSEG patch of one
// (n, to, 4-channel group).  Every input row of the patch's halo is loaded once per temporal tap, reduced along W into
// WSEG row maxima and folded into the (up to three) output rows whose window holds it: kT * NR * NC loads for
// HSEG * WSEG outputs (4 x 4, S = 1, kT = 3: 6.75 per output where the W-sliding kernel above needs 13.5 and the
// generic one 27) -- the stride-1 SAME pools of the Inception branches are L1/L2-request bound, not HBM bound.
template <int HSEG, int WSEG, int S, int NB>
__global__ void __launch_bounds__(256) maxpool3d_tile_kernel(ptx_pool3d_desc d, const float* __restrict__ x,
                                                             float* __restrict__ y, unsigned total, int xcd_local) {
    constexpr int NR = (HSEG - 1) * S + 3, NC = (WSEG - 1) * S + 3;
    constexpr int RB = NB == 0 ? 1 : (NR + NB - 1) / NB;       // halo rows whose loads are issued together
    const int f4r = (d.C + 3) / 4;
    const int ldy = d.ldy ? d.ldy : d.ld;
    const int wsegs = (d.Wo + WSEG - 1) / WSEG, hsegs = (d.Ho + HSEG - 1) / HSEG;
    const bool pad_zero = (d.flags & PTX_POOL_PAD_ZERO) != 0;
    // xcd_local: workgroup b runs on XCD b % 8 -- give every XCD a CONTIGUOUS chunk of the (n, to, hs, ws) list, so the halo
    // rows / frames two neighbouring patches share (1.27x spatially, 1.5x temporally for the 3x3x3 / 2 pool) are re-read
    // from that XCD's own L2 instead of missing in eight of them (round 4)
    const unsigned blk = (xcd_local & 1) ? (unsigned)xcd_remap((int)blockIdx.x, (int)gridDim.x) : blockIdx.x;
    for (unsigned i = blk * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const int q = (int)(i % (unsigned)f4r);
        unsigned r = i / (unsigned)f4r;
        int ws, hs, to;
        if (xcd_local & 2) {              // frames innermost: the output frames of one patch sit in one workgroup, so the
            to = (int)(r % (unsigned)d.To);   // input frames two of them share are re-read from L1 / L2 at once (a kT-frame
            r /= (unsigned)d.To;              // window of big frames does not survive in a 4 MiB L2 until the next `to`)
            ws = (int)(r % (unsigned)wsegs);
            r /= (unsigned)wsegs;
            hs = (int)(r % (unsigned)hsegs);
            r /= (unsigned)hsegs;
        } else {
            ws = (int)(r % (unsigned)wsegs);
            r /= (unsigned)wsegs;
            hs = (int)(r % (unsigned)hsegs);
            r /= (unsigned)hsegs;
            to = (int)(r % (unsigned)d.To);
            r /= (unsigned)d.To;
        }
        const int n = (int)r;
        const int t_lo = max(0, to * d.sT - d.pT), t_hi = min(d.Ti, to * d.sT - d.pT + d.kT);
        const bool t_clipped = (t_hi - t_lo) != d.kT;
        const int ho0 = hs * HSEG, wo0 = ws * WSEG;
        const int h0 = ho0 * S - d.pH, w0 = wo0 * S - d.pW;
        unsigned live_c = 0, live_r = 0;                      // halo columns / rows inside the input that a live window uses
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (w0 + c >= 0 && w0 + c < d.Wi && wo0 + (c < 3 ? 0 : (c - 2 + S - 1) / S) < d.Wo) live_c |= 1u << c;
#pragma unroll
        for (int rr = 0; rr < NR; ++rr)
            if (h0 + rr >= 0 && h0 + rr < d.Hi && ho0 + (rr < 3 ? 0 : (rr - 2 + S - 1) / S) < d.Ho) live_r |= 1u << rr;
        f32x4 o[HSEG][WSEG];
#pragma unroll
        for (int a = 0; a < HSEG; ++a)
#pragma unroll
            for (int b = 0; b < WSEG; ++b) o[a][b] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        const float* base = x + ((size_t)n * d.Ti * d.Hi * d.Wi) * d.ld + q * 4;
        for (int t = t_lo; t < t_hi; ++t) {
            const float* plane = base + (((ptrdiff_t)t * d.Hi + h0) * d.Wi + w0) * (ptrdiff_t)d.ld;
#pragma unroll
            for (int r0 = 0; r0 < NR; r0 += RB) {
                // NB = 0: one row at a time; the (wave-divergent) skip of a dead row also keeps the compiler from hoisting
                // every row's loads to the top (256 VGPRs + scratch, measured 2x slower on the big stride-2 pools)
                if (RB == 1 && !(live_r >> r0 & 1)) continue;
                f32x4 v[RB][NC];
#pragma unroll
                for (int k = 0; k < RB; ++k)
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        v[k][c] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                        if (r0 + k < NR && (live_r >> (r0 + k) & 1) && (live_c >> c & 1))
                            v[k][c] = *reinterpret_cast<const f32x4*>(plane + ((ptrdiff_t)(r0 + k) * d.Wi + c) * d.ld);
                    }
#pragma unroll
                for (int k = 0; k < RB; ++k) {
                    const int rr = r0 + k;                    // a dead row folds -inf: harmless
                    if (rr >= NR) break;
#pragma unroll
                    for (int b = 0; b < WSEG; ++b) {
                        f32x4 m;
                        m.x = fmaxf(v[k][b * S].x, fmaxf(v[k][b * S + 1].x, v[k][b * S + 2].x));
                        m.y = fmaxf(v[k][b * S].y, fmaxf(v[k][b * S + 1].y, v[k][b * S + 2].y));
                        m.z = fmaxf(v[k][b * S].z, fmaxf(v[k][b * S + 1].z, v[k][b * S + 2].z));
                        m.w = fmaxf(v[k][b * S].w, fmaxf(v[k][b * S + 1].w, v[k][b * S + 2].w));
#pragma unroll
                        for (int a = 0; a < HSEG; ++a)
                            if (rr >= a * S && rr <= a * S + 2) {
                                o[a][b].x = fmaxf(o[a][b].x, m.x); o[a][b].y = fmaxf(o[a][b].y, m.y);
                                o[a][b].z = fmaxf(o[a][b].z, m.z); o[a][b].w = fmaxf(o[a][b].w, m.w);
                            }
                    }
                }
            }
        }
#pragma unroll
        for (int a = 0; a < HSEG; ++a) {
            const int ho = ho0 + a;
            if (ho >= d.Ho) break;
            const int hl = ho * S - d.pH;
            const bool th_clipped = t_clipped || hl < 0 || hl + 3 > d.Hi;
            float* yrow = y + ((((size_t)n * d.To + to) * d.Ho + ho) * d.Wo) * ldy + q * 4;
#pragma unroll
            for (int b = 0; b < WSEG; ++b) {
                const int wo = wo0 + b;
                if (wo < d.Wo) {
                    f32x4 v = o[a][b];
                    const int wl = wo * S - d.pW;
                    if (pad_zero && (th_clipped || wl < 0 || wl + 3 > d.Wi)) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                    *reinterpret_cast<f32x4*>(yrow + (size_t)wo * ldy) = v;
                }
            }
        }
    }
}

// 3 x 3 x 3 / stride 2 / pad 1 (the ResNet3D stem pool, resnet3D.py:156), ROLLING over the frames: a thread owns a 2 x 2 output patch
// x 4 channels of one clip for ALL output frames, reads every input frame of its 5 x 5 halo window exactly once (25 loads), pools
// it in (H, W) and folds the 2-D result into the (at most two) output frames whose temporal window holds it -- output frame
// `to` covers input frames 2 to - 1, 2 to, 2 to + 1.  The tile kernel above re-reads every input frame for 1.5 output frames
// (15.2 loads per output at 4 x 4 patches); this one issues 25 / 4 x 2 = 12.5.  Same maxima in a different association: exact.
__global__ void __launch_bounds__(256) maxpool3d_roll_kernel(ptx_pool3d_desc d, const float* __restrict__ x, float* __restrict__ y,
                                                             unsigned total, int xcd_local) {
    const int f4r = (d.C + 3) / 4;
    const int ldy = d.ldy ? d.ldy : d.ld;
    const int wsegs = (d.Wo + 1) / 2, hsegs = (d.Ho + 1) / 2;
    const unsigned blk = xcd_local ? (unsigned)xcd_remap((int)blockIdx.x, (int)gridDim.x) : blockIdx.x;
    for (unsigned i = blk * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const int q = (int)(i % (unsigned)f4r);
        unsigned r = i / (unsigned)f4r;
        const int ws = (int)(r % (unsigned)wsegs);
        r /= (unsigned)wsegs;
        const int hs = (int)(r % (unsigned)hsegs);
        const int n = (int)(r / (unsigned)hsegs);
        const int ho0 = hs * 2, wo0 = ws * 2;
        const int h0 = ho0 * 2 - 1, w0 = wo0 * 2 - 1;
        unsigned live_c = 0, live_r = 0;
#pragma unroll
        for (int c = 0; c < 5; ++c)
            if (w0 + c >= 0 && w0 + c < d.Wi && wo0 + (c < 3 ? 0 : 1) < d.Wo) live_c |= 1u << c;
#pragma unroll
        for (int rr = 0; rr < 5; ++rr)
            if (h0 + rr >= 0 && h0 + rr < d.Hi && ho0 + (rr < 3 ? 0 : 1) < d.Ho) live_r |= 1u << rr;
        const f32x4 ninf = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        f32x4 acc[2][2] = {{ninf, ninf}, {ninf, ninf}};          // running maximum of the output frame being assembled
        const float* base = x + ((size_t)n * d.Ti * d.Hi * d.Wi) * d.ld + q * 4;
        float* ybase = y + (((size_t)n * d.To) * d.Ho * d.Wo) * ldy + q * 4;
        for (int t = 0; t < d.Ti; ++t) {
            const float* plane = base + (((ptrdiff_t)t * d.Hi + h0) * d.Wi + w0) * (ptrdiff_t)d.ld;
            f32x4 v[5][5];
#pragma unroll
            for (int rr = 0; rr < 5; ++rr)
#pragma unroll
                for (int c = 0; c < 5; ++c) {
                    v[rr][c] = ninf;
                    if ((live_r >> rr & 1) && (live_c >> c & 1))
                        v[rr][c] = *reinterpret_cast<const f32x4*>(plane + ((ptrdiff_t)rr * d.Wi + c) * d.ld);
                }
            f32x4 f[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    f32x4 m = ninf;
#pragma unroll
                    for (int rr = 2 * a; rr < 2 * a + 3; ++rr)
#pragma unroll
                        for (int c = 2 * b; c < 2 * b + 3; ++c) {
                            m.x = fmaxf(m.x, v[rr][c].x); m.y = fmaxf(m.y, v[rr][c].y);
                            m.z = fmaxf(m.z, v[rr][c].z); m.w = fmaxf(m.w, v[rr][c].w);
                        }
                    f[a][b] = m;
                }
            const bool closes = (t & 1) || t == d.Ti - 1;         // frame 2 to + 1 (or the clip's last frame) completes output `to`
            const int to = t >> 1;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    f32x4 m = acc[a][b];
                    m.x = fmaxf(m.x, f[a][b].x); m.y = fmaxf(m.y, f[a][b].y); m.z = fmaxf(m.z, f[a][b].z); m.w = fmaxf(m.w, f[a][b].w);
                    if (closes) {
                        if (to < d.To && ho0 + a < d.Ho && wo0 + b < d.Wo)
                            *reinterpret_cast<f32x4*>(ybase + (((size_t)to * d.Ho + ho0 + a) * d.Wo + wo0 + b) * ldy) = m;
                        acc[a][b] = (t & 1) ? f[a][b] : ninf;     // an odd frame 2 to + 1 is also frame 2 (to + 1) - 1 of the next output
                    } else {
                        acc[a][b] = m;
                    }
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// global average pool
// ---------------------------------------------------------------------------------------------
// channels-last: x [N][S][ld] -> y [N][C]; block = 4 row-groups x 64 channels
__global__ void __launch_bounds__(256) avgpool_cl_kernel(const float* __restrict__ x, float* __restrict__ y, int C,
                                                         long long S, int ld) {
    __shared__ float part[4][64];
    const int n = blockIdx.y;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int g = threadIdx.x >> 6;
    float acc = 0.f;
    if (c < C) {
        const float* xn = x + (size_t)n * S * ld + c;
        for (long long s = g; s < S; s += 4) acc += xn[(size_t)s * ld];
    }
    part[g][threadIdx.x & 63] = acc;
    __syncthreads();
    if (g == 0 && c < C) {
        const int l = threadIdx.x & 63;
        y[(size_t)n * C + c] = (part[0][l] + part[1][l] + part[2][l] + part[3][l]) / (float)S;
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// channels-first: x [N][C][S] -> y [N][C]; one wave per (n, c)
__global__ void __launch_bounds__(256) avgpool_cf_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         long long NC, long long S) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= NC) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + (size_t)row * S;
    float acc = 0.f;
    for (long long s = lane; s < S; s += 64) acc += xr[s];
    acc = wave_sum(acc);
    if (lane == 0) y[row] = acc / (float)S;
}

// ---------------------------------------------------------------------------------------------
// small-M linear: one wave per output feature, MR rows of x at a time (weight row read once)
// ---------------------------------------------------------------------------------------------
template <int MR>
__global__ void __launch_bounds__(256) linear_smallm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, float* __restrict__ y,
                                                            int M, int K, int Nout, int ldx, int ldy, unsigned flags) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= Nout) return;
    const int lane = threadIdx.x & 63;
    const int m0 = blockIdx.y * MR;
    const bool relu_in = (flags & PTX_PRO_RELU) != 0;
    const float* wr = w + (size_t)j * K;
    float acc[MR];
#pragma unroll
    for (int r = 0; r < MR; ++r) acc[r] = 0.f;
    const bool vec = (K % 4 == 0) && (ldx % 4 == 0) && ((((uintptr_t)x | (uintptr_t)w) & 15) == 0);
    if (vec) {
        for (int k = lane * 4; k < K; k += 256) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + k);
#pragma unroll
            for (int r = 0; r < MR; ++r) {
                if (m0 + r < M) {
                    f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)(m0 + r) * ldx + k);
                    if (relu_in) {
                        xv.x = fmaxf(xv.x, 0.f); xv.y = fmaxf(xv.y, 0.f);
                        xv.z = fmaxf(xv.z, 0.f); xv.w = fmaxf(xv.w, 0.f);
                    }
                    acc[r] = fmaf(xv.x, wv.x, acc[r]);
                    acc[r] = fmaf(xv.y, wv.y, acc[r]);
                    acc[r] = fmaf(xv.z, wv.z, acc[r]);
                    acc[r] = fmaf(xv.w, wv.w, acc[r]);
                }
            }
        }
    } else {
        for (int k = lane; k < K; k += 64) {
            const float wv = wr[k];
#pragma unroll
            for (int r = 0; r < MR; ++r) {
                if (m0 + r < M) {
                    float xv = x[(size_t)(m0 + r) * ldx + k];
                    if (relu_in) xv = fmaxf(xv, 0.f);
                    acc[r] = fmaf(xv, wv, acc[r]);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < MR; ++r) {
        const float s = wave_sum(acc[r]);
        if (lane == 0 && m0 + r < M) {
            float v = s + (b ? b[j] : 0.f);
            if (flags & PTX_EPI_ACCUM) v += y[(size_t)(m0 + r) * ldy + j];
            if (flags & PTX_EPI_RELU) v = fmaxf(v, 0.f);
            y[(size_t)(m0 + r) * ldy + j] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// skinny GEMM (M <= 8 rows per pass, weight-bandwidth bound): a 256-thread workgroup owns FB = 4
// output features and the whole K extent.  Every thread streams float4 chunks of the FB weight rows
// (coalesced: consecutive lanes -> consecutive 16-byte pieces of a row) against the MR = 8 input rows
// (L2 resident, shared by all workgroups), holding FB x MR partial sums; the 32 sums are combined
// across the wave with a 32-shuffle reduce-scatter butterfly and across the 4 waves through LDS.
// N/FB workgroups cover the chip with many 16-byte weight loads in flight per CU, which is what an
// HBM-bound GEMV needs (the one-wave-per-feature kernel above reached 0.45 TB/s of weight traffic, this
// one 1.5 TB/s; the remaining gap is L2 traffic of the input rows, re-read once per workgroup).
//
// Input modes (SkinnyArgs::mode):
//   0 dense      x_eff[m][k] = x[m*ldx + k]
//   1 gather     row m = r*B + b reads the frames idx[r][0..n_seg) of video b:  x[b*ldx + idx[r][k / F]*F + k % F]
//   2 set-sum    x_eff[b][k] = sum_r x[(r*B + b)*ldx + k]           (bias is scaled by n_sets)
// ---------------------------------------------------------------------------------------------
struct SkinnyArgs {
    const float* x;
    const float* w;
    const float* b;
    float* y;
    int M, K, Nout, ldx, ldy;
    unsigned flags;
    int mode, B, n_sets, seg_len, n_seg;
    float bias_scale;
    int idx[PTX_REL_MAX_SETS][PTX_REL_MAX_FRAMES];
};

constexpr int kSkMR = 8;

// FB features per workgroup, NT threads: long K -> 256 threads x 4 features; short K / few features ->
// one-wave workgroups and 2 features, so that the grid still covers the chip.
template <int kSkFB, int kSkNT>
__global__ void __launch_bounds__(kSkNT) skinny_linear_kernel(const SkinnyArgs p, int row_tiles) {
    __shared__ int sidx[PTX_REL_MAX_SETS * PTX_REL_MAX_FRAMES];
    __shared__ float red[kSkNT / 64][kSkFB * kSkMR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rt = blockIdx.x % row_tiles;             // row tile fastest: neighbours share the weight rows in L2
    const int j0 = (blockIdx.x / row_tiles) * kSkFB;
    const int m0 = rt * kSkMR;
    if (p.mode == 1) {
        for (int i = tid; i < PTX_REL_MAX_SETS * PTX_REL_MAX_FRAMES; i += kSkNT)
            sidx[i] = p.idx[i / PTX_REL_MAX_FRAMES][i % PTX_REL_MAX_FRAMES];
        __syncthreads();
    }
    const bool relu_in = (p.flags & PTX_PRO_RELU) != 0;
    // per-row base offsets (elements) and subset ids
    int xoff[kSkMR], rset[kSkMR];
    bool rok[kSkMR];
#pragma unroll
    for (int m = 0; m < kSkMR; ++m) {
        const int row = m0 + m;
        rok[m] = row < p.M;
        const int rr = rok[m] ? row : 0;
        if (p.mode == 1) {
            rset[m] = rr / p.B;
            xoff[m] = (rr % p.B) * p.ldx;
        } else {
            rset[m] = 0;
            xoff[m] = rr * p.ldx;
        }
    }
    const float* wr[kSkFB];
#pragma unroll
    for (int f = 0; f < kSkFB; ++f) wr[f] = p.w + (size_t)min(j0 + f, p.Nout - 1) * p.K;
    float acc[kSkFB * kSkMR];
#pragma unroll
    for (int i = 0; i < kSkFB * kSkMR; ++i) acc[i] = 0.f;

    const int seg_chunks = p.seg_len / 4;
    const size_t set_stride = (size_t)p.B * p.ldx;
    for (int seg = 0; seg < p.n_seg; ++seg) {
        int foff[kSkMR];
#pragma unroll
        for (int m = 0; m < kSkMR; ++m)
            foff[m] = xoff[m] + (p.mode == 1 ? sidx[rset[m] * PTX_REL_MAX_FRAMES + seg] * p.seg_len : seg * p.seg_len);
        const int wcol = seg * p.seg_len;
#pragma unroll 2
        for (int c = tid; c < seg_chunks; c += kSkNT) {
            const int k = c * 4;
            f32x4 wv[kSkFB];
#pragma unroll
            for (int f = 0; f < kSkFB; ++f) wv[f] = *reinterpret_cast<const f32x4*>(wr[f] + wcol + k);
#pragma unroll
            for (int m = 0; m < kSkMR; ++m) {
                f32x4 xv = *reinterpret_cast<const f32x4*>(p.x + foff[m] + k);
                if (p.mode == 2)
                    for (int r = 1; r < p.n_sets; ++r) xv += *reinterpret_cast<const f32x4*>(p.x + r * set_stride + foff[m] + k);
                if (relu_in) {
                    xv.x = fmaxf(xv.x, 0.f); xv.y = fmaxf(xv.y, 0.f);
                    xv.z = fmaxf(xv.z, 0.f); xv.w = fmaxf(xv.w, 0.f);
                }
#pragma unroll
                for (int f = 0; f < kSkFB; ++f) {
                    float a = acc[f * kSkMR + m];
                    a = fmaf(xv.x, wv[f].x, a);
                    a = fmaf(xv.y, wv[f].y, a);
                    a = fmaf(xv.z, wv[f].z, a);
                    a = fmaf(xv.w, wv[f].w, a);
                    acc[f * kSkMR + m] = a;
                }
            }
        }
    }
    // reduce-scatter butterfly over NV = FB*MR values: after log2(NV) halving steps lane l holds the sum
    // (over the lanes that differ from it in the exchanged bits) of value v(l); the remaining lane bits
    // are folded with plain xor-shuffles
    constexpr int NV = kSkFB * kSkMR;                 // 16 or 32
    constexpr int STEPS = NV == 32 ? 5 : 4;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int half = (NV / 2) >> s, off = 32 >> s;
        const bool up = (lane & off) != 0;
#pragma unroll
        for (int i = 0; i < half; ++i) {
            const float send = up ? acc[i] : acc[i + half];
            const float keep = up ? acc[i + half] : acc[i];
            acc[i] = keep + __shfl_xor(send, off, 64);
        }
    }
#pragma unroll
    for (int off = (32 >> STEPS); off > 0; off >>= 1) acc[0] += __shfl_xor(acc[0], off, 64);
    int v = 0;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) v += ((lane >> (5 - s)) & 1) * ((NV / 2) >> s);
    if ((lane & ((64 >> STEPS) - 1)) == 0) red[wave][v] = acc[0];
    __syncthreads();
    if (tid < kSkFB * kSkMR) {
        float sum = red[0][tid];
#pragma unroll
        for (int wv_ = 1; wv_ < kSkNT / 64; ++wv_) sum += red[wv_][tid];
        const int f = tid / kSkMR, m = tid % kSkMR;
        const int j = j0 + f, row = m0 + m;
        if (j < p.Nout && row < p.M) {
            float o = sum + (p.b ? p.bias_scale * p.b[j] : 0.f);
            float* dst = p.y + (size_t)row * p.ldy + j;
            if (p.flags & PTX_EPI_ACCUM) o += *dst;
            if (p.flags & PTX_EPI_RELU) o = fmaxf(o, 0.f);
            *dst = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// in-place row softmax (or 1/cols scaling): ONE WAVEFRONT PER ROW, the row held in registers,
// max / sum reduced with wave shuffles only -- no LDS, no barrier (4 rows per workgroup).
// ---------------------------------------------------------------------------------------------
constexpr int kSoftmaxMaxPerLane = 64;   // rows up to 4096 columns

template <int PER_LANE>
__global__ void __launch_bounds__(256) softmax_rows_kernel(float* __restrict__ x, long long rows, int cols, int ld,
                                                           int scale_only) {
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float* row = x + (size_t)r * ld;
    const int lane = threadIdx.x & 63;
    float v[PER_LANE];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < PER_LANE; ++i) {
        const int c = lane + 64 * i;
        v[i] = (c < cols) ? row[c] : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
    float inv;
    if (scale_only) {
        inv = 1.0f / (float)cols;
    } else {
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < PER_LANE; ++i) {
            const int c = lane + 64 * i;
            v[i] = (c < cols) ? expf(v[i] - mx) : 0.f;
            sum += v[i];
        }
        inv = 1.0f / wave_sum(sum);
    }
#pragma unroll
    for (int i = 0; i < PER_LANE; ++i) {
        const int c = lane + 64 * i;
        if (c < cols) row[c] = v[i] * inv;
        else if (c < ld) row[c] = 0.f;
    }
}

// ---------------------------------------------------------------------------------------------
// concat plumbing and temporal window means (tiny, bandwidth bound)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) copy2d_kernel(const float* __restrict__ x, float* __restrict__ y, size_t total4,
                                                     int cols4, long long ldx, long long ldy) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const size_t r = i / cols4;
        const int q = (int)(i - r * cols4);
        *reinterpret_cast<f32x4*>(y + r * ldy + q * 4) = *reinterpret_cast<const f32x4*>(x + r * ldx + q * 4);
    }
}

__global__ void __launch_bounds__(256) cbn_fold_kernel(const float* __restrict__ gain, const float* __restrict__ bias,
                                                       const float* __restrict__ mean, const float* __restrict__ var,
                                                       float eps, float* __restrict__ scale, float* __restrict__ shift,
                                                       int N, int C, int ld_gain, int ld_bias, int ld_out,
                                                       int plus_one) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C;
    const float g = gain ? gain[(size_t)n * ld_gain + c] : 0.f;
    const float m = plus_one ? 1.f + g : g;
    const float s = m / sqrtf(var[c] + eps);
    scale[(size_t)n * ld_out + c] = s;
    shift[(size_t)n * ld_out + c] = (bias ? bias[(size_t)n * ld_bias + c] : 0.f) - mean[c] * s;
}

__global__ void __launch_bounds__(256) affine_act_upsample_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                  const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, int lds_,
                                                                  size_t total4, int H, int W, int C, int ldx, int ldy,
                                                                  int up, int relu_in) {
    const bool out_f16 = (relu_in & PTX_ACT_OUT_F16) != 0;
    const int relu = relu_in & 0xff;
    const int c4 = (C + 3) / 4;
    const int Ho = H * up, Wo = W * up;
    const bool vec4 = scale && (C % 4 == 0) && (lds_ % 4 == 0) && ((((uintptr_t)scale | (uintptr_t)shift) & 15) == 0);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const int q = (int)(i % c4);
        size_t pos = i / c4;
        const int wo = (int)(pos % Wo);
        size_t r = pos / Wo;
        const int ho = (int)(r % Ho);
        const int n = (int)(r / Ho);
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((size_t)n * H + ho / up) * W + wo / up) * ldx + q * 4);
        float o[4] = {v.x, v.y, v.z, v.w};
        float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
        if (scale) {
            const float* sp = scale + (size_t)n * lds_ + q * 4;
            const float* hp = shift + (size_t)n * lds_ + q * 4;
            if (vec4) {                                   // one 16-byte load each instead of four scalar ones
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(sp), h4 = *reinterpret_cast<const f32x4*>(hp);
                sc[0] = s4.x; sc[1] = s4.y; sc[2] = s4.z; sc[3] = s4.w;
                sh[0] = h4.x; sh[1] = h4.y; sh[2] = h4.z; sh[3] = h4.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (q * 4 + e < C) { sc[e] = sp[e]; sh[e] = hp[e]; }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = q * 4 + e;
            if (c < C) {
                if (scale) o[e] = fmaf(o[e], sc[e], sh[e]);
                if (relu == 1) o[e] = fmaxf(o[e], 0.f);
                else if (relu == 2) o[e] = tanhf(o[e]);
            } else {
                o[e] = 0.f;
            }
        }
        if (out_f16) {
            typedef _Float16 half4 __attribute__((ext_vector_type(4)));
            const half4 h4 = {(_Float16)o[0], (_Float16)o[1], (_Float16)o[2], (_Float16)o[3]};
            *reinterpret_cast<half4*>(reinterpret_cast<_Float16*>(y) + pos * ldy + q * 4) = h4;
        } else {
            f32x4 w4 = {o[0], o[1], o[2], o[3]};
            *reinterpret_cast<f32x4*>(y + pos * ldy + q * 4) = w4;
        }
    }
}

__global__ void __launch_bounds__(256) outer_sum_relu_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             float* __restrict__ f, size_t total, int rows, int cols, int ldf) {
    const float inv = 1.0f / (float)cols;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int j = (int)(e % ldf);
        const size_t r = e / ldf;                    // n * rows + i
        const size_t n = r / rows;
        f[e] = j < cols ? fmaxf(a[r] + b[n * cols + j], 0.f) * inv : 0.f;
    }
}

__global__ void __launch_bounds__(256) window_mean_kernel(const float* __restrict__ x, float* __restrict__ y, size_t total,
                                                          int T, int To, int inner, int k, int stride) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int i = (int)(e % inner);
        size_t r = e / inner;
        const int j = (int)(r % To);
        const size_t o = r / To;
        const float* src = x + (o * T + (size_t)j * stride) * inner + i;
        float acc = 0.f;
        for (int d = 0; d < k; ++d) acc += src[(size_t)d * inner];
        y[e] = acc / (float)k;
    }
}

static unsigned grid_for(size_t work_items) {
    size_t b = (work_items + 255) / 256;
    const size_t cap = (size_t)kNumCU * 8;
    if (b > cap) b = cap;
    if (b == 0) b = 1;
    return (unsigned)b;
}

}  // namespace ptx

using namespace ptx;

extern "C" int ptx_maxpool3d_fwd(const ptx_pool3d_desc* d, const float* x, float* y, ptx_stream_t stream) {
    if (!d || !x || !y) return fail(PTX_ERR_INVALID, "maxpool3d: null pointer");
    if (d->N <= 0 || d->C <= 0 || d->ld < d->C || d->ld % 4 || d->kT <= 0 || d->kH <= 0 || d->kW <= 0 ||
        d->sT <= 0 || d->sH <= 0 || d->sW <= 0 || d->pT < 0 || d->pH < 0 || d->pW < 0 || d->To <= 0 || d->Ho <= 0 ||
        d->Wo <= 0)
        return fail(PTX_ERR_INVALID, "maxpool3d: bad descriptor");
    const int c4 = (d->C + 3) / 4 * 4;
    if (d->ldy < 0 || (d->ldy > 0 && (d->ldy < c4 || d->ldy % 4)))
        return fail(PTX_ERR_INVALID, "maxpool3d: output stride %d must cover C=%d and be a multiple of 4", d->ldy, d->C);
    if (d->flags & ~(PTX_POOL_SAME | PTX_POOL_PAD_ZERO)) return fail(PTX_ERR_INVALID, "maxpool3d: unknown flags");
    if (d->flags & PTX_POOL_SAME) {
        // explicit output extent: every window must start inside the (front-padded) input
        if (d->pT >= d->kT || d->pH >= d->kH || d->pW >= d->kW || (d->To - 1) * d->sT - d->pT >= d->Ti ||
            (d->Ho - 1) * d->sH - d->pH >= d->Hi || (d->Wo - 1) * d->sW - d->pW >= d->Wi)
            return fail(PTX_ERR_INVALID, "maxpool3d: SAME geometry leaves a window without a valid tap");
    } else {
        if (2 * d->pT > d->kT || 2 * d->pH > d->kH || 2 * d->pW > d->kW)
            return fail(PTX_ERR_INVALID, "maxpool3d: padding larger than half the window");
        const int to = (d->Ti + 2 * d->pT - d->kT) / d->sT + 1;
        const int ho = (d->Hi + 2 * d->pH - d->kH) / d->sH + 1;
        const int wo = (d->Wi + 2 * d->pW - d->kW) / d->sW + 1;
        if (to != d->To || ho != d->Ho || wo != d->Wo) return fail(PTX_ERR_INVALID, "maxpool3d: output extent mismatch");
    }
    if (((uintptr_t)x | (uintptr_t)y) & 15) return fail(PTX_ERR_INVALID, "maxpool3d: misaligned pointer");
    const hipStream_t st = (hipStream_t)stream;
    {
        // the ResNet3D stem pool on the rolling kernel when its frames are too big for the XCD-local tile order to pay
        // (kT-frame window above 6 MB: config 2's 112 x 112 x 64 frames, 111.9 -> 95.5 us = 4.8 TB/s; config 3's 56 x 56
        // frames run 40 us on the XCD-local tile kernel and 58 us here -- 25 k threads are too few).  PTX_POOL_ROLL=0 / 1 forces.
        static const int roll_env = getenv("PTX_POOL_ROLL") ? atoi(getenv("PTX_POOL_ROLL")) : -1;
        const bool fits = d->kT == 3 && d->kH == 3 && d->kW == 3 && d->sT == 2 && d->sH == 2 && d->sW == 2 && d->pT == 1 && d->pH == 1 &&
                          d->pW == 1 && !(d->flags & (PTX_POOL_SAME | PTX_POOL_PAD_ZERO)) && d->Ti >= 2;
        const size_t total = (size_t)d->N * cdiv(d->Ho, 2) * cdiv(d->Wo, 2) * (c4 / 4);
        const size_t window = (size_t)d->kT * d->Hi * d->Wi * d->ld * 4;
        if (fits && roll_env != 0 && (roll_env == 1 || window > (6u << 20)) && total < (1ull << 31)) {
            static const int xl_env2 = getenv("PTX_POOL_XCD") ? atoi(getenv("PTX_POOL_XCD")) : 1;
            hipLaunchKernelGGL(maxpool3d_roll_kernel, dim3(grid_for(total)), dim3(256), 0, st, *d, x, y, (unsigned)total, xl_env2 & 1);
            return hip_check(hipGetLastError(), "maxpool3d launch");
        }
    }
    if (d->kW == 3 && d->kH == 3 && d->sH == d->sW && (d->sW == 1 || d->sW == 2) && d->pW < 3 && d->pH < 3) {
        // measured on the config-2/3/4 geometries (scripts/gpu_pool_bench.py): 2 x 2 patches with a whole halo plane in
        // flight win wherever a launch is short of threads (latency bound); the big stride-1 pools (I3D at batch 8)
        // take 4 x 4 patches, one halo row in flight.  PTX_POOL_TILE=44|22 forces one of them.
        static const int tile_env = getenv("PTX_POOL_TILE") ? atoi(getenv("PTX_POOL_TILE")) : 0;
        const size_t total44 = (size_t)d->N * d->To * cdiv(d->Ho, 4) * cdiv(d->Wo, 4) * (c4 / 4);
        const bool big = tile_env ? tile_env == 44 : (d->sW == 1 && total44 >= 400000);
        const int seg = big ? 4 : 2;
        const size_t total = (size_t)d->N * d->To * cdiv(d->Ho, seg) * cdiv(d->Wo, seg) * (c4 / 4);
        if (total < (1ull << 31)) {
            const dim3 g(grid_for(total)), b(256);
            const unsigned n = (unsigned)total;
            // bit 0: XCD-local chunks; bit 1: frames innermost.  Measured per geometry (round 4, scripts/gpu_pool_bench.py,
            // profiles/r04_pool_order_ab.txt): XCD-local chunks pay when the kT-frame window of a 3-D pool fits the 4 MiB L2 --
            // config 3's stem pool 56 -> 41 us (5.7 TB/s), I3D's stride-2 Mixed_4 pool 30 -> 21 us, its small stride-1 SAME
            // pools at 2 clips per GPU 13-18 -> 9-11 us -- and cost elsewhere: config 2's 112 x 112 frames (9.6 MB window:
            // 112 -> 116 us), every kT = 1 pool (the 2-D stems: 44 -> 47 us), stride-1 pools at batch 8 (113 -> 174 us: eight
            // XCDs marching through eight clips in lock-step).  Frames-innermost alone or combined never won on the stem pools.
            // PTX_POOL_XCD=0..3 forces a mode.
            static const int xl_env = getenv("PTX_POOL_XCD") ? atoi(getenv("PTX_POOL_XCD")) : -1;
            const size_t window = (size_t)d->kT * d->Hi * d->Wi * d->ld * 4;
            int xl = 0;
            if (d->kT > 1 && d->To > 1) {
                if (d->sW == 2) xl = window <= (6u << 20) ? 1 : 0;
                else xl = (window <= (3u << 20) && d->N <= 4) ? 1 : 0;
            }
            if (xl_env >= 0) xl = xl_env;
            if (big && d->sW == 1) hipLaunchKernelGGL((maxpool3d_tile_kernel<4, 4, 1, 0>), g, b, 0, st, *d, x, y, n, xl);
            else if (big) hipLaunchKernelGGL((maxpool3d_tile_kernel<4, 4, 2, 0>), g, b, 0, st, *d, x, y, n, xl);
            else if (d->sW == 1) hipLaunchKernelGGL((maxpool3d_tile_kernel<2, 2, 1, 1>), g, b, 0, st, *d, x, y, n, xl);
            else hipLaunchKernelGGL((maxpool3d_tile_kernel<2, 2, 2, 1>), g, b, 0, st, *d, x, y, n, xl);
            return hip_check(hipGetLastError(), "maxpool3d launch");
        }
    }
    if (d->kW == 3 && (d->sW == 1 || d->sW == 2) && d->pW < 3 && d->Wo >= 4) {
        constexpr int WSEG = 4;                               // 8 measured 1.4x slower (fewer threads, same loads in flight)
        const size_t total = (size_t)d->N * d->To * d->Ho * cdiv(d->Wo, WSEG) * (c4 / 4);
        const dim3 g(grid_for(total)), b(256);
        if (d->sW == 2) hipLaunchKernelGGL((maxpool3d_slide_kernel<WSEG, 2>), g, b, 0, st, *d, x, y, total);
        else hipLaunchKernelGGL((maxpool3d_slide_kernel<WSEG, 1>), g, b, 0, st, *d, x, y, total);
        return hip_check(hipGetLastError(), "maxpool3d launch");
    }
    const size_t total4 = (size_t)d->N * d->To * d->Ho * d->Wo * (c4 / 4);
    hipLaunchKernelGGL(maxpool3d_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, *d, x, y, total4);
    return hip_check(hipGetLastError(), "maxpool3d launch");
}

extern "C" int ptx_cbn_fold(const float* gain, const float* bias, const float* mean, const float* var, float eps,
                            float* scale, float* shift, int32_t N, int32_t C, int32_t ld_gain, int32_t ld_bias,
                            int32_t ld_out, int32_t plus_one, ptx_stream_t stream) {
    if (!mean || !var || !scale || !shift) return fail(PTX_ERR_INVALID, "cbn_fold: null pointer");
    if (N <= 0 || C <= 0 || ld_gain < 0 || ld_bias < 0 || ld_out < C || (gain && ld_gain != 0 && ld_gain < C) ||
        (bias && ld_bias != 0 && ld_bias < C))
        return fail(PTX_ERR_INVALID, "cbn_fold: bad extents");
    if (!gain && !plus_one) return fail(PTX_ERR_INVALID, "cbn_fold: gain == NULL needs plus_one (scale 1)");
    hipLaunchKernelGGL(cbn_fold_kernel, dim3((unsigned)cdiv(N * C, 256)), dim3(256), 0, (hipStream_t)stream, gain, bias,
                       mean, var, eps, scale, shift, N, C, ld_gain, ld_bias, ld_out, plus_one);
    return hip_check(hipGetLastError(), "cbn_fold launch");
}

extern "C" int ptx_affine_act_upsample(const float* x, float* y, const float* scale, const float* shift,
                                       int32_t ld_scale, int32_t N, int32_t H, int32_t W, int32_t C, int32_t ldx,
                                       int32_t ldy, int32_t up, int32_t relu, ptx_stream_t stream) {
    if (!x || !y || ((scale == nullptr) != (shift == nullptr)))
        return fail(PTX_ERR_INVALID, "affine_act_upsample: null pointer (scale and shift go together)");
    if ((relu & ~PTX_ACT_OUT_F16) < 0 || (relu & ~PTX_ACT_OUT_F16) > 2)
        return fail(PTX_ERR_INVALID, "affine_act_upsample: act must be 0, 1 or 2 (| PTX_ACT_OUT_F16)");
    if ((relu & PTX_ACT_OUT_F16) && (ldy % 8 || ((uintptr_t)y & 7)))
        return fail(PTX_ERR_INVALID, "affine_act_upsample: fp16 output needs ldy %% 8 == 0");
    const int c4 = (C + 3) / 4 * 4;
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || up < 1 || up > 8 || ldx < c4 || ldy < c4 || ldx % 4 || ldy % 4 ||
        (scale && ld_scale != 0 && ld_scale < C))
        return fail(PTX_ERR_INVALID, "affine_act_upsample: bad extents");
    if (((uintptr_t)x | (uintptr_t)y) & 15) return fail(PTX_ERR_INVALID, "affine_act_upsample: misaligned pointer");
    const size_t total4 = (size_t)N * H * up * W * up * (c4 / 4);
    hipLaunchKernelGGL(affine_act_upsample_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, x, y, scale,
                       shift, ld_scale, total4, H, W, C, ldx, ldy, up, relu);
    return hip_check(hipGetLastError(), "affine_act_upsample launch");
}

extern "C" int ptx_copy2d(const float* x, float* y, int64_t rows, int32_t cols, int64_t ldx, int64_t ldy,
                          ptx_stream_t stream) {
    if (!x || !y) return fail(PTX_ERR_INVALID, "copy2d: null pointer");
    if (rows <= 0 || cols <= 0 || cols % 4 || ldx < cols || ldy < cols || ldx % 4 || ldy % 4)
        return fail(PTX_ERR_INVALID, "copy2d: cols and strides must be positive multiples of 4 covering cols");
    if (((uintptr_t)x | (uintptr_t)y) & 15) return fail(PTX_ERR_INVALID, "copy2d: misaligned pointer");
    const size_t total4 = (size_t)rows * (cols / 4);
    hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, x, y, total4, cols / 4,
                       (long long)ldx, (long long)ldy);
    return hip_check(hipGetLastError(), "copy2d launch");
}

extern "C" int ptx_outer_sum_relu(const float* a, const float* b, float* f, int32_t batch, int32_t rows, int32_t cols,
                                  int32_t ldf, ptx_stream_t stream) {
    if (!a || !b || !f) return fail(PTX_ERR_INVALID, "outer_sum_relu: null pointer");
    if (batch <= 0 || rows <= 0 || cols <= 0 || ldf < cols) return fail(PTX_ERR_INVALID, "outer_sum_relu: bad extents");
    const size_t total = (size_t)batch * rows * ldf;
    hipLaunchKernelGGL(outer_sum_relu_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, a, b, f, total, rows,
                       cols, ldf);
    return hip_check(hipGetLastError(), "outer_sum_relu launch");
}

extern "C" int ptx_window_mean(const float* x, float* y, int32_t outer, int32_t T, int32_t inner, int32_t k,
                               int32_t stride, ptx_stream_t stream) {
    if (!x || !y) return fail(PTX_ERR_INVALID, "window_mean: null pointer");
    if (outer <= 0 || T <= 0 || inner <= 0 || k <= 0 || k > T || stride <= 0)
        return fail(PTX_ERR_INVALID, "window_mean: bad extents (T=%d k=%d stride=%d)", T, k, stride);
    const int To = (T - k) / stride + 1;
    const size_t total = (size_t)outer * To * inner;
    hipLaunchKernelGGL(window_mean_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, total, T, To,
                       inner, k, stride);
    return hip_check(hipGetLastError(), "window_mean launch");
}

extern "C" int ptx_global_avgpool(const float* x, float* y, int32_t N, int32_t C, int64_t S, int32_t ld,
                                  int32_t channels_first, ptx_stream_t stream) {
    if (!x || !y) return fail(PTX_ERR_INVALID, "avgpool: null pointer");
    if (N <= 0 || C <= 0 || S <= 0) return fail(PTX_ERR_INVALID, "avgpool: non-positive extent");
    if (channels_first) {
        const long long NC = (long long)N * C;
        hipLaunchKernelGGL(avgpool_cf_kernel, dim3((unsigned)cdiv64(NC, 4)), dim3(256), 0, (hipStream_t)stream, x, y,
                           NC, (long long)S);
    } else {
        if (ld < C || N > 65535) return fail(PTX_ERR_INVALID, "avgpool: bad ld / N");
        hipLaunchKernelGGL(avgpool_cl_kernel, dim3((unsigned)cdiv(C, 64), (unsigned)N), dim3(256), 0,
                           (hipStream_t)stream, x, y, C, (long long)S, ld);
    }
    return hip_check(hipGetLastError(), "avgpool launch");
}

static int launch_skinny(SkinnyArgs& a, hipStream_t st) {
    const int row_tiles = cdiv(a.M, kSkMR);
    static int force_fb = -1, force_nt = -1;             // PTX_SKINNY="FB,NT": tuning switch
    if (force_fb < 0) {
        force_fb = force_nt = 0;
        if (const char* e = getenv("PTX_SKINNY")) sscanf(e, "%d,%d", &force_fb, &force_nt);
    }
    // measured on MI355X (scripts/gpu_trn_micro.py): 2 features x 256 threads wins for every head-sized
    // problem (1.5 TB/s of weight traffic at K = 16384; the L2 re-reads of the shared input rows bound it);
    // with many row tiles one-wave workgroups give the scheduler more independent work
    int nt = row_tiles >= 4 ? 64 : 256;
    int fb = 2;
    if (force_fb) fb = force_fb;
    if (force_nt) nt = force_nt;
    const long long blocks = (long long)row_tiles * cdiv(a.Nout, fb);
    if (blocks > 0x7fffffffLL) return fail(PTX_ERR_UNSUPPORTED, "linear: grid too large");
    const dim3 grid((unsigned)blocks);
    if (fb == 4 && nt == 256) hipLaunchKernelGGL((skinny_linear_kernel<4, 256>), grid, dim3(256), 0, st, a, row_tiles);
    else if (fb == 4) hipLaunchKernelGGL((skinny_linear_kernel<4, 64>), grid, dim3(64), 0, st, a, row_tiles);
    else if (nt == 256) hipLaunchKernelGGL((skinny_linear_kernel<2, 256>), grid, dim3(256), 0, st, a, row_tiles);
    else hipLaunchKernelGGL((skinny_linear_kernel<2, 64>), grid, dim3(64), 0, st, a, row_tiles);
    return hip_check(hipGetLastError(), "linear launch");
}

extern "C" int ptx_relation_linear_fwd(const ptx_relation_desc* d, const float* x, int32_t ldx, const float* w,
                                       const float* b, float* y, int32_t Nout, int32_t ldy, uint32_t flags,
                                       ptx_stream_t stream) {
    if (!d || !x || !w || !y) return fail(PTX_ERR_INVALID, "relation_linear: null pointer");
    if (d->B <= 0 || d->n_sets <= 0 || d->n_sets > PTX_REL_MAX_SETS || d->n_frames <= 0 ||
        d->n_frames > PTX_REL_MAX_FRAMES || d->frame_len <= 0 || d->frame_len % 4)
        return fail(PTX_ERR_INVALID, "relation_linear: need 1..%d subsets of 1..%d frames, frame_len %% 4 == 0",
                    PTX_REL_MAX_SETS, PTX_REL_MAX_FRAMES);
    if (Nout <= 0 || ldy < Nout || ldx % 4) return fail(PTX_ERR_INVALID, "relation_linear: bad extents");
    int max_idx = 0;
    for (int r = 0; r < d->n_sets; ++r)
        for (int f = 0; f < d->n_frames; ++f) {
            if (d->idx[r][f] < 0) return fail(PTX_ERR_INVALID, "relation_linear: negative frame index");
            max_idx = std::max(max_idx, (int)d->idx[r][f]);
        }
    if ((int64_t)(max_idx + 1) * d->frame_len > ldx)
        return fail(PTX_ERR_INVALID, "relation_linear: frame index %d outside a video of %d floats", max_idx, ldx);
    if (((uintptr_t)x | (uintptr_t)w) & 15) return fail(PTX_ERR_INVALID, "relation_linear: misaligned pointer");
    if ((int64_t)d->B * ldx > 0x7fffffffLL) return fail(PTX_ERR_UNSUPPORTED, "relation_linear: input too large");
    SkinnyArgs a{};
    a.x = x; a.w = w; a.b = b; a.y = y;
    a.M = d->n_sets * d->B; a.K = d->n_frames * d->frame_len; a.Nout = Nout; a.ldx = ldx; a.ldy = ldy;
    a.flags = flags; a.mode = 1; a.B = d->B; a.n_sets = d->n_sets; a.seg_len = d->frame_len; a.n_seg = d->n_frames;
    a.bias_scale = 1.f;
    for (int r = 0; r < PTX_REL_MAX_SETS; ++r)
        for (int f = 0; f < PTX_REL_MAX_FRAMES; ++f) a.idx[r][f] = d->idx[r][f];
    return launch_skinny(a, (hipStream_t)stream);
}

extern "C" int ptx_linear_setsum_fwd(const float* x, const float* w, const float* b, float* y, int32_t B,
                                     int32_t n_sets, int32_t K, int32_t Nout, int32_t ldx, int32_t ldy, uint32_t flags,
                                     ptx_stream_t stream) {
    if (!x || !w || !y) return fail(PTX_ERR_INVALID, "linear_setsum: null pointer");
    if (B <= 0 || n_sets <= 0 || K <= 0 || K % 4 || Nout <= 0 || ldx < K || ldx % 4 || ldy < Nout)
        return fail(PTX_ERR_INVALID, "linear_setsum: bad extents (K and ldx must be multiples of 4)");
    if (((uintptr_t)x | (uintptr_t)w) & 15) return fail(PTX_ERR_INVALID, "linear_setsum: misaligned pointer");
    if ((int64_t)n_sets * B * ldx > 0x7fffffffLL) return fail(PTX_ERR_UNSUPPORTED, "linear_setsum: input too large");
    SkinnyArgs a{};
    a.x = x; a.w = w; a.b = b; a.y = y;
    a.M = B; a.K = K; a.Nout = Nout; a.ldx = ldx; a.ldy = ldy;
    a.flags = flags; a.mode = 2; a.B = B; a.n_sets = n_sets; a.seg_len = K; a.n_seg = 1;
    a.bias_scale = (float)n_sets;
    return launch_skinny(a, (hipStream_t)stream);
}

extern "C" int ptx_linear_fwd(const float* x, const float* w, const float* b, float* y, int32_t M, int32_t K,
                              int32_t Nout, int32_t ldx, int32_t ldy, uint32_t flags, ptx_stream_t stream) {
    if (!x || !w || !y) return fail(PTX_ERR_INVALID, "linear: null pointer");
    if (M <= 0 || K <= 0 || Nout <= 0 || ldx < K || ldy < Nout) return fail(PTX_ERR_INVALID, "linear: bad extents");
    // M >= 32 rows is a GEMM, not a GEMV: the MFMA tiles read the weights once (flags they do not implement stay here)
    if (M >= 32 && K % 4 == 0 && Nout % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && !(flags & ~(uint32_t)PTX_EPI_RELU) &&
        ((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) == 0) && (uint64_t)M * ldx * 4ull < 0x80000000ull &&
        (uint64_t)Nout * K * 4ull < 0x80000000ull && (uint64_t)M * ldy * 4ull < 0x80000000ull)
        return linear_gemm(x, w, b, y, M, K, Nout, ldx, ldy, flags, (hipStream_t)stream);
    if (K % 4 == 0 && ldx % 4 == 0 && ((((uintptr_t)x | (uintptr_t)w) & 15) == 0) && (int64_t)M * ldx <= 0x7fffffffLL) {
        SkinnyArgs a{};
        a.x = x; a.w = w; a.b = b; a.y = y;
        a.M = M; a.K = K; a.Nout = Nout; a.ldx = ldx; a.ldy = ldy;
        a.flags = flags; a.mode = 0; a.B = M; a.n_sets = 1; a.seg_len = K; a.n_seg = 1; a.bias_scale = 1.f;
        return launch_skinny(a, (hipStream_t)stream);
    }
    constexpr int MR = 8;
    if (cdiv(M, MR) > 65535) return fail(PTX_ERR_UNSUPPORTED, "linear: M too large for the small-M kernel");
    dim3 grid((unsigned)cdiv(Nout, 4), (unsigned)cdiv(M, MR));
    hipLaunchKernelGGL(linear_smallm_kernel<MR>, grid, dim3(256), 0, (hipStream_t)stream, x, w, b, y, M, K, Nout, ldx,
                       ldy, flags);
    return hip_check(hipGetLastError(), "linear launch");
}

extern "C" int ptx_softmax_rows(float* x, int64_t rows, int32_t cols, int32_t ld, int32_t scale_only,
                                ptx_stream_t stream) {
    if (!x) return fail(PTX_ERR_INVALID, "softmax: null pointer");
    if (rows <= 0 || cols <= 0 || ld < cols) return fail(PTX_ERR_INVALID, "softmax: bad extents");
    if (ld > 64 * kSoftmaxMaxPerLane)
        return fail(PTX_ERR_UNSUPPORTED, "softmax: rows longer than %d are not supported", 64 * kSoftmaxMaxPerLane);
    if (rows > 4LL * 0x7fffffffLL) return fail(PTX_ERR_INVALID, "softmax: too many rows");
    const dim3 grid((unsigned)cdiv64(rows, 4));
    hipStream_t st = (hipStream_t)stream;
    if (ld <= 64 * 4)
        hipLaunchKernelGGL(softmax_rows_kernel<4>, grid, dim3(256), 0, st, x, (long long)rows, cols, ld, scale_only);
    else if (ld <= 64 * 16)
        hipLaunchKernelGGL(softmax_rows_kernel<16>, grid, dim3(256), 0, st, x, (long long)rows, cols, ld, scale_only);
    else if (ld <= 64 * 32)
        hipLaunchKernelGGL(softmax_rows_kernel<32>, grid, dim3(256), 0, st, x, (long long)rows, cols, ld, scale_only);
    else
        hipLaunchKernelGGL(softmax_rows_kernel<64>, grid, dim3(256), 0, st, x, (long long)rows, cols, ld, scale_only);
    return hip_check(hipGetLastError(), "softmax launch");
}

// Chained convolutions: conv -> BN -> ReLU -> 1x1x1 conv -> BN (-> + residual) -> ReLU in ONE launch
// (ptx_conv3d_chain_fwd).  The first conv's [BM x N1] output tile stays in the workgroup's LDS as the A operand of the
// second GEMM (conv_igemm_kernel<..., CHAIN = true>, conv_igemm_kernel.h): the intermediate tensor never reaches HBM.
//
// What it replaces in the reference:
//   * a bottleneck's tail -- conv2 (3x3x3) -> bn2 -> relu -> conv3 (1x1x1) -> bn3 -> += residual -> relu
//     (resnet3D.py:129-142).  For layer1 of resnet3d50 at 8x3x16x224x224 the 1x1x1 conv sits on the fp32 ridge
//     (AI = 26 FLOP/B): as its own launch it ran at 4.3 TB/s = 60 TF; chained, its HBM traffic
//     (205 MB residual + 205 MB output) overlaps the MFMA-bound 3x3x3 body of the co-resident workgroups;
//   * the pointwise pairs of the (2+1)D networks -- a "1x1x1" SpatioTemporalConv is two pointwise GEMMs through
//     M = floor(Cin Cout / (Cin + Cout)) mid channels with BN + ReLU between them (r2plus1d.py:68-88): 64 -> 51 -> 256,
//     256 -> 51 -> 64 ...: two launches of 15-60 us at 30-60 TF each become one.
// Limits: N1 (the first conv's output channels) <= the tile's BN (32 / 64 / 128: one N tile holds the whole intermediate
// row); dense fp32 convs, fp32 or split (PTX_F16X3_OPERANDS, both convs) operands; the tail is 1x1x1 / unit stride with
// PTX_EPI_RELU | PTX_EPI_RES_ADD; no split-K.
#include "conv_igemm_kernel.h"

namespace ptx {

typedef int (*chain_launch_fn)(const ConvArgs&, dim3, hipStream_t);

template <int BM, int BN, int BK, int WM, int WN, int MT, bool KTAIL, bool REPI, bool X3, int KWR>
static int launch_chain_one(const ConvArgs& a, dim3 grid, hipStream_t st) {
    constexpr size_t lds_tiles = (size_t)2 * ((KWR ? (BM + BM / 4 + 15) / 16 * 16 : BM) + (KWR ? KWR : 1) * BN) * BK * sizeof(float);
    // the parked intermediate tile aliases the two A stages when they are big enough (kernel: kAlias); the row-major
    // epilogue (REPI) adds one [MT][BN / WN + 4] parking block per wave behind everything else.  kw-reuse tiles: the parked
    // tile takes the whole tile area, the tail's two [BN][BK] filter stages follow it.
    constexpr bool alias = BM * BN <= 2 * BM * BK;
    constexpr size_t lds_plain = lds_tiles + (alias ? 0 : (size_t)BM * BN * sizeof(float)) +
                                 (REPI ? (size_t)WM * WN * MT * (BN / WN + 4) * sizeof(float) : 0);
    constexpr size_t lds_kwr = ((size_t)BM * BN + 2 * BN * BK) * sizeof(float);
    constexpr size_t lds = KWR ? (lds_kwr > lds_tiles ? lds_kwr : lds_tiles) : lds_plain;
    auto kern = conv_igemm_kernel<BM, BN, BK, WM, WN, MT, KTAIL, false, true, 2, false, X3, KWR, true, REPI>;
    static bool attr_set[64] = {};   // per device; benign race (idempotent call)
    int dev = 0;
    PTX_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        PTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, st, a);
    return hip_check(hipGetLastError(), "conv_chain launch");
}

template <int BM, int BN, int BK, int WM, int WN, int MT, bool REPI, bool X3 = false, int KWR = 0>
static int launch_chain(const ConvArgs& a, dim3 grid, hipStream_t st) {
    if ((a.kA % BK) || (a.kB % BK)) return launch_chain_one<BM, BN, BK, WM, WN, MT, true, REPI, X3, KWR>(a, grid, st);
    return launch_chain_one<BM, BN, BK, WM, WN, MT, false, REPI, X3, KWR>(a, grid, st);
}

struct ChainConfig {
    int BM, BN, BK, WM, WN, MT;
    const char* name;
    chain_launch_fn launch;
    bool x3;          // split fp32 operands (PTX_F16X3_OPERANDS) in both GEMMs
    int kwr;          // kw-reuse loader of the first conv (3-wide stride-1 filters, BM a whole number of output rows)
};
#define PTX_CHAIN_CFG(BM, BN, BK, WM, WN, MT) \
    { BM, BN, BK, WM, WN, MT, #BM "x" #BN "x" #BK "/" #WM "x" #WN "/m" #MT "/dma/chain", launch_chain<BM, BN, BK, WM, WN, MT, false>, false, 0 }
// the tail's epilogue row-major through LDS: 16-byte residual loads / output stores (conv_igemm_kernel.h, REPI)
#define PTX_CHAIN_CFG_RE(BM, BN, BK, WM, WN, MT) \
    { BM, BN, BK, WM, WN, MT, #BM "x" #BN "x" #BK "/" #WM "x" #WN "/m" #MT "/dma/chain/re", launch_chain<BM, BN, BK, WM, WN, MT, true>, false, 0 }
// split operands (x3): both GEMMs on v_mfma_f32_*_f16 with (hi, lo) half operands; the parked tile stays fp32 and is split
// at fragment-read time exactly like an activation tile staged from HBM
#define PTX_CHAIN_CFG_X3(BM, BN, BK, WM, WN, MT) \
    { BM, BN, BK, WM, WN, MT, #BM "x" #BN "x" #BK "/" #WM "x" #WN "/m" #MT "/dma/chain/x3", launch_chain<BM, BN, BK, WM, WN, MT, false, true>, true, 0 }
#define PTX_CHAIN_CFG_X3RE(BM, BN, BK, WM, WN, MT) \
    { BM, BN, BK, WM, WN, MT, #BM "x" #BN "x" #BK "/" #WM "x" #WN "/m" #MT "/dma/chain/re/x3", launch_chain<BM, BN, BK, WM, WN, MT, true, true>, true, 0 }
// ... with the kw-reuse loader for the first conv (a split-operand 3x3x3 conv2 is bound by its A traffic, not its MFMAs:
// conv_igemm_kernel.h, KWR): the bottleneck tails of a split-operand plan
#define PTX_CHAIN_CFG_KWRX3(BM, BN, BK, WM, WN, MT) \
    { BM, BN, BK, WM, WN, MT, #BM "x" #BN "x" #BK "/" #WM "x" #WN "/m" #MT "/dma/kwr/chain/x3", launch_chain<BM, BN, BK, WM, WN, MT, false, true, 3>, true, 3 }

static const ChainConfig kChain[] = {
    PTX_CHAIN_CFG(64, 64, 32, 2, 2, 32),     // 0  the 3x3x3 -> 1x1x1 bottleneck tail at 64 planes (same tile as the tuned conv2)
    PTX_CHAIN_CFG(64, 64, 16, 2, 2, 32),     // 1  short / ragged K (pointwise pairs: K = 64, 52 ...)
    PTX_CHAIN_CFG(128, 64, 16, 2, 2, 32),    // 2  large M, short K
    PTX_CHAIN_CFG(128, 64, 32, 4, 2, 32),    // 3  8 waves
    PTX_CHAIN_CFG(32, 64, 32, 2, 2, 16),     // 4  smaller M: twice the tiles
    PTX_CHAIN_CFG(64, 128, 32, 2, 2, 32),    // 5  mid widths up to 128 (85, 102, 128 planes)
    PTX_CHAIN_CFG(32, 128, 32, 2, 2, 16),    // 6
    PTX_CHAIN_CFG(64, 32, 32, 2, 2, 16),     // 7  mid width <= 32 (64 -> 32 -> 64)
    PTX_CHAIN_CFG(64, 128, 16, 2, 2, 32),    // 8  mid widths up to 128, short K
    PTX_CHAIN_CFG_RE(64, 64, 32, 2, 2, 32),  // 9
    PTX_CHAIN_CFG_RE(64, 64, 16, 2, 2, 32),  // 10
    PTX_CHAIN_CFG_RE(128, 64, 16, 2, 2, 32), // 11
    PTX_CHAIN_CFG_RE(128, 64, 32, 4, 2, 32), // 12
    PTX_CHAIN_CFG_RE(32, 64, 32, 2, 2, 16),  // 13
    PTX_CHAIN_CFG_RE(64, 128, 32, 2, 2, 32), // 14
    PTX_CHAIN_CFG_RE(32, 128, 32, 2, 2, 16), // 15
    PTX_CHAIN_CFG_RE(64, 32, 32, 2, 2, 16),  // 16
    PTX_CHAIN_CFG_RE(64, 128, 16, 2, 2, 32), // 17
    PTX_CHAIN_CFG_X3(64, 64, 32, 2, 2, 32),
    PTX_CHAIN_CFG_X3(128, 64, 32, 4, 2, 32),
    PTX_CHAIN_CFG_X3(64, 128, 32, 2, 2, 32),
    PTX_CHAIN_CFG_X3(32, 64, 64, 2, 2, 16),
    PTX_CHAIN_CFG_X3(64, 64, 64, 2, 2, 32),
    PTX_CHAIN_CFG_X3RE(64, 64, 32, 2, 2, 32),
    PTX_CHAIN_CFG_X3RE(128, 64, 32, 4, 2, 32),
    PTX_CHAIN_CFG_X3RE(64, 128, 32, 2, 2, 32),
    PTX_CHAIN_CFG_X3RE(32, 64, 64, 2, 2, 16),
    PTX_CHAIN_CFG_X3RE(64, 64, 64, 2, 2, 32),
    PTX_CHAIN_CFG_KWRX3(224, 64, 16, 7, 1, 32),
    PTX_CHAIN_CFG_KWRX3(224, 64, 32, 7, 1, 32),
};
constexpr int kNumChain = sizeof(kChain) / sizeof(kChain[0]);

static int validate_chain(const ptx_conv3d_desc* c, const ptx_conv3d_desc* t) {
    int s = validate_desc(c);
    if (s != PTX_OK) return s;
    s = validate_desc(t);
    if (s != PTX_OK) return s;
    if (c->flags & ~(uint32_t)(PTX_EPI_RELU | PTX_F16X3_OPERANDS))
        return fail(PTX_ERR_UNSUPPORTED, "conv3d_chain: the first conv takes PTX_EPI_RELU (| PTX_F16X3_OPERANDS) only (flags 0x%x)", c->flags);
    if (t->flags & ~(uint32_t)(PTX_EPI_RELU | PTX_EPI_RES_ADD | PTX_F16X3_OPERANDS))
        return fail(PTX_ERR_UNSUPPORTED, "conv3d_chain: the tail takes PTX_EPI_RELU | PTX_EPI_RES_ADD (| PTX_F16X3_OPERANDS) only (flags 0x%x)",
                    t->flags);
    if ((c->flags ^ t->flags) & PTX_F16X3_OPERANDS)
        return fail(PTX_ERR_INVALID, "conv3d_chain: both convs or neither take split operands (PTX_F16X3_OPERANDS)");
    if (c->groups > 1 || t->groups > 1) return fail(PTX_ERR_UNSUPPORTED, "conv3d_chain: dense convs only");
    if (t->kT * t->kH * t->kW != 1 || t->sT != 1 || t->sH != 1 || t->sW != 1 || t->pT || t->pH || t->pW)
        return fail(PTX_ERR_INVALID, "conv3d_chain: the tail must be a unit-stride 1x1x1 conv");
    if (t->N != c->N || t->Ti != c->To || t->Hi != c->Ho || t->Wi != c->Wo || t->To != c->To || t->Ho != c->Ho || t->Wo != c->Wo)
        return fail(PTX_ERR_INVALID, "conv3d_chain: the tail's positions must be the first conv's output positions");
    if (t->Ci != c->Co || t->Kc < (c->Co + 3) / 4 * 4)
        return fail(PTX_ERR_INVALID, "conv3d_chain: the tail's K axis (Ci=%d, Kc=%d) must cover the first conv's %d output channels",
                    t->Ci, t->Kc, c->Co);
    if ((t->flags & PTX_EPI_RES_ADD) && t->ldr < (t->Co + 3) / 4 * 4)
        return fail(PTX_ERR_INVALID, "conv3d_chain: residual stride %d does not cover Co=%d", t->ldr, t->Co);
    return PTX_OK;
}

}  // namespace ptx

using namespace ptx;

extern "C" int ptx_conv3d_chain_num_configs(void) { return kNumChain; }

extern "C" const char* ptx_conv3d_chain_config_name(int config) {
    if (config < 0 || config >= kNumChain) return "invalid";
    return kChain[config].name;
}

extern "C" int ptx_conv3d_chain_supported(const ptx_conv3d_desc* conv, const ptx_conv3d_desc* tail, int config) {
    if (!conv || !tail || config < 0 || config >= kNumChain) return 0;
    if (validate_chain(conv, tail) != PTX_OK) return 0;
    const ChainConfig& c = kChain[config];
    const int n1 = (conv->Co + 3) / 4 * 4;
    if (n1 > c.BN) return 0;                         // the whole intermediate row lives in ONE N tile
    if (c.x3 != ((conv->flags & PTX_F16X3_OPERANDS) != 0)) return 0;     // operand kind of the packed filters = the tile's
    if (c.kwr && (conv->kW != c.kwr || conv->sW != 1 || conv->Wo < 8 || c.BM % conv->Wo ||
                  conv->Wo != conv->Wi + 2 * conv->pW - conv->kW + 1))
        return 0;                                    // kw-reuse: a 3-wide stride-1 filter, the M tile = whole output rows
    return 1;
}

extern "C" int ptx_conv3d_chain_pick_config(const ptx_conv3d_desc* conv, const ptx_conv3d_desc* tail) {
    if (!conv || !tail || validate_chain(conv, tail) != PTX_OK) return -1;
    const int n1 = (conv->Co + 3) / 4 * 4;
    const int64_t M = (int64_t)conv->N * conv->To * conv->Ho * conv->Wo;
    const int K = conv->kT * conv->kH * conv->kW * conv->Kc;
    if (n1 > 128) return -1;
    const bool x3 = (conv->flags & PTX_F16X3_OPERANDS) != 0;
    // defaults resolved BY NAME (inserting / reordering kChain entries cannot remap them); the engine's tuner refines them
    auto named = [](const char* name) -> int {
        for (int i = 0; i < kNumChain; ++i)
            if (!strcmp(kChain[i].name, name)) return i;
        fprintf(stderr, "libptx_amd: default chained tile \"%s\" is not compiled into this build\n", name);
        abort();
        return -1;
    };
    static const int t64x32 = named("64x32x32/2x2/m16/dma/chain"), t32x64 = named("32x64x32/2x2/m16/dma/chain"),
                     t64x64x16 = named("64x64x16/2x2/m32/dma/chain"), t64x64x32 = named("64x64x32/2x2/m32/dma/chain"),
                     t32x128 = named("32x128x32/2x2/m16/dma/chain"), t64x128x16 = named("64x128x16/2x2/m32/dma/chain"),
                     t64x128x32 = named("64x128x32/2x2/m32/dma/chain");
    static const int x64x64 = named("64x64x32/2x2/m32/dma/chain/x3"),
                     x128x64 = named("128x64x32/4x2/m32/dma/chain/x3"), x64x128 = named("64x128x32/2x2/m32/dma/chain/x3");
    static const int xkwr = named("224x64x16/7x1/m32/dma/kwr/chain/x3");
    if (x3 && n1 <= 64 && M >= 32768 && ptx_conv3d_chain_supported(conv, tail, xkwr)) return xkwr;     // 3-wide filters, whole-row tiles
    if (x3) return n1 <= 64 ? (M >= 65536 ? x128x64 : x64x64) : x64x128;   // (the 16x16x32 tiles need BK = 64 > a 32-wide N tile)
    if (n1 <= 32) return t64x32;
    if (n1 <= 64) return M < 32768 ? t32x64 : (K <= 128 || conv->Kc % 32 ? t64x64x16 : t64x64x32);
    return M < 32768 ? t32x128 : (K <= 256 || conv->Kc % 32 ? t64x128x16 : t64x128x32);
}

extern "C" int ptx_conv3d_chain_fwd(const ptx_conv3d_desc* conv, const ptx_conv3d_desc* tail, const float* x, const float* w_packed,
                                    const float* bias, const float* w2_packed, const float* bias2, const float* res, float* y,
                                    int config, ptx_stream_t stream) {
    if (!conv || !tail) return fail(PTX_ERR_INVALID, "conv3d_chain: null descriptor");
    int s = validate_chain(conv, tail);
    if (s != PTX_OK) return s;
    if (!x || !w_packed || !w2_packed || !y) return fail(PTX_ERR_INVALID, "conv3d_chain: null tensor pointer");
    if ((tail->flags & PTX_EPI_RES_ADD) && !res) return fail(PTX_ERR_INVALID, "conv3d_chain: residual flag set but res == NULL");
    if (((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)w2_packed | (uintptr_t)y | (uintptr_t)res) & 15)
        return fail(PTX_ERR_INVALID, "conv3d_chain: pointers must be 16-byte aligned");
    if (config < 0) config = ptx_conv3d_chain_pick_config(conv, tail);
    if (config < 0 || config >= kNumChain || !ptx_conv3d_chain_supported(conv, tail, config))
        return fail(PTX_ERR_UNSUPPORTED, "conv3d_chain: no chained tile holds %d intermediate channels (config %d)", conv->Co, config);
    const ChainConfig& c = kChain[config];
    const ptx_conv3d_desc* d = conv;
    ConvArgs a{};
    a.x = x; a.w = w_packed; a.bias = bias; a.res = res; a.y = y;
    a.N = d->N; a.Ti = d->Ti; a.Hi = d->Hi; a.Wi = d->Wi; a.ldx = d->ldx; a.kA = d->ldx;
    a.To = d->To; a.Ho = d->Ho; a.Wo = d->Wo; a.Co = d->Co; a.ldy = tail->ldy; a.k_live = d->Ci;
    a.kT = d->kT; a.kH = d->kH; a.kW = d->kW; a.sT = d->sT; a.sH = d->sH; a.sW = d->sW;
    a.pT = d->pT; a.pH = d->pH; a.pW = d->pW;
    a.ldw = d->Kc; a.kB = d->Kc; a.w_rows = d->Co_pad; a.w_tap_stride = (long long)d->Co_pad * d->Kc;
    a.M = d->N * d->To * d->Ho * d->Wo;
    a.flags = d->flags & PTX_EPI_RELU;
    a.x3 = (d->flags & PTX_F16X3_OPERANDS) ? 1 : 0;
    {
        const uint64_t xb = (uint64_t)d->N * d->Ti * d->Hi * d->Wi * d->ldx * 4ull;
        const uint64_t wb = (uint64_t)d->kT * d->kH * d->kW * d->Co_pad * d->Kc * 4ull;
        const uint64_t w2b = (uint64_t)tail->Co_pad * tail->Kc * 4ull;
        const uint64_t yb = (uint64_t)a.M * tail->ldy * 4ull;
        const uint64_t rb = (tail->flags & PTX_EPI_RES_ADD) ? (uint64_t)a.M * tail->ldr * 4ull : 0ull;
        if (xb >= 0x80000000ull || wb >= 0x80000000ull || w2b >= 0x80000000ull || yb >= 0x80000000ull || rb >= 0x80000000ull)
            return fail(PTX_ERR_UNSUPPORTED, "conv3d_chain: every tensor of one launch must be < 2 GiB (32-bit buffer offsets); split the batch");
        a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb; a.w2_bytes = (unsigned)w2b;
        a.y_bytes = (unsigned)yb; a.r_bytes = (unsigned)rb;
    }
    a.groups = 1; a.cig = d->Ci; a.cog = d->Co;
    a.pps = d->To * d->Ho * d->Wo;
    a.ldr = tail->ldr;
    a.m_tiles = cdiv(a.M, c.BM);
    a.n_tiles = 1;
    fastdiv_make((unsigned)a.Wo, a.dv_wo);
    fastdiv_make((unsigned)a.Ho, a.dv_ho);
    fastdiv_make((unsigned)a.To, a.dv_to);
    if (c.kwr) fastdiv_make((unsigned)(a.Wo + c.kwr - 1), a.dv_hw);
    a.tiles_per_plane = 0;
    a.ncol = (a.Co + 3) / 4 * 4;
    a.kchunks = cdiv(std::max(a.kA, a.kB), c.BK);
    a.split_k = 1;
    a.unit_pointwise = (a.kT * a.kH * a.kW == 1 && a.sT == 1 && a.sH == 1 && a.sW == 1 && a.pT == 0 && a.pH == 0 && a.pW == 0 &&
                        a.Ti == a.To && a.Hi == a.Ho && a.Wi == a.Wo) ? 1 : 0;
    a.prune_analytic = prune_analytic_ok(a);
    a.w2 = w2_packed; a.bias2 = bias2;
    a.ldw2 = tail->Kc; a.kB2 = tail->Kc; a.w2_rows = tail->Co_pad; a.Co2 = tail->Co; a.ncol2 = (tail->Co + 3) / 4 * 4;
    a.flags2 = tail->flags & (PTX_EPI_RELU | PTX_EPI_RES_ADD);
    if ((int64_t)a.m_tiles > 0x7fffffffLL) return fail(PTX_ERR_INVALID, "conv3d_chain: grid too large");
    return c.launch(a, dim3((unsigned)a.m_tiles, 1, 1), (hipStream_t)stream);
}

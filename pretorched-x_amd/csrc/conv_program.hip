// conv_program.hip -- SEVERAL convolutions in ONE persistent launch (libptx_amd, gfx950).
//
// Why.  The tail of a video ResNet is a chain of small-M convolutions: layer3 / layer4 of resnet3d50 at 8 x 16 x 224^2 have
// M = 3136 / 392 output rows, the (2+1)D bottlenecks of config 3 M = 1568 / 256 -- six GEMMs per bottleneck of 0.1-4 GFLOP
// each (resnet3D.py:125-143, r2plus1d.py:68-88).  As one launch per conv they are bound by what happens AROUND the MFMAs:
// 196 tiles on 256 CUs leave 60 CUs idle for the whole launch, 392 tiles run as two rounds (1.53 -> 2), the smallest launches
// take 7.5 us for 1.6 us of matrix work (profiles/r04_rows_cfg3.txt: 58 launches, 1.17 ms, 21-57 TF).
//
// What.  A conv PROGRAM is an ordered list of convolutions ("stages") whose tiles are put on ONE work queue -- in WAVEFRONT
// order over (stage, clip group): the clips of the batch are cut into up to 8 groups, chunk (stage s, group g) sits on
// diagonal s + g, so while group 0 is in stage s + 1 group 1 is still in stage s and the queue always holds runnable tiles of
// several stages (clips never depend on each other: resnet3D.py:125-143, nonlocalnet.py:157 softmax per sample).  conv_program_kernel is launched once with 1-3 persistent 256-thread workgroups per CU; each
// workgroup repeatedly (1) takes the next queue index with one returning agent-scope atomicAdd, (2) waits until the row
// tiles of the producing stages that its tile reads are complete -- per-(stage, row tile) completion counters, polled by
// one wave with relaxed sc1 loads -- (3) runs the SAME tile body a plain launch would (conv_igemm_tile, conv_igemm_kernel.h)
// with write-through (sc1) output stores and L1-bypassing (sc1) activation / residual loads, (4) drains its stores and bumps
// the completion counter of its row tile.  There is no grid barrier: a workgroup that runs out of tiles of stage s starts
// on stage s + 1 wherever the rows it needs are done, so the partial last round of every stage overlaps the next stage's
// first, and stages that do not depend on each other (a bottleneck's shortcut conv and its conv1 -> conv2 chain) interleave.
// Split-K stages keep one fp32 partial slab per split IN THEIR OWN workspace region (stages overlap in time) and a ticket
// per tile; the last arriver sums the slabs in split order -- the order of splitk_reduce_kernel, so the result is
// bit-identical to the unfused launches -- applies bias / residual / ReLU and publishes.
//
// Progress.  A queue index's dependencies are tiles of EARLIER stages, i.e. smaller queue indices; indices are handed out in
// order, so the smallest unfinished index is always held by a running workgroup whose dependencies are complete: the
// launch finishes for any grid size and any residency -- nothing here needs co-residency, unlike a grid barrier.  Every spin
// is bounded all the same (PTX_PROG_SPIN_LIMIT polls, ~seconds): on expiry the workgroup sets the error word, every other
// workgroup sees it and leaves; ptx_conv_program_error reports it.
//
// Visibility (cdna_hip_programming.md Guideline 16, R1): payload stores carry sc1 (write-through), every storing wave
// drains (s_waitcnt vmcnt(0)), the workgroup meets, ONE lane bumps the counter with a relaxed agent-scope atomic; the
// consumer polls that word relaxed and then reads the payload with sc1 loads.  Correctness never depends on which XCD a
// workgroup runs on.  All polled words are zeroed by a memset node ahead of EVERY launch (ptx_conv_program_fwd).
#include "conv_igemm_kernel.h"
#include <vector>

namespace ptx {

constexpr int kProgMaxDeps = 4;
constexpr int kProgMaxStages = 256;
constexpr int kProgMaxGroups = 8;
constexpr int kProgCtrlHead = 16;          // ctrl words [0] queue head, [1] error code, [2..4] who waited for whom; counters from word 16
constexpr int kCoh = 16;                   // sc1

struct ProgDep {
    int stage;        // producing stage
    int kind;         // 0: this conv's input x (rows through stride / halo), 1: same rows (residual), 2: strided second source x2
    int bm;           // rows per row tile of the producer
    int mtiles;       // its row tiles
    int done_off;     // ctrl word of its first row-tile counter
    int target;       // completions per row tile = its N tiles
};

struct ProgStage {
    ConvArgs a;
    int cfg;          // tile shape (kProgTiles index)
    int bm, bn;
    int items;
    int done_off;     // ctrl word of this stage's first row-tile counter
    int tick_off;     // ctrl word of its first split-K ticket (one per tile), -1 without split-K
    int halo_lo, halo_hi;
    int clip_out, clip_in, clip_x2;      // rows of ONE clip in the output / input / second-source tensors
    int ndeps;
    ProgDep deps[kProgMaxDeps];
};

struct ProgChunk { int stage, mt_begin, items, group; };      // a run of a stage's row tiles: the clips of one group

struct ProgTileShape { int BM, BN, BK, WM, WN, NSTAGE; const char* name; };
static const ProgTileShape kProgTiles[] = {
    {32, 64, 64, 2, 2, 2, "32x64x64/2x2/m16/dma/re"},
    {32, 128, 32, 2, 2, 2, "32x128x32/2x2/m16/dma/re"},
    {32, 64, 32, 2, 2, 2, "32x64x32/2x2/m16/dma/re"},
    {32, 128, 32, 2, 2, 3, "32x128x32/2x2/m16/dma3/re"},
    {32, 64, 64, 2, 2, 3, "32x64x64/2x2/m16/dma3/re"},
};
constexpr int kNumProgTiles = sizeof(kProgTiles) / sizeof(kProgTiles[0]);
constexpr int kProgLdsCtrlBytes = 64;
static int prog_tile_lds(const ProgTileShape& t) {
    const int tiles = t.NSTAGE * (t.BM + t.BN) * t.BK * 4;
    const int epi = t.WM * t.WN * 16 * (t.BN / t.WN + 4) * 4;         // the row-major epilogue parks one 16-row block per wave
    return tiles > epi ? tiles : epi;
}

typedef const __attribute__((address_space(4))) ConvArgs ProgArgs;       // the stage table, read as constant memory

__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// input row (position index of the tensor a conv reads) under the centre tap of output row m
__device__ __forceinline__ int prog_in_row(int m, int To, int Ho, int Wo, int Ti, int Hi, int Wi, int sT, int sH, int sW) {
    const int wo = m % Wo;
    int t = m / Wo;
    const int ho = t % Ho;
    t /= Ho;
    const int to = t % To;
    const int n = t / To;
    return ((n * Ti + to * sT) * Hi + ho * sH) * Wi + wo * sW;
}

#define PTX_PROG_TILE(BM, BN, BK, WM, WN, NS) \
    conv_igemm_tile<BM, BN, BK, WM, WN, 16, true, false, true, NS, false, false, 0, false, true, kCoh, ProgArgs>

__global__ void __launch_bounds__(256) conv_program_kernel(const ProgStage* __restrict__ stages, const ProgChunk* __restrict__ chunks,
                                                          const int* __restrict__ chunk_begin, const int total_items, unsigned* ctrl,
                                                          const unsigned spin_limit, unsigned long long* trace, const int lds_ctrl_floats) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // `trace` (diagnostic launches only, ptx_conv_program_trace_fwd; NULL otherwise): 8 x u64 per queue item, written by
    // thread 0 -- the 100 MHz wall clock when the item was [0] taken, [1] cleared to run, [2] computed and drained,
    // [3] published; [4] CU id | workgroup << 32; [5] stage | tile << 32; [6] 1 + split slice, bit 32 = last arriver
    // workgroup control words behind the tile image: [0] queue index, [2] dependency wait ok, [3] last split arriver
    typedef __attribute__((address_space(3))) int lds_int;
    lds_int* const lc = (lds_int*)(smem + lds_ctrl_floats);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned* const err = ctrl + 1;
    // Every loop-control decision below is made on a value that was broadcast through LDS and read back with
    // readfirstlane: scalar branches, the same path for all four waves, so every wave meets every barrier.
    int c = 0;                                             // chunk of the last item: queue indices only grow
    for (;;) {
        // One queue index at a time, taken when the workgroup is free.  (Taking the next index early, to hide the atomic's
        // round trip under the tile, was measured 1.5-2.7x SLOWER: an index held by a busy workgroup is a tile nobody
        // else may run, and its consumers wait for it -- profiles/r05_program_probe.txt.)
        if (tid == 0) lc[0] = (int)__hip_atomic_fetch_add(ctrl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int item = __builtin_amdgcn_readfirstlane(lc[0]);
        if (item >= total_items) break;
        unsigned long long* const tr = trace ? trace + (size_t)item * 8 : nullptr;
        if (tr && tid == 0) tr[0] = wall_clock64();
        while (item >= chunk_begin[c + 1]) ++c;
        const ProgChunk ch = chunks[c];
        const int s = ch.stage;
        const ProgStage* const S = stages + s;
        const ProgArgs& p = *(ProgArgs*)(&S->a);           // constant address space: scalar loads, nothing to keep alive
        const int li = item - chunk_begin[c];
        const int split = p.split_k;
        const int zs = li % split;
        const int tile = ch.mt_begin * p.n_tiles + li / split;
        const int m_tile = tile / p.n_tiles;
        const int bm = S->bm;
        // ---- wait for the producers' row tiles this tile reads (one wave polls, relaxed, backing off)
        if (wave == 0) {
            const int m0 = m_tile * bm, m1 = min(m0 + bm, p.M) - 1;
            const int n0 = m0 / S->clip_out, n1 = m1 / S->clip_out;       // clips this tile's rows belong to
            int ok = 1;
            const int nd = S->ndeps;
            for (int d = 0; d < nd; ++d) {
                const ProgDep dep = S->deps[d];
                int lo, hi;
                if (dep.kind == 0) {
                    // rows under the filter footprint, clamped to the clips the tile belongs to (a tap never leaves its clip)
                    lo = max(prog_in_row(m0, p.To, p.Ho, p.Wo, p.Ti, p.Hi, p.Wi, p.sT, p.sH, p.sW) - S->halo_lo, n0 * S->clip_in);
                    hi = min(prog_in_row(m1, p.To, p.Ho, p.Wo, p.Ti, p.Hi, p.Wi, p.sT, p.sH, p.sW) + S->halo_hi, (n1 + 1) * S->clip_in - 1);
                } else if (dep.kind == 1) {
                    lo = m0;
                    hi = m1;
                } else {
                    lo = prog_in_row(m0, p.To, p.Ho, p.Wo, p.T2, p.H2, p.W2, p.s2T, p.s2H, p.s2W);
                    hi = prog_in_row(m1, p.To, p.Ho, p.Wo, p.T2, p.H2, p.W2, p.s2T, p.s2H, p.s2W);
                }
                const int t_lo = max(lo, 0) / dep.bm, t_hi = min(hi / dep.bm, dep.mtiles - 1);
                const unsigned* cnt = ctrl + dep.done_off;
                for (int base = t_lo; base <= t_hi; base += 64) {
                    const int idx = base + lane;
                    for (unsigned spins = 0; ok; ++spins) {
                        const unsigned v = idx <= t_hi ? ld_relaxed(cnt + idx) : (unsigned)dep.target;
                        if (__all(v >= (unsigned)dep.target)) break;
                        if ((spins & 15u) == 15u && ld_relaxed(err) != 0u) ok = 0;            // somebody else gave up
                        if (spins >= spin_limit) {
                            if (lane == 0) {
                                __hip_atomic_store(ctrl + 2, (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                __hip_atomic_store(ctrl + 3, (unsigned)item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                __hip_atomic_store(ctrl + 4, (unsigned)dep.stage, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                            ok = 0;
                        }
                        // a waiting workgroup shares its CU with working ones: poll rarely (polling-cost, MI355X_MICROARCH.md)
                        if (spins < 8u) __builtin_amdgcn_s_sleep(8);
                        else __builtin_amdgcn_s_sleep(32);
                    }
                }
            }
            if (lane == 0) lc[2] = ok;
        }
        __syncthreads();
        if (!__builtin_amdgcn_readfirstlane(lc[2])) break;
        if (tr && tid == 0) {
            tr[1] = wall_clock64();
            tr[4] = (unsigned long long)__smid() | ((unsigned long long)blockIdx.x << 32);
            tr[5] = (unsigned long long)(unsigned)s | ((unsigned long long)(unsigned)tile << 32);
        }
        // ---- the tile
        switch (S->cfg) {
            case 0: PTX_PROG_TILE(32, 64, 64, 2, 2, 2)(p, tile, 0, zs, smem); break;
            case 1: PTX_PROG_TILE(32, 128, 32, 2, 2, 2)(p, tile, 0, zs, smem); break;
            case 2: PTX_PROG_TILE(32, 64, 32, 2, 2, 2)(p, tile, 0, zs, smem); break;
            case 3: PTX_PROG_TILE(32, 128, 32, 2, 2, 3)(p, tile, 0, zs, smem); break;
            default: PTX_PROG_TILE(32, 64, 64, 2, 2, 3)(p, tile, 0, zs, smem); break;
        }
        // ---- publish: every storing wave drains its write-through stores, then one lane counts
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tr && tid == 0) tr[2] = wall_clock64();
        int publish = 1;
        if (split > 1) {
            if (tid == 0) {
                const unsigned t = __hip_atomic_fetch_add(ctrl + S->tick_off + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lc[3] = t == (unsigned)split - 1u ? 1 : 0;
            }
            __syncthreads();
            publish = __builtin_amdgcn_readfirstlane(lc[3]);
            if (publish) {
                splitk_reduce_tile(p, bm, S->bn, tile, tid, 256);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
        }
        if (publish && tid == 0)
            __hip_atomic_fetch_add(ctrl + S->done_off + m_tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tr && tid == 0) {
            tr[3] = wall_clock64();
            tr[6] = (unsigned long long)(1 + zs) | ((unsigned long long)(split > 1 && publish) << 32);
        }
    }
}

static bool overlaps(const void* a, size_t na, const void* b, size_t nb) {
    const uintptr_t a0 = (uintptr_t)a, b0 = (uintptr_t)b;
    return a && b && na && nb && a0 < b0 + nb && b0 < a0 + na;
}

struct HostStage {
    ConvArgs a;
    int cfg, split;
    size_t x_span, x2_span, res_span, y_span;      // bytes
};

struct ProgPlan {
    std::vector<ProgStage> stages;
    std::vector<ProgChunk> chunks;
    std::vector<int> chunk_begin;
    int groups = 1, clips_per_group = 1, lds_bytes = 0;
    size_t slab_bytes = 0;
};

// host twin of the kernel's dependency range: producer row tiles [t_lo, t_hi] that row tile `mt` of stage P reads through `dep`
static void dep_tiles(const ProgStage& P, const ProgDep& dep, int mt, int& t_lo, int& t_hi) {
    const ConvArgs& p = P.a;
    auto in_row = [&](int m, int Ti, int Hi, int Wi, int sT, int sH, int sW) {
        const int wo = m % p.Wo;
        int t = m / p.Wo;
        const int ho = t % p.Ho;
        t /= p.Ho;
        const int to = t % p.To, n = t / p.To;
        return ((n * Ti + to * sT) * Hi + ho * sH) * Wi + wo * sW;
    };
    const int m0 = mt * P.bm, m1 = std::min(m0 + P.bm, p.M) - 1;
    const int n0 = m0 / P.clip_out, n1 = m1 / P.clip_out;
    int lo, hi;
    if (dep.kind == 0) {
        lo = std::max(in_row(m0, p.Ti, p.Hi, p.Wi, p.sT, p.sH, p.sW) - P.halo_lo, n0 * P.clip_in);
        hi = std::min(in_row(m1, p.Ti, p.Hi, p.Wi, p.sT, p.sH, p.sW) + P.halo_hi, (n1 + 1) * P.clip_in - 1);
    } else if (dep.kind == 1) {
        lo = m0;
        hi = m1;
    } else {
        lo = in_row(m0, p.T2, p.H2, p.W2, p.s2T, p.s2H, p.s2W);
        hi = in_row(m1, p.T2, p.H2, p.W2, p.s2T, p.s2H, p.s2W);
    }
    t_lo = std::max(lo, 0) / dep.bm;
    t_hi = std::min(hi / dep.bm, dep.mtiles - 1);
}

// The queue: chunks (stage, clip group) in wavefront order -- diagonal stage + group ascending, and within a diagonal the
// HIGHER group first, because a row tile that straddles two clips belongs to the group of its first row and reads the
// producer's rows of the next group, which sit on the same diagonal.  `groups` == 1 is plain stage order.
static void build_queue(ProgPlan& pl, int groups, int cpg) {
    const int n = (int)pl.stages.size();
    pl.chunks.clear();
    for (int d = 0; d < n + groups - 1; ++d)
        for (int g = groups - 1; g >= 0; --g) {
            const int j = d - g;
            if (j < 0 || j >= n) continue;
            const ProgStage& P = pl.stages[(size_t)j];
            const int64_t r0 = (int64_t)g * cpg * P.clip_out, r1 = std::min<int64_t>((int64_t)(g + 1) * cpg * P.clip_out, P.a.M);
            const int mt0 = (int)cdiv64(r0, P.bm), mt1 = (int)cdiv64(r1, P.bm);       // row tiles whose FIRST row lies in the group
            if (mt1 <= mt0) continue;
            pl.chunks.push_back(ProgChunk{j, mt0, (mt1 - mt0) * P.a.n_tiles * P.a.split_k, g});
        }
    pl.chunk_begin.assign(pl.chunks.size() + 1, 0);
    for (size_t i = 0; i < pl.chunks.size(); ++i) pl.chunk_begin[i + 1] = pl.chunk_begin[i] + pl.chunks[i].items;
    pl.groups = groups;
    pl.clips_per_group = cpg;
}

// every row tile's producers must sit EARLIER in the queue (the progress argument of the file header)
static bool queue_is_topological(const ProgPlan& pl) {
    const int n = (int)pl.stages.size();
    std::vector<std::vector<int>> pos((size_t)n);              // queue position (chunk index) of every row tile of every stage
    for (int j = 0; j < n; ++j) pos[(size_t)j].assign((size_t)pl.stages[(size_t)j].a.m_tiles, -1);
    for (size_t c = 0; c < pl.chunks.size(); ++c) {
        const ProgChunk& ch = pl.chunks[c];
        const ProgStage& P = pl.stages[(size_t)ch.stage];
        const int nmt = ch.items / (P.a.n_tiles * P.a.split_k);
        for (int t = 0; t < nmt; ++t) pos[(size_t)ch.stage][(size_t)(ch.mt_begin + t)] = (int)c;
    }
    for (int j = 0; j < n; ++j) {
        const ProgStage& P = pl.stages[(size_t)j];
        for (int mt = 0; mt < P.a.m_tiles; ++mt) {
            if (pos[(size_t)j][(size_t)mt] < 0) return false;                       // a row tile nobody queued
            for (int d = 0; d < P.ndeps; ++d) {
                int lo, hi;
                dep_tiles(P, P.deps[d], mt, lo, hi);
                for (int t = lo; t <= hi; ++t)
                    if (pos[(size_t)P.deps[d].stage][(size_t)t] >= pos[(size_t)j][(size_t)mt]) return false;
            }
        }
    }
    return true;
}

// Shared front end of plan / build: arguments of every stage, tile + split choices, dependencies, the queue.
static int prog_prepare(const ptx_conv_stage* st, int n, ProgPlan& pl, ptx_conv_program_info* info) {
    if (!st || n <= 0 || n > kProgMaxStages) return fail(PTX_ERR_INVALID, "conv_program: 1..%d stages", kProgMaxStages);
    std::vector<HostStage> hs((size_t)n);
    for (int i = 0; i < n; ++i) {
        const ptx_conv3d_desc* d = &st[i].desc;
        const unsigned allowed = PTX_EPI_RELU | PTX_EPI_RES_ADD | PTX_SPLITK_FUSED;
        if (d->flags & ~allowed)
            return fail(PTX_ERR_UNSUPPORTED, "conv_program: stage %d: only fp32 convs with bias / ReLU / same-shape residual (flags 0x%x)", i, d->flags);
        if (d->groups > 1) return fail(PTX_ERR_UNSUPPORTED, "conv_program: stage %d: grouped convs keep their own launch", i);
        // the row-tile dependency range [in_row(m0) - halo_lo, in_row(m1) + halo_hi] (prog_tile_deps / dep_tiles) takes the
        // first and last output row of a tile as the extremes of what it reads: true when the input row of an output is
        // monotone in the output's raster index, i.e. the padding stays inside the filter's half width and a step of one
        // output row / frame never moves BACKWARDS in the input raster.  Anything else keeps its own launch.
        if (d->pT < 0 || d->pH < 0 || d->pW < 0 || d->pT > (d->kT - 1 + 1) / 2 || d->pH > (d->kH - 1 + 1) / 2 || d->pW > (d->kW - 1 + 1) / 2 ||
            (int64_t)(d->Wo - 1) * d->sW > (int64_t)d->sH * d->Wi || (int64_t)(d->Ho - 1) * d->sH * d->Wi > (int64_t)d->sT * d->Hi * d->Wi)
            return fail(PTX_ERR_UNSUPPORTED, "conv_program: stage %d: geometry outside the monotone row order the tile dependencies assume "
                        "(padding beyond the filter half width, or an output row / frame step that moves backwards in the input)", i);
        ptx_conv3d_desc dd = *d;
        dd.flags &= ~PTX_SPLITK_FUSED;
        HostStage& h = hs[(size_t)i];
        int s = make_conv_args(&dd, st[i].x, st[i].x2, st[i].w_packed, st[i].bias, st[i].res, st[i].y, nullptr, h.a);
        if (s != PTX_OK) return s;
        h.cfg = st[i].tile;
        if (h.cfg >= kNumProgTiles) return fail(PTX_ERR_INVALID, "conv_program: stage %d: tile %d out of range", i, h.cfg);
        const int ncol = (d->Co + 3) / 4 * 4;
        const int K = std::max(h.a.kA, h.a.kB);
        if (h.cfg < 0) {
            // defaults (the tuner's picks on the small-M problems of layer3 / layer4): 64-deep K steps for long K, 128-wide
            // N tiles when that still leaves a round of tiles, 32-deep steps for short K tails
            const int mt = cdiv(h.a.M, 32);
            if (K % 64 && K < 256) h.cfg = 2;
            else if (ncol % 128 == 0 && mt * (ncol / 128) >= kNumCU) h.cfg = 1;
            else h.cfg = 0;
        }
        const ProgTileShape& t = kProgTiles[h.cfg];
        int split = st[i].split_k;
        const int tiles = cdiv(h.a.M, t.BM) * cdiv(ncol, t.BN);
        const int steps = d->kT * d->kH * d->kW * (cdiv(h.a.kA, t.BK) + (h.a.dual ? cdiv(h.a.kA2, t.BK) : 0));
        if (split <= 0) {
            // enough queue items per stage to keep `target` workgroups per CU busy while the stage is the frontier, as long
            // as a split keeps `min_steps` k-steps (PTX_PROG_TARGET_ITEMS / PTX_PROG_MIN_STEPS: tuning knobs, read per plan)
            const char* e1 = getenv("PTX_PROG_TARGET_ITEMS");
            const char* e2 = getenv("PTX_PROG_MIN_STEPS");
            const int target = e1 ? std::max(1, atoi(e1)) : 2 * kNumCU;
            const int min_steps = e2 ? std::max(1, atoi(e2)) : 8;
            split = 1;
            if (tiles < target) {
                split = cdiv(target, tiles);
                if (split > 8) split = 8;
                while (split > 1 && steps / split < min_steps) --split;
            }
        }
        s = finalize_conv_args(h.a, t.BM, t.BN, t.BK, 0, false, split, 1);
        if (s != PTX_OK) return s;
        h.split = h.a.split_k;
        h.a.tiles_per_plane = 0;
        h.x_span = (size_t)h.a.x_bytes;
        h.x2_span = h.a.dual ? (size_t)h.a.x2_bytes : 0;
        h.res_span = (h.a.flags & PTX_EPI_RES_ADD) ? (size_t)h.a.M * h.a.ldr * 4 : 0;
        h.y_span = (size_t)h.a.M * h.a.ldy * 4;
        if ((int64_t)h.a.m_tiles * h.a.n_tiles * h.split > (1 << 24)) return fail(PTX_ERR_UNSUPPORTED, "conv_program: stage %d is not a small-M conv", i);
    }
    // ---- dependencies: the latest earlier stage(s) whose output a stage reads.  A stage that would overwrite what an
    // earlier stage reads or writes (buffer reuse inside one program) cannot be ordered by read-after-write counters alone.
    pl.stages.assign((size_t)n, ProgStage{});
    int done_words = 0, tick_words = 0;
    size_t slab_bytes = 0;
    int lds = 0;
    for (int j = 0; j < n; ++j) {
        HostStage& h = hs[(size_t)j];
        ProgStage& P = pl.stages[(size_t)j];
        const ProgTileShape& t = kProgTiles[h.cfg];
        lds = std::max(lds, prog_tile_lds(t));
        P.a = h.a;
        P.cfg = h.cfg;
        P.bm = t.BM;
        P.bn = t.BN;
        P.items = h.a.m_tiles * h.a.n_tiles * h.split;
        P.done_off = kProgCtrlHead + done_words;
        done_words += h.a.m_tiles;
        P.halo_lo = (h.a.pT * h.a.Hi + h.a.pH) * h.a.Wi + h.a.pW;
        P.halo_hi = ((h.a.kT - 1 - h.a.pT) * h.a.Hi + (h.a.kH - 1 - h.a.pH)) * h.a.Wi + (h.a.kW - 1 - h.a.pW);
        if (P.halo_lo < 0) P.halo_lo = 0;
        if (P.halo_hi < 0) P.halo_hi = 0;
        P.clip_out = h.a.To * h.a.Ho * h.a.Wo;
        P.clip_in = h.a.Ti * h.a.Hi * h.a.Wi;
        P.clip_x2 = h.a.dual ? h.a.T2 * h.a.H2 * h.a.W2 : 0;
        P.ndeps = 0;
        for (int i = 0; i < j; ++i) {
            const HostStage& e = hs[(size_t)i];
            if (overlaps(h.a.y, h.y_span, e.a.y, e.y_span) || overlaps(h.a.y, h.y_span, e.a.x, e.x_span) ||
                overlaps(h.a.y, h.y_span, e.a.x2, e.x2_span) || overlaps(h.a.y, h.y_span, e.a.res, e.res_span))
                return fail(PTX_ERR_UNSUPPORTED, "conv_program: stage %d writes a buffer stage %d reads or writes (every stage needs its own output)", j, i);
        }
        struct Src { const float* p; size_t span; int kind; int ld; };
        const Src srcs[3] = {{h.a.x, h.x_span, 0, h.a.ldx}, {h.a.res, h.res_span, 1, h.a.ldr}, {h.a.x2, h.x2_span, 2, h.a.ldx2}};
        for (const Src& sc : srcs) {
            if (!sc.p || !sc.span) continue;
            for (int i = j - 1; i >= 0; --i) {
                const HostStage& e = hs[(size_t)i];
                if (!overlaps(sc.p, sc.span, e.a.y, e.y_span)) continue;
                // the consumer must index the producer's rows as the producer wrote them: same row stride, starting inside row 0
                const ptrdiff_t off = (const char*)sc.p - (const char*)e.a.y;
                if (sc.ld != e.a.ldy || off < 0 || off >= (ptrdiff_t)e.a.ldy * 4)
                    return fail(PTX_ERR_UNSUPPORTED, "conv_program: stage %d reads stage %d's output through another row layout", j, i);
                if (P.ndeps == kProgMaxDeps) return fail(PTX_ERR_UNSUPPORTED, "conv_program: stage %d has more than %d producers", j, kProgMaxDeps);
                const int rows = sc.kind == 0 ? h.a.N * P.clip_in : sc.kind == 1 ? h.a.M : h.a.N * P.clip_x2;
                if (rows != e.a.M || e.a.N != h.a.N)
                    return fail(PTX_ERR_UNSUPPORTED, "conv_program: stage %d reads %d rows of stage %d's %d", j, rows, i, e.a.M);
                ProgDep& dp = P.deps[P.ndeps++];
                dp.stage = i;
                dp.kind = sc.kind;
                dp.bm = kProgTiles[e.cfg].BM;
                dp.mtiles = e.a.m_tiles;
                dp.done_off = pl.stages[(size_t)i].done_off;
                dp.target = e.a.n_tiles;
                // keep looking: a channel-concatenated tensor has several producers
            }
        }
    }
    for (int j = 0; j < n; ++j) {
        HostStage& h = hs[(size_t)j];
        ProgStage& P = pl.stages[(size_t)j];
        P.tick_off = -1;
        if (h.split > 1) {
            P.tick_off = kProgCtrlHead + done_words + tick_words;
            tick_words += h.a.m_tiles * h.a.n_tiles;
            slab_bytes += (((size_t)h.split * h.a.M * h.a.ncol * 4) + 255) / 256 * 256;
        }
    }
    // ---- the queue: clip groups when every stage sees the same clips (PTX_PROG_GROUPS: 0 / 1 = plain stage order)
    {
        const int N = hs[0].a.N;
        bool same = true;
        int min_clip = 1 << 30;
        for (int j = 0; j < n; ++j) {
            same = same && hs[(size_t)j].a.N == N;
            min_clip = std::min(min_clip, pl.stages[(size_t)j].clip_out);
        }
        const char* eg = getenv("PTX_PROG_GROUPS");
        int want = eg ? atoi(eg) : kProgMaxGroups;
        if (want > kProgMaxGroups) want = kProgMaxGroups;
        int groups = 1, cpg = N;
        if (same && want > 1 && N > 1) {
            cpg = cdiv(N, want);
            // a row tile spans at most two groups: a group holds at least one tile's worth of rows in every stage
            while (cpg < N && (int64_t)cpg * min_clip < 112) ++cpg;
            groups = cdiv(N, cpg);
        }
        build_queue(pl, groups, cpg);
        if (groups > 1 && !queue_is_topological(pl)) build_queue(pl, 1, N);
        if (!queue_is_topological(pl)) return fail(PTX_ERR_UNSUPPORTED, "conv_program: the stages do not form a forward chain");
    }
    pl.lds_bytes = lds + kProgLdsCtrlBytes;
    pl.slab_bytes = slab_bytes;
    if (info) {
        info->n_stages = n;
        info->total_items = pl.chunk_begin.back();
        info->ctrl_words = (kProgCtrlHead + done_words + tick_words + 63) / 64 * 64;
        info->lds_bytes = pl.lds_bytes;
        info->n_chunks = (int32_t)pl.chunks.size();
        info->image_bytes = (uint64_t)n * sizeof(ProgStage) + pl.chunks.size() * sizeof(ProgChunk) + (pl.chunks.size() + 1) * sizeof(int);
        info->image_bytes = (info->image_bytes + 15) / 16 * 16;
        info->workspace_bytes = (uint64_t)info->ctrl_words * 4 + slab_bytes;
        int split_stages = 0;
        for (int j = 0; j < n; ++j) split_stages += hs[(size_t)j].split > 1;
        info->launches_replaced = n + split_stages;
    }
    return PTX_OK;
}

}  // namespace ptx

using namespace ptx;

extern "C" int ptx_conv_program_num_tiles(void) { return kNumProgTiles; }

extern "C" const char* ptx_conv_program_tile_name(int tile) {
    return (tile >= 0 && tile < kNumProgTiles) ? kProgTiles[tile].name : "";
}

extern "C" int ptx_conv_program_plan(const ptx_conv_stage* stages, int32_t n, ptx_conv_program_info* info) {
    if (!info) return fail(PTX_ERR_INVALID, "conv_program: info == NULL");
    ProgPlan pl;
    return prog_prepare(stages, n, pl, info);
}

extern "C" int ptx_conv_program_describe(const ptx_conv_stage* stages, int32_t n, char* text, size_t text_bytes) {
    if (!text || !text_bytes) return fail(PTX_ERR_INVALID, "conv_program_describe: no buffer");
    ProgPlan pl;
    ptx_conv_program_info info{};
    const int s = prog_prepare(stages, n, pl, &info);
    if (s != PTX_OK) return s;
    size_t pos = 0;
    auto put = [&](const char* fmt, auto... a) {
        if (pos < text_bytes) {
            const int k = snprintf(text + pos, text_bytes - pos, fmt, a...);
            if (k > 0) pos += (size_t)k;
        }
    };
    put("items %d ctrl_words %d workspace %llu groups %d clips_per_group %d chunks %d lds %d\n", info.total_items, info.ctrl_words,
        (unsigned long long)info.workspace_bytes, pl.groups, pl.clips_per_group, (int)pl.chunks.size(), pl.lds_bytes);
    for (int j = 0; j < n; ++j) {
        const ProgStage& P = pl.stages[(size_t)j];
        put("stage %d tile %s split %d m_tiles %d n_tiles %d items %d halo %d %d deps", j, kProgTiles[P.cfg].name, P.a.split_k,
            P.a.m_tiles, P.a.n_tiles, P.items, P.halo_lo, P.halo_hi);
        for (int d = 0; d < P.ndeps; ++d) put(" %d:%s", P.deps[d].stage, P.deps[d].kind == 0 ? "x" : P.deps[d].kind == 1 ? "res" : "x2");
        put("%s", "\n");
    }
    put("%s", "queue");
    for (const ProgChunk& c : pl.chunks) put(" %d.%d", c.stage, c.group);
    put("%s", "\n");
    if (pos >= text_bytes) return fail(PTX_ERR_INVALID, "conv_program_describe: %zu bytes do not hold the description", text_bytes);
    return PTX_OK;
}

extern "C" int ptx_conv_program_build(const ptx_conv_stage* stages, int32_t n, void* workspace, size_t workspace_bytes,
                                      void* image_host, size_t image_bytes, ptx_conv_program_info* info) {
    if (!info || !image_host) return fail(PTX_ERR_INVALID, "conv_program: null info / image");
    ProgPlan pl;
    int s = prog_prepare(stages, n, pl, info);
    if (s != PTX_OK) return s;
    if (image_bytes < info->image_bytes) return fail(PTX_ERR_INVALID, "conv_program: image buffer of %zu bytes, need %llu", image_bytes, (unsigned long long)info->image_bytes);
    if (!workspace || ((uintptr_t)workspace & 255) || workspace_bytes < info->workspace_bytes)
        return fail(PTX_ERR_WORKSPACE, "conv_program: needs %llu bytes of 256-byte aligned workspace, got %zu",
                    (unsigned long long)info->workspace_bytes, workspace_bytes);
    char* slabs = static_cast<char*>(workspace) + (size_t)info->ctrl_words * 4;
    for (int j = 0; j < n; ++j) {
        ProgStage& P = pl.stages[(size_t)j];
        if (P.a.split_k > 1) {
            P.a.partial = reinterpret_cast<float*>(slabs);
            slabs += (((size_t)P.a.split_k * P.a.M * P.a.ncol * 4) + 255) / 256 * 256;
        }
    }
    std::memset(image_host, 0, (size_t)info->image_bytes);
    char* o = static_cast<char*>(image_host);
    std::memcpy(o, pl.stages.data(), (size_t)n * sizeof(ProgStage));
    o += (size_t)n * sizeof(ProgStage);
    std::memcpy(o, pl.chunks.data(), pl.chunks.size() * sizeof(ProgChunk));
    o += pl.chunks.size() * sizeof(ProgChunk);
    std::memcpy(o, pl.chunk_begin.data(), pl.chunk_begin.size() * sizeof(int));
    return PTX_OK;
}

static int program_launch(const ptx_conv_program_info* info, const void* image_dev, void* workspace, int32_t wgs_per_cu,
                          unsigned long long* trace, ptx_stream_t stream) {
    if (!info || !image_dev || !workspace) return fail(PTX_ERR_INVALID, "conv_program: null argument");
    if (info->n_stages <= 0 || info->n_stages > kProgMaxStages || info->total_items <= 0 || info->n_chunks <= 0 ||
        info->lds_bytes < kProgLdsCtrlBytes || info->lds_bytes > 160 * 1024)
        return fail(PTX_ERR_INVALID, "conv_program: info does not describe a built program");
    if (wgs_per_cu <= 0) wgs_per_cu = 2;
    const int fit = std::max(1, (160 * 1024) / (int)info->lds_bytes);           // workgroups the LDS of a CU holds
    if (wgs_per_cu > fit) wgs_per_cu = fit;
    if (wgs_per_cu > 3) wgs_per_cu = 3;                                          // 126 VGPRs + 24 AGPRs: three waves per SIMD
    hipStream_t st = (hipStream_t)stream;
    static unsigned spin_limit = 0;
    if (!spin_limit) {
        const char* e = getenv("PTX_PROG_SPIN_LIMIT");
        spin_limit = e ? (unsigned)strtoul(e, nullptr, 10) : 2000000u;
        if (!spin_limit) spin_limit = 1;
    }
    auto kern = conv_program_kernel;
    static int attr_set[64] = {};            // largest dynamic LDS size requested so far, per device (benign race: idempotent)
    int dev = 0;
    PTX_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || attr_set[dev] < info->lds_bytes) {
        PTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (dev >= 0 && dev < 64) attr_set[dev] = 160 * 1024;
    }
    PTX_HIP(hipMemsetAsync(workspace, 0, (size_t)info->ctrl_words * 4, st));
    const char* img = static_cast<const char*>(image_dev);
    const ProgStage* stages = reinterpret_cast<const ProgStage*>(img);
    const ProgChunk* chunks = reinterpret_cast<const ProgChunk*>(img + (size_t)info->n_stages * sizeof(ProgStage));
    const int* chunk_begin = reinterpret_cast<const int*>(img + (size_t)info->n_stages * sizeof(ProgStage) + (size_t)info->n_chunks * sizeof(ProgChunk));
    int grid = wgs_per_cu * kNumCU;
    if (grid > info->total_items) grid = info->total_items;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), (size_t)info->lds_bytes, st, stages, chunks, chunk_begin,
                       (int)info->total_items, static_cast<unsigned*>(workspace), spin_limit, trace,
                       (int)((info->lds_bytes - kProgLdsCtrlBytes) / 4));
    return hip_check(hipGetLastError(), "conv_program launch");
}

extern "C" int ptx_conv_program_fwd(const ptx_conv_program_info* info, const void* image_dev, void* workspace,
                                    int32_t wgs_per_cu, ptx_stream_t stream) {
    return program_launch(info, image_dev, workspace, wgs_per_cu, nullptr, stream);
}

extern "C" int ptx_conv_program_trace_fwd(const ptx_conv_program_info* info, const void* image_dev, void* workspace,
                                          int32_t wgs_per_cu, void* trace, size_t trace_bytes, ptx_stream_t stream) {
    if (!info || !trace || ((uintptr_t)trace & 7) || trace_bytes < (size_t)info->total_items * 64)
        return fail(PTX_ERR_INVALID, "conv_program_trace: needs an 8-byte aligned buffer of total_items x 64 bytes");
    return program_launch(info, image_dev, workspace, wgs_per_cu, static_cast<unsigned long long*>(trace), stream);
}

extern "C" int ptx_conv_program_error(const void* workspace, int32_t* code4, ptx_stream_t stream) {
    if (!workspace || !code4) return fail(PTX_ERR_INVALID, "conv_program_error: null argument");
    unsigned w[8] = {};
    PTX_HIP(hipMemcpyAsync(w, workspace, sizeof(w), hipMemcpyDeviceToHost, (hipStream_t)stream));
    PTX_HIP(hipStreamSynchronize((hipStream_t)stream));
    code4[0] = (int32_t)w[1];      // 0 = ok, 1 = a dependency wait ran out of polls
    code4[1] = (int32_t)w[2];      // waiting stage
    code4[2] = (int32_t)w[3];      // its queue index
    code4[3] = (int32_t)w[4];      // the producing stage it waited for
    return PTX_OK;
}

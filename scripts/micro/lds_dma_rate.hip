// What does ONE CU sustain when it pulls an L2-resident operand panel into LDS with `buffer_load ... lds` (the staging of every
// DMA tile of libptx_amd), and what when the same bytes go straight to registers?  Round-6 question behind the small-M
// launches (layer3 / layer4: ~one 4-wave workgroup per CU, 32 x 64 x 64 k-steps of 24 KiB): their k-step takes 0.75 us
// where the MFMAs need 0.43, deeper DMA rings change nothing (profiles/r06_smallm_probe_deep_rings.txt) -- is the
// per-CU DMA RATE the bound?
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/lds_dma_rate.hip -o scripts/micro/lds_dma_rate && scripts/micro/lds_dma_rate
// A workgroup (256 threads) streams `kib` KiB of a panel (re-read `reps` times: L2 / MALL resident) in 1-KiB-per-wave
// instructions; rows of `row_bytes` as a tile row is (64 floats = 256 B: each lane's 16 bytes come from its own row).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// mode 0: 16-byte LDS-DMA, lane-linear destination, contiguous source (a weight panel: [rows][64 floats], a wave covers 4 rows)
// mode 1: 16-byte LDS-DMA, each lane its own row (an activation tile: 16 lanes share a 256-byte row)
// mode 2: 16-byte loads to REGISTERS, contiguous source
// mode 3: 4-byte LDS-DMA, contiguous source
template <int MODE>
__global__ void __launch_bounds__(256) k_stream(const float* __restrict__ src, float* __restrict__ sink, unsigned panel_bytes, int per_wg_kib,
                                                int reps) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, panel_bytes, 0x00020000);
    // this workgroup's window of the panel (workgroups of one launch read DIFFERENT windows round-robin, all of them L2-resident)
    const unsigned win = (unsigned)per_wg_kib * 1024u;
    const unsigned base = (unsigned)(((size_t)blockIdx.x * win) % (panel_bytes - win + 1)) & ~1023u;
    f4 accv = {0.f, 0.f, 0.f, 0.f};
    const int n_instr = per_wg_kib / 4;                              // 1 KiB per wave-instruction, 4 waves
    for (int r = 0; r < reps; ++r) {
        for (int i = 0; i < n_instr; ++i) {
            const unsigned blk = base + (unsigned)(i * 4 + wave) * 1024u;
            unsigned off = blk + (unsigned)lane * 16u;
            if (MODE == 1) {
                // row-strided: lane l reads 16 bytes of row (i*16 + wave*4 + l/16) at column (l%16)*16 of a 256-byte row -- as tiles do
                off = base + (unsigned)(((i * 16 + wave * 4 + (lane >> 4)) * 256 + (lane & 15) * 16) % win);
            }
            if (MODE == 0 || MODE == 1)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + ((i & 7) * 4 + wave) * 256), 16, off, 0, 0, 0);
            else if (MODE == 3) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + ((i & 7) * 4 + wave) * 256 + q * 64), 4,
                                                             blk + (unsigned)(q * 256 + lane * 4), 0, 0, 0);
            } else
                accv += __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
            if ((i & 7) == 7) {
                if (MODE != 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // eight 1-KiB pieces per wave stay in flight
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (MODE != 2) accv.x = smem[tid];
    if (accv.x == 12345.678f) sink[0] = accv.x + accv.y + accv.z + accv.w;
}

template <int MODE>
static void run(const char* name, const float* src, float* sink, unsigned panel_bytes, int wgs, int kib, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k_stream<MODE>, dim3(wgs), dim3(256), 32 * 1024, 0, src, sink, panel_bytes, kib, reps);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_stream<MODE>, dim3(wgs), dim3(256), 32 * 1024, 0, src, sink, panel_bytes, kib, reps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double bytes = (double)wgs * kib * 1024.0 * reps;
    const double per_cu = bytes / (ms * 1e-3) / (wgs < 256 ? wgs : 256);
    printf("%-46s wgs %-4d %8.1f us   %6.2f TB/s chip   %6.1f GB/s per CU = %5.1f B/clk at 2.4 GHz\n", name, wgs, ms * 1e3, bytes / ms / 1e9,
           per_cu / 1e9, per_cu / 2.4e9);
}

int main() {
    const unsigned panel = 8u << 20;                                  // 8 MiB: L2 / MALL resident
    float *src, *sink;
    hipMalloc(&src, panel); hipMalloc(&sink, 64);
    hipMemset(src, 0, panel);
    const int kib = 1024, reps = 16;                                  // 16 MiB streamed per workgroup
    for (int wgs : {196, 256, 512, 768}) {
        run<0>("LDS-DMA 16 B, contiguous (weight panel)", src, sink, panel, wgs, kib, reps);
        run<1>("LDS-DMA 16 B, 256-byte rows (activation tile)", src, sink, panel, wgs, kib, reps);
        run<3>("LDS-DMA 4 B, contiguous", src, sink, panel, wgs, kib, reps);
        run<2>("loads to registers 16 B, contiguous", src, sink, panel, wgs, kib, reps);
    }
    return 0;
}

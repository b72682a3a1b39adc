// What the fp16 matrix cores SUSTAIN (not part of the product): a register-only loop of v_mfma_f32_32x32x16_f16 -- four
// independent accumulators per wave, operands fixed in registers -- with zero, constant and random operands, for short and
// long launches.  Everything the split-operand (x3) and fp16 paths do is priced against the 2.5 PFLOP/s dense figure; this
// is the yardstick for how much of it a kernel that does NOTHING but MFMAs gets on the box at hand.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_f16_peak.hip -o scripts/micro/mfma_f16_peak && scripts/micro/mfma_f16_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k(float* out, const _Float16* src, int iters) {
    half8 a[2], b[2];
    for (int i = 0; i < 2; ++i)
        for (int e = 0; e < 8; ++e) {
            a[i][e] = src[(threadIdx.x * 16 + i * 8 + e) & 4095];
            b[i][e] = src[(threadIdx.x * 16 + i * 8 + e + 2048) & 4095];
        }
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[1], acc[3], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
}

// the same loop on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: 64 cycles, 4096 FLOP): is the 157.3 TF figure data dependent?
template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k32(float* out, const _Float16* src, int iters) {
    float a[2], b[2];
    for (int i = 0; i < 2; ++i) { a[i] = (float)src[(threadIdx.x * 2 + i) & 4095]; b[i] = (float)src[(threadIdx.x * 2 + i + 2048) & 4095]; }
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[1], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[0], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc[3], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
}

template <int WAVES>
void run32(const char* data, const _Float16* src, int blocks_per_cu, int iters, float* out) {
    const int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k32<WAVES><<<blocks, 64 * WAVES>>>(out, src, 64);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k32<WAVES><<<blocks, 64 * WAVES>>>(out, src, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * WAVES * iters * 16.0 * 4096.0;
    printf("fp32 32x32x2  %-8s waves/WG=%d WG/CU=%d iters=%-6d %8.3f ms  %7.1f TF  (%4.1f %% of 157.3)\n", data, WAVES, blocks_per_cu, iters, ms,
           flop / ms / 1e9, flop / ms / 1e9 / 1.573);
}

template <int WAVES>
void run(const char* data, const _Float16* src, int blocks_per_cu, int iters, float* out) {
    const int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<WAVES><<<blocks, 64 * WAVES>>>(out, src, 64);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<WAVES><<<blocks, 64 * WAVES>>>(out, src, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * WAVES * iters * 16.0 * 32768.0;
    printf("%-8s waves/WG=%d WG/CU=%d iters=%-6d %8.3f ms  %7.1f TF  (%4.1f %% of 2500)\n", data, WAVES, blocks_per_cu, iters, ms,
           flop / ms / 1e9, flop / ms / 1e9 / 25.0);
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 512 * 4);
    _Float16 h[4096];
    _Float16* d[3];
    const char* names[3] = {"zeros", "ones", "random"};
    for (int v = 0; v < 3; ++v) {
        for (int i = 0; i < 4096; ++i) h[i] = v == 0 ? (_Float16)0.f : v == 1 ? (_Float16)1.f : (_Float16)((rand() % 2001 - 1000) / 1000.0f);
        hipMalloc(&d[v], sizeof(h));
        hipMemcpy(d[v], h, sizeof(h), hipMemcpyHostToDevice);
    }
    for (int iters : {500, 4000, 32000})          // ~0.05, 0.4 and 3 ms of pure MFMA at the nominal rate (4 waves per CU)
        for (int v = 0; v < 3; ++v) {
            run<4>(names[v], d[v], 1, iters, out);
            run<4>(names[v], d[v], 2, iters, out);
        }
    for (int v = 0; v < 3; ++v) run<7>(names[v], d[v], 1, 4000, out);      // the split-operand stem's 7 waves per CU
    for (int iters : {2000, 16000})
        for (int v = 0; v < 3; ++v) { run32<4>(names[v], d[v], 1, iters, out); run32<4>(names[v], d[v], 2, iters, out); }
    return 0;
}

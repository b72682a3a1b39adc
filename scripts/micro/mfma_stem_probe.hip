// What keeps the fp32 direct stem at 73 % MFMA-busy?  The stem's step loop (8 waves, 2x2 tiles of v_mfma_f32_32x32x2_f32
// per wave, 44 MFMAs per step, one barrier per step) rebuilt feature by feature (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_stem_probe.hip -o scripts/micro/mfma_stem_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// F bits: 1 = distinct operands per MFMA (2x2 tile), 2 = operands from LDS (ds_read_b32), 4 = 2 VALU per MFMA,
//         8 = barrier per 44-MFMA step, 16 = global -> LDS DMA (2 x 1 KiB per wave per step), 32 = 11 k-pairs read up front
template <int F, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k(float* out, const float* src, int steps, float seed) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64 * WAVES) lds[i] = seed + (i & 63);
    __syncthreads();
    f32x16 acc[2][2];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a >> 1][a & 1][r] = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a0 = seed + lane, a1 = seed * 2 + lane, b0 = seed * 3 + lane, b1 = seed * 5 + lane;
    int junk = threadIdx.x;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 1 << 24, 0x00020000);
    for (int s = 0; s < steps; ++s) {
        if (F & 8) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
        if (F & 16) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(lds + 4096 + wave * 256), 16, (unsigned)(((s * 977 + blockIdx.x * 64 + lane) & 65535) * 16), 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(lds + 6144 + wave * 256), 16, (unsigned)(((s * 577 + blockIdx.x * 64 + lane) & 65535) * 16), 0, 0, 0);
        }
        const float* A = lds + ((s & 3) * 64) + (lane & 31) * 2 + (lane >> 5);
        const float* B = lds + 2048 + ((s & 3) * 128) + lane;
#pragma unroll
        for (int j = 0; j < 11; ++j) {
            if (F & 2) {
                a0 = A[j * 2]; a1 = A[j * 2 + 700];
                b0 = B[j * 128 - 64 * (lane >> 5) * 0]; b1 = B[j * 128 + 32];
            }
            if (F & 4) { junk = junk * 3 + s; junk = junk ^ (junk >> 3); }
            if (F & 1) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            } else {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[1][1], 0, 0, 0);
            }
        }
    }
    float t = junk;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) t += acc[a >> 1][a & 1][r];
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = t;
}

template <int F, int WAVES>
void run(const char* name, int blocks_per_cu, float* out, const float* src) {
    const int steps = 400, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<F, WAVES><<<blocks, 64 * WAVES>>>(out, src, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<F, WAVES><<<blocks, 64 * WAVES>>>(out, src, steps, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * WAVES * steps * 44.0 * 4096.0;
    printf("%-64s waves/WG=%d WG/CU=%d %8.3f ms %7.1f TF  %6.3f us/step\n", name, WAVES, blocks_per_cu, ms, flop / ms / 1e9,
           ms * 1e3 / steps / blocks_per_cu);
}

int main() {
    float *out, *src;
    hipMalloc(&out, 256 * 8 * 512 * 4);
    hipMalloc(&src, 1 << 24);
    hipMemset(src, 0, 1 << 24);
    run<0, 4>("same operands, regs", 2, out, src);
    run<0, 8>("same operands, regs", 1, out, src);
    run<1, 8>("2x2 tile, regs", 1, out, src);
    run<1 | 8, 8>("2x2 tile, regs, barrier/step", 1, out, src);
    run<1 | 2, 8>("2x2 tile, LDS operands", 1, out, src);
    run<1 | 2 | 8, 8>("2x2 tile, LDS operands, barrier/step", 1, out, src);
    run<1 | 2 | 4 | 8, 8>("2x2 tile, LDS operands, VALU, barrier/step", 1, out, src);
    run<1 | 2 | 4 | 8 | 16, 8>("2x2 tile, LDS operands, VALU, barrier/step, DMA", 1, out, src);
    run<1 | 2 | 8 | 16, 8>("2x2 tile, LDS operands, barrier/step, DMA", 1, out, src);
    run<1 | 2 | 8, 4>("2x2 tile, LDS operands, barrier/step", 2, out, src);
    run<1 | 2 | 8, 4>("2x2 tile, LDS operands, barrier/step", 1, out, src);
    run<1 | 2 | 8, 16>("2x2 tile, LDS operands, barrier/step", 1, out, src);
    return 0;
}

// Speed-of-light calibration for the fp32 MFMA loop shapes used by conv_igemm (not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int MODE>   // MODE 0: registers only; 1: + ds_read_b128 operands; 2: + some VALU
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed + i;
    __syncthreads();
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float av = seed + threadIdx.x, bv = seed * 2 + threadIdx.x;
    int lane = threadIdx.x & 63;
    int junk = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 1) {
            f32x4 a4 = *reinterpret_cast<const f32x4*>(&lds[((it & 7) * 256 + lane * 4) & 4095]);
            f32x4 b4 = *reinterpret_cast<const f32x4*>(&lds[((it & 7) * 256 + 2048 + lane * 4) & 4095]);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[r], b4[r], acc[a], 0, 0, 0);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
        }
        if (MODE >= 2) {
#pragma unroll
            for (int v = 0; v < 8; ++v) junk = junk * 3 + it;
        }
    }
    float s = junk;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int MODE>
void run(const char* name, int blocks_per_cu, float* out) {
    int iters = 20000 / NACC;
    int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, MODE><<<blocks, 256>>>(out, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC, MODE><<<blocks, 256>>>(out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * 4 * iters * 4.0 * NACC * 4096.0;
    printf("%-34s blocks/CU=%d  %8.3f ms  %7.1f TF\n", name, blocks_per_cu, ms, flop / ms / 1e9);
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<4, 0>("regs only, 4 acc", 1, out);
    run<4, 0>("regs only, 4 acc", 2, out);
    run<1, 0>("regs only, 1 acc", 1, out);
    run<1, 0>("regs only, 1 acc", 4, out);
    run<4, 1>("ds_read_b128 operands, 4 acc", 1, out);
    run<4, 1>("ds_read_b128 operands, 4 acc", 2, out);
    run<1, 1>("ds_read_b128 operands, 1 acc", 1, out);
    run<1, 1>("ds_read_b128 operands, 1 acc", 4, out);
    run<2, 1>("ds_read_b128 operands, 2 acc", 2, out);
    run<1, 2>("ds_read + 8 VALU/4 MFMA, 1 acc", 4, out);
    run<4, 2>("ds_read + 8 VALU/16 MFMA, 4 acc", 1, out);
    return 0;
}

// HBM streaming ceilings for the access MIXES of the path's bandwidth-bound kernels (MI355X): what can a plain float4
// grid-stride kernel sustain for copy (1R:1W), residual add (2R:1W), the conv3 + residual epilogue mix (A read once:
// 0.25R + 1R + 1W), max-pool-like 8R:1W?  Rows of 256 floats as in layer1 of config 2 (M = 200704).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/hbm_mix.hip -o /tmp/hbm_mix && /tmp/hbm_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_copy(const f4* __restrict__ a, f4* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = a[i];
}
__global__ void __launch_bounds__(256) k_add(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        f4 v = a[i] + b[i];
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        y[i] = v;
    }
}
// the conv3 + residual mix: a quarter-width tensor (64 of 256 channels) is read and broadcast, the residual read, y written
__global__ void __launch_bounds__(256) k_conv3mix(const f4* __restrict__ a, const f4* __restrict__ r, f4* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t row = i >> 6, q = i & 63;                        // 64 float4 per 256-channel row
        f4 v = a[row * 16 + (q & 15)] + r[i];
        y[i] = v;
    }
}
__global__ void __launch_bounds__(256) k_read(const f4* __restrict__ a, f4* __restrict__ y, size_t n) {
    f4 s = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += a[i];
    if (s.x == 12345.678f) y[0] = s;
}
__global__ void __launch_bounds__(256) k_write(f4* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = f4{1.f, 2.f, 3.f, 4.f};
}

int main() {
    const size_t rows = 200704, n = rows * 64;                        // float4 elements of a [rows][256] tensor (205 MB)
    f4 *a, *b, *y;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&y, n * 16);
    hipMemset(a, 0, n * 16); hipMemset(b, 0, n * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {2048, 8192, 32768}) {
        auto timeit = [&](const char* name, double bytes, auto launch) {
            for (int i = 0; i < 3; ++i) launch();
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
            printf("%-34s blocks %-6d %8.1f us  %7.2f TB/s\n", name, blocks, ms * 1e3, bytes / ms / 1e9);
        };
        timeit("copy 1R:1W (410 MB)", 2.0 * n * 16, [&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, y, n); });
        timeit("relu(a + b) 2R:1W (615 MB)", 3.0 * n * 16, [&] { hipLaunchKernelGGL(k_add, dim3(blocks), dim3(256), 0, 0, a, b, y, n); });
        timeit("conv3 mix 1.25R:1W (461 MB)", 2.25 * n * 16, [&] { hipLaunchKernelGGL(k_conv3mix, dim3(blocks), dim3(256), 0, 0, a, b, y, n); });
        timeit("read only (205 MB)", 1.0 * n * 16, [&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, y, n); });
        timeit("write only (205 MB)", 1.0 * n * 16, [&] { hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, y, n); });
    }
    return 0;
}

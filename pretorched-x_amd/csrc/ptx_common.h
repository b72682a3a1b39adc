// Shared helpers for libptx_amd.so (gfx950 only; no portability layer on purpose).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "../../include/ptx_amd.h"

namespace ptx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kNumCU = 256;   // MI355X
constexpr int kNumXCD = 8;

char* last_error_buf();   // thread-local, 512 bytes

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

inline int hip_check(hipError_t e, const char* what) {
    if (e == hipSuccess) return PTX_OK;
    return fail(PTX_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

#define PTX_HIP(expr)                                            \
    do {                                                         \
        int _s = ::ptx::hip_check((expr), #expr);                \
        if (_s != PTX_OK) return _s;                             \
    } while (0)

// Observed dispatch: workgroup b runs on XCD b % 8 (MI355X_MICROARCH "Workgroup dispatch").
// Remap so each XCD owns a contiguous chunk of the logical tile list (neighbouring tiles share
// input halos / weight panels -> L2 hits).  Bijective for any grid size; speed-only assumption.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid % kNumXCD;
    const int q = nwg / kNumXCD, r = nwg % kNumXCD;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + bid / kNumXCD;
}

// conv_igemm.hip: a Linear layer on the implicit-GEMM MFMA tiles (M >= 32 rows)
int linear_gemm(const float* x, const float* w, const float* b, float* y, int M, int K, int Nout, int ldx, int ldy,
                unsigned flags, hipStream_t st);

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace ptx

// rcp_sweep.hip -- TEST translation unit (not part of libcppflow_hip.so): the exhaustive proof behind the collision stage's
// reciprocal.  Built by cppflow_amd/build.py:build_test_kernels() into tests/native/libcppf_testkernels.so with the library's own
// flags; includes the product's device header so that the function under test IS the product's rcp_rn.
//
// cppf_test_rcp_sweep compares rcp_rn (csrc/lmik_device.h: v_rcp_f32 + one Newton step) with the correctly rounded 1 / x on `count`
// consecutive fp32 BIT PATTERNS starting at `first` and adds, per biased exponent of x (0..255), the number of patterns on which
// the two differ in bits to mismatches[256] (DEVICE pointer, uint64, caller-zeroed).  NaN results on both sides count as equal.
// The oracle spells rcp_rn as `1 / x`, so the masks are bit-exact across the two only if this stays 0 over the range the kernels
// use (2^-100 <= |x| < 2^126).
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "../../cppflow_amd/csrc/lmik_device.h"

namespace {
__global__ __launch_bounds__(256) void rcp_sweep_kernel(uint64_t first, uint64_t count, unsigned long long* mism) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
        const uint32_t bits = (uint32_t)(first + i);
        const float x = __uint_as_float(bits);
        const float want = fabsf(x) >= 0x1p-100f ? 1.0f / x : 0.f;  // IEEE division (-fhip-fp32-correctly-rounded-divide-sqrt)
        const float got = cppf::rcp_rn(x);
        const bool both_nan = want != want && got != got;
        if (!both_nan && __float_as_uint(want) != __float_as_uint(got)) atomicAdd(&mism[(bits >> 23) & 0xffu], 1ull);
    }
}
}  // namespace

extern "C" int cppf_test_rcp_sweep(int device, uint64_t first, uint64_t count, uint64_t* mismatches, void* stream) {
    if (!mismatches || first > (1ull << 32) || count > (1ull << 32) - first) return -1;
    if (count == 0) return 0;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) return -2;
    if (prev != device && hipSetDevice(device) != hipSuccess) return -2;
    hipLaunchKernelGGL(rcp_sweep_kernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, first, count,
                       reinterpret_cast<unsigned long long*>(mismatches));
    const hipError_t e = hipGetLastError();
    if (prev != device) (void)hipSetDevice(prev);
    return e == hipSuccess ? 0 : -2;
}

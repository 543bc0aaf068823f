// Is  v_rcp_f32 + one Newton step  the CORRECTLY ROUNDED reciprocal?  Exhaustive over all 2^32 bit patterns on the MI355X itself.
//
// The collision stage's canonical operation order (csrc/lmik_device.h, mirrored by the fp32 oracle build) is defined with the
// correctly rounded reciprocal  RN(1/x).  hipcc's IEEE division is 10 VALU instructions (v_div_scale x2, v_rcp, 4 FMAs,
// v_div_fmas, v_div_fixup); if  y0 = v_rcp_f32(x); e = fma(-x, y0, 1); y = fma(e, y0, y0)  equals RN(1/x) for every x in the
// range the kernels guard, the device can use those 3 instructions and the CPU oracle plain `1.0f / x`: bit-identical by proof
// of exhaustion, not by hope.  This program counts the mismatches per exponent of x (and prints the first few), for one and for
// two Newton steps.
//
// Build / run:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt rcp_exhaustive.hip -o rcp_exhaustive && ./rcp_exhaustive
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

__device__ __forceinline__ float rcp_nr1(float x) {
    const float y0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}

__device__ __forceinline__ float rcp_nr2(float x) {
    float y = rcp_nr1(x);
    const float e = __builtin_fmaf(-x, y, 1.0f);
    return __builtin_fmaf(e, y, y);
}

// mism[variant][exponent]: number of x with that biased exponent whose fast reciprocal differs in bits from 1.0f / x
__global__ void sweep(unsigned long long* mism, uint32_t* first, unsigned long long* nfirst) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const uint32_t bits = (uint32_t)i;
        const float x = __uint_as_float(bits);
        const float want = 1.0f / x;  // correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt)
        const float got[2] = {rcp_nr1(x), rcp_nr2(x)};
        for (int v = 0; v < 2; ++v) {
            const bool both_nan = (want != want) && (got[v] != got[v]);
            if (!both_nan && __float_as_uint(want) != __float_as_uint(got[v])) {
                atomicAdd(&mism[v * 256 + ((bits >> 23) & 0xff)], 1ull);
                const unsigned long long k = atomicAdd(&nfirst[v], 1ull);
                if (k < 16) first[v * 16 + k] = bits;
            }
        }
    }
}

int main() {
    unsigned long long *d_mism, *d_n;
    uint32_t* d_first;
    hipMalloc(&d_mism, 2 * 256 * 8), hipMalloc(&d_n, 2 * 8), hipMalloc(&d_first, 2 * 16 * 4);
    hipMemset(d_mism, 0, 2 * 256 * 8), hipMemset(d_n, 0, 2 * 8), hipMemset(d_first, 0, 2 * 16 * 4);
    sweep<<<4096, 256>>>(d_mism, d_first, d_n);
    if (hipDeviceSynchronize() != hipSuccess) return printf("kernel failed\n"), 1;
    unsigned long long mism[512], n[2];
    uint32_t first[32];
    hipMemcpy(mism, d_mism, sizeof mism, hipMemcpyDeviceToHost), hipMemcpy(n, d_n, sizeof n, hipMemcpyDeviceToHost);
    hipMemcpy(first, d_first, sizeof first, hipMemcpyDeviceToHost);
    for (int v = 0; v < 2; ++v) {
        printf("== v_rcp_f32 + %d Newton step(s) vs RN(1/x): %llu mismatching bit patterns of 2^32\n", v + 1, n[v]);
        int lo = -1, hi = -1;
        for (int e = 0; e < 256; ++e)
            if (mism[v * 256 + e]) {
                printf("   biased exponent %3d (|x| in [2^%d, 2^%d)): %llu\n", e, e - 127, e - 126, mism[v * 256 + e]);
                if (lo < 0) lo = e;
                hi = e;
            }
        unsigned long long inside = 0;  // the range the kernels guard: 2^-100 <= |x| < 2^100
        for (int e = 27; e < 227; ++e) inside += mism[v * 256 + e];
        printf("   mismatches with 2^-100 <= |x| < 2^100: %llu\n", inside);
        for (unsigned long long k = 0; k < (n[v] < 16 ? n[v] : 16); ++k) {
            float x;
            memcpy(&x, &first[v * 16 + k], 4);
            printf("   e.g. x = %a (0x%08x)\n", x, first[v * 16 + k]);
        }
    }
    return 0;
}

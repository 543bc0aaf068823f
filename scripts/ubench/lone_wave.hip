// lone_wave.hip -- developer microbenchmark (not part of the library): how fast does a wavefront issue dependent VALU work as
// a function of how many other wavefronts run, and where they sit (same SIMD / same compute unit / elsewhere)?
//   build: hipcc --offload-arch=gfx950 -O3 -o build_var/lone_wave scripts/ubench/lone_wave.hip     run: build_var/lone_wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>  // 0: dependent FMA chain; 1: 4 independent FMA chains; 2: FMA + DPP mix (dependent)
__global__ void spin(float* out, int iters) {
    extern __shared__ float lds[];
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f, e = 0.125f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            if (MODE == 0) {
                a = __builtin_fmaf(a, b, c);
            } else if (MODE == 1) {
                a = __builtin_fmaf(a, b, c);
                d = __builtin_fmaf(d, b, c);
                e = __builtin_fmaf(e, b, a * 0.f + c);
                c = c + 0.f;
            } else {
                a = __builtin_fmaf(a, b, c);
                a = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a), 0x55, 0xf, 0xf, true));
            }
        }
    }
    if (a + d + e == 123.456f) out[0] = a;
}

template <int MODE>
int run(const char* what, int instr_per_iter) {
    float* out;
    CHECK(hipMalloc(&out, 4096));
    const int iters = 2000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&spin<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    printf("== %s  (%d VALU per loop iteration, %d iterations)\n", what, instr_per_iter, iters);
    const int wgs[] = {1, 8, 32, 64, 128, 256, 512, 1024, 2048};
    const int blocks[] = {64, 256};
    const size_t ldss[] = {0, 40 * 1024, 64 * 1024, 96 * 1024};
    for (int block : blocks)
        for (size_t lds : ldss)
            for (int g : wgs) {
                if ((size_t)g * block > 1024 * 64 * 2) continue;
                std::vector<float> t;
                for (int rep = 0; rep < 7; ++rep) {
                    CHECK(hipEventRecord(e0, 0));
                    hipLaunchKernelGGL(spin<MODE>, dim3(g), dim3(block), lds, 0, out, iters);
                    CHECK(hipEventRecord(e1, 0));
                    CHECK(hipEventSynchronize(e1));
                    float ms;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    t.push_back(ms * 1e3f);
                }
                std::sort(t.begin(), t.end());
                const double per = t[3] * 1e3 / ((double)iters * instr_per_iter);  // ns per VALU instruction of one wavefront
                printf("block %3d lds %6zu  wgs %5d  waves %6d : %8.1f us   %.2f ns / instr / wavefront\n", block, lds, g,
                       g * block / 64, t[3], per);
            }
    CHECK(hipFree(out));
    return 0;
}

int main() {
    if (run<0>("dependent FMA chain", 64)) return 1;
    if (run<2>("dependent FMA + DPP", 128)) return 1;
    if (run<1>("three FMA chains + add", 256)) return 1;
    return 0;
}

// VALU issue-rate calibration on gfx950: how many wave64 VALU instructions per cycle per SIMD do we get with
// (A) independent register-operand FMAs, (B) one dependent chain per wave, (C) FMAs carrying 32-bit literals (8-byte
// encodings), at 1/2/4/8 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p0 = {v0, v1}, p1 = {v2, v3}, p2 = {v4, v5}, p3 = {v6, v7}, pa = {a, a}, pb = {b, b};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                               "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a), "v"(b));)
        } else if (MODE == 1) {
            REP16(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                               "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                               : "+v"(v0) : "v"(a), "v"(b));)
        } else if (MODE == 2) {
            REP16(asm volatile("v_fmaak_f32 %0, %0, %8, 0x3f8ccccd\n v_fmaak_f32 %1, %1, %8, 0x3f8ccccd\n v_fmaak_f32 %2, %2, %8, 0x3f8ccccd\n v_fmaak_f32 %3, %3, %8, 0x3f8ccccd\n"
                               "v_fmaak_f32 %4, %4, %8, 0x3f8ccccd\n v_fmaak_f32 %5, %5, %8, 0x3f8ccccd\n v_fmaak_f32 %6, %6, %8, 0x3f8ccccd\n v_fmaak_f32 %7, %7, %8, 0x3f8ccccd\n"
                               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a));)
        } else if (MODE == 3) {  // VOP2 fmac (4-byte encoding), independent
            REP16(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                               "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a), "v"(b));)
        } else if (MODE == 4) {  // two interleaved dependent chains
            REP16(asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                               "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                               : "+v"(v0), "+v"(v1) : "v"(a), "v"(b));)
        } else if (MODE == 5) {  // packed fp32: 4 independent v_pk_fma_f32 on aligned register pairs (2 FMAs per lane each)
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                               "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));)
        } else if (MODE == 7) {  // transcendental unit: 8 independent v_rsq_f32
            REP16(asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n"
                               "v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7\n"
                               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));)
        } else if (MODE == 8) {  // 8 independent v_sin_f32
            REP16(asm volatile("v_sin_f32 %0, %0\n v_sin_f32 %1, %1\n v_sin_f32 %2, %2\n v_sin_f32 %3, %3\n"
                               "v_sin_f32 %4, %4\n v_sin_f32 %5, %5\n v_sin_f32 %6, %6\n v_sin_f32 %7, %7\n"
                               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));)
        } else if (MODE == 9) {  // 1 transcendental among 7 independent FMAs (the LM loop's mix with the hardware sine / cosine: 23 of 537)
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_sin_f32 %3, %3\n"
                               "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a), "v"(b));)
        } else if (MODE == 10) {  // 1 v_rsq among 7 independent FMAs
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_rsq_f32 %3, %3\n"
                               "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a), "v"(b));)
        } else if (MODE == 11) {  // the result of a transcendental consumed by the next instruction (v_sin -> v_fma on it)
            REP16(asm volatile("v_sin_f32 %0, %0\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n"
                               "v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n"
                               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a), "v"(b));)
        } else if (MODE == 6) {  // packed fp32, one dependent chain
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n"
                               "v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n"
                               : "+v"(p0) : "v"(pa), "v"(pb));)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE>
double run(int blocks, int iters, float* d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3;
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 256 * 8 * 256 * sizeof(float));
    const int iters = 2000;
    const int NM = 12;
    const char* names[NM] = {"indep fma (vop3 regs)", "1 dependent chain", "indep fmaak literal", "indep fmac (vop2)", "2 dependent chains", "indep v_pk_fma_f32", "dependent v_pk_fma_f32",
                             "indep v_rsq_f32", "indep v_sin_f32", "7 fma + 1 v_sin", "7 fma + 1 v_rsq", "v_sin -> dependent fma + 6 fma"};
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks = 256 * wps;  // one 256-thread block = 1 wave per SIMD on a CU
        double t[NM] = {run<0>(blocks, iters, d), run<1>(blocks, iters, d), run<2>(blocks, iters, d), run<3>(blocks, iters, d), run<4>(blocks, iters, d), run<5>(blocks, iters, d), run<6>(blocks, iters, d),
                        run<7>(blocks, iters, d), run<8>(blocks, iters, d), run<9>(blocks, iters, d), run<10>(blocks, iters, d), run<11>(blocks, iters, d)};
        for (int m = 0; m < NM; ++m) {
            const double winstr_per_simd = (double)iters * 128 * wps;  // 16*8 instr per iteration per wave
            printf("waves/SIMD %d  %-30s  %8.1f us   %.3f wave-instr/ns/SIMD (cycles/instr @2.4GHz: %.2f)\n", wps, names[m],
                   t[m] * 1e6, winstr_per_simd / (t[m] * 1e9), t[m] * 2.4e9 / winstr_per_simd);
        }
    }
    return 0;
}

// Do VGPR bank conflicts cost VALU issue slots on gfx950?  Independent v_fma_f32 / v_fmac_f32 streams whose three source registers
// sit in ONE register bank (register number mod 4 equal) against the same streams with the sources spread over three banks, at
// 1 / 2 / 4 / 8 wavefronts per SIMD.  hipcc allocates registers without regard to banks on gfx9, so if the two differ the fused
// kernel's issue rate depends on an accident of allocation.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/vgpr_banks.hip -o build_var/vgpr_banks
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x
#define CLOB "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23"

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    // sources v0..v11 (set once), destinations v16..v23
    asm volatile("v_mov_b32 v0, 1.0\n v_mov_b32 v1, 0.5\n v_mov_b32 v2, 0.25\n v_mov_b32 v3, 2.0\n v_mov_b32 v4, 1.0\n v_mov_b32 v5, 0.5\n"
                 "v_mov_b32 v6, 0.25\n v_mov_b32 v7, 2.0\n v_mov_b32 v8, 1.0\n v_mov_b32 v9, 0.5\n v_mov_b32 v10, 0.25\n v_mov_b32 v11, 2.0\n"
                 "v_mov_b32 v16, 0\n v_mov_b32 v17, 0\n v_mov_b32 v18, 0\n v_mov_b32 v19, 0\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n" ::: CLOB);
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // VOP3 fma, three sources in bank 0
            REP16(asm volatile("v_fma_f32 v16, v0, v4, v8\n v_fma_f32 v17, v0, v4, v8\n v_fma_f32 v18, v0, v4, v8\n v_fma_f32 v19, v0, v4, v8\n"
                               "v_fma_f32 v20, v0, v4, v8\n v_fma_f32 v21, v0, v4, v8\n v_fma_f32 v22, v0, v4, v8\n v_fma_f32 v23, v0, v4, v8\n" ::: CLOB);)
        } else if (MODE == 1) {  // VOP3 fma, sources in banks 0, 1, 2
            REP16(asm volatile("v_fma_f32 v16, v0, v5, v10\n v_fma_f32 v17, v0, v5, v10\n v_fma_f32 v18, v0, v5, v10\n v_fma_f32 v19, v0, v5, v10\n"
                               "v_fma_f32 v20, v0, v5, v10\n v_fma_f32 v21, v0, v5, v10\n v_fma_f32 v22, v0, v5, v10\n v_fma_f32 v23, v0, v5, v10\n" ::: CLOB);)
        } else if (MODE == 2) {  // VOP3 fma, two sources in one bank, one in another
            REP16(asm volatile("v_fma_f32 v16, v0, v4, v9\n v_fma_f32 v17, v0, v4, v9\n v_fma_f32 v18, v0, v4, v9\n v_fma_f32 v19, v0, v4, v9\n"
                               "v_fma_f32 v20, v0, v4, v9\n v_fma_f32 v21, v0, v4, v9\n v_fma_f32 v22, v0, v4, v9\n v_fma_f32 v23, v0, v4, v9\n" ::: CLOB);)
        } else if (MODE == 3) {  // VOP2 fmac: dst (also a source) + two sources, all in bank 0  (v16, v20: bank 0)
            REP16(asm volatile("v_fmac_f32 v16, v0, v4\n v_fmac_f32 v20, v0, v4\n v_fmac_f32 v16, v8, v4\n v_fmac_f32 v20, v8, v4\n"
                               "v_fmac_f32 v16, v0, v8\n v_fmac_f32 v20, v0, v8\n v_fmac_f32 v16, v4, v8\n v_fmac_f32 v20, v4, v8\n" ::: CLOB);)
        } else if (MODE == 4) {  // VOP2 fmac: dst bank 0/1, sources banks 2 and 3 (v2,v3,v6,v7,v10,v11)
            REP16(asm volatile("v_fmac_f32 v16, v2, v3\n v_fmac_f32 v17, v2, v3\n v_fmac_f32 v16, v6, v7\n v_fmac_f32 v17, v6, v7\n"
                               "v_fmac_f32 v16, v10, v11\n v_fmac_f32 v17, v10, v11\n v_fmac_f32 v16, v2, v7\n v_fmac_f32 v17, v2, v7\n" ::: CLOB);)
        } else if (MODE == 5) {  // v_mul_f32 VOP2, two sources in one bank
            REP16(asm volatile("v_mul_f32 v16, v0, v4\n v_mul_f32 v17, v0, v4\n v_mul_f32 v18, v0, v4\n v_mul_f32 v19, v0, v4\n"
                               "v_mul_f32 v20, v0, v4\n v_mul_f32 v21, v0, v4\n v_mul_f32 v22, v0, v4\n v_mul_f32 v23, v0, v4\n" ::: CLOB);)
        } else if (MODE == 6) {  // v_mul_f32 VOP2, sources in two banks
            REP16(asm volatile("v_mul_f32 v16, v0, v5\n v_mul_f32 v17, v0, v5\n v_mul_f32 v18, v0, v5\n v_mul_f32 v19, v0, v5\n"
                               "v_mul_f32 v20, v0, v5\n v_mul_f32 v21, v0, v5\n v_mul_f32 v22, v0, v5\n v_mul_f32 v23, v0, v5\n" ::: CLOB);)
        } else if (MODE == 7) {  // VOP3 fma with an SGPR / inline constant operand (one VGPR read fewer)
            REP16(asm volatile("v_fma_f32 v16, v0, v5, 1.0\n v_fma_f32 v17, v0, v5, 1.0\n v_fma_f32 v18, v0, v5, 1.0\n v_fma_f32 v19, v0, v5, 1.0\n"
                               "v_fma_f32 v20, v0, v5, 1.0\n v_fma_f32 v21, v0, v5, 1.0\n v_fma_f32 v22, v0, v5, 1.0\n v_fma_f32 v23, v0, v5, 1.0\n" ::: CLOB);)
        }
    }
    float r;
    asm volatile("v_add_f32 %0, v16, v17\n v_add_f32 %0, %0, v18\n v_add_f32 %0, %0, v19\n v_add_f32 %0, %0, v20\n v_add_f32 %0, %0, v21\n v_add_f32 %0, %0, v22\n v_add_f32 %0, %0, v23\n" : "=v"(r) :: CLOB);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
double run(int blocks, int iters, float* d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3;
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 256 * 8 * sizeof(float));
    const int iters = 2000;
    const char* names[8] = {"fma vop3, 3 srcs one bank", "fma vop3, 3 banks", "fma vop3, 2+1 banks", "fmac vop2, all one bank", "fmac vop2, 3 banks",
                            "mul vop2, one bank", "mul vop2, two banks", "fma vop3, 2 vgpr + const"};
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks = 256 * wps;
        double t[8] = {run<0>(blocks, iters, d), run<1>(blocks, iters, d), run<2>(blocks, iters, d), run<3>(blocks, iters, d),
                       run<4>(blocks, iters, d), run<5>(blocks, iters, d), run<6>(blocks, iters, d), run<7>(blocks, iters, d)};
        for (int m = 0; m < 8; ++m) {
            const double winstr_per_simd = (double)iters * 128 * wps;
            printf("waves/SIMD %d  %-28s %8.1f us   cycles/instr @2.4GHz: %.2f\n", wps, names[m], t[m] * 1e6, t[m] * 2.4e9 / winstr_per_simd);
        }
    }
    return 0;
}

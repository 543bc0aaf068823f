// What clock does the chip hold while the fused launch runs?  One resident wavefront samples  s_memtime (shader-clock ticks)
// against  s_memrealtime (a constant 100 MHz counter) every few microseconds; the quotient x 100 MHz is the shader clock during
// that window (MI355X_MICROARCH.md, "DVFS give-back").  Launched on its own stream beside the workload under test by
// scripts/clock_probe.py; it uses ~10 registers and no LDS, so it fits beside four 117-register wavefronts per SIMD.
//
// Build:  hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/ubench/clock_probe.hip -o build_var/libclock_probe.so
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ void clock_probe_kernel(uint64_t* __restrict__ out, int n_samples, int sleeps_per_sample) {
    if (threadIdx.x != 0) return;
    uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n_samples; ++i) {
        for (int k = 0; k < sleeps_per_sample; ++k) __builtin_amdgcn_s_sleep(127);  // ~127 x 64 cycles each
        const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        out[2 * i] = t1 - t0;
        out[2 * i + 1] = r1 - r0;
        t0 = t1, r0 = r1;
    }
}

extern "C" int clock_probe_launch(void* stream, uint64_t* out, int n_samples, int sleeps_per_sample) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out, n_samples, sleeps_per_sample);
    return (int)hipGetLastError();
}

// xcd_pingpong.hip -- developer microbenchmark (not part of the library): what does a flag hand-off between two workgroups cost
// when they sit on the SAME accelerator die (XCD: shared L2) and when they sit on different ones?  dp_search's resident kernels
// hand a cost row from step to step through such flags (kernels_dp.h); workgroup i of a launch goes to XCD i mod 8.
//   build: hipcc --offload-arch=gfx950 -O3 -o build_var/xcd_pingpong scripts/ubench/xcd_pingpong.hip     run: build_var/xcd_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

// SCOPE: 0 = agent-scope atomics (what the library uses), 1 = workgroup-scope atomics (only meaningful on one XCD: both ends then
// meet in that die's L2)
template <int SCOPE>
__global__ __launch_bounds__(64) void pingpong(uint32_t* flags, int iters, int a, int b, uint32_t* where) {
    const int me = blockIdx.x;
    if (me != a && me != b) return;
    if (threadIdx.x != 0) return;
    where[me == a ? 0 : 1] = xcc_id();
    uint32_t* mine = flags + (me == a ? 0 : 64);
    uint32_t* theirs = flags + (me == a ? 64 : 0);
    for (int i = 1; i <= iters; ++i) {
        if (me == a) {
            if (SCOPE == 0) __hip_atomic_store(mine, (uint32_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_store(mine, (uint32_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        uint32_t v = 0;
        long spins = 0;
        do {
            if (SCOPE == 0) v = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else v = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } while (v < (uint32_t)i && ++spins < (1l << 17));
        if (v < (uint32_t)i) {
            where[2] = 0xdead;  // gave up: the other end's store never became visible at this scope
            return;
        }
        if (me == b) {
            if (SCOPE == 0) __hip_atomic_store(mine, (uint32_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_store(mine, (uint32_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

template <int SCOPE>
int run(const char* what, int a, int b) {
    uint32_t *flags, *where;
    CHECK(hipMalloc(&flags, 4096));
    CHECK(hipMalloc(&where, 64));
    const int iters = 2000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<float> t;
    uint32_t h[3] = {0, 0, 0};
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipMemset(flags, 0, 4096));
        CHECK(hipMemset(where, 0, 64));
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(pingpong<SCOPE>, dim3(256), dim3(64), 0, 0, flags, iters, a, b, where);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms * 1e3f);
        CHECK(hipMemcpy(h, where, 12, hipMemcpyDeviceToHost));
        if (h[2] == 0xdead) break;
    }
    std::sort(t.begin(), t.end());
    while (t.size() < 3) t.push_back(t.back());
    printf("%-10s workgroups %3d (XCD %u) <-> %3d (XCD %u): %8.1f us for %d round trips = %6.0f ns per hand-off%s\n", what, a, h[0], b, h[1],
           t[2], iters, t[2] * 1e3f / (2.f * iters), h[2] == 0xdead ? "   GAVE UP (store never seen)" : "");
    CHECK(hipFree(flags));
    CHECK(hipFree(where));
    return 0;
}

int main() {
    const int pairs[][2] = {{0, 8}, {0, 16}, {0, 128}, {0, 1}, {0, 2}, {0, 4}, {0, 7}, {3, 11}, {3, 12}};
    for (auto& p : pairs) {
        if (run<0>("agent", p[0], p[1])) return 1;
    }
    for (auto& p : pairs) {
        if (run<1>("workgroup", p[0], p[1])) return 1;
    }
    return 0;
}

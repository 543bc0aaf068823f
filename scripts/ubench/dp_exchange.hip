// dp_exchange.hip -- developer microbenchmark (not part of the library): the per-waypoint exchange of dp_search's resident kernels
// (kernels_dp.h) WITHOUT the arithmetic.  W resident workgroups of 512 lanes, BP = 4 "destinations" each (k = 4 W words per step); in
// every step each workgroup needs all k words of the previous step and publishes its own four.  What does a step cost, by exchange form?
//   row      the library's form: the row of step t is its own flag (sentinel), every lane polls the word it needs (sc1 loads)
//   spread   the same with every 128-byte line of the row placed STRIDE bytes apart (other memory channels)
//   counter  data stored sc1, drained, then ONE agent-scope atomic add per workgroup to a counter sharded NSH ways (lines of their
//            own); a consumer polls the NSH counters with one wavefront, then loads its words once
//   push     every producer stores its four {value, step} granules into EVERY consumer's private mailbox; a consumer polls only
//            lines nobody else reads
// The dependent part of a step mimics the library's: polled words -> LDS image -> barrier -> four wavefronts reduce (minimum) by DPP ->
// publish minimum + 1 (so that step t really depends on all of step t - 1, and the final value is known: T - 1).
//   build: hipcc --offload-arch=gfx950 -O3 -o build_var/dp_exchange scripts/ubench/dp_exchange.hip     run: build_var/dp_exchange
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr uint32_t kSentinel = 0xFFFFFFFFu;
constexpr int BP = 4;
constexpr uint32_t kBudget = 1u << 20;

__device__ __forceinline__ uint32_t ld_sc1(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ uint32_t wave_min(uint32_t x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = min(x, (uint32_t)__shfl_xor((int)x, o));
    return x;
}

// the dependent tail of a step: image -> barrier -> wavefront i reduces destination i; returns the workgroup's published value in
// lane 0 of wavefronts 0..3 (all four destinations publish the same number here)
__device__ __forceinline__ uint32_t reduce_step(uint32_t (*img)[512], uint32_t mine, int tid) {
#pragma unroll
    for (int u = 0; u < BP; ++u) img[u][tid] = mine;
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    uint32_t v = kSentinel;
    if (wave < BP) {
        v = img[wave][lane];
#pragma unroll
        for (int w = 1; w < 8; ++w) v = min(v, img[wave][lane + 64 * w]);
        v = wave_min(v);
    }
    return v;
}

// word index -> position in a row whose 32-word lines are `line_words` words apart (32 = dense)
__device__ __forceinline__ size_t spread(int a, int line_words) { return (size_t)(a >> 5) * line_words + (a & 31); }

// FORM 0 / 1: row (line_words = 32) / spread
__global__ __launch_bounds__(512) void k_row(uint32_t* rows, size_t row_words, int line_words, int k, int T, uint32_t* fail) {
    __shared__ uint32_t img[2][BP][512];
    const int tid = threadIdx.x, b0 = blockIdx.x * BP;
    for (int t = 1; t < T; ++t) {
        const uint32_t* prev = rows + (size_t)(t - 1) * row_words;
        uint32_t mine = kSentinel - 1;
        for (int a = tid; a < k; a += 512) {
            const uint32_t* p = prev + spread(a, line_words);
            uint32_t v = ld_sc1(p);
            for (uint32_t s = 0; v == kSentinel && s < kBudget; ++s) {
                __builtin_amdgcn_s_sleep(1);
                v = ld_sc1(p);
                if ((s & 1023u) == 1023u && ld_sc1(fail)) break;  // some wait has expired: the launch has no result, drain fast
            }
            if (v == kSentinel) atomicAdd(fail, 1u);
            mine = min(mine, v);
        }
        const uint32_t v = reduce_step(img[t & 1], mine, tid);
        if ((tid & 63) == 0 && (tid >> 6) < BP && b0 + (tid >> 6) < k) st_sc1(rows + (size_t)t * row_words + spread(b0 + (tid >> 6), line_words), v + 1);
    }
}

// FORM 5: the row form with the workgroup's four words published by ONE store instruction (lanes 0..3 of wavefront 0, after an LDS
// gather and a second LDS-only barrier): one fabric write per workgroup and step instead of four partial ones into the same line
__global__ __launch_bounds__(512) void k_row_1store(uint32_t* rows, int k, int T, uint32_t* fail) {
    __shared__ uint32_t img[2][BP][512];
    __shared__ uint32_t red[2][BP];
    const int tid = threadIdx.x, b0 = blockIdx.x * BP;
    for (int t = 1; t < T; ++t) {
        const uint32_t* prev = rows + (size_t)(t - 1) * k;
        uint32_t mine = kSentinel - 1;
        for (int a = tid; a < k; a += 512) {
            const uint32_t* p = prev + a;
            uint32_t v = ld_sc1(p);
            for (uint32_t s = 0; v == kSentinel && s < kBudget; ++s) {
                __builtin_amdgcn_s_sleep(1);
                v = ld_sc1(p);
                if ((s & 1023u) == 1023u && ld_sc1(fail)) break;
            }
            if (v == kSentinel) atomicAdd(fail, 1u);
            mine = min(mine, v);
        }
        const uint32_t v = reduce_step(img[t & 1], mine, tid);
        if ((tid & 63) == 0 && (tid >> 6) < BP) red[t & 1][tid >> 6] = v;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (tid < BP && b0 + tid < k) st_sc1(rows + (size_t)t * k + b0 + tid, red[t & 1][tid] + 1);
    }
}

// FORM 4: the row form with all W workgroups on ONE accelerator die (XCD: one L2).  The grid is 8 W workgroups; the first to arrive
// claims its die (ctrl[0]), workgroups on other dies leave at once, the ones on the claimed die take logical ids (ctrl[1]).  PLAIN:
// the row is published with a plain store (the line stays in that die's L2: MI355X_MICROARCH.md, "stores of each flavour") and
// polled with sc1 loads (L1 bypassed, L2-served) -- an L2 round trip instead of a trip through the fabric.
__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
template <bool PLAIN>
__global__ __launch_bounds__(512) void k_row_xcd(uint32_t* rows, int k, int T, int W, uint32_t* ctrl, uint32_t* fail) {
    __shared__ uint32_t img[2][BP][512];
    __shared__ int s_id;
    const int tid = threadIdx.x;
    if (tid == 0) {
        const uint32_t me = xcc_id();
        const uint32_t old = atomicCAS(ctrl, 0xFFFFFFFFu, me);
        const uint32_t die = old == 0xFFFFFFFFu ? me : old;
        int id = -1;
        if (die == me) id = (int)atomicAdd(ctrl + 1, 1u);
        s_id = id < W ? id : -1;
    }
    __syncthreads();
    const int id = s_id;
    if (id < 0) return;
    const int b0 = id * BP;
    for (int t = 1; t < T; ++t) {
        const uint32_t* prev = rows + (size_t)(t - 1) * k;
        uint32_t mine = kSentinel - 1;
        for (int a = tid; a < k; a += 512) {
            const uint32_t* p = prev + a;
            uint32_t v = ld_sc1(p);
            for (uint32_t s = 0; v == kSentinel && s < kBudget; ++s) {
                __builtin_amdgcn_s_sleep(1);
                v = ld_sc1(p);
                if ((s & 1023u) == 1023u && ld_sc1(fail)) break;
            }
            if (v == kSentinel) atomicAdd(fail, 1u);
            mine = min(mine, v);
        }
        const uint32_t v = reduce_step(img[t & 1], mine, tid);
        if ((tid & 63) == 0 && (tid >> 6) < BP && b0 + (tid >> 6) < k) {
            uint32_t* dst = rows + (size_t)t * k + b0 + (tid >> 6);
            if (PLAIN)
                *reinterpret_cast<volatile uint32_t*>(dst) = v + 1;
            else
                st_sc1(dst, v + 1);
        }
    }
}

// FORM 2: counter.  data[t][k] written sc1, drained; lane 0 adds 1 to counter[t][blockIdx % nsh] (each on a line of its own).
__global__ __launch_bounds__(512) void k_counter(uint32_t* data, uint32_t* counters, int nsh, int k, int T, uint32_t* fail) {
    __shared__ uint32_t img[2][BP][512];
    const int tid = threadIdx.x, b0 = blockIdx.x * BP, W = gridDim.x;
    for (int t = 1; t < T; ++t) {
        // wait: wavefront 0, lane s polls shard s of step t - 1 (step 0 is pre-filled by the host: counters already full)
        if (tid < nsh) {
            const uint32_t want = (uint32_t)((W - tid + nsh - 1) / nsh);  // workgroups with blockIdx % nsh == tid
            const uint32_t* c = counters + ((size_t)(t - 1) * nsh + tid) * 32;
            uint32_t v = ld_sc1(c);
            for (uint32_t s = 0; v < want && s < kBudget; ++s) {
                __builtin_amdgcn_s_sleep(1);
                v = ld_sc1(c);
                if ((s & 1023u) == 1023u && ld_sc1(fail)) break;
            }
            if (v < want) atomicAdd(fail, 1u);
        }
        __syncthreads();
        uint32_t mine = kSentinel - 1;
        for (int a = tid; a < k; a += 512) mine = min(mine, ld_sc1(data + (size_t)(t - 1) * k + a));
        const uint32_t v = reduce_step(img[t & 1], mine, tid);
        if ((tid & 63) == 0 && (tid >> 6) < BP && b0 + (tid >> 6) < k) st_sc1(data + (size_t)t * k + b0 + (tid >> 6), v + 1);
        // drain the four stores, then signal (one lane, after a barrier that follows every storing wave's drain)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(counters + ((size_t)t * nsh + (blockIdx.x % nsh)) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// FORM 3: push.  box[consumer][t & 1][k] of 8-byte granules {value, step}; producer lane c (< W) stores its workgroup's four granules
// into consumer c's box (two 16-byte sc1 stores); consumer lane a polls box[me][(t - 1) & 1][a] until its tag is t - 1.
__global__ __launch_bounds__(512) void k_push(uint2* box, int k, int T, uint32_t* fail) {
    __shared__ uint32_t img[2][BP][512];
    __shared__ uint32_t pub[BP];
    const int tid = threadIdx.x, b0 = blockIdx.x * BP, W = gridDim.x;
    for (int t = 1; t < T; ++t) {
        const uint2* mybox = box + ((size_t)blockIdx.x * 2 + ((t - 1) & 1)) * k;
        uint32_t mine = kSentinel - 1;
        for (int a = tid; a < k; a += 512) {
            const unsigned long long* p = reinterpret_cast<const unsigned long long*>(mybox + a);
            unsigned long long g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (uint32_t s = 0; (uint32_t)(g >> 32) != (uint32_t)(t - 1) && s < kBudget; ++s) {
                __builtin_amdgcn_s_sleep(1);
                g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((s & 1023u) == 1023u && ld_sc1(fail)) break;
            }
            if ((uint32_t)(g >> 32) != (uint32_t)(t - 1)) atomicAdd(fail, 1u);
            mine = min(mine, (uint32_t)g);
        }
        const uint32_t v = reduce_step(img[t & 1], mine, tid);
        if ((tid & 63) == 0 && (tid >> 6) < BP) pub[tid >> 6] = v + 1;
        __syncthreads();
        for (int c = tid; c < W; c += 512) {
            uint2* dst = box + ((size_t)c * 2 + (t & 1)) * k + b0;
#pragma unroll
            for (int u = 0; u < BP; ++u)
                if (b0 + u < k)
                    __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst + u), ((unsigned long long)(uint32_t)t << 32) | pub[u], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

static uint32_t* g_fail = nullptr;
static uint32_t g_timeouts = 0;
static float time_launch(const std::function<void()>& reset, const std::function<void()>& launch) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int r = 0; r < 7; ++r) {
        reset();
        CHECK(hipMemset(g_fail, 0, 4));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        ts.push_back(ms);
        uint32_t f;
        CHECK(hipMemcpy(&f, g_fail, 4, hipMemcpyDeviceToHost));
        g_timeouts += f;
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main() {
    const int T = 256;
    uint32_t* fail;
    CHECK(hipMalloc(&fail, 4));
    CHECK(hipMemset(fail, 0, 4));
    g_fail = fail;
    printf("dp_exchange: W workgroups x 512 lanes, k = 4 W words per step, T = %d steps; us per step (median of 7 launches)\n", T);
    for (int W : {11, 16, 44, 64, 75, 128, 256}) {
        const int k = W * BP;
        printf("W %3d k %4d:", W, k);
        // row / spread
        for (int line_words : {32, 1056, 8224}) {
            const size_t row_words = (size_t)((k + 31) / 32) * line_words;
            uint32_t* rows;
            CHECK(hipMalloc(&rows, row_words * T * 4));
            std::vector<uint32_t> first(row_words, 0u);
            auto reset = [&] {
                CHECK(hipMemset(rows, 0xFF, row_words * T * 4));
                CHECK(hipMemcpy(rows, first.data(), row_words * 4, hipMemcpyHostToDevice));
            };
            const float ms = time_launch(reset, [&] { hipLaunchKernelGGL(k_row, dim3(W), dim3(512), 0, 0, rows, row_words, line_words, k, T, fail); });
            uint32_t last;
            CHECK(hipMemcpy(&last, rows + (size_t)(T - 1) * row_words, 4, hipMemcpyDeviceToHost));
            printf("  %s %6.2f%s", line_words == 32 ? "row" : (line_words == 1056 ? "spread4K" : "spread32K"), ms * 1e3 / (T - 1), last == (uint32_t)(T - 1) ? "" : "(!)");
            CHECK(hipFree(rows));
        }
        // one store instruction per workgroup
        {
            uint32_t* rows;
            CHECK(hipMalloc(&rows, (size_t)k * T * 4));
            std::vector<uint32_t> first(k, 0u);
            auto reset = [&] {
                CHECK(hipMemset(rows, 0xFF, (size_t)k * T * 4));
                CHECK(hipMemcpy(rows, first.data(), (size_t)k * 4, hipMemcpyHostToDevice));
            };
            const float ms = time_launch(reset, [&] { hipLaunchKernelGGL(k_row_1store, dim3(W), dim3(512), 0, 0, rows, k, T, fail); });
            uint32_t last;
            CHECK(hipMemcpy(&last, rows + (size_t)(T - 1) * k, 4, hipMemcpyDeviceToHost));
            printf("  row/1-store %6.2f%s", ms * 1e3 / (T - 1), last == (uint32_t)(T - 1) ? "" : "(!)");
            CHECK(hipFree(rows));
        }
        // one die
        if (W <= 64) {
            for (int plain = 0; plain < 2; ++plain) {
                uint32_t *rows, *ctrl;
                CHECK(hipMalloc(&rows, (size_t)k * T * 4));
                CHECK(hipMalloc(&ctrl, 64));
                std::vector<uint32_t> first(k, 0u);
                auto reset = [&] {
                    CHECK(hipMemset(rows, 0xFF, (size_t)k * T * 4));
                    CHECK(hipMemcpy(rows, first.data(), (size_t)k * 4, hipMemcpyHostToDevice));
                    const uint32_t c0[2] = {0xFFFFFFFFu, 0u};
                    CHECK(hipMemcpy(ctrl, c0, 8, hipMemcpyHostToDevice));
                };
                const float ms = time_launch(reset, [&] {
                    if (plain) hipLaunchKernelGGL(k_row_xcd<true>, dim3(8 * W), dim3(512), 0, 0, rows, k, T, W, ctrl, fail);
                    else hipLaunchKernelGGL(k_row_xcd<false>, dim3(8 * W), dim3(512), 0, 0, rows, k, T, W, ctrl, fail);
                });
                uint32_t last;
                CHECK(hipMemcpy(&last, rows + (size_t)(T - 1) * k, 4, hipMemcpyDeviceToHost));
                printf("  %s %6.2f%s", plain ? "one-die/plain-store" : "one-die/sc1-store", ms * 1e3 / (T - 1), last == (uint32_t)(T - 1) ? "" : "(!)");
                CHECK(hipFree(rows));
                CHECK(hipFree(ctrl));
            }
        }
        // counter
        for (int nsh : {1, 8, 32}) {
            uint32_t *data, *counters;
            CHECK(hipMalloc(&data, (size_t)k * T * 4));
            CHECK(hipMalloc(&counters, (size_t)T * nsh * 32 * 4));
            auto reset = [&] {
                CHECK(hipMemset(data, 0, (size_t)k * T * 4));
                CHECK(hipMemset(counters, 0, (size_t)T * nsh * 32 * 4));
                std::vector<uint32_t> c0((size_t)nsh * 32, 0u);
                for (int s = 0; s < nsh; ++s) c0[(size_t)s * 32] = (uint32_t)((W - s + nsh - 1) / nsh);
                CHECK(hipMemcpy(counters, c0.data(), c0.size() * 4, hipMemcpyHostToDevice));
            };
            const float ms = time_launch(reset, [&] { hipLaunchKernelGGL(k_counter, dim3(W), dim3(512), 0, 0, data, counters, nsh, k, T, fail); });
            uint32_t last;
            CHECK(hipMemcpy(&last, data + (size_t)(T - 1) * k, 4, hipMemcpyDeviceToHost));
            printf("  counter/%d %6.2f%s", nsh, ms * 1e3 / (T - 1), last == (uint32_t)(T - 1) ? "" : "(!)");
            CHECK(hipFree(data));
            CHECK(hipFree(counters));
        }
        // push
        {
            uint2* box;
            CHECK(hipMalloc(&box, (size_t)W * 2 * k * 8));
            auto reset = [&] {
                // tags 0xFFFFFFFF everywhere, then step 0 (tag 0, value 0) in every consumer's buffer 0
                CHECK(hipMemset(box, 0xFF, (size_t)W * 2 * k * 8));
                for (int c = 0; c < W; ++c) CHECK(hipMemset(box + (size_t)c * 2 * k, 0, (size_t)k * 8));
            };
            const float ms = time_launch(reset, [&] { hipLaunchKernelGGL(k_push, dim3(W), dim3(512), 0, 0, box, k, T, fail); });
            uint2 last;
            CHECK(hipMemcpy(&last, box + (size_t)((T - 1) & 1) * k, 8, hipMemcpyDeviceToHost));
            printf("  push %6.2f%s", ms * 1e3 / (T - 1), (last.x == (uint32_t)(T - 1) && last.y == (uint32_t)(T - 1)) ? "" : "(!)");
            CHECK(hipFree(box));
        }
        printf("  timeouts %u\n", g_timeouts);
        fflush(stdout);
    }
    return 0;
}

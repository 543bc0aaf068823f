// kernels_dp.h -- dp_search: time-major transpose, one (min,max) product per waypoint, back-trace; the mjac tensor.
// Part of the translation unit cppflow_hip.hip (included inside its anonymous namespace); gfx950 only.
#pragma once

// ---- dp_search (cppflow/search.py:100-191) -----------------------------------------------------------------------------------
// costs[b,t] = min_a { max(mjac(a->b,t-1), costs[a,t-1]) + ext[b,t] }, first minimal a recorded; one launch per timestep
// (the recurrence is sequential in t; each step is a k x k (min,max) product).  The reference materialises
// mjacs[k,k,T-1] (1 GB at k = 1024, T = 256); here every entry lives in a register for one compare.
// Work arrays are time-major so that a step reads two contiguous [k,d] slabs: qT[t][a][j], costsT[t][a], memoT[t][b].

__global__ __launch_bounds__(256) void dp_transpose_kernel(const float* __restrict__ q, const float* __restrict__ ext, int k,
                                                           int T, int d, float* __restrict__ qT, float* __restrict__ costsT) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)k * T * d;
    if (i < total) {
        const int j = (int)(i % d);
        const size_t r = i / d;
        const int t = (int)(r % T), a = (int)(r / T);
        qT[((size_t)t * k + a) * d + j] = q[i];
    }
    if (i < (size_t)k) {  // costs[:,0] = q_costs_external[:,0]  (search.py:151); 0xFFFFFFFF is the persistent kernel's "not yet" word
        const float c = ext[i * T];
        costsT[i] = __float_as_uint(c) == 0xFFFFFFFFu ? __uint_as_float(0x7FC00000u) : c;
    }
}

// order-preserving bits of a float (every non-NaN value; a NaN sorts above +inf, where the scalar rule `v < best` also leaves it)
__device__ __forceinline__ unsigned long long dp_key(float v, int a) {
    uint32_t b = __float_as_uint(v);
    b ^= (b >> 31) ? 0xFFFFFFFFu : 0x80000000u;
    return ((unsigned long long)b << 32) | (uint32_t)a;
}
__device__ __forceinline__ float dp_key_value(unsigned long long key) {
    uint32_t b = (uint32_t)(key >> 32);
    b ^= (b >> 31) ? 0x80000000u : 0xFFFFFFFFu;
    return __uint_as_float(b);
}
template <int CTRL>
__device__ __forceinline__ unsigned long long dp_dpp_min(unsigned long long x) {  // lanes without a source lane keep their own value
    const int lo = (int)(uint32_t)x, hi = (int)(uint32_t)(x >> 32);
    const uint32_t olo = (uint32_t)__builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    const uint32_t ohi = (uint32_t)__builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
    return o < x ? o : x;
}
__device__ __forceinline__ unsigned long long dp_wave_min_to_lane63(unsigned long long x) {
    x = dp_dpp_min<0x111>(x);  // row_shr:1
    x = dp_dpp_min<0x112>(x);  // row_shr:2
    x = dp_dpp_min<0x114>(x);  // row_shr:4
    x = dp_dpp_min<0x118>(x);  // row_shr:8   -> lane 15 of each row of 16 holds the row's minimum
    x = dp_dpp_min<0x142>(x);  // row_bcast:15 -> lanes 31 / 63 hold rows 0-1 / 2-3
    x = dp_dpp_min<0x143>(x);  // row_bcast:31 -> lane 63 holds the wavefront's minimum
    return x;
}

// BPB = destination candidates per workgroup: fewer for small k so that a step still fills the chip with workgroups
template <int D, int kDpBPB>
__global__ __launch_bounds__(256) void dp_step_kernel(const float* __restrict__ q_prev, const float* __restrict__ q_cur,
                                                      const float* __restrict__ cost_prev, const float* __restrict__ ext,
                                                      int k, int T, int t, uint32_t pris_mask, float pscale,
                                                      float* __restrict__ cost_cur, int32_t* __restrict__ memo_cur) {
    __shared__ unsigned long long red[kDpBPB][4];
    const int b0 = blockIdx.x * kDpBPB;
    float qb[kDpBPB][D], eb[kDpBPB], best[kDpBPB];
    int arg[kDpBPB];
#pragma unroll
    for (int i = 0; i < kDpBPB; ++i) {
        const int b = min(b0 + i, k - 1);
#pragma unroll
        for (int j = 0; j < D; ++j) qb[i][j] = q_cur[(size_t)b * D + j];
        eb[i] = ext[(size_t)b * T + t];
        best[i] = INFINITY;
        arg[i] = 0;
    }
    for (int a = threadIdx.x; a < k; a += 256) {
        float qa[D];
#pragma unroll
        for (int j = 0; j < D; ++j) qa[j] = q_prev[(size_t)a * D + j];
        const float c = cost_prev[a];
#pragma unroll
        for (int i = 0; i < kDpBPB; ++i) {
            float dq[D];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                dq[j] = qb[i][j] - qa[j];
                if ((pris_mask >> j) & 1u) dq[j] *= pscale;  // search.py:119-121
            }
            const float m = max_wrapped_change<D>(dq);
            const float v = fmaxf(m, c) + eb[i];  // search.py:157-158
            if (v < best[i]) {
                best[i] = v;
                arg[i] = a;
            }
        }
    }
    // lexicographic (value, index) minimum = first minimal index: ONE 64-bit unsigned minimum on (order key of the value, index),
    // over the wavefront by DPP (no LDS traffic), then over the 4 wavefronts
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < kDpBPB; ++i) {
        const unsigned long long key = dp_wave_min_to_lane63(dp_key(best[i], arg[i]));
        if (lane == 63) red[i][wave] = key;
    }
    __syncthreads();
    if (threadIdx.x < kDpBPB) {
        const int i = threadIdx.x;
        unsigned long long key = red[i][0];
#pragma unroll
        for (int w = 1; w < 4; ++w) key = red[i][w] < key ? red[i][w] : key;
        if (b0 + i < k) {
            cost_cur[b0 + i] = dp_key_value(key);
            memo_cur[b0 + i] = (int32_t)(uint32_t)key;
        }
    }
}

// ---- the whole recurrence in ONE launch ------------------------------------------------------------------------------------------
// T-1 dependent launches of a few microseconds each are launch-bound (1.05 ms at the reference's k = 175, T = 256).  Here
// the workgroups stay resident for all T-1 steps, each owning one (k <= 64) or four (k <= 256) destinations.  The only thing step t
// needs from the other workgroups is the cost row of step t-1 -- k floats -- and that row is its own flag: the cost table is
// pre-filled with a sentinel bit pattern (0xFFFFFFFF, which no cost can be: costs are sums of finite non-negative numbers,
// and a NaN computed by the hardware is 0x7FC00000), every cost is stored ONCE with an agent-scope (write-through, sc1)
// store, and a lane that needs costs[t-1][a] re-reads that one word with agent-scope loads until it is no longer the sentinel
// (MI355X_MICROARCH.md, hand-off form R2 "the data is the flag": a naturally aligned word written by one sc1 store needs no
// fence, no counter and no barrier).  No grid barrier: a lane waits exactly for the sources it reads.  The part of a step
// that does not depend on the costs -- the max joint change for this lane's (source, destination) pairs -- is computed BEFORE
// the wait, so the exposed time per step is the hand-off latency plus the argmin reduction.
// Same arithmetic, same reduction order as dp_step_kernel: the table, the argmins and the path stay bit-exact with the oracle.
// Every spin is bounded: a workgroup that never gets its inputs (the source workgroups are not resident with it -- a CU-masked or
// partitioned device, a profiler that serialises workgroups) falls through with +inf costs instead of hanging the GPU, AND says
// so: it raises kDpTimedOut in memoT[0] (row 0 of the memo table is zero-initialised and only ever read as a value), which
// dp_backtrace_kernel turns into best_idx[*] = -1 / a NaN path.  A caller that sees best_idx[0] < 0 repeats the call with
// CPPF_TUNE_DP_PERSISTENT = 0 (cppflow_amd.search.dp_search does).
constexpr uint32_t kDpSentinel = 0xFFFFFFFFu;
constexpr uint32_t kDpQuietNan = 0x7FC00000u;
constexpr int32_t kDpTimedOut = 0x40000000;

// `spin_budget`: re-reads before a wait gives up (the host passes 2^22, a few seconds; cppf_debug_set(CPPF_TUNE_DP_SPIN_LOG2) shrinks it
// so that a test can see the timeout path).  Once ANY wait of the launch has expired the launch has no result (dp_backtrace_kernel
// reports -1), so every later wait looks at the flag every 1024 re-reads and gives up at once when it is up: a CU-masked device pays
// for ONE expired wait, not for (T - 1) x workgroups of them (minutes of an apparently hung GPU at T = 256).
__device__ __forceinline__ float dp_wait_cost(const float* p, int32_t* timed_out_flag, uint32_t spin_budget) {
    uint32_t bits = __hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (uint32_t spins = 0; bits == kDpSentinel && spins < spin_budget; ++spins) {
        __builtin_amdgcn_s_sleep(1);
        bits = __hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((spins & 1023u) == 1023u &&
            (__hip_atomic_load(timed_out_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kDpTimedOut))
            break;
    }
    if (bits == kDpSentinel) atomicOr(timed_out_flag, kDpTimedOut);
    return bits == kDpSentinel ? INFINITY : __uint_as_float(bits);
}

__device__ __forceinline__ void dp_publish_cost(float* p, float v) {
    uint32_t bits = __float_as_uint(v);
    bits = bits == kDpSentinel ? kDpQuietNan : bits;
    __hip_atomic_store(reinterpret_cast<uint32_t*>(p), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One wavefront per destination b (grid = k single-wave workgroups): lane l handles sources l, l+64, ..., so a step needs neither
// LDS nor a workgroup barrier and the cost-independent part finishes before the hand-off does; every wavefront polls the whole
// cost row, though (k^2 flag reads per step), which is what its step time grows with: used for k <= 64.
template <int D>
__global__ __launch_bounds__(64) void dp_persistent_kernel(const float* __restrict__ qT, const float* __restrict__ ext, int k,
                                                           int T, uint32_t pris_mask, float pscale, float* costsT,
                                                           int32_t* __restrict__ memoT, uint32_t spin_budget) {
    constexpr int AMAX = 2;
    const int b = blockIdx.x, lane = threadIdx.x;
    const int A = (k + 63) >> 6;  // sources per lane (wave-uniform)
    for (int t = 1; t < T; ++t) {
        const float* q_prev = qT + (size_t)(t - 1) * k * D;
        const float* cost_prev = costsT + (size_t)(t - 1) * k;
        // ---- independent of the costs ----
        float qb[D], m[AMAX];
#pragma unroll
        for (int j = 0; j < D; ++j) qb[j] = qT[((size_t)t * k + b) * D + j];
        const float eb = ext[(size_t)b * T + t];
#pragma unroll
        for (int s = 0; s < AMAX; ++s) {
            m[s] = 0.f;
            if (s < A) {  // wave-uniform
                const int a = min(lane + 64 * s, k - 1);
                float dq[D];
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    dq[j] = qb[j] - q_prev[(size_t)a * D + j];
                    if ((pris_mask >> j) & 1u) dq[j] *= pscale;  // search.py:119-121
                }
                m[s] = max_wrapped_change<D>(dq);
            }
        }
        // ---- dependent: wait for exactly the costs this lane reads; sources ascend per lane, so `<` keeps the first minimum ----
        float best = INFINITY;
        int arg = 0;
#pragma unroll
        for (int s = 0; s < AMAX; ++s) {
            const int a = lane + 64 * s;
            if (s < A && a < k) {
                const float c = dp_wait_cost(cost_prev + a, memoT, spin_budget);
                const float v = fmaxf(m[s], c) + eb;  // search.py:157-158
                if (v < best) {
                    best = v;
                    arg = a;
                }
            }
        }
        const unsigned long long key = dp_wave_min_to_lane63(dp_key(best, arg));
        if (lane == 63) {
            memoT[(size_t)t * k + b] = (int32_t)(uint32_t)key;  // read only by the back-trace launch
            dp_publish_cost(costsT + (size_t)t * k + b, dp_key_value(key));
        }
    }
}

// The same recurrence with FOUR destinations per workgroup: half of the polling traffic of the one-wavefront-per-destination
// form (k^2 / 2 instead of k^2 flag reads per step, which is what bounds that form from k ~ 64 up), at the price of one LDS
// transpose + barrier per step (wavefront i reduces destination i by DPP; the image is double-buffered by the parity of t).
// SRC = sources per half-workgroup = the most candidates the instantiation takes: 256 (512 lanes, k <= 256) or 512 (1 024 lanes,
// 257 <= k <= 512, and with NS = 2 sources per lane 513 <= k <= 1024 up to 8 joints: sixteen wavefronts, four per SIMD -- the
// cost-independent part of a step is 2 NS (source, destination) pairs per lane where dp_resident_kernel's 512 lanes meet four
// destinations with NS sources each, twice as many).
template <int D, int SRC, int NS = 1>
__global__ __launch_bounds__(2 * SRC) void dp_persistent4_kernel(const float* __restrict__ qT, const float* __restrict__ ext, int k,
                                                                 int T, uint32_t pris_mask, float pscale, float* costsT,
                                                                 int32_t* __restrict__ memoT, uint32_t spin_budget) {
    // 2 SRC lanes: lane (h, a) = (tid / SRC, tid % SRC) handles sources a (and a + SRC: NS = 2, 513 <= k <= 1024) for destinations 2h
    // and 2h + 1 of the workgroup's four, so the cost-independent part is 2 NS (source, destination) pairs per lane -- short enough
    // to finish inside the hand-off latency -- while each cost word is still polled by only two lanes per workgroup.
    constexpr int BP = 4;
    static_assert(SRC == 256 || SRC == 512, "256 or 512 lanes per half-workgroup");
    __shared__ unsigned long long keys[2][BP][SRC];
    const int b0 = blockIdx.x * BP, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = tid / SRC, a = tid & (SRC - 1);  // k <= NS SRC
    // The operands of a step (this lane's source configurations at t - 1, its two destinations' at t, their external costs) depend
    // on nothing: they are loaded ONE STEP AHEAD, behind the arithmetic of the current step and in front of its wait, so that their
    // latency lies under the hand-off instead of in front of the next one (k = 175, T = 256: 491 -> measured in DESIGN.md 8.1).
    float qa[NS][D], qb[2][D], eb[2];
    auto load_operands = [&](int t, float (&qa_)[NS][D], float (&qb_)[2][D], float (&eb_)[2]) {
        const float* q_prev = qT + (size_t)(t - 1) * k * D;
        const float* q_cur = qT + (size_t)t * k * D;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int asrc = min(a + SRC * s, k - 1);
#pragma unroll
            for (int j = 0; j < D; ++j) qa_[s][j] = q_prev[(size_t)asrc * D + j];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int b = min(b0 + 2 * h + u, k - 1);
            eb_[u] = ext[(size_t)b * T + t];
#pragma unroll
            for (int j = 0; j < D; ++j) qb_[u][j] = q_cur[(size_t)b * D + j];
        }
    };
    if (T > 1) load_operands(1, qa, qb, eb);
    for (int t = 1; t < T; ++t) {
        const float* cost_prev = costsT + (size_t)(t - 1) * k;
        // ---- independent of the costs ----
        float m[NS][2], e_now[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) e_now[u] = eb[u];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float dq[D];
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    dq[j] = qb[u][j] - qa[s][j];
                    if ((pris_mask >> j) & 1u) dq[j] *= pscale;  // search.py:119-121
                }
                m[s][u] = max_wrapped_change<D>(dq);
            }
        if (t + 1 < T) load_operands(t + 1, qa, qb, eb);  // (in flight during the wait below)
        // ---- dependent: wait for exactly the costs this lane reads ----
        unsigned long long (*img)[SRC] = keys[t & 1];
        {
            // lanes / sources beyond k carry (+inf, 0), like the idle lanes of the per-waypoint kernel
            unsigned long long best[2] = {dp_key(INFINITY, 0), dp_key(INFINITY, 0)};
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int as = a + SRC * s;
                if (as < k) {
                    const float c = dp_wait_cost(cost_prev + as, memoT, spin_budget);
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const float v = fmaxf(m[s][u], c) + e_now[u];  // search.py:157-158
                        const unsigned long long key = dp_key(v < INFINITY ? v : INFINITY, v < INFINITY ? as : 0);
                        best[u] = NS == 1 || key < best[u] ? key : best[u];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) img[2 * h + u][a] = best[u];
        }
        __syncthreads();
        if (wave < BP) {
            const int i = wave;
            unsigned long long key = img[i][lane];
#pragma unroll
            for (int w = 1; w < SRC / 64; ++w) {
                const unsigned long long o = img[i][lane + 64 * w];
                key = o < key ? o : key;
            }
            key = dp_wave_min_to_lane63(key);
            if (lane == 63 && b0 + i < k) {
                memoT[(size_t)t * k + b0 + i] = (int32_t)(uint32_t)key;  // read only by the back-trace launch
                dp_publish_cost(costsT + (size_t)t * k + b0 + i, dp_key_value(key));
            }
        }
    }
}

// The resident form for 257 .. 1024 candidates (the reference's rerun searches k = 300, cppflow/planners.py:47, 253-258; eight ranks
// gather 8 x 128 = 1024): FOUR destinations per 512-lane workgroup again -- at most 256 workgroups, one per compute unit, all
// resident -- but every lane now owns NS = 1 or 2 SOURCES (tid, tid + 512) and meets all four destinations with them, so that each
// cost word of step t - 1 is polled by exactly ONE lane of a workgroup (k x k / 4 flag reads per step chip-wide: at k = 1024 one
// megabyte per polling round, where one wavefront per destination would read four).  Per step: the cost-independent part (NS x 4 wrapped
// joint changes per lane) and the operand loads of step t + 1 come BEFORE the wait; then each lane waits for its NS words, forms its
// four (value, index) keys, one 64-bit minimum per destination over its own sources, and the workgroup reduces the four columns of
// the 512-entry LDS image (wavefront i takes destination i: 8 entries per lane, then DPP); the image is double-buffered by the
// parity of t, one barrier per step.  Same arithmetic and the same first-minimal-index rule as dp_step_kernel -- the 64-bit key
// orders (value, index) lexicographically whatever order the sources are visited in.
template <int D, int NS>
__global__ __launch_bounds__(512) void dp_resident_kernel(const float* __restrict__ qT, const float* __restrict__ ext, int k, int T,
                                                          uint32_t pris_mask, float pscale, float* costsT,
                                                          int32_t* __restrict__ memoT, uint32_t spin_budget) {
    constexpr int BP = 4;
    __shared__ unsigned long long keys[2][BP][512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (the destinations' operands are the same for every lane, and the compiler would fetch them with scalar loads -- which count on
    // lgkmcnt like the LDS accesses: the barrier's wait then exposes their latency in EVERY step; fetched as vector loads they are
    // older than the polling loads and have returned when those do.  An opaque zero keeps the index per-lane for the compiler.)
    int lane_zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
    const int b0 = blockIdx.x * BP + lane_zero;
    float qa[NS][D], qb[BP][D], eb[BP];
    auto load_operands = [&](int t, float (&qa_)[NS][D], float (&qb_)[BP][D], float (&eb_)[BP]) {
        const float* q_prev = qT + (size_t)(t - 1) * k * D;
        const float* q_cur = qT + (size_t)t * k * D;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int a = min(tid + 512 * s, k - 1);
#pragma unroll
            for (int j = 0; j < D; ++j) qa_[s][j] = q_prev[(size_t)a * D + j];
        }
#pragma unroll
        for (int u = 0; u < BP; ++u) {
            const int b = min(b0 + u, k - 1);
            eb_[u] = ext[(size_t)b * T + t];
#pragma unroll
            for (int j = 0; j < D; ++j) qb_[u][j] = q_cur[(size_t)b * D + j];
        }
    };
    if (T > 1) load_operands(1, qa, qb, eb);
    for (int t = 1; t < T; ++t) {
        const float* cost_prev = costsT + (size_t)(t - 1) * k;
        // ---- independent of the costs ----
        float m[NS][BP], e_now[BP];
#pragma unroll
        for (int u = 0; u < BP; ++u) e_now[u] = eb[u];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int u = 0; u < BP; ++u) {
                float dq[D];
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    dq[j] = qb[u][j] - qa[s][j];
                    if ((pris_mask >> j) & 1u) dq[j] *= pscale;  // search.py:119-121
                }
                m[s][u] = max_wrapped_change<D>(dq);
            }
        if (t + 1 < T) load_operands(t + 1, qa, qb, eb);  // (in flight during the wait below)
        // ---- dependent: wait for exactly the costs this lane reads ----
        unsigned long long best[BP];
#pragma unroll
        for (int u = 0; u < BP; ++u) best[u] = dp_key(INFINITY, 0);  // lanes / sources beyond k carry (+inf, 0), like the idle lanes of dp_step_kernel
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int a = tid + 512 * s;
            if (a < k) {
                const float c = dp_wait_cost(cost_prev + a, memoT, spin_budget);
#pragma unroll
                for (int u = 0; u < BP; ++u) {
                    const float v = fmaxf(m[s][u], c) + e_now[u];  // search.py:157-158
                    const unsigned long long key = dp_key(v < INFINITY ? v : INFINITY, v < INFINITY ? a : 0);
                    best[u] = key < best[u] ? key : best[u];
                }
            }
        }
        unsigned long long (*img)[512] = keys[t & 1];
#pragma unroll
        for (int u = 0; u < BP; ++u) img[u][tid] = best[u];
        __syncthreads();
        if (wave < BP) {
            const int i = wave;
            unsigned long long key = img[i][lane];
#pragma unroll
            for (int w = 1; w < 8; ++w) {
                const unsigned long long o = img[i][lane + 64 * w];
                key = o < key ? o : key;
            }
            key = dp_wave_min_to_lane63(key);
            if (lane == 63 && b0 + i < k) {
                memoT[(size_t)t * k + b0 + i] = (int32_t)(uint32_t)key;  // read only by the back-trace launch
                dp_publish_cost(costsT + (size_t)t * k + b0 + i, dp_key_value(key));
            }
        }
    }
}

// 513 ... 1024 candidates in one resident launch: the 1 024-lane form of dp_persistent4_kernel with two sources per lane where its
// registers allow four wavefronts per SIMD (up to 9 joints: 128 VGPRs without scratch), dp_resident_kernel (512 lanes, four
// destinations x two sources per lane) beyond -- and as the A/B (`wide` = false: CPPF_TUNE_DP_PERSISTENT = 2)
struct DpResidentForm {
    const void* fn;  // the kernel (all forms share one argument list: qT, ext, k, T, pris_mask, pscale, costsT, memoT, spin)
    int block;
    unsigned grid;
};
template <int D>
inline DpResidentForm dp_resident_form(int k, int tune_persistent) {
    const unsigned g4 = (unsigned)((k + 3) / 4);
    if (k <= 64) return {(const void*)dp_persistent_kernel<D>, 64, (unsigned)k};  // one destination per wavefront
    if (k <= 256) return {(const void*)dp_persistent4_kernel<D, 256>, 512, g4};
    if (k <= 512) {  // the same form on 1 024 lanes, <= 128 workgroups (tune 2, the A/B: one source per lane and four destinations)
        if (tune_persistent != 2) return {(const void*)dp_persistent4_kernel<D, 512>, 1024, g4};
        return {(const void*)dp_resident_kernel<D, 1>, 512, g4};
    }
    // two sources per lane, <= 256 workgroups: one per compute unit
    if constexpr (D <= 9) {
        if (tune_persistent != 2) return {(const void*)dp_persistent4_kernel<D, 512, 2>, 1024, g4};
    }
    return {(const void*)dp_resident_kernel<D, 2>, 512, g4};
}

// ---- the recurrence on ONE compute unit, from a precomputed transition table (k <= 256) -------------------------------------------
// The resident kernels above pay one inter-workgroup hand-off per waypoint (~2-3 us with k cost words polled by every
// workgroup) for ~0.1 us of arithmetic.  The only part of a step that depends on the previous one is
//     c_t[b] = min_a max( m_t[a][b], c_{t-1}[a] ) + e_t[b]        (k^2 max / min pairs),
// and m_t -- the wrapped, scaled maximum joint change from candidate a at t-1 to candidate b at t, 95 % of the arithmetic --
// depends on nothing.  So: (1) dp_table_kernel fills m for all T-1 steps with the whole chip (the tensor the reference
// materialises too, search.py:100-125; [t][a][b] here, b fastest, rows padded to 64 floats); (2) dp_chain_kernel runs the
// chain in ONE workgroup of 8 wavefronts -- no hand-off, one __syncthreads per waypoint: lane <-> destination b, the sources
// split over the wavefronts, c_{t-1} held one value per lane and broadcast with v_readlane (an SGPR operand of v_max), the
// table rows streamed through registers one waypoint ahead (each register is reloaded for step t + 1 right after step t used it);
// (3) dp_memo_kernel recovers the argmins off the chain: memo_t[b] = the first a whose max(m, c) + e equals c_t[b] -- the same
// value the strict '<' scan of the per-waypoint kernel keeps, because fp32 addition is monotone ( min_a (f_a + e) = (min_a f_a)
// + e bit for bit ) and the comparison is on the sums.  Costs, argmins and path stay bit-exact with the oracle.
// The chain compares ORDER KEYS, not floats: key(x) = bits(x) ^ (x < 0 ? 0xFFFFFFFF : 0x80000000) is monotone in x over all
// non-NaN floats (dp_key above), so max / min of keys are max / min of the values, exactly, on the integer pipe (one
// instruction each, no NaN canonicalisation of loaded operands).  The table stores key(m) (m >= 0: bits | 0x80000000).
__device__ __forceinline__ uint32_t dp_okey(float x) {
    const uint32_t b = __float_as_uint(x);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float dp_okey_value(uint32_t key) {
    return __uint_as_float(key ^ ((key >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}
constexpr uint32_t kDpKeyInf = 0xFF800000u;  // key(+inf)

template <int D>
__global__ __launch_bounds__(256) void dp_table_kernel(const float* __restrict__ qT, int k, int kp, int T, uint32_t pris_mask,
                                                       float pscale, uint32_t* __restrict__ table) {
    // grid: x over chunks of 256 destinations, y over chunks of 8 sources, z = step - 1.  A thread keeps its destination's
    // configuration in registers for the 8 sources of its chunk, whose configurations are wavefront-uniform (scalar loads).
    const int b = (int)(blockIdx.x * 256 + threadIdx.x);
    if (b >= kp) return;
    const int t = (int)blockIdx.z + 1, a0 = (int)blockIdx.y * 8;
    float qb[D];
    {
        const float* src = qT + ((size_t)t * k + min(b, k - 1)) * D;
#pragma unroll
        for (int j = 0; j < D; ++j) qb[j] = src[j];
    }
    for (int a = a0; a < min(a0 + 8, k); ++a) {
        float m = INFINITY;  // row padding: a destination that does not exist
        if (b < k) {
            const float* qa = qT + ((size_t)(t - 1) * k + a) * D;
            float dq[D];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                dq[j] = qb[j] - qa[j];
                if ((pris_mask >> j) & 1u) dq[j] *= pscale;  // search.py:119-121
            }
            m = max_wrapped_change<D>(dq);
        }
        table[((size_t)(t - 1) * k + a) * kp + b] = dp_okey(m);
    }
}

// eight sources of the chain: c_{t-1}[a] comes out of lane I0 + i of cv as a scalar operand and meets the lane's V = KP / 64
// destinations (every lane of the wavefront owns V consecutive ones: one V-dword load per lane fetches a whole row, no lane
// idles); then the eight table rows of the step ahead go into the registers just used
template <int I0, int N, int KP>
__device__ __forceinline__ void dp_chain_chunk(uint32_t (&m)[N][KP / 64], uint32_t cv, uint32_t (&best)[KP / 64],
                                               const uint32_t* next, unsigned col) {
    constexpr int V = KP / 64;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cv, I0 + i);
#pragma unroll
        for (int v = 0; v < V; ++v) best[v] = min(best[v], max(m[I0 + i][v], c));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // rows past the group's range are read too -- the next group's, or the padding behind the
                                   // table: their cost lane is the largest key, they never win
        const uint32_t* src = next + (size_t)(I0 + i) * KP + col;
#pragma unroll
        for (int v = 0; v < V; ++v) m[I0 + i][v] = src[v];
    }
}

// every chunk of every step runs, unconditionally: the compiler can count the loads in flight (s_waitcnt vmcnt(N)) only through
// straight-line code -- a load under a branch makes it wait for (almost) everything, i.e. for the rows just requested
template <int I0, int N, int KP>
__device__ __forceinline__ void dp_chain_chunks(uint32_t (&m)[N][KP / 64], uint32_t cv, uint32_t (&best)[KP / 64],
                                                const uint32_t* next, unsigned col) {
    if constexpr (I0 < N) {
        dp_chain_chunk<I0, N, KP>(m, cv, best, next, col);
        dp_chain_chunks<I0 + 8, N, KP>(m, cv, best, next, col);
    }
}

// 8 wavefronts = 8 groups of <= NA sources; lane <-> destinations V lane .. V lane + V - 1.  SETS register sets hold the table
// rows of SETS consecutive steps: step t computes on set t % SETS and refills it, chunk by chunk, with the rows of step
// t + SETS -- a load has SETS steps to arrive.
template <int KP, int NA, int SETS>  // KP = row stride of the table = k rounded up to 64: compile-time, so that the row offsets are literals
__device__ __forceinline__ void dp_chain_body(const uint32_t* __restrict__ table, const float* __restrict__ ext, int k, int T,
                                              float* __restrict__ costsT, uint32_t (&part)[2][8][256], float (&hist)[32][256]) {
    constexpr int V = KP / 64;
    const int tid = threadIdx.x, lane = tid & 63, grp = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int per = (k + 7) / 8;
    const int a0 = min(grp * per, k - 1);
    const int na = max(0, min(per, k - grp * per));  // a short last group / empty groups when k < 8
    const unsigned col = (unsigned)(V * lane);
    uint32_t m[SETS][NA][V];
#pragma unroll
    for (int u = 0; u < SETS; ++u) {  // the table rows of steps 1 .. SETS (set u serves steps u, u + SETS, ...; step t -> set t % SETS)
        const int t = u == 0 ? SETS : u;  // the first step that uses set u
        const uint32_t* row = table + ((size_t)min(t - 1, max(T - 2, 0)) * k + a0) * KP + col;
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int v = 0; v < V; ++v) m[u][i][v] = row[(size_t)i * KP + v];
    }
    // the external cost of THIS wavefront's sources, SETS waypoints ahead of its use like the table rows (ext is [k][T]; a
    // short chain's step is shorter than the load's latency): er[w % SETS] holds ext[s0][w], slots indexed at compile time
    const int s0 = min(a0 + lane, k - 1);
    float er[SETS];
#pragma unroll
    for (int u = 0; u < SETS; ++u) er[u] = ext[(size_t)s0 * T + min(u == 0 ? SETS : u, T - 1)];  // the first waypoint >= 1 of each slot
    const float c_first = costsT[s0];  // costs[:, 0] as dp_transpose_kernel left them
    float e0 = 0.f;
    // Steps 1 .. T (step T only extracts the last cost row), in groups of SETS that are straight-line code: steps beyond T
    // in the last group recompute on stale operands and store nothing.  Two gfx950 facts shape the loop (see also
    // full_rows_eliminate_kernel): a branch around a step or around a load makes the compiler wait for (almost) every load in
    // flight where it needs one, and loads and stores share ONE counter (vmcnt) whose returns are ordered only within each
    // kind, so with a store in flight a prefetched load can only be waited for by draining it.  Hence: no global store in a
    // step -- a cost row goes to an LDS ring and 16 rows at a time are written out, followed by an explicit drain, so that no
    // store is ever pending where a load is waited for.
    for (int t0 = 1; t0 <= T; t0 += SETS) {
#pragma unroll
        for (int u = 0; u < SETS; ++u) {
            const int t = t0 + u;
            constexpr int kSet0 = 1 % SETS;  // set of step t0 (t0 = 1 mod SETS)
            const int set = (kSet0 + u) % SETS;
            // ---- c_{t-1} for this wavefront's sources, one per lane
            float c0;
            {
                const uint32_t (*p)[256] = part[(t - 1) & 1];
                uint32_t k0 = p[0][s0];
#pragma unroll
                for (int g = 1; g < 8; ++g) k0 = min(k0, p[g][s0]);
                c0 = t == 1 ? c_first : dp_okey_value(k0) + e0;  // search.py:157-158: min_a max(...) + ext (see the header)
            }
            if (lane < na && t >= 2 && t <= T) hist[(t - 1) & 31][a0 + lane] = c0;  // the cost row of step t - 1
            const uint32_t cv = lane < na ? dp_okey(c0) : 0xFFFFFFFFu;  // sources beyond this group's range never win
            // e_t, needed at step t + 1 (set = t % SETS); the slot is then reloaded with e_{t+SETS}.  The explicit move ends the
            // old value's life in the slot's register BEFORE the load: left to itself the compiler loads into a scratch
            // register and copies it into the slot at the loop's end -- a wait for a load just issued.
            asm volatile("v_mov_b32 %0, %1" : "=v"(e0) : "v"(er[set]));
            er[set] = ext[(size_t)s0 * T + min(t + SETS, T - 1)];
            // ---- k^2 / 8 max / min pairs per wavefront; step t + SETS's table rows go into the registers just used
            const uint32_t* next = table + ((size_t)min(t + SETS - 1, max(T - 2, 0)) * k + a0) * KP;
            uint32_t best[V];
#pragma unroll
            for (int v = 0; v < V; ++v) best[v] = 0xFFFFFFFFu;
            dp_chain_chunks<0, NA, KP>(m[set], cv, best, next, col);
#pragma unroll
            for (int v = 0; v < V; ++v) part[t & 1][grp][V * lane + v] = best[v];
            // A workgroup barrier for the LDS exchange ONLY (__syncthreads() would also drain the global loads in flight)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (((t - 1) & 15) == 15 && t <= T) {  // rows t - 16 .. t - 1 are complete in the ring: two per wavefront go out
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int r = t - 16 + 2 * grp + q;
                    if (r >= 1)
                        for (int jj = lane; jj < k; jj += 64) costsT[(size_t)r * k + jj] = hist[r & 31][jj];
                }
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): no store is pending beyond this point
            }
        }
    }
    // the rows since the last full group of 16 (the barrier of step T has passed: row T - 1 is in the ring)
    {
        const int done = T >= 2 ? ((T - 1) / 16) * 16 - 1 : 0;  // last row written by the in-loop flushes (rows 1 .. done), -1 -> none
        for (int r = max(done + 1, 1) + grp; r <= T - 1; r += 8)
            for (int jj = lane; jj < k; jj += 64) costsT[(size_t)r * k + jj] = hist[r & 31][jj];
    }
}

template <int KP>
__global__ __launch_bounds__(512) void dp_chain_kernel(const uint32_t* __restrict__ table, const float* __restrict__ ext, int k,
                                                       int T, float* __restrict__ costsT) {
    __shared__ uint32_t part[2][8][256];  // [parity of t][source group][destination]: partial minima (keys) over the group's sources
    __shared__ float hist[32][256];       // the last 32 cost rows (two windows of 16: one fills while the other is written out)
    // register sets in flight: a step of a long row (k > 128) is as long as the memory latency, one set ahead is enough there
    if constexpr (KP <= 64)
        dp_chain_body<KP, 8, 3>(table, ext, k, T, costsT, part, hist);
    else if constexpr (KP <= 128)
        dp_chain_body<KP, 16, 2>(table, ext, k, T, costsT, part, hist);
    else if constexpr (KP <= 192)
        dp_chain_body<KP, 24, 2>(table, ext, k, T, costsT, part, hist);
    else
        dp_chain_body<KP, 32, 1>(table, ext, k, T, costsT, part, hist);
}

// memo_t[b] = first a with max(m_t[a][b], c_{t-1}[a]) + e_t[b] == c_t[b]   (0 when c_t[b] is not below +inf: nothing was ever
// 'less than' the initial +inf of the scan, search.py:157-159 keeps index 0 then)
__global__ __launch_bounds__(256) void dp_memo_kernel(const uint32_t* __restrict__ table, const float* __restrict__ ext,
                                                      const float* __restrict__ costsT, int k, int kp, int T,
                                                      int32_t* __restrict__ memoT) {
    // grid: x over chunks of 64 destinations, y = step - 1; the four wavefronts of a workgroup scan a quarter of the sources each
    __shared__ int first[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int t = (int)blockIdx.y + 1, b = (int)blockIdx.x * 64 + lane;
    const int bb = min(b, k - 1);
    const float target = costsT[(size_t)t * k + bb], e = ext[(size_t)bb * T + t];
    const uint32_t* col = table + (size_t)(t - 1) * k * kp + bb;
    const float* cp = costsT + (size_t)(t - 1) * k;
    const int per = (k + 3) / 4, a_lo = w * per, a_hi = min(k, a_lo + per);
    int arg = 0x7fffffff;
    if (target < INFINITY) {
        for (int a = a_lo; a < a_hi; ++a) {
            const float v = dp_okey_value(max(col[(size_t)a * kp], dp_okey(cp[a]))) + e;
            if (arg == 0x7fffffff && v == target) arg = a;
        }
    }
    first[w][lane] = arg;
    __syncthreads();
    if (w == 0 && b < k) {
        int r = min(min(first[0][lane], first[1][lane]), min(first[2][lane], first[3][lane]));
        memoT[(size_t)t * k + b] = r == 0x7fffffff ? 0 : r;
    }
}

// _get_mjacs (cppflow/search.py:100-125): mjacs[i, j, t] = max over joints of |wrap(scale_j (q[i, t+1, j] - q[j_, t, j]))| -- the
// [k, k, T-1] tensor the reference's dp_search materialises (1 GB at k = 1024).  cppf_dp_search never builds it; this
// kernel exists for callers that want the tensor itself.  One lane per (i, j_, t), t fastest (coalesced writes).
template <int D>
__global__ __launch_bounds__(256) void mjacs_kernel(const float* __restrict__ q, int k, int T, uint32_t pris_mask, float pscale,
                                                    float* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t per = (size_t)(T - 1);
    const size_t total = (size_t)k * k * per;
    if (idx >= total) return;
    const int t = (int)(idx % per);
    const size_t ij = idx / per;
    const int j = (int)(ij % k), i = (int)(ij / k);
    const float* qi = q + ((size_t)i * T + t + 1) * D;
    const float* qj = q + ((size_t)j * T + t) * D;
    float dq[D];
#pragma unroll
    for (int c = 0; c < D; ++c) {
        dq[c] = qi[c] - qj[c];
        if ((pris_mask >> c) & 1u) dq[c] *= pscale;
    }
    out[idx] = max_wrapped_change<D>(dq);
}

// argmin over the final costs (first minimal index), walk the memo table back, gather the path
// The walk is T dependent reads of the memo table.  From global memory that is T memory latencies (63 us at T = 256); with
// `stage` the table is first copied into LDS as bytes (k <= 256, T k <= 64 KB: the launch passes T k bytes of dynamic LDS) and
// walked there (one LDS latency per step).
__global__ __launch_bounds__(256) void dp_backtrace_kernel(const float* __restrict__ q, const float* __restrict__ costsT,
                                                           const int32_t* __restrict__ memoT, int k, int T, int d, int stage,
                                                           int32_t* __restrict__ best_idx, float* __restrict__ best_path) {
    extern __shared__ uint8_t memo8[];
    __shared__ float red_v[4];
    __shared__ int red_a[4];
    if (memoT[0] & kDpTimedOut) {  // (uniform) the resident launch could not hand its cost rows over: no result, and say so
        for (int t = threadIdx.x; t < T; t += 256) best_idx[t] = -1;
        for (int n = threadIdx.x; n < T * d; n += 256) best_path[n] = __builtin_nanf("");
        return;
    }
    if (stage) {
        const int n = T * k;
        for (int i = threadIdx.x; i < n; i += 256) memo8[i] = (uint8_t)memoT[i];
    }
    const float* last = costsT + (size_t)(T - 1) * k;
    float v = INFINITY;
    int a = 0;
    for (int i = threadIdx.x; i < k; i += 256) {
        const float c = last[i];
        if (c < v) {
            v = c;
            a = i;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(v, off, 64);
        const int oa = __shfl_xor(a, off, 64);
        if (ov < v || (ov == v && oa < a)) {
            v = ov;
            a = oa;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        red_v[threadIdx.x >> 6] = v;
        red_a[threadIdx.x >> 6] = a;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (red_v[w] < v || (red_v[w] == v && red_a[w] < a)) {
                v = red_v[w];
                a = red_a[w];
            }
        int i = a;
        if (stage) {
            for (int t = T - 1; t >= 0; --t) {  // search.py:161-173
                best_idx[t] = i;
                i = memo8[t * k + i];
            }
        } else {
            for (int t = T - 1; t >= 0; --t) {
                best_idx[t] = i;
                i = memoT[(size_t)t * k + i];
            }
        }
    }
    __syncthreads();
    for (int n = threadIdx.x; n < T * d; n += 256) {
        const int t = n / d, j = n % d;
        best_path[n] = q[((size_t)best_idx[t] * T + t) * d + j];
    }
}

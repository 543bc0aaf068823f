// kernels_dp.h -- dp_search: time-major transpose, one (min,max) product per waypoint, back-trace; the mjac tensor.
// Part of the one translation unit cppflow_hip.hip (included inside its anonymous namespace); gfx950 only.
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
    if (i < (size_t)k) costsT[i] = ext[i * T];  // costs[:,0] = q_costs_external[:,0]  (search.py:151)
}

// BPB = destination candidates per workgroup: fewer for small k so that a step still fills the chip with workgroups
template <int D, int kDpBPB>
__global__ __launch_bounds__(256) void dp_step_kernel(const float* __restrict__ q_prev, const float* __restrict__ q_cur,
                                                      const float* __restrict__ cost_prev, const float* __restrict__ ext,
                                                      int k, int T, int t, uint32_t pris_mask, float pscale,
                                                      float* __restrict__ cost_cur, int32_t* __restrict__ memo_cur) {
    __shared__ float red_v[kDpBPB][4];
    __shared__ int red_a[kDpBPB][4];
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
            float m = 0.f;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                float dq = qb[i][j] - qa[j];
                if ((pris_mask >> j) & 1u) dq *= pscale;  // search.py:119-121
                m = fmaxf(m, fabsf(wrap_pi(dq)));
            }
            const float v = fmaxf(m, c) + eb[i];  // search.py:157-158
            if (v < best[i]) {
                best[i] = v;
                arg[i] = a;
            }
        }
    }
    // lexicographic (value, index) min: first minimal index, over the wave then over the 4 waves
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < kDpBPB; ++i) {
        float v = best[i];
        int a = arg[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(v, off, 64);
            const int oa = __shfl_xor(a, off, 64);
            if (ov < v || (ov == v && oa < a)) {
                v = ov;
                a = oa;
            }
        }
        if (lane == 0) {
            red_v[i][wave] = v;
            red_a[i][wave] = a;
        }
    }
    __syncthreads();
    if (threadIdx.x < kDpBPB) {
        const int i = threadIdx.x;
        float v = red_v[i][0];
        int a = red_a[i][0];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float ov = red_v[i][w];
            const int oa = red_a[i][w];
            if (ov < v || (ov == v && oa < a)) {
                v = ov;
                a = oa;
            }
        }
        if (b0 + i < k) {
            cost_cur[b0 + i] = v;
            memo_cur[b0 + i] = a;
        }
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
    float m = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        float dq = qi[c] - qj[c];
        if ((pris_mask >> c) & 1u) dq *= pscale;
        m = fmaxf(m, fabsf(wrap_pi(dq)));
    }
    out[idx] = m;
}

// argmin over the final costs (first minimal index), walk the memo table back, gather the path
__global__ __launch_bounds__(256) void dp_backtrace_kernel(const float* __restrict__ q, const float* __restrict__ costsT,
                                                           const int32_t* __restrict__ memoT, int k, int T, int d,
                                                           int32_t* __restrict__ best_idx, float* __restrict__ best_path) {
    __shared__ float red_v[4];
    __shared__ int red_a[4];
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
        for (int t = T - 1; t >= 0; --t) {  // search.py:161-173
            best_idx[t] = i;
            i = memoT[(size_t)t * k + i];
        }
    }
    __syncthreads();
    for (int n = threadIdx.x; n < T * d; n += 256) {
        const int t = n / d, j = n % d;
        best_path[n] = q[((size_t)best_idx[t] * T + t) * d + j];
    }
}

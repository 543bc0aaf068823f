// kernels_eval.h -- per-path reductions: Plan metrics and the per-seed summary of a fused launch's outputs.
// Part of the translation unit cppflow_hip.hip (included inside its anonymous namespace); gfx950 only.
#pragma once

// ---- Plan metrics for every seed at once (cppflow/data_types.py:140-264) ------------------------------------------------------
// One wavefront per seed; lanes stride over the W waypoints (FK + pose metrics per waypoint, joint changes to the next
// waypoint), then a 64-lane butterfly.  out[S,16] -- field order documented at cppf_plan_metrics in the header.
template <int D>
__global__ __launch_bounds__(64) void plan_metrics_kernel(const ChainK ch, const CollK co, int S, int W,
                                                          const float* __restrict__ x, const float* __restrict__ target,
                                                          const uint8_t* __restrict__ self_mask,
                                                          const uint8_t* __restrict__ env_mask,
                                                          const float* __restrict__ q_init, float* __restrict__ out) {
    using RB = DynRobot<D>;
    const RB rb{ch, co};
    const int s = blockIdx.x;
    if (s >= S) return;
    const float rad2deg = 57.29577951308232087680f;
    float mx[4] = {0.f, 0.f, 0.f, 0.f};              // max pos (cm), max rot (deg), mjac revolute (deg), mjac prismatic (cm)
    float sm[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // sum pos, sum rot, length rad, length m, # limit violations, # self, # env
    for (int w = threadIdx.x; w < W; w += 64) {
        const size_t row = (size_t)s * W + w;
        float q[D], R[9], p[3], Rt[9], tt[3], pe, re;
        load_x<D>(x, row, q);
        load_target(target, w, Rt, tt);
        fk_ee<RB>(rb, q, R, p);
        pose_metrics(Rt, tt, R, p, pe, re);
        const float pc = 100.f * pe, rd = rad2deg * re;
        mx[0] = fmaxf(mx[0], nan_to_inf(pc)), mx[1] = fmaxf(mx[1], nan_to_inf(rd));
        sm[0] += pc, sm[1] += rd;
#pragma unroll
        for (int j = 0; j < D; ++j) sm[4] += (float)((q[j] < ch.lo[j]) + (ch.hi[j] < q[j]));  // evaluation_utils.py:24
        if (self_mask) sm[5] += (float)self_mask[row];
        if (env_mask) sm[6] += (float)env_mask[row];
        if (w + 1 < W) {
            float qn[D], dq[D], wr[D];
            load_x<D>(x, row + 1, qn);
#pragma unroll
            for (int j = 0; j < D; ++j) dq[j] = wr[j] = qn[j] - q[j];
            wrap_pi_all<D>(wr);
#pragma unroll
            for (int j = 0; j < D; ++j) {
                if (rb.pris(j)) {
                    const float a = fabsf(dq[j]);
                    mx[3] = fmaxf(mx[3], nan_to_inf(100.f * a)), sm[3] += a;
                } else {
                    const float a = fabsf(wr[j]);
                    mx[2] = fmaxf(mx[2], nan_to_inf(rad2deg * a)), sm[2] += a;
                }
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off, 64));
#pragma unroll
        for (int k = 0; k < 7; ++k) sm[k] += __shfl_xor(sm[k], off, 64);
    }
    if (threadIdx.x == 0) {
        float qd = 0.f;
        if (q_init) {
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const float dq = q_init[j] - x[(size_t)s * W * D + j];
                qd = CPPF_FMA(dq, dq, qd);
            }
            qd = __builtin_sqrtf(qd);
        }
        float* o = out + (size_t)s * 16;
        o[0] = mx[0], o[1] = sm[0] / (float)W, o[2] = mx[1], o[3] = sm[1] / (float)W, o[4] = mx[2], o[5] = mx[3];
        o[6] = sm[2], o[7] = sm[3], o[8] = sm[4], o[9] = sm[5], o[10] = sm[6], o[11] = qd;
        o[12] = o[13] = o[14] = o[15] = 0.f;
    }
}

// ---- per-seed summary of a fused launch's per-row outputs ---------------------------------------------------------------------
// One wavefront per seed: reduces the packed per-row outputs of lm_fused_kernel (no FK) and the joint changes between
// consecutive waypoints of x into 8 floats -- what x_is_valid (cppflow/optimization_utils.py:845-884) and a cross-GPU seed
// selection need, and the payload of the per-step all-gather (32 B per seed instead of 15 B per row):
//   [0] max position error (cm)   [1] max rotation error (deg)   [2] max |revolute joint change| (deg)
//   [3] max |prismatic joint change| (cm)   [4] # self-colliding waypoints   [5] # env-colliding waypoints
//   [6] # waypoints within the joint-limit padding   [7] sum of the external cost (search.py:146-150)
template <int D>
__global__ __launch_bounds__(64) void seed_summary_kernel(const ChainK ch, int S, int W, const float* __restrict__ x,
                                                          const float* __restrict__ ext_cost,
                                                          const float* __restrict__ pos_err,
                                                          const float* __restrict__ rot_err,
                                                          const uint8_t* __restrict__ self_mask,
                                                          const uint8_t* __restrict__ env_mask,
                                                          const uint8_t* __restrict__ jlim_mask, float* __restrict__ out) {
    const int s = blockIdx.x;
    if (s >= S) return;
    const float rad2deg = 57.29577951308232087680f;
    float mp = 0.f, mr = 0.f, mrev = 0.f, mpri = 0.f, ns = 0.f, ne = 0.f, nj = 0.f, sc = 0.f;
    for (int w = threadIdx.x; w < W; w += 64) {
        const size_t row = (size_t)s * W + w;
        mp = fmaxf(mp, nan_to_inf(100.f * pos_err[row]));
        mr = fmaxf(mr, nan_to_inf(rad2deg * rot_err[row]));
        ns += (float)self_mask[row];
        ne += (float)env_mask[row];
        nj += (float)jlim_mask[row];
        sc += ext_cost[row];
        if (w + 1 < W) {
            float q[D], qn[D];
            load_x<D>(x, row, q);
            load_x<D>(x, row + 1, qn);
            float dq[D], wr[D];
#pragma unroll
            for (int j = 0; j < D; ++j) dq[j] = wr[j] = qn[j] - q[j];
            wrap_pi_all<D>(wr);
#pragma unroll
            for (int j = 0; j < D; ++j) {
                if ((ch.pris_mask >> j) & 1u)
                    mpri = fmaxf(mpri, nan_to_inf(fabsf(100.f * dq[j])));
                else
                    mrev = fmaxf(mrev, nan_to_inf(fabsf(rad2deg * wr[j])));
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mp = fmaxf(mp, __shfl_xor(mp, off, 64));
        mr = fmaxf(mr, __shfl_xor(mr, off, 64));
        mrev = fmaxf(mrev, __shfl_xor(mrev, off, 64));
        mpri = fmaxf(mpri, __shfl_xor(mpri, off, 64));
        ns += __shfl_xor(ns, off, 64);
        ne += __shfl_xor(ne, off, 64);
        nj += __shfl_xor(nj, off, 64);
        sc += __shfl_xor(sc, off, 64);
    }
    if (threadIdx.x == 0) {
        float* o = out + (size_t)s * 8;
        o[0] = mp, o[1] = mr, o[2] = mrev, o[3] = mpri, o[4] = ns, o[5] = ne, o[6] = nj, o[7] = sc;
    }
}

// ---- x_is_valid's seed selection over [S,8] summaries (cppflow/optimization_utils.py:856-909) ------------------------------------
// One workgroup per group of seeds: lanes stride over the seeds, then a min / sum / argmin reduction through LDS.  The seeds
// of group g are `n_chunks` chunks of `S_chunk` rows at row offsets (r * n_groups + g) * S_chunk -- the layout an
// all-gather of [n_groups, S_chunk, 8] buffers from n_chunks ranks leaves behind (n_chunks = n_groups = 1: a plain [S,8]).
struct SelectK {
    float thr[4];
    int32_t ignore_self, ignore_env;
    int32_t S_chunk, n_chunks, n_groups;
};

__global__ __launch_bounds__(256) void select_valid_seed_kernel(const float* __restrict__ summary, const SelectK k,
                                                                int32_t* __restrict__ out) {
    __shared__ int s_first[256];
    __shared__ int s_count[256];
    __shared__ float s_cost[256];
    __shared__ int s_arg[256];
    const int tid = threadIdx.x;
    int first = 0x7fffffff, count = 0, arg = 0x7fffffff;
    float best = INFINITY;
    const int g = blockIdx.x, S = k.S_chunk * k.n_chunks;
    out += 4 * g;
    for (int s = tid; s < S; s += 256) {
        const size_t row = ((size_t)(s / k.S_chunk) * k.n_groups + g) * k.S_chunk + (size_t)(s % k.S_chunk);
        const float4 a = reinterpret_cast<const float4*>(summary)[2 * row], b = reinterpret_cast<const float4*>(summary)[2 * row + 1];
        // strict '<' on the maxima (evaluation_utils.py:41-60): a NaN / inf maximum is "not below"
        bool ok = a.x < k.thr[0] && a.y < k.thr[1] && a.z < k.thr[2] && a.w < k.thr[3];
        ok = ok && (k.ignore_self || !(b.x > 0.f)) && (k.ignore_env || !(b.y > 0.f));
        if (ok) {
            first = s < first ? s : first;
            ++count;
        }
        if (b.w < best) best = b.w, arg = s;  // s ascends per lane: first minimum kept
    }
    s_first[tid] = first, s_count[tid] = count, s_cost[tid] = best, s_arg[tid] = arg;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) {
            s_first[tid] = s_first[tid + off] < s_first[tid] ? s_first[tid + off] : s_first[tid];
            s_count[tid] += s_count[tid + off];
            const float c = s_cost[tid + off];
            const int a = s_arg[tid + off];
            if (c < s_cost[tid] || (c == s_cost[tid] && a < s_arg[tid])) s_cost[tid] = c, s_arg[tid] = a;
        }
        __syncthreads();
    }
    if (tid == 0) {
        out[0] = s_first[0] == 0x7fffffff ? -1 : s_first[0];
        out[1] = s_count[0];
        out[2] = s_arg[0] == 0x7fffffff ? -1 : s_arg[0];
        out[3] = 0;
    }
}

// kernels_coupled.h -- the coupled LM step: waypoint-local blocks, distance Jacobians, the eliminations of the block-tridiagonal system.
// Part of the translation unit cppflow_hip.hip (included inside its anonymous namespace); gfx950 only.
#pragma once

// ---- coupled ("full") LM step: cppflow/optimization.py:95-144 + LmResidualFns.get_r_and_J (optimization_utils.py:486-731) -------
// The reference stacks pose / differencing / virtual-config / collision residuals of ONE trajectory into a dense
// J [(6T + d(T-1) + ...) x dT], forms the dense dT x dT normal matrix and factors it (O((dT)^3)).  Structurally
// A = J^T J + lambda I is block-tridiagonal with d x d blocks: pose and collision rows only touch their own waypoint's
// block, the differencing row (t,j) = a_j * wrap(x[t+1,j] - x[t,j]) couples (t,j) with (t+1,j) through -a_j^2 on the
// off-diagonal, virtual-config rows and lambda add to the diagonal.  So:
//   full_blocks_kernel  (one lane per (seed, waypoint) row): the waypoint-local part  M_t = sum Js^T Js + sum alpha^2 g g^T,
//                        m_t = Js^T es - sum alpha^2 dist g   (g = gradient of a colliding capsule distance)
//   full_solve_kernel   (one lane per seed): adds the analytic differencing / virtual-config / lambda terms and runs the
//                        block-tridiagonal elimination D'_t = A_tt - E G_{t-1} E,  G_t = D'_t^-1  (E = -diag(a^2)) forward and
//                        back -- O(T d^3) per trajectory, any number of trajectories at once (the reference: one, :128).
//                        Used when the pose block is on (rank-deficient blocks: Cholesky with floored pivots).
//   full_rows_eliminate_kernel + full_rows_substitute_kernel  (8 or 16 lanes per trajectory, one block row per lane, DPP
//                        Gauss-Jordan, both ends of the path at once): the default beyond ~512 trajectories x 256 waypoints.
//   full_solve_pcr_kernel (one workgroup per trajectory, parallel cyclic reduction over the waypoints, state in LDS for
//                        W <= 256): the default up to there -- the reference's cadence is ONE trajectory.
//   full_solve_wave_kernel (one wavefront per trajectory, round 1's form): kept as a cross-check behind a test hook.

struct FullK {
    float lm_lambda, a_pos, a_rot, a_diff, a_diff_pris, a_vq, a_self, a_env;
    int32_t use_pose, use_diff, use_vq, n_vq, use_self, use_env;
    int32_t S, W;
    int32_t fold;  // full_blocks_kernel also adds the differencing / virtual-config / lambda terms of its own waypoint (what
                   // full_rows_eliminate_kernel expects ready-made; the other elimination kernels add them in their own loops)
    // The "satisfied" row options of LmResidualFns.get_r_and_J (cppflow/optimization_utils.py:514-533, 548-606; off in both
    // presets).  Every one of them is a per-row weight (and, for rows left at full weight, a shift of the residual towards zero by
    // the threshold): pose rows whose |error| is below the threshold are scaled down (:288-333); differencing rows whose |joint
    // change| is below the threshold are dropped (filter_rows_from_r_J_differencing, :736-768) or scaled down (:352-398).
    int32_t pose_scale_satisfied;  // pose_do_scale_down_satisfied
    float pose_thr_m, pose_thr_rad, pose_scale;
    int32_t diff_mode;  // 0 none, 1 differencing_do_ignore_satisfied (filter + shift to threshold), 2 differencing_do_scale_satisfied
    float diff_thr_rad, diff_thr_m, diff_scale;
    int32_t diff_shift_invalid;  // differencing_scale_down_satisfied_shift_invalid_to_threshold
};

// Weight (squared) and residual of ONE differencing row under the "satisfied" options.  rho = the wrapped joint change of the row,
// a = alpha_differencing (x the prismatic scaling where the reference applies it: not in filter mode, optimization_utils.py:601).
__device__ __forceinline__ void diff_row_options(const FullK& prm, bool pris, float rho, float& w2, float& rho_out) {
    const float thr = pris ? prm.diff_thr_m : prm.diff_thr_rad;
    float a = prm.a_diff * ((pris && prm.diff_mode != 1) ? prm.a_diff_pris : 1.f);
    const float shifted = rho < -thr ? rho + thr : (rho > thr ? rho - thr : rho);
    rho_out = rho;
    if (prm.diff_mode == 1) {  // keep the rows with |rho| > thr, shifted to the threshold; the others are not in the system
        a = fabsf(rho) > thr ? a : 0.f;
        rho_out = shifted;
    } else if (prm.diff_mode == 2) {
        const bool valid = fabsf(rho) < thr;
        a = valid ? a * prm.diff_scale : a;
        rho_out = (!valid && prm.diff_shift_invalid) ? shifted : rho;
    }
    w2 = a * a;
}

// Closest points nearer than this (1 um) count as TOUCHING: a segment that crosses a cuboid (or another segment) has distance 0 in
// exact arithmetic and ~1e-9 in fp32, in a direction that is rounding noise -- normalised, that noise became a unit "gradient" on
// one side and a different one (or none) in the fp64 oracle, and in a system held only by its collision rows the two steps differed
// by their whole length (found by scripts/fuzz_coupled.py).  Below the bound the direction is undefined and the gradient is 0, on
// both sides (oracle/lmik_oracle.c ORC_TOUCH).
constexpr float kTouch = 1e-6f;

// gradient of a point rigidly attached to moving link `link`, projected on n:  n . d(c)/dq_j  for every joint j
template <class RB>
__device__ __forceinline__ void point_grad(const RB& rb, int link, const float (&n)[3], const float (&c)[3],
                                           const float (&ax)[RB::D][3], const float (&og)[RB::D][3], float sign,
                                           float (&g)[RB::D]) {
#pragma unroll
    for (int j = 0; j < RB::D; ++j) {
        float v;
        if (!rb.pris(j)) {
            const float rx = c[0] - og[j][0], ry = c[1] - og[j][1], rz = c[2] - og[j][2];
            const float cx = ax[j][1] * rz - ax[j][2] * ry, cy = ax[j][2] * rx - ax[j][0] * rz,
                        cz = ax[j][0] * ry - ax[j][1] * rx;
            v = n[0] * cx + n[1] * cy + n[2] * cz;
        } else {
            v = n[0] * ax[j][0] + n[1] * ax[j][1] + n[2] * ax[j][2];
        }
        g[j] += (j <= link) ? sign * v : 0.f;
    }
}

// M (upper triangle, row-major i <= j) += w * g g^T ;  m += wm * g
template <int D>
__device__ __forceinline__ void rank1(float (&M)[D * (D + 1) / 2], float (&m)[D], const float (&g)[D], float w, float wm) {
    int k = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const float wi = w * g[i];
#pragma unroll
        for (int j = i; j < D; ++j) {
            M[k] = CPPF_FMA(wi, g[j], M[k]);
            ++k;
        }
        m[i] = CPPF_FMA(wm, g[i], m[i]);
    }
}

// Robot.self_collision_distances_jacobian(x) / Robot.env_collision_distances_jacobian(x, cuboid, Tcuboid) (jrl; call sites
// cppflow/optimization_utils.py:670, 710): d(distance)/dq per pair / per capsule with the closest points held fixed on their
// links,  n . (dc1/dq - dc2/dq)  (0 where the segments touch: the direction is undefined).  Same FK, closest-point and
// gradient code as the coupled step (full_blocks_kernel), which only ever needs the colliding ones.
template <int D, bool ENV>
__global__ __launch_bounds__(kBlock) void distance_jacobians_kernel(const ChainK ch, const CollK co, int n,
                                                                    const float* __restrict__ x, float blo0, float blo1,
                                                                    float blo2, float bhi0, float bhi1, float bhi2,
                                                                    float* __restrict__ jac, float* __restrict__ dists) {
    extern __shared__ float lds[];
    using RB = DynRobot<D>;
    const RB rb{ch, co};
    const int tid = threadIdx.x;
    const size_t row = (size_t)blockIdx.x * kBlock + tid;
    if (row >= (size_t)n) return;
    float q[D], R[9], p[3], ax[D][3], og[D][3];
    load_x<D>(x, row, q);
    frame_identity(R, p);
    for (int c = co.cap_begin[0]; c < co.cap_begin[1]; ++c) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lds[(c * 6 + k) * kBlock + tid] = co.cap_c[c][k];
            lds[(c * 6 + 3 + k) * kBlock + tid] = co.cap_h[c][k];
        }
    }
#pragma unroll
    for (int j = 0; j < D; ++j) {
        fk_fixed_joint(rb, j, R, p);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            ax[j][i] = R[3 * i + 2];
            og[j][i] = p[i];
        }
        fk_joint(R, p, rb.pris(j), q[j]);
        for (int c = co.cap_begin[j + 1]; c < co.cap_begin[j + 2]; ++c) {
            float wc[3], wh[3];
            xform_point(R, p, co.cap_c[c][0], co.cap_c[c][1], co.cap_c[c][2], wc);
            xform_dir(R, co.cap_h[c][0], co.cap_h[c][1], co.cap_h[c][2], wh);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                lds[(c * 6 + k) * kBlock + tid] = wc[k];
                lds[(c * 6 + 3 + k) * kBlock + tid] = wh[k];
            }
        }
    }
    const int count = ENV ? co.ncaps : co.npairs;
    for (int e = 0; e < count; ++e) {
        float nrm[3] = {0.f, 0.f, 0.f}, g[D], sd, radius;
#pragma unroll
        for (int j = 0; j < D; ++j) g[j] = 0.f;
        if constexpr (ENV) {
            const float lo[3] = {blo0, blo1, blo2}, hi[3] = {bhi0, bhi1, bhi2};
            float wc[3], wh[3], cs[3], cb[3];
            lds_capsule(lds, tid, e, wc, wh);
            sd = seg_box_closest(wc, wh, lo, hi, cs, cb);
            radius = co.cap_r[e];
            if (sd > kTouch) {
#pragma unroll
                for (int i = 0; i < 3; ++i) nrm[i] = (cs[i] - cb[i]) / sd;
            }
            point_grad<RB>(rb, co.cap_link[e], nrm, cs, ax, og, 1.f, g);
        } else {
            const int a = co.pair_a[e], b = co.pair_b[e];
            float ca[3], ha[3], cb2[3], hb[3], c1[3], c2[3];
            lds_capsule(lds, tid, a, ca, ha);
            lds_capsule(lds, tid, b, cb2, hb);
            sd = seg_seg_closest(ca, ha, cb2, hb, co.cap_a[a], co.cap_ia[a], co.cap_a[b], co.cap_ia[b], c1, c2);
            radius = co.cap_r[a] + co.cap_r[b];
            if (sd > kTouch) {
#pragma unroll
                for (int i = 0; i < 3; ++i) nrm[i] = (c1[i] - c2[i]) / sd;
            }
            point_grad<RB>(rb, co.cap_link[a], nrm, c1, ax, og, 1.f, g);
            point_grad<RB>(rb, co.cap_link[b], nrm, c2, ax, og, -1.f, g);
        }
        float* o = jac + (row * count + e) * D;
#pragma unroll
        for (int j = 0; j < D; ++j) o[j] = g[j];
        if (dists) dists[row * count + e] = sd - radius;
    }
}

// A_tt's diagonal and b_t's differencing / virtual-config terms (optimization_utils.py:574-640) of one row, in the same
// operation order as the elimination kernels' own loops (prm.fold), then the row's packed block goes out
template <class RB>
__device__ __forceinline__ void full_block_store(const RB& rb, const FullK& prm, size_t row, const float (&q)[RB::D],
                                                 const float* __restrict__ x, const float* __restrict__ xv,
                                                 float (&M)[RB::D * (RB::D + 1) / 2], float (&m)[RB::D],
                                                 float* __restrict__ blocks, float* __restrict__ w2next) {
    constexpr int D = RB::D, NT = D * (D + 1) / 2;
    if (prm.fold && prm.diff_mode != 0) {
        // differencing rows carry individual weights: the two rows this waypoint is part of, (t, j) with its successor and
        // (t-1, j) with its predecessor (the latter recomputed here: the same value bit for bit that waypoint t-1 computes), and
        // the coupling of (t, j) goes to w2next for the elimination (full_solve_kernel<D, true>)
        const int t = (int)(row % (size_t)prm.W), T = prm.W;
        const bool has_next = t + 1 < T, has_prev = t > 0;
        const bool vq = prm.use_vq && (t < prm.n_vq || t >= T - prm.n_vq);
        const float beta = prm.a_vq * prm.a_diff, beta2 = beta * beta;
        int k = 0;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            float wn = 0.f, rn = 0.f, wp = 0.f, rp = 0.f;
            float w3[3] = {(prm.use_diff && has_next) ? x[(row + 1) * D + j] - q[j] : 0.f,
                           (prm.use_diff && has_prev) ? q[j] - x[(row - 1) * D + j] : 0.f, (vq && xv) ? q[j] - xv[row * D + j] : 0.f};
            wrap_pi_all<3>(w3);
            if (prm.use_diff && has_next) diff_row_options(prm, rb.pris(j), w3[0], wn, rn);
            if (prm.use_diff && has_prev) diff_row_options(prm, rb.pris(j), w3[1], wp, rp);
            M[k] += wn + wp + (vq ? beta2 : 0.f) + prm.lm_lambda;
            k += D - j;
            m[j] = CPPF_FMA(wn, rn, m[j]);
            m[j] = CPPF_FMA(-wp, rp, m[j]);
            if (vq && xv) m[j] = CPPF_FMA(-beta2, w3[2], m[j]);
            w2next[row * D + j] = wn;
        }
    } else if (prm.fold) {
        const int t = (int)(row % (size_t)prm.W), T = prm.W;
        const bool has_next = t + 1 < T, has_prev = t > 0;
        const bool vq = prm.use_vq && (t < prm.n_vq || t >= T - prm.n_vq);
        const float beta = prm.a_vq * prm.a_diff, beta2 = beta * beta;
        // the wrapped joint changes to the successor / from the predecessor / from the virtual configuration, each set through
        // wrap_pi_all (one rare branch per set instead of one per joint)
        float wn[D], wp[D], wv[D];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            wn[j] = has_next ? x[(row + 1) * D + j] - q[j] : 0.f;
            wp[j] = has_prev ? q[j] - x[(row - 1) * D + j] : 0.f;
            wv[j] = (vq && xv) ? q[j] - xv[row * D + j] : 0.f;
        }
        wrap_pi_all<D>(wn);
        wrap_pi_all<D>(wp);
        wrap_pi_all<D>(wv);
        int k = 0;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const float a = prm.use_diff ? prm.a_diff * (rb.pris(j) ? prm.a_diff_pris : 1.f) : 0.f;
            const float a2 = a * a;
            M[k] += ((has_next ? 1.f : 0.f) + (has_prev ? 1.f : 0.f)) * a2 + (vq ? beta2 : 0.f) + prm.lm_lambda;
            k += D - j;
            if (has_next) m[j] = CPPF_FMA(a2, wn[j], m[j]);
            if (has_prev) m[j] = CPPF_FMA(-a2, wp[j], m[j]);
            if (vq && xv) m[j] = CPPF_FMA(-beta2, wv[j], m[j]);
        }
    }
    float* o = blocks + row * (NT + D);
#pragma unroll
    for (int k = 0; k < NT; ++k) o[k] = M[k];
#pragma unroll
    for (int j = 0; j < D; ++j) o[NT + j] = m[j];
}

// RB = StaRobot<...>: a screening pass first -- capsule FK in registers, the wave-uniform broad phase, then the cheap squared
// distance against the (slightly widened) tabulated thresholds -- leaves two wavefront-uniform bit sets of the pairs /
// (cuboid, capsule) tests in which SOME lane may penetrate.  A wavefront with none (the common case on a planner's paths)
// writes its blocks straight away; the others run the general pass below over the flagged tests only (joint axes, closest
// points, gradients: ~3x the registers).  The general pass decides penetration exactly as before, so the screening changes no
// result -- it only has to be a superset, which the 1e-3 widening of the thresholds makes it.
// OCC4: held to 128 VGPRs (the rare gradient path spills ~23 registers to scratch) so that all four wavefronts per SIMD of a
// 262 144-row launch are resident.  Small launches use the unconstrained build: a lone wavefront pays a memory round trip per
// scratch access (one trajectory x 256 waypoints: 68 -> 66.6 us Panda, 95.7 -> 91.3 Fetch) and a scratch-using launch
// dispatches more slowly.
// Wavefronts per SIMD of the large-launch build (OCC != 0; n >= 131 072 rows, where residency is what hides the capsule stage's
// latency): four where the kernel fits 128 registers without scratch (Panda, Fetch, generic chains up to 6 joints), three (168
// registers) for FetchArm and the generic 7- / 8-joint kernels, which needed 20 / 48 / 144 B of scratch per lane at four.
template <class RB>
struct full_blocks_fits_128 {
    static constexpr bool value = true;
};
template <>
struct full_blocks_fits_128<StaRobot<gen::FetchArm>> {
    static constexpr bool value = false;
};
template <class RB>
constexpr int full_blocks_occ() {
    if (RB::D > 8) return 0;
    if (RB::kStatic) return full_blocks_fits_128<RB>::value ? 4 : 3;
    return RB::D <= 6 ? 4 : 3;
}
template <class RB, int OCC = 0>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(OCC ? OCC : 1, OCC ? OCC : 8))) void full_blocks_kernel(const ChainK ch, const CollK co, const FullK prm,
                                                             const float* __restrict__ x,
                                                             const float* __restrict__ target,
                                                             const float* __restrict__ xv, float* __restrict__ blocks,
                                                             float* __restrict__ w2next) {
    extern __shared__ float lds[];
    constexpr int D = RB::D;
    constexpr int NT = D * (D + 1) / 2;
    const RB rb{ch, co};
    const int tid = threadIdx.x;
    const size_t row = (size_t)blockIdx.x * kBlock + tid;
    const size_t n = (size_t)prm.S * prm.W;
    if (row >= n) return;
    float q[D];
    load_x<D>(x, row, q);
    unsigned long long near_self = ~0ull, near_env = ~0ull;  // wavefront-uniform
    if constexpr (RB::kStatic) {
        using T = typename RB::Table;
        constexpr int L = T::L > 0 ? T::L : 1;
        static_assert(T::P <= 64, "pair set is a 64-bit mask");
        near_self = 0ull;
        const bool env_bits = prm.use_env && co.nobs * T::L <= 64;
        if (env_bits || !prm.use_env) near_env = 0ull;
        if (prm.use_self || env_bits) {
            float R[9], p[3], wc[L][3], wh[L][3];
            capsule_fk_static<RB>(rb, q, R, p, wc, wh);
            if (prm.use_self) {
#pragma unroll
                for (int pi = 0; pi < T::P; ++pi) {
                    const int a = T::pair_a[pi], b = T::pair_b[pi];
                    if (cull_far(mid_dist2(wc[a], wc[b]), T::pair_cull[pi])) continue;
                    const float d2 = seg_seg_dist2(wc[a], wh[a], wc[b], wh[b], T::cap_a[a], T::cap_ia[a], T::cap_a[b], T::cap_ia[b]);
                    if (__builtin_amdgcn_ballot_w64(d2 < 1.001f * T::pair_thr[pi] + 1e-12f)) near_self |= 1ull << pi;
                }
            }
            if (env_bits) {
#pragma unroll
                for (int c = 0; c < T::L; ++c) {
                    const uint32_t reach = cuboids_in_reach(co, wc[c], T::cap_cull[c]);
                    if (reach == 0u) continue;
                    for (int o = 0; o < co.nobs; ++o) {
                        if (!((reach >> o) & 1u)) continue;
                        const float d2 = seg_box_dist2(wc[c], wh[c], co.obs_lo[o], co.obs_hi[o]);
                        if (__builtin_amdgcn_ballot_w64(d2 < 1.001f * T::cap_thr[c] + 1e-12f)) near_env |= 1ull << (o * T::L + c);
                    }
                }
            }
        }
        if (!prm.use_pose && near_self == 0ull && near_env == 0ull) {
            float M[NT], m[D];
#pragma unroll
            for (int k = 0; k < NT; ++k) M[k] = 0.f;
#pragma unroll
            for (int j = 0; j < D; ++j) m[j] = 0.f;
            full_block_store<RB>(rb, prm, row, q, x, xv, M, m, blocks, w2next);
            return;
        }
    }
    float R[9], p[3], ax[D][3], og[D][3], M[NT], m[D];
#pragma unroll
    for (int k = 0; k < NT; ++k) M[k] = 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) m[j] = 0.f;

    // FK with joint axes / origins, capsule end points to LDS (same canonical chain as everywhere else)
    frame_identity(R, p);
    for (int c = co.cap_begin[0]; c < co.cap_begin[1]; ++c) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lds[(c * 6 + k) * kBlock + tid] = co.cap_c[c][k];
            lds[(c * 6 + 3 + k) * kBlock + tid] = co.cap_h[c][k];
        }
    }
#pragma unroll
    for (int j = 0; j < D; ++j) {
        fk_fixed_joint(rb, j, R, p);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            ax[j][i] = R[3 * i + 2];
            og[j][i] = p[i];
        }
        fk_joint(R, p, rb.pris(j), q[j]);
        for (int c = co.cap_begin[j + 1]; c < co.cap_begin[j + 2]; ++c) {
            float wc[3], wh[3];
            xform_point(R, p, co.cap_c[c][0], co.cap_c[c][1], co.cap_c[c][2], wc);
            xform_dir(R, co.cap_h[c][0], co.cap_h[c][1], co.cap_h[c][2], wh);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                lds[(c * 6 + k) * kBlock + tid] = wc[k];
                lds[(c * 6 + 3 + k) * kBlock + tid] = wh[k];
            }
        }
    }

    if (prm.use_pose) {  // optimization_utils.py:503-543
        float Re[9], pe[3], Rt[9], tt[3], J[6][D], e[6];
#pragma unroll
        for (int k = 0; k < 9; ++k) Re[k] = R[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) pe[k] = p[k];
        fk_fixed_ee(rb, Re, pe);
        load_target(target, (int)(row % (size_t)prm.W), Rt, tt);
        pose_error(Rt, tt, Re, pe, e);
        jacobian_from_axes<RB>(rb, pe, ax, og, J);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            float a = i < 3 ? prm.a_rot : prm.a_pos;
            // pose_do_scale_down_satisfied (optimization_utils.py:288-333, applied BEFORE the alphas at :514-533): a row whose
            // unscaled |error| is below the threshold is scaled down, r and J alike
            if (prm.pose_scale_satisfied && fabsf(e[i]) < (i < 3 ? prm.pose_thr_rad : prm.pose_thr_m)) a *= prm.pose_scale;
            float g[D];
#pragma unroll
            for (int j = 0; j < D; ++j) g[j] = a * J[i][j];
            rank1<D>(M, m, g, 1.f, a * e[i]);
        }
    }
    if (prm.use_self) {  // :645-680: rows where -alpha * dist > 0
        const float w = prm.a_self * prm.a_self;
        for (int pi = 0; pi < co.npairs; ++pi) {
            if (!((near_self >> (pi & 63)) & 1ull)) continue;  // screened out above (specialised robots; else all ones)
            const int a = co.pair_a[pi], b = co.pair_b[pi];
            float ca[3], ha[3], cb2[3], hb[3], c1[3], c2[3];
            lds_capsule(lds, tid, a, ca, ha);
            lds_capsule(lds, tid, b, cb2, hb);
            // broad phase (cull_far): only pairs that penetrate contribute rows, and a far pair cannot penetrate
            if (cull_far(mid_dist2(ca, cb2), co.pair_cull[pi])) continue;
            const float sd = seg_seg_closest(ca, ha, cb2, hb, co.cap_a[a], co.cap_ia[a], co.cap_a[b], co.cap_ia[b], c1, c2);
            const float dist = sd - (co.cap_r[a] + co.cap_r[b]);
            if (dist < 0.f) {
                float nrm[3] = {0.f, 0.f, 0.f}, g[D];
                if (sd > kTouch) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) nrm[i] = (c1[i] - c2[i]) / sd;
                }
#pragma unroll
                for (int j = 0; j < D; ++j) g[j] = 0.f;
                point_grad<RB>(rb, co.cap_link[a], nrm, c1, ax, og, 1.f, g);
                point_grad<RB>(rb, co.cap_link[b], nrm, c2, ax, og, -1.f, g);
                rank1<D>(M, m, g, w, -w * dist);
            }
        }
    }
    if (prm.use_env) {  // :685-727
        const float w = prm.a_env * prm.a_env;
        for (int o = 0; o < co.nobs; ++o)
            for (int c = 0; c < co.ncaps; ++c) {
                if (!((near_env >> ((o * co.ncaps + c) & 63)) & 1ull)) continue;
                float wc[3], wh[3], cs[3], cb[3];
                lds_capsule(lds, tid, c, wc, wh);
                if (cull_far(point_box_dist2(wc, co.obs_lo[o], co.obs_hi[o]), co.cap_cull[c])) continue;
                const float sd = seg_box_closest(wc, wh, co.obs_lo[o], co.obs_hi[o], cs, cb);
                const float dist = sd - co.cap_r[c];
                if (dist < 0.f) {
                    float nrm[3] = {0.f, 0.f, 0.f}, g[D];
                    if (sd > kTouch) {
#pragma unroll
                        for (int i = 0; i < 3; ++i) nrm[i] = (cs[i] - cb[i]) / sd;
                    }
#pragma unroll
                    for (int j = 0; j < D; ++j) g[j] = 0.f;
                    point_grad<RB>(rb, co.cap_link[c], nrm, cs, ax, og, 1.f, g);
                    rank1<D>(M, m, g, w, -w * dist);
                }
            }
    }
    full_block_store<RB>(rb, prm, row, q, x, xv, M, m, blocks, w2next);
}

// inverse of a symmetric positive definite D x D matrix (full storage in, full storage out) by Cholesky; pivots floored
template <int D>
__device__ __forceinline__ void spd_inverse(const float (&A)[D][D], float floor_v, float (&G)[D][D]) {
    float L[D][D], Li[D][D], inv[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        float s = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) s = CPPF_FMA(-L[j][k], L[j][k], s);
        s = fmaxf(s, floor_v);
        inv[j] = __frsqrt_rn(s);
        L[j][j] = s * inv[j];
#pragma unroll
        for (int i = j + 1; i < D; ++i) {
            float t = A[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) t = CPPF_FMA(-L[i][k], L[j][k], t);
            L[i][j] = t * inv[j];
        }
    }
    // Li = L^-1 (lower triangular)
#pragma unroll
    for (int j = 0; j < D; ++j) {
        Li[j][j] = inv[j];
#pragma unroll
        for (int i = j + 1; i < D; ++i) {
            float t = 0.f;
#pragma unroll
            for (int k = j; k < i; ++k) t = CPPF_FMA(-L[i][k], Li[k][j], t);
            Li[i][j] = t * inv[i];
        }
    }
    // G = Li^T Li
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = i; j < D; ++j) {
            float t = 0.f;
#pragma unroll
            for (int k = j; k < D; ++k) t = CPPF_FMA(Li[k][i], Li[k][j], t);
            G[i][j] = t;
            G[j][i] = t;
        }
}

// VAR: the differencing rows carry individual weights (the "satisfied" options): the blocks arrive with every waypoint-local term
// folded in (full_block_store) and the coupling of waypoints t and t + 1 is -diag(w2next[t]) instead of the constant -diag(a^2).
template <int D, bool VAR = false>
__global__ __launch_bounds__(64) void full_solve_kernel(const ChainK ch, const FullK prm, const float* __restrict__ x,
                                                        const float* __restrict__ xv, const float* __restrict__ blocks,
                                                        float* __restrict__ workG, float* __restrict__ worky,
                                                        float* __restrict__ x_out, const float* __restrict__ w2next) {
    constexpr int NT = D * (D + 1) / 2;
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= prm.S) return;
    const int T = prm.W;
    const size_t base = (size_t)s * T;
    float a2[D];  // a_j^2 = (alpha_differencing * prismatic scaling)^2  (optimization_utils.py:607-612)
#pragma unroll
    for (int j = 0; j < D; ++j) {
        const float a = prm.use_diff ? prm.a_diff * (((ch.pris_mask >> j) & 1u) ? prm.a_diff_pris : 1.f) : 0.f;
        a2[j] = a * a;
    }
    const float beta = prm.a_vq * prm.a_diff, beta2 = beta * beta;

    float G[D][D], y[D], xp[D], xc[D], xn[D];
    load_x<D>(x, base, xc);
    // ---- forward elimination
    for (int t = 0; t < T; ++t) {
        const float* blk = blocks + (base + t) * (NT + D);
        float A[D][D], b[D];
        {
            int k = 0;
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = i; j < D; ++j) {
                    const float v = blk[k++];
                    A[i][j] = v;
                    A[j][i] = v;
                }
#pragma unroll
            for (int j = 0; j < D; ++j) b[j] = blk[NT + j];
        }
        const bool has_next = t + 1 < T, has_prev = t > 0;
        if constexpr (VAR) {
            if (has_prev) load_x<D>(w2next, base + t - 1, a2);  // the coupling with the predecessor
        } else {
            if (has_next) load_x<D>(x, base + t + 1, xn);
            const float cnt = (has_next ? 1.f : 0.f) + (has_prev ? 1.f : 0.f);
            const bool vq = prm.use_vq && (t < prm.n_vq || t >= T - prm.n_vq);
            float wn[D], wp[D];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                wn[j] = has_next ? xn[j] - xc[j] : 0.f;
                wp[j] = has_prev ? xc[j] - xp[j] : 0.f;
            }
            wrap_pi_all<D>(wn);
            wrap_pi_all<D>(wp);
#pragma unroll
            for (int j = 0; j < D; ++j) {
                A[j][j] += cnt * a2[j] + (vq ? beta2 : 0.f) + prm.lm_lambda;
                // J^T r of the differencing rows: +a^2 w_t at (t,j), -a^2 w_{t-1} at (t,j)   (w = wrapped joint change)
                if (has_next) b[j] = CPPF_FMA(a2[j], wn[j], b[j]);
                if (has_prev) b[j] = CPPF_FMA(-a2[j], wp[j], b[j]);
            }
            if (vq) {  // r = beta * wrap(x - x_virtual), J = -beta I  (optimization_utils.py:430-484)
                float v[D];
                if (xv) {
                    load_x<D>(xv, base + t, v);
#pragma unroll
                    for (int j = 0; j < D; ++j) v[j] = xc[j] - v[j];
                    wrap_pi_all<D>(v);
                }
#pragma unroll
                for (int j = 0; j < D; ++j) b[j] = CPPF_FMA(-beta2, xv ? v[j] : 0.f, b[j]);
            }
        }
        if (has_prev) {
            // D' = A - E G E ,  y = b - E G y_prev   with E = -diag(a2)
            float Gy[D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    acc = CPPF_FMA(G[i][j], y[j], acc);
                    A[i][j] = CPPF_FMA(-(a2[i] * a2[j]), G[i][j], A[i][j]);
                }
                Gy[i] = acc;
            }
#pragma unroll
            for (int i = 0; i < D; ++i) y[i] = CPPF_FMA(a2[i], Gy[i], b[i]);  // b - (-a2) * (G y_prev)
        } else {
#pragma unroll
            for (int i = 0; i < D; ++i) y[i] = b[i];
        }
        spd_inverse<D>(A, prm.lm_lambda, G);
        float* gout = workG + (base + t) * NT;
        float* yout = worky + (base + t) * D;
        {
            int k = 0;
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = i; j < D; ++j) gout[k++] = G[i][j];
#pragma unroll
            for (int j = 0; j < D; ++j) yout[j] = y[j];
        }
#pragma unroll
        for (int j = 0; j < D; ++j) {
            xp[j] = xc[j];
            xc[j] = xn[j];
        }
    }
    // ---- back substitution: delta_t = G_t (y_t - E delta_{t+1}) = G_t (y_t + a2 .* delta_{t+1})
    float dl[D];
#pragma unroll
    for (int j = 0; j < D; ++j) dl[j] = 0.f;
    for (int t = T - 1; t >= 0; --t) {
        const float* gin = workG + (base + t) * NT;
        const float* yin = worky + (base + t) * D;
        float rhs[D], Gt[D][D];
        {
            int k = 0;
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = i; j < D; ++j) {
                    const float v = gin[k++];
                    Gt[i][j] = v;
                    Gt[j][i] = v;
                }
        }
        if constexpr (VAR) {
            if (t + 1 < T) load_x<D>(w2next, base + t, a2);  // the coupling with the successor
        }
#pragma unroll
        for (int j = 0; j < D; ++j) rhs[j] = (t + 1 < T) ? CPPF_FMA(a2[j], dl[j], yin[j]) : yin[j];
        float nd[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < D; ++j) acc = CPPF_FMA(Gt[i][j], rhs[j], acc);
            nd[i] = acc;
        }
        float xr[D];
        load_x<D>(x, base + t, xr);
#pragma unroll
        for (int j = 0; j < D; ++j) {
            dl[j] = nd[j];
            xr[j] += nd[j];  // optimization.py:113: x + delta_x
        }
        store_x<D>(x_out, base + t, xr);
    }
}

// Wavefront-parallel form of full_solve_kernel for D <= 8: one wavefront per trajectory, lane l <-> element
// (i, j) = (l >> 3, l & 7) of the 8 x 8 matrix that holds the d x d block padded with the identity.  The elimination is
// inherently sequential in t, so the only parallelism inside a trajectory is inside the d x d operations: the block inverse
// is an in-place Gauss-Jordan sweep (per pivot: one v_readlane for the pivot, two cross-lane reads for its row and column,
// one fused update on all 64 lanes), matrix-vector products are a lane-local multiply plus a 3-step xor-butterfly over the
// row or column bits.  Operands of step t+1 are requested before step t is computed, so their latency hides behind it.
template <int D>
__global__ __launch_bounds__(64) void full_solve_wave_kernel(const ChainK ch, const FullK prm, const float* __restrict__ x,
                                                             const float* __restrict__ xv,
                                                             const float* __restrict__ blocks, float* __restrict__ workG,
                                                             float* __restrict__ worky, float* __restrict__ x_out) {
    static_assert(D <= 8, "one 8x8 tile per wavefront");
    constexpr int NT = D * (D + 1) / 2;
    const int s = blockIdx.x;
    if (s >= prm.S) return;
    const int lane = threadIdx.x, i = lane >> 3, j = lane & 7;
    const bool in = i < D && j < D, colv = j < D;
    const int T = prm.W;
    const size_t base = (size_t)s * T;
    const int ii = i < j ? i : j, jj = i < j ? j : i;
    const int tri = in ? ii * D - (ii * (ii - 1)) / 2 + (jj - ii) : 0;  // offset of (min, max) in the packed upper triangle
    auto a2_of = [&](int c) {
        if (!prm.use_diff || c >= D) return 0.f;
        const float a = prm.a_diff * (((ch.pris_mask >> c) & 1u) ? prm.a_diff_pris : 1.f);
        return a * a;
    };
    const float a2i = a2_of(i), a2j = a2_of(j);
    const float beta = prm.a_vq * prm.a_diff, beta2 = beta * beta;
    const int jc = colv ? j : 0;  // padded lanes read column 0 (their values are never used)

    // ---- forward elimination
    float G = 0.f, y_row = 0.f;
    float xp = 0.f, xc = x[(base + 0) * D + jc], xn = T > 1 ? x[(base + 1) * D + jc] : 0.f;
    float Mij = blocks[(base + 0) * (NT + D) + tri], bj = blocks[(base + 0) * (NT + D) + NT + jc];
    float vj = xv ? xv[(base + 0) * D + jc] : 0.f;
    for (int t = 0; t < T; ++t) {
        // request the operands of step t+1 now
        float nM = 0.f, nb = 0.f, nx2 = 0.f, nv = 0.f;
        if (t + 1 < T) {
            nM = blocks[(base + t + 1) * (NT + D) + tri];
            nb = blocks[(base + t + 1) * (NT + D) + NT + jc];
            if (xv) nv = xv[(base + t + 1) * D + jc];
        }
        if (t + 2 < T) nx2 = x[(base + t + 2) * D + jc];

        const bool has_next = t + 1 < T, has_prev = t > 0;
        const bool vq = prm.use_vq && (t < prm.n_vq || t >= T - prm.n_vq);
        float A = in ? Mij : (i == j ? 1.f : 0.f);
        if (in && i == j) A += ((has_next ? 1.f : 0.f) + (has_prev ? 1.f : 0.f)) * a2j + (vq ? beta2 : 0.f) + prm.lm_lambda;
        float b = colv ? bj : 0.f;
        if (colv) {
            float w3[3] = {has_next ? xn - xc : 0.f, has_prev ? xc - xp : 0.f, (vq && xv) ? xc - vj : 0.f};
            wrap_pi_all<3>(w3);  // (straight-line; this sits on the critical path of every elimination step)
            if (has_next) b = CPPF_FMA(a2j, w3[0], b);
            if (has_prev) b = CPPF_FMA(-a2j, w3[1], b);
            if (vq && xv) b = CPPF_FMA(-beta2, w3[2], b);
        }
        float ycol = b;
        if (has_prev) {
            A = CPPF_FMA(-(a2i * a2j), G, A);  // D' = A - E G E   (padding: a2 = 0)
            float pr = G * y_row;              // G_ij y_i, summed over i -> (G y)_j in every lane of column j
            pr += __shfl_xor(pr, 8, 64);
            pr += __shfl_xor(pr, 16, 64);
            pr += __shfl_xor(pr, 32, 64);
            ycol = CPPF_FMA(a2j, pr, b);  // y = b - E (G y_prev)
        }
        // in-place Gauss-Jordan inverse (SPD: no pivoting; pivots floored like the Cholesky pivots of the per-lane kernel)
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const float pv = fmaxf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(A), k * 9)), prm.lm_lambda);
            const float pinv = 1.f / pv;
            const float rk = __shfl(A, k * 8 + j, 64);        // A_kj
            const float ck = __shfl(A, (lane & 56) + k, 64);  // A_ik
            float nvl = CPPF_FMA(-(ck * pinv), rk, A);
            nvl = (i == k) ? rk * pinv : nvl;
            nvl = (j == k) ? -(ck * pinv) : nvl;
            nvl = (i == k && j == k) ? pinv : nvl;
            A = nvl;
        }
        G = A;
        if (in && i <= j) workG[(base + t) * NT + tri] = G;
        if (i == 0 && colv) worky[(base + t) * D + j] = ycol;
        y_row = __shfl(ycol, j * 8 + i, 64);  // lane (i,j) takes y_i from column i
        xp = xc, xc = xn, xn = nx2;
        Mij = nM, bj = nb, vj = nv;
    }
    // ---- back substitution: delta_t = G_t (y_t + a2 .* delta_{t+1})
    float dl = 0.f;  // delta_{t+1}, column-replicated
    float Gn = in ? workG[(base + T - 1) * NT + tri] : (i == j ? 1.f : 0.f);
    float yn = colv ? worky[(base + T - 1) * D + j] : 0.f;
    float xr = (in && j == 0) ? x[(base + T - 1) * D + i] : 0.f;
    for (int t = T - 1; t >= 0; --t) {
        const float Gt = Gn, yt = yn, xt = xr;
        if (t > 0) {
            Gn = in ? workG[(base + t - 1) * NT + tri] : (i == j ? 1.f : 0.f);
            yn = colv ? worky[(base + t - 1) * D + j] : 0.f;
            xr = (in && j == 0) ? x[(base + t - 1) * D + i] : 0.f;
        }
        const float rhs = (t + 1 < T) ? CPPF_FMA(a2j, dl, yt) : yt;
        float pr = Gt * rhs;  // G_ij rhs_j, summed over j -> delta_i in every lane of row i
        pr += __shfl_xor(pr, 1, 64);
        pr += __shfl_xor(pr, 2, 64);
        pr += __shfl_xor(pr, 4, 64);
        if (in && j == 0) x_out[(base + t) * D + i] = xt + pr;  // optimization.py:113: x + delta_x
        dl = __shfl(pr, j * 8 + i, 64);
    }
}

// Row-per-lane form for D <= 8: EIGHT trajectories per wavefront, lane 8 g + r holds ROW r of trajectory g's 8 x 8 padded block in
// registers.  The wavefront-per-trajectory kernel above spends its step in dependent ds_bpermute round trips (two per pivot,
// ~0.1 us each: 0.7 us per waypoint whatever else happens); here every cross-lane read is a broadcast inside a group of eight
// consecutive lanes from a compile-time lane, which is two DPP instructions (quad_perm picks lane k & 3 of each quad, a
// bank-masked row_shr:4 / row_shl:4 copies the right quad over the other) and never touches the LDS pipe.  Gauss-Jordan without
// pivoting (SPD; pivots floored like everywhere else): per pivot the pivot row is broadcast (16 DPP), and one FMA per entry with
// the factor f = A_rk / p (f = 1 - 1/p in the pivot's own lane, which turns its row into row / p) updates the whole row.
// ~300 instructions per waypoint forward, ~45 back, one wavefront for eight trajectories: 1024 trajectories are 128 wavefronts.
// d > 8 (up to 16): the same with SIXTEEN lanes per trajectory (one DPP row; four trajectories per wavefront) -- a broadcast is
// then quad_perm + three bank-masked row shifts (one per other quad of the row): four DPP instructions.
template <int D>
inline constexpr int kRowsGW = D <= 8 ? 8 : 16;  // lanes per trajectory = padded block size

template <int GW, int K>
__device__ __forceinline__ float g_bcast(float x) {  // value of lane (GW g + K) in every lane of group g
    int t = __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), (K & 3) * 0x55, 0xf, 0xf, true);
    if constexpr (GW == 8) {
        if constexpr (K < 4)  // the source sits in the even quad of its group: odd quads (banks 1, 3) read lane - 4
            t = __builtin_amdgcn_update_dpp(t, t, 0x114, 0xf, 0xa, false);
        else                  // ... in the odd quad: even quads (banks 0, 2) read lane + 4
            t = __builtin_amdgcn_update_dpp(t, t, 0x104, 0xf, 0x5, false);
    } else {
        constexpr int qk = K >> 2;  // the source's quad; every other quad q of the row reads 4 |q - qk| lanes down / up
        if constexpr (qk != 0) t = __builtin_amdgcn_update_dpp(t, t, 0x100 + 4 * qk, 0xf, 0x1, false);          // row_shl
        if constexpr (qk != 1) t = __builtin_amdgcn_update_dpp(t, t, qk > 1 ? 0x100 + 4 * (qk - 1) : 0x114, 0xf, 0x2, false);
        if constexpr (qk != 2) t = __builtin_amdgcn_update_dpp(t, t, qk > 2 ? 0x104 : 0x110 + 4 * (2 - qk), 0xf, 0x4, false);
        if constexpr (qk != 3) t = __builtin_amdgcn_update_dpp(t, t, 0x110 + 4 * (3 - qk), 0xf, 0x8, false);    // row_shr
    }
    return __builtin_bit_cast(float, t);
}

template <int D, int C = 0>
__device__ __forceinline__ void g8_bcast_all(float x, float (&out)[kRowsGW<D>]) {
    if constexpr (C < D) {
        out[C] = g_bcast<kRowsGW<D>, C>(x);
        g8_bcast_all<D, C + 1>(x, out);
    }
}

template <int D, int K, int C = 0>
__device__ __forceinline__ void g8_bcast_row(const float (&A)[kRowsGW<D>], float (&out)[kRowsGW<D>]) {  // out[c] = A[c] of lane K of the group
    if constexpr (C < D) {
        out[C] = g_bcast<kRowsGW<D>, K>(A[C]);
        g8_bcast_row<D, K, C + 1>(A, out);
    }
}

template <int D, int K = 0>
__device__ __forceinline__ void g8_gauss_jordan(float (&A)[kRowsGW<D>], int r, float floor_) {
    if constexpr (K < D) {
        float pr[kRowsGW<D>];
#pragma unroll
        for (int c = 0; c < kRowsGW<D>; ++c) pr[c] = 0.f;
        // pivot row K, every column
        g8_bcast_row<D, K>(A, pr);
        const float pv = __builtin_amdgcn_fmed3f(pr[K], floor_, INFINITY);  // max(pivot, floor) in one instruction
        const float pinv = __builtin_amdgcn_rcpf(pv);  // 1 ulp; the factorisation's own rounding is of that order per entry
        const bool own = r == K;
        const float f = own ? 1.f - pinv : A[K] * pinv;
#pragma unroll
        for (int c = 0; c < D; ++c) A[c] = CPPF_FMA(-f, pr[c], A[c]);
        A[K] = own ? pinv : -f;
        g8_gauss_jordan<D, K + 1>(A, r, floor_);
    }
}

// The elimination runs from BOTH ends of the path at once (a twisted factorisation): with m = T / 2, one wavefront eliminates
// waypoints 0 .. m-1 upwards ( D'_t = A_t - E G_{t-1} E ), another T-1 .. m+1 downwards ( D''_t = A_t - E G_{t+1} E ); the two
// chains are the same recurrence in mirrored time and touch disjoint waypoints, so they need no communication.  A second
// launch joins them at waypoint m,
//     ( A_m - E G'_{m-1} E - E G''_{m+1} E ) delta_m = b_m - E G'_{m-1} y_{m-1} - E G''_{m+1} z_{m+1}
// (both wavefronts of a pair compute it, redundantly, instead of exchanging it), and substitutes back outwards in both
// directions:  delta_t = G_t ( y_t + a^2 .* delta_{t +- 1} ).  Per-step work is unchanged and the chain each wavefront walks is
// half as long -- the launch is bound by that chain, not by throughput (1024 trajectories are 256 wavefronts on 1024 SIMDs).
// One wavefront per workgroup; the launch reserves enough (unused) LDS per workgroup that at most ceil(#workgroups / 256) of
// them fit a compute unit, which spreads a small launch over distinct compute units.
template <int D>
struct RowsLane {
    int r, rr;
    bool live;  // padded rows / idle groups work on a copy of a valid row and store nothing
    unsigned offM[D], offb, offG, offy;
    float a2r, pc[D];
    size_t ubase;
};

template <int D>
__device__ __forceinline__ RowsLane<D> rows_lane(const FullK& prm, uint32_t pris_mask, int wave, int lane) {
    constexpr int NT = D * (D + 1) / 2, STRIDE = NT + D;
    constexpr int GW = kRowsGW<D>, TPW = 64 / GW;  // lanes per trajectory, trajectories per wavefront
    RowsLane<D> L;
    L.r = lane & (GW - 1);
    const int s_raw = wave * TPW + lane / GW;
    L.live = s_raw < prm.S && L.r < D;
    const int s = s_raw < prm.S ? s_raw : prm.S - 1;
    L.rr = L.r < D ? L.r : 0;
    const int T = prm.W;
    // Addresses = a wavefront-uniform 64-bit part (this wavefront's first trajectory, moved by one waypoint per step) + a
    // 32-bit lane part fixed for the whole kernel (the host bounds 8 W d^2 floats to 2^31 bytes)
    L.ubase = (size_t)wave * TPW * (size_t)T;
    const unsigned lrow = (unsigned)(s - wave * TPW) * (unsigned)T;
#pragma unroll
    for (int c = 0; c < D; ++c) {  // (min(r,c), max(r,c)) in the packed upper triangle
        const int ii = L.rr < c ? L.rr : c, jj = L.rr < c ? c : L.rr;
        L.offM[c] = lrow * STRIDE + (unsigned)(ii * D - (ii * (ii - 1)) / 2 + (jj - ii));
    }
    L.offb = lrow * STRIDE + NT + L.rr, L.offG = lrow * (D * D) + L.rr * D, L.offy = lrow * D + L.rr;
    float a2c[D];
#pragma unroll
    for (int c = 0; c < D; ++c) {
        const float a = prm.use_diff ? prm.a_diff * (((pris_mask >> c) & 1u) ? prm.a_diff_pris : 1.f) : 0.f;
        a2c[c] = a * a;
    }
    L.a2r = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) L.a2r = L.r == c ? a2c[c] : L.a2r;
#pragma unroll
    for (int c = 0; c < D; ++c) L.pc[c] = L.a2r * a2c[c];  // E G E scales entry (r, c) of G by a_r^2 a_c^2
    return L;
}

// A -= E G E (this lane's row) and the return value  (G v)_r  for the vector v held one component per lane
template <int D>
__device__ __forceinline__ float rows_couple(const RowsLane<D>& L, const float (&G)[D], float v, float (&A)[kRowsGW<D>]) {
    float vb[kRowsGW<D>];
#pragma unroll
    for (int c = 0; c < kRowsGW<D>; ++c) vb[c] = 0.f;
    g8_bcast_all<D>(v, vb);
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        acc = CPPF_FMA(G[c], vb[c], acc);
        A[c] = CPPF_FMA(-L.pc[c], G[c], A[c]);
    }
    return acc;
}

template <int D>
__global__ __launch_bounds__(64) void full_rows_eliminate_kernel(const FullK prm, const uint32_t pris_mask,
                                                                 const float* __restrict__ blocks, float* __restrict__ workG,
                                                                 float* __restrict__ worky) {
    static_assert(D <= 16, "one row of the padded block per lane: 8 or 16 lanes per trajectory");
    constexpr int NT = D * (D + 1) / 2, STRIDE = NT + D;
    const int wave = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const int T = prm.W, m = T / 2;
    const int start = dir ? T - 1 : 0, sgn = dir ? -1 : 1, len = dir ? T - 1 - m : m;  // waypoint of step tau: start + sgn tau
    if (len <= 0) return;
    const RowsLane<D> L = rows_lane<D>(prm, pris_mask, wave, threadIdx.x);
    const float* blk = blocks + L.ubase * STRIDE;
    float* gw = workG + L.ubase * (D * D);
    float* yw = worky + L.ubase * D;
    // Block rows are requested PF steps ahead (a step is ~0.8 us of issue; a launch whose workspaces exceed the L2 sees
    // 2-3 us of load latency), into a ring of register sets indexed at compile time.
    constexpr int PF = 4;
    float G[D], y = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) G[c] = 0.f;
    float pM[PF][D], pb[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const float* nxt = blk + (size_t)(start + sgn * (u < len ? u : len - 1)) * STRIDE;
#pragma unroll
        for (int c = 0; c < D; ++c) pM[u][c] = nxt[L.offM[c]];
        pb[u] = nxt[L.offb];
    }
    // A block of PF steps is straight-line code (a step past the chain's end recomputes its last waypoint and stores nothing) and
    // its results are stored together behind it.  Both matter on gfx950: (1) a branch around a step makes the compiler copy
    // the freshly requested rows into the ring's registers at the end of the SAME step (a wait for loads just issued); (2) loads
    // and stores share one counter (vmcnt) whose returns are ordered only within each kind, so with a store in flight a
    // prefetched load can only be waited for by draining the counter -- clustered, the stores cost one drain per block.
    for (int t0 = 0; t0 < len; t0 += PF) {
        float sG[PF][D], sy[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int tau = t0 + u;
            float A[kRowsGW<D>];
#pragma unroll
            for (int c = 0; c < kRowsGW<D>; ++c) A[c] = c < D ? pM[u][c] : 0.f;
            const float b = pb[u];
            {  // step tau + PF into the slot just read (the last PF steps re-read the chain's last waypoint)
                const float* nxt = blk + (size_t)(start + sgn * (tau + PF < len ? tau + PF : len - 1)) * STRIDE;
#pragma unroll
                for (int c = 0; c < D; ++c) pM[u][c] = nxt[L.offM[c]];
                pb[u] = nxt[L.offb];
            }
            // y = b - E (G y_prev),  E = -diag(a^2); at the chain's first step G = 0, y = 0: A and b pass through exactly
            const float ynew = CPPF_FMA(L.a2r, rows_couple<D>(L, G, y, A), b);
            g8_gauss_jordan<D>(A, L.r, prm.lm_lambda);
            y = ynew;
#pragma unroll
            for (int c = 0; c < D; ++c) G[c] = A[c];
#pragma unroll
            for (int c = 0; c < D; ++c) sG[u][c] = G[c];
            sy[u] = y;
        }
        if (L.live) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int tau = t0 + u;
                if (tau < len) {
                    const int t = start + sgn * tau;
                    float* gt = gw + (size_t)t * (D * D);
#pragma unroll
                    for (int c = 0; c < D; ++c) gt[L.offG + c] = sG[u][c];
                    yw[(size_t)t * D + L.offy] = sy[u];
                }
            }
        }
    }
}

template <int D>
__global__ __launch_bounds__(64) void full_rows_substitute_kernel(const FullK prm, const uint32_t pris_mask,
                                                                  const float* __restrict__ x, const float* __restrict__ blocks,
                                                                  const float* __restrict__ workG,
                                                                  const float* __restrict__ worky, float* __restrict__ x_out) {
    constexpr int NT = D * (D + 1) / 2, STRIDE = NT + D;
    const int wave = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const int T = prm.W, m = T / 2;
    const int start = dir ? T - 1 : 0, sgn = dir ? -1 : 1, len = dir ? T - 1 - m : m;
    const RowsLane<D> L = rows_lane<D>(prm, pris_mask, wave, threadIdx.x);
    const float* blk = blocks + L.ubase * STRIDE;
    const float* gw = workG + L.ubase * (D * D);
    const float* yw = worky + L.ubase * D;
    const float* xin = x + L.ubase * D;
    float* xo = x_out + L.ubase * D;
    // ---- the chains' operands, PB steps ahead of their use (a step is ~0.15 us), requested before the join's arithmetic
    constexpr int PB = 16;
    float qG[PB][D], qy[PB], qx[PB];
#pragma unroll
    for (int u = 0; u < PB; ++u) {
        const int tau = len - 1 - u > 0 ? len - 1 - u : 0;
        const int tp = len > 0 ? start + sgn * tau : m;
        const float* gt = gw + (size_t)tp * (D * D);
#pragma unroll
        for (int c = 0; c < D; ++c) qG[u][c] = gt[L.offG + c];
        qy[u] = yw[(size_t)tp * D + L.offy];
        qx[u] = xin[(size_t)tp * D + L.offy];
    }
    // ---- the join at waypoint m
    float dl;
    {
        const float* bm = blk + (size_t)m * STRIDE;
        float A[kRowsGW<D>];
#pragma unroll
        for (int c = 0; c < kRowsGW<D>; ++c) A[c] = c < D ? bm[L.offM[c]] : 0.f;
        float rhs = bm[L.offb];
        const float xm = xin[(size_t)m * D + L.offy];
        if (m > 0) {
            float Gp[D];
            const float* gt = gw + (size_t)(m - 1) * (D * D);
#pragma unroll
            for (int c = 0; c < D; ++c) Gp[c] = gt[L.offG + c];
            rhs = CPPF_FMA(L.a2r, rows_couple<D>(L, Gp, yw[(size_t)(m - 1) * D + L.offy], A), rhs);
        }
        if (m + 1 < T) {
            float Gn[D];
            const float* gt = gw + (size_t)(m + 1) * (D * D);
#pragma unroll
            for (int c = 0; c < D; ++c) Gn[c] = gt[L.offG + c];
            rhs = CPPF_FMA(L.a2r, rows_couple<D>(L, Gn, yw[(size_t)(m + 1) * D + L.offy], A), rhs);
        }
        g8_gauss_jordan<D>(A, L.r, prm.lm_lambda);
        float rb[kRowsGW<D>];
#pragma unroll
        for (int c = 0; c < kRowsGW<D>; ++c) rb[c] = 0.f;
        g8_bcast_all<D>(rhs, rb);
        dl = 0.f;
#pragma unroll
        for (int c = 0; c < D; ++c) dl = CPPF_FMA(A[c], rb[c], dl);
        if (dir == 1 && L.live) xo[(size_t)m * D + L.offy] = xm + dl;  // optimization.py:113: x + delta_x
    }
    // ---- back substitution along this wavefront's chain, from the join outwards.  The PB results of a block are kept in
    // registers and stored together after it: gfx950 counts loads and stores on ONE counter (vmcnt) whose returns are ordered
    // only within each kind, so with a store in flight the compiler can wait for a prefetched load only by draining the counter
    // (s_waitcnt vmcnt(0)) -- a store per step made every step wait for the loads it had just issued, i.e. one memory latency
    // per step whatever the prefetch depth.  Clustered, the stores cost one drain per PB steps.
    for (int t0 = len - 1; t0 >= 0; t0 -= PB) {
        float xs[PB];  // (len = 0: the loop does not run; the prologue above then read waypoint m, harmlessly)
#pragma unroll
        for (int u = 0; u < PB; ++u) {  // straight-line: a step past the chain's end (tau < 0) works on stale operands, stores nothing
            const int tau = t0 - u;
            float Gt[D];
#pragma unroll
            for (int c = 0; c < D; ++c) Gt[c] = qG[u][c];
            const float yt = qy[u], xt = qx[u];
            {
                const int tp = start + sgn * (tau - PB > 0 ? tau - PB : 0);
                const float* gt = gw + (size_t)tp * (D * D);
#pragma unroll
                for (int c = 0; c < D; ++c) qG[u][c] = gt[L.offG + c];
                qy[u] = yw[(size_t)tp * D + L.offy];
                qx[u] = xin[(size_t)tp * D + L.offy];
            }
            float rb[kRowsGW<D>];
#pragma unroll
            for (int c = 0; c < kRowsGW<D>; ++c) rb[c] = 0.f;
            g8_bcast_all<D>(CPPF_FMA(L.a2r, dl, yt), rb);
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) acc = CPPF_FMA(Gt[c], rb[c], acc);
            xs[u] = xt + acc;
            dl = acc;
        }
        if (L.live) {
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int tau = t0 - u;
                if (tau >= 0) xo[(size_t)(start + sgn * tau) * D + L.offy] = xs[u];
            }
        }
    }
}

// Parallel-in-time form for few trajectories (the planner's cadence is ONE, cppflow/optimization.py:128): the two kernels
// above walk the T waypoints one after the other (2 T dependent block steps, ~1.4 us each), which leaves the chip idle when
// S is small.  Parallel cyclic reduction eliminates in ceil(log2 T) levels instead: one lane per waypoint, one workgroup per
// trajectory.  With A block-tridiagonal and symmetric, at stride s row t couples to t - s through L_t (and to t + s through
// L_{t+s}^T); one level replaces
//     alpha = -L_t D_{t-s}^-1 ,  gamma = -L_{t+s}^T D_{t+s}^-1
//     D_t <- D_t + alpha L_t^T + gamma L_{t+s} ,   y_t <- y_t + alpha y_{t-s} + gamma y_{t+s} ,   L_t <- alpha L_{t-s}
// which doubles the stride; after the last level delta_t = D_t^-1 y_t.  O(T log T d^3) work instead of O(T d^3), all of it
// parallel.  State lives in the caller's workspace (D_t packed | y_t in work_blocks, L_t dense in work_G) and is exchanged
// between the lanes of the workgroup through L1 / L2 (workgroup-scope fences of __syncthreads).  Used without the pose block
// (diagonal blocks = (cnt a^2 + lambda) I + collision terms: well conditioned, the Schur complements stay SPD).
// kLds (T <= 256): the state (D_t | y_t | L_t per waypoint, 84 floats at d = 7) lives in LDS for the whole launch instead of the
// caller's workspace -- a level's neighbour reads are LDS reads instead of L2 round trips behind a barrier, which was half of a
// level's time (86 KB of dynamic LDS at d = 7, 110 KB at d = 8: one workgroup per compute unit).
// kSplit (with kLds, BS = 512, T <= 256): TWO wavefront-uniform halves of the workgroup per waypoint -- lanes 0..255 do the
// inverse and the t - s side of a level, lanes 256..511 the t + s side into a zero accumulator, which is added to the state in
// a second write phase (two more barriers per level; the critical path of a level drops from inverse + both sides to inverse +
// the longer side).  Same-process A/B (scripts/pcr_ab.py, profiles/r4_pcr_ab.txt): one trajectory 64.4 -> 59.1 us, 512 trajectories
// 117 -> 108 us at d = 7; 85.6 -> 81.4 and 171 -> 160 us at d = 8.
// Two forms of the general level (CPPF_PCR_LEAN): up to 7 joints the row-by-row form, which holds the inverse and both coupling
// blocks in registers (250 of them); at 8 joints that form needs 340 and spilled 300 - 372 B per lane in the 512-lane
// instantiations, so there the blocks are streamed from the state as they are used (236 - 250 registers, nothing spilled; at 7
// joints the streamed form is 20 % slower than the row form -- 73 vs 59 us -- its loads sit in front of their uses).
// index of element (i, j) in the packed upper triangle (row-major, D(D+1)/2 entries) of a symmetric D x D matrix
template <int D>
__device__ __forceinline__ constexpr int sym_index(int i, int j) {
    const int a = i < j ? i : j, b = i < j ? j : i;
    return a * D - a * (a - 1) / 2 + (b - a);
}

#ifndef CPPF_PCR_FENCE
#define CPPF_PCR_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
#ifndef CPPF_PCR_LEAN
#define CPPF_PCR_LEAN(D, BS, kLds, kSplit) ((D) >= 8)
#endif
// a pointer the compiler knows nothing about: loads through it are not merged with earlier loads of the same address (which would
// keep a whole coupling block in registers between its two uses instead of reading it again where it is needed)
__device__ __forceinline__ const float* opaque(const float* p) {
    asm volatile("" : "+v"(p));
    return p;
}

template <int D, int BS, bool kLds = false, bool kSplit = false>
__global__ __launch_bounds__(BS) void full_solve_pcr_kernel(const ChainK ch, const FullK prm, const float* __restrict__ x,
                                                            const float* __restrict__ xv, float* blocks, float* workL,
                                                            float* __restrict__ x_out) {
    constexpr int NT = D * (D + 1) / 2, SB = NT + D, DD = D * D;
    constexpr int STR = (SB + DD) | 1;  // odd stride: lanes t, t + 1, ... of a wavefront hit distinct banks
    extern __shared__ float pcr_state[];
    static_assert(!kSplit || (kLds && BS == 512), "the split form keeps its state in LDS and runs 512 lanes");
    constexpr int TW = kSplit ? BS / 2 : BS;  // waypoints per workgroup
    constexpr bool kLean = CPPF_PCR_LEAN(D, BS, kLds, kSplit);  // which form of the general level (below)
    const int s = blockIdx.x, t = kSplit ? (int)(threadIdx.x & (TW - 1)) : (int)threadIdx.x, T = prm.W;
    const int h = kSplit ? (int)(threadIdx.x / TW) : 0;  // wavefront-uniform
    const bool act = t < T;
    const size_t base = (size_t)s * T;
    auto st_blk = [&](int u) {
        if constexpr (kLds)
            return pcr_state + u * STR;
        else
            return blocks + (base + u) * SB;
    };
    auto st_L = [&](int u) {
        if constexpr (kLds)
            return pcr_state + u * STR + SB;
        else
            return workL + (base + u) * DD;
    };
    float a2[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        const float a = prm.use_diff ? prm.a_diff * (((ch.pris_mask >> j) & 1u) ? prm.a_diff_pris : 1.f) : 0.f;
        a2[j] = a * a;
    }
    const float beta = prm.a_vq * prm.a_diff, beta2 = beta * beta;

    // ---- assemble row t in place: D_t = M_t + (cnt a^2 + [vq] beta^2 + lambda) I,  y_t = m_t + analytic J^T r terms
    if (act && h == 0) {
        const float* blk = blocks + (base + t) * SB;  // the row-parallel kernel's M_t | m_t
        float* dst = st_blk(t);
        const bool has_next = t + 1 < T, has_prev = t > 0;
        const float cnt = (has_next ? 1.f : 0.f) + (has_prev ? 1.f : 0.f);
        const bool vq = prm.use_vq && (t < prm.n_vq || t >= T - prm.n_vq);
        float xc[D], xo[D], b[D];
        load_x<D>(x, base + t, xc);
#pragma unroll
        for (int j = 0; j < D; ++j) b[j] = blk[NT + j];
        if (has_next) {
            load_x<D>(x, base + t + 1, xo);
#pragma unroll
            for (int j = 0; j < D; ++j) xo[j] = xo[j] - xc[j];
            wrap_pi_all<D>(xo);
#pragma unroll
            for (int j = 0; j < D; ++j) b[j] = CPPF_FMA(a2[j], xo[j], b[j]);
        }
        if (has_prev) {
            load_x<D>(x, base + t - 1, xo);
#pragma unroll
            for (int j = 0; j < D; ++j) xo[j] = xc[j] - xo[j];
            wrap_pi_all<D>(xo);
#pragma unroll
            for (int j = 0; j < D; ++j) b[j] = CPPF_FMA(-a2[j], xo[j], b[j]);
        }
        if (vq && xv) {
            load_x<D>(xv, base + t, xo);
#pragma unroll
            for (int j = 0; j < D; ++j) xo[j] = xc[j] - xo[j];
            wrap_pi_all<D>(xo);
#pragma unroll
            for (int j = 0; j < D; ++j) b[j] = CPPF_FMA(-beta2, xo[j], b[j]);
        }
        {
            int k = 0, kd = 0;
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = i; j < D; ++j) {
                    float v = blk[k];
                    if (j == i) v += cnt * a2[i] + (vq ? beta2 : 0.f) + prm.lm_lambda;  // diagonal of the packed upper triangle
                    dst[k++] = v;
                }
            (void)kd;
        }
#pragma unroll
        for (int j = 0; j < D; ++j) dst[NT + j] = b[j];
        float* Lt = st_L(t);
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) Lt[i * D + j] = (i == j && has_prev) ? -a2[i] : 0.f;  // E = -diag(a^2)
    }
    __syncthreads();

    auto load_sym = [&](const float* src, float (&M)[D][D]) {
        int k = 0;
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = i; j < D; ++j) {
                const float v = src[k++];
                M[i][j] = v;
                M[j][i] = v;
            }
    };

    // D_t^-1 is needed by both neighbours of t: every lane inverts its own block once per level and shares it through LDS
    __shared__ float s_P[TW][NT + 1];  // +1: odd row stride, no bank conflicts on the strided neighbour reads
    auto load_inv = [&](int u, float (&M)[D][D]) {
        int k = 0;
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = i; j < D; ++j) {
                const float v = s_P[u][k++];
                M[i][j] = v;
                M[j][i] = v;
            }
    };
    for (int st = 1; st < T; st <<= 1) {
        float nD[D][D], ny[D], nL[D][D];
        if (act && h == 0) {
            float Dn[D][D], P[D][D];
            load_sym(st_blk(t), Dn);
            spd_inverse<D>(Dn, prm.lm_lambda, P);
            int k = 0;
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = i; j < D; ++j) s_P[t][k++] = P[i][j];
        }
        __syncthreads();
        if (act) {
            const float* own = st_blk(t);
            // (h == 1, the t + s side of the split form, accumulates from zero)
#pragma unroll
            for (int j = 0; j < D; ++j) ny[j] = h == 0 ? own[NT + j] : 0.f;
            const bool do_m = !kSplit || h == 0, do_p = !kSplit || h == 1;
            const int tm = do_m ? t - st : -1, tp = do_p ? t + st : T;
            if (st == 1) {
                if (h == 0) {
                    load_sym(own, nD);
                } else {
#pragma unroll
                    for (int i = 0; i < D; ++i)
#pragma unroll
                        for (int j = 0; j < D; ++j) nD[i][j] = 0.f;
                }
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) nL[i][j] = 0.f;
                // First level: every coupling block is still the diagonal E = -diag(a^2) the assembly wrote (0 above waypoint
                // 0), so each matrix product below collapses to a row / column scaling.  The results are bit-identical to the
                // general branch (its other terms are exact zeros): ~250 instead of ~1 800 multiply-adds for this level.
                float lt[D], lm[D];
                int t1 = t;  // (opaque: these sixteen selects are otherwise hoisted out of the level loop and held in registers across it)
                asm volatile("" : "+v"(t1));
#pragma unroll
                for (int j = 0; j < D; ++j) lt[j] = t1 > 0 ? -a2[j] : 0.f, lm[j] = t1 > 1 && tm >= 0 ? -a2[j] : 0.f;
                if (tm >= 0) {
                    float P[D][D];
                    const float* nb = st_blk(tm);
                    load_inv(tm, P);
#pragma unroll
                    for (int i = 0; i < D; ++i) {
                        float al[D], accy = ny[i];
#pragma unroll
                        for (int j = 0; j < D; ++j) al[j] = -(lt[i] * P[i][j]);
#pragma unroll
                        for (int j = 0; j < D; ++j) {
                            nD[i][j] = CPPF_FMA(al[j], lt[j], nD[i][j]);  // alpha L_t^T
                            nL[i][j] = al[j] * lm[j];                      // alpha L_{t-1}
                        }
#pragma unroll
                        for (int k = 0; k < D; ++k) accy = CPPF_FMA(al[k], nb[NT + k], accy);
                        ny[i] = accy;
                    }
                }
                if (tp < T) {  // L_{t+1} = E (t + 1 > 0)
                    float P[D][D];
                    const float* nb = st_blk(tp);
                    load_inv(tp, P);
#pragma unroll
                    for (int i = 0; i < D; ++i) {
                        float ga[D], accy = ny[i];
#pragma unroll
                        for (int j = 0; j < D; ++j) ga[j] = -(-a2[i] * P[i][j]);  // -(L_{t+1}^T D_{t+1}^-1), row i
#pragma unroll
                        for (int j = 0; j < D; ++j) nD[i][j] = CPPF_FMA(ga[j], -a2[j], nD[i][j]);  // gamma L_{t+1}
#pragma unroll
                        for (int k = 0; k < D; ++k) accy = CPPF_FMA(ga[k], nb[NT + k], accy);
                        ny[i] = accy;
                    }
                }
            } else {
                if constexpr (kLean) {
                // General level, laid out for registers (the row-by-row form held P, L_t, L_{t-s} and all three
                // results at once -- 300+ values at d = 8, 372 B of scratch per lane at 512 lanes): first alpha = -L_t P_{t-s} and
                // gamma = -L_{t+s}^T P_{t+s} as whole matrices (the inverses are dead after that), then D_t and y_t column by
                // column with the coupling blocks streamed from the state (LDS or L2) as they are used, then -- gamma dead, D_t
                // folded to its packed symmetric form -- L_t.  Every sum is formed in the order of the row-by-row form.
                // A side a lane does not have (t - s < 0, t + s >= T) runs on the lane's own row with alpha / gamma set to zero --
                // its terms are exact zeros -- and is skipped when no lane of the wavefront has it (the last levels: half the
                // wavefronts have one side only).  Every array is assigned on every path: a conditionally assigned one is a
                // loop-carried value to the compiler (64 registers held across the whole level for each).
                float A[D][D], G[D][D];
                const bool m_on = tm >= 0, p_on = tp < T;
                const bool m_any = __builtin_amdgcn_ballot_w64(m_on) != 0ull, p_any = __builtin_amdgcn_ballot_w64(p_on) != 0ull;
                const int um = m_on ? tm : t, up = p_on ? tp : t;
                if (m_any) {
                    float P[D][D];
                    load_inv(um, P);
                    const float* Ltp = opaque(st_L(t));
#pragma unroll
                    for (int i = 0; i < D; ++i) {
                        CPPF_PCR_FENCE();
                        float lt[D];
#pragma unroll
                        for (int k = 0; k < D; ++k) lt[k] = Ltp[i * D + k];
#pragma unroll
                        for (int j = 0; j < D; ++j) {
                            float acc = 0.f;
#pragma unroll
                            for (int k = 0; k < D; ++k) acc = CPPF_FMA(lt[k], P[k][j], acc);
                            A[i][j] = m_on ? -acc : 0.f;
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < D; ++i)
#pragma unroll
                        for (int j = 0; j < D; ++j) A[i][j] = 0.f;
                }
                if (p_any) {
                    float P[D][D];
                    load_inv(up, P);
                    const float* Lpp = opaque(st_L(up));
#pragma unroll
                    for (int i = 0; i < D; ++i) {
                        CPPF_PCR_FENCE();
                        float lp[D];
#pragma unroll
                        for (int k = 0; k < D; ++k) lp[k] = Lpp[k * D + i];
#pragma unroll
                        for (int j = 0; j < D; ++j) {
                            float acc = 0.f;
#pragma unroll
                            for (int k = 0; k < D; ++k) acc = CPPF_FMA(lp[k], P[k][j], acc);  // L_{t+s}^T D_{t+s}^-1
                            G[i][j] = p_on ? -acc : 0.f;
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < D; ++i)
#pragma unroll
                        for (int j = 0; j < D; ++j) G[i][j] = 0.f;
                }
                {
                    const float *Ltp = st_L(t), *Lpp = st_L(up);
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        CPPF_PCR_FENCE();
#pragma unroll
                        for (int i = 0; i < D; ++i) nD[i][j] = h == 0 ? own[sym_index<D>(i, j)] : 0.f;
                        if (m_any) {
                            float lt[D];
#pragma unroll
                            for (int k = 0; k < D; ++k) lt[k] = Ltp[j * D + k];
#pragma unroll
                            for (int i = 0; i < D; ++i) {
                                float accD = nD[i][j];
#pragma unroll
                                for (int k = 0; k < D; ++k) accD = CPPF_FMA(A[i][k], lt[k], accD);  // alpha L_t^T
                                nD[i][j] = accD;
                            }
                        }
                        CPPF_PCR_FENCE();
                        if (p_any) {
                            float lp[D];
#pragma unroll
                            for (int k = 0; k < D; ++k) lp[k] = Lpp[k * D + j];
#pragma unroll
                            for (int i = 0; i < D; ++i) {
                                float accD = nD[i][j];
#pragma unroll
                                for (int k = 0; k < D; ++k) accD = CPPF_FMA(G[i][k], lp[k], accD);  // gamma L_{t+s}
                                nD[i][j] = accD;
                            }
                        }
                    }
                    CPPF_PCR_FENCE();
                    if (m_any) {
                        const float* nb = st_blk(um);
                        float ym[D];
#pragma unroll
                        for (int k = 0; k < D; ++k) ym[k] = nb[NT + k];
#pragma unroll
                        for (int i = 0; i < D; ++i) {
                            float accy = ny[i];
#pragma unroll
                            for (int k = 0; k < D; ++k) accy = CPPF_FMA(A[i][k], ym[k], accy);
                            ny[i] = accy;
                        }
                    }
                    if (p_any) {
                        const float* nb = st_blk(up);
                        float yp[D];
#pragma unroll
                        for (int k = 0; k < D; ++k) yp[k] = nb[NT + k];
#pragma unroll
                        for (int i = 0; i < D; ++i) {
                            float accy = ny[i];
#pragma unroll
                            for (int k = 0; k < D; ++k) accy = CPPF_FMA(G[i][k], yp[k], accy);
                            ny[i] = accy;
                        }
                    }
                }
                CPPF_PCR_FENCE();
                // D_t to its packed symmetric form (what the write phase stores): gamma and half of D_t are dead from here
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = i; j < D; ++j) nD[i][j] = 0.5f * (nD[i][j] + nD[j][i]), nD[j][i] = nD[i][j];
                if (m_any) {
                    const float* Lmp = st_L(um);
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        CPPF_PCR_FENCE();
                        float lm[D];
#pragma unroll
                        for (int k = 0; k < D; ++k) lm[k] = Lmp[k * D + j];
#pragma unroll
                        for (int i = 0; i < D; ++i) {
                            float accL = 0.f;
#pragma unroll
                            for (int k = 0; k < D; ++k) accL = CPPF_FMA(A[i][k], lm[k], accL);  // alpha L_{t-s}
                            nL[i][j] = accL;
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < D; ++i)
#pragma unroll
                        for (int j = 0; j < D; ++j) nL[i][j] = 0.f;
                }
                } else {
                    // the row-by-row form (faster where its 280 registers are there: one lane per waypoint at <= 7 joints -- 61 vs 73 us
                    // for one Panda trajectory, profiles/r4_pcr_ab.txt): the own block first, nothing assigned conditionally
                    if (h == 0) {
                        load_sym(own, nD);
                    } else {
#pragma unroll
                        for (int i = 0; i < D; ++i)
#pragma unroll
                            for (int j = 0; j < D; ++j) nD[i][j] = 0.f;
                    }
#pragma unroll
                    for (int i = 0; i < D; ++i)
#pragma unroll
                        for (int j = 0; j < D; ++j) nL[i][j] = 0.f;
            if (tm >= 0) {
                float P[D][D], Lt[D][D], Lm[D][D], ym[D];
                const float* nb = st_blk(tm);
                const float *Ltp = st_L(t), *Lmp = st_L(tm);
                load_inv(tm, P);
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        Lt[i][j] = Ltp[i * D + j];
                        Lm[i][j] = Lmp[i * D + j];
                    }
#pragma unroll
                for (int j = 0; j < D; ++j) ym[j] = nb[NT + j];
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    float al[D];
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        float acc = 0.f;
#pragma unroll
                        for (int k = 0; k < D; ++k) acc = CPPF_FMA(Lt[i][k], P[k][j], acc);
                        al[j] = -acc;
                    }
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        float accD = nD[i][j], accL = 0.f;
#pragma unroll
                        for (int k = 0; k < D; ++k) {
                            accD = CPPF_FMA(al[k], Lt[j][k], accD);  // alpha L_t^T
                            accL = CPPF_FMA(al[k], Lm[k][j], accL);  // alpha L_{t-s}
                        }
                        nD[i][j] = accD;
                        nL[i][j] = accL;
                    }
                    float accy = ny[i];
#pragma unroll
                    for (int k = 0; k < D; ++k) accy = CPPF_FMA(al[k], ym[k], accy);
                    ny[i] = accy;
                }
            }
            if (tp < T) {
                float P[D][D], Lp[D][D], yp[D];
                const float* nb = st_blk(tp);
                const float* Lpp = st_L(tp);
                load_inv(tp, P);
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) Lp[i][j] = Lpp[i * D + j];
#pragma unroll
                for (int j = 0; j < D; ++j) yp[j] = nb[NT + j];
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    float ga[D];
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        float acc = 0.f;
#pragma unroll
                        for (int k = 0; k < D; ++k) acc = CPPF_FMA(Lp[k][i], P[k][j], acc);  // L_{t+s}^T D_{t+s}^-1
                        ga[j] = -acc;
                    }
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        float accD = nD[i][j];
#pragma unroll
                        for (int k = 0; k < D; ++k) accD = CPPF_FMA(ga[k], Lp[k][j], accD);  // gamma L_{t+s}
                        nD[i][j] = accD;
                    }
                    float accy = ny[i];
#pragma unroll
                    for (int k = 0; k < D; ++k) accy = CPPF_FMA(ga[k], yp[k], accy);
                    ny[i] = accy;
                }
            }
                }
            }  // st > 1
        }
        __syncthreads();  // every lane has read its neighbours' old state
        if (act && h == 0) {
            float* own = st_blk(t);
            float* Lo = st_L(t);
            int k = 0;
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = i; j < D; ++j) own[k++] = 0.5f * (nD[i][j] + nD[j][i]);  // symmetric in exact arithmetic
#pragma unroll
            for (int j = 0; j < D; ++j) own[NT + j] = ny[j];
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) Lo[i * D + j] = nL[i][j];
        }
        __syncthreads();
        if constexpr (kSplit) {
            if (act && h == 1) {  // add the t + s side
                float* own = st_blk(t);
                int k = 0;
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = i; j < D; ++j) own[k++] += 0.5f * (nD[i][j] + nD[j][i]);
#pragma unroll
                for (int j = 0; j < D; ++j) own[NT + j] += ny[j];
            }
            __syncthreads();
        }
    }
    if (act && h == 0) {
        float Dn[D][D], P[D][D], xr[D];
        const float* own = st_blk(t);
        load_sym(own, Dn);
        spd_inverse<D>(Dn, prm.lm_lambda, P);
        load_x<D>(x, base + t, xr);
#pragma unroll
        for (int i = 0; i < D; ++i) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < D; ++k) acc = CPPF_FMA(P[i][k], own[NT + k], acc);
            xr[i] += acc;  // optimization.py:113: x + delta_x
        }
        store_x<D>(x_out, base + t, xr);
    }
}

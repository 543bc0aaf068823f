// cppflow_hip.hip -- kernels + C ABI of libcppflow_hip.so (gfx950 only; see include/cppflow_hip.h for the contract).
//
// Layout of the work: one (seed, waypoint) row per lane, 256 rows per workgroup; rows are independent (no reduction
// across rows anywhere in the reference's pose-only step, cppflow/optimization.py:61-92), so there is no inter-workgroup
// communication and the blockIdx -> rows map needs no XCD awareness: every workgroup streams its own contiguous slab of
// x, and the only shared data (the [W,7] target path, <= 14 KB) sits in every XCD's L2.
//
// The fused kernel keeps x in registers across K iterations of
//     FK -> pose error -> geometric Jacobian -> row scaling -> damped solve -> x += delta -> clamp
// and then evaluates the pose-error metrics and (optionally) the capsule collision masks / search cost of the result
// in the same launch.  The damped normal equations are solved in their dual form
//     delta = Js^T (Js Js^T + lambda I6)^-1 es        ( == (Js^T Js + lambda I)^-1 Js^T es exactly, push-through identity)
// which is a 6x6 SPD system for every ndof, conditioned like Js Js^T instead of the rank-deficient Js^T Js the reference
// hands to LU, and cheaper than the primal form for ndof >= 6.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>

#include "lmik_device.h"
#include "robots_gen.h"

using namespace cppf;

namespace {

#ifndef CPPF_BLOCK
#define CPPF_BLOCK 256
#endif
constexpr int kBlock = CPPF_BLOCK;

// minimum resident waves per SIMD the register allocator must leave room for (2nd __launch_bounds__ argument)
#ifndef CPPF_WAVES_LM
#define CPPF_WAVES_LM 2
#endif
#ifndef CPPF_WAVES_COLL
#define CPPF_WAVES_COLL 2
#endif

// ---- per-row chain evaluation ---------------------------------------------------------------------------------------------

template <int D>
__device__ __forceinline__ void load_x(const float* __restrict__ x, size_t row, float (&q)[D]) {
    const float* p = x + row * D;
    if constexpr (D % 4 == 0) {
#pragma unroll
        for (int k = 0; k < D / 4; ++k) {
            const float4 v = reinterpret_cast<const float4*>(p)[k];
            q[4 * k] = v.x, q[4 * k + 1] = v.y, q[4 * k + 2] = v.z, q[4 * k + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < D; ++j) q[j] = p[j];
    }
}

template <int D>
__device__ __forceinline__ void store_x(float* __restrict__ x, size_t row, const float (&q)[D]) {
    float* p = x + row * D;
    if constexpr (D % 4 == 0) {
#pragma unroll
        for (int k = 0; k < D / 4; ++k)
            reinterpret_cast<float4*>(p)[k] = make_float4(q[4 * k], q[4 * k + 1], q[4 * k + 2], q[4 * k + 3]);
    } else {
#pragma unroll
        for (int j = 0; j < D; ++j) p[j] = q[j];
    }
}

// FK to the end-effector frame only
template <class RB>
__device__ __forceinline__ void fk_ee(const RB& rb, const float (&q)[RB::D], float (&R)[9], float (&p)[3]) {
    frame_identity(R, p);
#pragma unroll
    for (int j = 0; j < RB::D; ++j) {
        fk_fixed_joint(rb, j, R, p);
        fk_joint(R, p, rb.pris(j), q[j]);
    }
    fk_fixed_ee(rb, R, p);
}

// FK keeping every joint's world axis and origin (for the Jacobian)
template <class RB>
__device__ __forceinline__ void fk_ee_axes(const RB& rb, const float (&q)[RB::D], float (&R)[9], float (&p)[3],
                                           float (&ax)[RB::D][3], float (&og)[RB::D][3]) {
    frame_identity(R, p);
#pragma unroll
    for (int j = 0; j < RB::D; ++j) {
        fk_fixed_joint(rb, j, R, p);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            ax[j][i] = R[3 * i + 2];
            og[j][i] = p[i];
        }
        fk_joint(R, p, rb.pris(j), q[j]);
    }
    fk_fixed_ee(rb, R, p);
}

// geometric Jacobian, rows 0:3 angular / 3:6 linear
template <class RB>
__device__ __forceinline__ void jacobian_from_axes(const RB& rb, const float (&pe)[3], const float (&ax)[RB::D][3],
                                                   const float (&og)[RB::D][3], float (&J)[6][RB::D]) {
#pragma unroll
    for (int j = 0; j < RB::D; ++j) {
        const float z0 = ax[j][0], z1 = ax[j][1], z2 = ax[j][2];
        if (!rb.pris(j)) {
            const float rx = pe[0] - og[j][0], ry = pe[1] - og[j][1], rz = pe[2] - og[j][2];
            J[0][j] = z0, J[1][j] = z1, J[2][j] = z2;
            J[3][j] = CPPF_FMA(z1, rz, -(z2 * ry));
            J[4][j] = CPPF_FMA(z2, rx, -(z0 * rz));
            J[5][j] = CPPF_FMA(z0, ry, -(z1 * rx));
        } else {
            J[0][j] = J[1][j] = J[2][j] = 0.f;
            J[3][j] = z0, J[4][j] = z1, J[5][j] = z2;
        }
    }
}

// get_6d_pose_errors without the quaternion detour: the five terms quaternion_to_rpy reads from q_target * q_cur^-1 are
// entries of R_err = R_target * R_cur^T  (cppflow/optimization_utils.py:813-819)
__device__ __forceinline__ void pose_error(const float (&Rt)[9], const float (&tt)[3], const float (&R)[9],
                                           const float (&p)[3], float (&e)[6]) {
    const float e20 = dot3(Rt[6], Rt[7], Rt[8], R[0], R[1], R[2]);
    const float e21 = dot3(Rt[6], Rt[7], Rt[8], R[3], R[4], R[5]);
    const float e22 = dot3(Rt[6], Rt[7], Rt[8], R[6], R[7], R[8]);
    const float e10 = dot3(Rt[3], Rt[4], Rt[5], R[0], R[1], R[2]);
    const float e00 = dot3(Rt[0], Rt[1], Rt[2], R[0], R[1], R[2]);
    float sp = -e20;
    sp = sp > 1.f ? 1.f : (sp < -1.f ? -1.f : sp);
    e[0] = atan2f(e21, e22);
    e[1] = asinf(sp);
    e[2] = atan2f(e10, e00);
    e[3] = tt[0] - p[0];
    e[4] = tt[1] - p[1];
    e[5] = tt[2] - p[2];
}

// positional / geodesic rotational error (cppflow/evaluation_utils.py:134-141).  The reference evaluates
// 2*acos(clamp(q_t . q_c, -1+1e-7, 1-1e-7)) folded to [0, pi]; that is the rotation angle theta of R_err, floored at
// 2*acos(1 - 1e-7) = 8.944e-4 rad by the clamp.  theta is taken from atan2(|skew(R_err)|/2, (tr - 1)/2), which keeps
// full relative accuracy for small angles (acos near 1 does not).
__device__ __forceinline__ void pose_metrics(const float (&Rt)[9], const float (&tt)[3], const float (&R)[9],
                                             const float (&p)[3], float& pos_err, float& rot_err) {
    const float dx = tt[0] - p[0], dy = tt[1] - p[1], dz = tt[2] - p[2];
    pos_err = __builtin_sqrtf(CPPF_FMA(dz, dz, CPPF_FMA(dy, dy, dx * dx)));
    float E[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            E[3 * i + j] = dot3(Rt[3 * i], Rt[3 * i + 1], Rt[3 * i + 2], R[3 * j], R[3 * j + 1], R[3 * j + 2]);
    const float a0 = E[7] - E[5], a1 = E[2] - E[6], a2 = E[3] - E[1];
    const float sn = 0.5f * __builtin_sqrtf(CPPF_FMA(a2, a2, CPPF_FMA(a1, a1, a0 * a0)));
    const float cs = 0.5f * (E[0] + E[4] + E[8] - 1.f);
    const float theta = atan2f(sn, cs);
    rot_err = fmaxf(theta, 8.94427191e-4f);
}

// One damped Gauss-Newton update in dual form.  With S = diag(a_rot x3, a_pos x3) the reference scales J and e in place
// (optimization.py:77-80) and solves (Js^T Js + lambda I) delta = Js^T es; here the scaling is folded into the 6x6 system:
//     A = S (J J^T) S + lambda I,   A y = S e,   delta = J^T (S y)
// (21 + 6 + 6 multiplies instead of 6 d + 6), identical in exact arithmetic.  J and e are left UNscaled.
template <int D>
__device__ __forceinline__ void lm_dual_solve(const float (&J)[6][D], const float (&e)[6], float lambda, float a_pos,
                                              float a_rot, float (&delta)[D]) {
    const float srr = a_rot * a_rot, srp = a_rot * a_pos, spp = a_pos * a_pos;
    // Cholesky A = L L^T with reciprocal pivots and pivot floor lambda (every exact pivot of A is >= lambda_min(A) >=
    // lambda, so the floor only acts on rounding noise)
    float L[6][6], inv[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
#pragma unroll
        for (int i = j; i < 6; ++i) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < D; ++k) s = CPPF_FMA(J[i][k], J[j][k], s);
            s *= (i < 3 ? (j < 3 ? srr : srp) : (j < 3 ? srp : spp));
            if (i == j) s += lambda;
#pragma unroll
            for (int k = 0; k < j; ++k) s = CPPF_FMA(-L[i][k], L[j][k], s);
            if (i == j) {
                s = fmaxf(s, lambda);
                inv[j] = __frsqrt_rn(s);
            } else {
                L[i][j] = s * inv[j];
            }
        }
    }
    float y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float s = e[i] * (i < 3 ? a_rot : a_pos);
#pragma unroll
        for (int k = 0; k < i; ++k) s = CPPF_FMA(-L[i][k], y[k], s);
        y[i] = s * inv[i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        float s = y[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) s = CPPF_FMA(-L[k][i], y[k], s);
        y[i] = s * inv[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i] *= (i < 3 ? a_rot : a_pos);
#pragma unroll
    for (int k = 0; k < D; ++k) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) s = CPPF_FMA(J[i][k], y[i], s);
        delta[k] = s;
    }
}

// Fewer than 6 joints: J J^T (6x6) is rank-deficient and the dual form loses its conditioning advantage, while J^T J (d x d)
// is well conditioned -- solve the reference's primal system (optimization.py:85-88) by Cholesky.
template <int D>
__device__ __forceinline__ void lm_primal_solve(const float (&J)[6][D], const float (&e)[6], float lambda, float a_pos,
                                                float a_rot, float (&delta)[D]) {
    const float s2[2] = {a_rot * a_rot, a_pos * a_pos};
    float L[D][D], inv[D], y[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
#pragma unroll
        for (int i = j; i < D; ++i) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 6; ++k) s = CPPF_FMA(s2[k >= 3] * J[k][i], J[k][j], s);
            if (i == j) s += lambda;
#pragma unroll
            for (int k = 0; k < j; ++k) s = CPPF_FMA(-L[i][k], L[j][k], s);
            if (i == j) {
                s = fmaxf(s, lambda);
                inv[j] = __frsqrt_rn(s);
            } else {
                L[i][j] = s * inv[j];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 6; ++k) s = CPPF_FMA(s2[k >= 3] * J[k][i], e[k], s);
#pragma unroll
        for (int k = 0; k < i; ++k) s = CPPF_FMA(-L[i][k], y[k], s);
        y[i] = s * inv[i];
    }
#pragma unroll
    for (int i = D - 1; i >= 0; --i) {
        float s = y[i];
#pragma unroll
        for (int k = i + 1; k < D; ++k) s = CPPF_FMA(-L[k][i], y[k], s);
        y[i] = s * inv[i];
        delta[i] = y[i];
    }
}

template <int D>
__device__ __forceinline__ void lm_solve(const float (&J)[6][D], const float (&e)[6], float lambda, float a_pos, float a_rot,
                                         float (&delta)[D]) {
    if constexpr (D < 6)
        lm_primal_solve<D>(J, e, lambda, a_pos, a_rot, delta);
    else
        lm_dual_solve<D>(J, e, lambda, a_pos, a_rot, delta);
}

template <class RB>
__device__ __forceinline__ void clamp_row(const RB& rb, float (&q)[RB::D]) {
#pragma unroll
    for (int j = 0; j < RB::D; ++j) q[j] = fminf(fmaxf(q[j], rb.lo(j)), rb.hi(j));
}

// ---- collision stage --------------------------------------------------------------------------------------------------------
// Capsule end points are wave-private scratch indexed by a wave-uniform but run-time capsule id, which registers cannot
// do without spilling; they go to LDS as [capsule*6 + k][lane] so that a wave's 64 lanes hit 64 consecutive banks.
struct CollOut {
    float min_self, min_env;
    int self_hit, env_hit;
};

// ---- broad phase (mask-only launches) -----------------------------------------------------------------------------------------
// A capsule's segment lies in the ball of radius h (half its length, a constant of the rigid link) about its mid point m,
// so  dist(seg_a, seg_b) >= |m_a - m_b| - h_a - h_b  and  dist(seg_c, box) >= dist(m_c, box) - h_c.  A pair is skipped when
// EVERY active lane of the wavefront has   |m_a - m_b|^2 > (h_a + h_b + r_a + r_b + 1 cm)^2 (1 + 1e-4)   (tabulated, rounded up;
// evaluated on doubled mid points against 4 x the threshold, which is the same comparison bit for bit).
// The exact functions return the squared distance between two points ON the segments / box (whatever parameters the
// fp32 arithmetic lands on), which is >= the true squared distance up to the ~1e-6 relative rounding of the final
// difference and dot product; with the 1 cm margin the skipped test could only have said "no hit", so the masks are
// unchanged bit for bit.  The branch is wave-uniform (ballot), so nothing diverges; consecutive lanes are consecutive
// waypoints of one seed, which makes far pairs far for the whole wavefront on real paths.
__device__ __forceinline__ bool cull_far(float lower2, float cull2) {
    return __builtin_amdgcn_ballot_w64(!(lower2 > cull2)) == 0ull;
}

__device__ __forceinline__ float mid_dist2(const float (&a)[3], const float (&b)[3]) {
    const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return CPPF_FMA(dz, dz, CPPF_FMA(dy, dy, dx * dx));
}

__device__ __forceinline__ float point_box_dist2(const float (&m)[3], const float* __restrict__ lo,
                                                 const float* __restrict__ hi) {
    const float ex = m[0] - clampf(m[0], lo[0], hi[0]), ey = m[1] - clampf(m[1], lo[1], hi[1]),
                ez = m[2] - clampf(m[2], lo[2], hi[2]);
    return CPPF_FMA(ez, ez, CPPF_FMA(ey, ey, ex * ex));
}

template <class RB>
__device__ __forceinline__ void fk_capsules_to_lds(const RB& rb, const CollK& co, const float (&q)[RB::D],
                                                   float* __restrict__ lds, int tid, float (&R)[9], float (&p)[3]) {
    frame_identity(R, p);
    for (int c = co.cap_begin[0]; c < co.cap_begin[1]; ++c) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lds[(c * 6 + k) * kBlock + tid] = co.cap_p0[c][k];
            lds[(c * 6 + 3 + k) * kBlock + tid] = co.cap_p1[c][k];
        }
    }
#pragma unroll
    for (int j = 0; j < RB::D; ++j) {
        fk_fixed_joint(rb, j, R, p);
        fk_joint(R, p, rb.pris(j), q[j]);
        for (int c = co.cap_begin[j + 1]; c < co.cap_begin[j + 2]; ++c) {
            float w0[3], w1[3];
            xform_point(R, p, co.cap_p0[c][0], co.cap_p0[c][1], co.cap_p0[c][2], w0);
            xform_point(R, p, co.cap_p1[c][0], co.cap_p1[c][1], co.cap_p1[c][2], w1);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                lds[(c * 6 + k) * kBlock + tid] = w0[k];
                lds[(c * 6 + 3 + k) * kBlock + tid] = w1[k];
            }
        }
    }
}

// Robot-specialised variant: capsule ids, link ids and the pair list are compile-time, so the end points live in VGPRs
// (static indices after unrolling) and no LDS is touched.  Same canonical operation order as the LDS variant.  Two phases so
// that a caller can retire everything else it holds (target pose, q, the frame) between them: the pair / cuboid tests then
// run with the capsule end points as the only long-lived registers, which keeps the fused kernel at <= 128 VGPRs, i.e. all
// four wavefronts per SIMD of a 262 144-row launch resident at once (no half-empty second round).
template <class RB>
__device__ __forceinline__ void capsule_fk_static(const RB& rb, const float (&q)[RB::D], float (&R)[9], float (&p)[3],
                                                  float (&w0)[(RB::Table::L > 0 ? RB::Table::L : 1)][3],
                                                  float (&w1)[(RB::Table::L > 0 ? RB::Table::L : 1)][3]) {
    using T = typename RB::Table;
    frame_identity(R, p);
#pragma unroll
    for (int c = 0; c < T::L; ++c) {
        if (T::cap_link[c] < 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                w0[c][k] = T::cap_p0[c][k];
                w1[c][k] = T::cap_p1[c][k];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < RB::D; ++j) {
        fk_fixed_joint(rb, j, R, p);
        fk_joint(R, p, rb.pris(j), q[j]);
#pragma unroll
        for (int c = 0; c < T::L; ++c) {
            if (T::cap_link[c] == j) {
                xform_point(R, p, T::cap_p0[c][0], T::cap_p0[c][1], T::cap_p0[c][2], w0[c]);
                xform_point(R, p, T::cap_p1[c][0], T::cap_p1[c][1], T::cap_p1[c][2], w1[c]);
            }
        }
    }
}

// twice the capsule's mid point: the broad phase works on doubled coordinates (|s_a - s_b|^2 against 4 x the tabulated
// threshold, the cuboid corners doubled by the host) -- exactly the same comparison as on the mid points themselves (scaling
// by powers of two is exact), without the three multiplies per capsule
__device__ __forceinline__ void capsule_mid(const float (&a0)[3], const float (&a1)[3], float (&m)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) m[k] = a0[k] + a1[k];
}

template <class RB, bool WANT_MIN>
__device__ __forceinline__ CollOut collide_tests_static(const CollK& co,
                                                        const float (&w0)[(RB::Table::L > 0 ? RB::Table::L : 1)][3],
                                                        const float (&w1)[(RB::Table::L > 0 ? RB::Table::L : 1)][3],
                                                        bool do_self, bool do_env) {
    using T = typename RB::Table;
    // Broad phase of the mask-only launches (see cull_far): one bounding-sphere test per pair / per (capsule, cuboid) on
    // the capsule mid points (recomputed per test: 6 adds are cheaper than 27 more live registers); the exact distance is
    // evaluated only when some lane of the wavefront is within reach.
    CollOut r;
    r.min_self = INFINITY;
    r.self_hit = 0;
    if (do_self) {
#pragma unroll
        for (int pi = 0; pi < T::P; ++pi) {
            const int a = T::pair_a[pi], b = T::pair_b[pi];
            if constexpr (!WANT_MIN) {
                float ma[3], mb[3];
                capsule_mid(w0[a], w1[a], ma);
                capsule_mid(w0[b], w1[b], mb);
                if (cull_far(mid_dist2(ma, mb), 4.f * T::pair_cull[pi])) continue;
            }
            const float d2 = seg_seg_dist2(w0[a], w1[a], w0[b], w1[b]);
            if constexpr (WANT_MIN) {
                const float v = __builtin_sqrtf(d2) - (T::cap_r[a] + T::cap_r[b]);
                r.min_self = v < r.min_self ? v : r.min_self;
            } else {
                r.self_hit |= d2 < T::pair_thr[pi];
            }
        }
    }
    if constexpr (WANT_MIN) r.self_hit = r.min_self < 0.f;
    r.min_env = INFINITY;
    r.env_hit = 0;
    if (do_env) {
        for (int o = 0; o < co.nobs; ++o) {
            float me = INFINITY;
#pragma unroll
            for (int c = 0; c < T::L; ++c) {
                if constexpr (!WANT_MIN) {
                    float m[3];
                    capsule_mid(w0[c], w1[c], m);
                    if (cull_far(point_box_dist2(m, co.obs_lo2[o], co.obs_hi2[o]), 4.f * T::cap_cull[c])) continue;
                }
                const float d2 = seg_box_dist2(w0[c], w1[c], co.obs_lo[o], co.obs_hi[o]);
                if constexpr (WANT_MIN) {
                    const float v = __builtin_sqrtf(d2) - T::cap_r[c];
                    me = v < me ? v : me;
                } else {
                    r.env_hit |= d2 < T::cap_thr[c];
                }
            }
            if constexpr (WANT_MIN) {
                r.env_hit |= (me < 0.f);
                r.min_env = me < r.min_env ? me : r.min_env;
            }
        }
    }
    return r;
}

__device__ __forceinline__ void lds_capsule(const float* __restrict__ lds, int tid, int c, float (&w0)[3], float (&w1)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        w0[k] = lds[(c * 6 + k) * kBlock + tid];
        w1[k] = lds[(c * 6 + 3 + k) * kBlock + tid];
    }
}

// WANT_MIN = false: masks only.  sqrt(d2) - r < 0  <=>  d2 < thr(r) exactly (thr = smallest fp32 y with sqrt_rn(y) >= r,
// tabulated per pair / capsule), so the correctly rounded square root -- ~18 instructions and a branch each on gfx950 -- is
// skipped without changing a bit of the masks.  WANT_MIN = true additionally tracks the signed minimum distances.
template <bool WANT_MIN>
__device__ __forceinline__ CollOut collide_from_lds(const CollK& co, const float* __restrict__ lds, int tid,
                                                    bool do_self, bool do_env) {
    CollOut r;
    r.min_self = INFINITY;
    r.self_hit = 0;
    if (do_self) {
        for (int pi = 0; pi < co.npairs; ++pi) {
            const int a = co.pair_a[pi], b = co.pair_b[pi];
            float a0[3], a1[3], b0[3], b1[3];
            lds_capsule(lds, tid, a, a0, a1);
            lds_capsule(lds, tid, b, b0, b1);
            if constexpr (!WANT_MIN) {
                float ma[3], mb[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) ma[k] = a0[k] + a1[k], mb[k] = b0[k] + b1[k];  // doubled mid points
                if (cull_far(mid_dist2(ma, mb), co.pair_cull4[pi])) continue;
            }
            const float d2 = seg_seg_dist2(a0, a1, b0, b1);
            if constexpr (WANT_MIN) {
                const float v = __builtin_sqrtf(d2) - (co.cap_r[a] + co.cap_r[b]);
                r.min_self = v < r.min_self ? v : r.min_self;
            } else {
                r.self_hit |= d2 < co.pair_thr[pi];
            }
        }
    }
    if constexpr (WANT_MIN) r.self_hit = r.min_self < 0.f;  // collision_detection.py:66-68
    r.min_env = INFINITY;
    r.env_hit = 0;
    if (do_env) {
        for (int o = 0; o < co.nobs; ++o) {
            float me = INFINITY;
            for (int c = 0; c < co.ncaps; ++c) {
                float w0[3], w1[3];
                lds_capsule(lds, tid, c, w0, w1);
                if constexpr (!WANT_MIN) {
                    float m[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) m[k] = w0[k] + w1[k];  // doubled mid point
                    if (cull_far(point_box_dist2(m, co.obs_lo2[o], co.obs_hi2[o]), co.cap_cull4[c])) continue;
                }
                const float d2 = seg_box_dist2(w0, w1, co.obs_lo[o], co.obs_hi[o]);
                if constexpr (WANT_MIN) {
                    const float v = __builtin_sqrtf(d2) - co.cap_r[c];
                    me = v < me ? v : me;
                } else {
                    r.env_hit |= d2 < co.cap_thr[c];
                }
            }
            if constexpr (WANT_MIN) {
                r.env_hit |= (me < 0.f);  // collision_detection.py:39-43
                r.min_env = me < r.min_env ? me : r.min_env;
            }
        }
    }
    return r;
}

// capsule FK + distances for one row; leaves the LAST LINK frame in R, p (the caller applies F_ee for the metrics)
template <class RB, bool WANT_MIN>
__device__ __forceinline__ CollOut collide_row(const RB& rb, const CollK& co, const float (&q)[RB::D], float* lds, int tid,
                                               float (&R)[9], float (&p)[3], bool do_self, bool do_env) {
    if constexpr (RB::kStatic) {
        constexpr int L = RB::Table::L > 0 ? RB::Table::L : 1;
        float w0[L][3], w1[L][3];
        capsule_fk_static<RB>(rb, q, R, p, w0, w1);
        return collide_tests_static<RB, WANT_MIN>(co, w0, w1, do_self, do_env);
    } else {
        fk_capsules_to_lds<RB>(rb, co, q, lds, tid, R, p);
        return collide_from_lds<WANT_MIN>(co, lds, tid, do_self, do_env);
    }
}

template <int D>
__device__ __forceinline__ int jlim_hit(const CollK& co, const float (&q)[D]) {
    int jl = 0;
    if (co.has_jl) {
#pragma unroll
        for (int j = 0; j < D; ++j) jl |= (q[j] < co.jl_lo[j]) | (q[j] > co.jl_hi[j]);  // search.py:52
    }
    return jl;
}

__device__ __forceinline__ void write_coll_outputs(size_t row, const CollOut& c, int jl, uint8_t* self_mask,
                                                   uint8_t* env_mask, uint8_t* jlim_mask, float* ext_cost,
                                                   float* min_self, float* min_env) {
    if (self_mask) self_mask[row] = (uint8_t)c.self_hit;
    if (env_mask) env_mask[row] = (uint8_t)c.env_hit;
    if (jlim_mask) jlim_mask[row] = (uint8_t)jl;
    if (ext_cost) ext_cost[row] = 100.f * (float)jl + 1000.f * (float)c.env_hit + 1000.f * (float)c.self_hit;
    if (min_self) min_self[row] = c.min_self;
    if (min_env) min_env[row] = c.min_env;
}

// ---- kernels ----------------------------------------------------------------------------------------------------------------

__device__ __forceinline__ void load_target(const float* __restrict__ target, int w, float (&Rt)[9], float (&tt)[3]) {
    const float* t = target + (size_t)w * 7;
    tt[0] = t[0], tt[1] = t[1], tt[2] = t[2];
    quat_to_mat(t[3], t[4], t[5], t[6], Rt);
}

// What one row hands to the in-kernel per-seed summary (block_seed_summary)
struct RowSummary {
    float pos_err = 0.f, rot_err = 0.f, cost = 0.f;
    int self_hit = 0, env_hit = 0, jl = 0;
};

// DPP reduction of a NON-NEGATIVE value over the 64 lanes of a wavefront into lane 63 (other lanes end up with partial
// results).  Lanes without a source in a DPP step read the identity 0 (`old` operand), valid for max and for sums here
// because every reduced quantity is >= 0.  No LDS traffic (ds_bpermute butterflies cost ~50 LDS-pipe ops per wave).
template <int CTRL>
__device__ __forceinline__ float dpp_or_zero(float x) {  // lanes without a source lane read 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}

template <bool IS_MAX>
__device__ __forceinline__ float wave_reduce_to_lane63(float v) {
    auto op = [](float a, float b) { return IS_MAX ? fmaxf(a, b) : a + b; };
    const float s1 = dpp_or_zero<0x111>(v), s2 = dpp_or_zero<0x112>(v), s3 = dpp_or_zero<0x113>(v);  // row_shr:1,2,3
    v = op(op(v, s1), op(s2, s3));         // the 4 lanes ending here (within a row of 16)
    v = op(v, dpp_or_zero<0x114>(v));      // row_shr:4   -> 8 lanes
    v = op(v, dpp_or_zero<0x118>(v));      // row_shr:8   -> lane 15 of each row holds its row
    v = op(v, dpp_or_zero<0x142>(v));      // row_bcast:15 -> lanes 31 / 63 hold rows 0-1 / 2-3
    v = op(v, dpp_or_zero<0x143>(v));      // row_bcast:31 -> lane 63 holds all four rows
    return v;
}

// Per-seed summary inside the fused launch (same 8 numbers, bit for bit, as seed_summary_kernel; every reduction is a max
// or a sum of small integers / multiples of 100, so the order does not matter).  Requires W in {64, 128, 256}: a workgroup
// then covers whole seeds and a seed is 1, 2 or 4 whole wavefronts.  Joint changes need the NEXT waypoint's final q:
// lane + 1 through DPP wave_shl:1, the first lane of the next wavefront through LDS.
template <class RB>
__device__ __forceinline__ void block_seed_summary(const RB& rb, int W, size_t row, bool active, const float (&q)[RB::D],
                                                   const RowSummary& rs, float* __restrict__ out) {
    constexpr int D = RB::D;
    __shared__ float s_q[kBlock / 64][D];
    __shared__ float s_red[kBlock / 64][8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float rad2deg = 57.29577951308232087680f;
    const int wps = W >> 6;  // wavefronts per seed: 1, 2 or 4
    if (wps > 1) {
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < D; ++j) s_q[wave][j] = q[j];
        }
        __syncthreads();
    }
    const bool seed_ends_here = ((wave + 1) & (wps - 1)) == 0;  // this wavefront holds the seed's last waypoints
    const bool has_next = active && !(lane == 63 && seed_ends_here);
    const int nw = wave + 1 < kBlock / 64 ? wave + 1 : wave;
    float mrev = 0.f, mpri = 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        float qn = dpp_or_zero<0x130>(q[j]);  // wave_shl:1 -- lane i reads lane i + 1
        qn = (lane == 63) ? s_q[nw][j] : qn;  // (stale but unused when wps == 1: has_next is false there)
        const float dq = qn - q[j];
        const bool pr = rb.pris(j);
        const float a = pr ? fabsf(100.f * dq) : fabsf(rad2deg * wrap_pi(dq));
        mpri = fmaxf(mpri, pr ? a : 0.f);
        mrev = fmaxf(mrev, pr ? 0.f : a);
    }
    float v[8] = {100.f * rs.pos_err, rad2deg * rs.rot_err, has_next ? mrev : 0.f, has_next ? mpri : 0.f,
                  (float)rs.self_hit, (float)rs.env_hit, (float)rs.jl, rs.cost};
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = active ? v[k] : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = wave_reduce_to_lane63<true>(v[k]);
#pragma unroll
    for (int k = 4; k < 8; ++k) v[k] = wave_reduce_to_lane63<false>(v[k]);
    float* o = out + ((uint32_t)row >> (31 - __builtin_clz((uint32_t)W))) * 8;  // seed = row / W, W a power of two
    if (wps == 1) {
        if (active && lane == 63) {
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = v[k];
        }
        return;
    }
    if (lane == 63) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s_red[wave][k] = v[k];
    }
    __syncthreads();
    if (active && lane == 63 && seed_ends_here) {
        for (int i = 1; i < wps; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], s_red[wave - i][k]);
#pragma unroll
            for (int k = 4; k < 8; ++k) v[k] += s_red[wave - i][k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = v[k];
    }
}

// ---- one row of the fused launch, in three pieces: load, one LM iteration, finish (store, metrics, collision stage) -----------
template <class RB>
__device__ __forceinline__ void lm_row_load(const LmK& prm, const float* __restrict__ x_in, const float* __restrict__ target,
                                            size_t row, float (&q)[RB::D], float (&Rt)[9], float (&tt)[3]) {
    load_x<RB::D>(x_in, row, q);
    load_target(target, (int)(row % (size_t)prm.W), Rt, tt);
}

template <class RB>
__device__ __forceinline__ void lm_row_iterate(const RB& rb, const LmK& prm, const cppf_lm_outputs& out, size_t row, bool last,
                                               const float (&Rt)[9], const float (&tt)[3], float (&q)[RB::D]) {
    constexpr int D = RB::D;
    float R[9], p[3], ax[D][3], og[D][3], J[6][D], e[6], delta[D];
    fk_ee_axes<RB>(rb, q, R, p, ax, og);
    pose_error(Rt, tt, R, p, e);
    jacobian_from_axes<RB>(rb, p, ax, og, J);
    lm_solve<D>(J, e, prm.lm_lambda, prm.a_pos, prm.a_rot, delta);
    if (last) {
        // the reference returns J and e scaled in place (optimization.py:77-80, 90-92)
        if (out.J_out) {
            float* Jo = out.J_out + row * 6 * D;
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) Jo[i * D + j] = J[i][j] * (i < 3 ? prm.a_rot : prm.a_pos);
        }
        if (out.e_out) {
#pragma unroll
            for (int i = 0; i < 6; ++i) out.e_out[row * 6 + i] = e[i] * (i < 3 ? prm.a_rot : prm.a_pos);
        }
    }
#pragma unroll
    for (int j = 0; j < D; ++j) q[j] += delta[j];
    if (prm.clamp) clamp_row<RB>(rb, q);
}

template <class RB, int COLL>
__device__ __forceinline__ void lm_row_finish(const RB& rb, const CollK& co, const cppf_lm_outputs& out, float* lds, int tid,
                                              size_t row, const float (&Rt)[9], const float (&tt)[3], const float (&q)[RB::D],
                                              RowSummary& rs) {
    constexpr int D = RB::D;
    if (out.x_out) store_x<D>(out.x_out, row, q);
    const bool want_metrics = out.pos_err_m || out.rot_err_rad || out.seed_summary;
    if constexpr (COLL != 0) {
        const bool do_self = out.self_mask || out.min_self || out.ext_cost || out.seed_summary;
        const bool do_env = out.env_mask || out.min_env || out.ext_cost || out.seed_summary;
        rs.jl = jlim_hit<D>(co, q);
        // capsule FK first, then the pose metrics off its last-link frame (after which the target pose and the frame are
        // dead), then the pair / cuboid tests
        CollOut c;
        auto metrics = [&](float (&R)[9], float (&p)[3]) {
            if (want_metrics) {
                fk_fixed_ee(rb, R, p);
                pose_metrics(Rt, tt, R, p, rs.pos_err, rs.rot_err);
                if (out.pos_err_m) out.pos_err_m[row] = rs.pos_err;
                if (out.rot_err_rad) out.rot_err_rad[row] = rs.rot_err;
            }
        };
        if constexpr (RB::kStatic) {
            constexpr int L = RB::Table::L > 0 ? RB::Table::L : 1;
            float w0[L][3], w1[L][3];
            {
                float R[9], p[3];
                capsule_fk_static<RB>(rb, q, R, p, w0, w1);
                metrics(R, p);
            }
            c = collide_tests_static<RB, COLL == 2>(co, w0, w1, do_self, do_env);
        } else {
            {
                float R[9], p[3];
                fk_capsules_to_lds<RB>(rb, co, q, lds, tid, R, p);
                metrics(R, p);
            }
            c = collide_from_lds<COLL == 2>(co, lds, tid, do_self, do_env);
        }
        rs.self_hit = c.self_hit, rs.env_hit = c.env_hit;
        rs.cost = 100.f * (float)rs.jl + 1000.f * (float)c.env_hit + 1000.f * (float)c.self_hit;
        write_coll_outputs(row, c, rs.jl, out.self_mask, out.env_mask, out.jlim_mask, out.ext_cost, out.min_self,
                           out.min_env);
    } else {
        if (want_metrics) {
            float R[9], p[3];
            fk_ee<RB>(rb, q, R, p);
            pose_metrics(Rt, tt, R, p, rs.pos_err, rs.rot_err);
            if (out.pos_err_m) out.pos_err_m[row] = rs.pos_err;
            if (out.rot_err_rad) out.rot_err_rad[row] = rs.rot_err;
        }
    }
}

// COLL: 0 = no collision stage, 1 = masks / cost only (no square roots), 2 = masks / cost and the signed minimum distances.
// out.seed_summary (host: only when W is 64, 128 or 256 and COLL != 0) adds the per-seed reduction as an epilogue.
template <class RB, int COLL>
__global__ __launch_bounds__(kBlock, CPPF_WAVES_LM) void lm_fused_kernel(const ChainK ch, const CollK co, const LmK prm,
                                                          const float* __restrict__ x_in,
                                                          const float* __restrict__ target, const cppf_lm_outputs out) {
    extern __shared__ float lds[];
    constexpr int D = RB::D;
    const RB rb{ch, co};
    const int tid = threadIdx.x;
    const size_t row = (size_t)blockIdx.x * kBlock + tid;
    const bool active = row < (size_t)prm.n;
    float q[D];
#pragma unroll
    for (int j = 0; j < D; ++j) q[j] = 0.f;
    RowSummary rs;
    if (active) {
        float Rt[9], tt[3];
        lm_row_load<RB>(prm, x_in, target, row, q, Rt, tt);
        for (int it = 0; it < prm.n_steps; ++it) lm_row_iterate<RB>(rb, prm, out, row, it == prm.n_steps - 1, Rt, tt, q);
        lm_row_finish<RB, COLL>(rb, co, out, lds, tid, row, Rt, tt, q, rs);
    }
    if constexpr (COLL != 0) {
        if (out.seed_summary) block_seed_summary<RB>(rb, prm.W, row, active, q, rs, out.seed_summary);
    }
}

template <class RB, bool WANT_MIN>
__global__ __launch_bounds__(kBlock, CPPF_WAVES_COLL) void collision_kernel(const ChainK ch, const CollK co, int n,
                                                           const float* __restrict__ x, uint8_t* self_mask,
                                                           uint8_t* env_mask, uint8_t* jlim_mask, float* ext_cost,
                                                           float* min_self, float* min_env) {
    extern __shared__ float lds[];
    constexpr int D = RB::D;
    const RB rb{ch, co};
    const int tid = threadIdx.x;
    const size_t row = (size_t)blockIdx.x * kBlock + tid;
    if (row >= (size_t)n) return;
    float q[D], R[9], p[3];
    load_x<D>(x, row, q);
    // wave-uniform: which halves of the work the caller asked for (jlim-only calls skip FK altogether)
    const bool do_self = self_mask || min_self || ext_cost;
    const bool do_env = env_mask || min_env || ext_cost;
    CollOut c;
    c.min_self = c.min_env = INFINITY;
    c.self_hit = c.env_hit = 0;
    if (do_self || do_env) c = collide_row<RB, WANT_MIN>(rb, co, q, lds, tid, R, p, do_self, do_env);
    write_coll_outputs(row, c, jlim_hit<D>(co, q), self_mask, env_mask, jlim_mask, ext_cost, min_self, min_env);
}

// full distance matrices (Robot.self_collision_distances / env_collision_distances); box = the single cuboid, co.nobs unused
template <int D, bool ENV>
__global__ __launch_bounds__(kBlock) void distances_kernel(const ChainK ch, const CollK co, int n,
                                                           const float* __restrict__ x, float blo0, float blo1,
                                                           float blo2, float bhi0, float bhi1, float bhi2,
                                                           float* __restrict__ dists) {
    extern __shared__ float lds[];
    using RB = DynRobot<D>;
    const RB rb{ch, co};
    const int tid = threadIdx.x;
    const size_t row = (size_t)blockIdx.x * kBlock + tid;
    if (row >= (size_t)n) return;
    float q[D], R[9], p[3];
    load_x<D>(x, row, q);
    fk_capsules_to_lds<RB>(rb, co, q, lds, tid, R, p);
    if constexpr (ENV) {
        const float lo[3] = {blo0, blo1, blo2}, hi[3] = {bhi0, bhi1, bhi2};
        for (int c = 0; c < co.ncaps; ++c) {
            float w0[3], w1[3];
            lds_capsule(lds, tid, c, w0, w1);
            dists[row * co.ncaps + c] = seg_box_dist(w0, w1, lo, hi) - co.cap_r[c];
        }
    } else {
        for (int pi = 0; pi < co.npairs; ++pi) {
            const int a = co.pair_a[pi], b = co.pair_b[pi];
            float a0[3], a1[3], b0[3], b1[3];
            lds_capsule(lds, tid, a, a0, a1);
            lds_capsule(lds, tid, b, b0, b1);
            dists[row * co.npairs + pi] = seg_seg_dist(a0, a1, b0, b1) - (co.cap_r[a] + co.cap_r[b]);
        }
    }
}

template <int D>
__global__ __launch_bounds__(kBlock) void fk_kernel(const ChainK ch, const CollK co, int n, const float* __restrict__ x,
                                                    float* __restrict__ poses) {
    using RB = DynRobot<D>;
    const RB rb{ch, co};
    const size_t row = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (row >= (size_t)n) return;
    float q[D], R[9], p[3], qt[4];
    load_x<D>(x, row, q);
    fk_ee<RB>(rb, q, R, p);
    mat_to_quat(R, qt);
    float* o = poses + row * 7;
    o[0] = p[0], o[1] = p[1], o[2] = p[2], o[3] = qt[0], o[4] = qt[1], o[5] = qt[2], o[6] = qt[3];
}

template <int D>
__global__ __launch_bounds__(kBlock) void jacobian_kernel(const ChainK ch, const CollK co, int n,
                                                          const float* __restrict__ x, float* __restrict__ Jout) {
    using RB = DynRobot<D>;
    const RB rb{ch, co};
    const size_t row = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (row >= (size_t)n) return;
    float q[D], R[9], p[3], ax[D][3], og[D][3], J[6][D];
    load_x<D>(x, row, q);
    fk_ee_axes<RB>(rb, q, R, p, ax, og);
    jacobian_from_axes<RB>(rb, p, ax, og, J);
    float* Jo = Jout + row * 6 * D;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) Jo[i * D + j] = J[i][j];
}

template <int D>
__global__ __launch_bounds__(kBlock) void pose_errors_kernel(const ChainK ch, const CollK co, int n, int W,
                                                             const float* __restrict__ x,
                                                             const float* __restrict__ target, float* __restrict__ e_out,
                                                             float* __restrict__ cur_out) {
    using RB = DynRobot<D>;
    const RB rb{ch, co};
    const size_t row = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (row >= (size_t)n) return;
    float q[D], R[9], p[3], Rt[9], tt[3], e[6];
    load_x<D>(x, row, q);
    load_target(target, (int)(row % (size_t)W), Rt, tt);
    fk_ee<RB>(rb, q, R, p);
    pose_error(Rt, tt, R, p, e);
    if (e_out) {
#pragma unroll
        for (int i = 0; i < 6; ++i) e_out[row * 6 + i] = e[i];
    }
    if (cur_out) {
        float qt[4];
        mat_to_quat(R, qt);
        float* o = cur_out + row * 7;
        o[0] = p[0], o[1] = p[1], o[2] = p[2], o[3] = qt[0], o[4] = qt[1], o[5] = qt[2], o[6] = qt[3];
    }
}

template <int D>
__global__ __launch_bounds__(kBlock) void pose_metrics_kernel(const ChainK ch, const CollK co, int n, int W,
                                                              const float* __restrict__ x,
                                                              const float* __restrict__ target,
                                                              float* __restrict__ pos_err, float* __restrict__ rot_err) {
    using RB = DynRobot<D>;
    const RB rb{ch, co};
    const size_t row = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (row >= (size_t)n) return;
    float q[D], R[9], p[3], Rt[9], tt[3], pe, re;
    load_x<D>(x, row, q);
    load_target(target, (int)(row % (size_t)W), Rt, tt);
    fk_ee<RB>(rb, q, R, p);
    pose_metrics(Rt, tt, R, p, pe, re);
    if (pos_err) pos_err[row] = pe;
    if (rot_err) rot_err[row] = re;
}

__global__ __launch_bounds__(kBlock) void clamp_kernel(const ChainK ch, size_t total, float* __restrict__ x) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i % (size_t)ch.ndof);
    x[i] = fminf(fmaxf(x[i], ch.lo[j]), ch.hi[j]);
}

// one wavefront per seed: lanes stride over the seed's W waypoints, then a 64-lane butterfly max
template <int D>
__global__ __launch_bounds__(64) void seed_validity_kernel(const ChainK ch, const CollK co, int S, int W,
                                                           const float* __restrict__ x,
                                                           const float* __restrict__ target, float* __restrict__ out) {
    using RB = DynRobot<D>;
    const RB rb{ch, co};
    const int s = blockIdx.x;
    if (s >= S) return;
    const float rad2deg = 57.29577951308232087680f;
    float mp = 0.f, mr = 0.f, mrev = 0.f, mpri = 0.f;
    for (int w = threadIdx.x; w < W; w += 64) {
        const size_t row = (size_t)s * W + w;
        float q[D], R[9], p[3], Rt[9], tt[3], pe, re;
        load_x<D>(x, row, q);
        load_target(target, w, Rt, tt);
        fk_ee<RB>(rb, q, R, p);
        pose_metrics(Rt, tt, R, p, pe, re);
        mp = fmaxf(mp, 100.f * pe);
        mr = fmaxf(mr, rad2deg * re);
        if (w + 1 < W) {
            float qn[D];
            load_x<D>(x, row + 1, qn);
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const float dq = qn[j] - q[j];
                if (rb.pris(j))
                    mpri = fmaxf(mpri, fabsf(100.f * dq));
                else
                    mrev = fmaxf(mrev, fabsf(rad2deg * wrap_pi(dq)));
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mp = fmaxf(mp, __shfl_xor(mp, off, 64));
        mr = fmaxf(mr, __shfl_xor(mr, off, 64));
        mrev = fmaxf(mrev, __shfl_xor(mrev, off, 64));
        mpri = fmaxf(mpri, __shfl_xor(mpri, off, 64));
    }
    if (threadIdx.x == 0) {
        out[s * 4 + 0] = mp, out[s * 4 + 1] = mr, out[s * 4 + 2] = mrev, out[s * 4 + 3] = mpri;
    }
}

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
        mx[0] = fmaxf(mx[0], pc), mx[1] = fmaxf(mx[1], rd);
        sm[0] += pc, sm[1] += rd;
#pragma unroll
        for (int j = 0; j < D; ++j) sm[4] += (float)((q[j] < ch.lo[j]) + (ch.hi[j] < q[j]));  // evaluation_utils.py:24
        if (self_mask) sm[5] += (float)self_mask[row];
        if (env_mask) sm[6] += (float)env_mask[row];
        if (w + 1 < W) {
            float qn[D];
            load_x<D>(x, row + 1, qn);
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const float dq = qn[j] - q[j];
                if (rb.pris(j)) {
                    const float a = fabsf(dq);
                    mx[3] = fmaxf(mx[3], 100.f * a), sm[3] += a;
                } else {
                    const float a = fabsf(wrap_pi(dq));
                    mx[2] = fmaxf(mx[2], rad2deg * a), sm[2] += a;
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
        mp = fmaxf(mp, 100.f * pos_err[row]);
        mr = fmaxf(mr, rad2deg * rot_err[row]);
        ns += (float)self_mask[row];
        ne += (float)env_mask[row];
        nj += (float)jlim_mask[row];
        sc += ext_cost[row];
        if (w + 1 < W) {
            float q[D], qn[D];
            load_x<D>(x, row, q);
            load_x<D>(x, row + 1, qn);
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const float dq = qn[j] - q[j];
                if ((ch.pris_mask >> j) & 1u)
                    mpri = fmaxf(mpri, fabsf(100.f * dq));
                else
                    mrev = fmaxf(mrev, fabsf(rad2deg * wrap_pi(dq)));
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

struct FullK {
    float lm_lambda, a_pos, a_rot, a_diff, a_diff_pris, a_vq, a_self, a_env;
    int32_t use_pose, use_diff, use_vq, n_vq, use_self, use_env;
    int32_t S, W;
};

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
            lds[(c * 6 + k) * kBlock + tid] = co.cap_p0[c][k];
            lds[(c * 6 + 3 + k) * kBlock + tid] = co.cap_p1[c][k];
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
            float w0[3], w1[3];
            xform_point(R, p, co.cap_p0[c][0], co.cap_p0[c][1], co.cap_p0[c][2], w0);
            xform_point(R, p, co.cap_p1[c][0], co.cap_p1[c][1], co.cap_p1[c][2], w1);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                lds[(c * 6 + k) * kBlock + tid] = w0[k];
                lds[(c * 6 + 3 + k) * kBlock + tid] = w1[k];
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
            float w0[3], w1[3], cs[3], cb[3];
            lds_capsule(lds, tid, e, w0, w1);
            sd = seg_box_closest(w0, w1, lo, hi, cs, cb);
            radius = co.cap_r[e];
            if (sd > 0.f) {
#pragma unroll
                for (int i = 0; i < 3; ++i) nrm[i] = (cs[i] - cb[i]) / sd;
            }
            point_grad<RB>(rb, co.cap_link[e], nrm, cs, ax, og, 1.f, g);
        } else {
            const int a = co.pair_a[e], b = co.pair_b[e];
            float a0[3], a1[3], b0[3], b1[3], c1[3], c2[3];
            lds_capsule(lds, tid, a, a0, a1);
            lds_capsule(lds, tid, b, b0, b1);
            sd = seg_seg_closest(a0, a1, b0, b1, c1, c2);
            radius = co.cap_r[a] + co.cap_r[b];
            if (sd > 0.f) {
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

template <int D>
__global__ __launch_bounds__(kBlock) void full_blocks_kernel(const ChainK ch, const CollK co, const FullK prm,
                                                             const float* __restrict__ x,
                                                             const float* __restrict__ target,
                                                             float* __restrict__ blocks) {
    extern __shared__ float lds[];
    using RB = DynRobot<D>;
    constexpr int NT = D * (D + 1) / 2;
    const RB rb{ch, co};
    const int tid = threadIdx.x;
    const size_t row = (size_t)blockIdx.x * kBlock + tid;
    const size_t n = (size_t)prm.S * prm.W;
    if (row >= n) return;
    float q[D], R[9], p[3], ax[D][3], og[D][3], M[NT], m[D];
    load_x<D>(x, row, q);
#pragma unroll
    for (int k = 0; k < NT; ++k) M[k] = 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) m[j] = 0.f;

    // FK with joint axes / origins, capsule end points to LDS (same canonical chain as everywhere else)
    frame_identity(R, p);
    for (int c = co.cap_begin[0]; c < co.cap_begin[1]; ++c) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lds[(c * 6 + k) * kBlock + tid] = co.cap_p0[c][k];
            lds[(c * 6 + 3 + k) * kBlock + tid] = co.cap_p1[c][k];
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
            float w0[3], w1[3];
            xform_point(R, p, co.cap_p0[c][0], co.cap_p0[c][1], co.cap_p0[c][2], w0);
            xform_point(R, p, co.cap_p1[c][0], co.cap_p1[c][1], co.cap_p1[c][2], w1);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                lds[(c * 6 + k) * kBlock + tid] = w0[k];
                lds[(c * 6 + 3 + k) * kBlock + tid] = w1[k];
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
            const float a = i < 3 ? prm.a_rot : prm.a_pos;
            float g[D];
#pragma unroll
            for (int j = 0; j < D; ++j) g[j] = a * J[i][j];
            rank1<D>(M, m, g, 1.f, a * e[i]);
        }
    }
    if (prm.use_self) {  // :645-680: rows where -alpha * dist > 0
        const float w = prm.a_self * prm.a_self;
        for (int pi = 0; pi < co.npairs; ++pi) {
            const int a = co.pair_a[pi], b = co.pair_b[pi];
            float a0[3], a1[3], b0[3], b1[3], c1[3], c2[3];
            lds_capsule(lds, tid, a, a0, a1);
            lds_capsule(lds, tid, b, b0, b1);
            {  // broad phase (cull_far): only pairs that penetrate contribute rows, and a far pair cannot penetrate
                float ma[3], mb[3];
                capsule_mid(a0, a1, ma);
                capsule_mid(b0, b1, mb);
                if (cull_far(mid_dist2(ma, mb), co.pair_cull4[pi])) continue;
            }
            const float sd = seg_seg_closest(a0, a1, b0, b1, c1, c2);
            const float dist = sd - (co.cap_r[a] + co.cap_r[b]);
            if (dist < 0.f) {
                float nrm[3] = {0.f, 0.f, 0.f}, g[D];
                if (sd > 0.f) {
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
                float w0[3], w1[3], cs[3], cb[3];
                lds_capsule(lds, tid, c, w0, w1);
                {
                    float m[3];
                    capsule_mid(w0, w1, m);
                    if (cull_far(point_box_dist2(m, co.obs_lo2[o], co.obs_hi2[o]), co.cap_cull4[c])) continue;
                }
                const float sd = seg_box_closest(w0, w1, co.obs_lo[o], co.obs_hi[o], cs, cb);
                const float dist = sd - co.cap_r[c];
                if (dist < 0.f) {
                    float nrm[3] = {0.f, 0.f, 0.f}, g[D];
                    if (sd > 0.f) {
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
    float* o = blocks + row * (NT + D);
#pragma unroll
    for (int k = 0; k < NT; ++k) o[k] = M[k];
#pragma unroll
    for (int j = 0; j < D; ++j) o[NT + j] = m[j];
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

template <int D>
__global__ __launch_bounds__(64) void full_solve_kernel(const ChainK ch, const FullK prm, const float* __restrict__ x,
                                                        const float* __restrict__ xv, const float* __restrict__ blocks,
                                                        float* __restrict__ workG, float* __restrict__ worky,
                                                        float* __restrict__ x_out) {
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
        if (has_next) load_x<D>(x, base + t + 1, xn);
        const float cnt = (has_next ? 1.f : 0.f) + (has_prev ? 1.f : 0.f);
        const bool vq = prm.use_vq && (t < prm.n_vq || t >= T - prm.n_vq);
#pragma unroll
        for (int j = 0; j < D; ++j) {
            A[j][j] += cnt * a2[j] + (vq ? beta2 : 0.f) + prm.lm_lambda;
            // J^T r of the differencing rows: +a^2 w_t at (t,j), -a^2 w_{t-1} at (t,j)   (w = wrapped joint change)
            if (has_next) b[j] = CPPF_FMA(a2[j], wrap_pi(xn[j] - xc[j]), b[j]);
            if (has_prev) b[j] = CPPF_FMA(-a2[j], wrap_pi(xc[j] - xp[j]), b[j]);
        }
        if (vq) {  // r = beta * wrap(x - x_virtual), J = -beta I  (optimization_utils.py:430-484)
            float v[D];
            if (xv) load_x<D>(xv, base + t, v);
#pragma unroll
            for (int j = 0; j < D; ++j) b[j] = CPPF_FMA(-beta2, xv ? wrap_pi(xc[j] - v[j]) : 0.f, b[j]);
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
            if (has_next) b = CPPF_FMA(a2j, wrap_pi(xn - xc), b);
            if (has_prev) b = CPPF_FMA(-a2j, wrap_pi(xc - xp), b);
            if (vq && xv) b = CPPF_FMA(-beta2, wrap_pi(xc - vj), b);
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
template <int D, int BS>
__global__ __launch_bounds__(BS) void full_solve_pcr_kernel(const ChainK ch, const FullK prm, const float* __restrict__ x,
                                                            const float* __restrict__ xv, float* blocks, float* workL,
                                                            float* __restrict__ x_out) {
    constexpr int NT = D * (D + 1) / 2, SB = NT + D, DD = D * D;
    const int s = blockIdx.x, t = threadIdx.x, T = prm.W;
    const bool act = t < T;
    const size_t base = (size_t)s * T;
    float a2[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        const float a = prm.use_diff ? prm.a_diff * (((ch.pris_mask >> j) & 1u) ? prm.a_diff_pris : 1.f) : 0.f;
        a2[j] = a * a;
    }
    const float beta = prm.a_vq * prm.a_diff, beta2 = beta * beta;

    // ---- assemble row t in place: D_t = M_t + (cnt a^2 + [vq] beta^2 + lambda) I,  y_t = m_t + analytic J^T r terms
    if (act) {
        float* blk = blocks + (base + t) * SB;
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
            for (int j = 0; j < D; ++j) b[j] = CPPF_FMA(a2[j], wrap_pi(xo[j] - xc[j]), b[j]);
        }
        if (has_prev) {
            load_x<D>(x, base + t - 1, xo);
#pragma unroll
            for (int j = 0; j < D; ++j) b[j] = CPPF_FMA(-a2[j], wrap_pi(xc[j] - xo[j]), b[j]);
        }
        if (vq && xv) {
            load_x<D>(xv, base + t, xo);
#pragma unroll
            for (int j = 0; j < D; ++j) b[j] = CPPF_FMA(-beta2, wrap_pi(xc[j] - xo[j]), b[j]);
        }
        int k = 0;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            blk[k] += cnt * a2[i] + (vq ? beta2 : 0.f) + prm.lm_lambda;  // diagonal entry (i, i) of the packed upper triangle
            k += D - i;
        }
#pragma unroll
        for (int j = 0; j < D; ++j) blk[NT + j] = b[j];
        float* Lt = workL + (base + t) * DD;
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
    __shared__ float s_P[BS][NT + 1];  // +1: odd row stride, no bank conflicts on the strided neighbour reads
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
        if (act) {
            float Dn[D][D], P[D][D];
            load_sym(blocks + (base + t) * SB, Dn);
            spd_inverse<D>(Dn, prm.lm_lambda, P);
            int k = 0;
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = i; j < D; ++j) s_P[t][k++] = P[i][j];
        }
        __syncthreads();
        if (act) {
            const float* own = blocks + (base + t) * SB;
            load_sym(own, nD);
#pragma unroll
            for (int j = 0; j < D; ++j) ny[j] = own[NT + j];
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) nL[i][j] = 0.f;
            const int tm = t - st, tp = t + st;
            if (tm >= 0) {
                float P[D][D], Lt[D][D], Lm[D][D], ym[D];
                const float* nb = blocks + (base + tm) * SB;
                load_inv(tm, P);
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        Lt[i][j] = workL[(base + t) * DD + i * D + j];
                        Lm[i][j] = workL[(base + tm) * DD + i * D + j];
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
                const float* nb = blocks + (base + tp) * SB;
                load_inv(tp, P);
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) Lp[i][j] = workL[(base + tp) * DD + i * D + j];
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
        __syncthreads();  // every lane has read its neighbours' old state
        if (act) {
            float* own = blocks + (base + t) * SB;
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
                for (int j = 0; j < D; ++j) workL[(base + t) * DD + i * D + j] = nL[i][j];
        }
        __syncthreads();
    }
    if (act) {
        float Dn[D][D], P[D][D], xr[D];
        const float* own = blocks + (base + t) * SB;
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

// ---- host side --------------------------------------------------------------------------------------------------------------

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define CPPF_REQUIRE(cond, msg) \
    do {                        \
        if (!(cond)) return fail(CPPF_ERR_INVALID, std::string("cppflow_hip: ") + (msg)); \
    } while (0)

#define CPPF_HIP(call)                                                                                       \
    do {                                                                                                     \
        hipError_t e__ = (call);                                                                             \
        if (e__ != hipSuccess)                                                                               \
            return fail(CPPF_ERR_HIP, std::string("cppflow_hip: " #call " failed: ") + hipGetErrorString(e__)); \
    } while (0)

inline unsigned grid_for(size_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

}  // namespace

struct cppf_robot {
    cppf_robot_desc desc;
    ChainK chain;
    CollK coll;
    int device;
    int static_id;     // index into robots_gen.h when the description equals a generated table, else -1
    size_t lds_bytes;  // generic path only: capsule end points, 6 floats per capsule per lane
};

namespace {

// smallest fp32 y with sqrt_rn(y) >= r (host sqrtf is correctly rounded): sqrtf(d2) - r < 0  <=>  d2 < y for all d2 >= 0
float sqrt_threshold(float r) {
    if (!(r > 0.f)) return 0.f;
    float y = r * r;
    while (std::sqrt(y) >= r) y = std::nextafterf(y, 0.f);
    while (std::sqrt(y) < r) y = std::nextafterf(y, INFINITY);
    return y;
}

// broad-phase thresholds (see cull_far): (reach + 1 cm)^2 (1 + 1e-4), rounded up to fp32
double cap_half_length(const cppf_robot_desc& d, int c) {
    double s = 0.0;
    for (int k = 0; k < 3; ++k) {
        const double v = (double)d.cap_p1[c][k] - (double)d.cap_p0[c][k];
        s += v * v;
    }
    return 0.5 * std::sqrt(s);
}

float cull_threshold(double reach) {
    const double y = (reach + 0.01) * (reach + 0.01) * (1.0 + 1e-4);
    float f = (float)y;
    if ((double)f < y) f = std::nextafterf(f, INFINITY);
    return f;
}

// does a description equal a generated compile-time table exactly?
template <class T>
bool desc_matches(const cppf_robot_desc& d) {
    if (d.ndof != T::D || d.n_capsules != T::L || d.n_pairs != T::P) return false;
    uint32_t pm = 0;
    for (int j = 0; j < T::D; ++j) {
        if (d.jtype[j] == CPPF_JOINT_PRISMATIC) pm |= 1u << j;
        for (int k = 0; k < 12; ++k)
            if (d.F[j][k] != T::F[j][k]) return false;
        if (d.lo[j] != T::lo[j] || d.hi[j] != T::hi[j]) return false;
    }
    if (pm != T::pris_mask) return false;
    for (int k = 0; k < 12; ++k)
        if (d.F_ee[k] != T::Fee[k]) return false;
    for (int c = 0; c < T::L; ++c) {
        if (d.cap_link[c] != T::cap_link[c] || d.cap_r[c] != T::cap_r[c]) return false;
        for (int k = 0; k < 3; ++k)
            if (d.cap_p0[c][k] != T::cap_p0[c][k] || d.cap_p1[c][k] != T::cap_p1[c][k]) return false;
    }
    for (int p = 0; p < T::P; ++p)
        if (d.pairs[p][0] != T::pair_a[p] || d.pairs[p][1] != T::pair_b[p]) return false;
    return true;
}

int find_static_robot(const cppf_robot_desc& d) {
#define CPPF_MATCH(idx, Type) \
    if (desc_matches<Type>(d)) return idx;
    CPPF_FOR_EACH_STATIC_ROBOT(CPPF_MATCH)
#undef CPPF_MATCH
    return -1;
}

// dispatch of the heavy kernels on the robot: a generated table if the description matched one, else the generic
// instantiation for its ndof.  Inside __VA_ARGS__ the accessor type is `RB`.
#define CPPF_STATIC_CASE(idx, Type)       \
    case idx: {                           \
        using RB = StaRobot<Type>;        \
        CPPF_BODY;                        \
    } break;

#define CPPF_DISPATCH_RB(robot)                                                                                       \
    if ((robot)->static_id >= 0 && !g_force_generic) {                                                                \
        switch ((robot)->static_id) { CPPF_FOR_EACH_STATIC_ROBOT(CPPF_STATIC_CASE) default: break; }                  \
    } else {                                                                                                          \
        switch ((robot)->desc.ndof) {                                                                                 \
            case 3: { using RB = DynRobot<3>; CPPF_BODY; } break;                                                     \
            case 4: { using RB = DynRobot<4>; CPPF_BODY; } break;                                                     \
            case 5: { using RB = DynRobot<5>; CPPF_BODY; } break;                                                     \
            case 6: { using RB = DynRobot<6>; CPPF_BODY; } break;                                                     \
            case 7: { using RB = DynRobot<7>; CPPF_BODY; } break;                                                     \
            case 8: { using RB = DynRobot<8>; CPPF_BODY; } break;                                                     \
            case 9: { using RB = DynRobot<9>; CPPF_BODY; } break;                                                     \
            case 10: { using RB = DynRobot<10>; CPPF_BODY; } break;                                                   \
            case 12: { using RB = DynRobot<12>; CPPF_BODY; } break;                                                   \
            default: return fail(CPPF_ERR_UNSUPPORTED, "cppflow_hip: kernels are built for ndof in {3..10, 12}");     \
        }                                                                                                             \
    }

bool g_force_generic = false;  // test hook (cppf_debug_force_generic): run the generic kernels even for shipped robots
int g_pcr_max_rows = 131072;  // coupled step: parallel-in-time elimination up to this many (trajectory, waypoint) rows (measured crossover)

// dispatch on ndof: the light kernels are instantiated for the degrees of freedom of the shipped robots
#define CPPF_DISPATCH_D(d, ...)                                                                               \
    switch (d) {                                                                                              \
        case 3: { constexpr int D = 3; __VA_ARGS__; } break;                                                  \
        case 4: { constexpr int D = 4; __VA_ARGS__; } break;                                                  \
        case 5: { constexpr int D = 5; __VA_ARGS__; } break;                                                  \
        case 6: { constexpr int D = 6; __VA_ARGS__; } break;                                                  \
        case 7: { constexpr int D = 7; __VA_ARGS__; } break;                                                  \
        case 8: { constexpr int D = 8; __VA_ARGS__; } break;                                                  \
        case 9: { constexpr int D = 9; __VA_ARGS__; } break;                                                  \
        case 10: { constexpr int D = 10; __VA_ARGS__; } break;                                                \
        case 12: { constexpr int D = 12; __VA_ARGS__; } break;                                                \
        default: return fail(CPPF_ERR_UNSUPPORTED, "cppflow_hip: kernels are built for ndof in {3..10, 12}");  \
    }

int check_launch(const cppf_robot* rb) {
    CPPF_HIP(hipGetLastError());
    (void)rb;
    return CPPF_OK;
}

// Launches go to the robot's device; the calling thread's current device is put back on return (a caller -- torch included --
// that had another device current must not find it changed behind its back).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) {
            err = hipSetDevice(device);
            switched = err == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

#define CPPF_ENTER(rb)                                                                                              \
    CPPF_REQUIRE((rb) != nullptr, "robot handle is NULL");                                                          \
    DeviceGuard device_guard__((rb)->device);                                                                        \
    if (device_guard__.err != hipSuccess)                                                                            \
        return fail(CPPF_ERR_HIP, std::string("cppflow_hip: selecting the robot's device failed: ") +               \
                                      hipGetErrorString(device_guard__.err))

}  // namespace

extern "C" {

int cppf_abi_version(void) { return CPPF_ABI_VERSION; }

const char* cppf_last_error(void) { return g_err.c_str(); }

int cppf_robot_create(const cppf_robot_desc* desc, int device, cppf_robot** out) {
    CPPF_REQUIRE(desc && out, "desc / out is NULL");
    *out = nullptr;
    const int d = desc->ndof;
    CPPF_REQUIRE(d >= 1 && d <= CPPF_MAX_DOF, "ndof out of range");
    CPPF_REQUIRE(desc->n_capsules >= 0 && desc->n_capsules <= CPPF_MAX_CAPSULES, "n_capsules out of range");
    CPPF_REQUIRE(desc->n_pairs >= 0 && desc->n_pairs <= CPPF_MAX_PAIRS, "n_pairs out of range");
    for (int j = 0; j < d; ++j) {
        CPPF_REQUIRE(desc->jtype[j] == CPPF_JOINT_REVOLUTE || desc->jtype[j] == CPPF_JOINT_PRISMATIC, "bad joint type");
        CPPF_REQUIRE(desc->lo[j] <= desc->hi[j], "joint limits: lo > hi");
        for (int k = 0; k < 12; ++k) CPPF_REQUIRE(std::isfinite(desc->F[j][k]), "non-finite chain constant");
    }
    int prev = -1;
    for (int c = 0; c < desc->n_capsules; ++c) {
        const int l = desc->cap_link[c];
        CPPF_REQUIRE(l >= -1 && l < d, "capsule link index out of range");
        CPPF_REQUIRE(l >= prev, "capsules must be ordered by link (base first)");
        prev = l;
        float len2 = 0.f;
        for (int k = 0; k < 3; ++k) {
            const float dd = desc->cap_p1[c][k] - desc->cap_p0[c][k];
            len2 += dd * dd;
        }
        CPPF_REQUIRE(len2 > 1e-12f, "degenerate capsule (p0 == p1)");
        CPPF_REQUIRE(desc->cap_r[c] >= 0.f, "negative capsule radius");
    }
    for (int p = 0; p < desc->n_pairs; ++p) {
        const int a = desc->pairs[p][0], b = desc->pairs[p][1];
        CPPF_REQUIRE(a >= 0 && a < desc->n_capsules && b >= 0 && b < desc->n_capsules && a != b, "bad capsule pair");
    }
    int ndev = 0;
    CPPF_HIP(hipGetDeviceCount(&ndev));
    CPPF_REQUIRE(device >= 0 && device < ndev, "device index out of range");

    cppf_robot* rb = new (std::nothrow) cppf_robot();
    if (!rb) return fail(CPPF_ERR_HIP, "cppflow_hip: out of host memory");
    rb->desc = *desc;
    rb->device = device;
    std::memset(&rb->chain, 0, sizeof(ChainK));
    std::memset(&rb->coll, 0, sizeof(CollK));
    rb->chain.ndof = d;
    for (int j = 0; j < d; ++j) {
        std::memcpy(rb->chain.F[j], desc->F[j], sizeof(float) * 12);
        rb->chain.lo[j] = desc->lo[j];
        rb->chain.hi[j] = desc->hi[j];
        if (desc->jtype[j] == CPPF_JOINT_PRISMATIC) rb->chain.pris_mask |= (1u << j);
    }
    std::memcpy(rb->chain.Fee, desc->F_ee, sizeof(float) * 12);
    CollK& co = rb->coll;
    co.ncaps = desc->n_capsules;
    co.npairs = desc->n_pairs;
    for (int c = 0; c < co.ncaps; ++c) {
        for (int k = 0; k < 3; ++k) {
            co.cap_p0[c][k] = desc->cap_p0[c][k];
            co.cap_p1[c][k] = desc->cap_p1[c][k];
        }
        co.cap_r[c] = desc->cap_r[c];
        co.cap_link[c] = (int8_t)desc->cap_link[c];
        co.cap_thr[c] = sqrt_threshold(desc->cap_r[c]);
        co.cap_cull4[c] = 4.f * cull_threshold(cap_half_length(*desc, c) + (double)desc->cap_r[c]);
    }
    // cap_begin[l+1] = first capsule whose link >= l
    for (int l = -1; l <= d; ++l) {
        int first = co.ncaps;
        for (int c = co.ncaps - 1; c >= 0; --c)
            if (desc->cap_link[c] >= l) first = c;
        co.cap_begin[l + 1] = first;
    }
    for (int p = 0; p < co.npairs; ++p) {
        co.pair_a[p] = (uint8_t)desc->pairs[p][0];
        co.pair_b[p] = (uint8_t)desc->pairs[p][1];
        co.pair_thr[p] = sqrt_threshold(desc->cap_r[desc->pairs[p][0]] + desc->cap_r[desc->pairs[p][1]]);
        const int a = desc->pairs[p][0], b = desc->pairs[p][1];
        co.pair_cull4[p] = 4.f * cull_threshold(cap_half_length(*desc, a) + cap_half_length(*desc, b) + (double)desc->cap_r[a] +
                                         (double)desc->cap_r[b]);
    }
    rb->lds_bytes = (size_t)co.ncaps * 6 * kBlock * sizeof(float);
    rb->static_id = find_static_robot(*desc);
    *out = rb;
    return CPPF_OK;
}

void cppf_robot_destroy(cppf_robot* robot) { delete robot; }

int cppf_robot_ndof(const cppf_robot* robot) { return robot ? robot->desc.ndof : CPPF_ERR_INVALID; }

int cppf_robot_specialization(const cppf_robot* robot) { return robot ? robot->static_id : CPPF_ERR_INVALID; }

void cppf_debug_force_generic(int on) { g_force_generic = on != 0; }

void cppf_debug_set_pcr_max_rows(int n) { g_pcr_max_rows = n; }

int cppf_set_obstacles(cppf_robot* robot, int n_obs, const float* cuboids, const float* Rt) {
    CPPF_REQUIRE(robot, "robot handle is NULL");
    CPPF_REQUIRE(n_obs >= 0 && n_obs <= CPPF_MAX_OBSTACLES, "n_obs out of range");
    CPPF_REQUIRE(n_obs == 0 || (cuboids && Rt), "cuboids / Rt is NULL");
    for (int o = 0; o < n_obs; ++o) {
        const float* R = Rt + o * 12;
        const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        for (int k = 0; k < 9; ++k)
            CPPF_REQUIRE(std::fabs(R[k] - I[k]) < 1e-8f, "only axis-aligned cuboids are supported (R must be I)");
        for (int k = 0; k < 3; ++k) {
            CPPF_REQUIRE(cuboids[o * 6 + k] <= cuboids[o * 6 + 3 + k], "cuboid min corner > max corner");
            robot->coll.obs_lo[o][k] = R[9 + k] + cuboids[o * 6 + k];
            robot->coll.obs_hi[o][k] = R[9 + k] + cuboids[o * 6 + 3 + k];
            robot->coll.obs_lo2[o][k] = 2.f * robot->coll.obs_lo[o][k];
            robot->coll.obs_hi2[o][k] = 2.f * robot->coll.obs_hi[o][k];
        }
    }
    robot->coll.nobs = n_obs;
    return CPPF_OK;
}

int cppf_set_joint_limit_padding(cppf_robot* robot, const float* lo_padded, const float* hi_padded) {
    CPPF_REQUIRE(robot, "robot handle is NULL");
    if (!lo_padded || !hi_padded) {
        robot->coll.has_jl = 0;
        return CPPF_OK;
    }
    for (int j = 0; j < robot->desc.ndof; ++j) {
        robot->coll.jl_lo[j] = lo_padded[j];
        robot->coll.jl_hi[j] = hi_padded[j];
    }
    robot->coll.has_jl = 1;
    return CPPF_OK;
}

int cppf_forward_kinematics(const cppf_robot* robot, const float* x, int n, float* poses, void* stream) {
    CPPF_ENTER(robot);
    CPPF_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return CPPF_OK;
    CPPF_REQUIRE(x && poses, "x / poses is NULL");
    hipStream_t st = (hipStream_t)stream;
    CPPF_DISPATCH_D(robot->desc.ndof, hipLaunchKernelGGL((fk_kernel<D>), dim3(grid_for(n)), dim3(kBlock), 0, st,
                                                        robot->chain, robot->coll, n, x, poses));
    return check_launch(robot);
}

int cppf_jacobian(const cppf_robot* robot, const float* x, int n, float* J, void* stream) {
    CPPF_ENTER(robot);
    CPPF_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return CPPF_OK;
    CPPF_REQUIRE(x && J, "x / J is NULL");
    hipStream_t st = (hipStream_t)stream;
    CPPF_DISPATCH_D(robot->desc.ndof, hipLaunchKernelGGL((jacobian_kernel<D>), dim3(grid_for(n)), dim3(kBlock), 0, st,
                                                        robot->chain, robot->coll, n, x, J));
    return check_launch(robot);
}

int cppf_pose_errors(const cppf_robot* robot, const float* x, const float* target, int S, int W, float* e,
                     float* current_poses, void* stream) {
    CPPF_ENTER(robot);
    CPPF_REQUIRE(S >= 0 && W >= 0, "S / W < 0");
    const size_t n = (size_t)S * W;
    if (n == 0) return CPPF_OK;
    CPPF_REQUIRE(n <= 0x7fffffffu, "S*W exceeds 2^31-1 rows");
    CPPF_REQUIRE(x && target, "x / target is NULL");
    hipStream_t st = (hipStream_t)stream;
    CPPF_DISPATCH_D(robot->desc.ndof, hipLaunchKernelGGL((pose_errors_kernel<D>), dim3(grid_for(n)), dim3(kBlock), 0, st,
                                                        robot->chain, robot->coll, (int)n, W, x, target, e, current_poses));
    return check_launch(robot);
}

int cppf_clamp_to_joint_limits(const cppf_robot* robot, float* x, int n, void* stream) {
    CPPF_ENTER(robot);
    CPPF_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return CPPF_OK;
    CPPF_REQUIRE(x, "x is NULL");
    const size_t total = (size_t)n * robot->desc.ndof;
    hipLaunchKernelGGL(clamp_kernel, dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream, robot->chain, total, x);
    return check_launch(robot);
}

int cppf_lm_pose_steps(const cppf_robot* robot, const float* x_in, const float* target, int S, int W,
                       const cppf_lm_params* params, const cppf_lm_outputs* out, void* stream) {
    CPPF_ENTER(robot);
    CPPF_REQUIRE(params && out, "params / out is NULL");
    CPPF_REQUIRE(S >= 0 && W >= 0, "S / W < 0");
    CPPF_REQUIRE(params->n_steps >= 1, "n_steps must be >= 1");
    CPPF_REQUIRE(params->clamp == 1 || params->n_steps == 1, "clamp = 0 is only defined for a single step");
    CPPF_REQUIRE(params->lm_lambda > 0.f, "lm_lambda must be > 0");
    const size_t n = (size_t)S * W;
    if (n == 0) return CPPF_OK;
    CPPF_REQUIRE(n <= 0x7fffffffu, "S*W exceeds 2^31-1 rows");
    CPPF_REQUIRE(x_in && target, "x_in / target is NULL");
    LmK prm;
    prm.lm_lambda = params->lm_lambda;
    prm.a_pos = params->alpha_position;
    prm.a_rot = params->alpha_rotation;
    prm.n_steps = params->n_steps;
    prm.clamp = params->clamp;
    prm.n = (int)n;
    prm.W = W;
    const bool coll = out->self_mask || out->env_mask || out->jlim_mask || out->ext_cost || out->min_self ||
                      out->min_env || out->seed_summary;
    hipStream_t st = (hipStream_t)stream;
    // per-seed summary: fused into the launch when a workgroup holds whole seeds, else the separate reduction afterwards
    cppf_lm_outputs outk = *out;
    float* const summary_dst = out->seed_summary;
    const bool summary_after = out->seed_summary && !(W >= 64 && kBlock % W == 0 && (W & (W - 1)) == 0);  // 64, 128, 256
    if (summary_after) {
        CPPF_REQUIRE(out->x_out && out->pos_err_m && out->rot_err_rad && out->self_mask && out->env_mask &&
                         out->jlim_mask && out->ext_cost,
                     "seed_summary with W not in {64, 128, 256} needs x_out, pos_err_m, rot_err_rad, the three masks and ext_cost");
        outk.seed_summary = nullptr;
    }
    out = &outk;
    const size_t lds = (robot->static_id >= 0 && !g_force_generic) ? 0 : robot->lds_bytes;
    if (coll && (out->min_self || out->min_env)) {
#define CPPF_BODY                                                                                                 \
    hipLaunchKernelGGL((lm_fused_kernel<RB, 2>), dim3(grid_for(n)), dim3(kBlock), lds, st, robot->chain, robot->coll, \
                       prm, x_in, target, *out)
        CPPF_DISPATCH_RB(robot)
#undef CPPF_BODY
    } else if (coll) {
#define CPPF_BODY                                                                                                 \
    hipLaunchKernelGGL((lm_fused_kernel<RB, 1>), dim3(grid_for(n)), dim3(kBlock), lds, st, robot->chain, robot->coll, \
                       prm, x_in, target, *out)
        CPPF_DISPATCH_RB(robot)
#undef CPPF_BODY
    } else {
#define CPPF_BODY                                                                                               \
    hipLaunchKernelGGL((lm_fused_kernel<RB, 0>), dim3(grid_for(n)), dim3(kBlock), 0, st, robot->chain, robot->coll, \
                       prm, x_in, target, *out)
        CPPF_DISPATCH_RB(robot)
#undef CPPF_BODY
    }
    if (int rc = check_launch(robot)) return rc;
    if (summary_after)
        return cppf_seed_summary(robot, outk.x_out, S, W, outk.ext_cost, outk.pos_err_m, outk.rot_err_rad, outk.self_mask,
                                 outk.env_mask, outk.jlim_mask, summary_dst, stream);
    return CPPF_OK;
}

int cppf_collision_masks(const cppf_robot* robot, const float* q, int S, int W, uint8_t* self_mask, uint8_t* env_mask,
                         uint8_t* jlim_mask, float* ext_cost, float* min_self, float* min_env, void* stream) {
    CPPF_ENTER(robot);
    CPPF_REQUIRE(S >= 0 && W >= 0, "S / W < 0");
    const size_t n = (size_t)S * W;
    if (n == 0) return CPPF_OK;
    CPPF_REQUIRE(n <= 0x7fffffffu, "S*W exceeds 2^31-1 rows");
    CPPF_REQUIRE(q, "q is NULL");
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (robot->static_id >= 0 && !g_force_generic) ? 0 : robot->lds_bytes;
    if (min_self || min_env) {
#define CPPF_BODY                                                                                                    \
    hipLaunchKernelGGL((collision_kernel<RB, true>), dim3(grid_for(n)), dim3(kBlock), lds, st, robot->chain, robot->coll, \
                       (int)n, q, self_mask, env_mask, jlim_mask, ext_cost, min_self, min_env)
        CPPF_DISPATCH_RB(robot)
#undef CPPF_BODY
    } else {
#define CPPF_BODY                                                                                                     \
    hipLaunchKernelGGL((collision_kernel<RB, false>), dim3(grid_for(n)), dim3(kBlock), lds, st, robot->chain, robot->coll, \
                       (int)n, q, self_mask, env_mask, jlim_mask, ext_cost, min_self, min_env)
        CPPF_DISPATCH_RB(robot)
#undef CPPF_BODY
    }
    return check_launch(robot);
}

int cppf_self_collision_distances(const cppf_robot* robot, const float* x, int n, float* dists, void* stream) {
    CPPF_ENTER(robot);
    CPPF_REQUIRE(n >= 0, "n < 0");
    if (n == 0 || robot->coll.npairs == 0) return CPPF_OK;
    CPPF_REQUIRE(x && dists, "x / dists is NULL");
    hipStream_t st = (hipStream_t)stream;
    CPPF_DISPATCH_D(robot->desc.ndof,
                    hipLaunchKernelGGL((distances_kernel<D, false>), dim3(grid_for(n)), dim3(kBlock), robot->lds_bytes,
                                       st, robot->chain, robot->coll, n, x, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, dists));
    return check_launch(robot);
}

int cppf_env_collision_distances(const cppf_robot* robot, const float* x, int n, const float* cuboid, const float* Rt,
                                 float* dists, void* stream) {
    CPPF_ENTER(robot);
    CPPF_REQUIRE(n >= 0, "n < 0");
    CPPF_REQUIRE(cuboid && Rt, "cuboid / Rt is NULL");
    const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < 9; ++k)
        CPPF_REQUIRE(std::fabs(Rt[k] - I[k]) < 1e-8f, "only axis-aligned cuboids are supported (R must be I)");
    if (n == 0 || robot->coll.ncaps == 0) return CPPF_OK;
    CPPF_REQUIRE(x && dists, "x / dists is NULL");
    float lo[3], hi[3];
    for (int k = 0; k < 3; ++k) {
        lo[k] = Rt[9 + k] + cuboid[k];
        hi[k] = Rt[9 + k] + cuboid[3 + k];
    }
    hipStream_t st = (hipStream_t)stream;
    CPPF_DISPATCH_D(robot->desc.ndof,
                    hipLaunchKernelGGL((distances_kernel<D, true>), dim3(grid_for(n)), dim3(kBlock), robot->lds_bytes,
                                       st, robot->chain, robot->coll, n, x, lo[0], lo[1], lo[2], hi[0], hi[1], hi[2],
                                       dists));
    return check_launch(robot);
}

int cppf_self_collision_distances_jacobian(const cppf_robot* robot, const float* x, int n, float* jac, float* dists,
                                           void* stream) {
    CPPF_ENTER(robot);
    CPPF_REQUIRE(n >= 0, "n < 0");
    if (n == 0 || robot->coll.npairs == 0) return CPPF_OK;
    CPPF_REQUIRE(x && jac, "x / jac is NULL");
    hipStream_t st = (hipStream_t)stream;
    CPPF_DISPATCH_D(robot->desc.ndof,
                    hipLaunchKernelGGL((distance_jacobians_kernel<D, false>), dim3(grid_for(n)), dim3(kBlock),
                                       robot->lds_bytes, st, robot->chain, robot->coll, n, x, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, jac,
                                       dists));
    return check_launch(robot);
}

int cppf_env_collision_distances_jacobian(const cppf_robot* robot, const float* x, int n, const float* cuboid,
                                          const float* Rt, float* jac, float* dists, void* stream) {
    CPPF_ENTER(robot);
    CPPF_REQUIRE(n >= 0, "n < 0");
    CPPF_REQUIRE(cuboid && Rt, "cuboid / Rt is NULL");
    const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < 9; ++k)
        CPPF_REQUIRE(std::fabs(Rt[k] - I[k]) < 1e-8f, "only axis-aligned cuboids are supported (R must be I)");
    if (n == 0 || robot->coll.ncaps == 0) return CPPF_OK;
    CPPF_REQUIRE(x && jac, "x / jac is NULL");
    float lo[3], hi[3];
    for (int k = 0; k < 3; ++k) {
        lo[k] = Rt[9 + k] + cuboid[k];
        hi[k] = Rt[9 + k] + cuboid[3 + k];
    }
    hipStream_t st = (hipStream_t)stream;
    CPPF_DISPATCH_D(robot->desc.ndof,
                    hipLaunchKernelGGL((distance_jacobians_kernel<D, true>), dim3(grid_for(n)), dim3(kBlock), robot->lds_bytes,
                                       st, robot->chain, robot->coll, n, x, lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], jac,
                                       dists));
    return check_launch(robot);
}

int cppf_pose_error_metrics(const cppf_robot* robot, const float* x, const float* target, int S, int W, float* pos_err_m,
                            float* rot_err_rad, void* stream) {
    CPPF_ENTER(robot);
    CPPF_REQUIRE(S >= 0 && W >= 0, "S / W < 0");
    const size_t n = (size_t)S * W;
    if (n == 0) return CPPF_OK;
    CPPF_REQUIRE(n <= 0x7fffffffu, "S*W exceeds 2^31-1 rows");
    CPPF_REQUIRE(x && target, "x / target is NULL");
    hipStream_t st = (hipStream_t)stream;
    CPPF_DISPATCH_D(robot->desc.ndof,
                    hipLaunchKernelGGL((pose_metrics_kernel<D>), dim3(grid_for(n)), dim3(kBlock), 0, st, robot->chain,
                                       robot->coll, (int)n, W, x, target, pos_err_m, rot_err_rad));
    return check_launch(robot);
}

int cppf_seed_validity(const cppf_robot* robot, const float* x, const float* target, int S, int W, float* out,
                       void* stream) {
    CPPF_ENTER(robot);
    CPPF_REQUIRE(S >= 0 && W >= 1, "S < 0 or W < 1");
    if (S == 0) return CPPF_OK;
    CPPF_REQUIRE(x && target && out, "x / target / out is NULL");
    hipStream_t st = (hipStream_t)stream;
    CPPF_DISPATCH_D(robot->desc.ndof, hipLaunchKernelGGL((seed_validity_kernel<D>), dim3(S), dim3(64), 0, st,
                                                        robot->chain, robot->coll, S, W, x, target, out));
    return check_launch(robot);
}

int cppf_seed_summary(const cppf_robot* robot, const float* x, int S, int W, const float* ext_cost, const float* pos_err_m,
                      const float* rot_err_rad, const uint8_t* self_mask, const uint8_t* env_mask,
                      const uint8_t* jlim_mask, float* out, void* stream) {
    CPPF_ENTER(robot);
    CPPF_REQUIRE(S >= 0 && W >= 1, "S < 0 or W < 1");
    if (S == 0) return CPPF_OK;
    CPPF_REQUIRE(x && ext_cost && pos_err_m && rot_err_rad && self_mask && env_mask && jlim_mask && out, "NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    CPPF_DISPATCH_D(robot->desc.ndof,
                    hipLaunchKernelGGL((seed_summary_kernel<D>), dim3(S), dim3(64), 0, st, robot->chain, S, W, x, ext_cost,
                                       pos_err_m, rot_err_rad, self_mask, env_mask, jlim_mask, out));
    return check_launch(robot);
}

int cppf_plan_metrics(const cppf_robot* robot, const float* x, const float* target, int S, int W, const uint8_t* self_mask,
                      const uint8_t* env_mask, const float* q_init, float* out, void* stream) {
    CPPF_ENTER(robot);
    CPPF_REQUIRE(S >= 0 && W >= 1, "S < 0 or W < 1");
    if (S == 0) return CPPF_OK;
    CPPF_REQUIRE((size_t)S * W <= 0x7fffffffu, "S*W exceeds 2^31-1 rows");
    CPPF_REQUIRE(x && target && out, "x / target / out is NULL");
    hipStream_t st = (hipStream_t)stream;
    CPPF_DISPATCH_D(robot->desc.ndof,
                    hipLaunchKernelGGL((plan_metrics_kernel<D>), dim3(S), dim3(64), 0, st, robot->chain, robot->coll, S, W, x,
                                       target, self_mask, env_mask, q_init, out));
    return check_launch(robot);
}

int cppf_lm_full_step(const cppf_robot* robot, const float* x_in, const float* target, const float* virtual_configs, int S,
                      int W, const cppf_full_params* params, float* work_blocks, float* work_G, float* work_y,
                      float* x_out, void* stream) {
    CPPF_ENTER(robot);
    CPPF_REQUIRE(params, "params is NULL");
    CPPF_REQUIRE(S >= 0 && W >= 1, "S < 0 or W < 1");
    CPPF_REQUIRE(params->lm_lambda > 0.f, "lm_lambda must be > 0");
    CPPF_REQUIRE(!params->use_virtual_configs || (params->n_virtual_configs > 0 && 2 * params->n_virtual_configs < W),
                 "2 * n_virtual_configs must be < number of waypoints (optimization_utils.py:449-451)");
    CPPF_REQUIRE(x_out != x_in, "x_out must not alias x_in (the back substitution reads x_in)");
    if (S == 0) return CPPF_OK;
    CPPF_REQUIRE(x_in && target && work_blocks && work_G && work_y && x_out, "NULL pointer");
    const size_t n = (size_t)S * W;
    CPPF_REQUIRE(n <= 0x7fffffffu, "S*W exceeds 2^31-1 rows");
    FullK prm;
    prm.lm_lambda = params->lm_lambda;
    prm.a_pos = params->alpha_position;
    prm.a_rot = params->alpha_rotation;
    prm.a_diff = params->alpha_differencing;
    prm.a_diff_pris = params->alpha_differencing_prismatic_scaling;
    prm.a_vq = params->alpha_virtual_configs;
    prm.a_self = params->alpha_self_collision;
    prm.a_env = params->alpha_env_collision;
    prm.use_pose = params->use_pose;
    prm.use_diff = params->use_differencing;
    prm.use_vq = params->use_virtual_configs;
    prm.n_vq = params->n_virtual_configs;
    prm.use_self = params->use_self_collisions;
    prm.use_env = params->use_env_collisions;
    prm.S = S;
    prm.W = W;
    hipStream_t st = (hipStream_t)stream;
    CPPF_DISPATCH_D(robot->desc.ndof,
                    hipLaunchKernelGGL((full_blocks_kernel<D>), dim3(grid_for(n)), dim3(kBlock), robot->lds_bytes, st,
                                       robot->chain, robot->coll, prm, x_in, target, work_blocks));
    // Trajectories are eliminated one per wavefront (8 x 8 lane tile) up to 8 joints, one per lane beyond.  With the pose
    // block the d x d blocks are J^T J + a small diagonal (rank 6 of 7, cond ~1e7): the per-lane kernel's Cholesky with
    // floored pivots copes with that better than the explicit Gauss-Jordan inverse, so it keeps that case.
    // Up to ~128k rows (the planner's cadence is one trajectory): parallel cyclic reduction, one workgroup per trajectory, one
    // lane per waypoint; beyond that its O(T log T) work and traffic lose against the waypoint-after-waypoint kernels
    if (!prm.use_pose && W <= 512 && n <= (size_t)(g_pcr_max_rows > 0 ? g_pcr_max_rows : 0) && robot->desc.ndof >= 3 &&
        robot->desc.ndof <= 8) {
        switch (robot->desc.ndof) {
#define CPPF_PCR_CASE(DD)                                                                                              \
    case DD:                                                                                                           \
        if (W <= 256)                                                                                                  \
            hipLaunchKernelGGL((full_solve_pcr_kernel<DD, 256>), dim3((unsigned)S), dim3(256), 0, st, robot->chain, prm, \
                               x_in, virtual_configs, work_blocks, work_G, x_out);                                     \
        else                                                                                                           \
            hipLaunchKernelGGL((full_solve_pcr_kernel<DD, 512>), dim3((unsigned)S), dim3(512), 0, st, robot->chain, prm, \
                               x_in, virtual_configs, work_blocks, work_G, x_out);                                     \
        break;
            CPPF_PCR_CASE(3) CPPF_PCR_CASE(4) CPPF_PCR_CASE(5) CPPF_PCR_CASE(6) CPPF_PCR_CASE(7) CPPF_PCR_CASE(8)
#undef CPPF_PCR_CASE
            default: break;
        }
        return check_launch(robot);
    }
    switch (prm.use_pose ? 0 : robot->desc.ndof) {
#define CPPF_WAVE_CASE(DD)                                                                                          \
    case DD:                                                                                                        \
        hipLaunchKernelGGL((full_solve_wave_kernel<DD>), dim3((unsigned)S), dim3(64), 0, st, robot->chain, prm, x_in, \
                           virtual_configs, work_blocks, work_G, work_y, x_out);                                    \
        break;
        CPPF_WAVE_CASE(3) CPPF_WAVE_CASE(4) CPPF_WAVE_CASE(5) CPPF_WAVE_CASE(6) CPPF_WAVE_CASE(7) CPPF_WAVE_CASE(8)
#undef CPPF_WAVE_CASE
        default:
            CPPF_DISPATCH_D(robot->desc.ndof,
                            hipLaunchKernelGGL((full_solve_kernel<D>), dim3((unsigned)((S + 63) / 64)), dim3(64), 0, st,
                                               robot->chain, prm, x_in, virtual_configs, work_blocks, work_G, work_y,
                                               x_out));
    }
    return check_launch(robot);
}

int cppf_mjacs(const cppf_robot* robot, const float* q, int k, int T, float prismatic_scaling, float* mjacs, void* stream) {
    CPPF_ENTER(robot);
    CPPF_REQUIRE(k >= 1 && T >= 1, "k, T must be >= 1");
    if (T == 1) return CPPF_OK;
    CPPF_REQUIRE(q && mjacs, "q / mjacs is NULL");
    const size_t total = (size_t)k * k * (size_t)(T - 1);
    CPPF_REQUIRE(total <= ((size_t)1 << 40), "k*k*(T-1) exceeds 2^40 entries");
    CPPF_REQUIRE((total + 255) / 256 <= 0x7fffffffu, "grid too large");
    hipStream_t st = (hipStream_t)stream;
    CPPF_DISPATCH_D(robot->desc.ndof, hipLaunchKernelGGL((mjacs_kernel<D>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                                                         st, q, k, T, robot->chain.pris_mask, prismatic_scaling, mjacs));
    return check_launch(robot);
}

int cppf_dp_search(const cppf_robot* robot, const float* q, const float* ext_cost, int k, int T, float prismatic_scaling,
                   float* work_qT, float* work_costsT, int32_t* work_memoT, float* best_path, int32_t* best_idx,
                   void* stream) {
    CPPF_ENTER(robot);
    CPPF_REQUIRE(k >= 1 && T >= 1, "k, T must be >= 1");
    CPPF_REQUIRE(q && ext_cost && work_qT && work_costsT && work_memoT && best_path && best_idx, "NULL pointer");
    CPPF_REQUIRE((size_t)k * T * robot->desc.ndof <= 0x7fffffffu, "k*T*d exceeds 2^31-1");
    hipStream_t st = (hipStream_t)stream;
    const int d = robot->desc.ndof;
    const size_t total = (size_t)k * T * d;
    hipLaunchKernelGGL(dp_transpose_kernel, dim3(grid_for(total > (size_t)k ? total : (size_t)k)), dim3(256), 0, st, q,
                       ext_cost, k, T, d, work_qT, work_costsT);
    // memo[:,0] is never read by the back-trace's result but is read as a value: define it (search.py:154 zero-inits memo)
    CPPF_HIP(hipMemsetAsync(work_memoT, 0, sizeof(int32_t) * (size_t)k, st));
    const int bpb = k >= 2048 ? 4 : (k >= 512 ? 2 : 1);
    const unsigned blocks = (unsigned)((k + bpb - 1) / bpb);
    for (int t = 1; t < T; ++t) {
        const float* qp = work_qT + (size_t)(t - 1) * k * d;
        const float* qc = work_qT + (size_t)t * k * d;
#define CPPF_DP_LAUNCH(BPB)                                                                                             \
    CPPF_DISPATCH_D(d, hipLaunchKernelGGL((dp_step_kernel<D, BPB>), dim3(blocks), dim3(256), 0, st, qp, qc,               \
                                         work_costsT + (size_t)(t - 1) * k, ext_cost, k, T, t, robot->chain.pris_mask, \
                                         prismatic_scaling, work_costsT + (size_t)t * k, work_memoT + (size_t)t * k))
        if (bpb == 4) {
            CPPF_DP_LAUNCH(4);
        } else if (bpb == 2) {
            CPPF_DP_LAUNCH(2);
        } else {
            CPPF_DP_LAUNCH(1);
        }
#undef CPPF_DP_LAUNCH
    }
    hipLaunchKernelGGL(dp_backtrace_kernel, dim3(1), dim3(256), 0, st, q, work_costsT, work_memoT, k, T, d, best_idx,
                       best_path);
    return check_launch(robot);
}

}  // extern "C"

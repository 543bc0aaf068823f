// lmik_device.h -- device-side building blocks of the fused LM-IK kernels (gfx950 / CDNA4 only).
//
// Everything a wavefront shares (the canonical chain, joint limits, capsules, pair list, obstacles) arrives in the
// kernel-argument segment (structs passed by value): every lane reads the same constant, so the compiler turns each
// access into a scalar load (s_load_dword*) and the constants live in SGPRs -- no VGPRs, no LDS traffic, no bank
// conflicts.  Per-row state (x, frames, Jacobian, the 6x6 dual system) lives in VGPRs, one row per lane.
//
// Canonical operation order.  FK, capsule end points and the two distance functions are written with explicit fmaf in
// a fixed order so that they can be compared bit for bit with the fp32 build of the CPU oracle (tests only).  This file
// is compiled with -ffp-contract=off; the only fused multiply-adds are the ones spelled out below.
#pragma once

#include <hip/hip_runtime.h>
#ifndef __HIPCC_RTC__
#include <stdint.h>
#endif

#include "../../include/cppflow_hip.h"

#ifndef INFINITY
#define INFINITY __builtin_huge_valf()
#endif

#define CPPF_FMA(a, b, c) __builtin_fmaf((a), (b), (c))

namespace cppf {

// ---- kernel-argument structs (wave-uniform constants) -----------------------------------------------------------------
struct ChainK {
    float F[CPPF_MAX_DOF][12];
    float Fee[12];
    float lo[CPPF_MAX_DOF];
    float hi[CPPF_MAX_DOF];
    uint32_t pris_mask;  // bit j set: joint j is prismatic
    int32_t ndof;
};

struct CollK {
    float cap_p0[CPPF_MAX_CAPSULES][3];
    float cap_p1[CPPF_MAX_CAPSULES][3];
    float cap_r[CPPF_MAX_CAPSULES];
    int32_t cap_begin[CPPF_MAX_DOF + 2];  // capsules of link l (-1..d-1) are [cap_begin[l+1], cap_begin[l+2])
    int8_t cap_link[CPPF_MAX_CAPSULES];   // moving link of each capsule (-1 = base)
    float pair_thr[CPPF_MAX_PAIRS];       // smallest y with sqrt_rn(y) >= r_a + r_b:  sqrt(d2) - (r_a+r_b) < 0  <=>  d2 < y
    float cap_thr[CPPF_MAX_CAPSULES];     // the same for r alone (capsule vs cuboid)
    float pair_cull4[CPPF_MAX_PAIRS];     // broad phase: 4 (h_a + h_b + r_a + r_b + 1 cm)^2 (1 + 1e-4), h = half length (x4: doubled mid points)
    float cap_cull4[CPPF_MAX_CAPSULES];   // broad phase: 4 (h + r + 1 cm)^2 (1 + 1e-4)
    uint8_t pair_a[CPPF_MAX_PAIRS];
    uint8_t pair_b[CPPF_MAX_PAIRS];
    float obs_lo[CPPF_MAX_OBSTACLES][3];  // world-frame box corners
    float obs_hi[CPPF_MAX_OBSTACLES][3];
    float obs_lo2[CPPF_MAX_OBSTACLES][3];  // 2 x the corners (broad phase on doubled mid points)
    float obs_hi2[CPPF_MAX_OBSTACLES][3];
    float jl_lo[CPPF_MAX_DOF];  // padded limits of search.py:46-51
    float jl_hi[CPPF_MAX_DOF];
    int32_t ncaps, npairs, nobs, has_jl;
};

struct LmK {
    float lm_lambda, a_pos, a_rot;
    int32_t n_steps, clamp;
    int32_t n, W;
    float tol_pos2, tol_rot2;  // early-out (cppf_lm_params.tol_*), squared; 0 = off
};

// ---- multiply / fma by a chain constant --------------------------------------------------------------------------------------
// In the robot-specialised instantiations every chain constant is a literal after unrolling; 0 and +-1 are peeled off
// here.  Each shortcut returns exactly what the general fmaf would (x*1 and acc + x*1 round once either way; x*0 and
// acc + x*0 are exact for finite x), so the specialised and the generic kernels agree bit for bit (up to the sign of a zero).
__device__ __forceinline__ float cmul(float x, float c) {
    if (__builtin_constant_p(c)) {
        if (c == 0.f) return 0.f;
        if (c == 1.f) return x;
        if (c == -1.f) return -x;
    }
    return x * c;
}

__device__ __forceinline__ float cfma(float x, float c, float acc) {
    if (__builtin_constant_p(c)) {
        if (c == 0.f) return acc;
        if (__builtin_constant_p(acc) && acc == 0.f) return cmul(x, c);
        if (c == 1.f) return acc + x;
        if (c == -1.f) return acc - x;
    }
    return CPPF_FMA(x, c, acc);
}

// ---- robot accessors ------------------------------------------------------------------------------------------------------------
// DynRobot<D>: constants come from the kernel-argument structs (scalar loads).  StaRobot<T>: constants are the
// generated compile-time tables of robots_gen.h; obstacles and the joint-limit padding stay run-time (CollK) in both.
template <int D_>
struct DynRobot {
    static constexpr int D = D_;
    static constexpr bool kStatic = false;
    const ChainK& ch;
    const CollK& co;
    __device__ __forceinline__ float F(int j, int k) const { return ch.F[j][k]; }
    __device__ __forceinline__ float Fee(int k) const { return ch.Fee[k]; }
    __device__ __forceinline__ bool pris(int j) const { return (ch.pris_mask >> j) & 1u; }
    __device__ __forceinline__ float lo(int j) const { return ch.lo[j]; }
    __device__ __forceinline__ float hi(int j) const { return ch.hi[j]; }
};

template <class T>
struct StaRobot {
    using Table = T;
    static constexpr int D = T::D;
    static constexpr bool kStatic = true;
    const ChainK& ch;
    const CollK& co;
    __device__ __forceinline__ float F(int j, int k) const { return T::F[j][k]; }
    __device__ __forceinline__ float Fee(int k) const { return T::Fee[k]; }
    __device__ __forceinline__ bool pris(int j) const { return (T::pris_mask >> j) & 1u; }
    __device__ __forceinline__ float lo(int j) const { return T::lo[j]; }
    __device__ __forceinline__ float hi(int j) const { return T::hi[j]; }
};

// ---- sin / cos ------------------------------------------------------------------------------------------------------------
// Cody-Waite reduction by pi/2 in three exact pieces + Cephes single-precision minimax polynomials on [-pi/4, pi/4].
// ~24 VALU instructions, no slow path: joint angles are bounded by the joint limits (|q| < 2^10 is ample; the magic-number
// rounding needs |x * 2/pi| < 2^22).
// The fp32 oracle build uses the identical formula, so FK agrees bit for bit.
__device__ __forceinline__ void sincos_cw(float x, float& s, float& c) {
    // k = round-to-nearest-even(x * 2/pi) by the magic-number trick: adding 1.5 * 2^23 leaves the integer in the low mantissa
    // bits of t (one rounding, of the exact fma), so t's bit pattern also carries k mod 4 -- no v_rndne / v_cvt_i32
    const float magic = 12582912.0f;
    const float t = CPPF_FMA(x, 0.63661977236758134f, magic);
    const float k = t - magic;
    float r = CPPF_FMA(-k, 1.5703125f, x);
    r = CPPF_FMA(-k, 4.837512969970703125e-4f, r);
    r = CPPF_FMA(-k, 7.54978995489188e-8f, r);
    const float z = r * r;
    const float ps = CPPF_FMA(CPPF_FMA(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    const float sn = CPPF_FMA(r * z, ps, r);
    const float pc = CPPF_FMA(CPPF_FMA(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    const float cs = CPPF_FMA(z * z, pc, CPPF_FMA(-0.5f, z, 1.0f));
    const uint32_t ki = __float_as_uint(t);
    const bool odd = ki & 1u;  // odd quadrants swap the two polynomials
    const uint32_t so = __float_as_uint(odd ? cs : sn) ^ ((ki << 30) & 0x80000000u);         // sin negated in quadrants 2, 3
    const uint32_t co = __float_as_uint(odd ? sn : cs) ^ (((ki + 1u) << 30) & 0x80000000u);  // cos in quadrants 1, 2
    s = __uint_as_float(so);
    c = __uint_as_float(co);
}

// ---- canonical FK steps ---------------------------------------------------------------------------------------------------
// frame <- frame * F      (F = 12 wave-uniform floats: R row-major, t)
__device__ __forceinline__ void fk_fixed(float (&R)[9], float (&p)[3], const float (&F)[12]) {
    float A[9], np[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float r0 = R[3 * i], r1 = R[3 * i + 1], r2 = R[3 * i + 2];
        np[i] = cfma(r2, F[11], cfma(r1, F[10], cfma(r0, F[9], p[i])));
#pragma unroll
        for (int c = 0; c < 3; ++c) A[3 * i + c] = cfma(r2, F[6 + c], cfma(r1, F[3 + c], cmul(r0, F[c])));
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = A[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = np[k];
}

template <class RB>
__device__ __forceinline__ void fk_fixed_joint(const RB& rb, int j, float (&R)[9], float (&p)[3]) {
    float F[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) F[k] = rb.F(j, k);
    fk_fixed(R, p, F);
}

template <class RB>
__device__ __forceinline__ void fk_fixed_ee(const RB& rb, float (&R)[9], float (&p)[3]) {
    float F[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) F[k] = rb.Fee(k);
    fk_fixed(R, p, F);
}

// frame <- frame * M_z(q): rotation about (revolute) or translation along (prismatic) the local z axis
__device__ __forceinline__ void fk_joint(float (&R)[9], float (&p)[3], bool prismatic, float q) {
    if (!prismatic) {
        float s, c;
        sincos_cw(q, s, c);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            // a0, a1 are literals for the first joint of a specialised chain (R = F_0): cmul / cfma fold the 0 / +-1 cases
            const float a0 = R[3 * i], a1 = R[3 * i + 1];
            R[3 * i] = cfma(s, a1, cmul(c, a0));
            R[3 * i + 1] = cfma(c, a1, -cmul(s, a0));
        }
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) p[i] = CPPF_FMA(R[3 * i + 2], q, p[i]);
    }
}

__device__ __forceinline__ void frame_identity(float (&R)[9], float (&p)[3]) {
    R[0] = 1.f, R[1] = 0.f, R[2] = 0.f, R[3] = 0.f, R[4] = 1.f, R[5] = 0.f, R[6] = 0.f, R[7] = 0.f, R[8] = 1.f;
    p[0] = p[1] = p[2] = 0.f;
}

__device__ __forceinline__ float dot3(float a0, float a1, float a2, float b0, float b1, float b2) {
    return CPPF_FMA(a2, b2, CPPF_FMA(a1, b1, a0 * b0));
}

// torch.max propagates a NaN, fmaxf drops it: the per-seed maxima map a NaN to +inf first, so that a seed holding a NaN row
// compares "not below threshold" exactly as the reference's `error.max() < thr` does (cppflow/evaluation_utils.py:41-42)
__device__ __forceinline__ float nan_to_inf(float v) { return v != v ? INFINITY : v; }

// clamps as one v_med3_f32: for lo <= hi and a non-NaN x the median of (x, lo, hi) IS the clamp, value for value
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
__device__ __forceinline__ float clamp01(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, 1.f); }

// world point of a link-frame constant point (canonical order shared with the oracle)
__device__ __forceinline__ void xform_point(const float (&R)[9], const float (&p)[3], float c0, float c1, float c2,
                                            float (&w)[3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) w[i] = cfma(R[3 * i + 2], c2, cfma(R[3 * i + 1], c1, cfma(R[3 * i], c0, p[i])));
}

// ---- rotation matrix -> quaternion (w first; branch on the largest of w,x,y,z so the divisor is >= 1) -----------------------
__device__ __forceinline__ void mat_to_quat(const float (&R)[9], float (&q)[4]) {
    const float m00 = R[0], m01 = R[1], m02 = R[2], m10 = R[3], m11 = R[4], m12 = R[5], m20 = R[6], m21 = R[7], m22 = R[8];
    const float q0 = 1.f + m00 + m11 + m22, q1 = 1.f + m00 - m11 - m22, q2 = 1.f - m00 + m11 - m22,
                q3 = 1.f - m00 - m11 + m22;
    int best = 0;
    float bv = q0;
    if (q1 > bv) best = 1, bv = q1;
    if (q2 > bv) best = 2, bv = q2;
    if (q3 > bv) best = 3, bv = q3;
    const float d = __builtin_sqrtf(bv > 0.f ? bv : 0.f);
    const float inv = 0.5f / d;
    if (best == 0) {
        q[0] = 0.5f * d, q[1] = (m21 - m12) * inv, q[2] = (m02 - m20) * inv, q[3] = (m10 - m01) * inv;
    } else if (best == 1) {
        q[0] = (m21 - m12) * inv, q[1] = 0.5f * d, q[2] = (m10 + m01) * inv, q[3] = (m02 + m20) * inv;
    } else if (best == 2) {
        q[0] = (m02 - m20) * inv, q[1] = (m10 + m01) * inv, q[2] = 0.5f * d, q[3] = (m12 + m21) * inv;
    } else {
        q[0] = (m10 - m01) * inv, q[1] = (m20 + m02) * inv, q[2] = (m21 + m12) * inv, q[3] = 0.5f * d;
    }
}

// target quaternion (w,x,y,z) -> the matrix whose entries are the terms quaternion_to_rpy reads (unit quaternion assumed)
__device__ __forceinline__ void quat_to_mat(float w, float x, float y, float z, float (&R)[9]) {
    R[0] = 1.f - 2.f * (y * y + z * z);
    R[1] = 2.f * (x * y - w * z);
    R[2] = 2.f * (x * z + w * y);
    R[3] = 2.f * (x * y + w * z);
    R[4] = 1.f - 2.f * (x * x + z * z);
    R[5] = 2.f * (y * z - w * x);
    R[6] = 2.f * (x * z - w * y);
    R[7] = 2.f * (y * z + w * x);
    R[8] = 1.f - 2.f * (x * x + y * y);
}

// ---- distances (canonical order) ------------------------------------------------------------------------------------------
// closest distance between two non-degenerate segments (Ericson, Real-Time Collision Detection 5.1.9); also returns the
// closest points c1 (on P1Q1) and c2 (on P2Q2)
__device__ __forceinline__ float seg_seg_closest2(const float (&P1)[3], const float (&Q1)[3], const float (&P2)[3],
                                                  const float (&Q2)[3], float (&c1)[3], float (&c2)[3]) {
    float d1[3], d2[3], rr[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        d1[i] = Q1[i] - P1[i];
        d2[i] = Q2[i] - P2[i];
        rr[i] = P1[i] - P2[i];
    }
    const float a = dot3(d1[0], d1[1], d1[2], d1[0], d1[1], d1[2]);
    const float e = dot3(d2[0], d2[1], d2[2], d2[0], d2[1], d2[2]);
    const float f = dot3(d2[0], d2[1], d2[2], rr[0], rr[1], rr[2]);
    const float c = dot3(d1[0], d1[1], d1[2], rr[0], rr[1], rr[2]);
    const float b = dot3(d1[0], d1[1], d1[2], d2[0], d2[1], d2[2]);
    const float denom = CPPF_FMA(a, e, -(b * b));
    const float inv_a = 1.f / a, inv_e = 1.f / e;  // one IEEE division per capsule: shared by every pair it is in
    float s = denom > 0.f ? clamp01(CPPF_FMA(b, f, -(c * e)) / denom) : 0.f;
    float t = CPPF_FMA(b, s, f) * inv_e;
    // branch-free form of { t < 0: t = 0, s = clamp(-c/a) ; t > 1: t = 1, s = clamp((b-c)/a) } -- lanes of a wave
    // disagree on these cases all the time, so both candidates are always computed and selected
    const float s_lo = clamp01(-c * inv_a), s_hi = clamp01((b - c) * inv_a);
    s = t < 0.f ? s_lo : (t > 1.f ? s_hi : s);
    t = clamp01(t);
    float df[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        c1[i] = CPPF_FMA(d1[i], s, P1[i]);
        c2[i] = CPPF_FMA(d2[i], t, P2[i]);
        df[i] = c1[i] - c2[i];
    }
    return dot3(df[0], df[1], df[2], df[0], df[1], df[2]);
}

// squared distance (what the mask-only kernels compare against the sqrt thresholds) and the distance itself
__device__ __forceinline__ float seg_seg_dist2(const float (&P1)[3], const float (&Q1)[3], const float (&P2)[3],
                                               const float (&Q2)[3]) {
    float c1[3], c2[3];
    return seg_seg_closest2(P1, Q1, P2, Q2, c1, c2);
}

__device__ __forceinline__ float seg_seg_closest(const float (&P1)[3], const float (&Q1)[3], const float (&P2)[3],
                                                 const float (&Q2)[3], float (&c1)[3], float (&c2)[3]) {
    return __builtin_sqrtf(seg_seg_closest2(P1, Q1, P2, Q2, c1, c2));
}

__device__ __forceinline__ float seg_seg_dist(const float (&P1)[3], const float (&Q1)[3], const float (&P2)[3],
                                              const float (&Q2)[3]) {
    return __builtin_sqrtf(seg_seg_dist2(P1, Q1, P2, Q2));
}

// exact distance from segment P0P1 to the axis-aligned box [lo, hi] (0 when they intersect): root of the nondecreasing,
// piecewise-linear half-derivative g of dist^2, bracketed among t = 0, 1 and the six (clamped) face-crossing parameters.
__device__ __forceinline__ float seg_box_closest2(const float (&P0)[3], const float (&P1)[3], const float* __restrict__ lo,
                                                  const float* __restrict__ hi, float (&cs)[3], float (&cb)[3]) {
    // Along the segment x_i(t) = P0_i + t D_i the excess over the slab [lo_i, hi_i] is D_i (t - clamp(t, a_i, b_i)) with
    // [a_i, b_i] the parameter interval in which coordinate i is inside the slab, so the half-derivative of dist^2 is
    //     g(t) = sum_i w_i (t - clamp(t, a_i, b_i)),   w_i = D_i^2
    // nondecreasing and piecewise linear with break points a_i, b_i.  (D_i = 0: w_i = 0, the term vanishes.)
    float D[3], w[3], ta[3], tb[3], cand[8], gv[8];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        D[i] = P1[i] - P0[i];
        const float inv = D[i] != 0.f ? 1.f / D[i] : 0.f;
        const float t0 = (lo[i] - P0[i]) * inv, t1 = (hi[i] - P0[i]) * inv;
        ta[i] = fminf(t0, t1);
        tb[i] = fmaxf(t0, t1);
        w[i] = D[i] * D[i];
        cand[2 + 2 * i] = clamp01(ta[i]);
        cand[3 + 2 * i] = clamp01(tb[i]);
    }
    cand[0] = 0.f;
    cand[1] = 1.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float t = cand[k];
        gv[k] = CPPF_FMA(w[2], t - clampf(t, ta[2], tb[2]),
                         CPPF_FMA(w[1], t - clampf(t, ta[1], tb[1]), w[0] * (t - clampf(t, ta[0], tb[0]))));
    }
    // g is monotone, so the bracket of its root is two independent max / min reductions over the candidates:
    //   tl = max{c : g(c) <= 0}, gl = max{g(c) : g(c) <= 0};  tr = min{c : g(c) > 0}, gr = min{g(c) : g(c) > 0}
    float tl = 0.f, gl = gv[0], tr = 1.f, gr = gv[1];
#pragma unroll
    for (int k = 2; k < 8; ++k) {
        const bool neg = gv[k] <= 0.f;
        tl = fmaxf(tl, neg ? cand[k] : 0.f);
        gl = fmaxf(gl, neg ? gv[k] : gv[0]);
        tr = fminf(tr, neg ? 1.f : cand[k]);
        gr = fminf(gr, neg ? gv[1] : gv[k]);
    }
    const float dg = gr - gl;
    const float t_in = dg > 0.f ? CPPF_FMA(tr - tl, (-gl) / dg, tl) : tl;
    const float t = gv[0] >= 0.f ? 0.f : (gv[1] <= 0.f ? 1.f : t_in);
    float ex[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float x = CPPF_FMA(D[i], t, P0[i]);
        cs[i] = x;
        cb[i] = clampf(x, lo[i], hi[i]);
        ex[i] = x - cb[i];
    }
    return dot3(ex[0], ex[1], ex[2], ex[0], ex[1], ex[2]);
}

__device__ __forceinline__ float seg_box_dist2(const float (&P0)[3], const float (&P1)[3], const float* __restrict__ lo,
                                               const float* __restrict__ hi) {
    float cs[3], cb[3];
    return seg_box_closest2(P0, P1, lo, hi, cs, cb);
}

__device__ __forceinline__ float seg_box_closest(const float (&P0)[3], const float (&P1)[3], const float* __restrict__ lo,
                                                 const float* __restrict__ hi, float (&cs)[3], float (&cb)[3]) {
    return __builtin_sqrtf(seg_box_closest2(P0, P1, lo, hi, cs, cb));
}

__device__ __forceinline__ float seg_box_dist(const float (&P0)[3], const float (&P1)[3], const float* __restrict__ lo,
                                              const float* __restrict__ hi) {
    return __builtin_sqrtf(seg_box_dist2(P0, P1, lo, hi));
}

// torch.remainder(dq + pi, 2 pi) - pi   (cppflow/evaluation_utils.py:151-153).  For |dq + pi| < 4 pi -- every difference of
// two in-limit joint values -- fmodf is one exact +-2 pi step (Sterbenz), spelled out here; beyond that the fmodf expansion.
__device__ __forceinline__ float wrap_pi(float dq) {
    const float pi = 3.14159265358979323846f, p2 = 2.f * pi;
    const float x = dq + pi;
    float r;
    if (__builtin_expect(fabsf(x) < 2.f * p2, 1)) {
        r = x >= p2 ? x - p2 : x;
        r = r <= -p2 ? r + p2 : r;
    } else {
        r = __builtin_fmodf(x, p2);
    }
    if (r < 0.f) r += p2;
    return r - pi;
}

}  // namespace cppf

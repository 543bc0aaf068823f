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
    float cap_c[CPPF_MAX_CAPSULES][3];    // link-frame centre  fp32(0.5 (p0 + p1))  and half-axis  fp32(0.5 (p1 - p0))  of each capsule
    float cap_h[CPPF_MAX_CAPSULES][3];    // (host: double arithmetic on the fp32 end points of the description, rounded once)
    float cap_a[CPPF_MAX_CAPSULES];       // |h|^2 and 1 / |h|^2 (double arithmetic on the fp32 h, rounded once): a rigid link's
    float cap_ia[CPPF_MAX_CAPSULES];      // constants -- the segment-segment test never divides by a capsule's length
    float cap_r[CPPF_MAX_CAPSULES];
    int32_t cap_begin[CPPF_MAX_DOF + 2];  // capsules of link l (-1..d-1) are [cap_begin[l+1], cap_begin[l+2])
    int8_t cap_link[CPPF_MAX_CAPSULES];   // moving link of each capsule (-1 = base)
    float pair_thr[CPPF_MAX_PAIRS];       // smallest y with sqrt_rn(y) >= r_a + r_b:  sqrt(d2) - (r_a+r_b) < 0  <=>  d2 < y
    float cap_thr[CPPF_MAX_CAPSULES];     // the same for r alone (capsule vs cuboid)
    float pair_cull[CPPF_MAX_PAIRS];      // broad phase: (h_a + h_b + r_a + r_b + 1 cm)^2 (1 + 1e-4), h = half length
    float cap_cull[CPPF_MAX_CAPSULES];    // broad phase: (h + r + 1 cm)^2 (1 + 1e-4)
    uint8_t pair_a[CPPF_MAX_PAIRS];
    uint8_t pair_b[CPPF_MAX_PAIRS];
    float obs_lo[CPPF_MAX_OBSTACLES][3];  // world-frame box corners
    float obs_hi[CPPF_MAX_OBSTACLES][3];
    float jl_lo[CPPF_MAX_DOF];  // padded limits of search.py:46-51
    float jl_hi[CPPF_MAX_DOF];
    int32_t ncaps, npairs, nobs, has_jl;
};

struct LmK {
    double lam_r_d, lam_p_d;  // the damping per row of the dual system, lambda / a_rot^2 and lambda / a_pos^2, formed by the host:
    float lam_r, lam_p;       // wave-uniform values a kernel would otherwise compute with the VALU and park in VGPRs for the whole loop
    float lm_lambda, a_pos, a_rot;
    int32_t n_steps, clamp;
    int32_t n, W;
    float tol_pos2, tol_rot2;  // early-out (cppf_lm_params.tol_*), squared; 0 = off
    float gate_thr;            // conditioning gate of the damped solve: a row whose  max diag(A) * max |y|  exceeds this redoes the
                               // solve in double precision (lm_solve_gated); +inf = never (pure fp32), -inf = always (pure fp64)
    float gate_rel2;           // lean iterations only (kernels_fused.h): ... and exceeds sqrt(gate_rel2) x the scaled residual norm
    int32_t pace_ticks;        // 0 = off; else the fair-share schedule of a launch that fills the chip by itself, in 10 ns ticks of
                               // s_memrealtime per LM iteration (kernels_fused.h: lm_pace; CPPF_TUNE_LM_PACE, the host decides)
};

// One problem of a fused launch: what cppf_lm_pose_steps takes as (x_in, target, S * W, W, outputs).  The same layout sits in the
// kernel-argument segment (FusedArgs::single, the plain launch) and, for a batched launch (cppf_lm_batch_*), in a device table of
// these behind a BatchHeadK -- the kernel picks one or the other base address and reads the fields with scalar loads either way.
struct BatchItemK {
    const float* x_in;
    const float* target;
    cppf_lm_outputs out;
    int32_t n, W;
};

// Head of the device table of a batched launch: block_end[i] = number of workgroups of items 0 .. i (cumulative; entries past the
// last item are 0xffffffff), so that a workgroup finds its item with CPPF_MAX_BATCH - 1 scalar compares on ONE 64-byte scalar load.
struct BatchHeadK {
    uint32_t block_end[CPPF_MAX_BATCH];
};

// The fused kernel's kernel-argument segment: lm_fused_kernel(ChainK, CollK, LmK, BatchItemK single, const void* table) takes these
// five by value, and the segment lays arguments out like the members of a struct (in order, natural alignment), so `single` sits at
// offsetof(FusedArgs, single) from the segment's base -- which is what lets the kernel address it through the same
// constant-address-space pointer type as a table entry.  (One struct argument instead was tried: clang copies a by-value struct
// argument to private memory and relies on the optimiser to elide the copy; with the generic kernels' run-time capsule indices it did
// not, and the 3.6 KB copy went to scratch.)  The host carries a launch's arguments in one of these.
struct FusedArgs {
    ChainK ch;
    CollK co;
    LmK prm;            // (prm.n / prm.W are NOT read by the fused kernel: the item's are)
    BatchItemK single;  // the problem of a plain launch
    const void* table;  // NULL, or the device table { BatchHeadK ; BatchItemK[n_items] } of a batched launch
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

// (a literal zero on either side: the products of two Jacobian entries)
__device__ __forceinline__ float cfma2(float a, float b, float acc) {
    if (__builtin_constant_p(a) && a == 0.f) return acc;
    return cfma(a, b, acc);
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

// The hardware's own sine / cosine (v_sin_f32 / v_cos_f32 take REVOLUTIONS, valid for |x| <= 256 of them; joint angles are bounded by
// the joint limits): 3 instructions instead of 23, 4e-7 of absolute error (scripts/ubench/hw_sincos.hip).  Only an A/B build uses it
// (CPPF_LEAD_SINCOS = 1, kernels_fused.h: in the leading iterations of a fused K-step launch it costs 3 % at the step rate); the shipped
// kernels and everything in the bit-exact set (FK, capsules, metrics, masks) use sincos_cw.
__device__ __forceinline__ void sincos_hw(float x, float& s, float& c) {
    const float r = x * 0.15915494309189535f;
    s = __builtin_amdgcn_sinf(r);
    c = __builtin_amdgcn_cosf(r);
}

// Sine / cosine of an angle inside the joint limits WITHOUT a quadrant reduction: one polynomial each on [-pi, pi] in u = r^2
// (sin r = r P5(u), cos r = Q6(u); least-squares on Chebyshev nodes, 4.9e-7 / 3.8e-7 of absolute error evaluated in fp32 -- the
// rounding of the alternating terms near +-pi, not the truncation), 13 multiply-adds where sincos_cw takes 23 instructions of which
// eight are the integer quadrant logic.  FOLD says how the argument gets into [-pi, pi]: 0 it is there (the joint's limits say so:
// every joint of the shipped robots but Panda's sixth), +1 / -1 one conditional turn down / up (limits within [-pi, 3 pi] /
// [-3 pi, pi]), 2 the general reduction by whole turns.  Like sincos_hw only for the leading iterations of a fused K-step launch
// (CPPF_LEAD_SINCOS = 2, kernels_fused.h), whose iterates nobody sees; never in the bit-exact set.
constexpr float kPiF = 3.14159274101257324f, kTwoPiF = 6.28318548202514648f;
template <int FOLD>
__device__ __forceinline__ void sincos_pi(float x, float& s, float& c) {
    float r = x;
    if constexpr (FOLD == 1) r = x > kPiF ? x - kTwoPiF : x;
    if constexpr (FOLD == -1) r = x < -kPiF ? x + kTwoPiF : x;
    if constexpr (FOLD == 2) {
        const float magic = 12582912.0f;
        const float k = CPPF_FMA(x, 0.15915494309189535f, magic) - magic;  // whole turns, round to nearest
        r = CPPF_FMA(-k, 1.9353071795864769e-3f, CPPF_FMA(-k, 6.28125f, x));
    }
    const float u = r * r;
    float ps = -2.0696598213e-08f, pc = 1.7243808603e-09f;
    ps = CPPF_FMA(ps, u, 2.7087969556e-06f);
    pc = CPPF_FMA(pc, u, -2.7078681342e-07f);
    ps = CPPF_FMA(ps, u, -1.9817604334e-04f);
    pc = CPPF_FMA(pc, u, 2.4769848096e-05f);
    ps = CPPF_FMA(ps, u, 8.3327908069e-03f);
    pc = CPPF_FMA(pc, u, -1.3887801906e-03f);
    ps = CPPF_FMA(ps, u, -1.6666620970e-01f);
    pc = CPPF_FMA(pc, u, 4.1666489094e-02f);
    ps = CPPF_FMA(ps, u, 9.9999994040e-01f);
    pc = CPPF_FMA(pc, u, -4.9999988079e-01f);
    s = ps * r;
    c = CPPF_FMA(pc, u, 1.0f);
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
// SC: 0 the canonical sine / cosine (sincos_cw: everything in the bit-exact set), 1 the hardware's (sincos_hw), 2 the full-range
// polynomials (sincos_pi; `lo`, `hi` = the joint's limits, compile-time constants in the robot-specialised kernels, pick the fold),
// 3 the same polynomials behind the reduction by whole turns (valid for any finite angle)
template <int SC = 0>
__device__ __forceinline__ void fk_joint(float (&R)[9], float (&p)[3], bool prismatic, float q, float lo = 0.f, float hi = 0.f) {
    if (!prismatic) {
        float s, c;
        if constexpr (SC == 1) {
            sincos_hw(q, s, c);
        } else if constexpr (SC == 3) {  // the polynomials behind the general reduction: any finite angle (a launch's own input)
            sincos_pi<2>(q, s, c);
        } else if constexpr (SC == 2) {
            if (lo >= -kPiF && hi <= kPiF)
                sincos_pi<0>(q, s, c);
            else if (lo >= -kPiF && hi <= 3.f * kPiF)
                sincos_pi<1>(q, s, c);
            else if (lo >= -3.f * kPiF && hi <= kPiF)
                sincos_pi<-1>(q, s, c);
            else
                sincos_pi<2>(q, s, c);
        } else {
            sincos_cw(q, s, c);
        }
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

// ---- reciprocal ------------------------------------------------------------------------------------------------------------
// RN(1/x), the correctly rounded reciprocal, in 3 VALU instructions instead of the 10 of hipcc's IEEE division: v_rcp_f32 (1 ulp)
// and ONE Newton step.  That this IS the correctly rounded result for every x with 2^-126 <= |x| < 2^126 -- all 2^32 bit patterns
// were compared on the MI355X itself (scripts/ubench/rcp_exhaustive.hip; profiles/r3_rcp_exhaustive.txt; the test
// tests/test_gpu_round3.py::test_fast_reciprocal_is_correctly_rounded repeats it) -- is what lets the CPU oracle spell the same
// value as a plain `1 / x`.  Below 2^-100 (zero, denormals: v_rcp_f32 does not handle them) the result is DEFINED as 0, which is
// what every caller wants from a vanishing denominator; beyond 2^126 (nothing geometric) the two sides may differ.
__device__ __forceinline__ float rcp_newton(float x) {
    const float y0 = __builtin_amdgcn_rcpf(x);
    return CPPF_FMA(CPPF_FMA(-x, y0, 1.0f), y0, y0);
}
__device__ __forceinline__ float rcp_rn(float x) { return fabsf(x) >= 0x1p-100f ? rcp_newton(x) : 0.f; }
// for denominators that must be positive: 0 for x < 2^-100 (negative and NaN included)
__device__ __forceinline__ float rcp_rn_pos(float x) { return x >= 0x1p-100f ? rcp_newton(x) : 0.f; }

__device__ __forceinline__ float clamp11(float v) { return __builtin_amdgcn_fmed3f(v, -1.f, 1.f); }

// world half-axis of a capsule: R * h (no translation); canonical order shared with the oracle
__device__ __forceinline__ void xform_dir(const float (&R)[9], float h0, float h1, float h2, float (&w)[3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) w[i] = cfma(R[3 * i + 2], h2, cfma(R[3 * i + 1], h1, cmul(R[3 * i], h0)));
}

// ---- distances (canonical order) ------------------------------------------------------------------------------------------
// A capsule's segment is { c + u h : u in [-1, 1] } (centre c, half-axis h; |h|^2 = a and 1 / a are constants of the rigid link).
// Closest points of two segments (Ericson, Real-Time Collision Detection 5.1.9, re-parametrised to [-1, 1]): minimise
// | r + s h1 - t h2 |^2, r = c1 - c2.  One reciprocal (of the Gram determinant) per test, no division by a length.
// Returns the squared distance; c1 / c2 = the closest points (dead code where the caller ignores them).
__device__ __forceinline__ float seg_seg_closest2(const float (&C1)[3], const float (&H1)[3], const float (&C2)[3],
                                                  const float (&H2)[3], float a, float ia, float e, float ie,
                                                  float (&c1)[3], float (&c2)[3]) {
    float rr[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) rr[i] = C1[i] - C2[i];
    const float b = dot3(H1[0], H1[1], H1[2], H2[0], H2[1], H2[2]);
    const float c = dot3(H1[0], H1[1], H1[2], rr[0], rr[1], rr[2]);
    const float f = dot3(H2[0], H2[1], H2[2], rr[0], rr[1], rr[2]);
    const float denom = CPPF_FMA(-b, b, a * e);
    float s = clamp11(CPPF_FMA(b, f, -(c * e)) * rcp_rn_pos(denom));  // parallel segments (denom < 2^-100): s = 0, the centre
    float t = CPPF_FMA(b, s, f) * ie;
    // branch-free form of { t < -1: t = -1, s = clamp((-b - c) / a) ; t > 1: t = 1, s = clamp((b - c) / a) } -- lanes of a wave
    // disagree on these cases all the time, so both candidates are always computed and selected
    const float s_lo = clamp11(-(b + c) * ia), s_hi = clamp11((b - c) * ia);
    s = t < -1.f ? s_lo : (t > 1.f ? s_hi : s);
    t = clamp11(t);
    float df[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        df[i] = CPPF_FMA(-t, H2[i], CPPF_FMA(s, H1[i], rr[i]));
        c1[i] = CPPF_FMA(s, H1[i], C1[i]);
        c2[i] = CPPF_FMA(t, H2[i], C2[i]);
    }
    return dot3(df[0], df[1], df[2], df[0], df[1], df[2]);
}

// squared distance (what the mask-only kernels compare against the sqrt thresholds) and the distance itself
__device__ __forceinline__ float seg_seg_dist2(const float (&C1)[3], const float (&H1)[3], const float (&C2)[3],
                                               const float (&H2)[3], float a, float ia, float e, float ie) {
    float c1[3], c2[3];
    return seg_seg_closest2(C1, H1, C2, H2, a, ia, e, ie, c1, c2);
}

__device__ __forceinline__ float seg_seg_closest(const float (&C1)[3], const float (&H1)[3], const float (&C2)[3],
                                                 const float (&H2)[3], float a, float ia, float e, float ie,
                                                 float (&c1)[3], float (&c2)[3]) {
    return __builtin_sqrtf(seg_seg_closest2(C1, H1, C2, H2, a, ia, e, ie, c1, c2));
}

__device__ __forceinline__ float seg_seg_dist(const float (&C1)[3], const float (&H1)[3], const float (&C2)[3],
                                              const float (&H2)[3], float a, float ia, float e, float ie) {
    return __builtin_sqrtf(seg_seg_dist2(C1, H1, C2, H2, a, ia, e, ie));
}

// exact distance from the segment { C + u H } to the axis-aligned box [lo, hi] (0 when they intersect): root of the
// nondecreasing, piecewise-linear half-derivative g of dist^2, bracketed among u = -1, 1 and the six (clamped) face-crossing
// parameters.
__device__ __forceinline__ float seg_box_closest2(const float (&C)[3], const float (&H)[3], const float* __restrict__ lo,
                                                  const float* __restrict__ hi, float (&cs)[3], float (&cb)[3]) {
    // Along the segment x_i(u) = C_i + u H_i the excess over the slab [lo_i, hi_i] is H_i (u - clamp(u, a_i, b_i)) with
    // [a_i, b_i] the parameter interval in which coordinate i is inside the slab, so the half-derivative of dist^2 is
    //     g(u) = sum_i w_i (u - clamp(u, a_i, b_i)),   w_i = H_i^2
    // nondecreasing and piecewise linear with break points a_i, b_i.  (H_i = 0: w_i = 0, the term vanishes.)
    float w[3], ua[3], ub[3], cand[8], gv[8];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float inv = rcp_rn(H[i]);
        const float u0 = (lo[i] - C[i]) * inv, u1 = (hi[i] - C[i]) * inv;
        ua[i] = fminf(u0, u1);
        ub[i] = fmaxf(u0, u1);
        w[i] = H[i] * H[i];
        cand[2 + 2 * i] = clamp11(ua[i]);
        cand[3 + 2 * i] = clamp11(ub[i]);
    }
    cand[0] = -1.f;
    cand[1] = 1.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float u = cand[k];
        gv[k] = CPPF_FMA(w[2], u - clampf(u, ua[2], ub[2]),
                         CPPF_FMA(w[1], u - clampf(u, ua[1], ub[1]), w[0] * (u - clampf(u, ua[0], ub[0]))));
    }
    // g is monotone, so the bracket of its root is two independent max / min reductions over the candidates:
    //   ul = max{c : g(c) <= 0}, gl = max{g(c) : g(c) <= 0};  ur = min{c : g(c) > 0}, gr = min{g(c) : g(c) > 0}
    float ul = -1.f, gl = gv[0], ur = 1.f, gr = gv[1];
#pragma unroll
    for (int k = 2; k < 8; ++k) {
        const bool neg = gv[k] <= 0.f;
        ul = fmaxf(ul, neg ? cand[k] : -1.f);
        gl = fmaxf(gl, neg ? gv[k] : gv[0]);
        ur = fminf(ur, neg ? 1.f : cand[k]);
        gr = fminf(gr, neg ? gv[1] : gv[k]);
    }
    const float u_in = CPPF_FMA(ur - ul, (-gl) * rcp_rn_pos(gr - gl), ul);  // a flat bracket (gr - gl < 2^-100): its left end
    const float u = gv[0] >= 0.f ? -1.f : (gv[1] <= 0.f ? 1.f : u_in);
    float ex[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float x = CPPF_FMA(H[i], u, C[i]);
        cs[i] = x;
        cb[i] = clampf(x, lo[i], hi[i]);
        ex[i] = x - cb[i];
    }
    return dot3(ex[0], ex[1], ex[2], ex[0], ex[1], ex[2]);
}

__device__ __forceinline__ float seg_box_dist2(const float (&C)[3], const float (&H)[3], const float* __restrict__ lo,
                                               const float* __restrict__ hi) {
    float cs[3], cb[3];
    return seg_box_closest2(C, H, lo, hi, cs, cb);
}

__device__ __forceinline__ float seg_box_closest(const float (&C)[3], const float (&H)[3], const float* __restrict__ lo,
                                                 const float* __restrict__ hi, float (&cs)[3], float (&cb)[3]) {
    return __builtin_sqrtf(seg_box_closest2(C, H, lo, hi, cs, cb));
}

__device__ __forceinline__ float seg_box_dist(const float (&C)[3], const float (&H)[3], const float* __restrict__ lo,
                                              const float* __restrict__ hi) {
    return __builtin_sqrtf(seg_box_dist2(C, H, lo, hi));
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

// The same for N values at once: the one-step form for all of them, straight-line, and ONE (never taken in practice) branch that
// redoes the set through wrap_pi when any |dq + pi| >= 4 pi or is NaN.  Same results bit for bit; what it saves is a divergent
// branch with the fmodf expansion behind it PER VALUE (a (min, max) product step of dp_search wraps 7 .. 12 differences per
// candidate pair: 1 179 instructions per source and two destinations, 442 of them scalar branch bookkeeping, became ~300).
template <int N>
__device__ __forceinline__ void wrap_pi_all(float (&dq)[N]) {
    const float pi = 3.14159265358979323846f, p2 = 2.f * pi;
    float out[N];
    bool far = false;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const float x = dq[j] + pi;
        far |= !(fabsf(x) < 2.f * p2);
        float r = x >= p2 ? x - p2 : x;
        r = r <= -p2 ? r + p2 : r;
        if (r < 0.f) r += p2;
        out[j] = r - pi;
    }
    if (__builtin_expect(far, 0)) {
#pragma unroll
        for (int j = 0; j < N; ++j) out[j] = wrap_pi(dq[j]);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) dq[j] = out[j];
}

// max_j |wrap_pi(dq[j])|: the joint-change measure of dp_search and the trajectory metrics (search.py:115-125)
template <int N>
__device__ __forceinline__ float max_wrapped_change(float (&dq)[N]) {
    wrap_pi_all<N>(dq);
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < N; ++j) m = fmaxf(m, fabsf(dq[j]));
    return m;
}

}  // namespace cppf

// kernels_chain.h -- per-row chain evaluation: FK with joint axes, pose error, geometric Jacobian, the damped solve, the clamp.
// Included inside the anonymous namespace of cppflow_hip.hip and of fused_static.hip (and handed to hipRTC); gfx950 only.
#pragma once

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

// FK keeping every joint's world axis and origin (for the Jacobian).  SC != 0: a cheaper sine / cosine (fk_joint) -- the leading
// iterations of a fused K-step launch only.
template <class RB, int SC = 0>
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
        fk_joint<SC>(R, p, rb.pris(j), q[j], rb.lo(j), rb.hi(j));
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
            // (cmul / cfma: the axis of a specialised chain's first joints has literal 0 / +-1 components)
            J[3][j] = cfma(rz, z1, -cmul(ry, z2));
            J[4][j] = cfma(rx, z2, -cmul(rz, z0));
            J[5][j] = cfma(ry, z0, -cmul(rx, z1));
        } else {
            J[0][j] = J[1][j] = J[2][j] = 0.f;
            J[3][j] = z0, J[4][j] = z1, J[5][j] = z2;
        }
    }
}

// atan2 / asin for the LM residual (roll / yaw / pitch of R_err, three per iteration -- 14 % of the loop with the library
// versions, whose IEEE divisions and special-case handling the residual does not need).  Minimax polynomials fitted for this
// kernel: atan(a) = a + a s P7(s) on a in [0, 1] after the min / max reduction (max abs error 8e-8, relative 1.6e-7),
// asin(x) = x + x z P4(z) on |x| <= 0.5 and pi/2 - 2 asin(sqrt((1 - |x|) / 2)) beyond (3.4e-8 / 1.5e-7).  Both keep full
// RELATIVE accuracy towards 0, which is what the iteration needs as the residual vanishes; exact at 0.
__device__ __forceinline__ float atan2_lm(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(fmaxf(ax, ay), 1e-30f), mn = fminf(ax, ay);
    float inv = __builtin_amdgcn_rcpf(mx);
    inv = CPPF_FMA(CPPF_FMA(-mx, inv, 1.f), inv, inv);  // one Newton step on the 1-ulp reciprocal
    const float a = mn * inv, s = a * a;
    float p = 0.002546269f;
    p = CPPF_FMA(p, s, -0.014814576f);
    p = CPPF_FMA(p, s, 0.040576745f);
    p = CPPF_FMA(p, s, -0.07317518f);
    p = CPPF_FMA(p, s, 0.10549001f);
    p = CPPF_FMA(p, s, -0.14178993f);
    p = CPPF_FMA(p, s, 0.19989419f);
    p = CPPF_FMA(p, s, -0.33332935f);
    float r = CPPF_FMA(a * s, p, a);
    r = ay > ax ? 1.57079632679489661923f - r : r;
    r = x < 0.f ? 3.14159265358979323846f - r : r;
    return __builtin_copysignf(r, y);
}

__device__ __forceinline__ float asin_lm(float x) {  // |x| <= 1
    const float ax = fabsf(x);
    const bool big = ax > 0.5f;
    const float z = big ? CPPF_FMA(-0.5f, ax, 0.5f) : x * x;
    const float t = big ? __builtin_amdgcn_sqrtf(z) : ax;
    float p = 0.04374494f;
    p = CPPF_FMA(p, z, 0.023150224f);
    p = CPPF_FMA(p, z, 0.04570716f);
    p = CPPF_FMA(p, z, 0.07493067f);
    p = CPPF_FMA(p, z, 0.16666822f);
    float r = CPPF_FMA(t * z, p, t);
    r = big ? CPPF_FMA(-2.f, r, 1.57079632679489661923f) : r;
    return __builtin_copysignf(r, x);
}

// The same three angle functions for the LEAN iterations of a fused launch (kernels_fused.h: iterations whose iterate nobody sees,
// already evaluated with 5e-7 sine / cosine polynomials): roll and yaw share ONE reciprocal -- 1 / (mx1 mx2), then times the other
// maximum: one transcendental and five multiplies where two Newton-refined reciprocals take two transcendentals and six
// instructions (a transcendental among multiply-adds costs the SIMD ~12 cycles, profiles/r4_valu_issue_rate_calibration.txt) --
// without a Newton step (v_rcp_f32 is good to 1 ulp; the quotient then to 3e-7 RELATIVE, which keeps the residual's accuracy
// towards 0), and shorter minimax polynomials: atan(a) = a + a s P5(s) on [0, 1] (4.0e-7 absolute in fp32; P7: 8e-8),
// asin(x) = x + x z P3(z) on |x| <= 0.5 (2.9e-8; 6e-8 through the half-angle form beyond).  Measured on one box, alternating builds
// (profiles/r5_ab_lean_trig.txt): 34.90 against 35.24 us per C4 step over 2 000 steps (-1.0 %), 35.86 against 36.53 with the driver's
// 20-step regions (-1.8 %); 586 -> 579 VALU and 9 -> 8 transcendentals per lean iteration.  CPPF_LEAN_TRIG = 0: the canonical
// functions everywhere (the A/B build).
#ifndef CPPF_LEAN_TRIG
#define CPPF_LEAN_TRIG 1
#endif
#ifndef CPPF_LEAN_RPY_FAST
#define CPPF_LEAN_RPY_FAST 1  // the principal-range fast path of pose_error<LEAN> (0: the A/B build)
#endif
__device__ __forceinline__ float atan_lean_poly(float a) {  // atan(a), |a| <= 1: a + a s P5(s), odd in a
    const float s = a * a;
    float p = 0.007374001666903496f;
    p = CPPF_FMA(p, s, -0.03551986813545227f);
    p = CPPF_FMA(p, s, 0.08216774463653564f);
    p = CPPF_FMA(p, s, -0.13398799300193787f);
    p = CPPF_FMA(p, s, 0.1986185610294342f);
    p = CPPF_FMA(p, s, -0.3332539498806f);
    return CPPF_FMA(a * s, p, a);
}
__device__ __forceinline__ float atan_lean_fixup(float a, float ax, float ay, float x, float y) {
    float r = atan_lean_poly(a);
    r = ay > ax ? 1.57079632679489661923f - r : r;
    r = x < 0.f ? 3.14159265358979323846f - r : r;
    return __builtin_copysignf(r, y);
}
__device__ __forceinline__ void atan2_pair_lean(float y1, float x1, float y2, float x2, float& r1, float& r2) {
    const float ax1 = fabsf(x1), ay1 = fabsf(y1), ax2 = fabsf(x2), ay2 = fabsf(y2);
    // (floors of 1e-15: the product of the two maxima must not underflow; atan2(0, 0) = 0 as in atan2_lm)
    const float mx1 = fmaxf(fmaxf(ax1, ay1), 1e-15f), mx2 = fmaxf(fmaxf(ax2, ay2), 1e-15f);
    const float inv = __builtin_amdgcn_rcpf(mx1 * mx2);
    r1 = atan_lean_fixup(fminf(ax1, ay1) * (inv * mx2), ax1, ay1, x1, y1);
    r2 = atan_lean_fixup(fminf(ax2, ay2) * (inv * mx1), ax2, ay2, x2, y2);
}
__device__ __forceinline__ float asin_lean(float x) {  // |x| <= 1
    const float ax = fabsf(x);
    const bool big = ax > 0.5f;

    const float z = big ? CPPF_FMA(-0.5f, ax, 0.5f) : x * x;
    const float t = big ? __builtin_amdgcn_sqrtf(z) : ax;
    float p = 0.05158697068691254f;
    p = CPPF_FMA(p, z, 0.03919339179992676f);
    p = CPPF_FMA(p, z, 0.07554031163454056f);
    p = CPPF_FMA(p, z, 0.16664926707744598f);
    float r = CPPF_FMA(t * z, p, t);
    r = big ? CPPF_FMA(-2.f, r, 1.57079632679489661923f) : r;
    return __builtin_copysignf(r, x);
}

// get_6d_pose_errors without the quaternion detour: the five terms quaternion_to_rpy reads from q_target * q_cur^-1 are
// entries of R_err = R_target * R_cur^T  (cppflow/optimization_utils.py:813-819).  LEAN: the angle functions above.
template <bool LEAN = false>
__device__ __forceinline__ void pose_error(const float (&Rt)[9], const float (&tt)[3], const float (&R)[9],
                                           const float (&p)[3], float (&e)[6]) {
    const float e20 = dot3(Rt[6], Rt[7], Rt[8], R[0], R[1], R[2]);
    const float e21 = dot3(Rt[6], Rt[7], Rt[8], R[3], R[4], R[5]);
    const float e22 = dot3(Rt[6], Rt[7], Rt[8], R[6], R[7], R[8]);
    const float e10 = dot3(Rt[3], Rt[4], Rt[5], R[0], R[1], R[2]);
    const float e00 = dot3(Rt[0], Rt[1], Rt[2], R[0], R[1], R[2]);
    float sp = -e20;
    sp = sp > 1.f ? 1.f : (sp < -1.f ? -1.f : sp);
    if constexpr (LEAN && CPPF_LEAN_TRIG != 0) {
        // The common case of an LM iteration -- every row of the wavefront has all three error angles in their principal ranges
        // (|pitch| <= 30 degrees, |roll|, |yaw| <= 45 degrees: true from the first iteration on for seeds within a few tenths of a
        // radian of the path) -- needs neither the octant / quadrant logic of atan2 nor asin's half-angle form with its square root:
        // roll = atan(e21 / e22), yaw = atan(e10 / e00), pitch = asin(sp) by the SAME polynomials on the SAME quotients, so a row's
        // value does not depend on which branch its wavefront took (the fused multiply-adds are odd in their argument; the
        // reciprocal is of the same product e22 e00 >= 0.37).  One scalar branch; the general lean form stays behind it.
        // A wave-uniform skip of asin's half-angle form alone measured -0.6 % per step (profiles/r5_ab_lean_asin_branch.txt).
        const bool general = !(fabsf(sp) <= 0.5f) | !(fabsf(e21) <= e22) | !(fabsf(e10) <= e00);  // (a NaN goes the general way)
        if (CPPF_LEAN_RPY_FAST != 0 && __builtin_expect(__builtin_amdgcn_ballot_w64(general) == 0ull, 1)) {
            const float inv = __builtin_amdgcn_rcpf(e22 * e00);
            const float a1 = e21 * (inv * e00), a2 = e10 * (inv * e22);
            e[0] = atan_lean_poly(a1);
            e[2] = atan_lean_poly(a2);
            const float z = sp * sp;
            float p = 0.05158697068691254f;
            p = CPPF_FMA(p, z, 0.03919339179992676f);
            p = CPPF_FMA(p, z, 0.07554031163454056f);
            p = CPPF_FMA(p, z, 0.16664926707744598f);
            e[1] = CPPF_FMA(sp * z, p, sp);
        } else {
            atan2_pair_lean(e21, e22, e10, e00, e[0], e[2]);
            e[1] = asin_lean(sp);
        }
    } else {
        e[0] = atan2_lm(e21, e22);
        e[1] = asin_lm(sp);
        e[2] = atan2_lm(e10, e00);
    }
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
// `est` = max diag(A) * max |y|: the a-posteriori size of the rounding error of this fp32 solve in task space, up to the factor
// eps * a_max (lm_solve_gated).
template <int D>
__device__ __forceinline__ void lm_dual_solve_y(const float (&J)[6][D], const float (&e)[6], float lam_r, float lam_p,
                                                float (&y)[6], float& est) {
    // With S = diag(a_rot x3, a_pos x3):  Js^T (Js Js^T + lambda I)^-1 es  =  J^T (J J^T + lambda S^-2)^-1 e,  so the row
    // scaling never has to be applied: it only changes the damping per row (lambda / a^2).  Cholesky A = L L^T with
    // reciprocal pivots and the damping as pivot floor (every exact pivot of A is >= its smallest eigenvalue >= the
    // smallest damping term, so the floor only acts on rounding noise).
    // (lam_r = lambda / a_rot^2, lam_p = lambda / a_pos^2: LmK, formed by the host)
    float L[6][6], inv[6], dmax = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const float lam = j < 3 ? lam_r : lam_p;
#pragma unroll
        for (int i = j; i < 6; ++i) {
            float s = (i == j) ? lam : 0.f;
#pragma unroll
            for (int k = 0; k < D; ++k) s = cfma2(J[i][k], J[j][k], s);  // (literal zeros of a specialised chain's J drop out)
            if (i == j) dmax = fmaxf(dmax, s);  // A[j][j], before the elimination terms
#pragma unroll
            for (int k = 0; k < j; ++k) s = CPPF_FMA(-L[i][k], L[j][k], s);
            if (i == j) {
                s = fmaxf(s, lam);
                inv[j] = __builtin_amdgcn_rsqf(s);  // the hardware reciprocal square root (1 ulp); the correctly rounded library form
                                                   // costs ~12 instructions each and the solve is not part of the bit-exact set
            } else {
                L[i][j] = s * inv[j];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float s = e[i];
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
    est = dmax * fmaxf(fmaxf(fmaxf(fabsf(y[0]), fabsf(y[1])), fmaxf(fabsf(y[2]), fabsf(y[3]))), fmaxf(fabsf(y[4]), fabsf(y[5])));
}

// The same 6x6 solve for the LEAN iterations of a fused launch: block L D L^T with 2x2 pivot blocks -- each block inverted through its
// adjugate, ONE v_rcp_f32 of the determinant per block -- i.e. three transcendentals where the Cholesky factorisation above takes six
// v_rsq_f32 (a transcendental among multiply-adds costs the SIMD ~12 cycles, four times a multiply-add), for three more plain
// instructions.  Algebraically two steps of the scalar elimination at once; the determinant a c - b^2 = a (c - b^2 / a) is floored at
// (its row's damping) x a, which is the scalar factorisation's floor on the second pivot; in fp32 its task-space error is 1.2x the
// Cholesky form's in the median and 1.8x at the 99th percentile (emulation on 1 024 random Panda rows), which an intermediate iterate
// does not notice (the relative gate's forcing term is 1e-3) -- the LAST iteration, a K = 1 launch and the early-out launches keep
// the Cholesky form.  Measured on one box, alternating builds (profiles/r5_ab_lean_block_solve.txt): C4 33.30 against 33.85 us per step
// over 2 000 steps (-1.6 %), 34.76 against 35.38 with the driver's 20-step regions (-1.8 %), C3 (Fetch, 8 joints) 16.05 against 17.44,
// independent random configurations 47.9 against 49.1; 579 -> 585 VALU and 8 -> 5 transcendentals per lean iteration, 128 -> 124 VGPRs.
// CPPF_LEAN_BLOCK_SOLVE = 0: the Cholesky form everywhere (the A/B build).
#ifndef CPPF_LEAN_BLOCK_SOLVE
#define CPPF_LEAN_BLOCK_SOLVE 1
#endif
template <int D>
__device__ __forceinline__ void lm_dual_solve_y_blk(const float (&J)[6][D], const float (&e)[6], float lam_r, float lam_p,
                                                    float (&y)[6], float& est) {
    float A[6][6], dmax = 0.f;  // lower triangle
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            float s = (i == j) ? (i < 3 ? lam_r : lam_p) : 0.f;
#pragma unroll
            for (int k = 0; k < D; ++k) s = cfma2(J[i][k], J[j][k], s);
            A[i][j] = s;
            if (i == j) dmax = fmaxf(dmax, s);
        }
    float w0[6][3], w1[6][3], ia[3], ib[3], ic[3];  // W = A21 A11^-1 per pivot block (rows below it), and the blocks' inverses
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int p = 2 * k;
        const float a = A[p][p], b = A[p + 1][p], c = A[p + 1][p + 1];
        const float det = fmaxf(CPPF_FMA(a, c, -(b * b)), (p + 1 < 3 ? lam_r : lam_p) * a);
        const float r = __builtin_amdgcn_rcpf(det);  // 1 ulp
        ia[k] = c * r, ib[k] = -(b * r), ic[k] = a * r;
#pragma unroll
        for (int i = p + 2; i < 6; ++i) {
            w0[i][k] = CPPF_FMA(A[i][p + 1], ib[k], A[i][p] * ia[k]);
            w1[i][k] = CPPF_FMA(A[i][p + 1], ic[k], A[i][p] * ib[k]);
        }
#pragma unroll
        for (int i = p + 2; i < 6; ++i)
#pragma unroll
            for (int j = p + 2; j <= i; ++j) A[i][j] = CPPF_FMA(-w1[i][k], A[j][p + 1], CPPF_FMA(-w0[i][k], A[j][p], A[i][j]));
    }
    float u[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) u[i] = e[i];
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // forward: u = L^-1 e
        const int p = 2 * k;
#pragma unroll
        for (int i = p + 2; i < 6; ++i) u[i] = CPPF_FMA(-w1[i][k], u[p + 1], CPPF_FMA(-w0[i][k], u[p], u[i]));
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // y = D^-1 u
        const int p = 2 * k;
        y[p] = CPPF_FMA(ib[k], u[p + 1], ia[k] * u[p]);
        y[p + 1] = CPPF_FMA(ic[k], u[p + 1], ib[k] * u[p]);
    }
#pragma unroll
    for (int k = 2; k >= 0; --k) {  // backward: y = L^-T y
        const int p = 2 * k;
#pragma unroll
        for (int i = p + 2; i < 6; ++i) {
            y[p] = CPPF_FMA(-w0[i][k], y[i], y[p]);
            y[p + 1] = CPPF_FMA(-w1[i][k], y[i], y[p + 1]);
        }
    }
    est = dmax * fmaxf(fmaxf(fmaxf(fabsf(y[0]), fabsf(y[1])), fmaxf(fabsf(y[2]), fabsf(y[3]))), fmaxf(fabsf(y[4]), fabsf(y[5])));
}

// delta = J^T y, the second half of the dual solve
template <int D>
__device__ __forceinline__ void lm_dual_apply(const float (&J)[6][D], const float (&y)[6], float (&delta)[D]) {
#pragma unroll
    for (int k = 0; k < D; ++k) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) s = cfma(y[i], J[i][k], s);
        delta[k] = s;
    }
}

template <int D>
__device__ __forceinline__ void lm_dual_solve(const float (&J)[6][D], const float (&e)[6], float lam_r, float lam_p,
                                              float (&delta)[D], float& est) {
    float y[6];
    lm_dual_solve_y<D>(J, e, lam_r, lam_p, y, est);
    lm_dual_apply<D>(J, y, delta);
}

// The same dual solve with everything after the Jacobian in fp64 (cppf_lm_params.solver = CPPF_SOLVER_F64).  In fp32 the step is
// exact to rounding only while cond(J J^T + lambda S^-2) * 6e-8 << 1: with y = A^-1 e of size |e| / (sigma_min^2 + lambda), a
// backward error eps |A| |y| of the factorisation reaches the TASK-space step J delta as eps |A| |e| / (sigma_min^2 + lambda) --
// 1e-4 .. 1e-2 on the 1 - 10 % of rows of a 7-DoF arm with sigma_min(J_s) < 2e-2 (the reference's own fp32 primal LU loses
// 1e-5 .. 7e-4 there; measured tables in DESIGN.md 5.1).  The entries of J are fp32, their products are exact in fp64, so
// this path's step equals the exactly solved step of the fp32 Jacobian to ~1e-13: |J_s (delta - delta_fp64 oracle)| <= 6e-7 on
// ALL rows.  It needs A (21 sums), the factorisation, the BACK substitution and J^T y in fp64 (measured: leaving any of them in
// fp32 gives the fp32 tail back); ~300 v_fma_f64 at half the fp32 rate plus 84 conversions: ~1.5x the iteration time.
template <int D>
__device__ __forceinline__ void lm_dual_solve_f64(const float (&J)[6][D], const float (&e)[6], double lam_r, double lam_p,
                                                  float (&delta)[D]) {
    double A[6][6];  // lower triangle
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) A[i][j] = (i == j) ? (i < 3 ? lam_r : lam_p) : 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        double c[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) c[i] = (double)J[i][k];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) A[i][j] = __builtin_fma(c[i], c[j], A[i][j]);
    }
    double inv[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const double lam = j < 3 ? lam_r : lam_p;
#pragma unroll
        for (int i = j; i < 6; ++i) {
            double s = A[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) s = __builtin_fma(-A[i][k], A[j][k], s);
            if (i == j) {
                s = s > lam ? s : lam;
                double r = __builtin_amdgcn_rsq(s);  // v_rsq_f64 (~2^-26), two Newton steps on r = s^-1/2
                r = __builtin_fma(r * 0.5, __builtin_fma(-s * r, r, 1.0), r);
                r = __builtin_fma(r * 0.5, __builtin_fma(-s * r, r, 1.0), r);
                inv[j] = r;
            } else {
                A[i][j] = s * inv[j];  // L overwrites A
            }
        }
    }
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = (double)e[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s = __builtin_fma(-A[i][k], y[k], s);
        y[i] = s * inv[i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) s = __builtin_fma(-A[k][i], y[k], s);
        y[i] = s * inv[i];
    }
#pragma unroll
    for (int k = 0; k < D; ++k) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) s = __builtin_fma((double)J[i][k], y[i], s);
        delta[k] = (float)s;
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
                inv[j] = __builtin_amdgcn_rsqf(s);  // the hardware reciprocal square root (1 ulp); the correctly rounded library form
                                                   // costs ~12 instructions each and the solve is not part of the bit-exact set
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

// ---- the conditioning-gated damped solve (cppf_lm_params.solver = CPPF_SOLVER_AUTO, the default) -------------------------------
// The task-space error of the fp32 dual solve is E y with E the rounding error of forming / factoring A = J J^T + lambda S^-2,
// |E| ~ eps |A|: it is  ~ eps * a_max * max diag(A) * max |y|  (measured: the true error is 0.2x that in the median, <= 3x over
// 16 384 random rows of the four robots), known only AFTER the fp32 solve because |y| = |A^-1 e| is what blows up
// on a near-singular row whose residual has a component in the weak direction.  A row whose estimate exceeds tau (1e-5 by
// default, in the scaled task-space units of the residual) redoes the solve in double precision.  gate_thr = tau / (eps a_max)
// is formed by the host; +inf gives the pure fp32 solve, -inf the pure fp64 one.  Fewer than 6 joints: the primal fp32 solve.
//
// Row-per-lane kernels: the double-precision solve must not cost the hot path a register (inlined next to the fp32 solve it
// takes the kernel from 115 to 232 VGPRs; capped at 128 the compiler spills 83 registers INSIDE the LM loop).  So the flagged
// lanes of a wavefront hand their J and e to up to kGateSlots LDS slots of their wavefront, and lanes 0 .. n-1 of the same
// wavefront solve one slot each reading J from LDS (twice: forming A, then J^T y), so the solver holds only L, the pivots and y
// besides its own row's state.  Everything stays inside one wavefront: no barrier, LDS accesses of a wavefront complete in
// order.  On a planner's inputs 17 % of the wavefronts take the detour in the first iteration and <= 3 % later (bench.py's C4
// workload, scripts/gate_census.py); independent random configurations (13 % of the rows flagged at first) are the worst case.
constexpr int kGateSlots = 8;
// lean iterations of a fused launch re-solve a flagged row only when the estimated step error also exceeds this fraction of the
// (scaled) residual norm (kernels_fused.h: lm_row_iterate<LEAD = true>)
constexpr float kGateRel = 1e-3f;
// floats of LDS per wavefront (also what the host sizes the residency claim of small launches by, cppflow_hip.hip)
constexpr int gate_lds_floats(int d) { return (6 * d + 6 + 2 * (21 + 6)) * kGateSlots; }
// Upper bound of a fused kernel's STATIC LDS (bytes): the gate's slots of its kBlock / 64 wavefronts (declared for every ndof) + the per-seed
// summary's staging ((ndof + 8) floats per wavefront) -- static_asserted in lm_fused_kernel next to the declarations, and what the
// host sizes its residency claim and its 160 KB feasibility check by (cppflow_hip.hip: fused_static_lds).
constexpr size_t fused_static_lds_bound(int d) {
    return (size_t)(kBlock / 64) * ((size_t)gate_lds_floats(d) + (size_t)d + 8) * sizeof(float) + 64;
}
template <int D>
struct GateLds {
    // per wavefront: slot s, float element i (J [6][D] at i * D + k, e at 6 D + i) at [i * kGateSlots + s]; behind them, as doubles,
    // the 21 entries of the lower triangle of A and the 6 of y: double element m of slot s at [m * kGateSlots + s]
    static constexpr int kJe = (6 * D + 6) * kGateSlots;
    static constexpr int kFloats = gate_lds_floats(D);
};

// entry m (0 .. 20) of the lower triangle, row-major: (i, j), three bits each
constexpr unsigned long long gate_tri_pack(bool rows) {
    unsigned long long p = 0;
    int m = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j <= i; ++j, ++m) p |= (unsigned long long)(rows ? i : j) << (3 * m);
    return p;
}

// The double-precision re-solve of up to kGateSlots rows of a wavefront, by the WHOLE wavefront (it is waiting for it anyway; in a
// launch of one wavefront per SIMD a round is pure latency): (1) the 21 entries of A = J J^T + lambda S^-2 of every slot are one
// task each -- 7 .. 12 FMAs -- spread over the 64 lanes (one lane per slot forming its own A was 21 D FMAs in a row, more than
// half of the round); (2) lanes 0 .. cnt-1 factor and substitute, one slot each, A coming from LDS; (3) the D entries of
// delta = J^T y of every slot are one task each again.  J is only ever read from LDS, by whoever needs an entry.
// Everything stays inside one wavefront: no barrier, the LDS accesses of a wavefront complete in order (s_waitcnt lgkmcnt(0)
// between the stages).
template <int D>
__device__ __forceinline__ void gate_solve_slots(float* __restrict__ w, int cnt, int lane, int nact, double lam_r, double lam_p) {
    double* const wd = reinterpret_cast<double*>(w + GateLds<D>::kJe);
    constexpr unsigned long long kRow = gate_tri_pack(true), kCol = gate_tri_pack(false);
    // (1) A
    // (the tasks go round the ACTIVE lanes: lanes 0 .. nact-1, a prefix of the wavefront -- the last wavefront of a launch may be
    // partly empty)
    const int n_a = cnt * 21;
#pragma unroll 1
    for (int base = 0; base < n_a; base += nact) {
        const int tau = base + lane;
        if (tau < n_a) {
            const int s = (tau * 3121) >> 16;  // tau / 21 (exact for tau < 21 * kGateSlots)
            const int m = tau - 21 * s;
            const int i = (int)((kRow >> (3 * m)) & 7ull), j = (int)((kCol >> (3 * m)) & 7ull);
            double a = i == j ? (i < 3 ? lam_r : lam_p) : 0.0;
            const float* ji = w + (i * D) * kGateSlots + s;
            const float* jj = w + (j * D) * kGateSlots + s;
#pragma unroll
            for (int k = 0; k < D; ++k) a = __builtin_fma((double)ji[k * kGateSlots], (double)jj[k * kGateSlots], a);
            wd[m * kGateSlots + s] = a;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // (2) Cholesky with the damping as pivot floor, forward and back substitution: one lane per slot
    if (lane < cnt) {
        double A[6][6];  // lower triangle; L overwrites it
        {
            int m = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j, ++m) A[i][j] = wd[m * kGateSlots + lane];
        }
        double inv[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const double lam = j < 3 ? lam_r : lam_p;
#pragma unroll
            for (int i = j; i < 6; ++i) {
                double t = A[i][j];
#pragma unroll
                for (int k = 0; k < j; ++k) t = __builtin_fma(-A[i][k], A[j][k], t);
                if (i == j) {
                    t = t > lam ? t : lam;
                    double r = __builtin_amdgcn_rsq(t);  // v_rsq_f64 (~2^-26) + one Newton step: ~1e-15
                    r = __builtin_fma(r * 0.5, __builtin_fma(-t * r, r, 1.0), r);
                    inv[j] = r;
                } else {
                    A[i][j] = t * inv[j];
                }
            }
        }
        double y[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            double t = (double)w[(6 * D + i) * kGateSlots + lane];
#pragma unroll
            for (int k = 0; k < i; ++k) t = __builtin_fma(-A[i][k], y[k], t);
            y[i] = t * inv[i];
        }
#pragma unroll
        for (int i = 5; i >= 0; --i) {
            double t = y[i];
#pragma unroll
            for (int k = i + 1; k < 6; ++k) t = __builtin_fma(-A[k][i], y[k], t);
            y[i] = t * inv[i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) wd[(21 + i) * kGateSlots + lane] = y[i];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // (3) delta = J^T y; entry k of slot s goes to float element k of the slot (the task that overwrites J[0][k] is the only reader of it)
    const int n_d = cnt * D;
#pragma unroll 1
    for (int base = 0; base < n_d; base += nact) {
        const int tau = base + lane;
        if (tau < n_d) {
            const int s = tau / D, k = tau - D * s;
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < 6; ++i) t = __builtin_fma((double)w[(i * D + k) * kGateSlots + s], wd[(21 + i) * kGateSlots + s], t);
            w[k * kGateSlots + s] = (float)t;
        }
    }
}

// One round of the gate, in two halves.  `todo` = ballot of `flag`, non-zero; must be called by every active lane of the wavefront;
// `w`: this wavefront's GateLds<D>::kFloats floats of LDS.
// lm_gate_hand_over: up to kGateSlots of the wavefront's flagged rows write J and e to their slot.  Called BETWEEN the two halves of
// the fp32 solve (y is known, so is the estimate; J^T y is still to come), so that J is not kept in registers a moment longer than
// the fp32 solve itself needs it.  A wavefront with more flagged rows than slots comes back for another round with J recomputed
// (lm_row_iterate) rather than parking 48 floats per row in registers meanwhile.
// lm_gate_solve: lanes 0 .. cnt-1 solve one slot each in double precision; the flagged rows pick their delta up and clear `flag`.
template <int D>
__device__ __forceinline__ int lm_gate_hand_over(const float (&J)[6][D], const float (&e)[6], unsigned long long todo,
                                                 float* __restrict__ w, bool flag) {
    // rank of this lane among the flagged lanes
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(todo >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)todo, 0u));
    if (flag && rank < kGateSlots) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
            for (int k = 0; k < D; ++k) w[(i * D + k) * kGateSlots + rank] = J[i][k];
            w[(6 * D + i) * kGateSlots + rank] = e[i];
        }
    }
    return rank;
}

template <int D>
__device__ __forceinline__ void lm_gate_solve(double lam_r, double lam_p, unsigned long long todo, int rank,
                                              float* __restrict__ w, bool& flag, float (&delta)[D]) {
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int cnt = min((int)__builtin_popcountll(todo), kGateSlots);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slots are written (one wavefront: its LDS accesses complete in order)
    // Every lane of the wavefront works on the slots (gate_solve_slots), flagged itself or not.
    const int nact = (int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(true));
    gate_solve_slots<D>(w, cnt, lane, nact, lam_r, lam_p);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (flag && rank < kGateSlots) {
#pragma unroll
        for (int k = 0; k < D; ++k) delta[k] = w[k * kGateSlots + rank];
        flag = false;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // read back before a later round's writes
}

// clamp_to_joint_limits for one row (cppflow/optimization_utils.py:823-833).  CPPF_CLAMP_MED3 = 1 would make it one v_med3_f32 per
// joint (for lo <= hi and a non-NaN q the median of (q, lo, hi) IS fmin(fmax(q, lo), hi), value for value): 7 instructions fewer per
// iteration and 0.4 % MORE time on one box, alternating builds (35.40 against 35.24 us per C4 step, profiles/r5_ab_lean_trig.txt) --
// one 8-byte VOP3 issues no faster than the two 4-byte VOP2 it replaces.  Not taken.
#ifndef CPPF_CLAMP_MED3
#define CPPF_CLAMP_MED3 0
#endif
template <class RB>
__device__ __forceinline__ void clamp_row(const RB& rb, float (&q)[RB::D]) {
#pragma unroll
    for (int j = 0; j < RB::D; ++j) q[j] = CPPF_CLAMP_MED3 != 0 ? clampf(q[j], rb.lo(j), rb.hi(j)) : fminf(fmaxf(q[j], rb.lo(j)), rb.hi(j));
}

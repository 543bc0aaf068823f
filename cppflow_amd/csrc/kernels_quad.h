// kernels_quad.h -- the latency shape of the fused launch: FOUR lanes cooperate on one (seed, waypoint) row.
// Part of the translation unit cppflow_hip.hip (included inside its anonymous namespace); gfx950 only.
//
// Why.  lm_fused_kernel (one row per lane) is the throughput shape: it needs >= 4 wavefronts per SIMD (262 144 rows) to reach
// the VALU issue ceiling.  A strong-scaling shard (32 768 rows = half a wavefront per SIMD), BASELINE config C2 (8 192 rows)
// and the reference's own cadence sit where one wavefront's instruction count IS the launch time (a lone wavefront issues one
// VALU instruction per ~5.5 cycles whatever it depends on).  This shape cuts the instructions per wavefront by spreading a
// row over a DPP quad (lanes 4r .. 4r+3; quad_perm reads any lane of the quad as an operand modifier, no LDS):
//
//   lane i = 0, 1, 2 holds ROW i of every 3x4 frame [R | p] of the chain; lane 3 holds zeros.  frame * F and the joint
//   rotation only mix entries WITHIN a row, so FK needs no communication at all and each lane does a third of it.
//   sin / cos: lane k evaluates joints k, k+4, k+8 and the quad broadcasts the results.
//   residual: roll / yaw / pitch are evaluated simultaneously in lanes 1 / 0 / 2 by ONE atan2 (pitch as
//   atan2(s, sqrt((1-s)(1+s)))).
//   Jacobian: lane i holds angular row i and linear row i; the cross product takes its other two components by quad rotation.
//   J J^T (6x6, 21 unique entries): 7 entries per lane, either as VALU FMAs on rotated operands or -- MFMA = true -- as
//   v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 rank-1 updates per instruction, i.e. exactly one per quad of the wavefront,
//   with the row-per-lane layout above as its native operand layout (three accumulating chains ZZ^T, ZV^T, VV^T).
//   The 6x6 Cholesky and the substitutions cannot be split three ways for less than they cost (every column needs a
//   broadcast of the pivot row): they run replicated on the broadcast matrix.  J^T y: two FMAs per joint and lane, then
//   a two-step quad butterfly, which leaves delta -- and so q -- replicated bit for bit in the four lanes.
//
// The collision stage stripes the pair / cuboid tests over the four lanes (capsule end points exchanged through LDS, 32 B per
// capsule and row; per-lane pair descriptors from a table the robot handle keeps in device memory).
//
// Numerics: same formulas as the row shape up to the order of a few sums (J J^T accumulation, the pitch formula), i.e. the two
// shapes agree to fp32 rounding on x, not bit for bit; FK / capsule end points / distances use the canonical order, so the
// masks and the cost of a given x are bit-identical between the shapes (and with the fp32 oracle).
#pragma once

constexpr int kQuadRows = kBlock / 4;  // rows per workgroup

typedef float v4f __attribute__((ext_vector_type(4)));

// device-resident per-robot tables of the striped collision stage (cppf_robot::d_quad): one 16-byte record per pair, then one
// per capsule
struct QuadPairRec {
    int32_t a, b;
    float thr, cull;
};
struct QuadCapRec {
    float thr, cull, a, ia;  // sqrt threshold of r, broad-phase threshold, |h|^2, 1 / |h|^2
};

template <int CTRL>
__device__ __forceinline__ float qperm(float x) {  // quad_perm: every lane has a source lane, so no `old` value is needed
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
constexpr int kQRot1 = 0xC9;  // lane i reads lane (i+1) % 3 for i < 3, lane 3 reads itself   [1,2,0,3]
constexpr int kQRot2 = 0xD2;  // lane i reads lane (i+2) % 3                                   [2,0,1,3]
constexpr int kQXor1 = 0xB1;  // [1,0,3,2]
constexpr int kQXor2 = 0x4E;  // [2,3,0,1]

__device__ __forceinline__ float quad_bcast(float x, int j) {  // j is a compile-time constant after unrolling
    switch (j & 3) {
        case 0: return qperm<0x00>(x);
        case 1: return qperm<0x55>(x);
        case 2: return qperm<0xAA>(x);
        default: return qperm<0xFF>(x);
    }
}
__device__ __forceinline__ float quad_sum(float v) {
    v += qperm<kQXor1>(v);
    return v + qperm<kQXor2>(v);
}
__device__ __forceinline__ int quad_or(int v) {
    v |= __builtin_amdgcn_mov_dpp(v, kQXor1, 0xf, 0xf, true);
    return v | __builtin_amdgcn_mov_dpp(v, kQXor2, 0xf, 0xf, true);
}
// Which lane of its quad a thread is, as three predicates computed once (chained comparisons of one integer would be turned
// into a switch, i.e. divergent branches; selects on independent predicates stay v_cndmask).
struct QuadLane {
    bool is0, is1, is2;
    __device__ __forceinline__ explicit QuadLane(int k) : is0(k == 0), is1(k == 1), is2(k == 2) {}
};
// value of lane-slot k out of four replicated candidates
__device__ __forceinline__ float lane_pick(const QuadLane& k, float v0, float v1, float v2, float v3) {
    float r = k.is2 ? v2 : v3;
    r = k.is1 ? v1 : r;
    return k.is0 ? v0 : r;
}

// ---- one row of a frame per lane ------------------------------------------------------------------------------------------------
struct RowFrame {
    float r[3];  // R[i][0..2]
    float p;     // p[i]
};

__device__ __forceinline__ RowFrame row_identity(const QuadLane& k) {
    RowFrame f;
    f.r[0] = k.is0 ? 1.f : 0.f, f.r[1] = k.is1 ? 1.f : 0.f, f.r[2] = k.is2 ? 1.f : 0.f, f.p = 0.f;
    return f;
}

// frame <- frame * F, row i only: the same operation order as fk_fixed (lmik_device.h), so rows agree bit for bit
__device__ __forceinline__ void fk_fixed_row(RowFrame& f, const float (&F)[12]) {
    const float r0 = f.r[0], r1 = f.r[1], r2 = f.r[2];
    f.p = cfma(r2, F[11], cfma(r1, F[10], cfma(r0, F[9], f.p)));
#pragma unroll
    for (int c = 0; c < 3; ++c) f.r[c] = cfma(r2, F[6 + c], cfma(r1, F[3 + c], cmul(r0, F[c])));
}

template <class RB>
__device__ __forceinline__ void fk_fixed_joint_row(const RB& rb, int j, RowFrame& f) {
    float F[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) F[k] = rb.F(j, k);
    fk_fixed_row(f, F);
}

template <class RB>
__device__ __forceinline__ void fk_fixed_ee_row(const RB& rb, RowFrame& f) {
    float F[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) F[k] = rb.Fee(k);
    fk_fixed_row(f, F);
}

__device__ __forceinline__ void fk_joint_row(RowFrame& f, bool prismatic, float q, float s, float c) {
    if (!prismatic) {
        const float a0 = f.r[0], a1 = f.r[1];
        f.r[0] = cfma(s, a1, cmul(c, a0));
        f.r[1] = cfma(c, a1, -cmul(s, a0));
    } else {
        f.p = CPPF_FMA(f.r[2], q, f.p);
    }
}

// sin / cos of every joint, evaluated once per quad: lane k does joints k, k+4, ...; results broadcast
template <class RB>
__device__ __forceinline__ void quad_sincos(const RB& rb, const QuadLane& k, const float (&q)[RB::D], float (&sn)[RB::D], float (&cs)[RB::D]) {
    constexpr int D = RB::D;
#pragma unroll
    for (int s0 = 0; s0 < D; s0 += 4) {
        auto at = [&](int j) { return q[j < D ? j : D - 1]; };
        const float qa = lane_pick(k, at(s0), at(s0 + 1), at(s0 + 2), at(s0 + 3));
        float s, c;
        sincos_cw(qa, s, c);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (s0 + t < D) {
                sn[s0 + t] = quad_bcast(s, t);
                cs[s0 + t] = quad_bcast(c, t);
            }
        }
    }
}

// 6x6 SPD solve on the replicated matrix: A = L L^T with reciprocal pivots and the damping as pivot floor (lm_dual_solve's
// factorisation, started from the assembled matrix).  A holds J J^T (lower triangle used); the damping is added here.
__device__ __forceinline__ void chol6_solve(const float (&A)[6][6], const float (&e)[6], float lam_r, float lam_p, float (&y)[6]) {
    float L[6][6], inv[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const float lam = j < 3 ? lam_r : lam_p;
#pragma unroll
        for (int i = j; i < 6; ++i) {
            float s = (i == j) ? A[i][j] + lam : A[i][j];
#pragma unroll
            for (int t = 0; t < j; ++t) s = CPPF_FMA(-L[i][t], L[j][t], s);
            if (i == j) {
                s = fmaxf(s, lam);
                inv[j] = __builtin_amdgcn_rsqf(s);
            } else {
                L[i][j] = s * inv[j];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float s = e[i];
#pragma unroll
        for (int t = 0; t < i; ++t) s = CPPF_FMA(-L[i][t], y[t], s);
        y[i] = s * inv[i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        float s = y[i];
#pragma unroll
        for (int t = i + 1; t < 6; ++t) s = CPPF_FMA(-L[t][i], y[t], s);
        y[i] = s * inv[i];
    }
}

// One LM iteration of the quad's row; q is replicated in the four lanes and stays so.  Returns true when the row was already
// below the early-out tolerances (then it is left untouched), like lm_row_iterate.
template <class RB, bool MFMA>
__device__ __forceinline__ bool quad_iterate(const RB& rb, const LmK& prm, const QuadLane& k, const float (&Rt)[9], float tt_own,
                                             float lam_r, float lam_p, float (&q)[RB::D], bool lean) {
    constexpr int D = RB::D;
    static_assert(D >= 6, "the quad shape solves the dual 6x6 system (ndof >= 6)");
    float sn[D], cs[D];
    quad_sincos<RB>(rb, k, q, sn, cs);
    RowFrame f = row_identity(k);
    float Zr[D], Or[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        fk_fixed_joint_row(rb, j, f);
        Zr[j] = f.r[2];
        Or[j] = f.p;
        fk_joint_row(f, rb.pris(j), q[j], sn[j], cs[j]);
    }
    fk_fixed_ee_row(rb, f);
    // residual (pose_error of kernels_chain.h): e2b = Rt row 2 . R row b lands in lane b, e10 / e00 in lane 0
    const float t2 = dot3(Rt[6], Rt[7], Rt[8], f.r[0], f.r[1], f.r[2]);
    const float t1 = dot3(Rt[3], Rt[4], Rt[5], f.r[0], f.r[1], f.r[2]);
    const float t0 = dot3(Rt[0], Rt[1], Rt[2], f.r[0], f.r[1], f.r[2]);
    const float ep = tt_own - f.p;
    float sp = -qperm<0x00>(t2);
    sp = sp > 1.f ? 1.f : (sp < -1.f ? -1.f : sp);
    const float e22 = qperm<0xAA>(t2);
    // lane 0: yaw = atan2(e10, e00); lane 1: roll = atan2(e21, e22); lane 2: pitch = asin(sp) = atan2(sp, sqrt((1-sp)(1+sp)))
    const float ay = lane_pick(k, t1, t2, sp, 0.f);
    const float ax = lane_pick(k, t0, e22, __builtin_amdgcn_sqrtf((1.f - sp) * (1.f + sp)), 1.f);
    const float ang = atan2_lm(ay, ax);
    float e[6];
    e[0] = qperm<0x55>(ang), e[1] = qperm<0xAA>(ang), e[2] = qperm<0x00>(ang);
    e[3] = qperm<0x00>(ep), e[4] = qperm<0x55>(ep), e[5] = qperm<0xAA>(ep);
    bool conv = false;
    if (prm.tol_pos2 > 0.f) {  // wave-uniform
        conv = dot3(e[3], e[4], e[5], e[3], e[4], e[5]) < prm.tol_pos2 && dot3(e[0], e[1], e[2], e[0], e[1], e[2]) < prm.tol_rot2;
        if (__builtin_amdgcn_ballot_w64(!conv) == 0ull) return true;
    }
    // Jacobian rows: Z[j] = angular row i, V[j] = linear row i of column j
    float Z[D], V[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        if (!rb.pris(j)) {
            const float rr = f.p - Or[j];
            const float z1 = qperm<kQRot1>(Zr[j]), z2 = qperm<kQRot2>(Zr[j]);
            const float r1 = qperm<kQRot1>(rr), r2 = qperm<kQRot2>(rr);
            Z[j] = Zr[j];
            V[j] = CPPF_FMA(z1, r2, -(z2 * r1));  // (z x r)_i = z_{i+1} r_{i+2} - z_{i+2} r_{i+1}
        } else {
            Z[j] = 0.f;
            V[j] = Zr[j];
        }
    }
    // A = J J^T, replicated into every lane of the quad (angular rows / columns 0..2, linear 3..5)
    float A[6][6];
    if constexpr (MFMA) {
        // one 4x4 block per quad: D[r][c] += a[r] * b[c]; lane c of the quad receives column c in its four result registers
        v4f zz = {0.f, 0.f, 0.f, 0.f}, zv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < D; ++j) {
            zz = __builtin_amdgcn_mfma_f32_4x4x1f32(Z[j], Z[j], zz, 0, 0, 0);
            zv = __builtin_amdgcn_mfma_f32_4x4x1f32(Z[j], V[j], zv, 0, 0, 0);
            vv = __builtin_amdgcn_mfma_f32_4x4x1f32(V[j], V[j], vv, 0, 0, 0);
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                if (b <= a) {
                    A[a][b] = quad_bcast(zz[a], b);          // ZZ^T[a][b]
                    A[3 + a][3 + b] = quad_bcast(vv[a], b);  // VV^T[a][b]
                }
                A[3 + b][a] = quad_bcast(zv[a], b);  // (Z V^T)[a][b] = A[a][3+b] = A[3+b][a]
            }
        }
    } else {
        float zz_d = 0.f, zz_o = 0.f, zv_d = 0.f, zv_1 = 0.f, zv_2 = 0.f, vv_d = 0.f, vv_o = 0.f;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const float Z1 = qperm<kQRot1>(Z[j]), V1 = qperm<kQRot1>(V[j]);
            zz_d = CPPF_FMA(Z[j], Z[j], zz_d);  // (i, i)
            zz_o = CPPF_FMA(Z[j], Z1, zz_o);    // (i, i+1)
            zv_d = CPPF_FMA(Z[j], V[j], zv_d);  // Z_i . V_i
            zv_1 = CPPF_FMA(Z[j], V1, zv_1);    // Z_i . V_{i+1}
            zv_2 = CPPF_FMA(Z1, V[j], zv_2);    // Z_{i+1} . V_i
            vv_d = CPPF_FMA(V[j], V[j], vv_d);
            vv_o = CPPF_FMA(V[j], V1, vv_o);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int n = (i + 1) % 3;  // lane i holds the entries (i, i), (i, n), (n, i)
            const int lo = i < n ? i : n, hi = i < n ? n : i;
            A[i][i] = quad_bcast(zz_d, i);
            A[hi][lo] = quad_bcast(zz_o, i);
            A[3 + i][3 + i] = quad_bcast(vv_d, i);
            A[3 + hi][3 + lo] = quad_bcast(vv_o, i);
            A[3 + i][i] = quad_bcast(zv_d, i);  // A[a][3+b] = Z_a . V_b, stored at [3+b][a]
            A[3 + n][i] = quad_bcast(zv_1, i);  // Z_i . V_n
            A[3 + i][n] = quad_bcast(zv_2, i);  // Z_n . V_i
        }
    }
    float y[6];
    chol6_solve(A, e, lam_r, lam_p, y);
    // delta = J^T y: this lane's two rows, then the quad sum (lane 3 holds zero rows)
    const float ya = lane_pick(k, y[0], y[1], y[2], 0.f), yl = lane_pick(k, y[3], y[4], y[5], 0.f);
    float delta[D];
#pragma unroll
    for (int j = 0; j < D; ++j) delta[j] = quad_sum(CPPF_FMA(V[j], yl, Z[j] * ya));
    // The conditioning gate of the damped solve (kernels_chain.h): a row whose fp32 solve is estimated to be off by more than the
    // gate's tolerance redoes it in double precision.  This shape runs one wavefront per SIMD with registers to spare, so the
    // double-precision solve is simply inlined -- replicated in the quad's four lanes like the fp32 one, on the gathered Jacobian.
    {
        const float dmax = fmaxf(fmaxf(fmaxf(A[0][0], A[1][1]), A[2][2]) + lam_r, fmaxf(fmaxf(A[3][3], A[4][4]), A[5][5]) + lam_p);
        const float ymax = fmaxf(fmaxf(fmaxf(fabsf(y[0]), fabsf(y[1])), fmaxf(fabsf(y[2]), fabsf(y[3]))), fmaxf(fabsf(y[4]), fabsf(y[5])));
        bool flag = !conv && dmax * ymax > prm.gate_thr;  // the same value in the quad's four lanes; NaN: not flagged
        if (__builtin_amdgcn_ballot_w64(flag) != 0ull) {  // wave-uniform
            // the same rule as the row shape (lm_row_iterate): under CPPF_SOLVER_AUTO inside a clamped launch a flagged row whose
            // fp32 step leaves the joint limits keeps that step -- a row's precision must not depend on which shape the batch size picked
            if (prm.clamp && prm.gate_thr > -INFINITY) {
                bool cut = false;
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const float v = q[j] + delta[j];
                    cut |= (v < rb.lo(j)) | (v > rb.hi(j));
                }
                flag = flag && !cut;
            }
            // ... and in an iteration in front of the last one of a plain launch (`lean`, wave-uniform) only when the estimate also
            // exceeds kGateRel of the residual norm the step reduces -- the row shape's rule again (lm_row_iterate<LEAD = true>)
            if (lean && prm.gate_thr > -INFINITY) {
                const float es2 = CPPF_FMA(prm.a_pos * prm.a_pos, dot3(e[3], e[4], e[5], e[3], e[4], e[5]),
                                           prm.a_rot * prm.a_rot * dot3(e[0], e[1], e[2], e[0], e[1], e[2]));
                flag = flag && (dmax * ymax) * (dmax * ymax) > prm.gate_rel2 * es2;
            }
            if (__builtin_amdgcn_ballot_w64(flag) != 0ull) {
            float Jf[6][D], d64[D];
#pragma unroll
            for (int j = 0; j < D; ++j) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    Jf[i][j] = quad_bcast(Z[j], i);
                    Jf[3 + i][j] = quad_bcast(V[j], i);
                    // (opaque: the compiler otherwise folds the DPP move into the f32 -> f64 conversion that follows, and
                    // v_cvt_f64_f32 with a quad_perm control is not an instruction -- hipRTC's build rejects it)
                    asm volatile("" : "+v"(Jf[i][j]), "+v"(Jf[3 + i][j]));
                }
            }
            lm_dual_solve_f64<D>(Jf, e, prm.lam_r_d, prm.lam_p_d, d64);
#pragma unroll
            for (int j = 0; j < D; ++j) delta[j] = flag ? d64[j] : delta[j];
            }
        }
    }
    if (prm.tol_pos2 > 0.f) {
        if (!conv) {
#pragma unroll
            for (int j = 0; j < D; ++j) q[j] += delta[j];
            if (prm.clamp) clamp_row<RB>(rb, q);
        }
        return conv;
    }
#pragma unroll
    for (int j = 0; j < D; ++j) q[j] += delta[j];
    if (prm.clamp) clamp_row<RB>(rb, q);
    return false;
}

// ---- finish: pose metrics and the striped collision stage -------------------------------------------------------------------
// LDS image of one workgroup: the pair / capsule tables, then per row (padded to an odd multiple of 4 words: conflict-free
// component writes) 8 floats per capsule: P0 (3) pad, P1 (3) pad.
__host__ __device__ __forceinline__ int quad_row_stride(int ncaps) { return ncaps * 8 + 4; }

// Registers: two wavefronts per SIMD (256 registers) hold every chain up to 10 joints; at 11 and 12 joints that took 12 ... 148 B of
// scratch per lane, so those instantiations are built for one wavefront per SIMD -- all this shape ever has resident anyway: it
// serves launches of at most 16 384 rows = 1 024 wavefronts, one per SIMD of the chip.
template <class RB>
constexpr int quad_min_waves() {
    return RB::D >= 11 ? 1 : 2;
}
template <class RB, int COLL, bool MFMA>
__global__ __launch_bounds__(kBlock, quad_min_waves<RB>()) void lm_quad_kernel(const ChainK ch, const CollK co, const LmK prm,
                                                            const float* __restrict__ x_in, const float* __restrict__ target,
                                                            const cppf_lm_outputs out, const uint4* __restrict__ tables) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int D = RB::D;
    const RB rb{ch, co};
    const int tid = threadIdx.x, kk = tid & 3, slot = tid >> 2;
    const QuadLane k(kk);
    const size_t row_raw = (size_t)blockIdx.x * kQuadRows + slot;
    const bool active = row_raw < (size_t)prm.n;
    const size_t row = active ? row_raw : (size_t)prm.n - 1;  // idle quads shadow the last row: no divergence, stores predicated
    // the collision stage's pair / capsule records: fetched now, parked in registers across the LM loop, written to LDS at the
    // end (a load issued there would put a trip to L2 on the critical path of a launch that is all latency)
    uint4 rec_pair = {0u, 0u, 0u, 0u}, rec_cap = {0u, 0u, 0u, 0u};
    if constexpr (COLL != 0) {
        if (tid < co.npairs) rec_pair = tables[tid];
        if (tid < co.ncaps) rec_cap = tables[CPPF_MAX_PAIRS + tid];
    }
    float q[D], Rt[9], tt[3];
    load_x<D>(x_in, row, q);
    load_target(target, (int)(row % (size_t)prm.W), Rt, tt);
    const float tt_own = lane_pick(k, tt[0], tt[1], tt[2], 0.f);
    float chk = tt[0] + tt[1] + tt[2];
#pragma unroll
    for (int j = 0; j < D; ++j) chk += q[j];
#pragma unroll
    for (int i = 0; i < 9; ++i) chk += Rt[i];
    const bool bad = !(fabsf(chk) < INFINITY);  // non-finite input: the reference propagates NaN (see lm_fused_kernel)
    const float lam_r = prm.lam_r, lam_p = prm.lam_p;
    int iters = 0;
    for (int it = 0; it < prm.n_steps; ++it) {
        const bool conv = quad_iterate<RB, MFMA>(rb, prm, k, Rt, tt_own, lam_r, lam_p, q, !(prm.tol_pos2 > 0.f) && it < prm.n_steps - 1);
        iters += conv ? 0 : 1;
        if (prm.tol_pos2 > 0.f && __builtin_amdgcn_ballot_w64(!conv) == 0ull) break;
    }
    if (bad) {
#pragma unroll
        for (int j = 0; j < D; ++j) q[j] = __builtin_nanf("");
    }
    if (active) {
        if (out.n_iters && k.is0) out.n_iters[row] = iters;
        if (out.x_out) {  // lane k stores joints k, k+4, ...
#pragma unroll
            for (int s0 = 0; s0 < D; s0 += 4) {
                auto at = [&](int j) { return q[j < D ? j : D - 1]; };
                const float v = lane_pick(k, at(s0), at(s0 + 1), at(s0 + 2), at(s0 + 3));
                if (s0 + kk < D) out.x_out[row * D + s0 + kk] = v;
            }
        }
    }
    const bool want_metrics = out.pos_err_m || out.rot_err_rad;
    if constexpr (COLL == 0) {
        if (!want_metrics) return;
    }
    // ---- final FK, row per lane: end-effector frame for the metrics, capsule end points (component i in lane i) to LDS ----
    const int ncaps = co.ncaps, npairs = co.npairs;
    const QuadPairRec* s_pairs = reinterpret_cast<const QuadPairRec*>(lds);
    const QuadCapRec* s_caps = reinterpret_cast<const QuadCapRec*>(lds) + CPPF_MAX_PAIRS;
    float* s_rows = lds + 4 * (CPPF_MAX_PAIRS + CPPF_MAX_CAPSULES);
    float* mine = s_rows + (size_t)slot * quad_row_stride(ncaps);
    if constexpr (COLL != 0) {
        if (tid < npairs) reinterpret_cast<uint4*>(lds)[tid] = rec_pair;
        if (tid < ncaps) reinterpret_cast<uint4*>(lds)[CPPF_MAX_PAIRS + tid] = rec_cap;
    }
    float sn[D], cs[D];
    quad_sincos<RB>(rb, k, q, sn, cs);
    RowFrame f = row_identity(k);
    auto put_caps = [&](int link) {
        if constexpr (COLL != 0) {
            if (kk < 3) {
                for (int c = co.cap_begin[link + 1]; c < co.cap_begin[link + 2]; ++c) {
                    // xform_point's / xform_dir's order, row i:  R[i][2] c2 + (R[i][1] c1 + (R[i][0] c0 + p[i]))  and
                    // R[i][2] h2 + (R[i][1] h1 + R[i][0] h0)
                    const float wc = CPPF_FMA(f.r[2], co.cap_c[c][2], CPPF_FMA(f.r[1], co.cap_c[c][1], CPPF_FMA(f.r[0], co.cap_c[c][0], f.p)));
                    const float wh = CPPF_FMA(f.r[2], co.cap_h[c][2], CPPF_FMA(f.r[1], co.cap_h[c][1], f.r[0] * co.cap_h[c][0]));
                    mine[c * 8 + kk] = wc;
                    mine[c * 8 + 4 + kk] = wh;
                }
            }
        }
    };
    put_caps(-1);
#pragma unroll
    for (int j = 0; j < D; ++j) {
        fk_fixed_joint_row(rb, j, f);
        fk_joint_row(f, rb.pris(j), q[j], sn[j], cs[j]);
        put_caps(j);
    }
    fk_fixed_ee_row(rb, f);
    float pos_err = 0.f, rot_err = 0.f;
    if (want_metrics || COLL != 0) {
        // pose_metrics of kernels_chain.h, one row of R per lane: E[a][i] = Rt row a . R row i (column i of R_err in lane i);
        // the scalars are then formed from broadcast components in pose_metrics' own operation order, so that for the same x
        // both kernel shapes report the same bits
        const float d = tt_own - f.p;
        const float dx = qperm<0x00>(d), dy = qperm<0x55>(d), dz = qperm<0xAA>(d);
        pos_err = __builtin_sqrtf(CPPF_FMA(dz, dz, CPPF_FMA(dy, dy, dx * dx)));
        const float E0 = dot3(Rt[0], Rt[1], Rt[2], f.r[0], f.r[1], f.r[2]);
        const float E1 = dot3(Rt[3], Rt[4], Rt[5], f.r[0], f.r[1], f.r[2]);
        const float E2 = dot3(Rt[6], Rt[7], Rt[8], f.r[0], f.r[1], f.r[2]);
        const float dg = lane_pick(k, E0, E1, E2, 0.f);  // E[i][i]
        const float m1 = lane_pick(k, E1, E2, E0, 0.f);  // E[i+1][i]
        const float m2 = lane_pick(k, E2, E0, E1, 0.f);  // E[i+2][i]
        const float w = m1 - qperm<kQRot1>(m2);         // lane 0: E10 - E01 (a2), lane 1: E21 - E12 (a0), lane 2: E02 - E20 (a1)
        const float a0 = qperm<0x55>(w), a1 = qperm<0xAA>(w), a2 = qperm<0x00>(w);
        const float sn_ = 0.5f * __builtin_sqrtf(CPPF_FMA(a2, a2, CPPF_FMA(a1, a1, a0 * a0)));
        const float cs_ = 0.5f * (qperm<0x00>(dg) + qperm<0x55>(dg) + qperm<0xAA>(dg) - 1.f);
        const float theta = atan2f(sn_, cs_);
        rot_err = fmaxf(theta, 8.94427191e-4f);
        if (active && k.is0) {
            if (out.pos_err_m) out.pos_err_m[row] = pos_err;
            if (out.rot_err_rad) out.rot_err_rad[row] = rot_err;
        }
    }
    if constexpr (COLL != 0) {
        __syncthreads();  // tables and every quad's end points are in LDS
        auto cap = [&](int c, float (&wc)[3], float (&wh)[3]) {  // centre, half-axis
            const float4 a = *reinterpret_cast<const float4*>(mine + c * 8), b = *reinterpret_cast<const float4*>(mine + c * 8 + 4);
            wc[0] = a.x, wc[1] = a.y, wc[2] = a.z, wh[0] = b.x, wh[1] = b.y, wh[2] = b.z;
        };
        int self_hit = 0, env_hit = 0;
        const bool do_self = out.self_mask || out.ext_cost, do_env = out.env_mask || out.ext_cost;
        if (do_self) {
            for (int p0 = 0; p0 < npairs; p0 += 4) {
                const int pi = p0 + kk;
                const QuadPairRec pr = s_pairs[pi < npairs ? pi : npairs - 1];
                float ca[3], ha[3], cb[3], hb[3];
                cap(pr.a, ca, ha);
                cap(pr.b, cb, hb);
                const bool near = pi < npairs && !(mid_dist2(ca, cb) > pr.cull);
                if (near) {
                    const QuadCapRec ra = s_caps[pr.a], rb2 = s_caps[pr.b];
                    self_hit |= seg_seg_dist2(ca, ha, cb, hb, ra.a, ra.ia, rb2.a, rb2.ia) < pr.thr;
                }
            }
        }
        if (do_env) {
            for (int o = 0; o < co.nobs; ++o) {
                for (int c0 = 0; c0 < ncaps; c0 += 4) {
                    const int c = c0 + kk;
                    const int cc = c < ncaps ? c : ncaps - 1;
                    const QuadCapRec cr = s_caps[cc];
                    float wc[3], wh[3];
                    cap(cc, wc, wh);
                    const bool near = c < ncaps && !(point_box_dist2(wc, co.obs_lo[o], co.obs_hi[o]) > cr.cull);
                    if (near) env_hit |= seg_box_dist2(wc, wh, co.obs_lo[o], co.obs_hi[o]) < cr.thr;
                }
            }
        }
        self_hit = quad_or(self_hit), env_hit = quad_or(env_hit);
        const int jl = jlim_hit<D>(co, q);
        if (active) {
            // one store per lane: lane 0 cost, lanes 1..3 the three masks
            const float cost = 100.f * (float)jl + 1000.f * (float)env_hit + 1000.f * (float)self_hit;
            if (k.is0 && out.ext_cost) out.ext_cost[row] = cost;
            uint8_t* mp = k.is1 ? out.self_mask : (k.is2 ? out.env_mask : out.jlim_mask);
            const int mv = k.is1 ? self_hit : (k.is2 ? env_hit : jl);
            if (!k.is0 && mp) mp[row] = (uint8_t)mv;
        }
    }
}

// kernels_collision.h -- collision stage of one row: capsule FK (registers or LDS), wave-uniform broad phase, exact pair / cuboid tests.
// Part of the one translation unit cppflow_hip.hip (included inside its anonymous namespace); gfx950 only.
#pragma once

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

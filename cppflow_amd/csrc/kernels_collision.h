// kernels_collision.h -- collision stage of one row: capsule FK (registers or LDS), wave-uniform broad phase, exact pair / cuboid tests.
// Included inside the anonymous namespace of cppflow_hip.hip and of fused_static.hip (and handed to hipRTC); gfx950 only.
#pragma once

// ---- collision stage --------------------------------------------------------------------------------------------------------
// A capsule in the world is (centre c, half-axis h).  In the generic kernels those are wave-private scratch indexed by a
// wave-uniform but run-time capsule id, which registers cannot do without spilling; they go to LDS as [capsule*6 + k][lane]
// (k = 0..2 the centre, 3..5 the half-axis) so that a wave's 64 lanes hit 64 consecutive banks.
struct CollOut {
    float min_self, min_env;
    int self_hit, env_hit;
};

// ---- broad phase (mask-only launches) -----------------------------------------------------------------------------------------
// A capsule's segment lies in the ball of radius h (half its length, a constant of the rigid link) about its mid point m,
// so  dist(seg_a, seg_b) >= |m_a - m_b| - h_a - h_b  and  dist(seg_c, box) >= dist(m_c, box) - h_c.  A pair is skipped when
// EVERY active lane of the wavefront has   |m_a - m_b|^2 > (h_a + h_b + r_a + r_b + 1 cm)^2 (1 + 1e-4)   (tabulated, rounded up).
// The mid point IS the capsule's centre, and m_a - m_b is the first thing the exact test needs as well.
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
            lds[(c * 6 + k) * kBlock + tid] = co.cap_c[c][k];
            lds[(c * 6 + 3 + k) * kBlock + tid] = co.cap_h[c][k];
        }
    }
#pragma unroll
    for (int j = 0; j < RB::D; ++j) {
        fk_fixed_joint(rb, j, R, p);
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
}

// Robot-specialised variant: capsule ids, link ids and the pair list are compile-time, so centres and half-axes live in VGPRs
// (static indices after unrolling) and no LDS is touched.  Same canonical operation order as the LDS variant.  Two phases so
// that a caller can retire everything else it holds (target pose, q, the frame) between them: the pair / cuboid tests then
// run with the capsules as the only long-lived registers, which keeps the fused kernel at <= 128 VGPRs, i.e. all
// four wavefronts per SIMD of a 262 144-row launch resident at once (no half-empty second round).
template <class RB>
__device__ __forceinline__ void capsule_fk_static(const RB& rb, const float (&q)[RB::D], float (&R)[9], float (&p)[3],
                                                  float (&wc)[(RB::Table::L > 0 ? RB::Table::L : 1)][3],
                                                  float (&wh)[(RB::Table::L > 0 ? RB::Table::L : 1)][3]) {
    using T = typename RB::Table;
    frame_identity(R, p);
#pragma unroll
    for (int c = 0; c < T::L; ++c) {
        if (T::cap_link[c] < 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                wc[c][k] = T::cap_c[c][k];
                wh[c][k] = T::cap_h[c][k];
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
                xform_point(R, p, T::cap_c[c][0], T::cap_c[c][1], T::cap_c[c][2], wc[c]);
                xform_dir(R, T::cap_h[c][0], T::cap_h[c][1], T::cap_h[c][2], wh[c]);
            }
        }
    }
}

// Which of the cuboids can a capsule reach at all?  Bit o set: SOME lane of the wavefront is within the broad-phase reach of
// cuboid o (wave-uniform).  Evaluated per capsule BEFORE anything of the exact test: the three reciprocals of the half-axis
// the exact test needs do not depend on the cuboid, and a compiler that sees them inside a loop over cuboids hoists them in
// front of it -- i.e. computes them for every capsule of every row whether or not a single test survives the broad phase.
__device__ __forceinline__ uint32_t cuboids_in_reach(const CollK& co, const float (&c)[3], float cull2) {
    uint32_t near = 0u;
    for (int o = 0; o < co.nobs; ++o)
        if (!cull_far(point_box_dist2(c, co.obs_lo[o], co.obs_hi[o]), cull2)) near |= 1u << o;
    return near;
}

template <class RB, bool WANT_MIN>
__device__ __forceinline__ CollOut collide_tests_static(const CollK& co,
                                                        const float (&wc)[(RB::Table::L > 0 ? RB::Table::L : 1)][3],
                                                        const float (&wh)[(RB::Table::L > 0 ? RB::Table::L : 1)][3],
                                                        bool do_self, bool do_env) {
    using T = typename RB::Table;
    // Broad phase of the mask-only launches (see cull_far): one bounding-sphere test per pair / per (capsule, cuboid) on
    // the capsule centres; the exact distance is evaluated only when some lane of the wavefront is within reach.
    CollOut r;
    r.min_self = INFINITY;
    r.self_hit = 0;
    if (do_self) {
#pragma unroll
        for (int pi = 0; pi < T::P; ++pi) {
            const int a = T::pair_a[pi], b = T::pair_b[pi];
            if constexpr (!WANT_MIN) {
                if (cull_far(mid_dist2(wc[a], wc[b]), T::pair_cull[pi])) continue;
            }
            const float d2 = seg_seg_dist2(wc[a], wh[a], wc[b], wh[b], T::cap_a[a], T::cap_ia[a], T::cap_a[b], T::cap_ia[b]);
            if constexpr (WANT_MIN) {
                const float v = __builtin_sqrtf(d2) - (T::cap_r[a] + T::cap_r[b]);
                r.min_self = v < r.min_self ? v : r.min_self;
            } else {
                r.self_hit |= d2 < T::pair_thr[pi];
            }
        }
    }
    if constexpr (WANT_MIN) r.self_hit = r.min_self < 0.f;
    // env: min over capsules per cuboid, "< 0", OR over cuboids (collision_detection.py:39-43) == any (capsule, cuboid) < 0,
    // and the minimum over all of them -- so the loops may nest capsule-outer
    r.min_env = INFINITY;
    r.env_hit = 0;
    if (do_env) {
#pragma unroll
        for (int c = 0; c < T::L; ++c) {
            uint32_t near = ~0u;
            if constexpr (!WANT_MIN) {
                near = cuboids_in_reach(co, wc[c], T::cap_cull[c]);
                if (near == 0u) continue;
            }
            for (int o = 0; o < co.nobs; ++o) {
                if (!((near >> o) & 1u)) continue;
                const float d2 = seg_box_dist2(wc[c], wh[c], co.obs_lo[o], co.obs_hi[o]);
                if constexpr (WANT_MIN) {
                    const float v = __builtin_sqrtf(d2) - T::cap_r[c];
                    r.min_env = v < r.min_env ? v : r.min_env;
                } else {
                    r.env_hit |= d2 < T::cap_thr[c];
                }
            }
        }
        if constexpr (WANT_MIN) r.env_hit = r.min_env < 0.f;
    }
    return r;
}

__device__ __forceinline__ void lds_capsule(const float* __restrict__ lds, int tid, int c, float (&wc)[3], float (&wh)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        wc[k] = lds[(c * 6 + k) * kBlock + tid];
        wh[k] = lds[(c * 6 + 3 + k) * kBlock + tid];
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
            float ca[3], ha[3], cb[3], hb[3];
            lds_capsule(lds, tid, a, ca, ha);
            lds_capsule(lds, tid, b, cb, hb);
            if constexpr (!WANT_MIN) {
                if (cull_far(mid_dist2(ca, cb), co.pair_cull[pi])) continue;
            }
            const float d2 = seg_seg_dist2(ca, ha, cb, hb, co.cap_a[a], co.cap_ia[a], co.cap_a[b], co.cap_ia[b]);
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
        for (int c = 0; c < co.ncaps; ++c) {
            float wc[3], wh[3];
            lds_capsule(lds, tid, c, wc, wh);
            uint32_t near = ~0u;
            if constexpr (!WANT_MIN) {
                near = cuboids_in_reach(co, wc, co.cap_cull[c]);
                if (near == 0u) continue;
            }
            for (int o = 0; o < co.nobs; ++o) {
                if (!((near >> o) & 1u)) continue;
                const float d2 = seg_box_dist2(wc, wh, co.obs_lo[o], co.obs_hi[o]);
                if constexpr (WANT_MIN) {
                    const float v = __builtin_sqrtf(d2) - co.cap_r[c];
                    r.min_env = v < r.min_env ? v : r.min_env;
                } else {
                    r.env_hit |= d2 < co.cap_thr[c];
                }
            }
        }
        if constexpr (WANT_MIN) r.env_hit = r.min_env < 0.f;  // collision_detection.py:39-43 (an OR over cuboids of min < 0)
    }
    return r;
}

// capsule FK + distances for one row; leaves the LAST LINK frame in R, p (the caller applies F_ee for the metrics)
template <class RB, bool WANT_MIN>
__device__ __forceinline__ CollOut collide_row(const RB& rb, const CollK& co, const float (&q)[RB::D], float* lds, int tid,
                                               float (&R)[9], float (&p)[3], bool do_self, bool do_env) {
    if constexpr (RB::kStatic) {
        constexpr int L = RB::Table::L > 0 ? RB::Table::L : 1;
        float wc[L][3], wh[L][3];
        capsule_fk_static<RB>(rb, q, R, p, wc, wh);
        return collide_tests_static<RB, WANT_MIN>(co, wc, wh, do_self, do_env);
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

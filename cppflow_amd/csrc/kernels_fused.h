// kernels_fused.h -- the fused launch and the single-purpose row kernels (FK, Jacobian, pose errors / metrics, clamp, distances, seed validity).
// Included inside the anonymous namespace of cppflow_hip.hip and of fused_static.hip (and handed to hipRTC); gfx950 only.
#pragma once

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
// A workgroup barrier for an exchange through LDS ONLY.  __syncthreads() also waits for every global access in flight
// (s_waitcnt vmcnt(0)) -- here the row's 13 output stores, issued just before: a memory round trip for nothing.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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
        lds_barrier();
    }
    const bool seed_ends_here = ((wave + 1) & (wps - 1)) == 0;  // this wavefront holds the seed's last waypoints
    const bool has_next = active && !(lane == 63 && seed_ends_here);
    const int nw = wave + 1 < kBlock / 64 ? wave + 1 : wave;
    float mrev = 0.f, mpri = 0.f;
    float dq[D], wr[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        float qn = dpp_or_zero<0x130>(q[j]);  // wave_shl:1 -- lane i reads lane i + 1
        qn = (lane == 63) ? s_q[nw][j] : qn;  // (stale but unused when wps == 1: has_next is false there)
        dq[j] = wr[j] = qn - q[j];
    }
    wrap_pi_all<D>(wr);  // (one rare branch for the row instead of one per joint)
#pragma unroll
    for (int j = 0; j < D; ++j) {
        const bool pr = rb.pris(j);
        const float a = nan_to_inf(pr ? fabsf(100.f * dq[j]) : fabsf(rad2deg * wr[j]));
        mpri = fmaxf(mpri, pr ? a : 0.f);
        mrev = fmaxf(mrev, pr ? 0.f : a);
    }
    float v[8] = {nan_to_inf(100.f * rs.pos_err), nan_to_inf(rad2deg * rs.rot_err), has_next ? mrev : 0.f, has_next ? mpri : 0.f,
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
    lds_barrier();
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

// ---- fair-share pacing of a launch that has the chip to itself (opt-in: CPPF_TUNE_LM_PACE) ------------------------------------
// Two wavefronts saturate a SIMD's VALU issue and arbitration is strictly oldest-first, so of the four wavefronts a full-size launch
// puts on every SIMD the two oldest run at the lone-wavefront rate, the third gets the remainder, the fourth nothing, and the last one
// ends up doing half its work alone at half the SIMD's rate (band ends 22 / 27 / 35 / 43 us, profiles/r5_pace_probe.txt).  With
// prm.pace_ticks set, a wavefront compares its progress before every LM iteration with the schedule  start + k x pace  on the chip-wide
// 100 MHz counter and takes one of four priorities (behind by more than half an iteration 3, behind 2, ahead 1, ahead by more than
// half an iteration 0): the bands equalise to 37 - 38 us and a launch in a dependency chain (one stream) takes 43.2 - 44.5 instead of
// 46.4 - 47.2 us.  It COSTS launches that overlap on several streams 2 - 5 % (equalised launches overlap less): off by default, and
// the throughput engine (cppflow_amd.distributed.ShardedRefiner with two streams) never turns it on.  Results do not depend on it.
#ifndef CPPF_LM_PACE
#define CPPF_LM_PACE 1  // 0: compiled out (the A/B build of the headline)
#endif
__device__ __forceinline__ void lm_pace(uint32_t t0, int k, int pace) {
    if (CPPF_LM_PACE == 0 || pace == 0) return;  // (everything here is scalar: wave-uniform integers)
    const int lag = (int)((uint32_t)__builtin_amdgcn_s_memrealtime() - t0) - k * pace, half = pace >> 1;  // (ticks; a launch lasts far less than 2^31 of them)
    if (lag > half)
        __builtin_amdgcn_s_setprio(3);
    else if (lag > 0)
        __builtin_amdgcn_s_setprio(2);
    else if (lag > -half)
        __builtin_amdgcn_s_setprio(1);
    else
        __builtin_amdgcn_s_setprio(0);
}

// ---- one row of the fused launch, in three pieces: load, one LM iteration, finish (store, metrics, collision stage) -----------
template <class RB>
__device__ __forceinline__ void lm_row_load(int W, const float* __restrict__ x_in, const float* __restrict__ target,
                                            size_t row, float (&q)[RB::D], float (&Rt)[9], float (&tt)[3]) {
    load_x<RB::D>(x_in, row, q);
    load_target(target, (int)((uint32_t)row % (uint32_t)W), Rt, tt);
}

// One LM iteration of one row.  Returns true when the row was ALREADY converged at this linearisation point (early-out
// tolerances of cppf_lm_params, off when 0): such a row is left untouched -- the reference's loop likewise stops stepping
// once the pose is valid (cppflow/optimization.py:251-258, 326-358).
// The damped solve is conditioning-gated (kernels_chain.h): a row whose fp32 solve is estimated to be off by more than the gate's
// tolerance is re-solved in double precision by its wavefront through LDS slots.  The common case -- no row of the wavefront
// flagged -- is straight-line code behind ONE scalar branch.  Under CPPF_SOLVER_AUTO inside a clamped loop a row whose fp32 step
// leaves the joint limits is not re-solved: where it lands is decided by the clamp, not by the last digits of the solve (these are
// the rows that sit against a limit iteration after iteration; re-solving them bought nothing and put one wavefront per such row
// 2 us per iteration behind the others).  CPPF_SOLVER_F64 (gate_thr = -inf) re-solves EVERY row, clamped or not, as the header
// promises; so does the single bare step (clamp = 0, the reference's own cadence).
//
// LEAD = true is a lean iteration of a plain K-step launch (iterations 0 .. K-2; FIRST = the launch's first one; 1 .. K-2 the hot
// loop): no early-out tests, no J / e outputs -- its iterate is an intermediate nobody sees.  Its residual's three angle functions are
// the lean ones of kernels_chain.h (pose_error<true>: one shared reciprocal, shorter polynomials, CPPF_LEAN_TRIG).  LEAD = false is the general iteration:
// the LAST iteration of every launch, and every iteration of an early-out launch, whose frozen intermediate iterates ARE results.
// CPPF_LEAD_SINCOS: which sine / cosine the leading iterations use -- in the robot-specialised AND the generic kernels alike (the
// generic ones pick a joint's fold from its limits with a scalar branch where the specialised ones know it at compile time: the same
// arithmetic either way, so the two stay bit-identical, tests/test_gpu_parity.py).
//   0  the canonical one (sincos_cw, 23 instructions per joint) -- the all-canonical A/B build.
//   1  v_sin_f32 / v_cos_f32 (sincos_hw: 3 instructions per joint, 4e-7 absolute).  Measured on one box, alternating builds
//      (profiles/r4_ab_hw_sincos.txt): the C4 step takes 37.45 - 37.51 us with it against 36.33 - 36.44 us with 0 -- 140 fewer VALU
//      instructions per iteration and 3 % MORE time: a transcendental among multiply-adds costs the SIMD ~12 cycles against 2.9
//      (profiles/r4_valu_issue_rate_calibration.txt: "7 fma + 1 v_sin" 4.08 cycles per instruction against 2.89) and its result is
//      needed by the very next instructions of the serial FK chain.  (Round 3 had measured -2.6 % for it on an isolated launch.)
//   2  (default) the full-range polynomials (sincos_pi: 13 plain multiply-adds per joint, no quadrant logic, 5e-7 absolute; one
//      conditional turn for a joint whose limits reach beyond +-pi).  Valid INSIDE the joint limits, which every iterate after the
//      first clamp is; the launch's own input need not be, which is why the FIRST iteration of a launch reduces by whole turns first
//      (mode 3 of fk_joint: 17 multiply-adds per joint, any finite angle).
//      Measured like 1 (profiles/r4_ab_sincos_poly.txt): 34.96 - 35.14 us against 36.35 - 36.49 us per C4 step (-3.8 %; -5.6 % with K - 1
//      lean iterations, which needed a per-wavefront decision on the input).
// Whatever the mode, the LAST iteration is canonical: x_out is one canonical LM step from its predecessor.
#ifndef CPPF_LEAD_SINCOS
#define CPPF_LEAD_SINCOS 2
#endif
#ifndef CPPF_FIRST_LEAN
#define CPPF_FIRST_LEAN 1  // the first iteration of a plain launch: 1 lean with the any-angle form of the polynomials, 0 general
#endif
template <class RB, bool LEAD, bool FIRST = false>
constexpr int lead_sincos() {
    return LEAD ? (FIRST && CPPF_LEAD_SINCOS == 2 ? 3 : CPPF_LEAD_SINCOS) : 0;
}
template <class RB, bool LEAD, bool FIRST = false>
__device__ __forceinline__ bool lm_row_iterate(const RB& rb, const LmK& prm, const cppf_lm_outputs& out, size_t row, bool last,
                                               const float (&Rt)[9], const float (&tt)[3], float* __restrict__ gate_lds,
                                               float (&q)[RB::D]) {
    constexpr int D = RB::D;
    float delta[D];
    bool conv = false;
    {
        float R[9], p[3], ax[D][3], og[D][3], J[6][D], e[6];
        fk_ee_axes<RB, lead_sincos<RB, LEAD, FIRST>()>(rb, q, R, p, ax, og);
        pose_error<LEAD>(Rt, tt, R, p, e);
        if constexpr (!LEAD) {
            if (prm.tol_pos2 > 0.f) {  // wave-uniform
                conv = dot3(e[3], e[4], e[5], e[3], e[4], e[5]) < prm.tol_pos2 && dot3(e[0], e[1], e[2], e[0], e[1], e[2]) < prm.tol_rot2;
                if (__builtin_amdgcn_ballot_w64(!conv) == 0ull) return true;  // every row of the wavefront is done: skip the solve
            }
        }
        jacobian_from_axes<RB>(rb, p, ax, og, J);
        float est = 0.f;
        if constexpr (D < 6) {
            lm_primal_solve<D>(J, e, prm.lm_lambda, prm.a_pos, prm.a_rot, delta);
        } else {
            float y[6];
            if constexpr (LEAD && CPPF_LEAN_BLOCK_SOLVE != 0)
                lm_dual_solve_y_blk<D>(J, e, prm.lam_r, prm.lam_p, y, est);
            else
                lm_dual_solve_y<D>(J, e, prm.lam_r, prm.lam_p, y, est);
            lm_dual_apply<D>(J, y, delta);
        }
        if constexpr (!LEAD) {
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
        }
        if constexpr (D >= 6) {
            bool flag = !conv && est > prm.gate_thr;  // NaN: not flagged (the row is NaN either way)
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(flag) != 0ull, 0)) {  // wave-uniform; rare
                if (prm.clamp && prm.gate_thr > -INFINITY) {  // (CPPF_SOLVER_AUTO only: the all-rows mode re-solves all rows)
                    bool cut = false;
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        const float v = q[j] + delta[j];
                        cut |= (v < rb.lo(j)) | (v > rb.hi(j));
                    }
                    flag = flag && !cut;
                }
                if constexpr (LEAD) {
                    // A lean iteration's iterate is an intermediate: its step has to be accurate RELATIVE to the residual it
                    // reduces, not to 1e-5 absolute -- far from the solution (independent random configurations: 2 - 5 % of the rows
                    // and 80 - 97 % of the wavefronts were flagged in EVERY iteration) a step error of a hundredth of the residual
                    // costs the iteration nothing, and the last iteration, the early-out iterations and the K = 1 launch keep the
                    // absolute bar.  (CPPF_SOLVER_AUTO only; here, behind the rare branch: nothing on the hot path.)
                    if (prm.gate_thr > -INFINITY) {
                        const float es2 = CPPF_FMA(prm.a_pos * prm.a_pos, dot3(e[3], e[4], e[5], e[3], e[4], e[5]),
                                                   prm.a_rot * prm.a_rot * dot3(e[0], e[1], e[2], e[0], e[1], e[2]));
                        flag = flag && est * est > prm.gate_rel2 * es2;
                    }
                }
                unsigned long long todo = __builtin_amdgcn_ballot_w64(flag);
                if (todo != 0ull) {
                    int rank = lm_gate_hand_over<D>(J, e, todo, gate_lds, flag);  // (the last use of J and e)
                    lm_gate_solve<D>(prm.lam_r_d, prm.lam_p_d, todo, rank, gate_lds, flag, delta);
                    // more flagged rows than slots (independent random configurations, first iteration): further rounds, each with
                    // the Jacobian formed again rather than parked in registers while other rows were being solved
                    while ((todo = __builtin_amdgcn_ballot_w64(flag)) != 0ull) {
                        float R2[9], p2[3], ax2[D][3], og2[D][3], J2[6][D], e2[6];
#pragma unroll
                        for (int j = 0; j < D; ++j) asm volatile("" : "+v"(q[j]));  // (unchanged, but the compiler must not know)
                        fk_ee_axes<RB, lead_sincos<RB, LEAD, FIRST>()>(rb, q, R2, p2, ax2, og2);
                        pose_error<LEAD>(Rt, tt, R2, p2, e2);
                        jacobian_from_axes<RB>(rb, p2, ax2, og2, J2);
                        rank = lm_gate_hand_over<D>(J2, e2, todo, gate_lds, flag);
                        lm_gate_solve<D>(prm.lam_r_d, prm.lam_p_d, todo, rank, gate_lds, flag, delta);
                    }
                }
            }
        }
    }
    if constexpr (!LEAD) {
        if (prm.tol_pos2 > 0.f) {  // wave-uniform: the predicated update only exists on the early-out path
            if (!conv) {
#pragma unroll
                for (int j = 0; j < D; ++j) q[j] += delta[j];
                if (prm.clamp) clamp_row<RB>(rb, q);
            }
            return conv;
        }
    }
#pragma unroll
    for (int j = 0; j < D; ++j) q[j] += delta[j];
    if (prm.clamp) clamp_row<RB>(rb, q);
    return false;
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
            float wc[L][3], wh[L][3];
            {
                float R[9], p[3];
                capsule_fk_static<RB>(rb, q, R, p, wc, wh);
                metrics(R, p);
            }
            c = collide_tests_static<RB, COLL == 2>(co, wc, wh, do_self, do_env);
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

// Wavefronts per SIMD the fused kernel is compiled for (128 / 168 / 256 VGPRs): the robot-specialised instantiations fit 128
// registers up to 7 joints (168 beyond) without touching scratch; the generic ones (chain constants in SGPRs, capsules in LDS) get
// one step more room.
template <class RB>
constexpr int lm_waves() {
    return RB::kStatic ? (RB::D <= 7 ? 4 : 3) : (RB::D <= 6 ? 4 : (RB::D <= 8 ? 3 : 2));
}

// A pointer that was loaded from memory is a generic one to the compiler (flat_load / flat_store, which also count on the LDS
// counter lgkmcnt -- the per-seed summary's LDS-only barrier would then wait for the row's output stores after all); the buffers of
// a launch are global memory by contract, so say so: the address-space round trip is what LLVM's InferAddressSpaces keys on.
template <class T>
__device__ __forceinline__ T* as_global(T* p) {
    return (T*)(__attribute__((address_space(1))) T*)p;
}

typedef const __attribute__((address_space(4))) BatchItemK* BatchItemPtr;

// The problem this workgroup belongs to, and the workgroup's index within it.  Plain launch (a.table == NULL): FusedArgs::single in
// the kernel-argument segment, blockIdx.x.  Batched launch: entry b of the device table, b from ONE 64-byte scalar load of the
// cumulative workgroup counts and CPPF_MAX_BATCH - 1 scalar compares.  Either way a constant-address-space pointer: every field is
// fetched with a scalar load when (and where) it is needed, nothing is copied into registers up front.
__device__ __forceinline__ BatchItemPtr fused_item(const void* table, uint32_t& blk) {
    typedef const __attribute__((address_space(4))) char* CPtr;
    blk = blockIdx.x;
    // (the kernel's arguments are laid out like the members of FusedArgs -- in order, each at its natural alignment -- which
    // tests/test_abi.py checks against the code object's own argument offsets)
    BatchItemPtr itp = (BatchItemPtr)((CPtr)__builtin_amdgcn_kernarg_segment_ptr() + __builtin_offsetof(FusedArgs, single));
    if (table) {  // wave-uniform
        const CPtr tab = (CPtr)(uintptr_t)table;
        const __attribute__((address_space(4))) BatchHeadK* hd = (const __attribute__((address_space(4))) BatchHeadK*)tab;
        uint32_t b = 0, begin = 0;
#pragma unroll
        for (int i = 0; i < CPPF_MAX_BATCH - 1; ++i) {
            const uint32_t e = hd->block_end[i];
            const bool past = blk >= e;
            b = past ? (uint32_t)(i + 1) : b;
            begin = past ? e : begin;
        }
        blk -= begin;
        itp = (BatchItemPtr)(tab + sizeof(BatchHeadK)) + b;
    }
    return itp;
}

__device__ __forceinline__ cppf_lm_outputs fused_outputs(BatchItemPtr itp) {
    cppf_lm_outputs o;
    o.x_out = as_global(itp->out.x_out), o.J_out = as_global(itp->out.J_out), o.e_out = as_global(itp->out.e_out);
    o.pos_err_m = as_global(itp->out.pos_err_m), o.rot_err_rad = as_global(itp->out.rot_err_rad);
    o.self_mask = as_global(itp->out.self_mask), o.env_mask = as_global(itp->out.env_mask), o.jlim_mask = as_global(itp->out.jlim_mask);
    o.ext_cost = as_global(itp->out.ext_cost), o.min_self = as_global(itp->out.min_self), o.min_env = as_global(itp->out.min_env);
    o.seed_summary = as_global(itp->out.seed_summary), o.n_iters = as_global(itp->out.n_iters);
    return o;
}

// COLL: 0 = no collision stage, 1 = masks / cost only (no square roots), 2 = masks / cost and the signed minimum distances.
// out.seed_summary (host: only when W is 64, 128 or 256 and COLL != 0) adds the per-seed reduction as an epilogue.
// The precision of the damped solve is a run-time parameter (prm.gate_thr, lm_solve_gated).
// Registers: the masks-only instantiations of the shipped robots need <= 128 VGPRs without being told to (Panda 115; round 2
// needed a second, occupancy-capped build with 36 B of scratch per lane for that), so all four wavefronts per SIMD of a
// 262 144-row launch are resident at once and no launch touches scratch.
// One grid serves ONE problem (cppf_lm_pose_steps) or up to CPPF_MAX_BATCH independent ones laid end to end (cppf_lm_batch_launch):
// each workgroup belongs to exactly one problem (fused_item), so a problem's rows see exactly the code of a launch of their own.
template <class RB, int COLL>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(lm_waves<RB>(), lm_waves<RB>()))) void lm_fused_kernel(const ChainK ch, const CollK co, const LmK prm,
                                                          const BatchItemK single, const void* table) {
    extern __shared__ float lds[];
    (void)single;  // read through the kernel-argument segment's address (fused_item), like a table entry
    constexpr int D = RB::D;
    __shared__ float s_gate[kBlock / 64][GateLds<D>::kFloats];  // the conditioning gate's slots, per wavefront (lm_solve_gated)
    // (+ block_seed_summary's s_q [kBlock / 64][D] and s_red [kBlock / 64][8]: the host's bound must cover all of it)
    static_assert(sizeof(s_gate) + (kBlock / 64) * (D + 8) * sizeof(float) <= fused_static_lds_bound(D), "fused_static_lds_bound is stale");
    const RB rb{ch, co};
    uint32_t blk;
    const BatchItemPtr itp = fused_item(table, blk);
    const cppf_lm_outputs out = fused_outputs(itp);
    const int tid = threadIdx.x;
    const size_t row = (size_t)(blk * (unsigned)kBlock + (unsigned)tid);  // n < 2^31 (host): one register across the LM loop
    const bool active = row < (size_t)itp->n;
    float q[D];
#pragma unroll
    for (int j = 0; j < D; ++j) q[j] = 0.f;
    RowSummary rs;
    if (active) {
        float Rt[9], tt[3];
        lm_row_load<RB>(itp->W, as_global(itp->x_in), as_global(itp->target), row, q, Rt, tt);
        // A non-finite input stays non-finite in the reference (torch.clamp and the solve propagate NaN); here fminf / fmaxf
        // of the clamp would turn it into a joint limit, so such a row is poisoned after the loop instead (once per launch).
        float chk = tt[0] + tt[1] + tt[2];
#pragma unroll
        for (int j = 0; j < D; ++j) chk += q[j];
#pragma unroll
        for (int k = 0; k < 9; ++k) chk += Rt[k];
        // (as a wavefront mask, formed NOW: left alone the compiler sinks the whole sum behind the LM loop and keeps its 19
        // inputs -- the row's initial q among them -- alive across it, which is what the last spilled registers were)
        unsigned long long bad_mask = __builtin_amdgcn_ballot_w64(!(fabsf(chk) < INFINITY));
        asm volatile("" : "+s"(bad_mask));
        float* const gate_lds = s_gate[__builtin_amdgcn_readfirstlane(tid >> 6)];  // wave-uniform: a scalar base
        const uint32_t pace_t0 = (CPPF_LM_PACE != 0 && prm.pace_ticks != 0) ? (uint32_t)__builtin_amdgcn_s_memrealtime() : 0u;  // (lm_pace)
        int iters = 0, it = 0;
        // A plain launch: the FIRST iteration lean too, but with the polynomials behind a reduction by whole turns (the launch's own
        // input need not lie inside the joint limits, which the plain lean iteration's sine / cosine assumes, CPPF_LEAD_SINCOS = 2;
        // every later iterate has been through the clamp -- K > 1 implies clamp = 1), then the hot loop: K - 2 lean iterations, then
        // -- below -- the last one, general and canonical.  Decided per ITERATION, never per wavefront: a row's result does not depend
        // on which rows share its wavefront.  Three straight-line regions: with ONE copy of each body -- both in one loop behind a
        // branch, or an outer loop of two passes -- the same arithmetic ran 6 % slower (36.6 against 34.3 us per C4 step) or spilled
        // 40 B per lane.  (CPPF_FIRST_LEAN = 0 makes the first iteration a general one: +0.6 %, profiles/r4_ab_first_lean.txt.)
        if (!(prm.tol_pos2 > 0.f)) {  // wave-uniform
            if (prm.n_steps >= 2) {
                (void)lm_row_iterate<RB, CPPF_FIRST_LEAN != 0, true>(rb, prm, out, row, false, Rt, tt, gate_lds, q);
                it = 1;
            }
            for (; it < prm.n_steps - 1; ++it) {
                lm_pace(pace_t0, it, prm.pace_ticks);
                (void)lm_row_iterate<RB, true>(rb, prm, out, row, false, Rt, tt, gate_lds, q);
            }
            iters = it;
            lm_pace(pace_t0, it, prm.pace_ticks);  // (the last, canonical iteration)
        }
        // the general iteration: the last one of a plain launch, every one of an early-out launch
        for (; it < prm.n_steps; ++it) {
            const bool conv = lm_row_iterate<RB, false>(rb, prm, out, row, it == prm.n_steps - 1, Rt, tt, gate_lds, q);
            iters += conv ? 0 : 1;
            if (prm.tol_pos2 > 0.f && __builtin_amdgcn_ballot_w64(!conv) == 0ull) break;
        }
        // Row index and everything derived from it (the byte offsets of a dozen outputs) are formed AGAIN behind the loop
        // from the work-item id -- opaque to the compiler, which would otherwise carry five registers of them across the loop
        // (the last ones it spilled to scratch at 128 VGPRs): the loop itself only needs the index when it stores J / e.
        int tid_b = threadIdx.x;
        asm volatile("" : "+v"(tid_b));
        const size_t row_b = (size_t)(blk * (unsigned)kBlock + (unsigned)tid_b);
        if (out.n_iters) out.n_iters[row_b] = iters;
        if ((bad_mask >> (tid_b & 63)) & 1ull) {
#pragma unroll
            for (int j = 0; j < D; ++j) q[j] = __builtin_nanf("");
        }
        lm_pace(pace_t0, prm.n_steps, prm.pace_ticks);  // (the finish stage: one more slot of the schedule)
        lm_row_finish<RB, COLL>(rb, co, out, lds, tid_b, row_b, Rt, tt, q, rs);
    }
    if constexpr (COLL != 0) {
        if (out.seed_summary) {
            int tid_c = threadIdx.x;
            asm volatile("" : "+v"(tid_c));
            block_seed_summary<RB>(rb, itp->W, (size_t)(blk * (unsigned)kBlock + (unsigned)tid_c), active, q, rs, out.seed_summary);
        }
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
            float wc[3], wh[3];
            lds_capsule(lds, tid, c, wc, wh);
            dists[row * co.ncaps + c] = seg_box_dist(wc, wh, lo, hi) - co.cap_r[c];
        }
    } else {
        for (int pi = 0; pi < co.npairs; ++pi) {
            const int a = co.pair_a[pi], b = co.pair_b[pi];
            float ca[3], ha[3], cb[3], hb[3];
            lds_capsule(lds, tid, a, ca, ha);
            lds_capsule(lds, tid, b, cb, hb);
            dists[row * co.npairs + pi] =
                seg_seg_dist(ca, ha, cb, hb, co.cap_a[a], co.cap_ia[a], co.cap_a[b], co.cap_ia[b]) - (co.cap_r[a] + co.cap_r[b]);
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
    const float v = x[i];
    x[i] = v != v ? v : fminf(fmaxf(v, ch.lo[j]), ch.hi[j]);  // torch.clamp keeps a NaN (fminf / fmaxf would drop it)
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
        mp = fmaxf(mp, nan_to_inf(100.f * pe));
        mr = fmaxf(mr, nan_to_inf(rad2deg * re));
        if (w + 1 < W) {
            float qn[D], dq[D], wr[D];
            load_x<D>(x, row + 1, qn);
#pragma unroll
            for (int j = 0; j < D; ++j) dq[j] = wr[j] = qn[j] - q[j];
            wrap_pi_all<D>(wr);
#pragma unroll
            for (int j = 0; j < D; ++j) {
                if (rb.pris(j))
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
    }
    if (threadIdx.x == 0) {
        out[s * 4 + 0] = mp, out[s * 4 + 1] = mr, out[s * 4 + 2] = mrev, out[s * 4 + 3] = mpri;
    }
}

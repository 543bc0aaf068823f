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

#include <dlfcn.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/cppflow_hip_debug.h"
#include "lmik_device.h"
#include "robots_gen.h"

using namespace cppf;

namespace cppf {
// fused_static.hip: lm_fused_kernel<StaRobot<table static_id>, coll> -- the row-shape fused kernel of the shipped robots lives in
// its own translation unit (another machine scheduler, see there)
bool launch_fused_static(int static_id, int coll, unsigned grid, size_t lds, hipStream_t st, const FusedArgs& args);
int fused_static_block();
}  // namespace cppf

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

// the device code, by topic (everything below lives in this anonymous namespace; the row-shape fused kernel of the shipped
// robots is instantiated in fused_static.hip instead)
#include "kernels_chain.h"
#include "kernels_collision.h"
#include "kernels_fused.h"
#include "kernels_quad.h"
#include "kernels_eval.h"
#include "kernels_coupled.h"
#include "kernels_dp.h"

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

namespace {
struct RtcModule;  // rtc_specialize.h
}

struct cppf_robot {
    cppf_robot_desc desc;
    ChainK chain;
    CollK coll;
    int device;
    int cu_count;      // compute units of `device` (0 for the host-only handle): what a resident launch's grid is held against
    int static_id;     // index into robots_gen.h when the description equals a generated table, else -1
    size_t lds_bytes;  // generic path only: capsule end points, 6 floats per capsule per lane
    void* d_quad;      // device: QuadPairRec[CPPF_MAX_PAIRS] then QuadCapRec[CPPF_MAX_CAPSULES] (quad shape's striped collision stage)
    RtcModule* rtc;    // kernels compiled for this description by cppf_robot_specialize (hipRTC), else NULL
    // Test / tuning switches of THIS handle (cppf_debug_set, include/cppflow_hip_debug.h): no process-wide dispatch state, two
    // handles on two threads can hold different settings.  Atomics: a setter may race with a launch on another thread.
    mutable std::atomic<int> tune[CPPF_TUNE_COUNT];
    // Lifetime (cppflow_hip.h, "Ownership"): the number of live cppf_lm_batch objects bound to this handle, and kRobotDead once
    // cppf_robot_destroy has been called.  ONE word, so that whoever takes it to { dead, 0 batches } with its single atomic
    // read-modify-write is the one who frees the handle -- and nobody else touches it afterwards.
    mutable std::atomic<uint32_t> life;
};
constexpr uint32_t kRobotDead = 0x80000000u;

namespace {
// defaults of the switches (the measured crossovers; see include/cppflow_hip_debug.h)
constexpr int kTuneDefaults[CPPF_TUNE_COUNT] = {
    /* FORCE_GENERIC */ 0, /* PCR_MAX_ROWS */ -1, /* QUAD_MAX_ROWS */ 16384, /* DP_PERSISTENT */ 1,
    /* FULL_ROWS */ 1,     /* PCR_LDS */ 2,       /* ROWS_POSE */ 0,         /* QUAD_MFMA */ 0,
    /* SPREAD_KB */ 42,    /* DP_SPIN_LOG2 */ 22,    /* GATE_REL_PPM */ (int)(kGateRel * 1e6f + 0.5f), /* CU_COUNT */ -1,
    /* LM_PACE */ 0,
};
inline int tune(const cppf_robot* rb, int key) { return rb->tune[key].load(std::memory_order_relaxed); }

// static LDS of a fused kernel: the gate's slots of its four wavefronts + the summary staging -- the bound the kernel itself
// static_asserts next to its __shared__ declarations (fused_static_lds_bound, kernels_fused.h)
inline size_t fused_static_lds(const cppf_robot* rb) { return fused_static_lds_bound(rb->desc.ndof); }
// Dynamic LDS (bytes) a fused row-shape launch of n rows claims purely to bound how many of its workgroups share a compute unit
// (CPPF_TUNE_SPREAD_KB; 0 = none): static LDS (the conditioning gate's slots: 13 KB at 7 joints, 17 KB at 12) + the claim stays
// under the 64 KB a workgroup may have without asking, and two such workgroups fit on a unit's 160 KB where a third does not.  Only
// launches of at most 128 workgroups take it -- half a workgroup per compute unit, so the four launches the hardware keeps in
// flight (profiles/r3_streams_sweep.txt) still all fit.  Measured on a 32 768-row shard, four launches in flight: 8.46 -> 7.1 ..
// 7.8 us per step (the dispatcher otherwise stacks the launches four deep on the units it tries first and leaves others idle);
// at 65 536 rows the same claim would hold two of the four launches back (10.1 -> 12.8 us), hence the bound.
inline size_t fused_spread_lds(const cppf_robot* rb, size_t n) {
    const int kb = tune(rb, CPPF_TUNE_SPREAD_KB);
    if (kb <= 0 || n > (size_t)128 * kBlock) return 0;
    const size_t stat = fused_static_lds(rb);
    const size_t room = stat + 512 < 65536 ? 65536 - 512 - stat : 0;
    return std::min((size_t)kb * 1024, room);
}
}  // namespace

#include "rtc_specialize.h"

namespace {

// smallest fp32 y with sqrt_rn(y) >= r (host sqrtf is correctly rounded): sqrtf(d2) - r < 0  <=>  d2 < y for all d2 >= 0
float sqrt_threshold(float r) {
    if (!(r > 0.f)) return 0.f;
    float y = r * r;
    while (std::sqrt(y) >= r) y = std::nextafterf(y, 0.f);
    while (std::sqrt(y) < r) y = std::nextafterf(y, INFINITY);
    return y;
}

// A capsule as the kernels use it: centre c = 0.5 (p0 + p1), half-axis h = 0.5 (p1 - p0), a = |h|^2, 1 / a -- double arithmetic
// on the fp32 end points of the description, each rounded to fp32 once (a over the ROUNDED h, summed (h0 h0 + h1 h1) + h2 h2).
// The same lines are in cppflow_amd/gen_robots.py (capsule_centred) and oracle/lmik_oracle.c (orc_robot_create).
struct CapsuleCentred {
    float c[3], h[3], a, ia;
    double half_length;  // sqrt of the unrounded a
};
CapsuleCentred capsule_centred(const cppf_robot_desc& d, int cap) {
    CapsuleCentred r;
    for (int k = 0; k < 3; ++k) {
        r.c[k] = (float)(0.5 * ((double)d.cap_p0[cap][k] + (double)d.cap_p1[cap][k]));
        r.h[k] = (float)(0.5 * ((double)d.cap_p1[cap][k] - (double)d.cap_p0[cap][k]));
    }
    const double h0 = r.h[0], h1 = r.h[1], h2 = r.h[2];
    const double a = (h0 * h0 + h1 * h1) + h2 * h2;
    r.a = (float)a;
    r.ia = a >= 0x1p-100 ? (float)(1.0 / a) : 0.f;  // a zero-length capsule is a sphere: its parameter stays 0 (as rcp_rn would have it)
    r.half_length = std::sqrt(a);
    return r;
}

// broad-phase thresholds (see cull_far): (reach + 1 cm)^2 (1 + 1e-4), rounded up to fp32
double cap_half_length(const cppf_robot_desc& d, int c) { return capsule_centred(d, c).half_length; }

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
    if ((robot)->static_id >= 0 && !tune((robot), CPPF_TUNE_FORCE_GENERIC)) {                                                                \
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
            case 11: { using RB = DynRobot<11>; CPPF_BODY; } break;                                                   \
            case 12: { using RB = DynRobot<12>; CPPF_BODY; } break;                                                   \
            default: return fail(CPPF_ERR_UNSUPPORTED, "cppflow_hip: kernels are built for ndof in 3..12");     \
        }                                                                                                             \
    }

// the generic instantiations only (the caller has dealt with the generated tables)
#define CPPF_DISPATCH_DYN(robot)                                                                                      \
    switch ((robot)->desc.ndof) {                                                                                     \
        case 3: { using RB = DynRobot<3>; CPPF_BODY; } break;                                                         \
        case 4: { using RB = DynRobot<4>; CPPF_BODY; } break;                                                         \
        case 5: { using RB = DynRobot<5>; CPPF_BODY; } break;                                                         \
        case 6: { using RB = DynRobot<6>; CPPF_BODY; } break;                                                         \
        case 7: { using RB = DynRobot<7>; CPPF_BODY; } break;                                                         \
        case 8: { using RB = DynRobot<8>; CPPF_BODY; } break;                                                         \
        case 9: { using RB = DynRobot<9>; CPPF_BODY; } break;                                                         \
        case 10: { using RB = DynRobot<10>; CPPF_BODY; } break;                                                       \
        case 11: { using RB = DynRobot<11>; CPPF_BODY; } break;                                                       \
        case 12: { using RB = DynRobot<12>; CPPF_BODY; } break;                                                       \
        default: return fail(CPPF_ERR_UNSUPPORTED, "cppflow_hip: kernels are built for ndof in 3..12");         \
    }

constexpr int kNoDevice = -12345;  // cppf_robot_create's host-only mode (no HIP call), for cppf_debug_rtc_compile
// coupled step: parallel-in-time elimination up to this many (trajectory, waypoint) rows -- the measured crossovers with the
// two-ended row-per-lane kernels at d <= 7 (x 0.5 at d = 8): 512 trajectories x 256 waypoints with the state in LDS (W <= 256: one
// workgroup per compute unit, so the time steps up at every multiple of 256 trajectories), 192 x 256 with the state in the workspace
constexpr int kPcrMaxRowsLds = 131072, kPcrMaxRowsGlobal = 49152;
// the split form of the LDS-resident reduction (two wavefront-uniform halves per waypoint) up to this many joints: -6 ... -8 % at
// d = 7 (one Panda trajectory 64.4 -> 59.1 us), -5 % at d = 8 (Fetch 85.6 -> 81.4 us) since round 4 -- in round 3 it spilled 300 B
// per lane there and lost 9 %; profiles/r4_pcr_ab.txt
constexpr int kPcrSplitMaxD = 8;

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
        case 11: { constexpr int D = 11; __VA_ARGS__; } break;                                                \
        case 12: { constexpr int D = 12; __VA_ARGS__; } break;                                                \
        default: return fail(CPPF_ERR_UNSUPPORTED, "cppflow_hip: kernels are built for ndof in 3..12");  \
    }

// launch one of the run-time-specialised kernels of a handle (same argument lists as the compiled-in instantiations)
int rtc_launch(const cppf_robot* rb, RtcKernel which, unsigned grid, size_t lds, hipStream_t st, void** args);

inline bool use_rtc(const cppf_robot* rb) { return rb->rtc != nullptr && !tune(rb, CPPF_TUNE_FORCE_GENERIC); }

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
    CPPF_REQUIRE(!((rb)->life.load(std::memory_order_acquire) & kRobotDead),                                         \
                 "the robot handle was destroyed (cppf_robot_destroy; it is only kept allocated for its live batches)"); \
    DeviceGuard device_guard__((rb)->device);                                                                        \
    if (device_guard__.err != hipSuccess)                                                                            \
        return fail(CPPF_ERR_HIP, std::string("cppflow_hip: selecting the robot's device failed: ") +               \
                                      hipGetErrorString(device_guard__.err))

}  // namespace

namespace {
int rtc_launch(const cppf_robot* rb, RtcKernel which, unsigned grid, size_t lds, hipStream_t st, void** args) {
    hipFunction_t f = rb->rtc->fn[which];
    if (!f) return fail(CPPF_ERR_UNSUPPORTED, "cppflow_hip: this kernel is not part of the handle's specialised module");
    CPPF_HIP(hipModuleLaunchKernel(f, grid, 1, 1, (unsigned)kBlock, 1, 1, (unsigned)lds, st, args, nullptr));
    return CPPF_OK;
}
}  // namespace

extern "C" {

int cppf_abi_version(void) { return CPPF_ABI_VERSION; }

#ifndef CPPF_BUILD_ID
#define CPPF_BUILD_ID "unversioned"
#endif
// (the marker lets cppflow_amd/build.py read the id out of the file without loading the library)
static const char kBuildIdMarker[] = "CPPF_BUILD_ID=" CPPF_BUILD_ID;
const char* cppf_build_id(void) { return kBuildIdMarker + 14; }

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
        // p0 == p1 exactly is a SPHERE (its 1 / |h|^2 is 0, capsule_centred); a segment shorter than a micrometre is a mistake
        CPPF_REQUIRE(len2 == 0.f || len2 > 1e-12f, "degenerate capsule (|p1 - p0| < 1e-6 but not 0; use p0 == p1 for a sphere)");
        CPPF_REQUIRE(desc->cap_r[c] >= 0.f, "negative capsule radius");
    }
    for (int p = 0; p < desc->n_pairs; ++p) {
        const int a = desc->pairs[p][0], b = desc->pairs[p][1];
        CPPF_REQUIRE(a >= 0 && a < desc->n_capsules && b >= 0 && b < desc->n_capsules && a != b, "bad capsule pair");
    }
    if (device != kNoDevice) {  // (kNoDevice: the host-only handle cppf_debug_rtc_compile builds)
        int ndev = 0;
        CPPF_HIP(hipGetDeviceCount(&ndev));
        CPPF_REQUIRE(device >= 0 && device < ndev, "device index out of range");
    }

    cppf_robot* rb = new (std::nothrow) cppf_robot();
    if (!rb) return fail(CPPF_ERR_HIP, "cppflow_hip: out of host memory");
    rb->desc = *desc;
    rb->device = device;
    rb->cu_count = 0;
    if (device != kNoDevice) (void)hipDeviceGetAttribute(&rb->cu_count, hipDeviceAttributeMultiprocessorCount, device);
    rb->life.store(0u, std::memory_order_relaxed);
    for (int k = 0; k < CPPF_TUNE_COUNT; ++k) rb->tune[k].store(kTuneDefaults[k], std::memory_order_relaxed);
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
        const CapsuleCentred cc = capsule_centred(*desc, c);
        for (int k = 0; k < 3; ++k) {
            co.cap_c[c][k] = cc.c[k];
            co.cap_h[c][k] = cc.h[k];
        }
        co.cap_a[c] = cc.a;
        co.cap_ia[c] = cc.ia;
        co.cap_r[c] = desc->cap_r[c];
        co.cap_link[c] = (int8_t)desc->cap_link[c];
        co.cap_thr[c] = sqrt_threshold(desc->cap_r[c]);
        co.cap_cull[c] = cull_threshold(cap_half_length(*desc, c) + (double)desc->cap_r[c]);
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
        co.pair_cull[p] = cull_threshold(cap_half_length(*desc, a) + cap_half_length(*desc, b) + (double)desc->cap_r[a] +
                                         (double)desc->cap_r[b]);
    }
    rb->lds_bytes = (size_t)co.ncaps * 6 * kBlock * sizeof(float);
    rb->static_id = find_static_robot(*desc);
    if (fused_static_block() != kBlock) rb->static_id = -1;  // (the two translation units were built for different workgroup sizes)
    rb->rtc = nullptr;
    rb->d_quad = nullptr;
    // device tables of the quad shape (static per robot: pair list, thresholds)
    if (device != kNoDevice) {
        std::vector<uint4> host(CPPF_MAX_PAIRS + CPPF_MAX_CAPSULES, uint4{0, 0, 0, 0});
        for (int p = 0; p < co.npairs; ++p) {
            QuadPairRec r{co.pair_a[p], co.pair_b[p], co.pair_thr[p], co.pair_cull[p]};
            std::memcpy(&host[p], &r, sizeof r);
        }
        for (int c = 0; c < co.ncaps; ++c) {
            QuadCapRec r{co.cap_thr[c], co.cap_cull[c], co.cap_a[c], co.cap_ia[c]};
            std::memcpy(&host[CPPF_MAX_PAIRS + c], &r, sizeof r);
        }
        DeviceGuard guard(device);
        hipError_t e = guard.err;
        rb->d_quad = nullptr;
        if (e == hipSuccess) e = hipMalloc(&rb->d_quad, host.size() * sizeof(uint4));
        if (e == hipSuccess) e = hipMemcpy(rb->d_quad, host.data(), host.size() * sizeof(uint4), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            if (rb->d_quad) (void)hipFree(rb->d_quad);
            delete rb;
            return fail(CPPF_ERR_HIP, std::string("cppflow_hip: device tables: ") + hipGetErrorString(e));
        }
    }
    *out = rb;
    return CPPF_OK;
}

namespace {
void robot_free(cppf_robot* robot) {
    {
        DeviceGuard guard(robot->device);
        if (robot->d_quad) (void)hipFree(robot->d_quad);
        if (robot->rtc) {
            if (robot->rtc->module) (void)hipModuleUnload(robot->rtc->module);
            delete robot->rtc;
        }
    }
    delete robot;
}
}  // namespace

void cppf_robot_destroy(cppf_robot* robot) {
    if (!robot) return;
    // Live batches keep the handle allocated (their launches then fail with CPPF_ERR_INVALID instead of reading freed memory); the
    // last cppf_lm_batch_destroy frees it.  A second destroy of a handle that is only kept for its batches is ignored.
    const uint32_t before = robot->life.fetch_or(kRobotDead, std::memory_order_acq_rel);
    if (before == 0u) robot_free(robot);
}

int cppf_robot_specialize(cppf_robot* robot, const char* cache_dir) {
    CPPF_REQUIRE(robot != nullptr, "robot handle is NULL");
    if (robot->static_id >= 0 || robot->rtc) return CPPF_OK;
    const bool with_quad = robot->desc.ndof >= 6;  // the quad shape solves the dual 6x6 system
    const std::string source = rtc_program_source(*robot);
    if (const char* dump = std::getenv("CPPF_RTC_DUMP")) {  // developer aid: the generated program, for scripts/rtc_try.py
        std::ofstream f(dump);
        f << source;
    }
    uint64_t key = fnv1a(source);
    for (int i = 0; i < kEmbeddedCount; ++i) key = fnv1a(kEmbeddedSources[i], key);
    for (int i = 0; i < kRtcOptionCount; ++i) key = fnv1a(kRtcOptions[i], key);
    key = fnv1a(rtc_version_string(), key);  // (and the compiler that would produce it)
    char hex[32];
    std::snprintf(hex, sizeof hex, "%016llx", (unsigned long long)key);
    const std::string dir = rtc_cache_dir(cache_dir), file = dir + "/robot_" + hex + ".cppfrtc";
    std::vector<std::string> names;
    std::string code;
    if (!rtc_cache_read(dir, file, names, code)) {
        if (int rc = rtc_compile(source, with_quad, names, code)) return rc;
        rtc_cache_write(dir, file, names, code);
    }
    // (compiling and caching need no device; loading does)
    DeviceGuard device_guard__(robot->device);
    if (device_guard__.err != hipSuccess)
        return fail(CPPF_ERR_HIP, std::string("cppflow_hip: selecting the robot's device failed: ") + hipGetErrorString(device_guard__.err));
    RtcModule* m = new (std::nothrow) RtcModule();
    if (!m) return fail(CPPF_ERR_HIP, "cppflow_hip: out of host memory");
    hipError_t e = hipModuleLoadData(&m->module, code.data());
    for (int i = 0; e == hipSuccess && i < RTC_COUNT; ++i) {
        if (!with_quad && (i == RTC_QUAD0 || i == RTC_QUAD1)) continue;
        e = hipModuleGetFunction(&m->fn[i], m->module, names[(size_t)i].c_str());
    }
    if (e != hipSuccess) {
        if (m->module) (void)hipModuleUnload(m->module);
        delete m;
        return fail(CPPF_ERR_HIP, std::string("cppflow_hip: loading the specialised code object failed: ") + hipGetErrorString(e));
    }
    robot->rtc = m;
    return CPPF_OK;
}

int cppf_robot_ndof(const cppf_robot* robot) { return robot ? robot->desc.ndof : CPPF_ERR_INVALID; }

int cppf_debug_fused_single_offset(void) { return (int)__builtin_offsetof(FusedArgs, single); }

int cppf_debug_rtc_compile(const cppf_robot_desc* desc, const char* cache_dir) {
    cppf_robot* rb = nullptr;
    if (int rc = cppf_robot_create(desc, kNoDevice, &rb)) return rc;
    rb->static_id = -1;  // compile even a shipped description: this hook exercises the run-time path itself
    rb->device = kNoDevice;
    const int rc = cppf_robot_specialize(rb, cache_dir);
    delete rb;
    // without a device the last stage (loading the code object) fails by design; the compile + cache stages have run
    return rc;
}

int cppf_robot_specialization(const cppf_robot* robot) {
    if (!robot) return CPPF_ERR_INVALID;
    return robot->static_id >= 0 ? robot->static_id : (robot->rtc ? CPPF_SPECIALIZATION_RTC : -1);
}

int cppf_debug_set(cppf_robot* robot, int key, int value) {
    CPPF_REQUIRE(robot, "robot handle is NULL");
    CPPF_REQUIRE(key >= 0 && key < CPPF_TUNE_COUNT, "unknown tuning key");
    robot->tune[key].store(value == CPPF_TUNE_DEFAULT ? kTuneDefaults[key] : value, std::memory_order_relaxed);
    return CPPF_OK;
}

int cppf_debug_get(const cppf_robot* robot, int key, int* value) {
    CPPF_REQUIRE(robot && value, "robot / value is NULL");
    CPPF_REQUIRE(key >= 0 && key < CPPF_TUNE_COUNT, "unknown tuning key");
    *value = tune(robot, key);
    return CPPF_OK;
}

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

}  // extern "C"

namespace {

// cppf_lm_params -> the kernels' LmK (validation included); n / W are the caller's to fill
int make_lm_kernel_params(const cppf_robot* robot, const cppf_lm_params* params, LmK& prm) {
    CPPF_REQUIRE(params, "params is NULL");
    CPPF_REQUIRE(params->n_steps >= 1, "n_steps must be >= 1");
    CPPF_REQUIRE(params->clamp == 1 || params->n_steps == 1, "clamp = 0 is only defined for a single step");
    CPPF_REQUIRE(params->lm_lambda > 0.f, "lm_lambda must be > 0");
    CPPF_REQUIRE(params->alpha_position > 0.f && params->alpha_rotation > 0.f, "alpha_position / alpha_rotation must be > 0");
    CPPF_REQUIRE(params->solver == CPPF_SOLVER_AUTO || params->solver == CPPF_SOLVER_F32 || params->solver == CPPF_SOLVER_F64,
                 "unknown solver");
    CPPF_REQUIRE(params->solver_gate >= 0.f, "solver_gate must be >= 0 (0 = the default tolerance)");
    CPPF_REQUIRE(params->shape == CPPF_SHAPE_AUTO || params->shape == CPPF_SHAPE_ROW || params->shape == CPPF_SHAPE_QUAD,
                 "unknown kernel shape");
    CPPF_REQUIRE(params->tol_pos_m >= 0.f && params->tol_rot_rad >= 0.f, "early-out tolerances must be >= 0");
    CPPF_REQUIRE((params->tol_pos_m > 0.f) == (params->tol_rot_rad > 0.f), "set both early-out tolerances or neither");
    prm.lm_lambda = params->lm_lambda;
    prm.a_pos = params->alpha_position;
    prm.a_rot = params->alpha_rotation;
    // the damping per row of the dual system (kernels_chain.h), formed once here instead of by every wavefront
    prm.lam_r_d = (double)params->lm_lambda / ((double)params->alpha_rotation * (double)params->alpha_rotation);
    prm.lam_p_d = (double)params->lm_lambda / ((double)params->alpha_position * (double)params->alpha_position);
    prm.lam_r = (float)prm.lam_r_d;
    prm.lam_p = (float)prm.lam_p_d;
    // conditioning gate (lm_solve_gated): redo a row's solve in double precision when eps * a_max * max diag(A) * max |y| > tau
    const float tau = params->solver_gate > 0.f ? params->solver_gate : CPPF_SOLVER_GATE_DEFAULT;
    const float a_max = std::fmax(params->alpha_position, params->alpha_rotation);
    prm.gate_thr = params->solver == CPPF_SOLVER_F32 ? INFINITY : params->solver == CPPF_SOLVER_F64 ? -INFINITY : tau / (6e-8f * a_max);
    // ... and, in the lean iterations of a fused launch, only when that estimate is also more than kGateRel of the residual the step
    // reduces (kernels_fused.h: an intermediate iterate needs a step that is accurate RELATIVE to its residual)
    const float rel = 1e-6f * (float)tune(robot, CPPF_TUNE_GATE_REL_PPM) / (6e-8f * a_max);  // (kGateRel unless a test changed it)
    prm.gate_rel2 = rel * rel;
    // fair-share pacing (kernels_fused.h: lm_pace), opt-in per handle: CPPF_TUNE_LM_PACE = 0 off (default), > 0 ticks per iteration,
    // < 0 the built-in estimate: 0.55 x the iteration's ~(200 + 56 d) VALU instructions ~ four wavefronts' share of a SIMD that issues
    // one every ~2.6 cycles (+ the longer first / last iterations), in 10 ns ticks (Panda: 326; the measured plateau is 300 .. 360).  launch_fused_rows drops it for launches
    // that do not fill the chip; early-out launches have no schedule.
    const int pace = tune(robot, CPPF_TUNE_LM_PACE);
    prm.pace_ticks = (pace == 0 || params->tol_pos_m > 0.f || params->n_steps < 3) ? 0 : (pace > 0 ? pace : (int)(0.55f * (200.f + 56.f * (float)robot->desc.ndof) + 0.5f));
    prm.n_steps = params->n_steps;
    prm.clamp = params->clamp;
    prm.n = 0;
    prm.W = 0;
    // (squared tolerances; a positive tolerance whose square underflows still means "early-out on": the smallest normal float)
    prm.tol_pos2 = params->tol_pos_m > 0.f ? std::fmax(params->tol_pos_m * params->tol_pos_m, 1.17549435e-38f) : 0.f;
    prm.tol_rot2 = params->tol_rot_rad > 0.f ? std::fmax(params->tol_rot_rad * params->tol_rot_rad, 1.17549435e-38f) : 0.f;
    return CPPF_OK;
}

inline bool outputs_want_collision(const cppf_lm_outputs& o) {
    return o.self_mask || o.env_mask || o.jlim_mask || o.ext_cost || o.min_self || o.min_env || o.seed_summary;
}
// the per-seed summary is an epilogue of the fused launch when a 256-row workgroup holds whole seeds (W = 64, 128, 256)
inline bool summary_in_launch(int W) { return W >= 64 && kBlock % W == 0 && (W & (W - 1)) == 0; }

// One launch of the row-shape fused kernel over `grid` workgroups: the problem `single` (table == NULL) or the problems of a device
// table (cppf_lm_batch_*).  coll = lm_fused_kernel's COLL; n_rows = all rows of the launch (what the residency claim goes by).
int launch_fused_rows(const cppf_robot* robot, int coll, size_t n_rows, unsigned grid, hipStream_t st, const LmK& prm,
                      const BatchItemK& single, const void* table) {
    // Dynamic LDS: what the generic kernels stage their capsules in, or -- for a launch of at most two workgroups per compute
    // unit -- a claim sized so that only two workgroups FIT on one (fused_spread_lds): several such launches in flight then
    // spread over the whole chip instead of stacking four deep on the compute units the dispatcher tries first.
    const bool generic = !use_rtc(robot) && !(robot->static_id >= 0 && !tune(robot, CPPF_TUNE_FORCE_GENERIC));
    const size_t spread = fused_spread_lds(robot, n_rows);
    const size_t lds_need = (generic && coll) ? robot->lds_bytes : 0;
    // the generic kernels stage 6 floats per capsule per lane: with the gate's slots beside them, 12 joints x 24 capsules no longer
    // fit the 160 KB of a compute unit (the launch would take the process down) -- such a robot has to be specialised
    if (generic && coll && lds_need + fused_static_lds(robot) > (size_t)160 * 1024)
        return fail(CPPF_ERR_UNSUPPORTED, "cppflow_hip: the generic kernels cannot stage this many capsules at this ndof; cppf_robot_specialize() the robot");
    FusedArgs fa;
    fa.ch = robot->chain;
    fa.co = robot->coll;
    fa.prm = prm;
    // (the schedule is that of ONE resident round of FOUR wavefronts per SIMD: a launch that leaves a quarter of the chip's slots empty, or
    // needs a second round, or whose kernel holds fewer wavefronts per SIMD -- lm_waves(): specialised chains beyond 7 joints, generic
    // ones beyond 6 -- is not paced: measured on Fetch, 8 joints, three per SIMD: +2 .. 3 %, profiles/r5_pace_sweep.txt)
    {
        const unsigned long long slots = (unsigned long long)(robot->cu_count > 0 ? robot->cu_count : 256) * 4ull;  // workgroups of 256 rows at four wavefronts per SIMD
        const int d = robot->desc.ndof;
        const bool four = (use_rtc(robot) || (robot->static_id >= 0 && !tune(robot, CPPF_TUNE_FORCE_GENERIC))) ? d <= 7 : d <= 6;
        if (!four || (unsigned long long)grid * 4ull < 3ull * slots || (unsigned long long)grid > slots) fa.prm.pace_ticks = 0;
    }
    fa.single = single;
    fa.table = table;
    if (use_rtc(robot)) {
        void* args[] = {(void*)&fa.ch, (void*)&fa.co, (void*)&fa.prm, (void*)&fa.single, (void*)&fa.table};
        if (int rc = rtc_launch(robot, coll == 0 ? RTC_FUSED0 : (coll == 2 ? RTC_FUSED2 : RTC_FUSED1), grid, spread, st, args)) return rc;
    } else if (!generic) {
        // a generated table: the kernel lives in fused_static.hip
        if (!launch_fused_static(robot->static_id, coll, grid, spread, st, fa))
            return fail(CPPF_ERR_UNSUPPORTED, "cppflow_hip: no fused kernel for this generated table");
    } else {
        const size_t lds = std::max(lds_need, spread);
#define CPPF_BODY                                                                                          \
    if (coll == 2)                                                                                         \
        hipLaunchKernelGGL((lm_fused_kernel<RB, 2>), dim3(grid), dim3(kBlock), lds, st, fa.ch, fa.co, fa.prm, fa.single, fa.table);               \
    else if (coll == 1)                                                                                    \
        hipLaunchKernelGGL((lm_fused_kernel<RB, 1>), dim3(grid), dim3(kBlock), lds, st, fa.ch, fa.co, fa.prm, fa.single, fa.table);               \
    else                                                                                                   \
        hipLaunchKernelGGL((lm_fused_kernel<RB, 0>), dim3(grid), dim3(kBlock), lds, st, fa.ch, fa.co, fa.prm, fa.single, fa.table)
        CPPF_DISPATCH_DYN(robot)
#undef CPPF_BODY
    }
    return check_launch(robot);
}

}  // namespace

// A batched fused launch (cppflow_hip.h): the item descriptors live in a device table the batch object owns.
struct cppf_lm_batch {
    const cppf_robot* robot;  // counted in robot->life from create to destroy: never dangling
    int device;               // the robot's, copied: destroy does not need the robot for anything but the count
    LmK prm;
    void* d_table;        // { BatchHeadK ; BatchItemK[n_items] }
    int n_items, coll;
    unsigned grid;
    size_t n_rows;
    std::vector<cppf_lm_batch_item> after;  // items whose per-seed summary needs the separate reduction launch (W not 64 / 128 / 256)
};

extern "C" {

int cppf_lm_pose_steps(const cppf_robot* robot, const float* x_in, const float* target, int S, int W,
                       const cppf_lm_params* params, const cppf_lm_outputs* out, void* stream) {
    CPPF_ENTER(robot);
    CPPF_REQUIRE(params && out, "params / out is NULL");
    CPPF_REQUIRE(S >= 0 && W >= 0, "S / W < 0");
    LmK prm;
    if (int rc = make_lm_kernel_params(robot, params, prm)) return rc;
    const size_t n = (size_t)S * W;
    if (n == 0) return CPPF_OK;
    CPPF_REQUIRE(n <= 0x7fffffffu, "S*W exceeds 2^31-1 rows");
    CPPF_REQUIRE(x_in && target, "x_in / target is NULL");
    prm.n = (int)n;
    prm.W = W;
    CPPF_REQUIRE(!(params->tol_pos_m > 0.f && (out->J_out || out->e_out)), "early-out is not combinable with J_out / e_out");
    const bool coll = outputs_want_collision(*out);
    hipStream_t st = (hipStream_t)stream;
    // per-seed summary: fused into the launch when a workgroup holds whole seeds, else the separate reduction afterwards
    cppf_lm_outputs outk = *out;
    float* const summary_dst = out->seed_summary;
    const bool summary_after = out->seed_summary && !summary_in_launch(W);
    if (summary_after) {
        CPPF_REQUIRE(out->x_out && out->pos_err_m && out->rot_err_rad && out->self_mask && out->env_mask &&
                         out->jlim_mask && out->ext_cost,
                     "seed_summary with W not in {64, 128, 256} needs x_out, pos_err_m, rot_err_rad, the three masks and ext_cost");
        outk.seed_summary = nullptr;
    }
    // kernel shape: four lanes per row for batches that cannot fill the chip (kernels_quad.h), one row per lane otherwise
    const int d = robot->desc.ndof;
    const bool quad_can = d >= 6 && !out->J_out && !out->e_out && !out->min_self && !out->min_env &&
                          (!out->seed_summary || (out->x_out && out->pos_err_m && out->rot_err_rad && out->self_mask &&
                                                  out->env_mask && out->jlim_mask && out->ext_cost));
    CPPF_REQUIRE(params->shape != CPPF_SHAPE_QUAD || quad_can,
                 "CPPF_SHAPE_QUAD needs ndof >= 6, no J_out / e_out / min_self / min_env, and (with seed_summary) every per-row output");
    // AUTO: the quad shape when the batch is at most one of its wavefronts per SIMD AND no per-seed summary is asked for (in
    // this shape a seed's rows span several workgroups, so the summary needs the separate reduction launch, which costs more
    // than the shape saves: measured 28.5 against 26.2 us for K = 10 + collision + summary at <= 16 384 rows)
    const bool quad = params->shape == CPPF_SHAPE_QUAD ||
                      (params->shape == CPPF_SHAPE_AUTO && quad_can && !out->seed_summary && n <= (size_t)tune(robot, CPPF_TUNE_QUAD_MAX_ROWS));
    if (quad) {
        cppf_lm_outputs oq = *out;
        oq.seed_summary = nullptr;  // rows of a seed span several workgroups in this shape: the reduction kernel follows
        const unsigned grid = (unsigned)((n + kQuadRows - 1) / kQuadRows);
        const size_t lds_q = coll ? sizeof(float) * (4 * (CPPF_MAX_PAIRS + CPPF_MAX_CAPSULES) +
                                                     (size_t)kQuadRows * quad_row_stride(robot->coll.ncaps))
                                  : 0;
        const uint4* tab = static_cast<const uint4*>(robot->d_quad);
        const bool mfma = tune(robot, CPPF_TUNE_QUAD_MFMA) && robot->static_id >= 0 && !tune(robot, CPPF_TUNE_FORCE_GENERIC);
        if (use_rtc(robot) && robot->rtc->fn[RTC_QUAD0]) {
            void* args[] = {(void*)&robot->chain, (void*)&robot->coll, (void*)&prm, (void*)&x_in, (void*)&target, (void*)&oq, (void*)&tab};
            if (int rc = rtc_launch(robot, coll ? RTC_QUAD1 : RTC_QUAD0, grid, lds_q, st, args)) return rc;
        } else {
#define CPPF_BODY                                                                                                          \
    if constexpr (RB::D >= 6) {                                                                                            \
        if (coll) {                                                                                                        \
            if constexpr (RB::kStatic) {                                                                                   \
                if (mfma)                                                                                                  \
                    hipLaunchKernelGGL((lm_quad_kernel<RB, 1, true>), dim3(grid), dim3(kBlock), lds_q, st, robot->chain,    \
                                       robot->coll, prm, x_in, target, oq, tab);                                           \
                else                                                                                                       \
                    hipLaunchKernelGGL((lm_quad_kernel<RB, 1, false>), dim3(grid), dim3(kBlock), lds_q, st, robot->chain,   \
                                       robot->coll, prm, x_in, target, oq, tab);                                           \
            } else {                                                                                                       \
                hipLaunchKernelGGL((lm_quad_kernel<RB, 1, false>), dim3(grid), dim3(kBlock), lds_q, st, robot->chain,       \
                                   robot->coll, prm, x_in, target, oq, tab);                                               \
            }                                                                                                              \
        } else {                                                                                                           \
            if constexpr (RB::kStatic) {                                                                                   \
                if (mfma)                                                                                                  \
                    hipLaunchKernelGGL((lm_quad_kernel<RB, 0, true>), dim3(grid), dim3(kBlock), 0, st, robot->chain,        \
                                       robot->coll, prm, x_in, target, oq, tab);                                           \
                else                                                                                                       \
                    hipLaunchKernelGGL((lm_quad_kernel<RB, 0, false>), dim3(grid), dim3(kBlock), 0, st, robot->chain,       \
                                       robot->coll, prm, x_in, target, oq, tab);                                           \
            } else {                                                                                                       \
                hipLaunchKernelGGL((lm_quad_kernel<RB, 0, false>), dim3(grid), dim3(kBlock), 0, st, robot->chain,           \
                                   robot->coll, prm, x_in, target, oq, tab);                                               \
            }                                                                                                              \
        }                                                                                                                  \
    }
        CPPF_DISPATCH_RB(robot)
#undef CPPF_BODY
        }
        if (int rc = check_launch(robot)) return rc;
        if (summary_dst)
            return cppf_seed_summary(robot, oq.x_out, S, W, oq.ext_cost, oq.pos_err_m, oq.rot_err_rad, oq.self_mask, oq.env_mask,
                                     oq.jlim_mask, summary_dst, stream);
        return CPPF_OK;
    }
    // (the kernel gets `outk`: with the separate reduction its seed_summary is NULL -- handing it `*out` there made the in-launch
    // summary run on a W it is not built for and write up to n / 64 - S rows past the caller's [S,8] buffer before the reduction
    // kernel put the right values in: found by the sentinel-arena test of tests/test_gpu_round3.py)
    BatchItemK item;
    item.x_in = x_in, item.target = target, item.out = outk, item.n = (int)n, item.W = W;
    const int c = !coll ? 0 : ((out->min_self || out->min_env) ? 2 : 1);
    if (int rc = launch_fused_rows(robot, c, n, grid_for(n), st, prm, item, nullptr)) return rc;
    if (summary_after)
        return cppf_seed_summary(robot, outk.x_out, S, W, outk.ext_cost, outk.pos_err_m, outk.rot_err_rad, outk.self_mask,
                                 outk.env_mask, outk.jlim_mask, summary_dst, stream);
    return CPPF_OK;
}

int cppf_lm_batch_create(const cppf_robot* robot, int n_items, const cppf_lm_batch_item* items, const cppf_lm_params* params,
                         cppf_lm_batch** out) {
    CPPF_ENTER(robot);
    CPPF_REQUIRE(out, "out is NULL");
    *out = nullptr;
    CPPF_REQUIRE(items && n_items >= 1 && n_items <= CPPF_MAX_BATCH, "n_items must be in 1 .. CPPF_MAX_BATCH");
    LmK prm;
    if (int rc = make_lm_kernel_params(robot, params, prm)) return rc;
    CPPF_REQUIRE(params->shape != CPPF_SHAPE_QUAD, "a batched launch is the row shape (CPPF_SHAPE_AUTO or CPPF_SHAPE_ROW)");
    std::vector<unsigned char> host(sizeof(BatchHeadK) + (size_t)n_items * sizeof(BatchItemK));
    BatchHeadK* head = reinterpret_cast<BatchHeadK*>(host.data());
    BatchItemK* tab = reinterpret_cast<BatchItemK*>(host.data() + sizeof(BatchHeadK));
    std::vector<cppf_lm_batch_item> after;
    size_t blocks = 0, rows = 0;
    bool coll = false;
    for (int i = 0; i < n_items; ++i) {
        const cppf_lm_batch_item& it = items[i];
        CPPF_REQUIRE(it.S >= 1 && it.W >= 1, "every item needs S >= 1 and W >= 1");
        const size_t n = (size_t)it.S * it.W;
        CPPF_REQUIRE(n <= 0x7fffffffu, "S*W exceeds 2^31-1 rows");
        CPPF_REQUIRE(it.x_in && it.target, "x_in / target is NULL");
        CPPF_REQUIRE(!it.out.J_out && !it.out.e_out && !it.out.min_self && !it.out.min_env,
                     "J_out / e_out / min_self / min_env are not available in a batched launch");
        BatchItemK k;
        k.x_in = it.x_in, k.target = it.target, k.out = it.out, k.n = (int)n, k.W = it.W;
        if (it.out.seed_summary && !summary_in_launch(it.W)) {
            CPPF_REQUIRE(it.out.x_out && it.out.pos_err_m && it.out.rot_err_rad && it.out.self_mask && it.out.env_mask &&
                             it.out.jlim_mask && it.out.ext_cost,
                         "seed_summary with W not in {64, 128, 256} needs x_out, pos_err_m, rot_err_rad, the three masks and ext_cost");
            k.out.seed_summary = nullptr;
            after.push_back(it);
        }
        coll = coll || outputs_want_collision(it.out);
        tab[i] = k;
        blocks += grid_for(n);
        rows += n;
        head->block_end[i] = (uint32_t)blocks;
    }
    CPPF_REQUIRE(blocks <= 0x7fffffffu, "the batch exceeds 2^31-1 workgroups");
    for (int i = n_items; i < CPPF_MAX_BATCH; ++i) head->block_end[i] = 0xffffffffu;
    cppf_lm_batch* b = new (std::nothrow) cppf_lm_batch();
    if (!b) return fail(CPPF_ERR_HIP, "cppflow_hip: out of host memory");
    b->robot = robot, b->device = robot->device, b->prm = prm, b->d_table = nullptr, b->n_items = n_items, b->coll = coll ? 1 : 0;
    b->grid = (unsigned)blocks, b->n_rows = rows;
    b->after = after;
    hipError_t e = hipMalloc(&b->d_table, host.size());
    if (e == hipSuccess) e = hipMemcpy(b->d_table, host.data(), host.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (b->d_table) (void)hipFree(b->d_table);
        delete b;
        return fail(CPPF_ERR_HIP, std::string("cppflow_hip: the batch's device table: ") + hipGetErrorString(e));
    }
    robot->life.fetch_add(1u, std::memory_order_acq_rel);  // (CPPF_ENTER saw the handle alive; the caller may not destroy it DURING this call)
    *out = b;
    return CPPF_OK;
}

int cppf_lm_batch_launch(const cppf_lm_batch* batch, void* stream) {
    CPPF_REQUIRE(batch, "batch is NULL");
    const cppf_robot* robot = batch->robot;
    CPPF_ENTER(robot);
    BatchItemK none;
    std::memset(&none, 0, sizeof none);
    if (int rc = launch_fused_rows(robot, batch->coll, batch->n_rows, batch->grid, (hipStream_t)stream, batch->prm, none, batch->d_table))
        return rc;
    for (const cppf_lm_batch_item& it : batch->after)
        if (int rc = cppf_seed_summary(robot, it.out.x_out, it.S, it.W, it.out.ext_cost, it.out.pos_err_m, it.out.rot_err_rad,
                                       it.out.self_mask, it.out.env_mask, it.out.jlim_mask, it.out.seed_summary, stream))
            return rc;
    return CPPF_OK;
}

void cppf_lm_batch_destroy(cppf_lm_batch* batch) {
    if (!batch) return;
    {
        DeviceGuard guard(batch->device);
        if (batch->d_table) (void)hipFree(batch->d_table);
    }
    cppf_robot* robot = const_cast<cppf_robot*>(batch->robot);
    delete batch;
    // the last batch of a handle that was destroyed meanwhile frees it
    if (robot->life.fetch_sub(1u, std::memory_order_acq_rel) == (kRobotDead | 1u)) robot_free(robot);
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
    const size_t lds = (robot->static_id >= 0 && !tune(robot, CPPF_TUNE_FORCE_GENERIC)) ? 0 : robot->lds_bytes;
    if (use_rtc(robot)) {
        int n_i = (int)n;
        void* args[] = {(void*)&robot->chain, (void*)&robot->coll, (void*)&n_i, (void*)&q, (void*)&self_mask, (void*)&env_mask,
                        (void*)&jlim_mask, (void*)&ext_cost, (void*)&min_self, (void*)&min_env};
        return rtc_launch(robot, (min_self || min_env) ? RTC_COLL_MIN : RTC_COLL_MASK, grid_for(n), 0, st, args);
    }
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

int cppf_select_valid_seed_gathered(const cppf_robot* robot, const float* gathered, int n_chunks, int n_groups, int S_chunk,
                                    const cppf_constraints* constraints, int32_t* out, void* stream) {
    CPPF_ENTER(robot);
    CPPF_REQUIRE(n_chunks >= 1 && n_groups >= 1 && S_chunk >= 0, "n_chunks, n_groups must be >= 1 and S_chunk >= 0");
    CPPF_REQUIRE((size_t)n_chunks * S_chunk <= 0x7fffffffu, "n_chunks * S_chunk exceeds 2^31-1 seeds");
    CPPF_REQUIRE(constraints && out && (gathered || S_chunk == 0), "NULL pointer");
    CPPF_REQUIRE(((uintptr_t)gathered & 15u) == 0, "the summaries must be 16-byte aligned");
    SelectK k;
    k.thr[0] = constraints->max_allowed_position_error_cm;
    k.thr[1] = constraints->max_allowed_rotation_error_deg;
    k.thr[2] = constraints->max_allowed_mjac_deg;
    k.thr[3] = constraints->max_allowed_mjac_cm;
    k.ignore_self = constraints->self_collisions_ignored;
    k.ignore_env = constraints->env_collisions_ignored;
    k.S_chunk = S_chunk, k.n_chunks = n_chunks, k.n_groups = n_groups;
    hipLaunchKernelGGL(select_valid_seed_kernel, dim3((unsigned)n_groups), dim3(256), 0, (hipStream_t)stream, gathered, k, out);
    return check_launch(robot);
}

int cppf_select_valid_seed(const cppf_robot* robot, const float* seed_summary, int S, const cppf_constraints* constraints,
                           int32_t* out, void* stream) {
    return cppf_select_valid_seed_gathered(robot, seed_summary, 1, 1, S, constraints, out, stream);
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
    // the "satisfied" row options (cppflow/optimization_utils.py:514-533, 548-606)
    CPPF_REQUIRE(params->differencing_mode >= 0 && params->differencing_mode <= 2, "differencing_mode must be 0, 1 or 2");
    CPPF_REQUIRE(!params->pose_do_scale_down_satisfied || (params->pose_scale_down >= 0.f && params->pose_scale_down < 1.f),
                 "pose scale-down must be in [0, 1) (optimization_utils.py:305)");
    CPPF_REQUIRE(params->differencing_mode != 2 || (params->differencing_scale_down >= 0.f && params->differencing_scale_down < 1.f),
                 "differencing scale-down must be in [0, 1) (optimization_utils.py:367)");
    prm.pose_scale_satisfied = params->use_pose && params->pose_do_scale_down_satisfied;
    prm.pose_thr_m = params->pose_threshold_m;
    prm.pose_thr_rad = params->pose_threshold_rad;
    prm.pose_scale = params->pose_scale_down;
    prm.diff_mode = params->use_differencing ? params->differencing_mode : 0;
    prm.diff_thr_rad = params->differencing_threshold_rad;
    prm.diff_thr_m = params->differencing_threshold_m;
    prm.diff_scale = params->differencing_scale_down;
    prm.diff_shift_invalid = params->differencing_shift_invalid_to_threshold;
    // Individually weighted differencing rows: the coupling between waypoints t and t + 1 is no longer the same constant for
    // every t, which the tuned elimination kernels assume; such a step goes through the one-lane-per-trajectory kernel with the
    // couplings read from memory (w2next: the spare tail of work_G -- that kernel keeps d(d+1)/2 of the d*d floats per row).
    const bool var_coupling = prm.diff_mode != 0;
    float* const w2next = work_G + n * (size_t)(robot->desc.ndof * (robot->desc.ndof + 1) / 2);
    // which elimination kernel: see the comments at the launches below
    const int t_pcr_rows = tune(robot, CPPF_TUNE_PCR_MAX_ROWS), t_pcr_lds = tune(robot, CPPF_TUNE_PCR_LDS);
    const bool g_pcr_lds = t_pcr_lds != 0, g_pcr_split = t_pcr_lds != 1;
    const bool g_rows_pose = tune(robot, CPPF_TUNE_ROWS_POSE) != 0, g_full_rows = tune(robot, CPPF_TUNE_FULL_ROWS) != 0;
    const size_t pcr_rows = t_pcr_rows >= 0 ? (size_t)t_pcr_rows : (size_t)((W <= 256 && g_pcr_lds) ? kPcrMaxRowsLds : kPcrMaxRowsGlobal);
    const size_t pcr_limit = pcr_rows * (robot->desc.ndof <= 7 ? 100 : 50) / 100;
    const bool use_pcr = !var_coupling && !prm.use_pose && W <= 512 && n <= pcr_limit && robot->desc.ndof >= 3 && robot->desc.ndof <= 8;
    const bool use_rows = !var_coupling && !use_pcr && (!prm.use_pose || g_rows_pose) && g_full_rows && robot->desc.ndof >= 3 &&
                          robot->desc.ndof <= 12 && W <= (1 << 19);
    prm.fold = use_rows || var_coupling;
    hipStream_t st = (hipStream_t)stream;
#define CPPF_BODY                                                                                                     \
    if (n >= 131072)                                                                                                  \
        hipLaunchKernelGGL((full_blocks_kernel<RB, full_blocks_occ<RB>()>), dim3(grid_for(n)), dim3(kBlock), robot->lds_bytes, st, \
                           robot->chain, robot->coll, prm, x_in, target, virtual_configs, work_blocks, w2next);      \
    else                                                                                                              \
        hipLaunchKernelGGL((full_blocks_kernel<RB>), dim3(grid_for(n)), dim3(kBlock), robot->lds_bytes, st, robot->chain, \
                           robot->coll, prm, x_in, target, virtual_configs, work_blocks, w2next)
    CPPF_DISPATCH_RB(robot)
#undef CPPF_BODY
    // Trajectories are eliminated one per wavefront (8 x 8 lane tile) up to 8 joints, one per lane beyond.  With the pose
    // block the d x d blocks are J^T J + a small diagonal (rank 6 of 7, cond ~1e7): the per-lane kernel's Cholesky with
    // floored pivots copes with that better than the explicit Gauss-Jordan inverse, so it keeps that case.
    // Up to ~128k rows (the planner's cadence is one trajectory): parallel cyclic reduction, one workgroup per trajectory, one
    // lane per waypoint; beyond that its O(T log T) work and traffic lose against the waypoint-after-waypoint kernels
    if (var_coupling) {
        CPPF_DISPATCH_D(robot->desc.ndof,
                        hipLaunchKernelGGL((full_solve_kernel<D, true>), dim3((unsigned)((S + 63) / 64)), dim3(64), 0, st,
                                           robot->chain, prm, x_in, virtual_configs, work_blocks, work_G, work_y, x_out, w2next));
        return check_launch(robot);
    }
    if (use_pcr) {
        switch (robot->desc.ndof) {
#define CPPF_PCR_CASE(DD)                                                                                              \
    case DD:                                                                                                           \
        if (W <= 256 && g_pcr_lds) { /* the state in LDS: 256 x ((d(d+1)/2 + d + d^2) | 1) floats */                     \
            constexpr size_t kState = 256 * (size_t)((DD * (DD + 1) / 2 + DD + DD * DD) | 1) * sizeof(float);           \
            CPPF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&full_solve_pcr_kernel<DD, 256, true>),         \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)kState));                    \
            if (g_pcr_split && (DD <= kPcrSplitMaxD || t_pcr_lds == 3)) { /* see kPcrSplitMaxD; 3 forces the split form */ \
                CPPF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&full_solve_pcr_kernel<DD, 512, true, true>), \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)kState));                \
                hipLaunchKernelGGL((full_solve_pcr_kernel<DD, 512, true, true>), dim3((unsigned)S), dim3(512), kState, st, \
                                   robot->chain, prm, x_in, virtual_configs, work_blocks, work_G, x_out);              \
            } else                                                                                                     \
            hipLaunchKernelGGL((full_solve_pcr_kernel<DD, 256, true>), dim3((unsigned)S), dim3(256), kState, st,        \
                               robot->chain, prm, x_in, virtual_configs, work_blocks, work_G, x_out);                  \
        } else if (W <= 256)                                                                                           \
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
    // row-per-lane kernels: 8 trajectories per one-wavefront workgroup and two workgroups (the two ends of the path) per 8
    // trajectories; the LDS reservation caps the workgroups per compute unit at ceil(#workgroups / 256) (160 KB per compute
    // unit), which spreads a small launch over distinct compute units
    const int rows_tpw = robot->desc.ndof <= 8 ? 8 : 4;  // trajectories per wavefront: 8 or 16 lanes each
    const unsigned rows_wgs = 2u * (unsigned)((S + rows_tpw - 1) / rows_tpw);
    const unsigned rows_per_cu = (rows_wgs + 255) / 256;
    const size_t rows_lds = rows_per_cu == 1 ? 96 * 1024 : rows_per_cu == 2 ? 64 * 1024 : rows_per_cu == 3 ? 48 * 1024 : 0;
    switch ((prm.use_pose && !use_rows) ? 0 : robot->desc.ndof) {
#define CPPF_WAVE_CASE(DD)                                                                                              \
    case DD:                                                                                                            \
        if (use_rows) {                                                                                                 \
            if (rows_lds > 64 * 1024) { /* per device: set whenever it is needed, a host-side table update */           \
                CPPF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&full_rows_eliminate_kernel<DD>),            \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));                   \
                CPPF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&full_rows_substitute_kernel<DD>),           \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));                   \
            }                                                                                                           \
            hipLaunchKernelGGL((full_rows_eliminate_kernel<DD>), dim3(rows_wgs), dim3(64), rows_lds, st, prm,           \
                               robot->chain.pris_mask, work_blocks, work_G, work_y);                                    \
            hipLaunchKernelGGL((full_rows_substitute_kernel<DD>), dim3(rows_wgs), dim3(64), rows_lds, st, prm,          \
                               robot->chain.pris_mask, x_in, work_blocks, work_G, work_y, x_out);                       \
        } else                                                                                                          \
            hipLaunchKernelGGL((full_solve_wave_kernel<DD>), dim3((unsigned)S), dim3(64), 0, st, robot->chain, prm,      \
                               x_in, virtual_configs, work_blocks, work_G, work_y, x_out);                              \
        break;
        CPPF_WAVE_CASE(3) CPPF_WAVE_CASE(4) CPPF_WAVE_CASE(5) CPPF_WAVE_CASE(6) CPPF_WAVE_CASE(7) CPPF_WAVE_CASE(8)
#undef CPPF_WAVE_CASE
#define CPPF_ROWS16_CASE(DD) /* 9 .. 12 joints: sixteen lanes per trajectory (no one-wavefront-per-trajectory form) */          \
    case DD:                                                                                                            \
        if (use_rows) {                                                                                                 \
            if (rows_lds > 64 * 1024) {                                                                                 \
                CPPF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&full_rows_eliminate_kernel<DD>),            \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));                   \
                CPPF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&full_rows_substitute_kernel<DD>),           \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));                   \
            }                                                                                                           \
            hipLaunchKernelGGL((full_rows_eliminate_kernel<DD>), dim3(rows_wgs), dim3(64), rows_lds, st, prm,           \
                               robot->chain.pris_mask, work_blocks, work_G, work_y);                                    \
            hipLaunchKernelGGL((full_rows_substitute_kernel<DD>), dim3(rows_wgs), dim3(64), rows_lds, st, prm,          \
                               robot->chain.pris_mask, x_in, work_blocks, work_G, work_y, x_out);                       \
        } else                                                                                                          \
            hipLaunchKernelGGL((full_solve_kernel<DD>), dim3((unsigned)((S + 63) / 64)), dim3(64), 0, st, robot->chain, \
                               prm, x_in, virtual_configs, work_blocks, work_G, work_y, x_out, nullptr);                \
        break;
        CPPF_ROWS16_CASE(9) CPPF_ROWS16_CASE(10) CPPF_ROWS16_CASE(11) CPPF_ROWS16_CASE(12)
#undef CPPF_ROWS16_CASE
        default:
            CPPF_DISPATCH_D(robot->desc.ndof,
                            hipLaunchKernelGGL((full_solve_kernel<D>), dim3((unsigned)((S + 63) / 64)), dim3(64), 0, st,
                                               robot->chain, prm, x_in, virtual_configs, work_blocks, work_G, work_y,
                                               x_out, nullptr));
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

constexpr int kDpResidentMaxK = 1024;  // four destinations per workgroup x 256 compute units (dp_resident_kernel)

// dynamic LDS of dp_backtrace_kernel: the memo table as bytes when it fits (k <= 256, T k <= 60 KB), else 0 = walk it in global memory
static size_t dp_stage_bytes(int k, int T) { return (k <= 256 && (size_t)k * T <= 60 * 1024) ? (size_t)k * T : 0; }

int cppf_dp_search(const cppf_robot* robot, const float* q, const float* ext_cost, int k, int T, float prismatic_scaling,
                   float* work_qT, float* work_costsT, int32_t* work_memoT, float* best_path, int32_t* best_idx, int mode,
                   void* stream) {
    CPPF_ENTER(robot);
    CPPF_REQUIRE(k >= 1 && T >= 1, "k, T must be >= 1");
    CPPF_REQUIRE(q && ext_cost && work_qT && work_costsT && work_memoT && best_path && best_idx, "NULL pointer");
    CPPF_REQUIRE((size_t)k * T * robot->desc.ndof <= 0x7fffffffu, "k*T*d exceeds 2^31-1");
    CPPF_REQUIRE(mode == CPPF_DP_AUTO || mode == CPPF_DP_RESIDENT || mode == CPPF_DP_LAUNCHES, "unknown dp_search mode");
    CPPF_REQUIRE(mode != CPPF_DP_RESIDENT || k <= kDpResidentMaxK, "CPPF_DP_RESIDENT: the resident launch holds at most 1024 candidates");
    hipStream_t st = (hipStream_t)stream;
    const int d = robot->desc.ndof;
    const size_t total = (size_t)k * T * d;
    bool persistent = T >= 2 && k <= kDpResidentMaxK &&
                      (mode == CPPF_DP_RESIDENT || (mode == CPPF_DP_AUTO && tune(robot, CPPF_TUNE_DP_PERSISTENT)));
    DpResidentForm form{nullptr, 0, 0};
    if (persistent) {
        // The resident form is only CORRECT while all its workgroups are on the device together (they wait for each other's cost words).
        // Hold the grid against what this device can hold of this kernel -- occupancy x compute units -- instead of finding out by
        // spinning 2^22 reads: a partitioned / CU-masked / smaller device takes the per-waypoint launches straight away (AUTO), and a
        // forced CPPF_DP_RESIDENT is refused.  (Other work on the device -- a second resident search on another stream, a full-width
        // fused launch -- can still delay workgroups; the bounded waits stay as the safety net for that.)
        CPPF_DISPATCH_D(d, form = dp_resident_form<D>(k, tune(robot, CPPF_TUNE_DP_PERSISTENT)));
        int per_cu = 0;
        CPPF_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, form.fn, form.block, 0));
        const int cus = tune(robot, CPPF_TUNE_CU_COUNT) >= 0 ? tune(robot, CPPF_TUNE_CU_COUNT) : robot->cu_count;  // (the test hook: a smaller device)
        if ((unsigned long long)per_cu * (unsigned long long)cus < form.grid) {
            if (mode == CPPF_DP_RESIDENT)
                return fail(CPPF_ERR_UNSUPPORTED, "cppflow_hip: CPPF_DP_RESIDENT: this device cannot hold the resident launch's workgroups "
                                                  "together (occupancy x compute units < grid): use CPPF_DP_AUTO or CPPF_DP_LAUNCHES");
            persistent = false;
        }
    }
    if (persistent)  // every cost word starts as "not yet" (kernels_dp.h); 16-byte multiple, from the allocation's start
        CPPF_HIP(hipMemsetAsync(work_costsT, 0xFF, sizeof(float) * (size_t)k * T, st));
    hipLaunchKernelGGL(dp_transpose_kernel, dim3(grid_for(total > (size_t)k ? total : (size_t)k)), dim3(256), 0, st, q,
                       ext_cost, k, T, d, work_qT, work_costsT);
    // memo[:,0] is never read by the back-trace's result but is read as a value: define it (search.py:154 zero-inits memo)
    CPPF_HIP(hipMemsetAsync(work_memoT, 0, sizeof(int32_t) * (size_t)k, st));
    if (persistent) {
        const int spin_log2 = tune(robot, CPPF_TUNE_DP_SPIN_LOG2);
        uint32_t spin = spin_log2 <= 0 ? 0u : (1u << (spin_log2 > 30 ? 30 : spin_log2));
        uint32_t pris_mask = robot->chain.pris_mask;
        // (the six resident kernels share one argument list)
        void* args[] = {(void*)&work_qT, (void*)&ext_cost, (void*)&k, (void*)&T, (void*)&pris_mask, (void*)&prismatic_scaling,
                        (void*)&work_costsT, (void*)&work_memoT, (void*)&spin};
        CPPF_HIP(hipLaunchKernel(form.fn, dim3(form.grid), dim3((unsigned)form.block), args, 0, st));
        hipLaunchKernelGGL(dp_backtrace_kernel, dim3(1), dim3(256), dp_stage_bytes(k, T), st, q, work_costsT, work_memoT, k, T, d,
                       dp_stage_bytes(k, T) != 0, best_idx,
                           best_path);
        return check_launch(robot);
    }
    const int bpb = k >= 4096 ? 4 : (k >= 2048 ? 2 : 1);
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
    hipLaunchKernelGGL(dp_backtrace_kernel, dim3(1), dim3(256), dp_stage_bytes(k, T), st, q, work_costsT, work_memoT, k, T, d,
                       dp_stage_bytes(k, T) != 0, best_idx,
                       best_path);
    return check_launch(robot);
}

int cppf_dp_table_floats(int k, int T, size_t* n_floats) {
    if (!n_floats || k < 1 || T < 1) return fail(CPPF_ERR_INVALID, "cppf_dp_table_floats: k, T must be >= 1, n_floats non-NULL");
    const size_t kp = (size_t)((k + 63) / 64 * 64);
    *n_floats = (size_t)(T > 1 ? T - 1 : 0) * (size_t)k * kp + 32 * kp;  // + 32 rows the chain may read past the last slab
    return CPPF_OK;
}

int cppf_dp_search_tabled(const cppf_robot* robot, const float* q, const float* ext_cost, int k, int T, float prismatic_scaling,
                          float* work_qT, float* work_costsT, int32_t* work_memoT, float* work_table, float* best_path,
                          int32_t* best_idx, void* stream) {
    CPPF_ENTER(robot);
    CPPF_REQUIRE(k >= 1 && k <= 256 && T >= 1, "cppf_dp_search_tabled: 1 <= k <= 256, T >= 1 (cppf_dp_search has no limit on k)");
    CPPF_REQUIRE(q && ext_cost && work_qT && work_costsT && work_memoT && best_path && best_idx, "NULL pointer");
    CPPF_REQUIRE(T == 1 || work_table, "work_table is NULL");
    hipStream_t st = (hipStream_t)stream;
    const int d = robot->desc.ndof, kp = (k + 63) / 64 * 64;
    const size_t total = (size_t)k * T * d;
    hipLaunchKernelGGL(dp_transpose_kernel, dim3(grid_for(total > (size_t)k ? total : (size_t)k)), dim3(256), 0, st, q,
                       ext_cost, k, T, d, work_qT, work_costsT);
    // memo[:,0] is never read by the back-trace's result but is read as a value: define it (search.py:154 zero-inits memo)
    CPPF_HIP(hipMemsetAsync(work_memoT, 0, sizeof(int32_t) * (size_t)k, st));
    if (T >= 2) {
        CPPF_REQUIRE(T - 1 <= 65535, "cppf_dp_search_tabled: T <= 65536");
        CPPF_DISPATCH_D(d, hipLaunchKernelGGL((dp_table_kernel<D>), dim3((unsigned)((kp + 255) / 256), (unsigned)((k + 7) / 8), (unsigned)(T - 1)),
                                             dim3(256), 0, st, work_qT, k, kp, T, robot->chain.pris_mask, prismatic_scaling,
                                             reinterpret_cast<uint32_t*>(work_table)));
        const uint32_t* tab = reinterpret_cast<const uint32_t*>(work_table);
        switch (kp) {
            case 64: hipLaunchKernelGGL(dp_chain_kernel<64>, dim3(1), dim3(512), 0, st, tab, ext_cost, k, T, work_costsT); break;
            case 128: hipLaunchKernelGGL(dp_chain_kernel<128>, dim3(1), dim3(512), 0, st, tab, ext_cost, k, T, work_costsT); break;
            case 192: hipLaunchKernelGGL(dp_chain_kernel<192>, dim3(1), dim3(512), 0, st, tab, ext_cost, k, T, work_costsT); break;
            default: hipLaunchKernelGGL(dp_chain_kernel<256>, dim3(1), dim3(512), 0, st, tab, ext_cost, k, T, work_costsT); break;
        }
        hipLaunchKernelGGL(dp_memo_kernel, dim3((unsigned)((k + 63) / 64), (unsigned)(T - 1)), dim3(256), 0, st, tab, ext_cost,
                           work_costsT, k, kp, T, work_memoT);
    }
    hipLaunchKernelGGL(dp_backtrace_kernel, dim3(1), dim3(256), dp_stage_bytes(k, T), st, q, work_costsT, work_memoT, k, T, d,
                       dp_stage_bytes(k, T) != 0, best_idx,
                       best_path);
    return check_launch(robot);
}

// ---- RCCL behind the C ABI ------------------------------------------------------------------------------------------------------
// Declared here instead of including <rccl/rccl.h>: the library is loaded with dlopen so that libcppflow_hip.so has no link-time
// dependency on it (a process that already holds PyTorch's copy must not get a second one).
namespace {
struct RcclUid {
    char bytes[CPPF_COMM_ID_BYTES];
};
typedef void* RcclComm;
struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(RcclUid*) = nullptr;
    int (*CommInitRank)(RcclComm*, int, RcclUid, int) = nullptr;
    int (*CommInitAll)(RcclComm*, int, const int*) = nullptr;
    int (*CommDestroy)(RcclComm) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, RcclComm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;      // (the communicator's side of the library: the one piece of process-wide state)
std::mutex g_rccl_mu;  // two threads may create their communicators at the same time

int load_rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl.handle) return CPPF_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return fail(CPPF_ERR_UNSUPPORTED, std::string("cppflow_hip: cannot load RCCL (librccl.so.1): ") + dlerror());
    RcclApi api;
    api.handle = h;
#define CPPF_SYM(field, name)                                                                        \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(h, name));                               \
    if (!api.field) return fail(CPPF_ERR_UNSUPPORTED, std::string("cppflow_hip: RCCL lacks ") + name)
    CPPF_SYM(GetUniqueId, "ncclGetUniqueId");
    CPPF_SYM(CommInitRank, "ncclCommInitRank");
    CPPF_SYM(CommInitAll, "ncclCommInitAll");
    CPPF_SYM(CommDestroy, "ncclCommDestroy");
    CPPF_SYM(AllGather, "ncclAllGather");
    CPPF_SYM(GroupStart, "ncclGroupStart");
    CPPF_SYM(GroupEnd, "ncclGroupEnd");
    CPPF_SYM(GetErrorString, "ncclGetErrorString");
#undef CPPF_SYM
    g_rccl = api;
    return CPPF_OK;
}

#define CPPF_RCCL(call)                                                                                          \
    do {                                                                                                         \
        int r__ = (call);                                                                                        \
        if (r__ != 0)                                                                                            \
            return fail(CPPF_ERR_HIP, std::string("cppflow_hip: " #call " failed: ") + g_rccl.GetErrorString(r__)); \
    } while (0)
}  // namespace

struct cppf_comm {
    RcclComm comm;
    int rank, world, device;
};

int cppf_comm_available(void) { return load_rccl(); }

int cppf_comm_unique_id(void* id_out) {
    CPPF_REQUIRE(id_out, "id_out is NULL");
    if (int rc = load_rccl()) return rc;
    RcclUid id;
    CPPF_RCCL(g_rccl.GetUniqueId(&id));
    std::memcpy(id_out, id.bytes, CPPF_COMM_ID_BYTES);
    return CPPF_OK;
}

int cppf_comm_init_rank(const void* id, int rank, int world, int device, cppf_comm** out) {
    CPPF_REQUIRE(id && out, "id / out is NULL");
    CPPF_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank / world out of range");
    *out = nullptr;
    if (int rc = load_rccl()) return rc;
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return fail(CPPF_ERR_HIP, std::string("cppflow_hip: selecting the device failed: ") + hipGetErrorString(guard.err));
    RcclUid uid;
    std::memcpy(uid.bytes, id, CPPF_COMM_ID_BYTES);
    RcclComm c = nullptr;
    CPPF_RCCL(g_rccl.CommInitRank(&c, world, uid, rank));
    *out = new cppf_comm{c, rank, world, device};
    return CPPF_OK;
}

int cppf_comm_init_all(int n_devices, const int* devices, cppf_comm** out) {
    CPPF_REQUIRE(n_devices >= 1 && devices && out, "n_devices < 1 or NULL pointer");
    if (int rc = load_rccl()) return rc;
    std::vector<RcclComm> comms((size_t)n_devices, nullptr);
    CPPF_RCCL(g_rccl.CommInitAll(comms.data(), n_devices, devices));
    for (int i = 0; i < n_devices; ++i) out[i] = new cppf_comm{comms[(size_t)i], i, n_devices, devices[i]};
    return CPPF_OK;
}

int cppf_comm_rank(const cppf_comm* comm) { return comm ? comm->rank : CPPF_ERR_INVALID; }

int cppf_comm_world(const cppf_comm* comm) { return comm ? comm->world : CPPF_ERR_INVALID; }

int cppf_allgather_bytes(cppf_comm* comm, const void* send, void* recv, size_t bytes_per_rank, void* stream) {
    CPPF_REQUIRE(comm, "comm is NULL");
    if (bytes_per_rank == 0) return CPPF_OK;
    CPPF_REQUIRE(send && recv, "send / recv is NULL");
    DeviceGuard guard(comm->device);
    if (guard.err != hipSuccess) return fail(CPPF_ERR_HIP, std::string("cppflow_hip: selecting the device failed: ") + hipGetErrorString(guard.err));
    CPPF_RCCL(g_rccl.AllGather(send, recv, bytes_per_rank, /* ncclUint8 */ 1, comm->comm, (hipStream_t)stream));
    return CPPF_OK;
}

int cppf_comm_group_begin(void) {
    if (int rc = load_rccl()) return rc;
    CPPF_RCCL(g_rccl.GroupStart());
    return CPPF_OK;
}

int cppf_comm_group_end(void) {
    if (int rc = load_rccl()) return rc;
    CPPF_RCCL(g_rccl.GroupEnd());
    return CPPF_OK;
}

void cppf_comm_destroy(cppf_comm* comm) {
    if (!comm) return;
    if (g_rccl.handle && comm->comm) (void)g_rccl.CommDestroy(comm->comm);
    delete comm;
}

}  // extern "C"

/*
 * lmik_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY) for the batched LM-IK hot path of jstmn/cppflow.
 *
 * This file is the checker, never the product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it.  The product (cppflow_amd/) never imports, links or calls anything under oracle/.
 *
 * It restates, in plain scalar C, the algorithm of the reference's pose-only Levenberg-Marquardt step and of its
 * batched collision masks, in the reference's own operation order:
 *
 *   orc_pose_errors      <- cppflow/optimization_utils.py:802-820   get_6d_pose_errors
 *   orc_lm_step          <- cppflow/optimization.py:61-92           levenberg_marquardt_only_pose
 *                           (scale rows :77-80, A = J^T J + lambda I :85-86, b = J^T e :87, LU solve :88, x+delta :90-92)
 *   orc_clamp            <- cppflow/optimization_utils.py:823-833   clamp_to_joint_limits
 *   orc_pose_metrics     <- cppflow/evaluation_utils.py:113-116,134-141 positional_errors / rotational_errors
 *                           (geodesic formula quoted at cppflow/data_types.py:408-411)
 *   orc_self_dists/orc_env_dists + orc_masks
 *                        <- cppflow/collision_detection.py:27-69    min over pairs/links, "< 0", OR over obstacles
 *   orc_jlim_mask        <- cppflow/search.py:25-52                 joint_limit_almost_violations_3d
 *   orc_ext_cost         <- cppflow/search.py:14-15,146-150         100*jlim + 1000*env + 1000*self
 *   orc_angular_changes  <- cppflow/evaluation_utils.py:144-154
 *   orc_seed_validity    <- cppflow/optimization_utils.py:845-884 + cppflow/evaluation_utils.py:29-75
 *   orc_plan_metrics     <- cppflow/data_types.py:140-264 (Plan properties) + cppflow/evaluation_utils.py:16-27, 83-99
 *
 * THIRD-PARTY ARITHMETIC.  Forward kinematics, the geometric Jacobian, capsule distances and the quaternion helpers
 * are methods of `jrl` 0.1.2 @ ef4c2f6eb1ba84395ff0bb01d5b7713854df6908 (reference pyproject.toml:12,
 * uv.lock:907-909), which is NOT vendored in /root/reference and not installable here.  Their published semantics are
 * restated from the reference's call sites:
 *   - FK pose layout [x y z qw qx qy qz]              (README.md:8, cppflow/ros2/ros2_utils.py:19-35)
 *   - Jacobian rows 0:3 angular, 3:6 linear, world    (cppflow/optimization.py:77-80, optimization_utils.py:806-808)
 *   - prismatic Jacobian column = [0; axis]           (tests/optimization_utils_test.py:377-402)
 *   - w-first Hamilton quaternions; geodesic = 2*acos(clamp(dot, -1+1e-7, 1-1e-7)) folded to [0, pi]
 *   - capsule-capsule: closed-form segment-segment distance minus radii
 *   - capsule-cuboid: exact segment / axis-aligned-box distance minus radius (jrl's own algorithm is unknown;
 *     this definition is the build's, DESIGN.md "capsule-cuboid distance")
 * PARITY PINNING: the reference holds no numeric FK / Jacobian / distance vector for its robots, so absolute model
 * values are "parity unpinned"; the oracle is pinned against every known-answer test the reference does hold for
 * this path (tests/test_oracle_kats.py lists them with file:line).
 *
 * Two builds of this one file:
 *   liborc64.so  (default)   REAL = double, libm sin/cos      -> ground truth
 *   liborc32.so  (-DORC_F32) REAL = float, canonical fp32 op order with explicit fmaf and the Cody-Waite sincos below,
 *                            compiled -ffp-contract=off       -> bit-level reference for masks / FK
 * The interface is double in both (inputs must be fp32-representable for the f32 build to be exact on entry).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef ORC_F32
typedef float REAL;
#define FMA(a, b, c) fmaf((a), (b), (c))
#define SQRT(a) sqrtf(a)
#define ATAN2(a, b) atan2f((a), (b))
#define ASIN(a) asinf(a)
#define ACOS(a) acosf(a)
#define FABS(a) fabsf(a)
#define FMOD(a, b) fmodf((a), (b))
#define RC(x) x##f
#else
typedef double REAL;
#define FMA(a, b, c) fma((a), (b), (c))
#define SQRT(a) sqrt(a)
#define ATAN2(a, b) atan2((a), (b))
#define ASIN(a) asin(a)
#define ACOS(a) acos(a)
#define FABS(a) fabs(a)
#define FMOD(a, b) fmod((a), (b))
#define RC(x) x
#endif

#define ORC_MAX_DOF 16
#define ORC_MAX_CAPS 24
#define ORC_MAX_PAIRS 128
#define ORC_MAX_OBS 8

typedef struct {
    int ndof;
    REAL F[ORC_MAX_DOF][12]; /* canonical fixed transforms: R row-major (9) then t (3) */
    REAL Fee[12];
    int jtype[ORC_MAX_DOF]; /* 0 revolute about local z, 1 prismatic along local z */
    REAL lo[ORC_MAX_DOF], hi[ORC_MAX_DOF];
    int ncaps;
    int cap_link[ORC_MAX_CAPS]; /* -1 = base */
    REAL cap_p0[ORC_MAX_CAPS][3], cap_p1[ORC_MAX_CAPS][3], cap_r[ORC_MAX_CAPS];
    /* the capsule as the distance functions use it: centre, half-axis, |h|^2, 1 / |h|^2 (see orc_robot_create) */
    REAL cap_c[ORC_MAX_CAPS][3], cap_h[ORC_MAX_CAPS][3], cap_a[ORC_MAX_CAPS], cap_ia[ORC_MAX_CAPS];
    int npairs;
    int pair_a[ORC_MAX_PAIRS], pair_b[ORC_MAX_PAIRS];
} orc_robot;

/* ------------------------------------------------------------------------------------------------------------- */
/* robot handle                                                                                                     */

void* orc_robot_create(int ndof, const double* F, const double* Fee, const int* jtype, const double* lo,
                       const double* hi, int ncaps, const int* cap_link, const double* cap_p0, const double* cap_p1,
                       const double* cap_r, int npairs, const int* pairs) {
    if (ndof < 1 || ndof > ORC_MAX_DOF || ncaps < 0 || ncaps > ORC_MAX_CAPS || npairs < 0 || npairs > ORC_MAX_PAIRS)
        return NULL;
    orc_robot* rb = (orc_robot*)calloc(1, sizeof(orc_robot));
    rb->ndof = ndof;
    for (int j = 0; j < ndof; ++j) {
        for (int k = 0; k < 12; ++k) rb->F[j][k] = (REAL)F[j * 12 + k];
        rb->jtype[j] = jtype[j];
        rb->lo[j] = (REAL)lo[j];
        rb->hi[j] = (REAL)hi[j];
    }
    for (int k = 0; k < 12; ++k) rb->Fee[k] = (REAL)Fee[k];
    rb->ncaps = ncaps;
    for (int c = 0; c < ncaps; ++c) {
        rb->cap_link[c] = cap_link[c];
        for (int k = 0; k < 3; ++k) {
            rb->cap_p0[c][k] = (REAL)cap_p0[c * 3 + k];
            rb->cap_p1[c][k] = (REAL)cap_p1[c * 3 + k];
        }
        rb->cap_r[c] = (REAL)cap_r[c];
        /* centre c = 0.5 (p0 + p1), half-axis h = 0.5 (p1 - p0), a = |h|^2, 1 / a: double arithmetic on the (fp32-valued) end
         * points, each rounded to fp32 ONCE, a over the rounded h summed (h0 h0 + h1 h1) + h2 h2, 1 / a (0 for a < 2^-100) -- the same lines as
         * cppflow_amd/gen_robots.py (capsule_centred) and csrc/cppflow_hip.hip (capsule_centred), so that all three hold the
         * same bits; the fp64 build keeps the fp32-valued c and h (it is the ground truth OF THAT capsule) with a, 1 / a unrounded. */
        float hf[3];
        for (int k = 0; k < 3; ++k) {
            const double p0 = (double)(float)cap_p0[c * 3 + k], p1 = (double)(float)cap_p1[c * 3 + k];
            rb->cap_c[c][k] = (REAL)(float)(0.5 * (p0 + p1));
            hf[k] = (float)(0.5 * (p1 - p0));
            rb->cap_h[c][k] = (REAL)hf[k];
        }
        const double a = ((double)hf[0] * (double)hf[0] + (double)hf[1] * (double)hf[1]) + (double)hf[2] * (double)hf[2];
        rb->cap_a[c] = (REAL)a; /* fp32 build: rounded once, the kernels' constants; fp64 build: the exact |h|^2 */
        rb->cap_ia[c] = a >= 0x1p-100 ? (REAL)(1.0 / a) : (REAL)0; /* a zero-length capsule is a sphere: its parameter stays 0 (rcp_rn) */
    }
    rb->npairs = npairs;
    for (int p = 0; p < npairs; ++p) {
        rb->pair_a[p] = pairs[2 * p];
        rb->pair_b[p] = pairs[2 * p + 1];
    }
    return rb;
}

void orc_robot_destroy(void* rb) { free(rb); }

void orc_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : 1);
#else
    (void)n;
#endif
}

int orc_is_f32(void) {
#ifdef ORC_F32
    return 1;
#else
    return 0;
#endif
}

/* ------------------------------------------------------------------------------------------------------------- */
/* sin / cos                                                                                                        */

#ifdef ORC_F32
/* Cody-Waite reduction by pi/2 (3 constants) + Cephes single-precision minimax polynomials on [-pi/4, pi/4].
 * Written with explicit fmaf so that the HIP kernels (which use the same formula) agree bit for bit. */
static void sincos_real(float x, float* s, float* c) {
    /* k = round-to-nearest-even(x * 2/pi) by the magic-number trick: adding 1.5 * 2^23 leaves the integer in the low mantissa
     * bits of t (one rounding, of the exact fma), so t's bit pattern also carries k mod 4 without a float->int conversion */
    const float magic = 12582912.0f;
    float t = fmaf(x, 0.63661977236758134f, magic);
    float k = t - magic;
    float r = fmaf(-k, 1.5703125f, x);
    r = fmaf(-k, 4.837512969970703125e-4f, r);
    r = fmaf(-k, 7.54978995489188e-8f, r);
    float z = r * r;
    float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    float sn = fmaf(r * z, ps, r);
    float pc = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    float cs = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
    uint32_t ki, so, co, sb, cb;
    memcpy(&ki, &t, 4);
    memcpy(&sb, &sn, 4);
    memcpy(&cb, &cs, 4);
    so = (ki & 1u) ? cb : sb; /* odd quadrants swap the two polynomials */
    co = (ki & 1u) ? sb : cb;
    so ^= (ki << 30) & 0x80000000u;        /* sin is negated in quadrants 2, 3 */
    co ^= ((ki + 1u) << 30) & 0x80000000u; /* cos in quadrants 1, 2 */
    memcpy(s, &so, 4);
    memcpy(c, &co, 4);
}
#else
static void sincos_real(double x, double* s, double* c) {
    *s = sin(x);
    *c = cos(x);
}
#endif

/* ------------------------------------------------------------------------------------------------------------- */
/* forward kinematics in canonical order                                                                            */

typedef struct {
    REAL R[9];
    REAL p[3];
} frame_t;

/* p' = R*t + p ;  A = R*Fr   (canonical order: k = 0 product first, then fma k = 1, k = 2) */
static void apply_fixed(const frame_t* in, const REAL* Fk, frame_t* out) {
    const REAL* Fr = Fk;
    const REAL* Ft = Fk + 9;
    for (int i = 0; i < 3; ++i) {
        const REAL r0 = in->R[3 * i], r1 = in->R[3 * i + 1], r2 = in->R[3 * i + 2];
        out->p[i] = FMA(r2, Ft[2], FMA(r1, Ft[1], FMA(r0, Ft[0], in->p[i])));
        for (int c = 0; c < 3; ++c) out->R[3 * i + c] = FMA(r2, Fr[6 + c], FMA(r1, Fr[3 + c], r0 * Fr[c]));
    }
}

/* motion about / along local z */
static void apply_joint(frame_t* f, int jtype, REAL q) {
    if (jtype == 0) {
        REAL s, c;
        sincos_real(q, &s, &c);
        for (int i = 0; i < 3; ++i) {
            const REAL a0 = f->R[3 * i], a1 = f->R[3 * i + 1];
            f->R[3 * i] = FMA(s, a1, c * a0);
            f->R[3 * i + 1] = FMA(c, a1, -(s * a0));
        }
    } else {
        for (int i = 0; i < 3; ++i) f->p[i] = FMA(f->R[3 * i + 2], q, f->p[i]);
    }
}

/* link frames after each joint's motion (frames[j]), joint axes/origins (for the Jacobian) and the ee frame */
static void fk_chain(const orc_robot* rb, const REAL* q, frame_t* links, REAL (*axis)[3], REAL (*origin)[3],
                     frame_t* ee) {
    frame_t cur;
    memset(&cur, 0, sizeof(cur));
    cur.R[0] = cur.R[4] = cur.R[8] = (REAL)1;
    for (int j = 0; j < rb->ndof; ++j) {
        frame_t nxt;
        apply_fixed(&cur, rb->F[j], &nxt);
        if (axis) {
            for (int i = 0; i < 3; ++i) {
                axis[j][i] = nxt.R[3 * i + 2];
                origin[j][i] = nxt.p[i];
            }
        }
        apply_joint(&nxt, rb->jtype[j], q[j]);
        cur = nxt;
        if (links) links[j] = cur;
    }
    if (ee) apply_fixed(&cur, rb->Fee, ee);
}

/* rotation matrix -> unit quaternion, w first; the branch with the largest of (w, x, y, z) is used so the divisor is
 * >= 1 (restates the 4-candidate scheme of jrl.math_utils.rotation_matrix_to_quaternion; sign: largest component > 0) */
static void mat_to_quat(const REAL* R, REAL* q) {
    const REAL m00 = R[0], m01 = R[1], m02 = R[2], m10 = R[3], m11 = R[4], m12 = R[5], m20 = R[6], m21 = R[7],
               m22 = R[8];
    const REAL one = (REAL)1;
    REAL qa[4];
    qa[0] = one + m00 + m11 + m22;
    qa[1] = one + m00 - m11 - m22;
    qa[2] = one - m00 + m11 - m22;
    qa[3] = one - m00 - m11 + m22;
    int best = 0;
    for (int i = 1; i < 4; ++i)
        if (qa[i] > qa[best]) best = i;
    const REAL d = SQRT(qa[best] > 0 ? qa[best] : 0); /* = 2*|component| */
    const REAL inv = RC(0.5) / d;
    switch (best) {
        case 0:
            q[0] = RC(0.5) * d;
            q[1] = (m21 - m12) * inv;
            q[2] = (m02 - m20) * inv;
            q[3] = (m10 - m01) * inv;
            break;
        case 1:
            q[0] = (m21 - m12) * inv;
            q[1] = RC(0.5) * d;
            q[2] = (m10 + m01) * inv;
            q[3] = (m02 + m20) * inv;
            break;
        case 2:
            q[0] = (m02 - m20) * inv;
            q[1] = (m10 + m01) * inv;
            q[2] = RC(0.5) * d;
            q[3] = (m12 + m21) * inv;
            break;
        default:
            q[0] = (m10 - m01) * inv;
            q[1] = (m20 + m02) * inv;
            q[2] = (m21 + m12) * inv;
            q[3] = RC(0.5) * d;
            break;
    }
}

static void frame_to_pose(const frame_t* f, REAL* pose) {
    pose[0] = f->p[0];
    pose[1] = f->p[1];
    pose[2] = f->p[2];
    mat_to_quat(f->R, pose + 3);
}

/* x[n,d] -> poses[n,7]   (Robot.forward_kinematics) */
void orc_fk(const void* h, const double* x, int n, double* poses) {
    const orc_robot* rb = (const orc_robot*)h;
    const int d = rb->ndof;
    for (int r = 0; r < n; ++r) {
        REAL q[ORC_MAX_DOF], pose[7];
        frame_t ee;
        for (int j = 0; j < d; ++j) q[j] = (REAL)x[(size_t)r * d + j];
        fk_chain(rb, q, NULL, NULL, NULL, &ee);
        frame_to_pose(&ee, pose);
        for (int k = 0; k < 7; ++k) poses[(size_t)r * 7 + k] = pose[k];
    }
}

/* x[n,d] -> link frames [n, d+1, 12] (R row-major, p); entry d is the ee frame.  Debug / test helper. */
void orc_link_frames(const void* h, const double* x, int n, double* out) {
    const orc_robot* rb = (const orc_robot*)h;
    const int d = rb->ndof;
    for (int r = 0; r < n; ++r) {
        REAL q[ORC_MAX_DOF];
        frame_t links[ORC_MAX_DOF + 1];
        for (int j = 0; j < d; ++j) q[j] = (REAL)x[(size_t)r * d + j];
        fk_chain(rb, q, links, NULL, NULL, &links[d]);
        for (int j = 0; j <= d; ++j) {
            double* o = out + ((size_t)r * (d + 1) + j) * 12;
            for (int k = 0; k < 9; ++k) o[k] = links[j].R[k];
            for (int k = 0; k < 3; ++k) o[9 + k] = links[j].p[k];
        }
    }
}

/* geometric Jacobian [6,d]: rows 0:3 angular, 3:6 linear, world frame */
static void jacobian_row(const orc_robot* rb, const REAL* q, REAL* J /*[6*d]*/, frame_t* ee) {
    const int d = rb->ndof;
    REAL axis[ORC_MAX_DOF][3], origin[ORC_MAX_DOF][3];
    fk_chain(rb, q, NULL, axis, origin, ee);
    for (int j = 0; j < d; ++j) {
        const REAL* z = axis[j];
        if (rb->jtype[j] == 0) {
            const REAL rx = ee->p[0] - origin[j][0], ry = ee->p[1] - origin[j][1], rz = ee->p[2] - origin[j][2];
            J[0 * d + j] = z[0];
            J[1 * d + j] = z[1];
            J[2 * d + j] = z[2];
            J[3 * d + j] = z[1] * rz - z[2] * ry;
            J[4 * d + j] = z[2] * rx - z[0] * rz;
            J[5 * d + j] = z[0] * ry - z[1] * rx;
        } else {
            J[0 * d + j] = J[1 * d + j] = J[2 * d + j] = 0;
            J[3 * d + j] = z[0];
            J[4 * d + j] = z[1];
            J[5 * d + j] = z[2];
        }
    }
}

void orc_jacobian(const void* h, const double* x, int n, double* Jout) {
    const orc_robot* rb = (const orc_robot*)h;
    const int d = rb->ndof;
    for (int r = 0; r < n; ++r) {
        REAL q[ORC_MAX_DOF], J[6 * ORC_MAX_DOF];
        frame_t ee;
        for (int j = 0; j < d; ++j) q[j] = (REAL)x[(size_t)r * d + j];
        jacobian_row(rb, q, J, &ee);
        for (int k = 0; k < 6 * d; ++k) Jout[(size_t)r * 6 * d + k] = J[k];
    }
}

/* ------------------------------------------------------------------------------------------------------------- */
/* quaternion helpers (jrl.math_utils semantics)                                                         */

static void quat_conj(const REAL* q, REAL* o) {
    o[0] = q[0];
    o[1] = -q[1];
    o[2] = -q[2];
    o[3] = -q[3];
}

static void quat_mul(const REAL* a, const REAL* b, REAL* o) {
    o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}

static void quat_to_rpy(const REAL* q, REAL* rpy) {
    const REAL q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    rpy[0] = ATAN2(2 * (q0 * q1 + q2 * q3), 1 - 2 * (q1 * q1 + q2 * q2));
    REAL sp = 2 * (q0 * q2 - q3 * q1);
    if (sp > 1) sp = 1;
    if (sp < -1) sp = -1;
    rpy[1] = ASIN(sp);
    rpy[2] = ATAN2(2 * (q0 * q3 + q1 * q2), 1 - 2 * (q2 * q2 + q3 * q3));
}

/* get_6d_pose_errors, cppflow/optimization_utils.py:802-820: e = [roll, pitch, yaw, x, y, z] errors */
static void pose_error_row(const REAL* cur, const REAL* tgt, REAL* e) {
    REAL inv[4], qe[4];
    for (int i = 0; i < 3; ++i) e[3 + i] = tgt[i] - cur[i]; /* :813-814 */
    quat_conj(cur + 3, inv);                                /* :816 */
    quat_mul(tgt + 3, inv, qe);                             /* :817 */
    quat_to_rpy(qe, e);                                     /* :818-819 */
}

void orc_pose_errors(const void* h, const double* x, const double* target, int n, double* e_out, double* cur_out) {
    const orc_robot* rb = (const orc_robot*)h;
    const int d = rb->ndof;
    for (int r = 0; r < n; ++r) {
        REAL q[ORC_MAX_DOF], cur[7], tgt[7], e[6];
        frame_t ee;
        for (int j = 0; j < d; ++j) q[j] = (REAL)x[(size_t)r * d + j];
        for (int k = 0; k < 7; ++k) tgt[k] = (REAL)target[(size_t)r * 7 + k];
        fk_chain(rb, q, NULL, NULL, NULL, &ee);
        frame_to_pose(&ee, cur);
        pose_error_row(cur, tgt, e);
        for (int k = 0; k < 6; ++k) e_out[(size_t)r * 6 + k] = e[k];
        if (cur_out)
            for (int k = 0; k < 7; ++k) cur_out[(size_t)r * 7 + k] = cur[k];
    }
}

/* ------------------------------------------------------------------------------------------------------------- */
/* dense solves                                                                                                     */

/* LU with partial pivoting, as torch.linalg.solve (getrf/getrs) does; A[d*d] row-major is destroyed. */
static int lu_solve(REAL* A, REAL* b, int d) {
    for (int k = 0; k < d; ++k) {
        int piv = k;
        REAL best = FABS(A[k * d + k]);
        for (int i = k + 1; i < d; ++i)
            if (FABS(A[i * d + k]) > best) {
                best = FABS(A[i * d + k]);
                piv = i;
            }
        if (best == 0) return -1;
        if (piv != k) {
            for (int c = 0; c < d; ++c) {
                REAL t = A[k * d + c];
                A[k * d + c] = A[piv * d + c];
                A[piv * d + c] = t;
            }
            REAL t = b[k];
            b[k] = b[piv];
            b[piv] = t;
        }
        for (int i = k + 1; i < d; ++i) {
            const REAL m = A[i * d + k] / A[k * d + k];
            A[i * d + k] = m;
            for (int c = k + 1; c < d; ++c) A[i * d + c] -= m * A[k * d + c];
            b[i] -= m * b[k];
        }
    }
    for (int i = d - 1; i >= 0; --i) {
        REAL s = b[i];
        for (int c = i + 1; c < d; ++c) s -= A[i * d + c] * b[c];
        b[i] = s / A[i * d + i];
    }
    return 0;
}

/* Cholesky A = L L^T (reference cppflow/optimization.py:95-113 uses this for the full step; test_cholesky property) */
static int chol_solve(REAL* A, REAL* b, int d) {
    for (int j = 0; j < d; ++j) {
        REAL s = A[j * d + j];
        for (int k = 0; k < j; ++k) s -= A[j * d + k] * A[j * d + k];
        if (!(s > 0)) return -1;
        const REAL l = SQRT(s);
        A[j * d + j] = l;
        for (int i = j + 1; i < d; ++i) {
            REAL t = A[i * d + j];
            for (int k = 0; k < j; ++k) t -= A[i * d + k] * A[j * d + k];
            A[i * d + j] = t / l;
        }
    }
    for (int i = 0; i < d; ++i) {
        REAL s = b[i];
        for (int k = 0; k < i; ++k) s -= A[i * d + k] * b[k];
        b[i] = s / A[i * d + i];
    }
    for (int i = d - 1; i >= 0; --i) {
        REAL s = b[i];
        for (int k = i + 1; k < d; ++k) s -= A[k * d + i] * b[k];
        b[i] = s / A[i * d + i];
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------- */
/* LM step                                                                                                          */

/* one row of levenberg_marquardt_only_pose (cppflow/optimization.py:61-92).  J and e are returned SCALED, as the
 * reference scales them in place before returning them (:77-80, :90-92).  Returns 0, or -1 if the solve broke down. */
static int lm_step_row(const orc_robot* rb, const REAL* q, const REAL* tgt, REAL lambda, REAL a_pos, REAL a_rot,
                       int solver, REAL* q_new, REAL* J, REAL* e) {
    const int d = rb->ndof;
    REAL cur[7];
    frame_t ee;
    jacobian_row(rb, q, J, &ee);  /* :74 */
    frame_to_pose(&ee, cur);
    pose_error_row(cur, tgt, e);  /* :73 */
    for (int i = 0; i < 3; ++i) { /* :77-80 */
        e[3 + i] *= a_pos;
        e[i] *= a_rot;
        for (int j = 0; j < d; ++j) {
            J[(3 + i) * d + j] *= a_pos;
            J[i * d + j] *= a_rot;
        }
    }
    REAL A[ORC_MAX_DOF * ORC_MAX_DOF], b[ORC_MAX_DOF];
    for (int i = 0; i < d; ++i) { /* :85-87 */
        for (int j = 0; j < d; ++j) {
            REAL s = 0;
            for (int k = 0; k < 6; ++k) s += J[k * d + i] * J[k * d + j];
            A[i * d + j] = s + (i == j ? lambda : (REAL)0);
        }
        REAL s = 0;
        for (int k = 0; k < 6; ++k) s += J[k * d + i] * e[k];
        b[i] = s;
    }
    const int rc = solver == 1 ? chol_solve(A, b, d) : lu_solve(A, b, d); /* :88 */
    for (int j = 0; j < d; ++j) q_new[j] = q[j] + b[j];                    /* :90-92 */
    return rc;
}

/* x[n,d], target[n,7] (already stacked, optimization.py:399-401) -> x_new[n,d], optional J[n,6,d], e[n,6].
 * solver: 0 = LU with partial pivoting (torch.linalg.solve), 1 = Cholesky.  Returns number of rows whose solve failed. */
int orc_lm_step(const void* h, const double* x, const double* target, int n, double lambda, double a_pos,
                double a_rot, int solver, double* x_new, double* J_out, double* e_out) {
    const orc_robot* rb = (const orc_robot*)h;
    const int d = rb->ndof;
    int fails = 0;
    for (int r = 0; r < n; ++r) {
        REAL q[ORC_MAX_DOF], qn[ORC_MAX_DOF], tgt[7], J[6 * ORC_MAX_DOF], e[6];
        for (int j = 0; j < d; ++j) q[j] = (REAL)x[(size_t)r * d + j];
        for (int k = 0; k < 7; ++k) tgt[k] = (REAL)target[(size_t)r * 7 + k];
        if (lm_step_row(rb, q, tgt, (REAL)lambda, (REAL)a_pos, (REAL)a_rot, solver, qn, J, e) != 0) ++fails;
        for (int j = 0; j < d; ++j) x_new[(size_t)r * d + j] = qn[j];
        if (J_out)
            for (int k = 0; k < 6 * d; ++k) J_out[(size_t)r * 6 * d + k] = J[k];
        if (e_out)
            for (int k = 0; k < 6; ++k) e_out[(size_t)r * 6 + k] = e[k];
    }
    return fails;
}

/* clamp_to_joint_limits, cppflow/optimization_utils.py:831-833 (in place) */
void orc_clamp(const void* h, double* x, int n) {
    const orc_robot* rb = (const orc_robot*)h;
    const int d = rb->ndof;
    for (int r = 0; r < n; ++r)
        for (int j = 0; j < d; ++j) {
            REAL v = (REAL)x[(size_t)r * d + j];
            if (v < rb->lo[j]) v = rb->lo[j];
            if (v > rb->hi[j]) v = rb->hi[j];
            x[(size_t)r * d + j] = v;
        }
}

/* K iterations of { pose-only step ; clamp } -- the pose branch of the loop at cppflow/optimization.py:251-259 */
int orc_lm_steps(const void* h, const double* x, const double* target, int n, int n_steps, double lambda,
                 double a_pos, double a_rot, int solver, double* x_out) {
    const orc_robot* rb = (const orc_robot*)h;
    const int d = rb->ndof;
    int fails = 0;
#pragma omp parallel for reduction(+ : fails) schedule(static)
    for (int r = 0; r < n; ++r) {
        REAL q[ORC_MAX_DOF], qn[ORC_MAX_DOF], tgt[7], J[6 * ORC_MAX_DOF], e[6];
        for (int j = 0; j < d; ++j) q[j] = (REAL)x[(size_t)r * d + j];
        for (int k = 0; k < 7; ++k) tgt[k] = (REAL)target[(size_t)r * 7 + k];
        for (int it = 0; it < n_steps; ++it) {
            if (lm_step_row(rb, q, tgt, (REAL)lambda, (REAL)a_pos, (REAL)a_rot, solver, qn, J, e) != 0) ++fails;
            for (int j = 0; j < d; ++j) {
                REAL v = qn[j];
                if (v < rb->lo[j]) v = rb->lo[j];
                if (v > rb->hi[j]) v = rb->hi[j];
                q[j] = v;
            }
        }
        for (int j = 0; j < d; ++j) x_out[(size_t)r * d + j] = q[j];
    }
    return fails;
}

/* ------------------------------------------------------------------------------------------------------------- */
/* pose-error metrics                                                                                               */

/* positional_errors / rotational_errors (cppflow/evaluation_utils.py:134-141): ||t_target - t_cur||_2 in metres and the
 * geodesic quaternion distance in radians: dot clipped to [-1,1], then 2*acos(clamp(dot, -1+1e-7, 1-1e-7))
 * (formula quoted at cppflow/data_types.py:408-411), folded into [0, pi] so that q and -q are the same rotation. */
void orc_pose_metrics(const void* h, const double* x, const double* target, int n, double* pos_err_m,
                      double* rot_err_rad) {
    const orc_robot* rb = (const orc_robot*)h;
    const int d = rb->ndof;
    const REAL pi = RC(3.14159265358979323846);
    for (int r = 0; r < n; ++r) {
        REAL q[ORC_MAX_DOF], cur[7], tgt[7];
        frame_t ee;
        for (int j = 0; j < d; ++j) q[j] = (REAL)x[(size_t)r * d + j];
        for (int k = 0; k < 7; ++k) tgt[k] = (REAL)target[(size_t)r * 7 + k];
        fk_chain(rb, q, NULL, NULL, NULL, &ee);
        frame_to_pose(&ee, cur);
        const REAL dx = tgt[0] - cur[0], dy = tgt[1] - cur[1], dz = tgt[2] - cur[2];
        pos_err_m[r] = SQRT(dx * dx + dy * dy + dz * dz);
        REAL dot = tgt[3] * cur[3] + tgt[4] * cur[4] + tgt[5] * cur[5] + tgt[6] * cur[6];
        if (dot > 1) dot = 1;
        if (dot < -1) dot = -1;
        const REAL eps = RC(1e-7);
        if (dot > 1 - eps) dot = 1 - eps;
        if (dot < -1 + eps) dot = -1 + eps;
        REAL dist = 2 * ACOS(dot);
        /* |remainder(dist + pi, 2 pi) - pi| */
        REAL m = FMOD(dist + pi, 2 * pi);
        if (m < 0) m += 2 * pi;
        rot_err_rad[r] = FABS(m - pi);
    }
}

/* The same two metrics with the geodesic term evaluated the numerically careful way: both quaternions normalised,
 * angle = 2*atan2(|vec(q_t * q_c^-1)|, |w(q_t * q_c^-1)|), floored at the reference's clamp value 2*acos(1 - 1e-7).
 * Mathematically identical to orc_pose_metrics; numerically it removes that formula's sensitivity to the quaternion
 * norm (d(2 acos)/d(dot) = -2/sin(theta/2): a 1e-7 norm error of an fp32-stored target moves the result by 1e-4 rad at
 * theta = 2e-3).  This is the quantity the HIP kernels report (DESIGN.md "rotation error"). */
void orc_pose_metrics_exact(const void* h, const double* x, const double* target, int n, double* pos_err_m,
                            double* rot_err_rad) {
    const orc_robot* rb = (const orc_robot*)h;
    const int d = rb->ndof;
    for (int r = 0; r < n; ++r) {
        REAL q[ORC_MAX_DOF], cur[7], tgt[7], inv[4], qe[4];
        frame_t ee;
        for (int j = 0; j < d; ++j) q[j] = (REAL)x[(size_t)r * d + j];
        for (int k = 0; k < 7; ++k) tgt[k] = (REAL)target[(size_t)r * 7 + k];
        fk_chain(rb, q, NULL, NULL, NULL, &ee);
        frame_to_pose(&ee, cur);
        const REAL dx = tgt[0] - cur[0], dy = tgt[1] - cur[1], dz = tgt[2] - cur[2];
        pos_err_m[r] = SQRT(dx * dx + dy * dy + dz * dz);
        REAL nt = SQRT(tgt[3] * tgt[3] + tgt[4] * tgt[4] + tgt[5] * tgt[5] + tgt[6] * tgt[6]);
        REAL nc = SQRT(cur[3] * cur[3] + cur[4] * cur[4] + cur[5] * cur[5] + cur[6] * cur[6]);
        for (int k = 3; k < 7; ++k) {
            tgt[k] /= nt;
            cur[k] /= nc;
        }
        quat_conj(cur + 3, inv);
        quat_mul(tgt + 3, inv, qe);
        const REAL vn = SQRT(qe[1] * qe[1] + qe[2] * qe[2] + qe[3] * qe[3]);
        REAL th = 2 * ATAN2(vn, FABS(qe[0]));
        const REAL floor_v = 2 * ACOS((REAL)1 - RC(1e-7));
        rot_err_rad[r] = th > floor_v ? th : floor_v;
    }
}

/* ------------------------------------------------------------------------------------------------------------- */
/* collision distances (canonical order)                                                                            */

static inline REAL dot3(const REAL* a, const REAL* b) { return FMA(a[2], b[2], FMA(a[1], b[1], a[0] * b[0])); }
static inline REAL clamp11(REAL v) { return v < -1 ? (REAL)-1 : (v > 1 ? (REAL)1 : v); }

/* The correctly rounded reciprocal, DEFINED as 0 below 2^-100 (zero, denormals, and -- for the positive-only form -- negative
 * denominators): what every caller wants from a vanishing denominator.  The HIP kernels evaluate it as v_rcp_f32 + one
 * Newton step, which equals RN(1/x) for every fp32 x with 2^-126 <= |x| < 2^126 (all 2^32 bit patterns compared on the GPU:
 * scripts/ubench/rcp_exhaustive.hip, tests/test_gpu_round3.py) -- so a plain division here is the same bits. */
static inline REAL rcp_rn(REAL x) { return FABS(x) >= RC(0x1p-100) ? (REAL)1 / x : (REAL)0; }
static inline REAL rcp_rn_pos(REAL x) { return x >= RC(0x1p-100) ? (REAL)1 / x : (REAL)0; }

/* A capsule's segment is { C + u H : u in [-1, 1] } (centre, half-axis; a = |H|^2 and ia = 1 / a are constants of the rigid
 * link).  Closest distance between two such segments (non-degenerate), Ericson "Real-Time Collision Detection" 5.1.9
 * re-parametrised to [-1, 1]: minimise | r + s H1 - t H2 |^2, r = C1 - C2; also returns the closest points c1, c2. */
static REAL seg_seg_closest(const REAL* C1, const REAL* H1, const REAL* C2, const REAL* H2, REAL a, REAL ia, REAL e,
                            REAL ie, REAL* c1, REAL* c2) {
    REAL rr[3];
    for (int i = 0; i < 3; ++i) rr[i] = C1[i] - C2[i];
    const REAL b = dot3(H1, H2), c = dot3(H1, rr), f = dot3(H2, rr);
    const REAL denom = FMA(-b, b, a * e);
    REAL s = clamp11(FMA(b, f, -(c * e)) * rcp_rn_pos(denom)); /* parallel: s = 0, the centre */
    REAL t = FMA(b, s, f) * ie;
    if (t < -1) {
        t = -1;
        s = clamp11(-(b + c) * ia);
    } else if (t > 1) {
        t = 1;
        s = clamp11((b - c) * ia);
    }
    REAL df[3];
    for (int i = 0; i < 3; ++i) {
        df[i] = FMA(-t, H2[i], FMA(s, H1[i], rr[i]));
        c1[i] = FMA(s, H1[i], C1[i]);
        c2[i] = FMA(t, H2[i], C2[i]);
    }
    return SQRT(dot3(df, df));
}

static REAL seg_seg_dist(const REAL* C1, const REAL* H1, const REAL* C2, const REAL* H2, REAL a, REAL ia, REAL e, REAL ie) {
    REAL c1[3], c2[3];
    return seg_seg_closest(C1, H1, C2, H2, a, ia, e, ie, c1, c2);
}

static inline REAL clampr(REAL x, REAL lo, REAL hi) { return x < lo ? lo : (x > hi ? hi : x); }
static inline REAL maxr(REAL a, REAL b) { return a > b ? a : b; }
static inline REAL minr(REAL a, REAL b) { return a < b ? a : b; }

/* exact distance from the segment { C + u H } to the axis-aligned box [lo, hi] (0 if they intersect), with the closest
 * points.  Along the segment x_i(u) = C_i + u H_i the excess over the slab [lo_i, hi_i] is H_i (u - clamp(u, a_i, b_i)),
 * [a_i, b_i] being the parameter interval in which coordinate i is inside the slab; so the half-derivative of dist^2 is
 *     g(u) = sum_i w_i (u - clamp(u, a_i, b_i)),  w_i = H_i^2,
 * nondecreasing and piecewise linear with break points a_i, b_i.  g is evaluated at u = -1, 1 and the six clamped break
 * points; because g is monotone the bracket of its root is two independent max / min reductions over those candidates
 * (ul = max{c : g(c) <= 0}, gl = max{g(c) : g(c) <= 0}; ur = min{c : g(c) > 0}, gr = min{g(c) : g(c) > 0}), and the root is
 * interpolated linearly inside the bracket. */
static REAL seg_box_closest(const REAL* C, const REAL* H, const REAL* lo, const REAL* hi, REAL* cs, REAL* cb) {
    REAL w[3], ua[3], ub[3], cand[8], gv[8];
    for (int i = 0; i < 3; ++i) {
        const REAL inv = rcp_rn(H[i]);
        const REAL u0 = (lo[i] - C[i]) * inv, u1 = (hi[i] - C[i]) * inv;
        ua[i] = minr(u0, u1);
        ub[i] = maxr(u0, u1);
        w[i] = H[i] * H[i];
        cand[2 + 2 * i] = clamp11(ua[i]);
        cand[3 + 2 * i] = clamp11(ub[i]);
    }
    cand[0] = -1;
    cand[1] = 1;
    for (int k = 0; k < 8; ++k) {
        const REAL u = cand[k];
        gv[k] = FMA(w[2], u - clampr(u, ua[2], ub[2]), FMA(w[1], u - clampr(u, ua[1], ub[1]), w[0] * (u - clampr(u, ua[0], ub[0]))));
    }
    REAL ul = -1, gl = gv[0], ur = 1, gr = gv[1];
    for (int k = 2; k < 8; ++k) {
        const int neg = gv[k] <= 0;
        ul = maxr(ul, neg ? cand[k] : (REAL)-1);
        gl = maxr(gl, neg ? gv[k] : gv[0]);
        ur = minr(ur, neg ? (REAL)1 : cand[k]);
        gr = minr(gr, neg ? gv[1] : gv[k]);
    }
    const REAL u_in = FMA(ur - ul, (-gl) * rcp_rn_pos(gr - gl), ul); /* a flat bracket: its left end */
    const REAL u = gv[0] >= 0 ? (REAL)-1 : (gv[1] <= 0 ? (REAL)1 : u_in);
    REAL ex[3];
    for (int i = 0; i < 3; ++i) {
        const REAL x = FMA(H[i], u, C[i]);
        cs[i] = x;
        cb[i] = clampr(x, lo[i], hi[i]);
        ex[i] = x - cb[i];
    }
    return SQRT(dot3(ex, ex));
}

static REAL seg_box_dist(const REAL* C, const REAL* H, const REAL* lo, const REAL* hi) {
    REAL cs[3], cb[3];
    return seg_box_closest(C, H, lo, hi, cs, cb);
}

/* world centre / half-axis of every capsule from the link frames (canonical order: the centre as a point, R c + p; the
 * half-axis as a direction, R h) */
static void capsules_from_links(const orc_robot* rb, const frame_t* links, REAL (*wc)[3], REAL (*wh)[3]) {
    for (int c = 0; c < rb->ncaps; ++c) {
        const int li = rb->cap_link[c];
        if (li < 0) {
            for (int i = 0; i < 3; ++i) {
                wc[c][i] = rb->cap_c[c][i];
                wh[c][i] = rb->cap_h[c][i];
            }
        } else {
            const frame_t* f = &links[li];
            for (int i = 0; i < 3; ++i) {
                const REAL r0 = f->R[3 * i], r1 = f->R[3 * i + 1], r2 = f->R[3 * i + 2];
                wc[c][i] = FMA(r2, rb->cap_c[c][2], FMA(r1, rb->cap_c[c][1], FMA(r0, rb->cap_c[c][0], f->p[i])));
                wh[c][i] = FMA(r2, rb->cap_h[c][2], FMA(r1, rb->cap_h[c][1], r0 * rb->cap_h[c][0]));
            }
        }
    }
}

static void capsule_endpoints(const orc_robot* rb, const REAL* q, REAL (*wc)[3], REAL (*wh)[3]) {
    frame_t links[ORC_MAX_DOF];
    fk_chain(rb, q, links, NULL, NULL, NULL);
    capsules_from_links(rb, links, wc, wh);
}

#define PAIR_CONSTS(rb, a, b) (rb)->cap_a[a], (rb)->cap_ia[a], (rb)->cap_a[b], (rb)->cap_ia[b]

/* Robot.self_collision_distances: x[n,d] -> dists[n,P] (signed: segment distance - r_a - r_b), SURVEY a11 */
void orc_self_dists(const void* h, const double* x, int n, double* dists) {
    const orc_robot* rb = (const orc_robot*)h;
    const int d = rb->ndof;
    for (int r = 0; r < n; ++r) {
        REAL q[ORC_MAX_DOF], w0[ORC_MAX_CAPS][3], w1[ORC_MAX_CAPS][3];
        for (int j = 0; j < d; ++j) q[j] = (REAL)x[(size_t)r * d + j];
        capsule_endpoints(rb, q, w0, w1);
        for (int p = 0; p < rb->npairs; ++p) {
            const int a = rb->pair_a[p], b = rb->pair_b[p];
            const REAL dist = seg_seg_dist(w0[a], w1[a], w0[b], w1[b], PAIR_CONSTS(rb, a, b));
            dists[(size_t)r * rb->npairs + p] = dist - (rb->cap_r[a] + rb->cap_r[b]);
        }
    }
}

/* Robot.env_collision_distances for ONE axis-aligned cuboid given by its world-frame corners lo/hi:
 * x[n,d] -> dists[n,L] (signed: segment-box distance - r), column i <-> i-th capsule, SURVEY a12 */
void orc_env_dists(const void* h, const double* x, int n, const double* box_lo, const double* box_hi, double* dists) {
    const orc_robot* rb = (const orc_robot*)h;
    const int d = rb->ndof;
    REAL lo[3], hi[3];
    for (int i = 0; i < 3; ++i) {
        lo[i] = (REAL)box_lo[i];
        hi[i] = (REAL)box_hi[i];
    }
    for (int r = 0; r < n; ++r) {
        REAL q[ORC_MAX_DOF], w0[ORC_MAX_CAPS][3], w1[ORC_MAX_CAPS][3];
        for (int j = 0; j < d; ++j) q[j] = (REAL)x[(size_t)r * d + j];
        capsule_endpoints(rb, q, w0, w1);
        for (int c = 0; c < rb->ncaps; ++c)
            dists[(size_t)r * rb->ncaps + c] = seg_box_dist(w0[c], w1[c], lo, hi) - rb->cap_r[c];
    }
}

/* capsule world endpoints [n, L, 6] (test helper) */
void orc_capsule_endpoints(const void* h, const double* x, int n, double* out) {
    const orc_robot* rb = (const orc_robot*)h;
    const int d = rb->ndof;
    for (int r = 0; r < n; ++r) {
        REAL q[ORC_MAX_DOF], w0[ORC_MAX_CAPS][3], w1[ORC_MAX_CAPS][3];
        for (int j = 0; j < d; ++j) q[j] = (REAL)x[(size_t)r * d + j];
        capsule_endpoints(rb, q, w0, w1);
        for (int c = 0; c < rb->ncaps; ++c)
            for (int i = 0; i < 3; ++i) {
                out[((size_t)r * rb->ncaps + c) * 6 + i] = w0[c][i] - w1[c][i];     /* centre - half-axis */
                out[((size_t)r * rb->ncaps + c) * 6 + 3 + i] = w0[c][i] + w1[c][i]; /* centre + half-axis */
            }
    }
}

/* qpaths_batched_{self,env}_collisions (cppflow/collision_detection.py:27-69) + joint_limit_almost_violations_3d
 * (cppflow/search.py:25-52) + q_costs_external (cppflow/search.py:146-150) for n = k*T rows.
 * jl_lo / jl_hi are the already-padded limits (l + eps, u - eps; search.py:46-51).  Any output may be NULL. */
void orc_masks(const void* h, const double* x, int n, int nobs, const double* box_lo /*[O,3]*/,
               const double* box_hi /*[O,3]*/, const double* jl_lo, const double* jl_hi, uint8_t* self_mask,
               uint8_t* env_mask, uint8_t* jlim_mask, double* ext_cost, double* min_self, double* min_env) {
    const orc_robot* rb = (const orc_robot*)h;
    const int d = rb->ndof;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < n; ++r) {
        REAL q[ORC_MAX_DOF], w0[ORC_MAX_CAPS][3], w1[ORC_MAX_CAPS][3];
        for (int j = 0; j < d; ++j) q[j] = (REAL)x[(size_t)r * d + j];
        capsule_endpoints(rb, q, w0, w1);
        REAL ms = INFINITY;
        for (int p = 0; p < rb->npairs; ++p) {
            const int a = rb->pair_a[p], b = rb->pair_b[p];
            const REAL v = seg_seg_dist(w0[a], w1[a], w0[b], w1[b], PAIR_CONSTS(rb, a, b)) - (rb->cap_r[a] + rb->cap_r[b]);
            if (v < ms) ms = v;
        }
        int env = 0;
        REAL me_all = INFINITY;
        for (int o = 0; o < nobs; ++o) {
            REAL lo[3], hi[3], me = INFINITY;
            for (int i = 0; i < 3; ++i) {
                lo[i] = (REAL)box_lo[o * 3 + i];
                hi[i] = (REAL)box_hi[o * 3 + i];
            }
            for (int c = 0; c < rb->ncaps; ++c) {
                const REAL v = seg_box_dist(w0[c], w1[c], lo, hi) - rb->cap_r[c];
                if (v < me) me = v;
            }
            env |= (me < 0); /* collision_detection.py:39-43 */
            if (me < me_all) me_all = me;
        }
        const int self = (ms < 0); /* collision_detection.py:66-68 */
        int jl = 0;
        if (jl_lo && jl_hi)
            for (int j = 0; j < d; ++j) jl |= (q[j] < (REAL)jl_lo[j]) | (q[j] > (REAL)jl_hi[j]); /* search.py:52 */
        if (self_mask) self_mask[r] = (uint8_t)self;
        if (env_mask) env_mask[r] = (uint8_t)env;
        if (jlim_mask) jlim_mask[r] = (uint8_t)jl;
        if (ext_cost) ext_cost[r] = 100.0 * jl + 1000.0 * env + 1000.0 * self; /* search.py:14-15,146-150 */
        if (min_self) min_self[r] = ms;
        if (min_env) min_env[r] = me_all;
    }
}

/* ------------------------------------------------------------------------------------------------------------- */
/* trajectory metrics                                                                                               */

static REAL wrap_pi(REAL dq) {
    /* torch.remainder(dq + pi, 2 pi) - pi   (cppflow/evaluation_utils.py:151-153) */
    const REAL pi = RC(3.14159265358979323846);
    REAL m = FMOD(dq + pi, 2 * pi);
    if (m < 0) m += 2 * pi;
    return m - pi;
}

/* angular_changes: qpath[T,c] -> [T-1,c] */
void orc_angular_changes(const double* qpath, int T, int c, double* out) {
    for (int t = 0; t + 1 < T; ++t)
        for (int j = 0; j < c; ++j)
            out[(size_t)t * c + j] =
                wrap_pi((REAL)qpath[(size_t)(t + 1) * c + j] - (REAL)qpath[(size_t)t * c + j]);
}

/* validity half of x_is_valid for every seed (cppflow/optimization_utils.py:845-884, evaluation_utils.py:29-75):
 * x[S*W,d], target[S*W,7] -> per seed: max pos err (cm), max rot err (deg), mjac revolute (deg), mjac prismatic (cm). */
void orc_seed_validity(const void* h, const double* x, const double* target, int S, int W, double* out /*[S,4]*/) {
    const orc_robot* rb = (const orc_robot*)h;
    const int d = rb->ndof;
    const REAL rad2deg = RC(57.29577951308232087680);
    double* pe = (double*)malloc(sizeof(double) * (size_t)W);
    double* re = (double*)malloc(sizeof(double) * (size_t)W);
    for (int s = 0; s < S; ++s) {
        const double* xs = x + (size_t)s * W * d;
        orc_pose_metrics(h, xs, target + (size_t)s * W * 7, W, pe, re);
        REAL mp = 0, mr = 0, mrev = 0, mpri = 0;
        for (int w = 0; w < W; ++w) {
            const REAL a = (REAL)100 * (REAL)pe[w], b = rad2deg * (REAL)re[w];
            if (a > mp) mp = a;
            if (b > mr) mr = b;
        }
        for (int w = 0; w + 1 < W; ++w)
            for (int j = 0; j < d; ++j) {
                const REAL dq = (REAL)xs[(size_t)(w + 1) * d + j] - (REAL)xs[(size_t)w * d + j];
                if (rb->jtype[j] == 0) {
                    const REAL v = FABS(rad2deg * wrap_pi(dq));
                    if (v > mrev) mrev = v;
                } else {
                    const REAL v = FABS((REAL)100 * dq);
                    if (v > mpri) mpri = v;
                }
            }
        out[s * 4 + 0] = mp;
        out[s * 4 + 1] = mr;
        out[s * 4 + 2] = mrev;
        out[s * 4 + 3] = mpri;
    }
    free(pe);
    free(re);
}

/* Plan metrics of S joint-space paths at once (cppflow/data_types.py:140-264 `Plan` properties, evaluated for every seed):
 * x[S*W,d], target[S*W,7], optional per-row collision flags and an optional initial configuration q_init[d] ->
 * out[S,16] = { max / mean positional error (cm), max / mean rotational error (deg)   (data_types.py:156-186),
 *               mjac revolute (deg), mjac prismatic (cm)                              (:189-210, evaluation_utils.py:83-99),
 *               path length revolute (rad), prismatic (m)                            (:141-149),
 *               # (waypoint, joint) entries outside the joint limits                 (evaluation_utils.py:16-27),
 *               # self-colliding, # env-colliding waypoints                           (:245-246),
 *               ||q_init - q_path[0]||_2 (0 without q_init)                          (:217-221),  0, 0, 0, 0 } */
void orc_plan_metrics(const void* h, const double* x, const double* target, int S, int W, const unsigned char* self_mask,
                      const unsigned char* env_mask, const double* q_init, double* out /*[S,16]*/) {
    const orc_robot* rb = (const orc_robot*)h;
    const int d = rb->ndof;
    const REAL rad2deg = RC(57.29577951308232087680);
    double* pe = (double*)malloc(sizeof(double) * (size_t)W);
    double* re = (double*)malloc(sizeof(double) * (size_t)W);
    for (int s = 0; s < S; ++s) {
        const double* xs = x + (size_t)s * W * d;
        orc_pose_metrics(h, xs, target + (size_t)s * W * 7, W, pe, re);
        REAL mp = 0, sp = 0, mr = 0, sr = 0, mrev = 0, mpri = 0, lrad = 0, lm = 0, njl = 0, ns = 0, ne = 0, qd = 0;
        for (int w = 0; w < W; ++w) {
            const REAL a = (REAL)100 * (REAL)pe[w], b = rad2deg * (REAL)re[w];
            if (a > mp) mp = a;
            if (b > mr) mr = b;
            sp += a;
            sr += b;
            for (int j = 0; j < d; ++j) {
                const REAL q = (REAL)xs[(size_t)w * d + j];
                njl += (REAL)((q < (REAL)rb->lo[j]) + ((REAL)rb->hi[j] < q));
            }
            if (self_mask) ns += (REAL)self_mask[(size_t)s * W + w];
            if (env_mask) ne += (REAL)env_mask[(size_t)s * W + w];
        }
        for (int w = 0; w + 1 < W; ++w)
            for (int j = 0; j < d; ++j) {
                const REAL dq = (REAL)xs[(size_t)(w + 1) * d + j] - (REAL)xs[(size_t)w * d + j];
                if (rb->jtype[j] == 0) {
                    const REAL v = FABS(wrap_pi(dq));
                    if (rad2deg * v > mrev) mrev = rad2deg * v;
                    lrad += v;
                } else {
                    const REAL v = FABS(dq);
                    if ((REAL)100 * v > mpri) mpri = (REAL)100 * v;
                    lm += v;
                }
            }
        if (q_init) {
            for (int j = 0; j < d; ++j) {
                const REAL dq = (REAL)q_init[j] - (REAL)xs[j];
                qd += dq * dq;
            }
            qd = SQRT(qd);
        }
        double* o = out + (size_t)s * 16;
        o[0] = mp, o[1] = sp / (REAL)W, o[2] = mr, o[3] = sr / (REAL)W, o[4] = mrev, o[5] = mpri, o[6] = lrad, o[7] = lm;
        o[8] = njl, o[9] = ns, o[10] = ne, o[11] = qd, o[12] = o[13] = o[14] = o[15] = 0;
    }
    free(pe);
    free(re);
}

/* ------------------------------------------------------------------------------------------------------------- */
/* dp_search (cppflow/search.py:100-191)                                                                            */

/* Min-max dynamic programme over k candidate paths: costs[:,0] = ext[:,0];
 *   costs[b,t] = min_a max(mjac(a->b, t-1), costs[a,t-1]) + ext[b,t],   memo[b,t] = argmin_a (first minimal a)
 * with mjac(a->b, t-1) = max_j |remainder(s_j * (q[b,t,j] - q[a,t-1,j]) + pi, 2 pi) - pi|, s_j = prismatic_scaling for
 * prismatic joints (search.py:119-123: the scaling is applied to dq BEFORE the wrap, to every joint it is given for) and
 * 1 otherwise.  Then the back-trace of search.py:161-173 from argmin_b costs[b,T-1].  Outputs: best_idx[T], costs[k,T]. */
void orc_dp_search(const void* h, const double* q, const double* ext, int k, int T, double prismatic_scaling, int* best_idx,
                   double* costs_out) {
    const orc_robot* rb = (const orc_robot*)h;
    const int d = rb->ndof;
    REAL* costs = (REAL*)malloc(sizeof(REAL) * (size_t)k * T);
    int* memo = (int*)calloc((size_t)k * T, sizeof(int));
    for (int b = 0; b < k; ++b) costs[(size_t)b * T] = (REAL)ext[(size_t)b * T];
    for (int t = 1; t < T; ++t) {
#pragma omp parallel for schedule(static)
        for (int b = 0; b < k; ++b) {
            REAL best = INFINITY;
            int arg = 0;
            for (int a = 0; a < k; ++a) {
                REAL m = 0;
                for (int j = 0; j < d; ++j) {
                    REAL dq = (REAL)q[((size_t)b * T + t) * d + j] - (REAL)q[((size_t)a * T + t - 1) * d + j];
                    if (rb->jtype[j] == 1) dq *= (REAL)prismatic_scaling;
                    const REAL v = FABS(wrap_pi(dq));
                    if (v > m) m = v;
                }
                const REAL c = costs[(size_t)a * T + t - 1];
                const REAL v = (m > c ? m : c) + (REAL)ext[(size_t)b * T + t];
                if (v < best) {
                    best = v;
                    arg = a;
                }
            }
            costs[(size_t)b * T + t] = best;
            memo[(size_t)b * T + t] = arg;
        }
    }
    int i = 0;
    for (int b = 1; b < k; ++b)
        if (costs[(size_t)b * T + T - 1] < costs[(size_t)i * T + T - 1]) i = b;
    for (int t = T - 1; t >= 0; --t) {
        best_idx[t] = i;
        i = memo[(size_t)i * T + t];
    }
    if (costs_out)
        for (size_t n = 0; n < (size_t)k * T; ++n) costs_out[n] = costs[n];
    free(costs);
    free(memo);
}

/* ------------------------------------------------------------------------------------------------------------- */
/* the coupled ("full") LM step: cppflow/optimization.py:95-144 + LmResidualFns.get_r_and_J, optimization_utils.py:486-731 */

typedef struct {
    double lm_lambda, alpha_position, alpha_rotation, alpha_differencing, alpha_differencing_prismatic_scaling,
        alpha_virtual_configs, alpha_self_collision, alpha_env_collision;
    int use_pose, use_differencing, use_virtual_configs, n_virtual_configs, use_self_collisions, use_env_collisions;
    /* the "satisfied" row options of LmResidualFns.get_r_and_J (cppflow/optimization_utils.py:514-533, 562-598); thresholds as
     * the reference forms them (oracle.py: FullParams.from_params) */
    int pose_do_scale_down_satisfied;
    double pose_threshold_m, pose_threshold_rad, pose_scale_down;
    int differencing_mode; /* 0 none, 1 differencing_do_ignore_satisfied, 2 differencing_do_scale_satisfied */
    double differencing_threshold_rad, differencing_threshold_m, differencing_scale_down;
    int differencing_shift_invalid_to_threshold;
} orc_full_params;

/* d(point rigidly attached to moving link `link`)/dq_j, world frame: z_j x (c - o_j) (revolute) or z_j (prismatic) for
 * j <= link, 0 otherwise (joint j moves link l iff j <= l) */
static void point_jacobian_col(const orc_robot* rb, int link, int j, const REAL (*axis)[3], const REAL (*origin)[3],
                               const REAL* c, REAL* out) {
    out[0] = out[1] = out[2] = 0;
    if (j > link) return;
    const REAL* z = axis[j];
    if (rb->jtype[j] == 0) {
        const REAL rx = c[0] - origin[j][0], ry = c[1] - origin[j][1], rz = c[2] - origin[j][2];
        out[0] = z[1] * rz - z[2] * ry;
        out[1] = z[2] * rx - z[0] * rz;
        out[2] = z[0] * ry - z[1] * rx;
    } else {
        out[0] = z[0], out[1] = z[1], out[2] = z[2];
    }
}

/* capsule world centres / half-axes (w0 / w1 below) + joint axes / origins for one configuration */
static void capsules_and_axes(const orc_robot* rb, const REAL* q, REAL (*w0)[3], REAL (*w1)[3], REAL (*axis)[3],
                              REAL (*origin)[3]) {
    frame_t links[ORC_MAX_DOF];
    fk_chain(rb, q, links, axis, origin, NULL);
    capsules_from_links(rb, links, w0, w1);
}

/* Closest points nearer than this (1 um) count as TOUCHING: the direction between them is rounding noise there -- a segment that
 * crosses a box has distance 0 in exact arithmetic and 1e-9 in fp32, in some direction -- so the gradient is 0 (the same constant
 * and rule as csrc/kernels_coupled.h kTouch). */
#define ORC_TOUCH RC(1e-6)

/* signed self-collision distances [P] and their gradients [P,d] for one configuration: the derivative of the minimum
 * distance equals the derivative with the closest points held fixed on their links (envelope theorem); where the
 * segments touch (distance below ORC_TOUCH) the direction is undefined and the gradient is set to 0.
 * (jrl.Robot.self_collision_distances_jacobian, call site cppflow/optimization_utils.py:670 -- un-vendored; this is the
 * analytic gradient of this build's own distance definition, checked against finite differences in the tests.) */
static void self_dists_and_grads(const orc_robot* rb, const REAL* q, REAL* dist, REAL* grad) {
    const int d = rb->ndof;
    REAL w0[ORC_MAX_CAPS][3], w1[ORC_MAX_CAPS][3], axis[ORC_MAX_DOF][3], origin[ORC_MAX_DOF][3];
    capsules_and_axes(rb, q, w0, w1, axis, origin);
    for (int p = 0; p < rb->npairs; ++p) {
        const int a = rb->pair_a[p], b = rb->pair_b[p];
        REAL c1[3], c2[3];
        const REAL sd = seg_seg_closest(w0[a], w1[a], w0[b], w1[b], PAIR_CONSTS(rb, a, b), c1, c2);
        dist[p] = sd - (rb->cap_r[a] + rb->cap_r[b]);
        REAL n[3] = {0, 0, 0};
        if (sd > ORC_TOUCH)
            for (int i = 0; i < 3; ++i) n[i] = (c1[i] - c2[i]) / sd;
        for (int j = 0; j < d; ++j) {
            REAL j1[3], j2[3];
            point_jacobian_col(rb, rb->cap_link[a], j, axis, origin, c1, j1);
            point_jacobian_col(rb, rb->cap_link[b], j, axis, origin, c2, j2);
            grad[p * d + j] = n[0] * (j1[0] - j2[0]) + n[1] * (j1[1] - j2[1]) + n[2] * (j1[2] - j2[2]);
        }
    }
}

static void env_dists_and_grads(const orc_robot* rb, const REAL* q, const REAL* lo, const REAL* hi, REAL* dist, REAL* grad) {
    const int d = rb->ndof;
    REAL w0[ORC_MAX_CAPS][3], w1[ORC_MAX_CAPS][3], axis[ORC_MAX_DOF][3], origin[ORC_MAX_DOF][3];
    capsules_and_axes(rb, q, w0, w1, axis, origin);
    for (int c = 0; c < rb->ncaps; ++c) {
        REAL cs[3], cb[3];
        const REAL sd = seg_box_closest(w0[c], w1[c], lo, hi, cs, cb);
        dist[c] = sd - rb->cap_r[c];
        REAL n[3] = {0, 0, 0};
        if (sd > ORC_TOUCH)
            for (int i = 0; i < 3; ++i) n[i] = (cs[i] - cb[i]) / sd;
        for (int j = 0; j < d; ++j) {
            REAL j1[3];
            point_jacobian_col(rb, rb->cap_link[c], j, axis, origin, cs, j1);
            grad[c * d + j] = n[0] * j1[0] + n[1] * j1[1] + n[2] * j1[2];
        }
    }
}

/* x[n,d] -> dists[n,P], grads[n,P,d] / dists[n,L], grads[n,L,d]  (test helpers; finite-difference checks) */
void orc_self_dists_grads(const void* h, const double* x, int n, double* dists, double* grads) {
    const orc_robot* rb = (const orc_robot*)h;
    const int d = rb->ndof, P = rb->npairs;
    for (int r = 0; r < n; ++r) {
        REAL q[ORC_MAX_DOF], dist[ORC_MAX_PAIRS], grad[ORC_MAX_PAIRS * ORC_MAX_DOF];
        for (int j = 0; j < d; ++j) q[j] = (REAL)x[(size_t)r * d + j];
        self_dists_and_grads(rb, q, dist, grad);
        for (int p = 0; p < P; ++p) {
            dists[(size_t)r * P + p] = dist[p];
            for (int j = 0; j < d; ++j) grads[((size_t)r * P + p) * d + j] = grad[p * d + j];
        }
    }
}

void orc_env_dists_grads(const void* h, const double* x, int n, const double* box_lo, const double* box_hi, double* dists,
                         double* grads) {
    const orc_robot* rb = (const orc_robot*)h;
    const int d = rb->ndof, L = rb->ncaps;
    REAL lo[3], hi[3];
    for (int i = 0; i < 3; ++i) lo[i] = (REAL)box_lo[i], hi[i] = (REAL)box_hi[i];
    for (int r = 0; r < n; ++r) {
        REAL q[ORC_MAX_DOF], dist[ORC_MAX_CAPS], grad[ORC_MAX_CAPS * ORC_MAX_DOF];
        for (int j = 0; j < d; ++j) q[j] = (REAL)x[(size_t)r * d + j];
        env_dists_and_grads(rb, q, lo, hi, dist, grad);
        for (int c = 0; c < L; ++c) {
            dists[(size_t)r * L + c] = dist[c];
            for (int j = 0; j < d; ++j) grads[((size_t)r * L + c) * d + j] = grad[c * d + j];
        }
    }
}

/* The residual rows of ONE trajectory x [T,d] in the reference's stacking order (LmResidual.get_r, optimization_utils.py:59-72):
 * pose, differencing, virtual configs, self collisions, env collisions.  Every row touches the d columns of ONE waypoint t,
 * except a differencing row, which touches one column of t and the same column of t + 1.  Rows are handed to a sink:
 *   emit(ctx, r, t, jt [d], jt2 [d] or NULL)   -- J[row, t*d + j] = jt[j],  J[row, (t+1)*d + j] = jt2[j]
 * which returns non-zero to abort (row storage exhausted).  Returns the number of rows, or -1. */
typedef int (*row_sink)(void* ctx, REAL r, int t, const REAL* jt, const REAL* jt2);

static int full_rows(const orc_robot* rb, const orc_full_params* pm, const REAL* x, const REAL* target, const REAL* xv, int T,
                     int nobs, const REAL* box_lo, const REAL* box_hi, row_sink emit, void* ctx) {
    const int d = rb->ndof;
    int row = 0;
    REAL jt[ORC_MAX_DOF], jt2[ORC_MAX_DOF];
    if (pm->use_pose) { /* optimization_utils.py:503-543: r = pose errors, J = FK Jacobian, rows scaled by the alphas */
        for (int t = 0; t < T; ++t) {
            REAL Jt[6 * ORC_MAX_DOF], e[6], cur[7];
            frame_t ee;
            jacobian_row(rb, x + (size_t)t * d, Jt, &ee);
            frame_to_pose(&ee, cur);
            pose_error_row(cur, target + (size_t)t * 7, e);
            for (int i = 0; i < 6; ++i) {
                REAL a = (REAL)(i < 3 ? pm->alpha_rotation : pm->alpha_position);
                /* _scale_down_rows_from_r_J_pose_below_error (:288-333), applied to the UNSCALED rows before the alphas (:514-533):
                 * rotation rows against the rad threshold, position rows against the m threshold */
                if (pm->pose_do_scale_down_satisfied &&
                    FABS(e[i]) < (REAL)(i < 3 ? pm->pose_threshold_rad : pm->pose_threshold_m))
                    a *= (REAL)pm->pose_scale_down;
                for (int j = 0; j < d; ++j) jt[j] = a * Jt[i * d + j];
                if (emit(ctx, a * e[i], t, jt, NULL)) return -1;
                ++row;
            }
        }
    }
    if (pm->use_differencing) { /* :549-613: r = alpha * angular_changes(x); J = +1 at (t,j), -1 at (t+1,j) */
        for (int t = 0; t + 1 < T; ++t)
            for (int j = 0; j < d; ++j) {
                REAL a = (REAL)pm->alpha_differencing;
                REAL rho = wrap_pi(x[(size_t)(t + 1) * d + j] - x[(size_t)t * d + j]);
                const REAL thr = (REAL)(rb->jtype[j] == 1 ? pm->differencing_threshold_m : pm->differencing_threshold_rad);
                const REAL shifted = rho < -thr ? rho + thr : (rho > thr ? rho - thr : rho);
                if (pm->differencing_mode == 1) { /* filter_rows_from_r_J_differencing (:736-768), shift_to_threshold = True (:572-580) */
                    if (!(FABS(rho) > thr)) continue; /* the row is not in the system */
                    rho = shifted;
                } else if (pm->differencing_mode == 2) { /* _scale_down_rows_from_r_J_differencing_below_error (:352-398) */
                    if (FABS(rho) < thr)
                        a *= (REAL)pm->differencing_scale_down;
                    else if (pm->differencing_shift_invalid_to_threshold)
                        rho = shifted;
                }
                /* the prismatic scaling is skipped in filter mode (:601) */
                if (rb->jtype[j] == 1 && pm->differencing_mode != 1) a *= (REAL)pm->alpha_differencing_prismatic_scaling;
                for (int c = 0; c < d; ++c) jt[c] = jt2[c] = 0;
                jt[j] = a;
                jt2[j] = -a;
                if (emit(ctx, a * rho, t, jt, jt2)) return -1;
                ++row;
            }
    }
    if (pm->use_virtual_configs) { /* :618-640, :430-484: first / last n_vq configs, r = b * (x - x_virtual), J = -b I */
        const REAL b = (REAL)(pm->alpha_virtual_configs * pm->alpha_differencing);
        const int nv = pm->n_virtual_configs;
        for (int side = 0; side < 2; ++side)
            for (int i = 0; i < nv; ++i) {
                const int t = side == 0 ? i : T - nv + i;
                for (int j = 0; j < d; ++j) {
                    for (int c = 0; c < d; ++c) jt[c] = 0;
                    jt[j] = -b;
                    if (emit(ctx, b * wrap_pi(x[(size_t)t * d + j] - xv[(size_t)t * d + j]), t, jt, NULL)) return -1;
                    ++row;
                }
            }
    }
    if (pm->use_self_collisions) { /* :645-680: r = -alpha * dist where that is > 0, J = alpha * d(dist)/dq */
        for (int t = 0; t < T; ++t) {
            REAL dist[ORC_MAX_PAIRS], grad[ORC_MAX_PAIRS * ORC_MAX_DOF];
            self_dists_and_grads(rb, x + (size_t)t * d, dist, grad);
            for (int p = 0; p < rb->npairs; ++p) {
                const REAL rv = -(REAL)pm->alpha_self_collision * dist[p];
                if (!(rv > 0)) continue;
                for (int j = 0; j < d; ++j) jt[j] = (REAL)pm->alpha_self_collision * grad[p * d + j];
                if (emit(ctx, rv, t, jt, NULL)) return -1;
                ++row;
            }
        }
    }
    if (pm->use_env_collisions) { /* :685-727: per obstacle, same rule */
        for (int o = 0; o < nobs; ++o)
            for (int t = 0; t < T; ++t) {
                REAL dist[ORC_MAX_CAPS], grad[ORC_MAX_CAPS * ORC_MAX_DOF];
                env_dists_and_grads(rb, x + (size_t)t * d, box_lo + 3 * o, box_hi + 3 * o, dist, grad);
                for (int c = 0; c < rb->ncaps; ++c) {
                    const REAL rv = -(REAL)pm->alpha_env_collision * dist[c];
                    if (!(rv > 0)) continue;
                    for (int j = 0; j < d; ++j) jt[j] = (REAL)pm->alpha_env_collision * grad[c * d + j];
                    if (emit(ctx, rv, t, jt, NULL)) return -1;
                    ++row;
                }
            }
    }
    return row;
}

/* sink 1: the dense r [rows] and J [rows, T*d] the reference builds (torch.block_diag of the per-waypoint Jacobians) */
typedef struct {
    REAL *r, *J;
    int row, max_rows, d, N;
} dense_sink;

static int dense_emit(void* ctx, REAL rv, int t, const REAL* jt, const REAL* jt2) {
    dense_sink* k = (dense_sink*)ctx;
    if (k->row >= k->max_rows) return 1;
    REAL* Jr = k->J + (size_t)k->row * k->N;
    for (int c = 0; c < k->N; ++c) Jr[c] = 0;
    for (int j = 0; j < k->d; ++j) Jr[t * k->d + j] = jt[j];
    if (jt2)
        for (int j = 0; j < k->d; ++j) Jr[(t + 1) * k->d + j] = jt2[j];
    k->r[k->row++] = rv;
    return 0;
}

static int full_r_and_J(const orc_robot* rb, const orc_full_params* pm, const REAL* x, const REAL* target, const REAL* xv,
                        int T, int nobs, const REAL* box_lo, const REAL* box_hi, REAL* r, REAL* J, int max_rows) {
    dense_sink k = {r, J, 0, max_rows, rb->ndof, T * rb->ndof};
    return full_rows(rb, pm, x, target, xv, T, nobs, box_lo, box_hi, dense_emit, &k);
}

/* sink 2: A = J^T J and b = J^T r accumulated straight into BAND storage.  Two columns of one row are at most d apart (a
 * differencing row: column j of waypoints t and t + 1), so A has half-bandwidth d:  Ab[i * (d + 1) + k] = A[i][i - k], k = 0..d. */
typedef struct {
    REAL *Ab, *b;
    int d;
} band_sink;

static int band_emit(void* ctx, REAL rv, int t, const REAL* jt, const REAL* jt2) {
    band_sink* k = (band_sink*)ctx;
    const int d = k->d, w = d + 1;
    /* the row's non-zeros: columns t*d + j (jt) and (t+1)*d + j (jt2) */
    for (int half = 0; half < (jt2 ? 2 : 1); ++half) {
        const REAL* ja = half ? jt2 : jt;
        for (int j = 0; j < d; ++j) {
            if (ja[j] == 0) continue;
            const int i = (t + half) * d + j;
            k->b[i] += ja[j] * rv;
            for (int half2 = 0; half2 <= half; ++half2) { /* columns c <= i */
                const REAL* jb = half2 ? jt2 : jt;
                for (int j2 = 0; j2 < d; ++j2) {
                    const int c = (t + half2) * d + j2;
                    if (c > i || jb[j2] == 0) continue;
                    k->Ab[(size_t)i * w + (i - c)] += ja[j] * jb[j2];
                }
            }
        }
    }
    return 0;
}

/* banded Cholesky A = L L^T in place (L in the same band storage) and the two substitutions; 0 on success */
static int band_chol_solve(REAL* Ab, REAL* b, int N, int bw) {
    const int w = bw + 1;
#define LB(i, c) Ab[(size_t)(i) * w + ((i) - (c))]
    for (int j = 0; j < N; ++j) {
        REAL s = LB(j, j);
        for (int c = j - bw > 0 ? j - bw : 0; c < j; ++c) s -= LB(j, c) * LB(j, c);
        if (!(s > 0)) return 1;
        const REAL ljj = SQRT(s);
        LB(j, j) = ljj;
        for (int i = j + 1; i < N && i <= j + bw; ++i) {
            REAL v = LB(i, j);
            for (int c = i - bw > 0 ? i - bw : 0; c < j; ++c) v -= LB(i, c) * LB(j, c);
            LB(i, j) = v / ljj;
        }
    }
    for (int i = 0; i < N; ++i) {
        REAL v = b[i];
        for (int c = i - bw > 0 ? i - bw : 0; c < i; ++c) v -= LB(i, c) * b[c];
        b[i] = v / LB(i, i);
    }
    for (int i = N - 1; i >= 0; --i) {
        REAL v = b[i];
        for (int c = i + 1; c < N && c <= i + bw; ++c) v -= LB(c, i) * b[c];
        b[i] = v / LB(i, i);
    }
#undef LB
    return 0;
}

/* The coupled step of orc_lm_full_step without the dense dT x dT matrix: the same residual rows (full_rows), A = J^T J +
 * lambda I in band storage (half-bandwidth d), banded Cholesky -- O(T d^3) per trajectory instead of O((dT)^3), so the oracle
 * reaches the planner's path lengths (T = 256 .. 512) and seed counts.  Identical in exact arithmetic to the dense step;
 * tests/test_oracle_kats.py holds the two together at T <= 64.  Returns the number of failed factorisations. */
int orc_lm_full_step_banded(const void* h, const double* x, const double* target, const double* xv, int S, int T,
                            const orc_full_params* pm, int nobs, const double* box_lo, const double* box_hi, double* x_new) {
    const orc_robot* rb = (const orc_robot*)h;
    const int d = rb->ndof, N = T * d, w = d + 1;
    REAL* tg = (REAL*)malloc(sizeof(REAL) * (size_t)T * 7);
    REAL lo[3 * ORC_MAX_OBS], hi[3 * ORC_MAX_OBS];
    for (int i = 0; i < T * 7; ++i) tg[i] = (REAL)target[i];
    for (int i = 0; i < 3 * nobs; ++i) lo[i] = (REAL)box_lo[i], hi[i] = (REAL)box_hi[i];
    int fails = 0;
#pragma omp parallel for schedule(dynamic) reduction(+ : fails)
    for (int s = 0; s < S; ++s) {
        REAL* xs = (REAL*)malloc(sizeof(REAL) * (size_t)N);
        REAL* xvs = (REAL*)malloc(sizeof(REAL) * (size_t)N);
        REAL* Ab = (REAL*)calloc((size_t)N * w, sizeof(REAL));
        REAL* b = (REAL*)calloc((size_t)N, sizeof(REAL));
        for (int i = 0; i < N; ++i) {
            xs[i] = (REAL)x[(size_t)s * N + i];
            xvs[i] = xv ? (REAL)xv[(size_t)s * N + i] : xs[i];
        }
        band_sink k = {Ab, b, d};
        full_rows(rb, pm, xs, tg, xvs, T, nobs, lo, hi, band_emit, &k);
        for (int i = 0; i < N; ++i) Ab[(size_t)i * w] += (REAL)pm->lm_lambda;
        if (band_chol_solve(Ab, b, N, d) != 0) ++fails;
        for (int i = 0; i < N; ++i) x_new[(size_t)s * N + i] = xs[i] + b[i];
        free(xs), free(xvs), free(Ab), free(b);
    }
    free(tg);
    return fails;
}

static int chol_solve_n(REAL* A, REAL* b, int n); /* below */

/* One coupled LM step for each of S trajectories: dense J^T J + lambda I, Cholesky, x + delta (optimization.py:95-113).
 * x [S*T,d], target [T,7] (shared), xv [S*T,d] (virtual configs; NULL = x, as the loop sets them at :253).
 * Optionally returns the first trajectory's stacked residual (r_out, *rows_out).  Returns the number of failed solves. */
int orc_lm_full_step(const void* h, const double* x, const double* target, const double* xv, int S, int T,
                     const orc_full_params* pm, int nobs, const double* box_lo, const double* box_hi, double* x_new,
                     double* r_out, int* rows_out) {
    const orc_robot* rb = (const orc_robot*)h;
    const int d = rb->ndof, N = T * d;
    const int max_rows = 6 * T + d * T + 2 * d * (pm->n_virtual_configs > 0 ? pm->n_virtual_configs : 0) +
                         T * (rb->npairs + rb->ncaps * (nobs > 0 ? nobs : 0)) + 8;
    REAL* tg = (REAL*)malloc(sizeof(REAL) * (size_t)T * 7);
    REAL lo[3 * ORC_MAX_OBS], hi[3 * ORC_MAX_OBS];
    for (int i = 0; i < T * 7; ++i) tg[i] = (REAL)target[i];
    for (int i = 0; i < 3 * nobs; ++i) lo[i] = (REAL)box_lo[i], hi[i] = (REAL)box_hi[i];
    int fails = 0;
    for (int s = 0; s < S; ++s) {
        REAL* xs = (REAL*)malloc(sizeof(REAL) * (size_t)N);
        REAL* xvs = (REAL*)malloc(sizeof(REAL) * (size_t)N);
        REAL* r = (REAL*)malloc(sizeof(REAL) * (size_t)max_rows);
        REAL* J = (REAL*)malloc(sizeof(REAL) * (size_t)max_rows * N);
        REAL* A = (REAL*)calloc((size_t)N * N, sizeof(REAL));
        REAL* b = (REAL*)calloc((size_t)N, sizeof(REAL));
        for (int i = 0; i < N; ++i) {
            xs[i] = (REAL)x[(size_t)s * N + i];
            xvs[i] = xv ? (REAL)xv[(size_t)s * N + i] : xs[i];
        }
        const int rows = full_r_and_J(rb, pm, xs, tg, xvs, T, nobs, lo, hi, r, J, max_rows);
        if (s == 0 && r_out) {
            for (int i = 0; i < rows; ++i) r_out[i] = r[i];
            if (rows_out) *rows_out = rows;
        }
        /* A = J^T J + lambda I, b = J^T r (optimization.py:108-110); J is sparse per row: skip zeros */
        for (int k = 0; k < rows; ++k) {
            const REAL* Jk = J + (size_t)k * N;
            int nz[2 * ORC_MAX_DOF + 2], nnz = 0;
            for (int c = 0; c < N && nnz < 2 * ORC_MAX_DOF; ++c)
                if (Jk[c] != 0) nz[nnz++] = c;
            for (int a = 0; a < nnz; ++a) {
                b[nz[a]] += Jk[nz[a]] * r[k];
                for (int c = 0; c < nnz; ++c) A[(size_t)nz[a] * N + nz[c]] += Jk[nz[a]] * Jk[nz[c]];
            }
        }
        for (int i = 0; i < N; ++i) A[(size_t)i * N + i] += (REAL)pm->lm_lambda;
        if (chol_solve_n(A, b, N) != 0) ++fails;
        for (int i = 0; i < N; ++i) x_new[(size_t)s * N + i] = xs[i] + b[i];
        free(xs), free(xvs), free(r), free(J), free(A), free(b);
    }
    free(tg);
    return fails;
}

/* Cholesky solve for a heap matrix of any size (same algorithm as chol_solve) */
static int chol_solve_n(REAL* A, REAL* b, int n) { return chol_solve(A, b, n); }

/*
 * cppflow_hip.h -- C ABI of libcppflow_hip.so: the MI355X (gfx950) implementation of jstmn/cppflow's batched LM-IK
 * refinement hot path (pose-only Levenberg-Marquardt step + batched self / environment collision masks).
 *
 * The reference has no FFI for this path (it is pure Python over torch + the un-vendored `jrl` package); the drop-in
 * boundary is therefore a set of Python call signatures, mirrored by the modules of cppflow_amd/, each of which lands on exactly
 * one entry point below.  Every entry point cites the reference interface it replaces (paths under /root/reference).
 *
 * Conventions
 *   - plain C, no torch types; all `x`, `target`, output pointers are DEVICE pointers to contiguous row-major fp32
 *     (masks: uint8) on the robot's device; robot / obstacle descriptions are HOST pointers, copied at call time.
 *   - the library never allocates or frees caller buffers; every launch is asynchronous on `stream` (a hipStream_t
 *     passed as void*, NULL = the default stream); no host synchronisation inside any compute entry point.
 *   - rows are (seed, waypoint) pairs: row r = s * W + w; `target` is [W, 7] and is indexed by r % W, which replaces the
 *     reference's `torch.vstack([target_path] * parallel_count)` (cppflow/optimization.py:399-401).  Pass W = n for an
 *     already-stacked [n, 7] target.
 *   - return value 0 on success; CPPF_ERR_INVALID (contract / shape violation -> the Python shim raises AssertionError,
 *     mirroring the reference's asserts), CPPF_ERR_HIP (runtime failure -> RuntimeError), CPPF_ERR_UNSUPPORTED.
 *     cppf_last_error() returns a thread-local message for the last failure.
 *
 * Ownership and lifetimes (the caller owns all tensors)
 *   - caller buffers: never allocated, freed or retained by the library beyond the launches that were handed them -- except by a
 *     cppf_lm_batch, which records its items' pointers: they must stay valid while the batch is launched.
 *   - cppf_robot: created / destroyed by the caller.  A cppf_lm_batch is bound to the robot it was created from and keeps that
 *     handle ALLOCATED: cppf_robot_destroy on a robot with live batches marks it destroyed and returns; from then on every entry
 *     point given the handle or one of its batches (cppf_lm_batch_launch included) returns CPPF_ERR_INVALID without launching
 *     anything, and the memory is released by the cppf_lm_batch_destroy of its last batch.  So "robot first, then its batches" is
 *     a legal order of destruction; using a robot handle after cppf_robot_destroy when it has NO live batch is a use after free,
 *     as for any C handle.  cppf_lm_batch_destroy needs nothing but the batch.
 *   - cppf_comm: independent of robots and batches.
 *   - one host thread per handle at a time; handles on different threads are independent (no process-wide mutable state besides
 *     the RCCL function table).
 */
#ifndef CPPFLOW_HIP_H
#define CPPFLOW_HIP_H

#ifndef __HIPCC_RTC__
#include <stddef.h>
#include <stdint.h>
#else /* hipRTC (cppf_robot_specialize compiles this header too) has no <stdint.h>: the compiler's own fixed-width types */
typedef __INT8_TYPE__ int8_t;
typedef __UINT8_TYPE__ uint8_t;
typedef __INT32_TYPE__ int32_t;
typedef __UINT32_TYPE__ uint32_t;
typedef __UINT64_TYPE__ uint64_t;
typedef __SIZE_TYPE__ size_t;
typedef __UINTPTR_TYPE__ uintptr_t;
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define CPPF_ABI_VERSION 6

#define CPPF_MAX_DOF 12 /* every ndof in 3..12 is built (the reference's robots have 7 and 8) */
#define CPPF_MAX_CAPSULES 24
#define CPPF_MAX_PAIRS 128
#define CPPF_MAX_OBSTACLES 8
#define CPPF_MAX_BATCH 16 /* independent problems one batched fused launch can carry (cppf_lm_batch_create) */

#define CPPF_OK 0
#define CPPF_ERR_INVALID (-1)
#define CPPF_ERR_HIP (-2)
#define CPPF_ERR_UNSUPPORTED (-3)

#define CPPF_JOINT_REVOLUTE 0
#define CPPF_JOINT_PRISMATIC 1

/* Canonical serial chain (cppflow_amd/robot_model.py): world_T_link_j = prod_{i<=j} F_i * M_z(q_i), ee = link_{d-1} * F_ee.
 * Stands in for the `jrl.robot.Robot` object the reference passes around (cppflow/data_types.py:383). */
typedef struct cppf_robot_desc {
    int32_t ndof;
    float F[CPPF_MAX_DOF][12]; /* per joint: rotation row-major (9) then translation (3) */
    float F_ee[12];
    int32_t jtype[CPPF_MAX_DOF]; /* CPPF_JOINT_* : motion about / along the local z axis */
    float lo[CPPF_MAX_DOF];     /* robot.actuated_joints_limits (cppflow/optimization_utils.py:825-832) */
    float hi[CPPF_MAX_DOF];
    int32_t n_capsules;
    int32_t cap_link[CPPF_MAX_CAPSULES]; /* moving link the capsule rides on, -1 = base; must be non-decreasing */
    float cap_p0[CPPF_MAX_CAPSULES][3]; /* axis end points in the link frame; p0 == p1 exactly makes the capsule a sphere */
    float cap_p1[CPPF_MAX_CAPSULES][3]; /* |p1 - p0| = 0 (sphere) or > 1e-6 */
    float cap_r[CPPF_MAX_CAPSULES];
    int32_t n_pairs;
    int32_t pairs[CPPF_MAX_PAIRS][2]; /* capsule index pairs checked by self_collision_distances */
} cppf_robot_desc;

/* cppflow/lm_hyper_parameters.py:14-81 (fields the pose-only step reads) + fused-loop controls */
typedef struct cppf_lm_params {
    float lm_lambda;      /* OptimizationParameters.lm_lambda      (ALT_LOSS_V2_1_POSE: 1e-6, :123) */
    float alpha_position; /* OptimizationParameters.alpha_position (3.5,  :125) */
    float alpha_rotation; /* OptimizationParameters.alpha_rotation (0.35, :126) */
    int32_t n_steps;      /* K >= 1 fused { step ; clamp } iterations (cppflow/optimization.py:258-259 per iteration).  The LAST
                           * iteration of a launch -- the one that produces x_out -- is evaluated in the canonical arithmetic (the
                           * sine / cosine the bit-exact FK uses); the K - 1 iterations before it, whose iterates are not outputs,
                           * use cheaper elementary functions in the row shape: for the residual's roll / pitch / yaw shorter polynomials
                           * behind ONE shared reciprocal (4e-7 absolute), the 6x6 solve as a block L D L^T with 2x2 pivots (three
                           * reciprocals for six reciprocal square roots; 1.2 - 1.8x the Cholesky form's fp32 rounding error), and a
                           * cheaper sine / cosine (polynomials on [-pi, pi], 5e-7 absolute; the first
                           * iteration behind a reduction by whole turns -- the input need not lie inside the joint limits --
                           * the others directly: their iterates have been through the clamp).  A K = 1 launch (the reference's
                           * cadence) and every iteration of an early-out launch are canonical throughout.
                           * What K fused steps are worth against K steps of the reference (cppflow/optimization.py:61-92, 258-259),
                           * measured at K = 2, 3, 5 against the fp64 reference-order oracle on the four shipped robots and on the C4
                           * planner inputs (tests/test_gpu_lean_parity.py, profiles/r5_lean_parity.txt), in the scaled task space of
                           * the residual, ts = |J_s (x_K - x_K^oracle)|: median 2e-7, 90th percentile <= 1.5e-6 (the reference's own
                           * fp32 arithmetic: median 8e-6 .. 2e-4), and ROW BY ROW
                           *     ts <= K * eps * (1 + 2 |largest step| / sigma_min(J_s)),  eps = 1e-4 (CPPF_SOLVER_AUTO), 2e-5 (CPPF_SOLVER_F64)
                           * -- K per-step errors of the K = 1 bar grown by the row's own conditioning; pose error after K steps within
                           * 1e-5 + 2 K |step| |dx| of the oracle's.  The polynomials and the relative gate by themselves (K launches of
                           * one step against one launch of K): median 2.5e-7, 90th percentile <= 1.6e-6. */
    int32_t clamp;        /* 1: clamp_to_joint_limits after every step (the reference loop); 0: bare step (K must be 1) */
    /* Early-out (0 = off): a row whose residual at the start of an iteration has ||t_target - t|| < tol_pos_m and
     * ||(roll, pitch, yaw)|| < tol_rot_rad is left untouched from then on, and a wavefront whose rows are all below
     * tolerance leaves the loop -- the in-launch counterpart of the reference loop stopping once the pose is valid
     * (cppflow/optimization.py:251-258, 326-358).  Not combinable with J_out / e_out. */
    float tol_pos_m;
    float tol_rot_rad;
    /* Kernel shape: CPPF_SHAPE_AUTO picks by batch size; CPPF_SHAPE_ROW = one (seed, waypoint) row per lane (throughput
     * shape); CPPF_SHAPE_QUAD = four lanes cooperate on one row (latency shape for batches that cannot fill the chip). */
    int32_t shape;
    /* Precision of the damped solve (ndof >= 6; below that the primal fp32 solve).
     *   CPPF_SOLVER_AUTO (0, the default): the reference's dtype, conditioning-gated -- every row is solved in fp32; a row whose
     *     a-posteriori error estimate  eps * a_max * max diag(A) * max |y|  (the task-space size of the rounding error of forming
     *     and factoring A = J J^T + lambda S^-2, known once y = A^-1 e is) exceeds `solver_gate` redoes the solve in double
     *     precision.  The step is then never farther from the exactly solved one than the reference's own fp32 arithmetic
     *     (torch.linalg.solve on the primal system, cppflow/optimization.py:85-88) gets -- near-singular rows included.
     *     Two exceptions, both inside a clamped launch (clamp = 1): a flagged row whose fp32 step leaves the joint limits keeps the
     *     fp32 step -- where it lands is decided by the clamp, not by the last digits of the solve; and in the K - 1 iterations in
     *     front of the LAST one (whose iterates are intermediates; both kernel shapes) a flagged row is re-solved only when the estimate also
     *     exceeds a thousandth of the residual norm the step reduces -- an intermediate step has to be accurate relative to its
     *     residual; the last iteration, a K = 1 launch and every iteration of an early-out launch keep the absolute bar.  Measured
     *     (tests/test_gpu_lean_parity.py): the relative gate declines the re-solve on 0.3 % of a planner's rows (7 - 15 % of independent
     *     random 7-joint configurations) and moves the K-step iterate by <= K (1e-3 |e_s| + 1e-4) (1 + 2 |step| / sigma_min) in scaled
     *     task space; a row that is never flagged is not moved at all.
     *   CPPF_SOLVER_F64: every row in double precision (J J^T, its factorisation, the substitutions and J^T y): the exactly
     *     solved step of the fp32 Jacobian on every row, clamped or not (task-space difference to the fp64 oracle <= 6e-7).  A verification mode: every
     *     wavefront runs eight re-solve rounds per iteration (~10x the iteration time).
     *   CPPF_SOLVER_F32: fp32 only, no gate (the round-2 behaviour; up to 6e-2 off in task space on near-singular rows). */
    int32_t solver;
    /* Tolerance of CPPF_SOLVER_AUTO's gate in the scaled task-space units of the residual (rad * alpha_rotation, m *
     * alpha_position); 0 = CPPF_SOLVER_GATE_DEFAULT. */
    float solver_gate;
} cppf_lm_params;

#define CPPF_SOLVER_AUTO 0
#define CPPF_SOLVER_F64 1
#define CPPF_SOLVER_F32 2
#define CPPF_SOLVER_GATE_DEFAULT 1e-5f

#define CPPF_SHAPE_AUTO 0
#define CPPF_SHAPE_ROW 1
#define CPPF_SHAPE_QUAD 2

/* Optional outputs of the fused launch; any pointer may be NULL. */
typedef struct cppf_lm_outputs {
    float* x_out;         /* [n, d]    x after K steps (may alias x_in) */
    float* J_out;         /* [n, 6, d] SCALED Jacobian at the last linearisation point (optimization.py:77-80, 90-92) */
    float* e_out;         /* [n, 6]    SCALED pose error there, rows [roll pitch yaw x y z] (optimization_utils.py:806) */
    float* pos_err_m;     /* [n]  ||t_target - t(x_out)||_2            (evaluation_utils.py:134-136) */
    float* rot_err_rad;   /* [n]  geodesic quaternion distance at x_out (evaluation_utils.py:139-141) */
    uint8_t* self_mask;   /* [n]  min self distance < 0 at x_out        (collision_detection.py:52-69) */
    uint8_t* env_mask;    /* [n]  OR over obstacles of min dist < 0     (collision_detection.py:27-49) */
    uint8_t* jlim_mask;   /* [n]  joint within padding of a limit       (search.py:25-52) */
    float* ext_cost;      /* [n]  100*jlim + 1000*env + 1000*self        (search.py:14-15, 146-150) */
    float* min_self;      /* [n]  min over pairs of the signed distance (+inf when no pairs) */
    float* min_env;       /* [n]  min over obstacles and capsules       (+inf when no obstacles) */
    float* seed_summary;  /* [S,8] the reduction of cppf_seed_summary over each seed's W rows, produced by the same launch
                           * when W is 64, 128 or 256 (a 256-row workgroup then holds whole seeds); for any other W the entry point runs the
                           * separate reduction kernel afterwards, which needs x_out, pos_err_m, rot_err_rad, the three masks
                           * and ext_cost to be non-NULL.  Implies the collision stage. */
    int32_t* n_iters;     /* [n]  LM steps actually applied to the row (< n_steps only with the early-out tolerances) */
} cppf_lm_outputs;

typedef struct cppf_robot cppf_robot; /* opaque: host copy of the description + launch state for one device */

int cppf_abi_version(void);
/* sha256 prefix of the sources this binary was compiled from (cppflow_amd/build.py:source_hash); the Python binding refuses a
 * library whose id differs from the sources next to it. */
const char* cppf_build_id(void);
const char* cppf_last_error(void);

/* Replaces jrl.robots.get_robot(name) (cppflow/data_type_utils.py:197): validates and binds a description to `device`. */
int cppf_robot_create(const cppf_robot_desc* desc, int device, cppf_robot** out);
void cppf_robot_destroy(cppf_robot* robot);
int cppf_robot_ndof(const cppf_robot* robot);
/* >= 0: index of the robot-specialised kernel set (generated tables, csrc/robots_gen.h) this handle runs; -1: the generic
 * kernels, driven by the description in the kernel-argument segment. */
int cppf_robot_specialization(const cppf_robot* robot);
/* Compile-time tables for a description that matches none of the generated ones (any real robot a caller brings): the fused,
 * collision and quad kernels are compiled for THIS robot with hipRTC -- chain constants become literals, capsule end points
 * stay in registers, exactly as for the shipped robots -- and cached on disk under `cache_dir` (NULL: $CPPF_CACHE_DIR, else
 * $HOME/.cache/cppflow_amd) keyed by a hash of the description and of the kernel sources, so that later processes load the code
 * object without compiling.  Results are bit-identical to the generic kernels.  (The generic fused kernel stages capsule end
 * points in LDS, 6 KB per capsule: at 12 joints x 24 capsules that no longer fits a compute unit and the fused launch with
 * collision outputs returns CPPF_ERR_UNSUPPORTED -- such a robot must be specialised.)  After success cppf_robot_specialization()
 * returns CPPF_SPECIALIZATION_RTC.  Stands in for jrl.robots.get_robot returning a robot class with baked-in kinematics
 * (cppflow/data_type_utils.py:197).  A no-op (CPPF_OK) for a handle that already runs a generated table. */
#define CPPF_SPECIALIZATION_RTC 1000
int cppf_robot_specialize(cppf_robot* robot, const char* cache_dir);
/* (Test and tuning hooks live in cppflow_hip_debug.h; none of them is process-wide.) */

/* Replaces Problem.obstacles_cuboids / obstacles_Tcuboids (cppflow/data_type_utils.py:87-145).
 * cuboids [O,6] = (-sx/2,-sy/2,-sz/2, sx/2,sy/2,sz/2); Rt [O,12] = rotation row-major (9) then translation (3), HOST
 * pointers.  Only axis-aligned cuboids are accepted (R = I), as the reference asserts at data_type_utils.py:108. */
int cppf_set_obstacles(cppf_robot* robot, int n_obs, const float* cuboids, const float* Rt);

/* Padded joint limits l + eps / u - eps of joint_limit_almost_violations_3d (cppflow/search.py:46-51), HOST [d] each. */
int cppf_set_joint_limit_padding(cppf_robot* robot, const float* lo_padded, const float* hi_padded);

/* Robot.forward_kinematics(x) -> [n,7] = [x y z qw qx qy qz] (call sites cppflow/optimization_utils.py:811,
 * cppflow/evaluation_utils.py:115) */
int cppf_forward_kinematics(const cppf_robot* robot, const float* x, int n, float* poses, void* stream);

/* Robot.jacobian(x) -> [n,6,d], rows 0:3 angular, 3:6 linear, world frame (call site cppflow/optimization.py:74) */
int cppf_jacobian(const cppf_robot* robot, const float* x, int n, float* J, void* stream);

/* get_6d_pose_errors(robot, x, target_poses) -> (e [n,6], current_poses [n,7]); cppflow/optimization_utils.py:802-820 */
int cppf_pose_errors(const cppf_robot* robot, const float* x, const float* target, int S, int W, float* e,
                     float* current_poses, void* stream);

/* clamp_to_joint_limits(robot, x): in place; cppflow/optimization_utils.py:823-833 */
int cppf_clamp_to_joint_limits(const cppf_robot* robot, float* x, int n, void* stream);

/* levenberg_marquardt_only_pose (cppflow/optimization.py:61-92) followed by clamp_to_joint_limits (:259), K times in one
 * launch, with the pose-error metrics of x_is_valid (cppflow/optimization_utils.py:847) and the collision masks / search
 * cost (cppflow/collision_detection.py:27-69, cppflow/search.py:146-150) of the result evaluated in the same launch. */
int cppf_lm_pose_steps(const cppf_robot* robot, const float* x_in, const float* target, int S, int W,
                       const cppf_lm_params* params, const cppf_lm_outputs* out, void* stream);

/* Several INDEPENDENT problems of the same robot in ONE launch of the same kernel (row shape).  The reference's entry point takes one
 * problem per call (run_lm_optimization, cppflow/optimization.py:376-426; its parallel_count replicates ONE target path); a planner
 * that serves several requests at once, or a GPU that holds only a shard of the candidate seeds (cppflow/planners.py:231-251 split
 * over 8 GPUs leaves 128 seeds = half a wavefront per SIMD on each), cannot fill the MI355X with one such launch: four small
 * launches in flight is all the hardware overlaps, and each costs one wavefront's full latency however few rows it has.  A batch
 * lays up to CPPF_MAX_BATCH problems end to end in one grid -- every workgroup belongs to exactly one problem, so each problem's
 * results are bit for bit those of cppf_lm_pose_steps on it alone -- and the launch is full width again.
 * Each item is what cppf_lm_pose_steps takes: x_in [S*W, d], target [W, 7] (items may have different targets, S and W), and its own
 * outputs (J_out / e_out / min_self / min_env are not available in a batch; seed_summary as in cppf_lm_pose_steps: in the launch
 * for W in {64, 128, 256}, else by one reduction launch per such item behind it, which needs the per-row outputs).
 * cppf_lm_batch_create copies the item descriptors to a device table owned by the batch object (the only allocation; the caller's
 * buffers stay the caller's, and must stay valid while the batch is used); cppf_lm_batch_launch is then one asynchronous launch on
 * `stream`, hipGraph-capturable, reading the robot's obstacles / joint-limit padding as they are at launch time.
 * cppf_lm_params.shape must be CPPF_SHAPE_AUTO or CPPF_SHAPE_ROW (a batch is the throughput shape by construction). */
typedef struct cppf_lm_batch_item {
    const float* x_in;   /* [S*W, d] */
    const float* target; /* [W, 7] */
    int32_t S, W;
    cppf_lm_outputs out;
} cppf_lm_batch_item;
typedef struct cppf_lm_batch cppf_lm_batch; /* opaque: device table of the items + the marshalled parameters */
int cppf_lm_batch_create(const cppf_robot* robot, int n_items, const cppf_lm_batch_item* items /* HOST */,
                         const cppf_lm_params* params, cppf_lm_batch** out);
int cppf_lm_batch_launch(const cppf_lm_batch* batch, void* stream); /* CPPF_ERR_INVALID once its robot has been destroyed */
void cppf_lm_batch_destroy(cppf_lm_batch* batch);                   /* also releases a robot that was destroyed before it (see "Ownership") */

/* qpaths_batched_self_collisions / qpaths_batched_env_collisions (cppflow/collision_detection.py:27-69) +
 * joint_limit_almost_violations_3d + q_costs_external (cppflow/search.py:25-52, 146-150) for q [S,W,d]. Outputs [S*W]. */
int cppf_collision_masks(const cppf_robot* robot, const float* q, int S, int W, uint8_t* self_mask, uint8_t* env_mask,
                         uint8_t* jlim_mask, float* ext_cost, float* min_self, float* min_env, void* stream);

/* Robot.self_collision_distances(x) -> [n, n_pairs] (call site cppflow/collision_detection.py:65) */
int cppf_self_collision_distances(const cppf_robot* robot, const float* x, int n, float* dists, void* stream);

/* Robot.env_collision_distances(x, cuboid, Tcuboid) -> [n, n_capsules] for ONE cuboid (HOST cuboid[6], Rt[12]);
 * call site cppflow/collision_detection.py:40 */
int cppf_env_collision_distances(const cppf_robot* robot, const float* x, int n, const float* cuboid, const float* Rt,
                                 float* dists, void* stream);

/* Robot.self_collision_distances_jacobian(x) -> [n, n_pairs, d] and Robot.env_collision_distances_jacobian(x, cuboid,
 * Tcuboid) -> [n, n_capsules, d] (jrl; call sites cppflow/optimization_utils.py:670, 710): d(distance)/dq with the
 * closest points held fixed on their links.  `dists` (same leading shape, may be NULL) receives the distances the
 * gradients belong to, so that a caller needs one launch for both. */
int cppf_self_collision_distances_jacobian(const cppf_robot* robot, const float* x, int n, float* jac, float* dists,
                                           void* stream);
int cppf_env_collision_distances_jacobian(const cppf_robot* robot, const float* x, int n, const float* cuboid,
                                          const float* Rt, float* jac, float* dists, void* stream);

/* calculate_pose_error_cm_deg's two per-row terms (cppflow/evaluation_utils.py:113-116) in metres / radians */
int cppf_pose_error_metrics(const cppf_robot* robot, const float* x, const float* target, int S, int W, float* pos_err_m,
                            float* rot_err_rad, void* stream);

/* Validity half of x_is_valid for every seed (cppflow/optimization_utils.py:845-884, evaluation_utils.py:29-75):
 * out [S,4] = max position error (cm), max rotation error (deg), max |revolute joint change| (deg),
 * max |prismatic joint change| (cm) over the seed's W waypoints. */
int cppf_seed_validity(const cppf_robot* robot, const float* x, const float* target, int S, int W, float* out,
                       void* stream);

/* Plan metrics of S joint-space paths at once -- the properties of `Plan` (cppflow/data_types.py:140-264), which the
 * reference evaluates one path at a time on the host: x [S*W,d], target [W,7] -> out [S,16] =
 *   [0] max, [1] mean positional error (cm)            (data_types.py:166-186, evaluation_utils.py:134-136)
 *   [2] max, [3] mean rotational error (deg)           (data_types.py:156-164, evaluation_utils.py:139-141)
 *   [4] mjac of the revolute joints (deg), [5] of the prismatic joints (cm)  (data_types.py:189-210, evaluation_utils.py:83-99)
 *   [6] path length of the revolute joints (rad), [7] of the prismatic joints (m)      (data_types.py:141-149)
 *   [8] number of (waypoint, joint) entries outside the joint limits       (evaluation_utils.py:16-27)
 *   [9] number of self-colliding, [10] of environment-colliding waypoints (sums of the given masks; 0 for a NULL mask)
 *   [11] ||q_init - q_path[0]||_2, 0 when q_init (DEVICE [d]) is NULL       (data_types.py:217-221)   [12..15] 0
 * self_mask / env_mask [S*W] are the outputs of cppf_collision_masks / cppf_lm_pose_steps for the same x (may be NULL). */
int cppf_plan_metrics(const cppf_robot* robot, const float* x, const float* target, int S, int W, const uint8_t* self_mask,
                      const uint8_t* env_mask, const float* q_init, float* out, void* stream);

/* Per-seed reduction of the per-row outputs of cppf_lm_pose_steps (no FK is repeated): out [S,8] =
 *   max position error (cm), max rotation error (deg), max |revolute joint change| (deg), max |prismatic joint change| (cm)
 *   -- the four quantities x_is_valid thresholds (cppflow/optimization_utils.py:861-884, evaluation_utils.py:29-75) --
 *   then # self-colliding, # env-colliding, # joint-limit-padding waypoints and the summed external cost (search.py:146-150).
 * 32 bytes per seed: the payload a multi-GPU run all-gathers every step (the per-row buffer is 15 B per row). */
int cppf_seed_summary(const cppf_robot* robot, const float* x, int S, int W, const float* ext_cost, const float* pos_err_m,
                      const float* rot_err_rad, const uint8_t* self_mask, const uint8_t* env_mask,
                      const uint8_t* jlim_mask, float* out, void* stream);

/* Constraints (cppflow/data_types.py:53-62; CLI values scripts/evaluate.py:51-56: 0.01 cm, 0.1 deg, 7 deg, 2 cm) and the two
 * collision switches of cppflow/config.py:23-24 that x_is_valid reads (cppflow/optimization_utils.py:889, 896). */
typedef struct cppf_constraints {
    float max_allowed_position_error_cm;
    float max_allowed_rotation_error_deg;
    float max_allowed_mjac_deg;
    float max_allowed_mjac_cm;
    int32_t self_collisions_ignored;
    int32_t env_collisions_ignored;
} cppf_constraints;

/* The seed selection of x_is_valid (cppflow/optimization_utils.py:856-909) over the per-seed summaries [S,8] of
 * cppf_seed_summary / cppf_lm_outputs.seed_summary -- of one GPU, or of every rank after the all-gather: a seed is valid
 * when its four maxima are strictly below the constraints (cppflow/evaluation_utils.py:29-75) and it has no self- /
 * environment-colliding waypoint.  out (DEVICE int32 [4]) = { first valid seed in order or -1, number of valid seeds,
 * seed of smallest summed external cost (first on ties), 0 }.  One small launch. */
int cppf_select_valid_seed(const cppf_robot* robot, const float* seed_summary, int S, const cppf_constraints* constraints,
                           int32_t* out, void* stream);
/* The same over what ONE all-gather of [n_groups, S_chunk, 8] buffers from n_chunks ranks leaves behind
 * (gathered [n_chunks, n_groups, S_chunk, 8]: the summaries of n_groups consecutive steps travel in one collective): group g
 * is the n_chunks * S_chunk seeds of step g in rank order, seed index = rank * S_chunk + s.  out: DEVICE int32 [n_groups, 4]. */
int cppf_select_valid_seed_gathered(const cppf_robot* robot, const float* gathered, int n_chunks, int n_groups, int S_chunk,
                                    const cppf_constraints* constraints, int32_t* out, void* stream);

/* cppflow/lm_hyper_parameters.py:14-56: the fields the coupled step reads (values of ALT_LOSS_V2_1_DIFF at :86-118) */
typedef struct cppf_full_params {
    float lm_lambda;
    float alpha_position, alpha_rotation;        /* pose block (only when use_pose) */
    float alpha_differencing;                    /* 0.00375 */
    float alpha_differencing_prismatic_scaling;  /* 1.0 */
    float alpha_virtual_configs;                 /* multiplies alpha_differencing */
    float alpha_self_collision, alpha_env_collision; /* 0.01, 0.01 */
    int32_t use_pose, use_differencing, use_virtual_configs, n_virtual_configs, use_self_collisions, use_env_collisions;
    /* The "satisfied" row options of LmResidualFns.get_r_and_J (off in both presets; all zero = off).
     *   pose_do_scale_down_satisfied (cppflow/optimization_utils.py:514-533, :288-333): a pose row whose unscaled |error| is below
     *     pose_threshold_rad (rotation rows) / pose_threshold_m (position rows) is multiplied by pose_scale_down, r and J alike.
     *     The caller forms the thresholds (the reference: pose_ignore_satisfied_threshold_scale x the constraint).
     *   differencing_mode (:548-606): 1 = differencing_do_ignore_satisfied -- rows with |joint change| <= threshold leave the system,
     *     the others are shifted towards zero by the threshold (filter_rows_from_r_J_differencing, :736-768; no prismatic scaling
     *     in this mode, :601); 2 = differencing_do_scale_satisfied -- rows below the threshold are multiplied by
     *     differencing_scale_down, the others optionally shifted (:352-398).  Thresholds: rad for revolute, m for prismatic
     *     joints (the reference: max_allowed_mjac - margin).  A step with differencing_mode != 0 is eliminated by the
     *     one-lane-per-trajectory kernel (the couplings between consecutive waypoints differ from row to row). */
    int32_t pose_do_scale_down_satisfied;
    float pose_threshold_m, pose_threshold_rad, pose_scale_down;
    int32_t differencing_mode;
    float differencing_threshold_rad, differencing_threshold_m, differencing_scale_down;
    int32_t differencing_shift_invalid_to_threshold;
} cppf_full_params;

/* levenberg_marquardt_full (cppflow/optimization.py:95-144) with the residual / Jacobian of LmResidualFns.get_r_and_J
 * (cppflow/optimization_utils.py:486-731, the "satisfied" scaling / filtering options included): one coupled LM step for each
 * of S trajectories x_in [S*W, d] (the reference: one trajectory, :128).
 * target [W,7]; virtual_configs [S*W, d] or NULL (= x_in, which is what the loop sets at optimization.py:253).
 * Obstacles are those of cppf_set_obstacles.  The normal matrix is never formed densely: it is block-tridiagonal and is
 * eliminated per trajectory.  Workspace (device): work_blocks [S*W * (d(d+1)/2 + d)], work_G [S*W * d*d],
 * work_y [S*W * d] floats.  x_out [S*W, d] must not alias x_in. */
int cppf_lm_full_step(const cppf_robot* robot, const float* x_in, const float* target, const float* virtual_configs, int S,
                      int W, const cppf_full_params* params, float* work_blocks, float* work_G, float* work_y,
                      float* x_out, void* stream);

/* _get_mjacs (cppflow/search.py:100-125): q [k,T,d] -> mjacs [k,k,T-1], mjacs[i,j,t] = max over joints of
 * |wrap(s (q[i,t+1] - q[j,t]))| with s = prismatic_scaling on prismatic joints.  cppf_dp_search does not need it (it never
 * materialises the tensor); provided for callers of the reference helper. */
int cppf_mjacs(const cppf_robot* robot, const float* q, int k, int T, float prismatic_scaling, float* mjacs, void* stream);

/* dp_search(robot, q, ...) of cppflow/search.py:128-191 given the external cost matrix q_costs_external [k,T]
 * (search.py:146-150, the `ext_cost` output of cppf_collision_masks / cppf_lm_pose_steps): the min-max dynamic programme
 * over the k candidate paths q [k,T,d] with mjacs as in search.py:100-125 (prismatic deltas scaled by `prismatic_scaling`,
 * 5.0 in the reference), then the back-trace.  Outputs best_path [T,d] and best_idx [T] (which candidate each waypoint
 * came from).  The caller supplies the workspace (device): work_qT [T*k*d] floats, work_costsT [T*k] floats (on return:
 * the cost table, time-major), work_memoT [T*k] int32.
 * mode = CPPF_DP_AUTO: for k <= 1024 (the reference plans with k = 175 and re-plans with 300, cppflow/planners.py:47, 253-258; eight
 *   ranks gather 1024) the whole recurrence runs in ONE resident launch -- one destination per wavefront (k <= 64) or four per
 *   workgroup, at most 256 workgroups; the cost row of step t-1 is handed from workgroup to workgroup as write-through words that are
 *   their own flags, no grid barrier -- else one small launch per waypoint; no host synchronisation either way.
 * mode = CPPF_DP_RESIDENT / CPPF_DP_LAUNCHES force one form for THIS call (no handle state involved).
 * The resident form needs its <= 256 workgroups on the device together.  The entry point holds the grid against what the device can
 * hold of that kernel (occupancy x compute units) BEFORE launching: a device that cannot (partitioned, CU-masked, smaller) gets one
 * launch per waypoint under CPPF_DP_AUTO and CPPF_ERR_UNSUPPORTED for a forced CPPF_DP_RESIDENT.  Work already on the device can
 * still delay workgroups (another resident search on a second stream, a full-width fused launch): the waits are bounded, and if one
 * expires the call has still returned CPPF_OK -- it is asynchronous -- but best_idx[*] = -1 and best_path is NaN: repeat the call
 * with CPPF_DP_LAUNCHES (cppflow_amd.search.dp_search does).  Do not run two resident searches of > 512 candidates concurrently. */
#define CPPF_DP_AUTO 0
#define CPPF_DP_RESIDENT 1
#define CPPF_DP_LAUNCHES 2
int cppf_dp_search(const cppf_robot* robot, const float* q, const float* ext_cost, int k, int T, float prismatic_scaling,
                   float* work_qT, float* work_costsT, int32_t* work_memoT, float* best_path, int32_t* best_idx, int mode,
                   void* stream);

/* The same dynamic programme for k <= 256 from a precomputed transition table (the mjacs tensor of search.py:100-125, which
 * the reference materialises as well): the table is filled by the whole chip, the recurrence then runs on ONE compute unit
 * (no hand-off between workgroups: one workgroup barrier per waypoint, the table streamed through registers two steps ahead)
 * and the argmins are recovered in a third, parallel launch.  Same outputs, bit for bit, as cppf_dp_search.  Measured at
 * T = 256: 560 vs 720 us at the reference's k = 175, 378 vs 734 us at k = 128, 168 vs 422 us at k = 64; slower at k = 256
 * (one compute unit streams ~80-90 GB/s).  Extra workspace: work_table, cppf_dp_table_floats(k, T) = ((T-1) * k + 32) *
 * roundup(k, 64) floats (34 MB at k = 175, T = 256).  T <= 65536. */
int cppf_dp_table_floats(int k, int T, size_t* n_floats);
int cppf_dp_search_tabled(const cppf_robot* robot, const float* q, const float* ext_cost, int k, int T, float prismatic_scaling,
                          float* work_qT, float* work_costsT, int32_t* work_memoT, float* work_table, float* best_path,
                          int32_t* best_idx, void* stream);

/* ---- seed sharding across the GPUs of one node: RCCL behind the C ABI ---------------------------------------
 * The reference is one process on one device (no collective anywhere in its tree); the MI355X build shards the candidate seeds
 * of cppflow/planners.py:231-251 over the GPUs and needs ONE collective: an all-gather of each rank's packed per-row / per-seed
 * outputs, so that every rank holds what cppflow/search.py:146-151 consumes.  The library loads RCCL lazily (dlopen of
 * librccl.so.1: the copy already in the process -- e.g. PyTorch's -- or the ROCm one), so a caller that never shards needs no
 * RCCL at all.  Two process models:
 *   - one process (or thread) per GPU: rank 0 calls cppf_comm_unique_id, ships the 128 bytes to the other ranks by any means,
 *     every rank calls cppf_comm_init_rank;
 *   - one process, all GPUs: cppf_comm_init_all creates one communicator per listed device.
 * cppf_allgather_bytes is asynchronous on `stream` (the stream of the communicator's device); with cppf_comm_init_all, issue
 * the calls for all communicators between cppf_comm_group_begin / cppf_comm_group_end. */
typedef struct cppf_comm cppf_comm;
#define CPPF_COMM_ID_BYTES 128

/* CPPF_OK when RCCL can be loaded into this process (dlopen + every symbol the library uses), else CPPF_ERR_UNSUPPORTED with
 * the reason in cppf_last_error(): lets every rank of a job agree on a transport BEFORE anyone enters a collective. */
int cppf_comm_available(void);
int cppf_comm_unique_id(void* id_out /* CPPF_COMM_ID_BYTES */);
int cppf_comm_init_rank(const void* id, int rank, int world, int device, cppf_comm** out);
int cppf_comm_init_all(int n_devices, const int* devices, cppf_comm** out /* [n_devices] */);
int cppf_comm_rank(const cppf_comm* comm);
int cppf_comm_world(const cppf_comm* comm);
/* recv [world * bytes_per_rank] <- every rank's send [bytes_per_rank] in rank order (DEVICE pointers; the packed buffer of
 * cppf_lm_outputs: ext_cost | pos_err_m | rot_err_rad | the three masks, or the [S,8] per-seed summaries). */
int cppf_allgather_bytes(cppf_comm* comm, const void* send, void* recv, size_t bytes_per_rank, void* stream);
int cppf_comm_group_begin(void);
int cppf_comm_group_end(void);
void cppf_comm_destroy(cppf_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* CPPFLOW_HIP_H */

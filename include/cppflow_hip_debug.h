/*
 * cppflow_hip_debug.h -- test and tuning hooks of libcppflow_hip.so.  NOT part of the drop-in boundary (cppflow_hip.h): nothing
 * here replaces an interface of the reference; the hooks exist so that tests can force every code path and measurement scripts
 * can A/B a dispatch decision.  Every switch belongs to ONE robot handle ("no global mutable state besides the
 * communicator"): two handles on two threads can hold different settings, and a test that flips a switch changes nothing for
 * any other handle in the process.
 */
#ifndef CPPFLOW_HIP_DEBUG_H
#define CPPFLOW_HIP_DEBUG_H

#include "cppflow_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* keys of cppf_debug_set / cppf_debug_get; CPPF_TUNE_DEFAULT as the value restores the built-in default */
#define CPPF_TUNE_DEFAULT (-2147483647 - 1)
/* non-zero: every launch of the handle runs the generic kernels (description in the kernel-argument segment) even when a
 * generated table or a run-time-specialised module exists.  Default 0. */
#define CPPF_TUNE_FORCE_GENERIC 0
/* cppf_lm_full_step eliminates in parallel over the waypoints (cyclic reduction, one workgroup per trajectory) when S*W <= n
 * rows (x 0.5 at d = 8; and W <= 512, d <= 8, no pose block), waypoint after waypoint from both ends of the path (eight
 * trajectories per wavefront) otherwise.  Default -1 = the measured crossovers: 131072 rows with the state in LDS (W <= 256),
 * 49152 with it in the workspace. */
#define CPPF_TUNE_PCR_MAX_ROWS 1
/* CPPF_SHAPE_AUTO runs four lanes per row up to n rows unless a per-seed summary is requested.  Default 16384 = one wavefront
 * of that shape per SIMD, the measured crossover. */
#define CPPF_TUNE_QUAD_MAX_ROWS 2
/* 0: cppf_dp_search in CPPF_DP_AUTO mode issues one launch per waypoint instead of the single resident launch (k <= 1024); 2: the
 * resident launch for 257 ... 512 candidates is dp_resident_kernel (512 lanes, four destinations per lane) instead of the 1 024-lane
 * form of dp_persistent4_kernel (the A/B of scripts/dp_bench.py).  Default 1. */
#define CPPF_TUNE_DP_PERSISTENT 3
/* 0: cppf_lm_full_step eliminates with one wavefront per trajectory, first waypoint to last (cross-lane reads through the LDS
 * pipe), instead of eight trajectories per wavefront, one block row per lane (DPP), from both ends of the path; d <= 8, beyond
 * the parallel-in-time range.  Default 1. */
#define CPPF_TUNE_FULL_ROWS 4
/* parallel-in-time elimination of cppf_lm_full_step, W <= 256: 0 = state in the caller's workspace (as for W > 256), 1 = in
 * LDS, one lane per waypoint, 2 (default) = in LDS, two half-workgroups per waypoint. */
#define CPPF_TUNE_PCR_LDS 5
/* non-zero: cppf_lm_full_step WITH the pose block (rank-deficient d x d blocks) goes through the row-per-lane Gauss-Jordan
 * kernels as well instead of the one-lane-per-trajectory Cholesky kernel.  Default 0. */
#define CPPF_TUNE_ROWS_POSE 6
/* non-zero: J J^T of the four-lanes-per-row shape by v_mfma_f32_4x4x1_16b_f32 in the robot-specialised instantiations (the
 * measured comparison of DESIGN.md section 4.2).  Default 0. */
#define CPPF_TUNE_QUAD_MFMA 7
/* KB of dynamic LDS a row-shape fused launch of at most 128 workgroups claims for nothing but residency control: with 42 (the
 * default) only two of its workgroups fit on one compute unit (12.5 KB static + 42 KB, 160 KB per unit), so that four such launches
 * in flight spread over the chip instead of stacking on the units the dispatcher fills first.  0 = no claim. */
#define CPPF_TUNE_SPREAD_KB 8
/* log2 of the re-reads after which a wait of the resident cppf_dp_search gives up (default 22: a few seconds).  0 makes every wait
 * that does not find its word at the first read expire at once: how the tests reach the timeout path (best_idx = -1) and the
 * caller's fall-back to one launch per waypoint. */
#define CPPF_TUNE_DP_SPIN_LOG2 9
/* parts per million of the (scaled) residual norm that the estimated error of a LEAN iteration's fp32 step must exceed, besides the
 * absolute gate, before the row is re-solved in double precision (CPPF_SOLVER_AUTO, both kernel shapes; the last iteration of a launch, the
 * early-out iterations and a K = 1 launch keep the absolute gate alone).  Default 1000 (kGateRel, csrc/kernels_chain.h); 0 = the
 * absolute gate in every iteration. */
#define CPPF_TUNE_GATE_REL_PPM 10
/* >= 0: the number of compute units cppf_dp_search holds its resident launch's grid against (occupancy x compute units >= grid, else
 * one launch per waypoint in CPPF_DP_AUTO and CPPF_ERR_UNSUPPORTED for a forced CPPF_DP_RESIDENT) instead of the device's own: how
 * the tests reach that decision on a full MI355X.  Default -1 = the device's. */
#define CPPF_TUNE_CU_COUNT 11
/* Fair-share pacing of a fused launch that has the chip to itself (kernels_fused.h: lm_pace): every wavefront takes one of four
 * priorities by its progress against a clock schedule, so that the four wavefronts of a SIMD finish together instead of in age
 * order.  0 (default) off; > 0: the schedule in 10 ns ticks per LM iteration; < 0: the built-in estimate for this robot.  For a
 * caller that steps ONE full-size batch in a dependency chain (one stream: -6 %); launches that overlap on several streams lose
 * 0 - 5 % with it.  Only launches of ONE resident round of four wavefronts per SIMD are paced (at least three quarters of the chip's
 * slots, not more than all of them; kernels that hold three per SIMD -- chains beyond 7 joints -- never).  No effect on results. */
#define CPPF_TUNE_LM_PACE 12
#define CPPF_TUNE_COUNT 13

int cppf_debug_set(cppf_robot* robot, int key, int value);
int cppf_debug_get(const cppf_robot* robot, int key, int* value);

/* Needs no GPU: generates the table for `desc`, compiles it with hipRTC and writes the cache entry (everything
 * cppf_robot_specialize does before it loads the code object); returns the error of the load stage (CPPF_ERR_HIP) on a machine
 * without a device and CPPF_OK never -- look for the cache file. */
int cppf_debug_rtc_compile(const cppf_robot_desc* desc, const char* cache_dir);

/* offsetof(FusedArgs, single) as the DEVICE code was compiled with it: the fused kernel reads the problem of a plain launch at this
 * offset from its kernel-argument segment (csrc/kernels_fused.h: fused_item).  tests/test_abi.py holds it against the offset the
 * code object's own metadata gives the kernel's fourth argument. */
int cppf_debug_fused_single_offset(void);

/* (Test-only KERNELS are not in this library: tests/native/ holds them as a translation unit of its own.) */

#ifdef __cplusplus
}
#endif
#endif /* CPPFLOW_HIP_DEBUG_H */

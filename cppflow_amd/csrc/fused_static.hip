// fused_static.hip -- the row-shape fused kernel (lm_fused_kernel, kernels_fused.h) of the SHIPPED robots, as its own translation
// unit of libcppflow_hip.so (gfx950 only) because it is compiled with another machine scheduler than the rest of the library:
//     -O2 -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -enable-post-misched=0
// The default strategy orders a kernel for the smallest register footprint first; on this kernel that leaves dependent
// instructions back to back where independent ones were available.  Scheduled for instruction-level parallelism the same code
// (same instructions, same results bit for bit: -ffp-contract=off, every FMA explicit) takes 124 instead of 118 VGPRs -- still
// four wavefronts per SIMD -- and the headline workload steps in 37.5 instead of 38.6 us, Fetch's in 18.7 instead of 19.4, a
// 32 768-row shard in 6.85 instead of 7.0 (profiles/README.md).  The post-register-allocation scheduler then moves instructions
// again, for latencies that four resident wavefronts hide anyway: without it the full-size launch gains another 1 % (36.9 -> 36.5
// us per step; with the DEFAULT pre-RA strategy and no post-RA pass it is as fast, 36.3, but the shard-sized launch falls back to
// 7.2 us); -O2 instead of -O3 measured 0.5 % better again on this unit.  Library-wide the ILP switch costs registers where they matter
// (the coupled step's block kernel 117 -> 167 VGPRs, scratch in three generic instantiations, the collision kernel one occupancy
// step), hence one translation unit for the kernel that gains.  cppflow_hip.hip calls launch_fused_static() for a handle whose
// description equals a generated table; everything else (generic and run-time-specialised kernels, every other stage) stays there.
#include <hip/hip_runtime.h>

#include <cmath>

#include "lmik_device.h"
#include "robots_gen.h"

using namespace cppf;

namespace {

#ifndef CPPF_BLOCK
#define CPPF_BLOCK 256
#endif
constexpr int kBlock = CPPF_BLOCK;
#ifndef CPPF_WAVES_LM
#define CPPF_WAVES_LM 2
#endif
#ifndef CPPF_WAVES_COLL
#define CPPF_WAVES_COLL 2
#endif

#include "kernels_chain.h"
#include "kernels_collision.h"
#include "kernels_fused.h"

template <class Type>
void launch_one(int coll, unsigned grid, size_t lds, hipStream_t st, const FusedArgs& args) {
    using RB = StaRobot<Type>;
    if (coll == 2)
        hipLaunchKernelGGL((lm_fused_kernel<RB, 2>), dim3(grid), dim3(kBlock), lds, st, args.ch, args.co, args.prm, args.single, args.table);
    else if (coll == 1)
        hipLaunchKernelGGL((lm_fused_kernel<RB, 1>), dim3(grid), dim3(kBlock), lds, st, args.ch, args.co, args.prm, args.single, args.table);
    else
        hipLaunchKernelGGL((lm_fused_kernel<RB, 0>), dim3(grid), dim3(kBlock), lds, st, args.ch, args.co, args.prm, args.single, args.table);
}

}  // namespace

namespace cppf {

// coll: 0 = no collision stage, 1 = masks / cost, 2 = masks / cost / signed minimum distances (lm_fused_kernel's COLL).  `args` is the
// kernel's whole argument segment: one problem (args.single) or a device table of them (args.table, cppf_lm_batch_launch).
// Returns false for a static_id this unit has no table for (nothing launched); launch errors are the caller's hipGetLastError.
bool launch_fused_static(int static_id, int coll, unsigned grid, size_t lds, hipStream_t st, const FusedArgs& args) {
    switch (static_id) {
#define CPPF_STATIC_LAUNCH(idx, Type)                  \
    case idx:                                          \
        launch_one<Type>(coll, grid, lds, st, args);   \
        return true;
        CPPF_FOR_EACH_STATIC_ROBOT(CPPF_STATIC_LAUNCH)
#undef CPPF_STATIC_LAUNCH
        default:
            return false;
    }
}

int fused_static_block() { return kBlock; }

}  // namespace cppf

// one-kernel translation unit for work on full_blocks_kernel (see scripts/README.md)
#include <hip/hip_runtime.h>
#include <cmath>
#include "lmik_device.h"
#include "robots_gen.h"
using namespace cppf;
namespace dev {
constexpr int kBlock = 256;
#define CPPF_WAVES_LM 2
#define CPPF_WAVES_COLL 2
#include "kernels_chain.h"
#include "kernels_collision.h"
#include "kernels_fused.h"
#include "kernels_coupled.h"
}
using namespace dev;
#define FB(RB) template __global__ void dev::full_blocks_kernel<RB, full_blocks_occ<RB>()>(const ChainK, const CollK, const FullK, const float*, const float*, const float*, float*, float*);
FB(StaRobot<gen::Panda>) FB(StaRobot<gen::Fetch>) FB(StaRobot<gen::FetchArm>) FB(DynRobot<6>) FB(DynRobot<7>) FB(DynRobot<8>)

// Developer TU: only the Panda / Chain12 quad-shape instantiations, for quick ISA inspection
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize \
//         -Icppflow_amd/csrc --cuda-device-only -c scripts/quad_dev.hip -o build_var/quad_dev.co
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
#include "kernels_quad.h"
}  // namespace dev
using namespace dev;

template __global__ void dev::lm_quad_kernel<StaRobot<gen::Panda>, 1, false>(const ChainK, const CollK, const LmK, const float*, const float*, const cppf_lm_outputs, const uint4*);
template __global__ void dev::lm_quad_kernel<StaRobot<gen::Chain12>, 1, true>(const ChainK, const CollK, const LmK, const float*, const float*, const cppf_lm_outputs, const uint4*);
template __global__ void dev::lm_quad_kernel<StaRobot<gen::Chain12>, 1, false>(const ChainK, const CollK, const LmK, const float*, const float*, const cppf_lm_outputs, const uint4*);
template __global__ void dev::lm_quad_kernel<StaRobot<gen::Chain12>, 0, false>(const ChainK, const CollK, const LmK, const float*, const float*, const cppf_lm_outputs, const uint4*);
template __global__ void dev::lm_quad_kernel<DynRobot<11>, 1, false>(const ChainK, const CollK, const LmK, const float*, const float*, const cppf_lm_outputs, const uint4*);
template __global__ void dev::lm_quad_kernel<DynRobot<12>, 1, false>(const ChainK, const CollK, const LmK, const float*, const float*, const cppf_lm_outputs, const uint4*);
template __global__ void dev::lm_quad_kernel<DynRobot<12>, 0, false>(const ChainK, const CollK, const LmK, const float*, const float*, const cppf_lm_outputs, const uint4*);
template __global__ void dev::lm_quad_kernel<DynRobot<10>, 1, false>(const ChainK, const CollK, const LmK, const float*, const float*, const cppf_lm_outputs, const uint4*);

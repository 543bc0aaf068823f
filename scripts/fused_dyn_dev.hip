#include <hip/hip_runtime.h>
#include <cmath>
#include "../include/cppflow_hip.h"
#include "lmik_device.h"
#include "robots_gen.h"
using namespace cppf;
namespace dev {
constexpr int kBlock = 256;
#ifndef CPPF_WAVES_LM
#define CPPF_WAVES_LM 2
#endif
#define CPPF_WAVES_COLL 2
#include "kernels_chain.h"
#include "kernels_collision.h"
#include "kernels_fused.h"
}
using namespace dev;
#ifndef DEV_D
#define DEV_D 7
#endif
template __global__ void dev::lm_fused_kernel<DynRobot<DEV_D>, 1>(const ChainK, const CollK, const LmK, const BatchItemK, const void*);

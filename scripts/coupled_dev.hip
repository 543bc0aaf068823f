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
template __global__ void dev::full_solve_wave_kernel<7>(const ChainK, const FullK, const float*, const float*, const float*, float*, float*, float*);
template __global__ void dev::full_blocks_kernel<StaRobot<gen::Panda>>(const ChainK, const CollK, const FullK, const float*, const float*, const float*, float*, float*);
template __global__ void dev::full_blocks_kernel<DynRobot<7>>(const ChainK, const CollK, const FullK, const float*, const float*, const float*, float*, float*);
template __global__ void dev::full_rows_eliminate_kernel<7>(const FullK, const uint32_t, const float*, float*, float*);
template __global__ void dev::full_rows_substitute_kernel<7>(const FullK, const uint32_t, const float*, const float*, const float*, const float*, float*);
template __global__ void dev::full_solve_pcr_kernel<7, 256, true>(const ChainK, const FullK, const float*, const float*, float*, float*, float*);
template __global__ void dev::full_rows_eliminate_kernel<12>(const FullK, const uint32_t, const float*, float*, float*);
template __global__ void dev::full_rows_substitute_kernel<12>(const FullK, const uint32_t, const float*, const float*, const float*, const float*, float*);
template __global__ void dev::full_solve_pcr_kernel<7, 512, true, true>(const ChainK, const FullK, const float*, const float*, float*, float*, float*);

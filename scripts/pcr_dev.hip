// one-kernel translation unit for work on full_solve_pcr_kernel (see scripts/README.md): hipcc -shared, then kernel_resources.py
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
#define PCR(D, BS, L, S) template __global__ void dev::full_solve_pcr_kernel<D, BS, L, S>(const ChainK, const FullK, const float*, const float*, float*, float*, float*);
PCR(7, 256, true, false) PCR(7, 512, true, true) PCR(7, 512, false, false) PCR(7, 256, false, false)
PCR(8, 256, true, false) PCR(8, 512, true, true) PCR(8, 512, false, false) PCR(8, 256, false, false)

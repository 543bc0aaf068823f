#include <hip/hip_runtime.h>
#include <cmath>
#include "lmik_device.h"
using namespace cppf;
namespace dev {
constexpr int kBlock = 256;
#include "kernels_dp.h"
}
using namespace dev;
template __global__ void dev::dp_table_kernel<7>(const float*, int, int, int, uint32_t, float, uint32_t*);
template __global__ void dev::dp_chain_kernel<192>(const uint32_t*, const float*, int, int, float*);
template __global__ void dev::dp_chain_kernel<64>(const uint32_t*, const float*, int, int, float*);
template __global__ void dev::dp_chain_kernel<128>(const uint32_t*, const float*, int, int, float*);
template __global__ void dev::dp_chain_kernel<256>(const uint32_t*, const float*, int, int, float*);

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
#define P4(D, SRC) template __global__ void dev::dp_persistent4_kernel<D, SRC>(const float*, const float*, int, int, uint32_t, float, float*, int32_t*, uint32_t);
P4(7, 256) P4(7, 512) P4(8, 512) P4(12, 512) P4(12, 256)
#define P4N(D, SRC, NS) template __global__ void dev::dp_persistent4_kernel<D, SRC, NS>(const float*, const float*, int, int, uint32_t, float, float*, int32_t*, uint32_t);
P4N(7, 512, 2) P4N(8, 512, 2) P4N(9, 512, 2) P4N(10, 512, 2) P4N(12, 512, 2)

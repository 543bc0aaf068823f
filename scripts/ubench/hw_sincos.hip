// accuracy of the hardware v_sin_f32 / v_cos_f32 (input in revolutions) against double sin/cos on [-4, 4]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, float* s, float* c, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float r = x[i] * 0.15915494309189535f;
    s[i] = __builtin_amdgcn_sinf(r);
    c[i] = __builtin_amdgcn_cosf(r);
}
int main() {
    const int n = 1 << 22;
    std::vector<float> hx(n), hs(n), hc(n);
    for (int i = 0; i < n; ++i) hx[i] = -4.f + 8.f * (float)i / (float)(n - 1);
    float *dx, *ds, *dc;
    hipMalloc(&dx, n * 4), hipMalloc(&ds, n * 4), hipMalloc(&dc, n * 4);
    hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, dx, ds, dc, n);
    hipMemcpy(hs.data(), ds, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hc.data(), dc, n * 4, hipMemcpyDeviceToHost);
    double es = 0, ec = 0, rs = 0;
    for (int i = 0; i < n; ++i) {
        es = fmax(es, fabs((double)hs[i] - sin((double)hx[i])));
        ec = fmax(ec, fabs((double)hc[i] - cos((double)hx[i])));
        rs += ((double)hs[i] - sin((double)hx[i])) * ((double)hs[i] - sin((double)hx[i]));
    }
    printf("max abs err: sin %.3e cos %.3e  rms sin %.3e\n", es, ec, sqrt(rs / n));
    return 0;
}

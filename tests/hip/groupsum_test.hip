#include <hip/hip_runtime.h>
#include <cstdio>
#include "cmf_kernels.hip.h"
using namespace cmfk;
__global__ void k(float* out) {
    int t = threadIdx.x;
    float v = (float)(t * t % 17) + 0.25f * t;
    out[t] = group_sum<8>(v); out[64 + t] = group_sum<16>(v); out[128 + t] = group_sum<32>(v); out[192 + t] = group_sum<64>(v);
    out[256 + t] = v;
}
int main() {
    float* d; (void)hipMalloc(&d, 320 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[320]; (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int gs[4] = {8, 16, 32, 64}; int bad = 0;
    for (int q = 0; q < 4; ++q) for (int t = 0; t < 64; ++t) {
        float ref = 0; int g0 = t / gs[q] * gs[q];
        for (int j = g0; j < g0 + gs[q]; ++j) ref += h[256 + j];
        if (fabsf(ref - h[64 * q + t]) > 1e-3f) { if (bad < 8) printf("GS %d lane %d got %f want %f\n", gs[q], t, h[64 * q + t], ref); ++bad; }
    }
    printf("bad=%d\n", bad);
    return 0;
}

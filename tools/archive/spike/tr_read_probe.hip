// Probe of ds_read_b64_tr_b16 (gfx950): fill an LDS tile [R][C] of 16-bit values with value = 256*row + col, let every
// lane issue one transposing read the way cdna_hip_programming.md T10 describes (lane 4q+p of a 16-lane group supplies
// the address of row q, columns 4p..4p+3; lane i receives column i of the 4 rows), and print what each lane got.
// hipcc --offload-arch=gfx950 -O3 tools/spike/tr_read_probe.hip -o /tmp/trp && /tmp/trp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned short u16;

__global__ void probe(uint32_t *out, int pitch_elems) {
    __shared__ __attribute__((aligned(16))) u16 tile[64 * 64];
    const int t = threadIdx.x;
    for (int i = t; i < 64 * 64; i += 64) tile[i] = (u16)(256 * (i / pitch_elems) + (i % pitch_elems));
    __syncthreads();
    const int g = t >> 4, li = t & 15, q = li >> 2, p = li & 3;
    // group g looks at rows 4g .. 4g+3, columns 0..15
    const unsigned addr = (unsigned)(uintptr_t)(&tile[(4 * g + q) * pitch_elems + 4 * p]);
    uint64_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[2 * t] = (uint32_t)v;
    out[2 * t + 1] = (uint32_t)(v >> 32);
}

int main() {
    uint32_t *d; (void)hipMalloc(&d, 64 * 2 * 4);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 64);
    std::vector<uint32_t> h(128);
    (void)hipMemcpy(h.data(), d, 128 * 4, hipMemcpyDeviceToHost);
    for (int t = 0; t < 64; ++t) {
        u16 e[4] = {(u16)(h[2 * t] & 0xFFFF), (u16)(h[2 * t] >> 16), (u16)(h[2 * t + 1] & 0xFFFF), (u16)(h[2 * t + 1] >> 16)};
        printf("lane %2d:", t);
        for (int k = 0; k < 4; ++k) printf("  (r%d,c%d)", e[k] / 256, e[k] % 256);
        printf("\n");
    }
    return 0;
}

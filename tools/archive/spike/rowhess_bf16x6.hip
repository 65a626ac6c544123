// Feasibility spike (NOT part of libcmfhip): the Hessian accumulation of row_hess_kernel<256> on the bf16 matrix pipe.
//   H = sum_j u_j u_j^T  over S rows u_j of 256 floats, one workgroup per H; every u_j is split exactly into three bf16
//   planes while it is staged; the contraction index j is the SLOW axis of the row-major LDS tile, so the k-contiguous
//   MFMA operands come from ds_read_b64_tr_b16 transposing reads; six cross products per block, 36 upper blocks.
// hipcc --offload-arch=gfx950 -O3 tools/spike/rowhess_bf16x6.hip -o /tmp/rh6 && /tmp/rh6
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__host__ __device__ inline float value_at(uint64_t r, uint64_t c) {
    uint64_t x = (r * 0x9E3779B97F4A7C15ull) ^ (c * 0xC2B2AE3D27D4EB4Full) ^ 0x1234567ull;
    x ^= x >> 33; x *= 0xFF51AFD7ED558CCDull; x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ull; x ^= x >> 33;
    return ((float)((x >> 40) + 1) * (1.0f / 16777217.0f) - 0.5f) * 0.25f;
}
__global__ void fill(float *U, int64_t n, int64_t cols) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) U[i] = value_at(i / cols, i % cols);
}
__device__ __forceinline__ u16 bf16_rn(float f) { unsigned u = __float_as_uint(f); u += 0x7FFFu + ((u >> 16) & 1u); return (u16)(u >> 16); }
__device__ __forceinline__ float bf16_f(u16 h) { return __uint_as_float((unsigned)h << 16); }

constexpr int KP = 256, PITCH = KP + 32;         // bf16 elements per LDS row: 576 B = 16 banks mod 64, so the four rows of a
                                                 // transposing read land on disjoint bank quarters (conflict-free)
constexpr int PLANE = 32 * PITCH;                // elements per plane of a 32-row tile
constexpr int STAGE = 3 * PLANE;                 // three planes
constexpr int LDS_BYTES = 2 * STAGE * 2;         // double buffered: 110592 bytes

__device__ __forceinline__ uint64_t tr_read(unsigned addr) {
    uint64_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

// wave -> blocks as in row_hess_kernel SYM: type 0 (waves 0-3): (w,4..7); type 1 (4, 6): (b,b..b+3), (b+3,b+3); type 2 (5, 7): ...
__host__ __device__ constexpr int s3_nf(int ty) { return ty == 0 ? 5 : (ty == 1 ? 4 : 3); }
__host__ __device__ constexpr int s3_np(int ty) { return ty == 0 ? 4 : 5; }
__host__ __device__ constexpr int s3_ai(int ty, int n) { return ty == 0 ? 0 : (ty == 1 ? (n == 4 ? 3 : 0) : (n >= 3 ? 1 : 0)); }
__host__ __device__ constexpr int s3_bi(int ty, int n) { return ty == 0 ? n + 1 : (ty == 1 ? (n == 4 ? 3 : n) : (n < 3 ? n : n - 2)); }
template <int V> struct IntC { static constexpr int value = V; };

template <int NPROD>
__global__ __launch_bounds__(512, 2) void rowhess6_kernel(const float *U, int64_t S, float *H) {
    extern __shared__ __attribute__((aligned(16))) u16 lds[];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6, l31 = lane & 31, lh = lane >> 5;
    const int uw = __builtin_amdgcn_readfirstlane(wid);
    const int wty = uw < 4 ? 0 : ((uw & 1) ? 2 : 1), sbase = uw >= 6 ? 4 : 0;
    int fblk[5];
    if (wty == 0) { fblk[0] = uw; fblk[1] = 4; fblk[2] = 5; fblk[3] = 6; fblk[4] = 7; }
    else if (wty == 1) { fblk[0] = sbase; fblk[1] = sbase + 1; fblk[2] = sbase + 2; fblk[3] = sbase + 3; fblk[4] = sbase + 3; }
    else { fblk[0] = sbase + 1; fblk[1] = sbase + 2; fblk[2] = sbase + 3; fblk[3] = sbase + 3; fblk[4] = sbase + 3; }
    const float *Ui = U + (int64_t)blockIdx.x * 64 * KP; // every workgroup starts 64 rows further (same work each)
    f32x16 hs[5];
#pragma unroll
    for (int n = 0; n < 5; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) hs[n][r] = 0.f;
    // staging: thread = (tile row t / 16, lane-in-row t % 16), 4 chunks of 4 floats: columns 4 (16 q + tl16) ..
    const int trow = t >> 4, tl16 = t & 15;
    f32x4 rr[4];
    auto gather = [&](int64_t tl) {
        const float *src = Ui + (tl * 32 + trow) * KP + 4 * tl16;
#pragma unroll
        for (int q = 0; q < 4; ++q) rr[q] = *reinterpret_cast<const f32x4 *>(src + 64 * q);
    };
    auto stage = [&](int buf) {
        u16 *base = lds + buf * STAGE + trow * PITCH + 4 * tl16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u16 p0[4], p1[4], p2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = rr[q][e];
                p0[e] = bf16_rn(x); const float r1 = x - bf16_f(p0[e]);
                p1[e] = bf16_rn(r1); const float r2 = r1 - bf16_f(p1[e]);
                p2[e] = bf16_rn(r2);
            }
            auto pack = [](const u16 *p) { return (uint64_t)p[0] | ((uint64_t)p[1] << 16) | ((uint64_t)p[2] << 32) | ((uint64_t)p[3] << 48); };
            *reinterpret_cast<uint64_t *>(base + 64 * q) = pack(p0);
            *reinterpret_cast<uint64_t *>(base + PLANE + 64 * q) = pack(p1);
            *reinterpret_cast<uint64_t *>(base + 2 * PLANE + 64 * q) = pack(p2);
        }
    };
    // transposing fragment read: block `blk` (32 columns), k16 sub-step ks (rows 16 ks ..), plane p
    const int g16 = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    auto frag = [&](int buf, int p, int blk, int ks) -> bf16x8 {
        // this lane supplies the address of row (16 ks + 8 lh + 4 r2 + tq), columns 32 blk + 16 (g16 & 1) + 4 tp ..
        const u16 *rowp = lds + buf * STAGE + p * PLANE + (16 * ks + 8 * lh + tq) * PITCH + 32 * blk + 16 * (g16 & 1) + 4 * tp;
        const unsigned a0 = (unsigned)(uintptr_t)rowp, a1 = (unsigned)(uintptr_t)(rowp + 4 * PITCH);
        union { uint64_t u[2]; bf16x8 v; } x;
        x.u[0] = tr_read(a0);
        x.u[1] = tr_read(a1);
        return x.v;
    };
    auto tile = [&](auto typ, int buf, bool do_stage, bool do_gather, int64_t tl_gather) {
        constexpr int TY = decltype(typ)::value;
        constexpr int NF = s3_nf(TY), NP = s3_np(TY);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 f[3][5];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int q = 0; q < NF; ++q) f[p][q] = frag(buf, p, fblk[q], ks);
            if (ks == 0 && do_stage) stage(buf ^ 1);
            if (ks == 1 && do_gather) gather(tl_gather);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < NP; ++n) {
                const int ai = s3_ai(TY, n), bi = s3_bi(TY, n);
#define MF(P, Q) hs[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[P][ai], f[Q][bi], hs[n], 0, 0, 0);
                if (NPROD >= 6) { MF(2, 0) MF(1, 1) MF(0, 2) }
                if (NPROD >= 3) { MF(1, 0) MF(0, 1) }
                MF(0, 0)
#undef MF
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const int64_t nt = S / 32;
    gather(0);
    stage(0);
    if (nt > 1) gather(1);
    __syncthreads();
    auto run = [&](auto typ) {
        for (int64_t tl = 0; tl < nt; ++tl) {
            tile(typ, (int)(tl & 1), tl + 1 < nt, tl + 2 < nt, tl + 2);
            __syncthreads();
        }
    };
    if (wty == 0) run(IntC<0>{}); else if (wty == 1) run(IntC<1>{}); else run(IntC<2>{});
    // store the upper blocks only (spike): lane = column l31 of block bb, register r = row (r & 3) + 8 (r >> 2) + 4 lh of block ba
    float *Hi = H + (int64_t)blockIdx.x * KP * KP;
    auto emit = [&](auto typ) {
        constexpr int TY = decltype(typ)::value;
#pragma unroll
        for (int n = 0; n < s3_np(TY); ++n) {
            const int ba = fblk[s3_ai(TY, n)], bb = fblk[s3_bi(TY, n)];
#pragma unroll
            for (int r = 0; r < 16; ++r) Hi[(32 * ba + (r & 3) + 8 * (r >> 2) + 4 * lh) * KP + 32 * bb + l31] = hs[n][r];
        }
    };
    if (wty == 0) emit(IntC<0>{}); else if (wty == 1) emit(IntC<1>{}); else emit(IntC<2>{});
}

int main(int argc, char **argv) {
    const int64_t S = argc > 1 ? atoll(argv[1]) : 8192, NWG = argc > 2 ? atoll(argv[2]) : 2048;
    const int64_t rows = S + 64 * NWG;
    float *U, *H;
    CK(hipMalloc(&U, rows * KP * 4)); CK(hipMalloc(&H, NWG * KP * KP * 4));
    hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, U, rows * KP, (int64_t)KP);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int nprod : {6, 3, 1}) {
        auto launch = [&]() {
            if (nprod == 6) { CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&rowhess6_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
                hipLaunchKernelGGL((rowhess6_kernel<6>), dim3((unsigned)NWG), dim3(512), LDS_BYTES, 0, U, S, H); }
            else if (nprod == 3) { CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&rowhess6_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
                hipLaunchKernelGGL((rowhess6_kernel<3>), dim3((unsigned)NWG), dim3(512), LDS_BYTES, 0, U, S, H); }
            else { CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&rowhess6_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
                hipLaunchKernelGGL((rowhess6_kernel<1>), dim3((unsigned)NWG), dim3(512), LDS_BYTES, 0, U, S, H); }
        };
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int it = 0; it < 3; ++it) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
        // check some upper-triangle entries of workgroup 3 against fp64
        std::vector<float> h(KP * KP);
        CK(hipMemcpy(h.data(), H + 3 * KP * KP, KP * KP * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int a = 0; a < KP; a += 17)
            for (int b = a; b < KP; b += 23) {
                double ref = 0, mag = 0;
                for (int64_t j = 0; j < S; ++j) { const double x = value_at(3 * 64 + j, a), y = value_at(3 * 64 + j, b); ref += x * y; mag += fabs(x * y); }
                worst = fmax(worst, fabs(h[a * KP + b] - ref) / mag);
            }
        printf("products=%d: %.3f ms for %lld Hessians of %lld rows (%.2f us per 32-row step per CU-slot)  max |err| / sum|terms| = %.2e\n", nprod, ms,
               (long long)NWG, (long long)S, ms * 1e3 / ((double)NWG / 256.0 * (S / 32)), worst);
    }
    return 0;
}

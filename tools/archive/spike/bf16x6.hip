// Feasibility spike (NOT part of libcmfhip): C[M x 256] = A[M x K] * B[256 x K]^T with every fp32 operand split
// exactly into three bf16 planes and the six leading cross products formed on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation ("bf16x6": fp32-equivalent products at 1/6 of the bf16 matrix rate = 2.6x the fp32 MFMA rate).
// Planes are stored tile-major ([row tile 256][k tile 16][256 rows][16 k] bf16 = 8 KB contiguous per tile).
// Tried on top of this and dropped (same-run A/B at the C4 shape, six products 9.9-10.0 ms): staging two groups later
// (-3 %, within noise), first-use order of the fragment loads (+8 %), conflict-free [k half][row] tile layout (+-0),
// mid-step barrier with the next step's fragments prefetched in groups 2 / 3 (10.35 ms).  PMC of the library kernel: the
// matrix pipe is busy 70 % of the cycles at an effective clock of 1.66 GHz -- the chip throttles under this load.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/spike/bf16x6.hip -o /tmp/bf16x6 && /tmp/bf16x6
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

__host__ __device__ inline float value_at(uint64_t seed, uint64_t r, uint64_t c) { // uniform (0, 1), exactly reproducible on the host
    uint64_t x = seed ^ (r * 0x9E3779B97F4A7C15ull) ^ (c * 0xC2B2AE3D27D4EB4Full);
    x ^= x >> 33; x *= 0xFF51AFD7ED558CCDull; x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ull; x ^= x >> 33;
    return (float)((x >> 40) + 1) * (1.0f / 16777217.0f);
}
__host__ __device__ inline u16 bf16_rn(float f) { // round to nearest even, as the hardware conversion does
    uint32_t u; memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (u16)(u >> 16);
}
__host__ __device__ inline float bf16_f(u16 h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

// planes[p] tile-major: element (r, k) at ((r/256 * KT + k/16) * 256 + r%256) * 16 + k%16
__global__ void fill_split_kernel(u16 *p0, u16 *p1, u16 *p2, int64_t R, int64_t K, uint64_t seed) {
    const int64_t KT = K / 16, total = R * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t kk = i % 16, rr = (i / 16) % 256, t = i / 4096, kt = t % KT, rt = t / KT;
        const float x = value_at(seed, (uint64_t)(rt * 256 + rr), (uint64_t)(kt * 16 + kk));
        const u16 a = bf16_rn(x); const float r1 = x - bf16_f(a);
        const u16 b = bf16_rn(r1); const float r2 = r1 - bf16_f(b);
        const u16 c = bf16_rn(r2);
        p0[i] = a; p1[i] = b; p2[i] = c;
    }
}

constexpr int ROWB = 48;                    // LDS bytes per tile row: 16 k x 2 B + 16 B pad (conflict-free b128 reads)
constexpr int PLANE = 256 * ROWB;           // one plane of one operand tile
constexpr int STAGE = 6 * PLANE;            // A planes 0..2, B planes 0..2
constexpr int LDS_BYTES = 2 * STAGE;        // 147456

template <int NPROD, int DIAG = 0>
__global__ __launch_bounds__(512, 2) void bf16x6_kernel(const u16 *A0, const u16 *A1, const u16 *A2, const u16 *B0, const u16 *B1,
                                                        const u16 *B2, float *C, int64_t KT) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6, l31 = lane & 31, lh = lane >> 5;
    const int wrow0 = (wid >> 1) * 64, wcol0 = (wid & 1) * 128;
    const u16 *Ap[3] = {A0, A1, A2}, *Bp[3] = {B0, B1, B2};
    const int64_t atile0 = (int64_t)blockIdx.x * KT; // first k tile of this row tile
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // staged 16-byte chunks (chunk t of each of the six 8 KB plane tiles), two sets: the tile written to LDS at step kt
    // was requested two steps earlier (one step = 3k cycles is shorter than the HBM latency under load)
    f32x4 st[2][6];
    auto gload = [&](int set, int64_t kt) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            st[set][p] = *reinterpret_cast<const f32x4 *>(Ap[p] + ((atile0 + kt) * 4096 + 8 * t));
            st[set][3 + p] = *reinterpret_cast<const f32x4 *>(Bp[p] + (kt * 4096 + 8 * t));
        }
    };
    const int srow = t >> 1, shalf = t & 1;
    auto lstore = [&](int set, int buf) {
        unsigned char *base = lds + buf * STAGE + srow * ROWB + 16 * shalf;
#pragma unroll
        for (int q = 0; q < 6; ++q) *reinterpret_cast<f32x4 *>(base + q * PLANE) = st[set][q];
    };
    // one K-step on LDS buffer `buf`; `set` = register set holding tile kt+1 (stored to the other buffer, then refilled with kt+3)
    auto compute = [&](int buf, int set, bool do_store, bool do_load, int64_t kt_load) {
        const unsigned char *As = lds + buf * STAGE, *Bs = As + 3 * PLANE;
        bf16x8 a[3][2];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                a[p][i] = *reinterpret_cast<const bf16x8 *>(As + p * PLANE + (wrow0 + 32 * i + l31) * ROWB + 16 * lh);
        bf16x8 b[2][3];
        auto ldb = [&](int j, bf16x8 *dst) {
#pragma unroll
            for (int p = 0; p < 3; ++p) dst[p] = *reinterpret_cast<const bf16x8 *>(Bs + p * PLANE + (wcol0 + 32 * j + l31) * ROWB + 16 * lh);
        };
        ldb(0, b[0]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j + 1 < 4) ldb(j + 1, b[(j + 1) & 1]);
            if (DIAG != 1 && j == 0 && do_store) lstore(set, buf ^ 1);
            if (DIAG != 1 && j == 1 && do_load) gload(set, kt_load);
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 *bb = b[j & 1];
            // smallest terms first: (a3 b1) (a2 b2) (a1 b3) (a2 b1) (a1 b2) (a1 b1); the two row blocks alternate so that
            // consecutive MFMAs never chain on the same accumulator
#define MF(P, Q)                                                                                                   \
    acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[P][0], bb[Q], acc[0][j], 0, 0, 0);                        \
    acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[P][1], bb[Q], acc[1][j], 0, 0, 0);
            if (NPROD >= 6) { MF(2, 0) MF(1, 1) MF(0, 2) }
            if (NPROD >= 3) { MF(1, 0) MF(0, 1) }
            MF(0, 0)
#undef MF
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // tile kt lives in LDS buffer kt & 1 and was staged through register set kt & 1
    gload(0, 0);
    lstore(0, 0);
    if (KT > 1) gload(1, 1);
    if (KT > 2) gload(0, 2);
    __syncthreads();
    for (int64_t kt = 0; kt < KT; kt += 2) {
        compute(0, 1, kt + 1 < KT, kt + 3 < KT, kt + 3);      // computes tile kt; stores tile kt+1 (set 1); set 1 <- tile kt+3
        __syncthreads();
        if (kt + 1 < KT) {
            compute(1, 0, kt + 2 < KT, kt + 4 < KT, kt + 4);  // computes tile kt+1; stores tile kt+2 (set 0); set 0 <- tile kt+4
            __syncthreads();
        }
    }
    // epilogue: lane = column l31, register r = row (r & 3) + 8 (r >> 2) + 4 lh
    const int64_t row0 = (int64_t)blockIdx.x * 256 + wrow0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                C[(row0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh) * 256 + wcol0 + 32 * j + l31] = acc[i][j][r];
}

template <int NPROD, int DIAG = 0>
static void run(u16 **A, u16 **B, float *C, int64_t M, int64_t KT) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&bf16x6_kernel<NPROD, DIAG>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipLaunchKernelGGL((bf16x6_kernel<NPROD, DIAG>), dim3((unsigned)(M / 256)), dim3(512), LDS_BYTES, 0, A[0], A[1], A[2], B[0], B[1], B[2], C, KT);
}

// short reduction (K = 64): the fp32 accumulation error is negligible, what is left is the product splitting
static void accuracy_small() {
    const int64_t M = 256, K = 64, N = 256;
    u16 *A[3], *B[3]; float *C;
    for (int p = 0; p < 3; ++p) { CK(hipMalloc(&A[p], M * K * 2)); CK(hipMalloc(&B[p], N * K * 2)); }
    CK(hipMalloc(&C, M * N * 4));
    hipLaunchKernelGGL(fill_split_kernel, dim3(64), dim3(256), 0, 0, A[0], A[1], A[2], M, K, 11ull);
    hipLaunchKernelGGL(fill_split_kernel, dim3(64), dim3(256), 0, 0, B[0], B[1], B[2], N, K, 22ull);
    std::vector<float> hc(M * N);
    for (int nprod : {6, 3, 1}) {
        if (nprod == 6) run<6>(A, B, C, M, K / 16); else if (nprod == 3) run<3>(A, B, C, M, K / 16); else run<1>(A, B, C, M, K / 16);
        CK(hipMemcpy(hc.data(), C, M * N * 4, hipMemcpyDeviceToHost));
        double worst = 0, worst32 = 0;
        for (int64_t r = 0; r < M; ++r)
            for (int64_t c = 0; c < N; ++c) {
                double ref = 0; float f32 = 0.f;
                for (int64_t k = 0; k < K; ++k) { const float x = value_at(11ull, r, k), y = value_at(22ull, c, k); ref += (double)x * y; f32 += x * y; }
                worst = fmax(worst, fabs(hc[r * N + c] - ref) / fabs(ref));
                worst32 = fmax(worst32, fabs(f32 - ref) / fabs(ref));
            }
        printf("K=64 accuracy, products=%d: max rel err %.2e over %lld entries (plain fp32 FMA chain: %.2e)\n", nprod, worst, (long long)(M * N), worst32);
    }
    for (int p = 0; p < 3; ++p) { CK(hipFree(A[p])); CK(hipFree(B[p])); }
    CK(hipFree(C));
}

int main(int argc, char **argv) {
    accuracy_small();
    const int64_t M = argc > 1 ? atoll(argv[1]) : 65536, K = argc > 2 ? atoll(argv[2]) : 65536, N = 256;
    const int64_t KT = K / 16;
    u16 *A[3], *B[3];
    float *C;
    for (int p = 0; p < 3; ++p) { CK(hipMalloc(&A[p], M * K * 2)); CK(hipMalloc(&B[p], N * K * 2)); }
    CK(hipMalloc(&C, M * N * 4));
    hipLaunchKernelGGL(fill_split_kernel, dim3(4096), dim3(256), 0, 0, A[0], A[1], A[2], M, K, 11ull);
    hipLaunchKernelGGL(fill_split_kernel, dim3(1024), dim3(256), 0, 0, B[0], B[1], B[2], N, K, 22ull);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int nprod : {6, 3, 1, 60}) {
        auto launch = [&]() {
            if (nprod == 6) run<6>(A, B, C, M, KT); else if (nprod == 3) run<3>(A, B, C, M, KT); else if (nprod == 1) run<1>(A, B, C, M, KT);
            else run<6, 1>(A, B, C, M, KT); // diagnostic: six products, no staging in the loop (wrong results)
        };
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int it = 0; it < 5; ++it) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        // accuracy on sampled entries against fp64 sums of the exact fp32 values
        std::vector<float> hc(256 * N);
        double worst = 0, worst32 = 0;
        for (int s = 0; s < 8; ++s) {
            const int64_t r = (int64_t)((s * 7919ll * 104729ll) % M);
            CK(hipMemcpy(hc.data(), C + r * N, N * 4, hipMemcpyDeviceToHost));
            for (int c = 0; c < N; c += 37) {
                double ref = 0; float f32 = 0.f;
                for (int64_t k = 0; k < K; ++k) { const float x = value_at(11ull, r, k), y = value_at(22ull, c, k); ref += (double)x * y; f32 += x * y; }
                worst = fmax(worst, fabs(hc[c] - ref) / fabs(ref));
                worst32 = fmax(worst32, fabs(f32 - ref) / fabs(ref));
            }
        }
        printf("products=%d: %.3f ms per launch  = %.1f TFLOP/s fp32-equivalent (2MKN)   max rel err %.2e   (sequential fp32 sum: %.2e)\n", nprod, ms,
               2.0 * M * K * N / ms / 1e9, worst, worst32);
    }
    return 0;
}

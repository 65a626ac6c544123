// cmf_bf16x6.hip.h -- OPTIONAL arithmetic for the data passes at k_pad = 256 (cmf_set_option "gemm_arith" = 1;
// default 0 = the fp32 MFMA kernels of cmf_kernels.hip.h).
//
// Every fp32 operand is split EXACTLY into three bf16 planes (x = x1 + x2 + x3, 8 + 8 + 8 mantissa bits, each plane
// rounded to nearest on the remainder of the previous ones) and the six leading cross products
//     x1 y1 + x1 y2 + x2 y1 + x1 y3 + x2 y2 + x3 y1          (dropped terms <= 2^-24 relative)
// are formed on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, smallest terms first.  Measured on products of
// K = 64 (no accumulation error): max relative error 3.8e-7, against 5.2e-7 for a plain fp32 FMA chain
// (tools/spike/bf16x6.hip) -- fp32-equivalent results at 1/6 of the bf16 matrix rate = 2.6x the fp32 MFMA rate.
//
// Data planes are resident (built once per data matrix and orientation) in a tile-major layout,
// [row tile of 256][k tile of 16][256 rows][16 k] bf16 = 8 KB contiguous per tile (factor operand: k_pad rows per tile), so that a workgroup fetches a
// whole operand tile with one 16-byte load per thread; the factor operand is split and transposed per product into
// the same layout.  One kernel form covers all four data passes:  C[R x k_pad] (+)= A[R x K] * B^T,  A = X, X^T, Y or Y^T,
// k_pad = 256 or 128.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cmfk {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short bf16_rn_bits(float f) { // round to nearest even
    const unsigned u = __float_as_uint(f);
    unsigned r = u + 0x7FFFu + ((u >> 16) & 1u);
    // a finite value just under FLT_MAX must not round up to infinity (the remainder would be -inf): truncate it instead
    if ((r & 0x7F800000u) == 0x7F800000u && (u & 0x7F800000u) != 0x7F800000u) r = u;
    return (unsigned short)(r >> 16);
}
__device__ __forceinline__ float bf16_bits_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

struct Bf16x3 {
    unsigned short a, b, c;
};
__device__ __forceinline__ Bf16x3 split3(float x) {
    Bf16x3 r;
    r.a = bf16_rn_bits(x);
    const float r1 = x - bf16_bits_f(r.a);
    r.b = bf16_rn_bits(r1);
    const float r2 = r1 - bf16_bits_f(r.b);
    r.c = bf16_rn_bits(r2);
    return r;
}

// planes of op(S): element (r, k) = trans ? S[k * ld + r] : S[r * ld + k], r < R (multiple of TR), k < K (multiple of 16).
// One thread = one (row, k tile): 16 k values -> 32 contiguous bytes per plane.
// TR = rows per tile (256 for the data operand, k_pad for the factor operand).
__global__ void bf16x3_split_kernel(const float *S, int64_t ld, int trans, int64_t R, int64_t K, unsigned short *P0, unsigned short *P1,
                                    unsigned short *P2, int TR) {
    const int64_t KT = K / 16, total = R * KT;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r, kt;
        if (trans) { r = i % R; kt = i / R; }          // consecutive threads walk a row of S: coalesced reads
        else { kt = i % KT; r = i / KT; }              // consecutive threads walk along a row of S too
        unsigned short pa[16], pb[16], pc[16];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int64_t k = kt * 16 + kk;
            const Bf16x3 s = split3(trans ? S[k * ld + r] : S[r * ld + k]);
            pa[kk] = s.a; pb[kk] = s.b; pc[kk] = s.c;
        }
        const int64_t o = (((r / TR) * KT + kt) * TR + (r % TR)) * 16;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) { P0[o + kk] = pa[kk]; P1[o + kk] = pb[kk]; P2[o + kk] = pc[kk]; }
    }
}

constexpr int BX_ROWB = 48;                 // LDS bytes per tile row: 16 k x 2 B + 16 B pad (conflict-free b128 reads)
constexpr int BX_PLANE = 256 * BX_ROWB;     // one plane of the A tile (256 rows)
template <int NJ>                           // NJ = 32-column blocks per wave: 4 -> output width 256, 2 -> 128
struct BxCfg {
    static constexpr int BN = 64 * NJ;
    static constexpr int BPLANE = BN * BX_ROWB;                 // one plane of the B tile (BN rows)
    static constexpr int STAGE = 3 * (BX_PLANE + BPLANE);       // A planes 0..2, B planes 0..2
    static constexpr int LDS_BYTES = 2 * STAGE;                 // 147456 (NJ = 4) / 110592 (NJ = 2)
};

struct Bf16x6Args {
    const unsigned short *A[3];   // planes of the data operand, rows R
    const unsigned short *B[3];   // planes of the transposed factor operand, BN = k_pad rows
    float *C;                     // [R x BN] row-major; split-K: slab blockIdx.y at C + blockIdx.y * slab_stride
    int64_t KT;                   // K / 16
    int64_t kt_per_split;         // k tiles per blockIdx.y (multiple of 2 unless it is the whole range)
    int64_t slab_stride;
    int accumulate;
};

// 512 threads, output tile 256 x BN, wave tile 64 x BN/2 (2 x NJ MFMA blocks), one 16-deep K-step per barrier through
// double-buffered LDS; register staging two steps ahead.
template <int NJ>
__global__ __launch_bounds__(512, 2) void bf16x6_gemm_kernel(Bf16x6Args g) {
    using Cf = BxCfg<NJ>;
    constexpr int BN = Cf::BN;
    extern __shared__ __attribute__((aligned(16))) unsigned char bxl[];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6, l31 = lane & 31, lh = lane >> 5;
    const int wrow0 = (wid >> 1) * 64, wcol0 = (wid & 1) * (32 * NJ);
    const bool bload = (2 * BN >= 512) || t < 2 * BN; // the B tile has 2 BN 16-byte chunks (compile-time true at BN = 256:
                                                      // a predicated load would make hipcc wait right behind it)
    // this workgroup reduces k tiles [kt0, kt0 + KT) of the g.KT in a row tile
    const int64_t kt0 = (int64_t)blockIdx.y * g.kt_per_split;
    const int64_t KT = (g.KT - kt0 < g.kt_per_split) ? g.KT - kt0 : g.kt_per_split;
    const int64_t atile0 = (int64_t)blockIdx.x * g.KT + kt0;
    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 st[2][6];
    auto gload = [&](int set, int64_t kt) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            st[set][p] = *reinterpret_cast<const f32x4 *>(g.A[p] + ((atile0 + kt) * 4096 + 8 * t));
            if (bload) st[set][3 + p] = *reinterpret_cast<const f32x4 *>(g.B[p] + ((kt0 + kt) * (BN * 16) + 8 * t));
        }
    };
    const int srow = t >> 1, shalf = t & 1;
    auto lstore = [&](int set, int buf) {
        unsigned char *base = bxl + buf * Cf::STAGE + srow * BX_ROWB + 16 * shalf;
#pragma unroll
        for (int q = 0; q < 3; ++q) *reinterpret_cast<f32x4 *>(base + q * BX_PLANE) = st[set][q];
        if (bload) {
#pragma unroll
            for (int q = 0; q < 3; ++q) *reinterpret_cast<f32x4 *>(base + 3 * BX_PLANE + q * Cf::BPLANE) = st[set][3 + q];
        }
    };
    auto compute = [&](int buf, int set, bool do_store, bool do_load, int64_t kt_load) {
        const unsigned char *As = bxl + buf * Cf::STAGE, *Bs = As + 3 * BX_PLANE;
        bf16x8 a[3][2];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                a[p][i] = *reinterpret_cast<const bf16x8 *>(As + p * BX_PLANE + (wrow0 + 32 * i + l31) * BX_ROWB + 16 * lh);
        bf16x8 b[2][3];
        auto ldb = [&](int j, bf16x8 *dst) {
#pragma unroll
            for (int p = 0; p < 3; ++p)
                dst[p] = *reinterpret_cast<const bf16x8 *>(Bs + p * Cf::BPLANE + (wcol0 + 32 * j + l31) * BX_ROWB + 16 * lh);
        };
        ldb(0, b[0]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (j + 1 < NJ) ldb(j + 1, b[(j + 1) & 1]);
            if (j == 0 && do_store) lstore(set, buf ^ 1);
            if (j == 1 && do_load) gload(set, kt_load);
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 *bb = b[j & 1];
            // smallest terms first; the two row blocks alternate so that consecutive MFMAs never chain on one accumulator
#define CMF_MF(P, Q)                                                                                       \
    acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[P][0], bb[Q], acc[0][j], 0, 0, 0);                \
    acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[P][1], bb[Q], acc[1][j], 0, 0, 0);
            CMF_MF(2, 0) CMF_MF(1, 1) CMF_MF(0, 2) CMF_MF(1, 0) CMF_MF(0, 1) CMF_MF(0, 0)
#undef CMF_MF
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
        compute(0, 1, kt + 1 < KT, kt + 3 < KT, kt + 3);
        __syncthreads();
        if (kt + 1 < KT) {
            compute(1, 0, kt + 2 < KT, kt + 4 < KT, kt + 4);
            __syncthreads();
        }
    }
    // lane = column l31, register r = row (r & 3) + 8 (r >> 2) + 4 lh
    float *Cw = g.C + (int64_t)blockIdx.y * g.slab_stride + ((int64_t)blockIdx.x * 256 + wrow0 + 4 * lh) * BN + wcol0 + l31;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float *dst = Cw + (32 * i + (r & 3) + 8 * (r >> 2)) * BN + 32 * j;
                *dst = acc[i][j][r] + (g.accumulate ? *dst : 0.f);
            }
}

} // namespace cmfk

// cmf_sparse.hip.h -- CSR kernels for sparse X / Y (gfx950), HBM-bound.
//
// The reference accepts scipy CSR/CSC (pycmf/cmf.py:679) and multiplies through
// sklearn's safe_sparse_dot (pycmf/cmf_solvers.py:232,238,244).  Here a sparse data
// matrix is kept twice on the device, as CSR of A and as CSR of A^T, so that both
// A*F and A^T*F are row-gather SpMMs:  out[r,:] = sum_q val[q] * F[idx[q],:].
// A group of k_pad/4 lanes owns one output row (float4 per lane = one coalesced
// 16-B-per-lane read of a factor row per nonzero); a 64-lane wave carries
// 64/(k_pad/4) rows.  No atomics, deterministic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cmfk {

struct CsrView {
    const int64_t *indptr; // rows + 1
    const int32_t *idx;    // column of each nonzero
    const float *val;
    int64_t rows;
};

// GL = lanes per row group = min(64, kp/4); CH = float4 chunks per lane (kp/4/GL)
template <int GL, int CH>
__global__ __launch_bounds__(256) void spmm_csr_kernel(CsrView A, const float *F, int kp, float *out, int accumulate) {
    constexpr int RPW = 64 / GL; // rows per wave
    const int lane = threadIdx.x & 63;
    const int gl = lane % GL, gsub = lane / GL;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t row = wave * RPW + gsub;
    if (row >= A.rows) return;
    const int64_t beg = A.indptr[row], end = A.indptr[row + 1];
    f32x4 acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    int64_t q = beg;
    for (; q + 4 <= end; q += 4) { // 4 independent gathers in flight per lane
        int32_t j[4];
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { j[u] = A.idx[q + u]; v[u] = A.val[q + u]; }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            f32x4 f[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) f[u] = *reinterpret_cast<const f32x4 *>(F + (int64_t)j[u] * kp + 4 * (gl + GL * c));
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[c] += v[u] * f[u];
        }
    }
    for (; q < end; ++q) {
        const int32_t j = A.idx[q];
        const float v = A.val[q];
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] += v * *reinterpret_cast<const f32x4 *>(F + (int64_t)j * kp + 4 * (gl + GL * c));
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        f32x4 *dst = reinterpret_cast<f32x4 *>(out + row * kp + 4 * (gl + GL * c));
        *dst = accumulate ? (*dst + acc[c]) : acc[c];
    }
}

// cross[wg] = sum over the nonzeros of this workgroup's rows of  a_ij * (L_i . R_j)
// (the 2 tr((A R)^T L) term of the expanded Frobenius error, sklearn _beta_divergence
// sparse branch used at pycmf/cmf_solvers.py:40)
template <int GL, int CH>
__global__ __launch_bounds__(256) void sddmm_cross_kernel(CsrView A, const float *L, const float *R, int kp, double *partials) {
    constexpr int RPW = 64 / GL;
    const int lane = threadIdx.x & 63;
    const int gl = lane % GL, gsub = lane / GL;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t row = wave * RPW + gsub;
    float acc = 0.f;
    if (row < A.rows) {
        f32x4 l[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) l[c] = *reinterpret_cast<const f32x4 *>(L + row * kp + 4 * (gl + GL * c));
        const int64_t beg = A.indptr[row], end = A.indptr[row + 1];
        for (int64_t q = beg; q < end; ++q) {
            const int32_t j = A.idx[q];
            const float v = A.val[q];
            float dot = 0.f;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const f32x4 r = *reinterpret_cast<const f32x4 *>(R + (int64_t)j * kp + 4 * (gl + GL * c));
                dot += l[c][0] * r[0] + l[c][1] * r[1] + l[c][2] * r[2] + l[c][3] * r[3];
            }
            acc += v * dot;
        }
    }
    double v = (double)acc;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __shared__ double red[4];
    if (lane == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// dense[r][idx] += val : expand CSR into the padded dense layout (per-row Newton images)
__global__ void csr_to_dense_kernel(CsrView A, float *dense, int64_t ld) {
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= A.rows) return;
    const int lane = threadIdx.x & 63;
    for (int64_t q = A.indptr[row] + lane; q < A.indptr[row + 1]; q += 64) dense[row * ld + A.idx[q]] += A.val[q];
}

// sum_ab A[a][b] * B[a][b] over kp x kp (fp64 accumulate, single block)
__global__ void frob_inner_kernel(const float *A, const float *B, int n, double *out) {
    __shared__ double red[256];
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) v += (double)A[i] * (double)B[i];
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0];
}

} // namespace cmfk

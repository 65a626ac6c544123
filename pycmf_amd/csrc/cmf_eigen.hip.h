// cmf_eigen.hip.h -- batched "safe inverse" of symmetric k x k Hessians on gfx950.
//
// Replaces NewtonSolver._safe_invert (pycmf/cmf_solvers.py:346-356):
//     lam, Q = eigh(H);  lam = |lam|;  lam[lam < pert] = pert;  return Q diag(1/lam) Q^T
// For a symmetric H the singular values are |lam| and the right singular vectors
// are the eigenvectors, so a one-sided (Hestenes) Jacobi SVD delivers exactly the
// two things the formula needs.  One workgroup owns one matrix; the k/2 disjoint
// column pairs of a round-robin step are rotated concurrently by the waves, with
// wave-level reductions for the three inner products of a pair.  The working
// images (H V)^T and V^T live in LDS when they fit (k_pad <= 128) and in an
// L2-resident global workspace otherwise.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cmfk {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Hin/Hout: nmat matrices, row-major, leading dimension kp, matrix stride `stride`.
// n = valid order (<= kp).  Only the leading n x n block is read; Hout's padding is zeroed.
template <bool USE_LDS>
__global__ __launch_bounds__(256) void jacobi_safe_inverse_kernel(const float *Hin, float *Hout, float *ws, int n, int kp,
                                                                  int64_t stride, float pert, int nmat, int max_sweeps,
                                                                  const int *need) {
    extern __shared__ __attribute__((aligned(16))) float esm[];
    const int mat = blockIdx.x;
    if (mat >= nmat) return;
    if (need && !need[mat]) return; // already inverted by the Cholesky fast path
    const float *H = Hin + (int64_t)mat * stride;
    float *O = Hout + (int64_t)mat * stride;
    const int ld = n;
    float *B, *Vt, *inv;
    if (USE_LDS) {
        B = esm;
        Vt = esm + n * ld;
        inv = esm + 2 * n * ld;
    } else {
        B = ws + (int64_t)mat * (2 * (int64_t)kp * kp);
        Vt = B + (int64_t)n * ld;
        inv = esm;
    }
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    constexpr int NW = 4;

    for (int idx = t; idx < n * n; idx += 256) {
        const int r = idx / n, c = idx % n;
        B[idx] = H[r * kp + c];
        Vt[idx] = (r == c) ? 1.0f : 0.0f;
    }
    __syncthreads();

    const int N = n + (n & 1); // even number of "players"; index n (if odd) is a bye
    const float tol = 2.0e-7f;
    for (int sweep = 0; sweep < max_sweeps && N >= 2; ++sweep) {
        int rotated = 0;
        for (int s = 0; s < N - 1; ++s) {
            for (int pi = wid; pi < N / 2; pi += NW) {
                int p, q;
                if (pi == 0) {
                    p = s;
                    q = N - 1;
                } else {
                    p = (s + pi) % (N - 1);
                    q = (s - pi + (N - 1)) % (N - 1);
                }
                if (p >= n || q >= n) continue;
                float *bp = B + p * ld, *bq = B + q * ld;
                float a = 0.f, b = 0.f, g = 0.f;
                for (int e = lane; e < n; e += 64) {
                    const float x = bp[e], y = bq[e];
                    a += x * x;
                    b += y * y;
                    g += x * y;
                }
                a = wave_sum(a);
                b = wave_sum(b);
                g = wave_sum(g);
                if (fabsf(g) > tol * sqrtf(a * b) && a > 0.f && b > 0.f) {
                    const float zeta = (b - a) / (2.0f * g);
                    const float tt = copysignf(1.0f, zeta) / (fabsf(zeta) + sqrtf(1.0f + zeta * zeta));
                    const float cs = 1.0f / sqrtf(1.0f + tt * tt);
                    const float sn = cs * tt;
                    float *vp = Vt + p * ld, *vq = Vt + q * ld;
                    for (int e = lane; e < n; e += 64) {
                        const float x = bp[e], y = bq[e];
                        bp[e] = cs * x - sn * y;
                        bq[e] = sn * x + cs * y;
                        const float u = vp[e], w = vq[e];
                        vp[e] = cs * u - sn * w;
                        vq[e] = sn * u + cs * w;
                    }
                    rotated = 1;
                }
            }
            __syncthreads();
        }
        if (!__syncthreads_or(rotated)) break;
    }

    // |lambda_j| = norm of row j of (H V)^T ; clamp ; invert
    for (int j = wid; j < n; j += NW) {
        float a = 0.f;
        for (int e = lane; e < n; e += 64) {
            const float x = B[j * ld + e];
            a += x * x;
        }
        a = wave_sum(a);
        if (lane == 0) {
            float sg = sqrtf(a);
            if (sg < pert) sg = pert;
            inv[j] = 1.0f / sg;
        }
    }
    __syncthreads();
    // O = V diag(inv) V^T  (zero on the padding)
    for (int idx = t; idx < kp * kp; idx += 256) {
        const int r = idx / kp, c = idx % kp;
        float acc = 0.f;
        if (r < n && c < n) {
            for (int j = 0; j < n; ++j) acc += inv[j] * Vt[j * ld + r] * Vt[j * ld + c];
        }
        O[idx] = acc;
    }
}

// ---------------------------------------------------------------------------------------
// Cholesky fast path of the safe inverse.  When every eigenvalue of the (symmetric) H is
// >= pert, the clamp of _safe_invert is the identity and the result is simply H^-1.  That is
// decided exactly by trying to factor H - pert*I: it is positive definite iff lambda_min > pert.
// If so, H = L L^T is factored in a packed lower triangle held in LDS (n(n+1)/2 floats: 131.6 KB
// at n = 256), L is inverted in place and H^-1 = L^-T L^-1 is written out.  Otherwise
// need_jacobi[mat] is set and the Jacobi kernel (which skips matrices whose flag is 0) does the
// general |lambda| / clamp computation.  One workgroup per matrix.
__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; } // j <= i

__device__ bool chol_packed(float *Lp, int n, float floor_) {
    // right-looking Cholesky on the packed lower triangle; returns false on a non-positive pivot
    __shared__ int ok;
    const int t = threadIdx.x;
    if (t == 0) ok = 1;
    __syncthreads();
    for (int j = 0; j < n; ++j) {
        const float dj = Lp[tri(j, j)];
        if (!(dj > floor_)) {
            if (t == 0) ok = 0;
            break; // uniform: every thread reads the same dj
        }
        const float ljj = sqrtf(dj);
        const float inv = 1.0f / ljj;
        __syncthreads();
        for (int i = j + t; i < n; i += 256) Lp[tri(i, j)] = (i == j) ? ljj : Lp[tri(i, j)] * inv;
        __syncthreads();
        for (int i = j + 1 + t; i < n; i += 256) {
            const float lij = Lp[tri(i, j)];
            float *row = Lp + tri(i, 0);
            for (int c = j + 1; c <= i; ++c) row[c] -= lij * Lp[tri(c, j)];
        }
        __syncthreads();
    }
    __syncthreads();
    return ok != 0;
}

__global__ __launch_bounds__(256) void chol_safe_inverse_kernel(const float *Hin, float *Hout, int *need_jacobi, int n, int kp,
                                                                 int64_t stride, float pert, int nmat) {
    extern __shared__ __attribute__((aligned(16))) float esm[];
    const int mat = blockIdx.x;
    if (mat >= nmat) return;
    const float *H = Hin + (int64_t)mat * stride;
    float *O = Hout + (int64_t)mat * stride;
    float *Lp = esm;
    const int t = threadIdx.x;
    const int ntri = n * (n + 1) / 2;
    // scale-aware pivot floor: a pivot must clear rounding noise of the largest diagonal entry
    float dmax = 0.f;
    for (int i = t; i < n; i += 256) dmax = fmaxf(dmax, fabsf(H[i * kp + i]));
    __shared__ float red[4];
    for (int off = 32; off > 0; off >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, off, 64));
    if ((t & 63) == 0) red[t >> 6] = dmax;
    __syncthreads();
    dmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float floor_ = 4.0e-6f * dmax;

    // 1) is H - pert*I positive definite?
    for (int idx = t; idx < ntri; idx += 256) {
        int i = (int)((sqrtf(8.0f * idx + 1.0f) - 1.0f) * 0.5f);
        while (tri(i + 1, 0) <= idx) ++i;
        while (tri(i, 0) > idx) --i;
        const int j = idx - tri(i, 0);
        Lp[idx] = H[i * kp + j] - (i == j ? pert : 0.f);
    }
    __syncthreads();
    const bool pd = chol_packed(Lp, n, floor_);
    if (!pd) {
        if (t == 0) need_jacobi[mat] = 1;
        return;
    }
    if (t == 0) need_jacobi[mat] = 0;
    // 2) factor H itself
    __syncthreads();
    for (int idx = t; idx < ntri; idx += 256) {
        int i = (int)((sqrtf(8.0f * idx + 1.0f) - 1.0f) * 0.5f);
        while (tri(i + 1, 0) <= idx) ++i;
        while (tri(i, 0) > idx) --i;
        const int j = idx - tri(i, 0);
        Lp[idx] = H[i * kp + j];
    }
    __syncthreads();
    (void)chol_packed(Lp, n, 0.0f);
    // 3) X = L^-1 in place (lower): column j from the already inverted trailing block
    for (int j = n - 1; j >= 0; --j) {
        const float xjj = 1.0f / Lp[tri(j, j)];
        float mine[1 + 255 / 256 + 1]; // rows handled by this thread: i = j+1+t, j+1+t+256 (n <= 512)
        int cnt = 0;
        for (int i = j + 1 + t; i < n; i += 256) {
            float sacc = 0.f;
            const float *row = Lp + tri(i, 0);
            for (int c = j + 1; c <= i; ++c) sacc += row[c] * Lp[tri(c, j)];
            mine[cnt++] = -sacc * xjj;
        }
        __syncthreads();
        cnt = 0;
        for (int i = j + 1 + t; i < n; i += 256) Lp[tri(i, j)] = mine[cnt++];
        if (t == 0) Lp[tri(j, j)] = xjj;
        __syncthreads();
    }
    // 4) H^-1 = X^T X (symmetric), zero on the padding
    for (int idx = t; idx < kp * kp; idx += 256) {
        const int a = idx / kp, b = idx % kp;
        float acc = 0.f;
        if (a < n && b < n) {
            const int lo = a > b ? a : b;
            for (int j = lo; j < n; ++j) acc += Lp[tri(j, a)] * Lp[tri(j, b)];
        }
        O[idx] = acc;
    }
}

// KR[j][a*kp + b] = F[j][a] * F[j][b]     (row-wise Khatri-Rao square of a factor)
__global__ void khatri_rao_kernel(float *KR, const float *F, int64_t rows, int kp) {
    const int64_t total4 = rows * kp * (kp / 4);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int b4 = (int)(i % (kp / 4));
        const int a = (int)((i / (kp / 4)) % kp);
        const int64_t j = i / ((int64_t)kp * (kp / 4));
        const float fa = F[j * kp + a];
        const f32x4 fb = *reinterpret_cast<const f32x4 *>(F + j * kp + 4 * b4);
        *reinterpret_cast<f32x4 *>(KR + (j * kp + a) * kp + 4 * b4) = fa * fb;
    }
}

// H_i = Hrows_i (or 0) + S (or 0) + diag * I   for a chunk of per-row Hessians
__global__ void hessian_finalize_kernel(float *Hrows, const float *S, float diag, int64_t nrows, int kp, int kvalid,
                                        int have_rows) {
    const int64_t total = nrows * kp * kp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(i % ((int64_t)kp * kp));
        const int r = e / kp, c = e % kp;
        float v = have_rows ? Hrows[i] : 0.f;
        if (S) v += S[e];
        if (r == c && r < kvalid) v += diag;
        Hrows[i] = v;
    }
}

// step_i = g_i * Hinv_i  (row vector times symmetric matrix), one wave per row
__global__ __launch_bounds__(256) void rowvec_mat_kernel(float *step, const float *grad, const float *Hinv, int64_t nrows,
                                                         int kp, int kvalid) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const float *g = grad + row * kp;
    const float *Hm = Hinv + row * (int64_t)kp * kp;
    for (int c = lane; c < kp; c += 64) {
        float acc = 0.f;
        if (c < kvalid)
            for (int a = 0; a < kvalid; ++a) acc += g[a] * Hm[a * kp + c];
        step[row * kp + c] = acc;
    }
}

// mask[rows[i]][cols[i]] = 1 built from per-row index lists
//   by_row = 1: list i (of `per` entries) holds column indices of row i      mask[i][idx]
//   by_row = 0: list i holds row indices of column i                          mask[idx][i]
__global__ void scatter_mask_kernel(uint8_t *mask, int64_t ld, const int32_t *idx, int64_t nlists, int64_t per, int by_row) {
    const int64_t total = nlists * per;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t l = i / per;
        const int64_t v = idx[i];
        if (by_row) mask[l * ld + v] = 1;
        else mask[v * ld + l] = 1;
    }
}

// out = a*A + b*B (B nullable)
__global__ void axpby_kernel(float *out, const float *A, float a, const float *B, float b, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        out[i] = a * A[i] + (B ? b * B[i] : 0.f);
}

} // namespace cmfk

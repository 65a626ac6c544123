// cmf_shared64.hip.h -- float64 treatment of the ONE shared k x k Hessian of a linear-link Newton sweep (gfx950).
//
// Reference: NewtonSolver._newton_update_U / _newton_update_V with sg_sample_ratio == 1 and linear links form a single
// Hessian  H = s F^T F (+ s' F'^T F') + l2 I  for all rows of the sweep and invert it once with _safe_invert
// (pycmf/cmf_solvers.py:407-410, :448-450, :346-356), everything in float64.  The factor data on the device is
// float32, but a step is  grad H^-1, so an error dH in H comes back multiplied by cond(H): Grams accumulated and
// inverted in float32 carry about cond(H) 1e-6 into the step.  H is one matrix per sweep, so it is cheap to do right:
//   * gram64_partial_kernel / gram64_reduce_kernel: F^T F with float64 products and float64 accumulation on the
//     float64 matrix pipe (v_mfma_f64_16x16x4_f64; the f32 inputs convert exactly), upper 64 x 64 tiles only, the
//     rows split over workgroups into slabs that are summed in a fixed order (deterministic, no atomics);
//   * safe_inverse64_small_kernel (n <= 64): one workgroup, everything in LDS: Cholesky test of H - pert I, then
//     H^-1 = L^-T L^-1, or -- when an eigenvalue lies under the perturbation -- the reference's own formula
//     Q diag(1 / max(|lambda|, pert)) Q^T by a float64 Hestenes-Jacobi sweep.  One launch, no host round trip;
//   * n > 64: chol64_kernel (blocked right-looking Cholesky of H - pert I and of H by two workgroups, matrix in L2),
//     tri_inverse64_kernel (X = L^-1, 16 columns per workgroup) and gemm64_kernel (H^-1 = X^T X); when the test
//     fails and H is positive semi-definite, the spectral clamp M = max(H, pert I) by float64 matrix polynomials of
//     sign(H - pert I) (the same Newton-Schulz construction as the float32 per-row path in cmf_newton.hip.h, here on
//     gemm64_kernel), then the same Cholesky inverse of M.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cmfk {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// sum over aligned groups of 16 lanes
__device__ __forceinline__ double group16_sum_f64(double v) {
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ------------------------------------------------------------------------------------------ Gram
// slab[split][tile][TS x TS] = sum over this split's rows of F[r, tile_i cols]^T F[r, tile_j cols]   (tile_j >= tile_i)
// F: float32, row-major, ld = kp, rows_pad rows (padding rows are zero).  Workgroup = (TS/32)^2 waves, wave tile 32 x 32
// = 2 x 2 MFMA blocks of 16 x 16; the rows pass through LDS 32 at a time (row pitch TS + 16 floats: the two 16-lane
// halves of a ds_read_b32 group read rows k and k + 1, 16 banks apart).
template <int TS>
__global__ __launch_bounds__((TS / 32) * (TS / 32) * 64) void gram64_partial_kernel(const float *F, int kp, int64_t rows_pad, int64_t chunk,
                                                                                   double *slab) {
    constexpr int WPS = TS / 32, NT = WPS * WPS * 64, LD = TS + 16;
    __shared__ __attribute__((aligned(16))) float sA[32 * LD];
    __shared__ __attribute__((aligned(16))) float sB[32 * LD];
    const int T = kp / TS;
    int b = blockIdx.x, ti = 0;
    while (b >= T - ti) { b -= T - ti; ++ti; }
    const int tj = ti + b;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wi = w / WPS, wj = w % WPS;
    const int l15 = lane & 15, lk = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.y * chunk;
    const int64_t r1 = r0 + chunk < rows_pad ? r0 + chunk : rows_pad;
    f64x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
    for (int64_t r = r0; r < r1; r += 32) {
        __syncthreads();
        for (int idx = t; idx < 8 * TS; idx += NT) {
            const int row = idx / (TS / 4), c4 = idx % (TS / 4);
            const float *src = F + (r + row) * kp + 4 * c4;
            *reinterpret_cast<f32x4 *>(sA + row * LD + 4 * c4) = *reinterpret_cast<const f32x4 *>(src + ti * TS);
            *reinterpret_cast<f32x4 *>(sB + row * LD + 4 * c4) = *reinterpret_cast<const f32x4 *>(src + tj * TS);
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            double a[2], bb[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                a[q] = (double)sA[(4 * s + lk) * LD + wi * 32 + 16 * q + l15];
                bb[q] = (double)sB[(4 * s + lk) * LD + wj * 32 + 16 * q + l15];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], bb[j], acc[i][j], 0, 0, 0);
        }
    }
    double *out = slab + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (TS * TS);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                out[(wi * 32 + 16 * i + lk + 4 * reg) * TS + wj * 32 + 16 * j + l15] = acc[i][j][reg];
}

// G64[r][c] (kp x kp, symmetric) = sum over the splits, in split order; optional float32 copy
__global__ __launch_bounds__(256) void gram64_reduce_kernel(const double *slab, int ts, int kp, int nsplit, double *G64, float *G32) {
    const int T = kp / ts, ntile = T * (T + 1) / 2;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < kp * kp; idx += gridDim.x * 256) {
        int r = idx / kp, c = idx % kp;
        const int rr = r < c ? r : c, cc = r < c ? c : r; // upper-triangle representative (tile row <= tile column)
        int ti = rr / ts, tj = cc / ts;
        int ir = rr % ts, ic = cc % ts;
        if (ti == tj) { ir = r % ts; ic = c % ts; } // diagonal tiles are stored whole
        const int tile = ti * T - ti * (ti - 1) / 2 + (tj - ti);
        const double *p = slab + (int64_t)tile * ts * ts + ir * ts + ic;
        double s = 0.0;
        for (int q = 0; q < nsplit; ++q) s += p[(int64_t)q * ntile * ts * ts];
        G64[idx] = s;
        if (G32) G32[idx] = (float)s;
    }
}

// H64 = a A + b B + diag I on the valid n x n block, identity on the padding; A, B float64 (B nullable)
__global__ __launch_bounds__(256) void hess64_build_kernel(double *H, const double *A, double a, const double *B, double b, double diag,
                                                           int kp, int n) {
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < kp * kp; idx += gridDim.x * 256) {
        const int r = idx / kp, c = idx % kp;
        double v = a * A[idx] + (B ? b * B[idx] : 0.0);
        if (r == c) v += diag;
        if (r >= n || c >= n) v = (r == c) ? 1.0 : 0.0;
        H[idx] = v;
    }
}
// the same from a float32 matrix (the all-reduced Gram of the sharded step)
__global__ __launch_bounds__(256) void hess64_from_f32_kernel(double *H, const float *A, double a, double diag, int kp, int n) {
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < kp * kp; idx += gridDim.x * 256) {
        const int r = idx / kp, c = idx % kp;
        double v = a * (double)A[idx];
        if (r == c) v += diag;
        if (r >= n || c >= n) v = (r == c) ? 1.0 : 0.0;
        H[idx] = v;
    }
}
__global__ __launch_bounds__(256) void axpby64_to_f32_kernel(float *out, const double *A, double a, const double *B, double b, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = (float)(a * A[i] + (B ? b * B[i] : 0.0));
}

// ------------------------------------------------------------------------------------------ n <= 64: one workgroup, LDS
// right-looking Cholesky of the n x n matrix A (LDS, pitch n), L written to Lm (LDS, pitch n, lower incl. diagonal);
// A is consumed.  One barrier per column: the trailing update divides by the pivot itself, so it reads only the
// UNscaled column j of A, which no thread writes in step j.  Returns false (uniformly) on a pivot <= floor.
__device__ __forceinline__ bool chol64_lds(double *A, double *Lm, int n, double floor_, int t, int nt) {
    for (int j = 0; j < n; ++j) {
        __syncthreads();
        const double d = A[j * n + j];
        if (!(d > floor_)) return false;
        const double inv = 1.0 / d, is = 1.0 / sqrt(d);
        const int rem = n - j; // rows j .. n-1
        for (int i = j + t; i < n; i += nt) Lm[i * n + j] = A[i * n + j] * is;
        // trailing (i, c), j < c <= i < n: flat index over the (rem-1) x (rem-1) lower triangle
        const int m1 = rem - 1, cnt = m1 * (m1 + 1) / 2;
        for (int e = t; e < cnt; e += nt) {
            int ii = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
            while ((ii + 1) * (ii + 2) / 2 <= e) ++ii;
            while (ii * (ii + 1) / 2 > e) --ii;
            const int cc = e - ii * (ii + 1) / 2;
            const int i = j + 1 + ii, c = j + 1 + cc;
            A[i * n + c] -= A[i * n + j] * A[c * n + j] * inv;
        }
    }
    __syncthreads();
    return true;
}

// Hinv32 (kp x kp float, zero on the padding) = safe_inverse(H64) for ONE matrix of valid order n <= 64; optional float64 copy
// (identity on the padding).
// dynamic LDS: 2 n^2 + n doubles.
__global__ __launch_bounds__(256) void safe_inverse64_small_kernel(const double *H, float *Hinv, int n, int kp, double pert, double *Hinv64) {
    extern __shared__ __attribute__((aligned(16))) double dsm[];
    double *A = dsm, *W = dsm + n * n, *inv = dsm + 2 * n * n;
    const int t = threadIdx.x, nt = 256, lane = t & 63, wid = t >> 6;
    __shared__ double red[4];
    double dmax = 0.0;
    for (int i = t; i < n; i += nt) dmax = fmax(dmax, fabs(H[i * kp + i]));
    for (int off = 32; off > 0; off >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, off, 64));
    if (lane == 0) red[wid] = dmax;
    __syncthreads();
    dmax = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    const double floor_ = 1.0e-13 * dmax;
    // 1) lambda_min(H) >= pert  <=>  H - pert I factors
    for (int idx = t; idx < n * n; idx += nt) {
        const int r = idx / n, c = idx % n;
        A[idx] = H[r * kp + c] - (r == c ? pert : 0.0);
    }
    const bool pd = chol64_lds(A, W, n, floor_, t, nt);
    __syncthreads();
    if (pd) {
        // 2) H = L L^T, X = L^-1 (column c by thread c), H^-1 = X^T X
        for (int idx = t; idx < n * n; idx += nt) A[idx] = H[(idx / n) * kp + idx % n];
        (void)chol64_lds(A, W, n, 0.0, t, nt);
        __syncthreads();
        // X into A (lower): four lanes share a column's dot products
        for (int idx = t; idx < n * n; idx += nt) A[idx] = 0.0;
        __syncthreads();
        if (t < n) {
            const int c = t;
            A[c * n + c] = 1.0 / W[c * n + c];
            for (int i = c + 1; i < n; ++i) {
                double s = 0.0;
                for (int q = c; q < i; ++q) s += W[i * n + q] * A[q * n + c];
                A[i * n + c] = -s / W[i * n + i];
            }
        }
        __syncthreads();
        for (int idx = t; idx < kp * kp; idx += nt) {
            const int r = idx / kp, c = idx % kp;
            double acc = 0.0;
            if (r < n && c < n)
                for (int q = (r > c ? r : c); q < n; ++q) acc += A[q * n + r] * A[q * n + c];
            Hinv[idx] = (float)acc;
            if (Hinv64) Hinv64[idx] = (r < n && c < n) ? acc : (r == c ? 1.0 : 0.0);
        }
        return;
    }
    // 3) the clamp acts: Q diag(1 / max(|lambda|, pert)) Q^T by one-sided Jacobi on B = H (rows of B = columns of H V)
    double *B = A, *Vt = W;
    for (int idx = t; idx < n * n; idx += nt) {
        const int r = idx / n, c = idx % n;
        B[idx] = H[r * kp + c];
        Vt[idx] = (r == c) ? 1.0 : 0.0;
    }
    __syncthreads();
    const int N = n + (n & 1);
    const double tol = 1.0e-15;
    for (int sweep = 0; sweep < 40 && N >= 2; ++sweep) {
        int rotated = 0;
        for (int s = 0; s < N - 1; ++s) {
            for (int pi = wid; pi < N / 2; pi += 4) {
                int p, q;
                if (pi == 0) { p = s; q = N - 1; }
                else { p = (s + pi) % (N - 1); q = (s - pi + (N - 1)) % (N - 1); }
                if (p >= n || q >= n) continue;
                double *bp = B + p * n, *bq = B + q * n;
                const double x = lane < n ? bp[lane] : 0.0, y = lane < n ? bq[lane] : 0.0; // n <= 64: one element per lane
                const double a = wave_sum_f64(x * x), b = wave_sum_f64(y * y), g = wave_sum_f64(x * y);
                if (fabs(g) > tol * sqrt(a * b) && a > 0.0 && b > 0.0) {
                    const double zeta = (b - a) / (2.0 * g);
                    const double tt = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
                    if (lane < n) {
                        bp[lane] = cs * x - sn * y;
                        bq[lane] = sn * x + cs * y;
                        const double u = Vt[p * n + lane], v = Vt[q * n + lane];
                        Vt[p * n + lane] = cs * u - sn * v;
                        Vt[q * n + lane] = sn * u + cs * v;
                    }
                    rotated = 1;
                }
            }
            __syncthreads();
        }
        if (!__syncthreads_or(rotated)) break;
    }
    for (int j = wid; j < n; j += 4) {
        const double x = lane < n ? B[j * n + lane] : 0.0;
        const double a = wave_sum_f64(x * x);
        if (lane == 0) {
            double sg = sqrt(a);
            if (sg < pert) sg = pert;
            inv[j] = 1.0 / sg;
        }
    }
    __syncthreads();
    for (int idx = t; idx < kp * kp; idx += nt) {
        const int r = idx / kp, c = idx % kp;
        double acc = 0.0;
        if (r < n && c < n)
            for (int j = 0; j < n; ++j) acc += inv[j] * Vt[j * n + r] * Vt[j * n + c];
        Hinv[idx] = (float)acc;
        if (Hinv64) Hinv64[idx] = (r < n && c < n) ? acc : (r == c ? 1.0 : 0.0);
    }
}

// ------------------------------------------------------------------------------------------ n > 64: matrix in L2
// Blocked right-looking Cholesky, ONE 1024-thread workgroup per matrix: workgroup b factors  H - shift[b] I  in its own
// workspace W_b (n x n doubles, pitch ld; lower triangle + diagonal hold L on exit).  flag[b] = 1 on a pivot <= floor.
// Panel width 32: (a) the diagonal block is factored in LDS (thread (i, c) of a 32 x 32 grid, one barrier per column),
// (b) the rows under it are solved against it one row per thread, and kept TRANSPOSED in LDS (Pt[t][row]) so that
// (c) the trailing update  W22 -= L21 L21^T  reads 4 x 4 register tiles with conflict-free 16-byte LDS reads.
// LDS: D 8.25 KB + Lb 8.25 KB + Pt 32 x PR doubles, PR = rows under the first panel (<= 992 -> 254 KB would not fit):
// up to n = 512 + 32 the panel lives in LDS (PT_ROWS = 512); larger n take the panel from global memory.
#define CMF_CHOL64_PT_ROWS 512
__global__ __launch_bounds__(1024) void chol64_kernel(const double *H, int n, int ldh, double *Wbase, int64_t wstride, int ld,
                                                      double shift0, double shift1, int *flag) {
    __shared__ double D[32 * 33];
    __shared__ double Lb[32 * 33];
    __shared__ __attribute__((aligned(16))) double Pt[32 * CMF_CHOL64_PT_ROWS];
    __shared__ double red[16];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    double *W = Wbase + (int64_t)blockIdx.x * wstride;
    const double shift = blockIdx.x == 0 ? shift0 : shift1;
    double dmax = 0.0;
    for (int i = t; i < n; i += 1024) dmax = fmax(dmax, fabs(H[(int64_t)i * ldh + i]));
    for (int off = 32; off > 0; off >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, off, 64));
    if (lane == 0) red[wid] = dmax;
    __syncthreads();
    dmax = 0.0;
    for (int q = 0; q < 16; ++q) dmax = fmax(dmax, red[q]);
    const double floor_ = shift != 0.0 ? 1.0e-13 * dmax : 0.0;
    for (int idx = t; idx < n * n; idx += 1024) {
        const int r = idx / n, c = idx % n;
        if (c <= r) W[(int64_t)r * ld + c] = H[(int64_t)r * ldh + c] - (r == c ? shift : 0.0);
    }
    if (t == 0) flag[blockIdx.x] = 0;
    __syncthreads();
    for (int j0 = 0; j0 < n; j0 += 32) {
        const int nb = n - j0 < 32 ? n - j0 : 32;
        // (a) diagonal block
        {
            const int i = t >> 5, c = t & 31;
            D[i * 33 + c] = (i < nb && c < nb && c <= i) ? W[(int64_t)(j0 + i) * ld + j0 + c] : (i == c ? 1.0 : 0.0);
            Lb[i * 33 + c] = 0.0;
            for (int j = 0; j < nb; ++j) {
                __syncthreads();
                const double d = D[j * 33 + j];
                if (!(d > floor_)) { // uniform: every thread reads the same pivot
                    if (t == 0) flag[blockIdx.x] = 1;
                    return;
                }
                if (c == j && i >= j) Lb[i * 33 + j] = D[i * 33 + j] / sqrt(d);
                if (c > j && i >= c) D[i * 33 + c] -= D[i * 33 + j] * D[c * 33 + j] / d;
            }
            __syncthreads();
            if (i < nb && c < nb && c <= i) W[(int64_t)(j0 + i) * ld + j0 + c] = Lb[i * 33 + c];
        }
        const int rbeg = j0 + nb, nrem = n - rbeg; // rows under the panel
        if (nrem <= 0) break;
        const bool in_lds = nrem <= CMF_CHOL64_PT_ROWS;
        // (b) L21 = A21 L11^-T : row per thread
        for (int r = t; r < nrem; r += 1024) {
            double x[32];
            double *row = W + (int64_t)(rbeg + r) * ld + j0;
#pragma unroll
            for (int c = 0; c < 32; ++c) x[c] = c < nb ? row[c] : 0.0;
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                if (c < nb) {
                    double s = x[c];
#pragma unroll
                    for (int q = 0; q < c; ++q) s -= x[q] * Lb[c * 33 + q];
                    x[c] = s / Lb[c * 33 + c];
                }
            }
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                if (c < nb) row[c] = x[c];
                if (in_lds) Pt[c * CMF_CHOL64_PT_ROWS + r] = x[c];
            }
        }
        __syncthreads();
        // (c) trailing update, 4 x 4 tiles over the lower triangle (tile row >= tile column)
        const int nt4 = (nrem + 3) / 4;
        const int ntile = nt4 * (nt4 + 1) / 2;
        for (int e = t; e < ntile; e += 1024) {
            int ti = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
            while ((ti + 1) * (ti + 2) / 2 <= e) ++ti;
            while (ti * (ti + 1) / 2 > e) --ti;
            const int tc = e - ti * (ti + 1) / 2;
            double acc[4][4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
            for (int q = 0; q < nb; ++q) {
                double ri[4], rc[4];
                if (in_lds) {
                    const f64x2 a0 = *reinterpret_cast<const f64x2 *>(Pt + q * CMF_CHOL64_PT_ROWS + 4 * ti);
                    const f64x2 a1 = *reinterpret_cast<const f64x2 *>(Pt + q * CMF_CHOL64_PT_ROWS + 4 * ti + 2);
                    const f64x2 b0 = *reinterpret_cast<const f64x2 *>(Pt + q * CMF_CHOL64_PT_ROWS + 4 * tc);
                    const f64x2 b1 = *reinterpret_cast<const f64x2 *>(Pt + q * CMF_CHOL64_PT_ROWS + 4 * tc + 2);
                    ri[0] = a0[0]; ri[1] = a0[1]; ri[2] = a1[0]; ri[3] = a1[1];
                    rc[0] = b0[0]; rc[1] = b0[1]; rc[2] = b1[0]; rc[3] = b1[1];
                } else {
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        ri[a] = 4 * ti + a < nrem ? W[(int64_t)(rbeg + 4 * ti + a) * ld + j0 + q] : 0.0;
                        rc[a] = 4 * tc + a < nrem ? W[(int64_t)(rbeg + 4 * tc + a) * ld + j0 + q] : 0.0;
                    }
                }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] += ri[a] * rc[b];
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int i = 4 * ti + a, c = 4 * tc + b;
                    if (i < nrem && c <= i) W[(int64_t)(rbeg + i) * ld + rbeg + c] -= acc[a][b];
                }
        }
        __syncthreads();
    }
}

// Register-resident Cholesky for n <= 32 NB (NB = 4: n <= 128, NB = 8: n <= 256): the float64 twin of chol_solve_kernel's
// factorisation.  1024 threads as a 32 x 32 grid; thread (ti, tc) owns the elements (i, c) = (32 a + ti, 32 b + tc), b <= a,
// of the lower triangle: NB (NB + 1) / 2 doubles (36 at n = 256) that never leave its registers.  A column step: the
// owners of column j publish it UNscaled, with the pivot, into one of two LDS buffers; after ONE barrier every thread
// fetches its NB row values and NB column values and applies the rank-1 update divided by the pivot (so it needs no
// scaled column), while the owners scale their own column.  256 steps of ~450 cycles: ~50 us, against 530 us for the
// blocked kernel below working out of L2.  Workgroup b factors H - shift_b I; L goes to W_b (lower triangle).
template <int NB>
__global__ __launch_bounds__(1024) void chol64_reg_kernel(const double *H, int n, int ldh, double *Wbase, int64_t wstride, int ld,
                                                         double shift0, double shift1, int *flag, int64_t hstride = 0, int per_matrix = 0) {
    // batched use (cmf_refine64.hip.h): workgroup b works on matrix b / per_matrix (stride hstride) with shift (b % per_matrix == 0 ?
    // shift0 : shift1); per_matrix = 0: the one matrix H, workgroup 0 with shift0, the others with shift1
    constexpr int CB = 32 * NB + 2;
    __shared__ __attribute__((aligned(16))) double col[2 * CB];
    __shared__ double red[16];
    const int t = threadIdx.x, ti = t & 31, tc = t >> 5, lane = t & 63, wid = t >> 6;
    double *W = Wbase + (int64_t)blockIdx.x * wstride;
    if (per_matrix > 0) H += (int64_t)(blockIdx.x / per_matrix) * hstride;
    const double shift = (per_matrix > 0 ? (blockIdx.x % per_matrix == 0) : (blockIdx.x == 0)) ? shift0 : shift1;
    double dmax = 0.0;
    for (int i = t; i < n; i += 1024) dmax = fmax(dmax, fabs(H[(int64_t)i * ldh + i]));
    for (int off = 32; off > 0; off >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, off, 64));
    if (lane == 0) red[wid] = dmax;
    __syncthreads();
    dmax = 0.0;
    for (int q = 0; q < 16; ++q) dmax = fmax(dmax, red[q]);
    const double floor_ = shift != 0.0 ? 1.0e-13 * dmax : 0.0;
    double M[NB][NB];
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) {
            const int i = 32 * a + ti, c = 32 * b + tc;
            double v = (i == c) ? 1.0 : 0.0; // identity outside the valid block
            if (i < n && c < n) v = H[(int64_t)c * ldh + i] - (i == c ? shift : 0.0); // H symmetric: read along the row of c
            M[a][b] = v;
        }
    if (t == 0) flag[blockIdx.x] = 0;
    bool ok = true;
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) {
        for (int jl = 0; jl < 32; ++jl) {
            const int j = 32 * jb + jl;
            if (j >= n) break;
            double *cb = col + (jl & 1) * CB;
            if (tc == jl) { // owners of column j: rows i = 32 a + ti
#pragma unroll
                for (int a = 0; a < NB; ++a) cb[a * 32 + ti] = (a >= jb && 32 * a + ti > j) ? M[a][jb] : 0.0; // [a][ti]: conflict-free
                if (ti == jl) cb[32 * NB] = M[jb][jb]; // the pivot sits in thread (jl, jl)
            }
            __syncthreads();
            const double d = cb[32 * NB];
            if (!(d > floor_)) { ok = false; break; }
            const double inv = 1.0 / d;
            double li[NB], lc[NB];
#pragma unroll
            for (int a = 0; a < NB; ++a) { li[a] = cb[a * 32 + ti]; lc[a] = cb[a * 32 + tc] * inv; }
#pragma unroll
            for (int a = jb; a < NB; ++a)
#pragma unroll
                for (int b = jb; b <= a; ++b)
                    if (b > jb || tc > jl) M[a][b] -= li[a] * lc[b];
            if (tc == jl) { // scale the finished column: L[i][j] = A[i][j] / sqrt(d), L[j][j] = sqrt(d)
                const double is = 1.0 / sqrt(d);
#pragma unroll
                for (int a = jb; a < NB; ++a) {
                    const int i = 32 * a + ti;
                    if (i > j) M[a][jb] *= is;
                    else if (i == j) M[a][jb] = sqrt(d);
                }
            }
        }
        if (!ok) break;
    }
    if (!ok) {
        if (t == 0) flag[blockIdx.x] = 1;
        return;
    }
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) {
            const int i = 32 * a + ti, c = 32 * b + tc;
            if (i < n && c <= i) W[(int64_t)i * ld + c] = M[a][b];
        }
}

// Xt[c][i] = (L^-1)[i][c]  (row c of Xt = column c of the inverse of the lower-triangular L; zero for i < c).
// Workgroup = 16 columns x 16 lanes; rows of L pass through LDS rp at a time; y lives in LDS ([n][16]).
// Rows of Xt beyond n are zero.
__global__ __launch_bounds__(256) void tri_inverse64_kernel(const double *L, int n, int ld, double *Xt, int ldx, int kp, int rp) {
    extern __shared__ __attribute__((aligned(16))) double tsm[];
    double *Y = tsm;               // [n][16]
    double *Ls = tsm + (size_t)n * 16; // [rp][n + 2]
    const int lp = n + 2;
    const int t = threadIdx.x, q = t & 15, cl = t >> 4;
    const int c0 = blockIdx.x * 16, c = c0 + cl;
    for (int i = t; i < n * 16; i += 256) Y[i] = 0.0;
    for (int i0 = (c0 / rp) * rp; i0 < n; i0 += rp) {
        const int nr = n - i0 < rp ? n - i0 : rp;
        __syncthreads();
        for (int idx = t; idx < nr * n; idx += 256) {
            const int r = idx / n, col = idx % n;
            Ls[r * lp + col] = col <= i0 + r ? L[(int64_t)(i0 + r) * ld + col] : 0.0;
        }
        __syncthreads();
        for (int r = 0; r < nr; ++r) {
            const int i = i0 + r;
            if (i >= c0) {
                double s = 0.0;
                for (int u = c0 + q; u < i; u += 16) s += Ls[r * lp + u] * Y[u * 16 + cl];
                s = group16_sum_f64(s);
                if (q == 0 && c < n && i >= c) Y[i * 16 + cl] = ((i == c ? 1.0 : 0.0) - s) / Ls[r * lp + i];
            }
            __syncthreads();
        }
    }
    for (int idx = t; idx < 16 * kp; idx += 256) {
        const int cc = idx / kp, i = idx % kp;
        if (c0 + cc < kp) Xt[(int64_t)(c0 + cc) * ldx + i] = (i < n && c0 + cc < n) ? Y[i * 16 + cc] : 0.0;
    }
}

// 16-lane sums of doubles on the VALU (DPP row operations on the two halves of the double) instead of ds_bpermute
template <int CTRL>
__device__ __forceinline__ double dpp_move_f64(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ double group16_sum_f64_dpp(double v) {
    v += dpp_move_f64<0xB1>(v);  // quad_perm [1,0,3,2]
    v += dpp_move_f64<0x4E>(v);  // quad_perm [2,3,0,1]
    v += dpp_move_f64<0x141>(v); // row_half_mirror
    v += dpp_move_f64<0x140>(v); // row_mirror
    return v;
}

// The same inverse with the solution in REGISTERS (n <= 16 NM; NM = 8: n <= 128, NM = 16: n <= 256): lane q of a column's
// 16-lane group holds y_u for the rows u = q (mod 16).  A row step is 16 (NM) LDS reads of the staged row of L, NM fused
// multiply-adds, four DPP additions and one multiplication by the pre-inverted diagonal -- no barrier inside a panel
// (the 16 lanes of a column sit in one wave), ~250 cycles per row instead of ~2000.
template <int NM>
__global__ __launch_bounds__(256) void tri_inverse64_reg_kernel(const double *L, int n, int ld, double *Xt, int ldx, int kp, int64_t lstride = 0,
                                                               int64_t xstride = 0) {
    // batched (grid y): matrix images lstride / xstride doubles apart
    L += (int64_t)blockIdx.y * lstride;
    Xt += (int64_t)blockIdx.y * xstride;
    constexpr int RP = 32;
    __shared__ __attribute__((aligned(16))) double Ls[RP * (16 * NM + 2)];
    __shared__ double dinv[RP];
    constexpr int LP = 16 * NM + 2;
    const int t = threadIdx.x, q = t & 15, cl = t >> 4;
    const int c0 = blockIdx.x * 16, c = c0 + cl;
    double y[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) y[m] = 0.0;
    for (int i0 = (c0 / RP) * RP; i0 < n; i0 += RP) {
        const int nr = n - i0 < RP ? n - i0 : RP;
        __syncthreads();
        for (int idx = t; idx < RP * 16 * NM; idx += 256) {
            const int r = idx / (16 * NM), col = idx % (16 * NM);
            Ls[r * LP + col] = (r < nr && col < i0 + r && col < n) ? L[(int64_t)(i0 + r) * ld + col] : 0.0; // strictly lower part
        }
        if (t < RP) dinv[t] = t < nr ? 1.0 / L[(int64_t)(i0 + t) * ld + i0 + t] : 0.0;
        __syncthreads();
        for (int r = 0; r < nr; ++r) {
            const int i = i0 + r;
            if (i < c0) continue; // rows above this block of columns: y stays 0 (uniform over the workgroup)
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int m = 0; m < NM; m += 2) {
                s0 += Ls[r * LP + q + 16 * m] * y[m];
                s1 += Ls[r * LP + q + 16 * (m + 1)] * y[m + 1];
            }
            const double s = group16_sum_f64_dpp(s0 + s1);
            const double val = (i >= c && c < n) ? ((i == c ? 1.0 : 0.0) - s) * dinv[r] : 0.0;
            const int qi = i & 15, mi = i >> 4;
#pragma unroll
            for (int m = 0; m < NM; ++m)
                if (m == mi && q == qi) y[m] = val;
        }
    }
    if (c < kp) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int u = q + 16 * m;
            if (u < kp) Xt[(int64_t)c * ldx + u] = (u < n && c < n) ? y[m] : 0.0;
        }
    }
    // columns of a k_pad wider than 16 NM (cannot happen: k_pad <= 16 NM by construction of the caller)
}

// C = alpha op(A) op(B) + beta D + gamma I, all kp x kp float64 (pitch kp), kp a multiple of 32.
//   TRANS_B = false: C = A B;  true: C = A B^T.   Optional float32 copy of C (C32, zeroing rows / columns >= nvalid).
// Runs only while  iter < *limit  when limit is given (device-side predicate of the Newton-Schulz chain).
// Workgroup = 4 waves, 32 x 32 output tile (one 16 x 16 MFMA block per wave), 32-deep K steps through LDS.
// LDS pitches: k-contiguous tiles 34 doubles, n-contiguous B tile 48 doubles (conflict-free ds_read_b64 fragments).
template <bool TRANS_B>
__global__ __launch_bounds__(256) void gemm64_kernel(const double *A, const double *B, double *C, const double *D, double alpha, double beta,
                                                     double gamma, int kp, float *C32, int nvalid, const int *limit, int iter, int64_t bstride = 0) {
    if (limit && iter >= *limit) return;
    if (bstride) { // batched (grid z): the same operation on matrix images bstride apart
        const int64_t off = (int64_t)blockIdx.z * bstride;
        A += off; B += off;
        if (C) C += off;
        if (D) D += off;
    }
    constexpr int LA = 34, LBN = 48;
    __shared__ __attribute__((aligned(16))) double As[32 * LA];
    __shared__ __attribute__((aligned(16))) double Bs[32 * LBN];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wi = w >> 1, wj = w & 1;
    const int l15 = lane & 15, lk = lane >> 4;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < kp; k0 += 32) {
        __syncthreads();
        for (int idx = t; idx < 512; idx += 256) { // 32 rows x 16 double2
            const int r = idx >> 4, c2 = idx & 15;
            *reinterpret_cast<f64x2 *>(As + r * LA + 2 * c2) = *reinterpret_cast<const f64x2 *>(A + (int64_t)(m0 + r) * kp + k0 + 2 * c2);
            if (TRANS_B) *reinterpret_cast<f64x2 *>(Bs + r * LA + 2 * c2) = *reinterpret_cast<const f64x2 *>(B + (int64_t)(n0 + r) * kp + k0 + 2 * c2);
            else *reinterpret_cast<f64x2 *>(Bs + r * LBN + 2 * c2) = *reinterpret_cast<const f64x2 *>(B + (int64_t)(k0 + r) * kp + n0 + 2 * c2);
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const double a = As[(wi * 16 + l15) * LA + 4 * s + lk];
            const double b = TRANS_B ? Bs[(wj * 16 + l15) * LA + 4 * s + lk] : Bs[(4 * s + lk) * LBN + wj * 16 + l15];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int r = m0 + wi * 16 + lk + 4 * reg, c = n0 + wj * 16 + l15;
        double v = alpha * acc[reg];
        if (D) v += beta * D[(int64_t)r * kp + c];
        if (r == c) v += gamma;
        if (C) C[(int64_t)r * kp + c] = v;
        if (C32) C32[(int64_t)r * kp + c] = (r < nvalid && c < nvalid) ? (float)v : 0.f;
    }
}

// ------------------------------------------------------------------------------------------ batched 256-wide float64 product
// C_b = alpha A_b B_b + beta D_b + gamma I for stacked kp x kp float64 matrices (kp a multiple of 128), bstride doubles apart:
// the Newton-Schulz polynomials of the batched float64 refinement (cmf_newton.hip.h, refine_rows64_batched: ~74 such products
// per clamped row).  gemm64_kernel gives a workgroup a 32 x 32 tile -- 4 flop per byte it stages, 8 MB of L2 -> LDS traffic per
// 256^3 product: the refinement of C3's 32768 rows of U at l2 = 0 moved 19 TB per iteration through it (26 TF/s).  Here a
// workgroup of eight waves owns a 128 x 128 tile (wave tile 32 x 64 = 2 x 4 blocks of v_mfma_f64_16x16x4_f64, six fragment reads
// per eight MFMAs, 16 flop per staged byte), 32-deep K-steps, the next step's operands waiting in registers under the MFMAs.
// LDS: A tile [128][34] doubles (k contiguous), B tile [32][144] doubles (n contiguous): 71 KB, two workgroups per CU.
constexpr int GEMM64_TILE128_LDS = (128 * 34 + 32 * 144) * (int)sizeof(double);
__global__ __launch_bounds__(512, 4) void gemm64_tile128_kernel(const double *A, const double *B, double *C, const double *D, double alpha,
                                                                double beta, double gamma, int kp, int64_t bstride) {
    constexpr int LA = 34, LB = 144;
    extern __shared__ __attribute__((aligned(16))) double g64sm[];   // GEMM64_TILE128_LDS bytes (dynamic: above the 64 KB static limit)
    double *As = g64sm, *Bs = g64sm + 128 * LA;
    {
        const int64_t off = (int64_t)blockIdx.z * bstride;
        A += off; B += off; C += off;
        if (D) D += off;
    }
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wi = w >> 1, wj = w & 1;               // wave rows wi * 32 .., columns wj * 64 ..
    const int l15 = lane & 15, lk = lane >> 4;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    f64x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
    // staging: A tile 128 rows x 16 double2, B tile 32 rows x 64 double2: four of each per thread
    f64x2 ra[4], rb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = t + 512 * q;
            ra[q] = *reinterpret_cast<const f64x2 *>(A + (int64_t)(m0 + (idx >> 4)) * kp + k0 + 2 * (idx & 15));
            rb[q] = *reinterpret_cast<const f64x2 *>(B + (int64_t)(k0 + (idx >> 6)) * kp + n0 + 2 * (idx & 63));
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = t + 512 * q;
            *reinterpret_cast<f64x2 *>(As + (idx >> 4) * LA + 2 * (idx & 15)) = ra[q];
            *reinterpret_cast<f64x2 *>(Bs + (idx >> 6) * LB + 2 * (idx & 63)) = rb[q];
        }
    };
    gload(0);
    for (int k0 = 0; k0 < kp; k0 += 32) {
        __syncthreads();            // the previous step's fragment reads are done
        lstore();
        __syncthreads();
        if (k0 + 32 < kp) gload(k0 + 32);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            double a[2], b[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[(wi * 32 + 16 * i + l15) * LA + 4 * s + lk];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[(4 * s + lk) * LB + wj * 64 + 16 * j + l15];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = m0 + wi * 32 + 16 * i + lk + 4 * reg, c = n0 + wj * 64 + 16 * j + l15;
                double v = alpha * acc[i][j][reg];
                if (D) v += beta * D[(int64_t)r * kp + c];
                if (r == c) v += gamma;
                C[(int64_t)r * kp + c] = v;
            }
}

// ------------------------------------------------------------------------------------------ factor times float64 matrix
// O32[rows x kp] = scale * A32[rows x kp] * B64[kp x kp], float64 products and sums on v_mfma_f64_16x16x4_f64, rounded ONCE to
// float32.  This is the "pre-conditioned operand" of the re-associated Newton sweep (cmf_newton.hip.h): with the safe inverse
// Hinv of the ONE shared Hessian the reference's  F - grad Hinv,  grad = s (F G - T O) + l2 F  (pycmf/cmf_solvers.py:399-410,
// :436-450, :321-326)  equals  F (I - H Hinv) + s T (O Hinv):  O Hinv is formed here in float64, so the float32 rounding of the
// data contraction is no longer multiplied by cond(H).
// Workgroup = 4 waves, output tile 64 x NW (NW = 128: wave tile 32 x 64 = 2 x 4 MFMA blocks, six fragment reads per eight MFMAs),
// the reduction in steps of 32 through LDS:
// A tile [64][36] floats (k contiguous; 16 rows x 4 k of a ds_read_b32 group fall on 64 distinct banks), B tile [32][80] doubles
// (the two 16-lane halves of a ds_read_b64 group read rows k and k + 1, 128 bytes apart mod 256).
template <int NW, int RW = 64>
__global__ __launch_bounds__(4 * RW) void factor_times64_kernel(const float *A, const double *B, float *O, int kp, double scale) {
    static_assert(NW == 128 || NW == 64 || NW == 32, "column tile 128 (k_pad >= 128), 64 (k_pad = 64) or 32 (k_pad = 32)");
    static_assert(RW == 64 || RW == 128, "row tile 64 (four waves) or 128 (eight waves: half the B traffic per flop)");
    // wave (wi, wj) owns rows wi * 32 .. + 31 and columns wj * NW / 2 .. : 2 x JB MFMA blocks
    constexpr int NT = 4 * RW, LA = 36, LB = NW + 16, WN = NW / 2, JB = WN / 16;
    constexpr int AL = RW * 8 / NT, BL = 32 * (NW / 2) / NT; // float4 loads of the A tile / double2 loads of the B tile per thread
    __shared__ __attribute__((aligned(16))) float As[RW * LA];
    __shared__ __attribute__((aligned(16))) double Bs[32 * LB];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wi = w >> 1, wj = w & 1;
    const int l15 = lane & 15, lk = lane >> 4;
    const int64_t m0 = (int64_t)blockIdx.y * RW;
    const int n0 = blockIdx.x * NW;
    f64x4 acc[2][JB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < JB; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < kp; k0 += 32) {
        f32x4 av[AL];
        f64x2 bv[BL];
#pragma unroll
        for (int q = 0; q < AL; ++q) { // RW rows x 8 float4
            const int idx = t + NT * q, r = idx >> 3, c4 = idx & 7;
            av[q] = *reinterpret_cast<const f32x4 *>(A + (m0 + r) * kp + k0 + 4 * c4);
        }
#pragma unroll
        for (int q = 0; q < BL; ++q) { // 32 rows x NW / 2 double2
            const int idx = t + NT * q, r = idx / (NW / 2), c2 = idx % (NW / 2);
            bv[q] = *reinterpret_cast<const f64x2 *>(B + (int64_t)(k0 + r) * kp + n0 + 2 * c2);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < AL; ++q) {
            const int idx = t + NT * q, r = idx >> 3, c4 = idx & 7;
            *reinterpret_cast<f32x4 *>(As + r * LA + 4 * c4) = av[q];
        }
#pragma unroll
        for (int q = 0; q < BL; ++q) {
            const int idx = t + NT * q, r = idx / (NW / 2), c2 = idx % (NW / 2);
            *reinterpret_cast<f64x2 *>(Bs + r * LB + 2 * c2) = bv[q];
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            double a[2], b[JB];
#pragma unroll
            for (int q = 0; q < 2; ++q) a[q] = (double)As[(wi * 32 + 16 * q + l15) * LA + 4 * s + lk];
#pragma unroll
            for (int q = 0; q < JB; ++q) b[q] = Bs[(4 * s + lk) * LB + wj * WN + 16 * q + l15];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < JB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                O[(m0 + wi * 32 + 16 * i + lk + 4 * reg) * kp + n0 + wj * WN + 16 * j + l15] = (float)(scale * acc[i][j][reg]);
}

// identity on the padding (rows / columns >= n) of a kp x kp float64 matrix
__global__ __launch_bounds__(256) void pad_identity64_kernel(double *M, int kp, int n) {
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < kp * kp; idx += gridDim.x * 256) {
        const int r = idx / kp, c = idx % kp;
        if (r >= n || c >= n) M[idx] = (r == c) ? 1.0 : 0.0;
    }
}

__global__ __launch_bounds__(256) void f32_to_f64_kernel(double *out, const float *in, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = (double)in[i];
}

// Newton-Schulz start: Bm = H - pert I on the valid block (padding: c on the diagonal), c = min(||B||_F, ||B||_inf),
// X = Bm / c;  out[0] = c.  One 1024-thread workgroup (H is symmetric: column sums = row sums).
__global__ __launch_bounds__(1024) void ns64_prepare_kernel(const double *H, double *Bm, double *X, int n, int kp, double pert, double *out,
                                                           int64_t bstride = 0) {
    __shared__ double red_f[16], red_m[16];
    if (bstride) { // batched (grid x): matrix images bstride apart, one norm per image
        const int64_t off = (int64_t)blockIdx.x * bstride;
        H += off; Bm += off; X += off; out += blockIdx.x;
    }
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    double fro = 0.0, cmaxv = 0.0;
    for (int col = t; col < n; col += 1024) {
        double cs = 0.0;
        for (int r = 0; r < n; ++r) {
            const double v = H[(int64_t)r * kp + col] - (r == col ? pert : 0.0);
            fro += v * v;
            cs += fabs(v);
        }
        cmaxv = fmax(cmaxv, cs);
    }
    for (int off = 32; off > 0; off >>= 1) {
        fro += __shfl_xor(fro, off, 64);
        cmaxv = fmax(cmaxv, __shfl_xor(cmaxv, off, 64));
    }
    if (lane == 0) { red_f[wid] = fro; red_m[wid] = cmaxv; }
    __syncthreads();
    fro = 0.0; cmaxv = 0.0;
    for (int q = 0; q < 16; ++q) { fro += red_f[q]; cmaxv = fmax(cmaxv, red_m[q]); }
    double c = fmin(sqrt(fro), cmaxv);
    if (!(c > 1e-300)) c = 1.0;
    const double ci = 1.0 / c;
    if (t == 0) out[0] = c;
    for (int idx = t; idx < kp * kp; idx += 1024) {
        const int r = idx / kp, col = idx % kp;
        double v = (r == col) ? c : 0.0;
        if (r < n && col < n) v = H[idx] - (r == col ? pert : 0.0);
        Bm[idx] = v;
        X[idx] = v * ci;
    }
}

// ------------------------------------------------------------------ float64 refinement of single rows (cmf_newton.hip.h: refine_rows64)
// A row whose float32 Hessian went through the spectral clamp with ||H||_F / pert beyond what float32 resolves is redone here
// in float64, one row at a time: weights, Hessian, safe inverse (shared_inverse64), step.  The factor rows themselves are the
// float32 values both sides of the comparison start from, so every product below is exact and only float64 sums round.

// Per sample j of one row (list position q): z = o_j . f in float64, residual r[q] = link(z) - t_ij, Hessian weight w[q] = 1 (linear)
// or sigma'(z) (logit); one wave per sample.  t_ij = T[t_off + j * t_col] (a zero buffer with stride 0 on natively sparse sides,
// whose stored values enter through row_sparse_grad64_kernel).                                (:419-428, :459-484, :495-506)
__global__ __launch_bounds__(256) void row_terms64_kernel(const float *O, int kp, const float *f, const int32_t *list, int s, int link,
                                                          const float *T, int64_t t_off, int64_t t_col, double *r, double *w) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= s) return;
    const int64_t j = list ? list[q] : q;
    const float *o = O + j * kp;
    double z = 0.0;
    for (int e = lane; e < kp; e += 64) z += (double)o[e] * (double)f[e];
    for (int off = 32; off > 0; off >>= 1) z += __shfl_xor(z, off, 64);
    if (lane == 0) {
        const double t = (double)T[t_off + j * t_col];
        if (link) {
            const double sg = 1.0 / (1.0 + exp(-z));
            r[q] = sg - t;
            w[q] = sg * (1.0 - sg);
        } else {
            r[q] = z - t;
            w[q] = 1.0;
        }
    }
}

// g[col] (+)= scale * sum_q r[q] o_{j(q)}[col]: 32 columns per workgroup, 64 samples per LDS stage, eight partial sums per column
// (sample position mod 8) added in a fixed order -- deterministic.
__global__ __launch_bounds__(256) void row_grad64_kernel(const float *O, int kp, const int32_t *list, int s, const double *r, double scale,
                                                         double *g, int first) {
    __shared__ float so[64][33];
    __shared__ double sr[64];
    __shared__ double part[8][32];
    const int t = threadIdx.x, c0 = blockIdx.x * 32, col = t & 31, ph = t >> 5;
    double acc = 0.0;
    for (int j0 = 0; j0 < s; j0 += 64) {
        __syncthreads();
        for (int e = t; e < 64 * 32; e += 256) {
            const int jj = e >> 5, q = e & 31;
            so[jj][q] = (j0 + jj < s) ? O[(int64_t)(list ? list[j0 + jj] : j0 + jj) * kp + c0 + q] : 0.f;
        }
        if (t < 64) sr[t] = (j0 + t < s) ? r[j0 + t] : 0.0;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += sr[ph + 8 * u] * (double)so[ph + 8 * u][col];
    }
    part[ph][col] = acc;
    __syncthreads();
    if (t < 32) {
        double v = 0.0;
        for (int u = 0; u < 8; ++u) v += part[u][t];
        v *= scale;
        g[c0 + t] = first ? v : g[c0 + t] + v;
    }
}

// natively sparse side: g[col] -= scale * sum over the stored values t_ij of data row `row` with j in S_i (ascending list L, or all)
// of t_ij o_j[col]; one thread per column, entries in storage order (see csr_sampled_sub_kernel)
__global__ __launch_bounds__(256) void row_sparse_grad64_kernel(const int64_t *indptr, const int32_t *idx, const float *val, int64_t row,
                                                                const float *O, int kp, const int32_t *L, int64_t per, double scale,
                                                                double *g) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= kp) return;
    const int64_t beg = indptr[row], end = indptr[row + 1];
    double acc = 0.0;
    for (int64_t q = beg; q < end; ++q) {
        const int32_t j = idx[q];
        if (L) {
            int64_t lo = 0, hi = per;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (L[mid] < j) lo = mid + 1;
                else hi = mid;
            }
            if (lo >= per || L[lo] != j) continue;
        }
        acc += (double)val[q] * (double)O[(int64_t)j * kp + col];
    }
    g[col] -= scale * acc;
}

// g += l1 sign(f) + l2 f on the valid columns, zero on the padding                                        (:399-400, :420, :498)
__global__ __launch_bounds__(256) void row_grad_finish64_kernel(double *g, const float *f, double l1, double l2, int n, int kp) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= kp) return;
    if (col >= n) { g[col] = 0.0; return; }
    const double fv = (double)f[col];
    g[col] += l1 * (fv > 0.0 ? 1.0 : (fv < 0.0 ? -1.0 : 0.0)) + l2 * fv;
}

// H (kp x kp float64) (+)= scale * sum_j w[j] o_j o_j^T over the s samples of one row: one 32 x 32 tile per workgroup, samples
// staged through LDS 64 at a time, every thread four entries; samples are added in list order (deterministic).
// first != 0: the tile starts from diag on the diagonal of the valid block plus sscale * S (S: float64 shared part, nullable).
__global__ __launch_bounds__(256) void weighted_gram64_kernel(const float *O, int kp, int n, const int32_t *list, int s, const double *w,
                                                              double scale, double *H, int first, double diag, const double *S,
                                                              double sscale) {
    __shared__ float sa[64][33], sb[64][33];
    __shared__ double sw[64];
    const int t = threadIdx.x, a0 = blockIdx.y * 32, b0 = blockIdx.x * 32;
    const int ta = (t >> 4) * 2, tb = (t & 15) * 2;
    double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    for (int j0 = 0; j0 < s; j0 += 64) {
        __syncthreads();
        for (int e = t; e < 64 * 32; e += 256) {
            const int jj = e >> 5, q = e & 31;
            float va = 0.f, vb = 0.f;
            if (j0 + jj < s) {
                const float *o = O + (int64_t)(list ? list[j0 + jj] : j0 + jj) * kp;
                va = o[a0 + q];
                vb = o[b0 + q];
            }
            sa[jj][q] = va;
            sb[jj][q] = vb;
        }
        if (t < 64) sw[t] = (j0 + t < s) ? w[j0 + t] : 0.0;
        __syncthreads();
        for (int jj = 0; jj < 64; ++jj) {
            const double wj = sw[jj];
            const double x0 = wj * (double)sa[jj][ta], x1 = wj * (double)sa[jj][ta + 1];
            const double y0 = (double)sb[jj][tb], y1 = (double)sb[jj][tb + 1];
            acc[0][0] += x0 * y0; acc[0][1] += x0 * y1;
            acc[1][0] += x1 * y0; acc[1][1] += x1 * y1;
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jx = 0; jx < 2; ++jx) {
            const int r = a0 + ta + i, cidx = b0 + tb + jx;
            const int64_t at = (int64_t)r * kp + cidx;
            double v = scale * acc[i][jx];
            if (first) {
                if (r < n && cidx < n) {
                    if (r == cidx) v += diag;
                    if (S) v += sscale * S[at];
                } else v = 0.0;
            } else v += H[at];
            H[at] = v;
        }
}

// step[q] = sum_r g[r] Hinv[r][q]   (row vector times matrix, :321-326) in float64, rounded once to float32
__global__ __launch_bounds__(256) void rowvec_mat64_kernel(const double *g, const double *Hinv, float *step, int n, int kp) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= kp) return;
    double a = 0.0;
    if (q < n)
        for (int r = 0; r < n; ++r) a += g[r] * Hinv[(int64_t)r * kp + q];
    step[q] = (float)a;
}

} // namespace cmfk

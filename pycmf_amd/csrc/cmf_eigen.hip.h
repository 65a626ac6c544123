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

__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); } // VALU only (DPP + permlane swaps)

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
// One SINGLE matrix (the shared Hessian of the linear-link sweeps) whose spectrum dips under the
// perturbation: the same Hestenes sweep spread over the whole chip, one launch per round-robin step.
// A step's n/2 column pairs are independent, so each gets its own one-wave workgroup; the kernel
// boundary between steps is what makes a step's rotations visible to the next (no in-kernel grid
// barrier, no cross-XCD coherence protocol).  ~4 us per step: n = 256 -> ~1 ms per sweep instead of
// ~30 ms for the one-workgroup kernel working out of L2.
__global__ __launch_bounds__(64) void jacobi_init_kernel(const float *H, float *B, float *Vt, int n, int kp) {
    for (int idx = blockIdx.x * 64 + threadIdx.x; idx < n * n; idx += gridDim.x * 64) {
        const int r = idx / n, c = idx % n;
        B[idx] = H[r * kp + c];
        Vt[idx] = (r == c) ? 1.0f : 0.0f;
    }
}

__global__ __launch_bounds__(64) void jacobi_pair_step_kernel(float *B, float *Vt, int n, int N, int s, int *rotated) {
    const int pi = blockIdx.x, lane = threadIdx.x;
    int p, q;
    if (pi == 0) { p = s; q = N - 1; }
    else { p = (s + pi) % (N - 1); q = (s - pi + (N - 1)) % (N - 1); }
    if (p >= n || q >= n) return;
    float *bp = B + p * n, *bq = B + q * n;
    float a = 0.f, b = 0.f, g = 0.f;
    for (int e = lane; e < n; e += 64) {
        const float x = bp[e], y = bq[e];
        a += x * x; b += y * y; g += x * y;
    }
    a = wave_sum(a); b = wave_sum(b); g = wave_sum(g);
    if (fabsf(g) > 2.0e-7f * sqrtf(a * b) && a > 0.f && b > 0.f) {
        const float zeta = (b - a) / (2.0f * g);
        const float tt = copysignf(1.0f, zeta) / (fabsf(zeta) + sqrtf(1.0f + zeta * zeta));
        const float cs = 1.0f / sqrtf(1.0f + tt * tt), sn = cs * tt;
        float *vp = Vt + p * n, *vq = Vt + q * n;
        for (int e = lane; e < n; e += 64) {
            const float x = bp[e], y = bq[e];
            bp[e] = cs * x - sn * y; bq[e] = sn * x + cs * y;
            const float u = vp[e], w = vq[e];
            vp[e] = cs * u - sn * w; vq[e] = sn * u + cs * w;
        }
        if (lane == 0) *rotated = 1;
    }
}

// inv[j] = 1 / max(||row j of (HV)^T||, pert)
__global__ __launch_bounds__(64) void jacobi_sigma_kernel(const float *B, float *inv, int n, float pert) {
    const int j = blockIdx.x, lane = threadIdx.x;
    float a = 0.f;
    for (int e = lane; e < n; e += 64) { const float x = B[j * n + e]; a += x * x; }
    a = wave_sum(a);
    if (lane == 0) { float sg = sqrtf(a); if (sg < pert) sg = pert; inv[j] = 1.0f / sg; }
}

// O = V diag(inv) V^T over the padded kp x kp output (zero on the padding)
__global__ __launch_bounds__(256) void jacobi_compose_kernel(const float *Vt, const float *inv, float *O, int n, int kp) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= kp * kp) return;
    const int r = idx / kp, c = idx % kp;
    float acc = 0.f;
    if (r < n && c < n)
        for (int j = 0; j < n; ++j) acc += inv[j] * Vt[j * n + r] * Vt[j * n + c];
    O[idx] = acc;
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

// ---------------------------------------------------------------------------------------
// Batched per-row Newton solve  step_i = g_i H_i^-1  through a register-resident Cholesky.
// (cmf_solvers.py:321-326 with :346-356, for the rows whose Hessian has lambda_min >= pert, where
// the eigenvalue clamp is the identity; the others are flagged for the Jacobi kernel.)
// One 256-thread workgroup per matrix; thread (ti, tc) = (t & 15, t >> 4) owns the elements
// (i, c) with i = ti (mod 16), c = tc (mod 16) of the lower triangle, i.e. NB(NB+1)/2 blocks of
// one register each (136 VGPRs at n = 256).  A column step publishes column j through LDS,
// every thread scales the 2x16 values it needs and applies the rank-1 update to its own
// registers; no element of the trailing matrix ever travels through LDS.  The triangular solves
// keep the right-hand side in LDS and reduce over the 16 consecutive lanes that own a column.
template <int NB>
struct CholRegs {
    float M[NB][NB]; // only b <= a is used
};

template <int NB>
__device__ __forceinline__ void chol_load(CholRegs<NB> &R, const float *H, int n, int ldh, float shift, int t) {
    // element (i, c) = (16 a + ti, 16 b + tc) is fetched as H[c][i] (H is symmetric): the 16 lanes of a DPP row
    // then read 64 contiguous bytes, nothing goes through LDS and all loads of a thread are in flight together
    const int ti = t & 15, tc = t >> 4;
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) {
            const int i = 16 * a + ti, cc = 16 * b + tc;
            float v = (i == cc) ? 1.0f : 0.0f; // identity outside the valid n x n block
            if (i < n && cc < n) v = H[cc * ldh + i] - (i == cc ? shift : 0.f);
            R.M[a][b] = v;
        }
}

// in-register Cholesky; returns false (uniformly) on a pivot <= floor_.  A column step: the 16 lanes that own
// column j take the pivot by readlane, scale their 16 column entries (zero on and above the diagonal) and publish
// them -- row i = ti + 16 a at col[ti * NB + a], so that afterwards every thread fetches its NB row values and its
// NB column values with wide LDS reads and goes straight to the rank-1 update of its own registers.
// With RHS the forward substitution L y = g rides along (vec holds g on entry, y on exit), between the two
// barriers the column step has anyway.
template <int NB>
__device__ __forceinline__ void chol_col_io(float *p, float (&v)[NB], bool write, int first = 0) {
    // `first`: entries below it are neither produced nor consumed any more (blocks above the current block column): whole
    // quads under it are skipped -- the column steps are bound by LDS traffic, and late block columns need few entries
    if constexpr (NB >= 4) {
#pragma unroll
        for (int k = 0; k < NB / 4; ++k) {
            if (4 * k + 3 < first) continue;
            f32x4 *q = reinterpret_cast<f32x4 *>(p + 4 * k);
            if (write) *q = f32x4{v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
            else { const f32x4 x = *q; v[4 * k] = x[0]; v[4 * k + 1] = x[1]; v[4 * k + 2] = x[2]; v[4 * k + 3] = x[3]; }
        }
    } else {
        f32x2 *q = reinterpret_cast<f32x2 *>(p);
        if (write) *q = f32x2{v[0], v[1]};
        else { const f32x2 x = *q; v[0] = x[0]; v[1] = x[1]; }
    }
}

template <int NB, bool RHS = false>
__device__ __forceinline__ bool chol_factor(CholRegs<NB> &R, int n, float floor_, float *col, int t, float *vec = nullptr) {
    // col = two buffers of (16 NB column entries + pivot slot + pad): column j goes to buffer j & 1, so the owners
    // of column j + 1 may publish while slower waves still read column j -- one barrier per column step
    constexpr int CB = 16 * NB + 4;
    const int ti = t & 15, tc = t >> 4;
    bool ok = true;
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) {
        for (int jl = 0; jl < 16; ++jl) {
            const int j = 16 * jb + jl;
            if (j >= n) break;
            float *cb = col + (jl & 1) * CB;
            if (tc == jl) {
                // the diagonal element sits in lane ti == jl of this 16-lane row
                const float piv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, R.M[jb][jb]), 16 * (jl & 3) + jl));
                const float inv = piv > floor_ ? __builtin_amdgcn_rsqf(piv) : 0.f;
                const float ljj = piv * inv;
                float sc[NB];
#pragma unroll
                for (int a = 0; a < NB; ++a) {
                    const int i = ti + 16 * a;
                    sc[a] = (a >= jb && i > j) ? R.M[a][jb] * inv : 0.f;
                    if (a >= jb) {
                        if (i > j) R.M[a][jb] = sc[a];
                        else if (i == j) R.M[a][jb] = ljj;
                    }
                }
                chol_col_io<NB>(cb + ti * NB, sc, true, jb);
                if (ti == 0) cb[16 * NB] = piv;
                if constexpr (RHS) {
                    const float yj = vec[j] * inv;
#pragma unroll
                    for (int a = jb; a < NB; ++a) {
                        const int i = ti + 16 * a;
                        if (i > j) vec[i] -= sc[a] * yj;
                    }
                    if (ti == jl) vec[j] = yj;
                }
            }
            __syncthreads();
            if (!(cb[16 * NB] > floor_)) { ok = false; break; }
            float li[NB], lc[NB];
            chol_col_io<NB>(cb + ti * NB, li, false, jb);
            chol_col_io<NB>(cb + tc * NB, lc, false, jb);
#pragma unroll
            for (int a = jb; a < NB; ++a)
#pragma unroll
                for (int b = jb; b <= a; ++b) R.M[a][b] -= li[a] * lc[b];
        }
        if (!ok) break;
    }
    __syncthreads(); // vec / col are free again
    return ok;
}

// Forward substitution L y = g, blocked like the back substitution below: per 16-column block (i) every thread's running
// partial sum s[jb] = sum over the solved block columns of L[row][col] y[col] (its own columns only) is reduced over the
// 16 column owners of the row -- two lane exchanges inside the wave, four wave partials through LDS -- (ii) wave 0
// finishes the 16 x 16 triangle with lane broadcasts, (iii) every thread adds the new y values to the partial sums of the
// block rows below.  32 barriers per solve.  (Riding along with the factorisation -- the owners of column j updating 16
// right-hand-side entries in LDS inside the critical path of every column step -- made the second factorisation 60 %
// longer than the first: 20.0 against 12.4 ms over C3's 57 344 matrices.)
template <int NB>
__device__ __forceinline__ void chol_forward(const CholRegs<NB> &R, int n, float *vec, float *dblk, float *part, int t) {
    const int ti = t & 15, tc = t >> 4, lane = t & 63, wv = t >> 6;
    float s[NB];
#pragma unroll
    for (int a = 0; a < NB; ++a) s[a] = 0.f;
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) {
        if (16 * jb >= n) continue;
        float p = s[jb];
        p += __shfl_xor(p, 16, 64);
        p += __shfl_xor(p, 32, 64);
        if (lane < 16) part[wv * 16 + ti] = p;
        dblk[ti * 17 + tc] = R.M[jb][jb];
        __syncthreads();
        if (t < 64) {
            const int c = t & 15;
            float r = vec[16 * jb + c] - (part[c] + part[16 + c] + part[32 + c] + part[48 + c]);
            const float invd = 1.0f / dblk[c * 17 + c];
            float lrow[16]; // L[c][j], j = 0..15
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) lrow[jj] = dblk[c * 17 + jj];
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const float xj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r * invd), jj));
                if (c == jj) r = xj;
                else if (c > jj) r -= lrow[jj] * xj;
            }
            if (t < 16) vec[16 * jb + c] = r;
        }
        __syncthreads();
        const float yv = vec[16 * jb + tc];
#pragma unroll
        for (int a = jb + 1; a < NB; ++a) s[a] += R.M[a][jb] * yv;
    }
}

template <int NB>
__global__ __launch_bounds__(256, 2) void chol_solve_kernel(const float *Hin, const float *grad, float *step, int *need_jacobi,
                                                            int n, int kp, int64_t stride, float pert, int nmat, int diag = 0,
                                                            const int *rowidx = nullptr, int sub = 1, const int *cert = nullptr,
                                                            int cert_rows = 1, int cert_split = 0, float *condest = nullptr) {
    __shared__ __attribute__((aligned(16))) float col[2 * (16 * NB + 4)]; // two buffers of (published column + pivot slot)
    __shared__ float vec[16 * NB];
    __shared__ float stage[16 * 17 + 16]; // diagonal block + block right-hand side of the back substitution
    __shared__ float red[4];
    __shared__ float fpart[64]; // forward substitution: the four waves' partial sums of one block row
    const int mat = blockIdx.x;
    if (mat >= nmat) return;
    // sub > 1: matrix `mat` is diagonal block mat % sub of the (sub k_pad)^2 block-diagonal image mat / sub
    const int ldh = sub * kp;
    const float *H = Hin + (int64_t)(mat / sub) * stride + (int64_t)(mat % sub) * kp * (ldh + 1);
    const int t = threadIdx.x, ti = t & 15, tc = t >> 4;
    // rowidx: the matrices are a compacted subset; gradient / step / flag live at the original row
    const int64_t orow = rowidx ? rowidx[mat] : mat;

    float dmax = 0.f;
    for (int i = t; i < n; i += 256) dmax = fmaxf(dmax, fabsf(H[i * ldh + i]));
    for (int off = 32; off > 0; off >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, off, 64));
    if ((t & 63) == 0) red[t >> 6] = dmax;
    __syncthreads();
    dmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float floor_ = 4.0e-6f * dmax;

    CholRegs<NB> R;
    // certificate (cmf_newton.hip.h, fused_rows_finish): a positive semi-definite part of this matrix, shared with the
    // neighbouring rows of its group, was already shown to exceed the threshold -- so does the matrix; no test of its own
    const bool certified = cert && cert[(mat / cert_rows) * 2 + ((mat % cert_rows) >= cert_split ? 1 : 0)] == 0;
    if (!certified) {
    chol_load<NB>(R, H, n, ldh, pert, t);
    if (diag == 3) { // keep the loaded values alive
        float sink = 0.f;
#pragma unroll
        for (int a = 0; a < NB; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) sink += R.M[a][b];
        if (sink == 12345.678f) need_jacobi[orow] = 2;
        return;
    }
    if (!chol_factor<NB>(R, n, floor_, col, t)) { // lambda_min < pert: the clamp matters -> Jacobi
        if (t == 0) need_jacobi[orow] = 1;
        return;
    }
    }
    if (t == 0) need_jacobi[orow] = 0;
    if (diag == 1) return; // timing diagnostics (cmf_set_option "chol_diag"): 1 = PD test only, 2 = no back substitution,
    __syncthreads();       // 3 = loads only
    for (int i = t; i < 16 * NB; i += 256) vec[i] = (i < n) ? grad[orow * kp + i] : 0.f;
    chol_load<NB>(R, H, n, ldh, 0.f, t);
    __syncthreads(); // publish vec
    (void)chol_factor<NB>(R, n, 0.f, col, t); // H = L L^T
    if (condest) { // max_i H_ii / min_i L_ii^2 <= cond(H): how far float32 can be trusted with this solve (cmf_newton.hip.h, clamp_stats)
        float pmin = 3.0e38f;
        if (ti == tc) {
#pragma unroll
            for (int a = 0; a < NB; ++a)
                if (16 * a + ti < n) pmin = fminf(pmin, R.M[a][a]);
        }
        for (int off = 32; off > 0; off >>= 1) pmin = fminf(pmin, __shfl_xor(pmin, off, 64));
        if ((t & 63) == 0) red[t >> 6] = pmin;
        __syncthreads();
        pmin = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
        if (t == 0) condest[orow] = dmax / fmaxf(pmin * pmin, 1.0e-37f);
    }
    chol_forward<NB>(R, n, vec, stage, fpart, t); // L y = g

    if (diag == 2) return;
    // back substitution  L^T x = y, one 16-column block per pair of barriers: (i) every 16-lane group subtracts
    // the solved blocks from its own column, (ii) wave 0 finishes the 16 x 16 triangle with lane broadcasts
    float *dblk = stage;           // [16][17] diagonal block
    float *rblk = stage + 16 * 17; // [16] right-hand side of the block
#pragma unroll
    for (int jb = NB - 1; jb >= 0; --jb) {
        if (16 * jb >= n) continue;
        float sacc = 0.f;
#pragma unroll
        for (int a = jb + 1; a < NB; ++a) sacc += R.M[a][jb] * vec[ti + 16 * a];
        sacc = group_sum<16>(sacc); // t = 16 tc + ti: the 16 lanes of a DPP row own one column
        dblk[ti * 17 + tc] = R.M[jb][jb];
        if (ti == 0) rblk[tc] = vec[16 * jb + tc] - sacc;
        __syncthreads();
        if (t < 64) {
            const int c = t & 15;
            float r = rblk[c];
            const float invd = 1.0f / dblk[c * 17 + c];
            float drow[16]; // L[j][c], j = 0..15: what x_j contributes to column c
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) drow[jj] = dblk[jj * 17 + c];
#pragma unroll
            for (int jj = 15; jj >= 0; --jj) {
                const float xj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r * invd), jj));
                if (c == jj) r = xj;
                else if (c < jj) r -= drow[jj] * xj;
            }
            if (t < 16) vec[16 * jb + c] = r;
        }
        __syncthreads();
    }
    for (int i = t; i < kp; i += 256) step[orow * kp + i] = (i < n) ? vec[i] : 0.f;
}

// step_i = g_i * Hinv_i only for the rows flagged for the Jacobi path
__global__ __launch_bounds__(256) void rowvec_mat_flagged_kernel(float *step, const float *grad, const float *Hinv, const int *need,
                                                                 int64_t nrows, int kp, int kvalid) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows || !need[row]) return;
    const float *g = grad + row * kp;
    const float *Hm = Hinv + row * (int64_t)kp * kp;
    for (int c = lane; c < kp; c += 64) {
        float acc = 0.f;
        if (c < kvalid)
            for (int a = 0; a < kvalid; ++a) acc += g[a] * Hm[a * kp + c];
        step[row * kp + c] = acc;
    }
}

// ---- low-rank per-row side (cmf_newton.hip.h, sweep_v_lowrank) --------------------------------------------------------------------
// A V row whose per-row Hessian part has FEWER samples than components -- p label columns of Y, p <= 64 < k -- has
//   H_i = S + Z^T C_i Z,   S = shared part (alpha U^T U + l2 I, one k x k matrix),  C_i = diag(c_ij) >= 0  (p x p),
// and when the safe inverse is the plain inverse (lambda_min >= pert: certified by l2 >= pert) the Woodbury identity gives
//   g H_i^-1 = a - (sqrt(C) y)^T B,   a = g S^-1,  B = Z S^-1 (p x k),  K = B Z^T (p x p),
//   (I + sqrt(C) K sqrt(C)) y = sqrt(C) (B g^T)                      -- a p x p positive definite system per row
// instead of forming and factoring a k x k matrix per row (reference: cmf_solvers.py:463-486 builds H_i and eigen-decomposes it).
// K = B Z^T for p <= 64 valid rows (pitch kp in, pitch pk out); workgroup j forms row j, four lanes share a dot product
__global__ __launch_bounds__(256) void lowrank_k_kernel(const float *B, const float *Z, float *K, int p, int pk, int kp) {
    const int j = blockIdx.x, l = threadIdx.x >> 2, q = threadIdx.x & 3;   // l = 0 .. 63
    float acc = 0.f;
    if (j < p && l < p)
        for (int c = q; c < kp; c += 4) acc += B[(int64_t)j * kp + c] * Z[(int64_t)l * kp + c];
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (q == 0 && l < pk) K[j * pk + l] = acc;
}
// per row i (one workgroup of 256 threads per 4 rows): M_i = I + sqrt(c_i) K sqrt(c_i)^T (pk x pk, row-major), rhs_i = sqrt(c_i) o b_i
__global__ __launch_bounds__(256) void lowrank_build_kernel(const float *W, const float *Bv, int64_t ldw, const float *K, float *M, float *rhs,
                                                           int64_t nrows, int p, int pk) {
    __shared__ float sq[4][64];
    const int sub = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + sub;
    const bool live = i < nrows;
    float s = 0.f;
    if (live && lane < p) s = __builtin_amdgcn_sqrtf(fmaxf(W[i * ldw + lane], 0.f));
    sq[sub][lane] = s;
    __syncthreads();
    if (!live) return;
    if (lane < pk) rhs[i * pk + lane] = (lane < p) ? s * Bv[i * ldw + lane] : 0.f;
    float *Mi = M + i * (int64_t)pk * pk;
    for (int idx = lane; idx < pk * pk; idx += 64) {
        const int j = idx / pk, l = idx % pk;
        Mi[idx] = (j == l ? 1.f : 0.f) + sq[sub][j] * K[idx] * sq[sub][l];
    }
}
// Q (nrows x ldw, zero outside the p valid columns) = sqrt(c_i) o y_i
__global__ __launch_bounds__(256) void lowrank_scale_kernel(const float *W, const float *y, float *Q, int64_t ldw, int64_t nrows, int p, int pk) {
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < nrows * ldw; idx += (int64_t)gridDim.x * 256) {
        const int64_t i = idx / ldw;
        const int j = (int)(idx % ldw);
        Q[idx] = (j < p) ? __builtin_amdgcn_sqrtf(fmaxf(W[idx], 0.f)) * y[i * pk + j] : 0.f;
    }
}
// The p x p system of one row solved by ONE WAVE: lane j holds row j of M = I + sqrt(c) K sqrt(c)^T in registers.  Right-looking
// Cholesky with a ROTATING register image: at step t the live columns t .. P - 1 sit in registers 0 .. P - 1 - t, so the pivot
// column is always register 0, every lane scales its entry at once, and the rank-1 update of column t + c takes L[t + c][t] from
// lane t + c with v_readlane and lands one register to the left (M[c - 1] = M[c] - L[j][t] L[t + c][t]: the shift is free).  The
// loop over t is a runtime loop with one compact body (a body unrolled over t needs compile-time register indices; hipcc does
// not finish that form for P >= 40).  The forward solve rides along (y_t is final as soon as column t is); column t of L goes
// to LDS for the backward solve (one wave sum per unknown).  M never exists in memory: the kernel reads c_i, b_i (p floats
// each) and K, and overwrites b_i with q_i = sqrt(c_i) o y_i.  MEASURED SLOWER than the route through memory and is therefore an
// A/B option only (lowrank_rows = 2): C5L, 1e5 rows, P = 64: 3.06 ms per launch against ~1.7 ms for build + chol_solve_kernel<4> +
// scale -- the v_readlane with a run-time lane index costs an SALU add + select and SGPR hazard slots per multiply-add, and
// 64 KB of LDS per workgroup leaves two waves per SIMD to hide them.
template <int P>
__global__ __launch_bounds__(256) void lowrank_solve_kernel(const float *W, float *Bq, int64_t ldw, const float *K, int64_t nrows, int p) {
    static_assert(P == 32 || P == 64, "padded order 32 or 64");
    __shared__ float Ls[4][P][64];                  // [wave][column t][lane j] = L[j][t]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + wv;
    if (i >= nrows) return;                         // whole waves leave: no workgroup barrier below
    const int j = lane < P ? lane : P - 1;          // P = 32: lanes 32 .. 63 mirror lane 31 and write nothing
    const bool live = lane < p;
    const float s = live ? __builtin_amdgcn_sqrtf(fmaxf(W[i * ldw + lane], 0.f)) : 0.f;
    float rhs = live ? s * Bq[i * ldw + lane] : 0.f;
    float M[P];
#pragma unroll
    for (int c = 0; c < P; ++c) {
        const float sc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), c));
        M[c] = (j == c ? 1.f : 0.f) + s * K[j * P + c] * sc;
    }
    float y = 0.f;
    for (int t = 0; t < P; ++t) {                   // runtime loop: register 0 = column t
        const float piv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, M[0]), t));
        const float inv = __builtin_amdgcn_rsqf(piv);
        const float Ljt = j >= t ? M[0] * inv : 0.f;                          // L[j][t]; L[t][t] = sqrt(piv) in lane t
        Ls[wv][t][lane] = Ljt;
        // forward substitution with the finished column: y_t = rhs_t / L[t][t], rhs_j -= L[j][t] y_t
        const float yt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rhs), t)) * inv;
        if (j == t) y = yt;
        if (j > t) rhs -= Ljt * yt;
#pragma unroll
        for (int c = 1; c < P; ++c) {
            const int src = t + c < P ? t + c : P - 1;                          // dead columns: any lane will do
            const float Lct = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, Ljt), src));
            M[c - 1] = M[c] - Ljt * Lct;
        }
    }
    // backward  L^T x = y:  x_t = (y_t - sum_{j > t} L[j][t] x_j) / L[t][t]
    float x = 0.f;
    for (int t = P - 1; t >= 0; --t) {
        const float Ljt = Ls[wv][t][lane];
        float part = (j > t && lane < P) ? Ljt * x : 0.f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
        const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, Ljt), t));
        const float yt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y), t));
        const float xt = (yt - part) * __builtin_amdgcn_rcpf(d);
        if (j == t) x = xt;
    }
    if (lane < P) Bq[i * ldw + lane] = live ? s * x : 0.f;
}

// out (cols x ld_out) = in (rows x ld_in)^T over the padded extents (small factor images)
__global__ __launch_bounds__(256) void transpose_small_kernel(const float *in, float *out, int rows, int cols, int ld_in, int ld_out) {
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < rows * cols; idx += gridDim.x * 256) {
        const int r = idx / cols, c = idx % cols;
        out[(int64_t)c * ld_out + r] = in[(int64_t)r * ld_in + c];
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

// caller-supplied sample lists: any index outside [0, n) sets *bad (the gather kernels index factor rows, LDS bytes and
// mask bytes with them)
__global__ void index_range_kernel(const int32_t *idx, int64_t total, int n, int *bad) {
    int seen = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int v = idx[i];
        if (v < 0 || v >= n) seen = 1;
    }
    if (seen) *bad = 1;
}

// Device-side sampler ("throughput mode" of sg_sample_ratio < 1): for every list l pick exactly
// `s` of the `n` candidates uniformly at random, as the reference's permutation(n)[:s] does
// (cmf_solvers.py:328-344), but from a counter-based hash instead of NumPy's MT19937 stream.
// Candidate j of list l gets the key hash(seed, l, j); the s smallest keys win.  The s-th smallest
// key is found exactly by a 3-level radix select (12 + 12 + 8 bits) on LDS histograms; the keys of a list are
// pairwise distinct (bijective hash of the candidate index), so there are no ties to break.  One workgroup per list; keys are recomputed, never stored.
// Key of candidate j of list l: a 64-bit mix of (seed, l) once per list picks the offset of an odd-multiplier walk over the
// 32-bit integers, a 32-bit avalanche finaliser (two multiplies: "lowbias32") scrambles it.  Both maps are bijections, so the
// keys of one list are pairwise distinct (no ties), and a candidate costs three 32-bit multiplies instead of two 64-bit mixes
// (the selection hashes every candidate four times: 5.0 -> 2.x ms per C3 iteration).
__device__ __forceinline__ uint64_t sample_list_offset(uint64_t seed, uint64_t l) {
    return mix64(seed ^ (l * 0xD1342543DE82EF95ull));
}
__device__ __forceinline__ uint32_t sample_key(uint64_t list_offset, uint32_t j) {
    uint32_t x = j * 0x9E3779B1u + (uint32_t)(list_offset >> 32);
    x ^= x >> 16; x *= 0x21f0aaadu;
    x ^= (uint32_t)list_offset;            // the other half of the list's 64 bits: still a bijection of j
    x ^= x >> 15; x *= 0x735a2d97u;
    x ^= x >> 15;
    return x;
}

// Output: the 0/1 byte mask (masked-dense formulation) and / or the ascending list of the s winners (fused row
// kernel); either pointer may be null.
// `l0` = global index of list 0 (a shard of the rows draws what the unsharded problem draws for those rows).
__global__ __launch_bounds__(256) void sample_select_kernel(uint8_t *mask, int64_t ld, int by_row, int32_t *lists, int64_t nlists,
                                                            int n, int s, uint64_t seed, int64_t l0) {
    __shared__ unsigned hist[4096];
    __shared__ unsigned wsum[4];
    __shared__ unsigned sel_prefix, sel_remaining;
    __shared__ int wave_cnt[8];
    const int64_t l = blockIdx.x;
    if (l >= nlists) return;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    unsigned prefix = 0, remaining = (unsigned)s; // keys < prefix region already counted as winners
    const uint64_t loff = sample_list_offset(seed, (uint64_t)(l + l0));
    // level 0: bits 31..20, level 1: bits 19..8, level 2: bits 7..0
    const int shifts[3] = {20, 8, 0};
    const int widths[3] = {12, 12, 8};
    unsigned known_mask = 0;
    for (int lev = 0; lev < 3; ++lev) {
        const int nb = 1 << widths[lev];
        for (int b = t; b < nb; b += 256) hist[b] = 0;
        if (t == 0) { // fallback if no bin reaches `remaining` (s > n, never for s = int(n * ratio))
            sel_prefix = prefix | ((unsigned)(nb - 1) << shifts[lev]);
            sel_remaining = 0;
        }
        __syncthreads();
        for (int j = t; j < n; j += 256) {
            const uint32_t k = sample_key(loff, (uint32_t)j);
            if ((k & known_mask) == prefix) atomicAdd(&hist[(k >> shifts[lev]) & (nb - 1)], 1u);
        }
        __syncthreads();
        // the bin holding the `remaining`-th smallest key: every thread sums its nb/256 consecutive bins, a block
        // scan places those sums, and the one thread whose range crosses `remaining` walks its own bins
        const int per_t = nb / 256;
        unsigned loc = 0;
        for (int q = 0; q < per_t; ++q) loc += hist[t * per_t + q];
        unsigned incl = loc;
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        unsigned excl = incl - loc;
        for (int w = 0; w < wid; ++w) excl += wsum[w];
        if (excl < remaining && remaining <= excl + loc) {
            unsigned acc = excl, b = (unsigned)(t * per_t);
            for (int q = 0; q < per_t - 1; ++q) {
                if (acc + hist[b] >= remaining) break;
                acc += hist[b];
                ++b;
            }
            sel_prefix = prefix | (b << shifts[lev]);
            sel_remaining = remaining - acc; // how many winners still to pick inside the chosen bin
        }
        __syncthreads();
        prefix = sel_prefix;
        remaining = sel_remaining;
        known_mask |= ((unsigned)(nb - 1)) << shifts[lev];
        __syncthreads();
    }
    // winners: key < prefix, plus the candidate whose key IS the prefix (keys are pairwise distinct inside a list: it is
    // the s-th smallest one; `remaining` is 0 only in the s > n fallback, where everything below the prefix wins)
    int base = 0; // winners emitted so far (uniform)
    for (int j0 = 0; j0 < n; j0 += 256) {
        const int j = j0 + t;
        bool win = false;
        if (j < n) {
            const uint32_t k = sample_key(loff, (uint32_t)j);
            win = k < prefix || (k == prefix && remaining > 0);
        }
        if (mask && j < n) {
            const int64_t off = by_row ? (l * ld + j) : ((int64_t)j * ld + l);
            mask[off] = win ? 1 : 0;
        }
        if (lists) { // ordered compaction of this 256-candidate slice (wave counts double-buffered: one barrier per slice)
            const unsigned long long bal = __ballot(win);
            int *wc = wave_cnt + 4 * ((j0 >> 8) & 1);
            if (lane == 0) wc[wid] = __popcll(bal);
            __syncthreads();
            int off = base;
            for (int w = 0; w < wid; ++w) off += wc[w];
            off += __popcll(bal & ((1ull << lane) - 1ull));
            if (win && off < s) lists[l * (int64_t)s + off] = j;
            base += wc[0] + wc[1] + wc[2] + wc[3];
        }
    }
}

// out = a*A + b*B (B nullable)
__global__ void axpby_kernel(float *out, const float *A, float a, const float *B, float b, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        out[i] = a * A[i] + (B ? b * B[i] : 0.f);
}

} // namespace cmfk
